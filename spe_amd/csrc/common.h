// Shared device helpers for the SPE hot-path kernels (gfx950 / CDNA4 only).
// Wave = 64 lanes everywhere in this tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPE_WAVE 64

// Timing-experiment switches (SPE_DBG_*, SPE_ABL_*, plain-store variants) exist only in -DSPE_ABLATE builds (tools/ab.py); the
// product library (spe_amd/build.py) never defines SPE_ABLATE, so none of them can change what a parity or benchmark run executes.
#ifndef SPE_ABLATE
#undef SPE_DBG_NOEXP
#undef SPE_DBG_NOMIX
#undef SPE_DBG_TAILNOP
#undef SPE_DBG_NOKEEP
#undef SPE_DBG_NOSTAGE
#undef SPE_DBG_NOLOAD
#undef SPE_DBG_NOGW
#undef SPE_DBG_NOMIX4
#undef SPE_DBG_NOMM
#undef SPE_DBG_NOGWM
#undef SPE_DBG_NOST3
#undef SPE_DBG_LN_NOATOMIC
#undef SPE_ABL_NOSTORE
#undef SPE_ABL_NOLOOP
#undef SPE_PLAIN_STORES
#undef FUSED_PLAIN_STORE
#endif

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// fp32 -> bf16, round-to-nearest-even (inputs here are finite activations/weights).
__device__ __forceinline__ unsigned short spe_f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float spe_bf2f(unsigned short h) {
    return __uint_as_float(((uint32_t)h) << 16);
}

// erf(x) in fp32, branch-free (both polynomial branches are evaluated and selected: ~20 instructions against the ~100
// of the device library's erff, whose two paths a wave usually executes one after the other).  Two minimax branches
// split at |x| = 0.9277: a degree-11 odd polynomial below, 1 - exp(p(|x|)) above (N. Juffa's single-precision
// formulation).  Maximum error against erf in fp64: 0.98 ulp with an exact exp (checked on 2M points in [-6, 6]);
// __expf adds <= 2 ulp of a term <= 0.19.  Used by every GELU / GELU' of the library so fused and unfused paths agree.
__device__ __forceinline__ float spe_erff(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __expf(r), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}

// Streaming stores (global_store ... nt) for the GEMM epilogues: outputs that the kernel never reads again - kept out of
// the L2 they do not evict the operand panels nor leave dirty lines behind (same-box A/B: 65.6 -> 65.2 ms per step).  The
// row-wise, conversion and contraction kernels keep ordinary stores: with nt there the step was 0.5 ms SLOWER.
typedef float spe_f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned spe_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void spe_store4_stream(float* p, float a, float b, float c, float d) {
#ifdef SPE_PLAIN_STORES
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
#else
    __builtin_nontemporal_store((spe_f32x4_t){a, b, c, d}, reinterpret_cast<spe_f32x4_t*>(p));
#endif
}
__device__ __forceinline__ void spe_store16_stream(void* p, spe_u32x4_t q) {
#ifdef SPE_PLAIN_STORES
    *reinterpret_cast<spe_u32x4_t*>(p) = q;
#else
    __builtin_nontemporal_store(q, reinterpret_cast<spe_u32x4_t*>(p));
#endif
}

__device__ __forceinline__ float spe_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float spe_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide reductions for <=1024-thread blocks; `red` is >=16 floats of LDS.
__device__ __forceinline__ float spe_block_sum(float v, float* red) {
    v = spe_wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}
__device__ __forceinline__ float spe_block_max(float v, float* red) {
    v = spe_wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, red[i]);
    return r;
}

// Counter-based RNG for dropout: Philox4x32 with SPE_PHILOX_ROUNDS rounds.  The mask for element `idx` of a
// tensor is a pure function of (seed, offset, idx), so backward regenerates it.  7 rounds: the smallest round count of Philox4x32 that
// passes BigCrush (Salmon et al., SC'11, table 2: "Philox4x32-7"; 10 is that paper's default with a safety margin a dropout mask does
// not need) - the attention passes regenerate N^2 H keep flags per block four times per step, 30 % of that cost is these rounds.
// Every mask of the library comes from this one function, so all passes stay consistent whatever the value.
#ifndef SPE_PHILOX_ROUNDS
#define SPE_PHILOX_ROUNDS 7
#endif
__device__ __forceinline__ void spe_philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < SPE_PHILOX_ROUNDS; ++r) {
        // one 32 x 32 -> 64 product per multiplier (v_mad_u64_u32) instead of a high and a low multiply: integer multiplies run at a quarter of the vector rate
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// Uniform in [0,1) for linear element index idx (one Philox call serves 4 consecutive idx).
__device__ __forceinline__ float spe_uniform(uint64_t seed, uint64_t offset, uint64_t idx) {
    uint32_t o[4];
    const uint64_t blk = idx >> 2;
    spe_philox4((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)offset, (uint32_t)(offset >> 32),
                (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return (float)(o[idx & 3] >> 8) * (1.0f / 16777216.0f);
}
// keep-scale of element idx under dropout prob p: 0 (dropped) or 1/(1-p).
__device__ __forceinline__ float spe_drop_scale(uint64_t seed, uint64_t offset, uint64_t idx, float p) {
    return (spe_uniform(seed, offset, idx) >= p) ? 1.0f / (1.0f - p) : 0.0f;
}

// keep-scales of the 4 consecutive elements idx0 .. idx0+3, idx0 a multiple of 4: ONE Philox call
__device__ __forceinline__ void spe_drop_scale4(uint64_t seed, uint64_t offset, uint64_t idx0, float p, float out[4]) {
    uint32_t o[4];
    const uint64_t blk = idx0 >> 2;
    spe_philox4((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
    const float inv = 1.0f / (1.0f - p);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = ((float)(o[i] >> 8) * (1.0f / 16777216.0f) >= p) ? inv : 0.0f;
}

// second 16-bit copy of 4 values next to their bf16 copy h: the low part of a split bf16 operand, bf16(v - bf16(v)), or - lo_f16 - the
// IEEE fp16 copy (saturating, NaN kept) that single-term fp16 forward products read (round 5: the backbone MLP of precision mode bf16s)
__device__ __forceinline__ uint2 spe_second16(const float (&v)[4], uint2 hi_bits, bool lo_f16) {
    typedef __bf16 spe_bf16x4s_t __attribute__((ext_vector_type(4)));
    typedef _Float16 spe_h4s_t __attribute__((ext_vector_type(4)));
    if (lo_f16) {
        spe_h4s_t h;
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = (_Float16)((v[k] != v[k]) ? v[k] : __builtin_amdgcn_fmed3f(v[k], -65504.f, 65504.f));
        return __builtin_bit_cast(uint2, h);
    }
    const spe_bf16x4s_t hb = __builtin_bit_cast(spe_bf16x4s_t, hi_bits);
    spe_bf16x4s_t l;
#pragma unroll
    for (int k = 0; k < 4; ++k) l[k] = (__bf16)(v[k] - (float)hb[k]);
    return __builtin_bit_cast(uint2, l);
}

// Tuning switches of the launchers.  The shipped library has none: every SPE_KNOB is its default, a compile-time constant.  The timing-
// experiment builds of tools/ab.py (-DSPE_ABLATE) read them from the environment, so that an A/B needs no rebuild per setting.
#ifdef SPE_ABLATE
#include <cstdlib>
#define SPE_KNOB(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define SPE_KNOB(name, dflt) (dflt)
#endif

#define SPE_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
