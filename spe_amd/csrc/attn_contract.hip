// Streaming contractions of the bf16 score tensors written by the fused talking-heads kernels
// (attn_fused.hip modes 1 and 3) with a [N, dh] operand:
//   trans = 0 :  out[q, :]   = alpha * sum_key T[q, key] X[key, :]     (O = P'd V ; dQ = scale dS K)
//   trans = 1 :  out[key, :] = alpha * sum_q   T[q, key] X[q, :]       (dV = P'd^T dO ; dK = scale dS^T Q)
// Reference: the `attn @ v` of models/cait.py:388 and the autograd of cait.py:377-388.
//
// T is stored in 16 x 16 blocks, T[b][h][qt][kt][lane][4]: lane l of a block holds query qt*16 + (l&15) and the
// 4 consecutive keys kt*16 + 4*(l>>4) + i - exactly the accumulator layout of the producer AND the A/B operand
// layout of v_mfma_f32_16x16x16_bf16, so a block is written with one fully coalesced 512-B wave store and read
// back straight into an MFMA operand (8 B per lane, no LDS, no shuffles):
//   trans = 0 : the block is the B operand  B[k = key][n = q]                        -> C[m = d][n = q]
//   trans = 1 : the block is the A operand  A[m = q][k = key] of an MFMA against the identity; the product
//               C[m = q][n = key] has lane = (key, 4 queries), i.e. it IS the B operand B[k = q][n = key] of the
//               real contraction (one extra matrix instruction transposes the lane ownership; exact in bf16).
// The other operand comes packed by spe_attn_pack16 as A[m = d][k = row]: X16[b][h][rowtile][dtile][lane][4] =
// x[row = rowtile*16 + 4*(lane>>4) + i][d = dtile*16 + (lane&15)].  The output tile C[m = d][n = row] gives every
// lane 4 consecutive d of one output row: 16-B stores.
//
// This path is HBM-streaming (2 B per score element, read once).  A workgroup owns R = 4 consecutive output
// tiles of one (b, h) (so the X fragments of a step are loaded once for 4 blocks), its 4 waves take the
// contraction steps round-robin and the partial tiles are summed through LDS.
#include <cstdlib>
#include "common.h"
#include "attn_pack.h"

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4p_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4c_t __attribute__((ext_vector_type(4)));

#ifndef CONTRACT_R
#define CONTRACT_R 4
#endif
#ifndef CONTRACT_PD
#define CONTRACT_PD 4
#endif

// LDS-DMA variant of the streaming loop (LDSRING = true): every wave owns a private ring of CL_D step slots in LDS and fills it
// with global_load_lds_dwordx4 (1 KB per instruction: two 512-B pieces - score blocks or X fragments - lanes 0-31 fetch one,
// lanes 32-63 the other; no staging registers, no VGPR write-back), counted s_waitcnt vmcnt(N) keeps CL_D - 1 steps in flight,
// the MFMA operands are read back with conflict-free ds_read_b64.  The guide's LDS-DMA stream reaches ~25 GB/s per CU against
// the ~19 GB/s register loads reach here.  Measured at cfg2 (isolated, same box): 0.131 / 0.129 / 0.133 / 0.128 ms (PV / dV / dQ / dK)
// -> 0.125 / 0.114 / 0.122 / 0.115 with 4 slots (2 slots: 0.119 / 0.120 / 0.119 / 0.117; 6 and 8 slots and `nt` loads are slower);
// in the training step 0.135 -> 0.128 ms per launch (4.1 -> 4.3 TB/s), 60.0 -> 59.3 ms per step.
#ifndef CL_D
#define CL_D 4
#endif
#ifdef CL_USE_NT
#define CL_NT " nt"
#else
#define CL_NT ""
#endif
#define CL_SLOT (CONTRACT_R * 512 + 2048)
__device__ __forceinline__ void cl_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" CL_NT "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Element formats (FMT): 0 = bf16 score blocks x bf16 fragments (dQ, dK: gradients);  1 = fp16 blocks x fp16 fragments (O = P'd V
// of the forward pass: probabilities and values are O(1), 3 more mantissa bits at the same bytes and MFMA rate);  2 = fp16 blocks
// x bf16 fragments (dV = P'd^T dO: the lane-transposing product against the identity runs in fp16 - exact - and its result is
// rounded to bf16 like the bf16 blocks' was, to meet the bf16 gradient dO).
typedef _Float16 f16x4c_t __attribute__((ext_vector_type(4)));
template <bool F16>
__device__ __forceinline__ f32x4_t cl_mfma(s16x4_t a, s16x4_t b, f32x4_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4c_t, a), __builtin_bit_cast(f16x4c_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ s16x4_t cl_round(f32x4_t t) {
    if constexpr (F16) {
        f16x4c_t tv;
        tv[0] = (_Float16)t[0]; tv[1] = (_Float16)t[1]; tv[2] = (_Float16)t[2]; tv[3] = (_Float16)t[3];
        return __builtin_bit_cast(s16x4_t, tv);
    } else {
        bf16x4c_t tv;
        tv[0] = (__bf16)t[0]; tv[1] = (__bf16)t[1]; tv[2] = (__bf16)t[2]; tv[3] = (__bf16)t[3];
        return __builtin_bit_cast(s16x4_t, tv);
    }
}

template <int DT, bool TRANS, bool LDSRING = false, int FMT = 0>
__global__ __launch_bounds__(256) void attn_contract_kernel(const uint2* __restrict__ T, const uint2* __restrict__ X,
                                                            float* __restrict__ out, long ob, long on, long oh,
                                                            int H, int N, int nt, int dh, int ngrp, float alpha,
                                                            int bhn, int nfull, float* ws, unsigned* counters,
                                                            unsigned short* __restrict__ out16, unsigned short* __restrict__ out16lo) {
    constexpr int R = CONTRACT_R;
    constexpr bool TF16 = FMT >= 1, XF16 = FMT == 1;            // element format of the score blocks / of the fragments
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4_t* red = reinterpret_cast<f32x4_t*>(smem_raw);          // [2 waves][R][DT][64]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // scalar: step indices and block addresses stay in SGPRs
    // Workgroups [0, bhn*nfull): one group of R output tiles over the whole contraction range.  The ngrp - nfull
    // left-over groups per (b, h) come LAST in the grid, each as 4 workgroups over a quarter of the range: they start
    // when the first full workgroups retire and run alone, so their length is the tail of the launch (see the launcher).
    int grp, bh, quarter = -1;
    if ((int)blockIdx.x < bhn * nfull) { grp = blockIdx.x % nfull; bh = blockIdx.x / nfull; }
    else {
        int li = blockIdx.x - bhn * nfull;
        quarter = li & 3; li >>= 2;
        const int nlo = ngrp - nfull;
        grp = nfull + li % nlo; bh = li / nlo;
    }
    const int clen = (nt + 3) / 4;
    const int cbeg = quarter < 0 ? 0 : quarter * clen, cend = quarter < 0 ? nt : min(nt, cbeg + clen);
    const int t0 = grp * R;
    const uint2* Tb = T + (long)bh * nt * nt * 64;
    const uint2* Xb = X + (long)bh * nt * DT * 64;

    f32x4_t acc[R][DT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int d = 0; d < DT; ++d) acc[r][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    s16x4_t ident;
#pragma unroll
    for (int i = 0; i < 4; ++i) ident[i] = ((lane & 15) == 4 * (lane >> 4) + i) ? (short)(TF16 ? 0x3C00 : 0x3F80) : (short)0;

    // tile indices of the R output tiles, clamped (a clamped duplicate is computed and never stored)
    int tr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) tr[r] = min(t0 + r, nt - 1);

    if constexpr (LDSRING) {
        unsigned char* ring = smem_raw + wave * (CL_D * CL_SLOT);
        const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ring);
        const int first = cbeg + wave;
        const int ns = (cend > first) ? (cend - first + 3) / 4 : 0;
        const int half = lane >> 5, l16 = (lane & 31) * 16;
        auto issue = [&](int st) {
            const int cc = min(first + 4 * st, nt - 1);                       // clamped: steps past the range fetch valid memory, unused
            const unsigned dst = ring_lds + (unsigned)((st % CL_D) * CL_SLOT);
#pragma unroll
            for (int pp = 0; pp < R / 2; ++pp) {                                // score blocks of output tiles 2pp, 2pp+1
                const int r = 2 * pp + half;
                const uint2* src = TRANS ? Tb + ((long)cc * nt + tr[r]) * 64 : Tb + ((long)tr[r] * nt + cc) * 64;
                cl_glds16(reinterpret_cast<const unsigned char*>(src) + l16, dst + pp * 1024);
            }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {                                    // X fragments of d tiles 2pp, 2pp+1 (clamped to DT-1)
                const int d = min(2 * pp + half, DT - 1);
                const uint2* src = Xb + ((long)cc * DT + d) * 64;
                cl_glds16(reinterpret_cast<const unsigned char*>(src) + l16, dst + R * 512 + pp * 1024);
            }
        };
        static_assert(CONTRACT_R % 2 == 0 && DT <= 4, "ring slot layout: R score blocks + up to 4 X fragments");
#pragma unroll
        for (int st = 0; st < CL_D; ++st) issue(st);
        for (int st = 0; st < ns; ++st) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((CL_D - 1) * (R / 2 + 2)) : "memory");     // the pieces of step st have landed
            const unsigned char* sl = ring + (st % CL_D) * CL_SLOT;
            uint2 tbs[R], xfs[DT];
#pragma unroll
            for (int r = 0; r < R; ++r) tbs[r] = *reinterpret_cast<const uint2*>(sl + r * 512 + lane * 8);
#pragma unroll
            for (int d = 0; d < DT; ++d) xfs[d] = *reinterpret_cast<const uint2*>(sl + R * 512 + d * 512 + lane * 8);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          // slot read: free for the refill
            issue(st + CL_D);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                s16x4_t bt;
                if (TRANS) {
                    const f32x4_t t = cl_mfma<TF16>(__builtin_bit_cast(s16x4_t, tbs[r]), ident, (f32x4_t){0.f, 0.f, 0.f, 0.f});
                    bt = cl_round<XF16>(t);
                } else {
                    bt = __builtin_bit_cast(s16x4_t, tbs[r]);
                }
#pragma unroll
                for (int d = 0; d < DT; ++d)
                    acc[r][d] = cl_mfma<XF16>(__builtin_bit_cast(s16x4_t, xfs[d]), bt, acc[r][d]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                // drain the ring before LDS is reused below
        __syncthreads();
    } else {
    // PD contraction steps are requested together (loads are unconditional with clamped indices - a branch around a
    // load costs a full vmcnt(0) drain)
    constexpr int PD = CONTRACT_PD;
    uint2 tb[PD][R], xf[PD][DT];
#define LOAD_STEP(c_, tb_, xf_)                                                                         \
    {                                                                                                   \
        const int cc = min((c_), nt - 1);                                                               \
        _Pragma("unroll") for (int d = 0; d < DT; ++d) xf_[d] = Xb[((long)cc * DT + d) * 64 + lane];    \
        _Pragma("unroll") for (int r = 0; r < R; ++r)                                                   \
            tb_[r] = TRANS ? Tb[((long)cc * nt + tr[r]) * 64 + lane] : Tb[((long)tr[r] * nt + cc) * 64 + lane]; \
    }
    // hipcc drains vmcnt(0) at the head of a loop whose loads are carried across the back edge, so a register ring
    // gives no overlap; instead every iteration requests PD steps at once and then consumes them in order (the
    // waits inside the straight-line body are exact: vmcnt(7*(PD-1)), ..., vmcnt(0)).
    for (int c0 = cbeg + wave; c0 < cend; c0 += 4 * PD) {
#pragma unroll
        for (int s = 0; s < PD; ++s) LOAD_STEP(c0 + 4 * s, tb[s], xf[s]);
#pragma unroll
        for (int s = 0; s < PD; ++s) {
            const int c = c0 + 4 * s;
            if (c < cend) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    s16x4_t bt;
                    if (TRANS) {
                        const f32x4_t t = cl_mfma<TF16>(__builtin_bit_cast(s16x4_t, tb[s][r]), ident, (f32x4_t){0.f, 0.f, 0.f, 0.f});
                        bt = cl_round<XF16>(t);
                    } else {
                        bt = __builtin_bit_cast(s16x4_t, tb[s][r]);
                    }
#pragma unroll
                    for (int d = 0; d < DT; ++d)
                        acc[r][d] = cl_mfma<XF16>(__builtin_bit_cast(s16x4_t, xf[s][d]), bt, acc[r][d]);
                }
            }
        }
    }
#undef LOAD_STEP
    }

    // ---- sum the 4 waves' partial tiles: (2,3) -> LDS -> (0,1) ; then 1 -> LDS -> 0
    if (wave >= 2) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int d = 0; d < DT; ++d) red[(((wave - 2) * R + r) * DT + d) * 64 + lane] = acc[r][d];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int d = 0; d < DT; ++d) acc[r][d] += red[((wave * R + r) * DT + d) * 64 + lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int d = 0; d < DT; ++d) red[(r * DT + d) * 64 + lane] = acc[r][d];
    }
    __syncthreads();
    if (wave == 0 && quarter >= 0) {
        // ---- quarter of a left-over group: publish the partial tiles; the workgroup that arrives last adds the four
        // partials in fixed order and writes the rows (device-scope fences: the quarters run on different XCDs)
        const int slot = bh * (ngrp - nfull) + (grp - nfull);
        float* wsp = ws + ((long)slot * 4 + quarter) * (R * DT * 256);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                acc[r][d] += red[(r * DT + d) * 64 + lane];
                *reinterpret_cast<f32x4_t*>(wsp + ((r * DT + d) * 64 + lane) * 4) = acc[r][d];
            }
        __threadfence();
        unsigned old = 0;
        if (lane == 0) old = atomicAdd(counters + slot, 1u);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old != 3u) return;
        __threadfence();
        const float* w0 = ws + (long)slot * 4 * (R * DT * 256);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int o = ((r * DT + d) * 64 + lane) * 4;
                // plain loads: the fence above orders them after the counter (volatile ones would be issued one by one)
                f32x4_t v = *reinterpret_cast<const f32x4_t*>(w0 + o);
                v += *reinterpret_cast<const f32x4_t*>(w0 + 1 * (R * DT * 256) + o);
                v += *reinterpret_cast<const f32x4_t*>(w0 + 2 * (R * DT * 256) + o);
                v += *reinterpret_cast<const f32x4_t*>(w0 + 3 * (R * DT * 256) + o);
                acc[r][d] = v;
                red[(r * DT + d) * 64 + lane] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            }
        if (lane == 0) counters[slot] = 0u;            // ready for the next launch on this stream
    }
    if (wave == 0) {
        const int b = bh / H, h = bh % H;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = (t0 + r) * 16 + (lane & 15);
            if (t0 + r < nt && row < N) {
                float* dst = out + b * ob + (long)row * on + h * oh;
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    f32x4_t v = acc[r][d] + red[(r * DT + d) * 64 + lane];
                    v *= alpha;
                    const int dc = d * 16 + 4 * (lane >> 4);
                    if (!out) {             // 16-bit result only (the gradient goes straight into the next GEMM's operand)
                    } else if (dc + 3 < dh && ((((uintptr_t)(dst + dc)) & 15) == 0)) {
                        *reinterpret_cast<f32x4_t*>(dst + dc) = v;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (dc + i < dh) dst[dc + i] = v[i];
                    }
                    if (out16) {        // bf16 copy with the same element strides: the operand of the Linear that consumes the result
                        unsigned short* d16 = out16 + b * ob + (long)row * on + h * oh + dc;
                        bf16x4c_t hv;
                        hv[0] = (__bf16)v[0]; hv[1] = (__bf16)v[1]; hv[2] = (__bf16)v[2]; hv[3] = (__bf16)v[3];
                        if (dc + 3 < dh && ((((uintptr_t)d16) & 7) == 0)) *reinterpret_cast<uint2*>(d16) = __builtin_bit_cast(uint2, hv);
                        else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) if (dc + i < dh) d16[i] = __builtin_bit_cast(s16x4_t, hv)[i];
                        }
                        if (out16lo) {  // low part of the split operand (precision mode bf16s), same addressing
                            unsigned short* l16 = out16lo + b * ob + (long)row * on + h * oh + dc;
                            bf16x4c_t lv;
                            lv[0] = (__bf16)(v[0] - (float)hv[0]); lv[1] = (__bf16)(v[1] - (float)hv[1]);
                            lv[2] = (__bf16)(v[2] - (float)hv[2]); lv[3] = (__bf16)(v[3] - (float)hv[3]);
                            if (dc + 3 < dh && ((((uintptr_t)l16) & 7) == 0)) *reinterpret_cast<uint2*>(l16) = __builtin_bit_cast(uint2, lv);
                            else {
#pragma unroll
                                for (int i = 0; i < 4; ++i) if (dc + i < dh) l16[i] = __builtin_bit_cast(s16x4_t, lv)[i];
                            }
                        }
                    }
                }
            }
        }
    }
}

// X16[b][h][rowtile][dtile][lane][4] = bf16(x[b, rowtile*16 + 4*(lane>>4) + i, h, dtile*16 + (lane&15)])  (0 outside)
__global__ __launch_bounds__(256) void attn_pack16_kernel(const float* __restrict__ x, long sb, long sn, long sh, int B, int N, int H,
                                                          int dh, int nt, int DT, uint2* __restrict__ out) {
    const long total = (long)B * H * nt * DT * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ln = (int)(i & 63); long t = i >> 6;
        const int dt = (int)(t % DT); t /= DT;
        const int tile = (int)(t % nt); t /= nt;
        const int h = (int)(t % H); const int b = (int)(t / H);
        const int d = dt * 16 + (ln & 15), r0 = tile * 16 + 4 * (ln >> 4);
        bf16x4c_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r0 + j;
            const float f = x[b * sb + (long)min(row, N - 1) * sn + h * sh + min(d, dh - 1)];
            o[j] = (__bf16)((row < N && d < dh) ? f : 0.f);
        }
        out[i] = __builtin_bit_cast(uint2, o);
    }
}

extern "C" int spe_attn_pack16(const float* x, long sb, long sn, long sh, int B, int N, int H, int dh, void* out, hipStream_t st) {
    const int nt = (N + 15) / 16, DT = (dh + 15) / 16;
    const long total = (long)B * H * nt * DT * 64;
    if (total <= 0) return 0;
    long nb = (total + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(attn_pack16_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, sb, sn, sh, B, N, H, dh, nt, DT,
                       reinterpret_cast<uint2*>(out));
    SPE_CHECK_LAUNCH();
    return 0;
}

// Several packs of one attention call in ONE launch (forward: q, k -> 32-wide fragments, v -> 16-wide; backward: v, dO
// -> 32-wide, dO, k, q -> 16-wide): the packs are ~10 us each and launch-latency bound.  blockIdx.y = job.
#define PACK_MAXJOBS 6
// kind & 15 = layout: 0 spe_attn_pack, 1 spe_attn_pack16, 2 spe_attn_pack without the tail step; kind & 16: fp16 elements (else bf16)
struct PackJob { const float* x; long sb, sn, sh; float scale; int kind; void* out; int N, dh; };
struct PackJobs { PackJob j[PACK_MAXJOBS]; int B, H; };
template <typename IT>
__device__ __forceinline__ void attn_pack_job(const PackJob& jb, int B, int H) {
    const int N = jb.N, dh = jb.dh, nt = (N + 15) / 16;
    const IT stride = (IT)gridDim.x * 256;
    const int lay = jb.kind & 15, f16 = (jb.kind >> 4) & 1;
    if (lay == 0 || lay == 2) {
        const int notail = lay == 2;
        const IT total = (IT)attn_pack_units(B, N, H, dh, notail);
        for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < total; i += stride)
            attn_pack_unit_t<IT>(jb.x, jb.sb, jb.sn, jb.sh, N, H, dh, nt, jb.scale, i, reinterpret_cast<uint2*>(jb.out), notail, f16);
    } else {
        const int DT = (dh + 15) / 16;
        const IT total = (IT)((long)B * H * nt * DT * 64);
        uint2* out = reinterpret_cast<uint2*>(jb.out);
        for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
            const int ln = (int)(i & 63); IT t = i >> 6;
            const IT t1 = t / (IT)DT; const int dt = (int)(t - t1 * (IT)DT);
            const IT t2 = t1 / (IT)nt; const int tile = (int)(t1 - t2 * (IT)nt);
            const int b = (int)(t2 / (IT)H), h = (int)(t2 - (IT)b * (IT)H);
            const int d = dt * 16 + (ln & 15), r0 = tile * 16 + 4 * (ln >> 4);
            const float* src = jb.x + b * jb.sb + h * jb.sh + min(d, dh - 1);
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = r0 + j;
                const float f = src[(long)min(row, N - 1) * jb.sn];
                o[j] = (row < N && d < dh) ? f * jb.scale : 0.f;
            }
            out[i] = spe_cvt4_16(o[0], o[1], o[2], o[3], f16);
        }
    }
}
// One WAVE per (b, h, 16-row tile) record: the per-unit version above spends its time on index arithmetic (three integer
// divisions per 8-B unit) and, for the 16-wide layout, on 4-B loads from four different rows per lane.  Here the record's
// coordinates are wave-uniform, every lane loads 16 B of ONE row (row = lane & 15, 8 consecutive head dims per 32-wide step:
// 128 contiguous bytes per row and step) and
//   kinds 0 / 2: converts and stores its own 16-B / 8-B fragment pieces directly (the fragment layout IS that load pattern);
//   kind 1     : passes the 16 x dh tile through a per-wave LDS tile and gathers the 4 rows x 1 column of its unit.
// Requires 16-B aligned rows (base, strides and dh multiples of 4 floats) and dh <= 64; otherwise the per-unit kernel runs.
#define PACKR_LD 68                                      // floats per LDS row: 64 + 4 (16-B aligned float4 stores)
__global__ __launch_bounds__(256) void attn_pack_rec_kernel(PackJobs a) {
    __shared__ __attribute__((aligned(16))) float tile_s[4][16 * PACKR_LD];
    const PackJob jb = a.j[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = jb.N, dh = jb.dh, nt = (N + 15) / 16, H = a.H;
    const long nrec = (long)a.B * H * nt;
    const int r = lane & 15, g = lane >> 4;
    const int lay = jb.kind & 15, f16 = (jb.kind >> 4) & 1;
    for (long rec = (long)blockIdx.x * 4 + wave; rec < nrec; rec += (long)gridDim.x * 4) {
        const long bh = rec / nt; const int tile = (int)(rec - bh * nt);
        const int b = (int)(bh / H), h = (int)(bh - (long)b * H);
        const int row = tile * 16 + r;
        const bool rv = row < N;
        const float* src = jb.x + b * jb.sb + (long)min(row, N - 1) * jb.sn + h * jb.sh;
        if (lay == 0 || lay == 2) {
            const int notail = lay == 2;
            const int rem = dh % 32, full = notail ? (dh + 31) / 32 : dh / 32 + (rem > 16 ? 1 : 0), tail = (!notail && rem > 0 && rem <= 16) ? 1 : 0;
            uint2* orec = reinterpret_cast<uint2*>(jb.out) + rec * (full * 128 + tail * 64);
            for (int st = 0; st < full; ++st) {
                const int d0 = st * 32 + g * 8;
                float f[8];
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4) {
                    const int d = d0 + 4 * q4;
                    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (d < dh) t = *reinterpret_cast<const float4*>(src + d);       // dh % 4 == 0: a float4 is inside or outside
                    f[4 * q4] = t.x; f[4 * q4 + 1] = t.y; f[4 * q4 + 2] = t.z; f[4 * q4 + 3] = t.w;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = rv ? f[e] * jb.scale : 0.f;
                const uint2 lo = spe_cvt4_16(f[0], f[1], f[2], f[3], f16), hi = spe_cvt4_16(f[4], f[5], f[6], f[7], f16);
                *reinterpret_cast<u32x4p_t*>(orec + st * 128 + lane * 2) = (u32x4p_t){lo.x, lo.y, hi.x, hi.y};
            }
            if (tail) {
                const int d = full * 32 + g * 4;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (d < dh) t = *reinterpret_cast<const float4*>(src + d);
                orec[full * 128 + lane] = spe_cvt4_16(rv ? t.x * jb.scale : 0.f, rv ? t.y * jb.scale : 0.f, rv ? t.z * jb.scale : 0.f,
                                                      rv ? t.w * jb.scale : 0.f, f16);
            }
        } else {
            const int DT = (dh + 15) / 16;
            float* tl = tile_s[wave];
            // 16 rows x up to 64 dims: lane (r, g) brings dims g*16 .. g*16+15 of row r
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int d = g * 16 + 4 * q4;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (d < dh && rv) t = *reinterpret_cast<const float4*>(src + d);
                *reinterpret_cast<float4*>(tl + r * PACKR_LD + d) = t;
            }
            // wave-private tile: the LDS writes of this wave are visible to it after the wait the compiler places before the reads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint2* orec = reinterpret_cast<uint2*>(jb.out) + rec * (DT * 64);
            for (int dt = 0; dt < DT; ++dt) {
                const int d = dt * 16 + r;                      // this lane's column; its rows: 4 g .. 4 g + 3
                orec[dt * 64 + lane] = spe_cvt4_16(tl[(4 * g) * PACKR_LD + d] * jb.scale, tl[(4 * g + 1) * PACKR_LD + d] * jb.scale,
                                                   tl[(4 * g + 2) * PACKR_LD + d] * jb.scale, tl[(4 * g + 3) * PACKR_LD + d] * jb.scale, f16);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
    }
}

__global__ __launch_bounds__(256) void attn_pack_multi_kernel(PackJobs a, int narrow) {
    const PackJob jb = a.j[blockIdx.y];
    if (narrow) attn_pack_job<unsigned>(jb, a.B, a.H); else attn_pack_job<long>(jb, a.B, a.H);
}

// C-ABI: see include/spe_hip.h (spe_attn_pack_multi).  xs/outs/scales/kinds/Ns/dhs: njobs entries (host arrays).
extern "C" int spe_attn_pack_multi(int njobs, const float* const* xs, const long* strides, const float* scales, const int* kinds,
                                   void* const* outs, const int* Ns, const int* dhs, int B, int H, hipStream_t st) {
    if (njobs <= 0 || B <= 0) return 0;
    if (njobs > PACK_MAXJOBS) return -2;
    PackJobs a;
    long total = 0;
    for (int i = 0; i < njobs; ++i) {
        a.j[i].x = xs[i]; a.j[i].sb = strides[3 * i]; a.j[i].sn = strides[3 * i + 1]; a.j[i].sh = strides[3 * i + 2];
        a.j[i].scale = scales[i]; a.j[i].kind = kinds[i]; a.j[i].out = outs[i]; a.j[i].N = Ns[i]; a.j[i].dh = dhs[i];
        if (Ns[i] <= 0 || dhs[i] <= 0) return -2;
        const long t = (long)B * H * ((Ns[i] + 15) / 16) * ((dhs[i] + 31) / 32) * 128;    // the largest item count of the layouts
        if (t > total) total = t;
    }
    a.B = B; a.H = H;
    {   // wave-per-record kernel when every job has 16-B aligned rows and dh <= 64
        static const int recpath = SPE_KNOB("SPE_PACK_REC", 1);      // 0: per-unit kernel (A/B)
        bool ok = recpath != 0;
        long nrec = 0;
        for (int i = 0; i < njobs && ok; ++i) {
            ok = (reinterpret_cast<uintptr_t>(xs[i]) & 15) == 0 && (a.j[i].sb & 3) == 0 && (a.j[i].sn & 3) == 0 && (a.j[i].sh & 3) == 0 &&
                 (dhs[i] & 3) == 0 && dhs[i] <= 64;
            const long n = (long)B * H * ((Ns[i] + 15) / 16);
            if (n > nrec) nrec = n;
        }
        if (ok) {
            long nbr = (nrec + 3) / 4; if (nbr > 4096) nbr = 4096;
            hipLaunchKernelGGL(attn_pack_rec_kernel, dim3((unsigned)nbr, njobs), dim3(256), 0, st, a);
            SPE_CHECK_LAUNCH();
            return 0;
        }
    }
    long nb = (total + 255) / 256; if (nb > 2048) nb = 2048;
    const int narrow = total + 2048L * 256 < (1L << 31);          // unit indices (and one grid stride beyond) fit 32 bits
    hipLaunchKernelGGL(attn_pack_multi_kernel, dim3((unsigned)nb, njobs), dim3(256), 0, st, a, narrow);
    SPE_CHECK_LAUNCH();
    return 0;
}

// The launch lasts as long as its most loaded CU (a CU streams ~19 GB/s of blocks whatever its resident waves), and
// workgroups beyond the resident slots run alone after the rest, latency-bound: 1040 workgroups on 1024 slots cost
// 0.136 ms against 0.117 ms for the first 1024 (cfg2).  With a workspace the groups beyond the last full round are
// therefore split into quarters of the contraction range (4x shorter) and combined by the last arriver.
template <int DT, bool TRANS, int FMT>
static int launch_contract(const void* T, const void* X, float* out, long ob, long on, long oh, int B, int H, int N, int nt, int dh,
                           float alpha, float* ws, unsigned* counters, long ws_floats, void* out16, void* out16lo, hipStream_t st) {
    const int ngrp = (nt + CONTRACT_R - 1) / CONTRACT_R;
    const long bhn = (long)B * H, full = bhn * ngrp;
    int nlo = 0;
    if (ws && counters && full > 1024) {
        const long over = full - (full / 1024) * 1024;                     // workgroups past the last full round
        const long l = (over + bhn - 1) / bhn;
        if (over > 0 && l <= 2 && l < ngrp && bhn * l * 4 <= 256 && bhn * l * 4 * (CONTRACT_R * DT * 256) <= ws_floats) nlo = (int)l;
    }
    const int nfull = ngrp - nlo;
    static const int use_ring = SPE_KNOB("SPE_CONTRACT_LDS", 1);   // 0: register-load loop (A/B)
    if (use_ring) {
        const int smem_r = (4 * CL_D * CL_SLOT > 2 * CONTRACT_R * DT * 64 * 16) ? 4 * CL_D * CL_SLOT : 2 * CONTRACT_R * DT * 64 * 16;
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_contract_kernel<DT, TRANS, true, FMT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, smem_r);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        hipLaunchKernelGGL((attn_contract_kernel<DT, TRANS, true, FMT>), dim3((unsigned)(bhn * nfull + bhn * nlo * 4)), dim3(256), smem_r, st,
                           reinterpret_cast<const uint2*>(T), reinterpret_cast<const uint2*>(X), out, ob, on, oh, H, N, nt, dh, ngrp, alpha,
                           (int)bhn, nfull, ws, counters, reinterpret_cast<unsigned short*>(out16), reinterpret_cast<unsigned short*>(out16lo));
        SPE_CHECK_LAUNCH();
        return 0;
    }
    const int smem = 2 * CONTRACT_R * DT * 64 * 16;
    hipLaunchKernelGGL((attn_contract_kernel<DT, TRANS, false, FMT>), dim3((unsigned)(bhn * nfull + bhn * nlo * 4)), dim3(256), smem, st,
                       reinterpret_cast<const uint2*>(T), reinterpret_cast<const uint2*>(X), out, ob, on, oh, H, N, nt, dh, ngrp, alpha,
                       (int)bhn, nfull, ws, counters, reinterpret_cast<unsigned short*>(out16), reinterpret_cast<unsigned short*>(out16lo));
    SPE_CHECK_LAUNCH();
    return 0;
}

// C-ABI: see include/spe_hip.h (spe_attn_contract).  Returns -2 for head dims above 64 or an unsupported (trans, fmt) pair.
extern "C" int spe_attn_contract(const void* T, const void* X16, float* out, long ob, long on, long oh, int B, int H, int N, int dh,
                                 int trans, int fmt, float alpha, float* ws, unsigned int* counters, long ws_floats, void* out16,
                                 void* out16lo, hipStream_t st) {
    const int nt = (N + 15) / 16, DT = (dh + 15) / 16;
    if (B <= 0 || H <= 0 || N <= 0) return 0;
    if (fmt < 0 || fmt > 2 || (fmt == 1 && trans) || (fmt == 2 && !trans) || (out16lo && !out16)) return -2;
#define SPE_CONTRACT_ARGS T, X16, out, ob, on, oh, B, H, N, nt, dh, alpha, ws, counters, ws_floats, out16, out16lo, st
#define SPE_CONTRACT_CASE(D) case D:                                                          \
        if (fmt == 1) return launch_contract<D, false, 1>(SPE_CONTRACT_ARGS);                 \
        if (fmt == 2) return launch_contract<D, true, 2>(SPE_CONTRACT_ARGS);                  \
        return trans ? launch_contract<D, true, 0>(SPE_CONTRACT_ARGS) : launch_contract<D, false, 0>(SPE_CONTRACT_ARGS);
    switch (DT) {
        SPE_CONTRACT_CASE(1)
        SPE_CONTRACT_CASE(2)
        SPE_CONTRACT_CASE(3)
        SPE_CONTRACT_CASE(4)
    }
#undef SPE_CONTRACT_CASE
#undef SPE_CONTRACT_ARGS
    return -2;
}
