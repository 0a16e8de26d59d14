// Flash-style multi-head attention for the encoder / decoder / cross-attention of the SPE transformer
// (reference models/attention.py:277-383 `multi_head_attention_forward`: softmax(scale q k^T + key_padding_mask),
// dropout on the probabilities, . v, with q/k head dim != v head dim in the decoder cross-attention; and the
// nn.MultiheadAttention core of the encoder, models/transformer.py:275-277) - forward and backward without ever
// writing the [B,H,Lq,Lk] score tensor (the materialising path moves 4 such fp32 tensors per layer: 53 MB each for
// the decoder's 200 x 4150 cross-attention, 1.1 GB each for an encoder layer at N = 4150).
//
// Element formats as in attn_fused.hip: the FORWARD operands - q * scale * log2(e), k (Qf, Kf), v (V16) and the probabilities of
// the P.V product - are O(1) and go through fp16 (3 more mantissa bits than bf16, same size and MFMA rate: the bf16 version of
// this kernel alone cost 1.1e-3 of pred_logits at cfg2, tools/error_budget.py); the backward recomputes S from the same fp16
// fragments and keeps bf16 wherever a gradient is an operand (Vf.dOf, P^T dO16, dS K16, dS^T Q16).
//
// Same building blocks as the talking-heads kernels: 16-bit MFMA operand fragments packed by spe_attn_pack_multi, the
// swapped orientation S^T = K_tile . Q_tile^T so that a lane owns one query and 4 keys, and the 16x16 C tile reused
// directly as the B operand of v_mfma_f32_16x16x16_bf16 for the second contraction (P.V, dS.K, P^T.dO, dS^T.Q).
// A wave owns one 16-row tile of the non-streamed axis; the 4 waves of a workgroup share the streamed tile's fragments
// through a double-buffered LDS stage:
//   forward   item = (b, h, 16-query tile, key chunk): online softmax in the log2 domain, O += P_drop V; partial
//             (O, m, l) per chunk, combined by mha_merge_kernel (few query tiles -> many key chunks keep the chip busy)
//   backward  dq  item = (b, h, query tile, key chunk): dS = P (keep dPd - D), dQ += dS K  (atomics across chunks)
//             dkv item = (b, h, 16-key tile): loops over all query tiles in the non-swapped orientation, dV += Pd^T dO,
//             dK += dS^T Q, plain stores
// Head dims are run-time (<= 96 for q/k, <= 64 for v); fragments are full 32-wide steps (kind 2 of spe_attn_pack_multi).
#include "common.h"

typedef unsigned int u32x4m_t __attribute__((ext_vector_type(4)));
typedef short s16x4m_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4m_t __attribute__((ext_vector_type(4)));
#define MHA_DSK 3          // max 32-steps of the q/k head dim
#define MHA_DSV 2          // max 32-steps of the v head dim
#define MHA_DVT 4          // max 16-tiles of the v head dim
#define MHA_DKT 6          // max 16-tiles of the q/k head dim
#define MHA_LOG2E 1.4426950408889634f
#define MHA_LN2 0.6931471805599453f

struct MhaArgs {
    const u32x4m_t* Qf; const u32x4m_t* Kf; const u32x4m_t* Vf; const u32x4m_t* dOf;   // 32-wide fragment records [B,H,nt][ds][64]
    const uint2* V16; const uint2* K16; const uint2* Q16; const uint2* dO16;            // 16-wide fragments [B,H,nt][dt][64]
    const unsigned char* mask;                 // [B, Lk] key padding (1 = padded) or null
    float* Opart; float* ML;                   // forward partials: [item][dvt][64][4], [item][16][2]
    const float* LSE; const float* Dd;         // [B,H,Lq] log2-domain log-sum-exp, D = rowsum(dO . O)
    float* dq; float* dk; float* dv;           // [B,L,H,d] fp32 (dq zero-initialised when chunks add into it atomically)
    float* dq_ws;                              // optional [nch][B,Lq,H,dk] slabs: one private partial dq per key chunk
    unsigned long long* keepbits;              // dropout keep flags [B*H*ntq*ntk][4] (written by the forward, p_drop > 0)
    int B, H, Lq, Lk, dk_dim, dv_dim, ntq, ntk, nch, ch_len;
    float scale, p_drop; uint64_t seed, offset;
};

__device__ __forceinline__ u32x4m_t mha_frag(const u32x4m_t* base, long rec, int ds, int st, int lane) {
    return base[(rec * ds + st) * 64 + lane];
}
typedef _Float16 f16x8f_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mha_mfma32(u32x4m_t a, u32x4m_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// fp16 operands: the K.Q^T products of all three kernels
__device__ __forceinline__ f32x4_t mha_mfma32h(u32x4m_t a, u32x4m_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8f_t, a), __builtin_bit_cast(f16x8f_t, b), c, 0, 0, 0);
}
// fp16 operands: P.V of the forward (probabilities relative to the running maximum are in [0, 1])
__device__ __forceinline__ f32x4_t mha_mfma16h(uint2 a, float p0, float p1, float p2, float p3, f32x4_t c) {
    f16x4f_t v; v[0] = (_Float16)p0; v[1] = (_Float16)p1; v[2] = (_Float16)p2; v[3] = (_Float16)p3;
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4f_t, a), v, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mha_mfma16(uint2 a, s16x4m_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4m_t, a), b, c, 0, 0, 0);
}
__device__ __forceinline__ s16x4m_t mha_bf16x4(float a, float b, float c, float d) {
    bf16x4m_t v; v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
    return __builtin_bit_cast(s16x4m_t, v);
}

// Workgroup = 4 waves working on 4 consecutive tiles of the NON-streamed axis (query tiles in the forward / dQ
// kernels, key tiles in the dK/dV kernel); the fragments of the streamed tile are fetched once per workgroup into a
// double-buffered LDS stage (one barrier per step) instead of once per wave: a quarter of the L2 traffic.
struct StageRegs { u32x4m_t a, b; uint2 c0, c1, d; float e; };

// keep bits of a 16x16 tile, written by the forward: word r, bit l = keep flag of the element lane l holds in register r
// of the swapped-orientation tile (query l & 15, key 4*(l >> 4) + r)
__device__ __forceinline__ float mha_keep_swapped(const unsigned long long* kb, int lane, int r, float inv) {
    return ((kb[r] >> lane) & 1ull) ? inv : 0.f;
}
// the same element seen from the non-swapped orientation: lane holds key (lane & 15) and query 4*(lane >> 4) + r
__device__ __forceinline__ float mha_keep_plain(const unsigned long long* kb, int lane, int r, float inv) {
    const int key = lane & 15, ql = 4 * (lane >> 4) + r;
    return ((kb[key & 3] >> (ql + 16 * (key >> 2))) & 1ull) ? inv : 0.f;
}

// ---- forward ---------------------------------------------------------------------------------------
// TDK / TDV: head dims fixed at compile time (0: taken from the arguments - every step guard is then a branch around an
// MFMA; the instantiated pairs fold them away: 131 -> 12 branches, 192 -> ~120 registers in the dK/dV kernel)
template <int TDK, int TDV>
__global__ __launch_bounds__(256) void mha_fwd_kernel(MhaArgs a) {
    __shared__ u32x4m_t sK[2][MHA_DSK * 64];
    __shared__ uint2 sV[2][MHA_DVT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const int nqg = (a.ntq + 3) / 4;
    const int ch = blockIdx.x % a.nch; const int qg = (blockIdx.x / a.nch) % nqg; const long bh = blockIdx.x / ((long)a.nch * nqg);
    const int b = (int)(bh / a.H);
    const int dkd = TDK ? TDK : a.dk_dim, dvd = TDV ? TDV : a.dv_dim;
    const int dsk = (dkd + 31) / 32, dvt = (dvd + 15) / 16;
    const int qt = qg * 4 + wave;
    const bool active = qt < a.ntq;
    const int qtc = min(qt, a.ntq - 1);
    const int q = qtc * 16 + (lane & 15);
    const long ld4 = (a.Lk + 3) & ~3;
    const float inv_keep = 1.f / (1.f - a.p_drop);

    u32x4m_t qf[MHA_DSK];
#pragma unroll
    for (int st = 0; st < MHA_DSK; ++st) qf[st] = (st < dsk) ? mha_frag(a.Qf, bh * a.ntq + qtc, dsk, st, lane) : (u32x4m_t){0u, 0u, 0u, 0u};
    f32x4_t o[MHA_DVT];
#pragma unroll
    for (int d = 0; d < MHA_DVT; ++d) o[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    const int kt0 = ch * a.ch_len, kt1 = min(kt0 + a.ch_len, a.ntk);
    auto fetch = [&](int kt, StageRegs& r) {
        if (tid < dsk * 64) r.a = a.Kf[(bh * a.ntk + kt) * dsk * 64 + tid];
        if (tid < dvt * 64) r.c0 = a.V16[(bh * a.ntk + kt) * dvt * 64 + tid];
    };
    auto commit = [&](int buf, const StageRegs& r) {
        if (tid < dsk * 64) sK[buf][tid] = r.a;
        if (tid < dvt * 64) sV[buf][tid] = r.c0;
    };
    StageRegs sr;
    fetch(kt0, sr); commit(0, sr);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) fetch(kt + 1, sr);
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < MHA_DSK; ++st)
            if (st < dsk) s = mha_mfma32h(sK[buf][st * 64 + lane], qf[st], s);
        const int kb = kt * 16 + 4 * (lane >> 4);
        float sv[4], tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kb + r;
            const bool ok = key < a.Lk && !(a.mask && a.mask[(long)b * a.Lk + min(key, a.Lk - 1)]);
            sv[r] = ok ? s[r] : -INFINITY;
            tmax = fmaxf(tmax, sv[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));           // max over the tile's 16 keys of this lane's query
        const float mn = fmaxf(m, tmax);
        // branch-free (the MFMAs below must run for the whole wave): m = -inf means o = l = 0, so alpha is moot
        const float alpha = (m > -INFINITY) ? __builtin_amdgcn_exp2f(m - mn) : 0.f;
        float p[4], pd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[r] = (sv[r] > -INFINITY) ? __builtin_amdgcn_exp2f(sv[r] - mn) : 0.f; pd[r] = p[r]; }
        if (a.p_drop > 0.f) {
            // one Philox block per lane: element index (row * ld4 + key), 4 consecutive keys, ld4 % 4 == 0
            float ks[4];
            spe_drop_scale4(a.seed, a.offset, (uint64_t)((bh * a.Lq + min(q, a.Lq - 1)) * ld4 + kb), a.p_drop, ks);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pd[r] *= ks[r];
                const unsigned long long w = __ballot(ks[r] != 0.f);
                if (active && lane == r) a.keepbits[((bh * a.ntq + qt) * a.ntk + kt) * 4 + r] = w;
            }
        }
        l = l * alpha + (p[0] + p[1] + p[2] + p[3]);
        m = mn;
#pragma unroll
        for (int d = 0; d < MHA_DVT; ++d)
            if (d < dvt) {
                o[d] *= alpha;
                o[d] = mha_mfma16h(sV[buf][d * 64 + lane], pd[0], pd[1], pd[2], pd[3], o[d]);
            }
        if (kt + 1 < kt1) commit(buf ^ 1, sr);
        __syncthreads();
    }
    if (!active) return;
    // partials: O (relative to m), and per query m and the lane-group-summed l
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const long item = (bh * a.ntq + qt) * a.nch + ch;
#pragma unroll
    for (int d = 0; d < MHA_DVT; ++d)
        if (d < dvt) *reinterpret_cast<f32x4_t*>(a.Opart + ((item * dvt + d) * 64 + lane) * 4) = o[d];
    if (lane < 16) { a.ML[(item * 16 + lane) * 2] = m; a.ML[(item * 16 + lane) * 2 + 1] = l; }
}

// O[b, q, h, :] = sum_c Opart_c * 2^(m_c - M) / L ; LSE = M + log2(L)
__global__ __launch_bounds__(256) void mha_merge_kernel(const float* __restrict__ Opart, const float* __restrict__ ML, float* __restrict__ O,
                                                        float* __restrict__ LSE, int B, int H, int Lq, int ntq, int nch, int dv_dim) {
    const int dvt = (dv_dim + 15) / 16;
    const long total = (long)B * H * ntq * dvt * 64;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63); long t = i >> 6;
    const int d = (int)(t % dvt); t /= dvt;
    const int qt = (int)(t % ntq); const long bh = t / ntq;
    const int b = (int)(bh / H), h = (int)(bh % H);
    const int ql = lane & 15, q = qt * 16 + ql;
    if (q >= Lq) return;
    const long item0 = (bh * ntq + qt) * nch;
    float M = -INFINITY;
    for (int c = 0; c < nch; ++c) M = fmaxf(M, ML[((item0 + c) * 16 + ql) * 2]);
    float L = 0.f; f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nch; ++c) {
        const float mc = ML[((item0 + c) * 16 + ql) * 2];
        if (mc == -INFINITY) continue;
        const float w = __builtin_amdgcn_exp2f(mc - M);
        L += ML[((item0 + c) * 16 + ql) * 2 + 1] * w;
        acc += *reinterpret_cast<const f32x4_t*>(Opart + (((item0 + c) * dvt + d) * 64 + lane) * 4) * w;
    }
    const float inv = 1.f / L;
    const int dc = d * 16 + 4 * (lane >> 4);
    float* dst = O + ((long)b * Lq + q) * ((long)H * dv_dim) + (long)h * dv_dim;
#pragma unroll
    for (int r = 0; r < 4; ++r) if (dc + r < dv_dim) dst[dc + r] = acc[r] * inv;
    if (d == 0 && lane < 16) LSE[bh * Lq + q] = M + __builtin_amdgcn_logf(L);
}

// ---- backward: dQ ------------------------------------------------------------------------------------
template <int TDK, int TDV>
__global__ __launch_bounds__(256) void mha_bwd_dq_kernel(MhaArgs a) {
    __shared__ u32x4m_t sK[2][MHA_DSK * 64];
    __shared__ u32x4m_t sV[2][MHA_DSV * 64];
    __shared__ uint2 sK16[2][MHA_DKT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const int nqg = (a.ntq + 3) / 4;
    const int ch = blockIdx.x % a.nch; const int qg = (blockIdx.x / a.nch) % nqg; const long bh = blockIdx.x / ((long)a.nch * nqg);
    const int b = (int)(bh / a.H), h = (int)(bh % a.H);
    const int dkd = TDK ? TDK : a.dk_dim, dvd = TDV ? TDV : a.dv_dim;
    const int dsk = (dkd + 31) / 32, dsv = (dvd + 31) / 32, dkt = (dkd + 15) / 16;
    const int qt = qg * 4 + wave;
    const bool active = qt < a.ntq;
    const int qtc = min(qt, a.ntq - 1);
    const int q = qtc * 16 + (lane & 15);
    const bool qv = active && q < a.Lq;
    const float inv_keep = 1.f / (1.f - a.p_drop);

    u32x4m_t qf[MHA_DSK], dof[MHA_DSV];
#pragma unroll
    for (int st = 0; st < MHA_DSK; ++st) qf[st] = (st < dsk) ? mha_frag(a.Qf, bh * a.ntq + qtc, dsk, st, lane) : (u32x4m_t){0u, 0u, 0u, 0u};
#pragma unroll
    for (int st = 0; st < MHA_DSV; ++st) dof[st] = (st < dsv) ? mha_frag(a.dOf, bh * a.ntq + qtc, dsv, st, lane) : (u32x4m_t){0u, 0u, 0u, 0u};
    const float lse = qv ? a.LSE[bh * a.Lq + q] : 0.f, Dq = qv ? a.Dd[bh * a.Lq + q] : 0.f;
    f32x4_t g[MHA_DKT];
#pragma unroll
    for (int d = 0; d < MHA_DKT; ++d) g[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int kt0 = ch * a.ch_len, kt1 = min(kt0 + a.ch_len, a.ntk);
    auto fetch = [&](int kt, StageRegs& r) {
        if (tid < dsk * 64) r.a = a.Kf[(bh * a.ntk + kt) * dsk * 64 + tid];
        if (tid < dsv * 64) r.b = a.Vf[(bh * a.ntk + kt) * dsv * 64 + tid];
        if (tid < dkt * 64) r.c0 = a.K16[(bh * a.ntk + kt) * dkt * 64 + tid];
        if (tid + 256 < dkt * 64) r.c1 = a.K16[(bh * a.ntk + kt) * dkt * 64 + tid + 256];
    };
    auto commit = [&](int buf, const StageRegs& r) {
        if (tid < dsk * 64) sK[buf][tid] = r.a;
        if (tid < dsv * 64) sV[buf][tid] = r.b;
        if (tid < dkt * 64) sK16[buf][tid] = r.c0;
        if (tid + 256 < dkt * 64) sK16[buf][tid + 256] = r.c1;
    };
    StageRegs sr;
    fetch(kt0, sr); commit(0, sr);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) fetch(kt + 1, sr);
        f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < MHA_DSK; ++st)
            if (st < dsk) s = mha_mfma32h(sK[buf][st * 64 + lane], qf[st], s);
#pragma unroll
        for (int st = 0; st < MHA_DSV; ++st)
            if (st < dsv) dp = mha_mfma32(sV[buf][st * 64 + lane], dof[st], dp);
        unsigned long long kbits[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        if (a.p_drop > 0.f) {
#pragma unroll
            for (int r = 0; r < 4; ++r) kbits[r] = a.keepbits[((bh * a.ntq + qtc) * a.ntk + kt) * 4 + r];
        }
        const int kb = kt * 16 + 4 * (lane >> 4);
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kb + r;
            const bool ok = qv && key < a.Lk && !(a.mask && a.mask[(long)b * a.Lk + min(key, a.Lk - 1)]);
            const float p = ok ? __builtin_amdgcn_exp2f(s[r] - lse) : 0.f;
            const float keep = (a.p_drop > 0.f) ? mha_keep_swapped(kbits, lane, r, inv_keep) : 1.f;
            ds[r] = p * (dp[r] * keep - Dq);
        }
        const s16x4m_t db = mha_bf16x4(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
        for (int d = 0; d < MHA_DKT; ++d)
            if (d < dkt) g[d] = mha_mfma16(sK16[buf][d * 64 + lane], db, g[d]);
        if (kt + 1 < kt1) commit(buf ^ 1, sr);
        __syncthreads();
    }
    if (!qv) return;
    // dq[b, q, h, d] += scale * g ; lane = (query, 4 consecutive d)
    const bool slab = a.nch > 1 && a.dq_ws;     // private slab per chunk (summed by the caller): no atomics, fixed order
    float* dst = (slab ? a.dq_ws + (long)ch * a.B * a.Lq * a.H * dkd : a.dq) + (((long)b * a.Lq + q) * a.H + h) * dkd;
#pragma unroll
    for (int d = 0; d < MHA_DKT; ++d)
        if (d < dkt) {
            const int dc = d * 16 + 4 * (lane >> 4);
            if ((slab || a.nch == 1) && dc + 3 < dkd && (dkd & 3) == 0) {
                *reinterpret_cast<float4*>(dst + dc) = make_float4(g[d][0] * a.scale, g[d][1] * a.scale, g[d][2] * a.scale, g[d][3] * a.scale);
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (dc + r < dkd) {
                    if (a.nch > 1 && !slab) atomicAdd(dst + dc + r, g[d][r] * a.scale); else dst[dc + r] = g[d][r] * a.scale;
                }
        }
}

// ---- backward: dK, dV ----------------------------------------------------------------------------------
template <int TDK, int TDV>
__global__ __launch_bounds__(256) void mha_bwd_dkv_kernel(MhaArgs a) {
    __shared__ u32x4m_t sQ[2][MHA_DSK * 64];
    __shared__ u32x4m_t sdO[2][MHA_DSV * 64];
    __shared__ uint2 sQ16[2][MHA_DKT * 64];
    __shared__ uint2 sdO16[2][MHA_DVT * 64];
    __shared__ float sLD[2][32];                           // LSE[16], D[16] of the query tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const int nkg = (a.ntk + 3) / 4;
    const int kg = blockIdx.x % nkg; const long bh = blockIdx.x / nkg;
    const int b = (int)(bh / a.H), h = (int)(bh % a.H);
    const int dkd = TDK ? TDK : a.dk_dim, dvd = TDV ? TDV : a.dv_dim;
    const int dsk = (dkd + 31) / 32, dsv = (dvd + 31) / 32, dkt = (dkd + 15) / 16, dvt = (dvd + 15) / 16;
    const int kt = kg * 4 + wave;
    const bool active = kt < a.ntk;
    const int ktc = min(kt, a.ntk - 1);
    const int key = ktc * 16 + (lane & 15);                // non-swapped orientation: a lane owns one key and 4 queries
    const bool kv = active && key < a.Lk && !(a.mask && a.mask[(long)b * a.Lk + min(key, a.Lk - 1)]);
    const float inv_keep = 1.f / (1.f - a.p_drop);

    u32x4m_t kf[MHA_DSK], vf[MHA_DSV];
#pragma unroll
    for (int st = 0; st < MHA_DSK; ++st) kf[st] = (st < dsk) ? mha_frag(a.Kf, bh * a.ntk + ktc, dsk, st, lane) : (u32x4m_t){0u, 0u, 0u, 0u};
#pragma unroll
    for (int st = 0; st < MHA_DSV; ++st) vf[st] = (st < dsv) ? mha_frag(a.Vf, bh * a.ntk + ktc, dsv, st, lane) : (u32x4m_t){0u, 0u, 0u, 0u};
    f32x4_t gk[MHA_DKT], gv[MHA_DVT];
#pragma unroll
    for (int d = 0; d < MHA_DKT; ++d) gk[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < MHA_DVT; ++d) gv[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int qt, StageRegs& r) {
        if (tid < dsk * 64) r.a = a.Qf[(bh * a.ntq + qt) * dsk * 64 + tid];
        if (tid < dsv * 64) r.b = a.dOf[(bh * a.ntq + qt) * dsv * 64 + tid];
        if (tid < dkt * 64) r.c0 = a.Q16[(bh * a.ntq + qt) * dkt * 64 + tid];
        if (tid + 256 < dkt * 64) r.c1 = a.Q16[(bh * a.ntq + qt) * dkt * 64 + tid + 256];
        if (tid < dvt * 64) r.d = a.dO16[(bh * a.ntq + qt) * dvt * 64 + tid];
        if (tid < 32) {
            const int qq = min(qt * 16 + (tid & 15), a.Lq - 1);
            r.e = (tid < 16) ? a.LSE[bh * a.Lq + qq] : a.Dd[bh * a.Lq + qq];
        }
    };
    auto commit = [&](int buf, const StageRegs& r) {
        if (tid < dsk * 64) sQ[buf][tid] = r.a;
        if (tid < dsv * 64) sdO[buf][tid] = r.b;
        if (tid < dkt * 64) sQ16[buf][tid] = r.c0;
        if (tid + 256 < dkt * 64) sQ16[buf][tid + 256] = r.c1;
        if (tid < dvt * 64) sdO16[buf][tid] = r.d;
        if (tid < 32) sLD[buf][tid] = r.e;
    };
    StageRegs sr;
    fetch(0, sr); commit(0, sr);
    __syncthreads();
    for (int qt = 0; qt < a.ntq; ++qt) {
        const int buf = qt & 1;
        if (qt + 1 < a.ntq) fetch(qt + 1, sr);
        f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < MHA_DSK; ++st)
            if (st < dsk) s = mha_mfma32h(sQ[buf][st * 64 + lane], kf[st], s);         // C[q, key]
#pragma unroll
        for (int st = 0; st < MHA_DSV; ++st)
            if (st < dsv) dp = mha_mfma32(sdO[buf][st * 64 + lane], vf[st], dp);
        unsigned long long kbits[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        if (a.p_drop > 0.f) {
#pragma unroll
            for (int r = 0; r < 4; ++r) kbits[r] = a.keepbits[((bh * a.ntq + qt) * a.ntk + ktc) * 4 + r];
        }
        const int qb = qt * 16 + 4 * (lane >> 4);
        float pd[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = kv && qb + r < a.Lq;
            const float lse = sLD[buf][4 * (lane >> 4) + r], Dq = sLD[buf][16 + 4 * (lane >> 4) + r];
            const float p = ok ? __builtin_amdgcn_exp2f(s[r] - lse) : 0.f;
            const float keep = (a.p_drop > 0.f) ? mha_keep_plain(kbits, lane, r, inv_keep) : 1.f;
            pd[r] = p * keep;
            ds[r] = p * (dp[r] * keep - Dq);
        }
        const s16x4m_t pb = mha_bf16x4(pd[0], pd[1], pd[2], pd[3]), db = mha_bf16x4(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
        for (int d = 0; d < MHA_DVT; ++d)
            if (d < dvt) gv[d] = mha_mfma16(sdO16[buf][d * 64 + lane], pb, gv[d]);
#pragma unroll
        for (int d = 0; d < MHA_DKT; ++d)
            if (d < dkt) gk[d] = mha_mfma16(sQ16[buf][d * 64 + lane], db, gk[d]);
        if (qt + 1 < a.ntq) commit(buf ^ 1, sr);
        __syncthreads();
    }
    if (!active || key >= a.Lk) return;
    float* dk_ = a.dk + (((long)b * a.Lk + key) * a.H + h) * dkd;
    float* dv_ = a.dv + (((long)b * a.Lk + key) * a.H + h) * dvd;
    // the Q fragments carry scale * log2(e): dK = dS^T (scale q) = dS^T Qpacked * ln 2
#pragma unroll
    for (int d = 0; d < MHA_DKT; ++d)
        if (d < dkt) {
            const int dc = d * 16 + 4 * (lane >> 4);
            if (dc + 3 < dkd && (dkd & 3) == 0) { *reinterpret_cast<float4*>(dk_ + dc) = make_float4(gk[d][0] * MHA_LN2, gk[d][1] * MHA_LN2, gk[d][2] * MHA_LN2, gk[d][3] * MHA_LN2); continue; }
#pragma unroll
            for (int r = 0; r < 4; ++r) if (dc + r < dkd) dk_[dc + r] = gk[d][r] * MHA_LN2;
        }
#pragma unroll
    for (int d = 0; d < MHA_DVT; ++d)
        if (d < dvt) {
            const int dc = d * 16 + 4 * (lane >> 4);
            if (dc + 3 < dvd && (dvd & 3) == 0) { *reinterpret_cast<float4*>(dv_ + dc) = make_float4(gv[d][0], gv[d][1], gv[d][2], gv[d][3]); continue; }
#pragma unroll
            for (int r = 0; r < 4; ++r) if (dc + r < dvd) dv_[dc + r] = gv[d][r];
        }
}

// ---- C ABI ----------------------------------------------------------------------------------------------
static int mha_fill(MhaArgs& a, int B, int H, int Lq, int Lk, int dk, int dv, int nch) {
    if (dk < 1 || dk > 32 * MHA_DSK || dv < 1 || dv > 16 * MHA_DVT) return -2;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.dk_dim = dk; a.dv_dim = dv;
    a.ntq = (Lq + 15) / 16; a.ntk = (Lk + 15) / 16;
    if (nch < 1) nch = 1; if (nch > a.ntk) nch = a.ntk;
    a.ch_len = (a.ntk + nch - 1) / nch;
    a.nch = (a.ntk + a.ch_len - 1) / a.ch_len;
    return 0;
}

// see include/spe_hip.h
extern "C" int spe_mha_plan(int B, int H, int Lq, int Lk, int* nch) {
    const int ntq = (Lq + 15) / 16, ntk = (Lk + 15) / 16;
    long items = (long)B * H * ((ntq + 3) / 4);                // workgroups of 4 query tiles
#ifndef MHA_TARGET
#define MHA_TARGET 1024                                         // developer knob (tools/ab.py): workgroups the plan aims at
#endif
    int c = (int)((MHA_TARGET + items - 1) / items);           // ~4 workgroups per CU
    if (c > ntk / 4) c = ntk / 4;                              // at least 4 key tiles per chunk
    if (c < 1) c = 1;
    const int len = (ntk + c - 1) / c;
    *nch = (ntk + len - 1) / len;
    return 0;
}
extern "C" int spe_mha_fwd(const void* Qf, const void* Kf, const void* V16, const void* mask, float* Opart, float* ML, float* O,
                           float* LSE, void* keepbits, int B, int H, int Lq, int Lk, int dk, int dv, int nch, float p_drop,
                           uint64_t seed, uint64_t offset, hipStream_t st) {
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
    MhaArgs a = {};
    const int rc = mha_fill(a, B, H, Lq, Lk, dk, dv, nch);
    if (rc) return rc;
    if (a.nch != nch) return -5;                               // the caller sized the workspaces with spe_mha_plan
    a.Qf = (const u32x4m_t*)Qf; a.Kf = (const u32x4m_t*)Kf; a.V16 = (const uint2*)V16; a.mask = (const unsigned char*)mask;
    a.Opart = Opart; a.ML = ML; a.p_drop = p_drop; a.seed = seed; a.offset = offset;
    a.keepbits = (unsigned long long*)keepbits;
    if (p_drop > 0.f && !keepbits) return -2;
    const long wgs = (long)B * H * ((a.ntq + 3) / 4) * a.nch;
    if (dk == 96 && dv == 48) hipLaunchKernelGGL((mha_fwd_kernel<96, 48>), dim3((unsigned)wgs), dim3(256), 0, st, a);
    else if (dk == 48 && dv == 48) hipLaunchKernelGGL((mha_fwd_kernel<48, 48>), dim3((unsigned)wgs), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((mha_fwd_kernel<0, 0>), dim3((unsigned)wgs), dim3(256), 0, st, a);
    SPE_CHECK_LAUNCH();
    const long total = (long)B * H * a.ntq * ((dv + 15) / 16) * 64;
    hipLaunchKernelGGL(mha_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Opart, ML, O, LSE, B, H, Lq, a.ntq,
                       a.nch, dv);
    SPE_CHECK_LAUNCH();
    return 0;
}
extern "C" int spe_mha_bwd(const void* Qf, const void* Kf, const void* Vf, const void* dOf, const void* K16, const void* Q16,
                           const void* dO16, const void* mask, const float* LSE, const float* D, const void* keepbits, float* dq,
                           float* dq_ws, float* dk_, float* dv_, int B, int H, int Lq, int Lk, int dk, int dv, int nch, float scale, float p_drop,
                           hipStream_t st) {
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
    MhaArgs a = {};
    const int rc = mha_fill(a, B, H, Lq, Lk, dk, dv, nch);
    if (rc) return rc;
    a.Qf = (const u32x4m_t*)Qf; a.Kf = (const u32x4m_t*)Kf; a.Vf = (const u32x4m_t*)Vf; a.dOf = (const u32x4m_t*)dOf;
    a.K16 = (const uint2*)K16; a.Q16 = (const uint2*)Q16; a.dO16 = (const uint2*)dO16; a.mask = (const unsigned char*)mask;
    a.LSE = LSE; a.Dd = D; a.dq = dq; a.dq_ws = dq_ws; a.dk = dk_; a.dv = dv_;
    a.scale = scale; a.p_drop = p_drop;
    a.keepbits = (unsigned long long*)const_cast<void*>(keepbits);
    if (p_drop > 0.f && !keepbits) return -2;
    long wgs = (long)B * H * ((a.ntq + 3) / 4) * a.nch;
    if (dk == 96 && dv == 48) hipLaunchKernelGGL((mha_bwd_dq_kernel<96, 48>), dim3((unsigned)wgs), dim3(256), 0, st, a);
    else if (dk == 48 && dv == 48) hipLaunchKernelGGL((mha_bwd_dq_kernel<48, 48>), dim3((unsigned)wgs), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((mha_bwd_dq_kernel<0, 0>), dim3((unsigned)wgs), dim3(256), 0, st, a);
    SPE_CHECK_LAUNCH();
    wgs = (long)B * H * ((a.ntk + 3) / 4);
    if (dk == 96 && dv == 48) hipLaunchKernelGGL((mha_bwd_dkv_kernel<96, 48>), dim3((unsigned)wgs), dim3(256), 0, st, a);
    else if (dk == 48 && dv == 48) hipLaunchKernelGGL((mha_bwd_dkv_kernel<48, 48>), dim3((unsigned)wgs), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((mha_bwd_dkv_kernel<0, 0>), dim3((unsigned)wgs), dim3(256), 0, st, a);
    SPE_CHECK_LAUNCH();
    return 0;
}
