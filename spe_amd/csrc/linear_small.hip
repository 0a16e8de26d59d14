// nn.Linear on a few hundred rows - the decoder / encoder / head side of the SPE hot path (reference models/transformer.py:
// 21-33, 206-250, 355-427; models/conditional_detr.py:68-116: every projection of the [B, 2Q, d] query stream, R = 400 rows
// at cfg2) - as ONE launch each way.
//
// These products are latency-, not throughput-bound (400 x 384 x 384 = 0.1 GFLOP): what they cost is the number of dependent
// launches and the serial load round trips inside each.  The bf16-copy path of gemm_bf16.hip ran the forward as  conversion
// (x -> x_hi, x_lo) + split GEMM  (5 + 14 us) and the backward as  conversion (dy -> dy16, + bias gradient) + input-gradient
// GEMM + weight-gradient GEMM  (5 + 6 + 7.5 us, three kernel boundaries); here
//   spe_linear_small_fwd : y = act(x W^T + b) straight from the fp32 activations: a workgroup converts its slab of x to (hi, lo)
//                          bf16 pairs while staging it (the split of precision mode bf16s; lo is skipped for single-term
//                          products), multiplies it with the cached bf16 weight parts, and - the column-0 workgroups - also
//                          leaves x_hi [R, K] behind, which is all the backward needs of x;
//   spe_linear_small_bwd : dx = dy' W, dW = dy'^T x and db = colsum(dy') from ONE grid (dy' = dy times the derivative of a fused
//                          ReLU / GELU): the first workgroups own tiles of dx (contraction over the outputs, operands contiguous
//                          along it: the forward's program with the activation derivative applied while dy is staged), the others
//                          tiles of dW (contraction over the rows: LDS transpose reads, as gemm_bf16tn.hip), each rounding the
//                          fp32 dy slab it needs itself; the dW workgroups of k-tile 0 add up the bias gradient of their columns in
//                          a fixed order (no atomics).
// Geometry for latency: 32 x 32 output tiles (400 x 384 -> 156 workgroups), 4 waves = 2 x 2 MFMA tiles of 16 x 16, and the WHOLE
// contraction (up to 384 elements / 512 rows) staged in LDS at once - every load of a workgroup is in flight together, ONE round
// trip, one barrier, 12-36 MFMAs per wave, store.  Longer contractions (the FFN's 2048) loop over such chunks with the next chunk's
// loads in flight while the current one is multiplied.
#include "common.h"
#include "gemm16_epilogue.h"

#define LS_T 32                     // tile edge
// contraction chunk KC of the k-contiguous products: 384 ("latency": one workgroup per CU, everything in flight at once - grids of
// at most ~1.25 waves of workgroups) or 128 ("throughput": 3-4 workgroups per CU overlap each other's round trips - the FFN's 2048-wide
// sides); LDS rows of KC + 8 bf16 = 196 / 68 dwords: 16 consecutive rows start 4 dwords apart (conflict-free b128 reads).
// Row chunk RC of the row-contraction product: 512 / 256 likewise.
#define LS_LDT (LS_T + 8)
#define LS_MAXBLK 16

typedef short s16x4l_t __attribute__((ext_vector_type(4)));
typedef short s16x8l_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4l_t __attribute__((ext_vector_type(4)));

struct LinSmallArgs {
    Gemm16Args g;                   // k-contiguous product: B / Blo = the [N, K] operand (hi / lo), epilogue fields (C, C2, bias, act, M, N, K, ldc)
    const float* x; long ldx;       // its fp32 [R, K] operand (forward: activations; backward dx part: dy)
    unsigned short* x16;            // forward: optional bf16(x) [R, K] side output
    const float* aux; int act;      // backward: derivative of the fused activation applied to dy while staging
    const unsigned short* xs16;     // backward dW part: x_hi [R, K]
    float* dW; float* db;           // [N, K], [N]
    int R, N, K, tiles_dx;
    // group mode (nblk > 0): nblk Linears [Nblk, K] sharing the input x - the column blocks of one stacked product, each with its own
    // weight / bias / output (forward) and dy / W^T / dW / db (backward) pointers: no stacked copies of anything
    int nblk, Nblk;
    const unsigned short* Wb[LS_MAXBLK]; const unsigned short* Wlob[LS_MAXBLK]; const float* biasb[LS_MAXBLK]; float* yb[LS_MAXBLK];
    const float* addb[LS_MAXBLK];   // forward, optional: y[i] += add[i] ([R, Nblk] fp32) - q = sa_qcontent_proj(tgt) + sa_qpos_proj(query_pos) without an add launch
    const float* dyb[LS_MAXBLK]; const unsigned short* WTb[LS_MAXBLK]; float* dWb[LS_MAXBLK]; float* dbb[LS_MAXBLK];
};

__device__ __forceinline__ float ls_dact(float v, float h, int act) {
    if (act == 1) return h > 0.f ? v : 0.f;
    if (act == 2) {                 // the arithmetic of cvt_bf16_kernel / act_bwd_kernel
        const float cdf = 0.5f * (1.f + spe_erff(h * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * h * h);
        return v * (cdf + h * pdf);
    }
    return v;
}
__device__ __forceinline__ uint2 ls_pack4(float a, float b, float c, float d) {
    bf16x4l_t v; v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
    return __builtin_bit_cast(uint2, v);
}

// ---- C[r][n] = sum_k A[r][k] B[n][k] on a 32 x 32 tile: A fp32 [rows][lda] (rounded / split to bf16 while staged, optionally times
// act'(aux)), B bf16 [ncols][K] (+ low part).  DACT: backward dx part.  Returns the tile in acc (one 16 x 16 MFMA tile per wave).
// Ab / Bb (group backward, dx part): the contraction runs over the nblk column blocks of width Nblk; chunk k0 lies in block k0 / Nblk,
// whose A rows (dy_i [R][Nblk]) and B rows (W_i^T [K][Nblk]) have row stride Nblk; a NULL A block contributes nothing.
template <int LS_KC, bool SPLIT, bool DACT>
__device__ __forceinline__ void ls_nt_tile(const float* __restrict__ A, long lda, const float* __restrict__ aux, int act, int r0, int R,
                                           const unsigned short* __restrict__ B, const unsigned short* __restrict__ Blo, int n0, int N, int K,
                                           unsigned short* x16, bool write_x16, unsigned short* smem, f32x4_t& acc,
                                           const float* const* Ab = nullptr, const unsigned short* const* Bb = nullptr, int Nblk = 0) {
    constexpr int LS_LD = LS_KC + 8, LS_NA = LS_KC / 32, LS_NB = LS_KC / 64;      // float4 / 16-B chunks per thread of a 32 x KC fp32 / bf16 slab
    unsigned short* sA = smem;                         // [32][LS_LD]
    unsigned short* sB = sA + LS_T * LS_LD;
    unsigned short* sAl = sB + LS_T * LS_LD;           // SPLIT
    unsigned short* sBl = sAl + LS_T * LS_LD;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1, fr = lane & 15, fk = (lane >> 4) * 8;
    acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float4 xa[LS_NA], ha[DACT ? LS_NA : 1];
    u32x4g_t wb[LS_NB], wl[SPLIT ? LS_NB : 1];
    // thread t: float4 columns (t & 31) + 32 m of rows (t >> 5) + 8 j of the fp32 slab; 16-B chunks (t & 15) + 16 m of rows (t >> 4) + 16 j of the bf16 slab
    auto load = [&](int k0) {
        const int kw = min(LS_KC, K - k0), n4 = kw >> 2, n8 = kw >> 3;           // K % 8 == 0
        const float* Ak = A; const unsigned short* Bk = B; long lda_ = lda, ldb_ = K; int kk = k0;
        if (Ab) { const int blk = k0 / Nblk; kk = k0 - blk * Nblk; Ak = Ab[blk]; Bk = Bb[blk]; lda_ = ldb_ = Nblk; }
#pragma unroll
        for (int i = 0; i < LS_NA; ++i) {
            const int rl = (t >> 5) + 8 * (i & 3), c4 = (t & 31) + 32 * (i >> 2);
            if (c4 >= n4) continue;
            const long off = (long)min(r0 + rl, R - 1) * lda_ + kk + c4 * 4;
            xa[i] = Ak ? *reinterpret_cast<const float4*>(Ak + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (DACT) { if (aux) ha[i] = *reinterpret_cast<const float4*>(aux + off); }
        }
#pragma unroll
        for (int i = 0; i < LS_NB; ++i) {
            const int nl = (t >> 4) + 16 * (i & 1), c8 = (t & 15) + 16 * (i >> 1);
            if (c8 >= n8) continue;
            const long off = (long)min(n0 + nl, N - 1) * ldb_ + kk + c8 * 8;
            wb[i] = *reinterpret_cast<const u32x4g_t*>(Bk + off);
            if constexpr (SPLIT) wl[i] = *reinterpret_cast<const u32x4g_t*>(Blo + off);
        }
    };
    load(0);
    for (int k0 = 0; k0 < K; k0 += LS_KC) {
        const int kw = min(LS_KC, K - k0), n4 = kw >> 2, n8 = kw >> 3;
        const int kpad = (kw + 31) & ~31;               // the MFMA steps are 32 deep: columns kw .. kpad-1 of both slabs are zeroed
#pragma unroll
        for (int i = 0; i < LS_NA; ++i) {
            const int rl = (t >> 5) + 8 * (i & 3), c4 = (t & 31) + 32 * (i >> 2);
            if (c4 >= n4) continue;
            float4 v = xa[i];
            if (r0 + rl >= R) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (DACT) {
                if (aux) { v.x = ls_dact(v.x, ha[i].x, act); v.y = ls_dact(v.y, ha[i].y, act); v.z = ls_dact(v.z, ha[i].z, act); v.w = ls_dact(v.w, ha[i].w, act); }
            }
            bf16x4l_t h; h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
            const uint2 uh = __builtin_bit_cast(uint2, h);
            *reinterpret_cast<uint2*>(sA + rl * LS_LD + c4 * 4) = uh;
            if constexpr (SPLIT) *reinterpret_cast<uint2*>(sAl + rl * LS_LD + c4 * 4) = ls_pack4(v.x - (float)h[0], v.y - (float)h[1], v.z - (float)h[2], v.w - (float)h[3]);
            if (write_x16 && r0 + rl < R) *reinterpret_cast<uint2*>(x16 + (long)(r0 + rl) * K + k0 + c4 * 4) = uh;
        }
#pragma unroll
        for (int i = 0; i < LS_NB; ++i) {
            const int nl = (t >> 4) + 16 * (i & 1), c8 = (t & 15) + 16 * (i >> 1);
            if (c8 >= n8) continue;
            const bool nv = n0 + nl < N;
            *reinterpret_cast<u32x4g_t*>(sB + nl * LS_LD + c8 * 8) = nv ? wb[i] : (u32x4g_t){0u, 0u, 0u, 0u};
            if constexpr (SPLIT) *reinterpret_cast<u32x4g_t*>(sBl + nl * LS_LD + c8 * 8) = nv ? wl[i] : (u32x4g_t){0u, 0u, 0u, 0u};
        }
        if (kpad > kw && t < 2 * LS_T * ((kpad - kw) >> 3)) {        // zero the ragged tail of the last 32-deep step (kw % 32 in {8, 16, 24})
            const int per = (kpad - kw) >> 3, idx = t % (LS_T * per), rl = idx / per, c8 = idx % per;
            unsigned short* base = (t < LS_T * per) ? sA : sB;
            *reinterpret_cast<u32x4g_t*>(base + rl * LS_LD + kw + c8 * 8) = (u32x4g_t){0u, 0u, 0u, 0u};
            if constexpr (SPLIT) *reinterpret_cast<u32x4g_t*>((base == sA ? sAl : sBl) + rl * LS_LD + kw + c8 * 8) = (u32x4g_t){0u, 0u, 0u, 0u};
        }
        if (k0 + LS_KC < K) load(k0 + LS_KC);           // the next chunk's loads are in flight while this one is multiplied
        __syncthreads();
        const int arow = (wm * 16 + fr) * LS_LD + fk, brow = (wn * 16 + fr) * LS_LD + fk;
        for (int ks = 0; ks < kpad; ks += 32) {
            const bf16x8_t a = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sA + arow + ks));
            const bf16x8_t b = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sB + brow + ks));
            if constexpr (SPLIT) {
                const bf16x8_t al = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sAl + arow + ks));
                const bf16x8_t bl = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sBl + brow + ks));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, a, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, al, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0);
        }
        if (k0 + LS_KC < K) __syncthreads();
    }
}

template <int KC, bool SPLIT>
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(LinSmallArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short ls_smem[];
    const int tiles_r = (p.g.M + LS_T - 1) / LS_T;
    const int tr = blockIdx.x % tiles_r, tn = blockIdx.x / tiles_r;
    f32x4_t acc1;
    if (p.nblk > 0) {                   // group: column tile tn lies in block tn * 32 / Nblk, which has its own weight, bias and output
        const int blk = tn * LS_T / p.Nblk, tl = tn - blk * (p.Nblk / LS_T);
        ls_nt_tile<KC, SPLIT, false>(p.x, p.ldx, nullptr, 0, tr * LS_T, p.g.M, p.Wb[blk], p.Wlob[blk], tl * LS_T, p.Nblk, p.g.K, p.x16,
                                 p.x16 != nullptr && tn == 0, ls_smem, acc1);
        Gemm16Args g = p.g;
        g.C = p.yb[blk]; g.bias = p.biasb[blk]; g.N = p.Nblk; g.ldc = p.Nblk;
        if (p.addb[blk]) {          // the lane's 4 columns of its row (the epilogue's ownership: gemm16_epilogue.h), alpha = 1
            const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
            const int m = tr * LS_T + (w >> 1) * 16 + (lane & 15), n = tl * LS_T + (w & 1) * 16 + (lane >> 4) * 4;
            if (m < p.g.M && n + 3 < p.Nblk) {
                const float4 a4 = *reinterpret_cast<const float4*>(p.addb[blk] + (long)m * p.Nblk + n);
                acc1[0] += a4.x; acc1[1] += a4.y; acc1[2] += a4.z; acc1[3] += a4.w;
            }
        }
        f32x4_t accg[1][1] = {{acc1}};
        gemm16_epilogue_plain<LS_T, LS_T>(g, accg, g.C, tr * LS_T, tl * LS_T);
        return;
    }
    ls_nt_tile<KC, SPLIT, false>(p.x, p.ldx, nullptr, 0, tr * LS_T, p.g.M, p.g.B, p.g.Blo, tn * LS_T, p.g.N, p.g.K, p.x16, p.x16 != nullptr && tn == 0,
                             ls_smem, acc1);
    f32x4_t acc[1][1] = {{acc1}};
    gemm16_epilogue_plain<LS_T, LS_T>(p.g, acc, p.g.C, tr * LS_T, tn * LS_T);
}

// ---- backward -----------------------------------------------------------------------------------------------------------
// 8 consecutive contraction rows (32*ks + 8*(lane>>4) ..) of column c0 + (lane & 15) of an LDS slab [rows][LS_LDT] (see gemm_bf16tn.hip)
__device__ __forceinline__ bf16x8_t ls_operand_t(const unsigned short* tile, int ks, int c0, int lane) {
    typedef __attribute__((address_space(3))) s16x4l_t lds_s4;
    const unsigned short* q = tile + (32 * ks + 8 * (lane >> 4) + ((lane & 15) >> 2)) * LS_LDT + c0 + 4 * (lane & 3);
    const s16x4l_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(q));
    const s16x4l_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(q + 4 * LS_LDT));
    const s16x8l_t v = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

template <int KC, int LS_RC>
__global__ __launch_bounds__(256) void linear_small_bwd_kernel(LinSmallArgs p) {
    constexpr int LS_ND = LS_RC / 32, LS_NX = LS_RC / 64;      // float4 / 16-B chunks per thread of an RC x 32 fp32 / bf16 slab
    extern __shared__ __attribute__((aligned(16))) unsigned short ls_smem[];
    const int R = p.R, N = p.N, K = p.K;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1, fr = lane & 15;
    if ((int)blockIdx.x < p.tiles_dx) {
        // ---- dx[r][k] = sum_n dy'[r][n] W^T[k][n]: the forward's program, "x" = dy (with act'), "W" = W^T [K rows][N], single-term
        const int tiles_r = (R + LS_T - 1) / LS_T;
        const int tr = blockIdx.x % tiles_r, tk = blockIdx.x / tiles_r;
        f32x4_t acc1;
        if (p.nblk > 0) ls_nt_tile<KC, false, true>(nullptr, 0, nullptr, 0, tr * LS_T, R, nullptr, nullptr, tk * LS_T, K, N, nullptr, false, ls_smem, acc1,
                                                    p.dyb, p.WTb, p.Nblk);
        else ls_nt_tile<KC, false, true>(p.x, N, p.aux, p.act, tr * LS_T, R, p.g.B, nullptr, tk * LS_T, K, N, nullptr, false, ls_smem, acc1);
        const int r = tr * LS_T + wm * 16 + fr, k = tk * LS_T + wn * 16 + (lane >> 4) * 4;
        if (r < R && k < K) *reinterpret_cast<float4*>(p.g.C + (long)r * K + k) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);     // K % 4 == 0
        return;
    }
    // ---- dW[n][k] = sum_r dy'[r][n] x[r][k] (+ db[n] = sum_r dy'[r][n] in the k-tile-0 workgroups): contraction over rows
    unsigned short* sD = ls_smem;                        // [LS_RC rows][LS_LDT] dy' slab (columns n0 .. n0+31)
    unsigned short* sX = sD + LS_RC * LS_LDT;            // [LS_RC rows][LS_LDT] x slab  (columns k0 .. k0+31)
    float* sred = reinterpret_cast<float*>(sX + LS_RC * LS_LDT);     // [32 row lanes][32 columns] partials of db
    const int b2 = blockIdx.x - p.tiles_dx;
    const int tiles_n = (N + LS_T - 1) / LS_T;
    const int tn = b2 % tiles_n, tk = b2 / tiles_n;
    int n0 = tn * LS_T; const int k0 = tk * LS_T;
    const float* dyp = p.x; float* dWp = p.dW; float* dbp = p.db; int Nld = N;
    if (p.nblk > 0) {                   // group: this column tile's block has its own dy [R][Nblk], dW [Nblk][K] and db [Nblk]
        const int blk = n0 / p.Nblk;
        n0 -= blk * p.Nblk; Nld = p.Nblk;
        dyp = p.dyb[blk]; dWp = p.dWb[blk]; dbp = p.dbb[blk];
        if (!dyp) return;               // an output nobody differentiated: its parameter gradients are not touched
    }
    const bool want_dw = dWp != nullptr, want_db = dbp != nullptr && tk == 0;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 dv[LS_ND], hv[LS_ND]; u32x4g_t xv[LS_NX];
    // thread t: float4 column t % 8 of rows t / 8 + 32 i of the dy slab; 16-B chunk t % 4 of rows t / 4 + 64 i of the x slab
    auto load = [&](int rc0) {
        const int cn = n0 + (t & 7) * 4, ck = k0 + (t & 3) * 8;
#pragma unroll
        for (int i = 0; i < LS_ND; ++i) {
            const int r = rc0 + (t >> 3) + 32 * i;
            if (r >= R) continue;
            const long off = (long)min(r, R - 1) * Nld + min(cn, Nld - 4);
            dv[i] = *reinterpret_cast<const float4*>(dyp + off);
            if (p.aux) hv[i] = *reinterpret_cast<const float4*>(p.aux + off);
        }
        if (want_dw) {
#pragma unroll
            for (int i = 0; i < LS_NX; ++i) {
                const int r = rc0 + (t >> 2) + 64 * i;
                if (r >= R) continue;
                xv[i] = *reinterpret_cast<const u32x4g_t*>(p.xs16 + (long)min(r, R - 1) * K + min(ck, K - 8));
            }
        }
    };
    load(0);
    for (int rc0 = 0; rc0 < R; rc0 += LS_RC) {
        const int rows = min(LS_RC, R - rc0), rpad = (rows + 31) & ~31;
        const bool cnv = n0 + (t & 7) * 4 < Nld, ckv = k0 + (t & 3) * 8 < K;
#pragma unroll
        for (int i = 0; i < LS_ND; ++i) {
            const int rl = (t >> 3) + 32 * i;
            if (rl >= rpad) continue;
            float4 v = dv[i];
            if (p.aux) { v.x = ls_dact(v.x, hv[i].x, p.act); v.y = ls_dact(v.y, hv[i].y, p.act); v.z = ls_dact(v.z, hv[i].z, p.act); v.w = ls_dact(v.w, hv[i].w, p.act); }
            if (rl >= rows || !cnv) v = make_float4(0.f, 0.f, 0.f, 0.f);
            bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
            *reinterpret_cast<uint2*>(sD + rl * LS_LDT + (t & 7) * 4) = ls_pack4(v.x, v.y, v.z, v.w);
        }
        if (want_dw) {
#pragma unroll
            for (int i = 0; i < LS_NX; ++i) {
                const int rl = (t >> 2) + 64 * i;
                if (rl >= rpad) continue;
                *reinterpret_cast<u32x4g_t*>(sX + rl * LS_LDT + (t & 3) * 8) = (rl < rows && ckv) ? xv[i] : (u32x4g_t){0u, 0u, 0u, 0u};
            }
        }
        if (rc0 + LS_RC < R) load(rc0 + LS_RC);
        __syncthreads();
        if (want_dw) {
            for (int ks = 0; ks < rpad / 32; ++ks) {
                const bf16x8_t a = ls_operand_t(sD, ks, wm * 16, lane);
                const bf16x8_t b = ls_operand_t(sX, ks, wn * 16, lane);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0);
            }
        }
        if (rc0 + LS_RC < R) __syncthreads();
    }
    if (want_dw) {
        const int n = n0 + wm * 16 + fr, k = k0 + wn * 16 + (lane >> 4) * 4;
        if (n < Nld && k < K) *reinterpret_cast<float4*>(dWp + (long)n * K + k) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    if (want_db) {          // 32 row lanes per column quad, added in a fixed order: deterministic, no atomics
        __syncthreads();
        *reinterpret_cast<float4*>(sred + (t >> 3) * 32 + (t & 7) * 4) = bsum;
        __syncthreads();
        if (t < LS_T && n0 + t < Nld) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) s += sred[i * 32 + t];
            dbp[n0 + t] = s;
        }
    }
}

static int ls_latency_tiles() {       // grids up to this many workgroups take the one-round-trip configuration (SPE_LS_WIDE_MAX, tuning only)
    static int v = -1;
    if (v < 0) v = SPE_KNOB("SPE_LS_WIDE_MAX", 320);
    return v;
}

// C-ABI: see include/spe_hip.h.  -2: unsupported shape / alignment (K, N multiples of 8; 16-B aligned operands; contiguous rows).
extern "C" int spe_linear_small_fwd(const float* x, long ldx, const void* W16, const void* W16lo, const float* bias, float* y, float* pre,
                                    void* x16_out, int R, int N, int K, long ldc, int act, hipStream_t stream) {
    if (R <= 0 || N <= 0) return 0;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (K <= 0 || (K & 7) || (ldx & 3) || !al16(x) || !al16(W16) || !al16(W16lo) || !al16(x16_out)) return -2;
    LinSmallArgs p = {};
    p.g.B = reinterpret_cast<const unsigned short*>(W16); p.g.Blo = reinterpret_cast<const unsigned short*>(W16lo);
    p.g.C = y; p.g.C2 = pre; p.g.bias = bias; p.g.M = R; p.g.N = N; p.g.K = K; p.g.ldc = ldc; p.g.alpha = 1.f; p.g.act = act; p.g.splitk = 1;
    p.x = x; p.ldx = ldx; p.x16 = reinterpret_cast<unsigned short*>(x16_out);
    const int tiles = ((R + LS_T - 1) / LS_T) * ((N + LS_T - 1) / LS_T);
    const bool split = W16lo != nullptr;
    const bool wide = tiles <= ls_latency_tiles();
    const int smem = (split ? 4 : 2) * LS_T * ((wide ? 384 : 128) + 8) * (int)sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_small_fwd_kernel<384, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           4 * LS_T * 392 * (int)sizeof(unsigned short));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (wide) {
        if (split) hipLaunchKernelGGL((linear_small_fwd_kernel<384, true>), dim3(tiles), dim3(256), smem, stream, p);
        else hipLaunchKernelGGL((linear_small_fwd_kernel<384, false>), dim3(tiles), dim3(256), smem, stream, p);
    } else {
        if (split) hipLaunchKernelGGL((linear_small_fwd_kernel<128, true>), dim3(tiles), dim3(256), smem, stream, p);
        else hipLaunchKernelGGL((linear_small_fwd_kernel<128, false>), dim3(tiles), dim3(256), smem, stream, p);
    }
    SPE_CHECK_LAUNCH();
    return 0;
}

extern "C" int spe_linear_small_bwd(const float* dy, const float* aux, int act, const void* x16, const void* WT16, float* dx, float* dW,
                                    float* db, int R, int N, int K, hipStream_t stream) {
    if (R <= 0 || N <= 0 || K <= 0) return 0;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if ((K & 7) || (N & 7) || !al16(dy) || !al16(aux) || !al16(x16) || !al16(WT16) || !al16(dx) || !al16(dW)) return -2;
    if ((dx && !WT16) || (dW && !x16) || (aux && act != 1 && act != 2)) return -2;
    LinSmallArgs p = {};
    p.x = dy; p.aux = aux; p.act = act; p.xs16 = reinterpret_cast<const unsigned short*>(x16);
    p.g.B = reinterpret_cast<const unsigned short*>(WT16); p.g.C = dx;
    p.dW = dW; p.db = db; p.R = R; p.N = N; p.K = K;
    const int tr = (R + LS_T - 1) / LS_T, tk = (K + LS_T - 1) / LS_T, tn = (N + LS_T - 1) / LS_T;
    p.tiles_dx = dx ? tr * tk : 0;
    const int tiles_dw = dW ? tn * tk : (db ? tn : 0);          // bias gradient only: the k-tile-0 workgroups
    if (p.tiles_dx + tiles_dw == 0) return 0;
    const bool wide = p.tiles_dx + tiles_dw <= ls_latency_tiles();
    auto smem_of = [](int kc, int rc) { const int a = 2 * LS_T * (kc + 8) * 2, b = 2 * rc * LS_LDT * 2 + 32 * 32 * 4; return a > b ? a : b; };
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_small_bwd_kernel<384, 512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           smem_of(384, 512));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (wide) hipLaunchKernelGGL((linear_small_bwd_kernel<384, 512>), dim3(p.tiles_dx + tiles_dw), dim3(256), smem_of(384, 512), stream, p);
    else hipLaunchKernelGGL((linear_small_bwd_kernel<128, 256>), dim3(p.tiles_dx + tiles_dw), dim3(256), smem_of(128, 256), stream, p);
    SPE_CHECK_LAUNCH();
    return 0;
}

// ---- group entries: nblk Linears of equal shape [N, K] on ONE input (the decoder's query-side projections: reference models/transformer.py:
// 368-372 sa_qcontent / sa_kcontent / sa_v of tgt, 369-371 + 399 sa_qpos / sa_kpos / ca_qpos of query_pos for every layer) - one launch each way
// for all of them, every Linear keeping its own weight copies, output and gradient buffers (pointer tables, nothing is stacked).
static bool ls_group_kc_wide(int N) { return N % 384 == 0; }
extern "C" int spe_linear_small_group_fwd(const float* x, long ldx, const void* const* W16, const void* const* W16lo, const float* const* bias,
                                          const float* const* add, float* const* y, void* x16_out, int R, int nblk, int N, int K, hipStream_t stream) {
    if (R <= 0 || N <= 0 || nblk <= 0) return 0;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (nblk > LS_MAXBLK || K <= 0 || (K & 7) || (N % LS_T) || (ldx & 3) || !al16(x) || !al16(x16_out)) return -2;
    LinSmallArgs p = {};
    const bool split = W16lo != nullptr;
    for (int i = 0; i < nblk; ++i) {
        if (!W16[i] || !y[i] || !al16(W16[i]) || !al16(y[i]) || (split && (!W16lo[i] || !al16(W16lo[i])))) return -2;
        p.Wb[i] = reinterpret_cast<const unsigned short*>(W16[i]); p.Wlob[i] = split ? reinterpret_cast<const unsigned short*>(W16lo[i]) : nullptr;
        p.biasb[i] = bias ? bias[i] : nullptr; p.yb[i] = y[i];
        p.addb[i] = add ? add[i] : nullptr;
        if (p.addb[i] && (!al16(p.addb[i]) || (N & 3))) return -2;
    }
    p.nblk = nblk; p.Nblk = N;
    p.g.M = R; p.g.N = nblk * N; p.g.K = K; p.g.ldc = N; p.g.alpha = 1.f; p.g.act = 0; p.g.splitk = 1;
    p.x = x; p.ldx = ldx; p.x16 = reinterpret_cast<unsigned short*>(x16_out);
    const int tiles = ((R + LS_T - 1) / LS_T) * (nblk * N / LS_T);
    const bool wide = tiles <= ls_latency_tiles();
    const int smem = (split ? 4 : 2) * LS_T * ((wide ? 384 : 128) + 8) * (int)sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_small_fwd_kernel<384, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           4 * LS_T * 392 * (int)sizeof(unsigned short));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (wide) {
        if (split) hipLaunchKernelGGL((linear_small_fwd_kernel<384, true>), dim3(tiles), dim3(256), smem, stream, p);
        else hipLaunchKernelGGL((linear_small_fwd_kernel<384, false>), dim3(tiles), dim3(256), smem, stream, p);
    } else {
        if (split) hipLaunchKernelGGL((linear_small_fwd_kernel<128, true>), dim3(tiles), dim3(256), smem, stream, p);
        else hipLaunchKernelGGL((linear_small_fwd_kernel<128, false>), dim3(tiles), dim3(256), smem, stream, p);
    }
    SPE_CHECK_LAUNCH();
    return 0;
}

// dx [R][K] = sum_i dy_i W_i, dW_i [N][K] = dy_i^T x, db_i [N] = colsum(dy_i); dy[i] NULL: that output received no gradient (its dW / db are
// not touched, it adds nothing to dx); dW / db NULL or with NULL elements: not wanted.  WT16[i]: bf16 W_i^T [K][N].  N % 128 == 0.
extern "C" int spe_linear_small_group_bwd(const float* const* dy, const void* x16, const void* const* WT16, float* dx, float* const* dW,
                                          float* const* db, int R, int nblk, int N, int K, hipStream_t stream) {
    if (R <= 0 || N <= 0 || K <= 0 || nblk <= 0) return 0;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (nblk > LS_MAXBLK || (K & 7) || (N % 128) || !al16(x16) || !al16(dx) || (dx && !WT16) || (dW && !x16)) return -2;
    LinSmallArgs p = {};
    bool any_dw = false;
    for (int i = 0; i < nblk; ++i) {
        p.dyb[i] = dy[i]; p.WTb[i] = WT16 ? reinterpret_cast<const unsigned short*>(WT16[i]) : nullptr;
        p.dWb[i] = dW ? dW[i] : nullptr; p.dbb[i] = db ? db[i] : nullptr;
        if (!al16(p.dyb[i]) || !al16(p.WTb[i]) || !al16(p.dWb[i]) || (dx && !p.WTb[i])) return -2;
        any_dw = any_dw || (p.dyb[i] && (p.dWb[i] || p.dbb[i]));
    }
    p.nblk = nblk; p.Nblk = N;
    p.xs16 = reinterpret_cast<const unsigned short*>(x16); p.g.C = dx;
    p.R = R; p.N = nblk * N; p.K = K;
    const int tr = (R + LS_T - 1) / LS_T, tk = (K + LS_T - 1) / LS_T, tn = nblk * N / LS_T;
    p.tiles_dx = dx ? tr * tk : 0;
    const int tiles_dw = any_dw ? tn * tk : 0;
    if (p.tiles_dx + tiles_dw == 0) return 0;
    const bool wide = ls_group_kc_wide(N) && p.tiles_dx + tiles_dw <= ls_latency_tiles();
    auto smem_of = [](int kc, int rc) { const int a = 2 * LS_T * (kc + 8) * 2, b = 2 * rc * LS_LDT * 2 + 32 * 32 * 4; return a > b ? a : b; };
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_small_bwd_kernel<384, 512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           smem_of(384, 512));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (wide) hipLaunchKernelGGL((linear_small_bwd_kernel<384, 512>), dim3(p.tiles_dx + tiles_dw), dim3(256), smem_of(384, 512), stream, p);
    else hipLaunchKernelGGL((linear_small_bwd_kernel<128, 256>), dim3(p.tiles_dx + tiles_dw), dim3(256), smem_of(128, 256), stream, p);
    SPE_CHECK_LAUNCH();
    return 0;
}
