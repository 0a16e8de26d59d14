// MFMA GEMM for every Linear / batched contraction on the SPE hot path (gfx950).
//
//   C[z][m][n] = act( alpha * sum_k opA(z)[m][k] * opB(z)[k][n] + bias[n] )
//
// fp32 tensors in HBM; operands are rounded to bf16 while they are staged into LDS and
// multiplied with v_mfma_f32_16x16x32_bf16 (fp32 accumulate).  `precision == 1` selects the
// 3-term split (hi*hi + hi*lo + lo*hi) which recovers ~fp32 accuracy at 3x the MFMA work; it is
// what the 1e-3 parity mode uses.  Replaces the implicit ATen GEMMs behind nn.Linear / torch.bmm
// in reference models/cait.py:374-393, models/transformer.py:355-427, models/attention.py:353,375.
//
// Block = 256 threads = 4 waves (2x2); block tile 128x128x32; wave tile 64x64 = 4x4 MFMA tiles.
// Register-staged double buffering: tile t+1 is in flight from HBM while tile t is multiplied.
#include "common.h"
#include <stdlib.h>

#define BK 32
#define LDSLD 40  // bf16 per LDS row: 32 + 8 pad -> 80 B rows keep ds_read_b128 16-B aligned

struct GemmArgs {
    const float* A; const float* B; float* C; float* C2; const float* bias;
    int M, N, K;
    long lda, ldb, ldc;
    int nb1;
    long sA0, sA1, sB0, sB1, sC0, sC1;
    float alpha;
    int act;       // 0 none, 1 relu, 2 gelu(erf)
    int splitk;    // >1: K split over workgroups (no bias/act): atomically added into pre-zeroed C, or, when
    long slab;     //     slab != 0, split z stores its partial tile into C + z*slab (reduced by the caller)
    int kt_per_split;
    int vecA, vecB;
    int xcd_bind;  // 0: plain tile order, 1: M-panels bound to XCDs, 2: N-panels bound to XCDs
    int a_bf16;    // A operand is stored as bf16 (the transposed score tensors of attn_fused.hip)
};

// ---- HBM -> registers -------------------------------------------------------------------
// All loads are UNCONDITIONAL on clamped addresses and masked afterwards: a branch around a load
// makes hipcc serialise it behind an s_waitcnt vmcnt(0) (one round trip per element).
// `vec` (wave-uniform): rows are 16-B aligned and padded to a multiple of 4 floats, so a float4
// at any 4-aligned in-row offset is inside the allocation even when it straddles the logical edge.
//
// "KC": the contraction index is the contiguous one: X(r,k) = base[r*ld + k].
// thread t loads rows r = (t>>3) + 32*i, k-quad (t&7).
template <int R, bool MASK>
__device__ __forceinline__ void load_kc(const float* __restrict__ base, long ld, int row0, int nrows,
                                        int k0, int kend, bool vec, float v[4][4]) {
    const int t = threadIdx.x;
    const int k = k0 + (t & 7) * 4;
    if (!MASK) {   // interior tile, 16-B aligned rows: no clamps, no selects (the staging VALU work bounds this kernel)
        const float* p0 = base + (long)(row0 + (t >> 3)) * ld + k;
#pragma unroll
        for (int i = 0; i < R / 32; ++i) {
            const float4 q = *reinterpret_cast<const float4*>(p0 + (long)(32 * i) * ld);
            v[i][0] = q.x; v[i][1] = q.y; v[i][2] = q.z; v[i][3] = q.w;
        }
        return;
    }
    const int kq = min(k, (kend - 1) & ~3);            // clamped quad start (>= 0 since kend >= 1)
    if (vec) {
#pragma unroll
        for (int i = 0; i < R / 32; ++i) {
            const int r = row0 + (t >> 3) + 32 * i;
            const float4 q = *reinterpret_cast<const float4*>(base + (long)min(r, nrows - 1) * ld + kq);
            const bool rv = r < nrows;
            v[i][0] = (rv && k + 0 < kend) ? q.x : 0.f; v[i][1] = (rv && k + 1 < kend) ? q.y : 0.f;
            v[i][2] = (rv && k + 2 < kend) ? q.z : 0.f; v[i][3] = (rv && k + 3 < kend) ? q.w : 0.f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < R / 32; ++i) {
            const int r = row0 + (t >> 3) + 32 * i;
            const float* p = base + (long)min(r, nrows - 1) * ld;
            float q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = p[min(k + j, kend - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = (r < nrows && k + j < kend) ? q[j] : 0.f;
        }
    }
}
// "RC": the row (m or n) index is contiguous: X(r,k) = base[k*ld + r].
// R = 128: thread t loads k = (t&7)*4 + i (i < 4), row-quad (t>>3); v[i][j] = X(row0 + 4*(t>>3) + j, k).
// R =  64: row-quad (t>>3)&15, k = (t&7)*4 + 2*(t>>7) + i (i < 2).
template <int R, bool MASK>
__device__ __forceinline__ void load_rc(const float* __restrict__ base, long ld, int row0, int nrows,
                                        int k0, int kend, bool vec, float v[4][4]) {
    const int t = threadIdx.x;
    constexpr int NI = R / 32;
    const int r = row0 + ((R == 128) ? (t >> 3) : ((t >> 3) & 15)) * 4;
    const int kb = k0 + (t & 7) * 4 + ((R == 128) ? 0 : 2 * (t >> 7));
    if (!MASK) {
        const float* p0 = base + (long)kb * ld + r;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float4 q = *reinterpret_cast<const float4*>(p0 + (long)i * ld);
            v[i][0] = q.x; v[i][1] = q.y; v[i][2] = q.z; v[i][3] = q.w;
        }
        return;
    }
    const int rq = min(r, (nrows - 1) & ~3);
    if (vec) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int k = kb + i;
            const float4 q = *reinterpret_cast<const float4*>(base + (long)min(k, kend - 1) * ld + rq);
            const bool kv = k < kend;
            v[i][0] = (kv && r + 0 < nrows) ? q.x : 0.f; v[i][1] = (kv && r + 1 < nrows) ? q.y : 0.f;
            v[i][2] = (kv && r + 2 < nrows) ? q.z : 0.f; v[i][3] = (kv && r + 3 < nrows) ? q.w : 0.f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int k = kb + i;
            const float* p = base + (long)min(k, kend - 1) * ld;
            float q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = p[min(r + j, nrows - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = (k < kend && r + j < nrows) ? q[j] : 0.f;
        }
    }
}

// bf16-stored A operand (same thread mapping; 8-B vector loads of 4 elements).
__device__ __forceinline__ float bfbits(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
template <int R, bool MASK>
__device__ __forceinline__ void load_kc_bf16(const unsigned short* __restrict__ base, long ld, int row0, int nrows,
                                             int k0, int kend, bool vec, float v[4][4]) {
    const int t = threadIdx.x;
    const int k = k0 + (t & 7) * 4;
    if (!MASK) {
        const unsigned short* p0 = base + (long)(row0 + (t >> 3)) * ld + k;
#pragma unroll
        for (int i = 0; i < R / 32; ++i) {
            const uint2 u = *reinterpret_cast<const uint2*>(p0 + (long)(32 * i) * ld);
            v[i][0] = __uint_as_float(u.x << 16); v[i][1] = __uint_as_float(u.x & 0xffff0000u);
            v[i][2] = __uint_as_float(u.y << 16); v[i][3] = __uint_as_float(u.y & 0xffff0000u);
        }
        return;
    }
    const int kq = min(k, (kend - 1) & ~3);
#pragma unroll
    for (int i = 0; i < R / 32; ++i) {
        const int r = row0 + (t >> 3) + 32 * i;
        const unsigned short* p = base + (long)min(r, nrows - 1) * ld;
        unsigned short q[4];
        if (vec) {
            const uint2 u = *reinterpret_cast<const uint2*>(p + kq);
            q[0] = u.x & 0xffff; q[1] = u.x >> 16; q[2] = u.y & 0xffff; q[3] = u.y >> 16;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = p[min(k + j, kend - 1)];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = (r < nrows && k + j < kend) ? bfbits(q[j]) : 0.f;
    }
}
template <int R, bool MASK>
__device__ __forceinline__ void load_rc_bf16(const unsigned short* __restrict__ base, long ld, int row0, int nrows,
                                             int k0, int kend, bool vec, float v[4][4]) {
    const int t = threadIdx.x;
    const int r = row0 + ((R == 128) ? (t >> 3) : ((t >> 3) & 15)) * 4;
    const int kb = k0 + (t & 7) * 4 + ((R == 128) ? 0 : 2 * (t >> 7));
    if (!MASK) {
        const unsigned short* p0 = base + (long)kb * ld + r;
#pragma unroll
        for (int i = 0; i < R / 32; ++i) {
            const uint2 u = *reinterpret_cast<const uint2*>(p0 + (long)i * ld);
            v[i][0] = __uint_as_float(u.x << 16); v[i][1] = __uint_as_float(u.x & 0xffff0000u);
            v[i][2] = __uint_as_float(u.y << 16); v[i][3] = __uint_as_float(u.y & 0xffff0000u);
        }
        return;
    }
    const int rq = min(r, (nrows - 1) & ~3);
#pragma unroll
    for (int i = 0; i < R / 32; ++i) {
        const int k = kb + i;
        const unsigned short* p = base + (long)min(k, kend - 1) * ld;
        unsigned short q[4];
        if (vec) {
            const uint2 u = *reinterpret_cast<const uint2*>(p + rq);
            q[0] = u.x & 0xffff; q[1] = u.x >> 16; q[2] = u.y & 0xffff; q[3] = u.y >> 16;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = p[min(r + j, nrows - 1)];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = (k < kend && r + j < nrows) ? bfbits(q[j]) : 0.f;
    }
}

// ---- registers -> LDS (bf16 via v_cvt_pk_bf16_f32, optionally hi/lo split) -------------------
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
template <bool SPLIT>
__device__ __forceinline__ void store4(unsigned short* hi, unsigned short* lo, int off,
                                       float a, float b, float c, float d) {
    bf16x4_t h;
    h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
    *reinterpret_cast<bf16x4_t*>(hi + off) = h;
    if (SPLIT) {
        bf16x4_t l;
        l[0] = (__bf16)(a - (float)h[0]); l[1] = (__bf16)(b - (float)h[1]);
        l[2] = (__bf16)(c - (float)h[2]); l[3] = (__bf16)(d - (float)h[3]);
        *reinterpret_cast<bf16x4_t*>(lo + off) = l;
    }
}
template <int R, bool SPLIT>
__device__ __forceinline__ void stage_kc(unsigned short* hi, unsigned short* lo, const float v[4][4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < R / 32; ++i)
        store4<SPLIT>(hi, lo, ((t >> 3) + 32 * i) * LDSLD + (t & 7) * 4, v[i][0], v[i][1], v[i][2], v[i][3]);
}
template <bool SPLIT>
__device__ __forceinline__ void store2(unsigned short* hi, unsigned short* lo, int off, float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t h;
    h[0] = (__bf16)a; h[1] = (__bf16)b;
    *reinterpret_cast<bf16x2_t*>(hi + off) = h;
    if (SPLIT) {
        bf16x2_t l;
        l[0] = (__bf16)(a - (float)h[0]); l[1] = (__bf16)(b - (float)h[1]);
        *reinterpret_cast<bf16x2_t*>(lo + off) = l;
    }
}
template <int R, bool SPLIT>
__device__ __forceinline__ void stage_rc(unsigned short* hi, unsigned short* lo, const float v[4][4]) {
    const int t = threadIdx.x;
    if (R == 128) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            store4<SPLIT>(hi, lo, ((t >> 3) * 4 + j) * LDSLD + (t & 7) * 4, v[0][j], v[1][j], v[2][j], v[3][j]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            store2<SPLIT>(hi, lo, (((t >> 3) & 15) * 4 + j) * LDSLD + (t & 7) * 4 + 2 * (t >> 7), v[0][j], v[1][j]);
    }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + spe_erff(x * 0.70710678118654752f)); }

// TA: opA(m,k) = A[k*lda+m] (else A[m*lda+k]).  TB: opB(k,n) = B[n*ldb+k] (else B[k*ldb+n]).
template <int T, bool TA, bool TB, bool SPLIT>
__global__ __launch_bounds__(256) void spe_gemm_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    constexpr int TILE = T * LDSLD;            // elements per operand tile
    constexpr int NF = T / 32;                 // 16x16 MFMA tiles per wave in each direction (wave tile = T/2)
    constexpr int WT = T / 2;
    constexpr int NPL = SPLIT ? 2 : 1;         // planes (hi, lo)
    // layout: [buf][A hi, A lo, B hi, B lo]
    auto sA = [&](int buf, int pl) { return smem + (buf * 2 * NPL + pl) * TILE; };
    auto sB = [&](int buf, int pl) { return smem + (buf * 2 * NPL + NPL + pl) * TILE; };

    // XCD-aware tile mapping: workgroup b runs on XCD b % 8 (8 private L2s).  The panels of the LARGER operand
    // are bound to XCDs (all tiles that read one such panel run on the same XCD), so that operand is fetched
    // into one L2 only; the smaller operand is re-fetched by each XCD.
    const int tiles_m = (p.M + T - 1) / T, tiles_n = (p.N + T - 1) / T;
    int tm, tn;
    if (p.xcd_bind == 0) { tm = blockIdx.x % tiles_m; tn = blockIdx.x / tiles_m; }
    else {
        const int no = (p.xcd_bind == 1) ? tiles_n : tiles_m;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int tb = xcd + 8 * (idx / no), to = idx % no;
        tm = (p.xcd_bind == 1) ? tb : to; tn = (p.xcd_bind == 1) ? to : tb;
        if (tm >= tiles_m || tn >= tiles_n) return;
    }
    const int zb = blockIdx.z / p.splitk, zs = blockIdx.z % p.splitk;
    const int b0 = zb / p.nb1, b1 = zb % p.nb1;
    const float* A = p.a_bf16 ? nullptr : p.A + b0 * p.sA0 + b1 * p.sA1;
    const unsigned short* A16 = p.a_bf16 ? reinterpret_cast<const unsigned short*>(p.A) + b0 * p.sA0 + b1 * p.sA1 : nullptr;
    const float* B = p.B + b0 * p.sB0 + b1 * p.sB1;
    float* C = p.C + b0 * p.sC0 + b1 * p.sC1 + (long)zs * p.slab;
    float* C2 = p.C2 ? p.C2 + b0 * p.sC0 + b1 * p.sC1 : nullptr;

    const int m0 = tm * T, n0 = tn * T;
    const int ktiles = (p.K + BK - 1) / BK;
    const int kt_begin = zs * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split; if (kt_end > ktiles) kt_end = ktiles;
    const int nt = kt_end - kt_begin;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int fr = lane & 15, fk = (lane >> 4) * 8;

    f32x4_t acc[NF][NF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    float va[4][4], vb[4][4];
    // interior fast path: the whole operand tile is in bounds and rows are vector-aligned -> unmasked loads
    const bool rowsA_in = p.vecA && (m0 + T <= p.M), rowsB_in = p.vecB && (n0 + T <= p.N);
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
        const bool k_in = k0 + BK <= p.K;
        if (rowsA_in && k_in) {
            if (p.a_bf16) { if (TA) load_rc_bf16<T, false>(A16, p.lda, m0, p.M, k0, p.K, true, va); else load_kc_bf16<T, false>(A16, p.lda, m0, p.M, k0, p.K, true, va); }
            else          { if (TA) load_rc<T, false>(A, p.lda, m0, p.M, k0, p.K, true, va); else load_kc<T, false>(A, p.lda, m0, p.M, k0, p.K, true, va); }
        } else {
            if (p.a_bf16) { if (TA) load_rc_bf16<T, true>(A16, p.lda, m0, p.M, k0, p.K, p.vecA, va); else load_kc_bf16<T, true>(A16, p.lda, m0, p.M, k0, p.K, p.vecA, va); }
            else          { if (TA) load_rc<T, true>(A, p.lda, m0, p.M, k0, p.K, p.vecA, va); else load_kc<T, true>(A, p.lda, m0, p.M, k0, p.K, p.vecA, va); }
        }
        if (rowsB_in && k_in) { if (TB) load_kc<T, false>(B, p.ldb, n0, p.N, k0, p.K, true, vb); else load_rc<T, false>(B, p.ldb, n0, p.N, k0, p.K, true, vb); }
        else                  { if (TB) load_kc<T, true>(B, p.ldb, n0, p.N, k0, p.K, p.vecB, vb); else load_rc<T, true>(B, p.ldb, n0, p.N, k0, p.K, p.vecB, vb); }
    };
    auto stage = [&](int buf) {
        if (TA) stage_rc<T, SPLIT>(sA(buf, 0), sA(buf, NPL - 1), va); else stage_kc<T, SPLIT>(sA(buf, 0), sA(buf, NPL - 1), va);
        if (TB) stage_kc<T, SPLIT>(sB(buf, 0), sB(buf, NPL - 1), vb); else stage_rc<T, SPLIT>(sB(buf, 0), sB(buf, NPL - 1), vb);
    };

    if (nt > 0) {
        gload(kt_begin);
        stage(0);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            if (t + 1 < nt) gload(kt_begin + t + 1);
            bf16x8_t a[NF], b[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i)
                a[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sA(buf, 0) + (wm * WT + i * 16 + fr) * LDSLD + fk));
#pragma unroll
            for (int j = 0; j < NF; ++j)
                b[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sB(buf, 0) + (wn * WT + j * 16 + fr) * LDSLD + fk));
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            if (SPLIT) {
                bf16x8_t al[NF], bl[NF];
#pragma unroll
                for (int i = 0; i < NF; ++i)
                    al[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sA(buf, 1) + (wm * WT + i * 16 + fr) * LDSLD + fk));
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    bl[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sB(buf, 1) + (wn * WT + j * 16 + fr) * LDSLD + fk));
#pragma unroll
                for (int i = 0; i < NF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], a[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], al[i], acc[i][j], 0, 0, 0);
                    }
            }
            if (t + 1 < nt) stage(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue.  The MFMAs were issued as (B-frag, A-frag), i.e. they produced C^T tiles, so
    // acc[i][j][r] = C[m0 + wm*T/2 + i*16 + (lane&15)][n0 + wn*T/2 + j*16 + (lane>>4)*4 + r]:
    // every lane owns 4 consecutive columns of one row -> one 16-B store per tile.
    const bool vst = ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                     (!C2 || (reinterpret_cast<uintptr_t>(C2) & 15) == 0);
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = n0 + wn * WT + j * 16 + (lane >> 4) * 4;
        if (n >= p.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && p.splitk == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = p.bias[min(n + r, p.N - 1)];
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int m = m0 + wm * WT + i * 16 + fr;
            if (m >= p.M) continue;
            const long off = (long)m * p.ldc + n;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha;
            if (p.splitk > 1) {
                if (p.slab != 0) {      // private slab of this split: plain (vector) stores, zeros if the split was empty
                    if (vst && n + 3 < p.N) *reinterpret_cast<float4*>(C + off) = make_float4(v[0], v[1], v[2], v[3]);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) C[off + r] = v[r];
                    }
                } else if (nt > 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) atomicAdd(C + off + r, v[r]);
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bv[r];
            const bool full = vst && (n + 3 < p.N);
            if (C2) {
                if (full) *reinterpret_cast<float4*>(C2 + off) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) C2[off + r] = v[r];
                }
            }
            if (p.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (p.act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            if (full) *reinterpret_cast<float4*>(C + off) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.N) C[off + r] = v[r];
            }
        }
    }
}

template <int T, bool TA, bool TB, bool SPLIT>
static int launch_gemm_t(const GemmArgs& p, int nbatch, hipStream_t stream) {
    constexpr int smem = 2 * 2 * (SPLIT ? 2 : 1) * T * LDSLD * (int)sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spe_gemm_kernel<T, TA, TB, SPLIT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles_m = (p.M + T - 1) / T, tiles_n = (p.N + T - 1) / T;
    GemmArgs q = p;
    // bind the operand with more bytes (same K: more rows) if it has enough panels to balance 8 XCDs
    q.xcd_bind = 0;
    if (p.M >= p.N && tiles_m >= 16) q.xcd_bind = 1;
    else if (p.N > p.M && tiles_n >= 16) q.xcd_bind = 2;
    else if (tiles_m >= 16) q.xcd_bind = 1;
    else if (tiles_n >= 16) q.xcd_bind = 2;
    int tiles = tiles_m * tiles_n;
    if (q.xcd_bind == 1) tiles = 8 * ((tiles_m + 7) / 8) * tiles_n;
    if (q.xcd_bind == 2) tiles = 8 * ((tiles_n + 7) / 8) * tiles_m;
    dim3 grid(tiles, 1, nbatch * p.splitk);
    hipLaunchKernelGGL((spe_gemm_kernel<T, TA, TB, SPLIT>), grid, dim3(256), smem, stream, q);
    SPE_CHECK_LAUNCH();
    return 0;
}

// Tile choice: every GEMM of this model is HBM/latency bound, so what matters is enough workgroups in flight.
// 128x128 tiles when they already give >= 1 workgroup per CU (split-K slices count), 64x64 tiles (4x the
// workgroups, half the registers, twice the L2 re-reads) otherwise and for skinny outputs (N <= 64: the per-head attention contractions).
extern "C" int spe_gemm_tile(int M, int N, int nbatch) {
    static int forced = -1;
    if (forced < 0) forced = SPE_KNOB("SPE_GEMM_TILE", 0);
    if (forced == 64 || forced == 128) return forced;
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * nbatch;
    return (t128 >= 256 && N > 64 && M > 64) ? 128 : 64;
}

template <bool TA, bool TB, bool SPLIT>
static int launch_gemm(const GemmArgs& p, int nbatch, hipStream_t stream) {
    if (spe_gemm_tile(p.M, p.N, nbatch * p.splitk) == 128) return launch_gemm_t<128, TA, TB, SPLIT>(p, nbatch, stream);
    return launch_gemm_t<64, TA, TB, SPLIT>(p, nbatch, stream);
}

extern "C" int spe_gemm_ex(const void* A, int a_bf16, const float* B, float* C, const float* bias, float* C2,
                           int M, int N, int K, long lda, long ldb, long ldc, int transA, int transB,
                           int batch0, int batch1, long sA0, long sA1, long sB0, long sB1, long sC0, long sC1,
                           float alpha, int act, int splitk, int precision, hipStream_t stream);

// C-ABI: see include/spe_hip.h (spe_gemm_f32).
extern "C" int spe_gemm_f32(const float* A, const float* B, float* C, const float* bias, float* C2,
                            int M, int N, int K, long lda, long ldb, long ldc, int transA, int transB,
                            int batch0, int batch1, long sA0, long sA1, long sB0, long sB1, long sC0, long sC1,
                            float alpha, int act, int splitk, int precision, hipStream_t stream) {
    return spe_gemm_ex(A, 0, B, C, bias, C2, M, N, K, lda, ldb, ldc, transA, transB, batch0, batch1, sA0, sA1, sB0, sB1, sC0, sC1,
                       alpha, act, splitk, precision, stream);
}

// Same contraction with the A operand optionally stored as bf16 (a_bf16 = 1; lda / sA* in elements).
extern "C" int spe_gemm_ex(const void* A, int a_bf16, const float* B, float* C, const float* bias, float* C2,
                           int M, int N, int K, long lda, long ldb, long ldc, int transA, int transB,
                           int batch0, int batch1, long sA0, long sA1, long sB0, long sB1, long sC0, long sC1,
                           float alpha, int act, int splitk, int precision, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || batch0 <= 0 || batch1 <= 0) return K <= 0 && M > 0 && N > 0 ? -4 : 0;
    if (transA && transB) return -2;  // not needed on this path
    GemmArgs p;
    p.A = reinterpret_cast<const float*>(A); p.a_bf16 = a_bf16; p.B = B; p.C = C; p.C2 = C2; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.nb1 = batch1; p.sA0 = sA0; p.sA1 = sA1; p.sB0 = sB0; p.sB1 = sB1; p.sC0 = sC0; p.sC1 = sC1;
    p.alpha = alpha; p.act = act; p.slab = 0;
    const int ktiles = (K + BK - 1) / BK;
    if (splitk < 0) {           // slab mode: C must hold |splitk| slabs (M*ldc floats; batched: the extent of all batches)
        splitk = -splitk;
        p.slab = (batch0 * batch1 == 1) ? (long)M * ldc
                                        : ((((long)(batch0 - 1) * sC0 + (long)(batch1 - 1) * sC1 + (long)(M - 1) * ldc + N) + 3) & ~3L);
        if (splitk > ktiles) return -5;
    }
    if (splitk < 1) splitk = 1;
    if (splitk > ktiles) splitk = ktiles > 0 ? ktiles : 1;
    p.kt_per_split = (ktiles + splitk - 1) / splitk;
    p.splitk = splitk;
    if (splitk > 1 && (act != 0 || C2 != nullptr || bias != nullptr)) return -3;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    auto m4 = [](long v) { return (v & 3) == 0; };
    // float4 loads need 16-B aligned rows; quads that straddle an edge fall back to scalars
    p.vecA = (a_bf16 ? ((reinterpret_cast<uintptr_t>(A) & 7) == 0) : al16(A)) && m4(lda) && m4(sA0) && m4(sA1);
    p.vecB = al16(B) && m4(ldb) && m4(sB0) && m4(sB1);
    const int nbatch = batch0 * batch1;
    const bool split = precision == 1;
    if (!transA && transB)  return split ? launch_gemm<false, true, true>(p, nbatch, stream) : launch_gemm<false, true, false>(p, nbatch, stream);
    if (!transA && !transB) return split ? launch_gemm<false, false, true>(p, nbatch, stream) : launch_gemm<false, false, false>(p, nbatch, stream);
    return split ? launch_gemm<true, false, true>(p, nbatch, stream) : launch_gemm<true, false, false>(p, nbatch, stream);
}
