// bf16 "TN" GEMM: the weight-gradient product of a Linear on ROW-MAJOR operands (gfx950),
//
//   C[m][n] = alpha * sum_r A16[r][m] * B16[r][n]            dW = dy^T x :  A = dy16 [R, N_out],  B = x16 [R, K_in]
//
// (reference: the autograd of nn.Linear at models/cait.py:376,390,409 and models/transformer.py:368-425).  The contraction
// runs over the ROWS of both operands, so neither is k-contiguous; spe_gemm_bf16nt needs transposed bf16 copies for this
// product (x16T written by every producer, dy16T by every backward) - this kernel does not: a 64-row slab of each operand
// is copied to LDS as it lies in memory (16-B chunks along the contiguous axis) and the MFMA operands are formed by the
// LDS transpose read of CDNA4, ds_read_b64_tr_b16: within a 16-lane group lane L hands in the address of 4 contiguous
// bf16 and receives element (L & 3) of the rows handed in by lanes (L >> 2), 4 + (L >> 2), 8 + (L >> 2), 12 + (L >> 2)
// (probed on the hardware).  With lane rho pointing at tile[k0 + (rho >> 2)][c0 + 4 * (rho & 3)] lane L gets
// tile[k0 + i][c0 + L], i = 0..3: four consecutive contraction indices of one column - two such reads are one
// v_mfma_f32_16x16x32_bf16 operand.  LDS rows are padded by 8 elements (2-way bank conflicts at worst).
//
// Block = 256 threads = 4 waves (2 x 2), tile BM x BN x 64 rows, register-staged double-buffered LDS like the NT kernel;
// split over the row range into private slabs (no atomics; the caller sums them with spe_colsum).
#include "common.h"

#define GT_BR 64

typedef unsigned int u32x4t_t __attribute__((ext_vector_type(4)));
typedef short s16x4t_t __attribute__((ext_vector_type(4)));
typedef short s16x8t_t __attribute__((ext_vector_type(8)));

struct GemmTNArgs {
    const unsigned short* A; const unsigned short* B; float* C;
    int M, N, R; long lda, ldb, ldc; float alpha;
    int splitk; long slab; int rt_per_split;
};

// thread t copies 16-B chunks: chunk (t % (W/8)) of rows (t / (W/8)) + (2048/W) * i of a 64-row slab that is W columns wide
template <int W>
__device__ __forceinline__ void gt_load(const unsigned short* __restrict__ base, long ld, int r0, int R, int c0, int C, u32x4t_t (&v)[W / 32]) {
    constexpr int CPR = W / 8, RPP = 256 / CPR;
    const int t = threadIdx.x, cc = c0 + (t % CPR) * 8;
    const bool cv = cc < C;
    const int ccl = cv ? cc : 0;
#pragma unroll
    for (int i = 0; i < W / 32; ++i) {
        const int r = r0 + t / CPR + RPP * i;
        const bool ok = cv && r < R;
        const u32x4t_t q = *reinterpret_cast<const u32x4t_t*>(base + (long)min(r, R - 1) * ld + ccl);
        v[i] = ok ? q : (u32x4t_t){0u, 0u, 0u, 0u};
    }
}
template <int W>
__device__ __forceinline__ void gt_stage(unsigned short* lds, const u32x4t_t (&v)[W / 32]) {
    constexpr int CPR = W / 8, RPP = 256 / CPR, LD = W + 8;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < W / 32; ++i)
        *reinterpret_cast<u32x4t_t*>(lds + (t / CPR + RPP * i) * LD + (t % CPR) * 8) = v[i];
}
// 8 consecutive contraction rows (32*ks + 8*(lane>>4) ..) of column c0 + (lane & 15) of an LDS slab with row stride LD
template <int LD>
__device__ __forceinline__ bf16x8_t gt_operand(const unsigned short* tile, int ks, int c0, int lane) {
    typedef __attribute__((address_space(3))) s16x4t_t lds_s4;
    const unsigned short* p = tile + (32 * ks + 8 * (lane >> 4) + ((lane & 15) >> 2)) * LD + c0 + 4 * (lane & 3);
    const s16x4t_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p));
    const s16x4t_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p + 4 * LD));
    const s16x8t_t v = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_bf16tn_kernel(GemmTNArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_tn[];
    constexpr int NFM = BM / 32, NFN = BN / 32, WM = BM / 2, WN = BN / 2;
    constexpr int LDA = BM + 8, LDB = BN + 8;
    constexpr int TA_ = GT_BR * LDA, TB_ = GT_BR * LDB;
    auto sA = [&](int buf) { return smem_tn + buf * (TA_ + TB_); };
    auto sB = [&](int buf) { return smem_tn + buf * (TA_ + TB_) + TA_; };
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int zs = blockIdx.z;
    float* C = p.C + (long)zs * p.slab;
    const int m0 = tm * BM, n0 = tn * BN;
    const int rtiles = (p.R + GT_BR - 1) / GT_BR;
    const int rt_begin = zs * p.rt_per_split;
    int rt_end = rt_begin + p.rt_per_split; if (rt_end > rtiles) rt_end = rtiles;
    const int nt = rt_end - rt_begin;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1, fr = lane & 15;

    f32x4_t acc[NFM][NFN];
#pragma unroll
    for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    u32x4t_t ca[BM / 32], cb[BN / 32], na[BM / 32], nb[BN / 32];
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) na[i] = (u32x4t_t){0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) nb[i] = (u32x4t_t){0u, 0u, 0u, 0u};
    if (nt > 0) {
        gt_load<BM>(p.A, p.lda, rt_begin * GT_BR, p.R, m0, p.M, ca);
        gt_load<BN>(p.B, p.ldb, rt_begin * GT_BR, p.R, n0, p.N, cb);
        gt_stage<BM>(sA(0), ca);
        gt_stage<BN>(sB(0), cb);
        if (nt > 1) {
            gt_load<BM>(p.A, p.lda, (rt_begin + 1) * GT_BR, p.R, m0, p.M, ca);
            gt_load<BN>(p.B, p.ldb, (rt_begin + 1) * GT_BR, p.R, n0, p.N, cb);
        }
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            if (t + 2 < nt) {
                gt_load<BM>(p.A, p.lda, (rt_begin + t + 2) * GT_BR, p.R, m0, p.M, na);
                gt_load<BN>(p.B, p.ldb, (rt_begin + t + 2) * GT_BR, p.R, n0, p.N, nb);
            }
#pragma unroll
            for (int ks = 0; ks < GT_BR / 32; ++ks) {
                bf16x8_t a[NFM], b[NFN];
#pragma unroll
                for (int i = 0; i < NFM; ++i) a[i] = gt_operand<LDA>(sA(buf), ks, wm * WM + i * 16, lane);
#pragma unroll
                for (int j = 0; j < NFN; ++j) b[j] = gt_operand<LDB>(sB(buf), ks, wn * WN + j * 16, lane);
#pragma unroll
                for (int i = 0; i < NFM; ++i)
#pragma unroll
                    for (int j = 0; j < NFN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
            if (t + 1 < nt) {
                gt_stage<BM>(sA(buf ^ 1), ca);
                gt_stage<BN>(sB(buf ^ 1), cb);
            }
#pragma unroll
            for (int i = 0; i < BM / 32; ++i) ca[i] = na[i];
#pragma unroll
            for (int i = 0; i < BN / 32; ++i) cb[i] = nb[i];
            __syncthreads();
        }
    }
    // MFMAs issued as (B-frag, A-frag): acc[i][j][r] = C[m0 + wm*WM + i*16 + (lane&15)][n0 + wn*WN + j*16 + (lane>>4)*4 + r]
    const bool vst = ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        const int n = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < NFM; ++i) {
            const int m = m0 + wm * WM + i * 16 + fr;
            if (m >= p.M) continue;
            const long off = (long)m * p.ldc + n;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha;
            if (vst && n + 3 < p.N) spe_store4_stream(C + off, v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.N) C[off + r] = v[r];
            }
        }
    }
}

template <int BM, int BN>
static int launch_tn(const GemmTNArgs& p, hipStream_t stream) {
    constexpr int smem = 2 * GT_BR * ((BM + 8) + (BN + 8)) * (int)sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16tn_kernel<BM, BN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_bf16tn_kernel<BM, BN>), dim3(tiles, 1, p.splitk), dim3(256), smem, stream, p);
    SPE_CHECK_LAUNCH();
    return 0;
}

// C-ABI: see include/spe_hip.h (spe_gemm_bf16tn).  -2: unsupported alignment (operands 16-B aligned, lda / ldb / M / N
// multiples of 8), -5: more splits than 64-row tiles.
extern "C" int spe_gemm_bf16tn(const void* A16, const void* B16, float* C, int M, int N, int R, long lda, long ldb, long ldc,
                               float alpha, int splitk, hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    if (R <= 0) return -4;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(A16) || !al16(B16) || (lda & 7) || (ldb & 7) || (M & 7) || (N & 7)) return -2;
    GemmTNArgs p;
    p.A = reinterpret_cast<const unsigned short*>(A16); p.B = reinterpret_cast<const unsigned short*>(B16); p.C = C;
    p.M = M; p.N = N; p.R = R; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.alpha = alpha; p.slab = 0;
    const int rtiles = (R + GT_BR - 1) / GT_BR;
    if (splitk < 0) { splitk = -splitk; p.slab = (long)M * ldc; if (splitk > rtiles) return -5; }
    else if (splitk > 1) return -2;
    if (splitk < 1) splitk = 1;
    p.splitk = splitk;
    p.rt_per_split = (rtiles + splitk - 1) / splitk;
    // decoder-size problems (a few hundred rows, no split): 64x64 tiles put 4x the workgroups on the chip (384 x 384: 36 instead of 9)
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * splitk;
    static const int small_max = SPE_KNOB("SPE_TN_SMALL_TILES", 256);      // 128-tiles x splits below this: 64 x 64 tiles
    // fewer than half the 512 resident workgroup slots with the wide tiles (a 384 x 384 weight gradient: 9 tiles x 16 splits):
    // 64 x 64 tiles, for which the caller sized the split (kernels.auto_splitk)
    if (t128 < small_max) return launch_tn<64, 64>(p, stream);
    if (M > 64 && N > 64) return launch_tn<128, 128>(p, stream);
    if (M > 64) return launch_tn<128, 64>(p, stream);
    return launch_tn<64, 64>(p, stream);
}
