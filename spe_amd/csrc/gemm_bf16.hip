// bf16-operand GEMM for the Linear layers of the SPE hot path (gfx950), benchmark ("bf16") precision mode:
//
//   C[m][n] = act( alpha * sum_k A16[m][k] * B16[n][k] + bias[n] )        ("NT": both operands k-contiguous)
//
// spe_gemm_f32 rounds its fp32 operands to bf16 while staging them, so feeding it the SAME values already rounded
// (spe_cvt_bf16, round-to-nearest-even) gives bit-identical products with half the bytes moved and - because a
// thread's staging registers now hold twice the elements - two K tiles in flight per workgroup instead of one.
// The M = 8300 layers of the backbone (12 K-iterations of ~0.2 us of MFMA against > 1 us of memory latency) are
// bound by bytes-in-flight / latency, which is what this kernel raises.  All three products of a Linear map to NT:
//   y  = x  W^T          A = x16   [R, K]      B = W16   [N, K]
//   dx = dy W            A = dy16  [R, N]      B = W16T  [K, N]        (transposed weight copy, once per step)
//   dW = dy^T x          A = dy16T [N, Rp]     B = x16T  [K, Rp]       (transposed activations, zero padded to Rp)
// Replaces the nn.Linear GEMMs of reference models/cait.py:376,390,409 and models/transformer.py:368-425.
//
// SPLIT operands (precision mode "bf16s", forward products): A = A_hi + A_lo and B = B_hi + B_lo, each part bf16 (lo = the
// bf16 rounding of the residual, written next to hi by every producer), and the product is evaluated as
//   A_hi B_hi + A_lo B_hi + A_hi B_lo          (3 MFMAs per tile step; the lo x lo term is below 2^-17 of the result)
// i.e. with ~16 significant bits per operand instead of 8: north_star's 1e-3 on logits / losses needs this in the FORWARD
// GEMMs (a single bf16 rounding of the Linear operands alone costs 2e-3 of pred_logits over 24 blocks, tools/error_budget.py),
// the matrix pipe has the room (these kernels are load / store bound), and the backward products stay single-term.
//
// Block = 256 threads = 4 waves (2x2); block tile BM x BN x 64 (BM, BN in {128, 64}); v_mfma_f32_16x16x32_bf16.
// Pipeline per workgroup: tile t is multiplied out of LDS buffer t&1 while tile t+1 waits in registers and tile
// t+2 is being fetched (16-B loads of 8 elements, no conversion work).
#include <cstdlib>
#include "common.h"
#include "gemm16_epilogue.h"
#include "det_reduce.h"

// thread t fetches 16-B chunk (t & 7) of rows (t >> 3) + 32*i ; rows are clamped (never stored), chunks past K zeroed
template <int R>
__device__ __forceinline__ void g16_load(const unsigned short* __restrict__ base, long ld, int row0, int nrows, int k0, int K,
                                         u32x4g_t (&v)[R / 32]) {
    const int t = threadIdx.x;
    const int k = k0 + (t & 7) * 8;
    const bool kv = k < K;
    const int kc = kv ? k : 0;
#pragma unroll
    for (int i = 0; i < R / 32; ++i) {
        const int r = min(row0 + (t >> 3) + 32 * i, nrows - 1);
        const u32x4g_t q = *reinterpret_cast<const u32x4g_t*>(base + (long)r * ld + kc);
        v[i] = kv ? q : (u32x4g_t){0u, 0u, 0u, 0u};
    }
}
template <int R>
__device__ __forceinline__ void g16_stage(unsigned short* lds, const u32x4g_t (&v)[R / 32]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < R / 32; ++i)
        *reinterpret_cast<u32x4g_t*>(lds + ((t >> 3) + 32 * i) * GB_LDR + (t & 7) * 8) = v[i];
}

// NTS > 0 ("small" variant, decoder-size problems: 400 x 384 x 384 is 42 workgroups of 6 K tiles, bound by the chain of
// load round trips of the pipelined loop - 10 us for 0.1 GFLOP): all NTS K tiles of the workgroup are requested at once
// (NTS * (BM + BN) / 32 16-B registers per thread), staged into NTS LDS slots behind ONE wait and barrier, then multiplied
// back to back.  K tiles past the contraction length load zeros.
// RING > 0 ("ring" variant, long contractions): the K tiles travel global -> LDS by global_load_lds_dwordx4 (no staging
// registers, nothing for hipcc to drain at the loop head) into a ring of RING stages; RING - 1 tiles are in flight per
// workgroup and a counted s_waitcnt vmcnt admits the oldest.  The pipelined loop below keeps ONE tile in flight beyond the one
// waiting in registers, so an iteration lasts (memory latency) / 2 ~ 1 us when its MFMAs need 0.1 us (fc2 forward, K = 1536:
// 24 iterations = 20 of its 29 us).  A DMA instruction deposits lane l's 16 B at (base + 16 l): 8 rows x 128 B per wave
// instruction, rows unpadded - so the 16-B chunks of a row are XOR-swizzled by the row index through the choice of the GLOBAL
// chunk each lane fetches (LDS slot (r, c) holds chunk c ^ (r & 7)), which leaves the operand reads 2-way conflicted at worst.
// Needs K % 64 == 0 (every model dimension is).
__device__ __forceinline__ void gb_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int BM, int BN, bool EX, int NTS = 0, int RING = 0, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_bf16nt_kernel(Gemm16Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    static_assert(!SPLIT || (NTS == 0 && RING == 0), "split operands run on the register-pipelined loop");
    constexpr int NFM = BM / 32, NFN = BN / 32;        // 16x16 MFMA tiles per wave (wave tile BM/2 x BN/2)
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TA_ = BM * GB_LDR, TB_ = BN * GB_LDR;
    constexpr int BUF_ = (TA_ + TB_) * (SPLIT ? 2 : 1);  // one pipeline stage: [A hi][B hi]([A lo][B lo])
    auto sA = [&](int buf) { return smem16 + buf * BUF_; };
    auto sB = [&](int buf) { return smem16 + buf * BUF_ + TA_; };
    auto sAl = [&](int buf) { return smem16 + buf * BUF_ + TA_ + TB_; };
    auto sBl = [&](int buf) { return smem16 + buf * BUF_ + 2 * TA_ + TB_; };

    // workgroup b runs on XCD b % 8: the panels of the operand with more rows are bound to XCDs (all tiles reading
    // one such panel run on the same XCD), so that operand is fetched into one L2 only; the other one is re-fetched
    // per XCD.  Same mapping as spe_gemm_kernel.
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int tm, tn;
    if (p.xcd_bind == 0) { tm = blockIdx.x % tiles_m; tn = blockIdx.x / tiles_m; }
    else {
        const int no = (p.xcd_bind == 1) ? tiles_n : tiles_m;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int tb = xcd + 8 * (idx / no), to = idx % no;
        tm = (p.xcd_bind == 1) ? tb : to; tn = (p.xcd_bind == 1) ? to : tb;
        if (tm >= tiles_m || tn >= tiles_n) return;
    }
    const int zs = blockIdx.z;
    float* C = p.C + (long)zs * p.slab;
    const int m0 = tm * BM, n0 = tn * BN;
    const int ktiles = (p.K + GB_BK - 1) / GB_BK;
    const int kt_begin = zs * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split; if (kt_end > ktiles) kt_end = ktiles;
    const int nt = kt_end - kt_begin;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int fr = lane & 15, fk = (lane >> 4) * 8;

    f32x4_t acc[NFM][NFN];
#pragma unroll
    for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    if constexpr (RING > 0) {
        constexpr int STG = (BM + BN) * 64;               // bf16 per stage: rows of 64 elements = 128 B, unpadded
        constexpr int PW = (BM + BN) / 32;                // 1-KB pieces (8 rows) per wave and stage
        const int ws = __builtin_amdgcn_readfirstlane(w);
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)smem16);
        auto issue = [&](int t, int slot) {
            const int tc = min(t, nt - 1);                // past the range: a valid tile, never used (keeps the vmcnt arithmetic static)
            const long k0 = (long)(kt_begin + tc) * GB_BK;
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                const int piece = ws + 4 * i;             // [0, BM/8): A rows ; [BM/8, (BM+BN)/8): B rows
                const bool isA = piece < BM / 8;
                const int r8 = (isA ? piece : piece - BM / 8) * 8 + (lane >> 3);
                const int c = (lane & 7) ^ (r8 & 7);
                const unsigned short* src = isA ? p.A + (long)min(m0 + r8, p.M - 1) * p.lda + k0 + c * 8
                                                : p.B + (long)min(n0 + r8, p.N - 1) * p.ldb + k0 + c * 8;
                gb_glds16(src, lds0 + (unsigned)((slot * STG + piece * 512) * 2));
            }
        };
#pragma unroll
        for (int st = 0; st < RING - 1; ++st) issue(st, st);
        for (int t = 0; t < nt; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((RING - 2) * PW) : "memory");     // this wave's pieces of tile t have landed
            __syncthreads();                              // everybody's have, and everybody is done reading tile t - 1
            issue(t + RING - 1, (t + RING - 1) % RING);   // refill the slot of tile t - 1
            const unsigned short* tA = smem16 + (t % RING) * STG;
            const unsigned short* tB = tA + BM * 64;
#pragma unroll
            for (int ks = 0; ks < GB_BK / 32; ++ks) {
                const int kc = ks * 4 + (lane >> 4);
                bf16x8_t a[NFM], b[NFN];
#pragma unroll
                for (int i = 0; i < NFM; ++i) {
                    const int row = wm * WM + i * 16 + fr;
                    a[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(tA + row * 64 + ((kc ^ (row & 7)) * 8)));
                }
#pragma unroll
                for (int j = 0; j < NFN; ++j) {
                    const int row = wn * WN + j * 16 + fr;
                    b[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(tB + row * 64 + ((kc ^ (row & 7)) * 8)));
                }
#pragma unroll
                for (int i = 0; i < NFM; ++i)
#pragma unroll
                    for (int j = 0; j < NFN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the ring before the epilogue reuses the buffers
        __syncthreads();
    }
    if constexpr (NTS > 0) {
        u32x4g_t ra[NTS][BM / 32], rb[NTS][BN / 32];
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
            g16_load<BM>(p.A, p.lda, m0, p.M, t * GB_BK, p.K, ra[t]);
            g16_load<BN>(p.B, p.ldb, n0, p.N, t * GB_BK, p.K, rb[t]);
        }
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
            g16_stage<BM>(sA(0) + t * (TA_ + TB_), ra[t]);
            g16_stage<BN>(sA(0) + t * (TA_ + TB_) + TA_, rb[t]);
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
            const unsigned short* tA = sA(0) + t * (TA_ + TB_);
            const unsigned short* tB = tA + TA_;
#pragma unroll
            for (int ks = 0; ks < GB_BK / 32; ++ks) {
                bf16x8_t a[NFM], b[NFN];
#pragma unroll
                for (int i = 0; i < NFM; ++i)
                    a[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(tA + (wm * WM + i * 16 + fr) * GB_LDR + ks * 32 + fk));
#pragma unroll
                for (int j = 0; j < NFN; ++j)
                    b[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(tB + (wn * WN + j * 16 + fr) * GB_LDR + ks * 32 + fk));
#pragma unroll
                for (int i = 0; i < NFM; ++i)
#pragma unroll
                    for (int j = 0; j < NFN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
    }
    u32x4g_t ca[BM / 32], cb[BN / 32];     // tile t+1 (landed or landing)
    u32x4g_t na[BM / 32], nb[BN / 32];     // tile t+2 (being fetched)
    constexpr int SA_ = SPLIT ? BM / 32 : 1, SB_ = SPLIT ? BN / 32 : 1;
    u32x4g_t cal[SA_], cbl[SB_], nal[SA_], nbl[SB_];       // the low parts of the same tiles (SPLIT)
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) na[i] = (u32x4g_t){0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) nb[i] = (u32x4g_t){0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < SA_; ++i) nal[i] = (u32x4g_t){0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < SB_; ++i) nbl[i] = (u32x4g_t){0u, 0u, 0u, 0u};
    if (NTS == 0 && RING == 0 && nt > 0) {
        g16_load<BM>(p.A, p.lda, m0, p.M, kt_begin * GB_BK, p.K, ca);
        g16_load<BN>(p.B, p.ldb, n0, p.N, kt_begin * GB_BK, p.K, cb);
        if constexpr (SPLIT) {
            g16_load<BM>(p.Alo, p.lda, m0, p.M, kt_begin * GB_BK, p.K, cal);
            g16_load<BN>(p.Blo, p.ldb, n0, p.N, kt_begin * GB_BK, p.K, cbl);
        }
        g16_stage<BM>(sA(0), ca);
        g16_stage<BN>(sB(0), cb);
        if constexpr (SPLIT) { g16_stage<BM>(sAl(0), cal); g16_stage<BN>(sBl(0), cbl); }
        if (nt > 1) {
            g16_load<BM>(p.A, p.lda, m0, p.M, (kt_begin + 1) * GB_BK, p.K, ca);
            g16_load<BN>(p.B, p.ldb, n0, p.N, (kt_begin + 1) * GB_BK, p.K, cb);
            if constexpr (SPLIT) {
                g16_load<BM>(p.Alo, p.lda, m0, p.M, (kt_begin + 1) * GB_BK, p.K, cal);
                g16_load<BN>(p.Blo, p.ldb, n0, p.N, (kt_begin + 1) * GB_BK, p.K, cbl);
            }
        }
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            if (t + 2 < nt) {
                g16_load<BM>(p.A, p.lda, m0, p.M, (kt_begin + t + 2) * GB_BK, p.K, na);
                g16_load<BN>(p.B, p.ldb, n0, p.N, (kt_begin + t + 2) * GB_BK, p.K, nb);
                if constexpr (SPLIT) {
                    g16_load<BM>(p.Alo, p.lda, m0, p.M, (kt_begin + t + 2) * GB_BK, p.K, nal);
                    g16_load<BN>(p.Blo, p.ldb, n0, p.N, (kt_begin + t + 2) * GB_BK, p.K, nbl);
                }
            }
#pragma unroll
            for (int ks = 0; ks < GB_BK / 32; ++ks) {
                bf16x8_t a[NFM], b[NFN], al[SPLIT ? NFM : 1], bl[SPLIT ? NFN : 1];
#pragma unroll
                for (int i = 0; i < NFM; ++i) {
                    a[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sA(buf) + (wm * WM + i * 16 + fr) * GB_LDR + ks * 32 + fk));
                    if constexpr (SPLIT) al[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sAl(buf) + (wm * WM + i * 16 + fr) * GB_LDR + ks * 32 + fk));
                }
#pragma unroll
                for (int j = 0; j < NFN; ++j) {
                    b[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sB(buf) + (wn * WN + j * 16 + fr) * GB_LDR + ks * 32 + fk));
                    if constexpr (SPLIT) bl[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sBl(buf) + (wn * WN + j * 16 + fr) * GB_LDR + ks * 32 + fk));
                }
#pragma unroll
                for (int i = 0; i < NFM; ++i)
#pragma unroll
                    for (int j = 0; j < NFN; ++j) {
                        if constexpr (SPLIT) {        // the two cross terms first, then hi x hi (each accumulator: one MFMA shape)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], a[i], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], al[i], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                    }
            }
            if (t + 1 < nt) {
                g16_stage<BM>(sA(buf ^ 1), ca);
                g16_stage<BN>(sB(buf ^ 1), cb);
                if constexpr (SPLIT) { g16_stage<BM>(sAl(buf ^ 1), cal); g16_stage<BN>(sBl(buf ^ 1), cbl); }
            }
#pragma unroll
            for (int i = 0; i < BM / 32; ++i) ca[i] = na[i];
#pragma unroll
            for (int i = 0; i < BN / 32; ++i) cb[i] = nb[i];
            if constexpr (SPLIT) {
#pragma unroll
                for (int i = 0; i < SA_; ++i) cal[i] = nal[i];
#pragma unroll
                for (int i = 0; i < SB_; ++i) cbl[i] = nbl[i];
            }
            __syncthreads();
        }
    }

    if constexpr (EX) {
        static_assert((BM * (BN + 8) + BN * (BM + 8)) <= 2 * (BM + BN) * GB_LDR, "staged tiles must fit the operand buffers");
        static_assert(!SPLIT || (2 * BM * (BN + 8) + BN * (BM + 8)) <= 4 * (BM + BN) * GB_LDR, "staged tiles must fit the operand buffers");
        gemm16_epilogue_ex<BM, BN, SPLIT>(p, acc, smem16, m0, n0);
        return;
    }
    gemm16_epilogue_plain<BM, BN>(p, acc, C, m0, n0);
}

// gemm_nt2.hip: the 128-row LDS-DMA kernels for activation-sized problems; SPE_NT2_NA = not covered, run the kernels of this file
#define SPE_NT2_NA (-100)
int spe_nt2_dispatch(const Gemm16Args& p, bool ex, hipStream_t stream);

template <int BM, int BN, bool EX = false, int NTS = 0, int RING = 0, bool SPLIT = false>
static int launch_gemm16(const Gemm16Args& p, hipStream_t stream) {
    constexpr int smem = RING > 0 ? RING * (BM + BN) * 64 * (int)sizeof(unsigned short)
                                  : (NTS > 0 ? NTS : 2) * (SPLIT ? 2 : 1) * (BM + BN) * GB_LDR * (int)sizeof(unsigned short);
    static_assert(RING == 0 || !EX || RING * 64 >= 2 * GB_LDR, "the staged epilogue tiles must fit the ring");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16nt_kernel<BM, BN, EX, NTS, RING, SPLIT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    Gemm16Args q = p;
    q.xcd_bind = 0;
    if (p.M >= p.N && tiles_m >= 16) q.xcd_bind = 1;
    else if (p.N > p.M && tiles_n >= 16) q.xcd_bind = 2;
    else if (tiles_m >= 16) q.xcd_bind = 1;
    else if (tiles_n >= 16) q.xcd_bind = 2;
    int tiles = tiles_m * tiles_n;
    if (q.xcd_bind == 1) tiles = 8 * ((tiles_m + 7) / 8) * tiles_n;
    if (q.xcd_bind == 2) tiles = 8 * ((tiles_n + 7) / 8) * tiles_m;
    dim3 grid(tiles, 1, p.splitk);
    hipLaunchKernelGGL((gemm_bf16nt_kernel<BM, BN, EX, NTS, RING, SPLIT>), grid, dim3(256), smem, stream, q);
    SPE_CHECK_LAUNCH();
    return 0;
}

// C-ABI: see include/spe_hip.h (spe_gemm_bf16nt).  -2: unsupported alignment, -3: bias/act with split-K,
// -5: more splits than K tiles.
extern "C" int spe_gemm_bf16nt(const void* A16, const void* B16, const void* A16lo, const void* B16lo, float* C, const float* bias,
                               float* C2, int M, int N, int K, long lda, long ldb, long ldc, float alpha, int act, int splitk,
                               hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0) return -4;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(A16) || !al16(B16) || !al16(A16lo) || !al16(B16lo) || (lda & 7) || (ldb & 7) || (K & 7)) return -2;
    if ((A16lo != nullptr) != (B16lo != nullptr)) return -2;
    Gemm16Args p;
    p.A = reinterpret_cast<const unsigned short*>(A16); p.B = reinterpret_cast<const unsigned short*>(B16);
    p.Alo = reinterpret_cast<const unsigned short*>(A16lo); p.Blo = reinterpret_cast<const unsigned short*>(B16lo); p.out16lo = nullptr;
    p.C = C; p.C2 = C2; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    // act bits 8, 9: IEEE fp16 operands (single-term, v_mfma_f32_16x16x32_f16) / IEEE fp16 output C [M][ldc] - the 128-wide LDS-DMA
    // kernels only (gemm_nt2.hip); -2 when the problem is outside their domain
    p.h16 = (act >> 8) & 3; act &= 0xff;
    p.alpha = alpha; p.act = act; p.slab = 0;
    p.ws = DetWs{nullptr, nullptr, 0, 0}; p.half_flags = 0;
    p.out16 = nullptr; p.out16T = nullptr; p.colsum = nullptr; p.aux = nullptr; p.ld16 = 0; p.ld16t = 0; p.res = nullptr; p.rgamma = nullptr;
    p.drop_p = 0.f; p.drop_seed = p.drop_off = 0; p.sscale = nullptr; p.rps = 1;
    const int ktiles = (K + GB_BK - 1) / GB_BK;
    if (p.h16 && (splitk != 1 || C2 || (ldc & 3))) return -2;
    if ((p.h16 & 1) && A16lo) return -2;
    if (splitk < 0) {           // slab mode: C holds |splitk| slabs of M*ldc floats
        splitk = -splitk; p.slab = (long)M * ldc;
        if (splitk > ktiles) return -5;
    } else if (splitk > 1) return -2;     // no atomic mode here
    if (splitk < 1) splitk = 1;
    p.kt_per_split = (ktiles + splitk - 1) / splitk;
    p.splitk = splitk;
    if (splitk > 1 && (act != 0 || C2 != nullptr || bias != nullptr)) return -3;
    if (p.Alo && splitk != 1) return -3;
    { const int rc = spe_nt2_dispatch(p, false, stream); if (rc != SPE_NT2_NA) return rc; }
    if (p.h16 & 1) return -2;
    if (p.Alo) {            // split operands: 64x64 tiles at two workgroups per CU (73 KB of LDS each); 128x64 when that fills the chip less
        if (splitk != 1) return -3;
        return launch_gemm16<64, 64, false, 0, 0, true>(p, stream);
    }
    {   // developer knob (tools/bench_gemm.py): SPE_GEMM16_TILE = 1 / 2 / 3 forces 128x128 / 128x64 / 64x64
        static const int forced = SPE_KNOB("SPE_GEMM16_TILE", 0);
        if (forced == 1) return launch_gemm16<128, 128>(p, stream);
        if (forced == 2) return launch_gemm16<128, 64>(p, stream);
        if (forced == 3) return launch_gemm16<64, 64>(p, stream);
    }
    // Activation-sized products (many rows, one pass over a short or medium contraction): 64x64 tiles.  Measured inside the
    // training step they beat the wider tiles by 10-30 % (qkv forward 39 -> 27 us, fc1 + GELU 70 -> 51 us): a workgroup's
    // epilogue (bias / activation / up to three output tensors) is as long as its main loop here, 2000-3000 small
    // workgroups at 4 per CU overlap one's stores with another's loads, and the 1.1-1.5 rounds that 585 / 780 wide tiles
    // make on 512 slots disappear.  The weight-gradient products (few output tiles, split-K) keep the wide tiles.
    if (splitk == 1 && M >= 2048) {
        // long contraction into a narrow output (fc2 forward, fc1 / qkv input gradients, the stacked decoder projections'
        // input gradient): 128x64 tiles fed by the LDS-DMA ring - half the A-panel re-reads of the 64x64 tiles and 2 tiles
        // in flight per workgroup (K = 1536: 30.5 -> 25.6 us, K = 1152: 23.3 -> 20.7, K = 4608: 85 -> 64; with more stages
        // or on the K = 384 products the lost occupancy costs more than the ring brings).  SPE_GEMM16_RING=0 disables it (A/B).
        static const int ring = SPE_KNOB("SPE_GEMM16_RING", 1);
        if (ring > 1 && (K % GB_BK) == 0) {          // developer knob: force a ring configuration for every activation-sized product
            if (ring == 1282) return launch_gemm16<128, 128, false, 0, 2>(p, stream);
            if (ring == 1283) return launch_gemm16<128, 128, false, 0, 3>(p, stream);
            if (ring == 642) return launch_gemm16<128, 64, false, 0, 2>(p, stream);
            if (ring == 643) return launch_gemm16<128, 64, false, 0, 3>(p, stream);
        }
        if (ring && (K % GB_BK) == 0 && K >= 1024 && N <= 512) return launch_gemm16<128, 64, false, 0, 3>(p, stream);
        return launch_gemm16<64, 64>(p, stream);
    }
    // tile: 128x128 when that already fills the chip, else narrower tiles (more workgroups in flight)
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * splitk;
    if (t128 >= 384 && N > 64) return launch_gemm16<128, 128>(p, stream);
    const long t64n = (long)((M + 127) / 128) * ((N + 63) / 64) * splitk;
    if (t64n >= 256 && M > 64) return launch_gemm16<128, 64>(p, stream);
    // decoder-size problems (few workgroups, short contraction): every K tile in flight at once
    if (splitk == 1 && ktiles <= 6) return launch_gemm16<64, 64, false, 6>(p, stream);
    if (splitk == 1 && ktiles == 7) return launch_gemm16<64, 64, false, 7>(p, stream);
    return launch_gemm16<64, 64>(p, stream);
}

// C-ABI: see include/spe_hip.h (spe_gemm_bf16nt_ex).  -2: unsupported alignment / leading dimensions.
static int gemm_bf16nt_ex_impl(const void* A16, const void* B16, const void* A16lo, const void* B16lo, float* C, const float* bias,
                               float* C2, void* out16, void* out16lo, long ld16, void* out16T, long ld16t, float* colsum,
                               const float* aux, const float* res, const float* rgamma,
                               int M, int N, int K, long lda, long ldb, long ldc, float alpha, int act, int half_flags,
                               float p_drop, uint64_t seed, uint64_t offset, const float* sample_scale, long rows_per_sample, hipStream_t stream);
extern "C" int spe_gemm_bf16nt_ex(const void* A16, const void* B16, const void* A16lo, const void* B16lo, float* C, const float* bias,
                                  float* C2, void* out16, void* out16lo, long ld16, void* out16T, long ld16t, float* colsum,
                                  const float* aux, const float* res, const float* rgamma,
                                  int M, int N, int K, long lda, long ldb, long ldc, float alpha, int act, int half_flags, hipStream_t stream) {
    return gemm_bf16nt_ex_impl(A16, B16, A16lo, B16lo, C, bias, C2, out16, out16lo, ld16, out16T, ld16t, colsum, aux, res, rgamma, M, N, K, lda, ldb,
                               ldc, alpha, act, half_flags, 0.f, 0, 0, nullptr, 1, stream);
}
// C-ABI: see include/spe_hip.h.  spe_gemm_bf16nt_ex with dropout after the activation (or its derivative) and a per-sample scale on the residual.
extern "C" int spe_gemm_bf16nt_exd(const void* A16, const void* B16, const void* A16lo, const void* B16lo, float* C, const float* bias,
                                   float* C2, void* out16, void* out16lo, long ld16, void* out16T, long ld16t, float* colsum,
                                   const float* aux, const float* res, const float* rgamma,
                                   int M, int N, int K, long lda, long ldb, long ldc, float alpha, int act, int half_flags,
                                   float p_drop, uint64_t seed, uint64_t offset, const float* sample_scale, long rows_per_sample, hipStream_t stream) {
    if (p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && (N & 3)) || (sample_scale && (!res || rows_per_sample <= 0))) return -2;
    return gemm_bf16nt_ex_impl(A16, B16, A16lo, B16lo, C, bias, C2, out16, out16lo, ld16, out16T, ld16t, colsum, aux, res, rgamma, M, N, K, lda, ldb,
                               ldc, alpha, act, half_flags, p_drop, seed, offset, sample_scale, rows_per_sample, stream);
}
static int gemm_bf16nt_ex_impl(const void* A16, const void* B16, const void* A16lo, const void* B16lo, float* C, const float* bias,
                               float* C2, void* out16, void* out16lo, long ld16, void* out16T, long ld16t, float* colsum,
                               const float* aux, const float* res, const float* rgamma,
                               int M, int N, int K, long lda, long ldb, long ldc, float alpha, int act, int half_flags,
                               float p_drop, uint64_t seed, uint64_t offset, const float* sample_scale, long rows_per_sample, hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0) return -4;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(A16) || !al16(B16) || !al16(A16lo) || !al16(B16lo) || (lda & 7) || (ldb & 7) || (K & 7)) return -2;
    // half_flags bit 2: A16 / B16 hold IEEE fp16 (single-term product; no low parts) ; bit 3: out16lo receives IEEE fp16(v) (no low parts needed)
    const bool op_f16 = (half_flags & 4) != 0, lo_f16 = (half_flags & 8) != 0;
    if ((A16lo != nullptr) != (B16lo != nullptr) || (out16lo && !A16lo && !lo_f16) || (op_f16 && A16lo)) return -2;
    if (out16T && ld16t < M) return -2;
    if (aux && act != 1 && act != 2) return -2;
    if ((res != nullptr) != (rgamma != nullptr) || (res && (!C || act != 0 || aux))) return -2;
    Gemm16Args p;
    p.h16 = 0; p.drop_p = p_drop; p.drop_seed = seed; p.drop_off = offset; p.sscale = sample_scale; p.rps = sample_scale ? rows_per_sample : 1;
    p.A = reinterpret_cast<const unsigned short*>(A16); p.B = reinterpret_cast<const unsigned short*>(B16);
    p.C = C; p.C2 = C2; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.act = act; p.slab = 0; p.splitk = 1;
    p.Alo = reinterpret_cast<const unsigned short*>(A16lo); p.Blo = reinterpret_cast<const unsigned short*>(B16lo);
    p.out16lo = reinterpret_cast<unsigned short*>(out16lo);
    p.kt_per_split = (K + GB_BK - 1) / GB_BK;
    p.out16 = reinterpret_cast<unsigned short*>(out16); p.ld16 = ld16;
    p.out16T = reinterpret_cast<unsigned short*>(out16T); p.ld16t = ld16t;
    p.colsum = colsum; p.aux = aux; p.res = res; p.rgamma = rgamma;
    if (half_flags & ~15) return -2;
    p.half_flags = half_flags & 3;
    p.h16 = (op_f16 ? 1 : 0) | (lo_f16 ? 4 : 0);
    p.ws = spe_detws();
    if (colsum) DET_CHECK(p.ws, (N + 63) / 64, (M + 63) / 64, 64);      // bound for the smallest tiles
    // the transposed copy's zero columns M..ld16t-1 are written by the last row tile: it must reach ld16t
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const bool reach128 = !out16T || ld16t <= (long)((M + 127) / 128) * 128;
    const bool reach64 = !out16T || ld16t <= (long)((M + 63) / 64) * 64;
    if (!reach128 && !reach64) return -2;
    { const int rc = spe_nt2_dispatch(p, true, stream); if (rc != SPE_NT2_NA) return rc; }
    if (p.h16) return -2;                    // fp16 operands / fp16 second copy: the LDS-DMA kernels only
    if (p.Alo) return reach64 ? launch_gemm16<64, 64, true, 0, 0, true>(p, stream) : -2;
    {   // developer knob: SPE_GEMM16_TILE also applies here
        static const int forced = SPE_KNOB("SPE_GEMM16_TILE", 0);
        if (forced == 1 && reach128) return launch_gemm16<128, 128, true>(p, stream);
        if (forced == 2 && reach128) return launch_gemm16<128, 64, true>(p, stream);
        if (forced == 3 && reach64) return launch_gemm16<64, 64, true>(p, stream);
    }
    {   // see spe_gemm_bf16nt
        static const int ring = SPE_KNOB("SPE_GEMM16_RING", 1);
        if (ring && M >= 2048 && reach128 && (K % GB_BK) == 0 && K >= 1024 && N <= 512) return launch_gemm16<128, 64, true, 0, 3>(p, stream);
    }
    if (M >= 2048 && reach64) return launch_gemm16<64, 64, true>(p, stream);
    if (reach128 && t128 >= 384 && N > 64) return launch_gemm16<128, 128, true>(p, stream);
    const long t64n = (long)((M + 127) / 128) * ((N + 63) / 64);
    if (reach128 && ((t64n >= 256 && M > 64) || !reach64)) return launch_gemm16<128, 64, true>(p, stream);
    return launch_gemm16<64, 64, true>(p, stream);
}

// ---- fp32 -> bf16 (round to nearest even) copies of a [R, C] matrix: out[R][ldo] (row-major) and/or the
// transpose outT[C][ldt] whose columns R..ldt-1 are zero filled (the contraction padding of the dW GEMM).
// colsum (optional): colsum[c] += sum_r x[r][c] in fp32 - the bias gradient of a Linear, taken from the same read of dy; summed
// over the row tiles in a fixed order (det_reduce.h).
// aux (optional, same layout as x): the activation backward of the fused Linear+activation is applied while reading,
// x := x * act'(aux) (act 1: ReLU, aux = forward output; act 2: exact-erf GELU, aux = pre-activation; the
// arithmetic of act_bwd_kernel in rowops.hip) - the fp32 gradient w.r.t. the pre-activation never reaches HBM.
__device__ __forceinline__ void cvt_bf16_tile(const float* __restrict__ x, long ldx, int R, int C,
                                              unsigned short* __restrict__ out, unsigned short* __restrict__ out_lo, long ldo,
                                              unsigned short* __restrict__ outT, long ldt, float* __restrict__ colsum,
                                              const float* __restrict__ aux, int act, const int r0, const int c0, const DetWs& ws,
                                              const bool lo_f16 = false) {
    __shared__ unsigned short tile[64][66];
    __shared__ float csum[16][64];
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;        // 16 column quads x 16 row lanes
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 16 * i, c = c0 + tx * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < R) {
            if (c + 3 < C && ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
                const float4 q = *reinterpret_cast<const float4*>(x + (long)r * ldx + c);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (c + j < C) v[j] = x[(long)r * ldx + c + j];
            }
            if (aux) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (c + j >= C) continue;
                    const float h = aux[(long)r * ldx + c + j];
                    if (act == 1) v[j] = h > 0.f ? v[j] : 0.f;
                    else {
                        const float cdf = 0.5f * (1.f + spe_erff(h * 0.70710678118654752f));
                        const float pdf = 0.3989422804014327f * __expf(-0.5f * h * h);
                        v[j] = v[j] * (cdf + h * pdf);
                    }
                }
            }
        }
        cs[0] += v[0]; cs[1] += v[1]; cs[2] += v[2]; cs[3] += v[3];
        typedef __bf16 bf16x4v_t __attribute__((ext_vector_type(4)));
        bf16x4v_t h;
        h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
        const uint2 u = __builtin_bit_cast(uint2, h);
        if (out && r < R) {
            if (c + 3 < C && ((ldo & 3) == 0)) *reinterpret_cast<uint2*>(out + (long)r * ldo + c) = u;
            else {
                const unsigned short e[4] = {(unsigned short)(u.x & 0xffff), (unsigned short)(u.x >> 16), (unsigned short)(u.y & 0xffff), (unsigned short)(u.y >> 16)};
#pragma unroll
                for (int j = 0; j < 4; ++j) if (c + j < C) out[(long)r * ldo + c + j] = e[j];
            }
        }
        if (out_lo && r < R) {          // low part of the split operand: bf16(x - bf16(x)), same layout as `out` (lo_f16: the IEEE fp16 copy)
            const uint2 ul = spe_second16(v, u, lo_f16);
            if (c + 3 < C && ((ldo & 3) == 0)) *reinterpret_cast<uint2*>(out_lo + (long)r * ldo + c) = ul;
            else {
                const unsigned short e[4] = {(unsigned short)(ul.x & 0xffff), (unsigned short)(ul.x >> 16), (unsigned short)(ul.y & 0xffff), (unsigned short)(ul.y >> 16)};
#pragma unroll
                for (int j = 0; j < 4; ++j) if (c + j < C) out_lo[(long)r * ldo + c + j] = e[j];
            }
        }
        if (outT) {
            tile[ty + 16 * i][tx * 4 + 0] = (unsigned short)(u.x & 0xffff); tile[ty + 16 * i][tx * 4 + 1] = (unsigned short)(u.x >> 16);
            tile[ty + 16 * i][tx * 4 + 2] = (unsigned short)(u.y & 0xffff); tile[ty + 16 * i][tx * 4 + 3] = (unsigned short)(u.y >> 16);
        }
    }
    if (colsum && r0 < R) {          // block-uniform: rows of this tile are inside the matrix
#pragma unroll
        for (int j = 0; j < 4; ++j) csum[ty][tx * 4 + j] = cs[j];
        __syncthreads();
        // 16 row lanes in order, then the row tiles of this column tile in order (det_reduce.h): colsum += total
        det_reduce(ws, c0 / 64, r0 / 64, (R + 63) / 64, 64, threadIdx.x, 256,
                   [&](int k) {
                       float t = 0.f;
#pragma unroll
                       for (int i = 0; i < 16; ++i) t += csum[i][k];
                       return t;
                   },
                   [&](int k, float t) { if (c0 + k < C) colsum[c0 + k] += t; });
    }
    if (!outT) return;
    __syncthreads();
    // transposed write: thread -> (column c0 + ty + 16*i, rows r0 + 4*tx .. +3); rows beyond R were staged as zeros
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 16 * i;
        const int r = r0 + tx * 4;
        if (c >= C || r >= ldt) continue;
        const unsigned short e0 = tile[tx * 4 + 0][ty + 16 * i], e1 = tile[tx * 4 + 1][ty + 16 * i];
        const unsigned short e2 = tile[tx * 4 + 2][ty + 16 * i], e3 = tile[tx * 4 + 3][ty + 16 * i];
        if (r + 3 < ldt && ((ldt & 3) == 0)) {
            uint2 u; u.x = (unsigned)e0 | ((unsigned)e1 << 16); u.y = (unsigned)e2 | ((unsigned)e3 << 16);
            *reinterpret_cast<uint2*>(outT + (long)c * ldt + r) = u;
        } else {
            const unsigned short e[4] = {e0, e1, e2, e3};
#pragma unroll
            for (int j = 0; j < 4; ++j) if (r + j < ldt) outT[(long)c * ldt + r + j] = e[j];
        }
    }
}

__global__ __launch_bounds__(256) void cvt_bf16_kernel(const float* __restrict__ x, long ldx, int R, int C,
                                                       unsigned short* __restrict__ out, unsigned short* __restrict__ out_lo, long ldo,
                                                       unsigned short* __restrict__ outT, long ldt, float* __restrict__ colsum,
                                                       const float* __restrict__ aux, int act, DetWs ws, bool lo_f16) {
    cvt_bf16_tile(x, ldx, R, C, out, out_lo, ldo, outT, ldt, colsum, aux, act, blockIdx.y * 64, blockIdx.x * 64, ws, lo_f16);
}

// Many contiguous matrices in one launch (the bf16 copies of every Linear weight after an optimizer step: ~190
// launch-bound conversions of 0.1-0.6 M elements otherwise).  jobs (device memory, built once by the host side):
// tile0 = first 64x64 tile of the job in the launch, ascending; a workgroup finds its job by bisection.
struct CvtJob { const float* x; unsigned short* out; unsigned short* outT; long ldt; int R, C, tile0, tiles_c; unsigned short* out_lo; long flags; };   // flags bit 0: out_lo receives IEEE fp16(x)
static_assert(sizeof(CvtJob) == 64, "spe_cvt_job_t layout");
__global__ __launch_bounds__(256) void cvt_bf16_multi_kernel(const CvtJob* __restrict__ jobs, int njobs) {
    const int t = blockIdx.x;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].tile0 <= t) lo = mid; else hi = mid - 1;
    }
    const CvtJob j = jobs[lo];
    const int lt = t - j.tile0;
    cvt_bf16_tile(j.x, j.C, j.R, j.C, j.out, j.out_lo, j.C, j.outT, j.ldt, nullptr, nullptr, 0, (lt / j.tiles_c) * 64, (lt % j.tiles_c) * 64, DetWs{}, (j.flags & 1) != 0);
}

// C-ABI: see include/spe_hip.h (spe_cvt_bf16_multi).
extern "C" int spe_cvt_bf16_multi(const void* jobs_dev, int njobs, int total_tiles, hipStream_t stream) {
    if (njobs <= 0 || total_tiles <= 0) return 0;
    hipLaunchKernelGGL(cvt_bf16_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, stream,
                       reinterpret_cast<const CvtJob*>(jobs_dev), njobs);
    SPE_CHECK_LAUNCH();
    return 0;
}

// C-ABI: see include/spe_hip.h (spe_cvt_bf16).
static int cvt_bf16_launch(const float* x, long ldx, int R, int C, void* out, void* out_lo, long ldo, void* outT, long ldt, float* colsum,
                           const float* aux, int act, bool lo_f16, hipStream_t stream);
extern "C" int spe_cvt_bf16(const float* x, long ldx, int R, int C, void* out, void* out_lo, long ldo, void* outT, long ldt, float* colsum,
                            const float* aux, int act, hipStream_t stream) {
    return cvt_bf16_launch(x, ldx, R, C, out, out_lo, ldo, outT, ldt, colsum, aux, act, false, stream);
}
// C-ABI: see include/spe_hip.h (spe_cvt_bf16_h): bf16 copy + IEEE fp16 copy of x from one pass.
extern "C" int spe_cvt_bf16_h(const float* x, long ldx, int R, int C, void* out, void* out_h, long ldo, void* outT, long ldt, hipStream_t stream) {
    return cvt_bf16_launch(x, ldx, R, C, out, out_h, ldo, outT, ldt, nullptr, nullptr, 0, true, stream);
}
static int cvt_bf16_launch(const float* x, long ldx, int R, int C, void* out, void* out_lo, long ldo, void* outT, long ldt, float* colsum,
                           const float* aux, int act, bool lo_f16, hipStream_t stream) {
    if (R <= 0 || C <= 0) return 0;
    if (!out && !out_lo && !outT && !colsum) return 0;
    if (outT && ldt < R) return -2;
    // the grid covers the padded row range of the transpose so that its zero columns are written too
    const long rows = outT ? ((ldt > R) ? ldt : R) : R;
    dim3 grid((C + 63) / 64, (unsigned)((rows + 63) / 64));
    DetWs ws = spe_detws();
    const DetDeferSeg sg[1] = {{colsum, C}};
    float* region = colsum ? det_defer_try((C + 63) / 64, (R + 63) / 64, 64, 1, sg, stream) : nullptr;  // deferred: colsum += totals at the next flush
    if (region) ws.defer = region;
    else if (colsum) DET_CHECK(ws, (C + 63) / 64, (R + 63) / 64, 64);
    hipLaunchKernelGGL(cvt_bf16_kernel, grid, dim3(256), 0, stream, x, ldx, R, C, reinterpret_cast<unsigned short*>(out),
                       reinterpret_cast<unsigned short*>(out_lo), ldo,
                       reinterpret_cast<unsigned short*>(outT), ldt, colsum, aux, act, ws, lo_f16);
    if (region) det_defer_commit(region, (C + 63) / 64, (R + 63) / 64, 64, 1, sg, 1);
    SPE_CHECK_LAUNCH();
    return 0;
}
