// 128-row-tile bf16 NT GEMM for the activation-sized Linear products of the SPE hot path (M = B*N tokens >= 2048 rows):
//
//   C[m][n] = epilogue( alpha * sum_k A[m][k] * B[n][k] + bias[n] )       A [M, K], B [N, K] bf16, both k-contiguous
//
// the three GEMMs of every backbone nn.Linear (reference models/cait.py:376,390,409: qkv, proj, fc1, fc2 and their input
// gradients) and the memory-side projections of the decoder's cross attention (models/transformer.py:389-396 - north_star's
// "decoder cross-attention GEMM").  Same C-ABI entry points and epilogues as gemm_bf16.hip (gemm16_epilogue.h); this file only
// replaces the main loop for the shapes where it wins:
//
//   * 128 x BN x BK workgroup tile (BN = 128, or 64 for N < 1024 so that 8300 x 384 outputs still give 390 workgroups), 4 waves as
//     2 x 2, wave tile 64 x BN/2 = 4 x (BN/32) v_mfma_f32_16x16x32_bf16 accumulators: one ds_read_b128 feeds 2-4 MFMAs (the 64x64
//     tiles of gemm_bf16.hip: one read per MFMA - its split variant is LDS-issue bound);
//   * every operand byte travels global -> LDS by global_load_lds_dwordx4 (no staging registers, no VGPR write-back, nothing for
//     hipcc to drain) into a ring of NST stages; a wave instruction deposits 1 KB lane-linearly (8 rows of 128 B at BK = 64, 16 rows
//     of 64 B at BK = 32), so the 16-B chunks of a row are permuted through the choice of the GLOBAL chunk each lane fetches and
//     the fragment reads apply the same involution: chunk ^ (row & 7) for 128-B rows, chunk ^ F[(row >> 2) & 3], F = {0, 2, 3, 1},
//     for 64-B rows - both conflict-free for the lane groups ds_read_b128 is serviced in (MI355X_MICROARCH.md, LDS table);
//   * stage t is admitted by a counted s_waitcnt vmcnt((NST - 2) * pieces per wave) and a RAW s_barrier (a __syncthreads() would
//     drain the DMA queue to vmcnt(0): the LDS-DMA is a pending LDS write on the VM counter);
//   * SPLIT operands (precision mode bf16s, forward products): hi and lo parts are four arrays of one stage (BK = 32 keeps the
//     stage at 32 KB), three MFMAs per tile step: A_hi B_hi + A_lo B_hi + A_hi B_lo.
// Two workgroups share a CU (64-70 KB of LDS each), so one's epilogue (bias / GELU / residual / bf16 copies, up to 100 MB of
// stores per launch) overlaps the other's main loop.
#include <cstdlib>
#include "common.h"
#include "gemm16_epilogue.h"

__device__ __forceinline__ void nt2_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// chunk permutation of row `r`: an involution on the 16-B chunk index, applied to the global source and to the fragment reads
template <int CPR>
__device__ __forceinline__ int nt2_swz(int r) {
    if constexpr (CPR == 8) return r & 7;
    else return (0x78 >> (2 * ((r >> 2) & 3))) & 3;
}

// F16: the operands hold IEEE fp16 (single-term product on v_mfma_f32_16x16x32_f16: same bytes and rate as bf16, 3 more mantissa bits)
template <int BM, int BN, int BK, int NST, bool SPLIT, bool EX, bool F16 = false>
__global__ __launch_bounds__(256, 2) void gemm_nt2_kernel(Gemm16Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    constexpr int NFM = BM / 32, NFN = BN / 32, WM = BM / 2, WN = BN / 2;
    constexpr int CPR = BK / 8;                   // 16-B chunks per LDS row (8: 128-B rows, 4: 64-B rows)
    constexpr int RPP = 64 / CPR;                 // rows per 1-KB piece (one wave instruction)
    constexpr int PA = BM / RPP, PB = BN / RPP;   // pieces of an A / B tile
    constexpr int NP = (PA + PB) * (SPLIT ? 2 : 1), PW = NP / 4;
    constexpr int STE = NP * 512;                 // bf16 elements per stage
    static_assert(NP % 4 == 0 && BM % 32 == 0 && BN % 32 == 0 && (BK == 32 || BK == 64) && NST >= 2, "tile geometry");

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int tm, tn;
    if (p.xcd_bind == 0) { tm = blockIdx.x % tiles_m; tn = blockIdx.x / tiles_m; }
    else {      // the panels of the operand with more rows are bound to XCDs (workgroup b runs on XCD b % 8): see gemm_bf16.hip
        const int no = (p.xcd_bind == 1) ? tiles_n : tiles_m;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int tb = xcd + 8 * (idx / no), to = idx % no;
        tm = (p.xcd_bind == 1) ? tb : to; tn = (p.xcd_bind == 1) ? to : tb;
        if (tm >= tiles_m || tn >= tiles_n) return;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    // Phase stagger: the two workgroups of a CU start together and would run their main loops (operand loads + MFMA) and then their
    // epilogues (tens of MB of stores) in lockstep - neither overlaps the other's.  The workgroups that fill the second slot
    // of the CUs (grid indices 256-511 of the first round) start `stagger` x ~4 us late, so one's epilogue meets the other's loop;
    // later rounds inherit the offset.
    if (p.stagger > 0 && (blockIdx.x >> 8) == 1) {
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
#if defined(SPE_ABLATE) && defined(SPE_ABL_NOLOOP)
    const int nt = 1;                              // timing experiment: one stage of the main loop
#else
    const int nt = p.K / BK;
#endif

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ws = __builtin_amdgcn_readfirstlane(w);
    const int wm = w >> 1, wn = w & 1;
    const int fr = lane & 15, fc = lane >> 4;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)smem16);

    f32x4_t acc[NFM][NFN];
#pragma unroll
    for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // this lane's share of a piece: row (lane / CPR) of the piece's RPP rows, LDS chunk slot lane % CPR
    const int pr = lane / CPR, pc = lane % CPR;
    auto issue = [&](int t, int slot) {
        const int tc = min(t, nt - 1);            // past the range: a valid tile, never used (keeps the vmcnt arithmetic static)
        const long k0 = (long)tc * BK;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int piece = i * 4 + ws;         // wave-uniform: which array a piece belongs to is a scalar decision
            constexpr int HALF = PA + PB;
            const bool lo = SPLIT && (piece >= HALF);
            const int ph = lo ? piece - HALF : piece;
            const bool isA = ph < PA;
            const int r = (isA ? ph : ph - PA) * RPP + pr;
            const int gc = pc ^ nt2_swz<CPR>(r);
            const unsigned short* base = isA ? (lo ? p.Alo : p.A) : (lo ? p.Blo : p.B);
            const unsigned short* src = isA ? base + (long)min(m0 + r, p.M - 1) * p.lda + k0 + gc * 8
                                            : base + (long)min(n0 + r, p.N - 1) * p.ldb + k0 + gc * 8;
            nt2_glds16(src, lds0 + (unsigned)((slot * STE + piece * 512) * 2));
        }
    };
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) issue(st, st);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * PW) : "memory");      // this wave's pieces of stage t have landed
        __builtin_amdgcn_s_barrier();                                               // everybody's have, and stage t - 1 has been read
        asm volatile("" ::: "memory");
        issue(t + NST - 1, (t + NST - 1) % NST);                                    // refill the slot of stage t - 1
        const unsigned short* sA_ = smem16 + (t % NST) * STE;
        const unsigned short* sB_ = sA_ + PA * 512;
        const unsigned short* sAl = sA_ + (PA + PB) * 512;
        const unsigned short* sBl = sAl + PA * 512;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            const int kc = ks * 4 + fc;
            bf16x8_t a[NFM], b[NFN], al[SPLIT ? NFM : 1], bl[SPLIT ? NFN : 1];
#pragma unroll
            for (int i = 0; i < NFM; ++i) {
                const int row = wm * WM + i * 16 + fr;
                const int off = row * BK + ((kc ^ nt2_swz<CPR>(row)) * 8);
                a[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sA_ + off));
                if constexpr (SPLIT) al[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sAl + off));
            }
#pragma unroll
            for (int j = 0; j < NFN; ++j) {
                const int row = wn * WN + j * 16 + fr;
                const int off = row * BK + ((kc ^ nt2_swz<CPR>(row)) * 8);
                b[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sB_ + off));
                if constexpr (SPLIT) bl[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u16x8_t*>(sBl + off));
            }
            if constexpr (SPLIT) {      // the two cross terms first; every accumulator sees ONE MFMA shape
#pragma unroll
                for (int i = 0; i < NFM; ++i)
#pragma unroll
                    for (int j = 0; j < NFN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], a[i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NFM; ++i)
#pragma unroll
                    for (int j = 0; j < NFN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], al[i], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NFM; ++i)
#pragma unroll
                for (int j = 0; j < NFN; ++j) {
                    if constexpr (F16) {
                        typedef _Float16 nt2_h8_t __attribute__((ext_vector_type(8)));
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(nt2_h8_t, b[j]), __builtin_bit_cast(nt2_h8_t, a[i]), acc[i][j], 0, 0, 0);
                    } else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the (unused) tail stages before the epilogue reuses the ring
    __syncthreads();
#if defined(SPE_ABLATE) && defined(SPE_ABL_NOSTORE)
#pragma unroll
    for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j) asm volatile("" :: "v"(acc[i][j]));      // timing experiment: no epilogue
    return;
#endif
    if constexpr (EX) gemm16_epilogue_ex<BM, BN, SPLIT || F16, false>(p, acc, smem16, m0, n0);      // (fp16 operands: the second 16-bit tile is the fp16 copy of the result)
    else if (F16 && (p.h16 & 2) && (p.N & 7) == 0 && (p.ldc & 7) == 0) gemm16_epilogue_h16<BM, BN>(p, acc, smem16, m0, n0);
    else gemm16_epilogue_plain<BM, BN>(p, acc, p.C, m0, n0);
}

template <int BM, int BN, int BK, int NST, bool SPLIT, bool EX, bool F16 = false>
static int launch_nt2(const Gemm16Args& p, hipStream_t stream) {
    constexpr int ring = NST * ((BM + BN) / (64 / (BK / 8))) * (SPLIT ? 2 : 1) * 1024;
    constexpr int epi = EX ? BM * (BN + 8) * 2 * ((SPLIT || F16) ? 2 : 1) : 0;
    constexpr int smem = ring > epi ? ring : epi;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt2_kernel<BM, BN, BK, NST, SPLIT, EX, F16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    Gemm16Args q = p;
    static const int stagger = SPE_KNOB("SPE_NT2_STAGGER", 0);
    q.stagger = stagger;
    q.xcd_bind = 0;
    if (p.M >= p.N && tiles_m >= 16) q.xcd_bind = 1;
    else if (p.N > p.M && tiles_n >= 16) q.xcd_bind = 2;
    else if (tiles_m >= 16) q.xcd_bind = 1;
    else if (tiles_n >= 16) q.xcd_bind = 2;
    int tiles = tiles_m * tiles_n;
    if (q.xcd_bind == 1) tiles = 8 * ((tiles_m + 7) / 8) * tiles_n;
    if (q.xcd_bind == 2) tiles = 8 * ((tiles_n + 7) / 8) * tiles_m;
    // column sums of the extended epilogue (a bias gradient): deferred when the caller said so - sets = column tiles, members = row tiles
    const DetDeferSeg sg[1] = {{q.colsum, q.N}};
    float* region = (EX && q.colsum) ? det_defer_try(tiles_n, tiles_m, BN, 1, sg, stream) : nullptr;
    if (region) q.ws.defer = region;
    hipLaunchKernelGGL((gemm_nt2_kernel<BM, BN, BK, NST, SPLIT, EX, F16>), dim3(tiles), dim3(256), smem, stream, q);
    if (region) det_defer_commit(region, tiles_n, tiles_m, BN, 1, sg, 1);
    SPE_CHECK_LAUNCH();
    return 0;
}

// Called by spe_gemm_bf16nt / spe_gemm_bf16nt_ex (gemm_bf16.hip) with validated arguments.  Returns SPE_NT2_NA when this kernel
// family does not cover the problem (the caller then runs its own kernels): fewer than 2048 rows, a contraction that is not a
// multiple of the stage depth, a K split, or a transposed bf16 copy of the result.
#define SPE_NT2_NA (-100)
int spe_nt2_dispatch(const Gemm16Args& p, bool ex, hipStream_t stream) {
    static const int enabled = SPE_KNOB("SPE_GEMM_NT2", 1);      // developer knob (A/B against gemm_bf16.hip)
    const bool split = p.Alo != nullptr;
    if (!enabled || p.M < 2048 || p.splitk != 1 || p.out16T || (p.K % 64) != 0 || p.K < 128 || p.N < 64) return SPE_NT2_NA;
    if (!(p.h16 & 1) && (p.h16 & 4)) return -2;        // the fp16 second copy comes with fp16 operands only
    if ((p.h16 & 1) && ex) {
        // fp16 single-term operands with the extended epilogue (round 5: the backbone MLP's forward products in precision mode bf16s -
        // fc1 + GELU emitting the bf16 copy for the backward and the fp16 copy for fc2, fc2 + LayerScale residual)
        if (split) return -2;
        if (p.N >= 1024) return launch_nt2<128, 128, 64, 2, false, true, true>(p, stream);
        return launch_nt2<128, 64, 64, 2, false, true, true>(p, stream);
    }
    if (p.h16 & 1) {        // fp16 single-term operands (the decoder's memory-side projections): wide tiles, plain epilogue
        if (split) return -2;
        const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128), t160 = (long)((p.M + 159) / 160) * ((p.N + 127) / 128);
        if (((t160 + 511) / 512) * 160 < ((t128 + 511) / 512) * 128) return launch_nt2<160, 128, 64, 2, false, false, true>(p, stream);
        return launch_nt2<128, 128, 64, 2, false, false, true>(p, stream);
    }
    static const int wide_min = SPE_KNOB("SPE_NT2_WIDE_MIN", 1024);     // developer knob
    // single-term products with the extended epilogue (fc2 dh: GELU derivative from the saved pre-activation, bf16 output, column
    // sums) are bound by that epilogue: 128 x 64 tiles at three workgroups per CU overlap it with other workgroups' main loops
    // (8300 x 1536 x 384: 63 -> 51 us); the split forward products and the plain-epilogue ones are faster on the wide tiles
    static const int wide_min_ex1 = SPE_KNOB("SPE_NT2_WIDE_MIN_EX1", 2048);      // developer knob
    // ... and so are the split forward products with the extended epilogue (fc1 + GELU: fp16 pre-activation + hi / lo bf16 outputs):
    // 85 -> 73 us INSIDE the step on 128 x 64 tiles (the isolated launch prefers the wide tiles, 69 vs 76 us: measured in the step)
    static const int wide_min_ex3 = SPE_KNOB("SPE_NT2_WIDE_MIN_EX3", 2048);           // developer knob
    const bool wide = p.N >= (ex ? (split ? wide_min_ex3 : wide_min_ex1) : wide_min);
    static const int cfg = SPE_KNOB("SPE_NT2_CFG", 0);      // developer knob: ring depth / stage depth variants
#define NT2_GO(BK_, NST_, SP_)                                                                                                   \
    do {                                                                                                                         \
        if (ex) return wide ? launch_nt2<128, 128, BK_, NST_, SP_, true>(p, stream) : launch_nt2<128, 64, BK_, NST_, SP_, true>(p, stream);   \
        return wide ? launch_nt2<128, 128, BK_, NST_, SP_, false>(p, stream) : launch_nt2<128, 64, BK_, NST_, SP_, false>(p, stream);         \
    } while (0)
    // Tile quantisation: 8300 rows make 65 row tiles of 128; 65 x 9 = 585 tiles (qkv forward) take 2 rounds on the 512 resident workgroup
    // slots for 1.14 rounds of work.  160-row tiles (52 x 9 = 468) fit one round of 1.25x larger tiles.  Plain epilogue only (the staged
    // epilogue of a 160 x 128 tile does not fit two workgroups per CU).
    static const int tall = SPE_KNOB("SPE_NT2_TALL", 1);      // developer knob (A/B)
    if (tall && !ex && wide && cfg == 0) {
        const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128), t160 = (long)((p.M + 159) / 160) * ((p.N + 127) / 128);
        const long c128 = ((t128 + 511) / 512) * 128, c160 = ((t160 + 511) / 512) * 160;
        if (c160 < c128) return split ? launch_nt2<160, 128, 32, 2, true, false>(p, stream) : launch_nt2<160, 128, 64, 2, false, false>(p, stream);
    }
    // (A balanced single round for narrow outputs - 52 evenly spread 160-row panels x 4 column tiles of 96, one workgroup per CU, 3-stage ring - was
    // measured slower in round 4 and is gone: profiles/r04_bench_nt.txt, profiles/HISTORY_r05.md.)
    // Narrow outputs (N = 384: the input-gradient products, proj / fc2 forward): 8300 x 384 is 390 tiles of 128 x 64 - 1.5 workgroups
    // per CU.  64 x 64 tiles (780 workgroups, three to four per CU) hide each other's load latency: qkv dx 19.4 -> 17.0 us, fc1 dx
    // 24.0 -> 21.5, the stacked decoder dx (K = 4608) 59.3 -> 53.1.  SPE_NT2_SHORT: bit 0 single-term plain, bit 1 split plain,
    // bit 2 split extended epilogue, bit 3 single-term extended epilogue (developer knob, A/B).
    static const int short_rows = SPE_KNOB("SPE_NT2_SHORT", 9);      // in the step: fc2 dh 64.5 -> 56.3 us with bit 3; bits 1, 2 no gain
    if (!wide && cfg == 0) {
        if ((short_rows & 1) && !ex && !split) return launch_nt2<64, 64, 64, 2, false, false>(p, stream);
        if ((short_rows & 2) && !ex && split) return launch_nt2<64, 64, 32, 2, true, false>(p, stream);
        if ((short_rows & 4) && ex && split) return launch_nt2<64, 64, 32, 2, true, true>(p, stream);
        if ((short_rows & 8) && ex && !split) return launch_nt2<64, 64, 64, 2, false, true>(p, stream);
    }
    if (split) {
        if (cfg == 1) NT2_GO(32, 3, true);
        if (cfg == 2) NT2_GO(32, 4, true);
        NT2_GO(32, 2, true);
    }
    if (cfg == 1) NT2_GO(64, 3, false);
    if (cfg == 2) NT2_GO(64, 4, false);
    if (cfg == 3) NT2_GO(32, 4, false);
    if (cfg == 4) NT2_GO(32, 6, false);
    if (cfg == 5) NT2_GO(32, 2, false);
    if (cfg == 6) NT2_GO(32, 3, false);
    NT2_GO(64, 2, false);
#undef NT2_GO
}
