// A-resident persistent NT GEMM for a SHORT contraction (K = 64 KS, KS <= 6: the token width of the backbone, 384 at cfg2) against a wide weight:
//
//   C[m][n] = fp16( alpha * sum_k A[m][k] * B[n][k] + bias[n] )       A [M, K], B [N, K] IEEE fp16, both k-contiguous; C [M, N] fp16
//
// - north_star's "decoder cross-attention GEMM": the memory-side projections ca_kcontent_proj / ca_v_proj (and ca_kpos_proj of `pos`) of ALL decoder
// layers as one product [B*S, d] x [d, 2 L d] (reference models/transformer.py:389-396, ops._MemorySideKV), [8300 x 384] x [384 x 4608] at cfg2.
//
// Why another main loop.  The 160 x 128 tiles of gemm_nt2.hip re-request both operands for every output tile: 1872 tiles x 216 KB = 404 MB of L2 -> LDS
// traffic for a 29-GF product, which the L2 fabric delivers at ~10 TB/s - 42 us of main loop (profiles/r05_cagemm_phases.txt) against 11.8 us of matrix
// pipe.  At K = 384 a wave's 64 rows of A are only 48 KB - 192 registers per lane: they stay IN REGISTERS for a whole row panel, and the register file is
// the largest memory of a CU (512 KB).  A workgroup of 4 waves (one per SIMD, 512 registers each) owns a panel of 256 rows and walks over column tiles of
// 128: only the weight tile is streamed (LDS-DMA ring of 16-KB stages, 64 deep, shared by the four waves), one ds_read_b128 feeds FOUR matrix instructions
// (the 128 x 128 tiles of gemm_nt2.hip: two), the 128 accumulators of the 64 x 128 wave tile live in AccVGPRs.  Operand traffic: 33 panels x 3.5 MB of
// weights + A once = 123 MB (3.3x less).  The (panel, column tile) units are flattened panel-major and cut into equal contiguous ranges, one per CU (1188
// units on 256 workgroups: 4 or 5 each; a range touches at most two panels), and the ring keeps running across units: the first stages of the next tile
// land while this tile's epilogue (fp16 pack -> wave-private LDS tile -> 16-B row stores) runs.
//
// Matrix instructions are inline assembly with the accumulators constrained to AccVGPRs ("+a"): at one wave per SIMD hipcc (ROCm 7.2) schedules the
// builtin in its AccVGPR form at ~45 clk per instruction instead of the hardware's 18 (profiles/r05_mfma_form.txt, measured again on this kernel:
// profiles/r06_gemm_ares.txt).  No compiler-visible vector-memory load may be live inside the unit loop: hipcc's own s_waitcnt insertion does not see the
// LDS-DMA instructions and would drain the ring (a vmcnt(0) in front of the first use of a bias value: measured) - the A fragments are followed by an
// explicit vmcnt(0) the compiler knows about, the bias vector is copied to LDS once.
//
// Waits.  Loads (LDS-DMA and the A fragments) and stores share the VM counter and retire in issue order on gfx9-class hardware (the compiler's own
// s_waitcnt insertion relies on it), so the wait that admits a stage counts the younger operations - the stages in flight behind it and, in the first
// three stages after an epilogue, that epilogue's 16 row stores.
#include "common.h"
#include "gemm16_epilogue.h"

#define ARES_BM 256                      // rows per workgroup (64 per wave)
#define ARES_BN 128                      // columns per unit
#define ARES_NST 4                       // ring stages (3 in flight)
#define ARES_PW 4                        // 1-KB pieces of a stage per wave (16 pieces: 128 weight rows x 128 B)
#define ARES_EST 16                      // row-store instructions of one epilogue per wave (64 rows x 256 B / 1 KB)
#define ARES_LR (ARES_BN + 8)            // halves per row of the epilogue tile

struct AresArgs {
    const unsigned short* A; const unsigned short* B; unsigned short* C; const float* bias;
    int M, N; long lda, ldb, ldc; float alpha;
    int tiles_n, total_units, units_per_wg;
};

__device__ __forceinline__ void ares_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int KS>
__global__ __launch_bounds__(256, 1) void gemm_ares_kernel(AresArgs p) {
    typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
    constexpr int NST = ARES_NST, PW = ARES_PW, STE = 16 * 512;          // halves per stage
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];       // [NST stages][4 waves x 64 x ARES_LR epilogue tiles][bias of the workgroup's column range]
    const int lane = threadIdx.x & 63, ws = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)smem16);
    unsigned short* etile = smem16 + NST * STE + ws * (64 * ARES_LR);
    float* sbias = reinterpret_cast<float*>(smem16 + NST * STE + 4 * (64 * ARES_LR));        // [N] (all columns: a range may wrap around the tiles)

    const int u_begin = blockIdx.x * p.units_per_wg;
    int u_end = u_begin + p.units_per_wg; if (u_end > p.total_units) u_end = p.total_units;
    if (u_begin >= u_end) return;
    const int nunits = u_end - u_begin, nstages = nunits * KS;

    // stage t of this workgroup's stream = k-step t % KS of unit u_begin + t / KS ; piece i * 4 + wave = weight rows 8 piece .. + 7, this lane: row
    // lane / 8 of them, LDS chunk slot lane % 8 (the 16-B chunks of a 128-B row are permuted by chunk ^ (row & 7): conflict-free fragment reads)
    const int pr = lane >> 3, pc = lane & 7;
    auto issue = [&](int t) {
#if defined(SPE_ABLATE) && defined(ARES_DBG_NODMA)
        if (t >= ARES_NST - 1) return;                 // timing experiment: the ring is filled once, never refilled (results invalid)
#endif
        const int tc = min(t, nstages - 1);            // past the range: a valid stage, never used (keeps the wait arithmetic static)
        const int u = u_begin + tc / KS, ks = tc % KS;
        const int n0 = (u % p.tiles_n) * ARES_BN;
        const unsigned dst = lds0 + (unsigned)(((t % NST) * STE) * 2);
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int piece = i * 4 + ws;
            const int r = piece * 8 + pr;
            const unsigned short* src = p.B + (long)min(n0 + r, p.N - 1) * p.ldb + ks * 64 + ((pc ^ (r & 7)) * 8);
            ares_glds16(src, dst + (unsigned)(piece * 1024));
        }
    };

    h8_t a[4][2 * KS];                                  // this wave's 64 rows of A, all of K: fragment (16 rows i, 32-deep step k32)
    auto load_a = [&](int panel) {
        const int m0 = panel * ARES_BM + ws * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned short* row = p.A + (long)min(m0 + i * 16 + fr, p.M - 1) * p.lda + fc * 8;
#pragma unroll
            for (int k32 = 0; k32 < 2 * KS; ++k32) a[i][k32] = *reinterpret_cast<const h8_t*>(row + k32 * 32);
        }
        // vmcnt(0) as an instruction the compiler's wait-count insertion sees (simm16: vmcnt 0, expcnt 7, lgkmcnt 15 = no wait): without it every first use
        // of a fragment inside the unit loop gets its own vmcnt(k), k = 47 .. 0, which the LDS-DMA ring's entries fall under
        __builtin_amdgcn_s_waitcnt(0x0F70);
    };

    f32x4_t acc[4][8];
    // one k-step of 64: stage t has landed for everybody, the slot of stage t - 1 is refilled, 2 x (8 fragment reads, 32 matrix instructions)
    auto kstep = [&](int t, int ks) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(t + NST - 1);
        const unsigned short* sB = smem16 + (t % NST) * STE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h8_t b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = j * 16 + fr;
#if defined(SPE_ABLATE) && defined(ARES_DBG_NOLDS)
                b[j] = a[j & 3][(2 * ks + kk + 1) % (2 * KS)];         // timing experiment: no fragment reads (results invalid)
#else
                b[j] = *reinterpret_cast<const h8_t*>(sB + row * 64 + (((kk * 4 + fc) ^ (row & 7)) * 8));
#endif
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#if defined(SPE_ABLATE) && defined(ARES_DBG_NOMFMA)
                    if (i == 0) acc[i][j][0] += (float)b[j][0];            // timing experiment: the fragment reads stay, the matrix instructions go
#elif defined(SPE_ABLATE) && defined(ARES_DBG_MFMA32)
                    // timing experiment (results invalid): half as many 32x32x16 instructions on the same registers - the issue rate of the other shape
                    if ((j & 1) == 0) {
                        typedef float f32x16_t __attribute__((ext_vector_type(16)));
                        f32x16_t& c16 = *reinterpret_cast<f32x16_t*>(&acc[i][((j >> 1) & 1) * 4]);
                        asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c16) : "v"(b[j]), "v"(a[i][2 * ks + kk]));
                    }
#elif defined(SPE_ABLATE) && defined(ARES_DBG_BUILTIN)
                    acc[i][j] = (ks == 0 && kk == 0) ? __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i][2 * ks + kk], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0)
                                                     : __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i][2 * ks + kk], acc[i][j], 0, 0, 0);
#else
                    // the first product of a unit starts the accumulator from 0 (no zeroing pass)
                    if (ks == 0 && kk == 0) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc[i][j]) : "v"(b[j]), "v"(a[i][0]));
                    else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(b[j]), "v"(a[i][2 * ks + kk]));
#endif
                }
        }
    };
    // acc[i][j][r] = C[m0 + i*16 + (lane & 15)][n0 + j*16 + 4 (lane >> 4) + r]: fp16 through the wave's LDS tile, out as 16-B pieces of whole 256-B rows
    auto epilogue = [&](int u) {
        const int panel = u / p.tiles_n, n0 = (u % p.tiles_n) * ARES_BN;
        const int m0 = panel * ARES_BM + ws * 64;
        // the accumulators become readable: 8-pass matrix instructions -> vector reads need 13+ wait states behind the last one
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int nl = j * 16 + fc * 4;
            const float4 bv = *reinterpret_cast<const float4*>(sbias + n0 + nl);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<uint2*>(etile + (i * 16 + fr) * ARES_LR + nl) = ep_f2h4(acc[i][j][0] * p.alpha + bv.x, acc[i][j][1] * p.alpha + bv.y,
                                                                                      acc[i][j][2] * p.alpha + bv.z, acc[i][j][3] * p.alpha + bv.w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // wave-private tile: the other lanes' packets are read next
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < ARES_EST; ++it) {
            const int idx = it * 64 + lane, r = idx >> 4, c8 = idx & 15;
            const spe_u32x4_t q = *reinterpret_cast<const spe_u32x4_t*>(etile + r * ARES_LR + c8 * 8);
            // UNCONDITIONAL: every wave issues exactly ARES_EST full store instructions per epilogue (the counted waits below rely on it); the rows M ..
            // of the last panel go to the padding rows the caller allocated (c_rows >= 256 ceil(M / 256): checked by the launcher)
#if defined(SPE_ABLATE) && defined(ARES_DBG_NOSTORE)
            asm volatile("" :: "v"(q));                                 // timing experiment: no output stores (results invalid)
#elif defined(SPE_ABLATE) && defined(ARES_DBG_PLAINSTORE)
            *reinterpret_cast<spe_u32x4_t*>(p.C + (long)(m0 + r) * p.ldc + n0 + c8 * 8) = q;
#else
            spe_store16_stream(p.C + (long)(m0 + r) * p.ldc + n0 + c8 * 8, q);
#endif
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the tile is rewritten by the next unit
    };

    // the bias vector -> LDS (zeros without one): the epilogues then need no vector-memory load
    for (int n = threadIdx.x; n < p.N; n += 256) sbias[n] = p.bias ? p.bias[n] : 0.f;
    int panel = u_begin / p.tiles_n;
    load_a(panel);                                      // (ends in vmcnt(0): the bias loads are done as well)
    __syncthreads();
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) issue(st);
    // first unit: nothing but the ring in flight
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * PW) : "memory");
        kstep(ks, ks);
    }
    epilogue(u_begin);
    for (int ui = 1; ui < nunits; ++ui) {
        const int u = u_begin + ui;
        const int pn = u / p.tiles_n;
        if (pn != panel) {                              // next row panel (at most once per workgroup): everything in flight lands first
            panel = pn;
            load_a(panel);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // younger than stage t at this point: the stages t + 1, t + 2 and - in the first NST - 1 steps after an epilogue - its ARES_EST stores
            // (issued after the refill of step KS - 1 of the previous unit, i.e. after stage t + 2 of step 0, t + 1 of step 1, t of step 2)
            if (ks == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PW + ARES_EST) : "memory");
            else if (ks == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PW + ARES_EST) : "memory");
            else if (ks == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PW + ARES_EST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PW) : "memory");
            kstep(ui * KS + ks, ks);
        }
        epilogue(u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the (unused) tail stages
}

template <int KS>
static int launch_ares(const AresArgs& a, int nwg, hipStream_t stream) {
    const int smem = ARES_NST * 16 * 1024 + 4 * 64 * ARES_LR * 2 + a.N * 4;
    if (smem > 160 * 1024) return -2;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ares_kernel<KS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_ares_kernel<KS>), dim3(nwg), dim3(256), smem, stream, a);
    SPE_CHECK_LAUNCH();
    return 0;
}

// C-ABI: see include/spe_hip.h (spe_gemm_f16nt_wide).  -2: shape / alignment not covered (callers use spe_gemm_bf16nt).
extern "C" int spe_gemm_f16nt_wide(const void* A16, const void* B16, void* C16, const float* bias, int M, int N, int K, long lda, long ldb, long ldc,
                                   long c_rows, float alpha, hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    if (!A16 || !B16 || !C16) return -2;
    if ((K != 384 && K != 192) || (N % ARES_BN) != 0 || N < 4 * ARES_BN || N > 6144 || M < 2048) return -2;
    const int panels = (M + ARES_BM - 1) / ARES_BM;
    if (c_rows < (long)panels * ARES_BM) return -2;
    if ((lda & 7) || (ldb & 7) || (ldc & 7) || lda < K || ldb < K || ldc < N || (reinterpret_cast<uintptr_t>(A16) & 15) ||
        (reinterpret_cast<uintptr_t>(B16) & 15) || (reinterpret_cast<uintptr_t>(C16) & 15) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15))) return -2;
    AresArgs a;
    a.A = reinterpret_cast<const unsigned short*>(A16); a.B = reinterpret_cast<const unsigned short*>(B16);
    a.C = reinterpret_cast<unsigned short*>(C16); a.bias = bias;
    a.M = M; a.N = N; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.alpha = alpha;
    a.tiles_n = N / ARES_BN;
    a.total_units = panels * a.tiles_n;
    int nwg = a.total_units < 256 ? a.total_units : 256;          // one workgroup per CU
    a.units_per_wg = (a.total_units + nwg - 1) / nwg;
    nwg = (a.total_units + a.units_per_wg - 1) / a.units_per_wg;
    return K == 384 ? launch_ares<6>(a, nwg, stream) : launch_ares<3>(a, nwg, stream);
}
