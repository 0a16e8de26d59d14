// The two backward kernels of the talking-heads attention (K4 of SURVEY.md section 2.2; reference models/cait.py:377-389 and its autograd), built for
// ONE wave per SIMD and the whole 512-entry register file:
//
//   key-major   (spe_talking_bwdk_pass1): D[h', q] = sum_k dP[h'] P[h'] (the softmax backward's row term), dWw, dbw, dV += P'd^T dO in registers
//   query-major (spe_talking_bwdq_pass2): dS' = P (dP - D), dWl, dbl, dS = Wl^T dS' -> bf16 16 x 16 blocks (read once more, by the dK contraction)
//                                         AND dQ += dS K in registers
//
// with S = scale q k^T, S' = Wl S + bl, P = softmax_k(S'), P' = Ww P + bw, dP' = dropout-mask * (dO V^T), dP = Ww^T dP'.  Both kernels
// recompute S / S' / P from the forward's own fp16 fragments and statistics (P matches the forward exactly) and dP' from bf16 fragments.
//
// Register plan (why this file is compiled with -mllvm -amdgpu-mfma-vgpr-form=1, spe_amd/build.py).  A wave owns one 16-row tile for a whole
// segment of tiles of the other axis.  Its two fragment records (2 x 48 registers at cfg2) and the 96 accumulators live in the AccVGPR half of
// the register file: they are touched by matrix instructions only, as B operands resp. C / D, through inline assembly with "a" constraints.
// Everything else (the score tile of all heads, both head-mix accumulator sets, the mixing weights) stays under 256 ordinary VGPRs, and with
// the VGPR form forced for the builtin matrix instructions hipcc moves nothing between the two halves inside the tile loop (without the flag every
// builtin result lands in an AccVGPR once a kernel may use them and is copied out for the vector instructions: 4.4 moves per matrix instruction
// measured on the round-4 prototype of the forward kernel).  The resident operands in registers instead of LDS is what makes the flash skeleton
// fit: 4 resident tiles x 2 records would be 96 KB next to 72 KB of stage buffers and 32 KB of transpose tiles.
//
// LDS: two stages of the streamed tiles (two 32-wide fragment sets + one 16-wide set in FLB_NK16 slots), filled by global_load_lds_dwordx4 one
// step ahead (one barrier per step), plus the wave-private transpose tiles of the weight-gradient outer products.  Work split:
// attn_flash_common.h (fl_plan) with 4 tiles per workgroup; partial D / dQ / dV of a segment go to slot workspaces summed in fixed order by the
// small merge kernels below (bitwise reproducible, no atomics).
//
// What was tried on these loops and lost (exponentials inside an inline-assembly mix block, the reduce-scatter from DPP builtins,
// sched_group_barrier interleave requests, the back half first in source order, a query-major pass 1): profiles/HISTORY_r05.md.
// Timing ablations (-DSPE_ABLATE builds only, tools/ab.py): FLB_DBG_SAMETILE (every workgroup streams tile 0: all L2 hits), FLB_DBG_NOST (no dS store).
#include "attn_flash_common.h"
#include <type_traits>

#define FLB_NW 4                         // waves per workgroup = resident tiles per workgroup (one wave per SIMD)
#define FLB_HB 4                         // heads per operand batch of the score products
#define FLB_GWR 144                      // row pitch of the weight-gradient transpose tiles (bytes)
#define FLB_NK16 3                       // slots of the 16-wide tiles: tile i is read by the back half of step i, one step after its stage
#define FLB_RSFUSE 1                     // key-major kernel: the D reduce-scatter of tile i inside the score-product blocks of tile i + 1 (head dim 33 .. 48, 8 heads)
#define FLB_DQAHEAD 2                    // heads the 16-wide operand loads of the accumulator products run ahead of their matrix instructions
#define FLB_SB_NOMEM 0x00F               // sched_barrier mask: ALU / VALU / SALU / MFMA may cross, memory instructions (LDS reads) may not
#define FLB_PHASE() __builtin_amdgcn_sched_barrier(0)
#ifndef SPE_ABLATE
#undef FLB_DBG_SAMETILE
#undef FLB_DBG_NOST
#undef FLB_DBG_STAMP
#endif
// FLB_DBG_STAMP (timing experiment, query-major kernel): s_memtime at the boundaries of a pipelined step's scheduling regions, summed over the steps of
// workgroup 0 / wave 0 and left in the first bytes of the dS tensor (tools/debug/attn_time.py --stamps; profiles/r06_bwdq_stamps.txt)
#ifdef FLB_DBG_STAMP
#define FLB_STAMP(k) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[k]))     /* the wait: hipcc may copy the pair before it lands otherwise */
#else
#define FLB_STAMP(k)
#endif
#ifndef FLB_STAGGER
#define FLB_STAGGER 1                    // waves 2, 3 issue their share of the next tile's loads one scheduling region later than waves 0, 1
#endif

struct FlashBwdArgs {
    const unsigned char* Qf; const unsigned char* dOf;            // q-side fragment records (fp16 q * scale * log2 e ; bf16 dO)
    const unsigned char* Kf; const unsigned char* Vf; const unsigned char* K16;   // key-side records: fp16 k, bf16 v (32-wide layout), bf16 k (16-wide layout)
    const float* Wl; const float* Ww;
    const float* c0;                     // [B][Np][H]: bl log2(e) - m + log2(1 / l), rows >= N zero
    const float* Drows;                  // [B][Np][H] D of the key-major kernel, rows >= N zero
    int Np;
    float* ws_q;                         // partial dQ [B * nmaj][FL_MAXSLOT][FLB_NW][H][DT][64 lanes][4]
    float* ws_w;                         // weight-gradient partials [nwg * FLB_NW][2 * (H * H + H)], row = [dWl | dbl | dWw | dbw]: this kernel fills the first half, the key-major one the second
    unsigned short* dS;                  // bf16 blocks [B, H, nt, nt][64 lanes][4], lane = (query l & 15, keys 4 (l >> 4) + i)
    const unsigned* keepbits;            // dropout keep flags of spe_talking_flash_fwd [B][nt][nt][64]
    int B, N, nt, nmaj, spw; long total;
    float p_drop;
};

// ---- score products with the B operand in AccVGPRs.  One statement per head: FULL chained 32-deep steps into c, the 16-deep tail step into
// its OWN accumulator t (an accumulate chain never mixes two MFMA shapes: attn_stats.hip).  No wait states inside: the results are only
// read behind flb_fence*, which follows the whole batch.
// One statement per chunk of 4 heads: the FULL 32-deep steps of all four heads first, then (TAIL16) their 16-deep tail steps accumulating
// onto the same registers - a head's two shapes are then three instructions apart, the first has long left the pipe when the second reads
// its result (back to back, hipcc's placement in the round-2 kernels, the mixed-shape chain gave run-to-run different sums).  No wait states
// inside: the results are only read behind flb_fence4, which follows the statement.
#define FLB_SCORE_ASM(NAME, M32, M16)                                                                                                      \
    template <int FULL, bool TAIL16>                                                                                                       \
    __device__ __forceinline__ void NAME(const flu32x4_t (*k32)[FULL ? FULL : 1], const flu32x2_t* k16, const flu32x4_t (*q32)[FULL ? FULL : 1], \
                                         const flu32x2_t* q16, f32x4_t* c) {                                                               \
        if constexpr (FULL == 1 && TAIL16)                                                                                                 \
            asm(M32 " %0, %4, %12, 0\n\t" M32 " %1, %5, %13, 0\n\t" M32 " %2, %6, %14, 0\n\t" M32 " %3, %7, %15, 0\n\t"                 \
                M16 " %0, %8, %16, %0\n\t" M16 " %1, %9, %17, %1\n\t" M16 " %2, %10, %18, %2\n\t" M16 " %3, %11, %19, %3"                 \
                : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])                                                                       \
                : "v"(k32[0][0]), "v"(k32[1][0]), "v"(k32[2][0]), "v"(k32[3][0]), "v"(k16[0]), "v"(k16[1]), "v"(k16[2]), "v"(k16[3]),      \
                  "a"(q32[0][0]), "a"(q32[1][0]), "a"(q32[2][0]), "a"(q32[3][0]), "a"(q16[0]), "a"(q16[1]), "a"(q16[2]), "a"(q16[3]));     \
        else if constexpr (FULL == 2 && !TAIL16)                                                                                           \
            asm(M32 " %0, %4, %12, 0\n\t" M32 " %1, %5, %13, 0\n\t" M32 " %2, %6, %14, 0\n\t" M32 " %3, %7, %15, 0\n\t"                 \
                M32 " %0, %8, %16, %0\n\t" M32 " %1, %9, %17, %1\n\t" M32 " %2, %10, %18, %2\n\t" M32 " %3, %11, %19, %3"                 \
                : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])                                                                       \
                : "v"(k32[0][0]), "v"(k32[1][0]), "v"(k32[2][0]), "v"(k32[3][0]), "v"(k32[0][1]), "v"(k32[1][1]), "v"(k32[2][1]), "v"(k32[3][1]), \
                  "a"(q32[0][0]), "a"(q32[1][0]), "a"(q32[2][0]), "a"(q32[3][0]), "a"(q32[0][1]), "a"(q32[1][1]), "a"(q32[2][1]), "a"(q32[3][1])); \
        else if constexpr (FULL == 1 && !TAIL16)                                                                                           \
            asm(M32 " %0, %4, %8, 0\n\t" M32 " %1, %5, %9, 0\n\t" M32 " %2, %6, %10, 0\n\t" M32 " %3, %7, %11, 0"                        \
                : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])                                                                       \
                : "v"(k32[0][0]), "v"(k32[1][0]), "v"(k32[2][0]), "v"(k32[3][0]), "a"(q32[0][0]), "a"(q32[1][0]), "a"(q32[2][0]), "a"(q32[3][0])); \
        else                                                                                                                               \
            asm(M16 " %0, %4, %8, 0\n\t" M16 " %1, %5, %9, 0\n\t" M16 " %2, %6, %10, 0\n\t" M16 " %3, %7, %11, 0"                        \
                : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])                                                                       \
                : "v"(k16[0]), "v"(k16[1]), "v"(k16[2]), "v"(k16[3]), "a"(q16[0]), "a"(q16[1]), "a"(q16[2]), "a"(q16[3]));                 \
    }
FLB_SCORE_ASM(flb_score_f16, "v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x16_f16")
FLB_SCORE_ASM(flb_score_bf16, "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x16_bf16")

// results of the matrix instructions issued above become readable: 13 wait states behind the last one (8-pass instruction -> any reader),
// tied to the registers so that no consumer is scheduled in front of it
__device__ __forceinline__ void flb_fence4(f32x4_t& a, f32x4_t& b, f32x4_t& c, f32x4_t& d) {
    asm("s_nop 7\n\ts_nop 4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// dQ^T[d][q] += K^T[d][key] dS^T[key][q] for one head: DT accumulate instructions on AccVGPR accumulators; s_nop 1: pk was just written
// by the vector pipe
template <int DT>
__device__ __forceinline__ void flb_dq_mfma(f32x4_t* acc, const fls16x4_t* ka, fls16x4_t pk) {
    if constexpr (DT == 1)
        asm("s_nop 1\n\tv_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+a"(acc[0]) : "v"(ka[0]), "v"(pk));
    else if constexpr (DT == 2)
        asm("s_nop 1\n\tv_mfma_f32_16x16x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_16x16x16_bf16 %1, %3, %4, %1"
                     : "+a"(acc[0]), "+a"(acc[1]) : "v"(ka[0]), "v"(ka[1]), "v"(pk));
    else if constexpr (DT == 3)
        asm("s_nop 1\n\tv_mfma_f32_16x16x16_bf16 %0, %3, %6, %0\n\tv_mfma_f32_16x16x16_bf16 %1, %4, %6, %1\n\tv_mfma_f32_16x16x16_bf16 %2, %5, %6, %2"
                     : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]) : "v"(ka[0]), "v"(ka[1]), "v"(ka[2]), "v"(pk));
    else
        asm("s_nop 1\n\tv_mfma_f32_16x16x16_bf16 %0, %4, %8, %0\n\tv_mfma_f32_16x16x16_bf16 %1, %5, %8, %1\n\t"
                     "v_mfma_f32_16x16x16_bf16 %2, %6, %8, %2\n\tv_mfma_f32_16x16x16_bf16 %3, %7, %8, %3"
                     : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]) : "v"(ka[0]), "v"(ka[1]), "v"(ka[2]), "v"(ka[3]), "v"(pk));
}
// the accumulators become readable (segment end)
__device__ __forceinline__ void flb_acc_fence(f32x4_t& a) { asm volatile("s_nop 7\n\ts_nop 4" : "+a"(a)); }

template <int H, int DSTEPS, bool TAIL16, bool DROP>
__global__ __launch_bounds__(64 * FLB_NW, 1) void talking_bwdq_kernel(FlashBwdArgs a) {
    constexpr int FULL = DSTEPS - (TAIL16 ? 1 : 0), DT = 2 * FULL + (TAIL16 ? 1 : 0), REC = DT * 512;
    constexpr int NW = FLB_NW, HB = (H >= FLB_HB) ? FLB_HB : H;
    constexpr int TILEB = H * REC;                  // one operand, one 16-row tile, all heads
    constexpr int KVB = 2 * TILEB;                  // a K / V stage: K fragments, V fragments
    constexpr int NK16 = FLB_NK16;
    constexpr int F1 = FULL ? FULL : 1;
    // LDS: [K / V stage 0][K / V stage 1][FLB_NK16 slots of K in the 16-wide layout][constants 512 B][NW x 3 transpose tiles]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const int nt = a.nt, N = a.N;

    // ---- mixing weights as matrix-instruction operands (attn_flash_common.h)
    float Al4[H / 4][H];                            // S' = Wl S (fp32, 4x4x1)
    fl_mixA_f32<H, false>(a.Wl, lane, Al4);
    fls16x4_t Awt[H / 4][H / 4];                    // dP = Ww^T dP' (bf16, 4x4x4)
    fl_mixA_16<H, true, false>(a.Ww, lane, 1.0f, Awt);
    fls16x4_t Alt[H / 4][H / 4];                    // dS = Wl^T dS' (bf16, 4x4x4)
    fl_mixA_16<H, true, false>(a.Wl, lane, 1.0f, Alt);

    // ---- weight-gradient outer product on the matrix pipe through a wave-private LDS transpose: dW[g][h] = sum over (query, key) positions of x_g y_h is
    // a contraction over POSITIONS, i.e. D[m = g][n = h] += A[g][pos] B[pos][h] with 32 positions per v_mfma_f32_16x16x32_bf16 (round 6; rounds 4-5 issued two 16-deep instructions per packet pair, each as dear as this one).  A lane owns ONE query and 4
    // keys for all heads, the operands want one HEAD per lane (m = lane & 15) and 4 keys of query t for the t-th instruction: a (query x head) transpose inside
    // each 16-lane group.  Rows m >= H read a block of zeros (the key-major kernel: row m = H of the B operand a block of ones - column H of D is the bias
    // gradient).  Here: X = dS', Y = S -> dWl (dbl stays an fp32 vector sum) ; key-major kernel: X = dP', Y = P -> dWw, dbw.
    // Transpose tile: rows [key group 4][head H] of 16 packets (queries) x 8 B, row pitch FLB_GWR = 144 B - with the natural 128 B the 16-B
    // reads of 8 heads fall on 2 bank groups (4-way conflicts: SQ_LDS_BANK_CONFLICT was half of the LDS cycles of the round-3 passes); 144 B
    // spreads the 8 rows of a lane group over distinct banks, and the constant blocks sit on banks no row uses.  The operand the FRONT half
    // of a tile writes (here Y, in the key-major kernel X) is double-buffered by tile parity: the front half of tile j + 1 runs beside the back half of tile j.
    constexpr int GWR = FLB_GWR, GWT = 4 * H * GWR;
    unsigned char* gconst = smem + 2 * KVB + NK16 * TILEB;                  // 512 B: zeros at + 48 (128 B), bf16 ones at + 208 (128 B)
    constexpr int KBOFF = 2 * KVB + NK16 * TILEB + 512 + NW * 3 * GWT;     // dropout: 2 x 1 KB of keep-flag records (the four waves' 256 B of a key tile each), behind the transpose tiles
    unsigned char* sgw = gconst + 512 + wave * (3 * GWT);                   // [front operand, parity 0][front operand, parity 1][back operand]
    if (threadIdx.x < 64) reinterpret_cast<uint2*>(gconst)[threadIdx.x] = (threadIdx.x >= 26 && threadIdx.x < 42) ? make_uint2(0x3F803F80u, 0x3F803F80u) : make_uint2(0u, 0u);
    const int gm = lane & 15, gk = lane >> 4;
    unsigned char* gw_wr = sgw + (gk * H) * GWR + gm * 8;                   // + tile * GWT + h * GWR: packet of head h, query gm, key group gk
    const unsigned gw_rd = (unsigned)((gk * H + gm) * GWR);                 // + tile * GWT: 16 packets (queries 0..15) of head gm
    const unsigned char* gw_zero = gconst + 48;
    f32x4_t gwacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gwacc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float gb[H];                                    // dbl
#pragma unroll
    for (int g = 0; g < H; ++g) gb[g] = 0.f;

    constexpr int NP = TILEB / 1024, NPW = (NP + NW - 1) / NW;     // 1-KB pieces of an operand tile, pieces per wave
    unsigned voff[NPW];                                            // byte offset of this lane's 16 B of piece i * NW + wave inside a (b, tile) image
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int o = (i * NW + wave) * 1024 + lane * 16;
        voff[i] = (unsigned)((o / REC) * nt * REC + o % REC);
    }
#ifdef FLB_DBG_STAMP
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0}, stacc[5] = {0, 0, 0, 0, 0}, stn = 0;
#endif
    const float keep_inv = DROP ? 1.0f / (1.0f - a.p_drop) : 1.0f;

    const long s_begin = (long)blockIdx.x * a.spw;
    long s_end = s_begin + a.spw; if (s_end > a.total) s_end = a.total;
    long s = s_begin;
    while (s < s_end) {
        const int bm = (int)(s / nt), kt0 = (int)(s % nt);
        int seg = nt - kt0; if (seg > s_end - s) seg = (int)(s_end - s);
        const int b = bm / a.nmaj, mj = bm % a.nmaj;
        const int qt = mj * NW + wave;                          // this wave's q-tile (wave-uniform)
        const bool wvalid = qt < nt;
        const int qtc = wvalid ? qt : nt - 1;
        const int q = qtc * 16 + (lane & 15);
        // keep flags (dropout): the (q-tile, key tile) records of the four waves are nt * 256 B apart; one LDS-DMA instruction (wave 1) gathers them - lanes
        // 16 w .. 16 w + 15 fetch wave w's 256 B - beside the tile's fragments, so no load the compiler can see (and would wait for with vmcnt(0), draining the
        // tile loads in flight) is left in the loop
        unsigned kb_voff = 0u;
        if constexpr (DROP) {
            const int w = lane >> 4, qw = (mj * NW + w < nt) ? w : 0;
            kb_voff = (unsigned)(qw * nt * 256 + (lane & 15) * 16);
        }

        // ---- this wave's Q and dO records -> registers (AccVGPR operands of the score products)
        flu32x4_t qa[H][F1], da[H][F1];
        flu32x2_t qta[H], dta[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const unsigned char* qb = a.Qf + (((long)b * H + h) * nt + qtc) * REC;
            const unsigned char* db = a.dOf + (((long)b * H + h) * nt + qtc) * REC;
#pragma unroll
            for (int st = 0; st < FULL; ++st) {
                qa[h][st] = *reinterpret_cast<const flu32x4_t*>(qb + st * 1024 + lane * 16);
                da[h][st] = *reinterpret_cast<const flu32x4_t*>(db + st * 1024 + lane * 16);
            }
            if constexpr (TAIL16) {
                qta[h] = *reinterpret_cast<const flu32x2_t*>(qb + FULL * 1024 + lane * 8);
                dta[h] = *reinterpret_cast<const flu32x2_t*>(db + FULL * 1024 + lane * 8);
            } else { qta[h] = (flu32x2_t){0u, 0u}; dta[h] = (flu32x2_t){0u, 0u}; }
        }
        // ---- row constants of this lane's query
        f32x4_t c0v[H / 4], Dn[H / 4];                 // Dn = -D
#pragma unroll
        for (int gh = 0; gh < H / 4; ++gh) {
            c0v[gh] = *reinterpret_cast<const f32x4_t*>(a.c0 + ((long)b * a.Np + q) * H + 4 * gh);
            Dn[gh] = -*reinterpret_cast<const f32x4_t*>(a.Drows + ((long)b * a.Np + q) * H + 4 * gh);
        }
        f32x4_t dQ[H][DT];
#pragma unroll
        for (int g = 0; g < H; ++g)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dQ[g][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        // the streamed operand tiles of key tile kt (local index i): K, V fragments -> K / V stage i & 1, 16-wide K -> slot i % NK16
        auto issue_tiles = [&](int i) {
#ifdef FLB_DBG_SAMETILE
            const int kt = 0;                 // timing experiment: every workgroup streams the same tile (all L2 hits)
#else
            const int kt = kt0 + i;
#endif
#pragma unroll
            for (int op = 0; op < 3; ++op) {
                const unsigned char* base = (op == 0) ? a.Kf : ((op == 1) ? a.Vf : a.K16);
                const unsigned char* tb = base + ((long)b * H * nt + kt) * REC;
                const unsigned dst = (op < 2) ? lds0 + (i & 1) * KVB + op * TILEB : lds0 + 2 * KVB + (i % NK16) * TILEB;
                if constexpr (NP % NW == 0 && NPW <= 4) fl_glds16_run<NPW, NW * 1024>(tb, voff, dst + wave * 1024);
                else {
#pragma unroll
                    for (int ii = 0; ii < NPW; ++ii) {
                        const int p = ii * NW + wave;
                        if (NP % NW != 0 && p >= NP) break;
                        fl_glds16_s(tb, voff[ii], dst + p * 1024);
                    }
                }
            }
            if constexpr (DROP)
                if (wave == 1) fl_glds16_s(a.keepbits + (((long)b * nt + mj * NW) * nt + kt) * 64, kb_voff, lds0 + KBOFF + (i & 1) * 1024);
        };

        // ---- FRONT half of key tile i (matrix-heavy), in chunks of FLB_HB heads:
        //   front_k: S^T = K Q^T (lane = (query l & 15, keys 4 (l >> 4) + r)) of the chunk's heads, S' += Wl S on the fly (head-outer,
        //            like the flash forward) -> sp (the exponent of P; initialised with the row constants: MASK - the ragged last tile -
        //            gives keys >= N the exponent -inf, i.e. P = 0)
        //   front_v: dP'^T = V dO^T (bf16) of the chunk's heads, dropout, dP += Ww^T dP' -> dp
        // operand fragments of a chunk: LDS -> registers (issued a whole chunk ahead of their use in the pipelined step: a single wave per SIMD
        // has nobody to cover an exposed s_waitcnt)
        struct Frags { flu32x4_t f[HB][F1]; flu32x2_t t[HB]; };
        auto load_frags = [&](const unsigned char* tile, int h0, Frags& o) {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const unsigned char* r = tile + (h0 + hb) * REC;
#pragma unroll
                for (int st = 0; st < FULL; ++st) o.f[hb][st] = *reinterpret_cast<const flu32x4_t*>(r + st * 1024 + lane * 16);
                if constexpr (TAIL16) o.t[hb] = *reinterpret_cast<const flu32x2_t*>(r + FULL * 1024 + lane * 8);
                else o.t[hb] = (flu32x2_t){0u, 0u};
            }
        };
        auto load_k = [&](int i, int h0, Frags& o) { load_frags(smem + (i & 1) * KVB, h0, o); };
        auto load_v = [&](int i, int g0, Frags& o) { load_frags(smem + (i & 1) * KVB + TILEB, g0, o); };
        auto front_k = [&](int i, auto h0_c, const Frags& kfr, f32x4_t (&sp)[4][H / 4], auto mask_c) {
            constexpr bool MASK = decltype(mask_c)::value;
            constexpr int h0 = decltype(h0_c)::value;
            unsigned char* gwf = gw_wr + (i & 1) * GWT;         // this tile's front operand of the outer product
            if constexpr (h0 == 0) {
                const int key0 = (kt0 + i) * 16 + 4 * (lane >> 4);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int gh = 0; gh < H / 4; ++gh) {
                        if constexpr (MASK) {
                            const bool kv = key0 + r < N;
#pragma unroll
                            for (int k = 0; k < 4; ++k) sp[r][gh][k] = kv ? c0v[gh][k] : -INFINITY;
                        } else sp[r][gh] = c0v[gh];
                    }
            }
            f32x4_t c[HB];
            static_assert(HB == 4, "chunk of 4 heads");
            flb_score_f16<FULL, TAIL16>(kfr.f, kfr.t, &qa[h0], &qta[h0], c);
            flb_fence4(c[0], c[1], c[2], c[3]);
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const f32x4_t cs = c[hb];
                // bf16(S) of this head is the Y operand of the dWl outer product - straight into the wave's transpose tile
                *reinterpret_cast<fls16x4_t*>(gwf + (h0 + hb) * GWR) = fl_pack4<false>(cs[0], cs[1], cs[2], cs[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int gh = 0; gh < H / 4; ++gh) sp[r][gh] = __builtin_amdgcn_mfma_f32_4x4x1f32(Al4[gh][h0 + hb], cs[r], sp[r][gh], 0, 0, 0);
            }
        };
        auto front_v = [&](int i, auto g0_c, const Frags& vfr, f32x4_t (&dp)[4][H / 4], uint32_t kb) {
            constexpr int g0 = decltype(g0_c)::value;
            if constexpr (g0 == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int gh = 0; gh < H / 4; ++gh) dp[r][gh] = Dn[gh];         // the mix starts from -D, its result is dP - D
            }
            f32x4_t es[HB];
            flb_score_bf16<FULL, TAIL16>(vfr.f, vfr.t, &da[g0], &dta[g0], es);
            flb_fence4(es[0], es[1], es[2], es[3]);
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                if constexpr (DROP) {       // bit hp * 8 + 2 r + e: key r of the lane's group, head 2 hp + e
                    const int g = g0 + hb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) es[hb][r] *= ((kb >> ((g >> 1) * 8 + 2 * r + (g & 1))) & 1u) ? keep_inv : 0.f;
                }
            }
#pragma unroll
            for (int hq = 0; hq < HB / 4; ++hq)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const fls16x4_t bv = fl_pack4<false>(es[4 * hq][r], es[4 * hq + 1][r], es[4 * hq + 2][r], es[4 * hq + 3][r]);
#pragma unroll
                    for (int gh = 0; gh < H / 4; ++gh) dp[r][gh] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(Awt[gh][g0 / 4 + hq], bv, dp[r][gh], 0, 0, 0);
                }
        };
        auto load_kb = [&](int i) -> uint32_t {
            if constexpr (DROP) return *reinterpret_cast<const uint32_t*>(smem + KBOFF + (i & 1) * 1024 + wave * 256 + lane * 4);
            else return 0u;
        };

        // ---- BACK half of key tile i (vector-heavy), in four chunks:
        //   back_exp: P = exp2(sp)
        //   back_ds : dS' = P (dP - D), dbl, bf16(dS') -> transpose tile
        //   back_gw : the outer product dWl from the transpose tiles ; dS = Wl^T dS' -> ds
        //   back_dq : bf16 block of dS -> HBM, dQ^T += K^T dS^T
        auto back_exp = [&](f32x4_t (&sp)[4][H / 4]) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int gh = 0; gh < H / 4; ++gh)
#pragma unroll
                    for (int k = 0; k < 4; ++k) sp[r][gh][k] = fl_exp2(sp[r][gh][k]);
        };
        auto back_ds = [&](f32x4_t (&sp)[4][H / 4], f32x4_t (&dp)[4][H / 4]) {
            unsigned char* gwb = gw_wr + 2 * GWT;               // the back operand of the outer product
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int gh = 0; gh < H / 4; ++gh) sp[r][gh] = sp[r][gh] * dp[r][gh];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int g = 0; g < H; ++g) gb[g] += sp[r][g >> 2][g & 3];
#pragma unroll
            for (int g = 0; g < H; ++g)
                *reinterpret_cast<fls16x4_t*>(gwb + g * GWR) =
                    fl_pack4<false>(sp[0][g >> 2][g & 3], sp[1][g >> 2][g & 3], sp[2][g >> 2][g & 3], sp[3][g >> 2][g & 3]);
        };
        auto back_gw = [&](int i, f32x4_t (&sp)[4][H / 4], f32x4_t (&ds)[4][H / 4]) {
            {
                const unsigned rdf = gw_rd + (i & 1) * GWT, rdb = gw_rd + 2 * GWT;
                // lanes >= H of a 16-lane group read zeros
                const unsigned char* gw_xrd = (gm < H) ? sgw + rdb : gw_zero;
                const unsigned char* gw_yrd = (gm < H) ? sgw + rdf : gw_zero;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // wave-private tile: the other lanes' packets are read next
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const flu32x4_t xa = *reinterpret_cast<const flu32x4_t*>(gw_xrd + c * 16);
                    const flu32x4_t yb = *reinterpret_cast<const flu32x4_t*>(gw_yrd + c * 16);
                    // one 32-deep instruction per 16-B packet pair: a contraction over positions does not care which 8 positions a lane group brings
                    // (the 16-deep form costs the same issue slot for half the positions: profiles/r05_mfma_form.txt)
                    gwacc[c & 3] = fl_mfma32<false>(xa, yb, gwacc[c & 3]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the back tile is rewritten by the next key tile
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x[H];
#pragma unroll
                for (int g = 0; g < H; ++g) x[g] = sp[r][g >> 2][g & 3];
                fl_mix_16<H, false>(x, Alt, nullptr, ds[r]);
            }
        };
        auto back_dq = [&](int i, f32x4_t (&ds)[4][H / 4]) {
            {
                const int kt = kt0 + i;
                const unsigned char* sK16 = smem + 2 * KVB + (i % NK16) * TILEB;
                fls16x4_t ka[H][DT];                // requested FLB_DQAHEAD heads ahead of the matrix instructions that consume them
#pragma unroll
                for (int h = 0; h < ((FLB_DQAHEAD < H) ? FLB_DQAHEAD : H); ++h)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) ka[h][dt] = *reinterpret_cast<const fls16x4_t*>(sK16 + h * REC + dt * 512 + lane * 8);
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    if (h + FLB_DQAHEAD < H) {
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) ka[h + FLB_DQAHEAD][dt] = *reinterpret_cast<const fls16x4_t*>(sK16 + (h + FLB_DQAHEAD) * REC + dt * 512 + lane * 8);
                        __builtin_amdgcn_sched_barrier(FLB_SB_NOMEM);        // the loads stay ahead: only register-only instructions may cross
                    }
                    const fls16x4_t pk = fl_pack4<false>(ds[0][h >> 2][h & 3], ds[1][h >> 2][h & 3], ds[2][h >> 2][h & 3], ds[3][h >> 2][h & 3]);
#ifndef FLB_DBG_NOST
                    __builtin_nontemporal_store(__builtin_bit_cast(flu32x2_t, pk),
                                                reinterpret_cast<flu32x2_t*>(a.dS + (((((long)b * H + h) * nt + qt) * nt + kt) * 64 + lane) * 4));
#endif
                    flb_dq_mfma<DT>(dQ[h], ka[h], pk);
                }
            }
        };
        // the halves as a whole (pipeline prologue / epilogue, the ragged tile)
        auto front = [&](int i, f32x4_t (&sp)[4][H / 4], f32x4_t (&dp)[4][H / 4], auto mask_c) {
            const uint32_t kb = load_kb(i);
            Frags fa, fb;
            load_k(i, 0, fa);
            if constexpr (H > HB) load_k(i, HB, fb);
            front_k(i, std::integral_constant<int, 0>{}, fa, sp, mask_c);
            load_v(i, 0, fa);
            if constexpr (H > HB) front_k(i, std::integral_constant<int, HB>{}, fb, sp, mask_c);
            if constexpr (H > HB) load_v(i, HB, fb);
            front_v(i, std::integral_constant<int, 0>{}, fa, dp, kb);
            if constexpr (H > HB) front_v(i, std::integral_constant<int, HB>{}, fb, dp, kb);
        };
        auto back = [&](int i, f32x4_t (&sp)[4][H / 4], f32x4_t (&dp)[4][H / 4]) {
            f32x4_t ds[4][H / 4];
            back_exp(sp);
            back_ds(sp, dp);
            back_gw(i, sp, ds);
            back_dq(i, ds);
        };
        // One pipelined step: the front half of tile i + 1 beside the back half of tile i, chunk by chunk - every scheduling region holds one
        // matrix-heavy and one vector-heavy chunk of INDEPENDENT work, which the scheduler interleaves (a single wave per SIMD has no partner
        // wave to fill its stalls).  The LDS operands of a chunk are requested at the top of the PREVIOUS region (fr0 / fr1 alternate) and
        // pinned there by a scheduling barrier only register-only instructions may cross; the full barriers bound the live ranges.
        // fr0 arrives holding the K fragments of tile i + 1's first chunk (requested by the previous step or the prologue).
        auto step = [&](int i, f32x4_t (&sp)[4][H / 4], f32x4_t (&dp)[4][H / 4], f32x4_t (&spn)[4][H / 4], f32x4_t (&dpn)[4][H / 4], Frags& fr0, Frags& fr1,
                        bool more) {
            f32x4_t ds[4][H / 4];
            const uint32_t kb = load_kb(i + 1);
            if constexpr (H > HB) load_k(i + 1, HB, fr1); else load_v(i + 1, 0, fr1);
            __builtin_amdgcn_sched_barrier(FLB_SB_NOMEM);
            front_k(i + 1, std::integral_constant<int, 0>{}, fr0, spn, std::false_type{});
            back_exp(sp);
            FLB_PHASE();
            FLB_STAMP(2);
            // the late half of the workgroup's tile loads (see admit)
            if (FLB_STAGGER && more && wave >= NW / 2) issue_tiles(i + 2);
            if constexpr (H > HB) {
                load_v(i + 1, 0, fr0);
                __builtin_amdgcn_sched_barrier(FLB_SB_NOMEM);
                front_k(i + 1, std::integral_constant<int, HB>{}, fr1, spn, std::false_type{});
            }
            back_ds(sp, dp);
            FLB_PHASE();
            FLB_STAMP(3);
            if constexpr (H > HB) {
                load_v(i + 1, HB, fr1);
                __builtin_amdgcn_sched_barrier(FLB_SB_NOMEM);
                front_v(i + 1, std::integral_constant<int, 0>{}, fr0, dpn, kb);
            } else front_v(i + 1, std::integral_constant<int, 0>{}, fr1, dpn, kb);
            back_gw(i, sp, ds);
            FLB_PHASE();
            FLB_STAMP(4);
            if constexpr (H > HB) front_v(i + 1, std::integral_constant<int, HB>{}, fr1, dpn, kb);
            back_dq(i, ds);
        };

        // step i + 1 is admitted: this wave's pieces have landed, a barrier (everybody's have, and everybody is done with the buffers refilled
        // next), then the tiles of step i + 2 go out with a whole step to land
        // The 36 LDS-DMA instructions of a tile (9 per wave) are worth 576 cycles of the CU's one vector-memory pipeline (1 KB each at 64 B / clk); issued by all
        // four waves at once they hold every wave for that long (measured: 570 of a step's 5100 cycles - profiles/r06_bwdq_stamps.txt - a single wave per SIMD issues
        // nothing else meanwhile), a wave issuing alone or in a pair pays ~39 cycles apiece for its own 9.  `late`: inside the pipelined loop waves 2, 3 therefore
        // issue theirs at the first region boundary of the step (step()), while waves 0, 1 compute.
        auto admit = [&](int nxt, bool late) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (nxt + 1 < seg && !(late && wave >= NW / 2)) issue_tiles(nxt + 1);
        };

        // Software pipeline over the segment's key tiles: the matrix-heavy front half of tile i + 1 runs in ONE scheduling region with the
        // vector-heavy back half of tile i - a single wave per SIMD has no partner wave to fill its stalls, so the two independent instruction
        // streams are interleaved inside the wave.  The ragged last key tile of an image runs the masked front instance after the loop: one
        // instance per loop keeps the accumulators in place (with both instances inside one loop hipcc copied all 96 around every step).
        const int nfull = ((N & 15) != 0 && kt0 + seg == nt) ? seg - 1 : seg;
        f32x4_t spA[4][H / 4], dpA[4][H / 4], spB[4][H / 4], dpB[4][H / 4];
        Frags frA, frB;
        issue_tiles(0);
        admit(0, false);
        if (nfull > 0) {
            if (wvalid) front(0, spA, dpA, std::false_type{});
            for (int i = 0; i + 1 < nfull; ++i) {
                FLB_STAMP(0);
                admit(i + 1, FLB_STAGGER && wvalid);
                FLB_STAMP(1);
                if (wvalid) {
                    load_k(i + 1, 0, frA);       // (the stage was admitted above)
                    step(i, spA, dpA, spB, dpB, frA, frB, i + 2 < seg);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int gh = 0; gh < H / 4; ++gh) { spA[r][gh] = spB[r][gh]; dpA[r][gh] = dpB[r][gh]; }
                }
#ifdef FLB_DBG_STAMP
                FLB_STAMP(5);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(stamp[0]), "+s"(stamp[1]), "+s"(stamp[2]), "+s"(stamp[3]), "+s"(stamp[4]), "+s"(stamp[5]));
                for (int k = 0; k < 5; ++k) stacc[k] += stamp[k + 1] - stamp[k];
                ++stn;
#endif
            }
            if (nfull < seg) admit(nfull, false);
            if (wvalid) back(nfull - 1, spA, dpA);
        }
        if (nfull < seg) {
            if (wvalid) { front(nfull, spA, dpA, std::true_type{}); back(nfull, spA, dpA); }
        }
        __builtin_amdgcn_s_barrier();              // the last buffers have been read by everybody: the next segment may refill them

        // ---- partial results of this segment -> the major's slot
        const int first_wg = (int)(((long)bm * nt) / a.spw);
        const int slot = (int)blockIdx.x - first_wg;
        {
#pragma unroll
            for (int g = 0; g < H; ++g)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) flb_acc_fence(dQ[g][dt]);
            if (wvalid) {
                float* dst = a.ws_q + (((long)bm * FL_MAXSLOT + slot) * NW + wave) * (long)(H * DT * 256);
#pragma unroll
                for (int g = 0; g < H; ++g)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4_t*>(dst + (g * DT + dt) * 256 + lane * 4) = dQ[g][dt];
            }
        }
        s += seg;
    }

#ifdef FLB_DBG_STAMP
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.dS);
        for (int k = 0; k < 5; ++k) o[k] = stacc[k];
        o[5] = stn;
    }
#endif
    // ---- weight-gradient partials of this wave -> its row of ws_w
    {
        constexpr int NWG = 2 * (H * H + H);
        float* row = a.ws_w + ((long)blockIdx.x * NW + wave) * NWG;
        // D[m = g][n]: lane holds rows 4 (lane >> 4) + r of column lane & 15; columns < H = dWl[g][h]
        const f32x4_t dsum = (gwacc[0] + gwacc[1]) + (gwacc[2] + gwacc[3]);
        const int nn = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int g = 4 * (lane >> 4) + r;
            // accumulated dS' . (log2(e) S)^T: the weight gradient carries ln 2, the bias gradient does not
            if (g < H && nn < H) row[g * H + nn] = FL_LN2 * dsum[r];
        }
#pragma unroll
        for (int g = 0; g < H; ++g) {
            const float v = spe_wave_sum(gb[g]);
            if (lane == 0) row[H * H + g] = v;
        }
    }
}

// dq[b, q, g, d] = scale * sum over the slots (element strides ob, on, oh; fp32 and / or bf16 with the same addressing).  One thread per
// float4 of the fragment-ordered workspace; fixed summation order.
__global__ __launch_bounds__(256) void bwdq_dq_merge_kernel(const float* __restrict__ ws, float* __restrict__ O, unsigned short* __restrict__ O16,
                                                            long ob, long on, long oh, int B, int H, int N, int nt, int dh, int DT, int nmaj, int spw,
                                                            float scale, long nvec) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int lane = (int)(i & 63);
    long r = i >> 6;
    const int dt = (int)(r % DT); r /= DT;
    const int g = (int)(r % H); r /= H;
    const int wave = (int)(r % FLB_NW); r /= FLB_NW;
    const long bm = r;
    const int b = (int)(bm / nmaj), mj = (int)(bm % nmaj);
    const int qt = mj * FLB_NW + wave;
    const int q = qt * 16 + (lane & 15), d = dt * 16 + 4 * (lane >> 4);
    if (qt >= nt || q >= N || d >= dh) return;
    const int first_wg = (int)((bm * nt) / spw), last_wg = (int)(((bm + 1) * nt - 1) / spw);
    const long slot_stride = (long)FLB_NW * H * DT * 256;
    const float* src = ws + bm * FL_MAXSLOT * slot_stride + ((long)wave * H + g) * (long)(DT * 256) + dt * 256 + lane * 4;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl <= last_wg - first_wg; ++sl) acc += *reinterpret_cast<const f32x4_t*>(src + sl * slot_stride);
    acc *= scale;
    const long oi = (long)b * ob + (long)q * on + (long)g * oh + d;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (d + k >= dh) break;
        if (O) O[oi + k] = acc[k];
    }
    if (O16) {
        if (d + 3 < dh && ((ob | on | oh) & 3) == 0 && ((reinterpret_cast<uintptr_t>(O16) & 7) == 0)) {
            const unsigned short h0 = spe_f2bf(acc[0]), h1 = spe_f2bf(acc[1]), h2 = spe_f2bf(acc[2]), h3 = spe_f2bf(acc[3]);
            *reinterpret_cast<uint2*>(O16 + oi) = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { if (d + k >= dh) break; O16[oi + k] = spe_f2bf(acc[k]); }
        }
    }
}

// =====================================================================================================================================
// KEY-major pass (spe_talking_bwdk_pass1): backward pass 1 AND the dV pass in one walk.  A wave keeps one 16-KEY tile - its K (fp16) and V
// (bf16) fragment records as AccVGPR B operands, the 96 dV accumulators in AccVGPRs - and streams the q-tiles: Q, dO fragments (the score
// products' A operands), dO in the 16-wide layout (A operand of dV^T[d][key] += dO^T[d][q] P'd[q][key]) and the 512 B of row constants of
// the tile, all by LDS-DMA.  The score tile is S[q][key] (lane = (key l & 15, queries 4 (l >> 4) + r)): the transpose of the q-major
// passes', so that bf16(P'd) in the accumulator layout IS the B operand of the dV product.  Per tile:
//   S, S' (fp32 mix), P = exp2 ; dP' = dropout (dO V^T) ; dP = Ww^T dP' ; D[q, h'] = sum_key dP P: a 16-lane row reduction (4 DPP steps)
//   per (query row, head), summed over the workgroup's 4 key tiles through LDS one step later, one 512-B row block per (major, q-tile)
//   -> ws_d ; dWw / dbw outer products (the same transpose tiles as pass 1) ; P' = Ww P + bw in fp16 on P * 2^8 (the flash forward's
//   arithmetic), dropout, bf16, dV += P'd^T dO.
// No mask instance: a key >= N has K = V = 0 (zero-padded records), so dP' = dP = 0 there and its dV rows are dropped by the merge; a query
// >= N has Q = dO = 0 and c0 = 0 (finite P, zero dO).  S, S', P are recomputed once for D, dWw, dbw AND dV.
struct FlashBwdKArgs {
    const unsigned char* Qf; const unsigned char* dOf; const unsigned char* dO16;     // streamed: fp16 q * scale * log2 e, bf16 dO (32-wide), bf16 dO (16-wide)
    const unsigned char* Kf; const unsigned char* Vf;                                 // resident: fp16 k, bf16 v (32-wide records)
    const float* Wl; const float* Ww; const float* bw;
    const float* c0; int Np;             // [B][Np][H], rows >= N zero, Np >= 16 nt + 64 (whole 1-KB pieces are fetched)
    float* ws_d;                         // D summed over a major's key tiles [B * nmaj][Np][H]
    float* ws_v;                         // partial dV [B * nmaj][FL_MAXSLOT][FLB_NW][H][DT][64 lanes][4]
    float* ws_w;                         // weight-gradient partials, the [dWw | dbw] half of each row
    const unsigned* keepbits;
    int B, N, nt, nmaj, spw; long total;
    float p_drop;
};

// Row sums of a score tile in the key-major layout (lane = (key l & 15, queries 4 (l >> 4) + r)) for 4 heads: t[r][k] = the lane's term of
// (query row r, head k) -> o[k] = the sum over the 16 keys (the lanes of a DPP row) for query row (l & 15) >> 2, in every lane of that
// quad.  A reduce-scatter on the vector pipe, 32 DPP adds for 16 values (an all-reduce butterfly of every value costs 4 per value):
//   rows {0, 1} stay in lanes 0..7 of the DPP row, rows {2, 3} in lanes 8..15 (partner: lane + 8, bank-masked writes) ; then row p in the
//   even / odd bank of each half (partner: the mirrored lane of the other bank) ; then the quad's four lanes add up (xor 1, xor 2).
// Every instruction reads registers written at least three instructions earlier (a DPP read needs two wait states behind a vector write);
// the leading s_nop covers the products computed just before.
__device__ __forceinline__ void flb_rowsum4(const f32x4_t (&t)[4], f32x4_t& o) {
    float a0, a1, a2, a3, b0, b1, b2, b3;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %13, %13 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %14, %14 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %15, %15 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %20, %20 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %21, %21 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %22, %22 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %23, %23 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %16, %16 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %5, %17, %17 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %6, %18, %18 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %7, %19, %19 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %24, %24 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %5, %25, %25 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %6, %26, %26 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %7, %27, %27 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %8, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %9, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %10, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %11, %3, %3 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %8, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %9, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %10, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %11, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %9, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %10, %10, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %11, %11, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %9, %9, %9 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %10, %10, %10 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %11, %11, %11 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3),
          "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
        : "v"(t[0][0]), "v"(t[0][1]), "v"(t[0][2]), "v"(t[0][3]), "v"(t[1][0]), "v"(t[1][1]), "v"(t[1][2]), "v"(t[1][3]),
          "v"(t[2][0]), "v"(t[2][1]), "v"(t[2][2]), "v"(t[2][3]), "v"(t[3][0]), "v"(t[3][1]), "v"(t[3][2]), "v"(t[3][3]));
}

// The score products of a chunk of 4 heads at head dim 33 .. 48 (8 matrix instructions: flb_score_*<1, true>) with the row-sum reduce-scatter of 4 heads
// (flb_rowsum4: 32 DPP adds) in their shadow - four vector instructions behind every matrix instruction; a single wave per SIMD overlaps the two pipes only where
// the instruction stream alternates, and an inline-assembly block is one unit for the scheduler.
#define FLB_SCORE_RS_ASM(NAME, M32, M16)                                                                                                   \
    __device__ __forceinline__ void NAME(const flu32x4_t (*k32)[1], const flu32x2_t* k16, const flu32x4_t (*q32)[1], const flu32x2_t* q16, \
                                         f32x4_t* c, const f32x4_t (&t)[4], f32x4_t& o) {                                                  \
        float a0, a1, a2, a3, b0, b1, b2, b3;                                                                                              \
            asm("s_nop 1\n\t"   \
                M32 " %0, %16, %24, 0\n\t"   \
                "v_add_f32_dpp %4, %32, %32 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"   \
                "v_add_f32_dpp %5, %33, %33 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"   \
                "v_add_f32_dpp %6, %34, %34 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"   \
                "v_add_f32_dpp %7, %35, %35 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"   \
                M32 " %1, %17, %25, 0\n\t"   \
                "v_add_f32_dpp %4, %40, %40 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"   \
                "v_add_f32_dpp %5, %41, %41 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"   \
                "v_add_f32_dpp %6, %42, %42 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"   \
                "v_add_f32_dpp %7, %43, %43 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"   \
                M32 " %2, %18, %26, 0\n\t"   \
                "v_add_f32_dpp %8, %36, %36 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"   \
                "v_add_f32_dpp %9, %37, %37 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"   \
                "v_add_f32_dpp %10, %38, %38 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"   \
                "v_add_f32_dpp %11, %39, %39 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"   \
                M32 " %3, %19, %27, 0\n\t"   \
                "v_add_f32_dpp %8, %44, %44 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"   \
                "v_add_f32_dpp %9, %45, %45 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"   \
                "v_add_f32_dpp %10, %46, %46 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"   \
                "v_add_f32_dpp %11, %47, %47 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"   \
                M16 " %0, %20, %28, %0\n\t"   \
                "v_add_f32_dpp %12, %4, %4 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"   \
                "v_add_f32_dpp %13, %5, %5 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"   \
                "v_add_f32_dpp %14, %6, %6 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"   \
                "v_add_f32_dpp %15, %7, %7 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"   \
                M16 " %1, %21, %29, %1\n\t"   \
                "v_add_f32_dpp %12, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"   \
                "v_add_f32_dpp %13, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"   \
                "v_add_f32_dpp %14, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"   \
                "v_add_f32_dpp %15, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"   \
                M16 " %2, %22, %30, %2\n\t"   \
                "v_add_f32_dpp %12, %12, %12 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   \
                "v_add_f32_dpp %13, %13, %13 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   \
                "v_add_f32_dpp %14, %14, %14 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   \
                "v_add_f32_dpp %15, %15, %15 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   \
                M16 " %3, %23, %31, %3\n\t"   \
                "v_add_f32_dpp %12, %12, %12 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"   \
                "v_add_f32_dpp %13, %13, %13 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"   \
                "v_add_f32_dpp %14, %14, %14 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"   \
                "v_add_f32_dpp %15, %15, %15 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"   \
            : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2),  \
              "=&v"(b3), "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])                                                                 \
            : "v"(k32[0][0]), "v"(k32[1][0]), "v"(k32[2][0]), "v"(k32[3][0]), "v"(k16[0]), "v"(k16[1]), "v"(k16[2]), "v"(k16[3]),          \
              "a"(q32[0][0]), "a"(q32[1][0]), "a"(q32[2][0]), "a"(q32[3][0]), "a"(q16[0]), "a"(q16[1]), "a"(q16[2]), "a"(q16[3]),          \
              "v"(t[0][0]), "v"(t[0][1]), "v"(t[0][2]), "v"(t[0][3]), "v"(t[1][0]), "v"(t[1][1]), "v"(t[1][2]), "v"(t[1][3]),              \
              "v"(t[2][0]), "v"(t[2][1]), "v"(t[2][2]), "v"(t[2][3]), "v"(t[3][0]), "v"(t[3][1]), "v"(t[3][2]), "v"(t[3][3]));             \
    }
FLB_SCORE_RS_ASM(flb_score_rs_f16, "v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x16_f16")
FLB_SCORE_RS_ASM(flb_score_rs_bf16, "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x16_bf16")



template <int H, int DSTEPS, bool TAIL16, bool DROP>
__global__ __launch_bounds__(64 * FLB_NW, 1) void talking_bwdk_kernel(FlashBwdKArgs a) {
    constexpr int FULL = DSTEPS - (TAIL16 ? 1 : 0), DT = 2 * FULL + (TAIL16 ? 1 : 0), REC = DT * 512;
    constexpr int NW = FLB_NW, HB = (H >= FLB_HB) ? FLB_HB : H;
    constexpr int TILEB = H * REC;                  // one operand, one 16-row tile, all heads
    constexpr int STG = 2 * TILEB;                  // a stage: Q fragments, dO fragments
    constexpr int NK16 = FLB_NK16;
    constexpr int F1 = FULL ? FULL : 1;
    // LDS: [stage 0][stage 1][NK16 slots of dO in the 16-wide layout][c0 rows 2 x 1 KB][D exchange 2 x NW x (16 H floats)][constants 512 B][NW x 3 transpose tiles]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const int nt = a.nt;
    constexpr int C0OFF = 2 * STG + NK16 * TILEB, DXB = 16 * H * 4, DXOFF = C0OFF + 2048, GCOFF = DXOFF + 2 * NW * DXB;
    constexpr int KBOFF = GCOFF + 512 + NW * 3 * (4 * H * FLB_GWR);       // dropout: 2 x 1 KB of keep-flag records behind the transpose tiles (talking_bwdq_kernel)

    float Al4[H / 4][H];                            // S' = Wl S (fp32, 4x4x1)
    fl_mixA_f32<H, false>(a.Wl, lane, Al4);
    fls16x4_t Awt[H / 4][H / 4];                    // dP = Ww^T dP' (bf16, 4x4x4)
    fl_mixA_16<H, true, false>(a.Ww, lane, 1.0f, Awt);
    // P travels as P * 2^8 through this kernel (the exponent starts from c0 + 8), like in the flash forward / dV pass: the proj_w mix runs in
    // fp16 on values that stay normal numbers; D, dWw and dV carry the factor and drop it where they leave (d_flush, the ws_w row, the merge)
    fls16x4_t Awp[H / 4][H / 4];                    // P' 2^8 = Ww (P 2^8) + bw 2^8 (fp16, 4x4x4)
    fl_mixA_16<H, false, true>(a.Ww, lane, 1.0f, Awp);
    f32x4_t vbw[H / 4];
#pragma unroll
    for (int gh = 0; gh < H / 4; ++gh)
#pragma unroll
        for (int i = 0; i < 4; ++i) vbw[gh][i] = a.bw[4 * gh + i] * FL_PD_SCALE;

    // weight-gradient outer products: X = dP' (front operand, double-buffered by tile parity), Y = P (back operand) - see talking_bwdq_kernel
    constexpr int GWR = FLB_GWR, GWT = 4 * H * GWR;
    unsigned char* gconst = smem + GCOFF;
    unsigned char* sgw = gconst + 512 + wave * (3 * GWT);
    if (threadIdx.x < 64) reinterpret_cast<uint2*>(gconst)[threadIdx.x] = (threadIdx.x >= 26 && threadIdx.x < 42) ? make_uint2(0x3F803F80u, 0x3F803F80u) : make_uint2(0u, 0u);
    const int gm = lane & 15, gk = lane >> 4;
    unsigned char* gw_wr = sgw + (gk * H) * GWR + gm * 8;
    const unsigned gw_rd = (unsigned)((gk * H + gm) * GWR);
    const unsigned char* gw_zero = gconst + 48;
    const unsigned char* gw_ones = gconst + 208;
    f32x4_t gwacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gwacc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    constexpr int NP = TILEB / 1024, NPW = (NP + NW - 1) / NW;
    unsigned voff[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int o = (i * NW + wave) * 1024 + lane * 16;
        voff[i] = (unsigned)((o / REC) * nt * REC + o % REC);
    }
    const float keep_inv = DROP ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const int kbsh = 2 * (gm & 3);                                 // keep flag of (query row, this lane's key), head g: bit (g >> 1) * 8 + kbsh + (g & 1) ...
#ifdef FLB_DBG_STAMP
    unsigned long long stamp[7] = {0, 0, 0, 0, 0, 0, 0}, stacc[6] = {0, 0, 0, 0, 0, 0}, stn = 0;
#endif
    const int kbln = (gm >> 2) << 4;                               // ... of the word of lane (query row) | kbln of the (q-tile, key tile) block

    const long s_begin = (long)blockIdx.x * a.spw;
    long s_end = s_begin + a.spw; if (s_end > a.total) s_end = a.total;
    long s = s_begin;
    while (s < s_end) {
        const int bm = (int)(s / nt), qt0 = (int)(s % nt);
        int seg = nt - qt0; if (seg > s_end - s) seg = (int)(s_end - s);
        const int b = bm / a.nmaj, mj = bm % a.nmaj;
        const int kt = mj * NW + wave;                          // this wave's key tile (wave-uniform)
        const bool wvalid = kt < nt;
        const int ktc = wvalid ? kt : nt - 1;
        const int nvw = (nt - mj * NW < NW) ? nt - mj * NW : NW;   // waves of this major that own a key tile
        unsigned kb_voff = 0u;              // keep flags: the four waves' records of a q-tile are consecutive (256 B each)
        if constexpr (DROP) {
            const int w = lane >> 4, kw = (mj * NW + w < nt) ? w : 0;
            kb_voff = (unsigned)(kw * 256 + (lane & 15) * 16);
        }

        // ---- this wave's K and V records -> registers (AccVGPR B operands of the score products)
        flu32x4_t ka[H][F1], va[H][F1];
        flu32x2_t kta[H], vta[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const unsigned char* kb_ = a.Kf + (((long)b * H + h) * nt + ktc) * REC;
            const unsigned char* vb_ = a.Vf + (((long)b * H + h) * nt + ktc) * REC;
#pragma unroll
            for (int st = 0; st < FULL; ++st) {
                ka[h][st] = *reinterpret_cast<const flu32x4_t*>(kb_ + st * 1024 + lane * 16);
                va[h][st] = *reinterpret_cast<const flu32x4_t*>(vb_ + st * 1024 + lane * 16);
            }
            if constexpr (TAIL16) {
                kta[h] = *reinterpret_cast<const flu32x2_t*>(kb_ + FULL * 1024 + lane * 8);
                vta[h] = *reinterpret_cast<const flu32x2_t*>(vb_ + FULL * 1024 + lane * 8);
            } else { kta[h] = (flu32x2_t){0u, 0u}; vta[h] = (flu32x2_t){0u, 0u}; }
        }
        f32x4_t dV[H][DT];
#pragma unroll
        for (int g = 0; g < H; ++g)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dV[g][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        // the streamed operand tiles of q-tile qt0 + i: Q, dO fragments -> stage i & 1, 16-wide dO -> slot i % NK16, row constants -> c0 buffer i & 1
        auto issue_tiles = [&](int i) {
#ifdef FLB_DBG_SAMETILE
            const int qt = 0;                 // timing experiment: every workgroup streams the same tile (all L2 hits)
#else
            const int qt = qt0 + i;
#endif
#pragma unroll
            for (int op = 0; op < 3; ++op) {
                const unsigned char* base = (op == 0) ? a.Qf : ((op == 1) ? a.dOf : a.dO16);
                const unsigned char* tb = base + ((long)b * H * nt + qt) * REC;
                const unsigned dst = (op < 2) ? lds0 + (i & 1) * STG + op * TILEB : lds0 + 2 * STG + (i % NK16) * TILEB;
                if constexpr (NP % NW == 0 && NPW <= 4) fl_glds16_run<NPW, NW * 1024>(tb, voff, dst + wave * 1024);
                else {
#pragma unroll
                    for (int ii = 0; ii < NPW; ++ii) {
                        const int p = ii * NW + wave;
                        if (NP % NW != 0 && p >= NP) break;
                        fl_glds16_s(tb, voff[ii], dst + p * 1024);
                    }
                }
            }
            if (wave == NW - 1) fl_glds16_s(a.c0 + ((long)b * a.Np + qt * 16) * H, (unsigned)(lane * 16), lds0 + C0OFF + (i & 1) * 1024);
            if constexpr (DROP)
                if (wave == 1) fl_glds16_s(a.keepbits + (((long)b * nt + qt) * nt + mj * NW) * 64, kb_voff, lds0 + KBOFF + (i & 1) * 1024);
        };

        struct Frags { flu32x4_t f[HB][F1]; flu32x2_t t[HB]; };
        auto load_frags = [&](const unsigned char* tile, int h0, Frags& o) {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const unsigned char* r = tile + (h0 + hb) * REC;
#pragma unroll
                for (int st = 0; st < FULL; ++st) o.f[hb][st] = *reinterpret_cast<const flu32x4_t*>(r + st * 1024 + lane * 16);
                if constexpr (TAIL16) o.t[hb] = *reinterpret_cast<const flu32x2_t*>(r + FULL * 1024 + lane * 8);
                else o.t[hb] = (flu32x2_t){0u, 0u};
            }
        };
        auto load_q = [&](int i, int h0, Frags& o) { load_frags(smem + (i & 1) * STG, h0, o); };
        auto load_d = [&](int i, int g0, Frags& o) { load_frags(smem + (i & 1) * STG + TILEB, g0, o); };
        // ---- FRONT half of q-tile i (matrix-heavy), chunks of FLB_HB heads
        // rs_t / rs_o (both or neither): the terms of a 4-head row-sum reduce-scatter of the PREVIOUS tile's back half, run in the shadow of this chunk's score
        // products (head dim 33 .. 48 only), and its result
        auto front_q = [&](int i, auto h0_c, const Frags& qfr, f32x4_t (&sp)[4][H / 4], const f32x4_t (*rs_t)[4] = nullptr, f32x4_t* rs_o = nullptr) {
            constexpr int h0 = decltype(h0_c)::value;
            if constexpr (h0 == 0) {        // the exponent starts from the row constants of query 4 (l >> 4) + r: broadcast LDS reads
                const unsigned char* cr = smem + C0OFF + (i & 1) * 1024 + (4 * gk) * H * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int gh = 0; gh < H / 4; ++gh) sp[r][gh] = *reinterpret_cast<const f32x4_t*>(cr + (r * H + 4 * gh) * 4) + 8.0f;      // exp2(. + 8) = P * 2^8
            }
            f32x4_t c[HB];
            static_assert(HB == 4, "chunk of 4 heads");
            if constexpr (FULL == 1 && TAIL16) {
                if (rs_t) flb_score_rs_f16(qfr.f, qfr.t, &ka[h0], &kta[h0], c, *rs_t, *rs_o);
                else flb_score_f16<FULL, TAIL16>(qfr.f, qfr.t, &ka[h0], &kta[h0], c);
            } else flb_score_f16<FULL, TAIL16>(qfr.f, qfr.t, &ka[h0], &kta[h0], c);
            flb_fence4(c[0], c[1], c[2], c[3]);

#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const f32x4_t cs = c[hb];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int gh = 0; gh < H / 4; ++gh) sp[r][gh] = __builtin_amdgcn_mfma_f32_4x4x1f32(Al4[gh][h0 + hb], cs[r], sp[r][gh], 0, 0, 0);
            }
        };
        auto front_d = [&](int i, auto g0_c, const Frags& dfr, f32x4_t (&dp)[4][H / 4], const uint32_t (&kb)[4], const f32x4_t (*rs_t)[4] = nullptr,
                           f32x4_t* rs_o = nullptr) {
            constexpr int g0 = decltype(g0_c)::value;
            unsigned char* gwf = gw_wr + (i & 1) * GWT;
            if constexpr (g0 == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int gh = 0; gh < H / 4; ++gh) dp[r][gh] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            }
            f32x4_t es[HB];
            if constexpr (FULL == 1 && TAIL16) {
                if (rs_t) flb_score_rs_bf16(dfr.f, dfr.t, &va[g0], &vta[g0], es, *rs_t, *rs_o);
                else flb_score_bf16<FULL, TAIL16>(dfr.f, dfr.t, &va[g0], &vta[g0], es);
            } else flb_score_bf16<FULL, TAIL16>(dfr.f, dfr.t, &va[g0], &vta[g0], es);
            flb_fence4(es[0], es[1], es[2], es[3]);
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                if constexpr (DROP) {
                    const int g = g0 + hb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) es[hb][r] *= ((kb[r] >> ((g >> 1) * 8 + kbsh + (g & 1))) & 1u) ? keep_inv : 0.f;
                }
                // bf16(dP') of this head: the X operand of the dWw outer product
                *reinterpret_cast<fls16x4_t*>(gwf + (g0 + hb) * GWR) = fl_pack4<false>(es[hb][0], es[hb][1], es[hb][2], es[hb][3]);
            }
#pragma unroll
            for (int hq = 0; hq < HB / 4; ++hq)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const fls16x4_t bv = fl_pack4<false>(es[4 * hq][r], es[4 * hq + 1][r], es[4 * hq + 2][r], es[4 * hq + 3][r]);
#pragma unroll
                    for (int gh = 0; gh < H / 4; ++gh) dp[r][gh] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(Awt[gh][g0 / 4 + hq], bv, dp[r][gh], 0, 0, 0);
                }
        };
        auto load_kb = [&](int i, uint32_t (&kb)[4]) {
            if constexpr (DROP) {
                const flu32x4_t w4 = *reinterpret_cast<const flu32x4_t*>(smem + KBOFF + (i & 1) * 1024 + (wvalid ? wave : 0) * 256 + (kbln + 4 * gk) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) kb[r] = w4[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) kb[r] = 0u;
            }
        };

        // ---- BACK half of q-tile i (vector-heavy)
        auto back_exp = [&](f32x4_t (&sp)[4][H / 4]) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int gh = 0; gh < H / 4; ++gh)
#pragma unroll
                    for (int k = 0; k < 4; ++k) sp[r][gh][k] = fl_exp2(sp[r][gh][k]);
        };
        // D of q-tile i, head group gh: the lane's terms dP P ; the reduced sums -> this wave's block of the exchange buffer
        auto d_terms = [&](int gh, f32x4_t (&sp)[4][H / 4], f32x4_t (&dp)[4][H / 4], f32x4_t (&t)[4]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = dp[r][gh] * sp[r][gh];
        };
        auto d_store = [&](int i, int gh, const f32x4_t& o) {
            float* dx = reinterpret_cast<float*>(smem + DXOFF + ((i & 1) * NW + wave) * DXB) + (4 * gk + (gm >> 2)) * H;
            if ((gm & 3) == 0) *reinterpret_cast<f32x4_t*>(dx + 4 * gh) = o;
        };
        auto back_ds = [&](int i, f32x4_t (&sp)[4][H / 4], f32x4_t (&dp)[4][H / 4], bool with_d = true) {
            unsigned char* gwb = gw_wr + 2 * GWT;
            // D[q][g] = sum over this tile's 16 keys (the lanes of a DPP row) of dP P: after the reduce-scatter every quad holds the sums of query
            // row 4 gk + (gm >> 2) -> this wave's block of the exchange buffer
            float* dx = reinterpret_cast<float*>(smem + DXOFF + ((i & 1) * NW + wave) * DXB) + (4 * gk + (gm >> 2)) * H;
#pragma unroll
            for (int gh = 0; gh < (with_d ? H / 4 : 0); ++gh) {
                f32x4_t t[4], o;
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = dp[r][gh] * sp[r][gh];
                flb_rowsum4(t, o);
                if ((gm & 3) == 0) *reinterpret_cast<f32x4_t*>(dx + 4 * gh) = o;
            }
#pragma unroll
            for (int g = 0; g < H; ++g)
                *reinterpret_cast<fls16x4_t*>(gwb + g * GWR) =
                    fl_pack4<false>(sp[0][g >> 2][g & 3], sp[1][g >> 2][g & 3], sp[2][g >> 2][g & 3], sp[3][g >> 2][g & 3]);
        };
        auto back_gw = [&](int i, f32x4_t (&sp)[4][H / 4], f32x4_t (&pp)[4][H / 4]) {
            {
                const unsigned rdf = gw_rd + (i & 1) * GWT, rdb = gw_rd + 2 * GWT;
                const unsigned char* gw_xrd = (gm < H) ? sgw + rdf : gw_zero;
                const unsigned char* gw_yrd = (gm < H) ? sgw + rdb : ((gm == H) ? gw_ones : gw_zero);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const flu32x4_t xa = *reinterpret_cast<const flu32x4_t*>(gw_xrd + c * 16);
                    const flu32x4_t yb = *reinterpret_cast<const flu32x4_t*>(gw_yrd + c * 16);
                    // one 32-deep instruction per 16-B packet pair: a contraction over positions does not care which 8 positions a lane group brings
                    // (the 16-deep form costs the same issue slot for half the positions: profiles/r05_mfma_form.txt)
                    gwacc[c & 3] = fl_mfma32<false>(xa, yb, gwacc[c & 3]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
            // P' 2^8 = Ww (P 2^8) + bw 2^8 on fp16 operands (P 2^8 <= 256: no saturation needed)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                fls16x4_t bv[H / 4];
#pragma unroll
                for (int hh = 0; hh < H / 4; ++hh) bv[hh] = fl_pack4_f16(sp[r][hh][0], sp[r][hh][1], sp[r][hh][2], sp[r][hh][3]);
#pragma unroll
                for (int gh = 0; gh < H / 4; ++gh) {
                    f32x4_t d = vbw[gh];
#pragma unroll
                    for (int hh = 0; hh < H / 4; ++hh)
                        d = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(flf16x4_t, Awp[gh][hh]), __builtin_bit_cast(flf16x4_t, bv[hh]), d, 0, 0, 0);
                    pp[r][gh] = d;
                }
            }
        };
        auto back_dv = [&](int i, f32x4_t (&pp)[4][H / 4], const uint32_t (&kb)[4]) {
            const unsigned char* sD16 = smem + 2 * STG + (i % NK16) * TILEB;
            fls16x4_t da[H][DT];                // requested FLB_DQAHEAD heads ahead of the matrix instructions that consume them
#pragma unroll
            for (int h = 0; h < ((FLB_DQAHEAD < H) ? FLB_DQAHEAD : H); ++h)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) da[h][dt] = *reinterpret_cast<const fls16x4_t*>(sD16 + h * REC + dt * 512 + lane * 8);
#pragma unroll
            for (int h = 0; h < H; ++h) {
                if (h + FLB_DQAHEAD < H) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) da[h + FLB_DQAHEAD][dt] = *reinterpret_cast<const fls16x4_t*>(sD16 + (h + FLB_DQAHEAD) * REC + dt * 512 + lane * 8);
                    __builtin_amdgcn_sched_barrier(FLB_SB_NOMEM);
                }
                float p4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p4[r] = pp[r][h >> 2][h & 3];
                    if constexpr (DROP) p4[r] *= ((kb[r] >> ((h >> 1) * 8 + kbsh + (h & 1))) & 1u) ? keep_inv : 0.f;
                }
                const fls16x4_t pk = fl_pack4<false>(p4[0], p4[1], p4[2], p4[3]);
                flb_dq_mfma<DT>(dV[h], da[h], pk);
            }
        };
        // D of q-tile i: the sum over the major's key tiles (fixed order) of the exchange blocks -> ws_d ; every wave adds a quarter of the 16 x H values
        auto d_flush = [&](int i) {
            constexpr int PER = 16 * H / NW;            // values per wave (32 at H = 8)
            if (lane < PER) {
                const float* src = reinterpret_cast<const float*>(smem + DXOFF + (i & 1) * NW * DXB) + wave * PER + lane;
                float acc = src[0];
#pragma unroll
                for (int w = 1; w < NW; ++w) acc += (w < nvw) ? src[w * (DXB / 4)] : 0.f;
                a.ws_d[((long)bm * a.Np + (qt0 + i) * 16) * H + wave * PER + lane] = acc * (1.0f / FL_PD_SCALE);
            }
        };
        auto front = [&](int i, f32x4_t (&sp)[4][H / 4], f32x4_t (&dp)[4][H / 4], uint32_t (&kb)[4]) {
            load_kb(i, kb);
            Frags fa, fb;
            load_q(i, 0, fa);
            if constexpr (H > HB) load_q(i, HB, fb);
            front_q(i, std::integral_constant<int, 0>{}, fa, sp);
            load_d(i, 0, fa);
            if constexpr (H > HB) front_q(i, std::integral_constant<int, HB>{}, fb, sp);
            if constexpr (H > HB) load_d(i, HB, fb);
            front_d(i, std::integral_constant<int, 0>{}, fa, dp, kb);
            if constexpr (H > HB) front_d(i, std::integral_constant<int, HB>{}, fb, dp, kb);
        };
        auto back = [&](int i, f32x4_t (&sp)[4][H / 4], f32x4_t (&dp)[4][H / 4], const uint32_t (&kb)[4]) {
            f32x4_t pp[4][H / 4];
            back_exp(sp);
            back_ds(i, sp, dp);
            back_gw(i, sp, pp);
            back_dv(i, pp, kb);
        };
        // one pipelined step: the front half of q-tile i + 1 beside the back half of q-tile i (see talking_bwdq_kernel)
        auto step = [&](int i, f32x4_t (&sp)[4][H / 4], f32x4_t (&dp)[4][H / 4], f32x4_t (&spn)[4][H / 4], f32x4_t (&dpn)[4][H / 4], Frags& fr0, Frags& fr1,
                        const uint32_t (&kb)[4], uint32_t (&kbn)[4], bool more) {
            f32x4_t pp[4][H / 4];
            load_kb(i + 1, kbn);
            if constexpr (H > HB) load_q(i + 1, HB, fr1); else load_d(i + 1, 0, fr1);
            __builtin_amdgcn_sched_barrier(FLB_SB_NOMEM);
            front_q(i + 1, std::integral_constant<int, 0>{}, fr0, spn);
            back_exp(sp);
            FLB_PHASE();
            FLB_STAMP(3);
            if (FLB_STAGGER && more && wave >= NW / 2) issue_tiles(i + 2);       // the late half of the workgroup's tile loads (talking_bwdq_kernel: admit)
            constexpr bool RSF = FLB_RSFUSE && H == 2 * HB && FULL == 1 && TAIL16;      // the D reduce-scatters of tile i ride on two score-product blocks of tile i + 1
            f32x4_t rt[4], ro0, ro1;
            if constexpr (H > HB) {
                load_d(i + 1, 0, fr0);
                if constexpr (RSF) d_terms(0, sp, dp, rt);
                __builtin_amdgcn_sched_barrier(FLB_SB_NOMEM);
                if constexpr (RSF) front_q(i + 1, std::integral_constant<int, HB>{}, fr1, spn, &rt, &ro0);
                else front_q(i + 1, std::integral_constant<int, HB>{}, fr1, spn);
            }
            if constexpr (RSF) { d_store(i, 0, ro0); back_ds(i, sp, dp, false); d_terms(1, sp, dp, rt); }
            else back_ds(i, sp, dp);

            FLB_PHASE();
            FLB_STAMP(4);
            if constexpr (H > HB) {
                load_d(i + 1, HB, fr1);
                __builtin_amdgcn_sched_barrier(FLB_SB_NOMEM);
                if constexpr (RSF) { front_d(i + 1, std::integral_constant<int, 0>{}, fr0, dpn, kbn, &rt, &ro1); d_store(i, 1, ro1); }
                else front_d(i + 1, std::integral_constant<int, 0>{}, fr0, dpn, kbn);
            } else front_d(i + 1, std::integral_constant<int, 0>{}, fr1, dpn, kbn);
            back_gw(i, sp, pp);
            FLB_PHASE();
            FLB_STAMP(5);
            if constexpr (H > HB) front_d(i + 1, std::integral_constant<int, HB>{}, fr1, dpn, kbn);
            back_dv(i, pp, kb);
        };
        auto admit = [&](int nxt, bool late) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (nxt + 1 < seg && !(late && wave >= NW / 2)) issue_tiles(nxt + 1);
        };

        f32x4_t spA[4][H / 4], dpA[4][H / 4], spB[4][H / 4], dpB[4][H / 4];
        uint32_t kbA[4], kbB[4];
        Frags frA, frB;
        issue_tiles(0);
        admit(0, false);
        if (wvalid) front(0, spA, dpA, kbA);
        for (int i = 0; i + 1 < seg; ++i) {
            FLB_STAMP(0);
            admit(i + 1, FLB_STAGGER && wvalid);
            FLB_STAMP(1);
            if (i > 0) d_flush(i - 1);           // written during step i - 1, one barrier ago (tried INSIDE the step's first region: slower, profiles/r06_bwdq_stamps.txt)
            FLB_STAMP(2);
            if (wvalid) {
                load_q(i + 1, 0, frA);
                step(i, spA, dpA, spB, dpB, frA, frB, kbA, kbB, i + 2 < seg);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    kbA[r] = kbB[r];
#pragma unroll
                    for (int gh = 0; gh < H / 4; ++gh) { spA[r][gh] = spB[r][gh]; dpA[r][gh] = dpB[r][gh]; }
                }
            }
#ifdef FLB_DBG_STAMP
            FLB_STAMP(6);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(stamp[0]), "+s"(stamp[1]), "+s"(stamp[2]), "+s"(stamp[3]), "+s"(stamp[4]), "+s"(stamp[5]), "+s"(stamp[6]));
            for (int k = 0; k < 6; ++k) stacc[k] += stamp[k + 1] - stamp[k];
            ++stn;
#endif
        }
        if (wvalid) back(seg - 1, spA, dpA, kbA);
        __builtin_amdgcn_s_barrier();              // the last buffers have been read, the last D blocks written by everybody
        if (seg > 1) d_flush(seg - 2);
        d_flush(seg - 1);

        // ---- partial dV of this segment -> the major's slot
        const int first_wg = (int)(((long)bm * nt) / a.spw);
        const int slot = (int)blockIdx.x - first_wg;
#pragma unroll
        for (int g = 0; g < H; ++g)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) flb_acc_fence(dV[g][dt]);
        if (wvalid) {
            float* dst = a.ws_v + (((long)bm * FL_MAXSLOT + slot) * NW + wave) * (long)(H * DT * 256);
#pragma unroll
            for (int g = 0; g < H; ++g)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4_t*>(dst + (g * DT + dt) * 256 + lane * 4) = dV[g][dt];
        }
        s += seg;
    }

#ifdef FLB_DBG_STAMP
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.ws_w);       // (row 0, the half the query-major kernel fills later)
        for (int k = 0; k < 6; ++k) o[k] = stacc[k];
        o[6] = stn;
    }
#endif
    // ---- weight-gradient partials of this wave -> the [dWw | dbw] half of its row of ws_w
    {
        constexpr int NWG = 2 * (H * H + H);
        float* row = a.ws_w + ((long)blockIdx.x * NW + wave) * NWG + (H * H + H);
        const f32x4_t dsum = (gwacc[0] + gwacc[1]) + (gwacc[2] + gwacc[3]);
        const int nn = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int g = 4 * (lane >> 4) + r;
            if (g < H && nn < H) row[g * H + nn] = dsum[r] * (1.0f / FL_PD_SCALE);          // dWw: Y was P * 2^8
            if (g < H && nn == H) row[H * H + g] = dsum[r];                                     // dbw: the ones column
        }
    }
}

// D rows [B][Np][H] = sum over the majors of ws_d [B * nmaj][Np][H] ; rows >= N zero.  One thread per element, fixed order.
__global__ __launch_bounds__(256) void bwdk_rows_merge_kernel(const float* __restrict__ ws, float* __restrict__ out, int B, int H, int N, int Np, int nmaj) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long per = (long)Np * H;
    if (i >= (long)B * per) return;
    const long bq = i / H; const int q = (int)(bq % Np), b = (int)(bq / Np);
    if (q >= N) { out[i] = 0.f; return; }
    const float* src = ws + (long)b * nmaj * per + (i - (long)b * per);
    float acc = 0.f;
    int m = 0;
    for (; m + 8 <= nmaj; m += 8) {          // eight loads in flight, added in index order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[(long)(m + k) * per];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    for (; m < nmaj; ++m) acc += src[(long)m * per];
    out[i] = acc;
}

static inline int flb_dsteps(int dh, int* tail) {
    const int rem = dh % 32, full = dh / 32 + (rem > 16 ? 1 : 0);
    *tail = (rem > 0 && rem <= 16) ? 1 : 0;
    return full + *tail;
}

template <int H, int DSTEPS, bool TAIL16>
static int launch_bwdq(const FlashBwdArgs& a, int nwg, bool drop, hipStream_t st) {
    constexpr int FULL = DSTEPS - (TAIL16 ? 1 : 0), DT = 2 * FULL + (TAIL16 ? 1 : 0), REC = DT * 512;
    constexpr int smem0 = 4 * H * REC + FLB_NK16 * H * REC + 512 + FLB_NW * 3 * 4 * H * FLB_GWR;
    const int smem = smem0 + (drop ? 2048 : 0);          // + the keep-flag records
    if (smem > 160 * 1024) return -2;
    static bool attr_set[2] = {false, false};
    const void* fn = drop ? reinterpret_cast<const void*>(&talking_bwdq_kernel<H, DSTEPS, TAIL16, true>)
                          : reinterpret_cast<const void*>(&talking_bwdq_kernel<H, DSTEPS, TAIL16, false>);
    if (!attr_set[drop]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set[drop] = true;
    }
    if (drop) hipLaunchKernelGGL((talking_bwdq_kernel<H, DSTEPS, TAIL16, true>), dim3(nwg), dim3(64 * FLB_NW), smem, st, a);
    else hipLaunchKernelGGL((talking_bwdq_kernel<H, DSTEPS, TAIL16, false>), dim3(nwg), dim3(64 * FLB_NW), smem, st, a);
    SPE_CHECK_LAUNCH();
    return 0;
}

static int dispatch_bwdq(const FlashBwdArgs& a, int H, int dh, int nwg, hipStream_t st) {
    int tail; const int ds = flb_dsteps(dh, &tail);
    const bool drop = a.p_drop > 0.f;
#define SPE_BWDQ(HH)                                                                       \
    if (H == HH && ds == 2 && tail) return launch_bwdq<HH, 2, true>(a, nwg, drop, st);     \
    if (H == HH && ds == 2 && !tail) return launch_bwdq<HH, 2, false>(a, nwg, drop, st);   \
    if (H == HH && ds == 1 && tail) return launch_bwdq<HH, 1, true>(a, nwg, drop, st);     \
    if (H == HH && ds == 1 && !tail) return launch_bwdq<HH, 1, false>(a, nwg, drop, st);
    SPE_BWDQ(8)
    SPE_BWDQ(4)
#undef SPE_BWDQ
    return -2;
}

// C-ABI: see include/spe_hip.h
extern "C" int spe_talking_bwdq_plan(int B, int N, int nwg, int* steps_per_wg, int* nwg_used, int* nmajor) {
    const int nt = (N + 15) / 16;
    if ((long)B * nt <= 0 || nwg <= 0) { *steps_per_wg = 0; *nwg_used = 0; *nmajor = 0; return 0; }
    const FlashPlan p = fl_plan(B, nt, FLB_NW, nt, nwg);
    *steps_per_wg = p.spw; *nwg_used = p.nwg; *nmajor = p.nmaj;
    return 0;
}

static int bwdq_fill(FlashBwdArgs& a, FlashPlan& p, const void* Qf, const void* dOf, const void* Kf, const void* Vf, const void* K16, const float* Wl,
                     const float* Ww, const float* c0, int Np, const void* keepbits, int B, int H, int N, int dh, int nwg, float p_drop) {
    const int nt = (N + 15) / 16;
    if (dh < 1 || dh > 64 || nwg <= 0 || Np < nt * 16 || (p_drop > 0.f && !keepbits)) return -2;
    p = fl_plan(B, nt, FLB_NW, nt, nwg);
    a.Qf = (const unsigned char*)Qf; a.dOf = (const unsigned char*)dOf; a.Kf = (const unsigned char*)Kf; a.Vf = (const unsigned char*)Vf;
    a.K16 = (const unsigned char*)K16; a.Wl = Wl; a.Ww = Ww; a.c0 = c0; a.Np = Np; a.Drows = nullptr;
    a.ws_q = nullptr; a.ws_w = nullptr; a.dS = nullptr;
    a.keepbits = (p_drop > 0.f) ? reinterpret_cast<const unsigned*>(keepbits) : nullptr;
    a.B = B; a.N = N; a.nt = nt; a.nmaj = p.nmaj; a.spw = p.spw; a.total = p.total; a.p_drop = p_drop;
    return 0;
}


extern "C" int spe_talking_bwdq_pass2(const void* Qf, const void* dOf, const void* Kf, const void* Vf, const void* K16, const float* Wl, const float* Ww,
                                      const float* c0, const float* Drows, int Np, float* ws_q, float* ws_w, void* dS, float* dq, void* dq16,
                                      long ob, long on, long oh, float scale, const void* keepbits, int B, int H, int N, int dh, int nwg, float p_drop,
                                      hipStream_t st) {
    if ((long)B * ((N + 15) / 16) <= 0) return 0;
    FlashBwdArgs a; FlashPlan p;
    int rc = bwdq_fill(a, p, Qf, dOf, Kf, Vf, K16, Wl, Ww, c0, Np, keepbits, B, H, N, dh, nwg, p_drop);
    if (rc != 0) return rc;
    if (!ws_q || !ws_w || !dS || !Drows || !K16 || (!dq && !dq16)) return -2;
    a.ws_q = ws_q; a.ws_w = ws_w; a.dS = reinterpret_cast<unsigned short*>(dS); a.Drows = Drows;
    rc = dispatch_bwdq(a, H, dh, p.nwg, st);
    if (rc != 0) return rc;
    const int DT = (dh + 15) / 16;
    const long nvec = (long)B * p.nmaj * FLB_NW * H * DT * 64;
    hipLaunchKernelGGL(bwdq_dq_merge_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st, ws_q, dq, reinterpret_cast<unsigned short*>(dq16),
                       ob, on, oh, B, H, N, a.nt, dh, DT, p.nmaj, p.spw, scale, nvec);
    SPE_CHECK_LAUNCH();
    return 0;
}

// ---- key-major pass: launch, dispatch, C-ABI
template <int H, int DSTEPS, bool TAIL16>
static int launch_bwdk(const FlashBwdKArgs& a, int nwg, bool drop, hipStream_t st) {
    constexpr int FULL = DSTEPS - (TAIL16 ? 1 : 0), DT = 2 * FULL + (TAIL16 ? 1 : 0), REC = DT * 512;
    constexpr int smem0 = 4 * H * REC + FLB_NK16 * H * REC + 2048 + 2 * FLB_NW * 16 * H * 4 + 512 + FLB_NW * 3 * 4 * H * FLB_GWR;
    const int smem = smem0 + (drop ? 2048 : 0);          // + the keep-flag records
    if (smem > 160 * 1024) return -2;
    static bool attr_set[2] = {false, false};
    const void* fn = drop ? reinterpret_cast<const void*>(&talking_bwdk_kernel<H, DSTEPS, TAIL16, true>)
                          : reinterpret_cast<const void*>(&talking_bwdk_kernel<H, DSTEPS, TAIL16, false>);
    if (!attr_set[drop]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set[drop] = true;
    }
    if (drop) hipLaunchKernelGGL((talking_bwdk_kernel<H, DSTEPS, TAIL16, true>), dim3(nwg), dim3(64 * FLB_NW), smem, st, a);
    else hipLaunchKernelGGL((talking_bwdk_kernel<H, DSTEPS, TAIL16, false>), dim3(nwg), dim3(64 * FLB_NW), smem, st, a);
    SPE_CHECK_LAUNCH();
    return 0;
}

static int dispatch_bwdk(const FlashBwdKArgs& a, int H, int dh, int nwg, hipStream_t st) {
    int tail; const int ds = flb_dsteps(dh, &tail);
    const bool drop = a.p_drop > 0.f;
#define SPE_BWDK(HH)                                                                       \
    if (H == HH && ds == 2 && tail) return launch_bwdk<HH, 2, true>(a, nwg, drop, st);     \
    if (H == HH && ds == 2 && !tail) return launch_bwdk<HH, 2, false>(a, nwg, drop, st);   \
    if (H == HH && ds == 1 && tail) return launch_bwdk<HH, 1, true>(a, nwg, drop, st);     \
    if (H == HH && ds == 1 && !tail) return launch_bwdk<HH, 1, false>(a, nwg, drop, st);
    SPE_BWDK(8)
    SPE_BWDK(4)
#undef SPE_BWDK
    return -2;
}

extern "C" int spe_talking_bwdk_pass1(const void* Qf, const void* dOf, const void* dO16, const void* Kf, const void* Vf, const float* Wl, const float* Ww,
                                      const float* bw, const float* c0, int Np, float* ws_d, float* ws_v, float* ws_w, float* Drows, float* dv, void* dv16,
                                      long ob, long on, long oh, const void* keepbits, int B, int H, int N, int dh, int nwg, float p_drop, hipStream_t st) {
    const int nt = (N + 15) / 16;
    if ((long)B * nt <= 0) return 0;
    if (dh < 1 || dh > 64 || nwg <= 0 || Np < nt * 16 + 64 || (p_drop > 0.f && !keepbits)) return -2;
    if (!Qf || !dOf || !dO16 || !Kf || !Vf || !Wl || !Ww || !bw || !c0 || !ws_d || !ws_v || !ws_w || !Drows || (!dv && !dv16)) return -2;
    const FlashPlan p = fl_plan(B, nt, FLB_NW, nt, nwg);
    FlashBwdKArgs a;
    a.Qf = (const unsigned char*)Qf; a.dOf = (const unsigned char*)dOf; a.dO16 = (const unsigned char*)dO16;
    a.Kf = (const unsigned char*)Kf; a.Vf = (const unsigned char*)Vf; a.Wl = Wl; a.Ww = Ww; a.bw = bw; a.c0 = c0; a.Np = Np;
    a.ws_d = ws_d; a.ws_v = ws_v; a.ws_w = ws_w;
    a.keepbits = (p_drop > 0.f) ? reinterpret_cast<const unsigned*>(keepbits) : nullptr;
    a.B = B; a.N = N; a.nt = nt; a.nmaj = p.nmaj; a.spw = p.spw; a.total = p.total; a.p_drop = p_drop;
    int rc = dispatch_bwdk(a, H, dh, p.nwg, st);
    if (rc != 0) return rc;
    const long n = (long)B * Np * H;
    hipLaunchKernelGGL(bwdk_rows_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws_d, Drows, B, H, N, Np, p.nmaj);
    SPE_CHECK_LAUNCH();
    const int DT = (dh + 15) / 16;
    const long nvec = (long)B * p.nmaj * FLB_NW * H * DT * 64;
    hipLaunchKernelGGL(bwdq_dq_merge_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st, ws_v, dv, reinterpret_cast<unsigned short*>(dv16),
                       ob, on, oh, B, H, N, nt, dh, DT, p.nmaj, p.spw, 1.0f / FL_PD_SCALE, nvec);
    SPE_CHECK_LAUNCH();
    return 0;
}
