// Query-major backward passes of the talking-heads attention on the flash skeleton (K4 of SURVEY.md section 2.2; reference
// models/cait.py:377-389 and its autograd), built for ONE wave per SIMD and the whole 512-entry register file:
//
//   pass 1 (spe_talking_bwdq_pass1):  D[h', q] = sum_k dP[h'] P[h'] (the softmax backward's row term), dWw, dbw
//   pass 2 (spe_talking_bwdq_pass2):  dS' = P (dP - D), dWl, dbl, dS = Wl^T dS' -> bf16 16 x 16 blocks (read once more, by the dK
//                                     contraction) AND dQ += dS K in registers: the streaming dQ contraction of attn_contract.hip and
//                                     one of the two reads of the 554 MB (cfg2) dS tensor are gone.
//
// with S = scale q k^T, S' = Wl S + bl, P = softmax_k(S'), P' = Ww P + bw, dP' = dropout-mask * (dO V^T), dP = Ww^T dP'.  Both passes
// recompute S / S' / P from the forward's own fp16 fragments and statistics (P matches the forward exactly) and dP' from bf16 fragments.
//
// Register plan (why this file is compiled with -mllvm -amdgpu-mfma-vgpr-form=1, spe_amd/build.py).  A wave owns one 16-query tile for a
// whole segment of key tiles.  Its Q and dO fragment records (2 x 48 registers at cfg2) and - pass 2 - the 96 dQ accumulators live in
// the AccVGPR half of the register file: they are touched by matrix instructions only, as B operands resp. C / D, through inline
// assembly with "a" constraints.  Everything else (the score tile of all heads, both head-mix accumulator sets, the mixing weights) stays
// under 256 ordinary VGPRs, and with the VGPR form forced for the builtin matrix instructions hipcc moves nothing between the two halves
// inside the key loop (without the flag every builtin result lands in an AccVGPR once a kernel may use them and is copied out for the
// vector instructions: 4.4 moves per matrix instruction measured on the round-4 prototype of the forward kernel).
// The q-side operands in registers instead of LDS is what makes the flash skeleton fit: 4 resident q-tiles x (Q + dO) would be 96 KB
// next to 72 KB of stage buffers and 32 KB of transpose tiles.
//
// LDS: two stages of the streamed key-side tiles - K fp16 fragments, V bf16 fragments, pass 2: K bf16 in the 16-wide layout - filled by
// global_load_lds_dwordx4 one step ahead (one barrier per step), plus the wave-private transpose tiles of the weight-gradient outer
// products (attn_fused.hip, GWM).  Work split: attn_flash_common.h (fl_plan) with 4 q-tiles per workgroup; partial D / dQ of a segment
// go to slot workspaces summed in fixed order by the two small merge kernels below (bitwise reproducible, no atomics).
#include "attn_flash_common.h"
#include <type_traits>

#define FLB_NW 4                         // waves per workgroup = q-tiles per workgroup (one wave per SIMD)
#ifndef FLB_HB
#define FLB_HB 4                         // heads per operand batch of the score products
#endif
#ifndef FLB_PHASEFENCE
#define FLB_PHASEFENCE 1
#endif
#if FLB_PHASEFENCE
#define FLB_PHASE() __builtin_amdgcn_sched_barrier(0)
#else
#define FLB_PHASE() do {} while (0)
#endif
#ifndef SPE_ABLATE
#undef FLB_DBG_NOGWM
#undef FLB_DBG_NODQ
#undef FLB_DBG_NOST
#undef FLB_DBG_NOMIX1
#undef FLB_DBG_NOEXP
#undef FLB_DBG_NOMIX16
#undef FLB_DBG_NODMA
#endif

struct FlashBwdArgs {
    const unsigned char* Qf; const unsigned char* dOf;            // q-side fragment records (fp16 q * scale * log2 e ; bf16 dO)
    const unsigned char* Kf; const unsigned char* Vf; const unsigned char* K16;   // key-side records: fp16 k, bf16 v (32-wide layout), bf16 k (16-wide layout, pass 2)
    const float* Wl; const float* Ww;
    const float* c0;                     // [B][Np][H]: bl log2(e) - m + log2(1 / l), rows >= N zero
    const float* Drows;                  // pass 2: [B][Np][H] D of pass 1, rows >= N zero
    int Np;
    float* ws_d;                         // pass 1: partial D [B * nmaj][FL_MAXSLOT][FLB_NW][H][16]
    float* ws_q;                         // pass 2: partial dQ [B * nmaj][FL_MAXSLOT][FLB_NW][H][DT][64 lanes][4]
    float* ws_w;                         // weight-gradient partials [nwg * FLB_NW][2 * (H * H + H)], row = [dWl | dbl | dWw | dbw]: pass 2 fills the first half, pass 1 the second
    unsigned short* dS;                  // pass 2: bf16 blocks [B, H, nt, nt][64 lanes][4], lane = (query l & 15, keys 4 (l >> 4) + i)
    const unsigned* keepbits;            // dropout keep flags of spe_talking_flash_fwd [B][nt][nt][64]
    int B, N, nt, nmaj, spw; long total;
    float p_drop;
};

// ---- score products with the B operand in AccVGPRs.  One statement per head: FULL chained 32-deep steps into c, the 16-deep tail step into
// its OWN accumulator t (an accumulate chain never mixes two MFMA shapes: attn_fused.hip).  No wait states inside: the results are only
// read behind flb_fence*, which follows the whole batch.
#define FLB_SCORE_ASM(NAME, M32, M16)                                                                                                      \
    template <int FULL, bool TAIL16>                                                                                                       \
    __device__ __forceinline__ void NAME(const flu32x4_t* k32, flu32x2_t k16, const flu32x4_t* q32, flu32x2_t q16, f32x4_t& c, f32x4_t& t) { \
        if constexpr (FULL == 1 && TAIL16)                                                                                                 \
            asm volatile(M32 " %0, %2, %4, 0\n\t" M16 " %1, %3, %5, 0" : "=&v"(c), "=&v"(t) : "v"(k32[0]), "v"(k16), "a"(q32[0]), "a"(q16));    \
        else if constexpr (FULL == 2 && !TAIL16)                                                                                           \
            asm volatile(M32 " %0, %1, %3, 0\n\t" M32 " %0, %2, %4, %0" : "=&v"(c) : "v"(k32[0]), "v"(k32[1]), "a"(q32[0]), "a"(q32[1]));      \
        else if constexpr (FULL == 1 && !TAIL16)                                                                                           \
            asm volatile(M32 " %0, %1, %2, 0" : "=&v"(c) : "v"(k32[0]), "a"(q32[0]));                                                      \
        else                                                                                                                               \
            asm volatile(M16 " %0, %1, %2, 0" : "=&v"(t) : "v"(k16), "a"(q16));                                                            \
    }
FLB_SCORE_ASM(flb_score_f16, "v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x16_f16")
FLB_SCORE_ASM(flb_score_bf16, "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x16_bf16")

// results of the matrix instructions issued above become readable: 13 wait states behind the last one (8-pass instruction -> any reader),
// tied to the registers so that no consumer is scheduled in front of it
__device__ __forceinline__ void flb_fence4(f32x4_t& a, f32x4_t& b, f32x4_t& c, f32x4_t& d) {
    asm volatile("s_nop 7\n\ts_nop 4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// dQ^T[d][q] += K^T[d][key] dS^T[key][q] for one head: DT accumulate instructions on AccVGPR accumulators; s_nop 1: pk was just written
// by the vector pipe
template <int DT>
__device__ __forceinline__ void flb_dq_mfma(f32x4_t* acc, const fls16x4_t* ka, fls16x4_t pk) {
    if constexpr (DT == 1)
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+a"(acc[0]) : "v"(ka[0]), "v"(pk));
    else if constexpr (DT == 2)
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_16x16x16_bf16 %1, %3, %4, %1"
                     : "+a"(acc[0]), "+a"(acc[1]) : "v"(ka[0]), "v"(ka[1]), "v"(pk));
    else if constexpr (DT == 3)
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x16_bf16 %0, %3, %6, %0\n\tv_mfma_f32_16x16x16_bf16 %1, %4, %6, %1\n\tv_mfma_f32_16x16x16_bf16 %2, %5, %6, %2"
                     : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]) : "v"(ka[0]), "v"(ka[1]), "v"(ka[2]), "v"(pk));
    else
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x16_bf16 %0, %4, %8, %0\n\tv_mfma_f32_16x16x16_bf16 %1, %5, %8, %1\n\t"
                     "v_mfma_f32_16x16x16_bf16 %2, %6, %8, %2\n\tv_mfma_f32_16x16x16_bf16 %3, %7, %8, %3"
                     : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]) : "v"(ka[0]), "v"(ka[1]), "v"(ka[2]), "v"(ka[3]), "v"(pk));
}
// the accumulators become readable (segment end)
__device__ __forceinline__ void flb_acc_fence(f32x4_t& a) { asm volatile("s_nop 7\n\ts_nop 4" : "+a"(a)); }

template <int H, int DSTEPS, bool TAIL16, bool DROP, int PASS>
__global__ __launch_bounds__(64 * FLB_NW, 1) void talking_bwdq_kernel(FlashBwdArgs a) {
    constexpr int FULL = DSTEPS - (TAIL16 ? 1 : 0), DT = 2 * FULL + (TAIL16 ? 1 : 0), REC = DT * 512;
    constexpr int NW = FLB_NW, HB = (H >= FLB_HB) ? FLB_HB : H;
    constexpr int TILEB = H * REC;                  // one operand, one 16-row tile, all heads
    constexpr int NOP = (PASS == 2) ? 3 : 2;        // streamed operands per step: K fragments, V fragments, (pass 2) K in the 16-wide layout
    constexpr int STAGEB = NOP * TILEB;
    constexpr int F1 = FULL ? FULL : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [stage 0][stage 1][zeros 128 B][ones 128 B][NW transpose tiles of 1024 * H B]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const int nt = a.nt, N = a.N;

    // ---- mixing weights as matrix-instruction operands (attn_flash_common.h)
    float Al4[H / 4][H];                            // S' = Wl S (fp32, 4x4x1)
    fl_mixA_f32<H, false>(a.Wl, lane, Al4);
    fls16x4_t Awt[H / 4][H / 4];                    // dP = Ww^T dP' (bf16, 4x4x4)
    fl_mixA_16<H, true, false>(a.Ww, lane, 1.0f, Awt);
    fls16x4_t Alt[(PASS == 2) ? H / 4 : 1][(PASS == 2) ? H / 4 : 1];           // dS = Wl^T dS' (bf16, 4x4x4)
    if constexpr (PASS == 2) fl_mixA_16<H, true, false>(a.Wl, lane, 1.0f, Alt);

    // ---- weight-gradient outer products on the matrix pipe through a wave-private LDS transpose (see attn_fused.hip, GWM):
    // pass 1: X = dP', Y = P -> dWw (+ dbw from the ones column) ; pass 2: X = dS', Y = S -> dWl (dbl stays an fp32 vector sum)
    unsigned char* gconst = smem + 2 * STAGEB;
    unsigned char* sgw = gconst + 256 + wave * (1024 * H);
    if (threadIdx.x < 32) reinterpret_cast<uint2*>(gconst)[threadIdx.x] = (threadIdx.x < 16) ? make_uint2(0u, 0u) : make_uint2(0x3F803F80u, 0x3F803F80u);
    const int gm = lane & 15, gk = lane >> 4;
    unsigned char* gw_wr = sgw + ((gk * H) * 16 + gm) * 8;                                  // + h * 128: packet of head h, query gm, key group gk
    const unsigned char* gw_xrd = (gm < H) ? sgw + ((gk * H + gm) * 16) * 8 : gconst;       // 16 packets (queries 0..15) of head gm
    const unsigned char* gw_yrd = (gm < H) ? sgw + 512 * H + ((gk * H + gm) * 16) * 8 : ((gm == H) ? gconst + 128 : gconst);
    f32x4_t gwacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gwacc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float gb[(PASS == 2) ? H : 1];                  // dbl
#pragma unroll
    for (int g = 0; g < ((PASS == 2) ? H : 1); ++g) gb[g] = 0.f;

    constexpr int NP = TILEB / 1024, NPW = (NP + NW - 1) / NW;     // 1-KB pieces of an operand tile, pieces per wave
    unsigned voff[NPW];                                            // byte offset of this lane's 16 B of piece i * NW + wave inside a (b, tile) image
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int o = (i * NW + wave) * 1024 + lane * 16;
        voff[i] = (unsigned)((o / REC) * nt * REC + o % REC);
    }
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
        f32x4_t c0v[H / 4], Dv[(PASS == 2) ? H / 4 : 1];
#pragma unroll
        for (int gh = 0; gh < H / 4; ++gh) {
            c0v[gh] = *reinterpret_cast<const f32x4_t*>(a.c0 + ((long)b * a.Np + q) * H + 4 * gh);
            if constexpr (PASS == 2) Dv[gh] = *reinterpret_cast<const f32x4_t*>(a.Drows + ((long)b * a.Np + q) * H + 4 * gh);
        }
        float rD[(PASS == 1) ? H : 1];
#pragma unroll
        for (int g = 0; g < ((PASS == 1) ? H : 1); ++g) rD[g] = 0.f;
        f32x4_t dQ[(PASS == 2) ? H : 1][(PASS == 2) ? DT : 1];
#pragma unroll
        for (int g = 0; g < ((PASS == 2) ? H : 1); ++g)
#pragma unroll
            for (int dt = 0; dt < ((PASS == 2) ? DT : 1); ++dt) dQ[g][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        // the streamed operand tiles of key tile kt -> stage st: NOP * TILEB / 1024 pieces shared by the waves
        auto issue_tiles = [&](int kt, int st) {
#ifdef FLB_DBG_NODMA
            if (kt != kt0) return;
#endif
#pragma unroll
            for (int op = 0; op < NOP; ++op) {
                const unsigned char* base = (op == 0) ? a.Kf : ((op == 1) ? a.Vf : a.K16);
                const unsigned char* tb = base + ((long)b * H * nt + kt) * REC;
#pragma unroll
                for (int i = 0; i < NPW; ++i) {
                    const int p = i * NW + wave;
                    if (NP % NW != 0 && p >= NP) break;
                    fl_glds16_s(tb, voff[i], lds0 + st * STAGEB + op * TILEB + p * 1024);
                }
            }
        };

        // ---- one key tile.  MASK: the ragged last tile (keys >= N get P = 0)
        auto tile = [&](int i, auto mask_c) {
            constexpr bool MASK = decltype(mask_c)::value;
            const int kt = kt0 + i;
            const unsigned char* sK = smem + (i & 1) * STAGEB;
            const unsigned char* sV = sK + TILEB;
            const unsigned char* sK16 = sV + TILEB;
            const int key0 = kt * 16 + 4 * (lane >> 4);
            uint32_t kb = 0u;
            if constexpr (DROP) kb = a.keepbits[(((long)b * nt + qt) * nt + kt) * 64 + lane];

            // ---- S^T = K Q^T (lane = (query l & 15, keys 4 (l >> 4) + r)), S' = Wl S + c0 on the fly, head-outer (attn_flash.hip, FLF_MIXH)
            f32x4_t sp[4][H / 4];
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
#pragma unroll
            for (int h0 = 0; h0 < H; h0 += HB) {
                flu32x4_t kf[HB][F1]; flu32x2_t kt16[HB];
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) {
                    const unsigned char* kr = sK + (h0 + hb) * REC;
#pragma unroll
                    for (int st = 0; st < FULL; ++st) kf[hb][st] = *reinterpret_cast<const flu32x4_t*>(kr + st * 1024 + lane * 16);
                    if constexpr (TAIL16) kt16[hb] = *reinterpret_cast<const flu32x2_t*>(kr + FULL * 1024 + lane * 8);
                    else kt16[hb] = (flu32x2_t){0u, 0u};
                }
                f32x4_t c[HB], t[HB];
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) flb_score_f16<FULL, TAIL16>(kf[hb], kt16[hb], qa[h0 + hb], qta[h0 + hb], c[hb], t[hb]);
                if constexpr (FULL > 0) { static_assert(HB == 4, "fence arity"); flb_fence4(c[0], c[1], c[2], c[3]); }
                if constexpr (TAIL16) flb_fence4(t[0], t[1], t[2], t[3]);
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) {
                    f32x4_t cs;
                    if constexpr (FULL > 0 && TAIL16) cs = c[hb] + t[hb];
                    else if constexpr (FULL > 0) cs = c[hb];
                    else cs = t[hb];
#ifndef FLB_DBG_NOGWM
                    // pass 2: bf16(S) of this head is the Y operand of the dWl outer product - straight into the wave's transpose tile (its
                    // previous contents were read by this wave's own, older LDS instructions)
                    if constexpr (PASS == 2) *reinterpret_cast<fls16x4_t*>(gw_wr + 512 * H + (h0 + hb) * 128) = fl_pack4<false>(cs[0], cs[1], cs[2], cs[3]);
#endif
#ifndef FLB_DBG_NOMIX1
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int gh = 0; gh < H / 4; ++gh) sp[r][gh] = __builtin_amdgcn_mfma_f32_4x4x1f32(Al4[gh][h0 + hb], cs[r], sp[r][gh], 0, 0, 0);
#else
#pragma unroll
                    for (int r = 0; r < 4; ++r) sp[r][(h0 + hb) >> 2][(h0 + hb) & 3] += cs[r];
#endif
                }
            }
            FLB_PHASE();
            // ---- P = exp2(S' + c0)   (sp[r][gh][i]: head 4 gh + i at key r)
#ifndef FLB_DBG_NOEXP
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int gh = 0; gh < H / 4; ++gh)
#pragma unroll
                    for (int k = 0; k < 4; ++k) sp[r][gh][k] = fl_exp2(sp[r][gh][k]);
#endif

            FLB_PHASE();
            // ---- dP'^T = V dO^T (bf16 operands), dropout, dP = Ww^T dP' head group by head group
            f32x4_t dp[4][H / 4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int gh = 0; gh < H / 4; ++gh) dp[r][gh] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g0 = 0; g0 < H; g0 += HB) {
                flu32x4_t vf[HB][F1]; flu32x2_t vt16[HB];
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) {
                    const unsigned char* vr = sV + (g0 + hb) * REC;
#pragma unroll
                    for (int st = 0; st < FULL; ++st) vf[hb][st] = *reinterpret_cast<const flu32x4_t*>(vr + st * 1024 + lane * 16);
                    if constexpr (TAIL16) vt16[hb] = *reinterpret_cast<const flu32x2_t*>(vr + FULL * 1024 + lane * 8);
                    else vt16[hb] = (flu32x2_t){0u, 0u};
                }
                f32x4_t e[HB], u[HB];
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) flb_score_bf16<FULL, TAIL16>(vf[hb], vt16[hb], da[g0 + hb], dta[g0 + hb], e[hb], u[hb]);
                if constexpr (FULL > 0) flb_fence4(e[0], e[1], e[2], e[3]);
                if constexpr (TAIL16) flb_fence4(u[0], u[1], u[2], u[3]);
                f32x4_t es[HB];
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) {
                    if constexpr (FULL > 0 && TAIL16) es[hb] = e[hb] + u[hb];
                    else if constexpr (FULL > 0) es[hb] = e[hb];
                    else es[hb] = u[hb];
                    if constexpr (DROP) {       // bit hp * 8 + 2 r + e: key r of the lane's group, head 2 hp + e
                        const int g = g0 + hb;
#pragma unroll
                        for (int r = 0; r < 4; ++r) es[hb][r] *= ((kb >> ((g >> 1) * 8 + 2 * r + (g & 1))) & 1u) ? keep_inv : 0.f;
                    }
#ifndef FLB_DBG_NOGWM
                    // pass 1: bf16(dP') of this head is the X operand of the dWw outer product
                    if constexpr (PASS == 1) *reinterpret_cast<fls16x4_t*>(gw_wr + (g0 + hb) * 128) = fl_pack4<false>(es[hb][0], es[hb][1], es[hb][2], es[hb][3]);
#endif
                }
#pragma unroll
                for (int hq = 0; hq < HB / 4; ++hq)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#ifndef FLB_DBG_NOMIX16
                        const fls16x4_t bv = fl_pack4<false>(es[4 * hq][r], es[4 * hq + 1][r], es[4 * hq + 2][r], es[4 * hq + 3][r]);
#pragma unroll
                        for (int gh = 0; gh < H / 4; ++gh) dp[r][gh] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(Awt[gh][g0 / 4 + hq], bv, dp[r][gh], 0, 0, 0);
#else
#pragma unroll
                        for (int k = 0; k < 4; ++k) dp[r][g0 / 4 + hq][k] += es[4 * hq + k][r];
#endif
                    }
            }

            FLB_PHASE();
            if constexpr (PASS == 1) {
                // ---- D += dP . P over this lane's keys ; dWw += dP' P^T, dbw += dP' over the tile's 256 positions
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int g = 0; g < H; ++g) rD[g] = fmaf(dp[r][g >> 2][g & 3], sp[r][g >> 2][g & 3], rD[g]);
#ifndef FLB_DBG_NOGWM
#pragma unroll
                for (int g = 0; g < H; ++g)
                    *reinterpret_cast<fls16x4_t*>(gw_wr + 512 * H + g * 128) =
                        fl_pack4<false>(sp[0][g >> 2][g & 3], sp[1][g >> 2][g & 3], sp[2][g >> 2][g & 3], sp[3][g >> 2][g & 3]);
#endif
            } else {
                // ---- dS' = P (dP - D) (zero for keys >= N: P = 0 there) ; dbl ; dWl += dS' S^T
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int gh = 0; gh < H / 4; ++gh) sp[r][gh] = sp[r][gh] * (dp[r][gh] - Dv[gh]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int g = 0; g < H; ++g) gb[g] += sp[r][g >> 2][g & 3];
#ifndef FLB_DBG_NOGWM
#pragma unroll
                for (int g = 0; g < H; ++g)
                    *reinterpret_cast<fls16x4_t*>(gw_wr + g * 128) =
                        fl_pack4<false>(sp[0][g >> 2][g & 3], sp[1][g >> 2][g & 3], sp[2][g >> 2][g & 3], sp[3][g >> 2][g & 3]);
#endif
            }
#ifndef FLB_DBG_NOGWM
            {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // wave-private tile: the other lanes' packets are read next
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const flu32x4_t xa = *reinterpret_cast<const flu32x4_t*>(gw_xrd + c * 16);
                    const flu32x4_t yb = *reinterpret_cast<const flu32x4_t*>(gw_yrd + c * 16);
                    gwacc[(2 * c) & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(fls16x4_t, (flu32x2_t){xa[0], xa[1]}),
                                                                                     __builtin_bit_cast(fls16x4_t, (flu32x2_t){yb[0], yb[1]}), gwacc[(2 * c) & 3], 0, 0, 0);
                    gwacc[(2 * c + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(fls16x4_t, (flu32x2_t){xa[2], xa[3]}),
                                                                                         __builtin_bit_cast(fls16x4_t, (flu32x2_t){yb[2], yb[3]}), gwacc[(2 * c + 1) & 3], 0, 0, 0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the tile is rewritten by the next key tile
            }
#endif
            FLB_PHASE();
            if constexpr (PASS == 2) {
                // ---- dS = Wl^T dS' (bf16 operands, fp32 accumulate) -> bf16 ; store the block ; dQ^T += K^T dS^T
                f32x4_t ds[4][H / 4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#ifndef FLB_DBG_NOMIX16
                    float x[H];
#pragma unroll
                    for (int g = 0; g < H; ++g) x[g] = sp[r][g >> 2][g & 3];
                    fl_mix_16<H, false>(x, Alt, nullptr, ds[r]);
#else
#pragma unroll
                    for (int gh = 0; gh < H / 4; ++gh) ds[r][gh] = sp[r][gh];
#endif
                }
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const fls16x4_t pk = fl_pack4<false>(ds[0][h >> 2][h & 3], ds[1][h >> 2][h & 3], ds[2][h >> 2][h & 3], ds[3][h >> 2][h & 3]);
#ifndef FLB_DBG_NOST
                    __builtin_nontemporal_store(__builtin_bit_cast(flu32x2_t, pk),
                                                reinterpret_cast<flu32x2_t*>(a.dS + (((((long)b * H + h) * nt + qt) * nt + kt) * 64 + lane) * 4));
#endif
#ifndef FLB_DBG_NODQ
                    fls16x4_t ka[DT];
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) ka[dt] = *reinterpret_cast<const fls16x4_t*>(sK16 + h * REC + dt * 512 + lane * 8);
                    flb_dq_mfma<DT>(dQ[h], ka, pk);
#else
                    dQ[h][0][0] += __builtin_bit_cast(float, (unsigned)pk[0] << 16);
#endif
                }
            }
        };

        issue_tiles(kt0, 0);
        // the ragged last key tile of an image (keys >= N) runs the masked instance of the tile code AFTER the loop over the full tiles: one
        // instance per loop keeps the accumulators in place (with both instances inside one loop hipcc copied all 96 of them around
        // every step)
        const int nfull = ((N & 15) != 0 && kt0 + seg == nt) ? seg - 1 : seg;
        auto admit = [&](int i) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of step i have landed (and its stores of step i - 1 are out)
            __builtin_amdgcn_s_barrier();                          // everybody's have, and everybody is done with the stage refilled next
            asm volatile("" ::: "memory");
            if (i + 1 < seg) issue_tiles(kt0 + i + 1, (i + 1) & 1);
        };
        for (int i = 0; i < nfull; ++i) {
            admit(i);
            if (wvalid) tile(i, std::false_type{});
        }
        if (nfull < seg) {
            admit(nfull);
            if (wvalid) tile(nfull, std::true_type{});
        }
        __builtin_amdgcn_s_barrier();              // the last stage has been read by everybody: the next segment may refill it

        // ---- partial results of this segment -> the major's slot
        const int first_wg = (int)(((long)bm * nt) / a.spw);
        const int slot = (int)blockIdx.x - first_wg;
        if constexpr (PASS == 1) {
#pragma unroll
            for (int g = 0; g < H; ++g) {
                float d = rD[g];
                d += __shfl_xor(d, 16, 64);
                d += __shfl_xor(d, 32, 64);
                if (lane < 16 && wvalid) a.ws_d[((((long)bm * FL_MAXSLOT + slot) * NW + wave) * H + g) * 16 + lane] = d;
            }
        } else {
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

    // ---- weight-gradient partials of this wave -> its row of ws_w
    {
        constexpr int NWG = 2 * (H * H + H);
        float* row = a.ws_w + ((long)blockIdx.x * NW + wave) * NWG + ((PASS == 1) ? (H * H + H) : 0);
        // D[m = g][n]: lane holds rows 4 (lane >> 4) + r of column lane & 15; columns < H = dW[g][h], column H = the bias gradient (pass 1)
        const f32x4_t dsum = (gwacc[0] + gwacc[1]) + (gwacc[2] + gwacc[3]);
        const int nn = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int g = 4 * (lane >> 4) + r;
            // pass 2 accumulated dS' . (log2(e) S)^T: the weight gradient carries ln 2, the bias gradient does not
            if (g < H && nn < H) row[g * H + nn] = (PASS == 2) ? FL_LN2 * dsum[r] : dsum[r];
            if (PASS == 1 && g < H && nn == H) row[H * H + g] = dsum[r];
        }
        if constexpr (PASS == 2) {
#pragma unroll
            for (int g = 0; g < H; ++g) {
                const float v = spe_wave_sum(gb[g]);
                if (lane == 0) row[H * H + g] = v;
            }
        }
    }
}

// D rows [B][Np][H] = sum over the slots of each (major, wave) ; rows >= N zero.  One thread per element, fixed order.
__global__ __launch_bounds__(256) void bwdq_rows_merge_kernel(const float* __restrict__ ws, float* __restrict__ out, int B, int H, int N, int nt, int Np,
                                                              int nmaj, int spw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * Np * H) return;
    const int g = (int)(i % H); const long bq = i / H; const int q = (int)(bq % Np), b = (int)(bq / Np);
    if (q >= N) { out[i] = 0.f; return; }
    const int qt = q >> 4, mj = qt / FLB_NW, wave = qt % FLB_NW;
    const long bm = (long)b * nmaj + mj;
    const int first_wg = (int)((bm * nt) / spw), last_wg = (int)(((bm + 1) * nt - 1) / spw);
    const float* src = ws + (((bm * FL_MAXSLOT) * FLB_NW + wave) * H + g) * 16 + (q & 15);
    float acc = 0.f;
    for (int sl = 0; sl <= last_wg - first_wg; ++sl) acc += src[(long)sl * FLB_NW * H * 16];
    out[i] = acc;
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

static inline int flb_dsteps(int dh, int* tail) {
    const int rem = dh % 32, full = dh / 32 + (rem > 16 ? 1 : 0);
    *tail = (rem > 0 && rem <= 16) ? 1 : 0;
    return full + *tail;
}

template <int H, int DSTEPS, bool TAIL16, int PASS>
static int launch_bwdq(const FlashBwdArgs& a, int nwg, bool drop, hipStream_t st) {
    constexpr int FULL = DSTEPS - (TAIL16 ? 1 : 0), DT = 2 * FULL + (TAIL16 ? 1 : 0), REC = DT * 512;
    constexpr int smem = 2 * ((PASS == 2) ? 3 : 2) * H * REC + 256 + FLB_NW * 1024 * H;
    if (smem > 160 * 1024) return -2;
    static bool attr_set[2] = {false, false};
    const void* fn = drop ? reinterpret_cast<const void*>(&talking_bwdq_kernel<H, DSTEPS, TAIL16, true, PASS>)
                          : reinterpret_cast<const void*>(&talking_bwdq_kernel<H, DSTEPS, TAIL16, false, PASS>);
    if (!attr_set[drop]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set[drop] = true;
    }
    if (drop) hipLaunchKernelGGL((talking_bwdq_kernel<H, DSTEPS, TAIL16, true, PASS>), dim3(nwg), dim3(64 * FLB_NW), smem, st, a);
    else hipLaunchKernelGGL((talking_bwdq_kernel<H, DSTEPS, TAIL16, false, PASS>), dim3(nwg), dim3(64 * FLB_NW), smem, st, a);
    SPE_CHECK_LAUNCH();
    return 0;
}

template <int PASS>
static int dispatch_bwdq(const FlashBwdArgs& a, int H, int dh, int nwg, hipStream_t st) {
    int tail; const int ds = flb_dsteps(dh, &tail);
    const bool drop = a.p_drop > 0.f;
#define SPE_BWDQ(HH)                                                                             \
    if (H == HH && ds == 2 && tail) return launch_bwdq<HH, 2, true, PASS>(a, nwg, drop, st);     \
    if (H == HH && ds == 2 && !tail) return launch_bwdq<HH, 2, false, PASS>(a, nwg, drop, st);   \
    if (H == HH && ds == 1 && tail) return launch_bwdq<HH, 1, true, PASS>(a, nwg, drop, st);     \
    if (H == HH && ds == 1 && !tail) return launch_bwdq<HH, 1, false, PASS>(a, nwg, drop, st);
#ifdef FLB_ONLY_CFG2          // register audits: only the cfg2 instance
    if (H == 8 && ds == 2 && tail) return launch_bwdq<8, 2, true, PASS>(a, nwg, drop, st);
#else
    SPE_BWDQ(8)
    SPE_BWDQ(4)
#endif
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
    a.ws_d = nullptr; a.ws_q = nullptr; a.ws_w = nullptr; a.dS = nullptr;
    a.keepbits = (p_drop > 0.f) ? reinterpret_cast<const unsigned*>(keepbits) : nullptr;
    a.B = B; a.N = N; a.nt = nt; a.nmaj = p.nmaj; a.spw = p.spw; a.total = p.total; a.p_drop = p_drop;
    return 0;
}

extern "C" int spe_talking_bwdq_pass1(const void* Qf, const void* dOf, const void* Kf, const void* Vf, const float* Wl, const float* Ww,
                                      const float* c0, int Np, float* ws_d, float* ws_w, float* Drows, const void* keepbits, int B, int H, int N, int dh,
                                      int nwg, float p_drop, hipStream_t st) {
    if ((long)B * ((N + 15) / 16) <= 0) return 0;
    FlashBwdArgs a; FlashPlan p;
    int rc = bwdq_fill(a, p, Qf, dOf, Kf, Vf, nullptr, Wl, Ww, c0, Np, keepbits, B, H, N, dh, nwg, p_drop);
    if (rc != 0) return rc;
    if (!ws_d || !ws_w || !Drows) return -2;
    a.ws_d = ws_d; a.ws_w = ws_w;
    rc = dispatch_bwdq<1>(a, H, dh, p.nwg, st);
    if (rc != 0) return rc;
    const long n = (long)B * Np * H;
    hipLaunchKernelGGL(bwdq_rows_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws_d, Drows, B, H, N, a.nt, Np, p.nmaj, p.spw);
    SPE_CHECK_LAUNCH();
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
    rc = dispatch_bwdq<2>(a, H, dh, p.nwg, st);
    if (rc != 0) return rc;
    const int DT = (dh + 15) / 16;
    const long nvec = (long)B * p.nmaj * FLB_NW * H * DT * 64;
    hipLaunchKernelGGL(bwdq_dq_merge_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st, ws_q, dq, reinterpret_cast<unsigned short*>(dq16),
                       ob, on, oh, B, H, N, a.nt, dh, DT, p.nmaj, p.spw, scale, nvec);
    SPE_CHECK_LAUNCH();
    return 0;
}
