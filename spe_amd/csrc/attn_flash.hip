// Flash forward of the talking-heads attention (K4 of SURVEY.md section 2.2; reference models/cait.py:377-389): no N x N tensor in HBM.
//
//   spe_talking_flash_fwd:  O = dropout(Ww softmax_k(Wl S + bl) + bw) V, S = scale q k^T, with the row statistics of the statistics pass
//            (attn_stats.hip: spe_talking_stats + spe_attn_merge_rows).  P' goes from the mix straight into the P' V matrix instructions, O accumulates
//            in registers.  With dropout it also stores the keep flags of every tile (1 bit per element) for the backward kernels (attn_flash_bwd.hip).
//
// Two waves per SIMD at <= 256 registers each (all ordinary VGPRs); the streamed operand tiles of a step are staged in LDS with
// global_load_lds_dwordx4, multi-buffered: the DMA of step i + 1 is issued right after the barrier that admits step i and has the whole step to
// land.  Work is split by a flattened (batch, major tile group, streamed tile) numbering cut into equal ranges (attn_flash_common.h: fl_plan), so 260
// tile groups on 256 CUs cost no second round; a range's partial O goes to slot workspaces that one small merge kernel sums in fixed order (bitwise
// reproducible: no floating-point atomics anywhere).  (Round 4 / 5 also ran the dV pass of the backward on this skeleton - template parameter KV,
// spe_talking_flash_dv; the key-major backward kernel took it over: profiles/HISTORY_r05.md.)
#include "attn_flash_common.h"

struct FlashFwdArgs {
    // fragment records (DT * 512 B each): Qf = q (resident tiles, fp16), Kf = k (streamed, fp16), V16 = v (streamed, fp16)
    const unsigned char* Qf; const unsigned char* Kf; const unsigned char* V16;
    const float* Wl; const float* Ww; const float* bw;
    const float* c0;                     // [B][Np][H]: bl * log2(e) - m + log2(1 / l), rows >= N zero (spe_attn_merge_rows)
    int Np;                              // rows per image of c0
    float* ws_o;                         // partial O * 2^8: [B * nmaj][FL_MAXSLOT][NW waves][QS][H][DT][64 lanes][4]
    int B, N, nt, nmaj, spw; long total;
    float p_drop; uint64_t seed, offset;
    // forward with dropout (optional): the keep flags of every 16 x 16 tile, one dword per lane - bit hp * 8 + 2 r + e = key 4 (lane >> 4) + r,
    // head 2 hp + e of query lane & 15 - as [B][nt (q-tile)][nt (key tile)][64]: the lane layout of the q-major passes, which then load their
    // masks (34 MB per block at cfg2) instead of regenerating them (4 Philox calls per lane and tile, ~60 us per pass)
    unsigned* keepbits;
};

// Geometry: 8 waves per workgroup, one q-tile per wave: 8 q-tiles = 128 queries per workgroup (the K / V tiles a workgroup streams are 24 KB per
// step at cfg2 and a CU loads ~10 B / clk: fewer queries per workgroup would make the kernel load-bound).  Two waves per SIMD, <= 256 registers
// each, all of them ordinary VGPRs (a one-wave-per-SIMD variant with the accumulators in AccVGPRs was built in round 4: hipcc gives every matrix
// result an AccVGPR then and copies it out for the vector instructions - 1060 moves per step; profiles/r04_flash_variants.txt).
// SKEW.  A step has a matrix-heavy half H1 (K Q^T, the fp32 Wl mix: ~1000 matrix cycles, few vector instructions)
// and a vector-heavy half H2 (the fp16 Ww mix, dropout, packs, P' V).  With every wave running H1, H2 between the same barriers the
// two waves of a SIMD are in lock step: they queue on the matrix pipe in H1 and leave it idle in H2 (measured, rocprofv3 SQ
// counters at cfg2: matrix pipe 53 % + vector 46 % busy = no overlap).  So the two wave groups place their ONE barrier per step at
// different points of the same instruction stream - waves 0-3 in front of H1, waves 4-7 in front of H2 - which holds the groups
// half a step apart: between two barriers group 0 runs H1(k), H2(k) and group 1 H2(k), H1(k + 1).  Three K buffers, two V buffers.
#define FLF_NW 8
#define FLF_MAJ 8                        // q-tiles per workgroup
#define FLF_HB 4                         // heads per operand-fragment batch of the K Q^T products
#define FLF_QS (FLF_MAJ / FLF_NW)

template <int H, int DSTEPS, bool TAIL16, bool DROP>
__global__ __launch_bounds__(64 * FLF_NW, FLF_NW / 4) void talking_flash_fwd_kernel(FlashFwdArgs a) {
    constexpr int FULL = DSTEPS - (TAIL16 ? 1 : 0), DT = 2 * FULL + (TAIL16 ? 1 : 0), REC = DT * 512;
    constexpr int NW = FLF_NW, QS = FLF_QS;
    constexpr int TILEB = H * REC;                 // one operand, one 16-row tile, all heads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];           // [K 0][K 1][K 2][V 0][V 1][8 resident tiles x TILEB]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const unsigned char* sQ = smem + 5 * TILEB + wave * (QS * TILEB);
    const unsigned ldsQ = lds0 + 5 * TILEB + wave * (QS * TILEB);
    const int grp = wave >> 2;                     // 1: this wave's barrier sits between H1 and H2
    const int nt = a.nt, N = a.N;

    float Al4[H / 4][H];
    fl_mixA_f32<H, false>(a.Wl, lane, Al4);
    fls16x4_t Aw[H / 4][H / 4];
    fl_mixA_16<H, false, true>(a.Ww, lane, 1.0f, Aw);
    f32x4_t vbws[H / 4];                           // bw * 2^8: the mix runs on P * 2^8
#pragma unroll
    for (int gh = 0; gh < H / 4; ++gh)
#pragma unroll
        for (int i = 0; i < 4; ++i) vbws[gh][i] = a.bw[4 * gh + i] * FL_PD_SCALE;
    constexpr int NP = TILEB / 1024, NPW = (NP + NW - 1) / NW;     // 1-KB pieces of an operand tile, pieces per wave
    unsigned voff[NPW];                                            // byte offset of this lane's 16 B of piece i * NW + wave inside a (b, tile) image
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int o = (i * NW + wave) * 1024 + lane * 16;
        voff[i] = (unsigned)((o / REC) * nt * REC + o % REC);
    }

    const long s_begin = (long)blockIdx.x * a.spw;
    long s_end = s_begin + a.spw; if (s_end > a.total) s_end = a.total;
    long s = s_begin;
    while (s < s_end) {
        const int bm = (int)(s / nt), kt0 = (int)(s % nt);
        int seg = nt - kt0; if (seg > s_end - s) seg = (int)(s_end - s);
        const int b = bm / a.nmaj, mj = bm % a.nmaj;
        const int qt0 = (mj * NW + wave) * QS;                  // this wave's first q-tile (wave-uniform)
        const bool wvalid = qt0 < nt;

        // ---- this wave's Q records -> its LDS area (QS x H records; a 1-KB piece may straddle two records: per-lane source offset)
#pragma unroll
        for (int u = 0; u < QS; ++u) {
            constexpr int NPQ = TILEB / 1024;
            const unsigned char* qbase = a.Qf + ((long)b * H * nt + min(qt0 + u, nt - 1)) * REC;          // wave-uniform
#pragma unroll
            for (int p = 0; p < NPQ; ++p) {
                const int o = p * 1024 + lane * 16;
                fl_glds16_s(qbase, (unsigned)((o / REC) * nt * REC + o % REC), ldsQ + u * TILEB + p * 1024);
            }
        }
        // one operand tile (all heads) of key tile kt -> LDS byte address dst: TILEB / 1024 pieces shared by the waves.  The source is
        // (uniform base of the (b, tile)) + (per-lane 32-bit offset of the piece, computed once): no 64-bit vector arithmetic per piece
        auto issue_tile = [&](const unsigned char* base, int kt, unsigned dst) {
            const unsigned char* tb = base + ((long)b * H * nt + kt) * REC;
            if constexpr (NP % NW == 0 && NPW <= 4) fl_glds16_run<NPW, NW * 1024>(tb, voff, dst + wave * 1024);
            else {
#pragma unroll
                for (int i = 0; i < NPW; ++i) {
                    const int p = i * NW + wave;
                    if (NP % NW != 0 && p >= NP) break;
                    fl_glds16_s(tb, voff[i], dst + p * 1024);
                }
            }
        };
        issue_tile(a.Kf, kt0, lds0);

        // ---- row constants and accumulators
        f32x4_t c0v[QS][H / 4];
        int qrow[QS];
#pragma unroll
        for (int u = 0; u < QS; ++u) {
            qrow[u] = (qt0 + u) * 16 + (lane & 15);                         // this lane's query
            const float* cp = a.c0 + ((long)b * a.Np + (wvalid ? min(qrow[u], N - 1) : 0)) * H;
#pragma unroll
            for (int gh = 0; gh < H / 4; ++gh) {
                c0v[u][gh] = *reinterpret_cast<const f32x4_t*>(cp + 4 * gh);
#pragma unroll
                for (int i = 0; i < 4; ++i) c0v[u][gh][i] += 8.0f;       // exp2(. + 8) = P * 2^8
            }
        }
        f32x4_t O[QS][H][DT];
#pragma unroll
        for (int u = 0; u < QS; ++u)
#pragma unroll
            for (int g = 0; g < H; ++g)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) O[u][g][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        // barrier j admits K(j) / V(j - 1) and everything older; after it this wave's share of K(j + 1) and V(j) goes out, with a whole
        // step to land.  Group 0 passes barrier i + 1 in front of H1(i), group 1 between H1(i) and H2(i); barrier 0 is common.
        auto step_barrier = [&](int j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's DMA pieces (and its Q records) have landed
            __builtin_amdgcn_s_barrier();                                // everybody's have, and the buffers refilled below have been read
            asm volatile("" ::: "memory");
            if (j + 1 < seg) issue_tile(a.Kf, kt0 + j + 1, lds0 + ((j + 1) % 3) * TILEB);
            if (j < seg) issue_tile(a.V16, kt0 + j, lds0 + (3 + (j & 1)) * TILEB);
        };
        step_barrier(0);
        for (int i = 0; i < seg; ++i) {
            if (grp == 0) step_barrier(i + 1);
            if (!wvalid) { if (grp != 0) step_barrier(i + 1); continue; }
            const unsigned char* sK = smem + (i % 3) * TILEB;
            const unsigned char* sV = smem + (3 + (i & 1)) * TILEB;
            fls16x4_t bv[QS][4][H / 4];             // fp16(P * 2^8): [q-tile][key r][4 heads] - all that crosses from H1 to H2
            // ---- H1: S^T = K Q^T (M = keys, N = queries): lane = (query l & 15, keys 4 (l >> 4) + r) ; P * 2^8 = exp2(Wl S + c0 + 8)
#pragma unroll
            for (int u = 0; u < QS; ++u) {
                // head-outer order: as soon as the raw scores of head h exist, its 4 x H / 4 contributions to the fp32 Wl mix are issued -
                // 8 independent 4x4x1 instructions per head that fill the matrix pipe while the next head's K Q^T products wait for
                // their LDS operands; the mix accumulators (32 registers) replace the 32 registers of raw scores, which now live for
                // one head only.  (The key-outer order below ends the K Q^T phase in a dependency wall: every mix chain needs all heads.)
                f32x4_t sp[4][H / 4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int gh = 0; gh < H / 4; ++gh) sp[r][gh] = c0v[u][gh];
#pragma unroll
                for (int h0 = 0; h0 < H; h0 += FLF_HB) {
                    flu32x4_t kf[FLF_HB][FULL ? FULL : 1], qf[FLF_HB][FULL ? FULL : 1];
                    fls16x4_t kt16[FLF_HB], qt16[FLF_HB];
#pragma unroll
                    for (int hb = 0; hb < FLF_HB; ++hb) {
                        const unsigned char* kr = sK + (h0 + hb) * REC;
                        const unsigned char* qr = sQ + (u * H + h0 + hb) * REC;
#pragma unroll
                        for (int st = 0; st < FULL; ++st) {
                            kf[hb][st] = *reinterpret_cast<const flu32x4_t*>(kr + st * 1024 + lane * 16);
                            qf[hb][st] = *reinterpret_cast<const flu32x4_t*>(qr + st * 1024 + lane * 16);
                        }
                        if constexpr (TAIL16) {
                            kt16[hb] = *reinterpret_cast<const fls16x4_t*>(kr + FULL * 1024 + lane * 8);
                            qt16[hb] = *reinterpret_cast<const fls16x4_t*>(qr + FULL * 1024 + lane * 8);
                        }
                    }
#pragma unroll
                    for (int hb = 0; hb < FLF_HB; ++hb) {
                        f32x4_t c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int st = 0; st < FULL; ++st) c = fl_mfma32<true>(kf[hb][st], qf[hb][st], c);
                        if constexpr (TAIL16) {
                            // the 16-wide tail step as a 16x16x16 instruction into its OWN accumulator (an accumulate chain never mixes two MFMA
                            // shapes - attn_stats.hip), added on the vector pipe
                            const f32x4_t t = fl_mfma16<true>(kt16[hb], qt16[hb], (f32x4_t){0.f, 0.f, 0.f, 0.f});
                            c = (FULL > 0) ? c + t : t;
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int gh = 0; gh < H / 4; ++gh) sp[r][gh] = __builtin_amdgcn_mfma_f32_4x4x1f32(Al4[gh][h0 + hb], c[r], sp[r][gh], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int hh = 0; hh < H / 4; ++hh)          // P * 2^8 <= 256: no saturation needed
                        bv[u][r][hh] = fl_pack4_f16(fl_exp2(sp[r][hh][0]), fl_exp2(sp[r][hh][1]), fl_exp2(sp[r][hh][2]), fl_exp2(sp[r][hh][3]));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (grp != 0) step_barrier(i + 1);
            // ---- H2: P' * 2^8 = Ww (P * 2^8) + bw * 2^8 ; dropout ; fp16 B operands ; O^T[d][q] += V^T[d][key] P'^T[key][q]
#pragma unroll
            for (int u = 0; u < QS; ++u) {
                f32x4_t pr[4][H / 4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int gh = 0; gh < H / 4; ++gh) {
                        f32x4_t d = vbws[gh];
#pragma unroll
                        for (int hh = 0; hh < H / 4; ++hh)
                            d = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(flf16x4_t, Aw[gh][hh]), __builtin_bit_cast(flf16x4_t, bv[u][r][hh]), d, 0, 0, 0);
                        pr[r][gh] = d;
                    }
                if constexpr (DROP) {
                    const uint32_t thr = (uint32_t)(a.p_drop * 65536.0f);
                    const float inv = 1.0f / (1.0f - a.p_drop);
                    uint32_t kbw = 0u;
#pragma unroll
                    for (int hp = 0; hp < H / 2; ++hp) {
                        const int g0 = 2 * hp, g1 = 2 * hp + 1;
                        // lane = (query, 4 consecutive keys): one counter, lots r
                        uint32_t o[4];
                        fl_keep_lots<H>(a.seed, a.offset, b, hp, qrow[u], (kt0 + i) * 16 + 4 * (lane >> 4), N, o);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool e0 = fl_lot(o, 0, r) >= thr, e1 = fl_lot(o, 1, r) >= thr;
                            pr[r][g0 >> 2][g0 & 3] *= e0 ? inv : 0.f;
                            pr[r][g1 >> 2][g1 & 3] *= e1 ? inv : 0.f;
                            kbw |= ((e0 ? 1u : 0u) | (e1 ? 2u : 0u)) << (hp * 8 + 2 * r);
                        }
                    }
                    // the backward kernels load this tile's flags instead of drawing them again
                    if (a.keepbits && wvalid && qt0 + u < nt) a.keepbits[(((long)b * nt + (qt0 + u)) * nt + (kt0 + i)) * 64 + lane] = kbw;
                }
#pragma unroll
                for (int g = 0; g < H; ++g) {
                    const fls16x4_t pk = fl_pack4<true>(pr[0][g >> 2][g & 3], pr[1][g >> 2][g & 3], pr[2][g >> 2][g & 3], pr[3][g >> 2][g & 3]);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)     // A = the V16 record (lane: d = dt*16 + (l & 15), keys 4 (l >> 4) + i), B = pk
                        O[u][g][dt] = fl_mfma16<true>(*reinterpret_cast<const fls16x4_t*>(sV + g * REC + dt * 512 + lane * 8), pk, O[u][g][dt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_barrier();              // the last tiles have been read by everybody: the next segment may refill the buffers

        // ---- partial O of this segment -> the major's slot (fragment order: one coalesced 1-KB store per accumulator register group)
        if (wvalid) {
            const int first_wg = (int)(((long)bm * nt) / a.spw);
            const int slot = (int)blockIdx.x - first_wg;
#pragma unroll
            for (int u = 0; u < QS; ++u) {
                if (qt0 + u >= nt) continue;
                float* dst = a.ws_o + ((((long)bm * FL_MAXSLOT + slot) * NW + wave) * QS + u) * (long)(H * DT * 256);
#pragma unroll
                for (int g = 0; g < H; ++g)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4_t*>(dst + (g * DT + dt) * 256 + lane * 4) = O[u][g][dt];
            }
        }
        s += seg;
    }
}

// Sum of the partial result slots of each (major, wave, tile), times 2^-8 -> O[b, row, g, d] (element strides ob, on, oh) fp32 (+ its bf16 copy /
// low part with the same addressing: the operand of the output projection).  One thread per float4 of the fragment-ordered workspace; fixed order.
__global__ __launch_bounds__(256) void flash_merge_kernel(const float* __restrict__ ws, float* __restrict__ O, long ob, long on, long oh,
                                                          unsigned short* __restrict__ O16, unsigned short* __restrict__ O16lo,
                                                          int B, int H, int N, int nt, int dh, int DT, int nmaj, int spw, long nvec) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    constexpr int QS = FLF_QS;
    const int lane = (int)(i & 63);
    long r = i >> 6;
    const int dt = (int)(r % DT); r /= DT;
    const int g = (int)(r % H); r /= H;
    const int u = (int)(r % QS); r /= QS;
    const int wave = (int)(r % FLF_NW); r /= FLF_NW;
    const int bm = (int)r;
    const int b = bm / nmaj, mj = bm % nmaj;
    const int qt = (mj * FLF_NW + wave) * QS + u;
    const int q = qt * 16 + (lane & 15), d = dt * 16 + 4 * (lane >> 4);
    if (qt >= nt || q >= N || d >= dh) return;
    const int first_wg = (int)(((long)bm * nt) / spw), last_wg = (int)(((long)(bm + 1) * nt - 1) / spw);
    const long slot_stride = (long)FLF_MAJ * H * DT * 256;
    const float* src = ws + (long)bm * FL_MAXSLOT * slot_stride + (((long)wave * QS + u) * H + g) * (long)(DT * 256) + dt * 256 + lane * 4;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl <= last_wg - first_wg; ++sl) acc += *reinterpret_cast<const f32x4_t*>(src + sl * slot_stride);
    acc *= (1.0f / FL_PD_SCALE);
    const long oi = (long)b * ob + (long)q * on + (long)g * oh + d;
    if (d + 3 < dh && ((ob | on | oh) & 3) == 0) {        // the common case: one 16-B and up to two 8-B stores per lane
        if (O && ((reinterpret_cast<uintptr_t>(O) & 15) == 0)) *reinterpret_cast<f32x4_t*>(O + oi) = acc;
        else if (O) { O[oi] = acc[0]; O[oi + 1] = acc[1]; O[oi + 2] = acc[2]; O[oi + 3] = acc[3]; }
        if (O16) {
            unsigned short hi[4], lo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { hi[k] = spe_f2bf(acc[k]); lo[k] = spe_f2bf(acc[k] - spe_bf2f(hi[k])); }
            const uint2 h2 = make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
            if ((reinterpret_cast<uintptr_t>(O16) & 7) == 0) *reinterpret_cast<uint2*>(O16 + oi) = h2;
            else { O16[oi] = hi[0]; O16[oi + 1] = hi[1]; O16[oi + 2] = hi[2]; O16[oi + 3] = hi[3]; }
            if (O16lo) {
                const uint2 l2 = make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16));
                if ((reinterpret_cast<uintptr_t>(O16lo) & 7) == 0) *reinterpret_cast<uint2*>(O16lo + oi) = l2;
                else { O16lo[oi] = lo[0]; O16lo[oi + 1] = lo[1]; O16lo[oi + 2] = lo[2]; O16lo[oi + 3] = lo[3]; }
            }
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (d + k >= dh) break;
        if (O) O[oi + k] = acc[k];
        if (O16) {
            const unsigned short hi = spe_f2bf(acc[k]);
            O16[oi + k] = hi;
            if (O16lo) O16lo[oi + k] = spe_f2bf(acc[k] - spe_bf2f(hi));
        }
    }
}

template <int H, int DSTEPS, bool TAIL16>
static int launch_flash_fwd(const FlashFwdArgs& a, int nwg, bool drop, hipStream_t st) {
    constexpr int FULL = DSTEPS - (TAIL16 ? 1 : 0), DT = 2 * FULL + (TAIL16 ? 1 : 0), REC = DT * 512;
    constexpr int smem = (5 + FLF_MAJ) * H * REC;
    if (smem > 160 * 1024) return -2;              // H * head dim too large for the resident tiles + the stage buffers: use the materialising path
    static bool attr_set[2] = {false, false};
    const void* fn = drop ? reinterpret_cast<const void*>(&talking_flash_fwd_kernel<H, DSTEPS, TAIL16, true>)
                          : reinterpret_cast<const void*>(&talking_flash_fwd_kernel<H, DSTEPS, TAIL16, false>);
    if (!attr_set[drop]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set[drop] = true;
    }
    if (drop) hipLaunchKernelGGL((talking_flash_fwd_kernel<H, DSTEPS, TAIL16, true>), dim3(nwg), dim3(64 * FLF_NW), smem, st, a);
    else hipLaunchKernelGGL((talking_flash_fwd_kernel<H, DSTEPS, TAIL16, false>), dim3(nwg), dim3(64 * FLF_NW), smem, st, a);
    SPE_CHECK_LAUNCH();
    return 0;
}

static inline int flash_dsteps(int dh, int* tail) {
    const int rem = dh % 32, full = dh / 32 + (rem > 16 ? 1 : 0);
    *tail = (rem > 0 && rem <= 16) ? 1 : 0;
    return full + *tail;
}

// C-ABI: see include/spe_hip.h
extern "C" int spe_talking_flash_plan(int B, int N, int nwg, int* steps_per_wg, int* nwg_used, int* nmajor, int* rows_padded) {
    const int nt = (N + 15) / 16;
    if ((long)B * nt <= 0 || nwg <= 0) { *steps_per_wg = 0; *nwg_used = 0; *nmajor = 0; *rows_padded = 0; return 0; }
    const FlashPlan p = fl_plan(B, nt, FLF_MAJ, nt, nwg);
    *steps_per_wg = p.spw; *nwg_used = p.nwg; *nmajor = p.nmaj; *rows_padded = nt * 16 + 64;
    return 0;
}

// C-ABI: see include/spe_hip.h
extern "C" int spe_talking_flash_fwd(const void* Qf, const void* Kf, const void* V16, const float* Wl, const float* Ww, const float* bw,
                                     const float* c0, int Np, float* ws, float* O, void* O16, void* O16lo, void* keepbits, int B, int H, int N,
                                     int dh, int nwg, float p_drop, uint64_t seed, uint64_t offset, hipStream_t st) {
    const int nt = (N + 15) / 16;
    if ((long)B * nt <= 0) return 0;
    if (dh < 1 || dh > 64 || nwg <= 0 || (O16lo && !O16) || Np < nt * 16 + 64) return -2;
    int tail; const int ds = flash_dsteps(dh, &tail);
    const FlashPlan p = fl_plan(B, nt, FLF_MAJ, nt, nwg);
    FlashFwdArgs a;
    a.Qf = (const unsigned char*)Qf; a.Kf = (const unsigned char*)Kf; a.V16 = (const unsigned char*)V16;
    a.Wl = Wl; a.Ww = Ww; a.bw = bw; a.c0 = c0; a.Np = Np; a.ws_o = ws;
    a.B = B; a.N = N; a.nt = nt; a.nmaj = p.nmaj; a.spw = p.spw; a.total = p.total;
    a.p_drop = p_drop; a.seed = seed; a.offset = offset; a.keepbits = reinterpret_cast<unsigned*>(p_drop > 0.f ? keepbits : nullptr);
    const bool drop = p_drop > 0.f;
    int rc = -2;
#define SPE_FLASH_FWD(HH)                                                                   \
    if (H == HH && ds == 2 && tail) rc = launch_flash_fwd<HH, 2, true>(a, p.nwg, drop, st);       \
    else if (H == HH && ds == 2 && !tail) rc = launch_flash_fwd<HH, 2, false>(a, p.nwg, drop, st); \
    else if (H == HH && ds == 1 && tail) rc = launch_flash_fwd<HH, 1, true>(a, p.nwg, drop, st);  \
    else if (H == HH && ds == 1 && !tail) rc = launch_flash_fwd<HH, 1, false>(a, p.nwg, drop, st);
    SPE_FLASH_FWD(8) else SPE_FLASH_FWD(4)
#undef SPE_FLASH_FWD
    if (rc != 0) return rc;
    const int DT = (dh + 15) / 16;
    const long nvec = (long)B * p.nmaj * FLF_MAJ * H * DT * 64;
    hipLaunchKernelGGL(flash_merge_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st, ws, O, (long)N * H * dh, (long)H * dh, (long)dh,
                       reinterpret_cast<unsigned short*>(O16), reinterpret_cast<unsigned short*>(O16lo), B, H, N, nt, dh, DT, p.nmaj, p.spw, nvec);
    SPE_CHECK_LAUNCH();
    return 0;
}
