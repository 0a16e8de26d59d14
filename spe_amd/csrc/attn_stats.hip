// Statistics pass of the fused talking-heads attention (reference models/cait.py:377-383: S = scale q k^T, S' = proj_l(S), softmax over keys),
// its merge, and the small launches around the attention kernels (fragment pack, weight-gradient reduce).
//
// The N x N score tensors never exist in HBM: every (q-tile, k-tile) step recomputes the raw scores of ALL heads with MFMA and mixes them in
// registers.  Tile = 16 keys x 16 queries per step, v_mfma_f32_16x16x32_f16 in the swapped orientation S^T = K_tile . Q_tile^T (M = keys, N = queries,
// K = head dim in 32-wide steps).  In the 16x16 C layout a lane owns ONE query column (q = lane & 15) and 4 consecutive keys ((lane>>4)*4 + r), so
// the per-row softmax state (running max, running sum) is lane-local - 2 registers per head, no cross-lane traffic in the key loop - and the same
// acc[h][r] index across heads is the same (q, key) element: the H x H head mix is lane-local (see mix_keys_f32).
//
// Operands come from "row-fragment" packed fp16 arrays produced by spe_attn_pack_multi (one 16-B load per lane per MFMA operand, 1 KB contiguous
// per wave): X_f[b][h][tile16][dstep][lane][8]; q carries scale * log2(e), so every exponential is a bare v_exp_f32.
//
// This kernel computes the partial (max, sum) of softmax_k(S'_g) per (b, g, q); spe_attn_merge_rows turns them into the row constants
// c0 = bl log2(e) - max - log2(sum) that the flash forward (attn_flash.hip) and the two backward kernels (attn_flash_bwd.hip) fold into the mix's
// addend.  (Rounds 1-5 kept three more modes of this skeleton - the P'd write pass and the two backward passes with dS / D in HBM; they were
// replaced by the flash kernels and are gone: profiles/HISTORY_r05.md.)
//
// Work partition: the (b, q-tile pair, k-tile) steps are flattened q-major and split evenly over the workgroups (two per CU, 4 waves each); a
// workgroup's range covers 1-3 q-tile pairs ("segments"): waves (0, 1) work on q-tile 2p, waves (2, 3) on q-tile 2p + 1, both pairs walk the same
// key tiles in step (the second request of a K fragment is served by the vector L1).  Per-segment row statistics go to a workspace indexed by
// (q-tile, slot = workgroup - first workgroup of the q-tile) and are merged by spe_attn_merge_rows.
#include "common.h"
#include "attn_pack.h"

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8m_t __attribute__((ext_vector_type(8)));

struct StatsArgs {
    const u32x4_t* Qf; const u32x4_t* Kf;
    const float* Wl; const float* bl;
    float* ws_stats;                                      // partial row stats [B*nt][MAXSLOT][H][16][2]
    int B, N, nt;                                         // nt = ceil(N/16) tiles per axis
    int steps_per_wg;
};

#define FUSED_MAXSLOT 8
#define FUSED_QP 2            // q-tiles per workgroup (measured at cfg2: 0.135 -> 0.126 ms against one q-tile per workgroup)
__host__ __device__ __forceinline__ int fused_npair(int nt) { return (nt + FUSED_QP - 1) / FUSED_QP; }

// Key chunks bound to XCDs.  A workgroup with blockIdx b runs on XCD b % 8 (8 private 4 MB L2s).  The K and V
// fragments of an image are 4-8 MB; when every workgroup sweeps all keys each L2 thrashes on them (rocprof: 1.9 GB of
// L2 fetches per backward launch against 34 MB of operands).  So the key tiles are cut into NCH chunks and XCD x only
// works on chunk x % NCH: its L2 holds 1/NCH of the K/V fragments.  Within a chunk the (b, q-tile, k-tile) steps are
// flattened q-major and split evenly over the chunk's workgroups as before.
#define FUSED_NCH 4
__host__ __device__ __forceinline__ int fused_nch(int nt) { return (nt >= 16 * FUSED_NCH) ? FUSED_NCH : 1; }
__host__ __device__ __forceinline__ int fused_kbeg(int c, int nt, int nch) { return (int)((long)c * nt / nch); }

#define EXP2(x) __builtin_amdgcn_exp2f(x)
#define SPE_LOG2E 1.4426950408889634f

// The softmax-input mix S' = Wl S + bl on the matrix pipe IN FP32: v_mfma_f32_4x4x1_16b_f32 computes, in each of its 16 blocks of
// 4 lanes, the outer product D[i][j] += A[i] B[j] with D[i][.] in register i of lane 4b + j (probed on gfx950:
// tools/debug/probe_mfma4x4.hip) - i.e. register i of a lane accumulates (A of lane 4b + i) * (the lane's OWN B).  With
// A := W[4gh + (lane & 3)][h] (a per-lane constant) and B := the lane's score of head h, one instruction adds head h's
// contribution to output heads 4gh .. 4gh+3 of the lane's own (query, key) element: H * H/4 instructions per key are the whole
// H x H mix, bit-for-bit the fmaf chain the packed FMAs computed (an f32 MFMA is a k-ordered fmaf chain) at the same
// FLOP rate (64 / clk / SIMD) - but on the pipe that is ~10 % busy instead of the one that bounds these kernels.  Measured
// (cfg2, isolated): removing the packed-FMA mix altogether is worth 0.049 of the statistics pass' 0.176 ms.
template <int H>
__device__ __forceinline__ void mixA4_build(const float* __restrict__ W, int lane, float (&A)[H / 4][H]) {
#pragma unroll
    for (int gh = 0; gh < H / 4; ++gh)
#pragma unroll
        for (int h = 0; h < H; ++h) A[gh][h] = W[(4 * gh + (lane & 3)) * H + h];
}
// out[r][gh][i] = c[4gh + i] + sum_h W[4gh + i][h] s[h][r]   (r: the lane's 4 keys)
template <int H>
__device__ __forceinline__ void mix_keys_f32(const f32x4_t (&s)[H], const float (&A)[H / 4][H], const float (&c)[H], f32x4_t (&out)[4][H / 4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int gh = 0; gh < H / 4; ++gh) out[r][gh] = (f32x4_t){c[4 * gh], c[4 * gh + 1], c[4 * gh + 2], c[4 * gh + 3]};
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int gh = 0; gh < H / 4; ++gh) out[r][gh] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[gh][h], s[h][r], out[r][gh], 0, 0, 0);
}

// Fragment record of one (b, h, 16-row tile): FULL = DSTEPS - TAIL16 steps of 32 head dims (64 lanes x 16 B) followed,
// when TAIL16, by one step of 16 dims (64 lanes x 8 B: the v_mfma_f32_16x16x16_bf16 operand).  dh = 48 is 32 + 16:
// 1.5 KB per record instead of the 2 KB of two padded 32-steps - the score kernels are sensitive to exactly this
// L2 -> register traffic (measured: dh 32 vs 48-padded-to-64 differ by 0.2 ms per block over the four passes).
template <int DSTEPS, bool TAIL16>
__device__ __forceinline__ u32x4_t frag_load(const u32x4_t* __restrict__ base, long rec, int st, int lane) {
    constexpr int FULL = DSTEPS - (TAIL16 ? 1 : 0);
    constexpr int REC8 = FULL * 128 + (TAIL16 ? 64 : 0);           // record size in 8-B units
    const uint2* p = reinterpret_cast<const uint2*>(base) + rec * REC8;
    if (TAIL16 && st == FULL) {
        const uint2 v = p[FULL * 128 + lane];
        return (u32x4_t){v.x, v.y, 0u, 0u};
    }
    return *reinterpret_cast<const u32x4_t*>(p + st * 128 + lane * 2);
}
// The tail step's 8-B operands are zero-extended (frag_load) and go through the same 16x16x32 instruction: lane group
// g then holds k-slots 8g..8g+3 = head dims FULL*32 + 4g..4g+3 in BOTH operands and zeros in slots 8g+4..8g+7, so the
// products line up - the saving of the tail step is its load bytes, the matrix pipe is idle anyway.
// Accumulate chains stay within ONE MFMA shape everywhere in this library: a 16x16x16 MFMA whose SrcC is the destination of the 16x16x32 MFMA issued
// right before it gave run-to-run different results on gfx950 as hipcc (ROCm 7.2) schedules it (profiles/HISTORY_r01_r03.md).
__device__ __forceinline__ f32x4_t frag_mfma(u32x4_t a, u32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8m_t, a), __builtin_bit_cast(f16x8m_t, b), c, 0, 0, 0);
}
// same record, addressed as (uniform byte pointer of the (b, h) row of records) + (32-bit byte offset of the tile's record): the
// row pointers are computed once per segment, so a fragment costs no 64-bit address arithmetic in the tile loop
template <int DSTEPS, bool TAIL16>
__device__ __forceinline__ u32x4_t frag_load_row(const char* __restrict__ row, unsigned recoff, int st, int lane) {
    constexpr int FULL = DSTEPS - (TAIL16 ? 1 : 0);
    if (TAIL16 && st == FULL) {
        const uint2 v = *reinterpret_cast<const uint2*>(row + (recoff + (unsigned)(FULL * 1024) + (unsigned)lane * 8u));
        return (u32x4_t){v.x, v.y, 0u, 0u};
    }
    return *reinterpret_cast<const u32x4_t*>(row + (recoff + (unsigned)(st * 1024) + (unsigned)lane * 16u));
}

template <int H, int DSTEPS, bool TAIL16, int KT>
__global__ __launch_bounds__(256, 2) void talking_stats_kernel(StatsArgs a) {
    constexpr int NFR = H * DSTEPS;                        // fragments (16 B per lane) per q-tile
    constexpr int QP = FUSED_QP, WPQ = 4 / QP;             // q-tiles per workgroup, waves per q-tile
    constexpr int QG = (H >= 4) ? 4 : H;                   // MFMA jobs issued d-step outer / job inner (consecutive instructions independent)
    __shared__ float sred[4 * H * 16 * 2];                 // [4 waves][H][16][2]

    // the wave index as a SCALAR: everything derived from it (macro step, key tile, fragment record addresses, tail masks) then lives
    // in SGPRs and the fragment loads take an SGPR base + one per-lane offset
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nt = a.nt, N = a.N;
    const int nch = fused_nch(nt);
    const int chunk = (blockIdx.x & 7) % nch, wg_j = (blockIdx.x >> 3) * (8 / nch) + (blockIdx.x & 7) / nch;   // index within the chunk
    const int kbeg = fused_kbeg(chunk, nt, nch), klen = fused_kbeg(chunk + 1, nt, nch) - kbeg;
    const int npair = fused_npair(nt);
    const long total_c = (long)a.B * npair * klen;
    const long s_begin = (long)wg_j * a.steps_per_wg;
    long s_end = s_begin + a.steps_per_wg; if (s_end > total_c) s_end = total_c;

    // Scores arrive in the log2 domain (the pack folds scale * log2(e) into the Q fragments): Wl S + bl*log2(e) is log2(e) * S'
    float vbl2[H];
#pragma unroll
    for (int g = 0; g < H; ++g) vbl2[g] = a.bl[g] * SPE_LOG2E;
    float Al4[H / 4][H];                                   // f32 operand of the S' mix (see mix_keys_f32)
    mixA4_build<H>(a.Wl, lane, Al4);

    long s = s_begin;
    while (s < s_end) {
        const int bqp = (int)(s / klen), kt0 = kbeg + (int)(s % klen);
        int seg = kbeg + klen - kt0; if (seg > s_end - s) seg = (int)(s_end - s);
        const int b = bqp / npair, qp = bqp % npair;
        const int qt_own = qp * QP + wave / WPQ;              // this wave's q-tile; past the last tile (odd tile count): the wave idles
        const bool qt_valid = qt_own < nt;
        const int qt = qt_valid ? qt_own : nt - 1;
        // ---- the q-tile's Q fragments live in registers for the whole segment (64 VGPRs at H = 8, dh <= 64)
        u32x4_t qreg[NFR];
        __syncthreads();                                       // sred of the previous segment has been consumed
#pragma unroll
        for (int f = 0; f < NFR; ++f) qreg[f] = frag_load<DSTEPS, TAIL16>(a.Qf, ((long)b * H + f / DSTEPS) * nt + qt, f % DSTEPS, lane);
        // ---- per-lane row state
        float rm[H], rl[H];
#pragma unroll
        for (int g = 0; g < H; ++g) { rm[g] = -INFINITY; rl[g] = 0.f; }

        // A wave's unit of work is a macro step of KT consecutive 16-key tiles against the 16 queries of the q-tile.  Waves of a q-tile take
        // macro steps round-robin.  Operand-fragment staging registers for one batch of H head jobs, and the batch loader:
        u32x4_t fr[H * DSTEPS];
        constexpr unsigned RECB = (unsigned)((DSTEPS - (TAIL16 ? 1 : 0)) * 1024 + (TAIL16 ? 512 : 0));   // bytes per fragment record
        const char* krow[H];
#pragma unroll
        for (int h = 0; h < H; ++h) krow[h] = reinterpret_cast<const char*>(a.Kf) + ((long)b * H + h) * nt * (long)RECB;
        auto load_batch = [&](int tj, int kt_first_) {
            const int ktl = min(kt_first_ + tj, nt - 1);
#pragma unroll
            for (int h = 0; h < H; ++h)
#pragma unroll
                for (int st = 0; st < DSTEPS; ++st) fr[h * DSTEPS + st] = frag_load_row<DSTEPS, TAIL16>(krow[h], (unsigned)ktl * RECB, st, lane);
        };
        for (int km = wave % WPQ; qt_valid && km * KT < seg; km += WPQ) {
            const int kt_first = kt0 + km * KT;
            // ---- raw scores of all heads acc[j][h] = K_tile(h).Q_tile(h)^T.  The H*DSTEPS operand fragments of a tile are requested together (one
            // L2 round trip per batch instead of one per head), and the first batch of the NEXT macro step is requested right after the last MFMA of
            // this one, so it lands during the VALU phase.
            f32x4_t acc[KT][H];
            if (km == wave % WPQ) load_batch(0, kt_first);      // first macro step of the segment: nothing prefetched yet
#pragma unroll
            for (int tj = 0; tj < KT; ++tj) {
#pragma unroll
                for (int g0 = 0; g0 < H; g0 += QG) {
                    f32x4_t c[QG];
#pragma unroll
                    for (int jj = 0; jj < QG; ++jj) c[jj] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int st = 0; st < DSTEPS; ++st)
#pragma unroll
                        for (int jj = 0; jj < QG; ++jj) c[jj] = frag_mfma(fr[(g0 + jj) * DSTEPS + st], qreg[(g0 + jj) * DSTEPS + st], c[jj]);
#pragma unroll
                    for (int jj = 0; jj < QG; ++jj) acc[tj][g0 + jj] = c[jj];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (tj + 1 < KT) load_batch(tj + 1, kt_first);
                else if ((km + WPQ) * KT < seg) load_batch(0, kt0 + (km + WPQ) * KT);
                __builtin_amdgcn_sched_barrier(0);
            }
            // this lane's 4 consecutive keys of tile j: KB(j) + r ; tiles past the segment end belong to another workgroup
#define KB(j) ((kt_first + (j)) * 16 + 4 * (lane >> 4))
#define TV(j) (kt_first + (j) < kt0 + seg)
            // wave-uniform: tile j needs per-key masking (not this workgroup's tile, or the ragged last tile)
#define TMASK(j) (!TV(j) || (kt_first + (j) == nt - 1 && (N & 15) != 0))
#define KVAL(j, r) (TV(j) && KB(j) + (r) < N)
            f32x4_t sp[KT][4][H / 4];
            float tmax[H];
#pragma unroll
            for (int g = 0; g < H; ++g) tmax[g] = -INFINITY;
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                mix_keys_f32<H>(acc[j], Al4, vbl2, sp[j]);
                if (TMASK(j)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool kv = KVAL(j, r);
#pragma unroll
                        for (int g = 0; g < H; ++g) sp[j][r][g >> 2][g & 3] = kv ? sp[j][r][g >> 2][g & 3] : -INFINITY;
                    }
                }
#pragma unroll
                for (int g = 0; g < H; ++g) {
                    tmax[g] = fmaxf(fmaxf(tmax[g], sp[j][0][g >> 2][g & 3]), sp[j][1][g >> 2][g & 3]);
                    tmax[g] = fmaxf(fmaxf(tmax[g], sp[j][2][g >> 2][g & 3]), sp[j][3][g >> 2][g & 3]);
                }
            }
#pragma unroll
            for (int g = 0; g < H; ++g) {
                // branch-free: the subtrahend is clamped, so a row that has seen no valid key yet (max = -inf) gives
                // exp2(-inf - (-1e30)) = 0 everywhere instead of NaN
                const float mn = fmaxf(rm[g], tmax[g]), ms = fmaxf(mn, -1e30f);
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < KT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum += EXP2(sp[j][r][g >> 2][g & 3] - ms);
                rl[g] = rl[g] * EXP2(rm[g] - ms) + sum;
                rm[g] = mn;
            }
#undef TMASK
#undef KVAL
#undef KB
#undef TV
        }

        // ---- segment end: combine the row statistics of the 4 lane groups (same q, different keys) and of the waves of a q-tile
#pragma unroll
        for (int g = 0; g < H; ++g) {
            float m = rm[g], l = rl[g];
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float om = __shfl_xor(m, o, 64), ol = __shfl_xor(l, o, 64);
                const float mn = fmaxf(m, om);
                l = (mn > -INFINITY) ? l * EXP2(m - mn) + ol * EXP2(om - mn) : 0.f;
                m = mn;
            }
            if (lane < 16) { sred[((wave * H + g) * 16 + lane) * 2] = m; sred[((wave * H + g) * 16 + lane) * 2 + 1] = l; }
        }
        __syncthreads();
        const int first_j = (int)(((long)bqp * klen) / a.steps_per_wg);
        const int slot = chunk * (FUSED_MAXSLOT / nch) + (wg_j - first_j);
        for (int i2 = threadIdx.x; i2 < QP * H * 16; i2 += 256) {
            const int u = i2 / (H * 16), i = i2 % (H * 16);
            if (qp * QP + u >= nt) continue;
            const long bq_u = (long)b * nt + qp * QP + u;
            float* dst = a.ws_stats + (((bq_u * FUSED_MAXSLOT + slot) * H * 16) + i) * 2;
            float mn = -INFINITY;
            for (int w = u * WPQ; w < (u + 1) * WPQ; ++w) mn = fmaxf(mn, sred[((w * H * 16) + i) * 2]);
            float l = 0.f;
            if (mn > -INFINITY)
                for (int w = u * WPQ; w < (u + 1) * WPQ; ++w) l += sred[((w * H * 16) + i) * 2 + 1] * EXP2(sred[((w * H * 16) + i) * 2] - mn);
            dst[0] = mn; dst[1] = l;
        }
        s += seg;
    }
}

// Merge the per-slot partial statistics of each (b, q-tile) -> M = max, IL = 1 / sum [B,H,N] and the [B][Np][H] row constants of the flash kernels,
// bl[g] * log2(e) - max - log2(sum) (the addend that turns Wl S into log2 P), zero for the rows N .. Np-1.
__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ ws, float* __restrict__ out0, float* __restrict__ out1,
                                                         int B, int H, int N, int nt, int steps_per_wg,
                                                         const float* __restrict__ bl, float* __restrict__ rows, int Np) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;          // over B*(Np/16)*H*16
    const int ntr = Np / 16;
    if (i >= (long)B * ntr * H * 16) return;
    const int ql = (int)(i & 15); const int g = (int)((i >> 4) % H);
    const int b = (int)(i / (16L * H * ntr)), qt = (int)((i / (16L * H)) % ntr), q = qt * 16 + ql;
    const int bq = b * nt + qt;
    if (q >= N) { rows[((long)b * Np + q) * H + g] = 0.f; return; }
    const float* base = ws + (((long)bq * FUSED_MAXSLOT) * H * 16 + (long)g * 16 + ql) * 2;
    const long stride = (long)H * 16 * 2;
    const long o = ((long)b * H + g) * N + q;
    const int nch = fused_nch(nt), spc = FUSED_MAXSLOT / nch;
    // the slots of chunk c that received a partial: workgroups first_j..last_j of the chunk touch this q-tile
    float mn = -INFINITY, l = 0.f;
    for (int pass = 0; pass < 2; ++pass)
        for (int c = 0; c < nch; ++c) {
            const int klen = fused_kbeg(c + 1, nt, nch) - fused_kbeg(c, nt, nch);
            if (klen <= 0) continue;
            const long bqp = (long)b * fused_npair(nt) + qt / FUSED_QP;   // the workgroups walk (q-group, key tile) steps
            const int first_j = (int)((bqp * klen) / steps_per_wg), last_j = (int)(((bqp + 1) * klen - 1) / steps_per_wg);
            for (int s = 0; s <= last_j - first_j; ++s) {
                const float* e = base + (c * spc + s) * stride;
                if (pass == 0) mn = fmaxf(mn, e[0]);
                else l += e[1] * EXP2(e[0] - mn);
            }
        }
    const float il = 1.f / l;
    out0[o] = mn; out1[o] = il;
    rows[((long)b * Np + q) * H + g] = bl[g] * SPE_LOG2E - mn + __builtin_amdgcn_logf(il);
}

// Pack rows of x[b][n][h][d] (strides sb, sn, sh; unit d stride) into bf16 fragment records (see frag_load):
// per (b, h, tile): FULL steps of [lane][8] = scale * x[tile*16 + (lane&15)][st*32 + (lane>>4)*8 + i], then (tail) one
// step of [lane][4] = scale * x[tile*16 + (lane&15)][FULL*32 + (lane>>4)*4 + i]; 0 outside N x dh.  One thread per 8-B unit.
__global__ __launch_bounds__(256) void attn_pack_kernel(const float* __restrict__ x, long sb, long sn, long sh, int B, int N, int H,
                                                        int dh, int nt, float scale, uint2* __restrict__ out, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
        attn_pack_unit(x, sb, sn, sh, N, H, dh, nt, scale, i, out);
}

extern "C" int spe_attn_pack(const float* x, long sb, long sn, long sh, int B, int N, int H, int dh, float scale,
                             void* out, hipStream_t st) {
    const int nt = (N + 15) / 16;
    const long total = attn_pack_units(B, N, H, dh);
    if (total <= 0) return 0;
    long nb = (total + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(attn_pack_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, sb, sn, sh, B, N, H, dh, nt, scale,
                       reinterpret_cast<uint2*>(out), total);
    SPE_CHECK_LAUNCH();
    return 0;
}

// C-ABI: see include/spe_hip.h.  Merge of the statistics pass -> (M, IL, the flash kernels' row constants [B][Np][H]; Np a multiple of 16, >= N).
extern "C" int spe_attn_merge_rows(const float* ws, float* M, float* IL, const float* bl, float* rows, int Np, int B, int H, int N,
                                   int steps_per_wg, hipStream_t st) {
    const int nt = (N + 15) / 16;
    if ((long)B * nt * H <= 0) return 0;
    if (!rows || !bl || Np < nt * 16 || (Np & 15)) return -2;
    const long n = (long)B * (Np / 16) * H * 16;
    hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws, M, IL, B, H, N, nt, steps_per_wg, bl, rows, Np);
    SPE_CHECK_LAUNCH();
    return 0;
}

// Sum of the per-workgroup weight-gradient partials ws_w[nwg][2*(H*H+H)] (row layout [dWl | dbl | dWw | dbw]) written
// straight into the four parameter gradients (their all-reduce bucket views): one wave per column, fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws_w, int nwg, int H, float* __restrict__ dWl,
                                                           float* __restrict__ dbl, float* __restrict__ dWw, float* __restrict__ dbw) {
    const int hh = H * H, nw = 2 * (hh + H);
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (col >= nw) return;
    float s = 0.f;
    for (int r = lane; r < nwg; r += 64) s += ws_w[(long)r * nw + col];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        // STORES (never accumulates); a NULL destination is skipped
        if (col < hh) { if (dWl) dWl[col] = s; }
        else if (col < hh + H) { if (dbl) dbl[col - hh] = s; }
        else if (col < 2 * hh + H) { if (dWw) dWw[col - hh - H] = s; }
        else if (dbw) dbw[col - 2 * hh - H] = s;
    }
}

// C-ABI: see include/spe_hip.h (spe_talking_wgrad_reduce).
extern "C" int spe_talking_wgrad_reduce(const float* ws_w, int nwg, int H, float* dWl, float* dbl, float* dWw, float* dbw,
                                        hipStream_t st) {
    if (nwg <= 0 || H <= 0) return 0;
    const int nw = 2 * (H * H + H);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, st, ws_w, nwg, H, dWl, dbl, dWw, dbw);
    SPE_CHECK_LAUNCH();
    return 0;
}

// steps per workgroup of a chunk: even split over the chunk's workgroups, but the chunk's part of a q-tile may
// spread over at most FUSED_MAXSLOT / nch workgroups (its slots in the statistics workspace)
static void make_plan(int B, int nt, int nwg, int* spw_out, int* nwg_out) {
    const int nch = fused_nch(nt), spc = FUSED_MAXSLOT / nch;
    int nwg8 = nwg & ~7; if (nwg8 < 8) nwg8 = 8;
    const int wpc = nwg8 / nch;
    const int len_max = (nt + nch - 1) / nch;
    long spw = ((long)B * fused_npair(nt) * len_max + wpc - 1) / wpc;
    const long min_spw = (len_max + (spc - 1) - 1) / (spc - 1);       // ceil(len / (spc-1)): <= spc slots
    if (spw < min_spw) spw = min_spw;
    *spw_out = (int)spw; *nwg_out = nwg8;
}

// C-ABI: see include/spe_hip.h (spe_talking_stats).  Returns -2 for unsupported (H, head dim).
extern "C" int spe_talking_stats(const void* Qf, const void* Kf, const float* Wl, const float* bl, float* ws_stats,
                                 int B, int H, int N, int dh, int nwg, hipStream_t st) {
    StatsArgs a;
    a.Qf = (const u32x4_t*)Qf; a.Kf = (const u32x4_t*)Kf; a.Wl = Wl; a.bl = bl; a.ws_stats = ws_stats;
    a.B = B; a.N = N; a.nt = (N + 15) / 16;
    if ((long)B * a.nt * a.nt <= 0) return 0;
    make_plan(B, a.nt, nwg, &a.steps_per_wg, &nwg);
    // head dim -> d-steps: full 32-wide steps, plus a 16-wide tail step when the remainder is 1..16
    const int rem = dh % 32, full = dh / 32 + (rem > 16 ? 1 : 0), tail = (rem > 0 && rem <= 16) ? 1 : 0;
    const int ds = full + tail;
    if (dh < 1 || dh > 64) return -2;
    // macro step = 1 key tile (4 tiles measured slower inside the step: occupancy)
#define SPE_STATS_GO(HH, DS, TL)                                                                                        \
    if (H == HH && ds == DS && tail == TL) {                                                                            \
        hipLaunchKernelGGL((talking_stats_kernel<HH, DS, (TL != 0), 1>), dim3(nwg), dim3(256), 0, st, a);              \
        SPE_CHECK_LAUNCH();                                                                                             \
        return 0;                                                                                                       \
    }
    SPE_STATS_GO(8, 2, 1) SPE_STATS_GO(8, 2, 0) SPE_STATS_GO(8, 1, 1) SPE_STATS_GO(8, 1, 0)
    SPE_STATS_GO(4, 2, 1) SPE_STATS_GO(4, 2, 0) SPE_STATS_GO(4, 1, 1) SPE_STATS_GO(4, 1, 0)
#undef SPE_STATS_GO
    return -2;
}

// steps_per_wg the launcher will use for (B, N, nwg): callers size the statistics workspace with it.
extern "C" int spe_talking_stats_plan(int B, int N, int nwg, int* steps_per_wg, int* nwg_used) {
    const int nt = (N + 15) / 16;
    if ((long)B * nt * nt <= 0) { *steps_per_wg = 0; *nwg_used = 0; return 0; }
    make_plan(B, nt, nwg, steps_per_wg, nwg_used);
    return 0;
}
