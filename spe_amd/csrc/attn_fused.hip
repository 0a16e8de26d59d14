// Fused talking-heads attention score kernels (K4 of SURVEY.md section 2.2; reference
// models/cait.py:377-389 and its autograd).  The N x N score tensors never exist in fp32 in HBM:
// every (q-tile, k-tile) step recomputes the raw scores of ALL heads with MFMA and does the two
// head mixes, the softmax and the softmax backward in registers.
//
// Tile = 16 keys x 16 queries per step, v_mfma_f32_16x16x32_bf16 in the swapped orientation
// S^T = K_tile . Q_tile^T (M = keys, N = queries, K = head dim in 32-wide steps).  In the 16x16 C
// layout a lane owns ONE query column (q = lane & 15) and 4 consecutive keys ((lane>>4)*4 + r), so the
// per-row softmax state (max, sum, dP.P) is lane-local - 2 registers per head, no cross-lane traffic in
// the key loop - and the same acc[h][r] index across heads is the same (q,key) element: the H x H head
// mixes are plain FMAs on registers with the weights in SGPRs.  The small tile keeps all H score
// accumulators of BOTH operand pairs (QK^T and dO.V^T) in 8*H registers, so every mode - including the
// backward ones with their H*H weight-gradient accumulators - fits 256 registers without spilling and
// two workgroups share a CU (a 32x32 tile needs 256 accumulator registers and spills; measured 4x slower).
//
// Operands come from "row-fragment" packed bf16 arrays produced by spe_attn_pack (one 16-B load per
// lane per MFMA operand, 1 KB contiguous per wave): X_f[b][h][tile16][dstep][lane][8].
//
// Modes (one template, same skeleton):
//   0  forward statistics : partial (max, sum) of softmax_k(S'_g) per (b, g, q)
//   1  forward write      : P'd[b][g][q][key] = bf16( dropout( Ww . softmax(S') + bw ) )
//   2  backward pass 1    : dP' = (dO.V^T) * keepscale ; dWw, dbw ; D_h[q] = sum_k dP_h P_h
//   3  backward pass 2    : dS' = P (dP - D) ; dWl, dbl ; dS[b][h][q][key] = bf16( Wl^T dS' )
// The bf16 tensors feed the PV / dV / dQ / dK contractions (spe_gemm_ex with a bf16 A operand).
//
// Work partition: the (b, q-tile, k-tile) steps are flattened q-major and split evenly over the
// workgroups (two per CU, 4 waves each); a workgroup's range covers 1-3 q-tiles ("segments"), its 4 waves
// take the k-tiles of a segment round-robin and share the q-tile's Q (and dO) fragments through LDS.
// Per-segment row statistics go to a workspace indexed by (q-tile, slot = workgroup - first workgroup of
// the q-tile) and are merged by spe_attn_merge.
#include "common.h"

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

struct FusedArgs {
    const u32x4_t* Qf; const u32x4_t* Kf; const u32x4_t* Vf; const u32x4_t* dOf;
    const float* Wl; const float* bl; const float* Ww; const float* bw;
    const float* M; const float* IL; const float* D;      // final row stats [B,H,N] (modes 1-3), D (mode 3)
    float* ws_stats;                                      // partial row stats [B*nt][MAXSLOT][H][16][2] (modes 0, 2)
    float* ws_w;                                          // weight-gradient partials [nwg][2*(H*H+H)] (modes 2, 3)
    unsigned short* outT;                                 // bf16 [B,H,nt*16,ldq] (modes 1, 3), row = q, column = key
    int B, N, nt;                                         // nt = ceil(N/16) tiles per axis
    long ldq;
    int steps_per_wg; long total_steps;
    float p_drop; uint64_t seed, offset;
};

// H*H mixing weights -> SGPRs.  `wv` holds W[lane] (one weight per lane, loaded once per kernel); a phase
// pulls the H*H values into scalar registers with v_readlane.  The empty asm launders the vector register so
// that the readlanes cannot be hoisted out of the tile loop: a matrix then occupies SGPRs for one register
// phase only (both matrices at once do not fit the scalar file).  hipcc only emits s_load for pointers it can
// prove read-only; after any laundering it falls back to per-lane VMEM loads into VGPRs, hence this form.
template <int H>
__device__ __forceinline__ void load_w(float wv, float (&w)[H][H]) {
    asm volatile("" : "+v"(wv));
#pragma unroll
    for (int g = 0; g < H; ++g)
#pragma unroll
        for (int h = 0; h < H; ++h)
            w[g][h] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wv), g * H + h));
}

#define FUSED_MAXSLOT 8

template <int H, int DSTEPS, int MODE, bool DROP, int KT>
__global__ __launch_bounds__(256, 2) void talking_fused_kernel(FusedArgs a) {
    constexpr int NFR = H * DSTEPS;                        // fragments (16 B per lane) per q-tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4_t* sQ = reinterpret_cast<u32x4_t*>(smem_raw);    // [NFR][64]
    u32x4_t* sdO = sQ + NFR * 64;                          // [NFR][64]   (modes 2, 3)
    float* sred = reinterpret_cast<float*>(sdO + ((MODE >= 2) ? NFR * 64 : 0));   // [4][H][16][2]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt = a.nt, N = a.N;
    const long s_begin = (long)blockIdx.x * a.steps_per_wg;
    long s_end = s_begin + a.steps_per_wg; if (s_end > a.total_steps) s_end = a.total_steps;

    // The mixing matrices are (re)loaded into SGPRs at the start of each register phase (see load_w): both
    // together (2*H*H+2H values) do not fit the scalar file next to the addressing state.
    float vbl[H], vbw[H];
#pragma unroll
    for (int g = 0; g < H; ++g) { vbl[g] = a.bl[g]; vbw[g] = a.bw[g]; }
    const float wlv = a.Wl[(threadIdx.x & 63) < H * H ? (threadIdx.x & 63) : 0];   // lane i holds Wl[i / H][i % H]
    const float wwv = a.Ww[(threadIdx.x & 63) < H * H ? (threadIdx.x & 63) : 0];
    // weight-gradient accumulators (whole workgroup range)
    float gW[(MODE >= 2) ? H : 1][(MODE >= 2) ? H : 1], gb[(MODE >= 2) ? H : 1];
    if (MODE >= 2) {
#pragma unroll
        for (int g = 0; g < H; ++g) { gb[g] = 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) gW[g][h] = 0.f; }
    }

    long s = s_begin;
    while (s < s_end) {
        const int bq = (int)(s / nt), kt0 = (int)(s % nt);
        int seg = nt - kt0; if (seg > s_end - s) seg = (int)(s_end - s);
        const int b = bq / nt, qt = bq % nt;
        const int q = qt * 16 + (lane & 15);
        const bool qv = q < N;
        // ---- stage this q-tile's Q (and dO) fragments in LDS
        __syncthreads();
        for (int i = threadIdx.x; i < NFR * 64; i += 256) {
            const int fr = i >> 6, ln = i & 63, h = fr / DSTEPS, st = fr % DSTEPS;
            const long src = ((((long)b * H + h) * nt + qt) * DSTEPS + st) * 64 + ln;
            sQ[i] = a.Qf[src];
            if (MODE >= 2) sdO[i] = a.dOf[src];
        }
        __syncthreads();
        // ---- per-lane row state
        float rm[H], rl[H], rD[H];
#pragma unroll
        for (int g = 0; g < H; ++g) {
            if (MODE == 0) { rm[g] = -INFINITY; rl[g] = 0.f; }
            else {
                const long si = ((long)b * H + g) * N + (qv ? q : 0);
                rm[g] = a.M[si]; rl[g] = a.IL[si];
            }
            if (MODE == 2) rD[g] = 0.f;
            if (MODE == 3) rD[g] = a.D[((long)b * H + g) * N + (qv ? q : 0)];
        }

        // A wave's unit of work is a macro step of KT consecutive 16-key tiles against the 16 queries of the
        // q-tile: the mixing weights (SGPR reload), the Q / dO fragments (LDS) and the loop overhead are paid once
        // per macro step.  Waves take macro steps round-robin.
        for (int km = wave; km * KT < seg; km += 4) {
            const int kt_first = kt0 + km * KT;
            // ---- raw scores of all heads acc[j][h] = K_tile(h).Q_tile(h)^T (and acc2[j][g] = V_tile(g).dO_tile(g)^T),
            // software-pipelined: the operand fragments of the next (head, tile) job are in flight while the MFMAs of
            // the current job issue (bounds the staging registers to 2*DSTEPS fragments).
            constexpr int NH = (MODE >= 2) ? 2 * H : H;
            constexpr int NJ = NH * KT;
            f32x4_t acc[KT][H];
            f32x4_t acc2[(MODE >= 2) ? KT : 1][(MODE >= 2) ? H : 1];
            u32x4_t cur[DSTEPS], nxt[DSTEPS], qf[DSTEPS];
            {
                const int ktl = min(kt_first, nt - 1);
#pragma unroll
                for (int st = 0; st < DSTEPS; ++st) cur[st] = a.Kf[((((long)b * H + 0) * nt + ktl) * DSTEPS + st) * 64 + lane];
            }
#pragma unroll
            for (int jb = 0; jb < NJ; ++jb) {
                const int hj = jb / KT, tj = jb % KT;                 // head job (0..NH-1), tile within the macro step
                if (jb + 1 < NJ) {
                    const int hn = ((jb + 1) / KT) % H, tn = (jb + 1) % KT;
                    const u32x4_t* srcp = ((jb + 1) / KT < H) ? a.Kf : a.Vf;
                    const int ktl = min(kt_first + tn, nt - 1);
#pragma unroll
                    for (int st = 0; st < DSTEPS; ++st) nxt[st] = srcp[((((long)b * H + hn) * nt + ktl) * DSTEPS + st) * 64 + lane];
                }
                const int hh = hj % H;
                if (tj == 0) {
                    const u32x4_t* lds = (hj < H) ? sQ : sdO;
#pragma unroll
                    for (int st = 0; st < DSTEPS; ++st) qf[st] = lds[(hh * DSTEPS + st) * 64 + lane];
                }
                f32x4_t c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < DSTEPS; ++st)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, cur[st]), __builtin_bit_cast(bf16x8_t, qf[st]), c, 0, 0, 0);
                if (hj < H) acc[tj][hh] = c; else acc2[(MODE >= 2) ? tj : 0][(MODE >= 2) ? hh : 0] = c;
#pragma unroll
                for (int st = 0; st < DSTEPS; ++st) cur[st] = nxt[st];
                __builtin_amdgcn_sched_barrier(0);
            }
            // this lane's 4 consecutive keys of tile j: kb(j) + r ; tiles past the segment end belong to another workgroup
#define KB(j) ((kt_first + (j)) * 16 + 4 * (lane >> 4))
#define TV(j) (kt_first + (j) < kt0 + seg)

            // Every mode runs as two register phases that each use ONE mixing matrix, so that the 64 weights
            // of a phase stay in SGPRs (both matrices together do not fit the scalar file).
            if (MODE == 0) {
                // phase A (Wl): S' in place + macro-step max ; then one rescale + 4*KT exps per head
                float wl[H][H];
                load_w<H>(wlv, wl);
                float tmax[H];
#pragma unroll
                for (int g = 0; g < H; ++g) tmax[g] = -INFINITY;
#pragma unroll
                for (int j = 0; j < KT; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool kv = TV(j) && KB(j) + r < N;
                        float sv[H];
#pragma unroll
                        for (int h = 0; h < H; ++h) sv[h] = acc[j][h][r];
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            float v = vbl[g];
#pragma unroll
                            for (int h = 0; h < H; ++h) v = fmaf(wl[g][h], sv[h], v);
                            v = kv ? v : -INFINITY;
                            acc[j][g][r] = v;
                            tmax[g] = fmaxf(tmax[g], v);
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < H; ++g) {
                    const float mn = fmaxf(rm[g], tmax[g]);
                    float sum = 0.f;
                    if (mn > -INFINITY) {
#pragma unroll
                        for (int j = 0; j < KT; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) sum += __expf(acc[j][g][r] - mn);
                        rl[g] = rl[g] * __expf(rm[g] - mn) + sum;
                        rm[g] = mn;
                    }
                }
            } else if (MODE == 1 || MODE == 2) {
                // phase A (Wl): acc <- P = exp(S' - m) / l
                {
                float wl[H][H];
                load_w<H>(wlv, wl);
#pragma unroll
                for (int j = 0; j < KT; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool kv = TV(j) && KB(j) + r < N;
                        float sv[H];
#pragma unroll
                        for (int h = 0; h < H; ++h) sv[h] = acc[j][h][r];
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            float v = vbl[g];
#pragma unroll
                            for (int h = 0; h < H; ++h) v = fmaf(wl[g][h], sv[h], v);
                            acc[j][g][r] = kv ? __expf(v - rm[g]) * rl[g] : 0.f;
                        }
                        if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                }
                __builtin_amdgcn_sched_barrier(0);
                // phase B (Ww)
                float ww[H][H];
                load_w<H>(wwv, ww);
                if (MODE == 1) {
#pragma unroll
                    for (int j = 0; j < KT; ++j) {
                        float o4[H][4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = KB(j) + r;
                            float pv[H];
#pragma unroll
                            for (int h = 0; h < H; ++h) pv[h] = acc[j][h][r];
#pragma unroll
                            for (int g = 0; g < H; ++g) {
                                float v = vbw[g];
#pragma unroll
                                for (int h = 0; h < H; ++h) v = fmaf(ww[g][h], pv[h], v);
                                if (DROP) v *= spe_drop_scale(a.seed, a.offset, (uint64_t)((((long)b * H + g) * N + q) * (long)N + key), a.p_drop);
                                o4[g][r] = v;
                            }
                        }
                        // lane owns 4 consecutive keys of row q: one 8-B store per head
                        if (TV(j)) {
#pragma unroll
                            for (int g = 0; g < H; ++g) {
                                bf16x4_t o;
                                o[0] = (__bf16)o4[g][0]; o[1] = (__bf16)o4[g][1]; o[2] = (__bf16)o4[g][2]; o[3] = (__bf16)o4[g][3];
                                *reinterpret_cast<bf16x4_t*>(a.outT + (((long)b * H + g) * (nt * 16) + q) * a.ldq + KB(j)) = o;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < KT; ++j) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = KB(j) + r;
                            float pv[H];
#pragma unroll
                            for (int h = 0; h < H; ++h) pv[h] = acc[j][h][r];
                            const bool ev = qv && TV(j) && key < N;
                            float dpp[H];
#pragma unroll
                            for (int g = 0; g < H; ++g) {
                                float v = ev ? acc2[j][g][r] : 0.f;
                                if (DROP) v *= spe_drop_scale(a.seed, a.offset, (uint64_t)((((long)b * H + g) * N + q) * (long)N + key), a.p_drop);
                                dpp[g] = v;
                                gb[g] += v;
#pragma unroll
                                for (int h = 0; h < H; ++h) gW[g][h] = fmaf(v, pv[h], gW[g][h]);
                            }
#pragma unroll
                            for (int h = 0; h < H; ++h) {
                                float v = 0.f;
#pragma unroll
                                for (int g = 0; g < H; ++g) v = fmaf(ww[g][h], dpp[g], v);
                                rD[h] = fmaf(v, pv[h], rD[h]);
                            }
                        }
                    }
                }
            } else {
                // MODE 3.  phase A (Ww): acc2 <- dP = Ww^T (dP'd * keepscale)
                {
                float ww[H][H];
                load_w<H>(wwv, ww);
#pragma unroll
                for (int j = 0; j < KT; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = KB(j) + r;
                        float dpp[H];
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            float v = acc2[j][g][r];
                            if (DROP) v *= spe_drop_scale(a.seed, a.offset, (uint64_t)((((long)b * H + g) * N + q) * (long)N + key), a.p_drop);
                            dpp[g] = v;
                        }
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            float v = 0.f;
#pragma unroll
                            for (int g = 0; g < H; ++g) v = fmaf(ww[g][h], dpp[g], v);
                            acc2[j][h][r] = v;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                }
                __builtin_amdgcn_sched_barrier(0);
                float wl[H][H];
                load_w<H>(wlv, wl);
                // phase B (Wl both ways): P from raw S, dS' = P (dP - D), dWl += dS' S^T, dS = Wl^T dS'
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    float o4[H][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = KB(j) + r;
                        const bool ev = qv && TV(j) && key < N;
                        float sv[H], ds1[H];
#pragma unroll
                        for (int h = 0; h < H; ++h) sv[h] = acc[j][h][r];
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            float v = vbl[g];
#pragma unroll
                            for (int h = 0; h < H; ++h) v = fmaf(wl[g][h], sv[h], v);
                            const float pg = __expf(v - rm[g]) * rl[g];
                            const float d = ev ? pg * (acc2[j][g][r] - rD[g]) : 0.f;
                            ds1[g] = d;
                            gb[g] += d;
#pragma unroll
                            for (int h = 0; h < H; ++h) gW[g][h] = fmaf(d, sv[h], gW[g][h]);
                        }
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            float v = 0.f;
#pragma unroll
                            for (int g = 0; g < H; ++g) v = fmaf(wl[g][h], ds1[g], v);
                            o4[h][r] = v;
                        }
                    }
                    if (TV(j)) {
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            bf16x4_t o;
                            o[0] = (__bf16)o4[h][0]; o[1] = (__bf16)o4[h][1]; o[2] = (__bf16)o4[h][2]; o[3] = (__bf16)o4[h][3];
                            *reinterpret_cast<bf16x4_t*>(a.outT + (((long)b * H + h) * (nt * 16) + q) * a.ldq + KB(j)) = o;
                        }
                    }
                }
            }
#undef KB
#undef TV
        }

        // ---- segment end: combine the row statistics of the 4 lane groups (same q, different keys) and 4 waves
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int g = 0; g < H; ++g) {
                if (MODE == 0) {
                    float m = rm[g], l = rl[g];
#pragma unroll
                    for (int o = 16; o <= 32; o <<= 1) {
                        const float om = __shfl_xor(m, o, 64), ol = __shfl_xor(l, o, 64);
                        const float mn = fmaxf(m, om);
                        l = (mn > -INFINITY) ? l * __expf(m - mn) + ol * __expf(om - mn) : 0.f;
                        m = mn;
                    }
                    if (lane < 16) { sred[((wave * H + g) * 16 + lane) * 2] = m; sred[((wave * H + g) * 16 + lane) * 2 + 1] = l; }
                } else {
                    float d = rD[g];
                    d += __shfl_xor(d, 16, 64);
                    d += __shfl_xor(d, 32, 64);
                    if (lane < 16) sred[((wave * H + g) * 16 + lane) * 2] = d;
                }
            }
            __syncthreads();
            const int first_wg = (int)(((long)bq * nt) / a.steps_per_wg);
            const int slot = blockIdx.x - first_wg;
            for (int i = threadIdx.x; i < H * 16; i += 256) {
                float* dst = a.ws_stats + ((((long)bq * FUSED_MAXSLOT + slot) * H * 16) + i) * 2;
                if (MODE == 0) {
                    float mn = -INFINITY;
                    for (int w = 0; w < 4; ++w) mn = fmaxf(mn, sred[((w * H * 16) + i) * 2]);
                    float l = 0.f;
                    if (mn > -INFINITY)
                        for (int w = 0; w < 4; ++w) l += sred[((w * H * 16) + i) * 2 + 1] * __expf(sred[((w * H * 16) + i) * 2] - mn);
                    dst[0] = mn; dst[1] = l;
                } else {
                    float d = 0.f;
                    for (int w = 0; w < 4; ++w) d += sred[((w * H * 16) + i) * 2];
                    dst[0] = d; dst[1] = 0.f;
                }
            }
        }
        s += seg;
    }

    // ---- weight-gradient partials of this workgroup -> ws_w[blockIdx][2*(H*H+H)]
    if (MODE >= 2) {
        constexpr int NW = 2 * (H * H + H);
        __syncthreads();
        float* part = sred;                                 // [4][H*H+H]
#pragma unroll
        for (int g = 0; g < H; ++g) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float v = spe_wave_sum(gW[g][h]);
                if (lane == 0) part[wave * (H * H + H) + g * H + h] = v;
            }
            const float v = spe_wave_sum(gb[g]);
            if (lane == 0) part[wave * (H * H + H) + H * H + g] = v;
        }
        __syncthreads();
        // layout of a ws_w row: [dWl | dbl | dWw | dbw]; mode 2 fills the second half, mode 3 the first
        const int off = (MODE == 2) ? (H * H + H) : 0;
        for (int i = threadIdx.x; i < H * H + H; i += 256)
            a.ws_w[(long)blockIdx.x * NW + off + i] = part[i] + part[(H * H + H) + i] + part[2 * (H * H + H) + i] + part[3 * (H * H + H) + i];
    }
}

// Merge the per-slot partial statistics of each (b, q-tile): mode 0 -> M = max, IL = 1/sum; mode 2 -> D = sum.
__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ ws, float* __restrict__ out0, float* __restrict__ out1,
                                                         int B, int H, int N, int nt, int steps_per_wg, int mode) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;          // over B*nt*H*16
    if (i >= (long)B * nt * H * 16) return;
    const int ql = (int)(i & 15); const int g = (int)((i >> 4) % H); const int bq = (int)(i / (16L * H));
    const int b = bq / nt, qt = bq % nt, q = qt * 16 + ql;
    if (q >= N) return;
    const int first_wg = (int)(((long)bq * nt) / steps_per_wg), last_wg = (int)((((long)bq + 1) * nt - 1) / steps_per_wg);
    const float* base = ws + (((long)bq * FUSED_MAXSLOT) * H * 16 + (long)g * 16 + ql) * 2;
    const long stride = (long)H * 16 * 2;
    const long o = ((long)b * H + g) * N + q;
    if (mode == 0) {
        float mn = -INFINITY;
        for (int s = 0; s <= last_wg - first_wg; ++s) mn = fmaxf(mn, base[s * stride]);
        float l = 0.f;
        for (int s = 0; s <= last_wg - first_wg; ++s) l += base[s * stride + 1] * __expf(base[s * stride] - mn);
        out0[o] = mn; out1[o] = 1.f / l;
    } else {
        float d = 0.f;
        for (int s = 0; s <= last_wg - first_wg; ++s) d += base[s * stride];
        out0[o] = d;
    }
}

// Pack rows of x[b][n][h][d] (strides sb, sn, sh; unit d stride) into bf16 row fragments
// out[b][h][tile][dstep][lane][8] = scale * x[b, tile*16 + (lane&15), h, dstep*32 + (lane>>4)*8 + i]  (0 outside)
__global__ __launch_bounds__(256) void attn_pack_kernel(const float* __restrict__ x, long sb, long sn, long sh, int B, int N, int H,
                                                        int dh, int nt, int dsteps, float scale, u32x4_t* __restrict__ out) {
    const long total = (long)B * H * nt * dsteps * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ln = (int)(i & 63); long t = i >> 6;
        const int st = (int)(t % dsteps); t /= dsteps;
        const int tile = (int)(t % nt); t /= nt;
        const int h = (int)(t % H); const int b = (int)(t / H);
        const int row = tile * 16 + (ln & 15), d0 = st * 32 + (ln >> 4) * 8;
        float v[8];
        const float* src = x + b * sb + (long)min(row, N - 1) * sn + h * sh;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = src[min(d0 + j, dh - 1)]; v[j] = (row < N && d0 + j < dh) ? f * scale : 0.f; }
        bf16x8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
        out[i] = __builtin_bit_cast(u32x4_t, o);
    }
}

extern "C" int spe_attn_pack(const float* x, long sb, long sn, long sh, int B, int N, int H, int dh, float scale,
                             void* out, hipStream_t st) {
    const int nt = (N + 15) / 16, dsteps = (dh + 31) / 32;
    const long total = (long)B * H * nt * dsteps * 64;
    if (total <= 0) return 0;
    long nb = (total + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(attn_pack_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, sb, sn, sh, B, N, H, dh, nt, dsteps, scale,
                       reinterpret_cast<u32x4_t*>(out));
    SPE_CHECK_LAUNCH();
    return 0;
}

extern "C" int spe_attn_merge(const float* ws, float* out0, float* out1, int B, int H, int N, int steps_per_wg, int mode,
                              hipStream_t st) {
    const int nt = (N + 15) / 16;
    const long n = (long)B * nt * H * 16;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws, out0, out1, B, H, N, nt,
                       steps_per_wg, mode);
    SPE_CHECK_LAUNCH();
    return 0;
}

// steps per workgroup: even split, but a q-tile (nt steps) may spread over at most FUSED_MAXSLOT workgroups
static int plan_spw(long total, int nt, int nwg) {
    if (nwg < 1) nwg = 1;
    long spw = (total + nwg - 1) / nwg;
    const long min_spw = (nt + FUSED_MAXSLOT - 3) / (FUSED_MAXSLOT - 2);   // ceil(nt / (MAXSLOT-2)): <= MAXSLOT-1 slots
    if (spw < min_spw) spw = min_spw;
    return (int)spw;
}

template <int H, int DSTEPS, int MODE, bool DROP, int KT>
static int launch_fused(const FusedArgs& a, int nwg, hipStream_t st) {
    constexpr int NFR = H * DSTEPS;
    constexpr int smem = NFR * 64 * 16 * ((MODE >= 2) ? 2 : 1) + 4 * (H * 16 * 2 > (H * H + H) ? H * 16 * 2 : (H * H + H)) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&talking_fused_kernel<H, DSTEPS, MODE, DROP, KT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((talking_fused_kernel<H, DSTEPS, MODE, DROP, KT>), dim3(nwg), dim3(256), smem, st, a);
    SPE_CHECK_LAUNCH();
    return 0;
}

// macro-step sizes (measured at cfg2): 4 tiles for the statistics pass (0.26 -> 0.235 ms), 1 tile for the write
// pass (4 tiles: 0.43 -> 0.48 ms, occupancy loss) and for the backward passes (H*H weight-gradient accumulators +
// the second accumulator set)
#ifndef SPE_FUSED_KTF
#define SPE_FUSED_KTF 4
#endif
#ifndef SPE_FUSED_KTB
#define SPE_FUSED_KTB 1
#endif
template <int H, int DSTEPS>
static int dispatch_mode(const FusedArgs& a, int mode, bool drop, int nwg, hipStream_t st) {
    constexpr int KF = SPE_FUSED_KTF, KB_ = SPE_FUSED_KTB;
    switch (mode) {
        case 0: return launch_fused<H, DSTEPS, 0, false, KF>(a, nwg, st);
        case 1: return drop ? launch_fused<H, DSTEPS, 1, true, KB_>(a, nwg, st) : launch_fused<H, DSTEPS, 1, false, KB_>(a, nwg, st);
        case 2: return drop ? launch_fused<H, DSTEPS, 2, true, KB_>(a, nwg, st) : launch_fused<H, DSTEPS, 2, false, KB_>(a, nwg, st);
        case 3: return drop ? launch_fused<H, DSTEPS, 3, true, KB_>(a, nwg, st) : launch_fused<H, DSTEPS, 3, false, KB_>(a, nwg, st);
    }
    return -2;
}

// C-ABI: see include/spe_hip.h (spe_talking_fused).  Returns -2 for unsupported (H, head dim).
extern "C" int spe_talking_fused(int mode, const void* Qf, const void* Kf, const void* Vf, const void* dOf,
                                 const float* Wl, const float* bl, const float* Ww, const float* bw,
                                 const float* M, const float* IL, const float* D, float* ws_stats, float* ws_w, void* outT,
                                 int B, int H, int N, int dh, long ldq, int nwg, float p_drop, uint64_t seed, uint64_t offset,
                                 hipStream_t st) {
    FusedArgs a;
    a.Qf = (const u32x4_t*)Qf; a.Kf = (const u32x4_t*)Kf; a.Vf = (const u32x4_t*)Vf; a.dOf = (const u32x4_t*)dOf;
    a.Wl = Wl; a.bl = bl; a.Ww = Ww; a.bw = bw; a.M = M; a.IL = IL; a.D = D;
    a.ws_stats = ws_stats; a.ws_w = ws_w; a.outT = (unsigned short*)outT;
    a.B = B; a.N = N; a.nt = (N + 15) / 16; a.ldq = ldq;
    a.total_steps = (long)B * a.nt * a.nt;
    if (a.total_steps <= 0) return 0;
    a.steps_per_wg = plan_spw(a.total_steps, a.nt, nwg);
    nwg = (int)((a.total_steps + a.steps_per_wg - 1) / a.steps_per_wg);
    a.p_drop = p_drop; a.seed = seed; a.offset = offset;
    const bool drop = p_drop > 0.f;
    const int ds = (dh + 31) / 32;
    if (H == 8 && ds == 2) return dispatch_mode<8, 2>(a, mode, drop, nwg, st);
    if (H == 4 && ds == 2) return dispatch_mode<4, 2>(a, mode, drop, nwg, st);
    if (H == 4 && ds == 1) return dispatch_mode<4, 1>(a, mode, drop, nwg, st);
    if (H == 8 && ds == 1) return dispatch_mode<8, 1>(a, mode, drop, nwg, st);
    return -2;
}

// steps_per_wg the launcher will use for (B, N, nwg): callers size the workspaces with it.
extern "C" int spe_talking_fused_plan(int B, int N, int nwg, int* steps_per_wg, int* nwg_used) {
    const int nt = (N + 15) / 16;
    const long total = (long)B * nt * nt;
    if (total <= 0) { *steps_per_wg = 0; *nwg_used = 0; return 0; }
    const int spw = plan_spw(total, nt, nwg);
    *steps_per_wg = spw;
    *nwg_used = (int)((total + spw - 1) / spw);
    return 0;
}
