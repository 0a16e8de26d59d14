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
// Operands come from "row-fragment" packed 16-bit arrays produced by spe_attn_pack_multi (one 16-B load per
// lane per MFMA operand, 1 KB contiguous per wave): X_f[b][h][tile16][dstep][lane][8].
//
// Element formats.  The FORWARD quantities are O(1) and go through fp16: the Q (scaled) and K fragments, the probabilities that
// enter the second head mix, and the stored P'd (scaled by SPE_PD_SCALE = 2^8 so that probabilities of 1e-4 .. 1e-7 stay normal
// numbers; the PV contraction multiplies by 2^-8) - 11 significant bits instead of bf16's 8 at the same bytes and MFMA rate,
// which takes the attention's share of the forward error from ~2e-4 to ~3e-5 of the outputs (tools/error_budget.py).  The
// backward passes recompute S from the same fp16 fragments (so P matches the forward statistics exactly) and keep bf16 for
// everything that carries a gradient (V.dO^T operands, dP, dS): gradient magnitudes are not bounded.
//
// Modes (one template, same skeleton):
//   0  forward statistics : partial (max, sum) of softmax_k(S'_g) per (b, g, q)
//   1  forward write      : P'd = bf16( dropout( Ww . softmax(S') + bw ) )   (16x16 blocks, see attn_contract.hip)
//   2  backward pass 1    : dP' = (dO.V^T) * keepscale ; dWw, dbw ; D_h[q] = sum_k dP_h P_h
//   3  backward pass 2    : dS' = P (dP - D) ; dWl, dbl ; dS = bf16( Wl^T dS' )   (same block layout)
// The bf16 tensors are written as whole 512-B blocks (one coalesced wave store per head and tile; row-major
// 8-B pieces cost 0.2 ms per pass at cfg2) and feed the PV / dV / dQ / dK contractions of attn_contract.hip.
//
// Work partition: the (b, q-tile, k-tile) steps are flattened q-major and split evenly over the
// workgroups (two per CU, 4 waves each); a workgroup's range covers 1-3 q-tiles ("segments"), its 4 waves
// take the k-tiles of a segment round-robin and share the q-tile's Q (and dO) fragments through LDS.
// Per-segment row statistics go to a workspace indexed by (q-tile, slot = workgroup - first workgroup of
// the q-tile) and are merged by spe_attn_merge.
#include "common.h"
#include "attn_pack.h"
#include "attn_flash_common.h"

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

struct FusedArgs {
    const u32x4_t* Qf; const u32x4_t* Kf; const u32x4_t* Vf; const u32x4_t* dOf;
    const float* Wl; const float* bl; const float* Ww; const float* bw;
    const float* M; const float* IL; const float* D;      // final row stats [B,H,N] (modes 1-3), D (mode 3)
    float* ws_stats;                                      // partial row stats [B*nt][MAXSLOT][H][16][2] (modes 0, 2)
    float* ws_w;                                          // weight-gradient partials [nwg][2*(H*H+H)] (modes 2, 3)
    unsigned short* outT;                                 // bf16 16x16 blocks [B,H,nt,nt][64 lanes][4] (modes 1, 3): lane = (q, 4 keys)
    int B, N, nt;                                         // nt = ceil(N/16) tiles per axis
    int steps_per_wg; long total_steps;
    float p_drop; uint64_t seed, offset;
    const unsigned* keepbits;                             // modes 2, 3 with dropout (optional): the keep flags the flash forward stored, [B][nt][nt][64]
};

// H*H mixing weights -> SGPRs, once per register phase: H*H/16 s_load_dwordx16 through the scalar cache (no
// VALU slots; v_readlane from a staging VGPR cost ~30 % of a pass's vector instructions).  The pointer is cast to
// the constant address space - hipcc only selects SMEM for global loads it can prove unclobbered, constant loads
// always qualify - and laundered through an SGPR so that the invariant loads are not hoisted out of the tile
// loop: a matrix then occupies the scalar file for one phase only (both at once do not fit).
typedef const __attribute__((address_space(4))) float* spe_cfp;
template <int H>
__device__ __forceinline__ void load_w(const float* p, float (&w)[H][H]) {
    spe_cfp wp = (spe_cfp)p;
    asm volatile("" : "+s"(wp));
#pragma unroll
    for (int g = 0; g < H; ++g)
#pragma unroll
        for (int h = 0; h < H; ++h) w[g][h] = wp[g * H + h];
}

#define FUSED_MAXSLOT 8
// q-tiles per workgroup: with 2, waves (0,1) work on q-tile 2p and waves (2,3) on q-tile 2p+1 and BOTH pairs walk the same key
// tiles in step, so every K / V fragment is requested twice within a few hundred cycles by one CU and the second request is
// served by (or merged into the pending miss of) the vector L1 - the L1 -> L2 request stream, which bounds the backward passes
// (DESIGN.md 4.1), halves.  Measured with duplicated tile walks before the change: ~10 % per tile in passes 0, 2 and 3.
// Measured (cfg2, isolated): statistics pass 0.135 -> 0.126 ms, write pass 0.157 -> 0.152; the backward passes, which would
// have to keep the Q and dO fragments of both q-tiles in LDS, lose (0.305 -> 0.341, 0.354 -> 0.362 ms) and keep one q-tile.
#ifndef SPE_FUSED_QP
#define SPE_FUSED_QP 2
#endif
__host__ __device__ __forceinline__ int fused_qp(int mode) { return (mode <= 1) ? SPE_FUSED_QP : 1; }
__host__ __device__ __forceinline__ int fused_npair(int nt, int mode) { return (nt + fused_qp(mode) - 1) / fused_qp(mode); }

// Key chunks bound to XCDs.  A workgroup with blockIdx b runs on XCD b % 8 (8 private 4 MB L2s).  The K and V
// fragments of an image are 4-8 MB; when every workgroup sweeps all keys each L2 thrashes on them (rocprof: 1.9 GB of
// L2 fetches per backward launch against 34 MB of operands).  So the key tiles are cut into NCH chunks and XCD x only
// works on chunk x % NCH: its L2 holds 1/NCH of the K/V fragments.  Within a chunk the (b, q-tile, k-tile) steps are
// flattened q-major and split evenly over the chunk's workgroups as before.
#ifndef SPE_FUSED_NCH
#define SPE_FUSED_NCH 4
#endif
__host__ __device__ __forceinline__ int fused_nch(int nt) { return (nt >= 16 * SPE_FUSED_NCH) ? SPE_FUSED_NCH : 1; }
__host__ __device__ __forceinline__ int fused_kbeg(int c, int nt, int nch) { return (int)((long)c * nt / nch); }

typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define PLO(v) __builtin_shufflevector(v, v, 0, 1)
#define PHI(v) __builtin_shufflevector(v, v, 2, 3)
#define PCAT(lo, hi) __builtin_shufflevector(lo, hi, 0, 1, 2, 3)
#ifdef SPE_DBG_NOEXP
#define EXP2(x) ((x) * 0.5f)
#else
#define EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
#define SPE_PD_SCALE 256.0f
#define SPE_LOG2E 1.4426950408889634f
#define SPE_LN2 0.6931471805599453f
__device__ __forceinline__ f32x2_t fma2(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2_t splat2(float x) { return (f32x2_t){x, x}; }

// out[g] = c[g] + sum_h w[g][h] * s[h] for a lane's 4 keys, as two key pairs (lo = keys 0,1 ; hi = keys 2,3):
// H*H v_pk_fma_f32 with the weight broadcast from one SGPR.
template <int H>
__device__ __forceinline__ void mix_rows(const f32x4_t (&s)[H], const float (&w)[H][H], const float (&c)[H], f32x2_t (&lo)[H], f32x2_t (&hi)[H]) {
#pragma unroll
    for (int g = 0; g < H; ++g) { lo[g] = splat2(c[g]); hi[g] = lo[g]; }
#ifdef SPE_DBG_NOMIX
#pragma unroll
    for (int g = 0; g < H; ++g) { lo[g] += PLO(s[g]); hi[g] += PHI(s[g]); }
    return;
#endif
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const f32x2_t sl = PLO(s[h]), sh = PHI(s[h]);
#pragma unroll
        for (int g = 0; g < H; ++g) { lo[g] = fma2(sl, splat2(w[g][h]), lo[g]); hi[g] = fma2(sh, splat2(w[g][h]), hi[g]); }
    }
}

// Head mixes on the MATRIX pipe.  The three mixes whose operands the bf16 contract rounds anyway - P' = Ww P + bw (its result is
// stored as bf16), dP = Ww^T dP' (dP' is an MFMA product of bf16 operands) and dS = Wl^T dS' (stored as bf16) - do not need the
// fp32 vector FMAs that bound these kernels (H*H v_pk_fma_f32 per lane and tile each; the vector pipe is ~80 % busy, the
// matrix pipe ~10 %).  One v_mfma_f32_16x16x16_bf16 against a BLOCK-DIAGONAL weight operand mixes 4 input heads into 4 output
// heads for all 64 lanes at once without moving data between lanes:
//   B[k = 4j + i][n = q] := x[head 4hh + i] of (query q, key 4j + i0)        - lane (q, j)'s own 4 values (bf16)
//   A[m = 4j' + g'][k = 4j + i] := (j' == j) ? W[4gh + g'][4hh + i] : 0     - constant per (gh, hh): 2 registers
//   D[m = 4j + r][n = q] = sum_i W[4gh + r][4hh + i] x[4hh + i]              - lands in lane (q, j): its own key, heads 4gh + r
// i.e. (H/4)^2 MFMAs per key and 4 * (H/4)^2 per tile (16 at H = 8) instead of H*H/2 * 4 = 128 packed FMAs, plus H/2
// v_cvt_pk_bf16_f32 per key.  Measured at cfg2 (same box, isolated): write pass 0.218 -> 0.187 ms, backward pass 1 0.385 -> 0.370,
// backward pass 2 0.424 -> 0.377; errors against fp64 unchanged (out 2.4e-3, dqkv 3.8e-3 -> 4.0e-3).  The softmax-input mix
// S' = Wl S + bl stays in fp32 on the vector pipe (it feeds exp2); a 3-term bf16 split of it on the matrix pipe (hi*hi + lo*hi +
// hi*lo, 48 MFMAs per tile, same accuracy) was measured and is NOT faster: the split itself costs what the packed FMAs did.
#ifndef SPE_FUSED_MFMAMIX
#define SPE_FUSED_MFMAMIX 1
#endif
typedef short s16x4m_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4m_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8m_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ _Float16 f2h_sat(float f) { return (_Float16)((f != f) ? f : __builtin_amdgcn_fmed3f(f, -65504.f, 65504.f)); }      // NaN stays NaN
// 4 floats -> one 8-B MFMA operand (F16: saturating fp16, else bf16), both round to nearest even
template <bool F16>
__device__ __forceinline__ s16x4m_t pack4(float a, float b, float c, float d) {
    if constexpr (F16) {
        f16x4m_t v; v[0] = f2h_sat(a); v[1] = f2h_sat(b); v[2] = f2h_sat(c); v[3] = f2h_sat(d);
        return __builtin_bit_cast(s16x4m_t, v);
    } else {
        bf16x4_t v; v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
        return __builtin_bit_cast(s16x4m_t, v);
    }
}
template <int H, bool TRANSPOSE, bool F16 = false>
__device__ __forceinline__ void mixA_build(const float* __restrict__ W, int lane, s16x4m_t (&A)[H / 4][H / 4]) {
    const bool nz = ((lane & 15) >> 2) == (lane >> 4);
#pragma unroll
    for (int gh = 0; gh < H / 4; ++gh)
#pragma unroll
        for (int hh = 0; hh < H / 4; ++hh) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int go = 4 * gh + (lane & 3), hi = 4 * hh + i;
                const float w = TRANSPOSE ? W[hi * H + go] : W[go * H + hi];
                v[i] = nz ? w : 0.f;
            }
            A[gh][hh] = pack4<F16>(v[0], v[1], v[2], v[3]);
        }
}
// out[gh][r] = init[4gh + r] + sum_h A(4gh + r, h) x[h]  for ONE key of the lane (x[h]: the H heads' values at that key)
// F16: x and A in fp16 (the forward mix P' = Ww P + bw; x arrives scaled by SPE_PD_SCALE, so does init), else bf16
template <int H, bool F16 = false>
__device__ __forceinline__ void mix_mfma_key(const float (&x)[H], const s16x4m_t (&A)[H / 4][H / 4], const float* init, f32x4_t (&out)[H / 4]) {
#ifdef SPE_DBG_NOMM
#pragma unroll
    for (int gh = 0; gh < H / 4; ++gh) out[gh] = (f32x4_t){x[4 * gh], x[4 * gh + 1], x[4 * gh + 2], x[4 * gh + 3]};
    return;
#endif
    s16x4m_t bv[H / 4];
#pragma unroll
    for (int hh = 0; hh < H / 4; ++hh) bv[hh] = pack4<F16>(x[4 * hh], x[4 * hh + 1], x[4 * hh + 2], x[4 * hh + 3]);
#pragma unroll
    for (int gh = 0; gh < H / 4; ++gh) {
        f32x4_t d = init ? (f32x4_t){init[4 * gh], init[4 * gh + 1], init[4 * gh + 2], init[4 * gh + 3]} : (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int hh = 0; hh < H / 4; ++hh) {
            if constexpr (F16) d = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4m_t, A[gh][hh]), __builtin_bit_cast(f16x4m_t, bv[hh]), d, 0, 0, 0);
            else d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A[gh][hh], bv[hh], d, 0, 0, 0);
        }
        out[gh] = d;
    }
}

// The softmax-input mix S' = Wl S + bl on the matrix pipe IN FP32: v_mfma_f32_4x4x1_16b_f32 computes, in each of its 16 blocks of
// 4 lanes, the outer product D[i][j] += A[i] B[j] with D[i][.] in register i of lane 4b + j (probed on gfx950:
// tools/debug/probe_mfma4x4.hip) - i.e. register i of a lane accumulates (A of lane 4b + i) * (the lane's OWN B).  With
// A := W[4gh + (lane & 3)][h] (a per-lane constant) and B := the lane's score of head h, one instruction adds head h's
// contribution to output heads 4gh .. 4gh+3 of the lane's own (query, key) element: H * H/4 instructions per key are the whole
// H x H mix, bit-for-bit the fmaf chain the packed FMAs computed (an f32 MFMA is a k-ordered fmaf chain) at the same
// FLOP rate (64 / clk / SIMD) - but on the pipe that is ~10 % busy instead of the one that bounds these kernels.  Measured
// (cfg2, isolated): removing the packed-FMA mix altogether is worth 0.049 of the statistics pass' 0.176 ms.
#ifndef SPE_FUSED_MIX4
#define SPE_FUSED_MIX4 1
#endif
template <int H, bool TRANSPOSE>
__device__ __forceinline__ void mixA4_build(const float* __restrict__ W, int lane, float (&A)[H / 4][H]) {
#pragma unroll
    for (int gh = 0; gh < H / 4; ++gh)
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int go = 4 * gh + (lane & 3);
            A[gh][h] = TRANSPOSE ? W[h * H + go] : W[go * H + h];
        }
}
// out[r][gh][i] = c[4gh + i] + sum_h W[4gh + i][h] s[h][r]   (r: the lane's 4 keys)
template <int H>
__device__ __forceinline__ void mix_keys_f32(const f32x4_t (&s)[H], const float (&A)[H / 4][H], const float (&c)[H], f32x4_t (&out)[4][H / 4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int gh = 0; gh < H / 4; ++gh) out[r][gh] = (f32x4_t){c[4 * gh], c[4 * gh + 1], c[4 * gh + 2], c[4 * gh + 3]};
#ifdef SPE_DBG_NOMIX4
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < H; ++g) out[r][g >> 2][g & 3] += s[g][r];
    return;
#endif
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int gh = 0; gh < H / 4; ++gh) out[r][gh] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[gh][h], s[h][r], out[r][gh], 0, 0, 0);
}

// Dropout of the mixed probabilities (reference models/cait.py:387, attn_drop): keep flags for the lane's 4 keys of TWO heads
// from ONE Philox4x32-10 call - the 128 random bits are eight 16-bit lots (keep iff lot >= p * 65536: the rate is exact to
// 1.5e-5).  The counter is (b, head pair, query, 4-key group): aligned by construction, so the write pass and both backward
// passes regenerate identical masks with H/2 calls per lane and tile.  (Round 1 drew one call per ELEMENT on an index that
// is not 4-aligned when N % 4 != 0: 32 calls per lane and tile, 3-5x the cost of everything else in the pass.)
template <int H>
__device__ __forceinline__ void fused_keep_scales(uint64_t seed, uint64_t offset, float p, int b, int hp, int q, int key0, int N,
                                                  float (&s0)[4], float (&s1)[4]) {
    const uint64_t ctr = (((uint64_t)b * (H / 2) + hp) * (uint64_t)N + (uint64_t)q) * (uint64_t)((N + 3) >> 2) + (uint64_t)(key0 >> 2);
    uint32_t o[4];
    spe_philox4((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
    const uint32_t thr = (uint32_t)(p * 65536.0f);
    const float inv = 1.0f / (1.0f - p);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        s0[2 * i] = ((o[i] & 0xffffu) >= thr) ? inv : 0.f;      s0[2 * i + 1] = ((o[i] >> 16) >= thr) ? inv : 0.f;
        s1[2 * i] = ((o[2 + i] & 0xffffu) >= thr) ? inv : 0.f;  s1[2 * i + 1] = ((o[2 + i] >> 16) >= thr) ? inv : 0.f;
    }
}

// the same keep scales from the flags the flash forward stored for this lane and tile (bit hp * 8 + 2 r + e: key r of the lane's group, head 2 hp + e)
__device__ __forceinline__ void fused_keep_from_bits(uint32_t kb, float p, int hp, float (&s0)[4], float (&s1)[4]) {
    const float inv = 1.0f / (1.0f - p);
    const uint32_t w = kb >> (hp * 8);
#pragma unroll
    for (int r = 0; r < 4; ++r) { s0[r] = (w & (1u << (2 * r))) ? inv : 0.f; s1[r] = (w & (2u << (2 * r))) ? inv : 0.f; }
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
#ifndef SPE_FUSED_TAIL1K
#define SPE_FUSED_TAIL1K 0
#endif
template <int DSTEPS, bool TAIL16, bool F16 = false>
__device__ __forceinline__ f32x4_t frag_mfma(int st, u32x4_t a, u32x4_t b, f32x4_t c) {
    // F16: fp16 operands (the K.Q^T products of every pass); the zero-extended tail step works the same way
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8m_t, a), __builtin_bit_cast(f16x8m_t, b), c, 0, 0, 0);
    // SPE_FUSED_TAIL1K = 1 (OFF by default): the 16-dim tail step as v_mfma_f32_16x16x16_bf16 on the 8-B halves, accumulating onto
    // the 16x16x32 result.  Same products, 2 v_mov and 2 live registers less per staged fragment - and NOT SAFE on gfx950 as hipcc
    // (ROCm 7.2) schedules it: a 16x16x16 MFMA whose SrcC is the destination of the 16x16x32 MFMA issued right before it
    // (`v_mfma_f32_16x16x32_bf16 v[14:17], ..; s_waitcnt; v_mfma_f32_16x16x16_bf16 v[18:21], .., v[14:17]`) gives run-to-run
    // different results (tools/debug/race_mode2.py: 51-386 of 300-400 launches at H = 4, dropout on; 0 with 16 wait states in
    // front of the tail instruction, 0 with the zero-extended 32-deep form).  Accumulate chains therefore stay within ONE MFMA
    // shape everywhere in this library; the measured cost of the zero-extended form is < 1 % per pass.
    if (SPE_FUSED_TAIL1K == 1 && TAIL16 && st == DSTEPS - 1) {
        typedef unsigned u32x2f_t __attribute__((ext_vector_type(2)));
        typedef short s16x4f_t __attribute__((ext_vector_type(4)));
        const u32x2f_t al = {a[0], a[1]}, bl = {b[0], b[1]};
#ifdef SPE_DBG_TAILNOP
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#endif
        f32x4_t d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4f_t, al), __builtin_bit_cast(s16x4f_t, bl), c, 0, 0, 0);
#ifndef SPE_DBG_NOKEEP
        // keep the operands alive past the instruction: hipcc otherwise allocates the destination over a dying source operand
        // (`v_mfma_f32_16x16x16_bf16 v[76:79], v[86:87], v[78:79], v[14:17]`), and on gfx950 that form gives run-to-run different
        // results (found with tools/debug/race_mode2.py: 386 of 400 launches differ at H = 4, N = 1100, dropout on)
        asm volatile("" :: "v"(al), "v"(bl), "v"(d));
#endif
        return d;
    }
    if (SPE_FUSED_TAIL1K == 2 && TAIL16) {
        // every step on ONE shape: a 32-deep step as two 16-deep MFMAs on the low / high halves of the same operand registers
        // (lane group g holds dims 8g..8g+7 of the step in both operands, so the halves pair up), the tail as one
        typedef unsigned u32x2g_t __attribute__((ext_vector_type(2)));
        typedef short s16x4g_t __attribute__((ext_vector_type(4)));
        const u32x2g_t al = {a[0], a[1]}, bl = {b[0], b[1]};
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4g_t, al), __builtin_bit_cast(s16x4g_t, bl), c, 0, 0, 0);
        if (st == DSTEPS - 1) return c;
        const u32x2g_t ah = {a[2], a[3]}, bh = {b[2], b[3]};
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4g_t, ah), __builtin_bit_cast(s16x4g_t, bh), c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
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

template <int H, int DSTEPS, bool TAIL16, int MODE, bool DROP, int KT>
#ifndef SPE_FUSED_MINW
#define SPE_FUSED_MINW 2
#endif
#ifndef SPE_FUSED_JB2
#define SPE_FUSED_JB2 4
#endif
#ifndef SPE_FUSED_MINW01
#define SPE_FUSED_MINW01 SPE_FUSED_MINW
#endif
#ifndef SPE_FUSED_QREG01
#define SPE_FUSED_QREG01 1
#endif
__global__ __launch_bounds__(256, (MODE <= 1) ? SPE_FUSED_MINW01 : SPE_FUSED_MINW) void talking_fused_kernel(FusedArgs a) {
    constexpr int NFR = H * DSTEPS;                        // fragments (16 B per lane) per q-tile
    // MFMA jobs per operand-fragment batch: whole tile in the forward passes; backward pass 1 has registers for 4 heads at a
    // time, backward pass 2 - since its two transposed mixes left the vector pipe - for 8 (0.380 -> 0.356 ms at cfg2; pass 1
    // spills with 8: 0.37 -> 0.43 ms)
#ifndef SPE_FUSED_GWMFMA
#define SPE_FUSED_GWMFMA 1
#endif
#ifndef SPE_FUSED_JB2G
#define SPE_FUSED_JB2G 2
#endif
#ifndef SPE_FUSED_GWMFMA3
#define SPE_FUSED_GWMFMA3 1
#endif
    // weight-gradient outer products on the matrix pipe (below): backward pass 1 (dWw, dbw) and backward pass 2 (dWl, dbl)
#ifdef SPE_DBG_NOGWM
    constexpr bool GWM = SPE_FUSED_GWMFMA && (H % 4 == 0) && (MODE == 2 || (MODE == 3 && SPE_FUSED_GWMFMA3));
#define SPE_GWM_BODY 0
#else
    constexpr bool GWM = SPE_FUSED_GWMFMA && (H % 4 == 0) && (MODE == 2 || (MODE == 3 && SPE_FUSED_GWMFMA3));
#define SPE_GWM_BODY 1
#endif
    // backward pass 1 with the outer products on the matrix pipe has ~36 registers to spare: 8 heads per fragment batch there too
#ifndef SPE_FUSED_JB3
#define SPE_FUSED_JB3 8
#endif
#ifndef SPE_FUSED_PREF3
#define SPE_FUSED_PREF3 0
#endif
    constexpr int JBW = (MODE == 3) ? SPE_FUSED_JB3 : (GWM ? SPE_FUSED_JB2G * SPE_FUSED_JB2 : SPE_FUSED_JB2);
    constexpr int JB = (MODE >= 2) ? ((H >= JBW) ? JBW : ((H >= SPE_FUSED_JB2) ? SPE_FUSED_JB2 : H)) : H;
    // request the next macro step's first batch before the VALU phases (its registers stay live through them)
// (round 3: the prefetch of modes 1 and 2 is OFF - measured INSIDE the training step, interleaved same-box runs: 57.22 -> 56.68 ms per
    // step without it; the isolated launches it was tuned on preferred it.  The registers it holds through the VALU phases cost more there.)
#ifndef SPE_FUSED_PREF1
#define SPE_FUSED_PREF1 0
#endif
#ifndef SPE_FUSED_QG
#define SPE_FUSED_QG 2
#endif
    // modes 0, 1 keep the q-tile's Q fragments in registers (64 VGPRs at H = 8, dh <= 64); the backward modes have no
    // room and read them (and dO) from LDS in groups of QG jobs
    constexpr bool QREG = (MODE <= 1) && SPE_FUSED_QREG01;
    constexpr int QG = QREG ? ((JB >= 4) ? 4 : JB) : ((JB >= SPE_FUSED_QG) ? SPE_FUSED_QG : JB);
#ifndef SPE_FUSED_PREF2
#define SPE_FUSED_PREF2 0
#endif
#ifndef SPE_FUSED_PREF0
#define SPE_FUSED_PREF0 1
#endif
    constexpr bool PREF = (MODE == 0 && SPE_FUSED_PREF0) || (MODE == 1 && SPE_FUSED_PREF1) || (MODE == 2 && !DROP && SPE_FUSED_PREF2) || (MODE == 3 && !DROP && SPE_FUSED_PREF3);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int QP = (MODE <= 1) ? SPE_FUSED_QP : 1, WPQ = 4 / QP;        // q-tiles per workgroup, waves per q-tile
    u32x4_t* sQ = reinterpret_cast<u32x4_t*>(smem_raw);    // [QP][NFR][64]
    u32x4_t* sdO = sQ + QP * NFR * 64;                     // [QP][NFR][64]   (modes 2, 3)
    float* sred = reinterpret_cast<float*>(sdO + ((MODE >= 2) ? QP * NFR * 64 : 0));   // [4][H][16][2]

    // the wave index as a SCALAR: everything derived from it (macro step, key tile, fragment record addresses, tail masks) then lives
    // in SGPRs and the fragment loads take an SGPR base + one per-lane offset.  With threadIdx.x >> 6 the compiler cannot prove
    // uniformity and rebuilt every 64-bit record address on the vector pipe (16 v_mad_u64_u32 + ~70 moves/adds per tile in the
    // forward passes, twice that in the backward ones: ~20 % of the issue slots of kernels that are vector-pipe bound)
#ifndef SPE_FUSED_SWAVE
#define SPE_FUSED_SWAVE 1
#endif
    const int lane = threadIdx.x & 63, wave = SPE_FUSED_SWAVE ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (threadIdx.x >> 6);
    const int nt = a.nt, N = a.N;
    const int nch = fused_nch(nt);
    const int chunk = (blockIdx.x & 7) % nch, wg_j = (blockIdx.x >> 3) * (8 / nch) + (blockIdx.x & 7) / nch;   // index within the chunk
    const int kbeg = fused_kbeg(chunk, nt, nch), klen = fused_kbeg(chunk + 1, nt, nch) - kbeg;
    const int npair = fused_npair(nt, MODE);
    const long total_c = (long)a.B * npair * klen;
    const long s_begin = (long)wg_j * a.steps_per_wg;
    long s_end = s_begin + a.steps_per_wg; if (s_end > total_c) s_end = total_c;

    // Scores arrive in the log2 domain (spe_attn_pack folds scale * log2(e) into the Q fragments): Wl S + bl*log2(e)
    // is log2(e) * S', so every exponential is a bare v_exp_f32.
    float vbl2[H], vbw[H];
#pragma unroll
    for (int g = 0; g < H; ++g) { vbl2[g] = a.bl[g] * SPE_LOG2E; vbw[g] = a.bw[g]; }
    // block-diagonal weight operands of the matrix-pipe mixes (see mix_mfma_key)
    constexpr bool MM = SPE_FUSED_MFMAMIX && (H % 4 == 0);
    s16x4m_t Aw[(MM && MODE >= 1) ? H / 4 : 1][(MM && MODE >= 1) ? H / 4 : 1];       // mode 1: Ww ; modes 2, 3: Ww^T
    s16x4m_t Al[(MM && MODE == 3) ? H / 4 : 1][(MM && MODE == 3) ? H / 4 : 1];       // mode 3: Wl^T
    if constexpr (MM && MODE == 1) mixA_build<H, false, true>(a.Ww, lane, Aw);          // forward mix: fp16 operands
    // round 5: the backward's two 16-bit mixes as 4-lane-block matrix instructions (v_mfma_f32_4x4x4_16b_bf16, 11 cycles each - the form the
    // flash kernels use, attn_flash_common.h) instead of block-diagonal 16x16x16 ones (18 cycles each)
    if constexpr (MM && MODE >= 2) fl_mixA_16<H, true, false>(a.Ww, lane, 1.0f, Aw);
    if constexpr (MM && MODE == 3) fl_mixA_16<H, true, false>(a.Wl, lane, 1.0f, Al);
    constexpr bool M4 = SPE_FUSED_MIX4 && (H % 4 == 0);
    float Al4[M4 ? H / 4 : 1][M4 ? H : 1];                 // f32 operand of the S' mix (see mix_keys_f32)
    if constexpr (M4) mixA4_build<H, false>(a.Wl, lane, Al4);
    // Weight-gradient outer products on the MATRIX pipe (GWM): dW[g][h] = sum over (query, key) positions of x_g * y_h (mode 2:
    // x = dP', y = P -> dWw, dbw) is a contraction over POSITIONS, i.e. D[m = g][n = h] += A[g][pos] B[pos][h] with 16 positions per
    // v_mfma_f32_16x16x16_bf16.  A lane owns ONE query and 4 keys for all heads, the operands want one HEAD per lane (m = lane & 15)
    // and 4 keys of query t for the t-th instruction: a (query x head) transpose inside each 16-lane group, done through a
    // wave-private LDS tile [key group][head][query] of 8-B packets (written with 8 ds_write_b64 per operand, read back as 128
    // contiguous bytes per lane).  Rows m >= H read a block of zeros, row m = H of the B operand a block of ones: column H of D is
    // the bias gradient.  16 MFMAs per tile replace H*H/2 * 4 = 128 v_pk_fma_f32 and the H*H/2 accumulator registers (4 x 4 here)
    // - the vector pipe, not the matrix pipe, bounds these passes; packed fp32 FMAs next to MFMAs are an
    // anti-lever on this chip (MI355X_MICROARCH.md, price of a filler beside MFMAs).
    constexpr int GW_RED = 4 * (H * 16 * 2 > (H * H + H) ? H * 16 * 2 : (H * H + H));        // floats of sred
    // round 5: row pitch 144 B instead of 128 (the 16-B reads of 8 heads then fall on distinct bank groups - SQ_LDS_BANK_CONFLICT was half of
    // this kernel's LDS cycles: profiles/r05_fused_pmc.txt), constant blocks on banks no row uses (see attn_flash_bwd.hip)
    constexpr int GWR = 144, GWT = 4 * H * GWR;
    unsigned char* gconst = reinterpret_cast<unsigned char*>(sred + GW_RED);                  // 512 B: zeros at + 48 (128 B), bf16 ones at + 208 (128 B)
    unsigned char* sgw = gconst + 512 + wave * (2 * GWT);                                     // this wave's [X | Y] tiles
    f32x4_t gwacc[GWM ? 4 : 1];
    const unsigned char* gw_xrd = nullptr; const unsigned char* gw_yrd = nullptr; unsigned char* gw_wr = nullptr;
    if constexpr (GWM) {
        if (threadIdx.x < 64) reinterpret_cast<uint2*>(gconst)[threadIdx.x] = (threadIdx.x >= 26 && threadIdx.x < 42) ? make_uint2(0x3F803F80u, 0x3F803F80u) : make_uint2(0u, 0u);
        const int gm = lane & 15, gk = lane >> 4;
        gw_wr = sgw + (gk * H) * GWR + gm * 8;                                  // + h * GWR: packet of head h, query gm, key group gk
        gw_xrd = (gm < H) ? sgw + (gk * H + gm) * GWR : gconst + 48;            // 16 packets (queries 0..15) of head gm
        gw_yrd = (gm < H) ? sgw + GWT + (gk * H + gm) * GWR : ((gm == H) ? gconst + 208 : gconst + 48);
#pragma unroll
        for (int i = 0; i < 4; ++i) gwacc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    // weight-gradient accumulators (whole workgroup range)
    // as head pairs: mode 2 gWp[g*(H/2)+hp] = (dWw[g][2hp], dWw[g][2hp+1]) ; mode 3 gWp[gp*H+h] = (dWl[2gp][h], dWl[2gp+1][h])
    f32x2_t gWp[(MODE >= 2 && !GWM) ? H * H / 2 : 1], gb2[(MODE == 3) ? H / 2 : 1];
    float gb[(MODE == 2 && !GWM) ? H : 1];
    if (MODE == 3 && GWM) {
#pragma unroll
        for (int g = 0; g < H / 2; ++g) gb2[(MODE == 3) ? g : 0] = splat2(0.f);
    }
    if (MODE >= 2 && !GWM) {
#pragma unroll
        for (int i = 0; i < H * H / 2; ++i) gWp[i] = splat2(0.f);
        if (MODE == 2) {
#pragma unroll
            for (int g = 0; g < H; ++g) gb[g] = 0.f;
        } else {
#pragma unroll
            for (int g = 0; g < H / 2; ++g) gb2[g] = splat2(0.f);
        }
    }

    long s = s_begin;
    while (s < s_end) {
        const int bqp = (int)(s / klen), kt0 = kbeg + (int)(s % klen);
        int seg = kbeg + klen - kt0; if (seg > s_end - s) seg = (int)(s_end - s);
        const int b = bqp / npair, qp = bqp % npair;
        const int qt_own = qp * QP + wave / WPQ;              // this wave's q-tile; past the last tile (odd tile count): the wave idles
        const bool qt_valid = qt_own < nt;
        const int qt = qt_valid ? qt_own : nt - 1;
        const int bq = b * nt + qt;
        const int q = qt * 16 + (lane & 15);
        const bool qv = qt_valid && q < N;
        // ---- stage the q-tiles' Q (and dO) fragments in LDS
        u32x4_t qreg[QREG ? NFR : 1];
        const u32x4_t* sQw = sQ + (wave / WPQ) * NFR * 64;
        const u32x4_t* sdOw = sdO + (wave / WPQ) * NFR * 64;
        if (QREG) {
            __syncthreads();                                   // sred of the previous segment has been consumed
#pragma unroll
            for (int f = 0; f < NFR; ++f) qreg[QREG ? f : 0] = frag_load<DSTEPS, TAIL16>(a.Qf, ((long)b * H + f / DSTEPS) * nt + qt, f % DSTEPS, lane);
        } else {
            __syncthreads();
#ifdef SPE_DBG_NOSTAGE      // timing experiment: Q / dO staged for the workgroup's first segment only
            if (s == s_begin)
#endif
            for (int i = threadIdx.x; i < QP * NFR * 64; i += 256) {
                const int u = i / (NFR * 64), fr = (i >> 6) % NFR, ln = i & 63, h = fr / DSTEPS, st = fr % DSTEPS;
                const long rec = ((long)b * H + h) * nt + min(qp * QP + u, nt - 1);
                sQ[i] = frag_load<DSTEPS, TAIL16>(a.Qf, rec, st, ln);
                if (MODE >= 2) sdO[i] = frag_load<DSTEPS, TAIL16>(a.dOf, rec, st, ln);
            }
            __syncthreads();
        }
        // ---- per-lane row state
        float rm[(MODE == 0) ? H : 1], rl[(MODE == 0) ? H : 1], c0[(MODE >= 1) ? H : 1];
        f32x2_t rD2[(MODE >= 2) ? H / 2 : 1];
#pragma unroll
        for (int g = 0; g < H; ++g) {
            if (MODE == 0) { rm[g] = -INFINITY; rl[g] = 0.f; }
            else {
                // P = exp2(S' + c0), c0 = bl - m + log2(1/l): max subtraction and normalisation folded into the mix's addend
                const long si = ((long)b * H + g) * N + (qv ? q : 0);
                c0[g] = vbl2[g] - a.M[si] + __builtin_amdgcn_logf(a.IL[si]);
            }
            if (MODE == 2) rD2[g / 2][g & 1] = 0.f;
            if (MODE == 3) rD2[g / 2][g & 1] = qv ? a.D[((long)b * H + g) * N + q] : 0.f;
        }

        // A wave's unit of work is a macro step of KT consecutive 16-key tiles against the 16 queries of the
        // q-tile: the mixing weights (SGPR reload), the Q / dO fragments (LDS) and the loop overhead are paid once
        // per macro step.  Waves take macro steps round-robin.
        // operand-fragment staging registers for one batch of jobs, and the batch loader: job jb of a macro step is
        // (tile jb / NH, head job jb % NH); head jobs >= H read the V fragments
        u32x4_t fr[JB * DSTEPS];
        constexpr unsigned RECB = (unsigned)((DSTEPS - (TAIL16 ? 1 : 0)) * 1024 + (TAIL16 ? 512 : 0));   // bytes per fragment record
#ifndef SPE_FUSED_ROWPTR
#define SPE_FUSED_ROWPTR 1
#endif
        const char* krow[SPE_FUSED_ROWPTR ? H : 1];
        const char* vrow[(SPE_FUSED_ROWPTR && MODE >= 2) ? H : 1];
        if (SPE_FUSED_ROWPTR) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                krow[SPE_FUSED_ROWPTR ? h : 0] = reinterpret_cast<const char*>(a.Kf) + ((long)b * H + h) * nt * (long)RECB;
                if (MODE >= 2) vrow[(SPE_FUSED_ROWPTR && MODE >= 2) ? h : 0] = reinterpret_cast<const char*>(a.Vf) + ((long)b * H + h) * nt * (long)RECB;
            }
        }
        auto load_batch = [&](int bi, int kt_first_) {
            constexpr int NH_ = (MODE >= 2) ? 2 * H : H;
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) {
                const int jb = bi * JB + jj;
                const int hj = jb % NH_, tj = jb / NH_;
                const int ktl = min(kt_first_ + tj, nt - 1);
                if (SPE_FUSED_ROWPTR) {
                    const char* row = (hj < H) ? krow[SPE_FUSED_ROWPTR ? hj % H : 0] : vrow[(SPE_FUSED_ROWPTR && MODE >= 2) ? hj % H : 0];
#pragma unroll
                    for (int st = 0; st < DSTEPS; ++st) fr[jj * DSTEPS + st] = frag_load_row<DSTEPS, TAIL16>(row, (unsigned)ktl * RECB, st, lane);
                } else {
                    const u32x4_t* srcp = (hj < H) ? a.Kf : a.Vf;
#pragma unroll
                    for (int st = 0; st < DSTEPS; ++st) fr[jj * DSTEPS + st] = frag_load<DSTEPS, TAIL16>(srcp, ((long)b * H + (hj % H)) * nt + ktl, st, lane);
                }
            }
        };
#define KM0 (wave % WPQ)
#define KMS WPQ
        for (int km = KM0; qt_valid && km * KT < seg; km += KMS) {
            const int kt_first = kt0 + km * KT;
            // ---- raw scores of all heads acc[j][h] = K_tile(h).Q_tile(h)^T (and acc2[j][g] = V_tile(g).dO_tile(g)^T).
            // The (operand, tile, head) jobs are issued in batches of JB: the JB*DSTEPS operand fragments of a batch
            // are requested together (one L2 round trip per batch instead of one per head - the kernel was ~55 %
            // s_waitcnt-bound), and the first batch of the NEXT macro step is requested right after the last MFMA of
            // this one, so it lands during the VALU phases.
            constexpr int NH = (MODE >= 2) ? 2 * H : H;
            constexpr int NJ = NH * KT;
            constexpr int NB = NJ / JB;
            f32x4_t acc[KT][H];
            f32x4_t acc2[(MODE >= 2) ? KT : 1][(MODE >= 2) ? H : 1];
            // dropout masks of this macro step's tiles, when the forward stored them: requested here, ahead of the MFMA phase that hides the trip
            uint32_t kbits[(MODE >= 2 && DROP) ? KT : 1];
            const bool have_bits = MODE >= 2 && DROP && a.keepbits != nullptr;
            if (have_bits) {
#pragma unroll
                for (int j = 0; j < KT; ++j)
                    kbits[(MODE >= 2 && DROP) ? j : 0] = a.keepbits[(((long)b * nt + qt) * nt + min(kt_first + j, nt - 1)) * 64 + lane];
            }
#ifdef SPE_DBG_NOLOAD
            if (km == KM0) load_batch(0, kt_first);
#else
            if (!PREF || km == KM0) load_batch(0, kt_first);
#endif   // first macro step of the segment: nothing prefetched yet
#pragma unroll
            for (int bi = 0; bi < NB; ++bi) {
                // jobs of a batch go through the matrix pipe in groups of QG: the group's Q / dO fragments are read
                // from LDS together (modes 0, 1: they live in registers for the whole segment) and the MFMAs are
                // issued d-step outer / job inner, so that consecutive instructions are independent (a per-job
                // read -> wait -> 2 dependent MFMAs sequence serialises ~150 cycles per head).
#pragma unroll
                for (int g0 = 0; g0 < JB; g0 += QG) {
                    u32x4_t qf[QREG ? 1 : QG * DSTEPS];
                    f32x4_t c[QG];
                    if (!QREG) {
#pragma unroll
                        for (int jj = 0; jj < QG; ++jj) {
                            const int hj = (bi * JB + g0 + jj) % NH;
                            const u32x4_t* lds = (hj < H) ? sQw : sdOw;
#pragma unroll
                            for (int st = 0; st < DSTEPS; ++st) qf[jj * DSTEPS + st] = lds[((hj % H) * DSTEPS + st) * 64 + lane];
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < QG; ++jj) c[jj] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int st = 0; st < DSTEPS; ++st)
#pragma unroll
                        for (int jj = 0; jj < QG; ++jj) {
                            const int hj = (bi * JB + g0 + jj) % NH;
                            const u32x4_t qv4 = QREG ? qreg[QREG ? (hj % H) * DSTEPS + st : 0] : qf[QREG ? 0 : jj * DSTEPS + st];
                            c[jj] = (hj < H) ? frag_mfma<DSTEPS, TAIL16, true>(st, fr[(g0 + jj) * DSTEPS + st], qv4, c[jj])
                                             : frag_mfma<DSTEPS, TAIL16, false>(st, fr[(g0 + jj) * DSTEPS + st], qv4, c[jj]);
                        }
#pragma unroll
                    for (int jj = 0; jj < QG; ++jj) {
                        const int jb = bi * JB + g0 + jj;
                        const int hj = jb % NH, tj = jb / NH, hh = hj % H;   // head job (>= H: V/dO), tile within the macro step
                        if (hj < H) acc[tj][hh] = c[jj]; else acc2[(MODE >= 2) ? tj : 0][(MODE >= 2) ? hh : 0] = c[jj];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#ifndef SPE_DBG_NOLOAD
                if (bi + 1 < NB) load_batch(bi + 1, kt_first);
                else if (PREF && (km + KMS) * KT < seg) load_batch(0, kt0 + (km + KMS) * KT);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            // this lane's 4 consecutive keys of tile j: kb(j) + r ; tiles past the segment end belong to another workgroup
#define KB(j) ((kt_first + (j)) * 16 + 4 * (lane >> 4))
#define TV(j) (kt_first + (j) < kt0 + seg)
            // wave-uniform: tile j needs per-key masking (not this workgroup's tile, or the ragged last tile)
#define TMASK(j) (!TV(j) || (kt_first + (j) == nt - 1 && (N & 15) != 0))
#define KVAL(j, r) (TV(j) && KB(j) + (r) < N)

            // Every mode runs as two register phases that each use ONE mixing matrix, so that the 64 weights
            // of a phase stay in SGPRs (both matrices together do not fit the scalar file).  All H x H mixes are
            // packed-fp32 FMAs (v_pk_fma_f32): either pairs over a lane's adjacent keys with a broadcast SGPR weight
            // (mix_rows) or pairs over adjacent heads with an SGPR weight pair and a broadcast (op_sel) operand.
            if constexpr (MODE == 0 && M4) {
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
            } else if constexpr (MODE == 0) {
                // phase A (Wl): S' in place + macro-step max ; then one rescale + 4*KT exp2 per head
                float wl[H][H];
                load_w<H>(a.Wl, wl);
                float tmax[H];
#pragma unroll
                for (int g = 0; g < H; ++g) tmax[g] = -INFINITY;
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    f32x2_t lo[H], hi[H];
                    mix_rows<H>(acc[j], wl, vbl2, lo, hi);
                    if (TMASK(j)) {
                        const bool k0 = KVAL(j, 0), k1 = KVAL(j, 1), k2 = KVAL(j, 2), k3 = KVAL(j, 3);
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            lo[g][0] = k0 ? lo[g][0] : -INFINITY; lo[g][1] = k1 ? lo[g][1] : -INFINITY;
                            hi[g][0] = k2 ? hi[g][0] : -INFINITY; hi[g][1] = k3 ? hi[g][1] : -INFINITY;
                        }
                    }
#pragma unroll
                    for (int g = 0; g < H; ++g) {
                        tmax[g] = fmaxf(fmaxf(tmax[g], lo[g][0]), lo[g][1]);
                        tmax[g] = fmaxf(fmaxf(tmax[g], hi[g][0]), hi[g][1]);
                        acc[j][g] = PCAT(lo[g], hi[g]);
                    }
                }
#pragma unroll
                for (int g = 0; g < H; ++g) {
                    const float mn = fmaxf(rm[g], tmax[g]);
                    if (mn > -INFINITY) {
                        f32x2_t sum2 = {0.f, 0.f};
                        const f32x2_t mn2 = splat2(mn);
#pragma unroll
                        for (int j = 0; j < KT; ++j) {
                            const f32x2_t xl = PLO(acc[j][g]) - mn2, xh = PHI(acc[j][g]) - mn2;
                            sum2 += (f32x2_t){EXP2(xl[0]), EXP2(xl[1])};
                            sum2 += (f32x2_t){EXP2(xh[0]), EXP2(xh[1])};
                        }
                        rl[g] = rl[g] * EXP2(rm[g] - mn) + (sum2[0] + sum2[1]);
                        rm[g] = mn;
                    }
                }
            } else if constexpr (MODE == 1) {
                // phase A (Wl): acc <- P = exp2(S' + c0)   (c0 = bl - m + log2(1/l), all in the log2 domain)
                if constexpr (M4) {
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    f32x4_t sp[4][H / 4];
                    mix_keys_f32<H>(acc[j], Al4, c0, sp);
#pragma unroll
                    for (int g = 0; g < H; ++g)
                        acc[j][g] = (f32x4_t){EXP2(sp[0][g >> 2][g & 3]), EXP2(sp[1][g >> 2][g & 3]), EXP2(sp[2][g >> 2][g & 3]), EXP2(sp[3][g >> 2][g & 3])};
                    if (TMASK(j)) {
                        const bool k0 = KVAL(j, 0), k1 = KVAL(j, 1), k2 = KVAL(j, 2), k3 = KVAL(j, 3);
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            acc[j][g][0] = k0 ? acc[j][g][0] : 0.f; acc[j][g][1] = k1 ? acc[j][g][1] : 0.f;
                            acc[j][g][2] = k2 ? acc[j][g][2] : 0.f; acc[j][g][3] = k3 ? acc[j][g][3] : 0.f;
                        }
                    }
                }
                } else {
                float wl[H][H];
                load_w<H>(a.Wl, wl);
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    f32x2_t lo[H], hi[H];
                    mix_rows<H>(acc[j], wl, c0, lo, hi);
#pragma unroll
                    for (int g = 0; g < H; ++g) acc[j][g] = (f32x4_t){EXP2(lo[g][0]), EXP2(lo[g][1]), EXP2(hi[g][0]), EXP2(hi[g][1])};
                    if (TMASK(j)) {
                        const bool k0 = KVAL(j, 0), k1 = KVAL(j, 1), k2 = KVAL(j, 2), k3 = KVAL(j, 3);
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            acc[j][g][0] = k0 ? acc[j][g][0] : 0.f; acc[j][g][1] = k1 ? acc[j][g][1] : 0.f;
                            acc[j][g][2] = k2 ? acc[j][g][2] : 0.f; acc[j][g][3] = k3 ? acc[j][g][3] : 0.f;
                        }
                    }
                }
                }
                __builtin_amdgcn_sched_barrier(0);
                // phase B (Ww): P'd = dropout(Ww P + bw) -> bf16
                float ww[MM ? 1 : H][MM ? 1 : H];
                if constexpr (!MM) load_w<H>(a.Ww, ww);
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    f32x2_t lo[H], hi[H];
                    if constexpr (MM) {
                        // P * 2^8 in fp16 against fp16 Ww, onto bw * 2^8: the result is P' * 2^8, the stored scale
                        f32x4_t pr[4][H / 4];
                        float vbws[H];
#pragma unroll
                        for (int g = 0; g < H; ++g) vbws[g] = vbw[g] * SPE_PD_SCALE;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float x[H];
#pragma unroll
                            for (int h = 0; h < H; ++h) x[h] = acc[j][h][r] * SPE_PD_SCALE;
                            mix_mfma_key<H, true>(x, Aw, vbws, pr[r]);
                        }
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            lo[g] = (f32x2_t){pr[0][g >> 2][g & 3], pr[1][g >> 2][g & 3]};
                            hi[g] = (f32x2_t){pr[2][g >> 2][g & 3], pr[3][g >> 2][g & 3]};
                        }
                    } else {
                        mix_rows<H>(acc[j], ww, vbw, lo, hi);
#pragma unroll
                        for (int g = 0; g < H; ++g) { lo[g] *= splat2(SPE_PD_SCALE); hi[g] *= splat2(SPE_PD_SCALE); }
                    }
                    if (DROP) {
#pragma unroll
                        for (int hp = 0; hp < H / 2; ++hp) {
                            float k0[4], k1[4];
                            fused_keep_scales<H>(a.seed, a.offset, a.p_drop, b, hp, q, KB(j), N, k0, k1);
                            lo[2 * hp][0] *= k0[0]; lo[2 * hp][1] *= k0[1]; hi[2 * hp][0] *= k0[2]; hi[2 * hp][1] *= k0[3];
                            lo[2 * hp + 1][0] *= k1[0]; lo[2 * hp + 1][1] *= k1[1]; hi[2 * hp + 1][0] *= k1[2]; hi[2 * hp + 1][1] *= k1[3];
                        }
                    }
                    // lane owns 4 consecutive keys of row q: one 8-B store per head
                    if (TV(j)) {
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            const s16x4m_t o = pack4<true>(lo[g][0], lo[g][1], hi[g][0], hi[g][1]);       // fp16(P'd * 2^8)
#ifndef FUSED_PLAIN_STORE
                            {   // non-temporal: the 554 MB of blocks are read next by another kernel, never again by this one -
                                // kept out of L2 they neither evict the K / V fragments nor leave dirty lines for the
                                // contraction that follows to wait on (write pass + PV contraction 0.382 -> 0.347 ms)
                                typedef unsigned u32x2nt_t __attribute__((ext_vector_type(2)));
                                __builtin_nontemporal_store(__builtin_bit_cast(u32x2nt_t, o),
                                                            reinterpret_cast<u32x2nt_t*>(a.outT + (((((long)b * H + g) * nt + qt) * nt + (kt_first + j)) * 64 + lane) * 4));
                            }
#else
                            *reinterpret_cast<s16x4m_t*>(a.outT + (((((long)b * H + g) * nt + qt) * nt + (kt_first + j)) * 64 + lane) * 4) = o;
#endif
                        }
                    }
                }
            } else if constexpr (MODE == 2) {
                // phase A (Wl): PT[j][r][hp] <- P of heads (2hp, 2hp+1) at key r  (head pairs adjacent: the exp2
                // results can be written to any register, so the transposition is free)
                f32x2_t PT[KT][4][H / 2];
                {
                float wl[M4 ? 1 : H][M4 ? 1 : H];
                if constexpr (!M4) load_w<H>(a.Wl, wl);
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    const bool tm = TMASK(j);
                    const bool k0 = !tm || KVAL(j, 0), k1 = !tm || KVAL(j, 1), k2 = !tm || KVAL(j, 2), k3 = !tm || KVAL(j, 3);
                    if constexpr (M4) {
                        f32x4_t sp[4][H / 4];
                        mix_keys_f32<H>(acc[j], Al4, c0, sp);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int g = 0; g < H; ++g) PT[j][r][g / 2][g & 1] = EXP2(sp[r][g >> 2][g & 3]);
                    } else {
                    f32x2_t lo[H], hi[H];
                    mix_rows<H>(acc[j], wl, c0, lo, hi);
#pragma unroll
                    for (int g = 0; g < H; ++g) {
                        PT[j][0][g / 2][g & 1] = EXP2(lo[g][0]); PT[j][1][g / 2][g & 1] = EXP2(lo[g][1]);
                        PT[j][2][g / 2][g & 1] = EXP2(hi[g][0]); PT[j][3][g / 2][g & 1] = EXP2(hi[g][1]);
                    }
                    }
                    if (tm) {
#pragma unroll
                        for (int hp = 0; hp < H / 2; ++hp) {
                            if (!k0) PT[j][0][hp] = splat2(0.f);
                            if (!k1) PT[j][1][hp] = splat2(0.f);
                            if (!k2) PT[j][2][hp] = splat2(0.f);
                            if (!k3) PT[j][3][hp] = splat2(0.f);
                        }
                    }
                }
                }
                __builtin_amdgcn_sched_barrier(0);
                // phase B (Ww): dP' = (dO V^T) keepscale ; dWw += dP' P^T ; dbw += dP' ; D += (Ww^T dP') . P
                float ww[MM ? 1 : H][MM ? 1 : H];
                if constexpr (!MM) load_w<H>(a.Ww, ww);
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    // rows q >= N have zero dO fragments and keys >= N zero V fragments, so dP' is already 0 there
                    if (!TV(j)) {
#pragma unroll
                        for (int g = 0; g < H; ++g) acc2[j][g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                    }
                    if (DROP) {
#pragma unroll
                        for (int hp = 0; hp < H / 2; ++hp) {
                            float k0[4], k1[4];
                            if (have_bits) fused_keep_from_bits(kbits[(MODE >= 2 && DROP) ? j : 0], a.p_drop, hp, k0, k1);
                            else fused_keep_scales<H>(a.seed, a.offset, a.p_drop, b, hp, q, KB(j), N, k0, k1);
#pragma unroll
                            for (int r = 0; r < 4; ++r) { acc2[j][2 * hp][r] *= k0[r]; acc2[j][2 * hp + 1][r] *= k1[r]; }
                        }
                    }
                    if constexpr (GWM && SPE_GWM_BODY) {
                        // dWw += dP' P^T, dbw += dP' over this tile's 256 positions: transpose through the wave's LDS tile, 16 MFMAs
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            *reinterpret_cast<s16x4m_t*>(gw_wr + g * GWR) = pack4<false>(acc2[j][g][0], acc2[j][g][1], acc2[j][g][2], acc2[j][g][3]);
                            *reinterpret_cast<s16x4m_t*>(gw_wr + GWT + g * GWR) =
                                pack4<false>(PT[j][0][g / 2][g & 1], PT[j][1][g / 2][g & 1], PT[j][2][g / 2][g & 1], PT[j][3][g / 2][g & 1]);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // wave-private tile: the other lanes' packets are read next
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const u32x4_t xa = *reinterpret_cast<const u32x4_t*>(gw_xrd + c * 16);
                            const u32x4_t yb = *reinterpret_cast<const u32x4_t*>(gw_yrd + c * 16);
                            typedef unsigned u32x2w_t __attribute__((ext_vector_type(2)));
                            gwacc[(2 * c) & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4m_t, (u32x2w_t){xa[0], xa[1]}),
                                                                                             __builtin_bit_cast(s16x4m_t, (u32x2w_t){yb[0], yb[1]}), gwacc[(2 * c) & 3], 0, 0, 0);
                            gwacc[(2 * c + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4m_t, (u32x2w_t){xa[2], xa[3]}),
                                                                                                 __builtin_bit_cast(s16x4m_t, (u32x2w_t){yb[2], yb[3]}), gwacc[(2 * c + 1) & 3], 0, 0, 0);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the tile is rewritten by the next key tile
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        f32x2_t mt[H / 2];
                        if constexpr (MM) {
                            float x[H];
#pragma unroll
                            for (int g = 0; g < H; ++g) x[g] = acc2[j][g][r];
                            f32x4_t m4[H / 4];
                            fl_mix_16<H, false>(x, Aw, nullptr, m4);               // dP = Ww^T dP' of this key
#pragma unroll
                            for (int hp = 0; hp < H / 2; ++hp) mt[hp] = (f32x2_t){m4[hp >> 1][2 * (hp & 1)], m4[hp >> 1][2 * (hp & 1) + 1]};
                        } else {
#pragma unroll
                            for (int hp = 0; hp < H / 2; ++hp) mt[hp] = splat2(0.f);
                        }
                        if constexpr (!GWM || !MM) {
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            const float dv = acc2[j][g][r];
                            const f32x2_t db = splat2(dv);
                            if constexpr (!GWM) gb[(MODE == 2 && !GWM) ? g : 0] += dv;
#pragma unroll
                            for (int hp = 0; hp < H / 2; ++hp) {
#ifndef SPE_DBG_NOGW
                                if constexpr (!GWM) gWp[(MODE >= 2 && !GWM) ? g * (H / 2) + hp : 0] = fma2(db, PT[j][r][hp], gWp[(MODE >= 2 && !GWM) ? g * (H / 2) + hp : 0]);
#endif
                                if constexpr (!MM) mt[hp] = fma2(db, (f32x2_t){ww[g][2 * hp], ww[g][2 * hp + 1]}, mt[hp]);
                            }
                        }
                        }
#pragma unroll
                        for (int hp = 0; hp < H / 2; ++hp) rD2[hp] = fma2(mt[hp], PT[j][r][hp], rD2[hp]);
                    }
                }
            } else {
                // MODE 3.  phase A (Ww): dPT[j][r][hp] <- dP = Ww^T (dP'd * keepscale) of heads (2hp, 2hp+1) at key r
                f32x2_t dPT[KT][4][H / 2];
                {
                float ww[MM ? 1 : H][MM ? 1 : H];
                if constexpr (!MM) load_w<H>(a.Ww, ww);
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    if (DROP) {
#pragma unroll
                        for (int hp = 0; hp < H / 2; ++hp) {
                            float k0[4], k1[4];
                            if (have_bits) fused_keep_from_bits(kbits[(MODE >= 2 && DROP) ? j : 0], a.p_drop, hp, k0, k1);
                            else fused_keep_scales<H>(a.seed, a.offset, a.p_drop, b, hp, q, KB(j), N, k0, k1);
#pragma unroll
                            for (int r = 0; r < 4; ++r) { acc2[j][2 * hp][r] *= k0[r]; acc2[j][2 * hp + 1][r] *= k1[r]; }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        f32x2_t mt[H / 2];
                        if constexpr (MM) {
                            float x[H];
#pragma unroll
                            for (int g = 0; g < H; ++g) x[g] = acc2[j][g][r];
                            f32x4_t m4[H / 4];
                            fl_mix_16<H, false>(x, Aw, nullptr, m4);
#pragma unroll
                            for (int hp = 0; hp < H / 2; ++hp) mt[hp] = (f32x2_t){m4[hp >> 1][2 * (hp & 1)], m4[hp >> 1][2 * (hp & 1) + 1]};
                        } else {
#pragma unroll
                            for (int hp = 0; hp < H / 2; ++hp) mt[hp] = splat2(0.f);
#pragma unroll
                            for (int g = 0; g < H; ++g) {
                                const f32x2_t db = splat2(acc2[j][g][r]);
#pragma unroll
                                for (int hp = 0; hp < H / 2; ++hp) mt[hp] = fma2(db, (f32x2_t){ww[g][2 * hp], ww[g][2 * hp + 1]}, mt[hp]);
                            }
                        }
#pragma unroll
                        for (int hp = 0; hp < H / 2; ++hp) dPT[j][r][hp] = mt[hp];
                    }
                }
                }
                __builtin_amdgcn_sched_barrier(0);
                float wl[(M4 && MM) ? 1 : H][(M4 && MM) ? 1 : H];
                if constexpr (!(M4 && MM)) load_w<H>(a.Wl, wl);
                // phase B (Wl both ways): P from raw S, dS' = P (dP - D), dWl += dS' S^T, dS = Wl^T dS'
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    f32x2_t d2[4][H / 2];
                    if constexpr (M4) {
                        f32x4_t sp[4][H / 4];
                        mix_keys_f32<H>(acc[j], Al4, c0, sp);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int g = 0; g < H; ++g) d2[r][g / 2][g & 1] = EXP2(sp[r][g >> 2][g & 3]);
                    } else {
                        f32x2_t lo[H], hi[H];
                        mix_rows<H>(acc[j], wl, c0, lo, hi);
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            d2[0][g / 2][g & 1] = EXP2(lo[g][0]); d2[1][g / 2][g & 1] = EXP2(lo[g][1]);
                            d2[2][g / 2][g & 1] = EXP2(hi[g][0]); d2[3][g / 2][g & 1] = EXP2(hi[g][1]);
                        }
                    }
                    // rows q >= N: dP = 0 and rD2 = 0 (loaded as 0), so dS' = 0 there without a mask
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int gp = 0; gp < H / 2; ++gp) d2[r][gp] = d2[r][gp] * (dPT[j][r][gp] - rD2[gp]);
                    if (TMASK(j)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool kv = KVAL(j, r);
#pragma unroll
                            for (int gp = 0; gp < H / 2; ++gp) if (!kv) d2[r][gp] = splat2(0.f);
                        }
                    }
                    if constexpr (GWM && SPE_GWM_BODY) {
                        // dWl += dS' S^T over this tile's 256 positions on the matrix pipe (see backward pass 1).  dbl stays an fp32 sum
                        // on the vector pipe: its exact value is 0 (softmax is shift invariant), and the bf16 rounding of dS' in the
                        // MFMA operand would leave 2^-9-grade noise where the fp32 sum leaves 1e-7
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int gp = 0; gp < H / 2; ++gp) gb2[(MODE == 3) ? gp : 0] += d2[r][gp];
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            *reinterpret_cast<s16x4m_t*>(gw_wr + g * GWR) = pack4<false>(d2[0][g / 2][g & 1], d2[1][g / 2][g & 1], d2[2][g / 2][g & 1], d2[3][g / 2][g & 1]);
                            *reinterpret_cast<s16x4m_t*>(gw_wr + GWT + g * GWR) = pack4<false>(acc[j][g][0], acc[j][g][1], acc[j][g][2], acc[j][g][3]);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const u32x4_t xa = *reinterpret_cast<const u32x4_t*>(gw_xrd + c * 16);
                            const u32x4_t yb = *reinterpret_cast<const u32x4_t*>(gw_yrd + c * 16);
                            typedef unsigned u32x2w_t __attribute__((ext_vector_type(2)));
                            gwacc[(2 * c) & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4m_t, (u32x2w_t){xa[0], xa[1]}),
                                                                                             __builtin_bit_cast(s16x4m_t, (u32x2w_t){yb[0], yb[1]}), gwacc[(2 * c) & 3], 0, 0, 0);
                            gwacc[(2 * c + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4m_t, (u32x2w_t){xa[2], xa[3]}),
                                                                                                 __builtin_bit_cast(s16x4m_t, (u32x2w_t){yb[2], yb[3]}), gwacc[(2 * c + 1) & 3], 0, 0, 0);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    } else if constexpr (!GWM) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int gp = 0; gp < H / 2; ++gp) gb2[(MODE == 3) ? gp : 0] += d2[r][gp];
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            const f32x2_t sb = splat2(acc[j][h][r]);
#pragma unroll
#ifndef SPE_DBG_NOGW
                            for (int gp = 0; gp < H / 2; ++gp) gWp[(MODE >= 2 && !GWM) ? gp * H + h : 0] = fma2(d2[r][gp], sb, gWp[(MODE >= 2 && !GWM) ? gp * H + h : 0]);
#else
                            for (int gp = 0; gp < ((h == 0) ? H / 2 : 0); ++gp) gWp[(MODE >= 2 && !GWM) ? gp * H + h : 0] = fma2(d2[r][gp], sb, gWp[(MODE >= 2 && !GWM) ? gp * H + h : 0]);
#endif
                        }
                    }
                    }
                    f32x2_t ds[4][H / 2];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (MM) {
                            float x[H];
#pragma unroll
                            for (int g = 0; g < H; ++g) x[g] = d2[r][g / 2][g & 1];
                            f32x4_t m4[H / 4];
                            fl_mix_16<H, false>(x, Al, nullptr, m4);               // dS = Wl^T dS' of this key
#pragma unroll
                            for (int hp = 0; hp < H / 2; ++hp) ds[r][hp] = (f32x2_t){m4[hp >> 1][2 * (hp & 1)], m4[hp >> 1][2 * (hp & 1) + 1]};
                        } else {
#pragma unroll
                            for (int hp = 0; hp < H / 2; ++hp) ds[r][hp] = splat2(0.f);
#pragma unroll
                            for (int g = 0; g < H; ++g) {
                                const f32x2_t db = splat2(d2[r][g / 2][g & 1]);
#pragma unroll
                                for (int hp = 0; hp < H / 2; ++hp) ds[r][hp] = fma2(db, (f32x2_t){wl[g][2 * hp], wl[g][2 * hp + 1]}, ds[r][hp]);
                            }
                        }
                    }
#ifdef SPE_DBG_NOST3
                    if (TV(j) && a.N < 0) {
#else
                    if (TV(j)) {
#endif
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            bf16x4_t o;
                            o[0] = (__bf16)ds[0][h / 2][h & 1]; o[1] = (__bf16)ds[1][h / 2][h & 1];
                            o[2] = (__bf16)ds[2][h / 2][h & 1]; o[3] = (__bf16)ds[3][h / 2][h & 1];
#ifndef FUSED_PLAIN_STORE
                            {   // non-temporal: the 554 MB of blocks are read next by another kernel, never again by this one -
                                // kept out of L2 they neither evict the K / V fragments nor leave dirty lines for the
                                // contraction that follows to wait on (write pass + PV contraction 0.382 -> 0.347 ms)
                                typedef unsigned u32x2nt_t __attribute__((ext_vector_type(2)));
                                __builtin_nontemporal_store(__builtin_bit_cast(u32x2nt_t, o),
                                                            reinterpret_cast<u32x2nt_t*>(a.outT + (((((long)b * H + h) * nt + qt) * nt + (kt_first + j)) * 64 + lane) * 4));
                            }
#else
                            *reinterpret_cast<bf16x4_t*>(a.outT + (((((long)b * H + h) * nt + qt) * nt + (kt_first + j)) * 64 + lane) * 4) = o;
#endif
                        }
                    }
                }
            }
#undef TMASK
#undef KVAL
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
                        l = (mn > -INFINITY) ? l * EXP2(m - mn) + ol * EXP2(om - mn) : 0.f;
                        m = mn;
                    }
                    if (lane < 16) { sred[((wave * H + g) * 16 + lane) * 2] = m; sred[((wave * H + g) * 16 + lane) * 2 + 1] = l; }
                } else {
                    float d = rD2[g / 2][g & 1];
                    d += __shfl_xor(d, 16, 64);
                    d += __shfl_xor(d, 32, 64);
                    if (lane < 16) sred[((wave * H + g) * 16 + lane) * 2] = d;
                }
            }
            __syncthreads();
            const int first_j = (int)(((long)bqp * klen) / a.steps_per_wg);
            const int slot = chunk * (FUSED_MAXSLOT / nch) + (wg_j - first_j);
            for (int i2 = threadIdx.x; i2 < QP * H * 16; i2 += 256) {
                const int u = i2 / (H * 16), i = i2 % (H * 16);
                if (qp * QP + u >= nt) continue;
                const long bq_u = (long)b * nt + qp * QP + u;
                float* dst = a.ws_stats + (((bq_u * FUSED_MAXSLOT + slot) * H * 16) + i) * 2;
                if (MODE == 0) {
                    float mn = -INFINITY;
                    for (int w = u * WPQ; w < (u + 1) * WPQ; ++w) mn = fmaxf(mn, sred[((w * H * 16) + i) * 2]);
                    float l = 0.f;
                    if (mn > -INFINITY)
                        for (int w = u * WPQ; w < (u + 1) * WPQ; ++w) l += sred[((w * H * 16) + i) * 2 + 1] * EXP2(sred[((w * H * 16) + i) * 2] - mn);
                    dst[0] = mn; dst[1] = l;
                } else {
                    float d = 0.f;
                    for (int w = u * WPQ; w < (u + 1) * WPQ; ++w) d += sred[((w * H * 16) + i) * 2];
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
        if constexpr (GWM) {
            // D[m = g][n]: lane holds rows 4*(lane>>4) + r of column lane & 15; columns < H = dW[g][h], column H = the bias gradient
            const f32x4_t dsum = (gwacc[0] + gwacc[1]) + (gwacc[2] + gwacc[3]);
            const int nn = lane & 15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = 4 * (lane >> 4) + r;
                // mode 3 accumulated dS' . (log2(e) S)^T: the weight gradient carries ln 2, the bias gradient does not
                if (g < H && nn < H) part[wave * (H * H + H) + g * H + nn] = (MODE == 3) ? SPE_LN2 * dsum[r] : dsum[r];
                if (MODE == 2 && g < H && nn == H) part[wave * (H * H + H) + H * H + g] = dsum[r];
            }
            if constexpr (MODE == 3) {
#pragma unroll
                for (int g = 0; g < H; ++g) {
                    const float v = spe_wave_sum(gb2[(MODE == 3) ? g / 2 : 0][g & 1]);
                    if (lane == 0) part[wave * (H * H + H) + H * H + g] = v;
                }
            }
        } else {
#pragma unroll
        for (int g = 0; g < H; ++g) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                // mode 3 accumulated dS' . (log2(e) S)^T
                const float v = (MODE == 2) ? spe_wave_sum(gWp[GWM ? 0 : g * (H / 2) + h / 2][h & 1]) : SPE_LN2 * spe_wave_sum(gWp[GWM ? 0 : (g / 2) * H + h][g & 1]);
                if (lane == 0) part[wave * (H * H + H) + g * H + h] = v;
            }
            const float v = spe_wave_sum((MODE == 2) ? gb[(MODE == 2 && !GWM) ? g : 0] : gb2[(MODE == 3) ? g / 2 : 0][g & 1]);
            if (lane == 0) part[wave * (H * H + H) + H * H + g] = v;
        }
        }
        __syncthreads();
        // layout of a ws_w row: [dWl | dbl | dWw | dbw]; mode 2 fills the second half, mode 3 the first
        const int off = (MODE == 2) ? (H * H + H) : 0;
        for (int i = threadIdx.x; i < H * H + H; i += 256)
            a.ws_w[(long)blockIdx.x * NW + off + i] = part[i] + part[(H * H + H) + i] + part[2 * (H * H + H) + i] + part[3 * (H * H + H) + i];
    }
}

// Merge the per-slot partial statistics of each (b, q-tile): mode 0 -> M = max, IL = 1/sum; mode 2 -> D = sum.
// rows (mode 0, optional): [B][Np][H] row constants of the flash kernels, bl[g] * log2(e) - max - log2(sum) (the addend that turns
// Wl S into log2 P), zero for the rows N .. Np-1 - written here instead of by a launch of its own (spe_talking_flash_rows)
__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ ws, float* __restrict__ out0, float* __restrict__ out1,
                                                         int B, int H, int N, int nt, int steps_per_wg, int mode,
                                                         const float* __restrict__ bl, float* __restrict__ rows, int Np) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;          // over B*ntr*H*16, ntr = nt (or Np / 16 with rows)
    const int ntr = rows ? Np / 16 : nt;
    if (i >= (long)B * ntr * H * 16) return;
    const int ql = (int)(i & 15); const int g = (int)((i >> 4) % H);
    const int b = (int)(i / (16L * H * ntr)), qt = (int)((i / (16L * H)) % ntr), q = qt * 16 + ql;
    const int bq = b * nt + qt;
    if (q >= N) { if (rows) rows[((long)b * Np + q) * H + g] = 0.f; return; }
    const float* base = ws + (((long)bq * FUSED_MAXSLOT) * H * 16 + (long)g * 16 + ql) * 2;
    const long stride = (long)H * 16 * 2;
    const long o = ((long)b * H + g) * N + q;
    const int nch = fused_nch(nt), spc = FUSED_MAXSLOT / nch;
    // the slots of chunk c that received a partial: workgroups first_j..last_j of the chunk touch this q-tile
    float mn = -INFINITY, l = 0.f, d = 0.f;
    for (int pass = 0; pass < (mode == 0 ? 2 : 1); ++pass)
        for (int c = 0; c < nch; ++c) {
            const int klen = fused_kbeg(c + 1, nt, nch) - fused_kbeg(c, nt, nch);
            if (klen <= 0) continue;
            const long bqp = (long)b * fused_npair(nt, mode) + qt / fused_qp(mode);   // the workgroups walk (q-group, key tile) steps
            const int first_j = (int)((bqp * klen) / steps_per_wg), last_j = (int)(((bqp + 1) * klen - 1) / steps_per_wg);
            for (int s = 0; s <= last_j - first_j; ++s) {
                const float* e = base + (c * spc + s) * stride;
                if (mode != 0) d += e[0];
                else if (pass == 0) mn = fmaxf(mn, e[0]);
                else l += e[1] * EXP2(e[0] - mn);
            }
        }
    if (mode == 0) {
        const float il = 1.f / l;
        out0[o] = mn; out1[o] = il;
        if (rows) rows[((long)b * Np + q) * H + g] = bl[g] * 1.4426950408889634f - mn + __builtin_amdgcn_logf(il);
    }
    else out0[o] = d;
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

extern "C" int spe_attn_merge(const float* ws, float* out0, float* out1, int B, int H, int N, int steps_per_wg, int mode,
                              hipStream_t st) {
    const int nt = (N + 15) / 16;
    const long n = (long)B * nt * H * 16;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws, out0, out1, B, H, N, nt,
                       steps_per_wg, mode, nullptr, nullptr, 0);
    SPE_CHECK_LAUNCH();
    return 0;
}

// C-ABI: see include/spe_hip.h.  Mode-0 merge that also writes the flash kernels' row constants [B][Np][H] (Np a multiple of 16, >= N).
extern "C" int spe_attn_merge_rows(const float* ws, float* M, float* IL, const float* bl, float* rows, int Np, int B, int H, int N,
                                   int steps_per_wg, hipStream_t st) {
    const int nt = (N + 15) / 16;
    if ((long)B * nt * H <= 0) return 0;
    if (!rows || !bl || Np < nt * 16 || (Np & 15)) return -2;
    const long n = (long)B * (Np / 16) * H * 16;
    hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws, M, IL, B, H, N, nt, steps_per_wg, 0, bl, rows, Np);
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
        if (col < hh) dWl[col] = s;
        else if (col < hh + H) dbl[col - hh] = s;
        else if (col < 2 * hh + H) dWw[col - hh - H] = s;
        else dbw[col - 2 * hh - H] = s;
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
static void make_plan(int B, int nt, int nwg, int mode, int* spw_out, int* nwg_out) {
    const int nch = fused_nch(nt), spc = FUSED_MAXSLOT / nch;
    int nwg8 = nwg & ~7; if (nwg8 < 8) nwg8 = 8;
    const int wpc = nwg8 / nch;
    const int len_max = (nt + nch - 1) / nch;
    long spw = ((long)B * fused_npair(nt, mode) * len_max + wpc - 1) / wpc;
    const long min_spw = (len_max + (spc - 1) - 1) / (spc - 1);       // ceil(len / (spc-1)): <= spc slots
    if (spw < min_spw) spw = min_spw;
    *spw_out = (int)spw; *nwg_out = nwg8;
}

template <int H, int DSTEPS, bool TAIL16, int MODE, bool DROP, int KT>
static int launch_fused(const FusedArgs& a, int nwg, hipStream_t st) {
    constexpr int NFR = H * DSTEPS;
    constexpr int smem = ((MODE <= 1) ? SPE_FUSED_QP : 1) * NFR * 64 * 16 * ((MODE >= 2) ? 2 : 1) + 4 * (H * 16 * 2 > (H * H + H) ? H * 16 * 2 : (H * H + H)) * 4
                         + (((MODE == 2 || (MODE == 3 && SPE_FUSED_GWMFMA3)) && SPE_FUSED_GWMFMA && H % 4 == 0) ? 512 + 4 * 2 * 4 * H * 144 : 0);      // GWM: constants + the 4 waves' transpose tiles
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&talking_fused_kernel<H, DSTEPS, TAIL16, MODE, DROP, KT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((talking_fused_kernel<H, DSTEPS, TAIL16, MODE, DROP, KT>), dim3(nwg), dim3(256), smem, st, a);
    SPE_CHECK_LAUNCH();
    return 0;
}

// macro-step sizes (measured at cfg2): 4 tiles for the statistics pass (0.26 -> 0.235 ms), 1 tile for the write
// pass (4 tiles: 0.43 -> 0.48 ms, occupancy loss) and for the backward passes (H*H weight-gradient accumulators +
// the second accumulator set)
#ifndef SPE_FUSED_KTF
#define SPE_FUSED_KTF 1
#endif
#ifndef SPE_FUSED_KTB
#define SPE_FUSED_KTB 1
#endif
template <int H, int DSTEPS, bool TAIL16>
static int dispatch_mode(const FusedArgs& a, int mode, bool drop, int nwg, hipStream_t st) {
    constexpr int KF = SPE_FUSED_KTF, KB_ = SPE_FUSED_KTB;
    switch (mode) {
        case 0: return launch_fused<H, DSTEPS, TAIL16, 0, false, KF>(a, nwg, st);
        case 1: return drop ? launch_fused<H, DSTEPS, TAIL16, 1, true, KB_>(a, nwg, st) : launch_fused<H, DSTEPS, TAIL16, 1, false, KB_>(a, nwg, st);
        case 2: return drop ? launch_fused<H, DSTEPS, TAIL16, 2, true, KB_>(a, nwg, st) : launch_fused<H, DSTEPS, TAIL16, 2, false, KB_>(a, nwg, st);
        case 3: return drop ? launch_fused<H, DSTEPS, TAIL16, 3, true, KB_>(a, nwg, st) : launch_fused<H, DSTEPS, TAIL16, 3, false, KB_>(a, nwg, st);
    }
    return -2;
}

// C-ABI: see include/spe_hip.h (spe_talking_fused).  Returns -2 for unsupported (H, head dim).
extern "C" int spe_talking_fused_bits(int mode, const void* Qf, const void* Kf, const void* Vf, const void* dOf,
                                      const float* Wl, const float* bl, const float* Ww, const float* bw,
                                      const float* M, const float* IL, const float* D, float* ws_stats, float* ws_w, void* outT, const void* keepbits,
                                      int B, int H, int N, int dh, int nwg, float p_drop, uint64_t seed, uint64_t offset, hipStream_t st);
extern "C" int spe_talking_fused(int mode, const void* Qf, const void* Kf, const void* Vf, const void* dOf,
                                 const float* Wl, const float* bl, const float* Ww, const float* bw,
                                 const float* M, const float* IL, const float* D, float* ws_stats, float* ws_w, void* outT,
                                 int B, int H, int N, int dh, int nwg, float p_drop, uint64_t seed, uint64_t offset,
                                 hipStream_t st) {
    return spe_talking_fused_bits(mode, Qf, Kf, Vf, dOf, Wl, bl, Ww, bw, M, IL, D, ws_stats, ws_w, outT, nullptr, B, H, N, dh, nwg, p_drop, seed, offset, st);
}
// C-ABI: see include/spe_hip.h
extern "C" int spe_talking_fused_bits(int mode, const void* Qf, const void* Kf, const void* Vf, const void* dOf,
                                      const float* Wl, const float* bl, const float* Ww, const float* bw,
                                      const float* M, const float* IL, const float* D, float* ws_stats, float* ws_w, void* outT, const void* keepbits,
                                      int B, int H, int N, int dh, int nwg, float p_drop, uint64_t seed, uint64_t offset, hipStream_t st) {
    FusedArgs a;
    a.keepbits = (mode >= 2 && p_drop > 0.f) ? reinterpret_cast<const unsigned*>(keepbits) : nullptr;
    a.Qf = (const u32x4_t*)Qf; a.Kf = (const u32x4_t*)Kf; a.Vf = (const u32x4_t*)Vf; a.dOf = (const u32x4_t*)dOf;
    a.Wl = Wl; a.bl = bl; a.Ww = Ww; a.bw = bw; a.M = M; a.IL = IL; a.D = D;
    a.ws_stats = ws_stats; a.ws_w = ws_w; a.outT = (unsigned short*)outT;
    a.B = B; a.N = N; a.nt = (N + 15) / 16;
    a.total_steps = (long)B * a.nt * a.nt;
    if (a.total_steps <= 0) return 0;
    make_plan(B, a.nt, nwg, mode, &a.steps_per_wg, &nwg);
    a.p_drop = p_drop; a.seed = seed; a.offset = offset;
    const bool drop = p_drop > 0.f;
    // head dim -> d-steps: full 32-wide steps, plus a 16-wide tail step when the remainder is 1..16
    const int rem = dh % 32, full = dh / 32 + (rem > 16 ? 1 : 0), tail = (rem > 0 && rem <= 16) ? 1 : 0;
    const int ds = full + tail;
    if (dh < 1 || dh > 64) return -2;
#define SPE_FUSED_DISPATCH(HH)                                                                       \
    if (H == HH && ds == 2 && tail) return dispatch_mode<HH, 2, true>(a, mode, drop, nwg, st);       \
    if (H == HH && ds == 2 && !tail) return dispatch_mode<HH, 2, false>(a, mode, drop, nwg, st);     \
    if (H == HH && ds == 1 && tail) return dispatch_mode<HH, 1, true>(a, mode, drop, nwg, st);       \
    if (H == HH && ds == 1 && !tail) return dispatch_mode<HH, 1, false>(a, mode, drop, nwg, st);
    SPE_FUSED_DISPATCH(8)
    SPE_FUSED_DISPATCH(4)
#undef SPE_FUSED_DISPATCH
    return -2;
}

// steps_per_wg the launcher will use for (B, N, nwg): callers size the workspaces with it.
extern "C" int spe_talking_fused_plan(int B, int N, int nwg, int mode, int* steps_per_wg, int* nwg_used) {
    const int nt = (N + 15) / 16;
    if ((long)B * nt * nt <= 0) { *steps_per_wg = 0; *nwg_used = 0; return 0; }
    make_plan(B, nt, nwg, mode, steps_per_wg, nwg_used);
    return 0;
}
