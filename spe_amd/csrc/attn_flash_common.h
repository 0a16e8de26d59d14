// Shared pieces of the flash-style talking-heads kernels (attn_flash.hip): operand formats, the head mixes on the matrix pipe,
// the LDS-DMA stage copy and the flattened work split.  Reference: models/cait.py:377-389 (Attention_talking_head.forward).
//
// Orientation-free facts the kernels rely on (measured on gfx950, tools/micro/mfma_rates.hip):
//   * v_mfma_f32_16x16x16_{f16,bf16} issues in 18 cycles per SIMD - the same as v_mfma_f32_16x16x32 with twice the work;
//   * v_mfma_f32_4x4x4_16b_f16 / v_mfma_f32_4x4x1_16b_f32 issue in 11 cycles;
//   * global_load_lds_dwordx4 reaches every byte of the 160 KB LDS through M0.
// A 16 x 16 score tile of all H heads lives in the MFMA C layout: lane l owns column (l & 15) and rows 4 * (l >> 4) + r of the tile
// for every head, so the H x H head mixes are lane-local and run as 4-lane-block MFMAs:
//   4x4x1 (f32):  register i of lane 4b + j accumulates A(lane 4b + i) * B(lane 4b + j)
//   4x4x4 (f16 / bf16): register i of lane 4b + j accumulates sum_k A(lane 4b + i)[k] * B(lane 4b + j)[k]
// With A := W[4gh + (lane & 3)][..] (a per-lane constant) and B := the lane's own head values, register i of the result is output
// head 4gh + i of the lane's own element - no data moves between lanes.
#pragma once
#include "common.h"

typedef unsigned int flu32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int flu32x2_t __attribute__((ext_vector_type(2)));
typedef short fls16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 flf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 flf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 flbf16x4_t __attribute__((ext_vector_type(4)));

#define FL_PD_SCALE 256.0f           // P' travels as fp16(P' * 2^8): probabilities of 1e-4 .. 1e-7 stay normal numbers
#define FL_LOG2E 1.4426950408889634f
#define FL_LN2 0.6931471805599453f
#define FL_MAXSLOT 8                 // partial results of one major tile group come from at most this many workgroups

__device__ __forceinline__ float fl_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// saturating fp16 conversion; a NaN stays a NaN (a diverged run must stay visible)
__device__ __forceinline__ _Float16 fl_f2h_sat(float f) { return (_Float16)((f != f) ? f : __builtin_amdgcn_fmed3f(f, -65504.f, 65504.f)); }

// 4 floats -> one 8-B MFMA operand: saturating fp16 (forward quantities, O(1)) or bf16 (anything that carries a gradient)
template <bool F16>
__device__ __forceinline__ fls16x4_t fl_pack4(float a, float b, float c, float d) {
    if constexpr (F16) {
        flf16x4_t v;
        v[0] = fl_f2h_sat(a); v[1] = fl_f2h_sat(b); v[2] = fl_f2h_sat(c); v[3] = fl_f2h_sat(d);
        return __builtin_bit_cast(fls16x4_t, v);
    } else {
        flbf16x4_t v; v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
        return __builtin_bit_cast(fls16x4_t, v);
    }
}

__device__ __forceinline__ fls16x4_t fl_pack4_f16(float a, float b, float c, float d) {      // values known to be inside fp16's range
    flf16x4_t v; v[0] = (_Float16)a; v[1] = (_Float16)b; v[2] = (_Float16)c; v[3] = (_Float16)d;
    return __builtin_bit_cast(fls16x4_t, v);
}

// ---- S' = Wl S + bl in fp32 (feeds exp2): A operand of v_mfma_f32_4x4x1_16b_f32, A[gh][h] = W[4gh + (lane & 3)][h]
template <int H, bool TRANSPOSE>
__device__ __forceinline__ void fl_mixA_f32(const float* __restrict__ W, int lane, float (&A)[H / 4][H]) {
#pragma unroll
    for (int gh = 0; gh < H / 4; ++gh)
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int go = 4 * gh + (lane & 3);
            A[gh][h] = TRANSPOSE ? W[h * H + go] : W[go * H + h];
        }
}
// out[r][gh][i] = c[r][gh][i] + sum_h W[4gh + i][h] s[h][r]  (r: the lane's 4 tile rows; c: the addend, e.g. bl - m + log2(1/l))
template <int H>
__device__ __forceinline__ void fl_mix_f32(const f32x4_t (&s)[H], const float (&A)[H / 4][H], const f32x4_t (&c)[4][H / 4], f32x4_t (&out)[4][H / 4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int gh = 0; gh < H / 4; ++gh) {
            f32x4_t d = c[r][gh];
#pragma unroll
            for (int h = 0; h < H; ++h) d = __builtin_amdgcn_mfma_f32_4x4x1f32(A[gh][h], s[h][r], d, 0, 0, 0);
            out[r][gh] = d;
        }
}

// ---- 16-bit head mixes: A operand of v_mfma_f32_4x4x4_16b_{f16,bf16}, A[gh][hh] = 4 x W[4gh + (lane & 3)][4hh + k]
template <int H, bool TRANSPOSE, bool F16>
__device__ __forceinline__ void fl_mixA_16(const float* __restrict__ W, int lane, float wscale, fls16x4_t (&A)[H / 4][H / 4]) {
#pragma unroll
    for (int gh = 0; gh < H / 4; ++gh)
#pragma unroll
        for (int hh = 0; hh < H / 4; ++hh) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int go = 4 * gh + (lane & 3), hi = 4 * hh + k;
                v[k] = wscale * (TRANSPOSE ? W[hi * H + go] : W[go * H + hi]);
            }
            A[gh][hh] = fl_pack4<F16>(v[0], v[1], v[2], v[3]);
        }
}
// out[gh][i] = init[gh][i] + sum_h W[4gh + i][h] x[h] for ONE tile row of the lane (x: the H heads' values there)
template <int H, bool F16>
__device__ __forceinline__ void fl_mix_16(const float (&x)[H], const fls16x4_t (&A)[H / 4][H / 4], const f32x4_t* init, f32x4_t (&out)[H / 4]) {
    fls16x4_t bv[H / 4];
#pragma unroll
    for (int hh = 0; hh < H / 4; ++hh) bv[hh] = fl_pack4<F16>(x[4 * hh], x[4 * hh + 1], x[4 * hh + 2], x[4 * hh + 3]);
#pragma unroll
    for (int gh = 0; gh < H / 4; ++gh) {
        f32x4_t d = init ? init[gh] : (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int hh = 0; hh < H / 4; ++hh) {
            if constexpr (F16) d = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(flf16x4_t, A[gh][hh]), __builtin_bit_cast(flf16x4_t, bv[hh]), d, 0, 0, 0);
            else d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(A[gh][hh], bv[hh], d, 0, 0, 0);
        }
        out[gh] = d;
    }
}

// ---- one 1-KB piece global -> LDS (64 lanes x 16 B, LDS destination lane-linear from the wave-uniform byte address lds_dst)
__device__ __forceinline__ void fl_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// the same with the source as (wave-uniform base) + (per-lane 32-bit byte offset): the saddr form, no 64-bit vector arithmetic
__device__ __forceinline__ void fl_glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// CNT pieces of one tile image, `stride` bytes apart in LDS (a wave's share of a tile: pieces wave, wave + NW, ...): three instructions per piece (m0, the
// wait state behind its write, the load) instead of seven - the issuing wave is the one wave of its SIMD, every scalar instruction of the admission is
// ~4.5 cycles nothing hides (profiles/r06_bwdq_stamps.txt).  m0 is not restored: nothing hipcc generates in these kernels reads it (no LDS instruction of this
// target needs it), every LDS-DMA sets it itself.
template <int CNT, int STRIDE>
__device__ __forceinline__ void fl_glds16_run(const void* sbase, const unsigned* voff, unsigned lds_dst) {
    static_assert(CNT >= 1 && CNT <= 4, "pieces per wave and tile");
    if constexpr (CNT == 1)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0"
                     :: "s"(sbase), "s"(lds_dst), "v"(voff[0]) : "memory");
    else if constexpr (CNT == 2)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0\n\ts_add_u32 m0, m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %0"
                     :: "s"(sbase), "s"(lds_dst), "v"(voff[0]), "v"(voff[1]), "n"(STRIDE) : "memory", "scc");
    else if constexpr (CNT == 3)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0\n\ts_add_u32 m0, m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %0\n\t"
                     "s_add_u32 m0, m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %0"
                     :: "s"(sbase), "s"(lds_dst), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "n"(STRIDE) : "memory", "scc");
    else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0\n\ts_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %0\n\t"
                     "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %0\n\ts_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %0"
                     :: "s"(sbase), "s"(lds_dst), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "n"(STRIDE) : "memory", "scc");
}

// ---- fragment records (spe_attn_pack_multi): per (b, h, 16-row tile) FULL steps of 64 lanes x 16 B (32 head dims each) and, when
// TAIL16, one step of 64 lanes x 8 B (16 head dims); the 16-wide "X16" records are DT = 2 FULL + TAIL16 steps of 64 x 8 B.  Both are
// DT * 512 bytes.  Operands read back from an LDS image of a record:
template <int DSTEPS, bool TAIL16>
__device__ __forceinline__ flu32x4_t fl_frag_lds(const unsigned char* rec, int st, int lane) {
    constexpr int FULL = DSTEPS - (TAIL16 ? 1 : 0);
    if (TAIL16 && st == FULL) {
        const flu32x2_t v = *reinterpret_cast<const flu32x2_t*>(rec + FULL * 1024 + lane * 8);
        return (flu32x4_t){v[0], v[1], 0u, 0u};          // zero-extended: the tail step goes through the same 16x16x32 instruction
    }
    return *reinterpret_cast<const flu32x4_t*>(rec + st * 1024 + lane * 16);
}
template <bool F16>
__device__ __forceinline__ f32x4_t fl_mfma32(flu32x4_t a, flu32x4_t b, f32x4_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(flf16x8_t, a), __builtin_bit_cast(flf16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x4_t fl_mfma16(fls16x4_t a, fls16x4_t b, f32x4_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(flf16x4_t, a), __builtin_bit_cast(flf16x4_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// ---- flattened work split.  A "major" is the tile group a workgroup keeps resident (4 waves x QS q-tiles in the forward kernel,
// 4 key tiles in the backward one), a step is one streamed tile against it.  Steps are numbered (batch, major, streamed tile) with
// the streamed tile fastest and cut into equal ranges, one per workgroup; a range covers parts of 1-3 majors ("segments") and the
// partial result of a segment goes to slot (workgroup - first workgroup of the major) of the major's workspace row.
struct FlashPlan { int nmaj, nstream, spw, nwg; long total; };
__host__ __device__ static inline FlashPlan fl_plan(int B, int nt_major, int per_major, int nstream, int nwg_max) {
    FlashPlan p;
    p.nmaj = (nt_major + per_major - 1) / per_major;
    p.nstream = nstream;
    p.total = (long)B * p.nmaj * nstream;
    long spw = (p.total + nwg_max - 1) / nwg_max;
    const long min_spw = (nstream + (FL_MAXSLOT - 2)) / (FL_MAXSLOT - 1);       // a major spreads over <= FL_MAXSLOT workgroups
    if (spw < min_spw) spw = min_spw;
    if (spw < 1) spw = 1;
    p.spw = (int)spw;
    p.nwg = (int)((p.total + spw - 1) / spw);
    return p;
}

// ---- dropout keep-scales shared with attn_fused.hip (same counter layout: (b, head pair, query, 4-key group), eight 16-bit lots)
template <int H>
__device__ __forceinline__ void fl_keep_lots(uint64_t seed, uint64_t offset, int b, int hp, int q, int key0, int N, uint32_t (&o)[4]) {
    const uint64_t ctr = (((uint64_t)b * (H / 2) + hp) * (uint64_t)N + (uint64_t)q) * (uint64_t)((N + 3) >> 2) + (uint64_t)(key0 >> 2);
    spe_philox4((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
}
// lot `i` (0..3: the 4 keys of the group) of head 0 / head 1 of the pair
__device__ __forceinline__ uint32_t fl_lot(const uint32_t (&o)[4], int head, int i) {
    const uint32_t w = o[2 * head + (i >> 1)];
    return (i & 1) ? (w >> 16) : (w & 0xffffu);
}

// value of lane (l & ~3) + t of every quad (quad-permute DPP, t = 0..3)
__device__ __forceinline__ uint32_t fl_quad_bcast(uint32_t v, int t) {
    switch (t) {
        case 0: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xf, 0xf, false);
        case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xf, 0xf, false);
        case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xaa, 0xf, 0xf, false);
        default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xff, 0xf, 0xf, false);
    }
}
