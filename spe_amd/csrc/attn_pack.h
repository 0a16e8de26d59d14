// Shared by attn_fused.hip (spe_attn_pack) and attn_contract.hip (spe_attn_pack_multi): the 16-bit fragment record
// layout of the fused talking-heads score kernels (see frag_load in attn_fused.hip).
//
// Element format of a packed tensor: bf16, or IEEE fp16 (`f16` = 1).  The forward operands of the attention - q * scale *
// log2(e), k, v and the probabilities - are O(1) quantities, far inside fp16's range, and fp16 carries 3 more mantissa bits
// than bf16 at the same size and the same MFMA rate: the forward error of the fused attention drops 8x (measured on the
// reference fixtures: tools/error_budget.py).  Gradients (dO, dS) keep bf16 - their range is not bounded.
// fp16 conversions saturate at +-65504 instead of producing infinities.
#pragma once
#include "common.h"

__device__ __forceinline__ unsigned short spe_f2h_sat(float f) {
    // v_med3_f32 keeps a NaN (fminf / fmaxf would turn it into +-65504 and hide a diverged run); v_cvt_f16_f32: round to nearest even
    const _Float16 h = (_Float16)((f != f) ? f : __builtin_amdgcn_fmed3f(f, -65504.f, 65504.f));
    return __builtin_bit_cast(unsigned short, h);
}
// 4 floats -> 4 x 16 bit (bf16 RNE or saturating fp16 RNE) as one 8-B unit
__device__ __forceinline__ uint2 spe_cvt4_16(float a, float b, float c, float d, int f16) {
    if (f16) {
        const unsigned x = (unsigned)spe_f2h_sat(a) | ((unsigned)spe_f2h_sat(b) << 16);
        const unsigned y = (unsigned)spe_f2h_sat(c) | ((unsigned)spe_f2h_sat(d) << 16);
        return make_uint2(x, y);
    }
    typedef __bf16 bf16x4q_t __attribute__((ext_vector_type(4)));
    bf16x4q_t o;
    o[0] = (__bf16)a; o[1] = (__bf16)b; o[2] = (__bf16)c; o[3] = (__bf16)d;
    return __builtin_bit_cast(uint2, o);
}

// IT: index type of the unit counter - unsigned when the tensor has < 2^31 units (every realistic shape: 32-bit divisions
// cost a fraction of the 64-bit ones, and this kernel is otherwise instruction-bound on its index arithmetic).
template <typename IT>
__device__ __forceinline__ void attn_pack_unit_t(const float* __restrict__ x, long sb, long sn, long sh, int N, int H, int dh, int nt,
                                                 float scale, IT i, uint2* __restrict__ out, int notail = 0, int f16 = 0) {
    // notail: ceil(dh/32) full steps and no 16-wide tail step (the layout of mha_flash.hip)
    const int rem = dh % 32, full = notail ? (dh + 31) / 32 : dh / 32 + (rem > 16 ? 1 : 0), tail = (!notail && rem > 0 && rem <= 16) ? 1 : 0;
    const IT rec8 = (IT)(full * 128 + tail * 64);
    const IT rec = i / rec8; const int u = (int)(i - rec * rec8);
    const IT bh = rec / (IT)nt; const int tile = (int)(rec - bh * (IT)nt);
    const int b = (int)(bh / (IT)H), h = (int)(bh - (IT)b * (IT)H);
    int ln, d0;
    if (u < full * 128) { const int st = u >> 7, w = u & 127; ln = w >> 1; d0 = st * 32 + (ln >> 4) * 8 + (w & 1) * 4; }
    else { ln = u - full * 128; d0 = full * 32 + (ln >> 4) * 4; }
    const int row = tile * 16 + (ln & 15);
    const float* src = x + b * sb + (long)min(row, N - 1) * sn + h * sh;
    float o[4];
    if (d0 + 3 < dh && ((reinterpret_cast<uintptr_t>(src + d0) & 15) == 0)) {
        const float4 f = *reinterpret_cast<const float4*>(src + d0);
        const bool rv = row < N;
        o[0] = rv ? f.x * scale : 0.f; o[1] = rv ? f.y * scale : 0.f;
        o[2] = rv ? f.z * scale : 0.f; o[3] = rv ? f.w * scale : 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float f = src[min(d0 + j, dh - 1)]; o[j] = (row < N && d0 + j < dh) ? f * scale : 0.f; }
    }
    out[i] = spe_cvt4_16(o[0], o[1], o[2], o[3], f16);
}
__device__ __forceinline__ void attn_pack_unit(const float* __restrict__ x, long sb, long sn, long sh, int N, int H, int dh, int nt,
                                               float scale, long i, uint2* __restrict__ out, int notail = 0, int f16 = 0) {
    attn_pack_unit_t<long>(x, sb, sn, sh, N, H, dh, nt, scale, i, out, notail, f16);
}

// 8-B units of a packed tensor [B, H, nt] records
__host__ __device__ static inline long attn_pack_units(int B, int N, int H, int dh, int notail = 0) {
    const int nt = (N + 15) / 16, rem = dh % 32, full = notail ? (dh + 31) / 32 : dh / 32 + (rem > 16 ? 1 : 0), tail = (!notail && rem > 0 && rem <= 16) ? 1 : 0;
    return (long)B * H * nt * (full * 128 + tail * 64);
}

