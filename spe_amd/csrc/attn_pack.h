// Shared by attn_fused.hip (spe_attn_pack) and attn_contract.hip (spe_attn_pack_multi): the bf16 fragment record
// layout of the fused talking-heads score kernels (see frag_load in attn_fused.hip).
#pragma once
#include "common.h"

// IT: index type of the unit counter - unsigned when the tensor has < 2^31 units (every realistic shape: 32-bit divisions
// cost a fraction of the 64-bit ones, and this kernel is otherwise instruction-bound on its index arithmetic).
template <typename IT>
__device__ __forceinline__ void attn_pack_unit_t(const float* __restrict__ x, long sb, long sn, long sh, int N, int H, int dh, int nt,
                                                 float scale, IT i, uint2* __restrict__ out, int notail = 0) {
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
    typedef __bf16 bf16x4p_t __attribute__((ext_vector_type(4)));
    bf16x4p_t o;
    if (d0 + 3 < dh && ((reinterpret_cast<uintptr_t>(src + d0) & 15) == 0)) {
        const float4 f = *reinterpret_cast<const float4*>(src + d0);
        const bool rv = row < N;
        o[0] = (__bf16)(rv ? f.x * scale : 0.f); o[1] = (__bf16)(rv ? f.y * scale : 0.f);
        o[2] = (__bf16)(rv ? f.z * scale : 0.f); o[3] = (__bf16)(rv ? f.w * scale : 0.f);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float f = src[min(d0 + j, dh - 1)]; o[j] = (__bf16)((row < N && d0 + j < dh) ? f * scale : 0.f); }
    }
    out[i] = __builtin_bit_cast(uint2, o);
}
__device__ __forceinline__ void attn_pack_unit(const float* __restrict__ x, long sb, long sn, long sh, int N, int H, int dh, int nt,
                                               float scale, long i, uint2* __restrict__ out, int notail = 0) {
    attn_pack_unit_t<long>(x, sb, sn, sh, N, H, dh, nt, scale, i, out, notail);
}

// 8-B units of a packed tensor [B, H, nt] records
__host__ __device__ static inline long attn_pack_units(int B, int N, int H, int dh, int notail = 0) {
    const int nt = (N + 15) / 16, rem = dh % 32, full = notail ? (dh + 31) / 32 : dh / 32 + (rem > 16 ? 1 : 0), tail = (!notail && rem > 0 && rem <= 16) ? 1 : 0;
    return (long)B * H * nt * (full * 128 + tail * 64);
}

