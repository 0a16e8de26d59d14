// Shared by attn_fused.hip (spe_attn_pack) and attn_contract.hip (spe_attn_pack_multi): the bf16 fragment record
// layout of the fused talking-heads score kernels (see frag_load in attn_fused.hip).
#pragma once
#include "common.h"

__device__ __forceinline__ void attn_pack_unit(const float* __restrict__ x, long sb, long sn, long sh, int N, int H, int dh, int nt,
                                               float scale, long i, uint2* __restrict__ out, int notail = 0) {
    // notail: ceil(dh/32) full steps and no 16-wide tail step (the layout of mha_flash.hip)
    const int rem = dh % 32, full = notail ? (dh + 31) / 32 : dh / 32 + (rem > 16 ? 1 : 0), tail = (!notail && rem > 0 && rem <= 16) ? 1 : 0;
    const int rec8 = full * 128 + tail * 64;
    const long rec = i / rec8; const int u = (int)(i % rec8);
    const int tile = (int)(rec % nt); const int h = (int)((rec / nt) % H); const int b = (int)(rec / ((long)nt * H));
    int ln, d0;
    if (u < full * 128) { const int st = u >> 7, w = u & 127; ln = w >> 1; d0 = st * 32 + (ln >> 4) * 8 + (w & 1) * 4; }
    else { ln = u - full * 128; d0 = full * 32 + (ln >> 4) * 4; }
    const int row = tile * 16 + (ln & 15);
    const float* src = x + b * sb + (long)min(row, N - 1) * sn + h * sh;
    typedef __bf16 bf16x4p_t __attribute__((ext_vector_type(4)));
    bf16x4p_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float f = src[min(d0 + j, dh - 1)]; o[j] = (__bf16)((row < N && d0 + j < dh) ? f * scale : 0.f); }
    out[i] = __builtin_bit_cast(uint2, o);
}

// 8-B units of a packed tensor [B, H, nt] records
__host__ __device__ static inline long attn_pack_units(int B, int N, int H, int dh, int notail = 0) {
    const int nt = (N + 15) / 16, rem = dh % 32, full = notail ? (dh + 31) / 32 : dh / 32 + (rem > 16 ? 1 : 0), tail = (!notail && rem > 0 && rem <= 16) ? 1 : 0;
    return (long)B * H * nt * (full * 128 + tail * 64);
}

