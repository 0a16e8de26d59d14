// Epilogues of the bf16-operand NT GEMMs (gemm_bf16.hip: register-pipelined / ring kernels; gemm_nt2.hip: 128-wide LDS-DMA
// kernels), shared so that every kernel family writes bit-identical results for the same accumulators.
// Ownership (MFMAs issued as (B-frag, A-frag)): acc[i][j][r] = C[m0 + wm*WM + i*16 + (lane&15)][n0 + wn*WN + j*16 + (lane>>4)*4 + r],
// WM = BM/2, WN = BN/2, 4 waves as 2 x 2: a lane owns 4 consecutive columns of one row per 16x16 tile (16-B fp32 stores).
#pragma once
#include "common.h"
#include "det_reduce.h"

#define GB_BK 64
#define GB_LDR 72      // bf16 per LDS row of the register-pipelined kernels: 64 + 8 pad (144-B rows: 16-B aligned, conflict-light ds_read_b128)

typedef unsigned int u32x4g_t __attribute__((ext_vector_type(4)));

struct Gemm16Args {
    const unsigned short* A; const unsigned short* B; float* C; float* C2; const float* bias;
    const unsigned short* Alo; const unsigned short* Blo;     // SPLIT kernels: the low parts (same leading dimensions)
    unsigned short* out16lo;                                   // EX epilogue of SPLIT kernels: bf16(v - bf16(v)) next to out16
    int M, N, K;               // K: logical contraction length (multiple of 8; operands zero padded beyond it if needed)
    long lda, ldb, ldc;
    float alpha;
    int act;                   // 0 none, 1 relu, 2 gelu(erf)
    int splitk; long slab; int kt_per_split;
    int xcd_bind;              // 0: plain tile order, 1: M-panels bound to XCDs, 2: N-panels bound to XCDs
    int stagger;               // gemm_nt2.hip: start delay of the second-slot workgroups (units of ~4 us)
    // extended epilogue (spe_gemm_bf16nt_ex): bf16 copies of the result for the NEXT GEMMs, column sums, and the
    // derivative of a fused activation applied from its saved argument
    unsigned short* out16; long ld16;      // [M][ld16]  bf16(v)
    unsigned short* out16T; long ld16t;    // [N][ld16t] bf16(v) transposed, columns M..ld16t-1 zero
    float* colsum;                         // [N] += sum_m v (row tiles added in a fixed order: det_reduce.h)
    DetWs ws;                              // reduction workspace of the column sums
    const float* aux;                      // [M][ldc]: v *= act'(aux) (act 1: aux = forward output, 2: pre-activation)
    int half_flags;                        // bit 0: C2 is written as IEEE fp16 [M][ldc] ; bit 1: aux holds IEEE fp16 [M][ldc]  (the saved
                                           // pre-activation of the fused MLP: only act'(.) is ever taken of it - fp16 costs the gradient
                                           // ~3e-4 relative, an order below its bf16 operand rounding, and halves 51 MB per block each way)
    const float* res; const float* rgamma; // LayerScale residual: C = res[m][n] + rgamma[n] * v  (C2 still gets v)
    // dropout / DropPath inside the extended epilogue (the backbone block with its training rates: reference models/cait.py:390-391 proj_drop,
    // timm Mlp drop after GELU and after fc2, :404-416 drop_path): after the activation (or its derivative) v *= keepscale(element m * N + n) -
    // the stream of spe_dropout on the row-major [M, N] tensor; with res: C = res + sscale[m / rps] * rgamma * v.  C2 keeps the raw v.
    float drop_p; uint64_t drop_seed, drop_off;
    const float* sscale; long rps;
    int h16;                               // bit 0: A / B hold IEEE fp16 (single-term product on v_mfma_f32_16x16x32_f16) ; bit 1: C is IEEE fp16 [M][ldc] (plain epilogue) ;
                                           // bit 2: out16lo receives IEEE fp16(v) instead of bf16(v - bf16(v)) (extended epilogue; SPLIT there = "a second 16-bit tile is staged")
};

typedef _Float16 ep_h4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 ep_f2h4(float a, float b, float c, float d) {
    ep_h4_t h;
    // saturating, but a NaN stays a NaN (fminf / fmaxf would turn it into +-65504 and hide a diverged run)
    h[0] = (_Float16)((a != a) ? a : __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f)); h[1] = (_Float16)((b != b) ? b : __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f));
    h[2] = (_Float16)((c != c) ? c : __builtin_amdgcn_fmed3f(c, -65504.f, 65504.f)); h[3] = (_Float16)((d != d) ? d : __builtin_amdgcn_fmed3f(d, -65504.f, 65504.f));
    return __builtin_bit_cast(uint2, h);
}
__device__ __forceinline__ float4 ep_h2f4(uint2 u) {
    const ep_h4_t h = __builtin_bit_cast(ep_h4_t, u);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
__device__ __forceinline__ float gelu_erf16(float x) { return 0.5f * x * (1.0f + spe_erff(x * 0.70710678118654752f)); }


// ---- extended epilogue.  v = alpha*acc + bias ; C2 = v ; v = act(v) or v * act'(aux) ; C = v (optional);
// the bf16 copies go through LDS (smem16: the operand buffers, free by now; the caller has synchronised the workgroup) so that both
// the row-major and the transposed copy leave as 16-B stores of full rows; column sums: 16-lane reduction per wave row, then a
// fixed-order sum over the row tiles (det_reduce.h).  SPLIT kernels can also emit out16lo = bf16(v - bf16(v)).  LDS needed (bf16 elements):
// BM*(BN+8) + BN*(BM+8) (+ BM*(BN+8) with SPLIT); TRANSPOSED = false drops the out16T tile (BN*(BM+8)) and its support.
template <int BM, int BN, bool SPLIT, bool TRANSPOSED = true>
__device__ __forceinline__ void gemm16_epilogue_ex(const Gemm16Args& p, f32x4_t (&acc)[BM / 32][BN / 32], unsigned short* smem16,
                                                   const int m0, const int n0) {
    constexpr int NFM = BM / 32, NFN = BN / 32, WM = BM / 2, WN = BN / 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int fr = lane & 15;
    constexpr int LR = BN + 8, LT = BM + 8;            // bf16 per LDS row of the two staged tiles
    unsigned short* sR = smem16;                         // [BM][LR]  row-major tile
    unsigned short* sT = smem16 + BM * LR;               // [BN][LT]  transposed tile
    unsigned short* sRl = sT + (TRANSPOSED ? BN * LT : 0);   // [BM][LR]  low part of the row-major tile (SPLIT kernels)
    __shared__ float cs_red[2][BN];                        // column sums of the two wave rows
    __syncthreads();
    typedef short s16x4i_t __attribute__((ext_vector_type(4)));
    s16x4i_t ident;
#pragma unroll
    for (int e = 0; e < 4; ++e) ident[e] = ((lane & 15) == 4 * (lane >> 4) + e) ? (short)0x3F80 : (short)0;
    const bool vst = ((p.ldc & 3) == 0) && (!p.C || (reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                     (!p.C2 || (reinterpret_cast<uintptr_t>(p.C2) & 15) == 0) && (!p.aux || (reinterpret_cast<uintptr_t>(p.aux) & 15) == 0) &&
                     (!p.res || (reinterpret_cast<uintptr_t>(p.res) & 15) == 0);
    // interior tiles request the whole aux tile up front (the operand staging registers are free now): 16 loads in
    // flight per lane instead of one exposed round trip per 16x16 block
    const bool stage = p.out16 || (TRANSPOSED && p.out16T) || p.colsum || (SPLIT && p.out16lo);      // bf16 copies / column sums wanted at all
    const bool interior = vst && (m0 + BM <= p.M) && (n0 + BN <= p.N);
    float4 hq[NFN][NFM];
    const float* pre_src = p.aux ? p.aux : p.res;              // aux and res are mutually exclusive
    const bool aux16 = p.aux && (p.half_flags & 2), c216 = p.half_flags & 1;
    if (pre_src && interior) {
#pragma unroll
        for (int j = 0; j < NFN; ++j)
#pragma unroll
            for (int i = 0; i < NFM; ++i)
            {
                const long o = (long)(m0 + wm * WM + i * 16 + fr) * p.ldc + n0 + wn * WN + j * 16 + (lane >> 4) * 4;
                if (aux16) hq[j][i] = ep_h2f4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.aux) + o));
                else hq[j][i] = *reinterpret_cast<const float4*>(pre_src + o);
            }
    }
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        const int nl = wn * WN + j * 16 + (lane >> 4) * 4, n = n0 + nl;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = p.bias[min(n + r, p.N - 1)];
        }
        float gv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.rgamma) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = p.rgamma[min(n + r, p.N - 1)];
        }
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NFM; ++i) {
            const int ml = wm * WM + i * 16 + fr, m = m0 + ml;
            const bool rowv = m < p.M;
            const long off = (long)min(m, p.M - 1) * p.ldc + min(n, p.N - 1);
            const bool full = vst && rowv && (n + 3 < p.N);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha + bv[r];
            if (p.C2 && rowv) {
                if (c216) {
                    unsigned short* c2h = reinterpret_cast<unsigned short*>(p.C2);
                    const uint2 u = ep_f2h4(v[0], v[1], v[2], v[3]);
                    if (full) *reinterpret_cast<uint2*>(c2h + off) = u;
                    else {
                        const unsigned short e[4] = {(unsigned short)(u.x & 0xffff), (unsigned short)(u.x >> 16), (unsigned short)(u.y & 0xffff), (unsigned short)(u.y >> 16)};
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) c2h[(long)m * p.ldc + n + r] = e[r];
                    }
                } else if (full) spe_store4_stream(p.C2 + off, v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) p.C2[(long)m * p.ldc + n + r] = v[r];
                }
            }
            if (p.aux) {
                float h[4] = {0.f, 0.f, 0.f, 0.f};
                if (interior) { h[0] = hq[j][i].x; h[1] = hq[j][i].y; h[2] = hq[j][i].z; h[3] = hq[j][i].w; }
                else if (full) {
                    const float4 q = aux16 ? ep_h2f4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.aux) + off))
                                           : *reinterpret_cast<const float4*>(p.aux + off);
                    h[0] = q.x; h[1] = q.y; h[2] = q.z; h[3] = q.w;
                } else if (rowv) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N)
                            h[r] = aux16 ? (float)__builtin_bit_cast(_Float16, reinterpret_cast<const unsigned short*>(p.aux)[(long)m * p.ldc + n + r])
                                         : p.aux[(long)m * p.ldc + n + r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (p.act == 1) v[r] = h[r] > 0.f ? v[r] : 0.f;
                    else if (p.act == 2) {          // the arithmetic of cvt_bf16_kernel / act_bwd_kernel
                        const float cdf = 0.5f * (1.f + spe_erff(h[r] * 0.70710678118654752f));
                        const float pdf = 0.3989422804014327f * __expf(-0.5f * h[r] * h[r]);
                        v[r] = v[r] * (cdf + h[r] * pdf);
                    }
                }
            } else if (p.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (p.act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf16(v[r]);
            }
            if (p.drop_p > 0.f) {                     // same mask as spe_dropout on the [M, N] result (n % 4 == 0: one Philox call)
                float ks[4];
                spe_drop_scale4(p.drop_seed, p.drop_off, (uint64_t)min(m, p.M - 1) * (uint64_t)p.N + (uint64_t)n, p.drop_p, ks);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= ks[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (rowv && n + r < p.N) ? v[r] : 0.f;      // padding rows / columns stage zeros
            if (p.C && rowv) {
                float o[4] = {v[0], v[1], v[2], v[3]};
                if (p.res) {
                    float xr[4] = {0.f, 0.f, 0.f, 0.f};
                    if (interior) { xr[0] = hq[j][i].x; xr[1] = hq[j][i].y; xr[2] = hq[j][i].z; xr[3] = hq[j][i].w; }
                    else if (full) { const float4 q = *reinterpret_cast<const float4*>(p.res + off); xr[0] = q.x; xr[1] = q.y; xr[2] = q.z; xr[3] = q.w; }
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) xr[r] = p.res[(long)m * p.ldc + n + r];
                    }
                    const float ssb = p.sscale ? p.sscale[min(m, p.M - 1) / p.rps] : 1.f;      // DropPath keep scale of the row's sample
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = xr[r] + ssb * gv[r] * v[r];
                }
                if (full) spe_store4_stream(p.C + off, o[0], o[1], o[2], o[3]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) p.C[(long)m * p.ldc + n + r] = o[r];
                }
            }
            if (!stage) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[r] += v[r];
            typedef __bf16 bf16x4v_t __attribute__((ext_vector_type(4)));
            bf16x4v_t hb;
            hb[0] = (__bf16)v[0]; hb[1] = (__bf16)v[1]; hb[2] = (__bf16)v[2]; hb[3] = (__bf16)v[3];
            const uint2 u = __builtin_bit_cast(uint2, hb);
            *reinterpret_cast<uint2*>(sR + ml * LR + nl) = u;
            if constexpr (SPLIT) {
                if (p.out16lo) *reinterpret_cast<uint2*>(sRl + ml * LR + nl) = spe_second16(v, u, (p.h16 & 4) != 0);      // low part, or the fp16 copy
            }
            // transposed copy: one MFMA against the identity moves the lane ownership from (row m, 4 columns) to
            // (column n, 4 rows) - exact in bf16 - so the transposed tile is staged with 8-B writes as well
            if constexpr (TRANSPOSED) {
                typedef short s16x4e_t __attribute__((ext_vector_type(4)));
                const f32x4_t t = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4e_t, u), ident,
                                                                              (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                bf16x4v_t tb;
                tb[0] = (__bf16)t[0]; tb[1] = (__bf16)t[1]; tb[2] = (__bf16)t[2]; tb[3] = (__bf16)t[3];
                *reinterpret_cast<uint2*>(sT + (wn * WN + j * 16 + fr) * LT + wm * WM + i * 16 + (lane >> 4) * 4) = __builtin_bit_cast(uint2, tb);
            }
        }
        if (p.colsum) {
            // 4 column sums over the 16 row lanes in 5 exchanges: lane pairs split the columns (xor 1: even lanes keep
            // columns 0,1, odd lanes 2,3), then lane pairs of pairs (xor 2), then plain sums over xor 4 and 8
            const bool o1 = lane & 1, o2 = lane & 2;
            const float r0 = __shfl_xor(o1 ? cs[0] : cs[2], 1, 64), r1 = __shfl_xor(o1 ? cs[1] : cs[3], 1, 64);
            const float b0 = (o1 ? cs[2] : cs[0]) + r0, b1 = (o1 ? cs[3] : cs[1]) + r1;
            float c = (o2 ? b1 : b0) + __shfl_xor(o2 ? b0 : b1, 2, 64);
            c += __shfl_xor(c, 4, 64);
            c += __shfl_xor(c, 8, 64);
            const int col = 2 * (fr & 1) + ((fr >> 1) & 1);          // the column this lane ended up with
            if (fr < 4) cs_red[wm][nl + col] = c;
        }
    }
    __syncthreads();
    for (int part = 0; part < (SPLIT ? 2 : 1); ++part) {       // [BM][BN] row-major: 16 B = 8 columns per thread and pass
        unsigned short* o16 = part ? p.out16lo : p.out16;
        const unsigned short* sS = part ? sRl : sR;
        if (!o16) continue;
        constexpr int CH = BN / 8;
        for (int idx = threadIdx.x; idx < BM * CH; idx += 256) {
            const int ml = idx / CH, ch = idx % CH, m = m0 + ml, n = n0 + ch * 8;
            if (m >= p.M || n >= p.N) continue;
            const u32x4g_t q = *reinterpret_cast<const u32x4g_t*>(sS + ml * LR + ch * 8);
            unsigned short* dst = o16 + (long)m * p.ld16 + n;
            if (n + 7 < p.N && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) spe_store16_stream(dst, q);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (n + e < p.N) dst[e] = sS[ml * LR + ch * 8 + e];
            }
        }
    }
    if (TRANSPOSED && p.out16T) {      // [BN][BM] transposed: 16 B = 8 rows of the result per thread and pass; zero columns up to ld16t
        constexpr int CH = BM / 8;
        for (int idx = threadIdx.x; idx < BN * CH; idx += 256) {
            const int nl = idx / CH, ch = idx % CH, n = n0 + nl, m = m0 + ch * 8;
            if (n >= p.N || m >= p.ld16t) continue;
            const u32x4g_t q = *reinterpret_cast<const u32x4g_t*>(sT + nl * LT + ch * 8);
            unsigned short* dst = p.out16T + (long)n * p.ld16t + m;
            if (m + 7 < p.ld16t && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) spe_store16_stream(dst, q);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (m + e < p.ld16t) dst[e] = sT[nl * LT + ch * 8 + e];
            }
        }
    }
    if (p.colsum)       // the two wave rows, then the row tiles of this column tile in index order: colsum += total
        det_reduce(p.ws, n0 / BN, m0 / BM, (p.M + BM - 1) / BM, BN, threadIdx.x, 256,
                   [&](int c) { return cs_red[0][c] + cs_red[1][c]; },
                   [&](int c, float t) { if (n0 + c < p.N) p.colsum[n0 + c] += t; });
}

// ---- plain epilogue: C = act(alpha*acc + bias) (+ C2 = pre-activation), or the private slab of a K split
template <int BM, int BN>
__device__ __forceinline__ void gemm16_epilogue_plain(const Gemm16Args& p, f32x4_t (&acc)[BM / 32][BN / 32], float* C, const int m0, const int n0) {
    constexpr int NFM = BM / 32, NFN = BN / 32, WM = BM / 2, WN = BN / 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int fr = lane & 15;
    // acc[i][j][r] = C[m0 + wm*WM + i*16 + (lane&15)][n0 + wn*WN + j*16 + (lane>>4)*4 + r]: one 16-B store per tile)
    float* C2 = p.C2;
    const bool vst = ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                     (!C2 || (reinterpret_cast<uintptr_t>(C2) & 15) == 0);
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        const int n = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
        if (n >= p.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && p.splitk == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = p.bias[min(n + r, p.N - 1)];
        }
#pragma unroll
        for (int i = 0; i < NFM; ++i) {
            const int m = m0 + wm * WM + i * 16 + fr;
            if (m >= p.M) continue;
            const long off = (long)m * p.ldc + n;
            const bool full = vst && (n + 3 < p.N);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha;
            if (p.splitk == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bv[r];
                if (C2) {
                    if (full) spe_store4_stream(C2 + off, v[0], v[1], v[2], v[3]);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) C2[off + r] = v[r];
                    }
                }
                if (p.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                } else if (p.act == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf16(v[r]);
                }
            }
            if (p.h16 & 2) {        // fp16 output (saturating, NaN kept): the consumer packs these values into fp16 MFMA operands anyway
                unsigned short* C16 = reinterpret_cast<unsigned short*>(C);
                const uint2 h = ep_f2h4(v[0], v[1], v[2], v[3]);
                if (full) __builtin_nontemporal_store(h.x | ((unsigned long long)h.y << 32), reinterpret_cast<unsigned long long*>(C16 + off));
                else {
                    const unsigned short e[4] = {(unsigned short)(h.x & 0xffff), (unsigned short)(h.x >> 16), (unsigned short)(h.y & 0xffff), (unsigned short)(h.y >> 16)};
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) C16[off + r] = e[r];
                }
                continue;
            }
            // split-K: the private slab of this split (zeros if the split was empty), summed by the caller
            if (full) spe_store4_stream(C + off, v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.N) C[off + r] = v[r];
            }
        }
    }
}


// ---- fp16-output epilogue of the 128-wide LDS-DMA kernels (Gemm16Args::h16 bit 1): v = alpha * acc + bias -> IEEE fp16 (saturating,
// NaN kept), staged through LDS (the operand ring is free by now; the caller has synchronised the workgroup) so that the tile leaves
// as 16-B stores of whole 256-B row segments - a lane's own 4 columns would be 8-B stores in 32-B pieces (measured: slower than the
// fp32 stores they replace).  Needs N % 8 == 0, ldc % 8 == 0 and a 16-B aligned C.
template <int BM, int BN>
__device__ __forceinline__ void gemm16_epilogue_h16(const Gemm16Args& p, f32x4_t (&acc)[BM / 32][BN / 32], unsigned short* smem16, const int m0, const int n0) {
    constexpr int NFM = BM / 32, NFN = BN / 32, WM = BM / 2, WN = BN / 2, LR = BN + 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int fr = lane & 15;
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        const int nl = wn * WN + j * 16 + (lane >> 4) * 4, n = n0 + nl;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = p.bias[min(n + r, p.N - 1)];
        }
#pragma unroll
        for (int i = 0; i < NFM; ++i) {
            const int ml = wm * WM + i * 16 + fr;
            *reinterpret_cast<uint2*>(smem16 + ml * LR + nl) = ep_f2h4(acc[i][j][0] * p.alpha + bv[0], acc[i][j][1] * p.alpha + bv[1],
                                                                      acc[i][j][2] * p.alpha + bv[2], acc[i][j][3] * p.alpha + bv[3]);
        }
    }
    __syncthreads();
    unsigned short* C16 = reinterpret_cast<unsigned short*>(p.C);
    for (int idx = threadIdx.x; idx < BM * (BN / 8); idx += 256) {
        const int r = idx / (BN / 8), c8 = idx % (BN / 8);
        const int m = m0 + r, n = n0 + c8 * 8;
        if (m >= p.M || n >= p.N) continue;
        spe_store16_stream(C16 + (long)m * p.ldc + n, *reinterpret_cast<const spe_u32x4_t*>(smem16 + r * LR + c8 * 8));
    }
}
