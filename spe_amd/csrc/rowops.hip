// Wavefront-reduced row kernels: LayerNorm fwd/bwd, masked softmax (+dropout) fwd/bwd,
// column sums (bias / gamma gradients), LayerScale-residual, GELU/ReLU backward, dropout.
// All tensors fp32, row-major, contiguous unless a stride is passed.  HBM-bound kernels:
// every lane moves 16 B per access where the row length allows it.
#include <cstdlib>
#include "common.h"
#include "det_reduce.h"

// ------------------------------------------------------------------------------------------
// LayerNorm (reference: nn.LayerNorm at models/cait.py:403,407 eps=1e-6; transformer.py:264-265,
// 342-344 eps=1e-5).  One wave per row; C % 4 == 0 and C <= 1024.
// ------------------------------------------------------------------------------------------
#define LN_MAXV 4  // float4 per lane -> C <= 1024

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     long R, int C, float eps, unsigned short* __restrict__ y16,
                                                     unsigned short* __restrict__ y16lo, bool lo_f16) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int C4 = C >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + row * C);
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) { v[i] = xr[c]; s += v[i].x + v[i].y + v[i].z + v[i].w; }
    }
    const float mu = spe_wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) {
            const float a = v[i].x - mu, b = v[i].y - mu, d = v[i].z - mu, e = v[i].w - mu;
            q += a * a + b * b + d * d + e * e;
        }
    }
    const float rs = rsqrtf(spe_wave_sum(q) / (float)C + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    float4* yr = reinterpret_cast<float4*>(y + row * C);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) {
            const float4 g = g4[c], b = b4[c];
            float4 o;
            o.x = (v[i].x - mu) * rs * g.x + b.x; o.y = (v[i].y - mu) * rs * g.y + b.y;
            o.z = (v[i].z - mu) * rs * g.z + b.z; o.w = (v[i].w - mu) * rs * g.w + b.w;
            yr[c] = o;
            if (y16) {      // the bf16 operand of the Linear that consumes y (same rounding as spe_cvt_bf16), from the same pass
                typedef __bf16 bf16x4l_t __attribute__((ext_vector_type(4)));
                bf16x4l_t h;
                h[0] = (__bf16)o.x; h[1] = (__bf16)o.y; h[2] = (__bf16)o.z; h[3] = (__bf16)o.w;
                *reinterpret_cast<uint2*>(y16 + row * C + 4 * c) = __builtin_bit_cast(uint2, h);
                if (y16lo) {    // low part of the split operand (precision mode bf16s): bf16(y - bf16(y)) - or the fp16 copy (lo_f16)
                    const float ov[4] = {o.x, o.y, o.z, o.w};
                    *reinterpret_cast<uint2*>(y16lo + row * C + 4 * c) = spe_second16(ov, __builtin_bit_cast(uint2, h), lo_f16);
                }
            }
        }
    }
}

// dx per row; dgamma/dbeta accumulated per wave in registers over a grid-stride row loop, combined through LDS, then across
// the workgroups in a fixed order (det_reduce.h) and added to the running gradient.
// LS (round 5): the LayerScale backward of the node that CONSUMES dx - dx is the `dout` of out = res + ls_gamma * ls_y (the branch Linear before this
// norm's input, reference models/cait.py:404-405) - rides on the same pass: ls_dy16 = bf16(ls_gamma * dx) (the operand of that Linear's backward GEMMs),
// ls_db += sum_r ls_gamma * dx (its bias gradient), ls_dg += sum_r dx * ls_y (the LayerScale gradient) - what spe_layerscale_residual_bwd16 would compute
// from dx in a launch of its own after re-reading it.
template <int NW, bool LS, int MAXV>
__global__ __launch_bounds__(NW * 64) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     long R, int C, const float* __restrict__ add, float* __restrict__ dz,
                                                     float p, uint64_t seed, uint64_t offset, DetWs ws, const float* __restrict__ dy2,
                                                     const float* __restrict__ ls_y, const float* __restrict__ ls_gamma,
                                                     unsigned short* __restrict__ ls_dy16, float* __restrict__ ls_db, float* __restrict__ ls_dg) {
    extern __shared__ float red_raw[];                 // [2 (LS: 4)][NW][C + 4]
    const int ldr = C + 4;
    auto red = [&](int k, int wv, int c) -> float& { return red_raw[((long)k * NW + wv) * ldr + c]; };
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int C4 = C >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    float4 ag[MAXV], ab[MAXV];
    float4 lg[LS ? MAXV : 1], lb[LS ? MAXV : 1];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0); }
#pragma unroll
    for (int i = 0; i < (LS ? MAXV : 1); ++i) { lg[i] = make_float4(0, 0, 0, 0); lb[i] = make_float4(0, 0, 0, 0); }
    for (long row = (long)blockIdx.x * NW + w; row < R; row += (long)gridDim.x * NW) {
        const float4* xr = reinterpret_cast<const float4*>(x + row * C);
        const float4* dr = reinterpret_cast<const float4*>(dy + row * C);
        // dy2 (optional): the gradient of a SECOND consumer of the norm's output (post-norm layers: the output feeds a Linear and the next residual) - the
        // sum autograd would take in a launch of its own happens while the row is loaded
        const float4* dr2 = dy2 ? reinterpret_cast<const float4*>(dy2 + row * C) : nullptr;
        const float mu = mean[row], rs = rstd[row];
        float4 xh[MAXV], dg[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < C4) {
                const float4 xv = xr[c], g = g4[c];
                float4 dv = dr[c];
                if (dr2) { const float4 d2 = dr2[c]; dv.x += d2.x; dv.y += d2.y; dv.z += d2.z; dv.w += d2.w; }
                xh[i].x = (xv.x - mu) * rs; xh[i].y = (xv.y - mu) * rs; xh[i].z = (xv.z - mu) * rs; xh[i].w = (xv.w - mu) * rs;
                dg[i].x = dv.x * g.x; dg[i].y = dv.y * g.y; dg[i].z = dv.z * g.z; dg[i].w = dv.w * g.w;
                s1 += dg[i].x + dg[i].y + dg[i].z + dg[i].w;
                s2 += dg[i].x * xh[i].x + dg[i].y * xh[i].y + dg[i].z * xh[i].z + dg[i].w * xh[i].w;
                ag[i].x += dv.x * xh[i].x; ag[i].y += dv.y * xh[i].y; ag[i].z += dv.z * xh[i].z; ag[i].w += dv.w * xh[i].w;
                ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
            }
        }
        s1 = spe_wave_sum(s1) / (float)C;
        s2 = spe_wave_sum(s2) / (float)C;
        float4* dxr = reinterpret_cast<float4*>(dx + row * C);
        const float4* ar = add ? reinterpret_cast<const float4*>(add + row * C) : nullptr;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < C4) {
                float4 o;
                o.x = rs * (dg[i].x - s1 - xh[i].x * s2); o.y = rs * (dg[i].y - s1 - xh[i].y * s2);
                o.z = rs * (dg[i].z - s1 - xh[i].z * s2); o.w = rs * (dg[i].w - s1 - xh[i].w * s2);
                if (ar) { const float4 a4 = ar[c]; o.x += a4.x; o.y += a4.y; o.z += a4.z; o.w += a4.w; }   // + gradient of the skip path
                dxr[c] = o;
                if constexpr (LS) {
                    const float4 yv = reinterpret_cast<const float4*>(ls_y + row * C)[c], gg = reinterpret_cast<const float4*>(ls_gamma)[c];
                    lg[i].x += o.x * yv.x; lg[i].y += o.y * yv.y; lg[i].z += o.z * yv.z; lg[i].w += o.w * yv.w;
                    const float4 q = make_float4(o.x * gg.x, o.y * gg.y, o.z * gg.z, o.w * gg.w);
                    lb[i].x += q.x; lb[i].y += q.y; lb[i].z += q.z; lb[i].w += q.w;
                    typedef __bf16 bf16x4n_t __attribute__((ext_vector_type(4)));
                    bf16x4n_t h;
                    h[0] = (__bf16)q.x; h[1] = (__bf16)q.y; h[2] = (__bf16)q.z; h[3] = (__bf16)q.w;
                    *reinterpret_cast<uint2*>(ls_dy16 + row * C + 4 * c) = __builtin_bit_cast(uint2, h);
                }
                if (dz) {          // gradient of the dropped branch of norm(x + dropout(z)): same mask as ln_res_fwd_kernel
                    float ks[4];
                    spe_drop_scale4(seed, offset, (uint64_t)(row * C + 4 * c), p, ks);
                    reinterpret_cast<float4*>(dz + row * C)[c] = make_float4(o.x * ks[0], o.y * ks[1], o.z * ks[2], o.w * ks[3]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) {
            red(0, w, 4 * c + 0) = ag[i].x; red(0, w, 4 * c + 1) = ag[i].y; red(0, w, 4 * c + 2) = ag[i].z; red(0, w, 4 * c + 3) = ag[i].w;
            red(1, w, 4 * c + 0) = ab[i].x; red(1, w, 4 * c + 1) = ab[i].y; red(1, w, 4 * c + 2) = ab[i].z; red(1, w, 4 * c + 3) = ab[i].w;
            if constexpr (LS) {
                red(2, w, 4 * c + 0) = lg[i].x; red(2, w, 4 * c + 1) = lg[i].y; red(2, w, 4 * c + 2) = lg[i].z; red(2, w, 4 * c + 3) = lg[i].w;
                red(3, w, 4 * c + 0) = lb[i].x; red(3, w, 4 * c + 1) = lb[i].y; red(3, w, 4 * c + 2) = lb[i].z; red(3, w, 4 * c + 3) = lb[i].w;
            }
        }
    }
    __syncthreads();
    // the 16 waves' column sums in wave order, then across workgroups in a fixed order (det_reduce.h): dgamma / dbeta (/ ls_dg / ls_db) += total
    det_reduce(ws, 0, blockIdx.x, gridDim.x, (LS ? 4 : 2) * C, threadIdx.x, NW * 64,
               [&](int c) {
                   const int k = c / C, cc = c - k * C;
                   float t = 0.f;
#pragma unroll
                   for (int wv = 0; wv < NW; ++wv) t += red(k, wv, cc);
                   return t;
               },
               [&](int c, float t) {
                   const int k = c / C, cc = c - k * C;
                   float* d = (k == 0) ? dgamma : ((k == 1) ? dbeta : ((k == 2) ? ls_dg : ls_db));
                   d[cc] += t;
               });
}

// Post-norm residual site of the DETR encoder / decoder layers, `norm(x + dropout(z))` (reference models/transformer.py:
// 279-287, 384-386, 420-421, 426-427), as ONE pass: s = x + z * keepscale(row*C + c) is written (LayerNorm's input, kept for
// the backward), then normalised exactly like ln_fwd_kernel.  The dropout stream is that of dropout_kernel (element index
// row*C + c), so the fused and the unfused composition draw the same mask.
__global__ __launch_bounds__(256) void ln_res_fwd_kernel(const float* __restrict__ x, const float* __restrict__ z,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ sum, float* __restrict__ y, float* __restrict__ mean,
                                                         float* __restrict__ rstd, long R, int C, float eps, float p,
                                                         uint64_t seed, uint64_t offset) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int C4 = C >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + row * C);
    const float4* zr = reinterpret_cast<const float4*>(z + row * C);
    float4* sr = reinterpret_cast<float4*>(sum + row * C);
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) {
            const float4 a = xr[c];
            float4 b = zr[c];
            if (p > 0.f) {
                float ks[4];
                spe_drop_scale4(seed, offset, (uint64_t)(row * C + 4 * c), p, ks);
                b.x *= ks[0]; b.y *= ks[1]; b.z *= ks[2]; b.w *= ks[3];
            }
            v[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            sr[c] = v[i];
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
    }
    const float mu = spe_wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) {
            const float a = v[i].x - mu, b = v[i].y - mu, d = v[i].z - mu, e = v[i].w - mu;
            q += a * a + b * b + d * d + e * e;
        }
    }
    const float rs = rsqrtf(spe_wave_sum(q) / (float)C + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    float4* yr = reinterpret_cast<float4*>(y + row * C);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) {
            const float4 g = g4[c], b = b4[c];
            float4 o;
            o.x = (v[i].x - mu) * rs * g.x + b.x; o.y = (v[i].y - mu) * rs * g.y + b.y;
            o.z = (v[i].z - mu) * rs * g.z + b.z; o.w = (v[i].w - mu) * rs * g.w + b.w;
            yr[c] = o;
        }
    }
}

// Half-wave-per-row variant for C = 128 * NV (NV <= 4: 128, 256, 384, 512 - every model width here): a wave normalises TWO rows,
// 32 lanes each, so no lane idles (C = 384 is 96 float4: 64 + 32 in the wave-per-row mapping) and a lane keeps NV 16-B loads in
// flight instead of 1.5 on average; the reductions are 5 DPP steps inside the half.  Same arithmetic order per row element
// as ln_fwd_kernel up to the reduction tree.
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_hw_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        float* __restrict__ mean, float* __restrict__ rstd,
                                                        long R, float eps, unsigned short* __restrict__ y16,
                                                        unsigned short* __restrict__ y16lo, bool lo_f16) {
    constexpr int C4 = 32 * NV, C = 4 * C4;
    const int lane = threadIdx.x & 63, hl = lane & 31;
    const long stride = (long)gridDim.x * 8;
    long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    float4 gq[NV], bq[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { gq[i] = g4[hl + 32 * i]; bq[i] = b4[hl + 32 * i]; }
    float4 v[NV];
    {
        const float4* xr = reinterpret_cast<const float4*>(x + (row0 < R ? row0 : R - 1) * C);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = xr[hl + 32 * i];
    }
    // rows of this half-wave: row0, row0 + stride, ...; the loop count is wave-uniform (the other half may run one row less)
    const long first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    for (long base = first; base < R; base += stride, row0 += stride) {
        const bool rv = row0 < R;
        const long row = rv ? row0 : R - 1;
        float4 nv[NV];
        {   // next rows requested before this row's reductions
            const long nr = row0 + stride;
            const float4* xn = reinterpret_cast<const float4*>(x + (nr < R ? nr : R - 1) * C);
#pragma unroll
            for (int i = 0; i < NV; ++i) nv[i] = xn[hl + 32 * i];
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mu = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float a = v[i].x - mu, b = v[i].y - mu, d = v[i].z - mu, e = v[i].w - mu;
            q += a * a + b * b + d * d + e * e;
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rs = rsqrtf(q / (float)C + eps);
        if (rv) {
            if (hl == 0) { mean[row] = mu; rstd[row] = rs; }
            float4* yr = reinterpret_cast<float4*>(y + row * C);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = hl + 32 * i;
                float4 o;
                o.x = (v[i].x - mu) * rs * gq[i].x + bq[i].x; o.y = (v[i].y - mu) * rs * gq[i].y + bq[i].y;
                o.z = (v[i].z - mu) * rs * gq[i].z + bq[i].z; o.w = (v[i].w - mu) * rs * gq[i].w + bq[i].w;
                yr[c] = o;
                if (y16) {
                    typedef __bf16 bf16x4h_t __attribute__((ext_vector_type(4)));
                    bf16x4h_t h;
                    h[0] = (__bf16)o.x; h[1] = (__bf16)o.y; h[2] = (__bf16)o.z; h[3] = (__bf16)o.w;
                    *reinterpret_cast<uint2*>(y16 + row * C + 4 * c) = __builtin_bit_cast(uint2, h);
                    if (y16lo) {
                        const float ov[4] = {o.x, o.y, o.z, o.w};
                        *reinterpret_cast<uint2*>(y16lo + row * C + 4 * c) = spe_second16(ov, __builtin_bit_cast(uint2, h), lo_f16);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = nv[i];
    }
}

// C-ABI: see include/spe_hip.h (spe_layernorm_res_fwd).
extern "C" int spe_layernorm_res_fwd(const float* x, const float* z, const float* gamma, const float* beta, float* sum, float* y,
                                     float* mean, float* rstd, long R, int C, float eps, float p, uint64_t seed, uint64_t offset,
                                     hipStream_t st) {
    if (R <= 0) return 0;
    if ((C & 3) || C > 256 * LN_MAXV) return -2;
    hipLaunchKernelGGL(ln_res_fwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, x, z, gamma, beta, sum, y, mean, rstd, R, C,
                       eps, p, seed, offset);
    SPE_CHECK_LAUNCH();
    return 0;
}

static int ln_fwd_launch(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                         float* rstd, long R, int C, float eps, void* y16, void* y16lo, bool lo_f16, hipStream_t st) {
    if (R <= 0) return 0;
    if ((C & 3) || C > 256 * LN_MAXV || (y16lo && !y16)) return -2;
    if ((C % 128) == 0 && C <= 512) {          // half-wave-per-row kernel
        long nwg_ = (R + 7) / 8; if (nwg_ > 512) nwg_ = 512;
        const dim3 grid((unsigned)nwg_);
        unsigned short* h16 = reinterpret_cast<unsigned short*>(y16);
        unsigned short* l16 = reinterpret_cast<unsigned short*>(y16lo);
        switch (C / 128) {
            case 1: hipLaunchKernelGGL(ln_fwd_hw_kernel<1>, grid, dim3(256), 0, st, x, gamma, beta, y, mean, rstd, R, eps, h16, l16, lo_f16); break;
            case 2: hipLaunchKernelGGL(ln_fwd_hw_kernel<2>, grid, dim3(256), 0, st, x, gamma, beta, y, mean, rstd, R, eps, h16, l16, lo_f16); break;
            case 3: hipLaunchKernelGGL(ln_fwd_hw_kernel<3>, grid, dim3(256), 0, st, x, gamma, beta, y, mean, rstd, R, eps, h16, l16, lo_f16); break;
            default: hipLaunchKernelGGL(ln_fwd_hw_kernel<4>, grid, dim3(256), 0, st, x, gamma, beta, y, mean, rstd, R, eps, h16, l16, lo_f16); break;
        }
        SPE_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, R, C, eps,
                       reinterpret_cast<unsigned short*>(y16), reinterpret_cast<unsigned short*>(y16lo), lo_f16);
    SPE_CHECK_LAUNCH();
    return 0;
}
// C-ABI: see include/spe_hip.h
extern "C" int spe_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                 float* rstd, long R, int C, float eps, void* y16, void* y16lo, hipStream_t st) {
    return ln_fwd_launch(x, gamma, beta, y, mean, rstd, R, C, eps, y16, y16lo, false, st);
}
// ... with the second 16-bit copy as IEEE fp16 (the operand of a single-term fp16 forward product) instead of the low part of the split
extern "C" int spe_layernorm_fwd_h(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                   float* rstd, long R, int C, float eps, void* y16, void* yh16, hipStream_t st) {
    return ln_fwd_launch(x, gamma, beta, y, mean, rstd, R, C, eps, y16, yh16, true, st);
}
static int ln_bwd_launch(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                         float* dgamma, float* dbeta, long R, int C, const float* add, float* dz, float p, uint64_t seed,
                         uint64_t offset, hipStream_t st, const float* ls_y = nullptr, const float* ls_gamma = nullptr, void* ls_dy16 = nullptr,
                         float* ls_db = nullptr, float* ls_dg = nullptr, const float* dy2 = nullptr) {
    if (R <= 0) return 0;
    if ((C & 3) || C > 256 * LN_MAXV) return -2;
    const bool ls = ls_y != nullptr;
    if (ls && (!ls_gamma || !ls_dy16 || !ls_db || !ls_dg || dz || C > 512 ||
               ((reinterpret_cast<uintptr_t>(ls_y) | reinterpret_cast<uintptr_t>(ls_gamma) | reinterpret_cast<uintptr_t>(ls_dy16)) & 15))) return -2;
    // 16 waves per workgroup, at most 256 workgroups: every workgroup ends with a 2*C-value (LS: 4*C) partial for the cross-workgroup sum
    constexpr int NW = 16;
    static bool attr[3] = {false, false, false};
    const int variant = ls ? 1 : (C <= 512 ? 2 : 0);
    if (!attr[variant]) {
        const void* fn = ls ? reinterpret_cast<const void*>(&ln_bwd_kernel<NW, true, 2>)
                            : (C <= 512 ? reinterpret_cast<const void*>(&ln_bwd_kernel<NW, false, 2>) : reinterpret_cast<const void*>(&ln_bwd_kernel<NW, false, LN_MAXV>));
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           ls ? 4 * NW * (512 + 4) * (int)sizeof(float) : 2 * NW * (256 * LN_MAXV + 4) * (int)sizeof(float));
        if (e != hipSuccess) return (int)e;
        attr[variant] = true;
    }
    long nb = (R + NW - 1) / NW; if (nb > 256) nb = 256;
    DetWs ws = spe_detws();
    const int nk = ls ? 4 : 2;
    // deferred (destinations inside the registered bucket ranges): the workgroups leave their partials behind, the totals land at the next flush
    const DetDeferSeg sg[4] = {{dgamma, C}, {dbeta, C}, {ls_dg, C}, {ls_db, C}};
    float* region = det_defer_try(1, nb, nk * C, nk, sg, st);
    if (region) ws.defer = region; else DET_CHECK(ws, 1, nb, nk * C);
    if (ls) hipLaunchKernelGGL((ln_bwd_kernel<NW, true, 2>), dim3((unsigned)nb), dim3(NW * 64), 4 * NW * (C + 4) * (int)sizeof(float), st, dy, x, gamma, mean,
                               rstd, dx, dgamma, dbeta, R, C, add, dz, p, seed, offset, ws, dy2, ls_y, ls_gamma, reinterpret_cast<unsigned short*>(ls_dy16), ls_db, ls_dg);
    else if (C <= 512) hipLaunchKernelGGL((ln_bwd_kernel<NW, false, 2>), dim3((unsigned)nb), dim3(NW * 64), 2 * NW * (C + 4) * (int)sizeof(float), st, dy, x, gamma, mean,
                                          rstd, dx, dgamma, dbeta, R, C, add, dz, p, seed, offset, ws, dy2, nullptr, nullptr, nullptr, nullptr, nullptr);
    else hipLaunchKernelGGL((ln_bwd_kernel<NW, false, LN_MAXV>), dim3((unsigned)nb), dim3(NW * 64), 2 * NW * (C + 4) * (int)sizeof(float), st, dy, x, gamma, mean,
                            rstd, dx, dgamma, dbeta, R, C, add, dz, p, seed, offset, ws, dy2, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (region) det_defer_commit(region, 1, nb, nk * C, nk, sg, 1);
    SPE_CHECK_LAUNCH();
    return 0;
}
// C-ABI: see include/spe_hip.h (spe_layernorm_bwd_ls): LayerNorm backward + the LayerScale backward of the node that consumes dx, one pass.
extern "C" int spe_layernorm_bwd_ls(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                                    float* dgamma, float* dbeta, long R, int C, const float* add, const float* ls_y, const float* ls_gamma,
                                    void* ls_dy16, float* ls_db, float* ls_dg, hipStream_t st) {
    if (!ls_y) return -2;
    return ln_bwd_launch(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, R, C, add, nullptr, 0.f, 0, 0, st, ls_y, ls_gamma, ls_dy16, ls_db, ls_dg);
}
extern "C" int spe_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                 const float* rstd, float* dx, float* dgamma, float* dbeta, long R, int C,
                                 const float* add, hipStream_t st) {
    return ln_bwd_launch(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, R, C, add, nullptr, 0.f, 0, 0, st);
}
// C-ABI: see include/spe_hip.h (spe_layernorm_res_bwd): backward of spe_layernorm_res_fwd; ds = gradient of the sum (= of x),
// dz = ds * keepscale (null when p == 0: the branch gradient is ds itself); dy2 (optional): a second gradient of the output, added to dy on load.
extern "C" int spe_layernorm_res_bwd(const float* dy, const float* dy2, const float* sum, const float* gamma, const float* mean, const float* rstd,
                                     float* ds, float* dz, float* dgamma, float* dbeta, long R, int C, float p, uint64_t seed,
                                     uint64_t offset, hipStream_t st) {
    return ln_bwd_launch(dy, sum, gamma, mean, rstd, ds, dgamma, dbeta, R, C, nullptr, (p > 0.f) ? dz : nullptr, p, seed, offset, st,
                         nullptr, nullptr, nullptr, nullptr, nullptr, dy2);
}

// ------------------------------------------------------------------------------------------
// Masked softmax over the last axis of scores[B,H,Nq,ld] (Nk valid columns), one wave per row.
// key-padding mask [B,Nk] (1 = padded key -> -inf), reference models/attention.py:363-371.
// P = softmax(S) is written for backward; Pd = dropout(P) (reference :373) only when p_drop > 0.  The dropout element
// index is row*ld + k (ld = padded row stride): the same stream mha_flash.hip draws from, 4 keys per Philox block.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, const unsigned char* __restrict__ mask,
                                                          float* __restrict__ P, float* __restrict__ Pd,
                                                          long rows, int rows_per_batch, int Nk, long ld,
                                                          float p_drop, uint64_t seed, uint64_t offset) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = S + row * ld;
    const unsigned char* mk = mask ? mask + (row / rows_per_batch) * (long)Nk : nullptr;
    float m = -INFINITY, l = 0.f;
    for (int k = lane; k < Nk; k += 64) {
        float v = s[k];
        if (mk && mk[k]) v = -INFINITY;
        if (v > m) { l = l * __expf(m - v) + 1.f; m = v; }
        else if (v > -INFINITY) l += __expf(v - m);
    }
    const float M = spe_wave_max(m);
    l = (m > -INFINITY) ? l * __expf(m - M) : 0.f;
    const float inv = 1.f / spe_wave_sum(l);
    float* p = P + row * ld;
    float* pd = Pd ? Pd + row * ld : nullptr;
    for (int k = lane; k < Nk; k += 64) {
        float v = s[k];
        if (mk && mk[k]) v = -INFINITY;
        const float pr = __expf(v - M) * inv;
        p[k] = pr;
        if (pd) pd[k] = pr * spe_drop_scale(seed, offset, (uint64_t)(row * ld + k), p_drop);
    }
}

// dS = P * (dP - sum_k dP*P), dP = dPd * dropscale.  dS may alias dPd.
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dPd, const float* __restrict__ P,
                                                          float* __restrict__ dS, long rows, int Nk, long ld,
                                                          float p_drop, uint64_t seed, uint64_t offset) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* d = dPd + row * ld;
    const float* p = P + row * ld;
    float acc = 0.f;
    for (int k = lane; k < Nk; k += 64) {
        float g = d[k];
        if (p_drop > 0.f) g *= spe_drop_scale(seed, offset, (uint64_t)(row * ld + k), p_drop);
        acc += g * p[k];
    }
    acc = spe_wave_sum(acc);
    float* o = dS + row * ld;
    for (int k = lane; k < Nk; k += 64) {
        float g = d[k];
        if (p_drop > 0.f) g *= spe_drop_scale(seed, offset, (uint64_t)(row * ld + k), p_drop);
        o[k] = p[k] * (g - acc);
    }
}

extern "C" int spe_softmax_fwd(const float* S, const unsigned char* mask, float* P, float* Pd, int B, int H, int Nq,
                               int Nk, long ld, float p_drop, uint64_t seed, uint64_t offset, hipStream_t st) {
    const long rows = (long)B * H * Nq;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S, mask, P,
                       p_drop > 0.f ? Pd : nullptr, rows, H * Nq, Nk, ld, p_drop, seed, offset);
    SPE_CHECK_LAUNCH();
    return 0;
}
extern "C" int spe_softmax_bwd(const float* dPd, const float* P, float* dS, int B, int H, int Nq, int Nk, long ld,
                               float p_drop, uint64_t seed, uint64_t offset, hipStream_t st) {
    const long rows = (long)B * H * Nq;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, dPd, P, dS, rows, Nk, ld,
                       p_drop, seed, offset);
    SPE_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// Column sum: out[c] += sum_r in[r*ld + c]  (bias gradients, split-K slab sums).  out must be pre-zeroed / hold
// the running gradient.
// ------------------------------------------------------------------------------------------
// Two shapes occur: a few very wide rows (the split-K slabs of a weight gradient: R <= 32, C ~ 1e5..1e6) and
// tall narrow matrices (bias gradients: R ~ 1e4, C <= 2048).
//   wide: one thread per 4 columns, float4 loads down the R rows, plain read-modify-write of out (no atomics)
//   tall: block = 16 column quads (64 columns) x 16 row lanes, float4 loads, LDS reduce, fixed-order sum over the row segments
__global__ __launch_bounds__(256) void colsum_wide_kernel(const float* __restrict__ in, float* __restrict__ out, int R, long C, long ld,
                                                          int accumulate) {
    const long c = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= C) return;
    float4 acc = accumulate ? *reinterpret_cast<const float4*>(out + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
        const float4 v = *reinterpret_cast<const float4*>(in + r * ld + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(out + c) = acc;
}
__global__ __launch_bounds__(256) void colsum_tall4_kernel(const float* __restrict__ in, float* __restrict__ out, long R, int C, long ld,
                                                           int accumulate, DetWs ws) {
    __shared__ float4 red[16][16];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = (blockIdx.x * 16 + cq) * 4;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < C)
        for (long r = (long)blockIdx.y * 16 + rl; r < R; r += (long)gridDim.y * 16) {
            const float4 v = *reinterpret_cast<const float4*>(in + r * ld + c);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    red[rl][cq] = acc;
    __syncthreads();
    // 16 row lanes in order, then the row segments (blockIdx.y) in order: det_reduce.h
    det_reduce(ws, blockIdx.x, blockIdx.y, gridDim.y, 64, threadIdx.x, 256,
               [&](int cl) {
                   float s = 0.f;
#pragma unroll
                   for (int i = 0; i < 16; ++i) s += reinterpret_cast<const float*>(&red[i][cl >> 2])[cl & 3];
                   return s;
               },
               [&](int cl, float s) { const int cc = blockIdx.x * 64 + cl; if (cc < C) out[cc] = accumulate ? out[cc] + s : s; });
}
// generic fallback (unaligned pointers / leading dimension): block = 64 columns x 4 row lanes
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, float* __restrict__ out, long R, int C,
                                                     long ld, int accumulate, DetWs ws) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < C)
        for (long r = (long)blockIdx.y * 4 + rl; r < R; r += (long)gridDim.y * 4) acc += in[r * ld + c];
    red[rl][cl] = acc;
    __syncthreads();
    det_reduce(ws, blockIdx.x, blockIdx.y, gridDim.y, 64, threadIdx.x, 256,
               [&](int k) { return red[0][k] + red[1][k] + red[2][k] + red[3][k]; },
               [&](int k, float s) { const int cc = blockIdx.x * 64 + k; if (cc < C) out[cc] = accumulate ? out[cc] + s : s; });
}
extern "C" int spe_colsum(const float* in, float* out, long R, int C, long ld, int accumulate, hipStream_t st) {
    if (C <= 0) return 0;
    if (R <= 0) { if (!accumulate) { hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)C, st); if (e != hipSuccess) return (int)e; } return 0; }
    const bool al4 = (C % 4 == 0) && (ld % 4 == 0) && ((((uintptr_t)in) | ((uintptr_t)out)) % 16 == 0);
    const bool wide = al4 && R <= 64 && C >= 4096;
    const DetWs ws = spe_detws();
    if (wide) {
        hipLaunchKernelGGL(colsum_wide_kernel, dim3((unsigned)((C / 4 + 255) / 256)), dim3(256), 0, st, in, out, (int)R, (long)C, ld, accumulate);
    } else if (al4) {
        const int gx = (C / 4 + 15) / 16;
        long ry = (R + 63) / 64;                       // >= 4 rows per thread
        const long cap = (2048 + gx - 1) / gx;         // ~8 blocks per CU overall
        if (ry > cap) ry = cap; if (ry < 1) ry = 1;
        DET_CHECK(ws, gx, ry, 64);
        hipLaunchKernelGGL(colsum_tall4_kernel, dim3(gx, (unsigned)ry), dim3(256), 0, st, in, out, R, C, ld, accumulate, ws);
    } else {
        long ry = (R + 255) / 256; if (ry > 64) ry = 64; if (ry < 1) ry = 1;
        DET_CHECK(ws, (C + 63) / 64, ry, 64);
        hipLaunchKernelGGL(colsum_kernel, dim3((C + 63) / 64, (unsigned)ry), dim3(256), 0, st, in, out, R, C, ld, accumulate, ws);
    }
    SPE_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// LayerScale residual  out = x + scale_b * gamma[c] * y   (reference models/cait.py:413-416;
// scale_b = per-sample DropPath keep-scale, 1 when drop_path == 0).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lsres_fwd_kernel(const float4* __restrict__ x, const float4* __restrict__ y,
                                                        const float4* __restrict__ gamma, const float* __restrict__ sample_scale,
                                                        float4* __restrict__ out, long n4, int C4, long per_sample4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 g = gamma[i % C4], a = x[i], b = y[i];
        const float sc = sample_scale ? sample_scale[i / per_sample4] : 1.f;
        float4 o;
        o.x = a.x + sc * g.x * b.x; o.y = a.y + sc * g.y * b.y; o.z = a.z + sc * g.z * b.z; o.w = a.w + sc * g.w * b.w;
        out[i] = o;
    }
}
// dy = scale_b*gamma*dout ; dgamma[c] += sum_rows scale_b*dout*y   (dx = dout is the caller's alias)
__global__ __launch_bounds__(256) void lsres_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ sample_scale,
                                                        float* __restrict__ dy, float* __restrict__ dgamma, long R, int C,
                                                        long rows_per_sample, DetWs ws) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < C) {
        const float g = gamma[c];
        for (long r = (long)blockIdx.y * 4 + rl; r < R; r += (long)gridDim.y * 4) {
            const float sc = sample_scale ? sample_scale[r / rows_per_sample] : 1.f;
            const float d = dout[r * C + c] * sc;
            acc += d * y[r * C + c];
            dy[r * C + c] = d * g;
        }
    }
    red[rl][cl] = acc;
    __syncthreads();
    det_reduce(ws, blockIdx.x, blockIdx.y, gridDim.y, 64, threadIdx.x, 256,
               [&](int k) { return red[0][k] + red[1][k] + red[2][k] + red[3][k]; },
               [&](int k, float s) { const int cc = blockIdx.x * 64 + k; if (cc < C) dgamma[cc] += s; });
}
// The same, one wave per row with 16-B accesses and per-lane column accumulators (C % 4 == 0, C <= 256 * LN_MAXV):
// NW = 16 waves per workgroup so that few workgroups (few atomics per column) still fill the SIMDs.
template <int NW>
__global__ __launch_bounds__(NW * 64) void lsres_bwd_rows_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                                 const float* __restrict__ gamma, const float* __restrict__ sample_scale,
                                                                 float* __restrict__ dy, float* __restrict__ dgamma, long R, int C,
                                                                 long rows_per_sample, DetWs ws) {
    extern __shared__ float red_raw[];                 // [NW][C + 4]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int C4 = C >> 2, ldr = C + 4;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    float4 acc[LN_MAXV], g[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        acc[i] = make_float4(0, 0, 0, 0);
        const int c = lane + 64 * i;
        g[i] = c < C4 ? g4[c] : make_float4(0, 0, 0, 0);
    }
    for (long row = (long)blockIdx.x * NW + w; row < R; row += (long)gridDim.x * NW) {
        const float sc = sample_scale ? sample_scale[row / rows_per_sample] : 1.f;
        const float4* dr = reinterpret_cast<const float4*>(dout + row * C);
        const float4* yr = reinterpret_cast<const float4*>(y + row * C);
        float4* o = reinterpret_cast<float4*>(dy + row * C);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < C4) {
                float4 d = dr[c];
                const float4 yv = yr[c];
                d.x *= sc; d.y *= sc; d.z *= sc; d.w *= sc;
                acc[i].x += d.x * yv.x; acc[i].y += d.y * yv.y; acc[i].z += d.z * yv.z; acc[i].w += d.w * yv.w;
                o[c] = make_float4(d.x * g[i].x, d.y * g[i].y, d.z * g[i].z, d.w * g[i].w);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) *reinterpret_cast<float4*>(red_raw + (long)w * ldr + 4 * c) = acc[i];
    }
    __syncthreads();
    det_reduce(ws, 0, blockIdx.x, gridDim.x, C, threadIdx.x, NW * 64,
               [&](int c) {
                   float t = 0.f;
#pragma unroll
                   for (int wv = 0; wv < NW; ++wv) t += red_raw[(long)wv * ldr + c];
                   return t;
               },
               [&](int c, float t) { dgamma[c] += t; });
}
// LayerScale backward feeding a Linear backward directly: dy = gamma * dout is never written in fp32 - the kernel emits
// what the weight / input gradient GEMMs of the preceding Linear consume, dy16 [R][C] and dy16T [C][ldt] (zero padded
// to ldt = R rounded up to 64), plus db[c] += sum_r dy (that Linear's bias gradient) and dgamma[c] += sum_r dout * y.
// A workgroup of 16 waves owns 64-row tiles: wave w converts rows w, w+16, w+32, w+48 of the tile (16-B loads, 8-B bf16
// stores) and stages them in LDS, then the tile leaves transposed as 16-B stores of 8 consecutive rows per column.
template <int MAXV>
__global__ __launch_bounds__(1024) void lsres_bwd16_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                           const float* __restrict__ gamma, unsigned short* __restrict__ dy16,
                                                           unsigned short* __restrict__ dy16T, long ldt, float* __restrict__ db,
                                                           float* __restrict__ dgamma, long R, int C, DetWs ws, int y_f16,
                                                           float p_drop, uint64_t seed, uint64_t offset, const float* __restrict__ sscale, long rps) {
    extern __shared__ unsigned short lsT[];            // [64][C + 8] bf16 tile ; reused as float [16][C + 4] x 2 at the end
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int C4 = C >> 2, ldl = C + 8;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    float4 ag[MAXV], ab[MAXV], g[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0);
        const int c = lane + 64 * i;
        g[i] = c < C4 ? g4[c] : make_float4(0, 0, 0, 0);
    }
    typedef __bf16 bf16x4r_t __attribute__((ext_vector_type(4)));
    const long ntiles = (ldt + 63) / 64;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long r0 = tile * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rl = w + 16 * k;
            const long row = r0 + rl;
            const bool rv = row < R;
            const float4* dr = reinterpret_cast<const float4*>(dout + (rv ? row : 0) * C);
            const float4* yr = reinterpret_cast<const float4*>(y + (rv ? row : 0) * C);
            const uint2* yh = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(y) + (rv ? row : 0) * C);
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = lane + 64 * i;
                if (c < C4) {
                    float4 d = make_float4(0, 0, 0, 0), yv = d;
                    if (rv) {
                        d = dr[c];
                        if (y_f16) {          // y saved as IEEE fp16 by the producing GEMM's epilogue: it only enters the gamma gradient
                            typedef _Float16 ls_h4_t __attribute__((ext_vector_type(4)));
                            const ls_h4_t h = __builtin_bit_cast(ls_h4_t, yh[c]);
                            yv = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
                        } else yv = yr[c];
                    }
                    if (sscale && rv) { const float ssb = sscale[row / rps]; d.x *= ssb; d.y *= ssb; d.z *= ssb; d.w *= ssb; }      // DropPath keep scale of the sample
                    if (p_drop > 0.f) {           // the branch was dropout(y): the mask of the forward's epilogue (element row * C + 4 c)
                        float ks[4];
                        spe_drop_scale4(seed, offset, (uint64_t)(rv ? row : 0) * (uint64_t)C + 4u * (unsigned)c, p_drop, ks);
                        d.x *= ks[0]; d.y *= ks[1]; d.z *= ks[2]; d.w *= ks[3];
                    }
                    ag[i].x += d.x * yv.x; ag[i].y += d.y * yv.y; ag[i].z += d.z * yv.z; ag[i].w += d.w * yv.w;
                    const float4 o = make_float4(d.x * g[i].x, d.y * g[i].y, d.z * g[i].z, d.w * g[i].w);
                    ab[i].x += o.x; ab[i].y += o.y; ab[i].z += o.z; ab[i].w += o.w;
                    bf16x4r_t h;
                    h[0] = (__bf16)o.x; h[1] = (__bf16)o.y; h[2] = (__bf16)o.z; h[3] = (__bf16)o.w;
                    const uint2 u = __builtin_bit_cast(uint2, h);
                    if (rv && dy16) *reinterpret_cast<uint2*>(dy16 + row * C + 4 * c) = u;
                    *reinterpret_cast<uint2*>(lsT + rl * ldl + 4 * c) = u;        // rows past R stage zeros
                }
            }
        }
        __syncthreads();
        if (dy16T) {
            for (int item = threadIdx.x; item < C * 8; item += 1024) {
                const int c = item >> 3, ch = item & 7;
                const long rr = r0 + ch * 8;
                if (rr >= ldt) continue;
                unsigned short e[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = lsT[(ch * 8 + k) * ldl + c];
                uint4 q;
                q.x = (unsigned)e[0] | ((unsigned)e[1] << 16); q.y = (unsigned)e[2] | ((unsigned)e[3] << 16);
                q.z = (unsigned)e[4] | ((unsigned)e[5] << 16); q.w = (unsigned)e[6] | ((unsigned)e[7] << 16);
                *reinterpret_cast<uint4*>(dy16T + (long)c * ldt + rr) = q;       // ldt % 64 == 0: aligned, in range
            }
        }
        __syncthreads();
    }
    // column sums of the 16 waves through LDS, then across the workgroups in a fixed order (det_reduce.h)
    float* red = reinterpret_cast<float*>(lsT);
    const int ldr = C + 4;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) {
            *reinterpret_cast<float4*>(red + (long)w * ldr + 4 * c) = ag[i];
            *reinterpret_cast<float4*>(red + (long)(16 + w) * ldr + 4 * c) = ab[i];
        }
    }
    __syncthreads();
    det_reduce(ws, 0, blockIdx.x, gridDim.x, 2 * C, threadIdx.x, 1024,
               [&](int c) {
                   const int k = c >= C, cc = k ? c - C : c;
                   float t = 0.f;
#pragma unroll
                   for (int wv = 0; wv < 16; ++wv) t += red[(long)(16 * k + wv) * ldr + cc];
                   return t;
               },
               [&](int c, float t) { float* dst = (c >= C) ? db : dgamma; if (dst) dst[c >= C ? c - C : c] += t; });
}

// C-ABI: see include/spe_hip.h (spe_layerscale_residual_bwd16).  -2: C % 4 != 0, C > 1024, ldt not a multiple of 64
// or smaller than R, misaligned pointers.
static int lsres_bwd16_launch(const float* dout, const void* y, int y_f16, const float* gamma, void* dy16, void* dy16T, long ldt,
                              float* db, float* dgamma, long R, int C, float p_drop, uint64_t seed, uint64_t offset, const float* sscale, long rps,
                              hipStream_t st);
extern "C" int spe_layerscale_residual_bwd16(const float* dout, const void* y, int y_f16, const float* gamma, void* dy16, void* dy16T, long ldt,
                                             float* db, float* dgamma, long R, int C, hipStream_t st) {
    return lsres_bwd16_launch(dout, y, y_f16, gamma, dy16, dy16T, ldt, db, dgamma, R, C, 0.f, 0, 0, nullptr, 1, st);
}
// C-ABI: see include/spe_hip.h.  The same backward when the forward was res + s_b * gamma * dropout(y) (spe_gemm_bf16nt_exd).
extern "C" int spe_layerscale_residual_bwd16d(const float* dout, const void* y, int y_f16, const float* gamma, void* dy16, void* dy16T, long ldt,
                                              float* db, float* dgamma, long R, int C, float p_drop, uint64_t seed, uint64_t offset,
                                              const float* sample_scale, long rows_per_sample, hipStream_t st) {
    if (p_drop < 0.f || p_drop >= 1.f || (sample_scale && rows_per_sample <= 0)) return -2;
    return lsres_bwd16_launch(dout, y, y_f16, gamma, dy16, dy16T, ldt, db, dgamma, R, C, p_drop, seed, offset, sample_scale,
                              sample_scale ? rows_per_sample : 1, st);
}
static int lsres_bwd16_launch(const float* dout, const void* y, int y_f16, const float* gamma, void* dy16, void* dy16T, long ldt,
                              float* db, float* dgamma, long R, int C, float p_drop, uint64_t seed, uint64_t offset, const float* sscale, long rps,
                              hipStream_t st) {
    if (R <= 0) return 0;
    if ((C & 3) || C > 256 * LN_MAXV || (ldt & 63) || ldt < R) return -2;
    if ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) |
         reinterpret_cast<uintptr_t>(dy16) | reinterpret_cast<uintptr_t>(dy16T)) & 15) return -2;
    const int tile_bytes = 64 * (C + 8) * 2, red_bytes = 32 * (C + 4) * 4;
    const int smem = tile_bytes > red_bytes ? tile_bytes : red_bytes;
    static bool attr = false;
    if (!attr) {
        constexpr int mx = 64 * (256 * LN_MAXV + 8) * 2 > 32 * (256 * LN_MAXV + 4) * 4 ? 64 * (256 * LN_MAXV + 8) * 2 : 32 * (256 * LN_MAXV + 4) * 4;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lsres_bwd16_kernel<LN_MAXV>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lsres_bwd16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    long nb = (ldt + 63) / 64; if (nb > 256) nb = 256;
    DetWs ws = spe_detws();
    const DetDeferSeg sg[2] = {{dgamma, C}, {db, C}};
    float* region = det_defer_try(1, nb, 2 * C, 2, sg, st);        // deferred: dgamma / db += totals at the next flush
    if (region) ws.defer = region; else DET_CHECK(ws, 1, nb, 2 * C);
    // (2 float4 per lane up to C = 512: fewer live registers per wave than the 4 the widest rows need)
    if (C <= 512) hipLaunchKernelGGL(lsres_bwd16_kernel<2>, dim3((unsigned)nb), dim3(1024), smem, st, dout, reinterpret_cast<const float*>(y), gamma,
                                     reinterpret_cast<unsigned short*>(dy16), reinterpret_cast<unsigned short*>(dy16T), ldt, db, dgamma, R, C, ws, y_f16,
                                     p_drop, seed, offset, sscale, rps);
    else hipLaunchKernelGGL(lsres_bwd16_kernel<LN_MAXV>, dim3((unsigned)nb), dim3(1024), smem, st, dout, reinterpret_cast<const float*>(y), gamma,
                            reinterpret_cast<unsigned short*>(dy16), reinterpret_cast<unsigned short*>(dy16T), ldt, db, dgamma, R, C, ws, y_f16,
                            p_drop, seed, offset, sscale, rps);
    if (region) det_defer_commit(region, 1, nb, 2 * C, 2, sg, 1);
    SPE_CHECK_LAUNCH();
    return 0;
}

extern "C" int spe_layerscale_residual_fwd(const float* x, const float* y, const float* gamma, const float* sample_scale,
                                           float* out, long R, int C, long rows_per_sample, hipStream_t st) {
    if (R <= 0) return 0;
    if (C & 3) return -2;
    const long n4 = R * C / 4;
    long nb = (n4 + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(lsres_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const float4*)x, (const float4*)y,
                       (const float4*)gamma, sample_scale, (float4*)out, n4, C / 4, rows_per_sample * C / 4);
    SPE_CHECK_LAUNCH();
    return 0;
}
extern "C" int spe_layerscale_residual_bwd(const float* dout, const float* y, const float* gamma, const float* sample_scale,
                                           float* dy, float* dgamma, long R, int C, long rows_per_sample, hipStream_t st) {
    if (R <= 0) return 0;
    const DetWs ws = spe_detws();
    if ((C & 3) == 0 && C <= 256 * LN_MAXV && ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(y) |
                                                           reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(gamma)) & 15) == 0) {
        long nb = (R + 15) / 16; if (nb > 128) nb = 128;
        DET_CHECK(ws, 1, nb, C);
        hipLaunchKernelGGL(lsres_bwd_rows_kernel<16>, dim3((unsigned)nb), dim3(1024), 16 * (C + 4) * (int)sizeof(float), st, dout, y,
                           gamma, sample_scale, dy, dgamma, R, C, rows_per_sample, ws);
        SPE_CHECK_LAUNCH();
        return 0;
    }
    long ry = (R + 63) / 64; if (ry > 256) ry = 256; if (ry < 1) ry = 1;
    DET_CHECK(ws, (C + 63) / 64, ry, 64);
    hipLaunchKernelGGL(lsres_bwd_kernel, dim3((C + 63) / 64, (unsigned)ry), dim3(256), 0, st, dout, y, gamma, sample_scale,
                       dy, dgamma, R, C, rows_per_sample, ws);
    SPE_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// Activation backward + dropout (elementwise, grid-stride, 16 B per lane).
// mode 1: dx = dy * (out > 0)            (ReLU, `aux` = forward output)
// mode 2: dx = dy * gelu'(aux)           (exact-erf GELU, `aux` = pre-activation)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ aux,
                                                      float4* __restrict__ dx, long n4, int mode) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 d = dy[i], a = aux[i];
        float dv[4] = {d.x, d.y, d.z, d.w}, av[4] = {a.x, a.y, a.z, a.w}, o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (mode == 1) o[j] = av[j] > 0.f ? dv[j] : 0.f;
            else {
                const float h = av[j];
                const float cdf = 0.5f * (1.f + spe_erff(h * 0.70710678118654752f));
                const float pdf = 0.3989422804014327f * __expf(-0.5f * h * h);
                o[j] = dv[j] * (cdf + h * pdf);
            }
        }
        dx[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}
extern "C" int spe_act_bwd(const float* dy, const float* aux, float* dx, long n, int mode, hipStream_t st) {
    if (n <= 0) return 0;
    if (n & 3) return -2;
    long nb = (n / 4 + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const float4*)dy, (const float4*)aux, (float4*)dx, n / 4, mode);
    SPE_CHECK_LAUNCH();
    return 0;
}

// y = x * keepscale(idx); the same call with x := dy is the backward.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p,
                                                      uint64_t seed, uint64_t offset) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        y[i] = x[i] * spe_drop_scale(seed, offset, (uint64_t)i, p);
}
extern "C" int spe_dropout(const float* x, float* y, long n, float p, uint64_t seed, uint64_t offset, hipStream_t st) {
    if (n <= 0) return 0;
    long nb = (n + 255) / 256; if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, y, n, p, seed, offset);
    SPE_CHECK_LAUNCH();
    return 0;
}
