// Optimiser step of the SPE training loop on flat buffers: reference
// engine.py:161-165 (`clip_grad_norm_(model.parameters(), 0.1)` then `optimizer.step()`) with the AdamW parameter
// groups of main.py:177-191.  Gradients already live in the flat all-reduce buckets (spe_amd/dp.py); parameters and
// the two moment buffers use the same layout, so the whole step is two launches per bucket - a squared-norm pass and
// one fused clip + decoupled-weight-decay + Adam update - instead of ~20 multi-tensor launches over ~600 tensors.
#include "common.h"

// partials[b] = sum of g[i]^2 over the b-th contiguous chunk (fixed order: deterministic)
__global__ __launch_bounds__(256) void sqnorm_partials_kernel(const float* __restrict__ g, long n, float* __restrict__ partials) {
    __shared__ float red[16];
    const long per = ((n + gridDim.x - 1) / gridDim.x + 3) & ~3L;
    long beg = (long)blockIdx.x * per; if (beg > n) beg = n;       // chunks past the end are empty (end - beg must not go negative)
    long end = beg + per; if (end > n) end = n;
    float acc = 0.f;
    const bool al = (reinterpret_cast<uintptr_t>(g) & 15) == 0;
    if (al) {
        for (long i = beg + threadIdx.x * 4; i + 3 < end; i += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        const long tail = beg + ((end - beg) & ~3L);
        for (long i = tail + threadIdx.x; i < end; i += 256) acc += g[i] * g[i];
    } else {
        for (long i = beg + threadIdx.x; i < end; i += 256) acc += g[i] * g[i];
    }
    acc = spe_block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

extern "C" int spe_sqnorm_partials(const float* g, long n, float* partials, int nblocks, hipStream_t st) {
    if (nblocks <= 0) return 0;
    hipLaunchKernelGGL(sqnorm_partials_kernel, dim3(nblocks), dim3(256), 0, st, g, n, partials);
    SPE_CHECK_LAUNCH();
    return 0;
}

#define ADAMW_MAXSEG 64
struct AdamwArgs {
    float* p; float* g; float* m; float* v; long n;
    const long* seg_end; const float* seg_lr; const float* seg_wd; int nseg;
    float beta1, beta2, eps, bc1, bc2sqrt;
    const float* partials; int npartials; float max_norm; int write_grad; float gscale;
};

// torch.optim.AdamW (amsgrad = False, maximize = False):
//   p *= 1 - lr*wd ; m = lerp(m, g, 1-beta1) ; v = beta2*v + (1-beta2) g^2 ;
//   p -= (lr / (1-beta1^t)) * m / (sqrt(v)/sqrt(1-beta2^t) + eps)
// with g first scaled by clip = min(1, max_norm / (||g||_2 + 1e-6)) over ALL parameters (torch clip_grad_norm_).
// Element i belongs to the segment s with seg_end[s-1] <= i < seg_end[s] (parameter group: lr, weight decay).
__global__ __launch_bounds__(256) void adamw_flat_kernel(AdamwArgs a) {
    __shared__ float red[16];
    __shared__ long s_end[ADAMW_MAXSEG];
    __shared__ float s_lr[ADAMW_MAXSEG], s_wd[ADAMW_MAXSEG];
    for (int i = threadIdx.x; i < a.nseg; i += 256) { s_end[i] = a.seg_end[i]; s_lr[i] = a.seg_lr[i]; s_wd[i] = a.seg_wd[i]; }
    float clip = 1.f;
    if (a.max_norm > 0.f) {
        float acc = 0.f;
        for (int i = threadIdx.x; i < a.npartials; i += 256) acc += a.partials[i];
        const float tot = a.gscale * sqrtf(spe_block_sum(acc, red));
        clip = fminf(a.max_norm / (tot + 1e-6f), 1.f);
    }
    clip *= a.gscale;
    __syncthreads();
    // grid-stride over float4 chunks: the prologue above (segment table, clip factor from the norm partials) is paid once
    // per workgroup, not once per 1024 elements (26 K workgroups for the 27 M parameters of cfg2: 0.41 -> 0.2x ms)
    for (long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i0 < a.n; i0 += (long)gridDim.x * 1024) {
    int lo = 0, hi = a.nseg - 1;                         // first segment whose end is > i0
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_end[mid] > i0) hi = mid; else lo = mid + 1; }
    int seg = lo;
    float pv[4], gv[4], mv[4], vv[4];
    const bool full = i0 + 3 < a.n;
    if (full) {
        const float4 P = *reinterpret_cast<const float4*>(a.p + i0), G = *reinterpret_cast<const float4*>(a.g + i0);
        const float4 M = *reinterpret_cast<const float4*>(a.m + i0), V = *reinterpret_cast<const float4*>(a.v + i0);
        pv[0] = P.x; pv[1] = P.y; pv[2] = P.z; pv[3] = P.w; gv[0] = G.x; gv[1] = G.y; gv[2] = G.z; gv[3] = G.w;
        mv[0] = M.x; mv[1] = M.y; mv[2] = M.z; mv[3] = M.w; vv[0] = V.x; vv[1] = V.y; vv[2] = V.z; vv[3] = V.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = i0 + j < a.n;
            pv[j] = in ? a.p[i0 + j] : 0.f; gv[j] = in ? a.g[i0 + j] : 0.f; mv[j] = in ? a.m[i0 + j] : 0.f; vv[j] = in ? a.v[i0 + j] : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        while (seg < a.nseg - 1 && s_end[seg] <= i0 + j) ++seg;
        const float lr = s_lr[seg], wd = s_wd[seg];
        const float g = gv[j] * clip;
        float p = pv[j] * (1.f - lr * wd);
        const float m = mv[j] + (g - mv[j]) * (1.f - a.beta1);
        const float v = a.beta2 * vv[j] + (1.f - a.beta2) * g * g;
        const float denom = sqrtf(v) / a.bc2sqrt + a.eps;
        p -= (lr / a.bc1) * (m / denom);
        pv[j] = p; gv[j] = g; mv[j] = m; vv[j] = v;
    }
    if (full) {
        *reinterpret_cast<float4*>(a.p + i0) = make_float4(pv[0], pv[1], pv[2], pv[3]);
        *reinterpret_cast<float4*>(a.m + i0) = make_float4(mv[0], mv[1], mv[2], mv[3]);
        *reinterpret_cast<float4*>(a.v + i0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (a.write_grad) *reinterpret_cast<float4*>(a.g + i0) = make_float4(gv[0], gv[1], gv[2], gv[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j < a.n) { a.p[i0 + j] = pv[j]; a.m[i0 + j] = mv[j]; a.v[i0 + j] = vv[j]; if (a.write_grad) a.g[i0 + j] = gv[j]; }
    }
    }
}

// C-ABI: see include/spe_hip.h (spe_adamw_flat).  All flat buffers 16-B aligned, n elements.
extern "C" int spe_adamw_flat(float* p, float* g, float* m, float* v, long n, const long* seg_end, const float* seg_lr,
                              const float* seg_wd, int nseg, float beta1, float beta2, float eps, float bias_c1, float bias_c2,
                              const float* partials, int npartials, float max_norm, int write_grad, float grad_scale,
                              hipStream_t st) {
    if (n <= 0) return 0;
    if (nseg < 1 || nseg > ADAMW_MAXSEG) return -2;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) return -2;
    AdamwArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.seg_end = seg_end; a.seg_lr = seg_lr; a.seg_wd = seg_wd; a.nseg = nseg;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.bc1 = bias_c1; a.bc2sqrt = sqrtf(bias_c2);
    a.partials = partials; a.npartials = npartials; a.max_norm = max_norm; a.write_grad = write_grad; a.gscale = grad_scale;
    long nb = (n + 1023) / 1024; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)nb), dim3(256), 0, st, a);
    SPE_CHECK_LAUNCH();
    return 0;
}
