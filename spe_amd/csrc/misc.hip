// Small gather / elementwise kernels around the GEMMs.
#include "common.h"

// Patch gather for the 16x16/stride-16 patch-embed convolution (reference models/cait.py:518-528,
// timm PatchEmbed.proj = Conv2d(3,C,16,16)): img[B,Cin,Hi,Wi] -> cols[B*h*w, Cin*P*P] with the
// column order (c, py, px) of the flattened conv weight, so the conv becomes one GEMM.
// Each lane moves one float4 = 4 horizontally adjacent pixels.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, float* __restrict__ cols, int B, int Cin,
                                                       int Hi, int Wi, int P, int h, int w) {
    const int P4 = P >> 2;
    const long per_row = (long)Cin * P * P4;                 // float4 per output row
    const long total = (long)B * h * w * per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / per_row; int r = (int)(i % per_row);
        const int c = r / (P * P4); r %= P * P4;
        const int py = r / P4, px4 = r % P4;
        const int b = (int)(row / (h * w)); const int pr = (int)(row % (h * w));
        const int ph = pr / w, pw = pr % w;
        const float* src = img + (((long)b * Cin + c) * Hi + (ph * P + py)) * Wi + pw * P + px4 * 4;
        float4 v;
        if ((Wi & 3) == 0) v = *reinterpret_cast<const float4*>(src);
        else { v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3]; }
        reinterpret_cast<float4*>(cols)[i] = v;
    }
}
extern "C" int spe_patchify(const float* img, float* cols, int B, int Cin, int Hi, int Wi, int P, hipStream_t st) {
    if (P & 3) return -2;
    const int h = Hi / P, w = Wi / P;
    const long total = (long)B * h * w * Cin * P * (P / 4);
    if (total <= 0) return 0;
    long nb = (total + 255) / 256; if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)nb), dim3(256), 0, st, img, cols, B, Cin, Hi, Wi, P, h, w);
    SPE_CHECK_LAUNCH();
    return 0;
}

// out[r][c] = a[r][c] + b[(r % rb)][c]   (adds a broadcast table: pos-embed / bias rows), float4.
__global__ __launch_bounds__(256) void add_rows_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                       float4* __restrict__ out, long n4, long period4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 x = a[i], y = b[i % period4];
        out[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}
extern "C" int spe_add_rows(const float* a, const float* b, float* out, long n, long period, hipStream_t st) {
    if (n <= 0) return 0;
    if ((n & 3) || (period & 3)) return -2;
    long nb = (n / 4 + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const float4*)a, (const float4*)b, (float4*)out,
                       n / 4, period / 4);
    SPE_CHECK_LAUNCH();
    return 0;
}

extern "C" int spe_abi_version(void) { return 1; }
