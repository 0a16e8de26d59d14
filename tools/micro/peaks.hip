// Calibration micro-benchmarks for the roofline peaks ("calibrate both peaks with a
// micro-benchmark on the box"): HBM read / write / copy with plain grid-stride float4 kernels, and the dense bf16 MFMA
// rate of v_mfma_f32_16x16x32_bf16 from registers (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void rd(const float4* __restrict__ x, float* out, long n4) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) { const float4 v = x[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void rd8(const float2* __restrict__ x, float* out, long n2) {   // 8 B per lane, like the score blocks
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) { const float2 v = x[i]; acc += v.x + v.y; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void wr(float4* __restrict__ y, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) y[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void cp(const float4* __restrict__ x, float4* __restrict__ y, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) y[i] = x[i];
}
__global__ __launch_bounds__(256) void mfma(float* out, int iters) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    f32x4_t c[8];
    for (int j = 0; j < 8; ++j) c[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> float timeit(F f, int n) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < n; ++i) f();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / n;
}
int main() {
    const long bytes = 2L << 30, n4 = bytes / 16;
    float4 *x, *y; float* out;
    (void)hipMalloc(&x, bytes); (void)hipMalloc(&y, bytes); (void)hipMalloc(&out, 4 << 20);
    (void)hipMemset(x, 0, bytes);
    for (int g : {2048, 8192, 32768}) {
        float t = timeit([&] { hipLaunchKernelGGL(rd, dim3(g), dim3(256), 0, 0, x, out, n4); }, 5);
        printf("HBM read   2 GiB grid %6d: %7.0f GB/s\n", g, bytes / (t * 1e-3) / 1e9);
        t = timeit([&] { hipLaunchKernelGGL(rd8, dim3(g), dim3(256), 0, 0, (const float2*)x, out, n4 * 2); }, 5);
        printf("HBM read8  2 GiB grid %6d: %7.0f GB/s (8 B per lane)\n", g, bytes / (t * 1e-3) / 1e9);
        t = timeit([&] { hipLaunchKernelGGL(wr, dim3(g), dim3(256), 0, 0, y, n4); }, 5);
        printf("HBM write  2 GiB grid %6d: %7.0f GB/s\n", g, bytes / (t * 1e-3) / 1e9);
        t = timeit([&] { hipLaunchKernelGGL(cp, dim3(g), dim3(256), 0, 0, x, y, n4); }, 5);
        printf("HBM copy   2 GiB grid %6d: %7.0f GB/s (read + write)\n", g, 2.0 * bytes / (t * 1e-3) / 1e9);
    }
    for (int wgs : {256, 512, 1024}) {
        const int iters = 4000;
        float t = timeit([&] { hipLaunchKernelGGL(mfma, dim3(wgs), dim3(256), 0, 0, out, iters); }, 3);
        const double flop = 2.0 * 16 * 16 * 32 * 8.0 * iters * 4 /*waves*/ * wgs;
        printf("MFMA bf16 16x16x32, %4d workgroups: %7.1f TFLOP/s\n", wgs, flop / (t * 1e-3) / 1e12);
    }
    return 0;
}
