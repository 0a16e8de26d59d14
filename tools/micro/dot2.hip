// Micro-benchmark: issue rate of v_dot2_f32_f16 / v_dot2_f32_bf16 vs v_fma_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* w, int iters) {
    float acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = (float)(threadIdx.x + i);
    const float s0 = w[0], s1 = w[1];
    h2 hw = {(_Float16)w[0], (_Float16)w[1]};
    h2 hx = {(_Float16)w[threadIdx.x & 3], (_Float16)w[2]};
    b2 bw = {(__bf16)w[0], (__bf16)w[1]};
    b2 bx = {(__bf16)w[threadIdx.x & 3], (__bf16)w[2]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (MODE == 0) acc[i] = fmaf(acc[i], s0, s1);
                if (MODE == 1) acc[i] = __builtin_amdgcn_fdot2(hx, hw, acc[i], false);
                if (MODE == 2) acc[i] = __builtin_amdgcn_fdot2_f32_bf16(bx, bw, acc[i], false);
            }
        asm volatile("" : "+v"(acc[0]));
    }
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* out, float* w) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 2000, nblk = 256 * 2;
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, out, w, 10);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, out, w, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double insts_per_simd = 2.0 * iters * 128;
    printf("%-28s %.3f ms  -> %.2f cycles/instr @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
}
int main() {
    float *out, *w; (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&w, 64);
    float hw[4] = {1.0001f, 0.5f, 0.25f, 0.125f}; (void)hipMemcpy(w, hw, 16, hipMemcpyHostToDevice);
    run<0>("v_fma_f32", out, w);
    run<1>("v_dot2_f32_f16", out, w);
    run<2>("v_dot2_f32_bf16", out, w);
    return 0;
}
