// Micro-benchmark: issue rate of v_pk_fma_f32 (VGPR / SGPR-broadcast operands) vs v_fma_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* w, int iters) {
    f32x2_t acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x2_t){(float)threadIdx.x, (float)i};
    float s0 = w[0], s1 = w[1];
    f32x2_t v0 = {w[threadIdx.x & 3], w[2]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) acc[i] = __builtin_elementwise_fma(acc[i], (f32x2_t){s0, s0}, (f32x2_t){s1, s1});      // pk, SGPR broadcast
                if (MODE == 1) acc[i] = __builtin_elementwise_fma(acc[i], v0, v0);                                      // pk, all VGPR
                if (MODE == 2) { acc[i][0] = fmaf(acc[i][0], s0, s1); acc[i][1] = fmaf(acc[i][1], s0, s1); }       // 2 scalar fma
                if (MODE == 3) acc[i] = __builtin_elementwise_fma(acc[i], (f32x2_t){v0[1], v0[1]}, acc[(i + 1) & 15]); // pk, op_sel hi broadcast
            }
        asm volatile("" : "+v"(acc[0]));
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* out, float* w) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, nblk = 256 * 2;   // 2 WG/CU -> 2 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, out, w, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, out, w, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double insts_per_simd = 2.0 * iters * 64 * (MODE == 2 ? 2 : 1);   // waves per SIMD * instrs
    printf("%-28s %.3f ms  -> %.2f cycles/instr @2.4GHz (fp32 FMA lanes: %.1f TFLOP/s)\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_simd,
           2.0 * 2 * 64 * iters * 64.0 * nblk * 4 / (ms * 1e-3) / 1e12);
}
int main() {
    float *out, *w; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&w, 64);
    float hw[4] = {1.0001f, 0.5f, 0.25f, 0.125f}; hipMemcpy(w, hw, 16, hipMemcpyHostToDevice);
    run<0>("pk_fma sgpr-broadcast", out, w);
    run<1>("pk_fma vgpr", out, w);
    run<2>("2x v_fma_f32", out, w);
    run<3>("pk_fma op_sel-hi broadcast", out, w);
    return 0;
}
