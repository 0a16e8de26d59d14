// Does the destination register file of a matrix instruction change its issue interval?  One wave per SIMD (256 workgroups x 4 waves) or two
// (x 8 waves), 8 independent accumulators, 16x16x32 f16 / 16x16x16 bf16 / 4x4x1 f32:
//   v  : accumulators in ordinary VGPRs   (asm "+v")
//   a  : accumulators in AccVGPRs         (asm "+a")
//   b  : the builtin, compiler's choice
// hipcc --offload-arch=gfx950 -O3 mfma_form.hip -o /tmp/mfma_form && /tmp/mfma_form
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int KIND, int FORM, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void k(float* out, unsigned long long* cyc, int iters) {
    f16x8_t a8, b8; s16x4_t sa, sb;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.001f * (threadIdx.x + i)); b8[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    for (int i = 0; i < 4; ++i) { sa[i] = (short)(0x3c00 + threadIdx.x + i); sb[i] = (short)(0x3c00 + i); }
    float fa = 0.001f * threadIdx.x, fb = 0.5f;
    f32x4_t c[8];
    for (int j = 0; j < 8; ++j) c[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (FORM == 0) {
                if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a8), "v"(b8));
                if (KIND == 1) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(sa), "v"(sb));
                if (KIND == 2) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(c[j]) : "v"(fa), "v"(fb));
            } else if constexpr (FORM == 1) {
                if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c[j]) : "v"(a8), "v"(b8));
                if (KIND == 1) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+a"(c[j]) : "v"(sa), "v"(sb));
                if (KIND == 2) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(c[j]) : "v"(fa), "v"(fb));
            } else {
                if (KIND == 0) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[j], 0, 0, 0);
                if (KIND == 1) c[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, c[j], 0, 0, 0);
                if (KIND == 2) c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(fa, fb, c[j], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * 64 * NW + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND, int FORM, int NW> void run(const char* name, float* out, unsigned long long* cyc, double flop) {
    const int iters = 4000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<KIND, FORM, NW>), dim3(256), dim3(64 * NW), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<KIND, FORM, NW>), dim3(256), dim3(64 * NW), 0, 0, out, cyc, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 8.0 * iters;
    printf("%-22s form %c waves/SIMD %d: %7.2f clk per instr of one wave, %7.2f ns per instr per SIMD, chip %8.1f TFLOP/s\n", name, "vab"[FORM], NW / 4,
           (double)h / n, ms * 1e6 / (n * (NW / 4)), flop * n * 256 * NW / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 8 << 20); (void)hipMalloc(&cyc, 8);
#define ALL(KIND, NAME, FLOP) \
    run<KIND, 0, 4>(NAME, out, cyc, FLOP); run<KIND, 1, 4>(NAME, out, cyc, FLOP); run<KIND, 2, 4>(NAME, out, cyc, FLOP); \
    run<KIND, 0, 8>(NAME, out, cyc, FLOP); run<KIND, 1, 8>(NAME, out, cyc, FLOP); run<KIND, 2, 8>(NAME, out, cyc, FLOP);
    ALL(0, "mfma_f32_16x16x32_f16", 16384.0)
    ALL(1, "mfma_f32_16x16x16_bf16", 8192.0)
    ALL(2, "mfma_f32_4x4x1_16b_f32", 512.0)
    return 0;
}
