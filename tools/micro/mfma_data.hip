// What bounds a dense stream of v_mfma_f32_16x16x32_f16 from ONE wave per SIMD (or two) when the operands are real?  Round 6's A-resident GEMM
// (profiles/r06_gemm_ares.txt) issued them at ~22 ns per instruction and SIMD; mfma_form.hip measures 9 ns on ONE operand pair.  Variants:
//   NA x NB : distinct A / B fragments in registers (A_i x B_j -> accumulator (i, j), the register tiling of a GEMM main loop), all accumulators in AccVGPRs
//   DATA    : 0 = zeros, 1 = small smooth values (mfma_form's), 2 = pseudo-random fp16 in [-2, 2) (every bit toggles between consecutive instructions)
//   LONG    : 20x more iterations (a launch of ~10 ms instead of ~0.5 ms: does the clock sag under sustained load?)
// Prints ns per instruction per SIMD (wall) and s_memtime ticks per instruction of one wave; chip rate = all FLOP / wall.
// hipcc --offload-arch=gfx950 -O3 mfma_data.hip -o /tmp/mfma_data && /tmp/mfma_data
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ unsigned rnd_f16pair(unsigned& s) {
    // two fp16 with random sign / mantissa and exponent 12..15 (|x| in [2^-3, 2)): finite, all mantissa bits random
    const unsigned r = rnd(s) ^ (rnd(s) >> 7);
    const unsigned lo = (r & 0x83FFu) | ((12u + ((r >> 10) & 3u)) << 10);
    const unsigned hi = ((r >> 16) & 0x83FFu) | ((12u + ((r >> 26) & 3u)) << 10);
    return lo | (hi << 16);
}
template <int NA, int NB, int DATA, int NW>
__global__ __launch_bounds__(64 * NW, 1) void k(float* out, unsigned long long* cyc, int iters) {
    u32x4_t a[NA], b[NB];
    unsigned s = 12345u + 977u * threadIdx.x + 131071u * blockIdx.x;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[i][j] = DATA == 0 ? 0u : DATA == 1 ? 0x14001400u + (threadIdx.x << 2) + i : rnd_f16pair(s);
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) b[i][j] = DATA == 0 ? 0u : DATA == 1 ? 0x18001800u + (threadIdx.x << 1) + i : rnd_f16pair(s);
    f32x4_t c[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) c[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c[i][j]) : "v"(a[i]), "v"(b[j]));
        if (DATA == 2) {            // keep the accumulators finite over a long run: nothing (random signs: a random walk of ~sqrt(32 iters) * 4), cheap enough
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) r += c[i][j][0] + c[i][j][1] + c[i][j][2] + c[i][j][3];
    out[blockIdx.x * 64 * NW + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NA, int NB, int DATA, int NW> void run(float* out, unsigned long long* cyc, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NA, NB, DATA, NW>), dim3(256), dim3(64 * NW), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NA, NB, DATA, NW>), dim3(256), dim3(64 * NW), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)NA * NB * iters;
    static const char* dn[3] = {"zeros ", "smooth", "random"};
    printf("A x B = %d x %d  data %s  waves/SIMD %d  launch %8.3f ms: %6.2f ns per instr per SIMD, %6.2f ticks per instr of one wave, chip %7.1f TFLOP/s\n", NA, NB, dn[DATA], NW / 4, ms,
           ms * 1e6 / (n * (NW / 4)), (double)h / n, 16384.0 * n * 256 * NW / (ms * 1e-3) / 1e12);
    fflush(stdout);
}
int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 8 << 20); (void)hipMalloc(&cyc, 8);
    const int it = 2000;
#define SET(NA, NB, NW) run<NA, NB, 0, NW>(out, cyc, it); run<NA, NB, 1, NW>(out, cyc, it); run<NA, NB, 2, NW>(out, cyc, it);
    SET(1, 1, 4) SET(1, 8, 4) SET(4, 8, 4) SET(4, 4, 4) SET(2, 4, 4)
    SET(1, 1, 8) SET(4, 4, 8) SET(2, 4, 8)
    printf("-- sustained (20x longer launches)\n");
    run<4, 8, 0, 4>(out, cyc, 20 * it); run<4, 8, 2, 4>(out, cyc, 20 * it); run<4, 4, 2, 8>(out, cyc, 20 * it);
    printf("-- back to back: the same launch five times (does the first differ?)\n");
    for (int r = 0; r < 5; ++r) run<4, 8, 2, 4>(out, cyc, it);
    return 0;
}
