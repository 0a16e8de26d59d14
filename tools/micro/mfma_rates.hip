// Issue cost of the matrix instructions the talking-heads kernels use, one wave per SIMD (256 workgroups x 4 waves), 8 independent
// accumulators: cycles per instruction per SIMD from s_memtime, and the chip rate.  hipcc --offload-arch=gfx950 -O3 mfma_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
    f16x8_t a8, b8; f16x4_t a4, b4; s16x4_t sa, sb;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.001f * (threadIdx.x + i)); b8[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; sa[i] = (short)(0x3c00 + threadIdx.x + i); sb[i] = (short)(0x3c00 + i); }
    float fa = 0.001f * threadIdx.x, fb = 0.5f;
    f32x4_t c[8];
    for (int j = 0; j < 8; ++j) c[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a8), "v"(b8));
            if (KIND == 1) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a4), "v"(b4));
            if (KIND == 2) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(sa), "v"(sb));
            if (KIND == 3) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a4), "v"(b4));
            if (KIND == 4) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(c[j]) : "v"(fa), "v"(fb));
            if (KIND == 5) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[j]) : "v"(fa), "v"(fb));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 4000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %7.2f clk/instr (s_memtime)  %8.3f ns/instr/SIMD wall\n", name, (double)h / (8.0 * iters), ms * 1e6 / (8.0 * iters));
}
__global__ __launch_bounds__(64) void dma_probe(const unsigned* __restrict__ src, unsigned* out, unsigned off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned* w = reinterpret_cast<unsigned*>(sm);
    for (unsigned i = threadIdx.x; i < 160 * 256; i += 64) w[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sm);
    const void* g = src + threadIdx.x * 4;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(g), "s"(lds0 + off) : "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = w[off / 4 + threadIdx.x * 4 + i];
}
int main() {
    {
        unsigned *src, *out; (void)hipMalloc(&src, 1024); (void)hipMalloc(&out, 1024);
        unsigned h[256]; for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
        (void)hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (unsigned off : {0u, 32768u, 65536u, 100u * 1024u, 159u * 1024u}) {
            hipLaunchKernelGGL(dma_probe, dim3(1), dim3(64), 160 * 1024, 0, src, out, off);
            unsigned r[256]; (void)hipMemcpy(r, out, 1024, hipMemcpyDeviceToHost);
            int bad = 0; for (int i = 0; i < 256; ++i) bad += (r[i] != h[i]);
            printf("LDS-DMA at LDS offset %6u: %d of 256 words wrong (first %u)\n", off, bad, r[0]);
        }
    }
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 4 << 20); (void)hipMalloc(&cyc, 8);
    run<0>("mfma_f32_16x16x32_f16", out, cyc);
    run<1>("mfma_f32_16x16x16_f16", out, cyc);
    run<2>("mfma_f32_16x16x16_bf16_1k", out, cyc);
    run<3>("mfma_f32_4x4x4_16b_f16", out, cyc);
    run<4>("mfma_f32_4x4x1_16b_f32", out, cyc);
    run<5>("mfma_f32_16x16x4_f32", out, cyc);
    return 0;
}
