// layout probe of v_mfma_f32_4x4x1_16b_f32: A = 100*lane, B = lane  ->  which (lane pair) products land in which register
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    f32x4_t d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1000 * lane), (float)(lane + 1), d, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = d[i];
}
int main() {
    float* o; hipMalloc(&o, 256 * 4); k<<<1, 64>>>(o); float h[256]; hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        const float v = h[l * 4 + i];
        // hypothesis A: d[i] at lane l = A(lane 4*(l/4)+i) * B(lane l)
        if (v != 1000.f * (4 * (l / 4) + i) * (l + 1)) okA = 0;
        if (v != 1000.f * l * (4 * (l / 4) + i + 1)) okB = 0;
    }
    printf("hypA (reg i <- A of lane 4b+i, own B) %d ; hypB (own A, B of lane 4b+i) %d\n", okA, okB);
    for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
