// is a chain of v_mfma_f32_4x4x1_16b_f32 bit-identical to the fmaf chain it replaces?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ void k(const float* w, const float* x, const float* c, float* om, float* of) {
    const int lane = threadIdx.x;
    f32x4_t d = {c[0], c[1], c[2], c[3]};
    float f[4] = {c[0], c[1], c[2], c[3]};
    for (int h = 0; h < 8; ++h) {
        const float a = w[(lane & 3) * 8 + h];           // W[i = lane&3][h]
        const float b = x[h * 64 + lane];                // own value of head h
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
        for (int i = 0; i < 4; ++i) f[i] = fmaf(w[i * 8 + h], b, f[i]);
    }
    for (int i = 0; i < 4; ++i) { om[lane * 4 + i] = d[i]; of[lane * 4 + i] = f[i]; }
}
int main() {
    float hw[32], hx[512], hc[4];
    srand(1);
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 60.f;
    for (auto& v : hc) v = (rand() / (float)RAND_MAX - 0.5f) * 20.f;
    float *w, *x, *c, *om, *of;
    hipMalloc(&w, sizeof hw); hipMalloc(&x, sizeof hx); hipMalloc(&c, sizeof hc); hipMalloc(&om, 1024); hipMalloc(&of, 1024);
    hipMemcpy(w, hw, sizeof hw, hipMemcpyHostToDevice); hipMemcpy(x, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(c, hc, sizeof hc, hipMemcpyHostToDevice);
    k<<<1, 64>>>(w, x, c, om, of);
    float a[256], b[256]; hipMemcpy(a, om, 1024, hipMemcpyDeviceToHost); hipMemcpy(b, of, 1024, hipMemcpyDeviceToHost);
    int same = 0; double worst = 0;
    for (int i = 0; i < 256; ++i) { if (a[i] == b[i]) ++same; double r = fabs((double)a[i] - b[i]) / (fabs((double)b[i]) + 1e-30); if (r > worst) worst = r; }
    // fp64 reference of the chain
    double worst_m = 0, worst_f = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        double r = hc[i]; for (int h = 0; h < 8; ++h) r += (double)hw[i * 8 + h] * hx[h * 64 + l];
        worst_m = fmax(worst_m, fabs(a[l * 4 + i] - r)); worst_f = fmax(worst_f, fabs(b[l * 4 + i] - r));
    }
    printf("bit-identical %d / 256, worst relative difference %.3g ; abs err vs fp64: mfma %.3g fmaf %.3g\n", same, worst, worst_m, worst_f);
    for (int i = 0; i < 4; ++i) printf("  %.9g %.9g\n", a[i], b[i]);
    return 0;
}
