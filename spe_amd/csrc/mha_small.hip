// Multi-head attention over a FEW rows - the decoder's query self-attention (reference models/transformer.py:368-386 through
// models/attention.py:277-383: 100 queries per proposal stage, 8 heads of 48 dims) - as ONE launch each way instead of three
// (QK^T GEMM, masked softmax + dropout, PV GEMM) and six (their autograd).  The whole problem of one (batch, head) lives in the LDS of
// one workgroup: 0.5 MFLOP per head - what this costs is launches and dependent round trips, not arithmetic, so everything is plain
// fp32 on the vector pipe (no operand rounding at all: more exact than the MFMA path it replaces).
//   forward : S = scale q k^T (+ key padding mask) ; P = softmax_k(S) (saved for the backward) ; Pd = dropout(P) ; O = Pd v
//   backward: dPd = dO v^T ; dv = Pd^T dO ; dP = dPd * keep ; dS = P (dP - rowsum(dP P)) ; dq = scale dS k ; dk = scale dS^T q
// Dropout element index ((b H + h) Lq + q) ld + key, ld = Lk rounded up to 4: the stream spe_softmax_fwd / spe_mha_fwd draw from.
// LDS: (Lq + Lk)(dk + 1) + Lk dv [+ Lq dv] + Lq (Lk + 1) [x 2] floats; -2 when that exceeds 160 KB (callers keep the three-launch path).
#include "common.h"

#define MS_T 1024          // threads per workgroup: only B * H workgroups exist, so each one brings 16 waves

struct MhaSmallArgs {
    const float* q; const float* k; const float* v; const float* dO; const unsigned char* mask;
    float* O; float* P; float* dq; float* dk; float* dv;
    long qb, qn, qh, kb, kn, kh, vb, vn, vh;          // element strides of the [B, L, H, d] views (unit d stride)
    int B, H, Lq, Lk, dk_, dv_, ld;
    float scale, p_drop; uint64_t seed, offset;
};

__device__ __forceinline__ void ms_load(float* dst, int ldd, const float* src, long sn, int L, int d) {
    for (int e = threadIdx.x; e < L * d; e += MS_T) { const int i = e / d, c = e % d; dst[i * ldd + c] = src[(long)i * sn + c]; }
}

__global__ __launch_bounds__(MS_T) void mha_small_fwd_kernel(MhaSmallArgs a) {
    extern __shared__ float sm[];
    const int Lq = a.Lq, Lk = a.Lk, dk = a.dk_, dv = a.dv_, ldk = dk + 1, lds = Lk + 1;
    float* sQ = sm; float* sK = sQ + Lq * ldk; float* sV = sK + Lk * ldk; float* sS = sV + Lk * dv;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    ms_load(sQ, ldk, a.q + b * a.qb + h * a.qh, a.qn, Lq, dk);
    ms_load(sK, ldk, a.k + b * a.kb + h * a.kh, a.kn, Lk, dk);
    ms_load(sV, dv, a.v + b * a.vb + h * a.vh, a.vn, Lk, dv);
    __syncthreads();
    for (int e = threadIdx.x; e < Lq * Lk; e += MS_T) {
        const int i = e / Lk, j = e % Lk;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // independent chains: the loop is latency-, not throughput-bound
        int d = 0;
        for (; d + 3 < dk; d += 4) {
            s0 = fmaf(sQ[i * ldk + d], sK[j * ldk + d], s0); s1 = fmaf(sQ[i * ldk + d + 1], sK[j * ldk + d + 1], s1);
            s2 = fmaf(sQ[i * ldk + d + 2], sK[j * ldk + d + 2], s2); s3 = fmaf(sQ[i * ldk + d + 3], sK[j * ldk + d + 3], s3);
        }
        for (; d < dk; ++d) s0 = fmaf(sQ[i * ldk + d], sK[j * ldk + d], s0);
        float s = ((s0 + s1) + (s2 + s3)) * a.scale;
        if (a.mask && a.mask[(long)b * Lk + j]) s = -INFINITY;
        sS[i * lds + j] = s;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float inv = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
    for (int i = w; i < Lq; i += MS_T / 64) {
        float m = -INFINITY;
        for (int j = lane; j < Lk; j += 64) m = fmaxf(m, sS[i * lds + j]);
        m = spe_wave_max(m);
        float l = 0.f;
        for (int j = lane; j < Lk; j += 64) { const float p = (m > -INFINITY) ? __expf(sS[i * lds + j] - m) : 0.f; sS[i * lds + j] = p; l += p; }
        l = spe_wave_sum(l);
        const float il = l > 0.f ? 1.f / l : 0.f;
        const long row = ((long)b * a.H + h) * Lq + i;
        float* Pr = a.P + row * a.ld;
        for (int j = lane; j < Lk; j += 64) {
            const float p = sS[i * lds + j] * il;
            Pr[j] = p;
            float pd = p;
            if (a.p_drop > 0.f) pd = (spe_uniform(a.seed, a.offset, (uint64_t)(row * a.ld + j)) >= a.p_drop) ? p * inv : 0.f;
            sS[i * lds + j] = pd;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < Lq * dv; e += MS_T) {
        const int i = e / dv, c = e % dv;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        int j = 0;
        for (; j + 3 < Lk; j += 4) {
            o0 = fmaf(sS[i * lds + (j)], sV[(j) * dv + c], o0);
            o1 = fmaf(sS[i * lds + (j + 1)], sV[(j + 1) * dv + c], o1);
            o2 = fmaf(sS[i * lds + (j + 2)], sV[(j + 2) * dv + c], o2);
            o3 = fmaf(sS[i * lds + (j + 3)], sV[(j + 3) * dv + c], o3);
        }
        for (; j < Lk; ++j) o0 = fmaf(sS[i * lds + (j)], sV[(j) * dv + c], o0);
        float o = (o0 + o1) + (o2 + o3);
        a.O[((long)b * Lq + i) * (a.H * dv) + h * dv + c] = o;
    }
}

__global__ __launch_bounds__(MS_T) void mha_small_bwd_kernel(MhaSmallArgs a) {
    extern __shared__ float sm[];
    const int Lq = a.Lq, Lk = a.Lk, dk = a.dk_, dv = a.dv_, ldk = dk + 1, lds = Lk + 1;
    float* sQ = sm; float* sK = sQ + Lq * ldk; float* sV = sK + Lk * ldk; float* sdO = sV + Lk * dv;
    float* sP = sdO + Lq * dv; float* sG = sP + Lq * lds;       // sP: P ; sG: dPd, then dS
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    ms_load(sQ, ldk, a.q + b * a.qb + h * a.qh, a.qn, Lq, dk);
    ms_load(sK, ldk, a.k + b * a.kb + h * a.kh, a.kn, Lk, dk);
    ms_load(sV, dv, a.v + b * a.vb + h * a.vh, a.vn, Lk, dv);
    ms_load(sdO, dv, a.dO + (long)b * Lq * (a.H * dv) + h * dv, (long)a.H * dv, Lq, dv);
    const long row0 = ((long)b * a.H + h) * Lq;
    for (int e = threadIdx.x; e < Lq * Lk; e += MS_T) { const int i = e / Lk, j = e % Lk; sP[i * lds + j] = a.P[(row0 + i) * a.ld + j]; }
    __syncthreads();
    const float inv = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
    // Pd = P * keep (one Philox evaluation per element, here only) -> sG
    for (int e = threadIdx.x; e < Lq * Lk; e += MS_T) {
        const int i = e / Lk, j = e % Lk;
        float pd = sP[i * lds + j];
        if (a.p_drop > 0.f) pd = (spe_uniform(a.seed, a.offset, (uint64_t)((row0 + i) * a.ld + j)) >= a.p_drop) ? pd * inv : 0.f;
        sG[i * lds + j] = pd;
    }
    __syncthreads();
    // dv[j][c] = sum_i Pd[i][j] dO[i][c]
    for (int e = threadIdx.x; e < Lk * dv; e += MS_T) {
        const int j = e / dv, c = e % dv;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        int i = 0;
        for (; i + 3 < Lq; i += 4) {
            o0 = fmaf(sG[(i) * lds + j], sdO[(i) * dv + c], o0);
            o1 = fmaf(sG[(i + 1) * lds + j], sdO[(i + 1) * dv + c], o1);
            o2 = fmaf(sG[(i + 2) * lds + j], sdO[(i + 2) * dv + c], o2);
            o3 = fmaf(sG[(i + 3) * lds + j], sdO[(i + 3) * dv + c], o3);
        }
        for (; i < Lq; ++i) o0 = fmaf(sG[(i) * lds + j], sdO[(i) * dv + c], o0);
        float o = (o0 + o1) + (o2 + o3);
        a.dv[((long)b * Lk + j) * (a.H * dv) + h * dv + c] = o;
    }
    __syncthreads();
    // dP = (dO v^T) * keep ; keep is read off Pd (a dropped element has Pd = 0; where P itself is 0 the factor does not matter: dS = P (..))
    for (int e = threadIdx.x; e < Lq * Lk; e += MS_T) {
        const int i = e / Lk, j = e % Lk;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
        int c = 0;
        for (; c + 3 < dv; c += 4) {
            g0 = fmaf(sdO[i * dv + (c)], sV[j * dv + (c)], g0);
            g1 = fmaf(sdO[i * dv + (c + 1)], sV[j * dv + (c + 1)], g1);
            g2 = fmaf(sdO[i * dv + (c + 2)], sV[j * dv + (c + 2)], g2);
            g3 = fmaf(sdO[i * dv + (c + 3)], sV[j * dv + (c + 3)], g3);
        }
        for (; c < dv; ++c) g0 = fmaf(sdO[i * dv + (c)], sV[j * dv + (c)], g0);
        float g = (g0 + g1) + (g2 + g3);
        const float ks = (a.p_drop > 0.f) ? ((sG[i * lds + j] != 0.f) ? inv : 0.f) : 1.f;
        sG[i * lds + j] = g * ks;
    }
    __syncthreads();
    // dS = P (dP - sum_j dP P)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = w; i < Lq; i += MS_T / 64) {
        float d = 0.f;
        for (int j = lane; j < Lk; j += 64) d = fmaf(sG[i * lds + j], sP[i * lds + j], d);
        d = spe_wave_sum(d);
        for (int j = lane; j < Lk; j += 64) sG[i * lds + j] = sP[i * lds + j] * (sG[i * lds + j] - d);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < Lq * dk; e += MS_T) {
        const int i = e / dk, c = e % dk;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        int j = 0;
        for (; j + 3 < Lk; j += 4) {
            o0 = fmaf(sG[i * lds + (j)], sK[(j) * ldk + c], o0);
            o1 = fmaf(sG[i * lds + (j + 1)], sK[(j + 1) * ldk + c], o1);
            o2 = fmaf(sG[i * lds + (j + 2)], sK[(j + 2) * ldk + c], o2);
            o3 = fmaf(sG[i * lds + (j + 3)], sK[(j + 3) * ldk + c], o3);
        }
        for (; j < Lk; ++j) o0 = fmaf(sG[i * lds + (j)], sK[(j) * ldk + c], o0);
        float o = (o0 + o1) + (o2 + o3);
        a.dq[((long)b * Lq + i) * (a.H * dk) + h * dk + c] = o * a.scale;
    }
    for (int e = threadIdx.x; e < Lk * dk; e += MS_T) {
        const int j = e / dk, c = e % dk;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        int i = 0;
        for (; i + 3 < Lq; i += 4) {
            o0 = fmaf(sG[(i) * lds + j], sQ[(i) * ldk + c], o0);
            o1 = fmaf(sG[(i + 1) * lds + j], sQ[(i + 1) * ldk + c], o1);
            o2 = fmaf(sG[(i + 2) * lds + j], sQ[(i + 2) * ldk + c], o2);
            o3 = fmaf(sG[(i + 3) * lds + j], sQ[(i + 3) * ldk + c], o3);
        }
        for (; i < Lq; ++i) o0 = fmaf(sG[(i) * lds + j], sQ[(i) * ldk + c], o0);
        float o = (o0 + o1) + (o2 + o3);
        a.dk[((long)b * Lk + j) * (a.H * dk) + h * dk + c] = o * a.scale;
    }
}

static inline long ms_smem(int Lq, int Lk, int dk, int dv, bool bwd) {
    long f = (long)(Lq + Lk) * (dk + 1) + (long)Lk * dv + (long)Lq * (Lk + 1);
    if (bwd) f += (long)Lq * dv + (long)Lq * (Lk + 1);
    return f * 4;
}

// C-ABI: see include/spe_hip.h
extern "C" int spe_mha_small_fwd(const float* q, long qb, long qn, long qh, const float* k, long kb, long kn, long kh,
                                 const float* v, long vb, long vn, long vh, const void* mask, float* O, float* P,
                                 int B, int H, int Lq, int Lk, int dk, int dv, float scale, float p_drop, uint64_t seed, uint64_t offset,
                                 hipStream_t st) {
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
    const long smem = ms_smem(Lq, Lk, dk, dv, false);
    if (smem > 160 * 1024 || dk < 1 || dv < 1) return -2;
    static long set_for = 0;
    if (smem > set_for) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mha_small_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        set_for = smem;
    }
    MhaSmallArgs a{};
    a.q = q; a.k = k; a.v = v; a.mask = (const unsigned char*)mask; a.O = O; a.P = P;
    a.qb = qb; a.qn = qn; a.qh = qh; a.kb = kb; a.kn = kn; a.kh = kh; a.vb = vb; a.vn = vn; a.vh = vh;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.dk_ = dk; a.dv_ = dv; a.ld = (Lk + 3) & ~3;
    a.scale = scale; a.p_drop = p_drop; a.seed = seed; a.offset = offset;
    hipLaunchKernelGGL(mha_small_fwd_kernel, dim3(B * H), dim3(MS_T), (size_t)smem, st, a);
    SPE_CHECK_LAUNCH();
    return 0;
}

extern "C" int spe_mha_small_bwd(const float* q, long qb, long qn, long qh, const float* k, long kb, long kn, long kh,
                                 const float* v, long vb, long vn, long vh, const float* P, const float* dO, float* dq, float* dk_out, float* dv_out,
                                 int B, int H, int Lq, int Lk, int dk, int dv, float scale, float p_drop, uint64_t seed, uint64_t offset,
                                 hipStream_t st) {
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
    const long smem = ms_smem(Lq, Lk, dk, dv, true);
    if (smem > 160 * 1024 || dk < 1 || dv < 1) return -2;
    static long set_for = 0;
    if (smem > set_for) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mha_small_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        set_for = smem;
    }
    MhaSmallArgs a{};
    a.q = q; a.k = k; a.v = v; a.dO = dO; a.P = const_cast<float*>(P); a.dq = dq; a.dk = dk_out; a.dv = dv_out;
    a.qb = qb; a.qn = qn; a.qh = qh; a.kb = kb; a.kn = kn; a.kh = kh; a.vb = vb; a.vn = vn; a.vh = vh;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.dk_ = dk; a.dv_ = dv; a.ld = (Lk + 3) & ~3;
    a.scale = scale; a.p_drop = p_drop; a.seed = seed; a.offset = offset;
    hipLaunchKernelGGL(mha_small_bwd_kernel, dim3(B * H), dim3(MS_T), (size_t)smem, st, a);
    SPE_CHECK_LAUNCH();
    return 0;
}
