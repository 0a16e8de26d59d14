// Talking-heads score transform of CaiT self-attention (reference models/cait.py:379-387):
//
//   S'[g] = sum_h Wl[g,h] S[h] + bl[g]        (proj_l: Linear over the head axis, pre-softmax)
//   P[g]  = softmax_k(S'[g])
//   P'[g] = sum_h Ww[g,h] P[h] + bw[g]        (proj_w, post-softmax)
//   Pd    = dropout(P')                        (attn_drop)
//
// on materialised scores S[B,H,Nq,ld] (Nk valid columns).  One workgroup owns all H score rows
// of one (b,q) so the head mix happens in registers; the softmax statistics are an online
// (max,sum) pass, the second pass re-reads the rows from L2.  This is the HBM-bound first
// implementation of K4 in SURVEY.md section 2.2; the fused never-materialise kernel replaces it.
#include "common.h"

template <int H>
__global__ __launch_bounds__(256) void talking_fwd_kernel(const float* __restrict__ S, const float* __restrict__ Wl,
                                                          const float* __restrict__ bl, const float* __restrict__ Ww,
                                                          const float* __restrict__ bw, float* __restrict__ P,
                                                          float* __restrict__ Pd, int Nq, int Nk, long ld,
                                                          float p_drop, uint64_t seed, uint64_t offset) {
    __shared__ float red[16];
    __shared__ float sWl[H * H], sWw[H * H], sbl[H], sbw[H], sM[H], sInv[H];
    const int b = blockIdx.x / Nq, q = blockIdx.x % Nq;
    const long hs = (long)Nq * ld;                       // head stride
    const long base = ((long)b * H * Nq + q) * ld;       // (b, h=0, q, 0)
    for (int i = threadIdx.x; i < H * H; i += 256) { sWl[i] = Wl[i]; sWw[i] = Ww[i]; }
    if (threadIdx.x < H) { sbl[threadIdx.x] = bl[threadIdx.x]; sbw[threadIdx.x] = bw[threadIdx.x]; }
    __syncthreads();

    float m[H], l[H];
#pragma unroll
    for (int g = 0; g < H; ++g) { m[g] = -INFINITY; l[g] = 0.f; }
    for (int k = threadIdx.x; k < Nk; k += 256) {
        float s[H];
#pragma unroll
        for (int h = 0; h < H; ++h) s[h] = S[base + h * hs + k];
#pragma unroll
        for (int g = 0; g < H; ++g) {
            float v = sbl[g];
#pragma unroll
            for (int h = 0; h < H; ++h) v += sWl[g * H + h] * s[h];
            if (v > m[g]) { l[g] = l[g] * __expf(m[g] - v) + 1.f; m[g] = v; }
            else l[g] += __expf(v - m[g]);
        }
    }
#pragma unroll
    for (int g = 0; g < H; ++g) {
        const float M = spe_block_max(m[g], red);
        const float lg = (m[g] > -INFINITY) ? l[g] * __expf(m[g] - M) : 0.f;
        const float L = spe_block_sum(lg, red);
        if (threadIdx.x == 0) { sM[g] = M; sInv[g] = 1.f / L; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < Nk; k += 256) {
        float s[H], p[H];
#pragma unroll
        for (int h = 0; h < H; ++h) s[h] = S[base + h * hs + k];
#pragma unroll
        for (int g = 0; g < H; ++g) {
            float v = sbl[g];
#pragma unroll
            for (int h = 0; h < H; ++h) v += sWl[g * H + h] * s[h];
            p[g] = __expf(v - sM[g]) * sInv[g];
        }
#pragma unroll
        for (int g = 0; g < H; ++g) P[base + g * hs + k] = p[g];   // P may alias S: all S[.,k] already read
#pragma unroll
        for (int g = 0; g < H; ++g) {
            float v = sbw[g];
#pragma unroll
            for (int h = 0; h < H; ++h) v += sWw[g * H + h] * p[h];
            if (p_drop > 0.f) {
                const uint64_t idx = (uint64_t)(((long)(b * H + g) * Nq + q) * (long)Nk + k);
                v *= spe_drop_scale(seed, offset, idx, p_drop);
            }
            Pd[base + g * hs + k] = v;
        }
    }
}

// Backward.  Inputs: dPd (gradient w.r.t. Pd), P (saved), S (recomputed raw scores).
// Outputs: dS (may alias dPd) and per-workgroup partials of [dWl(H*H) | dbl(H) | dWw(H*H) | dbw(H)]
// written to ws[gridDim.x][2*(H*H+H)] (reduced by spe_colsum afterwards - deterministic, no
// same-address atomics).  A workgroup walks (b,q) pairs with a grid stride.
template <int H>
__global__ __launch_bounds__(256) void talking_bwd_kernel(const float* __restrict__ dPd, const float* __restrict__ P,
                                                          const float* __restrict__ S, const float* __restrict__ Wl,
                                                          const float* __restrict__ Ww, float* __restrict__ dS,
                                                          float* __restrict__ ws, int B, int Nq, int Nk, long ld,
                                                          float p_drop, uint64_t seed, uint64_t offset) {
    constexpr int NW = 2 * (H * H + H);
    __shared__ float red[16];
    __shared__ float sWl[H * H], sWw[H * H], sRS[H];
    __shared__ float part[4][NW];
    for (int i = threadIdx.x; i < H * H; i += 256) { sWl[i] = Wl[i]; sWw[i] = Ww[i]; }
    __syncthreads();
    float aWl[H][H], abl[H], aWw[H][H], abw[H];
#pragma unroll
    for (int g = 0; g < H; ++g) {
        abl[g] = 0.f; abw[g] = 0.f;
#pragma unroll
        for (int h = 0; h < H; ++h) { aWl[g][h] = 0.f; aWw[g][h] = 0.f; }
    }
    const long hs = (long)Nq * ld;
    for (int item = blockIdx.x; item < B * Nq; item += gridDim.x) {
        const int b = item / Nq, q = item % Nq;
        const long base = ((long)b * H * Nq + q) * ld;
        float rs[H];
#pragma unroll
        for (int h = 0; h < H; ++h) rs[h] = 0.f;
        // pass 1: dP' -> dWw, dbw, dP, row sums of dP*P
        for (int k = threadIdx.x; k < Nk; k += 256) {
            float d[H], p[H];
#pragma unroll
            for (int g = 0; g < H; ++g) {
                d[g] = dPd[base + g * hs + k];
                if (p_drop > 0.f) d[g] *= spe_drop_scale(seed, offset, (uint64_t)(((long)(b * H + g) * Nq + q) * (long)Nk + k), p_drop);
                p[g] = P[base + g * hs + k];
            }
#pragma unroll
            for (int g = 0; g < H; ++g) {
                abw[g] += d[g];
#pragma unroll
                for (int h = 0; h < H; ++h) aWw[g][h] += d[g] * p[h];
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float dp = 0.f;
#pragma unroll
                for (int g = 0; g < H; ++g) dp += sWw[g * H + h] * d[g];
                rs[h] += dp * p[h];
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float r = spe_block_sum(rs[h], red);
            if (threadIdx.x == 0) sRS[h] = r;
        }
        __syncthreads();
        // pass 2: dS' = P*(dP - rs) -> dWl, dbl, dS
        for (int k = threadIdx.x; k < Nk; k += 256) {
            float d[H], p[H], s[H], ds1[H];
#pragma unroll
            for (int g = 0; g < H; ++g) {
                d[g] = dPd[base + g * hs + k];
                if (p_drop > 0.f) d[g] *= spe_drop_scale(seed, offset, (uint64_t)(((long)(b * H + g) * Nq + q) * (long)Nk + k), p_drop);
                p[g] = P[base + g * hs + k];
                s[g] = S[base + g * hs + k];
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float dp = 0.f;
#pragma unroll
                for (int g = 0; g < H; ++g) dp += sWw[g * H + h] * d[g];
                ds1[h] = p[h] * (dp - sRS[h]);
            }
#pragma unroll
            for (int g = 0; g < H; ++g) {
                abl[g] += ds1[g];
#pragma unroll
                for (int h = 0; h < H; ++h) aWl[g][h] += ds1[g] * s[h];
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float v = 0.f;
#pragma unroll
                for (int g = 0; g < H; ++g) v += sWl[g * H + h] * ds1[g];
                dS[base + h * hs + k] = v;
            }
        }
        __syncthreads();
    }
    // combine the register accumulators: wave shuffle, then LDS across the 4 waves
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < H; ++g) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float a = spe_wave_sum(aWl[g][h]), c = spe_wave_sum(aWw[g][h]);
            if (lane == 0) { part[w][g * H + h] = a; part[w][H * H + H + g * H + h] = c; }
        }
        const float a = spe_wave_sum(abl[g]), c = spe_wave_sum(abw[g]);
        if (lane == 0) { part[w][H * H + g] = a; part[w][2 * H * H + H + g] = c; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NW; i += 256)
        ws[(long)blockIdx.x * NW + i] = part[0][i] + part[1][i] + part[2][i] + part[3][i];
}

#define TALK_DISPATCH(H_, CALL) \
    switch (H_) { case 4: { constexpr int HH = 4; CALL; break; } case 6: { constexpr int HH = 6; CALL; break; } \
                  case 8: { constexpr int HH = 8; CALL; break; } default: return -2; }

extern "C" int spe_talking_softmax_fwd(const float* S, const float* Wl, const float* bl, const float* Ww, const float* bw,
                                       float* P, float* Pd, int B, int H, int Nq, int Nk, long ld, float p_drop,
                                       uint64_t seed, uint64_t offset, hipStream_t st) {
    if (B * Nq <= 0) return 0;
    TALK_DISPATCH(H, hipLaunchKernelGGL((talking_fwd_kernel<HH>), dim3(B * Nq), dim3(256), 0, st, S, Wl, bl, Ww, bw, P, Pd,
                                        Nq, Nk, ld, p_drop, seed, offset));
    SPE_CHECK_LAUNCH();
    return 0;
}

// ws must hold nblocks * 2*(H*H+H) floats; returns partial rows to be column-summed by the caller.
extern "C" int spe_talking_softmax_bwd(const float* dPd, const float* P, const float* S, const float* Wl, const float* Ww,
                                       float* dS, float* ws, int nblocks, int B, int H, int Nq, int Nk, long ld,
                                       float p_drop, uint64_t seed, uint64_t offset, hipStream_t st) {
    if (B * Nq <= 0) return 0;
    if (nblocks > B * Nq) nblocks = B * Nq;
    TALK_DISPATCH(H, hipLaunchKernelGGL((talking_bwd_kernel<HH>), dim3(nblocks), dim3(256), 0, st, dPd, P, S, Wl, Ww, dS, ws,
                                        B, Nq, Nk, ld, p_drop, seed, offset));
    SPE_CHECK_LAUNCH();
    return 0;
}
