// Memory side of the decoder's conditional cross attention (reference models/transformer.py:389-419: ca_kcontent_proj / ca_v_proj of
// `memory`, ca_kpos_proj of `pos`, the per-head key concatenation [k_content | k_pos]) between the projection GEMMs and the flash
// MHA kernels (mha_flash.hip), for ALL decoder layers at once:
//
//   spe_kv_frags        the fp16 outputs of the two stacked projection GEMMs (spe_gemm_bf16nt, act bits 8 + 9)
//                         ym16 [B*S][2 L d]: column block 2l = ca_kcontent_proj_l(memory), 2l + 1 = ca_v_proj_l(memory)
//                         yp16 [B*S][L d]  : column block l = ca_kpos_proj_l(pos)
//                       -> the operand fragments the attention kernels consume, per layer: Kf (fp16, 32-wide steps of the 2 dh key
//                       dims [k_content | k_pos] of a head), V16 (fp16, 16-wide), and for the backward K16 (bf16, 16-wide) and Vf
//                       (bf16, 32-wide steps).  No fp32 key / value tensor, no torch.cat, no per-layer pack launch: 112 MB read,
//                       <= 240 MB written at cfg2 instead of ~1.4 GB of fp32 traffic.
//                       (Layer 0 of the reference adds k_pos onto k_content; here the QUERY side carries that term:
//                       q_c (k_c + k_p) + q_s k_p = q_c k_c + (q_s + q_c) k_p - models/transformer.py in this tree.)
//   spe_kv_grad_scatter the backward's dk [B,S,H,2 dh] / dv [B,S,H,dh] (fp32, spe_mha_bwd) of one layer -> bf16 column blocks of the
//                       stacked dY operands of the projection GEMMs' backward (dYm [B*S][2 L d], dYp [B*S][L d]).
#include <cstdlib>
#include "common.h"
#include "attn_pack.h"
#include "det_reduce.h"

typedef unsigned int kvu32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 kvh4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float kv_h2f(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }

struct KvFragArgs {
    const unsigned short* ym; long ldm; const unsigned short* yp; long ldp;
    unsigned short* Kf; unsigned short* V16; unsigned short* K16; unsigned short* Vf;
    int L, B, S, H, dh, nt;
};

#define KV_MAXDH 64
__global__ __launch_bounds__(256) void kv_frag_kernel(KvFragArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short sK[4][16 * (2 * KV_MAXDH + 8)];
    __shared__ __attribute__((aligned(16))) unsigned short sV[4][16 * (KV_MAXDH + 8)];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dh = a.dh, dk = 2 * dh, d = a.H * dh, nt = a.nt, S = a.S;
    const int LDK = dk + 8, LDV = dh + 8;
    const int PPK = dk / 8, PPV = dh / 8, PPR = PPK + PPV;            // 16-B pieces per token row: keys, values
    const int steps_k = (dk + 31) / 32, steps_v = (dh + 31) / 32, DTk = (dk + 15) / 16, DTv = (dh + 15) / 16;
    const long nrec = (long)a.L * a.B * a.H * nt;
    const int r = lane & 15, g = lane >> 4;
    unsigned short* tK = sK[wave];
    unsigned short* tV = sV[wave];
    for (long rec = (long)blockIdx.x * 4 + wave; rec < nrec; rec += (long)gridDim.x * 4) {
        const int tile = (int)(rec % nt); long t = rec / nt;
        const int h = (int)(t % a.H); t /= a.H;
        const int b = (int)(t % a.B); const int l = (int)(t / a.B);
        // ---- 16 token rows x (2 dh key dims + dh value dims) -> the wave's LDS tiles, 16 B per lane and piece
        for (int idx = lane; idx < 16 * PPR; idx += 64) {
            const int row = idx / PPR, pc = idx % PPR;
            const int tok = tile * 16 + row;
            kvu32x4_t v = {0u, 0u, 0u, 0u};
            if (tok < S) {
                const long rr = (long)b * S + tok;
                const unsigned short* src;
                if (pc < PPV) src = a.ym + rr * a.ldm + (long)(2 * l) * d + h * dh + pc * 8;                       // k_content
                else if (pc < PPK) src = a.yp + rr * a.ldp + (long)l * d + h * dh + (pc - PPV) * 8;                // k_pos
                else src = a.ym + rr * a.ldm + (long)(2 * l + 1) * d + h * dh + (pc - PPK) * 8;                   // v
                v = *reinterpret_cast<const kvu32x4_t*>(src);
            }
            if (pc < PPK) *reinterpret_cast<kvu32x4_t*>(tK + row * LDK + pc * 8) = v;
            else *reinterpret_cast<kvu32x4_t*>(tV + row * LDV + (pc - PPK) * 8) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // wave-private tiles: written and read by this wave only
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- Kf: fp16, steps of 32 key dims, lane (row r, dim group g) holds 8 consecutive dims
        for (int st = 0; st < steps_k; ++st) {
            const int d0 = st * 32 + g * 8;
            kvu32x4_t v = {0u, 0u, 0u, 0u};
            if (d0 < dk) v = *reinterpret_cast<const kvu32x4_t*>(tK + r * LDK + d0);
            *reinterpret_cast<kvu32x4_t*>(a.Kf + (rec * steps_k + st) * 512 + lane * 8) = v;
        }
        // ---- V16: fp16, 16-wide: lane (dim column r, rows 4 g .. 4 g + 3)
        for (int dt = 0; dt < DTv; ++dt) {
            const int dd = dt * 16 + r;
            uint2 v = make_uint2(0u, 0u);
            if (dd < dh) {
                v.x = (unsigned)tV[(4 * g) * LDV + dd] | ((unsigned)tV[(4 * g + 1) * LDV + dd] << 16);
                v.y = (unsigned)tV[(4 * g + 2) * LDV + dd] | ((unsigned)tV[(4 * g + 3) * LDV + dd] << 16);
            }
            *reinterpret_cast<uint2*>(a.V16 + (rec * DTv + dt) * 256 + lane * 4) = v;
        }
        if (a.K16) {        // backward operands (bf16: they meet gradients)
            for (int dt = 0; dt < DTk; ++dt) {
                const int dd = dt * 16 + r;
                uint2 v = make_uint2(0u, 0u);
                if (dd < dk) v = spe_cvt4_16(kv_h2f(tK[(4 * g) * LDK + dd]), kv_h2f(tK[(4 * g + 1) * LDK + dd]), kv_h2f(tK[(4 * g + 2) * LDK + dd]),
                                             kv_h2f(tK[(4 * g + 3) * LDK + dd]), 0);
                *reinterpret_cast<uint2*>(a.K16 + (rec * DTk + dt) * 256 + lane * 4) = v;
            }
            for (int st = 0; st < steps_v; ++st) {
                const int d0 = st * 32 + g * 8;
                uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
                if (d0 < dh) {
                    const unsigned short* p = tV + r * LDV + d0;
                    lo = spe_cvt4_16(kv_h2f(p[0]), kv_h2f(p[1]), kv_h2f(p[2]), kv_h2f(p[3]), 0);
                    hi = spe_cvt4_16(kv_h2f(p[4]), kv_h2f(p[5]), kv_h2f(p[6]), kv_h2f(p[7]), 0);
                }
                *reinterpret_cast<kvu32x4_t*>(a.Vf + (rec * steps_v + st) * 512 + lane * 8) = (kvu32x4_t){lo.x, lo.y, hi.x, hi.y};
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // the tiles are rewritten by the next record
    }
}

// C-ABI: see include/spe_hip.h
extern "C" int spe_kv_frags(const void* ym16, long ldm, const void* yp16, long ldp, void* Kf, void* V16, void* K16, void* Vf,
                            int L, int B, int S, int H, int dh, hipStream_t st) {
    if (L <= 0 || B <= 0 || S <= 0) return 0;
    if (dh < 8 || dh > KV_MAXDH || (dh & 7) || (ldm & 7) || (ldp & 7) || ((K16 == nullptr) != (Vf == nullptr))) return -2;
    KvFragArgs a;
    a.ym = (const unsigned short*)ym16; a.ldm = ldm; a.yp = (const unsigned short*)yp16; a.ldp = ldp;
    a.Kf = (unsigned short*)Kf; a.V16 = (unsigned short*)V16; a.K16 = (unsigned short*)K16; a.Vf = (unsigned short*)Vf;
    a.L = L; a.B = B; a.S = S; a.H = H; a.dh = dh; a.nt = (S + 15) / 16;
    const long nrec = (long)L * B * H * a.nt;
    long nb = (nrec + 3) / 4; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(kv_frag_kernel, dim3((unsigned)nb), dim3(256), 0, st, a);
    SPE_CHECK_LAUNCH();
    return 0;
}

// one thread per 4 consecutive dims of one (token, head): dk -> [content -> dYm block 2l | pos -> dYp block l], dv -> dYm block 2l + 1
__global__ __launch_bounds__(256) void kv_grad_scatter_kernel(const float* __restrict__ dk, const float* __restrict__ dv, unsigned short* __restrict__ dYm,
                                                              long ldm, unsigned short* __restrict__ dYp, long ldp, int l, long rows, int H, int dh) {
    const int q4 = dh / 4, per_row = H * 3 * q4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * per_row; i += (long)gridDim.x * 256) {
        const long row = i / per_row; int u = (int)(i % per_row);
        const int h = u / (3 * q4); u %= 3 * q4;
        const int which = u / q4, c = (u % q4) * 4;              // 0: k content, 1: k pos, 2: v
        const int d = H * dh;
        float4 v;
        if (which < 2) v = *reinterpret_cast<const float4*>(dk + (row * H + h) * (2L * dh) + which * dh + c);
        else v = *reinterpret_cast<const float4*>(dv + (row * H + h) * (long)dh + c);
        const uint2 o = spe_cvt4_16(v.x, v.y, v.z, v.w, 0);
        unsigned short* dst = (which == 1) ? dYp + row * ldp + (long)l * d + h * dh + c
                                           : dYm + row * ldm + (long)(2 * l + (which == 2 ? 1 : 0)) * d + h * dh + c;
        *reinterpret_cast<uint2*>(dst) = o;
    }
}

extern "C" int spe_kv_grad_scatter(const float* dk, const float* dv, void* dYm, long ldm, void* dYp, long ldp, int layer, int B, int S,
                                   int H, int dh, hipStream_t st) {
    const long rows = (long)B * S;
    if (rows <= 0) return 0;
    if ((dh & 3) || (ldm & 3) || (ldp & 3) || layer < 0) return -2;
    const long n = rows * H * 3 * (dh / 4);
    long nb = (n + 255) / 256; if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(kv_grad_scatter_kernel, dim3((unsigned)nb), dim3(256), 0, st, dk, dv, reinterpret_cast<unsigned short*>(dYm), ldm,
                       reinterpret_cast<unsigned short*>(dYp), ldp, layer, rows, H, dh);
    SPE_CHECK_LAUNCH();
    return 0;
}


// Column sums of a bf16 matrix [R][ld] whose columns are `nblk` blocks of `blkC`, block i summed into its OWN fp32 vector outs[i]
// (the bias gradients of the stacked projections: each lives in its parameter's all-reduce bucket).  Fixed order (det_reduce.h).
#define KV_MAXBLK 32
struct ColsumBlkArgs { const unsigned short* x; long ld; long R; int nblk, blkC, accumulate; float* out[KV_MAXBLK]; };
__global__ __launch_bounds__(256) void colsum_bf16_blocks_kernel(ColsumBlkArgs a, DetWs ws) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int C = a.nblk * a.blkC, c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < C)
        for (long r = (long)blockIdx.y * 4 + rl; r < a.R; r += (long)gridDim.y * 4) acc += spe_bf2f(a.x[r * a.ld + c]);
    red[rl][cl] = acc;
    __syncthreads();
    det_reduce(ws, blockIdx.x, blockIdx.y, gridDim.y, 64, threadIdx.x, 256,
               [&](int k) { return red[0][k] + red[1][k] + red[2][k] + red[3][k]; },
               [&](int k, float s) {
                   const int cc = blockIdx.x * 64 + k;
                   if (cc < C) { float* o = a.out[cc / a.blkC] + cc % a.blkC; *o = a.accumulate ? *o + s : s; }
               });
}

// The same sums with 16-byte loads: a workgroup owns 256 columns (32 lanes x 8) and 8 row lanes; thread (rl, cg) adds rows rl + 8 k of
// its slab for its 8 columns, the 8 row lanes meet in LDS, the slabs across workgroups in a fixed order (det_reduce.h).
__global__ __launch_bounds__(256) void colsum_bf16_blocks_v8_kernel(ColsumBlkArgs a, DetWs ws) {
    __shared__ float red[8][256 + 8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int C = a.nblk * a.blkC, c0 = blockIdx.x * 256 + cg * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < C) {
        const long rows_per = (a.R + gridDim.y - 1) / gridDim.y;
        const long r0 = (long)blockIdx.y * rows_per, r1 = r0 + rows_per < a.R ? r0 + rows_per : a.R;
        for (long r = r0 + rl; r < r1; r += 8) {
            const uint4 q = *reinterpret_cast<const uint4*>(a.x + r * a.ld + c0);
            const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[2 * i] += __uint_as_float(w[i] << 16); acc[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u); }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[rl][cg * 8 + i] = acc[i];
    __syncthreads();
    det_reduce(ws, blockIdx.x, blockIdx.y, gridDim.y, 256, threadIdx.x, 256,
               [&](int k) { float t = 0.f;
#pragma unroll
                            for (int j = 0; j < 8; ++j) t += red[j][k];
                            return t; },
               [&](int k, float s) {
                   const int cc = blockIdx.x * 256 + k;
                   if (cc < C) { float* o = a.out[cc / a.blkC] + cc % a.blkC; *o = a.accumulate ? *o + s : s; }
               });
}

extern "C" int spe_colsum_bf16_blocks(const void* x, long ld, long R, int nblk, int blkC, float* const* outs, int accumulate, hipStream_t st) {
    if (nblk <= 0 || blkC <= 0 || R <= 0) return 0;
    if (nblk > KV_MAXBLK) return -2;
    ColsumBlkArgs a;
    a.x = (const unsigned short*)x; a.ld = ld; a.R = R; a.nblk = nblk; a.blkC = blkC; a.accumulate = accumulate;
    for (int i = 0; i < nblk; ++i) a.out[i] = outs[i];
    if (((nblk * blkC) & 7) == 0 && (ld & 7) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const int gx = (nblk * blkC + 255) / 256;
        static const int ry_max = SPE_KNOB("SPE_COLSUM_RY", 96);       // developer knob (tuning)
        long ry = (R + 63) / 64; if (ry > ry_max) ry = ry_max; if (ry < 1) ry = 1;
        while (gx * ry > 1024 && ry > 16) ry /= 2;            // a few workgroups per CU are enough; the slabs cost a reduction each
        DetWs ws = spe_detws();
        DetDeferSeg sg[KV_MAXBLK];
        for (int i = 0; i < nblk; ++i) sg[i] = DetDeferSeg{outs[i], blkC};
        float* region = det_defer_try(gx, ry, 256, nblk, sg, st);         // deferred: the block sums land at the next flush
        if (region) ws.defer = region; else DET_CHECK(ws, gx, ry, 256);
        hipLaunchKernelGGL(colsum_bf16_blocks_v8_kernel, dim3(gx, (unsigned)ry), dim3(256), 0, st, a, ws);
        if (region) det_defer_commit(region, gx, ry, 256, nblk, sg, accumulate);
        SPE_CHECK_LAUNCH();
        return 0;
    }
    const int gx = (nblk * blkC + 63) / 64;
    long ry = (R + 255) / 256; if (ry > 64) ry = 64; if (ry < 1) ry = 1;
    const DetWs ws = spe_detws();
    DET_CHECK(ws, gx, ry, 64);
    hipLaunchKernelGGL(colsum_bf16_blocks_kernel, dim3(gx, (unsigned)ry), dim3(256), 0, st, a, ws);
    SPE_CHECK_LAUNCH();
    return 0;
}
