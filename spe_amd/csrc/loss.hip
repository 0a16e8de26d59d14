// Matcher cost matrix + set-criterion losses (reference models/matcher.py:62-83,
// util/box_ops.py:18-74, models/conditional_detr.py:237-265, 300-319, 468-494, 504-560).
// All decoder layers (and any number of images) go through ONE launch per kernel; only the
// block-diagonal (same-image) part of the cost matrix is ever computed.
#include "common.h"
#include "det_reduce.h"

struct Box { float x0, y0, x1, y1; };
__device__ __forceinline__ Box to_xyxy(const float* b) {
    Box r; r.x0 = b[0] - 0.5f * b[2]; r.y0 = b[1] - 0.5f * b[3]; r.x1 = b[0] + 0.5f * b[2]; r.y1 = b[1] + 0.5f * b[3];
    return r;
}
__device__ __forceinline__ float giou_xyxy(const Box& a, const Box& b) {
    const float area1 = (a.x1 - a.x0) * (a.y1 - a.y0), area2 = (b.x1 - b.x0) * (b.y1 - b.y0);
    const float iw = fmaxf(fminf(a.x1, b.x1) - fmaxf(a.x0, b.x0), 0.f), ih = fmaxf(fminf(a.y1, b.y1) - fmaxf(a.y0, b.y0), 0.f);
    const float inter = iw * ih, uni = area1 + area2 - inter, iou = inter / uni;
    const float ew = fmaxf(fmaxf(a.x1, b.x1) - fminf(a.x0, b.x0), 0.f), eh = fmaxf(fmaxf(a.y1, b.y1) - fminf(a.y0, b.y0), 0.f);
    const float ea = ew * eh;
    return iou - (ea - uni) / ea;
}

// cost[l] = packed concat over images b of row-major [Q, M_b] blocks (block b starts at Q*toff[b]).
//   C = w_bbox * L1(cxcywh) + w_class * (pos_focal - neg_focal)[label] - w_giou * GIoU
// alpha = 0.25, gamma = 2, eps = 1e-8 are hard-coded in the reference (matcher.py:70-73).
__global__ __launch_bounds__(256) void matcher_cost_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                           const int* __restrict__ tgt_ids, const float* __restrict__ tgt_boxes,
                                                           const int* __restrict__ toff, float* __restrict__ cost,
                                                           int* __restrict__ err, int L, int B, int Q, int Kc,
                                                           float w_class, float w_bbox, float w_giou) {
    const long total = (long)Q * toff[B];
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y;
    if (p >= total) return;
    int b = 0;
    while (b + 1 < B && p >= (long)Q * toff[b + 1]) ++b;
    const int Mb = toff[b + 1] - toff[b];
    const long rem = p - (long)Q * toff[b];
    const int q = (int)(rem / Mb), j = (int)(rem % Mb);
    const int t = toff[b] + j;
    const long row = ((long)l * B + b) * Q + q;
    const float x = logits[row * Kc + tgt_ids[t]];
    const float pr = 1.f / (1.f + expf(-x));
    const float neg = 0.75f * (pr * pr) * (-logf(1.f - pr + 1e-8f));
    const float pos = 0.25f * ((1.f - pr) * (1.f - pr)) * (-logf(pr + 1e-8f));
    const float* sb = boxes + row * 4;
    const float* tb = tgt_boxes + (long)t * 4;
    const float l1 = fabsf(sb[0] - tb[0]) + fabsf(sb[1] - tb[1]) + fabsf(sb[2] - tb[2]) + fabsf(sb[3] - tb[3]);
    const Box a = to_xyxy(sb), c = to_xyxy(tb);
    if (!(a.x1 >= a.x0 && a.y1 >= a.y0 && c.x1 >= c.x0 && c.y1 >= c.y0)) atomicOr(err, 1);  // box_ops.py:64-65 asserts
    cost[(long)l * total + p] = w_bbox * l1 + w_class * (pos - neg) - w_giou * giou_xyxy(a, c);
}

extern "C" int spe_matcher_cost(const float* logits, const float* boxes, const int* tgt_ids, const float* tgt_boxes,
                                const int* toff, int total_targets, float* cost, int* err, int L, int B, int Q, int Kc,
                                float w_class, float w_bbox, float w_giou, hipStream_t st) {
    const long total = (long)Q * total_targets;
    if (total <= 0 || L <= 0) return 0;
    hipLaunchKernelGGL(matcher_cost_kernel, dim3((unsigned)((total + 255) / 256), L), dim3(256), 0, st, logits, boxes, tgt_ids,
                       tgt_boxes, toff, cost, err, L, B, Q, Kc, w_class, w_bbox, w_giou);
    SPE_CHECK_LAUNCH();
    return 0;
}

// Device-side assignment (reference models/matcher.py:83-86 calls scipy.optimize.linear_sum_assignment on the host
// per image; here the optimum is computed where the costs are, so the criterion needs no device->host round trip
// and the host keeps running ahead of the GPU).  One wave per (layer, image) problem: the M_b targets of the image
// are assigned to distinct queries by the shortest-augmenting-path Hungarian algorithm with dual potentials, in
// fp64 like SciPy (which promotes the fp32 costs to double).  The minimum-cost assignment is unique unless two
// assignments tie exactly, so the result equals SciPy's.  Output for problem (l, b): M_b triples at offset
// l*total + toff[b], ordered by query index (the order linear_sum_assignment returns):
//   srow = (l*B + b)*Q + q        (row of the flattened [L*B*Q] predictions)
//   gidx = toff[b] + j            (index into the concatenated targets)
//   lidx = l
//
// The search is ONE serial chain of (rows x path length) steps - 300 x 300 costs ~45 000 of them - so what matters is the latency of a step.
// Everything a step touches lives in registers: lane t owns the columns t + 1 + 64 c, c < CPL, with their v, minv, way, p, used AND the row
// potential of the row matched to the column (u travels with p: the row potentials are only ever read through a column); the problem's cost
// matrix is staged once in LDS (transposed: a row's costs lie along the lanes) when it fits.  The wave-wide argmin - the bulk of a step in
// round 4's kernel, whose butterfly of 64-bit __shfl_xor pairs cost six dependent LDS-crossbar round trips - is two 32-bit unsigned minima
// over an order-preserving integer image of the doubles, each four DPP stages + four v_readlane, and the lowest column among the ties comes
// from one ballot per column slot.  No LDS traffic but the cost row, no barrier inside the search.  Same fp64 arithmetic in the same order and
// the same tie rule (lowest column index) as rounds 2-5: the pairs equal SciPy's at every tested size.
#define HUNG_QMAX 1024
#define HUNG_LDS_FLOATS (36 * 1024)

template <int CTRL>
__device__ __forceinline__ unsigned hung_dpp(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false); }
// minimum over the 64 lanes (all active), returned uniform: xor 1, xor 2 inside the quads, the mirrored quad of the half row, the mirrored half
// row - 16 lanes agree - then the four rows through the scalar unit
__device__ __forceinline__ unsigned hung_wave_umin(unsigned x) {
    x = min(x, hung_dpp<0xB1>(x));          // quad_perm [1,0,3,2]
    x = min(x, hung_dpp<0x4E>(x));          // quad_perm [2,3,0,1]
    x = min(x, hung_dpp<0x141>(x));         // row_half_mirror
    x = min(x, hung_dpp<0x140>(x));         // row_mirror
    const unsigned a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16), c = __builtin_amdgcn_readlane(x, 32),
                   d = __builtin_amdgcn_readlane(x, 48);
    return min(min(a, b), min(c, d));
}
// order-preserving map double -> uint64 (and back): a < b  <=>  key(a) < key(b) for all non-NaN values
__device__ __forceinline__ unsigned long long hung_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double hung_unkey(unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

template <int CPL>
__global__ __launch_bounds__(64) void hungarian_kernel(const float* __restrict__ cost, const int* __restrict__ toff,
                                                       long* __restrict__ srow, long* __restrict__ gidx, int* __restrict__ lidx,
                                                       int* __restrict__ err, int B, int Q) {
    extern __shared__ float a_t[];                       // [n][m] when n * m <= HUNG_LDS_FLOATS
    const int lane = threadIdx.x;
    const int b = blockIdx.x, l = blockIdx.y;
    const int total = toff[B];
    const int n = toff[b + 1] - toff[b], m = Q;          // n targets (rows, 1-based i), m queries (columns, 1-based j)
    if (n <= 0) return;
    const float* a = cost + (long)l * Q * total + (long)Q * toff[b];      // a(i, j) = a[(j-1)*n + (i-1)]
    const bool staged = (long)n * m <= HUNG_LDS_FLOATS;
    if (staged)
        for (int e = lane; e < n * m; e += 64) { const int j = e / n, i = e - j * n; a_t[i * m + j] = a[e]; }
    double v[CPL], minv[CPL], ucol[CPL]; int p[CPL], way[CPL]; bool used[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) { v[c] = 0.0; ucol[c] = 0.0; p[c] = 0; way[c] = 0; }
    // value of a per-column register of column j (1-based; wave-uniform j) as a uniform value: a select over the slots + one v_readlane
    auto col_i = [&](const int (&r)[CPL], int j) {
        const int slot = (j - 1) >> 6, own = __builtin_amdgcn_readfirstlane((j - 1) & 63);
        int x = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) if (c == slot) x = r[c];
        return __builtin_amdgcn_readlane(x, own);
    };
    auto col_d = [&](const double (&r)[CPL], int j) {
        const int slot = (j - 1) >> 6, own = __builtin_amdgcn_readfirstlane((j - 1) & 63);
        double x = 0.0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) if (c == slot) x = r[c];
        const long long bits = __double_as_longlong(x);
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)bits, own), hi = __builtin_amdgcn_readlane((unsigned)(bits >> 32), own);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    __syncthreads();                                     // the staged costs are in place
    for (int i = 1; i <= n; ++i) {
        const int p0 = i;                                 // the virtual column 0 holds the row being inserted ...
        double u0 = 0.0;                                  // ... and its potential (a fresh row starts at 0)
#pragma unroll
        for (int c = 0; c < CPL; ++c) { minv[c] = INFINITY; used[c] = false; }
        int j0 = 0, i0 = p0;
        double ui0 = 0.0;
        do {
            if (j0 != 0) {
                const int slot = (j0 - 1) >> 6, own = (j0 - 1) & 63;
#pragma unroll
                for (int c = 0; c < CPL; ++c) if (c == slot && lane == own) used[c] = true;
            }
            unsigned long long bk = ~0ull;                // this lane's best key over its free columns
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int j = lane + 64 * c + 1;
                if (j > m || used[c]) continue;
                const float av = staged ? a_t[(i0 - 1) * m + (j - 1)] : a[(long)(j - 1) * n + (i0 - 1)];
                const double cur = (double)av - ui0 - v[c];
                if (cur < minv[c]) { minv[c] = cur; way[c] = j0; }
                const unsigned long long k = hung_key(minv[c]);
                bk = (k < bk) ? k : bk;
            }
            // wave minimum of the 64-bit keys: high words first, then the low words of the lanes that hold the minimal high word
            const unsigned hi = hung_wave_umin((unsigned)(bk >> 32));
            const unsigned lo = hung_wave_umin(((unsigned)(bk >> 32) == hi) ? (unsigned)bk : 0xffffffffu);
            const unsigned long long best = ((unsigned long long)hi << 32) | lo;
            const double delta = hung_unkey(best);
            // lowest column index among the free columns whose minv equals the minimum (SciPy's tie rule)
            int j1 = 0x7fffffff;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int j = lane + 64 * c + 1;
                const unsigned long long mask = __ballot(j <= m && !used[c] && hung_key(minv[c]) == best);
                if (j1 == 0x7fffffff && mask != 0ull) j1 = 64 * c + __builtin_ctzll(mask) + 1;
            }
            if (best == ~0ull || j1 == 0x7fffffff || !(fabs(delta) <= 1.7e308)) {
                // every remaining cost of this row is NaN / +-inf (diverged logits or boxes): the augmenting-path search
                // has no column to move to.  SciPy raises ValueError here (matcher.py:86); the device path raises flag
                // bit 1 (inspected by the host without a stall) and emits the identity assignment so that nothing
                // indexes out of bounds or spins; the step's loss is non-finite anyway (engine.py:156-159 exits on it).
                if (err && lane == 0) atomicOr(err, 2);
                const long ob = (long)l * total + toff[b];
                for (int i2 = lane; i2 < n; i2 += 64) {
                    srow[ob + i2] = ((long)l * B + b) * Q + i2;
                    gidx[ob + i2] = toff[b] + i2;
                    lidx[ob + i2] = l;
                }
                return;
            }
            u0 += delta;                                  // column 0 is always in the tree
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int j = lane + 64 * c + 1;
                if (j > m) continue;
                if (used[c]) { ucol[c] += delta; v[c] -= delta; }
                else minv[c] -= delta;
            }
            j0 = j1;
            i0 = col_i(p, j0);
            ui0 = col_d(ucol, j0);
        } while (i0 != 0);
        // augment along the alternating path (uniform walk; the owner of a column rewrites its row and that row's potential)
        do {
            const int j1 = col_i(way, j0);
            const int pj1 = (j1 == 0) ? p0 : col_i(p, j1);
            const double uj1 = (j1 == 0) ? u0 : col_d(ucol, j1);
            const int slot = (j0 - 1) >> 6, own = (j0 - 1) & 63;
#pragma unroll
            for (int c = 0; c < CPL; ++c) if (c == slot && lane == own) { p[c] = pj1; ucol[c] = uj1; }
            j0 = j1;
        } while (j0);
    }
    // emit the pairs in ascending query order
    const long obase = (long)l * total + toff[b];
    int base = 0;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int j = lane + 64 * c + 1;
        const bool has = j <= m && p[c] != 0;
        const unsigned long long mask = __ballot(has);
        if (has) {
            const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            srow[obase + pos] = ((long)l * B + b) * Q + (j - 1);
            gidx[obase + pos] = toff[b] + (p[c] - 1);
            lidx[obase + pos] = l;
        }
        base += __popcll(mask);
    }
}

template <int CPL>
static int launch_hungarian(const float* cost, const int* toff, long* srow, long* gidx, int* lidx, int* err, int L, int B, int Q, hipStream_t st) {
    constexpr int smem = HUNG_LDS_FLOATS * (int)sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hungarian_kernel<CPL>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(hungarian_kernel<CPL>, dim3(B, L), dim3(64), smem, st, cost, toff, srow, gidx, lidx, err, B, Q);
    SPE_CHECK_LAUNCH();
    return 0;
}

// C-ABI: see include/spe_hip.h (spe_hungarian).  -2: Q above HUNG_QMAX (callers fall back to the host solver).
extern "C" int spe_hungarian(const float* cost, const int* toff, long* srow, long* gidx, int* lidx, int* err, int L, int B,
                             int Q, hipStream_t st) {
    if (L <= 0 || B <= 0 || Q <= 0) return 0;
    if (Q > HUNG_QMAX) return -2;
    if (Q <= 128) return launch_hungarian<2>(cost, toff, srow, gidx, lidx, err, L, B, Q, st);
    if (Q <= 320) return launch_hungarian<5>(cost, toff, srow, gidx, lidx, err, L, B, Q, st);
    if (Q <= 512) return launch_hungarian<8>(cost, toff, srow, gidx, lidx, err, L, B, Q, st);
    return launch_hungarian<16>(cost, toff, srow, gidx, lidx, err, L, B, Q, st);
}

// Weighted sigmoid focal loss, forward value + d(sum)/d(logit) in one pass (one wave per
// (l,b,q) row).  tclass[row] in [0,Kc] (Kc = no object -> all-zero one-hot), roww[row] = weight
// of the row (nullptr = 1).  loss[l] += sum_row sum_c w * alpha_t * ce * (1-clamp(p_t))^gamma.
// argmax[row] = top-1 class (for class_error / cardinality logging).
__global__ __launch_bounds__(256) void focal_kernel(const float* __restrict__ logits, const int* __restrict__ tclass,
                                                    const float* __restrict__ roww, float* __restrict__ grad,
                                                    float* __restrict__ loss, int* __restrict__ argmax,
                                                    long rows_per_l, int Kc, float alpha, float gamma, DetWs ws) {
    // grid (row blocks of a layer, layers): the layer's loss is the fixed-order sum of its row blocks (det_reduce.h)
    __shared__ float wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long rl = (long)blockIdx.x * 4 + wv;
    float acc = 0.f;
    if (rl < rows_per_l) {
        const long row = (long)blockIdx.y * rows_per_l + rl;
        const int tc = tclass[row];
        const float w = roww ? roww[row] : 1.f;
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int c = lane; c < Kc; c += 64) {
            const float x = logits[row * Kc + c];
            const float t = (c == tc) ? 1.f : 0.f;
            const float p = 1.f / (1.f + expf(-x));
            const float ce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));   // BCE-with-logits, stable form
            const float pt_raw = p * t + (1.f - p) * (1.f - t);
            const float pt = fminf(fmaxf(pt_raw, 1e-5f), 1.f - 1e-5f);
            const bool clamped = (pt_raw < 1e-5f) || (pt_raw > 1.f - 1e-5f);
            const float om = 1.f - pt;
            const float mod = powf(om, gamma);
            const float at = alpha >= 0.f ? alpha * t + (1.f - alpha) * (1.f - t) : 1.f;
            acc += w * at * ce * mod;
            const float dpt = clamped ? 0.f : p * (1.f - p) * (2.f * t - 1.f);
            const float dmod = -gamma * powf(om, gamma - 1.f) * dpt;
            grad[row * Kc + c] = w * at * ((p - t) * mod + ce * dmod);
            if (x > best) { best = x; bi = c; }
        }
        acc = spe_wave_sum(acc);
        // argmax with lowest-index tie break
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) argmax[row] = bi;
    }
    if (lane == 0) wsum[wv] = acc;
    __syncthreads();
    det_reduce(ws, blockIdx.y, blockIdx.x, gridDim.x, 1, threadIdx.x, 256,
               [&](int) { return wsum[0] + wsum[1] + wsum[2] + wsum[3]; },
               [&](int, float t) { loss[blockIdx.y] += t; });
}
extern "C" int spe_focal_loss(const float* logits, const int* tclass, const float* roww, float* grad, float* loss,
                              int* argmax, int L, long rows_per_l, int Kc, float alpha, float gamma, hipStream_t st) {
    if (L <= 0 || rows_per_l <= 0) return 0;
    const long nb = (rows_per_l + 3) / 4;
    const DetWs ws = spe_detws();
    DET_CHECK(ws, L, nb, 1);
    hipLaunchKernelGGL(focal_kernel, dim3((unsigned)nb, (unsigned)L), dim3(256), 0, st, logits, tclass, roww, grad, loss,
                       argmax, rows_per_l, Kc, alpha, gamma, ws);
    SPE_CHECK_LAUNCH();
    return 0;
}

// Matched-pair box losses: for pair i (prediction row srow[i] of pred_boxes, target box tbox[i],
// weight w[i] or 1, layer lidx[i]):  sums[l][0] += w*L1, sums[l][1] += w*(1-GIoU); the gradients
// w.r.t. the predicted cxcywh box are written to g_l1[i][4], g_giou[i][4].  One workgroup (a few hundred pairs): the per-pair
// values pass through LDS in chunks of 1024 and thread (l, k) adds the pairs of layer l in pair order - no atomics.
__device__ __forceinline__ void box_pair(const float* __restrict__ pred_boxes, const long* __restrict__ srow,
                                         const float* __restrict__ tbox, const float* __restrict__ w,
                                         float* __restrict__ g_l1, float* __restrict__ g_giou, long i, float& o_l1, float& o_giou) {
    const float* s = pred_boxes + srow[i] * 4;
    const float* t = tbox + i * 4;
    const float wt = w ? w[i] : 1.f;
    float l1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float d = s[j] - t[j];
        l1 += fabsf(d);
        g_l1[i * 4 + j] = wt * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
    const Box a = to_xyxy(s), b = to_xyxy(t);
    // value
    const float bw = a.x1 - a.x0, bh = a.y1 - a.y0;
    const float area1 = bw * bh, area2 = (b.x1 - b.x0) * (b.y1 - b.y0);
    const float iwr = fminf(a.x1, b.x1) - fmaxf(a.x0, b.x0), ihr = fminf(a.y1, b.y1) - fmaxf(a.y0, b.y0);
    const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
    const float inter = iw * ih, uni = area1 + area2 - inter, iou = inter / uni;
    const float ewr = fmaxf(a.x1, b.x1) - fminf(a.x0, b.x0), ehr = fmaxf(a.y1, b.y1) - fminf(a.y0, b.y0);
    const float ew = fmaxf(ewr, 0.f), eh = fmaxf(ehr, 0.f);
    const float ea = ew * eh;
    const float giou = iou - (ea - uni) / ea;
    // derivatives w.r.t. (x0,y0,x1,y1) of the predicted box; ties split 1/2 like torch.maximum/minimum
    auto sel_lt = [](float u, float v) { return u < v ? 1.f : (u == v ? 0.5f : 0.f); };   // d min(u,v)/du
    auto sel_gt = [](float u, float v) { return u > v ? 1.f : (u == v ? 0.5f : 0.f); };   // d max(u,v)/du
    const float on_iw = iwr >= 0.f ? 1.f : 0.f, on_ih = ihr >= 0.f ? 1.f : 0.f;
    const float on_ew = ewr >= 0.f ? 1.f : 0.f, on_eh = ehr >= 0.f ? 1.f : 0.f;
    float d_area[4] = {-bh, -bw, bh, bw};
    float d_iw[4] = {-on_iw * sel_gt(a.x0, b.x0), 0.f, on_iw * sel_lt(a.x1, b.x1), 0.f};
    float d_ih[4] = {0.f, -on_ih * sel_gt(a.y0, b.y0), 0.f, on_ih * sel_lt(a.y1, b.y1)};
    float d_ew[4] = {-on_ew * sel_lt(a.x0, b.x0), 0.f, on_ew * sel_gt(a.x1, b.x1), 0.f};
    float d_eh[4] = {0.f, -on_eh * sel_lt(a.y0, b.y0), 0.f, on_eh * sel_gt(a.y1, b.y1)};
    float dg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float d_inter = ih * d_iw[j] + iw * d_ih[j];
        const float d_uni = d_area[j] - d_inter;
        const float d_iou = (d_inter * uni - inter * d_uni) / (uni * uni);
        const float d_ea = eh * d_ew[j] + ew * d_eh[j];
        // giou = iou - (ea - uni)/ea  ->  d = d_iou - (d_ea - d_uni)/ea + (ea - uni) d_ea / ea^2
        dg[j] = d_iou - (d_ea - d_uni) / ea + (ea - uni) * d_ea / (ea * ea);
    }
    // loss = w (1 - giou); chain to cxcywh
    g_giou[i * 4 + 0] = -wt * (dg[0] + dg[2]);
    g_giou[i * 4 + 1] = -wt * (dg[1] + dg[3]);
    g_giou[i * 4 + 2] = -wt * 0.5f * (dg[2] - dg[0]);
    g_giou[i * 4 + 3] = -wt * 0.5f * (dg[3] - dg[1]);
    o_l1 = wt * l1;
    o_giou = wt * (1.f - giou);
}
__global__ __launch_bounds__(1024) void box_loss_kernel(const float* __restrict__ pred_boxes, const long* __restrict__ srow,
                                                        const float* __restrict__ tbox, const float* __restrict__ w,
                                                        const int* __restrict__ lidx, float* __restrict__ sums,
                                                        float* __restrict__ g_l1, float* __restrict__ g_giou, long n, int L) {
    __shared__ float val[1024][2];
    __shared__ int lay[1024];
    const int t = threadIdx.x;
    float mine = 0.f;                                   // thread t < 2L: sums[t >> 1][t & 1]
    for (long c0 = 0; c0 < n; c0 += 1024) {
        const long i = c0 + t;
        if (i < n) {
            float a, b;
            box_pair(pred_boxes, srow, tbox, w, g_l1, g_giou, i, a, b);
            val[t][0] = a; val[t][1] = b; lay[t] = lidx[i];
        }
        __syncthreads();
        if (t < 2 * L) {
            const int l = t >> 1, k = t & 1, cnt = (int)min((long)1024, n - c0);
            for (int j = 0; j < cnt; ++j) if (lay[j] == l) mine += val[j][k];
        }
        __syncthreads();
    }
    if (t < 2 * L) sums[t] += mine;
}
extern "C" int spe_box_loss(const float* pred_boxes, const long* srow, const float* tbox, const float* w, const int* lidx,
                            float* sums, float* g_l1, float* g_giou, long n, int L, hipStream_t st) {
    if (n <= 0 || L <= 0) return 0;
    if (L > 512) return -2;
    hipLaunchKernelGGL(box_loss_kernel, dim3(1), dim3(1024), 0, st, pred_boxes, srow, tbox, w, lidx, sums, g_l1, g_giou, n, L);
    SPE_CHECK_LAUNCH();
    return 0;
}

// dpred[srow[i]][:] += c1[lidx[i]] * g_l1[i][:] + c2[lidx[i]] * g_giou[i][:]   (scatter-add of the
// matched-row gradients; c1/c2 = upstream grad / num_boxes per layer).  A prediction row is matched at most once per
// criterion call (one-to-one assignment per layer and image), so every address receives a single addend and the result does
// not depend on the order; the atomic only keeps a hypothetical duplicate row correct.
__global__ __launch_bounds__(256) void box_loss_bwd_kernel(const long* __restrict__ srow, const int* __restrict__ lidx,
                                                           const float* __restrict__ g_l1, const float* __restrict__ g_giou,
                                                           const float* __restrict__ c1, const float* __restrict__ c2,
                                                           float* __restrict__ dpred, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 4) return;
    const long pi = i >> 2; const int j = (int)(i & 3);
    atomicAdd(dpred + srow[pi] * 4 + j, c1[lidx[pi]] * g_l1[i] + c2[lidx[pi]] * g_giou[i]);
}
extern "C" int spe_box_loss_bwd(const long* srow, const int* lidx, const float* g_l1, const float* g_giou, const float* c1,
                                const float* c2, float* dpred, long n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(box_loss_bwd_kernel, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, st, srow, lidx, g_l1, g_giou, c1,
                       c2, dpred, n);
    SPE_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------------------
// Per-class greedy NMS of one image's detections (reference engine_loc.py:154-174:
// for every predicted class, torchvision.ops.nms(boxes, scores, 0.5) and concatenation in ascending class order).
// Input: n detections ALREADY ordered by (label ascending, score descending) - exactly the order the reference's
// concatenated result has.  keep[i] = 1 iff detection i survives: it is visited in order and suppresses every later
// detection of the same class with IoU > thr (areas (x1-x0)*(y1-y0), intersection extents clamped at 0 - the
// arithmetic of torchvision's nms kernel).  One workgroup per image: thread j owns detection j (+256k), the visit
// order is serial, each visit is one parallel step.
// ------------------------------------------------------------------------------------------
#define NMS_MAXN 4096
__global__ __launch_bounds__(256) void nms_kernel(const float* __restrict__ boxes, const long* __restrict__ labels,
                                                  const int* __restrict__ counts, unsigned char* __restrict__ keep, int nmax, float thr) {
    __shared__ unsigned char dead[NMS_MAXN];
    const int img = blockIdx.x;
    const int n = counts ? counts[img] : nmax;
    const float* bx = boxes + (long)img * nmax * 4;
    const long* lb = labels + (long)img * nmax;
    unsigned char* kp = keep + (long)img * nmax;
    for (int j = threadIdx.x; j < n; j += 256) dead[j] = 0;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        if (dead[i]) continue;                               // uniform: every thread reads the same LDS byte
        const float x0 = bx[i * 4], y0 = bx[i * 4 + 1], x1 = bx[i * 4 + 2], y1 = bx[i * 4 + 3];
        const float ai = (x1 - x0) * (y1 - y0);
        const long li = lb[i];
        for (int j = i + 1 + threadIdx.x; j < n && lb[j] == li; j += 256) {
            const float u0 = bx[j * 4], v0 = bx[j * 4 + 1], u1 = bx[j * 4 + 2], v1 = bx[j * 4 + 3];
            const float w = fmaxf(fminf(x1, u1) - fmaxf(x0, u0), 0.f), h = fmaxf(fminf(y1, v1) - fmaxf(y0, v0), 0.f);
            const float inter = w * h, aj = (u1 - u0) * (v1 - v0);
            if (inter / (ai + aj - inter) > thr) dead[j] = 1;
        }
        __syncthreads();
    }
    for (int j = threadIdx.x; j < nmax; j += 256) kp[j] = (j < n && !dead[j]) ? 1 : 0;
}

// C-ABI: see include/spe_hip.h (spe_nms_sorted).  -2: more than NMS_MAXN detections per image.
extern "C" int spe_nms_sorted(const float* boxes, const long* labels, const int* counts, unsigned char* keep, int nimg, int nmax,
                              float iou_threshold, hipStream_t st) {
    if (nimg <= 0 || nmax <= 0) return 0;
    if (nmax > NMS_MAXN) return -2;
    hipLaunchKernelGGL(nms_kernel, dim3(nimg), dim3(256), 0, st, boxes, labels, counts, keep, nmax, iou_threshold);
    SPE_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// One-to-many target jitter (reference models/conditional_detr.py:409-431: every GT box becomes `ratio` rows - up to ratio - 1 copies
// scaled by uniform factors in [1 - j, 1 + j], each the first of <= 1000 attempts whose IoU with the original exceeds 0.7, the original
// last).  The host-side mirror draws the 1000 candidate scales per box with torch's generator in ONE launch; this kernel does the rest
// (candidate boxes, IoU, the first ratio - 1 kept candidates in attempt order) instead of ~25 elementwise / scan / gather launches
// over [M, 1000, 4] tensors per criterion call.  One wave per box; the arithmetic is written without contraction so that the
// decisions equal the elementwise composition's.
__global__ __launch_bounds__(64) void jitter_pick_kernel(const float* __restrict__ box, const float* __restrict__ scale, float* __restrict__ out,
                                                         int M, int ncand, int ratio) {
    const int m = blockIdx.x, lane = threadIdx.x;
    if (m >= M) return;
    const float b0 = box[m * 4], b1 = box[m * 4 + 1], b2 = box[m * 4 + 2], b3 = box[m * 4 + 3];
    const float bx0 = __fsub_rn(b0, __fmul_rn(0.5f, b2)), by0 = __fsub_rn(b1, __fmul_rn(0.5f, b3));
    const float bx1 = __fadd_rn(b0, __fmul_rn(0.5f, b2)), by1 = __fadd_rn(b1, __fmul_rn(0.5f, b3));
    const float area_b = __fmul_rn(__fsub_rn(bx1, bx0), __fsub_rn(by1, by0));
    float* o = out + (long)m * ratio * 4;
    int found = 0;                                       // kept candidates so far (wave-uniform)
    for (int c0 = 0; c0 < ncand && found < ratio - 1; c0 += 64) {
        const int c = c0 + lane;
        bool keep = false;
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
        if (c < ncand) {
            const float4 sc = *reinterpret_cast<const float4*>(scale + ((long)m * ncand + c) * 4);
            q0 = __fmul_rn(sc.x, b0); q1 = __fmul_rn(sc.y, b1); q2 = __fmul_rn(sc.z, b2); q3 = __fmul_rn(sc.w, b3);
            const float ax0 = __fsub_rn(q0, __fmul_rn(0.5f, q2)), ay0 = __fsub_rn(q1, __fmul_rn(0.5f, q3));
            const float ax1 = __fadd_rn(q0, __fmul_rn(0.5f, q2)), ay1 = __fadd_rn(q1, __fmul_rn(0.5f, q3));
            const float w = fmaxf(__fsub_rn(fminf(ax1, bx1), fmaxf(ax0, bx0)), 0.f), h = fmaxf(__fsub_rn(fminf(ay1, by1), fmaxf(ay0, by0)), 0.f);
            const float inter = __fmul_rn(w, h);
            const float area_a = __fmul_rn(__fsub_rn(ax1, ax0), __fsub_rn(ay1, ay0));
            const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_a, area_b), inter));
            keep = iou > 0.7f;
        }
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const int rank = found + __popcll(mask & ((1ull << lane) - 1ull));       // 0-based rank among the kept candidates
            if (rank < ratio - 1) { o[rank * 4] = q0; o[rank * 4 + 1] = q1; o[rank * 4 + 2] = q2; o[rank * 4 + 3] = q3; }
        }
        found += __popcll(mask);
    }
    if (found > ratio - 1) found = ratio - 1;
    // picks without a kept candidate, and the last row: the original box
    for (int s = found + lane; s < ratio; s += 64) { o[s * 4] = b0; o[s * 4 + 1] = b1; o[s * 4 + 2] = b2; o[s * 4 + 3] = b3; }
}

// C-ABI: see include/spe_hip.h
extern "C" int spe_jitter_pick(const float* box, const float* scale, float* out, int M, int ncand, int ratio, hipStream_t st) {
    if (M <= 0) return 0;
    if (ratio < 1 || ncand < 0 || (reinterpret_cast<uintptr_t>(scale) & 15)) return -2;
    hipLaunchKernelGGL(jitter_pick_kernel, dim3(M), dim3(64), 0, st, box, scale, out, M, ncand, ratio);
    SPE_CHECK_LAUNCH();
    return 0;
}
