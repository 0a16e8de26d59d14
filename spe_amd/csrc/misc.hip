// Small gather / elementwise kernels around the GEMMs.
#include "common.h"
#include "det_reduce.h"

// Patch gather for the 16x16/stride-16 patch-embed convolution (reference models/cait.py:518-528,
// timm PatchEmbed.proj = Conv2d(3,C,16,16)): img[B,Cin,Hi,Wi] -> cols[B*h*w, Cin*P*P] with the
// column order (c, py, px) of the flattened conv weight, so the conv becomes one GEMM.
// Each lane moves one float4 = 4 horizontally adjacent pixels.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, float* __restrict__ cols, int B, int Cin,
                                                       int Hi, int Wi, int P, int h, int w) {
    const int P4 = P >> 2;
    const long per_row = (long)Cin * P * P4;                 // float4 per output row
    const long total = (long)B * h * w * per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / per_row; int r = (int)(i % per_row);
        const int c = r / (P * P4); r %= P * P4;
        const int py = r / P4, px4 = r % P4;
        const int b = (int)(row / (h * w)); const int pr = (int)(row % (h * w));
        const int ph = pr / w, pw = pr % w;
        const float* src = img + (((long)b * Cin + c) * Hi + (ph * P + py)) * Wi + pw * P + px4 * 4;
        float4 v;
        if ((Wi & 3) == 0) v = *reinterpret_cast<const float4*>(src);
        else { v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3]; }
        reinterpret_cast<float4*>(cols)[i] = v;
    }
}
extern "C" int spe_patchify(const float* img, float* cols, int B, int Cin, int Hi, int Wi, int P, hipStream_t st) {
    if (P & 3) return -2;
    const int h = Hi / P, w = Wi / P;
    const long total = (long)B * h * w * Cin * P * (P / 4);
    if (total <= 0) return 0;
    long nb = (total + 255) / 256; if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)nb), dim3(256), 0, st, img, cols, B, Cin, Hi, Wi, P, h, w);
    SPE_CHECK_LAUNCH();
    return 0;
}

// out[r][c] = a[r][c] + b[(r % rb)][c]   (adds a broadcast table: pos-embed / bias rows), float4.
__global__ __launch_bounds__(256) void add_rows_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                       float4* __restrict__ out, long n4, long period4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 x = a[i], y = b[i % period4];
        out[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}
extern "C" int spe_add_rows(const float* a, const float* b, float* out, long n, long period, hipStream_t st) {
    if (n <= 0) return 0;
    if ((n & 3) || (period & 3)) return -2;
    long nb = (n / 4 + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const float4*)a, (const float4*)b, (float4*)out,
                       n / 4, period / 4);
    SPE_CHECK_LAUNCH();
    return 0;
}

// D[b][h][q] = sum_d x[b][q][h][d] * y[b][q][h][d] for contiguous [B, L, H, dh] tensors - the row term rowsum(dO . O) of a softmax backward
// (reference: the autograd of models/attention.py:277-383's softmax), one thread per (b, q, h); replaces an ATen mul + sum + permuted copy.
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ D, int B, int L, int H, int dh) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;          // over B * L * H, h fastest (the inputs' order)
    if (i >= (long)B * L * H) return;
    const int h = (int)(i % H); const long bq = i / H; const int q = (int)(bq % L), b = (int)(bq / L);
    const float* xp = x + i * dh; const float* yp = y + i * dh;
    float acc = 0.f;
    if ((dh & 3) == 0) {
        for (int d = 0; d < dh; d += 4) {
            const float4 u = *reinterpret_cast<const float4*>(xp + d), v = *reinterpret_cast<const float4*>(yp + d);
            acc = fmaf(u.x, v.x, acc); acc = fmaf(u.y, v.y, acc); acc = fmaf(u.z, v.z, acc); acc = fmaf(u.w, v.w, acc);
        }
    } else {
        for (int d = 0; d < dh; ++d) acc = fmaf(xp[d], yp[d], acc);
    }
    D[((long)b * H + h) * L + q] = acc;
}
extern "C" int spe_rowdot(const float* x, const float* y, float* D, int B, int L, int H, int dh, hipStream_t st) {
    const long n = (long)B * L * H;
    if (n <= 0) return 0;
    if (dh <= 0 || ((dh & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15))) return -2;
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, D, B, L, H, dh);
    SPE_CHECK_LAUNCH();
    return 0;
}

// Bicubic (A = -0.75, align_corners = false) resize of the learned position-embedding grid, token-major:
// in[gh*gw][C] -> out[h*w][C]  (reference models/cait.py:598-613, F.interpolate(mode='bicubic')).
__device__ __forceinline__ void cubic_w(float t, float w[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x3 = 2.f - t, x2 = 1.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}
__global__ __launch_bounds__(256) void bicubic_kernel(const float* __restrict__ src, float* __restrict__ dst, int gh, int gw, int h,
                                                      int w, int C) {
    const int C4 = C >> 2;
    const long total = (long)h * w * C4;
    const float sy = (float)gh / (float)h, sx = (float)gw / (float)w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % C4); const int o = (int)(i / C4);
        const int oy = o / w, ox = o % w;
        const float fy = sy * (oy + 0.5f) - 0.5f, fx = sx * (ox + 0.5f) - 0.5f;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        float wy[4], wx[4];
        cubic_w(fy - iy, wy); cubic_w(fx - ix, wx);
        {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int yy = min(max(iy - 1 + a, 0), gh - 1);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int xx = min(max(ix - 1 + b, 0), gw - 1);
                    const float4 v = reinterpret_cast<const float4*>(src + ((long)yy * gw + xx) * C)[c4];
                    const float wt = wy[a] * wx[b];
                    acc.x += wt * v.x; acc.y += wt * v.y; acc.z += wt * v.z; acc.w += wt * v.w;
                }
            }
            reinterpret_cast<float4*>(dst + (long)o * C)[c4] = acc;
        }
    }
}
// Backward as a GATHER (no atomics, fixed order): one thread per (source cell, channel quad) adds up, row by row, the output
// pixels whose 4 x 4 footprint - with the border clamp of the forward - contains the cell.  Output rows that can reach source row
// yy lie in a window of ~4 / sy rows around it (the clamp only adds rows next to the border, which the window's limits keep).
__global__ __launch_bounds__(256) void bicubic_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int gh, int gw, int h,
                                                          int w, int C) {
    const int C4 = C >> 2;
    const long total = (long)gh * gw * C4;
    const float sy = (float)gh / (float)h, sx = (float)gw / (float)w;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4); const int cell = (int)(i / C4);
    const int yy = cell / gw, xx = cell % gw;
    auto lo = [](int v, float s, int n) { return max(0, (int)floorf((v - 2 + 0.5f) / s - 0.5f) - 1); };
    auto hi = [](int v, float s, int n) { return min(n - 1, (int)ceilf((v + 2 + 0.5f) / s - 0.5f) + 1); };
    const int oy0 = lo(yy, sy, h), oy1 = hi(yy, sy, h), ox0 = lo(xx, sx, w), ox1 = hi(xx, sx, w);
    // weight of output coordinate o on source coordinate v along one axis: the taps that land on v after clamping
    auto axis_w = [](int o, float s, int v, int n) {
        const float f = s * (o + 0.5f) - 0.5f;
        const int i0 = (int)floorf(f);
        float wt[4];
        cubic_w(f - i0, wt);
        float r = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) if (min(max(i0 - 1 + a, 0), n - 1) == v) r += wt[a];
        return r;
    };
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = oy0; oy <= oy1; ++oy) {
        const float wyv = axis_w(oy, sy, yy, gh);
        if (wyv == 0.f) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            const float wt = wyv * axis_w(ox, sx, xx, gw);
            if (wt == 0.f) continue;
            const float4 g = reinterpret_cast<const float4*>(dout + ((long)oy * w + ox) * C)[c4];
            acc.x += wt * g.x; acc.y += wt * g.y; acc.z += wt * g.z; acc.w += wt * g.w;
        }
    }
    float4* d = reinterpret_cast<float4*>(din + (long)cell * C) + c4;      // += : the destination holds the running gradient
    const float4 o = *d;
    *d = make_float4(o.x + acc.x, o.y + acc.y, o.z + acc.z, o.w + acc.w);
}
extern "C" int spe_bicubic(const float* src, float* dst, int gh, int gw, int h, int w, int C, int backward, hipStream_t st) {
    if (C & 3) return -2;
    const long total = (long)h * w * (C / 4);
    if (total <= 0) return 0;
    long nb = (total + 255) / 256; if (nb > 4096) nb = 4096;
    if (backward) {
        const long cells = (long)gh * gw * (C / 4);
        hipLaunchKernelGGL(bicubic_bwd_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, src, dst, gh, gw, h, w, C);
    } else hipLaunchKernelGGL(bicubic_kernel, dim3((unsigned)nb), dim3(256), 0, st, src, dst, gh, gw, h, w, C);
    SPE_CHECK_LAUNCH();
    return 0;
}

// ---- diagnostic: occupy `nwg` workgroup slots for `micros` microseconds, optionally streaming copies through buf (two halves:
// the workgroups read one and write the other, over and over) - a stand-in for the CU and HBM share an RCCL ring takes
// (channels x one persistent workgroup each) on a box with one GPU.  Used by tools/dp_proxy.py and its test to measure how much
// the hand-balanced attention / GEMM grids lose when a collective runs beside them (reference main.py:172: DDP overlaps its
// all-reduce with the backward); never launched by the product.
__global__ __launch_bounds__(256) void occupy_kernel(long ticks, float4* __restrict__ buf, long n4_half) {
    const long t0 = wall_clock64();                     // constant 100 MHz counter
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        if (buf) {
#pragma unroll 4
            for (int k = 0; k < 16; ++k) {
                if (i >= n4_half) i -= n4_half;
                buf[n4_half + i] = buf[i];
                i += (long)gridDim.x * 256;
            }
        } else {
            __builtin_amdgcn_s_sleep(8);
        }
    }
}
extern "C" int spe_occupy(int nwg, long micros, float* buf, long buf_floats, hipStream_t st) {
    if (nwg <= 0 || micros <= 0) return 0;
    const long n4_half = buf ? buf_floats / 8 : 0;
    if (buf && ((reinterpret_cast<uintptr_t>(buf) & 15) || n4_half < (long)nwg * 256)) return -2;
    hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)nwg), dim3(256), 0, st, micros * 100, reinterpret_cast<float4*>(buf), n4_half);
    SPE_CHECK_LAUNCH();
    return 0;
}

// ---- reduction workspace of the deterministic cross-workgroup sums (det_reduce.h) -------------------------------------
static DetWs g_detws = {nullptr, nullptr, 0, 0, nullptr};
DetWs spe_detws() { return g_detws; }
// C-ABI: see include/spe_hip.h.  The first 64 KiB hold the tickets (zeroed here, on `st`), the rest the partial-sum slabs.
extern "C" int spe_set_reduce_workspace(void* ws, size_t bytes, hipStream_t st) {
    if (!ws) { g_detws = DetWs{nullptr, nullptr, 0, 0, nullptr}; return 0; }
    const size_t tbytes = 64 * 1024;
    if ((reinterpret_cast<uintptr_t>(ws) & 255) || bytes < tbytes + (1u << 20)) return -2;
    hipError_t e = hipMemsetAsync(ws, 0, tbytes, st);
    if (e != hipSuccess) return (int)e;
    g_detws.tickets = reinterpret_cast<unsigned*>(ws);
    g_detws.ntickets = (int)(tbytes / sizeof(unsigned));
    g_detws.slab = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + tbytes);
    g_detws.slab_floats = (long)((bytes - tbytes) / sizeof(float));
    return 0;
}

// ---- deferred reductions (det_reduce.h): arena, pending table, flush kernel ------------------------------------------------
#define DEFER_MAXENT 24
struct DeferEntry { const float* src; int members, L, sets, seg0, nseg, accumulate; };
struct DeferTable { DeferEntry e[DEFER_MAXENT]; DetDeferSeg seg[DET_DEFER_MAXSEG]; int n; };
static DeferTable g_dt = {};
static int g_dt_nseg = 0;
static float* g_arena = nullptr; static long g_arena_floats = 0, g_arena_cur = 0;
#define DEFER_MAXRANGE 64
static const char* g_rng_lo[DEFER_MAXRANGE]; static const char* g_rng_hi[DEFER_MAXRANGE]; static int g_nrange = 0;

// entry blockIdx.y: output o = blockIdx.x * 16 + (threadIdx.x >> 4); its 16 lanes take members lane, lane + 16, .. (all loads in flight),
// the lanes are added in lane order: a fixed order, whatever the schedule
__global__ __launch_bounds__(256) void reduce_flush_kernel(DeferTable t) {
    const DeferEntry e = t.e[blockIdx.y];
    const long total = (long)e.sets * e.L;
    const long o = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (o >= total) return;                     // whole 16-lane groups leave together
    const int ml = threadIdx.x & 15;
    const int set = (int)(o / e.L), c = (int)(o - (long)set * e.L);
    const float* p = e.src + ((long)set * e.members) * e.L + c;
    float s = 0.f;
    for (int m = ml; m < e.members; m += 16) s += p[(long)m * e.L];
    // lanes 0..15 in order
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += __shfl(s, (threadIdx.x & 48) + k, 64);
    if (ml != 0) return;
    long off = o;
    for (int k = 0; k < e.nseg; ++k) {
        const DetDeferSeg sg = t.seg[e.seg0 + k];
        if (off < sg.len) { if (sg.dst) sg.dst[off] = e.accumulate ? sg.dst[off] + tot : tot; return; }
        off -= sg.len;
    }
}

static int defer_flush(hipStream_t st) {
    if (g_dt.n == 0) { g_arena_cur = 0; g_dt_nseg = 0; return 0; }
    long maxo = 0;
    for (int i = 0; i < g_dt.n; ++i) { const long o = (long)g_dt.e[i].sets * g_dt.e[i].L; if (o > maxo) maxo = o; }
    hipLaunchKernelGGL(reduce_flush_kernel, dim3((unsigned)((maxo + 15) / 16), (unsigned)g_dt.n), dim3(256), 0, st, g_dt);
    g_dt.n = 0; g_dt_nseg = 0; g_arena_cur = 0;
    SPE_CHECK_LAUNCH();
    return 0;
}
float* det_defer_try(long sets, long members, long L, int nseg, const DetDeferSeg* segs, hipStream_t st) {
    if (!g_nrange || !g_arena || members <= 1 || nseg > DET_DEFER_MAXSEG) return nullptr;
    for (int k = 0; k < nseg; ++k) {              // every destination inside a registered range (NULL: not wanted, nothing to write)
        const char* d = reinterpret_cast<const char*>(segs[k].dst);
        if (!d) continue;
        bool in = false;
        for (int r = 0; r < g_nrange && !in; ++r) in = d >= g_rng_lo[r] && d + (long)segs[k].len * 4 <= g_rng_hi[r];
        if (!in) return nullptr;
    }
    const long need = (sets * members * L + 63) & ~63L;
    if (need > g_arena_floats) return nullptr;
    // one producer per destination and flush: the flush kernel adds every pending entry onto its destination with a plain read-modify-write,
    // one entry per blockIdx.y - a new entry whose destination overlaps a pending one (two launches into the same gradient, or an
    // accumulate = 0 entry mixed with an accumulate = 1 one) would race with it, so the pending ones are flushed first (stream-ordered)
    bool overlap = false;
    for (int k = 0; k < nseg && !overlap; ++k) {
        const char* lo = reinterpret_cast<const char*>(segs[k].dst);
        if (!lo) continue;
        const char* hi = lo + (long)segs[k].len * 4;
        for (int j = 0; j < g_dt_nseg && !overlap; ++j) {
            const char* plo = reinterpret_cast<const char*>(g_dt.seg[j].dst);
            if (plo && lo < plo + (long)g_dt.seg[j].len * 4 && plo < hi) overlap = true;
        }
    }
    if (overlap && defer_flush(st) != 0) return nullptr;
    if (g_dt.n >= DEFER_MAXENT || g_dt_nseg + nseg > DET_DEFER_MAXSEG || g_arena_cur + need > g_arena_floats) {
        if (defer_flush(st) != 0) return nullptr;      // stream-ordered before this launch reuses the arena
    }
    return g_arena + g_arena_cur;
}
void det_defer_commit(float* region, long sets, long members, long L, int nseg, const DetDeferSeg* segs, int accumulate) {
    DeferEntry& e = g_dt.e[g_dt.n++];
    e.src = region; e.members = (int)members; e.L = (int)L; e.sets = (int)sets; e.seg0 = g_dt_nseg; e.nseg = nseg; e.accumulate = accumulate;
    for (int k = 0; k < nseg; ++k) g_dt.seg[g_dt_nseg++] = segs[k];
    g_arena_cur += (sets * members * L + 63) & ~63L;
}
// C-ABI: see include/spe_hip.h
extern "C" int spe_reduce_defer_arena(void* arena, size_t bytes) {
    if (g_dt.n) return -3;                      // pending sums: flush first
    g_arena = reinterpret_cast<float*>(arena); g_arena_floats = arena ? (long)(bytes / sizeof(float)) : 0; g_arena_cur = 0;
    if (arena && (reinterpret_cast<uintptr_t>(arena) & 255)) { g_arena = nullptr; g_arena_floats = 0; return -2; }
    return 0;
}
extern "C" int spe_reduce_defer_ranges(const void* const* ptrs, const size_t* bytes, int n) {
    if (g_dt.n) return -3;
    if (n < 0 || n > DEFER_MAXRANGE) return -2;
    for (int i = 0; i < n; ++i) { g_rng_lo[i] = reinterpret_cast<const char*>(ptrs[i]); g_rng_hi[i] = g_rng_lo[i] + bytes[i]; }
    g_nrange = n;
    return 0;
}
extern "C" int spe_reduce_flush(hipStream_t st) { return defer_flush(st); }
extern "C" int spe_reduce_pending(void) { return g_dt.n; }

extern "C" int spe_abi_version(void) { return 7; }    // 2: round 2 (signatures of spe_hungarian, spe_adamw_flat, spe_layernorm_fwd, spe_attn_contract, spe_talking_fused_plan changed; new entry points)


// ------------------------------------------------------------------------------------------
// Sine position embedding of the padded feature map (reference models/position_encoding.py:37-57, normalize = True):
// out[b][y][x][k] for k < npf encodes the row coordinate, k >= npf the column coordinate:
//   e = (count of non-padded cells up to and including this one along the axis) / (count along the whole axis + eps) * scale
//   value = sin(e / dim_t[k']) for even k', cos(e / dim_t[k']) for odd k'   (dim_t = temperature^(2*(k'/2)/npf), passed in)
// One thread per (b, y, x, feature pair); the two prefix counts are recomputed from the mask (<= h + w byte loads).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pos_sine_kernel(const unsigned char* __restrict__ mask, const float* __restrict__ dim_t,
                                                       float* __restrict__ out, int B, int h, int w, int npf, float scale, float eps,
                                                       int normalize) {
    const int half = npf / 2;
    const long total = (long)B * h * w * half;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int kp = (int)(i % half); long t = i / half;
        const int x = (int)(t % w); t /= w;
        const int y = (int)(t % h); const int b = (int)(t / h);
        const unsigned char* m = mask + (long)b * h * w;
        int cy = 0, ty = 0, cx = 0, tx = 0;
        for (int yy = 0; yy < h; ++yy) { const int v = m[yy * w + x] ? 0 : 1; ty += v; if (yy <= y) cy += v; }
        for (int xx = 0; xx < w; ++xx) { const int v = m[y * w + xx] ? 0 : 1; tx += v; if (xx <= x) cx += v; }
        float ey = (float)cy, ex = (float)cx;
        if (normalize) { ey = ey / ((float)ty + eps) * scale; ex = ex / ((float)tx + eps) * scale; }
        const float d0 = dim_t[2 * kp], d1 = dim_t[2 * kp + 1];
        float* o = out + (((long)b * h + y) * w + x) * (2 * npf);
        o[2 * kp] = sinf(ey / d0); o[2 * kp + 1] = cosf(ey / d1);
        o[npf + 2 * kp] = sinf(ex / d0); o[npf + 2 * kp + 1] = cosf(ex / d1);
    }
}

// C-ABI: see include/spe_hip.h (spe_pos_sine).  npf even.
extern "C" int spe_pos_sine(const void* mask_u8, const float* dim_t, float* out, int B, int h, int w, int npf, float scale,
                            float eps, int normalize, hipStream_t st) {
    if (B <= 0 || h <= 0 || w <= 0 || npf <= 0) return 0;
    if (npf & 1) return -2;
    const long total = (long)B * h * w * (npf / 2);
    long nb = (total + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(pos_sine_kernel, dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const unsigned char*>(mask_u8), dim_t,
                       out, B, h, w, npf, scale, eps, normalize);
    SPE_CHECK_LAUNCH();
    return 0;
}


// fp32 [R, C] (row stride ldx) -> IEEE fp16 [R, C] (row stride ldo), saturating at +-65504, NaN kept: the single-term operands of
// the decoder's memory-side projections (spe_gemm_bf16nt with act bit 8; reference models/transformer.py:389-396).
__global__ __launch_bounds__(256) void cvt_f16_kernel(const float* __restrict__ x, long ldx, int R, int C4, unsigned short* __restrict__ out, long ldo) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)R * C4; i += (long)gridDim.x * 256) {
        const int r = (int)(i / C4), c = (int)(i % C4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + (long)r * ldx + c);
        typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
        h4_t h;
        h[0] = (_Float16)((v.x != v.x) ? v.x : __builtin_amdgcn_fmed3f(v.x, -65504.f, 65504.f)); h[1] = (_Float16)((v.y != v.y) ? v.y : __builtin_amdgcn_fmed3f(v.y, -65504.f, 65504.f));
        h[2] = (_Float16)((v.z != v.z) ? v.z : __builtin_amdgcn_fmed3f(v.z, -65504.f, 65504.f)); h[3] = (_Float16)((v.w != v.w) ? v.w : __builtin_amdgcn_fmed3f(v.w, -65504.f, 65504.f));
        *reinterpret_cast<uint2*>(out + (long)r * ldo + c) = __builtin_bit_cast(uint2, h);
    }
}
extern "C" int spe_cvt_f16(const float* x, long ldx, int R, int C, void* out, long ldo, hipStream_t st) {
    if (R <= 0 || C <= 0) return 0;
    if ((C & 3) || (ldx & 3) || (ldo & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 7)) return -2;
    const long n = (long)R * (C / 4);
    long nb = (n + 255) / 256; if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(cvt_f16_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, ldx, R, C / 4, reinterpret_cast<unsigned short*>(out), ldo);
    SPE_CHECK_LAUNCH();
    return 0;
}
