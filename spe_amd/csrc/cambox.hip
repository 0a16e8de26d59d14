// CAM -> pseudo boxes: reference cams_deit.py:9-13 (`resize_cam`: cv2.resize bilinear,
// min-max normalisation) and cams_deit.py:61-96 (`get_multi_bboxes`: uint8 quantisation, THRESH_TOZERO at
// int(cam_thr * max), cv2.findContours(RETR_TREE), contourArea, boundingRect, keep area >= ratio * largest), driven
// per (image, present class) by engine.py:356-398.  OpenCV is a third-party dependency that is absent here: its
// published algorithms are restated (bilinear with half-pixel centres and replicated borders; Suzuki-Abe border
// following, 8-connected, outer and hole borders; shoelace polygon area; inclusive bounding rectangle).
//
// Device: spe_cam_prepare turns M class maps [h, w] into M thresholded uint8 images [rows, cols] (resize, min/max,
// normalise, quantise, threshold) in two launches - that is 1 M pixels per map at 800x1333, the part worth a GPU.
// Host: spe_cam_contour_boxes follows the borders of one thresholded image and selects the boxes (serial, ~10^4
// border pixels per map; the reference does this on the host too).
#include "common.h"
#include <vector>
#include <algorithm>
#include <cmath>

// ---- device ------------------------------------------------------------------------------------
__device__ __forceinline__ float cam_bilinear(const float* __restrict__ src, int h, int w, int r, int c, float sy, float sx) {
    // cv2.resize INTER_LINEAR: source coordinate (d + 0.5) * scale - 0.5, floor, replicated border
    float fy = (r + 0.5f) * sy - 0.5f, fx = (c + 0.5f) * sx - 0.5f;
    int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    fy -= y0; fx -= x0;
    if (y0 < 0) { y0 = 0; fy = 0.f; }
    if (y0 >= h - 1) { y0 = h - 1; fy = 0.f; }
    if (x0 < 0) { x0 = 0; fx = 0.f; }
    if (x0 >= w - 1) { x0 = w - 1; fx = 0.f; }
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    // horizontal pass first, then vertical (the order of OpenCV's separable implementation)
    const float t0 = src[y0 * w + x0] * (1.f - fx) + src[y0 * w + x1] * fx;
    const float t1 = src[y1 * w + x0] * (1.f - fx) + src[y1 * w + x1] * fx;
    return t0 * (1.f - fy) + t1 * fy;
}
__device__ __forceinline__ void atomic_minmax(float* mm, float lo, float hi) {
    // float min/max through the ordered-int trick (values may be negative)
    int* mi = reinterpret_cast<int*>(mm);
    const int l = __float_as_int(lo), h = __float_as_int(hi);
    if (l >= 0) atomicMin(mi, l); else atomicMax(reinterpret_cast<unsigned*>(mi), (unsigned)l);
    if (h >= 0) atomicMax(mi + 1, h); else atomicMin(reinterpret_cast<unsigned*>(mi + 1), (unsigned)h);
}
__global__ __launch_bounds__(256) void cam_minmax_kernel(const float* __restrict__ cams, float* __restrict__ mm, int h, int w,
                                                         int rows, int cols) {
    __shared__ float red[16];
    const int m = blockIdx.y;
    const float* src = cams + (long)m * h * w;
    const float sy = (float)h / rows, sx = (float)w / cols;
    float lo = INFINITY, hi = -INFINITY;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)rows * cols; i += (long)gridDim.x * 256) {
        const float v = cam_bilinear(src, h, w, (int)(i / cols), (int)(i % cols), sy, sx);
        lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    lo = -spe_block_max(-lo, red);
    hi = spe_block_max(hi, red);
    if (threadIdx.x == 0) atomic_minmax(mm + 2 * m, lo, hi);
}
__global__ __launch_bounds__(256) void cam_quantise_kernel(const float* __restrict__ cams, const float* __restrict__ mm,
                                                           unsigned char* __restrict__ out, int h, int w, int rows, int cols, float cam_thr) {
    const int m = blockIdx.y;
    const float* src = cams + (long)m * h * w;
    const float sy = (float)h / rows, sx = (float)w / cols;
    const float lo = mm[2 * m], hi = mm[2 * m + 1];
    const float mx = hi - lo;                               // max of (cam - min)
    // after cam / cam.max() the largest pixel is exactly 1 -> 255; a constant map is 0/0 = NaN -> 0 everywhere
    const int top = (mx > 0.f) ? 255 : 0;
    const int thr = (int)((double)cam_thr * (double)top);   // int(cam_thr * np.max(uint8 image))
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)rows * cols; i += (long)gridDim.x * 256) {
        const float v = cam_bilinear(src, h, w, (int)(i / cols), (int)(i % cols), sy, sx);
        int q = 0;
        if (mx > 0.f) { const float n = (v - lo) / mx; q = (int)(n * 255.f); q = max(0, min(255, q)); }
        out[(long)m * rows * cols + i] = (unsigned char)((q > thr) ? q : 0);      // THRESH_TOZERO
    }
}

// C-ABI: see include/spe_hip.h (spe_cam_prepare).  minmax: workspace of 2*M floats.
extern "C" int spe_cam_prepare(const float* cams, int M, int h, int w, int rows, int cols, float cam_thr, float* minmax,
                               unsigned char* out, hipStream_t st) {
    if (M <= 0 || rows <= 0 || cols <= 0) return 0;
    // min slot = +inf, max slot = -inf
    std::vector<float> init(2 * (size_t)M);
    for (int i = 0; i < M; ++i) { init[2 * i] = INFINITY; init[2 * i + 1] = -INFINITY; }
    hipError_t e = hipMemcpyAsync(minmax, init.data(), init.size() * sizeof(float), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(st);                           // `init` is pageable and goes out of scope
    if (e != hipSuccess) return (int)e;
    long nb = ((long)rows * cols + 255) / 256; if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(cam_minmax_kernel, dim3((unsigned)nb, M), dim3(256), 0, st, cams, minmax, h, w, rows, cols);
    SPE_CHECK_LAUNCH();
    hipLaunchKernelGGL(cam_quantise_kernel, dim3((unsigned)nb, M), dim3(256), 0, st, cams, minmax, out, h, w, rows, cols, cam_thr);
    SPE_CHECK_LAUNCH();
    return 0;
}

// ---- host: border following ----------------------------------------------------------------------
namespace {
struct Contour { double area; int x0, y0, x1, y1; };

// 8-neighbourhood in clockwise order starting east (image coordinates: y grows downwards)
const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int DY[8] = {0, 1, 1, 1, 0, -1, -1, -1};

// Suzuki & Abe (1985), algorithm 1, on a zero-padded int image f (1 = foreground); every border (outer and hole) is
// reported once with its shoelace area and inclusive bounding box.
void follow_borders(std::vector<int>& f, int R, int C, std::vector<Contour>& out) {
    const int W = C + 2;
    auto at = [&](int y, int x) -> int& { return f[(size_t)(y + 1) * W + (x + 1)]; };
    int nbd = 1;
    for (int i = 0; i < R; ++i) {
        for (int j = 0; j < C; ++j) {
            const int v = at(i, j);
            if (v == 0) continue;
            int from = -1;                                  // direction index of the start neighbour (i2, j2) seen from (i, j)
            if (v == 1 && at(i, j - 1) == 0) { ++nbd; from = 4; }            // outer border: start neighbour = west
            else if (v >= 1 && at(i, j + 1) == 0) { ++nbd; from = 0; }       // hole border: start neighbour = east
            if (from < 0) continue;
            // (3.1) clockwise around (i, j) starting from the start neighbour: first non-zero pixel
            int d1 = -1;
            for (int k = 0; k < 8; ++k) { const int d = (from + k) & 7; if (at(i + DY[d], j + DX[d]) != 0) { d1 = d; break; } }
            Contour c; c.area = 0.0; c.x0 = c.x1 = j; c.y0 = c.y1 = i;
            if (d1 < 0) { at(i, j) = -nbd; out.push_back(c); continue; }     // isolated pixel
            const int i1 = i + DY[d1], j1 = j + DX[d1];
            int i2 = i1, j2 = j1, i3 = i, j3 = j;
            double acc = 0.0;
            while (true) {
                // (3.3) counter-clockwise around (i3, j3) starting after (i2, j2): first non-zero pixel (i4, j4)
                int dstart = 0;
                for (int d = 0; d < 8; ++d) if (i3 + DY[d] == i2 && j3 + DX[d] == j2) { dstart = d; break; }
                bool east_zero_examined = false;
                int d4 = -1;
                for (int k = 1; k <= 8; ++k) {
                    const int d = (dstart - k) & 7;          // counter-clockwise = decreasing index
                    if (at(i3 + DY[d], j3 + DX[d]) != 0) { d4 = d; break; }
                    if (d == 0) east_zero_examined = true;
                }
                const int i4 = i3 + DY[d4], j4 = j3 + DX[d4];
                // (3.4)
                if (east_zero_examined) at(i3, j3) = -nbd;
                else if (at(i3, j3) == 1) at(i3, j3) = nbd;
                // polygon edge (i3, j3) -> (i4, j4): shoelace with x = column, y = row
                acc += (double)j3 * i4 - (double)j4 * i3;
                c.x0 = std::min(c.x0, j3); c.x1 = std::max(c.x1, j3); c.y0 = std::min(c.y0, i3); c.y1 = std::max(c.y1, i3);
                // (3.5)
                if (i4 == i && j4 == j && i3 == i1 && j3 == j1) break;
                i2 = i3; j2 = j3; i3 = i4; j3 = j4;
            }
            c.area = std::fabs(acc) * 0.5;
            out.push_back(c);
        }
    }
}
}  // namespace

// C-ABI: see include/spe_hip.h (spe_cam_contour_boxes).  HOST function (img and boxes are host pointers).
extern "C" int spe_cam_contour_boxes(const unsigned char* img, int rows, int cols, float area_ratio, int* boxes, int max_boxes,
                                     int* nboxes) {
    std::vector<int> f((size_t)(rows + 2) * (cols + 2), 0);
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j) f[(size_t)(i + 1) * (cols + 2) + (j + 1)] = img[(size_t)i * cols + j] ? 1 : 0;
    std::vector<Contour> cs;
    follow_borders(f, rows, cols, cs);
    int n = 0;
    if (cs.empty()) {
        if (max_boxes > 0) { boxes[0] = 0; boxes[1] = 0; boxes[2] = 1; boxes[3] = 1; n = 1; }
    } else {
        std::vector<int> idx(cs.size());
        for (size_t k = 0; k < cs.size(); ++k) idx[k] = (int)k;
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cs[a].area > cs[b].area; });
        const double top = cs[idx[0]].area;
        for (int k : idx) {
            if (!(cs[k].area >= top * (double)area_ratio)) continue;
            if (n >= max_boxes) return -5;
            boxes[4 * n] = cs[k].x0; boxes[4 * n + 1] = cs[k].y0; boxes[4 * n + 2] = cs[k].x1 + 1; boxes[4 * n + 3] = cs[k].y1 + 1;
            ++n;
        }
    }
    *nboxes = n;
    return 0;
}
