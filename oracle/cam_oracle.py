"""TEST INFRASTRUCTURE ONLY.  NumPy restatement of the reference's CAM -> box step (cams_deit.py:9-13 `resize_cam`,
cams_deit.py:61-96 `get_multi_bboxes`) with the OpenCV calls replaced by restatements of their published algorithms
(OpenCV is a third-party dependency absent from this environment and from /root/reference: **parity unpinned** - no
golden vector from cv2 exists for this component).

  cv2.resize(INTER_LINEAR): source coordinate (d + 0.5) * scale - 0.5, replicated borders, horizontal then vertical.
  cv2.threshold(THRESH_TOZERO): keep src where src > thresh.
  cv2.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE): Suzuki & Abe 1985 border following, 8-connected, every outer and
      hole border (the approximation mode changes neither area nor bounding box).
  cv2.contourArea: |shoelace| / 2 over the border pixel sequence.  cv2.boundingRect: inclusive extent.
"""
import numpy as np

_DX = [1, 1, 0, -1, -1, -1, 0, 1]          # clockwise from east, y grows downwards
_DY = [0, 1, 1, 1, 0, -1, -1, -1]


def resize_bilinear(cam, rows, cols):
    h, w = cam.shape
    sy, sx = np.float32(h) / np.float32(rows), np.float32(w) / np.float32(cols)
    fy = (np.arange(rows, dtype=np.float32) + np.float32(0.5)) * sy - np.float32(0.5)
    fx = (np.arange(cols, dtype=np.float32) + np.float32(0.5)) * sx - np.float32(0.5)
    y0 = np.floor(fy).astype(np.int64); x0 = np.floor(fx).astype(np.int64)
    wy = (fy - y0).astype(np.float32); wx = (fx - x0).astype(np.float32)
    wy[y0 < 0] = 0; y0[y0 < 0] = 0
    wy[y0 >= h - 1] = 0; y0[y0 >= h - 1] = h - 1
    wx[x0 < 0] = 0; x0[x0 < 0] = 0
    wx[x0 >= w - 1] = 0; x0[x0 >= w - 1] = w - 1
    y1 = np.minimum(y0 + 1, h - 1); x1 = np.minimum(x0 + 1, w - 1)
    cam = cam.astype(np.float32)
    top = cam[y0][:, x0] * (1 - wx)[None, :] + cam[y0][:, x1] * wx[None, :]
    bot = cam[y1][:, x0] * (1 - wx)[None, :] + cam[y1][:, x1] * wx[None, :]
    return (top * (1 - wy)[:, None] + bot * wy[:, None]).astype(np.float32)


def threshold_image(cam, rows, cols, cam_thr):
    """resize_cam + the first half of get_multi_bboxes -> thresholded uint8 image."""
    r = resize_bilinear(cam, rows, cols)
    r = r - r.min()
    mx = r.max()
    if not mx > 0:
        return np.zeros((rows, cols), np.uint8)
    r = r / mx
    q = (r * np.float32(255.0)).astype(np.uint8)
    thr = int(cam_thr * float(q.max()))
    return np.where(q > thr, q, 0).astype(np.uint8)


def find_borders(img):
    """-> list of (area, x0, y0, x1, y1) in discovery (raster) order."""
    R, C = img.shape
    f = np.zeros((R + 2, C + 2), np.int64)
    f[1:-1, 1:-1] = (img != 0)
    nbd = 1
    res = []
    for i in range(1, R + 1):
        for j in range(1, C + 1):
            v = f[i, j]
            if v == 0:
                continue
            if v == 1 and f[i, j - 1] == 0:
                nbd += 1; start = 4
            elif v >= 1 and f[i, j + 1] == 0:
                nbd += 1; start = 0
            else:
                continue
            d1 = None
            for k in range(8):
                d = (start + k) % 8
                if f[i + _DY[d], j + _DX[d]] != 0:
                    d1 = d
                    break
            if d1 is None:
                f[i, j] = -nbd
                res.append((0.0, j - 1, i - 1, j - 1, i - 1))
                continue
            i1, j1 = i + _DY[d1], j + _DX[d1]
            i2, j2, i3, j3 = i1, j1, i, j
            acc = 0.0
            xs, ys = [], []
            while True:
                ds = [d for d in range(8) if (i3 + _DY[d], j3 + _DX[d]) == (i2, j2)][0]
                east_zero = False
                for k in range(1, 9):
                    d = (ds - k) % 8
                    if f[i3 + _DY[d], j3 + _DX[d]] != 0:
                        d4 = d
                        break
                    if d == 0:
                        east_zero = True
                i4, j4 = i3 + _DY[d4], j3 + _DX[d4]
                if east_zero:
                    f[i3, j3] = -nbd
                elif f[i3, j3] == 1:
                    f[i3, j3] = nbd
                acc += float(j3) * i4 - float(j4) * i3
                xs.append(j3); ys.append(i3)
                if (i4, j4) == (i, j) and (i3, j3) == (i1, j1):
                    break
                i2, j2, i3, j3 = i3, j3, i4, j4
            res.append((abs(acc) * 0.5, min(xs) - 1, min(ys) - 1, max(xs) - 1, max(ys) - 1))
    return res


def multi_bboxes_from_image(img, area_ratio):
    """second half of get_multi_bboxes (cams_deit.py:78-96)."""
    cs = find_borders(img)
    if not cs:
        return [[0, 0, 1, 1]]
    areas = [c[0] for c in cs]
    order = sorted(range(len(areas)), key=areas.__getitem__, reverse=True)
    out = []
    for idx in order:
        if areas[idx] >= areas[order[0]] * area_ratio:
            _, x0, y0, x1, y1 = cs[idx]
            out.append([x0, y0, x1 + 1, y1 + 1])
    return out
