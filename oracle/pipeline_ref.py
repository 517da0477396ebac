"""ORACLE (test infrastructure, not product code) — CPU restatement of everything around the two networks:
det pre-processing, DB post-processing, box ordering, perspective crop, rec batching/pre-processing, CTC greedy
decode, and the reference's own glue (OcrRecogniser.predict / get_coordinates / subtitle-area filter).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Pinning status
  * `ocr_predict_glue`, `get_coordinates`, `y_round` follow backend/tools/ocr.py:16-22,24-86,115-134 and are PINNED
    by tests/golden/ocr_glue.json, produced by executing that reference file in the build container
    (tests/golden/make_ocr_glue_golden.py).
  * `subtitle_area_keep` follows backend/tools/subtitle_ocr.py:42-67 (shapely rectangles restated as integer
    rectangle arithmetic) — unpinned (shapely absent).
  * everything else restates third-party `paddleocr~=2.10.0` (requirements.txt:16; tools/infer/predict_{det,rec,system}.py,
    ppocr/data/imaug/operators.py, ppocr/postprocess/{db_postprocess,rec_postprocess}.py) and the OpenCV 4.11
    primitives it calls, as recalled in SURVEY.md Appendix C — PARITY UNPINNED (neither package is installed,
    the reference holds no tests or golden outputs for this path).  Call sites: backend/tools/ocr.py:27,
    backend/tools/subtitle_detect.py:25.
Plain numpy/scipy; loops only over boxes/components.

Consistency checks (NOT pins: neither paddle nor cv2 can run here): tests/test_oracle_crosschecks.py compares the restated
primitives of this file with independent implementations of the same published definitions (torch.nn.LSTM, a float64
bilinear resize, brute-force rotation search, scipy's convex hull).
"""
import math

import numpy as np
from scipy import ndimage

DET_MEAN = np.array([0.485, 0.456, 0.406], np.float32)
DET_STD = np.array([0.229, 0.224, 0.225], np.float32)


# ------------------------------------------------------------------------------------------------ det pre-process
def det_resize_shape(h, w, limit_side_len=960, limit_type="max"):
    """paddleocr DetResizeForTest.resize_image_type0 (App. C.1)."""
    if limit_type == "max":
        ratio = float(limit_side_len) / max(h, w) if max(h, w) > limit_side_len else 1.0
    else:
        ratio = float(limit_side_len) / min(h, w) if min(h, w) < limit_side_len else 1.0
    rh, rw = int(h * ratio), int(w * ratio)
    rh = max(int(round(rh / 32) * 32), 32)
    rw = max(int(round(rw / 32) * 32), 32)
    return rh, rw


def _lin_coefs(dst, src):
    """OpenCV INTER_LINEAR coefficient table for 8-bit images: (s0 index, a0, a1) with 11-bit fixed point."""
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo] = 0
    s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0
    s[hi] = src - 1
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, a0, a1


def cv2_resize_linear_u8(img, dst_w, dst_h):
    """cv2.resize(img, (dst_w, dst_h)) for uint8 HxWxC, INTER_LINEAR, restated with OpenCV's integer arithmetic."""
    h, w = img.shape[:2]
    if (h, w) == (dst_h, dst_w):
        return img.copy()
    sx, ax0, ax1 = _lin_coefs(dst_w, w)
    sy, ay0, ay1 = _lin_coefs(dst_h, h)
    x1 = np.minimum(sx + 1, w - 1)
    y1 = np.minimum(sy + 1, h - 1)
    I = img.astype(np.int64)
    rows0 = I[sy][:, sx] * ax0[None, :, None] + I[sy][:, x1] * ax1[None, :, None]
    rows1 = I[y1][:, sx] * ax0[None, :, None] + I[y1][:, x1] * ax1[None, :, None]
    out = (((ay0[:, None, None] * (rows0 >> 4)) >> 16) + ((ay1[:, None, None] * (rows1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def det_preprocess(img, limit_side_len=960):
    """u8 BGR HxWx3 -> (float32 NCHW [1,3,rh,rw], (ratio_h, ratio_w))."""
    h, w = img.shape[:2]
    rh, rw = det_resize_shape(h, w, limit_side_len)
    r = cv2_resize_linear_u8(img, rw, rh)
    x = (r.astype(np.float32) * np.float32(1.0 / 255.0) - DET_MEAN) / DET_STD
    return np.ascontiguousarray(x.transpose(2, 0, 1)[None]), (rh / float(h), rw / float(w))


# ------------------------------------------------------------------------------------------------ DB post-process
def _convex_hull(pts):
    """Andrew monotone chain on integer points (same as csrc/db_geometry.h)."""
    pts = sorted(set(map(tuple, pts)))
    if len(pts) < 3:
        return [tuple(p) for p in pts]

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    h = []
    for p in pts:
        while len(h) >= 2 and cross(h[-2], h[-1], p) <= 0:
            h.pop()
        h.append(p)
    t = len(h) + 1
    for p in reversed(pts[:-1]):
        while len(h) >= t and cross(h[-2], h[-1], p) <= 0:
            h.pop()
        h.append(p)
    return h[:-1]


def _min_area_rect(hull):
    """-> (corners[4][2] float64, w, h); first strictly smallest area over hull edges."""
    n = len(hull)
    if n == 0:
        return np.zeros((4, 2)), 0.0, 0.0
    if n == 1:
        return np.array([hull[0]] * 4, dtype=np.float64), 0.0, 0.0
    H = np.asarray(hull, dtype=np.float64)
    best = None
    edges = 1 if n == 2 else n
    for i in range(edges):
        a = H[i]
        b = H[(i + 1) % n]
        ex, ey = b[0] - a[0], b[1] - a[1]
        len2 = ex * ex + ey * ey
        px, py = H[:, 0] - a[0], H[:, 1] - a[1]
        u = px * ex + py * ey
        v = py * ex - px * ey
        umin, umax, vmin, vmax = u.min(), u.max(), v.min(), v.max()
        area = (umax - umin) * (vmax - vmin) / len2
        if best is None or area < best[0]:
            ux, uy = ex / len2, ey / len2

            def at(uu, vv):
                return [a[0] + uu * ux - vv * uy, a[1] + uu * uy + vv * ux]
            ln = math.sqrt(len2)
            best = (area, np.array([at(umin, vmin), at(umax, vmin), at(umax, vmax), at(umin, vmax)]),
                    (umax - umin) / ln, (vmax - vmin) / ln)
    return best[1], best[2], best[3]


def _mini_box(corners, w, h):
    """paddleocr get_mini_boxes ordering (stable sort by x, then y inside the left / right pair)."""
    order = sorted(range(4), key=lambda i: corners[i][0])
    p = [corners[i] for i in order]
    i1, i4 = (0, 1) if p[1][1] > p[0][1] else (1, 0)
    i2, i3 = (2, 3) if p[3][1] > p[2][1] else (3, 2)
    return np.array([p[i1], p[i2], p[i3], p[i4]], dtype=np.float64), min(w, h)


def _box_score_fast(prob, box):
    """Mean of prob over lattice points inside-or-on the int-truncated quad, in its clipped bounding rectangle."""
    h, w = prob.shape
    b32 = box.astype(np.float32)
    xmin = int(np.clip(np.floor(b32[:, 0].min()), 0, w - 1))
    xmax = int(np.clip(np.ceil(b32[:, 0].max()), 0, w - 1))
    ymin = int(np.clip(np.floor(b32[:, 1].min()), 0, h - 1))
    ymax = int(np.clip(np.ceil(b32[:, 1].max()), 0, h - 1))
    q = np.stack([(b32[:, 0] - np.float32(xmin)), (b32[:, 1] - np.float32(ymin))], 1).astype(np.int32).astype(np.int64)
    ys, xs = np.mgrid[0:ymax - ymin + 1, 0:xmax - xmin + 1]
    pos = np.ones(xs.shape, bool)
    neg = np.ones(xs.shape, bool)
    for e in range(4):
        ax, ay = q[e]
        bx, by = q[(e + 1) % 4]
        cr = (bx - ax) * (ys - ay) - (by - ay) * (xs - ax)
        pos &= cr >= 0
        neg &= cr <= 0
    m = pos | neg
    if not m.any():
        return 0.0
    return float(np.float32(prob[ymin:ymax + 1, xmin:xmax + 1][m].astype(np.float64).sum() / m.sum()))


def _unclip_rect(box, ratio):
    """Closed-form rectangle offset (App. C.3): see csrc/db_geometry.h unclip_rect."""
    x, y = box[:, 0], box[:, 1]
    area = abs(sum(x[i] * y[(i + 1) % 4] - x[(i + 1) % 4] * y[i] for i in range(4))) * 0.5
    length = sum(math.sqrt((x[(i + 1) % 4] - x[i]) ** 2 + (y[(i + 1) % 4] - y[i]) ** 2) for i in range(4))
    dist = area * ratio / length
    q = np.trunc(box)
    u = q[1] - q[0]
    v = q[3] - q[0]
    ul, vl = math.sqrt(u[0] ** 2 + u[1] ** 2), math.sqrt(v[0] ** 2 + v[1] ** 2)
    u = u / ul if ul > 0 else np.array([1.0, 0.0])
    v = v / vl if vl > 0 else np.array([-u[1], u[0]])
    su, sv = [-1, 1, 1, -1], [-1, -1, 1, 1]
    out = []
    for i in range(4):
        px = q[i][0] + dist * (su[i] * u[0] + sv[i] * v[0])
        py = q[i][1] + dist * (su[i] * u[1] + sv[i] * v[1])
        out.append((int(np.rint(px)), int(np.rint(py))))
    return out


def order_points_clockwise(pts):
    """paddleocr 2.10 predict_det.order_points_clockwise (sum / diff rule)."""
    pts = np.asarray(pts, dtype=np.float64)
    s = pts.sum(axis=1)
    imin, imax = int(np.argmin(s)), int(np.argmax(s))
    rest = np.delete(pts, (imin, imax), axis=0)
    if len(rest) > 2:
        rest = rest[:2]
    d = rest[:, 1] - rest[:, 0]
    rect = np.zeros((4, 2))
    rect[0], rect[2] = pts[imin], pts[imax]
    rect[1] = rest[int(np.argmin(d))]
    rect[3] = rest[int(np.argmax(d))]
    return rect


def _hole_contours(mask):
    """Hole borders of cv2.findContours(RETR_LIST): 4-connected background regions that do not reach the image frame; the
    border of one = the foreground pixels 4-adjacent to it (Suzuki-Abe border points of the 8-connected case).
    -> [(raster index of the hole's first pixel, xs, ys)]"""
    h, w = mask.shape
    lab, k = ndimage.label(~mask)                       # default structure: 4-connectivity
    if k == 0:
        return []
    outside = np.unique(np.concatenate([lab[0], lab[-1], lab[:, 0], lab[:, -1]]))
    cross = ndimage.generate_binary_structure(2, 1)
    out = []
    for li, sl in enumerate(ndimage.find_objects(lab), 1):
        if li in outside or sl is None:
            continue
        y0, y1, x0, x1 = sl[0].start - 1, sl[0].stop + 1, sl[1].start - 1, sl[1].stop + 1      # a hole never touches the frame
        region = lab[y0:y1, x0:x1] == li
        ring = ndimage.binary_dilation(region, cross) & mask[y0:y1, x0:x1]
        ys, xs = np.nonzero(ring)
        fy, fx = np.nonzero(region)
        out.append(((int(fy[0]) + y0) * w + int(fx[0]) + x0, xs + x0, ys + y0))
    return out


def db_postprocess(prob, src_h, src_w, thresh=0.3, box_thresh=0.6, unclip_ratio=1.5, max_candidates=1000, min_size=3):
    """prob: float32 [h,w] -> (boxes float32 [k,4,2] in source pixels, scores float32 [k]).
    Restates DBPostProcess.boxes_from_bitmap + TextDetector.filter_tag_det_res (App. C.2): outer borders of the 8-connected
    components AND hole borders (cv2.findContours with RETR_LIST returns both), visited in reverse raster order of the pixel
    at which the scan finds them (a component's first pixel / a hole's first pixel), convex hull of the border points in
    place of the traced contour (the minimum-area rectangle only sees the hull)."""
    h, w = prob.shape
    mask = prob > np.float32(thresh)
    lab, k = ndimage.label(mask, structure=np.ones((3, 3), int))
    boxes, scores = [], []
    if k == 0:
        return np.zeros((0, 4, 2), np.float32), np.zeros((0,), np.float32)
    left = mask & ~np.pad(mask, ((0, 0), (1, 0)))[:, :-1]
    right = mask & ~np.pad(mask, ((0, 0), (0, 1)))[:, 1:]
    ys, xs = np.nonzero(left | right)
    labs = lab[ys, xs]
    order = np.argsort(labs, kind="stable")
    ys, xs, labs = ys[order], xs[order], labs[order]
    starts = np.searchsorted(labs, np.arange(1, k + 2))
    contours = []                                        # (raster index where the scan meets the border, xs, ys)
    for ci in range(1, k + 1):
        sl = slice(starts[ci - 1], starts[ci])
        contours.append((int((ys[sl] * w + xs[sl]).min()), xs[sl], ys[sl]))
    contours += _hole_contours(mask)
    contours.sort(key=lambda c: -c[0])
    for _key, cx, cy in contours[:max_candidates]:
        hull = _convex_hull(zip(cx.tolist(), cy.tolist()))
        corners, rw_, rh_ = _min_area_rect(hull)
        box, sside = _mini_box(corners, rw_, rh_)
        if sside < min_size:
            continue
        score = _box_score_fast(prob, box)
        if box_thresh > score:
            continue
        grown = _unclip_rect(box, unclip_ratio)
        corners2, w2, h2 = _min_area_rect(_convex_hull(grown))
        box2, sside2 = _mini_box(corners2, w2, h2)
        if sside2 < min_size + 2:
            continue
        b32 = box2.astype(np.float32)
        fx = np.clip(np.round(b32[:, 0] / np.float32(w) * np.float32(src_w)), 0, src_w)
        fy = np.clip(np.round(b32[:, 1] / np.float32(h) * np.float32(src_h)), 0, src_h)
        q = np.stack([fx, fy], 1).astype(np.int32).astype(np.float64)
        o = order_points_clockwise(q)
        o[:, 0] = np.clip(o[:, 0], 0, src_w - 1).astype(np.int64)
        o[:, 1] = np.clip(o[:, 1], 0, src_h - 1).astype(np.int64)
        rw = int(np.linalg.norm(o[0] - o[1]))
        rh = int(np.linalg.norm(o[0] - o[3]))
        if rw <= 3 or rh <= 3:
            continue
        boxes.append(o.astype(np.float32))
        scores.append(score)
    if not boxes:
        return np.zeros((0, 4, 2), np.float32), np.zeros((0,), np.float32)
    return np.stack(boxes), np.asarray(scores, np.float32)


# ------------------------------------------------------------------------------------------------ box order + crop
def sorted_boxes(dt_boxes):
    """paddleocr predict_system.sorted_boxes (App. C.4)."""
    n = len(dt_boxes)
    bs = sorted(list(dt_boxes), key=lambda b: (b[0][1], b[0][0]))
    for i in range(n - 1):
        for j in range(i, -1, -1):
            if abs(bs[j + 1][0][1] - bs[j][0][1]) < 10 and bs[j + 1][0][0] < bs[j][0][0]:
                bs[j], bs[j + 1] = bs[j + 1], bs[j]
            else:
                break
    return bs


def crop_geometry(pts):
    """-> (crop_w, crop_h, rotate) of get_rotate_crop_image."""
    pts = np.asarray(pts, dtype=np.float32)
    cw = int(max(np.linalg.norm(pts[0] - pts[1]), np.linalg.norm(pts[2] - pts[3])))
    ch = int(max(np.linalg.norm(pts[0] - pts[3]), np.linalg.norm(pts[1] - pts[2])))
    rotate = 1 if (cw > 0 and ch * 1.0 / cw >= 1.5) else 0
    return cw, ch, rotate


def _perspective_inverse(src, cw, ch):
    """cv2.getPerspectiveTransform + the inversion warpPerspective applies (no WARP_INVERSE_MAP), in cv2's own arithmetic as
    recalled: the 8x8 system solved by OpenCV's LU (partial pivoting on |a|, row updates a[j][k] += (a[j][i] * (-1 / a[i][i])) *
    a[i][k], back substitution s / a[i][i]), then the closed-form 3x3 inverse (cofactors times 1 / det3).  Plain float64
    scalar operations in that order: the engine's host code (prepost.hip perspective_inverse) performs the same sequence, so
    the two agree to the last bit — which matters because integer-cornered quads put many 1/32-pixel coordinates exactly on a
    rounding tie."""
    dst = [(0.0, 0.0), (float(cw), 0.0), (float(cw), float(ch)), (0.0, float(ch))]
    A = [[0.0] * 8 for _ in range(8)]
    b = [0.0] * 8
    for i in range(4):
        x, y = float(src[i][0]), float(src[i][1])
        X, Y = dst[i]
        A[i] = [x, y, 1.0, 0.0, 0.0, 0.0, -x * X, -y * X]
        A[i + 4] = [0.0, 0.0, 0.0, x, y, 1.0, -x * Y, -y * Y]
        b[i], b[i + 4] = X, Y
    eps = 2.220446049250313e-16 * 100
    for i in range(8):
        k = i
        for j in range(i + 1, 8):
            if abs(A[j][i]) > abs(A[k][i]):
                k = j
        if abs(A[k][i]) < eps:
            raise np.linalg.LinAlgError("singular")
        if k != i:
            A[i], A[k] = A[k], A[i]
            b[i], b[k] = b[k], b[i]
        d = -1.0 / A[i][i]
        for j in range(i + 1, 8):
            alpha = A[j][i] * d
            for c in range(i + 1, 8):
                A[j][c] += alpha * A[i][c]
            b[j] += alpha * b[i]
    for i in range(7, -1, -1):
        sacc = b[i]
        for c in range(i + 1, 8):
            sacc -= A[i][c] * b[c]
        b[i] = sacc / A[i][i]
    m = b + [1.0]
    det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6])
    if det == 0.0:
        raise np.linalg.LinAlgError("singular")
    d = 1.0 / det
    inv = [(m[4] * m[8] - m[5] * m[7]) * d, (m[2] * m[7] - m[1] * m[8]) * d, (m[1] * m[5] - m[2] * m[4]) * d,
           (m[5] * m[6] - m[3] * m[8]) * d, (m[0] * m[8] - m[2] * m[6]) * d, (m[2] * m[3] - m[0] * m[5]) * d,
           (m[3] * m[7] - m[4] * m[6]) * d, (m[1] * m[6] - m[0] * m[7]) * d, (m[0] * m[4] - m[1] * m[3]) * d]
    return np.array(inv, np.float64).reshape(3, 3)


def _cubic_w(x):
    A = np.float32(-0.75)
    x = x.astype(np.float32)
    c0 = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
    c1 = ((A + 2) * x - (A + 3)) * x * x + 1
    c2 = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
    c3 = np.float32(1.0) - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], -1).astype(np.float32)


def _bicubic_itab(wy, wx):
    """cv2's fixed-point bicubic table entry (imgwarp.cpp initInterTab2D, as recalled): the 4 x 4 products of the float32 1-D
    coefficients scaled by INTER_REMAP_COEF_SCALE = 32768 and rounded to short (cvRound: half to even); when they do not sum
    to 32768 the difference goes to the largest (sum too small) or smallest (too large) of the entries [2..3] x [2..3] — the
    loop bounds `ksize/2 .. ksize/2 + 2` of the original, scanned with its strict-less / else-strict-greater updates.
    wy, wx: float32 [..., 4] -> int32 [..., 4, 4]."""
    v = (wy[..., :, None] * wx[..., None, :]).astype(np.float32) * np.float32(32768.0)
    it = np.clip(np.rint(v), -32768, 32767).astype(np.int32)
    diff = it.reshape(it.shape[:-2] + (16,)).sum(-1) - 32768
    flat = it.reshape(it.shape[:-2] + (16,))
    mk = np.full(diff.shape, 10, np.int64)                # flat index of (2, 2)
    Mk = mk.copy()
    for pos in (10, 11, 14, 15):
        val = flat[..., pos]
        cur_m = np.take_along_axis(flat, mk[..., None], -1)[..., 0]
        cur_M = np.take_along_axis(flat, Mk[..., None], -1)[..., 0]
        less = val < cur_m
        more = (~less) & (val > cur_M)
        mk = np.where(less, pos, mk)
        Mk = np.where(more, pos, Mk)
    tgt = np.where(diff < 0, Mk, mk)
    fix = np.where(diff != 0, diff, 0)
    cur = np.take_along_axis(flat, tgt[..., None], -1)[..., 0]
    new = ((cur - fix + 32768) % 65536) - 32768            # (short) cast of the corrected entry
    np.put_along_axis(flat, tgt[..., None], new[..., None], -1)
    return flat.reshape(it.shape)


def get_rotate_crop_image(img, pts):
    """paddleocr get_rotate_crop_image: warpPerspective(INTER_CUBIC, BORDER_REPLICATE) with coordinates quantised
    to 1/32 px and cv2's FIXED-POINT bicubic (A = -0.75; 15-bit weight table, (sum + 2^14) >> 15), then np.rot90 when tall."""
    cw, ch, rotate = crop_geometry(pts)
    sh, sw = img.shape[:2]
    try:
        minv = _perspective_inverse(np.asarray(pts, np.float32), cw, ch)
    except np.linalg.LinAlgError:
        minv = np.array([[1, 0, pts[0][0]], [0, 1, pts[0][1]], [0, 0, 1]], dtype=np.float64)
    # WarpPerspectiveInvoker walks the destination in blocks (BLOCK_SZ = 32: bh0 = min(16, h), bw0 = min(1024 / bh0, w)) and
    # evaluates the homography from the block's left edge bx: X0 = M0 bx + M1 y + M2, then (X0 + M0 x1) * (32 / (W0 + M6 x1))
    ys, xs = np.mgrid[0:ch, 0:cw]
    bw0 = min(1024 // min(16, ch), cw)
    bx = ((xs // bw0) * bw0).astype(np.float64)
    x1 = (xs % bw0).astype(np.float64)
    ys = ys.astype(np.float64)
    X0 = minv[0, 0] * bx + minv[0, 1] * ys + minv[0, 2]
    Y0 = minv[1, 0] * bx + minv[1, 1] * ys + minv[1, 2]
    W0 = minv[2, 0] * bx + minv[2, 1] * ys + minv[2, 2]
    W = W0 + minv[2, 0] * x1
    W = np.where(W != 0, 32.0 / np.where(W != 0, W, 1), 0.0)
    X = np.rint(np.clip((X0 + minv[0, 0] * x1) * W, -2147483648.0, 2147483647.0)).astype(np.int64)
    Y = np.rint(np.clip((Y0 + minv[1, 0] * x1) * W, -2147483648.0, 2147483647.0)).astype(np.int64)
    sx, sy = (X >> 5) - 1, (Y >> 5) - 1
    wx = _cubic_w((X & 31).astype(np.float32) * np.float32(1 / 32))
    wy = _cubic_w((Y & 31).astype(np.float32) * np.float32(1 / 32))
    itab = _bicubic_itab(wy, wx)
    acc = np.zeros((ch, cw, 3), np.int64)
    I = img.astype(np.int64)
    for r in range(4):
        yy = np.clip(sy + r, 0, sh - 1)
        for q in range(4):
            xx = np.clip(sx + q, 0, sw - 1)
            acc += itab[..., r, q][..., None] * I[yy, xx]
    out = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    if rotate:
        out = np.rot90(out)
    return np.ascontiguousarray(out)


# ------------------------------------------------------------------------------------------------ rec pre-process
def rec_resized_width(w, h, img_w, img_h=48):
    return min(img_w, int(math.ceil(img_h * (w / float(h)))))


def resize_norm_img(crop, img_w, img_h=48):
    """paddleocr TextRecognizer.resize_norm_img for CTC models -> float32 [3,img_h,img_w]."""
    h, w = crop.shape[:2]
    rw = rec_resized_width(w, h, img_w, img_h)
    r = cv2_resize_linear_u8(crop, rw, img_h).astype(np.float32)
    r = (r.transpose(2, 0, 1) / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)
    out = np.zeros((3, img_h, img_w), np.float32)
    out[:, :, :rw] = r
    return out


def rec_batches(crops, rec_batch_num=6, img_h=48, base_w=320):
    """-> list of (indices, img_w) mini-batches in paddleocr order (sorted by w/h, chunks of rec_batch_num)."""
    ratios = [c.shape[1] / float(c.shape[0]) for c in crops]
    order = np.argsort(np.array(ratios), kind="stable") if crops else []
    out = []
    for b in range(0, len(crops), rec_batch_num):
        idx = [int(i) for i in order[b:b + rec_batch_num]]
        max_wh = base_w / float(img_h)
        for i in idx:
            max_wh = max(max_wh, ratios[i])
        out.append((idx, int(img_h * max_wh)))
    return out


def ctc_greedy(probs):
    """probs [T,C] -> (kept class ids list, mean kept max-prob) — CTCLabelDecode with is_remove_duplicate."""
    idx = probs.argmax(1)
    mp = probs.max(1)
    keep = np.ones(len(idx), bool)
    keep[1:] = idx[1:] != idx[:-1]
    keep &= idx != 0
    ids = idx[keep].tolist()
    conf = float(np.mean(mp[keep])) if keep.any() else 0.0
    return ids, conf


def en_charset():
    """Character table for lang='en' (en_dict.txt as recalled in App. C.6): blank + 0x30..0x7E + 0x21..0x2F + ' ' + ' '."""
    chars = [chr(c) for c in range(0x30, 0x7F)] + [chr(c) for c in range(0x21, 0x30)] + [" "]
    return ["blank"] + chars + [" "]


def standin_charset(ncls):
    """Stand-in table when the dictionary file is unavailable (ppocr_keys_v1.txt etc.): class i -> one code point."""
    return ["blank"] + [chr(0x4E00 + i) for i in range(ncls - 2)] + [" "]


def decode_text(ids, charset):
    return "".join(charset[i] for i in ids)


# ------------------------------------------------------------------------------------------------ reference glue
def y_round(y):
    """backend/tools/ocr.py:16-22."""
    y_min = y + 10 - y % 10
    y_max = y - y % 10
    return y_min if abs(y - y_min) < abs(y - y_max) else y_max


def _quad_to_aabb(q):
    x1, y1 = int(q[0][0]), int(q[0][1])
    x2, y2 = int(q[1][0]), int(q[1][1])
    x3, y3 = int(q[2][0]), int(q[2][1])
    x4, y4 = int(q[3][0]), int(q[3][1])
    return [max(x1, x4), min(x2, x3), max(y1, y2), min(y3, y4)]


def get_coordinates(dt_box):
    """backend/tools/ocr.py:115-134 (returns [] unless given a python list)."""
    if not isinstance(dt_box, list):
        return []
    return [tuple(_quad_to_aabb(list(q))) for q in dt_box]


def ocr_predict_glue(detection_box, recognise_result):
    """backend/tools/ocr.py:28-86: AABB conversion, line clustering on y_round(ymin), per-line x ordering."""
    if len(detection_box) == 0:
        return detection_box, recognise_result
    coords = [_quad_to_aabb(list(q)) for q in detection_box] if isinstance(detection_box, list) else []
    lines = []
    for c in coords:
        yr = y_round(c[2])
        if not lines:
            lines.append(yr)
        elif yr not in lines and yr + 10 not in lines and yr - 10 not in lines:
            lines.append(yr)
    lines = sorted(lines)
    for c in coords:
        for j in lines:
            if abs(j - y_round(c[2])) <= 10:
                c[2] = j
    pairs = list(zip(coords, recognise_result))
    ranked = []
    for line in lines:
        tmp = [p for p in pairs if p[0][2] == line]
        for l in range(1, len(tmp)):
            for j in range(0, len(tmp) - l):
                if tmp[j][0][0] > tmp[j + 1][0][0]:
                    tmp[j], tmp[j + 1] = tmp[j + 1], tmp[j]
        ranked += tmp
    dt_box = [[(c[0], c[2]), (c[1], c[2]), (c[1], c[3]), (c[0], c[3])] for c, _ in ranked]
    return dt_box, [r for _, r in ranked]


def subtitle_area_keep(coordinate, prob, area, deviation_rate=0.0, drop_score=0.75):
    """backend/tools/subtitle_ocr.py:42-67 with shapely rectangles restated: keep iff the box intersects the area with
    positive area... (shapely `is_empty` is False for touching rectangles too) and overflow <= rate and prob > drop.
    coordinate = (xmin,xmax,ymin,ymax); area = (ymin,ymax,xmin,xmax)."""
    # shapely polygons do not care about the corner order: an "inverted" box (xmin > xmax from a skewed quad) covers the
    # rectangle between its smaller and larger values
    xmin, xmax = min(coordinate[0], coordinate[1]), max(coordinate[0], coordinate[1])
    ymin, ymax = min(coordinate[2], coordinate[3]), max(coordinate[2], coordinate[3])
    aymin, aymax, axmin, axmax = min(area[0], area[1]), max(area[0], area[1]), min(area[2], area[3]), max(area[2], area[3])
    ix0, ix1 = max(xmin, axmin), min(xmax, axmax)
    iy0, iy1 = max(ymin, aymin), min(ymax, aymax)
    if ix0 > ix1 or iy0 > iy1:
        return False
    inter = max(0, ix1 - ix0) * max(0, iy1 - iy0)
    a_area = (axmax - axmin) * (aymax - aymin)
    b_area = (xmax - xmin) * (ymax - ymin)
    overflow = (a_area + b_area - inter) / a_area - 1
    return overflow <= deviation_rate and prob > drop_score


# ------------------------------------------------------------------------------------------------ whole system
def text_system(img, det_fn, rec_fn, charset, rec_batch_num=6, drop_score=0.0):
    """paddleocr TextSystem.__call__(img, cls=False) with pluggable networks:
       det_fn(float32 NCHW) -> prob [h,w];  rec_fn(float32 [B,3,48,W]) -> probs [B,T,C]."""
    x, _ = det_preprocess(img)
    prob = det_fn(x)
    boxes, _ = db_postprocess(prob, img.shape[0], img.shape[1])
    boxes = sorted_boxes(boxes)
    crops = [get_rotate_crop_image(img, b) for b in boxes]
    res = [("", 0.0)] * len(crops)
    for idx, img_w in rec_batches(crops, rec_batch_num):
        batch = np.stack([resize_norm_img(crops[i], img_w) for i in idx])
        probs = rec_fn(batch)
        for k, i in enumerate(idx):
            ids, conf = ctc_greedy(probs[k])
            res[i] = (decode_text(ids, charset), conf)
    fb, fr = [], []
    for b, r in zip(boxes, res):
        if r[1] >= drop_score:
            fb.append(b)
            fr.append(r)
    return fb, fr
