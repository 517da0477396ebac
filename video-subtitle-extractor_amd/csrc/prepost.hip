// Pre-/post-processing of the OCR hot path on the GPU (HBM-bound byte/integer work):
//   det pre-process   uint8 BGR -> fixed-point bilinear resize -> normalise -> fp16 NHWC(8)
//   DB post-process   threshold -> 8-connected components (atomic union-find) -> run end-points -> host
//                     geometry (db_geometry.h) -> device polygon means -> host unclip/scale/filter
//   rec pre-process   perspective bicubic crop from the original frame -> fixed-point bilinear resize to
//                     height 48 -> normalise -> zero right-pad -> fp16 NHWC(8)
//   CTC collapse      wavefront ballot scan over the arg-max sequence
// These restate paddleocr 2.10 / OpenCV 4.11 behaviour as recalled in SURVEY.md App. C (not verifiable here).
#include <cstring>
#include <vector>

#include <algorithm>
#include <mutex>
#include "common.h"
#include "db_geometry.h"
#include "resize_u8.h"

extern "C" const char* vse_last_error(void);
void vse_set_error(const char* msg);

// ================================================================================================ bilinear (cv2): resize_u8.h
__global__ __launch_bounds__(256) void det_preprocess_kernel(const uint8_t* __restrict__ src, int n, int sh, int sw,
                                                             long pitch, long fstride, half_t* __restrict__ dst, int dh,
                                                             int dw, float m0, float m1, float m2, float sd0, float sd1,
                                                             float sd2, int raw) {
    const long total = (long)n * dh * dw;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % dw);
        const long t = i / dw;
        const int y = (int)(t % dh);
        const long f = t / dh;
        const LinCoef cx = lin_coef(x, dw, sw), cy = lin_coef(y, dh, sh);
        const int x1 = min(cx.s0 + 1, sw - 1), y1 = min(cy.s0 + 1, sh - 1);
        const uint8_t* r0 = src + f * fstride + (long)cy.s0 * pitch;
        const uint8_t* r1 = src + f * fstride + (long)y1 * pitch;
        float v[3];
        const float mean[3] = {m0, m1, m2}, sd[3] = {sd0, sd1, sd2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int u;
            if (sw == dw && sh == dh) u = r0[cx.s0 * 3 + c];
            else u = cv_bilinear_u8(r0[cx.s0 * 3 + c], r0[x1 * 3 + c], r1[cx.s0 * 3 + c], r1[x1 * 3 + c], cx, cy);
            // paddleocr NormalizeImage: (img * (1/255) - mean) / std in float32; raw: the resized u8 value itself (exact in fp16)
            v[c] = raw ? (float)u : ((float)u * (1.f / 255.f) - mean[c]) / sd[c];
        }
        // raw: channel 3 is the constant 1 inside the image — the stem conv's weights for it carry -mean/std per tap, and the
        // conv's zero padding then stands for a NORMALISED zero exactly as in the reference (compiler input_norm)
        half8 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)(raw ? 1.f : 0.f), 0, 0, 0, 0};
        *reinterpret_cast<half8*>(dst + i * 8) = o;
    }
}

extern "C" int vse_det_preprocess(vse_ctx*, const void* d_bgr, int n, int src_h, int src_w, int64_t pitch,
                                  int64_t frame_stride, void* d_out, int dst_h, int dst_w, const float* mean3,
                                  const float* std3, void* stream) {
    if (!d_bgr || !d_out || n <= 0 || (!mean3) != (!std3)) return VSE_E_INVAL;
    const int raw = mean3 == nullptr;
    static const float zero3[3] = {0.f, 0.f, 0.f}, one3[3] = {1.f, 1.f, 1.f};
    if (raw) { mean3 = zero3; std3 = one3; }
    const long total = (long)n * dst_h * dst_w;
    int grid = (int)std::min<long>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(det_preprocess_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint8_t*>(d_bgr), n, src_h, src_w, (long)pitch, (long)frame_stride,
                       reinterpret_cast<half_t*>(d_out), dst_h, dst_w, mean3[0], mean3[1], mean3[2], std3[0],
                       std3[1], std3[2], raw);
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}

// ================================================================================================ DB post-process
// Labels are int32 pixel indices local to a frame (-1 = background).  Union-find with atomicMin: the root of a
// component is its smallest pixel index = first pixel in raster order.
__device__ __forceinline__ int uf_find(int* L, int a) {
    int r = a;
    while (true) {
        const int p = L[r];
        if (p == r) break;
        r = p;
    }
    return r;
}
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
    while (true) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }   // a > b: hang a under b
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

// The three label passes walk a [n, h, w] map of which ~1-2 % of the pixels are text: a thread owns FOUR consecutive pixels of one row
// (one 16-byte load), a block one row segment of 1024 pixels, blockIdx = (segment, row, frame) — no pixel index is ever divided (the first
// form, one pixel per thread with three 64-bit divisions by runtime extents each, spent 337 us on a 134 MB label map) — and a vector that
// holds no text pixel is done after that one load.
// Labels start as RUN STARTS: db_init_kernel gives every text pixel the index of the first pixel of its horizontal run inside the block's
// segment (a block-wide max-scan of "last background column"), so a run is one tree of depth 1 from the start and db_merge_kernel has
// only the vertical / diagonal contacts (and the seam between two segments of a row) left to union.  With pixel-index labels every pixel
// was hooked to its left neighbour concurrently: chains as long as the run, walked by every later find (258 us of the pass).
__global__ __launch_bounds__(256) void db_init_kernel(const float* __restrict__ prob, int* __restrict__ Lall, int h, int w, float thresh) {
    __shared__ int wave_last[4];
    const int y = blockIdx.y, xg = blockIdx.x * 1024, x0 = xg + threadIdx.x * 4;
    const long row = ((long)blockIdx.z * h + y) * w;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v[4] = {0.f, 0.f, 0.f, 0.f};                       // columns behind the row count as background
    const bool vec = x0 + 3 < w && ((reinterpret_cast<uintptr_t>(prob + row + x0) | reinterpret_cast<uintptr_t>(Lall + row + x0)) & 15) == 0;
    if (vec) {
        const float4v q = *reinterpret_cast<const float4v*>(prob + row + x0);
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
    } else {
        for (int k = 0; k < 4; ++k) if (x0 + k < w) v[k] = prob[row + x0 + k];
    }
    bool fg[4];
    int last = -1;                                           // last background column of this vector (-1: none)
#pragma unroll
    for (int k = 0; k < 4; ++k) { fg[k] = v[k] > thresh; if (!fg[k]) last = x0 + k; }
    // inclusive max-scan over the block's threads (columns ascend with the thread index)
    int scan = last;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(scan, d, 64);
        if (lane >= d) scan = max(scan, o);
    }
    if (lane == 63) wave_last[wave] = scan;
    __syncthreads();
    int before = __shfl_up(scan, 1, 64);                     // last background column left of this vector ...
    if (lane == 0) before = -1;
    for (int q = 0; q < wave; ++q) before = max(before, wave_last[q]);
    if (x0 >= w) return;
    const int p0 = y * w;
    int lab[4];
    int start = max(before + 1, xg);                         // ... so a run that reaches this vector starts here (at the segment's first column at the earliest)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!fg[k]) { lab[k] = -1; start = x0 + k + 1; }
        else lab[k] = p0 + start;
    }
    if (vec) {
        int4 l; l.x = lab[0]; l.y = lab[1]; l.z = lab[2]; l.w = lab[3];
        *reinterpret_cast<int4*>(Lall + row + x0) = l;
    } else {
        for (int k = 0; k < 4 && x0 + k < w; ++k) Lall[row + x0 + k] = lab[k];
    }
}
__global__ __launch_bounds__(256) void db_merge_kernel(int* __restrict__ Lall, int h, int w) {
    const int y = blockIdx.y, x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= w) return;
    int* L = Lall + (long)blockIdx.z * h * w;
    const int p0 = y * w + x0;
    int lab[4] = {-1, -1, -1, -1};
    if (x0 + 3 < w && (reinterpret_cast<uintptr_t>(L + p0) & 15) == 0) {
        const int4 l = *reinterpret_cast<const int4*>(L + p0);
        lab[0] = l.x; lab[1] = l.y; lab[2] = l.z; lab[3] = l.w;
    } else {
        for (int k = 0; k < 4 && x0 + k < w; ++k) lab[k] = L[p0 + k];
    }
    if (lab[0] < 0 && lab[1] < 0 && lab[2] < 0 && lab[3] < 0) return;      // no text pixel here
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (lab[k] < 0) continue;
        const int x = x0 + k, p = p0 + k;
        // (a run is already one tree: only the seam between two 1024-pixel segments of a row needs the horizontal union)
        if ((x & 1023) == 0 && x > 0 && L[p - 1] >= 0) uf_union(L, p, p - 1);
        if (y > 0) {
            if (L[p - w] >= 0) uf_union(L, p, p - w);
            if (x > 0 && L[p - w - 1] >= 0) uf_union(L, p, p - w - 1);
            if (x < w - 1 && L[p - w + 1] >= 0) uf_union(L, p, p - w + 1);
        }
    }
}
// Emits one record per horizontal run END-POINT: {root | end flags, x | y << 16}.  cnt[f] counts records of frame f.
#define DB_LEFT (1 << 28)
#define DB_RIGHT (1 << 29)
#define DB_ROOT_MASK ((1 << 28) - 1)
__global__ __launch_bounds__(256) void db_runs_kernel(int* __restrict__ Lall, int h, int w, int2* __restrict__ recs,
                                                      int* __restrict__ cnt, int cap) {
    const int y = blockIdx.y, x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= w) return;
    const int f = blockIdx.z;
    int* L = Lall + (long)f * h * w;
    const int p0 = y * w + x0;
    int lab[4] = {-1, -1, -1, -1};
    if (x0 + 3 < w && (reinterpret_cast<uintptr_t>(L + p0) & 15) == 0) {
        const int4 l = *reinterpret_cast<const int4*>(L + p0);
        lab[0] = l.x; lab[1] = l.y; lab[2] = l.z; lab[3] = l.w;
    } else {
        for (int k = 0; k < 4 && x0 + k < w; ++k) lab[k] = L[p0 + k];
    }
    if (lab[0] < 0 && lab[1] < 0 && lab[2] < 0 && lab[3] < 0) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (lab[k] < 0) continue;
        const int x = x0 + k, p = p0 + k;
        if (x >= w) continue;
        const bool left = (x == 0) || (k > 0 ? lab[k - 1] < 0 : L[p - 1] < 0);
        const bool right = (x == w - 1) || (k < 3 ? lab[k + 1] < 0 : L[p + 1] < 0);
        if (!(left || right)) continue;
        const int root = uf_find(L, p);
        const int slot = atomicAdd(&cnt[f], 1);
        if (slot < cap) recs[(long)f * cap + slot] = make_int2(root | (left ? DB_LEFT : 0) | (right ? DB_RIGHT : 0), x | (y << 16));
    }
}

// Mean of prob over the lattice points inside (or on) a convex integer quad, restricted to its clipped bounding
// rectangle — paddleocr box_score_fast (fillPoly mask + cv2.mean).  One block per candidate box.
struct ScoreBox { int frame; int x0, y0, x1, y1; int qx[4], qy[4]; };
__global__ __launch_bounds__(256) void db_score_kernel(const float* __restrict__ prob, int h, int w,
                                                       const ScoreBox* __restrict__ boxes, float* __restrict__ out) {
    __shared__ double ssum[256];
    __shared__ int scnt[256];
    const ScoreBox b = boxes[blockIdx.x];
    const float* P = prob + (long)b.frame * h * w;
    const int bw = b.x1 - b.x0 + 1, bh = b.y1 - b.y0 + 1;
    double sum = 0.0;
    int cnt = 0;
    for (int i = threadIdx.x; i < bw * bh; i += blockDim.x) {
        const int lx = i % bw, ly = i / bw;   // mask-local coordinates (the quad is given mask-local too)
        bool pos = true, neg = true;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ax = b.qx[e], ay = b.qy[e], bx = b.qx[(e + 1) & 3], by = b.qy[(e + 1) & 3];
            const long cr = (long)(bx - ax) * (ly - ay) - (long)(by - ay) * (lx - ax);
            pos &= (cr >= 0);
            neg &= (cr <= 0);
        }
        if (pos || neg) {
            sum += (double)P[(long)(b.y0 + ly) * w + b.x0 + lx];
            ++cnt;
        }
    }
    ssum[threadIdx.x] = sum;
    scnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { ssum[threadIdx.x] += ssum[threadIdx.x + s]; scnt[threadIdx.x] += scnt[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = scnt[0] ? (float)(ssum[0] / scnt[0]) : 0.f;
}

#define DB_RUN_CAP 32768   // run end-point records per frame kept on the device (more -> host pass for that frame)

// Host twin of db_runs_kernel over one frame's label image (parent pointers, -1 = background).
static void db_runs_host(const int* L, int h, int w, std::vector<int2>& out) {
    out.clear();
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int p = y * w + x;
            if (L[p] < 0) continue;
            const bool left = (x == 0) || (L[p - 1] < 0);
            const bool right = (x == w - 1) || (L[p + 1] < 0);
            if (!(left || right)) continue;
            int r = p;
            while (L[r] != r) r = L[r];
            out.push_back(make_int2(r | (left ? DB_LEFT : 0) | (right ? DB_RIGHT : 0), x | (y << 16)));
        }
}

extern "C" size_t vse_db_workspace_bytes(int n, int h, int w) {
    size_t b = (size_t)n * h * w * sizeof(int);          // labels
    b += (size_t)n * DB_RUN_CAP * sizeof(int2);          // run records
    b += (size_t)n * sizeof(int) + 256;                  // counters
    b += (size_t)n * 1024 * (sizeof(ScoreBox) + sizeof(float)) + 256;   // score boxes + scores
    return b + 1024;
}

extern "C" int vse_db_postprocess(vse_ctx*, const float* d_prob, int n, int h, int w, int src_h, int src_w,
                                  const vse_db_params* prm, void* d_ws, size_t ws_bytes, vse_box* boxes, int max_boxes,
                                  int* n_boxes, void* stream) {
    if (!d_prob || !prm || !d_ws || !boxes || !n_boxes || n <= 0) return VSE_E_INVAL;
    if (ws_bytes < vse_db_workspace_bytes(n, h, w)) return VSE_E_NOMEM;
    if (w >= 65536 || h >= 32768 || (long)h * w > DB_ROOT_MASK) return VSE_E_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(d_ws);
    int* L = reinterpret_cast<int*>(ws);
    size_t off = (size_t)n * h * w * sizeof(int);
    int2* recs = reinterpret_cast<int2*>(ws + off);
    off += (size_t)n * DB_RUN_CAP * sizeof(int2);
    int* cnt = reinterpret_cast<int*>(ws + off);
    off += ((size_t)n * sizeof(int) + 255) / 256 * 256;
    ScoreBox* sboxes = reinterpret_cast<ScoreBox*>(ws + off);
    off += (size_t)n * 1024 * sizeof(ScoreBox);
    float* scores = reinterpret_cast<float*>(ws + off);

    if (h > 65535 || n > 65535) return VSE_E_UNSUPPORTED;                 // grid = (column groups of 1024 pixels, rows, frames)
    const dim3 grid((unsigned)((w + 1023) / 1024), (unsigned)h, (unsigned)n);
    if (hipMemsetAsync(cnt, 0, n * sizeof(int), st) != hipSuccess) return VSE_E_HIP;
    hipLaunchKernelGGL(db_init_kernel, grid, dim3(256), 0, st, d_prob, L, h, w, prm->thresh);
    hipLaunchKernelGGL(db_merge_kernel, grid, dim3(256), 0, st, L, h, w);
    hipLaunchKernelGGL(db_runs_kernel, grid, dim3(256), 0, st, L, h, w, recs, cnt, DB_RUN_CAP);
    std::vector<int> hcnt(n);
    if (hipMemcpyAsync(hcnt.data(), cnt, n * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return VSE_E_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return VSE_E_HIP;
    std::vector<std::vector<int2>> hrecs(n);
    int maxc = 0;
    std::vector<int> over;                                // frames whose records did not fit (noise-like maps)
    for (int f = 0; f < n; ++f) {
        if (hcnt[f] > DB_RUN_CAP) { over.push_back(f); hcnt[f] = 0; }
        maxc = std::max(maxc, hcnt[f]);
    }
    if (maxc) {
        // ONE strided copy for all frames (64 per-frame copies from pageable memory cost ~1 ms of GPU idle per step)
        std::vector<int2> flat((size_t)n * maxc);
        if (hipMemcpy2DAsync(flat.data(), (size_t)maxc * sizeof(int2), recs, (size_t)DB_RUN_CAP * sizeof(int2),
                             (size_t)maxc * sizeof(int2), n, hipMemcpyDeviceToHost, st) != hipSuccess)
            return VSE_E_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return VSE_E_HIP;
        for (int f = 0; f < n; ++f) hrecs[f].assign(flat.begin() + (size_t)f * maxc, flat.begin() + (size_t)f * maxc + hcnt[f]);
    }
    // A frame with more run end-points than the device buffer holds (TV static, a detector gone wild) degrades alone:
    // its label image is copied back and the same records are extracted on the host, so the other frames of the batch are
    // unaffected and the frame itself still yields what the reference would (up to max_candidates boxes).
    for (int f : over) {
        std::vector<int> hl((size_t)h * w);
        if (hipMemcpyAsync(hl.data(), L + (size_t)f * h * w, hl.size() * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess)
            return VSE_E_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return VSE_E_HIP;
        db_runs_host(hl.data(), h, w, hrecs[f]);
    }

    // ---- host geometry, pass 1: components -> candidate quads ----------------------------------------
    struct Cand { int frame; dbgeo::Pt box[4]; };
    std::vector<Cand> cands;
    std::vector<ScoreBox> sb;
    for (int f = 0; f < n; ++f) {
        auto& r = hrecs[f];
        // contours in cv2.findContours(RETR_LIST) order: outer borders of the 8-connected components and hole borders, most
        // recently found first = descending raster index of the pixel at which the scan meets them (a component's first
        // pixel = its union-find root; a hole's first pixel)
        struct Contour { long key; std::vector<dbgeo::IPt> pts; };
        std::vector<Contour> contours;
        {
            // foreground runs per row (for the holes): records sorted by (y, x); a LEFT record opens a run, a RIGHT one closes it
            std::vector<int2> byrow(r);
            std::sort(byrow.begin(), byrow.end(), [](const int2& a, const int2& b) {
                const int ya = a.y >> 16, yb = b.y >> 16;
                return ya < yb || (ya == yb && (a.y & 0xffff) < (b.y & 0xffff));
            });
            if (!byrow.empty()) {
                const int y_first = byrow.front().y >> 16, y_last = byrow.back().y >> 16;
                std::vector<std::vector<dbgeo::Run>> fg(y_last - y_first + 1);
                for (const int2& e : byrow) {
                    auto& row = fg[(e.y >> 16) - y_first];
                    const int x = e.y & 0xffff;
                    if (e.x & DB_LEFT) row.push_back({x, x});
                    if (e.x & DB_RIGHT) row.back().b = x;
                }
                for (auto& hb : dbgeo::hole_borders(fg, y_first, h, w)) contours.push_back({hb.key, std::move(hb.pts)});
            }
        }
        for (auto& e : r) e.x &= DB_ROOT_MASK;
        std::sort(r.begin(), r.end(), [](const int2& a, const int2& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
        for (size_t i = 0; i < r.size();) {
            size_t j = i;
            Contour c;
            c.key = r[i].x;
            while (j < r.size() && r[j].x == r[i].x) { c.pts.push_back({r[j].y & 0xffff, r[j].y >> 16}); ++j; }
            contours.push_back(std::move(c));
            i = j;
        }
        std::sort(contours.begin(), contours.end(), [](const Contour& a, const Contour& b) { return a.key > b.key; });
        int used = 0;
        for (auto& ct : contours) {
            if (used++ >= prm->max_candidates) break;
            const auto& pts = ct.pts;
            const auto hull = dbgeo::convex_hull(pts);
            const auto mr = dbgeo::min_area_rect(hull);
            Cand c;
            c.frame = f;
            const double sside = dbgeo::mini_box(mr, c.box);
            if (sside < prm->min_size) continue;
            // box_score_fast geometry
            double xmin = c.box[0].x, xmax = c.box[0].x, ymin = c.box[0].y, ymax = c.box[0].y;
            for (int k = 1; k < 4; ++k) {
                xmin = std::min(xmin, c.box[k].x); xmax = std::max(xmax, c.box[k].x);
                ymin = std::min(ymin, c.box[k].y); ymax = std::max(ymax, c.box[k].y);
            }
            // the reference casts the box to float32 before floor/ceil (np.float32 arrays)
            ScoreBox s;
            s.frame = f;
            auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
            s.x0 = clampi((int)std::floor((float)xmin), 0, w - 1);
            s.x1 = clampi((int)std::ceil((float)xmax), 0, w - 1);
            s.y0 = clampi((int)std::floor((float)ymin), 0, h - 1);
            s.y1 = clampi((int)std::ceil((float)ymax), 0, h - 1);
            for (int k = 0; k < 4; ++k) {
                s.qx[k] = (int)((float)c.box[k].x - (float)s.x0);   // astype(int32) truncation
                s.qy[k] = (int)((float)c.box[k].y - (float)s.y0);
            }
            cands.push_back(c);
            sb.push_back(s);
        }
    }
    std::vector<float> hs(sb.size());
    if (!sb.empty()) {
        if (sb.size() > (size_t)n * 1024) return VSE_E_NOMEM;
        if (hipMemcpyAsync(sboxes, sb.data(), sb.size() * sizeof(ScoreBox), hipMemcpyHostToDevice, st) != hipSuccess)
            return VSE_E_HIP;
        hipLaunchKernelGGL(db_score_kernel, dim3((unsigned)sb.size()), dim3(256), 0, st, d_prob, h, w, sboxes, scores);
        if (hipMemcpyAsync(hs.data(), scores, sb.size() * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess)
            return VSE_E_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return VSE_E_HIP;
    }
    // ---- host geometry, pass 2: score filter -> unclip -> scale -> order -> size filter ----------------
    int nb = 0;
    for (size_t i = 0; i < cands.size(); ++i) {
        if (prm->box_thresh > hs[i]) continue;
        const auto grown = dbgeo::unclip_rect(cands[i].box, prm->unclip_ratio);
        const auto hull = dbgeo::convex_hull(grown);
        const auto mr = dbgeo::min_area_rect(hull);
        dbgeo::Pt b2[4];
        const double sside = dbgeo::mini_box(mr, b2);
        if (sside < prm->min_size + 2) continue;
        dbgeo::Pt q[4];
        for (int k = 0; k < 4; ++k) {
            // np.round (half to even) on float32 values, clip, astype(int32)
            float fx = (float)b2[k].x / (float)w * (float)src_w;
            float fy = (float)b2[k].y / (float)h * (float)src_h;
            fx = std::min(std::max(std::nearbyintf(fx), 0.f), (float)src_w);
            fy = std::min(std::max(std::nearbyintf(fy), 0.f), (float)src_h);
            q[k] = {(double)(int)fx, (double)(int)fy};
        }
        dbgeo::Pt o[4];
        dbgeo::order_clockwise(q, o);
        for (int k = 0; k < 4; ++k) {   // clip_det_res
            o[k].x = (double)(int)std::min(std::max(o[k].x, 0.0), (double)(src_w - 1));
            o[k].y = (double)(int)std::min(std::max(o[k].y, 0.0), (double)(src_h - 1));
        }
        const int rw = (int)std::sqrt((o[0].x - o[1].x) * (o[0].x - o[1].x) + (o[0].y - o[1].y) * (o[0].y - o[1].y));
        const int rh = (int)std::sqrt((o[0].x - o[3].x) * (o[0].x - o[3].x) + (o[0].y - o[3].y) * (o[0].y - o[3].y));
        if (rw <= 3 || rh <= 3) continue;
        if (nb >= max_boxes) return VSE_E_NOMEM;
        vse_box& ob = boxes[nb++];
        for (int k = 0; k < 4; ++k) { ob.pts[k][0] = (float)o[k].x; ob.pts[k][1] = (float)o[k].y; }
        ob.score = hs[i];
        ob.frame = cands[i].frame;
    }
    *n_boxes = nb;
    return VSE_OK;
}

// ================================================================================================ rec pre-process
// cv2.getPerspectiveTransform (8x8 linear system, Gaussian elimination with partial pivoting in double) and
// its inverse, as warpPerspective (without WARP_INVERSE_MAP) applies it.
static bool perspective_inverse(const float src[4][2], int cw, int ch, double minv[9]) {
    const double dst[4][2] = {{0, 0}, {(double)cw, 0}, {(double)cw, (double)ch}, {0, (double)ch}};
    double a[8][9];
    for (int i = 0; i < 4; ++i) {
        const double x = src[i][0], y = src[i][1], X = dst[i][0], Y = dst[i][1];
        double r0[9] = {x, y, 1, 0, 0, 0, -x * X, -y * X, X};
        double r1[9] = {0, 0, 0, x, y, 1, -x * Y, -y * Y, Y};
        memcpy(a[i], r0, sizeof r0);
        memcpy(a[i + 4], r1, sizeof r1);
    }
    // OpenCV's LU (hal::LU64f as recalled): partial pivoting, a[j][k] += (a[j][i] * (-1 / a[i][i])) * a[i][k], back
    // substitution s / a[i][i] — the same operation sequence as oracle/pipeline_ref.py _perspective_inverse
    const double eps = 2.220446049250313e-16 * 100;
    for (int i = 0; i < 8; ++i) {
        int k = i;
        for (int j = i + 1; j < 8; ++j)
            if (std::fabs(a[j][i]) > std::fabs(a[k][i])) k = j;
        if (std::fabs(a[k][i]) < eps) return false;
        if (k != i)
            for (int c = 0; c < 9; ++c) std::swap(a[i][c], a[k][c]);
        const double d = -1.0 / a[i][i];
        for (int j = i + 1; j < 8; ++j) {
            const double alpha = a[j][i] * d;
            for (int c = i + 1; c < 8; ++c) a[j][c] += alpha * a[i][c];
            a[j][8] += alpha * a[i][8];
        }
    }
    double m[9];
    for (int i = 7; i >= 0; --i) {
        double sacc = a[i][8];
        for (int c = i + 1; c < 8; ++c) sacc -= a[i][c] * m[c];
        m[i] = sacc / a[i][i];
    }
    m[8] = 1.0;
    // invert 3x3
    const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (det == 0.0) return false;
    const double id = 1.0 / det;
    minv[0] = (m[4] * m[8] - m[5] * m[7]) * id;
    minv[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    minv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    minv[3] = (m[5] * m[6] - m[3] * m[8]) * id;
    minv[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    minv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    minv[6] = (m[3] * m[7] - m[4] * m[6]) * id;
    minv[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    minv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

struct CropDev {
    double minv[9];
    long scratch_off;     // byte offset of this crop's uint8 image in the scratch
    int frame, cw, ch;    // warp size
    int iw, ih;           // image size after the optional rot90
    int resized_w, rotate;
};

__device__ __forceinline__ void cubic_w(float x, float* c) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
    c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    c[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

// grid (ceil(maxpix/256), n_crops): bicubic perspective warp with replicate border into the scratch image
// (already rotated: np.rot90(dst) -> out[i][j] = dst[j][cw-1-i]).
__global__ __launch_bounds__(256) void crop_warp_kernel(const uint8_t* __restrict__ src, int sh, int sw, long pitch,
                                                        long fstride, const CropDev* __restrict__ crops,
                                                        uint8_t* __restrict__ scratch) {
    const CropDev c = crops[blockIdx.y];
    const int npx = c.cw * c.ch;
    const uint8_t* S = src + c.frame * fstride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
        const int x = i % c.cw, y = i / c.cw;
        // WarpPerspectiveInvoker's blocked evaluation (BLOCK_SZ = 32): the homography is taken at the block's left edge bx
        // and advanced by x1 inside the block — the same operation order as the oracle, ties at 1/32 pixel included
        const int bw0 = min(1024 / min(16, c.ch), c.cw);
        const double bx = (double)((x / bw0) * bw0), x1 = (double)(x % bw0), yd = (double)y;
        const double X0 = c.minv[0] * bx + c.minv[1] * yd + c.minv[2];
        const double Y0 = c.minv[3] * bx + c.minv[4] * yd + c.minv[5];
        const double W0 = c.minv[6] * bx + c.minv[7] * yd + c.minv[8];
        double W = W0 + c.minv[6] * x1;
        W = W != 0.0 ? 32.0 / W : 0.0;                       // INTER_TAB_SIZE = 32
        const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + c.minv[0] * x1) * W));
        const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + c.minv[3] * x1) * W));
        const int X = (int)rint(fX), Y = (int)rint(fY);      // cv::saturate_cast<int>(double) = lrint
        const int sx = (X >> 5) - 1, sy = (Y >> 5) - 1;
        float wx[4], wy[4];
        cubic_w((float)(X & 31) * (1.f / 32.f), wx);
        cubic_w((float)(Y & 31) * (1.f / 32.f), wy);
        // cv2's fixed-point table entry for this (fy, fx) (imgwarp.cpp initInterTab2D, restated in oracle/pipeline_ref.py
        // _bicubic_itab): products scaled by 2^15 and rounded to short, the rounding residue moved into the largest / smallest
        // of the entries [2..3] x [2..3]
        int it[16], isum = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = (wy[r] * wx[q]) * 32768.f;
                it[r * 4 + q] = min(max((int)rintf(v), -32768), 32767);
                isum += it[r * 4 + q];
            }
        if (isum != 32768) {
            const int diff = isum - 32768;
            int mk = 10, Mk = 10;
            const int order[4] = {10, 11, 14, 15};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pos = order[k];
                if (it[pos] < it[mk]) mk = pos;
                else if (it[pos] > it[Mk]) Mk = pos;
            }
            const int tgt = diff < 0 ? Mk : mk;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k == tgt) it[k] = (int)(short)(it[k] - diff);
        }
        int acc[3] = {0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int yy = min(max(sy + r, 0), sh - 1);
            const uint8_t* row = S + (long)yy * pitch;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int xx = min(max(sx + q, 0), sw - 1);
                const int wgt = it[r * 4 + q];
                acc[0] += wgt * (int)row[xx * 3 + 0];
                acc[1] += wgt * (int)row[xx * 3 + 1];
                acc[2] += wgt * (int)row[xx * 3 + 2];
            }
        }
        int ox = x, oy = y;
        if (c.rotate) { oy = c.cw - 1 - x; ox = y; }
        uint8_t* o = scratch + c.scratch_off + ((long)oy * c.iw + ox) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = (uint8_t)min(max((acc[k] + (1 << 14)) >> 15, 0), 255);      // FixedPtCast<int, uchar, 15>
    }
}

// grid (ceil(rec_h*rec_w/256), n_crops): cv2 fixed-point bilinear resize of the crop to (resized_w, rec_h),
// (x/255 - 0.5)/0.5, zero right-pad, fp16 NHWC(8).
__global__ __launch_bounds__(256) void crop_resize_kernel(const CropDev* __restrict__ crops, const uint8_t* __restrict__ scratch,
                                                          half_t* __restrict__ dst, int rec_h, int rec_w) {
    const CropDev c = crops[blockIdx.y];
    const uint8_t* img = scratch + c.scratch_off;
    const int npx = rec_h * rec_w;
    half_t* out = dst + (long)blockIdx.y * npx * 8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
        const int x = i % rec_w, y = i / rec_w;
        half8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        if (x < c.resized_w) {
            const LinCoef cx = lin_coef(x, c.resized_w, c.iw), cy = lin_coef(y, rec_h, c.ih);
            const int x1 = min(cx.s0 + 1, c.iw - 1), y1 = min(cy.s0 + 1, c.ih - 1);
            const uint8_t* r0 = img + (long)cy.s0 * c.iw * 3;
            const uint8_t* r1 = img + (long)y1 * c.iw * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int u;
                if (c.iw == c.resized_w && c.ih == rec_h) u = r0[cx.s0 * 3 + k];
                else u = cv_bilinear_u8(r0[cx.s0 * 3 + k], r0[x1 * 3 + k], r1[cx.s0 * 3 + k], r1[x1 * 3 + k], cx, cy);
                o[k] = (half_t)(((float)u / 255.f - 0.5f) / 0.5f);
            }
        }
        *reinterpret_cast<half8*>(out + (long)i * 8) = o;
    }
}

static inline size_t crop_bytes(int w, int h) { return ((size_t)w * h * 3 + 255) / 256 * 256; }

extern "C" size_t vse_rec_preprocess_scratch_bytes(int n_crops, int max_crop_w, int max_crop_h) {
    return (size_t)n_crops * (crop_bytes(max_crop_w, max_crop_h) + ((sizeof(CropDev) + 255) / 256 * 256)) + 4096;
}

extern "C" int vse_rec_preprocess(vse_ctx*, const void* d_bgr, int n_frames, int src_h, int src_w, int64_t pitch,
                                  int64_t frame_stride, const vse_crop* crops, int n_crops, void* d_out, int rec_h,
                                  int rec_w, void* d_scratch, size_t scratch_bytes, void* stream) {
    if (!d_bgr || !crops || !d_out || !d_scratch || n_crops <= 0) return VSE_E_INVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    std::vector<CropDev> cd(n_crops);
    size_t off = ((size_t)n_crops * sizeof(CropDev) + 255) / 256 * 256;
    int maxpix = 0;
    for (int i = 0; i < n_crops; ++i) {
        const vse_crop& c = crops[i];
        if (c.frame < 0 || c.frame >= n_frames || c.crop_w <= 0 || c.crop_h <= 0 || c.resized_w <= 0 || c.resized_w > rec_w)
            return VSE_E_INVAL;
        CropDev& d = cd[i];
        if (!perspective_inverse(c.quad, c.crop_w, c.crop_h, d.minv)) {
            // degenerate quad: identity-like mapping from the first corner
            double id[9] = {1, 0, c.quad[0][0], 0, 1, c.quad[0][1], 0, 0, 1};
            memcpy(d.minv, id, sizeof id);
        }
        d.frame = c.frame; d.cw = c.crop_w; d.ch = c.crop_h; d.rotate = c.rotate;
        d.iw = c.rotate ? c.crop_h : c.crop_w;
        d.ih = c.rotate ? c.crop_w : c.crop_h;
        d.resized_w = c.resized_w;
        d.scratch_off = (long)off;
        off += crop_bytes(c.crop_w, c.crop_h);
        maxpix = std::max(maxpix, c.crop_w * c.crop_h);
    }
    if (off > scratch_bytes) return VSE_E_NOMEM;
    // crop records go through a small ring of pinned host slots so that the upload is truly asynchronous: a stream
    // synchronise here (needed for a pageable, stack-lifetime source) made every width group wait for the previous
    // recogniser launch on its stream and left the GPU idle while the host prepared the next group
    {
        constexpr int PIN_N = 8;
        constexpr size_t PIN_BYTES = 64 * 1024;
        static struct PinRing {
            void* buf[PIN_N] = {};
            hipEvent_t ev[PIN_N] = {};
            bool used[PIN_N] = {};
            int next = 0;
            std::mutex m;
        } ring;
        const size_t bytes = (size_t)n_crops * sizeof(CropDev);
        bool done = false;
        if (bytes <= PIN_BYTES) {
            std::lock_guard<std::mutex> lk(ring.m);
            const int sl = ring.next;
            ring.next = (ring.next + 1) % PIN_N;
            bool ok = true;
            if (!ring.buf[sl]) ok = hipHostMalloc(&ring.buf[sl], PIN_BYTES, hipHostMallocDefault) == hipSuccess &&
                                    hipEventCreateWithFlags(&ring.ev[sl], hipEventDisableTiming) == hipSuccess;
            if (ok && ring.used[sl]) ok = hipEventSynchronize(ring.ev[sl]) == hipSuccess;     // slot's previous upload has finished
            if (ok) {
                memcpy(ring.buf[sl], cd.data(), bytes);
                if (hipMemcpyAsync(d_scratch, ring.buf[sl], bytes, hipMemcpyHostToDevice, st) != hipSuccess) return VSE_E_HIP;
                if (hipEventRecord(ring.ev[sl], st) != hipSuccess) return VSE_E_HIP;
                ring.used[sl] = true;
                done = true;
            }
        }
        if (!done) {     // oversized batch or no pinned memory: pageable source, keep it alive until the copy has run
            if (hipMemcpyAsync(d_scratch, cd.data(), bytes, hipMemcpyHostToDevice, st) != hipSuccess) return VSE_E_HIP;
            if (hipStreamSynchronize(st) != hipSuccess) return VSE_E_HIP;
        }
    }
    const CropDev* dcd = reinterpret_cast<const CropDev*>(d_scratch);
    uint8_t* scr = reinterpret_cast<uint8_t*>(d_scratch);
    dim3 g1(std::min((maxpix + 255) / 256, 1024), n_crops);
    hipLaunchKernelGGL(crop_warp_kernel, g1, dim3(256), 0, st, reinterpret_cast<const uint8_t*>(d_bgr), src_h, src_w,
                       (long)pitch, (long)frame_stride, dcd, scr);
    dim3 g2((rec_h * rec_w + 255) / 256, n_crops);
    hipLaunchKernelGGL(crop_resize_kernel, g2, dim3(256), 0, st, dcd, scr, reinterpret_cast<half_t*>(d_out), rec_h, rec_w);
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}

// ================================================================================================ CTC collapse
// One wave per sequence: lane l looks at t = base + l, keep = idx[t] != 0 && idx[t] != idx[t-1]; the 64-bit
// ballot gives each kept element its output slot by a popcount of the lower lanes (wavefront scan).
// tlen (ragged batch): the sequence length of each row; time steps at or behind it are not part of the sample.
__global__ __launch_bounds__(256) void ctc_collapse_kernel(const int2* __restrict__ ip, int b, int tfull, const int* __restrict__ tlen,
                                                           int* __restrict__ out_idx, int* __restrict__ out_len,
                                                           float* __restrict__ out_conf) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= b) return;
    const int2* r = ip + (long)row * tfull;
    const int t = tlen != nullptr ? min(max(tlen[row], 0), tfull) : tfull;
    int n = 0;
    float sum = 0.f;
    for (int base = 0; base < t; base += 64) {
        const int tt = base + lane;
        int idx = 0, prev = -1;
        float p = 0.f;
        if (tt < t) {
            const int2 v = r[tt];
            idx = v.x;
            p = __int_as_float(v.y);
            prev = tt > 0 ? r[tt - 1].x : -1;
        }
        const bool keep = (tt < t) && idx != 0 && idx != prev;
        const unsigned long long m = __ballot(keep);
        if (keep) out_idx[(long)row * tfull + n + __popcll(m & ((1ull << lane) - 1ull))] = idx;
        float ps = keep ? p : 0.f;
        for (int o = 32; o >= 1; o >>= 1) ps += __shfl_xor(ps, o);
        sum += ps;
        n += __popcll(m);
    }
    if (lane == 0) {
        out_len[row] = n;
        out_conf[row] = n ? sum / (float)n : 0.f;
    }
}

extern "C" int vse_ctc_collapse_ragged(vse_ctx*, const void* d_idx_maxp, int b, int t, const int32_t* d_tlen, int32_t* d_out_idx,
                                       int32_t* d_out_len, float* d_out_conf, void* stream) {
    if (!d_idx_maxp || !d_out_idx || !d_out_len || !d_out_conf || b <= 0 || t <= 0) return VSE_E_INVAL;
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3((b + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const int2*>(d_idx_maxp), b, t, d_tlen, d_out_idx, d_out_len, d_out_conf);
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
extern "C" int vse_ctc_collapse(vse_ctx* c, const void* d_idx_maxp, int b, int t, int32_t* d_out_idx, int32_t* d_out_len,
                                float* d_out_conf, void* stream) {
    return vse_ctc_collapse_ragged(c, d_idx_maxp, b, t, nullptr, d_out_idx, d_out_len, d_out_conf, stream);
}
