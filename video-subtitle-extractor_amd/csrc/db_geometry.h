// Host-side geometry of DB post-processing (paddleocr DBPostProcess.boxes_from_bitmap),
// restated without OpenCV / pyclipper: convex hull of a component's run end-points, min-area rectangle by
// rotating the support edge over the hull, paddleocr's corner ordering, closed-form "unclip" of a rectangle,
// scaling to source pixels, clockwise ordering and the final size filter.
// Plain double arithmetic, no FMA contraction (compiled with -ffp-contract=off) so that the numpy oracle
// (oracle/db_ref.py), which follows the same formulas, agrees to the last bit in all but half-way roundings.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace dbgeo {

struct Pt { double x, y; };
struct IPt { int x, y; };

// Andrew monotone chain on integer points; returns hull in counter-clockwise order (y down => visually clockwise),
// collinear points removed.
inline std::vector<IPt> convex_hull(std::vector<IPt> p) {
    std::sort(p.begin(), p.end(), [](const IPt& a, const IPt& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    p.erase(std::unique(p.begin(), p.end(), [](const IPt& a, const IPt& b) { return a.x == b.x && a.y == b.y; }), p.end());
    const int n = (int)p.size();
    if (n < 3) return p;
    std::vector<IPt> h(2 * n);
    int k = 0;
    auto cross = [](const IPt& o, const IPt& a, const IPt& b) {
        return (int64_t)(a.x - o.x) * (b.y - o.y) - (int64_t)(a.y - o.y) * (b.x - o.x);
    };
    for (int i = 0; i < n; ++i) {
        while (k >= 2 && cross(h[k - 2], h[k - 1], p[i]) <= 0) --k;
        h[k++] = p[i];
    }
    for (int i = n - 2, t = k + 1; i >= 0; --i) {
        while (k >= t && cross(h[k - 2], h[k - 1], p[i]) <= 0) --k;
        h[k++] = p[i];
    }
    h.resize(k - 1);
    return h;
}

struct MinRect {
    Pt corner[4];   // in rectangle order around the perimeter
    double w, h;    // side lengths
};

// Min-area enclosing rectangle of a convex polygon: try every hull edge as the support direction, keep the
// first strictly smallest area (hull order as returned by convex_hull).  Degenerate hulls (1 or 2 points)
// give zero-area rectangles like cv2.minAreaRect.
inline MinRect min_area_rect(const std::vector<IPt>& hull) {
    MinRect r{};
    const int n = (int)hull.size();
    if (n == 0) return r;
    if (n == 1) {
        for (auto& c : r.corner) c = {(double)hull[0].x, (double)hull[0].y};
        return r;
    }
    double best = -1.0;
    const int edges = (n == 2) ? 1 : n;
    for (int i = 0; i < edges; ++i) {
        const IPt a = hull[i], b = hull[(i + 1) % n];
        const double ex = (double)(b.x - a.x), ey = (double)(b.y - a.y);
        const double len2 = ex * ex + ey * ey;
        double umin = 0, umax = 0, vmin = 0, vmax = 0;
        for (int j = 0; j < n; ++j) {
            const double px = (double)(hull[j].x - a.x), py = (double)(hull[j].y - a.y);
            const double u = px * ex + py * ey;       // along the edge (scaled by |e|)
            const double v = py * ex - px * ey;       // across the edge (scaled by |e|)
            if (j == 0) { umin = umax = u; vmin = vmax = v; }
            else { umin = std::min(umin, u); umax = std::max(umax, u); vmin = std::min(vmin, v); vmax = std::max(vmax, v); }
        }
        const double area = (umax - umin) * (vmax - vmin) / len2;
        if (best < 0 || area < best) {
            best = area;
            const double ux = ex / len2, uy = ey / len2;   // u/len2 * e = coordinates
            auto at = [&](double u, double v) -> Pt {
                return {(double)a.x + u * ux - v * uy, (double)a.y + u * uy + v * ux};
            };
            r.corner[0] = at(umin, vmin);
            r.corner[1] = at(umax, vmin);
            r.corner[2] = at(umax, vmax);
            r.corner[3] = at(umin, vmax);
            const double len = std::sqrt(len2);
            r.w = (umax - umin) / len;
            r.h = (vmax - vmin) / len;
        }
    }
    return r;
}

// paddleocr get_mini_boxes ordering: sort the 4 corners by x (stable), then pick tl,tr,br,bl by y inside the
// left pair and the right pair.  Returns min side.
inline double mini_box(const MinRect& r, Pt out[4]) {
    Pt p[4] = {r.corner[0], r.corner[1], r.corner[2], r.corner[3]};
    std::stable_sort(p, p + 4, [](const Pt& a, const Pt& b) { return a.x < b.x; });
    int i1, i2, i3, i4;
    if (p[1].y > p[0].y) { i1 = 0; i4 = 1; } else { i1 = 1; i4 = 0; }
    if (p[3].y > p[2].y) { i2 = 2; i3 = 3; } else { i2 = 3; i3 = 2; }
    out[0] = p[i1]; out[1] = p[i2]; out[2] = p[i3]; out[3] = p[i4];
    return std::min(r.w, r.h);
}

inline double poly_area(const Pt* p, int n) {
    double s = 0;
    for (int i = 0; i < n; ++i) {
        const Pt& a = p[i];
        const Pt& b = p[(i + 1) % n];
        s += a.x * b.y - b.x * a.y;
    }
    return std::fabs(s) * 0.5;
}
inline double poly_len(const Pt* p, int n) {
    double s = 0;
    for (int i = 0; i < n; ++i) {
        const double dx = p[(i + 1) % n].x - p[i].x, dy = p[(i + 1) % n].y - p[i].y;
        s += std::sqrt(dx * dx + dy * dy);
    }
    return s;
}

// Closed-form restatement of paddleocr unclip() for a rectangle (App. C.3): distance = area*ratio/perimeter on
// the float box; Clipper works on the integer-truncated corners and returns integer vertices; the min-area
// rectangle of a round-joined offset of a rectangle is that rectangle grown by `distance` on every side.
// Returns the 4 grown corners rounded to the integer grid (Clipper output precision), as hull input.
inline std::vector<IPt> unclip_rect(const Pt box[4], double ratio) {
    const double dist = poly_area(box, 4) * ratio / poly_len(box, 4);
    Pt q[4];
    for (int i = 0; i < 4; ++i) q[i] = {std::trunc(box[i].x), std::trunc(box[i].y)};
    // edge directions from the (tl,tr,br,bl) ordering
    double ux = q[1].x - q[0].x, uy = q[1].y - q[0].y;
    double vx = q[3].x - q[0].x, vy = q[3].y - q[0].y;
    const double ul = std::sqrt(ux * ux + uy * uy), vl = std::sqrt(vx * vx + vy * vy);
    if (ul > 0) { ux /= ul; uy /= ul; } else { ux = 1; uy = 0; }
    if (vl > 0) { vx /= vl; vy /= vl; } else { vx = -uy; vy = ux; }
    const double su[4] = {-1, 1, 1, -1}, sv[4] = {-1, -1, 1, 1};
    std::vector<IPt> out(4);
    for (int i = 0; i < 4; ++i) {
        const double x = q[i].x + dist * (su[i] * ux + sv[i] * vx);
        const double y = q[i].y + dist * (su[i] * uy + sv[i] * vy);
        out[i] = {(int)std::nearbyint(x), (int)std::nearbyint(y)};
    }
    return out;
}

// paddleocr 2.10 order_points_clockwise (tools/infer/predict_det.py): tl = argmin(x+y), br = argmax(x+y); of the
// two remaining points tr = argmin(y-x), bl = argmax(y-x) (numpy first-index tie rule).
inline void order_clockwise(const Pt in[4], Pt out[4]) {
    int imin = 0, imax = 0;
    for (int i = 1; i < 4; ++i) {
        if (in[i].x + in[i].y < in[imin].x + in[imin].y) imin = i;
        if (in[i].x + in[i].y > in[imax].x + in[imax].y) imax = i;
    }
    Pt rest[2];
    int k = 0;
    for (int i = 0; i < 4 && k < 2; ++i)
        if (i != imin && i != imax) rest[k++] = in[i];
    if (k < 2) {   // degenerate: argmin == argmax; numpy deletes one row and keeps three -> use the first two
        k = 0;
        for (int i = 0; i < 4 && k < 2; ++i)
            if (i != imin) rest[k++] = in[i];
    }
    const double d0 = rest[0].y - rest[0].x, d1 = rest[1].y - rest[1].x;
    out[0] = in[imin];
    out[2] = in[imax];
    out[1] = (d1 < d0) ? rest[1] : rest[0];
    out[3] = (d1 > d0) ? rest[1] : rest[0];
}


// ---- hole borders (cv2.findContours with RETR_LIST returns them beside the outer borders) -------------------------------
// Input: the foreground runs of one frame, per row, sorted by column.  A hole = a 4-connected background region that does
// not reach the image frame; its border = the foreground pixels 4-adjacent to it (Suzuki-Abe border points of the
// 8-connected case).  Only hull-relevant border points are emitted (run neighbours left / right, and the first / last
// foreground pixel above and below each hole run inside every foreground run that overlaps it).
struct Run { int a, b; };                     // columns [a, b] inclusive
struct HoleBorder { long key; std::vector<IPt> pts; };    // key = raster index of the hole's first pixel

inline std::vector<HoleBorder> hole_borders(const std::vector<std::vector<Run>>& fg, int y_first, int h, int w) {
    // fg[i] = runs of row y_first + i (rows outside [y_first, y_first + fg.size()) hold no foreground)
    struct Bg { int y, a, b, parent; bool frame; };
    std::vector<Bg> bg;
    std::vector<int> row_begin(fg.size() + 1, 0);
    for (size_t i = 0; i < fg.size(); ++i) {
        const int y = y_first + (int)i;
        row_begin[i] = (int)bg.size();
        const auto& r = fg[i];
        const bool edge_row = (y == 0) || (y == h - 1);
        if (r.empty()) { bg.push_back({y, 0, w - 1, 0, true}); continue; }
        if (r[0].a > 0) bg.push_back({y, 0, r[0].a - 1, 0, true});
        for (size_t k = 0; k + 1 < r.size(); ++k) bg.push_back({y, r[k].b + 1, r[k + 1].a - 1, 0, edge_row});
        if (r.back().b < w - 1) bg.push_back({y, r.back().b + 1, w - 1, 0, true});
    }
    row_begin[fg.size()] = (int)bg.size();
    for (size_t i = 0; i < bg.size(); ++i) bg[i].parent = (int)i;
    auto find = [&](int a) { while (bg[a].parent != a) { bg[a].parent = bg[bg[a].parent].parent; a = bg[a].parent; } return a; };
    auto unite = [&](int a, int b) {
        a = find(a); b = find(b);
        if (a == b) return;
        if (a > b) std::swap(a, b);
        bg[b].parent = a;                       // the smaller index (earlier in raster order) stays the root
        bg[a].frame = bg[a].frame || bg[b].frame;
    };
    // the rows just outside the band are all background and reach the frame
    auto touch_frame = [&](size_t i) { for (int k = row_begin[i]; k < row_begin[i + 1]; ++k) bg[find(k)].frame = true; };
    if (!fg.empty()) { touch_frame(0); touch_frame(fg.size() - 1); }
    for (size_t i = 1; i < fg.size(); ++i) {
        int p = row_begin[i - 1], q = row_begin[i];
        const int pe = row_begin[i], qe = row_begin[i + 1];
        while (p < pe && q < qe) {
            if (bg[p].a <= bg[q].b && bg[q].a <= bg[p].b) unite(p, q);
            if (bg[p].b < bg[q].b) ++p; else ++q;
        }
    }
    std::vector<HoleBorder> out;
    std::vector<int> slot(bg.size(), -1);
    for (size_t k = 0; k < bg.size(); ++k) {
        const int root = find((int)k);
        if (bg[root].frame) continue;
        if (slot[root] < 0) {
            slot[root] = (int)out.size();
            out.push_back({(long)bg[root].y * w + bg[root].a, {}});    // root = first run in raster order
        }
        auto& pts = out[slot[root]].pts;
        const Bg& g = bg[k];
        pts.push_back({g.a - 1, g.y});
        pts.push_back({g.b + 1, g.y});
        for (int dy = -1; dy <= 1; dy += 2) {
            const int i = g.y + dy - y_first;
            if (i < 0 || i >= (int)fg.size()) continue;               // cannot happen for a hole; kept for safety
            for (const Run& r : fg[i]) {
                if (r.b < g.a) continue;
                if (r.a > g.b) break;
                pts.push_back({std::max(r.a, g.a), g.y + dy});
                pts.push_back({std::min(r.b, g.b), g.y + dy});
            }
        }
    }
    return out;
}

}  // namespace dbgeo
