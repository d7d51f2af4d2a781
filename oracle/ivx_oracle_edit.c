/* ivx_oracle_edit.c -- TEST INFRASTRUCTURE ONLY (see ivx_oracle.c header).
 *
 * CPU restatements of the 3-D mask editing kernels either side of the hot path:
 *   orc_mask_cut        invesalius_rs/src/mask_cut.rs:7-61      (caller mask3d_editor_state.py:221)
 *   orc_brush_mask      invesalius_rs/src/brush_mask.rs:5-71    (caller mask3d_editor_state.py:259)
 *   orc_polygon2mask    invesalius_rs/src/polygon_mask.rs:4-79  (caller mask3d_editor_state.py:176)
 *   orc_count_regions   invesalius_rs/src/count_regions.rs:5-18 (python wrapper invesalius_rs/__init__.py:108-111)
 * The reference is Rust (no toolchain here) and its test-suite holds no vectors for them: PARITY UNPINNED beyond
 * these statement-by-statement restatements.  All arithmetic is float64, written in the reference's order
 * (matrix x vector as nalgebra accumulates it: column by column, left to right; no FMA contraction). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline double row_dot(const double *m, int i, const double p[4]) {
    return ((m[4 * i] * p[0] + m[4 * i + 1] * p[1]) + m[4 * i + 2] * p[2]) + m[4 * i + 3] * p[3];
}

/* Rust `f64 as usize`: NaN -> 0, saturating at both ends */
static inline uint64_t f2usize(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)v;
}
/* Rust `f64 as isize` */
static inline int64_t f2isize(double v) {
    if (v != v) return 0;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    if (v >= 9223372036854775807.0) return INT64_MAX;
    return (int64_t)v;
}

/* out: (d,h_,w_) uint8, modified in place.  mask: (mh, mw) bytes (numpy bool). m, mv: 4x4 row-major. */
int orc_mask_cut(double sx, double sy, double sz, double max_depth, const uint8_t *mask, int64_t mh, int64_t mw,
                 const double *m, const double *mv, uint8_t *out, const int64_t shape[3], int edit_mode) {
    for (int64_t z = 0; z < shape[0]; z++)
        for (int64_t y = 0; y < shape[1]; y++)
            for (int64_t x = 0; x < shape[2]; x++) {
                uint8_t *val = out + (z * shape[1] + y) * shape[2] + x;
                if (!(*val > 127)) continue;
                const double p[4] = {(double)x * sx, (double)y * sy, (double)z * sz, 1.0};
                const double q3 = row_dot(m, 3, p);
                if (!(q3 > 0.0)) continue;
                const double q0 = row_dot(m, 0, p) / q3, q1 = row_dot(m, 1, p) / q3;
                const double c3 = row_dot(mv, 3, p);
                const double c0 = row_dot(mv, 0, p) / c3, c1 = row_dot(mv, 1, p) / c3, c2 = row_dot(mv, 2, p) / c3;
                const double dist = sqrt((c0 * c0 + c1 * c1) + c2 * c2);
                if (!(dist <= max_depth)) continue;
                const double px = (q0 / 2.0 + 0.5) * (double)(mw - 1);
                const double py = (q1 / 2.0 + 0.5) * (double)(mh - 1);
                if (px >= 0.0 && px < (double)mw && py >= 0.0 && py < (double)mh) {
                    if (mask[(int64_t)f2usize(py) * mw + (int64_t)f2usize(px)]) *val = 0;
                } else if (edit_mode == 0) {
                    *val = 0;
                }
            }
    return 0;
}

int orc_brush_mask(uint8_t *out, const uint8_t *orig, const int64_t shape[3], const double spacing[3],
                   const double center[3], double radius, int edit_mode) {
    const int64_t d = shape[0], h = shape[1], w = shape[2];
    if (d == 0 || h == 0 || w == 0) return 0;
    const double sx = spacing[0], sy = spacing[1], sz = spacing[2], cx = center[0], cy = center[1], cz = center[2];
    const uint64_t min_x = f2usize(fmax(floor((cx - radius) / sx), 0.0));
    const uint64_t max_x = f2usize(fmin(fmax(ceil((cx + radius) / sx), 0.0), (double)(w - 1)));
    const uint64_t min_y = f2usize(fmax(floor((cy - radius) / sy), 0.0));
    const uint64_t max_y = f2usize(fmin(fmax(ceil((cy + radius) / sy), 0.0), (double)(h - 1)));
    const uint64_t min_z = f2usize(fmax(floor((cz - radius) / sz), 0.0));
    const uint64_t max_z = f2usize(fmin(fmax(ceil((cz + radius) / sz), 0.0), (double)(d - 1)));
    const double radius_sq = radius * radius;
    for (int64_t z = 0; z < d; z++)
        for (int64_t y = 0; y < h; y++)
            for (int64_t x = 0; x < w; x++) {
                if (!((uint64_t)z >= min_z && (uint64_t)z <= max_z && (uint64_t)y >= min_y && (uint64_t)y <= max_y &&
                      (uint64_t)x >= min_x && (uint64_t)x <= max_x))
                    continue;
                uint8_t *val = out + (z * h + y) * w + x;
                const double dx = (double)x * sx - cx, dy = (double)y * sy - cy, dz = (double)z * sz - cz;
                const double dist_sq = (dx * dx + dy * dy) + dz * dz;
                if (edit_mode == 1) {
                    if (*val > 0 && dist_sq <= radius_sq) *val = 0;
                } else if (edit_mode == 0) {
                    if (dist_sq <= radius_sq) {
                        if (orig) {
                            const uint8_t o = orig[(z * h + y) * w + x];
                            if (o > 0) *val = o;
                        } else {
                            *val = 255;
                        }
                    }
                }
            }
    return 0;
}

/* Rust f64::max / f64::min ignore a NaN operand; C fmax/fmin do the same */

/* out: (w, h) bytes; points: (n, 2) */
int orc_polygon2mask(int64_t w, int64_t h, const double *pts, int64_t n, uint8_t *out) {
    memset(out, 0, (size_t)(w * h));
    if (n == 0 || w == 0 || h == 0) return 0;
    double min_px = 1.7976931348623157e308, max_px = -1.7976931348623157e308, min_py = min_px, max_py = max_px;
    for (int64_t i = 0; i < n; i++) {
        if (pts[2 * i] < min_px) min_px = pts[2 * i];
        if (pts[2 * i] > max_px) max_px = pts[2 * i];
        if (pts[2 * i + 1] < min_py) min_py = pts[2 * i + 1];
        if (pts[2 * i + 1] > max_py) max_py = pts[2 * i + 1];
    }
    /* (x.floor() as isize - 1).max(0) as usize, then .min(w): isize arithmetic wraps in release builds; the
     * saturated extremes only arise from non-finite points, restated with saturating adds */
    int64_t a = f2isize(floor(min_px));
    a = a == INT64_MIN ? a : a - 1;
    int64_t b = f2isize(ceil(max_px));
    b = b == INT64_MAX ? b : b + 1;
    uint64_t min_x = (uint64_t)(a > 0 ? a : 0), max_x = (uint64_t)(b > 0 ? b : 0);
    if (min_x > (uint64_t)w) min_x = (uint64_t)w;
    if (max_x > (uint64_t)w) max_x = (uint64_t)w;
    a = f2isize(floor(min_py));
    a = a == INT64_MIN ? a : a - 1;
    b = f2isize(ceil(max_py));
    b = b == INT64_MAX ? b : b + 1;
    uint64_t min_y = (uint64_t)(a > 0 ? a : 0), max_y = (uint64_t)(b > 0 ? b : 0);
    if (min_y > (uint64_t)h) min_y = (uint64_t)h;
    if (max_y > (uint64_t)h) max_y = (uint64_t)h;
    for (int64_t r = 0; r < w; r++) {
        if (!((uint64_t)r >= min_x && (uint64_t)r <= max_x)) continue;
        const double px = (double)r;
        for (int64_t c = 0; c < h; c++) {
            if (!((uint64_t)c >= min_y && (uint64_t)c <= max_y)) continue;
            const double py = (double)c;
            int inside = 0;
            int64_t j = n - 1;
            for (int64_t i = 0; i < n; i++) {
                const double xi = pts[2 * i], yi = pts[2 * i + 1], xj = pts[2 * j], yj = pts[2 * j + 1];
                const int intersect = ((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi) + xi);
                if (intersect) inside = !inside;
                j = i;
            }
            out[r * h + c] = (uint8_t)inside;
        }
    }
    return 0;
}

/* labels as int64 (the wrapper widens i16/i32); returns -1 where the reference would panic (index out of bounds) */
int orc_count_regions(const int64_t *labels, int64_t n, int64_t number_regions, uint32_t *out) {
    uint32_t *counts = calloc((size_t)number_regions + 1, sizeof(uint32_t));
    if (!counts) return -2;
    for (int64_t i = 0; i < n; i++) {
        if (labels[i] < 0 || labels[i] > number_regions) {
            free(counts);
            return -1;
        }
        counts[labels[i]]++;
    }
    for (int64_t i = 0; i < n; i++) out[i] = counts[labels[i]];
    free(counts);
    return 0;
}

/* jump_flooding_internal                   invesalius_rs/src/floodfill.rs:298-507
 * 3-D jump flooding (Voronoi owners + distance to the owning site) with floor(log2(max_dim)) passes whose 26 taps sit at
 * offsets (size / 2) / 2^pass per axis; every pass reads the previous pass's arrays only (the Rust code double-buffers), so
 * the result does not depend on any traversal order.  Taps are visited z-major (zi, yi, xi ascending) and a tap wins
 * only with a strictly smaller distance -- or unconditionally while the voxel has no owner yet.  normalize: sites move to
 * the (integer) centroid of their cells, distances are recomputed to the new sites and divided by the cell's maximum.
 * Arrays are dense C-order [z][y][x]; sites are (z, y, x) int32 triples.  Test infrastructure only.
 * PARITY UNPINNED: the reference tree holds no test or golden vector for this function and its Rust source cannot be
 * built here; the restatement is checked against a brute-force Voronoi diagram (tests/test_oracle_golden.py). */
int orc_jump_flooding(float *dist, int32_t *owners, const int64_t shape[3], const int32_t *sites, int64_t nsites,
                      int normalize) {
    const int64_t sz = shape[0], sy = shape[1], sx = shape[2];
    if (nsites == 0 || sz == 0 || sy == 0 || sx == 0) return 0;
    const size_t n = (size_t)sz * sy * sx;
    int32_t *oc = (int32_t *)malloc(n * 4), *on = (int32_t *)malloc(n * 4);
    float *dc = (float *)malloc(n * 4), *dn = (float *)malloc(n * 4);
    if (!oc || !on || !dc || !dn) { free(oc); free(on); free(dc); free(dn); return -3; }
    memcpy(oc, owners, n * 4);
    memcpy(dc, dist, n * 4);
    for (int64_t i = 0; i < nsites; i++) {
        const int32_t z = sites[3 * i], y = sites[3 * i + 1], x = sites[3 * i + 2];
        if (z < 0 || y < 0 || x < 0 || z >= sz || y >= sy || x >= sx) continue;
        oc[((size_t)z * sy + y) * sx + x] = (int32_t)i + 1;
        dc[((size_t)z * sy + y) * sx + x] = 0.0f;
    }
    int64_t max_dim = sx > sy ? sx : sy;
    if (sz > max_dim) max_dim = sz;
    int n_steps = 0;
    if (max_dim > 1)
        while ((max_dim >> (n_steps + 1)) > 0) n_steps++;
    int64_t ox = sx / 2, oy = sy / 2, oz = sz / 2;
    memcpy(on, oc, n * 4);
    memcpy(dn, dc, n * 4);
    for (int s = 0; s < n_steps; s++) {
        for (int64_t z = 0; z < sz; z++)
            for (int64_t y = 0; y < sy; y++)
                for (int64_t x = 0; x < sx; x++) {
                    const size_t v = ((size_t)z * sy + y) * sx + x;
                    int32_t idx0 = oc[v];
                    float best = dc[v];
                    for (int zi = -1; zi <= 1; zi++)
                        for (int yi = -1; yi <= 1; yi++)
                            for (int xi = -1; xi <= 1; xi++) {
                                if (!xi && !yi && !zi) continue;
                                const int64_t tz = z + zi * oz, ty = y + yi * oy, tx = x + xi * ox;
                                if (tz < 0 || ty < 0 || tx < 0 || tz >= sz || ty >= sy || tx >= sx) continue;
                                const int32_t idx1 = oc[((size_t)tz * sy + ty) * sx + tx];
                                if (idx1 <= 0) continue;
                                const int64_t si = (int64_t)idx1 - 1;
                                if (si >= nsites) continue;
                                const float z1 = (float)sites[3 * si], y1 = (float)sites[3 * si + 1], x1 = (float)sites[3 * si + 2];
                                const float dz = (float)z - z1, dy = (float)y - y1, dx = (float)x - x1;
                                const float d1 = sqrtf(dz * dz + dy * dy + dx * dx);
                                if (idx0 > 0) {
                                    if (d1 < best) { idx0 = idx1; best = d1; }
                                } else { idx0 = idx1; best = d1; }
                            }
                    on[v] = idx0;
                    dn[v] = best;
                }
        { int32_t *t = oc; oc = on; on = t; }
        { float *t = dc; dc = dn; dn = t; }
        ox /= 2; oy /= 2; oz /= 2;
    }
    if (normalize) {
        uint32_t *cnt = (uint32_t *)calloc((size_t)nsites, 4);
        int64_t *sum = (int64_t *)calloc((size_t)nsites * 3, 8);
        int32_t *ns = (int32_t *)calloc((size_t)nsites * 3, 4);
        float *mx = (float *)calloc((size_t)nsites, 4);
        if (!cnt || !sum || !ns || !mx) { free(cnt); free(sum); free(ns); free(mx); free(oc); free(on); free(dc); free(dn); return -3; }
        for (int64_t z = 0; z < sz; z++)
            for (int64_t y = 0; y < sy; y++)
                for (int64_t x = 0; x < sx; x++) {
                    const int32_t o = oc[((size_t)z * sy + y) * sx + x];
                    if (o <= 0 || (int64_t)o - 1 >= nsites) continue;
                    cnt[o - 1]++;
                    sum[3 * (o - 1)] += z; sum[3 * (o - 1) + 1] += y; sum[3 * (o - 1) + 2] += x;
                }
        for (int64_t i = 0; i < nsites; i++)
            if (cnt[i]) for (int q = 0; q < 3; q++) ns[3 * i + q] = (int32_t)(sum[3 * i + q] / (int64_t)cnt[i]);
        for (int64_t z = 0; z < sz; z++)
            for (int64_t y = 0; y < sy; y++)
                for (int64_t x = 0; x < sx; x++) {
                    const size_t v = ((size_t)z * sy + y) * sx + x;
                    const int32_t o = oc[v];
                    if (o <= 0 || (int64_t)o - 1 >= nsites) continue;
                    const float dz = (float)z - (float)ns[3 * (o - 1)], dy = (float)y - (float)ns[3 * (o - 1) + 1],
                                dx = (float)x - (float)ns[3 * (o - 1) + 2];
                    const float d = sqrtf(dz * dz + dy * dy + dx * dx);
                    dc[v] = d;
                    if (d > mx[o - 1]) mx[o - 1] = d;
                }
        for (size_t v = 0; v < n; v++) {
            const int32_t o = oc[v];
            if (o <= 0 || (int64_t)o - 1 >= nsites) continue;
            if (mx[o - 1] > 0.0f) dc[v] /= mx[o - 1];
        }
        free(cnt); free(sum); free(ns); free(mx);
    }
    memcpy(owners, oc, n * 4);
    memcpy(dist, dc, n * 4);
    free(oc); free(on); free(dc); free(dn);
    return 0;
}
