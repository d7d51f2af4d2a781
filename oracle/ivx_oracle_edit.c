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
