// k_edit.hip -- 3-D mask editing kernels either side of the hot path.
//
// Reference semantics (bit-exact; all arithmetic float64 in the reference's order, no FMA contraction):
//   mask_cut        invesalius_rs/src/mask_cut.rs:7-61      zero the voxels (> 127) a screen-space polygon covers
//   brush_mask_rs   invesalius_rs/src/brush_mask.rs:5-71    spherical erase / reveal brush
//   polygon2mask_rs invesalius_rs/src/polygon_mask.rs:4-79  even-odd ray casting of a polygon onto a (w,h) grid
//   count_regions   invesalius_rs/src/count_regions.rs:5-18 voxel count of each voxel's label
//
// MI355X design: all four are streaming byte/word passes bound by HBM (1 B/voxel read, rare writes); no LDS tiling,
// no MFMA.  mask_cut moves 16 voxels per lane and does its float64 projection only for voxels that are set;
// brush_mask launches over the brush's bounding box only; count_regions aggregates a wave's shared label into one
// atomic (neighbouring voxels mostly share a label) and then gathers.
#include <algorithm>
#include <cmath>

#include "ivx_internal.h"

typedef unsigned char uchar16_t __attribute__((ext_vector_type(16)));

namespace {

struct Mat4 {
    double m[16];
};
__device__ __forceinline__ double row_dot(const Mat4 &a, int i, double p0, double p1, double p2, double p3) {
    // nalgebra Matrix4 * Vector4: column by column, left to right
    return ((a.m[4 * i] * p0 + a.m[4 * i + 1] * p1) + a.m[4 * i + 2] * p2) + a.m[4 * i + 3] * p3;
}
// Rust `f64 as usize` for the non-negative, in-range values that reach it here
__device__ __forceinline__ int64_t f2idx(double v) { return v > 0.0 ? (int64_t)v : 0; }

__device__ __forceinline__ bool cut_voxel(int64_t x, int64_t y, int64_t z, double sx, double sy, double sz, double max_depth,
                                          const uint8_t *__restrict__ mask, int64_t mh, int64_t mw, const Mat4 &m,
                                          const Mat4 &mv, int edit_mode) {
    const double p0 = (double)x * sx, p1 = (double)y * sy, p2 = (double)z * sz;
    const double q3 = row_dot(m, 3, p0, p1, p2, 1.0);
    if (!(q3 > 0.0)) return false;
    const double q0 = row_dot(m, 0, p0, p1, p2, 1.0) / q3, q1 = row_dot(m, 1, p0, p1, p2, 1.0) / q3;
    const double c3 = row_dot(mv, 3, p0, p1, p2, 1.0);
    const double c0 = row_dot(mv, 0, p0, p1, p2, 1.0) / c3, c1 = row_dot(mv, 1, p0, p1, p2, 1.0) / c3,
                 c2 = row_dot(mv, 2, p0, p1, p2, 1.0) / c3;
    const double dist = sqrt((c0 * c0 + c1 * c1) + c2 * c2);
    if (!(dist <= max_depth)) return false;
    const double px = (q0 / 2.0 + 0.5) * (double)(mw - 1);
    const double py = (q1 / 2.0 + 0.5) * (double)(mh - 1);
    if (px >= 0.0 && px < (double)mw && py >= 0.0 && py < (double)mh) return mask[f2idx(py) * mw + f2idx(px)] != 0;
    return edit_mode == 0;
}

__global__ __launch_bounds__(256) void k_mask_cut(uint8_t *__restrict__ out, int64_t n, int64_t h, int64_t w, double sx,
                                                  double sy, double sz, double max_depth, const uint8_t *__restrict__ mask,
                                                  int64_t mh, int64_t mw, Mat4 m, Mat4 mv, int edit_mode, int vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        const int64_t nchunks = n / 16;
        uchar16_t *o16 = reinterpret_cast<uchar16_t *>(out);
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += stride) {
            uchar16_t v = o16[c];
            bool any = false;
#pragma unroll
            for (int i = 0; i < 16; i++) any |= v[i] > 127;
            if (!any) continue;
            int64_t x = (c * 16) % w, r = (c * 16) / w;
            int64_t y = r % h, z = r / h;
            bool changed = false;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (v[i] > 127 && cut_voxel(x, y, z, sx, sy, sz, max_depth, mask, mh, mw, m, mv, edit_mode)) {
                    v[i] = 0;
                    changed = true;
                }
                if (++x == w) {
                    x = 0;
                    if (++y == h) {
                        y = 0;
                        z++;
                    }
                }
            }
            if (changed) o16[c] = v;
        }
    }
    const int64_t begin = vec ? (n / 16) * 16 : 0;
    for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!(out[i] > 127)) continue;
        const int64_t x = i % w, r = i / w;
        if (cut_voxel(x, r % h, r / h, sx, sy, sz, max_depth, mask, mh, mw, m, mv, edit_mode)) out[i] = 0;
    }
}

struct Box {
    int64_t x0, x1, y0, y1, z0, z1; // inclusive
};
__global__ __launch_bounds__(256) void k_brush(uint8_t *__restrict__ out, const uint8_t *__restrict__ orig, int64_t h,
                                               int64_t w, Box b, double sx, double sy, double sz, double cx, double cy,
                                               double cz, double radius_sq, int edit_mode) {
    const int64_t bw = b.x1 - b.x0 + 1, bh = b.y1 - b.y0 + 1, bd = b.z1 - b.z0 + 1;
    const int64_t total = bw * bh * bd;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t x = b.x0 + i % bw, r = i / bw, y = b.y0 + r % bh, z = b.z0 + r / bh;
        const int64_t at = (z * h + y) * w + x;
        const double dx = (double)x * sx - cx, dy = (double)y * sy - cy, dz = (double)z * sz - cz;
        const double dist_sq = (dx * dx + dy * dy) + dz * dz;
        if (!(dist_sq <= radius_sq)) continue;
        if (edit_mode == 1) {
            if (out[at] > 0) out[at] = 0;
        } else if (edit_mode == 0) {
            if (orig) {
                const uint8_t o = orig[at];
                if (o > 0) out[at] = o;
            } else {
                out[at] = 255;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_polygon2mask(uint8_t *__restrict__ out, int64_t w, int64_t h,
                                                      const double *__restrict__ pts, int64_t n, uint64_t min_x,
                                                      uint64_t max_x, uint64_t min_y, uint64_t max_y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += stride) {
        const int64_t r = i / h, c = i % h;
        bool inside = false;
        if ((uint64_t)r >= min_x && (uint64_t)r <= max_x && (uint64_t)c >= min_y && (uint64_t)c <= max_y) {
            const double px = (double)r, py = (double)c;
            int64_t j = n - 1;
            for (int64_t k = 0; k < n; k++) {
                const double xi = pts[2 * k], yi = pts[2 * k + 1], xj = pts[2 * j], yj = pts[2 * j + 1];
                if (((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi) + xi)) inside = !inside;
                j = k;
            }
        }
        out[i] = inside ? 1 : 0;
    }
}

// Histogram of the labels.  Up to LBINS labels are counted in a per-workgroup LDS histogram (the wave's most common
// label -- the one its first lane holds, usually the background -- goes in with ONE add, the rest with LDS atomics) and
// flushed with one global atomic per non-empty bin; larger label spaces fall back to global atomics.  A first version
// with one global atomic per voxel-that-differs-from-lane-0 took 160 ms at 512^3 on salt-like labels.
constexpr int LBINS = 4096;
template <typename T>
__global__ __launch_bounds__(256) void k_count_hist(const T *__restrict__ labels, int64_t n, int64_t nreg,
                                                    uint32_t *__restrict__ counts, int *__restrict__ status) {
    __shared__ uint32_t s_h[LBINS];
    const bool lds = nreg < LBINS;
    if (lds) {
        for (int b = threadIdx.x; b < LBINS; b += 256) s_h[b] = 0u;
        __syncthreads();
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += stride) {
        const int64_t i = base + threadIdx.x;
        bool live = i < n;
        const int64_t l = live ? (int64_t)labels[i] : -1;
        if (live && (l < 0 || l > nreg)) {
            *status = 1; // the reference indexes counts[label]: out-of-bounds panic
            live = false;
        }
        const unsigned long long livem = __ballot(live);
        if (!livem) continue;
        const int64_t first = __shfl(l, __builtin_ctzll(livem), 64);
        const unsigned long long same = __ballot(live && l == first);
        const bool lead = (threadIdx.x & 63) == (unsigned)__builtin_ctzll(same);
        if (lds) {
            if (lead) atomicAdd(&s_h[first], (uint32_t)__popcll(same));
            if (live && l != first) atomicAdd(&s_h[l], 1u);
        } else {
            if (lead) atomicAdd(&counts[first], (uint32_t)__popcll(same));
            if (live && l != first) atomicAdd(&counts[l], 1u);
        }
    }
    if (lds) {
        __syncthreads();
        for (int b = threadIdx.x; b <= (int)nreg; b += 256)
            if (s_h[b]) atomicAdd(&counts[b], s_h[b]);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_count_gather(const T *__restrict__ labels, int64_t n, int64_t nreg,
                                                      const uint32_t *__restrict__ counts, uint32_t *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t l = (int64_t)labels[i];
        out[i] = (l >= 0 && l <= nreg) ? counts[l] : 0u;
    }
}

static inline unsigned grid_for(int64_t n, int per = 1) {
    const int64_t b = ivx::cdiv(n, (int64_t)256 * per);
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(b, 16384));
}

// Rust float -> integer casts (saturating, NaN -> 0)
static inline uint64_t f2usize(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)v;
}
static inline int64_t f2isize(double v) {
    if (v != v) return 0;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    if (v >= 9223372036854775807.0) return INT64_MAX;
    return (int64_t)v;
}

} // namespace

extern "C" int ivx_dev_mask_cut(uint8_t *out, int64_t dz, int64_t dy, int64_t dx, double sx, double sy, double sz,
                                double max_depth, const uint8_t *mask2d, int64_t mh, int64_t mw, const double *m,
                                const double *mv, int edit_mode, void *stream) {
    IVX_REQUIRE(dz >= 0 && dy >= 0 && dx >= 0 && mh >= 0 && mw >= 0, IVX_EINVAL, "mask_cut: negative shape");
    const int64_t n = dz * dy * dx;
    if (n == 0) return IVX_OK;
    Mat4 a, b;
    for (int i = 0; i < 16; i++) {
        a.m[i] = m[i];
        b.m[i] = mv[i];
    }
    const int vec = (((uintptr_t)out) & 15) == 0 && n >= 16;
    hipLaunchKernelGGL(k_mask_cut, dim3(grid_for(n, 16)), dim3(256), 0, ivx::S(stream), out, n, dy, dx, sx, sy, sz, max_depth,
                       mask2d, mh, mw, a, b, edit_mode, vec);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_brush_mask(uint8_t *out, const uint8_t *orig, int64_t dz, int64_t dy, int64_t dx,
                                  const double spacing[3], const double center[3], double radius, int edit_mode,
                                  void *stream) {
    IVX_REQUIRE(dz >= 0 && dy >= 0 && dx >= 0, IVX_EINVAL, "brush_mask: negative shape");
    if (dz == 0 || dy == 0 || dx == 0) return IVX_OK;
    if (edit_mode != 0 && edit_mode != 1) return IVX_OK; // the reference touches nothing in any other mode
    const double sx = spacing[0], sy = spacing[1], sz = spacing[2], cx = center[0], cy = center[1], cz = center[2];
    const uint64_t min_x = f2usize(std::fmax(std::floor((cx - radius) / sx), 0.0));
    const uint64_t max_x = f2usize(std::fmin(std::fmax(std::ceil((cx + radius) / sx), 0.0), (double)(dx - 1)));
    const uint64_t min_y = f2usize(std::fmax(std::floor((cy - radius) / sy), 0.0));
    const uint64_t max_y = f2usize(std::fmin(std::fmax(std::ceil((cy + radius) / sy), 0.0), (double)(dy - 1)));
    const uint64_t min_z = f2usize(std::fmax(std::floor((cz - radius) / sz), 0.0));
    const uint64_t max_z = f2usize(std::fmin(std::fmax(std::ceil((cz + radius) / sz), 0.0), (double)(dz - 1)));
    if (min_x > max_x || min_y > max_y || min_z > max_z) return IVX_OK; // empty box (also: min beyond the volume)
    Box b = {(int64_t)min_x, (int64_t)max_x, (int64_t)min_y, (int64_t)max_y, (int64_t)min_z, (int64_t)max_z};
    const int64_t total = (b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1) * (b.z1 - b.z0 + 1);
    hipLaunchKernelGGL(k_brush, dim3(grid_for(total)), dim3(256), 0, ivx::S(stream), out, orig, dy, dx, b, sx, sy, sz, cx, cy,
                       cz, radius * radius, edit_mode);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_polygon2mask(int64_t w, int64_t h, const double *points_dev, const double *points_host, int64_t npts,
                                    uint8_t *out, void *stream) {
    IVX_REQUIRE(w >= 0 && h >= 0 && npts >= 0, IVX_EINVAL, "polygon2mask: negative size");
    if (w == 0 || h == 0) return IVX_OK;
    hipStream_t st = ivx::S(stream);
    if (npts == 0) {
        IVX_HIP(hipMemsetAsync(out, 0, (size_t)(w * h), st));
        return IVX_OK;
    }
    double min_px = 1.7976931348623157e308, max_px = -1.7976931348623157e308, min_py = min_px, max_py = max_px;
    for (int64_t i = 0; i < npts; i++) {
        const double x = points_host[2 * i], y = points_host[2 * i + 1];
        if (x < min_px) min_px = x;
        if (x > max_px) max_px = x;
        if (y < min_py) min_py = y;
        if (y > max_py) max_py = y;
    }
    auto lo = [](double v, int64_t lim) {
        int64_t a = f2isize(std::floor(v));
        a = a == INT64_MIN ? a : a - 1;
        const uint64_t u = (uint64_t)(a > 0 ? a : 0);
        return u > (uint64_t)lim ? (uint64_t)lim : u;
    };
    auto hi = [](double v, int64_t lim) {
        int64_t a = f2isize(std::ceil(v));
        a = a == INT64_MAX ? a : a + 1;
        const uint64_t u = (uint64_t)(a > 0 ? a : 0);
        return u > (uint64_t)lim ? (uint64_t)lim : u;
    };
    hipLaunchKernelGGL(k_polygon2mask, dim3(grid_for(w * h)), dim3(256), 0, st, out, w, h, points_dev, npts, lo(min_px, w),
                       hi(max_px, w), lo(min_py, h), hi(max_py, h));
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_count_regions(int ldtype, const void *labels, int64_t n, int64_t number_regions, uint32_t *counts,
                                     uint32_t *out, int *status, void *stream) {
    IVX_REQUIRE(ldtype == IVX_I16 || ldtype == IVX_I32 || ldtype == IVX_I64, IVX_EINVAL,
                "count_regions: labels must be int16, int32 or int64");
    IVX_REQUIRE(n >= 0 && number_regions >= 0, IVX_EINVAL, "count_regions: negative size");
    hipStream_t st = ivx::S(stream);
    IVX_HIP(hipMemsetAsync(counts, 0, ((size_t)number_regions + 1) * 4, st));
    IVX_HIP(hipMemsetAsync(status, 0, 4, st));
    if (n == 0) return IVX_OK;
#define IVX_CR(T)                                                                                                    \
    hipLaunchKernelGGL((k_count_hist<T>), dim3(std::min(grid_for(n, 4), 2048u)), dim3(256), 0, st, (const T *)labels, n, number_regions, \
                       counts, status);                                                                              \
    IVX_LAUNCH_CHECK();                                                                                              \
    hipLaunchKernelGGL((k_count_gather<T>), dim3(grid_for(n, 4)), dim3(256), 0, st, (const T *)labels, n,             \
                       number_regions, counts, out);                                                                 \
    IVX_LAUNCH_CHECK();
    if (ldtype == IVX_I16) {
        IVX_CR(int16_t)
    } else if (ldtype == IVX_I32) {
        IVX_CR(int32_t)
    } else {
        IVX_CR(int64_t)
    }
#undef IVX_CR
    return IVX_OK;
}

// ---- host forms (C-contiguous arrays, as PyO3's as_array views of the reference's callers) -------------------------------
extern "C" int ivx_mask_cut(uint8_t *out, const int64_t shape[3], const int64_t ost[3], double sx, double sy, double sz,
                            double max_depth, const uint8_t *mask2d, int64_t mh, int64_t mw, const int64_t mst[2],
                            const double *m, const double *mv, int edit_mode) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(shape[0] >= 0 && shape[1] >= 0 && shape[2] >= 0 && mh >= 0 && mw >= 0, IVX_EINVAL, "mask_cut: negative shape");
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    if (n == 0) return IVX_OK;
    void *d_out, *d_m;
    int rc;
    if ((rc = ws_get(WS_OUT, n, &d_out))) return rc;
    if ((rc = ws_get(WS_AUX0, (size_t)mh * mw + 16, &d_m))) return rc;
    if ((rc = upload_strided(d_out, out, shape, ost, 1, WS_OUT))) return rc;
    const int64_t ms3[3] = {1, mh, mw}, mst3[3] = {0, mst[0], mst[1]};
    if (mh * mw && (rc = upload_strided(d_m, mask2d, ms3, mst3, 1, WS_AUX0))) return rc;
    if ((rc = ivx_dev_mask_cut((uint8_t *)d_out, shape[0], shape[1], shape[2], sx, sy, sz, max_depth, (const uint8_t *)d_m, mh,
                               mw, m, mv, edit_mode, nullptr)))
        return rc;
    IVX_HIP(hipDeviceSynchronize());
    return download_strided(out, shape, ost, d_out, 1, WS_OUT);
}

extern "C" int ivx_brush_mask(uint8_t *out, const int64_t shape[3], const int64_t ost[3], const uint8_t *orig,
                              const int64_t orig_st[3], const double spacing[3], const double center[3], double radius,
                              int edit_mode) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(shape[0] >= 0 && shape[1] >= 0 && shape[2] >= 0, IVX_EINVAL, "brush_mask: negative shape");
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    if (n == 0) return IVX_OK;
    void *d_out, *d_orig = nullptr;
    int rc;
    if ((rc = ws_get(WS_OUT, n, &d_out))) return rc;
    if ((rc = upload_strided(d_out, out, shape, ost, 1, WS_OUT))) return rc;
    if (orig) {
        if ((rc = ws_get(WS_IN, n, &d_orig))) return rc;
        if ((rc = upload_strided(d_orig, orig, shape, orig_st, 1, WS_IN))) return rc;
    }
    if ((rc = ivx_dev_brush_mask((uint8_t *)d_out, (const uint8_t *)d_orig, shape[0], shape[1], shape[2], spacing, center, radius,
                                 edit_mode, nullptr)))
        return rc;
    IVX_HIP(hipDeviceSynchronize());
    return download_strided(out, shape, ost, d_out, 1, WS_OUT);
}

extern "C" int ivx_polygon2mask(int64_t w, int64_t h, const double *points, int64_t npts, uint8_t *out) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(w >= 0 && h >= 0 && npts >= 0, IVX_EINVAL, "polygon2mask: negative size");
    if (w == 0 || h == 0) return IVX_OK;
    void *d_out, *d_p;
    int rc;
    if ((rc = ws_get(WS_OUT, (size_t)(w * h), &d_out))) return rc;
    if ((rc = ws_get(WS_SMALL, (size_t)npts * 16 + 16, &d_p))) return rc;
    if (npts) IVX_HIP(hipMemcpy(d_p, points, (size_t)npts * 16, hipMemcpyHostToDevice));
    if ((rc = ivx_dev_polygon2mask(w, h, (const double *)d_p, points, npts, (uint8_t *)d_out, nullptr))) return rc;
    IVX_HIP(hipMemcpy(out, d_out, (size_t)(w * h), hipMemcpyDeviceToHost));
    return IVX_OK;
}

extern "C" int ivx_count_regions(int ldtype, const void *labels, const int64_t shape[3], const int64_t lst[3],
                                 int64_t number_regions, uint32_t *out) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(ldtype == IVX_I16 || ldtype == IVX_I32 || ldtype == IVX_I64, IVX_EINVAL,
                "count_regions: labels must be int16, int32 or int64");
    IVX_REQUIRE(shape[0] >= 0 && shape[1] >= 0 && shape[2] >= 0 && number_regions >= 0, IVX_EINVAL, "count_regions: negative size");
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    if (n == 0) return IVX_OK;
    const size_t isz = ldtype == IVX_I16 ? 2 : (ldtype == IVX_I32 ? 4 : 8);
    void *d_l, *d_o, *d_c, *d_s;
    int rc;
    if ((rc = ws_get(WS_IN, n * isz, &d_l))) return rc;
    if ((rc = ws_get(WS_OUT, n * 4, &d_o))) return rc;
    if ((rc = ws_get(WS_AUX0, ((size_t)number_regions + 1) * 4, &d_c))) return rc;
    if ((rc = ws_get(WS_SMALL, 64, &d_s))) return rc;
    if ((rc = upload_strided(d_l, labels, shape, lst, isz, WS_IN))) return rc;
    if ((rc = ivx_dev_count_regions(ldtype, d_l, (int64_t)n, number_regions, (uint32_t *)d_c, (uint32_t *)d_o, (int *)d_s,
                                    nullptr)))
        return rc;
    int bad = 0;
    IVX_HIP(hipMemcpy(&bad, d_s, 4, hipMemcpyDeviceToHost));
    IVX_REQUIRE(!bad, IVX_ERANGE, "count_regions: a label lies outside [0, number_regions] (the reference panics on counts[label])");
    IVX_HIP(hipMemcpy(out, d_o, n * 4, hipMemcpyDeviceToHost));
    return IVX_OK;
}
