// k_transform.hip -- resample a slab through a 4x4 view matrix: the step in front of the projections when the volume
// is re-oriented ("next" row 3).
//
// Reference semantics:
//   apply_view_matrix_transform   invesalius_rs/src/transforms_py.rs:12-49, 95-147
//   coord_transform                invesalius_rs/src/transforms.rs:9-55
//   get_value (single wrap-around) / trilinear / tricubic / Lanczos-4   invesalius_rs/src/interpolation.rs:6-188
// All arithmetic in double in the reference's evaluation order (this file is built with -ffp-contract=off; double
// add / mul / div / floor are IEEE on gfx950), so nearest, trilinear and tricubic are bit-exact.  Lanczos calls sin():
// device sin and glibc sin may differ in the last ulp, which can move a truncating integer cast by one LSB.
//
// One lane per output voxel, x fastest: neighbouring lanes sample neighbouring source positions, so the gathers of a
// wave hit a compact footprint in L1/L2 (8 taps trilinear, 64 tricubic, 343 Lanczos).  Memory-bound on the source
// slab for nearest/trilinear, instruction-bound for the two high-order kernels.
#include <math.h>

#include "ivx_internal.h"

namespace {

struct TGeom {
    int64_t dz, dy, dx;    // source volume
    int64_t oz, oy, ox;    // output block
    int64_t n;             // first slice of the block along the orientation axis
    int orientation;       // 0 AXIAL, 1 CORONAL, 2 SAGITAL, else none
    int minterpol;
    double sx, sy, sz;
    double m[16];
    double cval;
};

template <typename T> __device__ __forceinline__ bool numcast_d(double v, double *out);
template <> __device__ __forceinline__ bool numcast_d<int16_t>(double v, double *out) {
    if (!(v > -32769.0 && v < 32768.0)) return false;
    *out = (double)(int16_t)v;
    return true;
}
template <> __device__ __forceinline__ bool numcast_d<uint8_t>(double v, double *out) {
    if (!(v > -1.0 && v < 256.0)) return false;
    *out = (double)(uint8_t)v;
    return true;
}
template <> __device__ __forceinline__ bool numcast_d<double>(double v, double *out) {
    *out = v;
    return true;
}

template <typename T>
__device__ __forceinline__ double tget(const T *__restrict__ v, const TGeom &g, int64_t x, int64_t y, int64_t z) {
    if (x < 0) x += g.dx; else if (x >= g.dx) x -= g.dx; // interpolation.rs:6-35: ONE wrap per axis
    if (y < 0) y += g.dy; else if (y >= g.dy) y -= g.dy;
    if (z < 0) z += g.dz; else if (z >= g.dz) z -= g.dz;
    return (double)v[(z * g.dy + y) * g.dx + x];
}

__device__ __forceinline__ double cubic1(const double p[4], double x) { // interpolation.rs:37-43
    return p[1] + 0.5 * x * (p[2] - p[0] + x * (2.0 * p[0] - 5.0 * p[1] + 4.0 * p[2] - p[3] + x * (3.0 * (p[1] - p[2]) + p[3] - p[0])));
}

__device__ __forceinline__ double lanczos_k(double x, int a) { // interpolation.rs:55-64
    const double PI = 3.14159265358979323846264338327950288;
    if (x == 0.0) return 1.0;
    if (-(double)a <= x && x < (double)a) {
        const double af = (double)a;
        return (af * sin(PI * x) * sin(PI * (x / af))) / (PI * PI * x * x);
    }
    return 0.0;
}

template <typename T>
__device__ double trilinear(const T *__restrict__ v, const TGeom &g, double x, double y, double z) {
    const int64_t x0 = (int64_t)floor(x), x1 = x0 + 1, y0 = (int64_t)floor(y), y1 = y0 + 1, z0 = (int64_t)floor(z), z1 = z0 + 1;
    const double xd = x - (double)x0, yd = y - (double)y0, zd = z - (double)z0;
    const double v000 = tget(v, g, x0, y0, z0), v100 = tget(v, g, x1, y0, z0), v010 = tget(v, g, x0, y1, z0),
                 v001 = tget(v, g, x0, y0, z1), v110 = tget(v, g, x1, y1, z0), v101 = tget(v, g, x1, y0, z1),
                 v011 = tget(v, g, x0, y1, z1), v111 = tget(v, g, x1, y1, z1);
    const double c00 = v000 * (1.0 - xd) + v100 * xd, c10 = v010 * (1.0 - xd) + v110 * xd;
    const double c01 = v001 * (1.0 - xd) + v101 * xd, c11 = v011 * (1.0 - xd) + v111 * xd;
    const double c0 = c00 * (1.0 - yd) + c10 * yd, c1 = c01 * (1.0 - yd) + c11 * yd;
    return c0 * (1.0 - zd) + c1 * zd;
}

template <typename T>
__device__ double tricubic(const T *__restrict__ v, const TGeom &g, double x, double y, double z) {
    const int64_t xi = (int64_t)floor(x), yi = (int64_t)floor(y), zi = (int64_t)floor(z);
    const double fy = y - (double)yi, fz = z - (double)zi;
    double r[4];
    for (int i = 0; i < 4; i++) { // p[i][j][k] = get(xi+i-1, yi+j-1, zi+k-1); bicubic(p[i], y-yi, z-zi)
        double a[4];
        for (int j = 0; j < 4; j++) {
            double p[4];
#pragma unroll
            for (int k = 0; k < 4; k++) p[k] = tget(v, g, xi + i - 1, yi + j - 1, zi + k - 1);
            a[j] = cubic1(p, fz);
        }
        r[i] = cubic1(a, fy);
    }
    return cubic1(r, x - (double)xi);
}

template <typename T>
__device__ double lanczos(const T *__restrict__ v, const TGeom &g, double x, double y, double z) {
    const int a = 4;
    const int64_t xd = (int64_t)floor(x), yd = (int64_t)floor(y), zd = (int64_t)floor(z);
    const int64_t xi = xd - a + 1, yi = yd - a + 1, zi = zd - a + 1;
    double kx[7], ky[7], kz[7];
#pragma unroll
    for (int q = 0; q < 7; q++) { // the reference re-evaluates these inside its loops; same values either way
        kx[q] = lanczos_k(x - (double)(xi + q), a);
        ky[q] = lanczos_k(y - (double)(yi + q), a);
        kz[q] = lanczos_k(z - (double)(zi + q), a);
    }
    double lz = 0.0;
    for (int m = 0; m < 7; m++) {
        double ly = 0.0;
        for (int nn = 0; nn < 7; nn++) {
            double lx = 0.0;
#pragma unroll
            for (int q = 0; q < 7; q++) lx += tget(v, g, xi + q, yi + nn, zi + m) * kx[q];
            ly += lx * ky[nn];
        }
        lz += ly * kz[m];
    }
    return lz;
}

template <typename T>
__global__ __launch_bounds__(256) void k_view_transform(const T *__restrict__ vol, TGeom g, T *__restrict__ out,
                                                        int *__restrict__ status) {
    const int64_t total = g.oz * g.oy * g.ox;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double dz = (double)g.dz, dy = (double)g.dy, dx = (double)g.dx;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t cx = i % g.ox, r = i / g.ox, cy = r % g.oy, cz = r / g.oy;
        int64_t z = cz, y = cy, x = cx;
        if (g.orientation == 0) z = g.n + cz;
        else if (g.orientation == 1) y = g.n + cy;
        else if (g.orientation == 2) x = g.n + cx;
        const double c0 = (double)z * g.sz, c1 = (double)y * g.sy, c2 = (double)x * g.sx;
        double nc[4];
#pragma unroll
        for (int q = 0; q < 4; q++) nc[q] = ((g.m[4 * q] * c0 + g.m[4 * q + 1] * c1) + g.m[4 * q + 2] * c2) + g.m[4 * q + 3] * 1.0;
        const double nz = (nc[0] / nc[3]) / g.sz, ny = (nc[1] / nc[3]) / g.sy, nx = (nc[2] / nc[3]) / g.sx;
        double v = g.cval;
        if (nz >= 0.0 && nz < dz - 1.0 && ny >= 0.0 && ny < dy - 1.0 && nx >= 0.0 && nx < dx - 1.0) {
            if (g.minterpol == 0) v = (double)vol[((int64_t)nz * g.dy + (int64_t)ny) * g.dx + (int64_t)nx];
            else {
                const double f = g.minterpol == 1 ? trilinear(vol, g, nx, ny, nz)
                                 : g.minterpol == 2 ? tricubic(vol, g, nx, ny, nz) : lanczos(vol, g, nx, ny, nz);
                double c;
                if (!numcast_d<T>(f, &c)) { atomicMin(status, IVX_EDOM); c = g.cval; }
                else if (g.minterpol != 1 && c < g.cval) c = g.cval; // transforms.rs:38-50
                v = c;
            }
        }
        out[i] = (T)v;
    }
}

template <typename T>
static int launch(const void *vol, const TGeom &g, void *out, int *status, hipStream_t st) {
    const int64_t total = g.oz * g.oy * g.ox;
    if (!total) return IVX_OK;
    const int64_t b = ivx::cdiv(total, 256);
    hipLaunchKernelGGL(k_view_transform<T>, dim3((unsigned)(b < 65536 ? b : 65536)), dim3(256), 0, st, (const T *)vol, g, (T *)out, status);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

} // namespace

extern "C" int ivx_dev_apply_view_matrix_transform(int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx,
                                                   const double spacing[3], const double m[16], int64_t n,
                                                   int orientation, int minterpol, double cval, void *out, int64_t oz,
                                                   int64_t oy, int64_t ox, int *status, void *stream) {
    TGeom g;
    g.dz = dz; g.dy = dy; g.dx = dx; g.oz = oz; g.oy = oy; g.ox = ox; g.n = n;
    g.orientation = orientation; g.minterpol = minterpol;
    g.sx = spacing[0]; g.sy = spacing[1]; g.sz = spacing[2];
    for (int i = 0; i < 16; i++) g.m[i] = m[i];
    g.cval = cval;
    hipStream_t st = ivx::S(stream);
    switch (dtype) {
    case IVX_I16: return launch<int16_t>(vol, g, out, status, st);
    case IVX_U8: return launch<uint8_t>(vol, g, out, status, st);
    case IVX_F64: return launch<double>(vol, g, out, status, st);
    }
    ivx::set_error("Invalid volume or output type");
    return IVX_EINVAL;
}

extern "C" int ivx_apply_view_matrix_transform(int dtype, const void *vol, const int64_t shape[3], const int64_t strides[3],
                                               const double spacing[3], const double m[16], int64_t n, int orientation,
                                               int minterpol, double cval, void *out, const int64_t oshape[3],
                                               const int64_t ostrides[3]) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    const size_t isz = dtype_size(dtype);
    IVX_REQUIRE(dtype == IVX_I16 || dtype == IVX_U8 || dtype == IVX_F64, IVX_EINVAL, "Invalid volume or output type");
    const size_t nv = (size_t)shape[0] * shape[1] * shape[2], no = (size_t)oshape[0] * oshape[1] * oshape[2];
    if (no == 0) return IVX_OK;
    IVX_REQUIRE(nv > 0, IVX_EINVAL, "apply_view_matrix_transform: empty volume");
    void *d_v, *d_o, *d_s;
    int rc;
    if ((rc = ws_get(WS_IN, nv * isz, &d_v))) return rc;
    if ((rc = ws_get(WS_OUT, no * isz, &d_o))) return rc;
    if ((rc = ws_get(WS_SMALL, 256, &d_s))) return rc;
    IVX_HIP(hipMemset(d_s, 0, 256));
    if ((rc = upload_strided(d_v, vol, shape, strides, isz, WS_IN))) return rc;
    if ((rc = ivx_dev_apply_view_matrix_transform(dtype, d_v, shape[0], shape[1], shape[2], spacing, m, n, orientation,
                                                  minterpol, cval, d_o, oshape[0], oshape[1], oshape[2], (int *)d_s, nullptr)))
        return rc;
    IVX_HIP(hipDeviceSynchronize());
    if ((rc = download_strided(out, oshape, ostrides, d_o, isz, WS_OUT))) return rc;
    int stt = 0;
    IVX_HIP(hipMemcpy(&stt, d_s, 4, hipMemcpyDeviceToHost));
    IVX_REQUIRE(stt == 0, IVX_EDOM, "NumCast failure: an interpolated value does not fit the volume dtype (the reference panics)");
    return IVX_OK;
}
