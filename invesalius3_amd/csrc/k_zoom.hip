// k_zoom.hip -- the quality-preset resample of SurfaceManager.AddNewActor on the GPU.
//
// Replaces  imagedata_utils.resize_image_array(image, 1 / imagedata_resolution, True)  (invesalius/data/imagedata_utils.py:
// 121-130, called for image AND mask at invesalius/data/surface.py:1350-1353 for the Low / Medium presets), i.e.
//     scipy.ndimage.zoom(image, factor, image.dtype, order=2)          (mode="constant", cval=0, prefilter, grid_mode=False)
// scipy's routine (python: ndimage/_interpolation.py zoom / spline_filter; C: ni_splines.c, ni_interpolation.c
// NI_ZoomShift) restated:
//   1. quadratic B-spline prefilter in float64, axis 0, 1, 2 in turn: gain (1-z)(1-1/z), z = sqrt(8)-3 as scipy's decimal literal; causal start with
//      the exact mirror sum  c0 = (c0 + z^(n-1) c[n-1] + sum z^i (c[i] + z^(n-1) c[n-1-i])) / (1 - z^(2n-2));
//      c[i] += z c[i-1];  c[n-1] = (z c[n-2] + c[n-1]) z / (z z - 1);  c[i] = z (c[i+1] - c[i]);
//   2. output voxel k -> input coordinate k * (n_in - 1) / (n_out - 1) per axis; 3 taps from floor(x + 0.5) - 1, mirrored at
//      the ends; weights with d = x - floor(x + 0.5):  w1 = 0.75 - d d,  w0 = 0.5 (0.5 - d)^2,  w2 = 1 - w0 - w1
//      (pinned by probing scipy with impulses, tools/probe_zoom.py); 27 products coeff * w0 * w1 * w2 summed in raster order;
//   3. integer outputs: t > 0 ? t + 0.5 : t - 0.5, clamped to the type, truncated.
// One lane per line for the recursions (they are serial along the line), one lane per output voxel for the gather.
// Compiled with -ffp-contract=off like everything else here: no FMA where scipy's C has none.
// Parity: equal to live scipy on every volume of tests/test_gpu_zoom.py, binary masks included (whose interpolated values
// sit exactly on rounding ties in exact arithmetic, so the float64 coefficients have to be scipy's to the last bit: the
// prefilter restated here was checked bit for bit against scipy.ndimage.spline_filter1d, tools/probe_zoom.py).
#include <math.h>

#include <vector>

#include "ivx_internal.h"

namespace {
using namespace ivx;

template <typename T>
__global__ __launch_bounds__(256) void k_zoom_widen(const T *__restrict__ in, double *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

// lines along one axis: `nlines` = product of the other two extents; line l starts at (l / inner) * outer_stride + (l % inner)
__global__ __launch_bounds__(256) void k_zoom_prefilter(double *__restrict__ c, int64_t nlines, int64_t inner, int64_t outer_stride,
                                                        int64_t stride, int64_t n, double z, double gain, double z_n_1) {
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= nlines) return;
    double *p = c + (l / inner) * outer_stride + (l % inner);
    for (int64_t i = 0; i < n; i++) p[i * stride] *= gain;
    if (n < 2) return; // scipy filters only lines longer than one sample
    double c0 = p[0] + z_n_1 * p[(n - 1) * stride];
    double z_i = z;
    for (int64_t i = 1; i < n - 1; i++) {
        c0 += z_i * (p[i * stride] + z_n_1 * p[(n - 1 - i) * stride]);
        z_i *= z;
    }
    c0 /= 1 - z_n_1 * z_n_1;
    p[0] = c0;
    double prev = c0;
    for (int64_t i = 1; i < n; i++) {
        prev = p[i * stride] + z * prev;
        p[i * stride] = prev;
    }
    double last = (z * p[(n - 2) * stride] + p[(n - 1) * stride]) * z / (z * z - 1);
    p[(n - 1) * stride] = last;
    for (int64_t i = n - 2; i >= 0; i--) {
        last = z * (last - p[i * stride]);
        p[i * stride] = last;
    }
}

struct ZoomTab { // per output index of one axis
    int32_t idx[3];
    int32_t zero;
    double w[3];
};

template <typename T>
__global__ __launch_bounds__(256) void k_zoom_gather(const double *__restrict__ f, int64_t iy, int64_t ix, const ZoomTab *__restrict__ tz,
                                                     const ZoomTab *__restrict__ ty, const ZoomTab *__restrict__ tx, int64_t oz,
                                                     int64_t oy, int64_t ox, double lo, double hi, T *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= oz * oy * ox) return;
    const int64_t x = i % ox, r = i / ox, y = r % oy, zc = r / oy;
    const ZoomTab a = tz[zc], b = ty[y], c = tx[x];
    double t = 0.0;
    if (!(a.zero | b.zero | c.zero)) {
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    double coeff = f[((int64_t)a.idx[p] * iy + b.idx[q]) * ix + c.idx[s]];
                    coeff *= a.w[p];
                    coeff *= b.w[q];
                    coeff *= c.w[s];
                    t += coeff;
                }
    }
    t = t > 0 ? t + 0.5 : t - 0.5;
    t = t > hi ? hi : (t < lo ? lo : t);
    out[i] = (T)t;
}

static void make_tab(int64_t n_in, int64_t n_out, std::vector<ZoomTab> &tab) {
    tab.resize((size_t)n_out);
    const double zoom = n_out > 1 ? (double)(n_in - 1) / (double)(n_out - 1) : 1.0;
    for (int64_t k = 0; k < n_out; k++) {
        ZoomTab &t = tab[(size_t)k];
        double cc = (double)k;
        cc *= zoom;
        t.zero = 0;
        if (cc < 0 || cc > (double)(n_in - 1)) { // NI_EXTEND_CONSTANT: outside -> cval
            t.zero = 1;
            t.idx[0] = t.idx[1] = t.idx[2] = 0;
            t.w[0] = t.w[1] = t.w[2] = 0.0;
            continue;
        }
        const int64_t start = (int64_t)floor(cc + 0.5) - 1;
        for (int h = 0; h < 3; h++) {
            int64_t idx = start + h;
            if (n_in <= 1) idx = 0;
            else {
                const int64_t s2 = 2 * n_in - 2;
                if (idx < 0) {
                    idx = s2 * (int64_t)(-idx / s2) + idx;
                    idx = idx <= 1 - n_in ? idx + s2 : -idx;
                } else if (idx >= n_in) {
                    idx -= s2 * (int64_t)(idx / s2);
                    if (idx >= n_in) idx = s2 - idx;
                }
            }
            t.idx[h] = (int32_t)idx;
        }
        const double d = cc - floor(cc + 0.5);
        t.w[1] = 0.75 - d * d;
        const double y = 0.5 - d;
        t.w[0] = 0.5 * y * y;
        t.w[2] = 1.0 - t.w[0] - t.w[1];
    }
}

template <typename T>
static int zoom_run(const T *d_in, const int64_t ish[3], T *d_out, const int64_t osh[3], double *d_f, ZoomTab *d_tab, hipStream_t st) {
    const int64_t n = ish[0] * ish[1] * ish[2], no = osh[0] * osh[1] * osh[2];
    hipLaunchKernelGGL(k_zoom_widen<T>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, d_in, d_f, n);
    IVX_LAUNCH_CHECK();
    // ni_splines.c get_filter_poles writes the pole as a decimal literal: it differs from sqrt(8.0) - 3.0 evaluated in double
    // by several ulp, and ties of a 0 / 255 mask (exactly x.5 in exact arithmetic) round by those ulp
    const double z = -0.171572875253809902396622551580603843, gain = (1.0 - z) * (1.0 - 1.0 / z);
    // axis 0: lines indexed by (y, x); axis 1: by (z, x); axis 2: by (z, y)
    const int64_t strides[3] = {ish[1] * ish[2], ish[2], 1};
    for (int ax = 0; ax < 3; ax++) {
        const int64_t len = ish[ax];
        int64_t nlines, inner, outer_stride;
        if (ax == 0) { nlines = ish[1] * ish[2]; inner = nlines; outer_stride = 0; }
        else if (ax == 1) { nlines = ish[0] * ish[2]; inner = ish[2]; outer_stride = strides[0]; }
        else { nlines = ish[0] * ish[1]; inner = 1; outer_stride = ish[2]; }
        if (len > 1) { // scipy: `if (len > 1) apply_filter(...)`: a line of one sample is not even scaled
            hipLaunchKernelGGL(k_zoom_prefilter, dim3((unsigned)cdiv(nlines, 256)), dim3(256), 0, st, d_f, nlines, inner, outer_stride,
                               strides[ax], len, z, gain, pow(z, (double)(len - 1)));
            IVX_LAUNCH_CHECK();
        }
    }
    std::vector<ZoomTab> tabs[3];
    size_t off = 0;
    ZoomTab *d_t[3];
    for (int ax = 0; ax < 3; ax++) {
        make_tab(ish[ax], osh[ax], tabs[ax]);
        d_t[ax] = d_tab + off;
        IVX_HIP(hipMemcpyAsync(d_t[ax], tabs[ax].data(), tabs[ax].size() * sizeof(ZoomTab), hipMemcpyHostToDevice, st));
        off += tabs[ax].size();
    }
    IVX_HIP(hipStreamSynchronize(st)); // the tables live on this stack frame
    const double lo = sizeof(T) == 2 ? -32768.0 : 0.0, hi = sizeof(T) == 2 ? 32767.0 : 255.0;
    hipLaunchKernelGGL(k_zoom_gather<T>, dim3((unsigned)cdiv(no, 256)), dim3(256), 0, st, d_f, ish[1], ish[2], d_t[0], d_t[1], d_t[2],
                       osh[0], osh[1], osh[2], lo, hi, d_out);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}
} // namespace

// device form: `scratch` holds the float64 coefficient volume and the three index / weight tables
extern "C" int ivx_zoom_scratch_bytes(const int64_t ishape[3], const int64_t oshape[3], size_t *nbytes) {
    *nbytes = (size_t)(ishape[0] * ishape[1] * ishape[2]) * 8 + 256 + (size_t)(oshape[0] + oshape[1] + oshape[2]) * sizeof(ZoomTab);
    return IVX_OK;
}

extern "C" int ivx_dev_zoom_order2(int dtype, const void *in, const int64_t ishape[3], void *out, const int64_t oshape[3],
                                   void *scratch, void *stream) {
    IVX_REQUIRE(dtype == IVX_I16 || dtype == IVX_U8, IVX_EINVAL, "zoom: int16 or uint8 volumes");
    for (int a = 0; a < 3; a++) IVX_REQUIRE(ishape[a] > 0 && oshape[a] > 0, IVX_EINVAL, "zoom: empty shape");
    IVX_REQUIRE(in && out && scratch, IVX_EINVAL, "zoom: null buffer");
    const int64_t n = ishape[0] * ishape[1] * ishape[2];
    double *d_f = (double *)scratch;
    ZoomTab *d_tab = (ZoomTab *)((char *)scratch + (((size_t)n * 8 + 255) & ~(size_t)255));
    if (dtype == IVX_I16) return zoom_run<int16_t>((const int16_t *)in, ishape, (int16_t *)out, oshape, d_f, d_tab, S(stream));
    return zoom_run<uint8_t>((const uint8_t *)in, ishape, (uint8_t *)out, oshape, d_f, d_tab, S(stream));
}

extern "C" int ivx_zoom_order2(int dtype, const void *in, const int64_t ishape[3], void *out, const int64_t oshape[3]) {
    HostCallGuard guard;
    IVX_REQUIRE(dtype == IVX_I16 || dtype == IVX_U8, IVX_EINVAL, "zoom: int16 or uint8 volumes");
    const size_t isz = dtype == IVX_I16 ? 2 : 1;
    const size_t n = (size_t)(ishape[0] * ishape[1] * ishape[2]), no = (size_t)(oshape[0] * oshape[1] * oshape[2]);
    if (n == 0 || no == 0) return IVX_OK;
    void *d_in = nullptr, *d_out = nullptr, *d_s = nullptr;
    size_t sb = 0;
    int rc = ivx_zoom_scratch_bytes(ishape, oshape, &sb);
    if ((rc = ws_get(WS_IN, n * isz, &d_in)) != IVX_OK) return rc;
    if ((rc = ws_get(WS_OUT, no * isz, &d_out)) != IVX_OK) return rc;
    if ((rc = ws_get(WS_AUX0, sb, &d_s)) != IVX_OK) return rc;
    IVX_HIP(hipMemcpy(d_in, in, n * isz, hipMemcpyHostToDevice));
    rc = ivx_dev_zoom_order2(dtype, d_in, ishape, d_out, oshape, d_s, nullptr);
    if (rc != IVX_OK) return rc;
    IVX_HIP(hipMemcpy(out, d_out, no * isz, hipMemcpyDeviceToHost));
    return IVX_OK;
}
