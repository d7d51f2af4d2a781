// k_rays.hip -- order-dependent slab projections: MIDA, LMIP, fast contour MIP.
//
// Reference semantics (bit-exact for the integer outputs; f32 state evaluated in the reference's operation order,
// this file is compiled with -ffp-contract=off so nothing is fused):
//   lmip                          invesalius_rs/src/mips.rs:7-86
//   get_opacity / mida_internal   invesalius_rs/src/mips.rs:88-168   (dtype pairs: mips_py.rs:161-202)
//   finite_difference / calc_fcm_intensity / fast_countour_mip_internal   invesalius_rs/src/mips.rs:171-279
//
// MI355X design (memory-bound: 2 B/voxel read once, one output pixel per ray; no MFMA):
//   axis 0 / 1  a ray is strided in memory but ADJACENT rays are adjacent in x: lane <-> x, every step of the
//               walk is one fully coalesced row segment per wave.
//   axis 2      a ray is contiguous (an x-row): a wave takes 64 consecutive rows, stages a 64-row x 128-byte
//               chunk in LDS with 16-B coalesced loads (8 lanes per row segment) and each lane then walks its own
//               row out of LDS (row pitch 136 B: no more than 2-way bank conflicts).  The chunk loop stops as soon
//               as every lane's ray has terminated (MIDA alpha >= 1, LMIP first fall after the threshold).
//   k_fcm_volume  one lane per voxel, x fastest; the six clamped neighbours come through L1/L2 (each row is
//               re-used by its two y- and two z-neighbours, so HBM still sees each voxel about once).
#include <math.h>

#include "ivx_internal.h"
#include "glibc_powf.h"

namespace {

// ---- NumCast f32 -> T (num-traits: None unless MIN-1 < x < MAX+1, then truncation) -----------------------
template <typename T> __device__ __forceinline__ bool numcast(float v, T *dst);
template <> __device__ __forceinline__ bool numcast<int16_t>(float v, int16_t *dst) {
    if (!(v > -32769.0f && v < 32768.0f)) return false;
    *dst = (int16_t)v;
    return true;
}
template <> __device__ __forceinline__ bool numcast<uint8_t>(float v, uint8_t *dst) {
    if (!(v > -1.0f && v < 256.0f)) return false;
    *dst = (uint8_t)v;
    return true;
}
template <> __device__ __forceinline__ bool numcast<double>(float v, double *dst) {
    *dst = (double)v;
    return true;
}

// ---- ray functors --------------------------------------------------------------------------------------
// Both functors are written branch-free (selects on an `active` predicate): a ray that has terminated, or a sample past
// the end of the ray, leaves the state untouched.  Per-sample branches cost more than the arithmetic they skip.
template <typename T> struct LmipRay { // mips.rs:24-37
    T maxv, tmin, tmax;
    bool start, first;
    __device__ __forceinline__ void init(T lo, T hi) { tmin = lo; tmax = hi; first = true; start = false; maxv = (T)0; }
    __device__ __forceinline__ bool step(T v, bool active) { // returns true when the ray finishes on this sample
        const bool inwin = v >= tmin && v <= tmax;
        // first sample: max_val = image[0], start = in-window(max_val); then the loop body sees val == max_val
        const T m0 = first ? v : maxv;
        const bool s0 = first ? inwin : start;
        const bool gt = v > m0, lt = v < m0;
        const bool fin = active && lt && s0;               // `else if val < max_val && start { break }`
        const bool upd = active && !fin;
        maxv = upd ? (gt ? v : m0) : maxv;
        start = upd ? (s0 || inwin) : start;
        first = first && !active;
        return fin;
    }
};

struct MidaRay { // mips.rs:136-163
    float fmax, alpha_p, colour_p, final_colour;
    float img_min, inv_range, wl, ww;
    __device__ __forceinline__ void init(float mn, float range, float wl_, float ww_) {
        fmax = alpha_p = colour_p = final_colour = 0.0f;
        img_min = mn;
        inv_range = 1.0f / range;
        wl = wl_;
        ww = ww_;
    }
    __device__ __forceinline__ bool step(float vl, bool active) {
        const float fpi = inv_range * (vl - img_min);
        const bool rise = fpi > fmax;
        const float dl = rise ? fpi - fmax : 0.0f;
        const float bt = 1.0f - dl;
        // get_opacity, mips.rs:88-100
        const float min_value = wl - (ww / 2.0f), max_value = wl + (ww / 2.0f);
        const float ramp = (vl - min_value) / (max_value - min_value);
        const float alpha = vl < min_value ? 0.0f : (vl > max_value ? 1.0f : ramp);
        const float colour = (bt * colour_p) + (1.0f - bt * alpha_p) * fpi * alpha;
        const float current_alpha = (bt * alpha_p) + (1.0f - bt * alpha_p) * alpha;
        fmax = (active && rise) ? fpi : fmax;
        colour_p = active ? colour : colour_p;
        alpha_p = active ? current_alpha : alpha_p;
        final_colour = active ? colour : final_colour;
        return active && current_alpha >= 1.0f;
    }
};

// ---- geometry of the three axes --------------------------------------------------------------------------
struct RayGeom {
    int64_t nr, nc, len; // output rows, cols, ray length
    int64_t sr, sc, sl;  // element strides
};
static RayGeom ray_geom(int axis, int64_t dz, int64_t dy, int64_t dx) {
    RayGeom g;
    if (axis == 0) { g.nr = dy; g.nc = dx; g.len = dz; g.sr = dx; g.sc = 1; g.sl = dy * dx; }
    else if (axis == 1) { g.nr = dz; g.nc = dx; g.len = dy; g.sr = dy * dx; g.sc = 1; g.sl = dx; }
    else { g.nr = dz; g.nc = dy; g.len = dx; g.sr = dy * dx; g.sc = dx; g.sl = 1; }
    return g;
}

// MODE 0 = lmip (out T), 1 = mida (out U)
template <typename T, typename U, int MODE>
__device__ __forceinline__ void finish(const LmipRay<T> &lr, const MidaRay &mr, float range, U *out, int *status) {
    if (MODE == 0) {
        *out = (U)lr.maxv;
    } else {
        U v = (U)0;
        if (!numcast<U>(range * mr.final_colour + mr.img_min, &v)) atomicMin(status, IVX_EDOM);
        *out = v;
    }
}

// ---- axis 0 / 1: lane <-> x -------------------------------------------------------------------------------
template <typename T, typename U, int MODE>
__global__ __launch_bounds__(256) void k_rays_strided(const T *__restrict__ vol, RayGeom g, double p0, double p1,
                                                      const float *__restrict__ minmax, U *__restrict__ out,
                                                      int *__restrict__ status, const double *__restrict__ state_in,
                                                      double *__restrict__ state_out) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= g.nr * g.nc) return;
    const int64_t r = pix / g.nc, c = pix - r * g.nc;
    const T *p = vol + r * g.sr + c * g.sc;
    LmipRay<T> lr;
    MidaRay mr;
    float range = 0.0f;
    if (MODE == 0) lr.init((T)p0, (T)p1);
    else {
        range = minmax[1] - minmax[0];
        mr.init(minmax[0], range, (float)p0, (float)p1);
    }
    bool done = false;
    // A ray that continues from the previous Z-slab (sharded volumes, rays along the sharding axis) resumes from the
    // state that slab left: five doubles per pixel (every state variable is exactly representable in one).
    if (state_in) {
        const double *si = state_in + pix * 5;
        if (MODE == 0) {
            lr.maxv = (T)si[0];
            lr.start = si[1] != 0.0;
            lr.first = si[2] != 0.0;
        } else {
            mr.fmax = (float)si[0];
            mr.alpha_p = (float)si[1];
            mr.colour_p = (float)si[2];
            mr.final_colour = (float)si[3];
        }
        done = si[4] != 0.0;
    }
    // software pipeline: the 8 loads of the NEXT block are in flight while the current block's 8 samples are
    // composited (the walk is a serial dependence chain, the loads are not); one wave vote per block for early exit
    constexpr int B = 16; // samples per block: up to 2 x 16 loads in flight per lane
    T cur[B], nxt[B];
#pragma unroll
    for (int k = 0; k < B; k++) cur[k] = (k < g.len) ? p[k * g.sl] : (T)0;
    int64_t l0 = 0;
    // whole blocks with a whole block behind them: no per-sample bounds tests, the next block's samples come off one running
    // pointer (the 64-bit index arithmetic and compares were ~10 of the ~55 vector instructions a sample cost)
    const T *q = p + (int64_t)B * g.sl;
    for (; l0 + 2 * B <= g.len; l0 += B) {
#pragma unroll
        for (int k = 0; k < B; k++) {
            nxt[k] = *q;
            q += g.sl;
        }
#pragma unroll
        for (int k = 0; k < B; k++) done |= MODE == 0 ? lr.step(cur[k], !done) : mr.step((float)cur[k], !done);
        if (__all(done)) break; // the whole wave's rays have terminated
#pragma unroll
        for (int k = 0; k < B; k++) cur[k] = nxt[k];
    }
    if (!__all(done))
        for (; l0 < g.len; l0 += B) { // the last one or two blocks: guarded
#pragma unroll
            for (int k = 0; k < B; k++) nxt[k] = (l0 + B + k < g.len) ? p[(l0 + B + k) * g.sl] : (T)0;
#pragma unroll
            for (int k = 0; k < B; k++) {
                const bool active = !done && l0 + k < g.len;
                done |= MODE == 0 ? lr.step(cur[k], active) : mr.step((float)cur[k], active);
            }
            if (__all(done)) break;
#pragma unroll
            for (int k = 0; k < B; k++) cur[k] = nxt[k];
        }
    if (state_out) { // more slabs follow: hand the ray over instead of finishing it
        double *so = state_out + pix * 5;
        if (MODE == 0) {
            so[0] = (double)lr.maxv;
            so[1] = lr.start ? 1.0 : 0.0;
            so[2] = lr.first ? 1.0 : 0.0;
            so[3] = 0.0;
        } else {
            so[0] = (double)mr.fmax;
            so[1] = (double)mr.alpha_p;
            so[2] = (double)mr.colour_p;
            so[3] = (double)mr.final_colour;
        }
        so[4] = done ? 1.0 : 0.0;
        return;
    }
    finish<T, U, MODE>(lr, mr, range, out + pix, status);
}

// ---- axis 2: wave <-> 64 rows, LDS-staged chunks -------------------------------------------------------------
constexpr int PITCH = 136; // bytes per staged row (128 data + 8 pad)
template <typename T, typename U, int MODE>
__global__ __launch_bounds__(256) void k_rays_rows(const T *__restrict__ vol, int64_t nrays, int64_t len, double p0,
                                                   double p1, const float *__restrict__ minmax, U *__restrict__ out,
                                                   int *__restrict__ status) {
    constexpr int CH = 128 / sizeof(T); // elements per staged row chunk
    __shared__ __attribute__((aligned(16))) unsigned char s_tile[4][64 * PITCH];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ray0 = ((int64_t)blockIdx.x * 4 + wv) * 64;
    if (ray0 >= nrays) return; // whole wave out of range (no block-level barriers below)
    const int64_t ray = ray0 + lane;
    const bool live = ray < nrays;
    unsigned char *tile = s_tile[wv];
    LmipRay<T> lr;
    MidaRay mr;
    float range = 0.0f;
    if (MODE == 0) lr.init((T)p0, (T)p1);
    else {
        range = minmax[1] - minmax[0];
        mr.init(minmax[0], range, (float)p0, (float)p1);
    }
    bool done = !live;
    const bool vec_ok = ((len * sizeof(T)) % 16 == 0) && (((uintptr_t)vol & 15) == 0);
    for (int64_t c0 = 0; c0 < len; c0 += CH) {
        // stage rows ray0..ray0+63, elements [c0, c0+CH): 8 lanes x 16 B per row, 8 rows per pass
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int row = i * 8 + (lane >> 3), seg = lane & 7;
            const int64_t rr = ray0 + row;
            const int64_t e0 = c0 + seg * (16 / sizeof(T));
            uint4 q = make_uint4(0, 0, 0, 0);
            if (rr < nrays && e0 < len) {
                const T *src = vol + rr * len + e0;
                if (vec_ok && e0 + (int64_t)(16 / sizeof(T)) <= len) q = *reinterpret_cast<const uint4 *>(src);
                else {
                    T tmp[16 / sizeof(T)];
#pragma unroll
                    for (int e = 0; e < (int)(16 / sizeof(T)); e++) tmp[e] = (e0 + e < len) ? src[e] : (T)0;
                    q = *reinterpret_cast<uint4 *>(tmp);
                }
            }
            // 8-byte LDS stores (row pitch 136 is 8-aligned, not 16)
            unsigned long long *d = reinterpret_cast<unsigned long long *>(tile + row * PITCH + seg * 16);
            d[0] = ((unsigned long long)q.y << 32) | q.x;
            d[1] = ((unsigned long long)q.w << 32) | q.z;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const T *mine = reinterpret_cast<const T *>(tile + lane * PITCH);
        const int n = (int)((len - c0) < CH ? (len - c0) : CH);
        // eight bytes of the lane's row per LDS read (the pitch keeps every row 8-byte aligned), then the samples one by one
        constexpr int PER = 8 / (int)sizeof(T);
        int e = 0;
#pragma unroll 2
        for (; e + PER <= n; e += PER) {
            const unsigned long long w = *reinterpret_cast<const unsigned long long *>(mine + e);
            T v[PER];
            memcpy(v, &w, 8);
#pragma unroll
            for (int u = 0; u < PER; u++) done |= MODE == 0 ? lr.step(v[u], !done) : mr.step((float)v[u], !done);
        }
        for (; e < n; e++) done |= MODE == 0 ? lr.step(mine[e], !done) : mr.step((float)mine[e], !done);
        __builtin_amdgcn_wave_barrier();
        if (__all(done)) break;
    }
    if (live) finish<T, U, MODE>(lr, mr, range, out + ray, status);
}

// ---- min / max of the volume as f32 (mida_internal's pre-pass, mips.rs:113-123) ------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_minmax_part(const T *__restrict__ vol, int64_t n, T *__restrict__ part) {
    __shared__ T s_mn[4], s_mx[4];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    T mn = vol[0], mx = vol[0];
    constexpr int V = 16 / sizeof(T);
    if ((((uintptr_t)vol) & 15) == 0) { // 16-B loads over the aligned body, scalar tail below
        typedef T vec_t __attribute__((ext_vector_type(V)));
        const int64_t nv = n / V;
        for (int64_t q = i; q < nv; q += stride) {
            const vec_t x = reinterpret_cast<const vec_t *>(vol)[q];
#pragma unroll
            for (int e = 0; e < V; e++) {
                mn = x[e] < mn ? x[e] : mn;
                mx = x[e] > mx ? x[e] : mx;
            }
        }
        i += nv * V;
    }
    for (; i < n; i += stride) {
        const T v = vol[i];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; q++) {
            mn = s_mn[q] < mn ? s_mn[q] : mn;
            mx = s_mx[q] > mx ? s_mx[q] : mx;
        }
        part[2 * blockIdx.x] = mn;
        part[2 * blockIdx.x + 1] = mx;
    }
}
template <typename T>
__global__ __launch_bounds__(64) void k_minmax_final(const T *__restrict__ part, int nparts, float *__restrict__ out) {
    T mn = part[0], mx = part[1];
    for (int i = threadIdx.x; i < nparts; i += 64) {
        mn = part[2 * i] < mn ? part[2 * i] : mn;
        mx = part[2 * i + 1] > mx ? part[2 * i + 1] : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if (threadIdx.x == 0) {
        out[0] = (float)mn; // the conversion is monotone, so min/max commute with it
        out[1] = (float)mx;
    }
}

// ---- contour volume (calc_fcm_intensity, mips.rs:171-213) -------------------------------------------------------
template <typename T> __device__ __forceinline__ float fd_sub(T a, T b);
// the subtraction happens in T and wraps in a release build
template <> __device__ __forceinline__ float fd_sub<int16_t>(int16_t a, int16_t b) {
    return (float)(int16_t)((uint16_t)a - (uint16_t)b);
}
template <> __device__ __forceinline__ float fd_sub<uint8_t>(uint8_t a, uint8_t b) { return (float)(uint8_t)(a - b); }
template <> __device__ __forceinline__ float fd_sub<double>(double a, double b) { return (float)(a - b); }

// base^n as the reference computes it: Rust's f32::powf = the platform libm's powf (mips.rs:211).  pmode 1: n == 1 -- glibc's
// powf(x, 1) IS x for every float in [-1, 1] (checked exhaustively against libm, both signs); pmode 2 / 3: glibc's algorithm
// restated (glibc_powf.h), the build with fused multiply-adds that x86-64 glibc selects on CPUs with FMA + AVX2 / the plain
// build -- the host passes the one this machine's libm runs (ivx_powf_variant).  n == 2 is NOT base * base: glibc's powf is
// not correctly rounded and differs from the product in the last bit for 0.07 % of the bases in [2^-24, 1].
struct PowTabs { // glibc's two powf tables (512 bytes), copied to LDS once per workgroup: two dynamic look-ups per voxel
    double lt[32];
    unsigned long long et[32];
};
__device__ __forceinline__ void fcm_pow_setup(PowTabs *t) { // (every thread of the workgroup, before any early return)
    if (threadIdx.x < 32) glibc_powf::copy_table_entry((int)threadIdx.x, t->lt, (uint64_t *)t->et);
    __syncthreads();
}
__device__ __forceinline__ float fcm_pow(float base, float n, int pmode, const PowTabs *t) {
    if (pmode == 1) return base;
    return pmode == 2 ? glibc_powf::powf_glibc<true>(base, n, t->lt, (const uint64_t *)t->et)
                      : glibc_powf::powf_glibc<false>(base, n, t->lt, (const uint64_t *)t->et);
}

// base^n in float32 on the transcendental unit, with a bound: the contour MaxIP's pixel is T(max over the ray of gm * base^n),
// truncated to an integer, so a value known to (1 +- rel) decides the pixel unless an integer lies inside that interval -- and
// only then is glibc's powf (two table look-ups and a dozen double-precision operations per voxel: the exponent-2 sweep was
// ALU-bound at 0.09 of the roofline) worth its price.  v_log_f32 and v_exp_f32 are good to 1 ulp (CDNA ISA); with y = n * log2(base)
// the result's relative error is below (|y| ln 2 + 2) * 2^-23, glibc's own result is within 1 ulp of the true power: the bound
// returned, (|y| + 4) * 2^-21, keeps a factor of four in hand (tests/test_gpu_rays.py checks it on millions of inputs).
// base in [0, 1] (1 - |d / gm|), n > 0.
__device__ __forceinline__ float fcm_pow_fast(float base, float n, float *rel) {
    *rel = 0.0f;
    if (base <= 0.0f) return 0.0f; // powf(+0, n > 0) = +0
    if (base >= 1.0f) return 1.0f; // powf(1, n) = 1
    const float y = n * __builtin_amdgcn_logf(base); // (v_log_f32: log2)
    if (!(y > -160.0f)) return 0.0f;                 // below 2^-160: nothing a 16-bit pixel can see (and no 0 * inf in the bound)
    *rel = (fabsf(y) + 4.0f) * 4.76837158203125e-7f;
    return __builtin_amdgcn_exp2f(y); // (v_exp_f32: 2^y; results below 2^-126 flush to zero -- covered: see the test)
}

// one voxel's contour value the exact way, from the volume (clamped neighbours): the few rays the fast fold cannot decide
__device__ __forceinline__ float fcm_voxel_exact(const int16_t *__restrict__ img, int64_t sz, int64_t sy, int64_t sx, int64_t z, int64_t y,
                                                 int64_t x, float n, int axis, int pmode, const PowTabs *pt) {
    const int64_t px = x == 0 ? 0 : x - 1, fx = x == sx - 1 ? sx - 1 : x + 1;
    const int64_t py = y == 0 ? 0 : y - 1, fy = y == sy - 1 ? sy - 1 : y + 1;
    const int64_t pz = z == 0 ? 0 : z - 1, fz = z == sz - 1 ? sz - 1 : z + 1;
    const int16_t *row = img + (z * sy + y) * sx;
    const float g0 = fd_sub<int16_t>(row[fx], row[px]) / (2.0f * 1.0f);
    const float g1 = fd_sub<int16_t>(img[(z * sy + fy) * sx + x], img[(z * sy + py) * sx + x]) / (2.0f * 1.0f);
    const float g2 = fd_sub<int16_t>(img[(fz * sy + y) * sx + x], img[(pz * sy + y) * sx + x]) / (2.0f * 1.0f);
    const float gm = sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
    if (gm == 0.0f) return 0.0f;
    const float d = axis == 0 ? g2 : axis == 1 ? g1 : g0;
    const float base = 1.0f - fabsf(d / gm);
    return gm * fcm_pow(base, n, pmode, pt);
}

template <typename T>
__global__ __launch_bounds__(256) void k_fcm_volume(const T *__restrict__ img, int64_t sz, int64_t sy, int64_t sx,
                                                    float n, int pmode, int axis, T *__restrict__ tmp, int *__restrict__ status) {
    __shared__ PowTabs s_pt;
    fcm_pow_setup(&s_pt);
    const PowTabs *pt = &s_pt;
    const int64_t total = sz * sy * sx;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t x = i % sx, r = i / sx, y = r % sy, z = r / sy;
        const int64_t px = x == 0 ? 0 : x - 1, fx = x == sx - 1 ? sx - 1 : x + 1;
        const int64_t py = y == 0 ? 0 : y - 1, fy = y == sy - 1 ? sy - 1 : y + 1;
        const int64_t pz = z == 0 ? 0 : z - 1, fz = z == sz - 1 ? sz - 1 : z + 1;
        const T *row = img + (z * sy + y) * sx;
        const float g0 = fd_sub<T>(row[fx], row[px]) / (2.0f * 1.0f);
        const float g1 = fd_sub<T>(img[(z * sy + fy) * sx + x], img[(z * sy + py) * sx + x]) / (2.0f * 1.0f);
        const float g2 = fd_sub<T>(img[(fz * sy + y) * sx + x], img[(pz * sy + y) * sx + x]) / (2.0f * 1.0f);
        const float gm = sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
        float v = 0.0f;
        if (gm != 0.0f) {
            const float d = axis == 0 ? g2 : axis == 1 ? g1 : axis == 2 ? g0 : 0.0f; // dir = unit axis
            const float base = 1.0f - fabsf(d / gm);
            const float sf = fcm_pow(base, n, pmode, pt);
            v = gm * sf;
        }
        T o = (T)0;
        if (!numcast<T>(v, &o)) atomicMin(status, IVX_EDOM);
        tmp[i] = o;
    }
}

// int16 fast path: one lane = 8 consecutive voxels of a row (sx % 8 == 0, 16-B aligned): five 16-B loads (the chunk,
// the chunks of rows y-+1 and slices z-+1) + two 2-B loads (x neighbours across the chunk edge) and one 16-B store,
// instead of 7 two-byte loads and a two-byte store per voxel
typedef short rshort8_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_fcm_volume_i16x8(const int16_t *__restrict__ img, int64_t sz, int64_t sy, int64_t sx,
                                                          float n, int pmode, int axis, int16_t *__restrict__ tmp,
                                                          int *__restrict__ status) {
    __shared__ PowTabs s_pt;
    fcm_pow_setup(&s_pt);
    const PowTabs *pt = &s_pt;
    const int64_t cpr = sx / 8, total = sz * sy * cpr;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t q = i % cpr, r = i / cpr, y = r % sy, z = r / sy;
        const int64_t x0 = q * 8;
        const int64_t py = y == 0 ? 0 : y - 1, fy = y == sy - 1 ? sy - 1 : y + 1;
        const int64_t pz = z == 0 ? 0 : z - 1, fz = z == sz - 1 ? sz - 1 : z + 1;
        const int16_t *row = img + (z * sy + y) * sx;
        const rshort8_t c = *reinterpret_cast<const rshort8_t *>(row + x0);
        const rshort8_t yp = *reinterpret_cast<const rshort8_t *>(img + (z * sy + py) * sx + x0);
        const rshort8_t yf = *reinterpret_cast<const rshort8_t *>(img + (z * sy + fy) * sx + x0);
        const rshort8_t zp = *reinterpret_cast<const rshort8_t *>(img + (pz * sy + y) * sx + x0);
        const rshort8_t zf = *reinterpret_cast<const rshort8_t *>(img + (fz * sy + y) * sx + x0);
        const int16_t left = x0 == 0 ? (int16_t)c[0] : row[x0 - 1];          // clamped: px = x == 0 ? 0 : x - 1
        const int16_t right = x0 + 8 == sx ? (int16_t)c[7] : row[x0 + 8];    // fx = x == sx-1 ? sx-1 : x + 1
        rshort8_t o;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int16_t xm = e == 0 ? left : (int16_t)c[e - 1], xp = e == 7 ? right : (int16_t)c[e + 1];
            const float g0 = fd_sub<int16_t>(xp, xm) / (2.0f * 1.0f);
            const float g1 = fd_sub<int16_t>((int16_t)yf[e], (int16_t)yp[e]) / (2.0f * 1.0f);
            const float g2 = fd_sub<int16_t>((int16_t)zf[e], (int16_t)zp[e]) / (2.0f * 1.0f);
            const float gm = sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
            float v = 0.0f;
            if (gm != 0.0f) {
                const float d = axis == 0 ? g2 : axis == 1 ? g1 : axis == 2 ? g0 : 0.0f;
                const float base = 1.0f - fabsf(d / gm);
                const float sf = fcm_pow(base, n, pmode, pt);
                v = gm * sf;
            }
            int16_t ov = 0;
            if (!numcast<int16_t>(v, &ov)) atomicMin(status, IVX_EDOM);
            o[e] = ov;
        }
        *reinterpret_cast<rshort8_t *>(tmp + (z * sy + y) * sx + x0) = o;
    }
}

// ---- fused contour MaxIP (fast_countour_mip with tmip == 0, mips.rs:237-247) ------------------------------------------
// The reference materialises tmp[z, y, x] = T(calc_fcm_intensity) and folds it with max along the axis.  NumCast's
// truncation is monotone, so max over T(v) == T(max over v): the contour value is folded in f32 as it is computed and cast
// once per pixel -- no 2 B/voxel temp written and read back.  A value NumCast would refuse (the reference panics while
// building tmp) is reported through `status` exactly as k_fcm_volume does.
// int16, rows of whole 16-byte chunks.  Lane = 8 consecutive voxels of a row.
//   AXIS 0 / 1: the lane walks a segment of the ray axis with a 3-chunk sliding window in registers (previous, centre,
//               next along the ray axis); per step it loads the next chunk, the two chunks of the OTHER in-slice axis
//               (neighbour lanes' centre chunks: L1 / L2 hits) and the two 2-byte x-neighbours across the chunk edge.
//   AXIS 2:     a wave per row: each lane folds its chunks' 8 values, then a shuffle tree.
// the same value through fcm_pow_fast: lo <= the exact value <= hi
__device__ __forceinline__ void fcm_value_fast(int16_t xm, int16_t xp, int16_t ym, int16_t yp, int16_t zm, int16_t zp, float n, int axis,
                                               float *lo, float *hi) {
    const float g0 = fd_sub<int16_t>(xp, xm) / (2.0f * 1.0f);
    const float g1 = fd_sub<int16_t>(yp, ym) / (2.0f * 1.0f);
    const float g2 = fd_sub<int16_t>(zp, zm) / (2.0f * 1.0f);
    const float gm = sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
    *lo = *hi = 0.0f;
    if (gm != 0.0f) {
        const float d = axis == 0 ? g2 : axis == 1 ? g1 : g0;
        const float base = 1.0f - fabsf(d / gm);
        float rel;
        const float v = gm * fcm_pow_fast(base, n, &rel);
        const float e = v * rel;
        *lo = v - e;
        *hi = v + e;
    }
}

// Exponent 1 (the GUI's default): v = gm * (1 - |d| / gm) IS gm - |d| up to rounding, and with D = the wrapped int16 differences
// (g = D / 2) that is (sqrt(D0^2 + D1^2 + D2^2) - |Dray|) / 2.  The fold keeps t = v_sqrt_f32(S) - |Dray| (S exact in 32 bits:
// three squares of at most 2^30; the unit's root is good to 1 ulp) and the ray's largest S; the pixel's bounds are
// max(t) / 2 -+ sqrt(max S) / 2 * 2^-20 (k_fcm_max_decide).  Why that holds, with gm and v the real values: the reference's float
// sequence gives gm (1 + a) * (base + b) * (1 + r) with |a| <= 2.5 * 2^-24 (squares and sums rounded, correctly rounded root),
// |b| <= 2^-22 (quotient 3.5 * 2^-24, 1 - q half an ulp of 1), |r| <= 2^-24: within gm * 2^-22 + v * 3.5 * 2^-24 of gm - |d|; this
// one (S rounded once, the unit's root, one subtraction) within gm * 1.25 * 2^-23 + v * 2^-24 -- together, v <= gm, at most
// 0.69 * gm * 2^-20 for the voxel, and the ray's largest gm bounds every voxel's.  ~16 instructions a voxel instead of ~45 with the
// correctly rounded root and quotient (the exponent-1 fold was ALU-bound at 0.17 of the roofline); tests/test_gpu_rays.py checks
// the bound on 8 M gradients (worst observed: 0.2 of it).
__device__ __forceinline__ void fcm_lin_fold(int dray, int da, int db, float *t_max, uint32_t *s_max) {
    const uint32_t S = (uint32_t)(dray * dray) + (uint32_t)(da * da) + (uint32_t)(db * db);
    const float t = __builtin_amdgcn_sqrtf((float)S) - fabsf((float)dray);
    *t_max = t > *t_max ? t : *t_max;
    *s_max = S > *s_max ? S : *s_max;
}

// Exponents >= 1 other than 1: the same integer S, the unit's reciprocal root for both gm and the quotient, fcm_pow_fast for the
// power.  Against the reference's float sequence: base within 2^-21 (2^-22 each side of the real 1 - |d| / gm: see above; here
// the quotient is |D| * v_rsq_f32(S), 1.75 * 2^-23 relative, and 1 - q half an ulp), and for x, y in [0, 1], n >= 1,
// |x^n - y^n| <= n |x - y|: the power within n * 2^-21 + the fast power's own bound; gm (S * rsq / 2) within 2.5 * 2^-23 relative.
// Bounds: v (1 -+ (rel + 2^-21)) -+ gm * n * 2^-21.  ~35 instructions a voxel against ~80 with the correctly rounded root and
// quotient in front of the same fast power.
__device__ __forceinline__ void fcm_pow_fold(int dray, int da, int db, float n, float *lo_max, float *hi_max) {
    const uint32_t S = (uint32_t)(dray * dray) + (uint32_t)(da * da) + (uint32_t)(db * db);
    const float Sf = (float)S;
    const float r = __builtin_amdgcn_rsqf(Sf);
    const float gm2 = S ? Sf * r : 0.0f; // sqrt(S) = 2 gm
    const float base = S ? 1.0f - fabsf((float)dray) * r : 0.0f;
    float rel;
    const float v = 0.5f * gm2 * fcm_pow_fast(base, n, &rel);
    const float e = v * (rel + 4.76837158203125e-7f) + gm2 * (n * 2.384185791015625e-7f); // gm * n * 2^-21 = gm2 * n * 2^-22
    const float lo = v - e, hi = v + e;
    *lo_max = lo > *lo_max ? lo : *lo_max;
    *hi_max = hi > *hi_max ? hi : *hi_max;
}

__device__ __forceinline__ float fcm_value(int16_t xm, int16_t xp, int16_t ym, int16_t yp, int16_t zm, int16_t zp, float n,
                                           int axis, int pmode, const PowTabs *pt) {
    const float g0 = fd_sub<int16_t>(xp, xm) / (2.0f * 1.0f);
    const float g1 = fd_sub<int16_t>(yp, ym) / (2.0f * 1.0f);
    const float g2 = fd_sub<int16_t>(zp, zm) / (2.0f * 1.0f);
    const float gm = sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
    float v = 0.0f;
    if (gm != 0.0f) {
        const float d = axis == 0 ? g2 : axis == 1 ? g1 : g0;
        const float base = 1.0f - fabsf(d / gm);
        const float sf = fcm_pow(base, n, pmode, pt);
        v = gm * sf;
    }
    return v;
}

// FAST: fcm_value_fast's bounds folded instead of the exact value -- two partial planes per segment (lower, upper; the upper ones
// behind all the lower ones), decided or handed to k_fcm_fix by k_fcm_max_combine
template <int AXIS, int FAST> // FAST 0: exact; 1: bounds of the power; 2: exponent 1 (fcm_lin_fold); 3: exponents >= 1 (fcm_pow_fold)
__global__ __launch_bounds__(256) void k_fcm_max_walk(const int16_t *__restrict__ img, int64_t sz, int64_t sy, int64_t sx,
                                                      float n, int pmode, int64_t seg, float *__restrict__ partial,
                                                      int *__restrict__ status, uint32_t *nlist) {
    __shared__ PowTabs s_pt;
    if (!FAST) fcm_pow_setup(&s_pt);
    const PowTabs *pt = &s_pt;
    // output pixel row r = y (AXIS 0) or z (AXIS 1); the ray runs along l = z (AXIS 0) or y (AXIS 1)
    const int64_t cpr = sx / 8;
    const int64_t nr = AXIS == 0 ? sy : sz, len = AXIS == 0 ? sz : sy;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (FAST && nlist && t == 0 && blockIdx.y == 0) *nlist = 0; // (k_fcm_max_decide's list starts empty: it runs behind this kernel)
    if (t >= nr * cpr) return;
    const int64_t r = t / cpr, x0 = (t - r * cpr) * 8;
    const int64_t l0 = (int64_t)blockIdx.y * seg, l1 = l0 + seg < len ? l0 + seg : len;
    // strides of the ray axis (sl) and of the other in-slice axis (so), in voxels; o = this lane's fixed coordinate there
    const int64_t sl = AXIS == 0 ? sy * sx : sx, so = AXIS == 0 ? sx : sy * sx, no = nr;
    const int64_t om = r == 0 ? 0 : r - 1, op = r == no - 1 ? no - 1 : r + 1;
    const int16_t *col = img + r * so + x0;             // + l * sl
    const int16_t *colm = img + om * so + x0, *colp = img + op * so + x0;
    auto ld = [](const int16_t *p) { return *reinterpret_cast<const rshort8_t *>(p); };
    rshort8_t prev = ld(col + (l0 == 0 ? 0 : l0 - 1) * sl), cur = ld(col + l0 * sl);
    float acc[8], acc_hi[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        acc[e] = -INFINITY;
        acc_hi[e] = FAST == 2 ? 0.0f : -INFINITY; // (FAST == 2: the bits of the largest S, an unsigned integer)
    }
    bool bad = false;
#pragma unroll 2
    for (int64_t l = l0; l < l1; l++) {
        const int64_t ln = l == len - 1 ? len - 1 : l + 1;
        const rshort8_t next = ld(col + ln * sl);
        const rshort8_t am = ld(colm + l * sl), ap = ld(colp + l * sl);
        const int16_t *row = col + l * sl - x0;
        const int16_t left = x0 == 0 ? (int16_t)cur[0] : row[x0 - 1];
        const int16_t right = x0 + 8 == sx ? (int16_t)cur[7] : row[x0 + 8];
        rshort8_t dx8, da8, dl8; // FAST >= 2: the three differences of the whole chunk, wrapping in 16 bits like the reference's T
        if (FAST >= 2) {
            rshort8_t xm8 = __builtin_shufflevector(cur, cur, 0, 0, 1, 2, 3, 4, 5, 6), xp8 = __builtin_shufflevector(cur, cur, 1, 2, 3, 4, 5, 6, 7, 7);
            xm8[0] = left;
            xp8[7] = right;
            dx8 = xp8 - xm8;
            da8 = ap - am;
            dl8 = next - prev;
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int16_t xm = e == 0 ? left : (int16_t)cur[e - 1], xp = e == 7 ? right : (int16_t)cur[e + 1];
            if (FAST == 2) { // (the ray runs along prev -> next for both axes)
                uint32_t sm = __float_as_uint(acc_hi[e]);
                fcm_lin_fold((int)dl8[e], (int)dx8[e], (int)da8[e], &acc[e], &sm);
                acc_hi[e] = __uint_as_float(sm);
            } else if (FAST == 3) {
                fcm_pow_fold((int)dl8[e], (int)dx8[e], (int)da8[e], n, &acc[e], &acc_hi[e]);
            } else if (FAST) {
                float lo, hi;
                if (AXIS == 0) fcm_value_fast(xm, xp, (int16_t)am[e], (int16_t)ap[e], (int16_t)prev[e], (int16_t)next[e], n, 0, &lo, &hi);
                else fcm_value_fast(xm, xp, (int16_t)prev[e], (int16_t)next[e], (int16_t)am[e], (int16_t)ap[e], n, 1, &lo, &hi);
                acc[e] = lo > acc[e] ? lo : acc[e];
                acc_hi[e] = hi > acc_hi[e] ? hi : acc_hi[e];
            } else {
                const float v = AXIS == 0 ? fcm_value(xm, xp, (int16_t)am[e], (int16_t)ap[e], (int16_t)prev[e], (int16_t)next[e], n, 0, pmode, pt)
                                          : fcm_value(xm, xp, (int16_t)prev[e], (int16_t)next[e], (int16_t)am[e], (int16_t)ap[e], n, 1, pmode, pt);
                bad |= !(v > -32769.0f && v < 32768.0f);
                acc[e] = v > acc[e] ? v : acc[e];
            }
        }
        prev = cur;
        cur = next;
    }
    if (bad) atomicMin(status, IVX_EDOM);
    float *o = partial + ((int64_t)blockIdx.y * nr + r) * sx + x0;
    *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4 *>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    if (FAST) {
        float *oh = o + (int64_t)gridDim.y * nr * sx;
        *reinterpret_cast<float4 *>(oh) = make_float4(acc_hi[0], acc_hi[1], acc_hi[2], acc_hi[3]);
        *reinterpret_cast<float4 *>(oh + 4) = make_float4(acc_hi[4], acc_hi[5], acc_hi[6], acc_hi[7]);
    }
}

template <int FAST> // FAST (1 / 2 as above): the bounds' two planes (partial: lower then upper, one segment) instead of the pixel
__global__ __launch_bounds__(256) void k_fcm_max_rows(const int16_t *__restrict__ img, int64_t sz, int64_t sy, int64_t sx, float n,
                                                      int pmode, int16_t *__restrict__ out, float *__restrict__ partial,
                                                      int *__restrict__ status, uint32_t *nlist) {
    __shared__ PowTabs s_pt;
    if (!FAST) fcm_pow_setup(&s_pt);
    const PowTabs *pt = &s_pt;
    if (FAST && nlist && blockIdx.x == 0 && threadIdx.x == 0) *nlist = 0;
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= sz * sy) return;
    const int64_t z = ray / sy, y = ray - z * sy;
    const int64_t py = y == 0 ? 0 : y - 1, fy = y == sy - 1 ? sy - 1 : y + 1;
    const int64_t pz = z == 0 ? 0 : z - 1, fz = z == sz - 1 ? sz - 1 : z + 1;
    const int16_t *row = img + (z * sy + y) * sx;
    const int16_t *rym = img + (z * sy + py) * sx, *ryp = img + (z * sy + fy) * sx;
    const int16_t *rzm = img + (pz * sy + y) * sx, *rzp = img + (fz * sy + y) * sx;
    float acc = -INFINITY, acc_hi = -INFINITY;
    uint32_t acc_s = 0; // (FAST == 2: the largest S)
    bool bad = false;
    for (int64_t x0 = (int64_t)lane * 8; x0 < sx; x0 += 512) {
        const rshort8_t c = *reinterpret_cast<const rshort8_t *>(row + x0);
        const rshort8_t ym = *reinterpret_cast<const rshort8_t *>(rym + x0), yp = *reinterpret_cast<const rshort8_t *>(ryp + x0);
        const rshort8_t zm = *reinterpret_cast<const rshort8_t *>(rzm + x0), zp = *reinterpret_cast<const rshort8_t *>(rzp + x0);
        const int16_t left = x0 == 0 ? (int16_t)c[0] : row[x0 - 1];
        const int16_t right = x0 + 8 == sx ? (int16_t)c[7] : row[x0 + 8];
        const rshort8_t dy8 = yp - ym, dz8 = zp - zm;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int16_t xm = e == 0 ? left : (int16_t)c[e - 1], xp = e == 7 ? right : (int16_t)c[e + 1];
            if (FAST == 2) { // (the ray runs along x)
                fcm_lin_fold((int)(int16_t)((uint16_t)xp - (uint16_t)xm), (int)dy8[e], (int)dz8[e], &acc, &acc_s);
            } else if (FAST == 3) {
                fcm_pow_fold((int)(int16_t)((uint16_t)xp - (uint16_t)xm), (int)dy8[e], (int)dz8[e], n, &acc, &acc_hi);
            } else if (FAST) {
                float lo, hi;
                fcm_value_fast(xm, xp, (int16_t)ym[e], (int16_t)yp[e], (int16_t)zm[e], (int16_t)zp[e], n, 2, &lo, &hi);
                acc = lo > acc ? lo : acc;
                acc_hi = hi > acc_hi ? hi : acc_hi;
            } else {
                const float v = fcm_value(xm, xp, (int16_t)ym[e], (int16_t)yp[e], (int16_t)zm[e], (int16_t)zp[e], n, 2, pmode, pt);
                bad |= !(v > -32769.0f && v < 32768.0f);
                acc = v > acc ? v : acc;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float b = __shfl_xor(acc, o, 64);
        acc = b > acc ? b : acc;
        if (FAST == 2) {
            const uint32_t bs = (uint32_t)__shfl_xor((int)acc_s, o, 64);
            acc_s = bs > acc_s ? bs : acc_s;
        } else if (FAST) {
            const float bh = __shfl_xor(acc_hi, o, 64);
            acc_hi = bh > acc_hi ? bh : acc_hi;
        }
    }
    if (bad) atomicMin(status, IVX_EDOM);
    if (lane == 0) {
        if (FAST) {
            partial[ray] = acc;
            partial[sz * sy + ray] = FAST == 2 ? __uint_as_float(acc_s) : acc_hi;
        } else {
            out[ray] = (int16_t)acc; // (sx >= 8 here, so acc is a real value; out of range only with `bad` set)
        }
    }
}

__global__ __launch_bounds__(256) void k_fcm_max_combine(const float *__restrict__ partial, int64_t npix, int split,
                                                         int16_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    float acc = partial[i];
    for (int s = 1; s < split; s++) {
        const float b = partial[(int64_t)s * npix + i];
        acc = b > acc ? b : acc;
    }
    out[i] = (int16_t)acc;
}

// FAST fold: partial = `split` planes of lower bounds, then `split` planes of upper bounds.  A pixel whose bounds truncate to the
// same integer is decided; the others (an integer inside the interval: a few in a thousand) go on a list for k_fcm_fix.
// lin (exponent 1): the first planes hold max t, the second ones the bits of max S (fcm_lin_fold): bounds max t / 2 -+ sqrt(max S) / 2 * 2^-20.
__global__ __launch_bounds__(256) void k_fcm_max_decide(const float *__restrict__ partial, int64_t npix, int split, int lin,
                                                        int16_t *__restrict__ out, uint32_t *__restrict__ list, uint32_t *nlist) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    float lo = partial[i], hi = partial[(int64_t)split * npix + i], e_lin = 0.0f;
    if (lin) {
        uint32_t sm = __float_as_uint(hi);
        for (int s = 1; s < split; s++) {
            const float a = partial[(int64_t)s * npix + i];
            const uint32_t b = __float_as_uint(partial[(int64_t)(split + s) * npix + i]);
            lo = a > lo ? a : lo;
            sm = b > sm ? b : sm;
        }
        const float v = 0.5f * lo, e = sqrtf((float)sm) * 4.76837158203125e-7f; // 2^-21 = 2^-20 / 2
        lo = v - e;
        hi = v + e;
        e_lin = e;
    } else {
        for (int s = 1; s < split; s++) {
            const float a = partial[(int64_t)s * npix + i], b = partial[(int64_t)(split + s) * npix + i];
            lo = a > lo ? a : lo;
            hi = b > hi ? b : hi;
        }
    }
    // (values are >= 0; anything near the top of int16 goes to the exact fold, which also owns the out-of-range report)
    if (hi < 32766.0f && (int32_t)lo == (int32_t)hi) {
        out[i] = (int16_t)lo;
    } else {
        // only the segments whose upper bound reaches the pixel's lower bound can hold the maximum: the exact pass walks those
        // (one or two of 8 / 16: its scattered loads, not its arithmetic, are what it costs)
        uint32_t m = 0;
        if (!(hi < 32766.0f)) {
            m = 0xFFFFFFFFu;
        } else {
            for (int s = 0; s < split; s++) { // (lin: the ray's largest S bounds every segment's)
                const float hs = lin ? 0.5f * partial[(int64_t)s * npix + i] + e_lin : partial[(int64_t)(split + s) * npix + i];
                if (hs >= lo) m |= 1u << s;
            }
        }
        out[i] = 0;
        const uint32_t k = atomicAdd(nlist, 1u);
        list[2 * k] = (uint32_t)i;
        list[2 * k + 1] = m;
    }
}

// the undecided pixels, exactly: one workgroup per pixel, lanes along the ray (a 512-voxel ray is two dependent rounds of seven
// loads, not eight), the reference's float sequence -- glibc's powf, correctly rounded root and quotient -- for every voxel of it
__global__ __launch_bounds__(256) void k_fcm_fix(const int16_t *__restrict__ img, int64_t sz, int64_t sy, int64_t sx, float n, int axis,
                                                 int pmode, int64_t seg, const uint32_t *__restrict__ list, const uint32_t *__restrict__ nlist,
                                                 int16_t *__restrict__ out, int *__restrict__ status) {
    __shared__ PowTabs s_pt;
    __shared__ float s_acc[4];
    const uint32_t nl = *nlist;
    if (blockIdx.x >= nl) return; // (whole workgroups: a few hundred pixels of 2^18 are open, most of the grid leaves here)
    const uint2 first = reinterpret_cast<const uint2 *>(list)[blockIdx.x];
    if (pmode != 1) fcm_pow_setup(&s_pt); // (exponent 1 takes no table)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t len = axis == 0 ? sz : axis == 1 ? sy : sx, nc = axis == 2 ? sy : sx;
    for (uint32_t w = blockIdx.x; w < nl; w += gridDim.x) {
        const uint2 ent = w == blockIdx.x ? first : reinterpret_cast<const uint2 *>(list)[w]; // (pixel, the segments of its ray still open)
        const int64_t i = ent.x, r = i / nc, c = i - r * nc; // pixel (r, c): axis 0 -> (y, x), 1 -> (z, x), 2 -> (z, y)
        float acc = -INFINITY;
        bool bad = false;
#pragma unroll 2
        for (int64_t l = threadIdx.x; l < len; l += 256) {
            if (!((ent.y >> (uint32_t)(l / seg)) & 1u)) continue;
            const int64_t z = axis == 0 ? l : r, y = axis == 0 ? r : axis == 1 ? l : c, x = axis == 2 ? l : c;
            const float v = fcm_voxel_exact(img, sz, sy, sx, z, y, x, n, axis, pmode, &s_pt);
            bad |= !(v > -32769.0f && v < 32768.0f);
            acc = v > acc ? v : acc;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float b = __shfl_xor(acc, o, 64);
            acc = b > acc ? b : acc;
        }
        if (bad) atomicMin(status, IVX_EDOM);
        if (lane == 0) s_acc[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = s_acc[0];
            for (int k = 1; k < 4; k++) m = s_acc[k] > m ? s_acc[k] : m;
            out[i] = (int16_t)m;
        }
        __syncthreads();
    }
}

template <typename T, typename U, int MODE>
static int launch_rays(const void *vol, int64_t dz, int64_t dy, int64_t dx, int axis, double p0, double p1,
                       const float *minmax, void *out, int *status, hipStream_t st, const double *state_in = nullptr,
                       double *state_out = nullptr) {
    const RayGeom g = ray_geom(axis, dz, dy, dx);
    const int64_t npix = g.nr * g.nc;
    if (npix == 0) return IVX_OK;
    if (axis == 2) {
        hipLaunchKernelGGL((k_rays_rows<T, U, MODE>), dim3((unsigned)ivx::cdiv(npix, 256)), dim3(256), 0, st,
                           (const T *)vol, npix, g.len, p0, p1, minmax, (U *)out, status);
    } else {
        hipLaunchKernelGGL((k_rays_strided<T, U, MODE>), dim3((unsigned)ivx::cdiv(npix, 256)), dim3(256), 0, st,
                           (const T *)vol, g, p0, p1, minmax, (U *)out, status, state_in, state_out);
    }
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

template <typename T> static int run_minmax(const void *vol, int64_t n, float *out, hipStream_t st) {
    void *part;
    const int nb = (int)(ivx::cdiv(n, 256) < 2048 ? ivx::cdiv(n, 256) : 2048);
    int rc = ivx::ws_get_s(ivx::WS_AUX3, st, (size_t)nb * 2 * sizeof(T) + 64, &part);
    if (rc) return rc;
    hipLaunchKernelGGL(k_minmax_part<T>, dim3(nb), dim3(256), 0, st, (const T *)vol, n, (T *)part);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_minmax_final<T>, dim3(1), dim3(64), 0, st, (const T *)part, nb, out);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

} // namespace

extern "C" int ivx_dev_minmax_f32(int dtype, const void *vol, int64_t n, float *minmax2, void *stream) {
    IVX_REQUIRE(n > 0, IVX_EDOM, "minmax: empty volume (the reference unwraps a None)");
    hipStream_t st = ivx::S(stream);
    switch (dtype) {
    case IVX_I16: return run_minmax<int16_t>(vol, n, minmax2, st);
    case IVX_U8: return run_minmax<uint8_t>(vol, n, minmax2, st);
    case IVX_F64: return run_minmax<double>(vol, n, minmax2, st);
    }
    ivx::set_error("minmax: unsupported dtype %d", dtype);
    return IVX_EINVAL;
}

extern "C" int ivx_dev_mida(int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx, int axis, float wl, float ww,
                            const float *minmax2, int out_dtype, void *out, int *status, void *stream) {
    hipStream_t st = ivx::S(stream);
    if (axis > 2 || axis < 0) axis = 2; // `_ => image.slice(s![r, c, ..])`, mips.rs:131
    if (dtype == IVX_I16 && out_dtype == IVX_I16)
        return launch_rays<int16_t, int16_t, 1>(vol, dz, dy, dx, axis, wl, ww, minmax2, out, status, st);
    if (dtype == IVX_U8 && out_dtype == IVX_U8)
        return launch_rays<uint8_t, uint8_t, 1>(vol, dz, dy, dx, axis, wl, ww, minmax2, out, status, st);
    if (dtype == IVX_F64 && out_dtype == IVX_U8)
        return launch_rays<double, uint8_t, 1>(vol, dz, dy, dx, axis, wl, ww, minmax2, out, status, st);
    if (dtype == IVX_F64 && out_dtype == IVX_F64) // only reached from fast_countour_mip (tmp f64 -> out f64)
        return launch_rays<double, double, 1>(vol, dz, dy, dx, axis, wl, ww, minmax2, out, status, st);
    ivx::set_error("mida: Invalid image or output type (in %d, out %d)", dtype, out_dtype);
    return IVX_EINVAL;
}

extern "C" int ivx_dev_lmip(int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx, int axis, double tmin,
                            double tmax, void *out, void *stream) {
    hipStream_t st = ivx::S(stream);
    if (axis < 0 || axis > 2) return IVX_OK; // `_ => ()`, mips.rs:84
    switch (dtype) {
    case IVX_I16: return launch_rays<int16_t, int16_t, 0>(vol, dz, dy, dx, axis, tmin, tmax, nullptr, out, nullptr, st);
    case IVX_U8: return launch_rays<uint8_t, uint8_t, 0>(vol, dz, dy, dx, axis, tmin, tmax, nullptr, out, nullptr, st);
    case IVX_F64: return launch_rays<double, double, 0>(vol, dz, dy, dx, axis, tmin, tmax, nullptr, out, nullptr, st);
    }
    ivx::set_error("lmip: unsupported dtype %d", dtype);
    return IVX_EINVAL;
}

// Rays along Z through ONE slab of a Z-sharded volume (axis 0 only): resume from `state_in` (NULL on the first slab),
// leave the state in `state_out` (NULL on the last slab, which writes `out` instead).  kind 0 = LMIP (p0, p1 = tmin,
// tmax), 1 = MIDA (p0, p1 = wl, ww; minmax2 = the min / max of the WHOLE volume as float32, on the device).
extern "C" int ivx_dev_rays_z_slab(int kind, int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx, double p0, double p1,
                                   const float *minmax2, const double *state_in, double *state_out, int out_dtype, void *out,
                                   int *status, void *stream) {
    hipStream_t st = ivx::S(stream);
    IVX_REQUIRE(kind == 0 || kind == 1, IVX_EINVAL, "rays_z_slab: kind must be 0 (lmip) or 1 (mida)");
    IVX_REQUIRE(dz > 0 && dy >= 0 && dx >= 0, IVX_EINVAL, "rays_z_slab: empty slab");
    IVX_REQUIRE(state_out || out, IVX_EINVAL, "rays_z_slab: the last slab needs an output image");
    if (kind == 0) {
        switch (dtype) {
        case IVX_I16: return launch_rays<int16_t, int16_t, 0>(vol, dz, dy, dx, 0, p0, p1, nullptr, out, nullptr, st, state_in, state_out);
        case IVX_U8: return launch_rays<uint8_t, uint8_t, 0>(vol, dz, dy, dx, 0, p0, p1, nullptr, out, nullptr, st, state_in, state_out);
        case IVX_F64: return launch_rays<double, double, 0>(vol, dz, dy, dx, 0, p0, p1, nullptr, out, nullptr, st, state_in, state_out);
        }
    } else {
        if (dtype == IVX_I16 && out_dtype == IVX_I16)
            return launch_rays<int16_t, int16_t, 1>(vol, dz, dy, dx, 0, p0, p1, minmax2, out, status, st, state_in, state_out);
        if (dtype == IVX_U8 && out_dtype == IVX_U8)
            return launch_rays<uint8_t, uint8_t, 1>(vol, dz, dy, dx, 0, p0, p1, minmax2, out, status, st, state_in, state_out);
        if (dtype == IVX_F64 && out_dtype == IVX_U8)
            return launch_rays<double, uint8_t, 1>(vol, dz, dy, dx, 0, p0, p1, minmax2, out, status, st, state_in, state_out);
    }
    ivx::set_error("rays_z_slab: unsupported dtype pair (in %d, out %d)", dtype, out_dtype);
    return IVX_EINVAL;
}

// Which build of glibc's powf this machine's libm runs (sysdeps/x86_64/fpu/multiarch/e_powf.c: `__powf_fma` when the CPU has
// FMA and AVX2, the plain one otherwise; other architectures build the plain source with their own contraction rules -- on
// those, and under another libm, IVX_POWF_VARIANT=fma|plain says which).  The two differ in the last bit of ~3 results in 10^9.
extern "C" int ivx_powf_variant(void) {
    static const int v = []() {
        const char *e = getenv("IVX_POWF_VARIANT");
        if (e) return strcmp(e, "fma") == 0 ? 1 : 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
        __builtin_cpu_init();
        return (__builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2")) ? 1 : 0;
#else
        return 0;
#endif
    }();
    return v;
}
static int fcm_pmode(float n) { return n == 1.0f ? 1 : (ivx_powf_variant() ? 2 : 3); }

// the power function of the contour MIP on its own (parity probe: tests compare it with the host libm's powf bit for bit)
__global__ __launch_bounds__(256) void k_powf_probe(const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ out,
                                                    int64_t n, int fma_build) {
    __shared__ PowTabs s_pt;
    fcm_pow_setup(&s_pt);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (fma_build >= 2) { // 2: the fast power, 3: its error bound (relative)
        float rel;
        const float v = fcm_pow_fast(x[i], y[i], &rel);
        out[i] = fma_build == 2 ? v : rel;
    } else {
        out[i] = fcm_pow(x[i], y[i], fma_build ? 2 : 3, &s_pt);
    }
}
extern "C" int ivx_dev_powf(const float *x, const float *y, float *out, int64_t n,
                            int variant /* -1: this machine's glibc build, 0 plain, 1 fma; 2: the fast power of the contour MaxIP, 3: its relative bound */,
                            void *stream) {
    IVX_REQUIRE(n >= 0 && (n == 0 || (x && y && out)), IVX_EINVAL, "powf: bad arguments");
    if (n == 0) return IVX_OK;
    hipLaunchKernelGGL(k_powf_probe, dim3((unsigned)ivx::cdiv(n, 256)), dim3(256), 0, ivx::S(stream), x, y, out, n,
                       variant < 0 ? ivx_powf_variant() : variant >= 2 ? variant : (variant ? 1 : 0));
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

// one voxel's bounds as the fused contour MaxIP folds them (tests): the wrapped int16 differences along the ray (d) and across it
// (two int16 in a word), the exponent (1: fcm_lin_fold with k_fcm_max_decide's bound for that voxel alone; > 1: fcm_pow_fold)
__global__ __launch_bounds__(256) void k_fcm_bounds_probe(const int32_t *__restrict__ d, const uint32_t *__restrict__ other, float n,
                                                          float *__restrict__ lo, float *__restrict__ hi, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int da = (int)(int16_t)(other[i] & 0xFFFFu), db = (int)(int16_t)(other[i] >> 16);
    if (n == 1.0f) {
        float t = -INFINITY;
        uint32_t sm = 0;
        fcm_lin_fold(d[i], da, db, &t, &sm);
        const float e = sqrtf((float)sm) * 4.76837158203125e-7f;
        lo[i] = 0.5f * t - e;
        hi[i] = 0.5f * t + e;
    } else {
        float l = -INFINITY, h = -INFINITY;
        fcm_pow_fold(d[i], da, db, n, &l, &h);
        lo[i] = l;
        hi[i] = h;
    }
}
extern "C" int ivx_dev_fcm_bounds(const int32_t *d, const uint32_t *other, float n, float *lo, float *hi, int64_t count, void *stream) {
    IVX_REQUIRE(count >= 0 && n >= 1.0f && n <= 64.0f && (count == 0 || (d && other && lo && hi)), IVX_EINVAL, "fcm_bounds: bad arguments");
    if (count == 0) return IVX_OK;
    hipLaunchKernelGGL(k_fcm_bounds_probe, dim3((unsigned)ivx::cdiv(count, 256)), dim3(256), 0, ivx::S(stream), d, other, n, lo, hi, count);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_fcm_volume(int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx, float n, int axis,
                                  void *tmp, int *status, void *stream) {
    hipStream_t st = ivx::S(stream);
    const int64_t total = dz * dy * dx;
    if (!total) return IVX_OK;
    const int64_t blocks = ivx::cdiv(total, 256);
    const int grid = (int)(blocks < 65536 ? blocks : 65536);
    const int pmode = fcm_pmode(n);
    if (dtype == IVX_I16 && dx % 8 == 0 && (((uintptr_t)vol | (uintptr_t)tmp) & 15) == 0) {
        const int64_t b8 = ivx::cdiv(total / 8, 256);
        hipLaunchKernelGGL(k_fcm_volume_i16x8, dim3((unsigned)(b8 < 65536 ? b8 : 65536)), dim3(256), 0, st, (const int16_t *)vol,
                           dz, dy, dx, n, pmode, axis, (int16_t *)tmp, status);
        IVX_LAUNCH_CHECK();
        return IVX_OK;
    }
    switch (dtype) {
    case IVX_I16: hipLaunchKernelGGL(k_fcm_volume<int16_t>, dim3(grid), dim3(256), 0, st, (const int16_t *)vol, dz, dy, dx, n, pmode, axis, (int16_t *)tmp, status); break;
    case IVX_U8: hipLaunchKernelGGL(k_fcm_volume<uint8_t>, dim3(grid), dim3(256), 0, st, (const uint8_t *)vol, dz, dy, dx, n, pmode, axis, (uint8_t *)tmp, status); break;
    case IVX_F64: hipLaunchKernelGGL(k_fcm_volume<double>, dim3(grid), dim3(256), 0, st, (const double *)vol, dz, dy, dx, n, pmode, axis, (double *)tmp, status); break;
    default: ivx::set_error("fcm: unsupported dtype %d", dtype); return IVX_EINVAL;
    }
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

// fast_countour_mip with tmip == 0 (mips.rs:237-247) without the contour volume: fused for int16 rows of whole 16-byte
// chunks (every volume the GUI holds), through a materialised temp from this stream's workspace otherwise.  Same bits.
extern "C" int ivx_dev_fcm_maxip(int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx, float n, int axis,
                                 void *out, int *status, void *stream) {
    hipStream_t st = ivx::S(stream);
    IVX_REQUIRE(axis >= 0 && axis <= 2, IVX_EINVAL, "fast_countour_mip: axis %d", axis);
    IVX_REQUIRE(dtype == IVX_I16 || dtype == IVX_U8, IVX_EINVAL, "fcm_maxip: int16 or uint8 images (float64 takes the LMIP fold)");
    const int64_t total = dz * dy * dx;
    if (!total) return IVX_OK;
    static const bool fused_ok = []() { const char *e = getenv("IVX_FCM_FUSED"); return !(e && e[0] == '0'); }();
    int rc;
    if (fused_ok && dtype == IVX_I16 && dx % 8 == 0 && (((uintptr_t)vol) & 15) == 0) {
        const int pmode = fcm_pmode(n);
        const int16_t *img = (const int16_t *)vol;
        // the fold on bounds from the transcendental unit, the exact value (glibc's powf, correctly rounded root and quotient) only
        // for the pixels the bounds leave open (IVX_FCM_FAST=0: glibc's powf for every voxel, as in round 5 -- A/B, tests)
        static const bool fast_ok = []() { const char *e = getenv("IVX_FCM_FAST"); return !(e && e[0] == '0'); }();
        const bool lin = fast_ok && pmode == 1; // exponent 1: bounds from the unit's root and reciprocal (fcm_value_fast<true>)
        const bool fast = lin || (fast_ok && n > 0.0f && n <= 64.0f);
        const bool unit = !lin && fast && n >= 1.0f; // the unit's root in front of the fast power (fcm_pow_fold)
        auto decide_and_fix = [&](float *part, int64_t npix, int split, int64_t seg, uint32_t *list, uint32_t *nlist) -> int {
            hipLaunchKernelGGL(k_fcm_max_decide, dim3((unsigned)ivx::cdiv(npix, 256)), dim3(256), 0, st, (const float *)part, npix, split,
                               lin ? 1 : 0, (int16_t *)out, list, nlist);
            IVX_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_fcm_fix, dim3(2048), dim3(256), 0, st, img, dz, dy, dx, n, axis, pmode, seg, (const uint32_t *)list,
                               (const uint32_t *)nlist, (int16_t *)out, status);
            IVX_LAUNCH_CHECK();
            static const bool debug = getenv("IVX_FCM_DEBUG") != nullptr; // (tools/probe_fcm_fix.py: how many pixels the bounds left open)
            if (debug) { uint32_t h = 0; hipStreamSynchronize(st); hipMemcpy(&h, nlist, 4, hipMemcpyDeviceToHost); fprintf(stderr, "fcm undecided %u of %lld (lin %d)\n", h, (long long)npix, (int)lin); }
            return IVX_OK;
        };
        if (axis == 2) {
            if (!fast) {
                hipLaunchKernelGGL(k_fcm_max_rows<0>, dim3((unsigned)ivx::cdiv(dz * dy, 4)), dim3(256), 0, st, img, dz, dy, dx, n, pmode,
                                   (int16_t *)out, (float *)nullptr, status, (uint32_t *)nullptr);
                IVX_LAUNCH_CHECK();
                return IVX_OK;
            }
            const int64_t npix = dz * dy;
            void *part;
            if ((rc = ivx::ws_get_s(ivx::WS_AUX3, st, (size_t)npix * 16 + 256, &part))) return rc;
            uint32_t *list = (uint32_t *)((float *)part + 2 * npix), *nlist = list + 2 * npix; // (list: pixel, open segments)
            hipLaunchKernelGGL(lin ? k_fcm_max_rows<2> : unit ? k_fcm_max_rows<3> : k_fcm_max_rows<1>, dim3((unsigned)ivx::cdiv(dz * dy, 4)), dim3(256), 0, st, img, dz, dy, dx,
                               n, pmode, (int16_t *)out, (float *)part, status, nlist);
            IVX_LAUNCH_CHECK();
            return decide_and_fix((float *)part, npix, 1, dx, list, nlist);
        }
        const int64_t nr = axis == 0 ? dy : dz, len = axis == 0 ? dz : dy, npix = nr * dx;
        const int64_t nblk = ivx::cdiv(nr * (dx / 8), 256);
        // segments of the ray: enough workgroups to fill the chip, each long enough to amortise its two window chunks
        // (the exponent-1 fold is light enough for its partial planes to show: 8 segments at 512^3 measured 0.113 ms against 0.120 with 16;
        // the power folds want the 16: 0.253 against 0.276)
        static const int64_t wg_env = []() { const char *e = getenv("IVX_FCM_WGS"); return (int64_t)(e && atoi(e) > 0 ? atoi(e) : 0); }();
        int64_t split = ivx::cdiv(wg_env ? wg_env : lin ? 1024 : 2048, nblk);
        if (split > len / 32) split = len / 32;
        if (split > 32) split = 32; // (k_fcm_max_decide names the open segments in one word)
        if (split < 1) split = 1;
        const int64_t seg = ivx::cdiv(len, split);
        split = ivx::cdiv(len, seg);
        void *part;
        if ((rc = ivx::ws_get_s(ivx::WS_AUX3, st, (size_t)npix * sizeof(float) * split * (fast ? 2 : 1) + (fast ? (size_t)npix * 8 + 256 : 64), &part)))
            return rc;
        if (fast) {
            uint32_t *list = (uint32_t *)((float *)part + 2 * split * npix), *nlist = list + 2 * npix;
            auto walk = axis == 0 ? (lin ? k_fcm_max_walk<0, 2> : unit ? k_fcm_max_walk<0, 3> : k_fcm_max_walk<0, 1>)
                                  : (lin ? k_fcm_max_walk<1, 2> : unit ? k_fcm_max_walk<1, 3> : k_fcm_max_walk<1, 1>);
            hipLaunchKernelGGL(walk, dim3((unsigned)nblk, (unsigned)split), dim3(256), 0, st, img, dz, dy, dx, n, pmode, seg, (float *)part, status, nlist);
            IVX_LAUNCH_CHECK();
            return decide_and_fix((float *)part, npix, (int)split, seg, list, nlist);
        }
        if (axis == 0)
            hipLaunchKernelGGL((k_fcm_max_walk<0, 0>), dim3((unsigned)nblk, (unsigned)split), dim3(256), 0, st, img, dz, dy, dx, n, pmode, seg,
                               (float *)part, status, (uint32_t *)nullptr);
        else
            hipLaunchKernelGGL((k_fcm_max_walk<1, 0>), dim3((unsigned)nblk, (unsigned)split), dim3(256), 0, st, img, dz, dy, dx, n, pmode, seg,
                               (float *)part, status, (uint32_t *)nullptr);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_fcm_max_combine, dim3((unsigned)ivx::cdiv(npix, 256)), dim3(256), 0, st, (const float *)part, npix,
                           (int)split, (int16_t *)out);
        IVX_LAUNCH_CHECK();
        return IVX_OK;
    }
    void *tmp;
    if ((rc = ivx::ws_get_s(ivx::WS_AUX0, st, (size_t)total * ivx::dtype_size(dtype) + 64, &tmp))) return rc;
    if ((rc = ivx_dev_fcm_volume(dtype, vol, dz, dy, dx, n, axis, tmp, status, stream))) return rc;
    return ivx_dev_mip_reduce(dtype, tmp, dz, dy, dx, axis, IVX_MIP_MAX, out, stream);
}

// ---- host forms -----------------------------------------------------------------------------------------------
namespace {
struct HostRay {
    void *d_in, *d_out, *d_small;
    int64_t osh[2];
    size_t isz;
};
static int host_prep(int dtype, const void *img, const int64_t shape[3], const int64_t strides[3], int axis,
                     size_t osz, HostRay *h) {
    using namespace ivx;
    h->isz = dtype_size(dtype);
    IVX_REQUIRE(dtype == IVX_I16 || dtype == IVX_U8 || dtype == IVX_F64, IVX_EINVAL, "Invalid image or output type");
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    const int a = (axis < 0 || axis > 2) ? 2 : axis;
    if (a == 0) { h->osh[0] = shape[1]; h->osh[1] = shape[2]; }
    else if (a == 1) { h->osh[0] = shape[0]; h->osh[1] = shape[2]; }
    else { h->osh[0] = shape[0]; h->osh[1] = shape[1]; }
    int rc;
    if ((rc = ws_get(WS_IN, n * h->isz, &h->d_in))) return rc;
    if ((rc = ws_get(WS_OUT, (size_t)h->osh[0] * h->osh[1] * osz, &h->d_out))) return rc;
    if ((rc = ws_get(WS_SMALL, 256, &h->d_small))) return rc;
    IVX_HIP(hipMemset(h->d_small, 0, 256));
    return upload_strided(h->d_in, img, shape, strides, h->isz, WS_IN);
}
static int host_status(const HostRay &h) {
    int st = 0;
    IVX_HIP(hipMemcpy(&st, (char *)h.d_small + 64, 4, hipMemcpyDeviceToHost));
    IVX_REQUIRE(st == 0, IVX_EDOM, "NumCast failure: a projected value does not fit the output dtype (the reference panics)");
    return IVX_OK;
}
} // namespace

extern "C" int ivx_mida(int dtype, const void *img, const int64_t shape[3], const int64_t strides[3], int axis,
                        double wl, double ww, int out_dtype, void *out, const int64_t out_strides[2]) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE((dtype == IVX_I16 && out_dtype == IVX_I16) || (dtype == IVX_U8 && out_dtype == IVX_U8) ||
                    (dtype == IVX_F64 && out_dtype == IVX_U8),
                IVX_EINVAL, "Invalid image or output type");
    IVX_REQUIRE(shape[0] * shape[1] * shape[2] > 0, IVX_EDOM, "mida: empty image (the reference unwraps a None)");
    HostRay h;
    int rc = host_prep(dtype, img, shape, strides, axis, dtype_size(out_dtype), &h);
    if (rc) return rc;
    float *mm = (float *)h.d_small;
    int *status = (int *)((char *)h.d_small + 64);
    if ((rc = ivx_dev_minmax_f32(dtype, h.d_in, shape[0] * shape[1] * shape[2], mm, nullptr))) return rc;
    if ((rc = ivx_dev_mida(dtype, h.d_in, shape[0], shape[1], shape[2], axis, (float)wl, (float)ww, mm, out_dtype,
                           h.d_out, status, nullptr)))
        return rc;
    IVX_HIP(hipDeviceSynchronize());
    if ((rc = download_strided2(out, h.osh, out_strides, h.d_out, dtype_size(out_dtype), WS_OUT))) return rc;
    return host_status(h);
}

extern "C" int ivx_lmip(int dtype, const void *img, const int64_t shape[3], const int64_t strides[3], int axis,
                        double tmin, double tmax, void *out, const int64_t out_strides[2]) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    if (axis < 0 || axis > 2) return IVX_OK;
    if (shape[0] * shape[1] * shape[2] == 0) return IVX_OK;
    HostRay h;
    int rc = host_prep(dtype, img, shape, strides, axis, dtype_size(dtype), &h);
    if (rc) return rc;
    if ((rc = ivx_dev_lmip(dtype, h.d_in, shape[0], shape[1], shape[2], axis, tmin, tmax, h.d_out, nullptr))) return rc;
    IVX_HIP(hipDeviceSynchronize());
    return download_strided2(out, h.osh, out_strides, h.d_out, h.isz, WS_OUT);
}

extern "C" int ivx_fast_countour_mip(int dtype, const void *img, const int64_t shape[3], const int64_t strides[3],
                                     float n, int axis, double wl, double ww, int tmip, void *out,
                                     const int64_t out_strides[2]) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    const int64_t nvox = shape[0] * shape[1] * shape[2];
    HostRay h;
    int rc = host_prep(dtype, img, shape, strides, axis, dtype_size(dtype), &h);
    if (rc) return rc;
    if (nvox == 0) return tmip == 2 ? IVX_EDOM : IVX_OK;
    float *mm = (float *)h.d_small;
    int *status = (int *)((char *)h.d_small + 64);
    if (tmip == 0 && dtype != IVX_F64 && axis >= 0 && axis <= 2) {
        // max fold: the contour value is folded as it is computed (no temp volume); a NumCast failure anywhere in the volume
        // is reported like the reference's panic while it builds tmp
        if ((rc = ivx_dev_fcm_maxip(dtype, h.d_in, shape[0], shape[1], shape[2], n, axis, h.d_out, status, nullptr))) return rc;
        IVX_HIP(hipDeviceSynchronize());
        if ((rc = host_status(h))) return rc;
        return download_strided2(out, h.osh, out_strides, h.d_out, h.isz, WS_OUT);
    }
    void *d_tmp;
    if ((rc = ws_get(WS_AUX0, (size_t)nvox * h.isz, &d_tmp))) return rc;
    if ((rc = ivx_dev_fcm_volume(dtype, h.d_in, shape[0], shape[1], shape[2], n, axis, d_tmp, status, nullptr))) return rc;
    if ((rc = host_status(h))) return rc; // the reference panics while building tmp, before any projection
    if (tmip == 0) {
        IVX_REQUIRE(axis >= 0 && axis <= 2, IVX_EINVAL, "fast_countour_mip: axis %d", axis);
        if (dtype == IVX_F64) {
            // fold_axis max over f64: same as LMIP with an unreachable threshold window is NOT equivalent; use the
            // ray kernel in LMIP mode with start never set: tmin > tmax
            rc = ivx_dev_lmip(dtype, d_tmp, shape[0], shape[1], shape[2], axis, 1.0, -1.0, h.d_out, nullptr);
        } else {
            rc = ivx_dev_mip_reduce(dtype, d_tmp, shape[0], shape[1], shape[2], axis, IVX_MIP_MAX, h.d_out, nullptr);
        }
        if (rc) return rc;
    } else if (tmip == 1) {
        IVX_REQUIRE(dtype != IVX_U8, IVX_EDOM, "fast_countour_mip: NumCast::from(700) does not fit uint8 (the reference panics)");
        if ((rc = ivx_dev_lmip(dtype, d_tmp, shape[0], shape[1], shape[2], axis, 700.0, 3033.0, h.d_out, nullptr))) return rc;
    } else if (tmip == 2) {
        if ((rc = ivx_dev_minmax_f32(dtype, d_tmp, nvox, mm, nullptr))) return rc;
        if ((rc = ivx_dev_mida(dtype, d_tmp, shape[0], shape[1], shape[2], axis, (float)wl, (float)ww, mm, dtype, h.d_out,
                               status, nullptr)))
            return rc;
    } else {
        return IVX_OK; // `_ => ()`: out untouched
    }
    IVX_HIP(hipDeviceSynchronize());
    if ((rc = host_status(h))) return rc;
    return download_strided2(out, h.osh, out_strides, h.d_out, h.isz, WS_OUT);
}
