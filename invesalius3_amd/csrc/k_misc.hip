// k_misc.hip -- the small streaming kernels around the segmentation tools.
//
//   fill_holes        fill_holes_automatically_internal           invesalius_rs/src/floodfill.rs:51-94
//   lut_u16           get_LUT_value(...).astype("uint16")         invesalius/data/imagedata_utils.py:555-564,
//                                                                 invesalius/data/watershed_process.py:34,42
//   shift_min_u16     (image - image.min()).astype("uint16")      invesalius/data/watershed_process.py:47,55
//   morph_gradient    scipy.ndimage.morphological_gradient(size)  invesalius/data/watershed_process.py:36-38,49-51
//                     == maximum_filter - minimum_filter, mode="reflect", uint16
//   watershed_merge   styles.py:2147-2152 (3-D) / 1984-1989 (2-D)
//   masked_stats      np.mean / np.std over image[bool_mask]      invesalius/data/styles.py:3237-3238
//
// All are HBM streaming passes (bytes per voxel in DESIGN.md section 3); no LDS tiling is needed except for the
// gradient, whose 27 taps are served by L1/L2 (each row is re-read by its y/z neighbours while still cached).
#include <algorithm>
#include <chrono>
#include <cstring>

#include "ivx_internal.h"

namespace {

static inline int grid_for(int64_t n, int per_thread = 1) {
    const int64_t b = ivx::cdiv(ivx::cdiv(n, per_thread), 256);
    return (int)(b < 1 ? 1 : (b < 65536 ? b : 65536));
}

// ---- fill holes ---------------------------------------------------------------------------------------------
// histogram of u32 labels.  Neighbouring voxels mostly share a label (background, big regions), so a wave first
// checks whether all its lanes hold the same label and then issues ONE atomic for the whole wave.
__global__ __launch_bounds__(256) void k_label_hist(const uint32_t *__restrict__ labels, int64_t n, uint32_t nlabels,
                                                    uint32_t *__restrict__ sizes, int *__restrict__ status) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += stride) {
        const int64_t i = base + threadIdx.x;
        bool live = i < n;
        const uint32_t l = live ? labels[i] : 0xffffffffu;
        if (live && l > nlabels) {
            atomicMin(status, IVX_ERANGE); // the reference indexes sizes[label]: out-of-bounds panic
            live = false;
        }
        if (!__ballot(live)) continue; // wave-uniform
        const uint32_t first = __shfl(l, __builtin_ctzll(__ballot(live)), 64);
        const unsigned long long same = __ballot(live && l == first);
        const unsigned long long livem = __ballot(live);
        if (same == livem) {
            if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(livem)) atomicAdd(&sizes[first], (uint32_t)__popcll(livem));
        } else if (live) {
            atomicAdd(&sizes[l], 1u);
        }
    }
}
__global__ void k_any_small(const uint32_t *__restrict__ sizes, uint32_t nlabels, uint32_t max_size, int *__restrict__ modified) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= (int64_t)nlabels && sizes[i] > 0 && sizes[i] <= max_size) *modified = 1; // benign race: all write 1
}
__global__ __launch_bounds__(256) void k_fill_small(uint8_t *__restrict__ mask, const uint32_t *__restrict__ labels,
                                                    int64_t n, const uint32_t *__restrict__ sizes, uint32_t max_size,
                                                    const int *__restrict__ modified) {
    if (!*modified) return;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (sizes[labels[i]] <= max_size) mask[i] = 254; // includes label 0 when it is small (faithful quirk Q5)
}

// ---- window / level LUT -------------------------------------------------------------------------------------
// np.piecewise on an int16 array: result dtype int16, the float64 expression is truncated toward zero.
// The input is int16, so the whole float64 expression has only 65536 possible arguments: evaluate it once per value into
// a 128 KiB table and turn the volume pass into a gather
// from L1/L2 -- the per-voxel float64 division made the direct form ALU-bound (0.15 ms at 512^3 instead of ~0.1).
__global__ __launch_bounds__(256) void k_lut_table(double window, double level, double top, int16_t *__restrict__ tab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x; // 0..65535 = the uint16 reinterpretation of the sample
    const double lo = level - 0.5 - (window - 1.0) / 2.0;
    const double hi = level - 0.5 + (window - 1.0) / 2.0;
    const double d = (double)(int16_t)(uint16_t)i;
    int16_t r;
    if (d <= lo) r = 0;
    else if (d > hi) r = (int16_t)top;
    else r = (int16_t)(((d - (level - 0.5)) / (window - 1.0) + 0.5) * top);
    tab[i] = r;
}
typedef short s8_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_lut_apply(const int16_t *__restrict__ img, int64_t n, const int16_t *__restrict__ tab,
                                                   int16_t *__restrict__ out, int vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        const int64_t nc = n / 8;
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += stride) {
            const s8_t v = __builtin_nontemporal_load(reinterpret_cast<const s8_t *>(img) + c);
            s8_t r;
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = tab[(uint16_t)v[j]];
            reinterpret_cast<s8_t *>(out)[c] = r;
        }
    }
    for (int64_t i = (vec ? (n / 8) * 8 : 0) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = tab[(uint16_t)img[i]];
}

__global__ __launch_bounds__(256) void k_shift_min_u16(const int16_t *__restrict__ img, int64_t n, int imin,
                                                       uint16_t *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint16_t m = (uint16_t)(int16_t)imin;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = (uint16_t)((uint16_t)img[i] - m); // int16 wrap-around then .astype("uint16") == mod 2^16
}

// ---- morphological gradient, cubic flat footprint of odd `size`, mode="reflect" --------------------------------
__device__ __forceinline__ int64_t reflect(int64_t i, int64_t n) { // d c b a | a b c d | d c b a
    if (i >= 0 && i < n) return i; // interior: no 64-bit modulo (it made the whole kernel ~10x slower)
    if (n == 1) return 0;
    const int64_t p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}
template <typename T>
__global__ __launch_bounds__(256) void k_morph_gradient(const T *__restrict__ in, int64_t dz, int64_t dy, int64_t dx, int rz,
                                                        int ry, int rx, T *__restrict__ out) {
    const int64_t total = dz * dy * dx;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t x = i % dx, q = i / dx, y = q % dy, z = q / dy;
        T mx = in[i], mn = in[i];
        for (int c = -rz; c <= rz; c++) {
            const int64_t zz = reflect(z + c, dz);
            for (int b = -ry; b <= ry; b++) {
                const int64_t yy = reflect(y + b, dy);
                const T *row = in + (zz * dy + yy) * dx;
                for (int a = -rx; a <= rx; a++) {
                    const T v = row[reflect(x + a, dx)];
                    mx = v > mx ? v : mx;
                    mn = v < mn ? v : mn;
                }
            }
        }
        out[i] = (T)(mx - mn);
    }
}

// 3x3x3 (the reference's default mg_size) on rows of whole 16-byte chunks: a lane owns 8 consecutive voxels; each of the
// nine (dz, dy) rows costs one 16-byte load plus the two neighbours at the chunk ends, the 3-wide max/min along x is
// formed in registers.  27 taps per voxel become ~3.4 loads per voxel, all but one served by L1/L2.
typedef unsigned short us8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int64_t reflect1(int64_t i, int64_t n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i); }
__global__ __launch_bounds__(256) void k_morph_gradient_3(const uint16_t *__restrict__ in, int64_t dz, int64_t dy, int64_t dx,
                                                          uint16_t *__restrict__ out) {
    const int64_t cpr = dx / 8, total = dz * dy * cpr;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t ch = t % cpr, q = t / cpr, y = q % dy, z = q / dy;
        const int64_t x0 = ch * 8;
        unsigned mx[8], mn[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            mx[j] = 0u;
            mn[j] = 0xffffu;
        }
#pragma unroll
        for (int c = -1; c <= 1; c++)
#pragma unroll
            for (int b = -1; b <= 1; b++) {
                const uint16_t *row = in + (reflect1(z + c, dz) * dy + reflect1(y + b, dy)) * dx;
                const us8_t v = *reinterpret_cast<const us8_t *>(row + x0);
                unsigned e[10];
                e[0] = row[x0 > 0 ? x0 - 1 : 0];               // reflect: index -1 -> 0
                e[9] = row[x0 + 8 < dx ? x0 + 8 : dx - 1];     //          index dx -> dx - 1
#pragma unroll
                for (int j = 0; j < 8; j++) e[j + 1] = v[j];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const unsigned a = e[j], m = e[j + 1], p = e[j + 2];
                    const unsigned hi3 = a > m ? (a > p ? a : p) : (m > p ? m : p);
                    const unsigned lo3 = a < m ? (a < p ? a : p) : (m < p ? m : p);
                    mx[j] = hi3 > mx[j] ? hi3 : mx[j];
                    mn[j] = lo3 < mn[j] ? lo3 : mn[j];
                }
            }
        us8_t r;
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = (unsigned short)(mx[j] - mn[j]);
        *reinterpret_cast<us8_t *>(out + (z * dy + y) * dx + x0) = r;
    }
}

// The same 3x3x3 gradient walking along z (round 6): a lane owns 8 consecutive voxels of one (y, chunk) column over a segment of
// slices and keeps the 3x3 in-slice maximum / minimum of three consecutive slices in registers -- a slice's rows are loaded once
// per segment instead of three times (three 16-byte loads per output chunk instead of nine, a third of the compares).  Same bits.
constexpr int MG_SEG = 32;
__device__ __forceinline__ void mg_slice(const uint16_t *__restrict__ in, int64_t zz, int64_t y, int64_t x0, int64_t dy, int64_t dx,
                                         unsigned *mx, unsigned *mn) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        mx[j] = 0u;
        mn[j] = 0xffffu;
    }
#pragma unroll
    for (int b = -1; b <= 1; b++) {
        const uint16_t *row = in + (zz * dy + reflect1(y + b, dy)) * dx;
        const us8_t v = *reinterpret_cast<const us8_t *>(row + x0);
        unsigned e[10];
        e[0] = row[x0 > 0 ? x0 - 1 : 0];
        e[9] = row[x0 + 8 < dx ? x0 + 8 : dx - 1];
#pragma unroll
        for (int j = 0; j < 8; j++) e[j + 1] = v[j];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned a = e[j], m = e[j + 1], p = e[j + 2];
            const unsigned hi3 = a > m ? (a > p ? a : p) : (m > p ? m : p);
            const unsigned lo3 = a < m ? (a < p ? a : p) : (m < p ? m : p);
            mx[j] = hi3 > mx[j] ? hi3 : mx[j];
            mn[j] = lo3 < mn[j] ? lo3 : mn[j];
        }
    }
}
__global__ __launch_bounds__(256) void k_morph_gradient_3_walk(const uint16_t *__restrict__ in, int64_t dz, int64_t dy, int64_t dx,
                                                               uint16_t *__restrict__ out) {
    const int64_t cpr = dx / 8;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= dy * cpr) return;
    const int64_t y = t / cpr, x0 = (t - y * cpr) * 8;
    const int64_t z0 = (int64_t)blockIdx.y * MG_SEG, z1 = z0 + MG_SEG < dz ? z0 + MG_SEG : dz;
    unsigned amx[8], amn[8], bmx[8], bmn[8], cmx[8], cmn[8];
    mg_slice(in, reflect1(z0 - 1, dz), y, x0, dy, dx, amx, amn);
    mg_slice(in, z0, y, x0, dy, dx, bmx, bmn);
    for (int64_t z = z0; z < z1; z++) {
        mg_slice(in, reflect1(z + 1, dz), y, x0, dy, dx, cmx, cmn);
        us8_t r;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned hi = amx[j] > bmx[j] ? (amx[j] > cmx[j] ? amx[j] : cmx[j]) : (bmx[j] > cmx[j] ? bmx[j] : cmx[j]);
            const unsigned lo = amn[j] < bmn[j] ? (amn[j] < cmn[j] ? amn[j] : cmn[j]) : (bmn[j] < cmn[j] ? bmn[j] : cmn[j]);
            r[j] = (unsigned short)(hi - lo);
            amx[j] = bmx[j];
            amn[j] = bmn[j];
            bmx[j] = cmx[j];
            bmn[j] = cmn[j];
        }
        *reinterpret_cast<us8_t *>(out + (z * dy + y) * dx + x0) = r;
    }
}

// ---- watershed merge -------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t ws_merge1(uint8_t m, uint8_t t, int overwrite) {
    if (overwrite) return t == 1 ? 253 : 0;
    const bool sel = m == 0 || m == 2 || m == 253;
    if (t == 2 && sel) m = 2;
    if (t == 1 && sel) m = 253;
    return m;
}
typedef unsigned char uc16_t __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_ws_merge(uint8_t *__restrict__ mask, const uint8_t *__restrict__ tmp, int64_t n,
                                                  int overwrite, int vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        const int64_t nc = n / 16;
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += stride) {
            const uc16_t t = reinterpret_cast<const uc16_t *>(tmp)[c];
            uc16_t m = reinterpret_cast<uc16_t *>(mask)[c];
#pragma unroll
            for (int i = 0; i < 16; i++) m[i] = ws_merge1(m[i], t[i], overwrite);
            reinterpret_cast<uc16_t *>(mask)[c] = m;
        }
    }
    for (int64_t i = (vec ? (n / 16) * 16 : 0) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        mask[i] = ws_merge1(mask[i], tmp[i], overwrite);
}

// ---- masked statistics: count, sum, sum of squares as exact integers ---------------------------------------------
__global__ __launch_bounds__(256) void k_masked_stats(const int16_t *__restrict__ img, const uint8_t *__restrict__ sel,
                                                      int64_t n, unsigned long long *__restrict__ acc) {
    long long cnt = 0, s = 0, s2 = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (sel[i]) {
            const long long v = img[i];
            cnt++;
            s += v;
            s2 += v * v;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o, 64);
        s += __shfl_xor(s, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    // one set of atomics per WORKGROUP of a capped grid: millions of same-address atomics (one per wave of a
    // voxel-sized grid) took 15 ms at 512^3
    __shared__ long long sh[4][3];
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6][0] = cnt;
        sh[threadIdx.x >> 6][1] = s;
        sh[threadIdx.x >> 6][2] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long c4 = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];
        if (c4) {
            atomicAdd(&acc[0], (unsigned long long)c4);
            atomicAdd(&acc[1], (unsigned long long)(sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1])); // two's complement sum
            atomicAdd(&acc[2], (unsigned long long)(sh[0][2] + sh[1][2] + sh[2][2] + sh[3][2]));
        }
    }
}
// count, sum, sum of squares, min, max of img where mask > 127 (Slice.calc_image_density, slice_.py:2284-2297)
__global__ __launch_bounds__(256) void k_density(const int16_t *__restrict__ img, const uint8_t *__restrict__ mask, int64_t n,
                                                 unsigned long long *__restrict__ acc, int *__restrict__ mm, int vec) {
    long long cnt = 0, s = 0, s2 = 0;
    int lo = 32767, hi = -32768;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (vec) { // 8 voxels per lane: one 16-byte image load + one 8-byte mask load
        typedef short s8v_t __attribute__((ext_vector_type(8)));
        const int64_t nc = n / 8;
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += stride) {
            const unsigned long long m8 = reinterpret_cast<const unsigned long long *>(mask)[c];
            if (!(m8 & 0x8080808080808080ull)) continue; // > 127 <=> top bit set
            const s8v_t v8 = reinterpret_cast<const s8v_t *>(img)[c];
#pragma unroll
            for (int j = 0; j < 8; j++)
                if ((m8 >> (8 * j + 7)) & 1ull) {
                    const int v = v8[j];
                    cnt++;
                    s += v;
                    s2 += (long long)v * v;
                    lo = v < lo ? v : lo;
                    hi = v > hi ? v : hi;
                }
        }
    }
    for (int64_t i = (vec ? (n / 8) * 8 : 0) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (mask[i] > 127) {
            const int v = img[i];
            cnt++;
            s += v;
            s2 += (long long)v * v;
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o, 64);
        s += __shfl_xor(s, o, 64);
        s2 += __shfl_xor(s2, o, 64);
        const int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    __shared__ long long sh[4][3];
    __shared__ int shm[4][2];
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6][0] = cnt;
        sh[threadIdx.x >> 6][1] = s;
        sh[threadIdx.x >> 6][2] = s2;
        shm[threadIdx.x >> 6][0] = lo;
        shm[threadIdx.x >> 6][1] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long c4 = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];
        if (c4) {
            atomicAdd(&acc[0], (unsigned long long)c4);
            atomicAdd(&acc[1], (unsigned long long)(sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1]));
            atomicAdd(&acc[2], (unsigned long long)(sh[0][2] + sh[1][2] + sh[2][2] + sh[3][2]));
            int l = shm[0][0], h = shm[0][1];
            for (int q = 1; q < 4; q++) {
                l = shm[q][0] < l ? shm[q][0] : l;
                h = shm[q][1] > h ? shm[q][1] : h;
            }
            atomicMin(&mm[0], l);
            atomicMax(&mm[1], h);
        }
    }
}

// Slice.do_boolean_op (slice_.py:1906-1916): "selected" here means > 2, the result is 0 / 255
typedef unsigned char bool16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ unsigned char bool_op(int op, unsigned char a, unsigned char b) {
    const bool x = a > 2, y = b > 2;
    const bool r = op == 1 ? (x || y) : op == 2 ? (x != (x && y)) : op == 3 ? (x && y) : (x != y);
    return r ? 255 : 0;
}
__global__ __launch_bounds__(256) void k_mask_boolean(int op, const uint8_t *__restrict__ m1, const uint8_t *__restrict__ m2,
                                                      uint8_t *__restrict__ out, int64_t n, int vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        const int64_t nc = n / 16;
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += stride) {
            const bool16_t a = reinterpret_cast<const bool16_t *>(m1)[c], b = reinterpret_cast<const bool16_t *>(m2)[c];
            bool16_t r;
#pragma unroll
            for (int i = 0; i < 16; i++) r[i] = bool_op(op, a[i], b[i]);
            reinterpret_cast<bool16_t *>(out)[c] = r;
        }
    }
    for (int64_t i = (vec ? (n / 16) * 16 : 0) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = bool_op(op, m1[i], m2[i]);
}

__global__ __launch_bounds__(256) void k_set_where(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int64_t n,
                                                   int value, int fill) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (src[i] == (uint8_t)value) dst[i] = (uint8_t)fill;
}
__global__ __launch_bounds__(256) void k_or_eq(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int64_t n, int value) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (src[i] == (uint8_t)value) dst[i] = 1;
}

// markers.astype("int16" | "int8") (watershed_process.py:39,45,52,57) on the device: the caller's integer array goes up in its
// own dtype and is narrowed / widened here like numpy does it (two's complement truncation), 4 voxels per lane
template <typename S, typename D>
__global__ __launch_bounds__(256) void k_cast_markers(const S *__restrict__ src, D *__restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (i + q < n) dst[i + q] = (D)src[i + q];
    }
}
template <typename D> static int cast_markers_to(int sdt, const void *src, D *dst, int64_t n, hipStream_t st) {
    const dim3 gr(grid_for(n, 4)), bl(256);
    switch (sdt) {
    case IVX_U8: hipLaunchKernelGGL((k_cast_markers<uint8_t, D>), gr, bl, 0, st, (const uint8_t *)src, dst, n); break;
    case IVX_I8: hipLaunchKernelGGL((k_cast_markers<int8_t, D>), gr, bl, 0, st, (const int8_t *)src, dst, n); break;
    case IVX_I16: hipLaunchKernelGGL((k_cast_markers<int16_t, D>), gr, bl, 0, st, (const int16_t *)src, dst, n); break;
    case IVX_U16: hipLaunchKernelGGL((k_cast_markers<uint16_t, D>), gr, bl, 0, st, (const uint16_t *)src, dst, n); break;
    case IVX_I32: hipLaunchKernelGGL((k_cast_markers<int32_t, D>), gr, bl, 0, st, (const int32_t *)src, dst, n); break;
    case IVX_I64: hipLaunchKernelGGL((k_cast_markers<int64_t, D>), gr, bl, 0, st, (const int64_t *)src, dst, n); break;
    default: ivx::set_error("do_watershed: markers of dtype code %d cannot be cast on the device", sdt); return IVX_EINVAL;
    }
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}
static inline size_t marker_size(int dt) {
    switch (dt) {
    case IVX_U8: case IVX_I8: return 1;
    case IVX_I16: case IVX_U16: return 2;
    case IVX_I32: return 4;
    case IVX_I64: return 8;
    default: return 0;
    }
}

} // namespace

extern "C" int ivx_dev_fill_holes(uint8_t *mask, const uint32_t *labels, int64_t n, uint32_t nlabels, uint32_t max_size,
                                  uint32_t *sizes, int *status2 /* device: [0] modified, [1] error */, void *stream) {
    hipStream_t st = ivx::S(stream);
    IVX_HIP(hipMemsetAsync(sizes, 0, ((size_t)nlabels + 1) * 4, st));
    IVX_HIP(hipMemsetAsync(status2, 0, 8, st));
    if (n == 0) return IVX_OK;
    hipLaunchKernelGGL(k_label_hist, dim3(grid_for(n, 4)), dim3(256), 0, st, labels, n, nlabels, sizes, status2 + 1);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_any_small, dim3((unsigned)ivx::cdiv((int64_t)nlabels + 1, 256)), dim3(256), 0, st, sizes, nlabels,
                       max_size, status2);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fill_small, dim3(grid_for(n, 4)), dim3(256), 0, st, mask, labels, n, sizes, max_size, status2);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_fill_holes_automatically(uint8_t *mask, const int64_t shape[3], const int64_t mst[3],
                                            const uint32_t *labels, const int64_t lst[3], uint32_t nlabels,
                                            uint32_t max_size, int *modified) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    *modified = 0;
    void *d_mask, *d_lab, *d_sizes, *d_stat;
    int rc;
    if ((rc = ws_get(WS_OUT, n, &d_mask))) return rc;
    if ((rc = ws_get(WS_IN, n * 4, &d_lab))) return rc;
    if ((rc = ws_get(WS_AUX0, ((size_t)nlabels + 1) * 4, &d_sizes))) return rc;
    if ((rc = ws_get(WS_SMALL, 256, &d_stat))) return rc;
    if ((rc = upload_strided(d_mask, mask, shape, mst, 1, WS_OUT))) return rc;
    if ((rc = upload_strided(d_lab, labels, shape, lst, 4, WS_IN))) return rc;
    if ((rc = ivx_dev_fill_holes((uint8_t *)d_mask, (const uint32_t *)d_lab, (int64_t)n, nlabels, max_size,
                                 (uint32_t *)d_sizes, (int *)d_stat, nullptr)))
        return rc;
    int h[2] = {0, 0};
    IVX_HIP(hipMemcpy(h, d_stat, 8, hipMemcpyDeviceToHost));
    IVX_REQUIRE(h[1] == 0, IVX_ERANGE, "fill_holes: a label exceeds nlabels (the reference panics on sizes[label])");
    *modified = h[0];
    if (h[0]) return download_strided(mask, shape, mst, d_mask, 1, WS_OUT);
    return IVX_OK;
}

// int16 result of np.piecewise; the uint16 form is its .astype("uint16"), i.e. the same 16 bits
static int lut_run(const int16_t *img, int64_t n, double window, double level, int top255, int16_t *out, void *stream) {
    if (n == 0) return IVX_OK;
    hipStream_t st = ivx::S(stream);
    void *tab;
    int rc;
    if ((rc = ivx::ws_get_s(ivx::WS_LUT, st, 65536 * 2, &tab))) return rc;
    hipLaunchKernelGGL(k_lut_table, dim3(256), dim3(256), 0, st, window, level, top255 ? 255.0 : window, (int16_t *)tab);
    IVX_LAUNCH_CHECK();
    const int vec = ((((uintptr_t)img | (uintptr_t)out) & 15) == 0) && n >= 8;
    hipLaunchKernelGGL(k_lut_apply, dim3(grid_for(n, 8)), dim3(256), 0, st, img, n, (const int16_t *)tab, out, vec);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}
extern "C" int ivx_dev_lut_u16(const int16_t *img, int64_t n, double window, double level, int top255, uint16_t *out,
                               void *stream) {
    return lut_run(img, n, window, level, top255, (int16_t *)out, stream);
}
extern "C" int ivx_dev_lut_i16(const int16_t *img, int64_t n, double window, double level, int top255, int16_t *out,
                               void *stream) {
    return lut_run(img, n, window, level, top255, out, stream);
}
extern "C" int ivx_dev_shift_min_u16(const int16_t *img, int64_t n, int imin, uint16_t *out, void *stream) {
    if (n == 0) return IVX_OK;
    hipLaunchKernelGGL(k_shift_min_u16, dim3(grid_for(n, 4)), dim3(256), 0, ivx::S(stream), img, n, imin, out);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}
extern "C" int ivx_dev_morph_gradient_u16(const uint16_t *in, int64_t dz, int64_t dy, int64_t dx, const int size[3],
                                          uint16_t *out, void *stream) {
    for (int a = 0; a < 3; a++)
        IVX_REQUIRE(size[a] >= 1 && (size[a] & 1), IVX_EINVAL,
                    "morphological_gradient: only odd footprint sizes are supported (got %d)", size[a]);
    const int64_t n = dz * dy * dx;
    if (n == 0) return IVX_OK;
    if (size[0] == 3 && size[1] == 3 && size[2] == 3 && dx % 8 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
        // (IVX_MG_WALK=0: a lane per chunk, nine row loads each, as in rounds 2 - 5 -- A/B, tests)
        static const bool walk = []() { const char *e = getenv("IVX_MG_WALK"); return !(e && e[0] == '0'); }();
        if (walk && dz >= 4)
            hipLaunchKernelGGL(k_morph_gradient_3_walk, dim3((unsigned)ivx::cdiv(dy * (dx / 8), 256), (unsigned)ivx::cdiv(dz, MG_SEG)), dim3(256), 0,
                               ivx::S(stream), in, dz, dy, dx, out);
        else
            hipLaunchKernelGGL(k_morph_gradient_3, dim3(grid_for(n / 8)), dim3(256), 0, ivx::S(stream), in, dz, dy, dx, out);
        IVX_LAUNCH_CHECK();
        return IVX_OK;
    }
    hipLaunchKernelGGL(k_morph_gradient<uint16_t>, dim3(grid_for(n)), dim3(256), 0, ivx::S(stream), in, dz, dy, dx, size[0] / 2, size[1] / 2, size[2] / 2, out);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}
extern "C" int ivx_dev_watershed_merge(uint8_t *mask, const uint8_t *tmp, int64_t n, int overwrite, void *stream) {
    if (n == 0) return IVX_OK;
    const int vec = ((((uintptr_t)mask | (uintptr_t)tmp) & 15) == 0) && n >= 16;
    hipLaunchKernelGGL(k_ws_merge, dim3(grid_for(n, 16)), dim3(256), 0, ivx::S(stream), mask, tmp, n, overwrite, vec);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}
extern "C" int ivx_dev_masked_stats_i16(const int16_t *img, const uint8_t *sel, int64_t n, int64_t out3[3], void *stream) {
    void *d_acc;
    int rc = ivx::ws_get_s(ivx::WS_SMALL, ivx::S(stream), 256, &d_acc);
    if (rc) return rc;
    hipStream_t st = ivx::S(stream);
    unsigned long long *acc = (unsigned long long *)((char *)d_acc + 128);
    IVX_HIP(hipMemsetAsync(acc, 0, 24, st));
    if (n) {
        hipLaunchKernelGGL(k_masked_stats, dim3(std::min(grid_for(n, 8), 2048)), dim3(256), 0, st, img, sel, n, acc);
        IVX_LAUNCH_CHECK();
    }
    IVX_HIP(hipMemcpyAsync(out3, acc, 24, hipMemcpyDeviceToHost, st));
    IVX_HIP(hipStreamSynchronize(st));
    return IVX_OK;
}
extern "C" int ivx_dev_or_equal_u8(uint8_t *dst, const uint8_t *src, int64_t n, int value, void *stream) {
    if (n == 0) return IVX_OK;
    hipLaunchKernelGGL(k_or_eq, dim3(grid_for(n, 4)), dim3(256), 0, ivx::S(stream), dst, src, n, value);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_flood_apply_where(uint8_t *dst, const uint8_t *src, int64_t n, int value, int fill, void *stream) {
    if (n == 0) return IVX_OK;
    hipLaunchKernelGGL(k_set_where, dim3(grid_for(n, 4)), dim3(256), 0, ivx::S(stream), dst, src, n, value, fill);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

// host forms used by the watershed_process mirror ---------------------------------------------------------------------
extern "C" int ivx_watershed_prepare(const int16_t *img, const int64_t shape[3], const int64_t strides[3], int use_ww_wl,
                                     double window, double level, const int gradient_size[3] /* NULL = none */,
                                     uint16_t *out) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    const int64_t n = shape[0] * shape[1] * shape[2];
    if (n == 0) return IVX_OK;
    void *d_img, *d_a, *d_b;
    int rc;
    if ((rc = ws_get(WS_IN, (size_t)n * 2, &d_img))) return rc;
    if ((rc = ws_get(WS_AUX0, (size_t)n * 2, &d_a))) return rc;
    if ((rc = upload_strided(d_img, img, shape, strides, 2, WS_IN))) return rc;
    if (use_ww_wl) {
        if ((rc = ivx_dev_lut_u16((const int16_t *)d_img, n, window, level, 0, (uint16_t *)d_a, nullptr))) return rc;
    } else {
        float *mm;
        void *d_small;
        if ((rc = ws_get(WS_SMALL, 256, &d_small))) return rc;
        mm = (float *)d_small;
        if ((rc = ivx_dev_minmax_f32(IVX_I16, d_img, n, mm, nullptr))) return rc;
        float h[2];
        IVX_HIP(hipMemcpy(h, mm, 8, hipMemcpyDeviceToHost));
        if ((rc = ivx_dev_shift_min_u16((const int16_t *)d_img, n, (int)h[0], (uint16_t *)d_a, nullptr))) return rc;
    }
    void *res = d_a;
    if (gradient_size) {
        if ((rc = ws_get(WS_AUX1, (size_t)n * 2, &d_b))) return rc;
        if ((rc = ivx_dev_morph_gradient_u16((const uint16_t *)d_a, shape[0], shape[1], shape[2], gradient_size,
                                             (uint16_t *)d_b, nullptr)))
            return rc;
        res = d_b;
    }
    IVX_HIP(hipDeviceSynchronize());
    IVX_HIP(hipMemcpy(out, res, (size_t)n * 2, hipMemcpyDeviceToHost));
    return IVX_OK;
}

// do_watershed in one host call: the image and the markers go up once, the uint8 labels come back once; the cost /
// gradient image and the labels in the markers' width never cross PCIe (watershed_process.py:19-60 moves them through
// host memory between its numpy / scipy / scikit-image steps).  The markers travel in the CALLER's dtype (any integer
// width, dense or a sub-box view) and are cast to `mdtype` on the device -- the host-side `markers.astype(...)` of the
// reference is a 134 M-element pass on one core at 512^3 --, and the labels land straight in the caller's array (the memmap
// behind `tfile`: its untouched pages are faulted in by the lane threads of copy_d2h, not by one numpy assignment).
extern "C" int ivx_do_watershed_into(const int16_t *img, const int64_t shape[3], const int64_t strides[3], int mk_src_dtype,
                                     const void *markers, const int64_t mk_strides[3], int mdtype, const uint8_t strct[27], int algorithm,
                                     const int gradient_size[3], int use_ww_wl, double window, double level, uint8_t *out_u8,
                                     const int64_t out_strides[3], int64_t stats[16]) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    // IVX_HOST_TIMING=1: where the call's wall time goes (stderr; every mark waits for the device)
    const bool timing = getenv("IVX_HOST_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = now();
    auto mark = [&](const char *what) {
        if (!timing) return;
        (void)hipDeviceSynchronize();
        const auto t = now();
        fprintf(stderr, "ivx do_watershed: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    IVX_REQUIRE(algorithm == 0 || algorithm == 1, IVX_EINVAL, "do_watershed: algorithm must be 0 (Watershed IFT) or 1 (Watershed)");
    IVX_REQUIRE(mdtype == IVX_I16 || mdtype == IVX_I8, IVX_EINVAL, "do_watershed: markers must be cast to int16 or int8");
    IVX_REQUIRE(algorithm == 0 || gradient_size, IVX_EINVAL, "do_watershed: the Watershed branch needs the gradient size");
    const size_t ssz = marker_size(mk_src_dtype);
    IVX_REQUIRE(ssz != 0, IVX_EINVAL, "do_watershed: markers must be an integer array (dtype code %d)", mk_src_dtype);
    const int64_t n = shape[0] * shape[1] * shape[2];
    if (n == 0) return IVX_OK;
    const size_t msz = mdtype == IVX_I16 ? 2 : 1;
    void *d_img, *d_a, *d_b = nullptr, *d_mk, *d_out;
    int rc;
    if ((rc = ws_get(WS_IN, (size_t)n * 2, &d_img))) return rc;
    if ((rc = ws_get(WS_AUX0, (size_t)n * 2, &d_a))) return rc;
    if ((rc = ws_get(WS_AUX2, (size_t)n * msz, &d_mk))) return rc;
    if ((rc = ws_get(WS_OUT, (size_t)n, &d_out))) return rc;
    mark("workspaces");
    if ((rc = upload_strided(d_img, img, shape, strides, 2, WS_IN))) return rc;
    mark("image up");
    if (mk_src_dtype == mdtype) {
        if ((rc = upload_strided(d_mk, markers, shape, mk_strides, msz, WS_AUX2))) return rc;
    } else {
        void *d_raw;
        if ((rc = ws_get(WS_AUX3, (size_t)n * ssz, &d_raw))) return rc;
        if ((rc = upload_strided(d_raw, markers, shape, mk_strides, ssz, WS_AUX3))) return rc;
        rc = mdtype == IVX_I16 ? cast_markers_to<int16_t>(mk_src_dtype, d_raw, (int16_t *)d_mk, n, nullptr)
                               : cast_markers_to<int8_t>(mk_src_dtype, d_raw, (int8_t *)d_mk, n, nullptr);
        if (rc) return rc;
    }
    mark("markers up (+ cast)");
    if (use_ww_wl) {
        if ((rc = ivx_dev_lut_u16((const int16_t *)d_img, n, window, level, 0, (uint16_t *)d_a, nullptr))) return rc;
    } else {
        void *d_small;
        if ((rc = ws_get(WS_SMALL, 256, &d_small))) return rc;
        if ((rc = ivx_dev_minmax_f32(IVX_I16, d_img, n, (float *)d_small, nullptr))) return rc;
        float h[2];
        IVX_HIP(hipMemcpy(h, d_small, 8, hipMemcpyDeviceToHost));
        if ((rc = ivx_dev_shift_min_u16((const int16_t *)d_img, n, (int)h[0], (uint16_t *)d_a, nullptr))) return rc;
    }
    const uint16_t *cost = (const uint16_t *)d_a;
    if (algorithm == 1) { // gradient image first (watershed_process.py:36-38,49-51)
        if ((rc = ws_get(WS_AUX1, (size_t)n * 2, &d_b))) return rc;
        if ((rc = ivx_dev_morph_gradient_u16((const uint16_t *)d_a, shape[0], shape[1], shape[2], gradient_size, (uint16_t *)d_b, nullptr)))
            return rc;
        cost = (const uint16_t *)d_b;
        rc = ivx_dev_watershed_sk(cost, mdtype, d_mk, shape[0], shape[1], shape[2], strct, nullptr, nullptr, (uint8_t *)d_out, nullptr, stats,
                                  nullptr);
    } else {
        rc = ivx_dev_watershed_ift(cost, mdtype, d_mk, shape[0], shape[1], shape[2], strct, nullptr, (uint8_t *)d_out, nullptr, stats, nullptr);
    }
    if (rc != IVX_OK) return rc;
    IVX_HIP(hipDeviceSynchronize());
    mark("cost image + flood");
    rc = download_strided(out_u8, shape, out_strides, d_out, 1, WS_OUT);
    mark("labels down");
    return rc;
}

extern "C" int ivx_do_watershed(const int16_t *img, const int64_t shape[3], const int64_t strides[3], int mdtype, const void *markers,
                                const uint8_t strct[27], int algorithm, const int gradient_size[3], int use_ww_wl, double window,
                                double level, uint8_t *out_u8, int64_t stats[16]) {
    IVX_REQUIRE(mdtype == IVX_I16 || mdtype == IVX_I8, IVX_EINVAL, "do_watershed: markers must be int16 or int8");
    const int64_t msz = mdtype == IVX_I16 ? 2 : 1;
    const int64_t mst[3] = {shape[1] * shape[2] * msz, shape[2] * msz, msz};
    const int64_t ost[3] = {shape[1] * shape[2], shape[2], 1};
    return ivx_do_watershed_into(img, shape, strides, mdtype, markers, mst, mdtype, strct, algorithm, gradient_size, use_ww_wl, window, level,
                                 out_u8, ost, stats);
}

extern "C" int ivx_watershed_merge(uint8_t *mask, const int64_t shape[3], const int64_t mst[3], const uint8_t *tmp,
                                   const int64_t tst[3], int overwrite) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    const int64_t n = shape[0] * shape[1] * shape[2];
    if (n == 0) return IVX_OK;
    void *d_m, *d_t;
    int rc;
    if ((rc = ws_get(WS_OUT, (size_t)n, &d_m))) return rc;
    if ((rc = ws_get(WS_IN, (size_t)n, &d_t))) return rc;
    if ((rc = upload_strided(d_m, mask, shape, mst, 1, WS_OUT))) return rc;
    if ((rc = upload_strided(d_t, tmp, shape, tst, 1, WS_IN))) return rc;
    if ((rc = ivx_dev_watershed_merge((uint8_t *)d_m, (const uint8_t *)d_t, n, overwrite, nullptr))) return rc;
    IVX_HIP(hipDeviceSynchronize());
    return download_strided(mask, shape, mst, d_m, 1, WS_OUT);
}


extern "C" int ivx_dev_mask_boolean(int op, const uint8_t *m1, const uint8_t *m2, uint8_t *out, int64_t n, void *stream) {
    IVX_REQUIRE(op >= 1 && op <= 4, IVX_EINVAL, "mask_boolean: op must be 1 (union), 2 (diff), 3 (and) or 4 (xor)");
    IVX_REQUIRE(n >= 0, IVX_EINVAL, "mask_boolean: negative size");
    if (n == 0) return IVX_OK;
    const int vec = ((((uintptr_t)m1 | (uintptr_t)m2 | (uintptr_t)out) & 15) == 0) && n >= 16;
    hipLaunchKernelGGL(k_mask_boolean, dim3(grid_for(n, 16)), dim3(256), 0, ivx::S(stream), op, m1, m2, out, n, vec);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

// acc5 (device): count, sum, sum of squares (int64), then min and max as two int32 packed behind them (40 bytes total)
extern "C" int ivx_dev_masked_density_i16(const int16_t *img, const uint8_t *mask, int64_t n, void *acc5, void *stream) {
    IVX_REQUIRE(n >= 0, IVX_EINVAL, "masked_density: negative size");
    hipStream_t st = ivx::S(stream);
    IVX_HIP(hipMemsetAsync(acc5, 0, 24, st));
    const int init[2] = {32767, -32768};
    IVX_HIP(hipMemcpyAsync((char *)acc5 + 24, init, 8, hipMemcpyHostToDevice, st));
    if (n == 0) return IVX_OK;
    const int vec = ((((uintptr_t)img) & 15) == 0 && (((uintptr_t)mask) & 7) == 0) && n >= 8;
    hipLaunchKernelGGL(k_density, dim3(std::min(grid_for(n, 8), 2048)), dim3(256), 0, st, img, mask, n, (unsigned long long *)acc5,
                       (int *)((char *)acc5 + 24), vec);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

// host forms ---------------------------------------------------------------------------------------------------------
extern "C" int ivx_mask_boolean(int op, const uint8_t *m1, const int64_t st1[3], const uint8_t *m2, const int64_t st2[3],
                                uint8_t *out, const int64_t sto[3], const int64_t shape[3]) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(shape[0] >= 0 && shape[1] >= 0 && shape[2] >= 0, IVX_EINVAL, "mask_boolean: negative shape");
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    if (n == 0) return IVX_OK;
    void *d1, *d2, *d3;
    int rc;
    if ((rc = ws_get(WS_IN, n, &d1))) return rc;
    if ((rc = ws_get(WS_AUX0, n, &d2))) return rc;
    if ((rc = ws_get(WS_OUT, n, &d3))) return rc;
    if ((rc = upload_strided(d1, m1, shape, st1, 1, WS_IN))) return rc;
    if ((rc = upload_strided(d2, m2, shape, st2, 1, WS_AUX0))) return rc;
    if ((rc = ivx_dev_mask_boolean(op, (const uint8_t *)d1, (const uint8_t *)d2, (uint8_t *)d3, (int64_t)n, nullptr))) return rc;
    IVX_HIP(hipDeviceSynchronize());
    return download_strided(out, shape, sto, d3, 1, WS_OUT);
}

// out5: count, sum, sum of squares (as doubles: exact below 2^53), min, max
extern "C" int ivx_masked_density_i16(const int16_t *img, const int64_t ist[3], const uint8_t *mask, const int64_t mst[3],
                                      const int64_t shape[3], double *out5) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(shape[0] >= 0 && shape[1] >= 0 && shape[2] >= 0, IVX_EINVAL, "masked_density: negative shape");
    for (int q = 0; q < 5; q++) out5[q] = 0.0;
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    if (n == 0) return IVX_OK;
    void *d_i, *d_m, *d_a;
    int rc;
    if ((rc = ws_get(WS_IN, n * 2, &d_i))) return rc;
    if ((rc = ws_get(WS_AUX0, n, &d_m))) return rc;
    if ((rc = ws_get(WS_SMALL, 64, &d_a))) return rc;
    if ((rc = upload_strided(d_i, img, shape, ist, 2, WS_IN))) return rc;
    if ((rc = upload_strided(d_m, mask, shape, mst, 1, WS_AUX0))) return rc;
    if ((rc = ivx_dev_masked_density_i16((const int16_t *)d_i, (const uint8_t *)d_m, (int64_t)n, d_a, nullptr))) return rc;
    long long h[4];
    IVX_HIP(hipMemcpy(h, d_a, 32, hipMemcpyDeviceToHost));
    int mm[2];
    memcpy(mm, &h[3], 8);
    out5[0] = (double)h[0];
    out5[1] = (double)h[1];
    out5[2] = (double)h[2];
    out5[3] = h[0] ? (double)mm[0] : 0.0;
    out5[4] = h[0] ? (double)mm[1] : 0.0;
    return IVX_OK;
}
