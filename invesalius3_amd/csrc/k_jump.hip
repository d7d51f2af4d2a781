// k_jump.hip -- 3-D jump flooding (Voronoi owners + distance to the owning site).
//
// Reference: jump_flooding_internal, invesalius_rs/src/floodfill.rs:298-507 (binding floodfill_py.rs:262-275, wrapper
// invesalius_rs/__init__.py:76-80; exported, no caller in the reference tree).  floor(log2(max_dim)) passes; pass s reads
// the arrays of pass s-1 only (double-buffered like the Rust code), each voxel looks at 26 taps at per-axis offsets
// (size / 2) >> s in z-major order and adopts a tap's site when it is STRICTLY nearer (float32, sqrt of the sum of
// squares in (dz, dy, dx) order) -- or unconditionally while it has no owner.  normalize: sites move to the integer
// centroid of their cells (exact int64 sums), distances are recomputed and divided by the cell's maximum.
// Everything is order-independent, so the GPU result is bit-identical: float32 sqrt / divide are correctly rounded
// (hipcc default) and -ffp-contract=off keeps the sum of squares unfused.
//
//   k_jf_seed     lane = site
//   k_jf_step     lane = voxel; 26 gathers from the previous owner array (rows at +-offset: coalesced), site table via L2
//   k_jf_accum    normalize: per-site count + coordinate sums; a wave whose lanes share one owner adds once
//   k_jf_redist   distance to the moved site + per-site maximum (float bits are ordered for d >= 0: atomicMax on uint)
//   k_jf_divide   dist /= max
#include <array>
#include <map>
#include <utility>
#include <vector>

#include "ivx_internal.h"

namespace {

struct JShape {
    int64_t sz, sy, sx;
};

__global__ void k_jf_seed(const int32_t *__restrict__ sites, int64_t nsites, JShape g, int32_t *__restrict__ owners,
                          float *__restrict__ dist) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsites) return;
    const int32_t z = sites[3 * i], y = sites[3 * i + 1], x = sites[3 * i + 2];
    if (z < 0 || y < 0 || x < 0 || z >= g.sz || y >= g.sy || x >= g.sx) return;
    // two sites on one voxel: the Rust loop lets the LAST one win -- seeding is serial there; see ivx_dev_jump_flooding
    owners[((int64_t)z * g.sy + y) * g.sx + x] = (int32_t)i + 1;
    dist[((int64_t)z * g.sy + y) * g.sx + x] = 0.0f;
}

__global__ __launch_bounds__(256) void k_jf_step(const int32_t *__restrict__ oc, const float *__restrict__ dc,
                                                 int32_t *__restrict__ on, float *__restrict__ dn,
                                                 const int32_t *__restrict__ sites, int64_t nsites, JShape g, int64_t oz,
                                                 int64_t oy, int64_t ox) {
    const int64_t n = g.sz * g.sy * g.sx;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += stride) {
        const int64_t x = v % g.sx, r = v / g.sx, y = r % g.sy, z = r / g.sy;
        int32_t idx0 = oc[v];
        float best = dc[v];
#pragma unroll
        for (int zi = -1; zi <= 1; zi++)
#pragma unroll
            for (int yi = -1; yi <= 1; yi++)
#pragma unroll
                for (int xi = -1; xi <= 1; xi++) {
                    if (!xi && !yi && !zi) continue;
                    const int64_t tz = z + zi * oz, ty = y + yi * oy, tx = x + xi * ox;
                    if (tz < 0 || ty < 0 || tx < 0 || tz >= g.sz || ty >= g.sy || tx >= g.sx) continue;
                    const int32_t idx1 = oc[(tz * g.sy + ty) * g.sx + tx];
                    if (idx1 <= 0) continue;
                    const int64_t si = (int64_t)idx1 - 1;
                    if (si >= nsites) continue;
                    const float z1 = (float)sites[3 * si], y1 = (float)sites[3 * si + 1], x1 = (float)sites[3 * si + 2];
                    const float dz = (float)z - z1, dy = (float)y - y1, dx = (float)x - x1;
                    const float d1 = sqrtf(dz * dz + dy * dy + dx * dx);
                    if (idx0 > 0) {
                        if (d1 < best) {
                            idx0 = idx1;
                            best = d1;
                        }
                    } else {
                        idx0 = idx1;
                        best = d1;
                    }
                }
        on[v] = idx0;
        dn[v] = best;
    }
}

// acc per site: [count, sum z, sum y, sum x] as u64 / i64.  LDS_SITES > 0: the workgroup accumulates in LDS first and
// flushes only the sites it touched (a few hundred sites over 10^8 voxels would otherwise hammer a handful of words:
// 573 ms at 512^3 with 64 sites, against 31 ms for the nine passes themselves).
constexpr int JF_LDS_SITES = 1024;
template <bool USE_LDS>
__global__ __launch_bounds__(256) void k_jf_accum(const int32_t *__restrict__ owners, int64_t nsites, JShape g,
                                                  unsigned long long *__restrict__ acc) {
    __shared__ unsigned long long s_acc[USE_LDS ? 4 * JF_LDS_SITES : 1];
    if (USE_LDS) {
        for (int i = threadIdx.x; i < 4 * (int)nsites; i += 256) s_acc[i] = 0ull;
        __syncthreads();
    }
    unsigned long long *dst = USE_LDS ? s_acc : acc;
    const int64_t n = g.sz * g.sy * g.sx;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v0 = (int64_t)blockIdx.x * blockDim.x; v0 < n; v0 += stride) { // wave-uniform trip count (ballots below)
        const int64_t v = v0 + threadIdx.x;
        int32_t o = v < n ? owners[v] : 0;
        if (o <= 0 || (int64_t)o - 1 >= nsites) o = 0;
        const int64_t x = v % g.sx, r = v / g.sx, y = r % g.sy, z = r / g.sy;
        // Voronoi cells are compact: most waves see one owner.  Then one lane adds the wave's totals.
        const int32_t o0 = __shfl(o, 0, 64);
        if (__all(o == o0)) {
            if (o0 == 0) continue;
            unsigned long long sz_ = (unsigned long long)z, sy_ = (unsigned long long)y, sx_ = (unsigned long long)x;
#pragma unroll
            for (int k = 32; k > 0; k >>= 1) {
                sz_ += __shfl_xor(sz_, k, 64);
                sy_ += __shfl_xor(sy_, k, 64);
                sx_ += __shfl_xor(sx_, k, 64);
            }
            if ((threadIdx.x & 63) == 0) {
                unsigned long long *a = dst + 4 * ((int64_t)o0 - 1);
                atomicAdd(a, 64ull);
                atomicAdd(a + 1, sz_);
                atomicAdd(a + 2, sy_);
                atomicAdd(a + 3, sx_);
            }
        } else if (o) {
            unsigned long long *a = dst + 4 * ((int64_t)o - 1);
            atomicAdd(a, 1ull);
            atomicAdd(a + 1, (unsigned long long)z);
            atomicAdd(a + 2, (unsigned long long)y);
            atomicAdd(a + 3, (unsigned long long)x);
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * (int)nsites; i += 256)
            if (s_acc[i]) atomicAdd(acc + i, s_acc[i]);
    }
}

__global__ void k_jf_newsites(const unsigned long long *__restrict__ acc, int64_t nsites, int32_t *__restrict__ ns) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsites) return;
    const long long c = (long long)acc[4 * i];
#pragma unroll
    for (int q = 0; q < 3; q++) ns[3 * i + q] = c > 0 ? (int32_t)((long long)acc[4 * i + 1 + q] / c) : 0;
}

template <bool USE_LDS>
__global__ __launch_bounds__(256) void k_jf_redist(const int32_t *__restrict__ owners, float *__restrict__ dist,
                                                   const int32_t *__restrict__ ns, int64_t nsites, JShape g,
                                                   unsigned int *__restrict__ mx) {
    __shared__ unsigned int s_mx[USE_LDS ? JF_LDS_SITES : 1];
    if (USE_LDS) {
        for (int i = threadIdx.x; i < (int)nsites; i += 256) s_mx[i] = 0u;
        __syncthreads();
    }
    unsigned int *dst = USE_LDS ? s_mx : mx;
    const int64_t n = g.sz * g.sy * g.sx;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v0 = (int64_t)blockIdx.x * blockDim.x; v0 < n; v0 += stride) {
        const int64_t v = v0 + threadIdx.x;
        int32_t o = v < n ? owners[v] : 0;
        if (o <= 0 || (int64_t)o - 1 >= nsites) o = 0;
        float d = 0.0f;
        if (o) {
            const int64_t x = v % g.sx, r = v / g.sx, y = r % g.sy, z = r / g.sy;
            const float dz = (float)z - (float)ns[3 * ((int64_t)o - 1)], dy = (float)y - (float)ns[3 * ((int64_t)o - 1) + 1],
                        dx = (float)x - (float)ns[3 * ((int64_t)o - 1) + 2];
            d = sqrtf(dz * dz + dy * dy + dx * dx);
            dist[v] = d;
        }
        const int32_t o0 = __shfl(o, 0, 64);
        unsigned int bits = __float_as_uint(d); // d >= 0: the bit patterns order like the values
        if (__all(o == o0)) {
            if (o0 == 0) continue;
#pragma unroll
            for (int k = 32; k > 0; k >>= 1) {
                const unsigned int t = __shfl_xor(bits, k, 64);
                bits = t > bits ? t : bits;
            }
            if ((threadIdx.x & 63) == 0) atomicMax(dst + ((int64_t)o0 - 1), bits);
        } else if (o) {
            atomicMax(dst + ((int64_t)o - 1), bits);
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < (int)nsites; i += 256)
            if (s_mx[i]) atomicMax(mx + i, s_mx[i]);
    }
}

__global__ __launch_bounds__(256) void k_jf_divide(const int32_t *__restrict__ owners, float *__restrict__ dist,
                                                   const unsigned int *__restrict__ mx, int64_t nsites, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += stride) {
        const int32_t o = owners[v];
        if (o <= 0 || (int64_t)o - 1 >= nsites) continue;
        const float m = __uint_as_float(mx[(int64_t)o - 1]);
        if (m > 0.0f) dist[v] = dist[v] / m;
    }
}

static inline unsigned grid_for(int64_t n) {
    const int64_t b = ivx::cdiv(n, 256);
    return (unsigned)(b < 1 ? 1 : (b < 65536 ? b : 65536));
}

} // namespace

// dist / owners: dense [sz][sy][sx] device arrays, updated in place; sites: nsites x (z, y, x) int32 on the HOST.
extern "C" int ivx_dev_jump_flooding(float *dist, int32_t *owners, const int64_t shape[3], const int32_t *sites_host,
                                     int64_t nsites, int normalize, void *stream) {
    using namespace ivx;
    IVX_REQUIRE(shape && shape[0] >= 0 && shape[1] >= 0 && shape[2] >= 0 && nsites >= 0, IVX_EINVAL, "jump_flooding: bad shape");
    JShape g = {shape[0], shape[1], shape[2]};
    const int64_t n = g.sz * g.sy * g.sx;
    if (nsites == 0 || n == 0) return IVX_OK; // floodfill.rs:310-312
    IVX_REQUIRE(dist && owners && sites_host, IVX_EINVAL, "jump_flooding: NULL argument");
    IVX_REQUIRE(nsites < 0x7fffffffll, IVX_EINVAL, "jump_flooding: too many sites");
    hipStream_t st = S(stream);
    void *d_on, *d_dn, *d_small;
    int rc;
    if ((rc = ws_get_s(WS_AUX0, st, (size_t)n * 4, &d_on))) return rc;
    if ((rc = ws_get_s(WS_AUX1, st, (size_t)n * 4, &d_dn))) return rc;
    // small block: sites | moved sites | acc (4 x u64 per site) | max bits
    const size_t sb = (size_t)nsites * 12, off_ns = (sb + 255) & ~(size_t)255, off_acc = (off_ns + sb + 255) & ~(size_t)255,
                 off_mx = off_acc + (size_t)nsites * 32, tot = off_mx + (size_t)nsites * 4;
    if ((rc = ws_get_s(WS_SMALL, st, tot, &d_small))) return rc;
    int32_t *d_sites = (int32_t *)d_small, *d_ns = (int32_t *)((char *)d_small + off_ns);
    unsigned long long *d_acc = (unsigned long long *)((char *)d_small + off_acc);
    unsigned int *d_mx = (unsigned int *)((char *)d_small + off_mx);
    IVX_HIP(hipMemcpyAsync(d_sites, sites_host, sb, hipMemcpyHostToDevice, st));
    // Seeding is serial in the reference, so of several sites on one voxel the LAST one owns it: keep only that one
    // among the seeding lanes (the site table itself stays complete -- owners may refer to any site).
    {
        std::vector<int32_t> seed(sites_host, sites_host + 3 * nsites);
        std::map<std::array<int32_t, 3>, int64_t> last;
        bool dup = false;
        for (int64_t i = 0; i < nsites; i++) {
            std::array<int32_t, 3> k = {seed[3 * i], seed[3 * i + 1], seed[3 * i + 2]};
            auto it = last.find(k);
            if (it != last.end()) dup = true;
            last[k] = i;
        }
        if (dup) { // rare: seed the duplicates on the host-determined winner only, by masking the losers out of range
            void *d_seed;
            if ((rc = ws_get_s(WS_AUX2, st, sb, &d_seed))) return rc;
            for (int64_t i = 0; i < nsites; i++) {
                std::array<int32_t, 3> k = {seed[3 * i], seed[3 * i + 1], seed[3 * i + 2]};
                if (last[k] != i) seed[3 * i] = -1;
            }
            IVX_HIP(hipMemcpyAsync(d_seed, seed.data(), sb, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_jf_seed, dim3(grid_for(nsites)), dim3(256), 0, st, (const int32_t *)d_seed, nsites, g, owners, dist);
            IVX_LAUNCH_CHECK();
            IVX_HIP(hipStreamSynchronize(st)); // `seed` is host memory about to go out of scope
        } else {
            hipLaunchKernelGGL(k_jf_seed, dim3(grid_for(nsites)), dim3(256), 0, st, (const int32_t *)d_sites, nsites, g, owners, dist);
            IVX_LAUNCH_CHECK();
        }
    }
    int64_t max_dim = g.sx > g.sy ? g.sx : g.sy;
    if (g.sz > max_dim) max_dim = g.sz;
    int n_steps = 0;
    if (max_dim > 1)
        while ((max_dim >> (n_steps + 1)) > 0) n_steps++;
    int64_t ox = g.sx / 2, oy = g.sy / 2, oz = g.sz / 2;
    int32_t *oc = owners, *on = (int32_t *)d_on;
    float *dc = dist, *dn = (float *)d_dn;
    for (int s = 0; s < n_steps; s++) {
        hipLaunchKernelGGL(k_jf_step, dim3(grid_for(n)), dim3(256), 0, st, (const int32_t *)oc, (const float *)dc, on, dn,
                           (const int32_t *)d_sites, nsites, g, oz, oy, ox);
        IVX_LAUNCH_CHECK();
        std::swap(oc, on);
        std::swap(dc, dn);
        ox /= 2; oy /= 2; oz /= 2;
    }
    if (normalize) {
        IVX_HIP(hipMemsetAsync(d_acc, 0, (size_t)nsites * 32 + (size_t)nsites * 4, st)); // acc and max are adjacent
        // workgroup-level accumulation while the site table fits in LDS; a capped grid keeps the flush small
        const bool lds = nsites <= JF_LDS_SITES;
        const unsigned ng = lds ? (grid_for(n) < 2048u ? grid_for(n) : 2048u) : grid_for(n);
        if (lds) hipLaunchKernelGGL(k_jf_accum<true>, dim3(ng), dim3(256), 0, st, (const int32_t *)oc, nsites, g, d_acc);
        else hipLaunchKernelGGL(k_jf_accum<false>, dim3(ng), dim3(256), 0, st, (const int32_t *)oc, nsites, g, d_acc);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_jf_newsites, dim3(grid_for(nsites)), dim3(256), 0, st, (const unsigned long long *)d_acc, nsites, d_ns);
        IVX_LAUNCH_CHECK();
        if (lds) hipLaunchKernelGGL(k_jf_redist<true>, dim3(ng), dim3(256), 0, st, (const int32_t *)oc, dc, (const int32_t *)d_ns,
                                    nsites, g, d_mx);
        else hipLaunchKernelGGL(k_jf_redist<false>, dim3(ng), dim3(256), 0, st, (const int32_t *)oc, dc, (const int32_t *)d_ns,
                                nsites, g, d_mx);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_jf_divide, dim3(grid_for(n)), dim3(256), 0, st, (const int32_t *)oc, dc, (const unsigned int *)d_mx,
                           nsites, n);
        IVX_LAUNCH_CHECK();
    }
    if (oc != owners) { // an odd number of passes left the result in the work buffers
        IVX_HIP(hipMemcpyAsync(owners, oc, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        IVX_HIP(hipMemcpyAsync(dist, dc, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    }
    return IVX_OK;
}

// Host form: numpy-style strided float32 / int32 arrays updated in place.
extern "C" int ivx_jump_flooding(float *dist, const int64_t dstrides[3], int32_t *owners, const int64_t ostrides[3],
                                 const int64_t shape[3], const int32_t *sites, int64_t nsites, int normalize) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(shape && shape[0] >= 0 && shape[1] >= 0 && shape[2] >= 0, IVX_EINVAL, "jump_flooding: bad shape");
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    if (nsites == 0 || n == 0) return IVX_OK;
    void *d_d, *d_o;
    int rc;
    if ((rc = ws_get(WS_IN, n * 4, &d_d))) return rc;
    if ((rc = ws_get(WS_OUT, n * 4, &d_o))) return rc;
    if ((rc = upload_strided(d_d, dist, shape, dstrides, 4, WS_IN))) return rc;
    if ((rc = upload_strided(d_o, owners, shape, ostrides, 4, WS_OUT))) return rc;
    if ((rc = ivx_dev_jump_flooding((float *)d_d, (int32_t *)d_o, shape, sites, nsites, normalize, nullptr))) return rc;
    IVX_HIP(hipDeviceSynchronize());
    if ((rc = download_strided(dist, shape, dstrides, d_d, 4, WS_IN))) return rc;
    return download_strided(owners, shape, ostrides, d_o, 4, WS_OUT);
}
