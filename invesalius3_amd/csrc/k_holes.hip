// k_holes.hip -- Mask.fill_holes_auto in one device pass: labelling + hole filling without a label volume.
//
// Reference (invesalius/data/mask.py:519-562): imask = ~(matrix > 127); labels, n = scipy.ndimage.label(imask, bstruct,
// uint32); floodfill.fill_holes_automatically(matrix, labels, n, size) (invesalius_rs/src/floodfill.rs:51-94): voxel
// counts per label, and -- if any label has 0 < count <= size -- every voxel whose label's count is <= size becomes 254,
// label 0 (the voxels that are NOT in imask) included (reference quirk Q5).  The result depends on the labels only
// through the component sizes, so the 4-byte label volume (537 MB at 512^3) and scipy's serial labelling -- the step
// that dominates the reference's time -- are replaced by the run-based union-find of k_ccl.hip over the imask BIT plane:
// sizes are sums of run lengths per root, components of at most `size` voxels are painted into a bit plane, one sparse
// pass writes 254.  ivx_fill_holes_automatically (explicit labels, the Rust function's own signature) stays as it is.
#include "ivx_internal.h"

namespace {

__global__ __launch_bounds__(256) void k_holes_popcount(const unsigned long long *__restrict__ plane, int64_t nwords,
                                                        unsigned long long *__restrict__ total) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long n = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) n += (unsigned)__popcll(plane[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(total, n);
}

// status: [0] any small imask component, [1] modified (out); cnt: voxels in imask
__global__ void k_holes_decide(int64_t nvox, const unsigned long long *__restrict__ cnt, uint32_t max_size, int *status) {
    const unsigned long long n0 = (unsigned long long)nvox - *cnt; // label 0: the voxels > 127
    const int zero_small = n0 > 0 && n0 <= (unsigned long long)max_size;
    status[2] = zero_small;
    status[1] = status[0] || zero_small;
}

// lane = one byte of the bit plane = 8 voxels of a row; rows are padded to whole words
__global__ __launch_bounds__(256) void k_holes_apply(uint8_t *__restrict__ mask, const uint8_t *__restrict__ plane, int64_t dz,
                                                     int64_t dy, int64_t dx, int64_t wx, const int *__restrict__ status) {
    if (!status[1]) return;
    const bool zero_small = status[2] != 0;
    const int64_t total = dz * dy * wx * 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const unsigned m = plane[t];
        if (!m && !zero_small) continue;
        const int64_t row = t / (wx * 8), x0 = (t - row * wx * 8) * 8;
        uint8_t *p = mask + row * dx + x0;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            if (x0 + b >= dx) break;
            if ((m >> b) & 1u) p[b] = 254;
            else if (zero_small && p[b] > 127) p[b] = 254;
        }
    }
}

} // namespace

extern "C" int ivx_dev_fill_holes_auto(uint8_t *mask, int64_t dz, int64_t dy, int64_t dx, const uint8_t *strct,
                                       const int64_t sshape[3], uint32_t max_size, int *modified, void *stream) {
    IVX_REQUIRE(dz >= 0 && dy >= 0 && dx >= 0, IVX_EINVAL, "fill_holes_auto: negative shape");
    *modified = 0;
    if (dz == 0 || dy == 0 || dx == 0) return IVX_OK;
    ivx_flood_plan plan = {dz, dy, dx, ivx::cdiv(dx, 64), 0};
    int rc;
    if ((rc = ivx_flood_strct_bits(strct, sshape, &plan.strct_bits))) return rc;
    IVX_REQUIRE(ivx::ccl_supported(plan.strct_bits), IVX_EINVAL,
                "fill_holes_auto: the structuring element must be symmetric and hold both x neighbours (4/8, 6/18/26)");
    hipStream_t st = ivx::S(stream);
    const int64_t nwords = dz * dy * plan.wx;
    void *ws;
    if ((rc = ivx::ws_get_s(ivx::WS_HOLES, st, (size_t)nwords * 16 + 256, &ws))) return rc;
    uint64_t *cand = (uint64_t *)ws, *small = cand + nwords;
    int *status = (int *)(small + nwords);                        // [0] any small, [1] modified, [2] label 0 small
    unsigned long long *cnt = (unsigned long long *)(status + 4); // voxels in imask
    IVX_HIP(hipMemsetAsync(small, 0, (size_t)nwords * 8 + 64, st)); // plane + status + counter
    // imask = ~(mask > 127)  <=>  0 <= mask <= 127
    if ((rc = ivx_dev_flood_candidates(&plan, IVX_U8, mask, 0.0, 127.0, nullptr, 0, 0.0, cand, stream))) return rc;
    if ((rc = ivx::ccl_small_components(&plan, cand, small, max_size, status, cand, st))) return rc;
    const int64_t blocks = ivx::cdiv(nwords, 256);
    hipLaunchKernelGGL(k_holes_popcount, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st,
                       (const unsigned long long *)cand, nwords, cnt);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_holes_decide, dim3(1), dim3(1), 0, st, dz * dy * dx, cnt, max_size, status);
    IVX_LAUNCH_CHECK();
    const int64_t nbytes = nwords * 8, ablocks = ivx::cdiv(nbytes, 256);
    hipLaunchKernelGGL(k_holes_apply, dim3((unsigned)(ablocks < 16384 ? ablocks : 16384)), dim3(256), 0, st, mask,
                       (const uint8_t *)small, dz, dy, dx, plan.wx, status);
    IVX_LAUNCH_CHECK();
    uint32_t seq, got[3];
    if ((rc = ivx::mailbox_publish(status, 3, st, &seq))) return rc;
    if ((rc = ivx::mailbox_wait(seq, st, got, 3))) return rc;
    *modified = got[1] != 0;
    return IVX_OK;
}

// host form: mask = the (strided) uint8 view matrix[1:,1:,1:] or a (1,h,w) slice view; edited in place
extern "C" int ivx_fill_holes_auto(uint8_t *mask, const int64_t shape[3], const int64_t mst[3], const uint8_t *strct,
                                   const int64_t sshape[3], uint32_t max_size, int *modified) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(shape[0] >= 0 && shape[1] >= 0 && shape[2] >= 0, IVX_EINVAL, "fill_holes_auto: negative shape");
    *modified = 0;
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    if (n == 0) return IVX_OK;
    void *d_mask;
    int rc;
    if ((rc = ws_get(WS_OUT, n, &d_mask))) return rc;
    if ((rc = upload_strided(d_mask, mask, shape, mst, 1, WS_OUT))) return rc;
    if ((rc = ivx_dev_fill_holes_auto((uint8_t *)d_mask, shape[0], shape[1], shape[2], strct, sshape, max_size, modified, nullptr)))
        return rc;
    if (*modified) return download_strided(mask, shape, mst, d_mask, 1, WS_OUT);
    return IVX_OK;
}
