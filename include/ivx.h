/*
 * ivx.h -- C ABI of libivx.so: the MI355X (gfx950) implementation of the InVesalius dense-voxel
 * hot path (threshold / region growing / watershed masks, MIP-family projections, marching cubes)
 * and of the stages SURVEY.md 8(f) ranks next to it (indexed surface, largest region, area / volume,
 * context-aware smoothing, view-matrix resampling, 3-D mask editing, mask measurements).
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes plain pointers, sizes and
 * byte strides (no torch / numpy / VTK types) and returns an int status (IVX_OK or a negative
 * IVX_E* code; ivx_last_error() gives the text).  Each declaration cites the reference interface it
 * replaces (paths relative to the invesalius3 checkout).  INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add; invesalius3_amd/ is that stub, complete.
 *
 * Two layers:
 *   ivx_dev_*   operate on DEVICE pointers (dense C-order [z][y][x] arrays already resident in HBM)
 *               on a caller-supplied HIP stream (void* == hipStream_t, NULL = default stream).
 *               Used by the resident pipeline (bench.py, the multi-GPU slab driver).
 *   ivx_*       operate on HOST pointers with byte strides (numpy arrays / np.memmap views such as
 *               mask.matrix[1:,1:,1:]); they stage through HBM, run the same kernels and write the
 *               result in place into the caller-owned output, exactly like the PyO3 functions.
 *
 * There is NO CPU fallback anywhere in this library: without a HIP device every compute entry
 * point returns IVX_EHIP.
 *
 * Conventions (identical to the reference): arrays are [z][y][x]; seeds are (x,y,z);
 * axis 0/1/2 = AXIAL/CORONAL/SAGITAL = reduce over z/y/x; strct is the 3x3x3 (or 1x3x3 ...) uint8
 * structuring element from scipy.ndimage.generate_binary_structure.
 */
#ifndef IVX_H
#define IVX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVX_OK 0
#define IVX_EINVAL (-1) /* bad dtype / shape / argument        -> Python TypeError / ValueError   */
#define IVX_ERANGE (-2) /* seed or label out of bounds          -> IndexError (Rust: index panic)  */
#define IVX_ENOMEM (-3) /* host or device allocation failed     -> MemoryError                     */
#define IVX_EDOM (-4)   /* NumCast failure                      -> ValueError (Rust: unwrap panic) */
#define IVX_EHIP (-5)   /* HIP runtime error / no device        -> RuntimeError                    */

/* dtype codes (image dtypes accepted by the reference's ImageTypes3 enum, invesalius_rs/src/types.rs:4-70) */
#define IVX_U8 0
#define IVX_I16 1
#define IVX_F64 2
#define IVX_U16 3
#define IVX_F32 4 /* mesh vertices */
#define IVX_I32 5 /* label volumes (LabelsTypes3) */
#define IVX_I64 6
#define IVX_I8 7 /* watershed markers (watershed_process.py:57 casts to int8) */

/* projection ops for ivx_*mip_reduce (numpy .max/.min/.mean, invesalius/data/slice_.py:885-889) */
#define IVX_MIP_MAX 0
#define IVX_MIP_MIN 1
#define IVX_MIP_MEAN 2
#define IVX_MIP_SUM 3 /* exact int64 sums: what a Z-sharded MeanIP all-reduces before dividing */

/* ------------------------------------------------------------------------------------------------
 * runtime
 * ---------------------------------------------------------------------------------------------- */
int ivx_version(void);
const char *ivx_last_error(void);
int ivx_device_count(int *count);
int ivx_set_device(int device);
int ivx_device_synchronize(void);
int ivx_device_name(char *buf, size_t buflen);
int ivx_malloc(void **dptr, size_t nbytes);
int ivx_free(void *dptr);
int ivx_memset(void *dptr, int value, size_t nbytes, void *stream);
int ivx_memcpy_h2d(void *dst, const void *src, size_t nbytes);
int ivx_memcpy_d2h(void *dst, const void *src, size_t nbytes);
int ivx_memcpy_d2d(void *dst, const void *src, size_t nbytes, void *stream);
/* the sharded flood's vote words (invesalius3_amd/parallel.py, slab_region_grow; no counterpart upstream -- the reference has no
 * multi-GPU code): votes[0] = what ivx_comm_exchange_vote all-reduces, votes[1] = the words this rank's last OR gained.
 * _set: both words by a one-thread kernel; _read: both words to the host through the pinned mailbox (no stream
 * synchronisation), after which votes[0] <- votes[1] and votes[1] <- 0 on the device -- staged for the next round's all-reduce and
 * for ivx_dev_flood_or_planes_acc */
int ivx_dev_vote_set(int32_t *votes, int32_t v0, int32_t v1, void *stream);
int ivx_dev_vote_read(int32_t *votes, int32_t out[2], void *stream);
/* page-locked host memory (hipHostMalloc): arrays kept there cross PCIe at the link's rate instead of through the runtime's
 * bounce buffers; every host pointer of this header may point into it */
int ivx_host_alloc(void **hptr, size_t nbytes);
int ivx_host_free(void *hptr);
int ivx_stream_create(void **stream);
int ivx_stream_create_low_priority(void **stream); /* yields to default-priority streams when both have work */
int ivx_stream_destroy(void *stream);
int ivx_stream_synchronize(void *stream);
/* HIP events on the stream the kernels are launched on (bench.py roofline timing) */
int ivx_event_create(void **event);
int ivx_event_create_sync(void **event); /* for ordering streams (record + ivx_stream_wait_event), not for timing */
int ivx_event_destroy(void *event);
int ivx_event_record(void *event, void *stream);
int ivx_stream_wait_event(void *stream, void *event); /* work queued on `stream` after this call waits for `event` */
int ivx_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on `stop` */
/* free the cached device workspaces the host-level entry points keep between calls */
int ivx_release_workspace(void);

/* ------------------------------------------------------------------------------------------------
 * threshold -> uint8 mask
 *   replaces Slice.do_threshold_to_a_slice / do_threshold_to_all_slices
 *            invesalius/data/slice_.py:1722-1737, 1739-1769            (preserve = 1)
 *        and Slice.SetMaskThreshold whole-volume loop slice_.py:1240-1247 (preserve = 0)
 * mask[v] = (lo <= img[v] <= hi) ? 255 : 0; with preserve, existing 1/2/253/254 are kept.
 * skip_flags (dz bytes, may be NULL): slices with a non-zero flag are left untouched
 * (mask.matrix[n,0,0] != 0 test, slice_.py:1761).
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_threshold_i16(const int16_t *img, int64_t dz, int64_t dy, int64_t dx, int lo, int hi,
                          int preserve, const uint8_t *skip_flags, uint8_t *mask, void *stream);
/* threshold (no preserve rule, no skip flags) that also emits the mask's inside-bit plane (mask >= 127), in the
 * layout of the region-growing / marching-cubes planes (64 x-voxels per uint64).  Needs dx % 64 == 0 and 16-byte
 * aligned buffers (IVX_EINVAL otherwise: use ivx_dev_threshold_i16).  bits: dz*dy*(dx/64) words. */
int ivx_dev_threshold_i16_bits(const int16_t *img, int64_t dz, int64_t dy, int64_t dx, int lo, int hi, uint8_t *mask,
                               uint64_t *bits, void *stream);
/* Host form.  `mask` points at the FULL (dz+1,dy+1,dx+1) matrix of invesalius/data/mask.py:422-431
 * (flag cells in the index-0 planes); strides in bytes.  With honour_flags the per-slice flag
 * mask[n,0,0] is tested and then set to 1 (slice_.py:1761-1767); without it every slice is
 * processed and the flag forced to 1 (slice_.py:1240-1247). */
int ivx_threshold_all_slices(const int16_t *img, const int64_t shape[3], const int64_t img_strides[3],
                             int lo, int hi, int preserve, int honour_flags, uint8_t *mask,
                             const int64_t mask_strides[3]);

/* ------------------------------------------------------------------------------------------------
 * MaxIP / MinIP / MeanIP along an axis
 *   replaces numpy .max/.min/.mean(axis) in Slice.get_image_slice slice_.py:885-889,969-973,1056-1060
 * out is 2-D: axis0 -> (dy,dx), axis1 -> (dz,dx), axis2 -> (dz,dy); dtype == image dtype for
 * max/min, float64 for mean (numpy semantics).
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_mip_reduce(int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx, int axis, int op,
                       void *out, void *stream);
int ivx_mip_reduce(int dtype, const void *vol, const int64_t shape[3], const int64_t strides[3], int axis,
                   int op, void *out, const int64_t out_strides[2]);
/* viewport magnification of a projection: dst (h*f, w*f) = every pixel of src (h, w) repeated f x f times -- what the
 * reference's ray caster renders for axis-aligned rays at SetImageSampleDistance(1/f) (invesalius/data/volume.py:678) */
int ivx_dev_replicate_i16(const int16_t *src, int64_t h, int64_t w, int factor, int16_t *dst, void *stream);

/* ------------------------------------------------------------------------------------------------
 * MIDA, LMIP, fast contour MIP
 *   replaces mida / lmip / fast_countour_mip of invesalius_rs/src/mips.rs:7-279
 *   (bindings mips_py.rs:161-253, wrappers invesalius_rs/__init__.py:91-101)
 * dtype pairs (in -> out): i16->i16, u8->u8, f64->u8 for mida; out dtype == in dtype otherwise.
 * A NumCast failure on any output pixel returns IVX_EDOM (the reference panics).
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_minmax_f32(int dtype, const void *vol, int64_t n, float *minmax2, void *stream);
int ivx_dev_mida(int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx, int axis, float wl, float ww,
                 const float *minmax2 /* device, from ivx_dev_minmax_f32 */, int out_dtype, void *out,
                 int *status /* device int, set to IVX_EDOM on cast failure */, void *stream);
int ivx_dev_lmip(int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx, int axis, double tmin,
                 double tmax, void *out, void *stream);
/* Z-sharded volumes, rays along Z: one slab's share of every ray.  state = 5 doubles per output pixel
 * (dy*dx*5), NULL state_in on the first slab, NULL state_out on the last one (which writes `out`).  kind 0 = LMIP
 * (p0, p1 = tmin, tmax), 1 = MIDA (p0, p1 = wl, ww; minmax2 = min / max of the WHOLE volume, device float32[2]). */
int ivx_dev_rays_z_slab(int kind, int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx, double p0, double p1,
                        const float *minmax2, const double *state_in, double *state_out, int out_dtype, void *out,
                        int *status, void *stream);
/* The contour volume of fast_countour_mip (calc_fcm_intensity, invesalius_rs/src/mips.rs:197-213).  `base^n` is Rust's f32::powf =
 * the platform libm's powf: reproduced bit for bit for glibc (csrc/glibc_powf.h restates its algorithm and tables; checked
 * against libm on 1.5e9 inputs) in the build this machine's glibc selects -- ivx_powf_variant(): 1 = the FMA build x86-64 glibc
 * runs on CPUs with FMA + AVX2, 0 = the plain build (override: IVX_POWF_VARIANT=fma|plain). */
int ivx_powf_variant(void);
/* out[i] = powf(x[i], y[i]) as the contour MIP computes it (device float arrays; variant 1 / 0 as above, -1 = this machine's).
 * Probes of the fused MaxIP's fast power (tests): 2 / 3 = its value and its relative bound. */
int ivx_dev_powf(const float *x, const float *y, float *out, int64_t n, int variant, void *stream);
/* (tests) the bounds the fused contour MaxIP folds for one voxel: d = the wrapped int16 difference along the ray, other = the two
 * across it (int16 pair in a word), n = the exponent in [1, 64]; the reference's float32 value must lie in [lo, hi]. */
int ivx_dev_fcm_bounds(const int32_t *d, const uint32_t *other, float n, float *lo, float *hi, int64_t count, void *stream);
int ivx_dev_fcm_volume(int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx, float n, int axis,
                       void *tmp /* same dtype/shape */, int *status, void *stream);
/* fast_countour_mip_internal with tmip == 0 (invesalius_rs/src/mips.rs:237-247: tmp = contour volume, out = fold_axis max):
 * the contour value is folded into the running maximum as it is computed -- no temp volume -- for int16 rows of whole
 * 16-byte chunks; other inputs materialise the volume in this stream's workspace.  int16 / uint8 images; `out` has the
 * image's dtype and the shape of the projection; *status receives IVX_EDOM where the reference's NumCast would panic. */
int ivx_dev_fcm_maxip(int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx, float n, int axis, void *out,
                      int *status, void *stream);
int ivx_mida(int dtype, const void *img, const int64_t shape[3], const int64_t strides[3], int axis,
             double wl, double ww, int out_dtype, void *out, const int64_t out_strides[2]);
int ivx_lmip(int dtype, const void *img, const int64_t shape[3], const int64_t strides[3], int axis,
             double tmin, double tmax, void *out, const int64_t out_strides[2]);
int ivx_fast_countour_mip(int dtype, const void *img, const int64_t shape[3], const int64_t strides[3],
                          float n, int axis, double wl, double ww, int tmip, void *out,
                          const int64_t out_strides[2]);

/* ------------------------------------------------------------------------------------------------
 * marching cubes == geometry of create_surface_piece
 *   replaces pad_image + converters.to_vtk + vtkImageFlip + vtkContourFilter
 *            invesalius/data/surface_process.py:52-68,100-186; invesalius/data/converters.py:34-101
 * `a` is the piece (mask[roi+1,1:,1:] u8 or image[roi] i16), dense (nz,ny,nx).  The padding
 * (pad_xy, pad_bottom, pad_top, pad_value) and the Y flip are folded into the addressing, never
 * materialised.  Output: triangle soup, 9 float32 per triangle, in iso-major then (k,j,i) raster
 * order of the padded+flipped cell grid.  Two-call protocol: count, then emit into a buffer of at
 * least `count` triangles.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ivx_mc_params {
    int32_t dtype;      /* IVX_U8 / IVX_I16 / IVX_U16 */
    int32_t pad_xy;     /* 1: one pad voxel on both sides in y and x */
    int32_t pad_bottom; /* 1: one pad slice before z=0 */
    int32_t pad_top;    /* 1: one pad slice after the last */
    int32_t vtk_pz;     /* z padding reported to to_vtk (== pad_bottom when fill_border_holes) */
    int32_t niso;       /* 1 or 2 */
    int64_t nz, ny, nx; /* piece shape */
    int64_t roi_start;  /* first image slice of the piece */
    double pad_value;
    double spacing[3]; /* (sx, sy, sz) */
    double iso[2];
} ivx_mc_params;
/* scratch needed by count/emit for this piece (bytes); pass a device buffer of that size */
int ivx_dev_mc_scratch_bytes(const ivx_mc_params *p, size_t *nbytes);
/* classify + per-row-group triangle counts + scan; *ntris (host) receives the total */
int ivx_dev_mc_count(const ivx_mc_params *p, const void *a, void *scratch, int64_t *ntris, void *stream);
/* Queue-only forms: nothing comes back to the host, so ivx_dev_mc_emit can be queued right behind them with the
 * CAPACITY of `tris` as max_tris (the emit kernel reads the real count, and the iso-0 / iso-1 split, on the device and
 * writes min(count, max_tris) triangles).  ivx_dev_mc_total then fetches the count: if it exceeds the capacity, grow the
 * buffer and call ivx_dev_mc_emit again. */
int ivx_dev_mc_count_async(const ivx_mc_params *p, const void *a, void *scratch, void *stream);
int ivx_dev_mc_count_bits_async(const ivx_mc_params *p, const uint64_t *inside_bits, void *scratch, void *stream);
int ivx_dev_mc_total(const ivx_mc_params *p, void *scratch, int64_t *ntris, void *stream);
/* same, from an inside plane (value >= iso[0]) the caller already holds; niso must be 1.  The plane is read IN PLACE by
 * this call and by the ivx_dev_mc_emit / ivx_dev_mc_indexed_* calls that follow on the same scratch: keep it unchanged
 * until they have run. */
int ivx_dev_mc_count_bits(const ivx_mc_params *p, const uint64_t *inside_bits, void *scratch, int64_t *ntris,
                          void *stream);
/* emit; must follow ivx_dev_mc_count with the same params/scratch */
int ivx_dev_mc_emit(const ivx_mc_params *p, const void *a, const void *scratch, float *tris, int64_t max_tris,
                    void *stream);
/* ivx_dev_mc_emit for a uint8 mask (from_binary) whose bytes are KNOWN: v_out outside the inside plane the count ran on
 * (ivx_dev_mc_count_bits), v_sel where `sel_bits` (same layout) has a bit, v_in elsewhere inside -- the state a resident
 * threshold + region growing leaves behind.  No voxel is read: which end of an edge is inside is in the case index, so a
 * triangle costs three bit look-ups instead of six byte gathers; same arithmetic on the same numbers, same soup. */
int ivx_dev_mc_emit_levels(const ivx_mc_params *p, const void *scratch, const uint64_t *sel_bits, double v_out, double v_in,
                           double v_sel, float *tris, int64_t max_tris, void *stream);
/* The whole surface in ONE launch (k_mc_fused): cell words -> active cells -> triangle counts -> output offsets by a decoupled
 * look-back across the workgroups -> triangles, without per-word counts, a scan launch or a triangle list.  One iso-value
 * (from_binary surfaces: surface_process.py:172-186 with the mask at iso 127; two-iso "Default" pieces keep count + emit).
 * inside_bits: the plane "value >= iso[0]" in source coordinates if the caller holds it, else NULL (derived from `a`).  Writes at
 * most max_tris triangles -- same soup, same order, same bits as ivx_dev_mc_count + ivx_dev_mc_emit (measured SLOWER than those at
 * 512^3, equal at 1024^3: an opt-in, IVX_MC_ONE_LAUNCH=1, see csrc/k_mc.hip); ivx_dev_mc_total then
 * returns the count (if it exceeds max_tris, call again with a larger buffer).  The *_levels form reads no voxel: the mask's
 * bytes are v_out outside inside_bits, v_sel where sel_bits has a bit, v_in elsewhere inside (see ivx_dev_mc_emit_levels). */
int ivx_dev_mc_surface(const ivx_mc_params *p, const void *a, const uint64_t *inside_bits /* may be NULL */, void *scratch,
                       float *tris, int64_t max_tris, void *stream);
int ivx_dev_mc_surface_levels(const ivx_mc_params *p, const uint64_t *inside_bits, const uint64_t *sel_bits, double v_out, double v_in,
                              double v_sel, void *scratch, float *tris, int64_t max_tris, void *stream);
/* the list pass of ivx_dev_mc_emit on its own (it needs the counts, not the voxels): queue it early, on the stream the
 * emit will use; the emit that follows with max_tris <= this max_tris skips its own list pass */
int ivx_dev_mc_list(const ivx_mc_params *p, const void *scratch, int64_t max_tris, void *stream);
/* The host form in two halves that share the device work (replaces the count call + emit call of the form below, which upload the
 * piece and count twice): _begin uploads, counts and emits into the library's own output block and returns the count; _fetch,
 * which must be the NEXT host-level call of the process, copies the soup into the array the caller sized from it (IVX_EINVAL when
 * anything came in between: take ivx_marching_cubes then).  Same soup, same order, same bits.
 * Replaces vtkContourFilter::Update of create_surface_piece (invesalius/data/surface_process.py:172-186). */
int ivx_marching_cubes_begin(const ivx_mc_params *p, const void *a, const int64_t strides[3], int64_t *ntris);
int ivx_marching_cubes_fetch(float *tris, int64_t ntris);
/* Host form: strided piece in, soup out.  tris == NULL -> count only.  Returns count in *ntris. */
int ivx_marching_cubes(const ivx_mc_params *p, const void *a, const int64_t strides[3], float *tris,
                       int64_t max_tris, int64_t *ntris);

/* Cross-slab stitch on the device (replaces vtkAppendPolyData + vtkCleanPolyData of join_process_surface,
 * invesalius/data/surface_process.py:229-268, across Z-slabs).  Follows ivx_dev_mc_indexed_emit on the same params / scratch /
 * stream, one iso-value.  Rank r's top point plane and rank r+1's bottom point plane are the same voxels: the vertices both
 * pieces carry there are matched by edge identity (point word, kind, bit), never by float compares.
 *   1. ivx_dev_mc_stitch_top_sig   -> 32 bytes per point word of the top plane; send them to the rank above
 *   2. ivx_dev_mc_stitch_match     bottom plane vs the signature received from below (NULL on the lowest rank):
 *                                  vd[0] = nverts, vd[1] = copies this piece drops (device words); all-gather the pairs
 *   3. ivx_dev_mc_stitch_apply     faces -> global vertex ids in place, kept vertices -> verts_out (nverts - dropped of
 *                                  them, old order; global id of the first = sum over lower ranks of nverts - dropped) */
int ivx_dev_mc_stitch_sig_bytes(const ivx_mc_params *p, size_t *nbytes);
int ivx_dev_mc_stitch_top_sig(const ivx_mc_params *p, const void *scratch, void *sig, void *stream);
int ivx_dev_mc_stitch_match(const ivx_mc_params *p, const void *scratch, const void *nbr_sig, int64_t nverts, uint32_t *vd,
                            void *stream);
int ivx_dev_mc_stitch_apply(const ivx_mc_params *p, const void *scratch, const void *nbr_sig, const uint32_t *vd_all, int rank,
                            const float *verts, int64_t nverts, int32_t *faces, int64_t ntris, float *verts_out, void *stream);
/* ------------------------------------------------------------------------------------------------
 * indexed surface ("point merge"): unique vertices + int32 faces instead of a soup
 *   replaces vtkAppendPolyData + vtkCleanPolyData of join_process_surface
 *            invesalius/data/surface_process.py:229-268
 * A vertex is a grid edge that crosses the iso-surface, so ids come from popcount scans of the crossing bit planes
 * (no hashing, no sorting); verts[faces] is bit-identical to the soup of ivx_dev_mc_emit, in the same triangle
 * order.  Edges that cross exactly AT a grid point (the sample equals the iso-value) share that point's one vertex,
 * so coincident positions are merged exactly as an exact-arithmetic point merge would; faces that thereby collapse
 * (two equal ids) are kept, so the triangle count and order stay those of the soup.  Protocol: ivx_dev_mc_count -> ivx_dev_mc_indexed_count -> ivx_dev_mc_indexed_emit (same params/scratch).
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_mc_indexed_count(const ivx_mc_params *p, const void *a, const void *scratch, int64_t *nverts, void *stream);
int ivx_dev_mc_indexed_emit(const ivx_mc_params *p, const void *a, const void *scratch, float *verts, int64_t max_verts,
                            int32_t *faces, int64_t max_tris, void *stream);
/* The two calls above for a uint8 mask whose bytes are KNOWN -- `v_out` outside the inside plane handed to
 * ivx_dev_mc_count_bits (and as padding), `v_sel` where `sel_bits` has a bit, `v_in` elsewhere inside; v_out < iso < v_in,
 * v_sel -- as a resident pipeline knows them after its threshold and region growing (see ivx_dev_mc_emit_levels): neither
 * the strictly-inside plane nor a vertex reads a voxel.  Same vertices and faces, bit for bit. */
int ivx_dev_mc_indexed_count_levels(const ivx_mc_params *p, const void *scratch, int64_t *nverts, void *stream);
int ivx_dev_mc_indexed_emit_levels(const ivx_mc_params *p, const void *scratch, const uint64_t *sel_bits, double v_out,
                                   double v_in, double v_sel, float *verts, int64_t max_verts, int32_t *faces,
                                   int64_t max_tris, void *stream);
int ivx_marching_cubes_indexed(const ivx_mc_params *p, const void *a, const int64_t strides[3], float *verts,
                               int64_t max_verts, int32_t *faces, int64_t max_tris, int64_t *nverts, int64_t *ntris);

/* ------------------------------------------------------------------------------------------------
 * surface post-processing on the indexed mesh (join_process_surface, invesalius/data/surface_process.py)
 *   ivx_*_mesh_keep_largest     replaces vtkPolyDataConnectivityFilter + SetExtractionModeToLargestRegion
 *                               surface_process.py:376-391.  Regions = triangles connected through shared vertex
 *                               ids; the region with the most triangles is kept (the one whose first triangle comes
 *                               first wins a tie); kept triangles stay in order, vertices are compacted in order.
 *   ivx_*_mesh_mass_properties  replaces vtkMassProperties (GetVolume / GetSurfaceArea) surface_process.py:452-458.
 *                               out8 = {volume, area, vol_x, vol_y, vol_z, kx, ky, kz}; faces == NULL -> `verts` is a
 *                               soup (3 vertices per triangle).
 * Device forms take device pointers; sizes come back on the host.  out_verts/out_faces == NULL -> sizes only.
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_mesh_keep_largest(const float *verts, int64_t nverts, const int32_t *faces, int64_t ntris, float *out_verts,
                              int64_t max_verts, int32_t *out_faces, int64_t max_tris, int64_t *out_nverts,
                              int64_t *out_ntris, int64_t *nregions, void *stream);
int ivx_dev_mesh_mass_properties(const float *verts, const int32_t *faces, int64_t ntris, double *out8, void *stream);
int ivx_mesh_keep_largest(const float *verts, int64_t nverts, const int32_t *faces, int64_t ntris, float *out_verts,
                          int32_t *out_faces, int64_t *out_nverts, int64_t *out_ntris, int64_t *nregions);
int ivx_mesh_mass_properties(const float *verts, int64_t nverts, const int32_t *faces, int64_t ntris, double *out8);
/* The last two filters of join_process_surface (invesalius/data/surface_process.py:396-435), host arrays in and out:
 *   ivx_mesh_fill_holes     replaces vtkFillHolesFilter + SetHoleSize(hole_size) (:396-416): every closed rim of boundary
 *                           edges whose bounding sphere (half the diagonal of the rim's bounding box) has a radius <= hole_size
 *                           is capped by a fan from its centroid -- ONE new point per hole (index nverts + hole), cap triangles
 *                           wound so that the patch continues the surface.  Rims ordered by their smallest directed edge
 *                           3 f + k, triangles inside a rim in walking order from that edge.
 *   ivx_mesh_point_normals  replaces vtkPolyDataNormals (:420-435: FeatureAngle, SplittingOn, AutoOrientNormalsOn,
 *                           ComputeCellNormalsOn): unit cell normals; corners joined across edges whose cell normals' dot
 *                           product exceeds cos_feature_angle form fans, a vertex's smallest fan keeps the point, the others get
 *                           copies appended in (vertex, fan) order; point normal = normalised sum of the fan's cell normals;
 *                           with auto_orient a surface whose signed volume is negative is turned inside out first.
 * Both are two-call: with NULL output arrays the sizes come back (*n_new_verts / *n_new_tris, *out_nverts); the second call
 * passes the capacities in through the same arguments.  PARITY UNPINNED vs VTK 9.3 (third party, not installable here). */
int ivx_mesh_fill_holes(const float *verts, int64_t nverts, const int32_t *faces, int64_t ntris, double hole_size,
                        float *new_verts, int32_t *new_faces, int64_t *n_new_verts, int64_t *n_new_tris);
int ivx_mesh_point_normals(const float *verts, int64_t nverts, const int32_t *faces, int64_t ntris, double cos_feature_angle,
                           int splitting, int auto_orient, float *out_verts, int32_t *out_faces, float *point_normals,
                           float *cell_normals /* may be NULL */, int64_t *out_nverts);

/* ------------------------------------------------------------------------------------------------
 * context-aware smoothing of the indexed surface
 *   replaces context_aware_smoothing           invesalius_rs/src/mesh_py.rs:7-330 -> mesh.rs:27-86
 *            (python: invesalius_rs.Mesh.ca_smoothing / ca_smoothing, invesalius_rs/__init__.py:220-275;
 *             caller: join_process_surface, invesalius/data/surface_process.py:313-317)
 * verts: (nverts,3) IVX_F32 or IVX_F64, smoothed IN PLACE; faces: (ntris,3) int32 (the reference's (M,4) rows minus
 * their leading count column, which must be 3); normals: (ntris,3) float64 cell normals; t / tmax / bmin / n_iters as in
 * the reference ("angle", "max distance", "min weight", "steps").  Optional outputs (NULL to skip): the staircase
 * flags (nverts bytes) and the per-vertex weights (nverts doubles).  Results are bit-identical to a statement-by-
 * statement restatement of mesh.rs (oracle/ivx_oracle_mesh.c), its quirks included.
 *   ivx_*_mesh_propagate_weights  propagate_weights (mesh.rs:204-288) alone, from explicit seed flags
 *   ivx_*_mesh_face_normals       unit cell normals of the triangles (what vtkPolyDataNormals hands the reference)
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_context_aware_smoothing(void *verts, int vdtype, int64_t nverts, const int32_t *faces, int64_t ntris,
                                    const double *normals, double t, double tmax, double bmin, int n_iters,
                                    uint8_t *staircase_out, double *weights_out, void *stream);
int ivx_dev_mesh_propagate_weights(const void *verts, int vdtype, int64_t nverts, const int32_t *faces, int64_t ntris,
                                   const uint8_t *seed_flags, double tmax, double bmin, double *weights, void *stream);
int ivx_dev_mesh_face_normals(const void *verts, int vdtype, const int32_t *faces, int64_t ntris, double *normals,
                              void *stream);
/* host forms; normals == NULL -> computed from the geometry */
int ivx_context_aware_smoothing(void *verts, int vdtype, int64_t nverts, const int32_t *faces, int64_t ntris,
                                const double *normals, double t, double tmax, double bmin, int n_iters,
                                uint8_t *staircase_out, double *weights_out);
int ivx_mesh_propagate_weights(const void *verts, int vdtype, int64_t nverts, const int32_t *faces, int64_t ntris,
                               const uint8_t *seed_flags, double tmax, double bmin, double *weights);
int ivx_mesh_face_normals(const void *verts, int vdtype, int64_t nverts, const int32_t *faces, int64_t ntris,
                          double *normals);

/* ------------------------------------------------------------------------------------------------
 * 3-D mask editing (callers: invesalius/data/mask3d_editor_state.py:176,221,259)
 *   ivx_*_mask_cut       replaces mask_cut        invesalius_rs/src/mask_cut_py.rs:9-69 -> mask_cut.rs:7-61
 *                        out (dz,dy,dx) uint8 edited in place; mask2d (mh,mw) bytes (numpy bool); m, mv 4x4 row-major
 *                        float64 (world->screen, world->camera); edit_mode 0 = include, 1 = exclude
 *   ivx_*_brush_mask     replaces brush_mask_rs   invesalius_rs/src/brush_mask_py.rs:8-28 -> brush_mask.rs:5-71
 *                        orig may be NULL (edit_mode 0 then paints 255)
 *   ivx_*_polygon2mask   replaces polygon2mask_rs invesalius_rs/src/polygon_mask_py.rs:7-27 -> polygon_mask.rs:4-79
 *                        out (w,h) bytes; points (npts,2) float64
 *   ivx_*_count_regions  replaces count_regions   invesalius_rs/src/count_regions_py.rs:9-26 -> count_regions.rs:5-18
 *                        labels IVX_I16 / IVX_I32 / IVX_I64; IVX_ERANGE where the reference panics (label outside
 *                        [0, number_regions])
 * Host forms take numpy-style byte strides.
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_mask_cut(uint8_t *out, int64_t dz, int64_t dy, int64_t dx, double sx, double sy, double sz, double max_depth,
                     const uint8_t *mask2d, int64_t mh, int64_t mw, const double *m, const double *mv, int edit_mode,
                     void *stream);
int ivx_dev_brush_mask(uint8_t *out, const uint8_t *orig, int64_t dz, int64_t dy, int64_t dx, const double spacing[3],
                       const double center[3], double radius, int edit_mode, void *stream);
/* points_dev: device copy read by the kernel; points_host: the same points on the host (bounding box) */
int ivx_dev_polygon2mask(int64_t w, int64_t h, const double *points_dev, const double *points_host, int64_t npts,
                         uint8_t *out, void *stream);
/* counts: device scratch of number_regions + 1 words; status: device int, non-zero after the call = label out of range */
int ivx_dev_count_regions(int ldtype, const void *labels, int64_t n, int64_t number_regions, uint32_t *counts,
                          uint32_t *out, int *status, void *stream);
int ivx_mask_cut(uint8_t *out, const int64_t shape[3], const int64_t out_strides[3], double sx, double sy, double sz,
                 double max_depth, const uint8_t *mask2d, int64_t mh, int64_t mw, const int64_t mask_strides[2],
                 const double *m, const double *mv, int edit_mode);
int ivx_brush_mask(uint8_t *out, const int64_t shape[3], const int64_t out_strides[3], const uint8_t *orig,
                   const int64_t orig_strides[3], const double spacing[3], const double center[3], double radius,
                   int edit_mode);
int ivx_polygon2mask(int64_t w, int64_t h, const double *points, int64_t npts, uint8_t *out);
int ivx_count_regions(int ldtype, const void *labels, const int64_t shape[3], const int64_t label_strides[3],
                      int64_t number_regions, uint32_t *out);

/* ------------------------------------------------------------------------------------------------
 * whole-mask numpy expressions of invesalius/data/slice_.py
 *   ivx_*_mask_boolean        Slice.do_boolean_op slice_.py:1906-1916: out = 255 where op(m1 > 2, m2 > 2) else 0;
 *                             op = 1 union, 2 diff (m1 and not m2), 3 intersection, 4 xor (constants.py:818-821)
 *   ivx_*_masked_density_i16  Slice.calc_image_density slice_.py:2284-2297: count / sum / sum of squares (exact
 *                             integers) / min / max of the int16 image where mask > 127; mean and std are formed
 *                             from them in float64 by the caller
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_mask_boolean(int op, const uint8_t *m1, const uint8_t *m2, uint8_t *out, int64_t n, void *stream);
/* acc5: 32 bytes on the device: int64 count, sum, sum of squares, then int32 min, max */
int ivx_dev_masked_density_i16(const int16_t *img, const uint8_t *mask, int64_t n, void *acc5, void *stream);
int ivx_mask_boolean(int op, const uint8_t *m1, const int64_t st1[3], const uint8_t *m2, const int64_t st2[3], uint8_t *out,
                     const int64_t sto[3], const int64_t shape[3]);
int ivx_masked_density_i16(const int16_t *img, const int64_t img_strides[3], const uint8_t *mask,
                           const int64_t mask_strides[3], const int64_t shape[3], double *out5);

/* ------------------------------------------------------------------------------------------------
 * convolve_non_zero and the mask-area measurement built on it
 *   ivx_*_convolve_non_zero  replaces convolve_non_zero  invesalius_rs/src/transforms_py.rs:51-93
 *                            (float64 volume and kernel; correlation, outside = cval (an int16), zero where the
 *                            volume is zero; every output value bit-identical to the reference's (k, j, i) sum)
 *   ivx_*_mask_area          replaces Slice.calc_image_area  invesalius/data/slice_.py:2296-2322  (mask > 127, the
 *                            7-point exposed-face kernel of the spacing, cval = 1, summed) without materialising
 *                            the float64 volume; spacing = (sx, sy, sz)
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_convolve_non_zero(const double *volume, int64_t sz, int64_t sy, int64_t sx, const double *kernel, int64_t skz,
                              int64_t sky, int64_t skx, int cval, double *out, void *stream);
int ivx_mask_area_scratch_bytes(int64_t sz, int64_t sy, int64_t sx, size_t *nbytes);
/* kernel27_dev: the 27 doubles of the kernel on the device; area_dev: one double on the device */
int ivx_dev_mask_area_u8(const uint8_t *mask, int64_t sz, int64_t sy, int64_t sx, const double *kernel27_dev,
                         double *scratch, double *area_dev, void *stream);
int ivx_convolve_non_zero(const double *volume, const int64_t shape[3], const int64_t vol_strides[3], const double *kernel,
                          const int64_t kshape[3], int cval, double *out);
int ivx_mask_area(const uint8_t *mask, const int64_t shape[3], const int64_t mask_strides[3], const double spacing_xyz[3],
                  double *area);

/* ------------------------------------------------------------------------------------------------
 * seeded region growing
 *   replaces generic_floodfill_threshold          invesalius_rs/src/floodfill.rs:96-166
 *            generic_floodfill_threshold_inplace  invesalius_rs/src/floodfill.rs:168-237
 *   (bindings floodfill_py.rs:137-231, wrappers invesalius_rs/__init__.py:21-54)
 * Device form works on a bit-packed candidate / reached pair (1 bit per voxel, 64 voxels of an
 * x-row per uint64 word, rows padded to a whole word): see DESIGN.md "region growing".
 * ---------------------------------------------------------------------------------------------- */
typedef struct ivx_flood_plan {
    int64_t dz, dy, dx;
    int64_t wx;            /* uint64 words per x-row = ceil(dx/64) */
    uint32_t strct_bits;   /* bit (kk*9+jj*3+ii) set <=> strct[kk][jj][ii] != 0 (3x3x3, centre ignored) */
} ivx_flood_plan;
/* bytes of ONE bit volume (candidate or reached) for this plan */
int ivx_flood_bits_bytes(const ivx_flood_plan *p, size_t *nbytes);
/* scratch for the tile work-list */
int ivx_flood_scratch_bytes(const ivx_flood_plan *p, size_t *nbytes);
/* strct (dims 1..3 each, centre offset dim/2 as floodfill.rs:108-110) -> plan.strct_bits */
int ivx_flood_strct_bits(const uint8_t *strct, const int64_t sshape[3], uint32_t *bits);
/* zero `reached` and the tile work-list in `scratch` */
int ivx_dev_flood_clear(const ivx_flood_plan *p, uint64_t *reached, void *scratch, void *stream);
/* cand[v] = (t0 <= data[v] <= t1) && not blocked(v); barrier_mode 0: nothing blocks, 1: barrier[v] == (uint8)fill
 * blocks (out-of-place form, floodfill.rs:154), 2: data[v] == fill blocks (in-place form, floodfill.rs:225) */
int ivx_dev_flood_candidates(const ivx_flood_plan *p, int dtype, const void *data, double t0, double t1,
                             const uint8_t *barrier, int barrier_mode, double fill, uint64_t *cand, void *stream);
/* reached := seeds (x,y,z triples on the HOST) that are in [t0,t1]; such seeds are also forced into cand
 * (a pre-filled seed still expands, floodfill.rs:121-128).  Out-of-bounds seed -> IVX_ERANGE. */
int ivx_dev_flood_seed(const ivx_flood_plan *p, int dtype, const void *data, double t0, double t1,
                       const int64_t *seeds_xyz, int64_t nseeds, uint64_t *cand, uint64_t *reached,
                       void *scratch, void *stream);
/* grow `reached` inside `cand` to the fix-point; *rounds (host, may be NULL) = global rounds used */
int ivx_dev_flood_run(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, void *scratch,
                      int *rounds, void *stream);
/* ivx_dev_flood_clear + ivx_dev_flood_seed + ivx_dev_flood_run in one call: floodfill.rs:96-166 from `seeds` over `cand`
 * into `reached`, whose previous contents are discarded.  With <= 16 seeds and scipy's 6 / 18 / 26 structure the clearing
 * and the seeding ride on the coarse pass (no separate pass over the plane, two launches fewer); IVX_FLOOD_FUSED=0 runs the
 * three calls one after the other (same bits). */
int ivx_dev_flood_grow(const ivx_flood_plan *p, int dtype, const void *data, double t0, double t1,
                       const int64_t *seeds_xyz, int64_t nseeds, uint64_t *cand, uint64_t *reached, void *scratch,
                       int *rounds, void *stream);
/* The rounds of a flood run in ONE resident launch (k_flood_resident: round boundaries are device-wide barriers; the launch
 * ends at the first empty round) -- IVX_FLOOD_RESIDENT=0 brings back one launch per round.  ivx_dev_flood_grow waits for the
 * launch's last word before it returns; ivx_dev_flood_grow_async returns right behind the launch (*pending = 1; *rounds is
 * not set then), so the stages the caller queues next start the moment the flood ends.  ivx_dev_flood_wait (same p, cand,
 * reached, scratch) then fetches the round count; *late = 1 when the launch had ended early (round cap of a long thin
 * region -> the union-find engine, or a timed-out barrier -> launches per round) and the flood was completed only inside
 * ivx_dev_flood_wait: work queued between the two calls has seen an incomplete `reached` and must be queued again.
 * With nothing pending ivx_dev_flood_wait returns at once and leaves *rounds alone. */
int ivx_dev_flood_grow_async(const ivx_flood_plan *p, int dtype, const void *data, double t0, double t1,
                             const int64_t *seeds_xyz, int64_t nseeds, uint64_t *cand, uint64_t *reached, void *scratch,
                             int *rounds, int *pending, void *stream);
int ivx_dev_flood_wait(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, void *scratch, int *rounds,
                       int *late, void *stream);
/* Gate for background work (e.g. the next stage's mask-independent passes on a second, low-priority stream): the next
 * ivx_dev_flood_run on `scratch` stores `value` to the device word `word` from the first round that starts with fewer
 * than `below_tiles` tiles -- its throughput-bound head is over -- or when it returns, if no round did.
 * ivx_dev_gate_wait parks `stream` behind a one-wave kernel polling the word; it gives up after `timeout_us`. */
int ivx_dev_flood_arm_gate(const void *scratch, uint32_t *word, uint32_t value, uint32_t below_tiles);
/* drop an arm no flood has consumed (gate opened by hand / buffers about to be freed): the arm is keyed by the scratch
 * address, which a later allocation may reuse */
int ivx_dev_flood_disarm_gate(const void *scratch);
int ivx_dev_gate_wait(const uint32_t *word, uint32_t value, uint32_t timeout_us, void *stream);
int ivx_dev_gate_open(uint32_t *word, uint32_t value, void *stream); /* open it by hand (no flood came) */
/* mark every tile of the plan whose z-range touches [z0,z1) dirty (multi-GPU halo re-seeding) */
int ivx_dev_flood_mark_slab(const ivx_flood_plan *p, void *scratch, int64_t z0, int64_t z1, void *stream);
/* multi-GPU slab halo: reached[z] |= plane & cand[z] (plane = the Z-neighbour's boundary plane, dy*wx words);
 * *changed (host) = number of words that gained bits; the tiles touching z are re-marked dirty */
/* whole-plane dst |= src (op 0) / dst &= ~src (op 1) */
int ivx_dev_bits_combine(uint64_t *dst, const uint64_t *src, int64_t nwords, int op, void *stream);
int ivx_dev_flood_or_plane(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, int64_t z,
                           const uint64_t *plane, void *scratch, int *changed, void *stream);
/* both halo planes of a slab in one call (either plane may be NULL): same update per plane, the tiles of the words that
 * gained bits are marked dirty on the device, ONE read-back: *changed (host) = words that gained bits in either plane */
int ivx_dev_flood_or_planes(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, int64_t z_a,
                            const uint64_t *plane_a, int64_t z_b, const uint64_t *plane_b, void *scratch, int *changed,
                            void *stream);
/* the same with NO read-back: the count is left in the caller's DEVICE word, where ivx_comm_exchange_vote's all-reduce
 * reads it in the next round */
int ivx_dev_flood_or_planes_dev(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, int64_t z_a,
                                const uint64_t *plane_a, int64_t z_b, const uint64_t *plane_b, void *scratch,
                                uint32_t *changed_dev, void *stream);
/* the same, the count ADDED to the caller's device word (ivx_dev_vote_read leaves it at zero: no fill per round) */
int ivx_dev_flood_or_planes_acc(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, int64_t z_a,
                                const uint64_t *plane_a, int64_t z_b, const uint64_t *plane_b, void *scratch,
                                uint32_t *changed_dev, void *stream);
/* out[v] = fill where reached (uint8 out), or data[v] = fill (in-place form, dtype of data) */
int ivx_dev_flood_apply(const ivx_flood_plan *p, const uint64_t *reached, int dtype, void *target,
                        double fill, void *stream);
/* both writes of a GUI region-grow in one pass: out[v] = fill and mask[v] = select where reached
 * (floodfill_threshold's out + `mask[out_mask.astype(bool)] = 254`, styles.py:3200-3214) */
int ivx_dev_flood_apply2(const ivx_flood_plan *p, const uint64_t *reached, uint8_t *out, int fill, uint8_t *mask,
                         int select, void *stream);
/* number of reached voxels */
int ivx_dev_flood_count(const ivx_flood_plan *p, const uint64_t *reached, int64_t *count, void *stream);
/* directed floods (floodfill_auto_threshold): six edge planes (+x, -x, +y, -y, +z, -z; bit = SOURCE voxel, layout of the
 * candidate plane, packed back to back) instead of a candidate plane.  ivx_dev_flood_edges_auto builds them for
 * floodfill_py.rs:32-35: from a voxel of value v the flood may step to a neighbour whose value lies in
 * [ceil(v(1-p)), floor(v(1+p))] (f32 products, `as i16`) and whose `out` byte is not `fill`. */
int ivx_flood_edges_bytes(const ivx_flood_plan *p, size_t *nbytes);
int ivx_dev_flood_edges_auto(const ivx_flood_plan *p, const int16_t *data, const uint8_t *out, float pfrac, int fill,
                             uint64_t *edges, void *stream);
/* seeds set unconditionally (floodfill.rs:21, floodfill_py.rs:30); cand may be NULL */
int ivx_dev_flood_seed_forced(const ivx_flood_plan *p, const int64_t *seeds_xyz, int64_t nseeds, uint64_t *cand,
                              uint64_t *reached, void *scratch, void *stream);
/* ivx_dev_flood_run over edge planes */
int ivx_dev_flood_run_edges(const ivx_flood_plan *p, const uint64_t *edges, uint64_t *reached, void *scratch, int *rounds,
                            void *stream);
/* Host forms.  floodfill (invesalius_rs/__init__.py:10 = floodfill_py.rs:88-135, floodfill.rs:5-49): 6-neighbour
 * component of data == v around (x, y, z), the seed filled whatever its value.  floodfill_auto_threshold
 * (invesalius_rs/__init__.py:57-65 = floodfill_py.rs:12-85): int16 data, seeds as (x, y, z) triples. */
int ivx_floodfill(int dtype, const void *data, const int64_t shape[3], const int64_t strides[3], int64_t x, int64_t y,
                  int64_t z, double v, int fill, uint8_t *out, const int64_t out_strides[3]);
int ivx_floodfill_auto_threshold(const int16_t *data, const int64_t shape[3], const int64_t strides[3],
                                 const int64_t *seeds_xyz, int64_t nseeds, float pfrac, int fill, uint8_t *out,
                                 const int64_t out_strides[3]);
/* jump_flooding (invesalius_rs/__init__.py:76-80 = floodfill_py.rs:262-275, floodfill.rs:298-507): Voronoi owners (int32,
 * 1-based site index, 0 = none) and float32 distance to the owning site, floor(log2(max dim)) double-buffered passes of
 * 26 taps; normalize != 0: sites move to their cells' integer centroids, distances are recomputed and divided by the
 * cell maximum.  sites = nsites rows of (z, y, x) int32 in HOST memory.  Device form: dense arrays, in place. */
int ivx_dev_jump_flooding(float *dist, int32_t *owners, const int64_t shape[3], const int32_t *sites_host, int64_t nsites,
                          int normalize, void *stream);
int ivx_jump_flooding(float *dist, const int64_t dist_strides[3], int32_t *owners, const int64_t owner_strides[3],
                      const int64_t shape[3], const int32_t *sites, int64_t nsites, int normalize);
int ivx_floodfill_threshold(int dtype, const void *data, const int64_t shape[3], const int64_t strides[3],
                            const int64_t *seeds_xyz, int64_t nseeds, double t0, double t1, int fill,
                            const uint8_t *strct, const int64_t sshape[3], uint8_t *out,
                            const int64_t out_strides[3]);
int ivx_floodfill_threshold_inplace(int dtype, void *data, const int64_t shape[3], const int64_t strides[3],
                                    const int64_t *seeds_xyz, int64_t nseeds, double t0, double t1,
                                    double fill, const uint8_t *strct, const int64_t sshape[3]);

/* ------------------------------------------------------------------------------------------------
 * fill small holes
 *   replaces fill_holes_automatically_internal invesalius_rs/src/floodfill.rs:51-94
 *   (labels come from scipy.ndimage.label on the host, invesalius/data/mask.py:529-531)
 * *modified = 1 when any label had 0 < size <= max_size (then voxels whose label size <= max_size -> 254,
 * label 0 included: faithful quirk).  A label > nlabels is IVX_ERANGE (the reference panics).
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_fill_holes(uint8_t *mask, const uint32_t *labels, int64_t n, uint32_t nlabels, uint32_t max_size,
                       uint32_t *sizes /* device, nlabels+1 */, int *status2 /* device: [0] modified, [1] error */,
                       void *stream);
int ivx_fill_holes_automatically(uint8_t *mask, const int64_t shape[3], const int64_t mask_strides[3],
                                 const uint32_t *labels, const int64_t label_strides[3], uint32_t nlabels,
                                 uint32_t max_size, int *modified);
/* Mask.fill_holes_auto (invesalius/data/mask.py:519-562) in one call: imask = ~(mask > 127), connected components of
 * imask under `strct` (what scipy.ndimage.label computes there), then the rule above -- without a label volume: the
 * result depends on the labels only through the component sizes.  strct: generate_binary_structure(3, 1|2|3) or a
 * (1,3,3) 4-/8-neighbour element for a (1,h,w) slice.  *modified = the bool the reference's call returns. */
int ivx_dev_fill_holes_auto(uint8_t *mask, int64_t dz, int64_t dy, int64_t dx, const uint8_t *strct, const int64_t sshape[3],
                            uint32_t max_size, int *modified, void *stream);
int ivx_fill_holes_auto(uint8_t *mask, const int64_t shape[3], const int64_t mask_strides[3], const uint8_t *strct,
                        const int64_t sshape[3], uint32_t max_size, int *modified);

/* ------------------------------------------------------------------------------------------------
 * watershed pre- and post-processing (the deterministic parts of do_watershed,
 * invesalius/data/watershed_process.py:19-60)
 *   ivx_dev_lut_u16        get_LUT_value / get_LUT_value_255 (imagedata_utils.py:540-564): np.piecewise on an
 *                          int16 array -> int16 (float64 expression truncated toward zero) -> .astype(uint16)
 *   ivx_dev_shift_min_u16  (image - image.min()).astype("uint16")   (watershed_process.py:47,55)
 *   ivx_dev_morph_gradient_u16  scipy.ndimage.morphological_gradient(size) == max - min filter, mode="reflect"
 *   ivx_dev_watershed_merge     merge rule of styles.py:2147-2152: tmp (labels 0/1/2) into mask
 *   ivx_watershed_prepare  host form: LUT or min-shift, then optional gradient, -> the uint16 cost image
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_lut_u16(const int16_t *img, int64_t n, double window, double level, int top255, uint16_t *out,
                    void *stream);
/* same LUT, int16 out: the image get_LUT_value_255 hands to floodfill_threshold in the "dynamic" / "confidence"
 * region-growing modes with use_ww_wl (invesalius/data/styles.py:3166-3171, 3222-3225) */
int ivx_dev_lut_i16(const int16_t *img, int64_t n, double window, double level, int top255, int16_t *out,
                    void *stream);
int ivx_dev_shift_min_u16(const int16_t *img, int64_t n, int imin, uint16_t *out, void *stream);
int ivx_dev_morph_gradient_u16(const uint16_t *in, int64_t dz, int64_t dy, int64_t dx, const int size[3],
                               uint16_t *out, void *stream);
int ivx_dev_watershed_merge(uint8_t *mask, const uint8_t *tmp, int64_t n, int overwrite, void *stream);
int ivx_watershed_prepare(const int16_t *img, const int64_t shape[3], const int64_t strides[3], int use_ww_wl,
                          double window, double level, const int gradient_size[3] /* NULL = none */, uint16_t *out);
int ivx_watershed_merge(uint8_t *mask, const int64_t shape[3], const int64_t mask_strides[3], const uint8_t *tmp,
                        const int64_t tmp_strides[3], int overwrite);

/* The IFT cost map level by level on bit planes: C[p] = min over paths from a marker of the largest arc |I(a) - I(b)|
 * (6 neighbours, scipy's linear-index neighbourhood) for every voxel of cost < the number of levels done; levels run until
 * `stop_frac` of the voxels are in or `max_levels` are done.  Other voxels keep the caller's value (0xFFFF).  Used by
 * ivx_dev_watershed_ift before its relaxation; needs dx % 64 == 0 and dy % 16 == 0 (IVX_EINVAL otherwise).
 * ivx_dev_sk_cost_levels: the same for the scikit-image branch's cost (largest image VALUE on the path, markers cost their own
 * value, lattice neighbours under `strct`): level c = what the markers of value <= c reach inside {image <= c}, on the
 * region-growing engine; stops after level 0 when that level holds < 5 % of the voxels (no plateau).  Needs dx % 64 == 0 and
 * 16-byte aligned image and markers (IVX_EINVAL otherwise). */
int ivx_dev_sk_cost_levels(const uint16_t *image, int mdtype, const void *markers, int64_t dz, int64_t dy, int64_t dx,
                           const uint8_t strct[27], uint16_t *C, int max_levels, double stop_frac, int *levels_done,
                           int64_t *reached, int64_t *rounds, void *stream);
int ivx_dev_ws_cost_levels(const uint16_t *cost_image, int mdtype, const void *markers, int64_t dz, int64_t dy, int64_t dx,
                           uint16_t *C, int max_levels, double stop_frac, int *levels_done, int64_t *reached,
                           int64_t *rounds, void *stream);
/* ------------------------------------------------------------------------------------------------
 * the marker flood of do_watershed's IFT branch
 *   replaces scipy.ndimage.watershed_ift(tmp_image, markers, bstruct) as called at
 *            invesalius/data/watershed_process.py:44-46,54-57 (3-D) and invesalius/data/styles.py:1958-1983 (one slice,
 *            shape (1, h, w) with the 3x3 structure in strct[9..17])
 * cost: uint16 (the LUT / min-shifted image above), markers int16 or int8 (>= 0; the reference passes 0 / 1 / 2),
 * strct: 3x3x3 uint8, symmetric (generate_binary_structure).  Labels come back in the markers' dtype (what scipy
 * returns) and/or as uint8 (what `mask[:] = tmp_mask` stores, watershed_process.py:58).  Arc weight |I(p)-I(q)|, path
 * cost = largest arc, LIFO inside a cost bucket, neighbours by linear index (row wrap-around like scipy): the
 * defect-free statement of NI_WatershedIFT, see k_wsift.hip.  cost_out (optional): the minimax cost map.
 * stats (optional, host): [0] relaxation rounds, [1] tile visits, [2] non-empty cost levels, [3] time stamps used,
 * [4] marker voxels, [5] entry voxels, [6] tiles, [7] LDS sweeps over all tile visits, [8..12] microseconds of: costs, zones, bucketing, level chain, labels.
 * The host form uploads / downloads dense C-order arrays.
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_watershed_ift(const uint16_t *cost, int mdtype, const void *markers, int64_t dz, int64_t dy, int64_t dx,
                          const uint8_t strct[27], void *out_labels /* markers' dtype, may be NULL */,
                          uint8_t *out_u8 /* may be NULL */, uint16_t *cost_out /* may be NULL */, int64_t stats[16],
                          void *stream);
int ivx_watershed_ift(int idtype /* IVX_U8 | IVX_U16 */, const void *input, const int64_t shape[3], int mdtype,
                      const void *markers, const uint8_t strct[27], void *output, uint16_t *cost_out, int64_t stats[16]);

/* ------------------------------------------------------------------------------------------------
 * the marker flood of do_watershed's scikit-image branch (algorithm == "Watershed", the GUI's default)
 *   replaces skimage.segmentation.watershed(tmp_image, markers, bstruct) as called at
 *            invesalius/data/watershed_process.py:36-39,49-52 (3-D) and invesalius/data/styles.py:1958,1975 (one slice,
 *            shape (1, h, w) with the 3x3 structure in strct[9..17]); no mask, no compactness, no watershed lines.
 * image: uint16 (the gradient of the LUT / min-shifted image; 65535 is refused), markers int16 or int8 (any non-zero
 * label, negative ones included), strct: 3x3x3 uint8, symmetric.  Labels come back in the markers' dtype, as int32
 * (scikit-image's output dtype) and / or as uint8 (what `mask[:] = tmp_mask` stores).  The serial rule -- pop the
 * smallest (image value, age), age = push counter, label at push time -- is evaluated without the heap: k_wssk.hip.
 * Marker voxels of equal image value are taken in raster order (scikit-image's heap takes them in an order that
 * depends on its array layout); stats[6] counts the tied neighbours in that order that carry different labels:
 * 0 = identical to scikit-image by construction.  cost_out (optional): the minimax map of image values.
 * stats (optional, host): [0] relaxation rounds, [1] tile visits, [2] non-empty levels, [3] generations, [4] marker
 * voxels, [5] generation-0 voxels, [6] tied markers of different labels, [7] frontier launches, [8..11] microseconds
 * of: costs, generation 0, level chain, labels, [12] basin relay launches, [13] generation steps, [14] launches that took a run of
 * small levels in one workgroup, [15] tile rounds of levels relaxed tile-wise.
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_watershed_sk(const uint16_t *image, int mdtype, const void *markers, int64_t dz, int64_t dy, int64_t dx,
                         const uint8_t strct[27], void *out_labels /* markers' dtype, may be NULL */,
                         int32_t *out_i32 /* may be NULL */, uint8_t *out_u8 /* may be NULL */,
                         uint16_t *cost_out /* may be NULL */, int64_t stats[16], void *stream);
int ivx_watershed_sk(int idtype /* IVX_U8 | IVX_U16 */, const void *input, const int64_t shape[3], int mdtype,
                     const void *markers, const uint8_t strct[27], int32_t *output, uint16_t *cost_out, int64_t stats[16]);

/* do_watershed (invesalius/data/watershed_process.py:19-60) in one host call: int16 image (dense or a strided view) and
 * markers (dense, int16 / int8) up once, uint8 labels (what `mask[:] = tmp_mask` stores) back once.  algorithm 0 =
 * "Watershed IFT" (LUT or min-shift -> ivx_dev_watershed_ift), 1 = "Watershed" (the same, then the morphological gradient of
 * gradient_size -> ivx_dev_watershed_sk).  stats as for the flood that ran. */
int ivx_do_watershed(const int16_t *img, const int64_t shape[3], const int64_t strides[3], int mdtype, const void *markers,
                     const uint8_t strct[27], int algorithm, const int gradient_size[3] /* NULL for algorithm 0 */,
                     int use_ww_wl, double window, double level, uint8_t *out_u8, int64_t stats[16]);
/* The same, as the Python hook calls it: `markers` in the caller's own integer dtype (IVX_U8 / I8 / I16 / U16 / I32 / I64, dense or a
 * sub-box view: byte strides) and cast on the device to `mdtype` (IVX_I16 | IVX_I8 -- watershed_process.py:39,45,52,57 do
 * `markers.astype("int16" | "int8")` on the host, two's-complement truncation like numpy); the uint8 labels are written through
 * `out_strides` straight into the caller's array -- the memmap of `tfile` (watershed_process.py:21,58) -- so the reference's
 * `mask[:] = tmp_mask` second pass over the volume does not exist. */
int ivx_do_watershed_into(const int16_t *img, const int64_t shape[3], const int64_t strides[3], int mk_src_dtype, const void *markers,
                          const int64_t mk_strides[3], int mdtype, const uint8_t strct[27], int algorithm,
                          const int gradient_size[3] /* NULL for algorithm 0 */, int use_ww_wl, double window, double level,
                          uint8_t *out_u8, const int64_t out_strides[3], int64_t stats[16]);

/* ------------------------------------------------------------------------------------------------
 * confidence-connected region growing support (do_rg_confidence, invesalius/data/styles.py:3220-3251):
 * exact integer count / sum / sum-of-squares of image[sel != 0]; dst[v] = 1 where src[v] == value.
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_masked_stats_i16(const int16_t *img, const uint8_t *sel, int64_t n, int64_t out3[3], void *stream);
int ivx_dev_or_equal_u8(uint8_t *dst, const uint8_t *src, int64_t n, int value, void *stream);
/* dst[v] = fill where src[v] == value  (mask[out_mask.astype(bool)] = 254, styles.py:3214,3249) */
int ivx_dev_flood_apply_where(uint8_t *dst, const uint8_t *src, int64_t n, int value, int fill, void *stream);

/* ------------------------------------------------------------------------------------------------
 * resample a slab through the 4x4 view matrix (re-oriented volumes, the step in front of the projections)
 *   replaces apply_view_matrix_transform  invesalius_rs/src/transforms_py.rs:12-49,95-147
 *            coord_transform               invesalius_rs/src/transforms.rs:9-55
 *            nearest / trilinear / tricubic / Lanczos-4 with single wrap-around  invesalius_rs/src/interpolation.rs
 * m is row-major; orientation 0/1/2 = AXIAL/CORONAL/SAGITAL adds n to z/y/x of the output index; minterpol
 * 0 nearest, 1 trilinear, 2 tricubic, else Lanczos; out has the volume's dtype.  A NumCast failure -> IVX_EDOM.
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_apply_view_matrix_transform(int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx,
                                        const double spacing[3], const double m[16], int64_t n, int orientation,
                                        int minterpol, double cval, void *out, int64_t oz, int64_t oy, int64_t ox,
                                        int *status, void *stream);
int ivx_apply_view_matrix_transform(int dtype, const void *vol, const int64_t shape[3], const int64_t strides[3],
                                    const double spacing[3], const double m[16], int64_t n, int orientation,
                                    int minterpol, double cval, void *out, const int64_t oshape[3],
                                    const int64_t ostrides[3]);

/* ------------------------------------------------------------------------------------------------
 * quality-preset resample of the surface pipeline
 *   replaces imagedata_utils.resize_image_array (invesalius/data/imagedata_utils.py:121-130) =
 *            scipy.ndimage.zoom(image, factor, image.dtype, order=2), applied to image and mask by
 *            SurfaceManager.AddNewActor for the Low / Medium presets (invesalius/data/surface.py:1350-1353)
 * int16 or uint8 volumes, dense C order; oshape[a] = round(ishape[a] * factor) is the caller's (scipy's rule).
 * ---------------------------------------------------------------------------------------------- */
int ivx_zoom_scratch_bytes(const int64_t ishape[3], const int64_t oshape[3], size_t *nbytes);
int ivx_dev_zoom_order2(int dtype, const void *in, const int64_t ishape[3], void *out, const int64_t oshape[3],
                        void *scratch, void *stream);
int ivx_zoom_order2(int dtype, const void *in, const int64_t ishape[3], void *out, const int64_t oshape[3]);

/* ------------------------------------------------------------------------------------------------
 * bench / test input made in HBM (no reference counterpart): a CT-like int16 phantom -- six Gaussian blobs + sinusoid +
 * hashed N(0,25) noise, clipped to [-1024, 3071] -- for the slices [z0, z0 + dz) of a z_total-slice volume; deterministic
 * in (seed, global voxel index), so slabs made by different ranks tile the whole volume.  centres_zyx_sigma: 6 x (cz, cy,
 * cx, sigma) in normalised coordinates.  Used by bench.py --config sharded2048 (BASELINE configs[3]).
 * ---------------------------------------------------------------------------------------------- */
int ivx_dev_synth_volume(int16_t *out, int64_t dz, int64_t dy, int64_t dx, int64_t z0, int64_t z_total, uint32_t seed,
                         const float centres_zyx_sigma[24], void *stream);

/* ------------------------------------------------------------------------------------------------
 * Z-slab communicator: RCCL over xGMI behind the C ABI (one process per GPU).
 *   the reference's decomposition: Z pieces + one overlap slice, invesalius/data/surface.py:1362-1380;
 *   it has no multi-GPU code, so these entry points replace nothing upstream -- they are what the sharded
 *   driver (invesalius3_amd/parallel.py, bench.py --gpus N) binds instead of torch.distributed.
 * librccl.so is dlopen'ed by the first call.  All buffers are DEVICE pointers; every call only enqueues on `stream`.
 *   ivx_comm_unique_id   rank 0 makes the 128-byte id; it reaches the other ranks out of band (a file / env)
 *   ivx_comm_init        collective over all ranks, on the calling process's current device
 *   ivx_comm_exchange    to_down -> rank-1, to_up -> rank+1 and the mirror receives, one group (halo slices, planes)
 *   ivx_comm_exchange_vote   the same followed by an in-place int32 sum all-reduce of `vote` on the same stream
 *   ivx_comm_allreduce   in place; op 0 sum / 1 max / 2 min; IVX_I32 / I64 / F32 / F64 / U8 / I8 (no 16-bit integers)
 *   ivx_comm_allgather   recv = world * nbytes, rank order;   ivx_comm_bcast / send / recv: raw bytes
 * ---------------------------------------------------------------------------------------------- */
int ivx_comm_unique_id(uint8_t id[128]);
int ivx_comm_init(const uint8_t id[128], int rank, int world, void **comm);
int ivx_comm_destroy(void *comm);
int ivx_comm_rank(const void *comm, int *rank, int *world);
int ivx_comm_exchange(void *comm, const void *to_down, void *from_down, const void *to_up, void *from_up, size_t nbytes,
                      void *stream);
int ivx_comm_exchange_vote(void *comm, const void *to_down, void *from_down, const void *to_up, void *from_up,
                           size_t nbytes, int32_t *vote, int nvote, void *stream);
int ivx_comm_allreduce(void *comm, void *buf, size_t count, int dtype, int op, void *stream);
int ivx_comm_allgather(void *comm, const void *send, void *recv, size_t nbytes, void *stream);
int ivx_comm_bcast(void *comm, void *buf, size_t nbytes, int root, void *stream);
int ivx_comm_send(void *comm, const void *buf, size_t nbytes, int peer, void *stream);
int ivx_comm_recv(void *comm, void *buf, size_t nbytes, int peer, void *stream);
/* Every entry point above once, on 4 KB buffers, checked against the analytic answer; the error names the collective that
 * failed.  Collective: every rank calls it.  At world 1, where the entry points never reach RCCL, it drives RCCL directly
 * on the one-rank communicator (all-reduce, all-gather, broadcast, a grouped send + receive to itself). */
int ivx_comm_selftest(void *comm, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* IVX_H */
