/*
 * ivx_oracle_ws.c -- CPU ORACLE for the watershed floods (test infrastructure).
 *
 * The reference calls third-party code here (invesalius/data/watershed_process.py:36-57):
 *   - scipy.ndimage.watershed_ift (scipy 1.14.0 pinned by the reference's uv.lock; scipy 1.15.3 is
 *     importable in this image) -> orc_watershed_ift restates scipy's published algorithm
 *     (ndimage/src/ni_measure.c, NI_WatershedIFT: bucket queue over max-arc path cost, positive
 *     labels pushed at the FRONT of a bucket, negative at the back, relabel on strictly smaller
 *     cost) and is PINNED against the live scipy in tests/test_oracle_watershed.py.
 *   - skimage.segmentation.watershed (scikit-image 0.24.0 pinned by the reference; 0.18.3 sits under
 *     /opt/conda in this image): restated in ivx_oracle_wssk.c and PINNED to that compiled kernel
 *     (tests/golden/watershed_sk.npz, tests/golden/make_golden_sk.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_EINVAL (-1)
#define ORC_ENOMEM (-3)

typedef struct ws_el {
    int64_t index;
    int32_t cost;
    struct ws_el *next, *prev;
    uint8_t done;
} ws_el;

/* input: uint8 (idt=0) or uint16 (idt=3), C-contiguous, rank 3 (use shape[0]=1 for 2-D).
 * markers/output: int16 (mdt=1) or int8 (mdt=4), C-contiguous.  strct: 3x3x3 uint8. */
/* flags (optional, one byte per voxel): 1 = popped LATE (from a bucket above its cost), 2 = reachable but never popped,
 * 4 = the defect's trigger (sole element of a bucket re-queued without being unlinked), 8 = popped twice.
 * lvl (optional, 3 x 65536 counters): pops / late pops / triggers per bucket level. */
static int ws_ift_impl(int idt, const void *input, const int64_t shape[3], int mdt, const void *markers,
                       const uint8_t *strct, void *output, int64_t *ev, uint8_t *flags, int64_t *lvl) {
    const int64_t dims[3] = {shape[0], shape[1], shape[2]};
    const int64_t size = dims[0] * dims[1] * dims[2];
    const int64_t strides[3] = {dims[1] * dims[2], dims[2], 1};
    if (size == 0) return ORC_OK;
#define IN(i) (idt == 0 ? (int32_t)((const uint8_t *)input)[i] : (int32_t)((const uint16_t *)input)[i])
#define MK(i) (mdt == 1 ? (int32_t)((const int16_t *)markers)[i] : (int32_t)((const int8_t *)markers)[i])
#define OUT(i) (mdt == 1 ? (int32_t)((int16_t *)output)[i] : (int32_t)((int8_t *)output)[i])
#define SETOUT(i, v) do { if (mdt == 1) ((int16_t *)output)[i] = (int16_t)(v); else ((int8_t *)output)[i] = (int8_t)(v); } while (0)
    int32_t maxval = 0;
    for (int64_t i = 0; i < size; i++) { int32_t v = IN(i); if (v > maxval) maxval = v; }
    ws_el *temp = (ws_el *)malloc((size_t)size * sizeof(ws_el));
    ws_el **first = (ws_el **)calloc((size_t)maxval + 1, sizeof(ws_el *));
    ws_el **last = (ws_el **)calloc((size_t)maxval + 1, sizeof(ws_el *));
    if (!temp || !first || !last) { free(temp); free(first); free(last); return ORC_ENOMEM; }
    for (int64_t jj = 0; jj < size; jj++) {
        int32_t label = MK(jj);
        SETOUT(jj, label);
        temp[jj].index = jj;
        temp[jj].done = 0;
        if (label != 0) {
            temp[jj].cost = 0;
            if (!first[0]) {
                first[0] = &temp[jj]; temp[jj].next = NULL; temp[jj].prev = NULL; last[0] = first[0];
            } else if (label > 0) { /* object markers: front of the queue */
                temp[jj].next = first[0]; temp[jj].prev = NULL; first[0]->prev = &temp[jj]; first[0] = &temp[jj];
            } else { /* background markers: back of the queue */
                temp[jj].next = NULL; temp[jj].prev = last[0]; last[0]->next = &temp[jj]; last[0] = &temp[jj];
            }
        } else {
            temp[jj].cost = maxval + 1; temp[jj].next = NULL; temp[jj].prev = NULL;
        }
    }
    /* neighbour offsets, structure scanned in raster order, centre skipped */
    int64_t nstrides[27];
    int nneigh = 0;
    for (int kk = 0; kk < 27; kk++) {
        if (!strct[kk]) continue;
        int cz = kk / 9 - 1, cy = (kk / 3) % 3 - 1, cx = kk % 3 - 1;
        int64_t off = cz * strides[0] + cy * strides[1] + cx * strides[2];
        if (off != 0) nstrides[nneigh++] = off;
    }
    for (int32_t jj = 0; jj <= maxval; jj++) {
        while (first[jj]) {
            ws_el *v = first[jj];
            if (ev) { if (v->done) ev[2]++; else if (v->cost != jj) ev[1]++; }
            if (flags) { if (v->done) flags[v->index] |= 8; else if (v->cost != jj) flags[v->index] |= 1; }
            if (lvl) { lvl[jj]++; if (!v->done && v->cost != jj) lvl[65536 + jj]++; }
            first[jj] = first[jj]->next;
            if (first[jj]) first[jj]->prev = NULL;
            v->prev = NULL; v->next = NULL;
            v->done = 1;
            for (int hh = 0; hh < nneigh; hh++) {
                int64_t v_index = v->index, p_index = v->index + nstrides[hh];
                int outside = 0;
                /* scipy's extent test decomposes the LINEAR index, so it only rejects indices
                 * outside [0,size): row/plane wrap-around neighbours are accepted (faithful). */
                int64_t idx = p_index;
                for (int qq = 0; qq < 3; qq++) {
                    int64_t cc = idx / strides[qq];
                    if (cc < 0 || cc >= dims[qq]) { outside = 1; break; }
                    idx -= cc * strides[qq];
                }
                if (p_index < 0) outside = 1;
                if (outside) continue;
                ws_el *p = &temp[p_index];
                if (p->done) continue;
                int32_t pval = IN(p_index), vval = IN(v_index);
                int32_t wvp = pval - vval;
                if (wvp < 0) wvp = -wvp;
                int32_t pcost = p->cost;
                int32_t max = v->cost > wvp ? v->cost : wvp;
                if (max < pcost) {
                    p->cost = max;
                    int32_t label = OUT(v_index);
                    SETOUT(p_index, label);
                    if (pcost <= maxval && !(p->next || p->prev) && first[pcost] == p) {
                        if (ev) ev[0]++;
                        if (flags) flags[p_index] |= 4;
                        if (lvl) lvl[2 * 65536 + jj]++;
                    }
                    if (p->next || p->prev) {
                        ws_el *prev = p->prev, *next = p->next;
                        /* the splice goes wrong exactly when p's recorded neighbours are not its neighbours in the list it
                         * is being taken out of: a predecessor that does not point at p (ev[4]) or a successor that does not
                         * point back at p (ev[5]) -- only possible after the defect's trigger left p in two lists at once */
                        if (ev && flags) { if (prev && prev->next != p) ev[4]++; if (next && next->prev != p) ev[5]++; }
                        if (first[pcost] == p) first[pcost] = next;
                        if (last[pcost] == p) last[pcost] = prev;
                        if (prev) prev->next = next;
                        if (next) next->prev = prev;
                    }
                    if (label < 0) {
                        p->prev = last[max]; p->next = NULL;
                        if (last[max]) last[max]->next = p;
                        last[max] = p;
                        if (!first[max]) first[max] = p;
                    } else {
                        p->next = first[max]; p->prev = NULL;
                        if (first[max]) first[max]->prev = p;
                        first[max] = p;
                        if (!last[max]) last[max] = p;
                    }
                }
            }
        }
    }
    if (ev || flags)
        for (int64_t i = 0; i < size; i++) if (!temp[i].done && temp[i].cost <= maxval) { if (ev) ev[3]++; if (flags) flags[i] |= 2; }
    free(temp); free(first); free(last);
    return ORC_OK;
#undef IN
#undef MK
#undef OUT
#undef SETOUT
}

int orc_watershed_ift(int idt, const void *input, const int64_t shape[3], int mdt, const void *markers,
                      const uint8_t *strct, void *output) {
    return ws_ift_impl(idt, input, shape, mdt, markers, strct, output, NULL, NULL, NULL);
}

/* Same flood, plus what scipy's linked-list defect did on this input:
 * ev[0] = a bucket's only element re-queued without being unlinked (the defect's trigger),
 * ev[1] = elements popped from a bucket that is not their cost's (processed LATE: the only way the defect
 *         changes labels), ev[2] = elements popped twice (harmless), ev[3] = reachable elements never popped. */
int orc_watershed_ift_events(int idt, const void *input, const int64_t shape[3], int mdt, const void *markers,
                             const uint8_t *strct, void *output, int64_t ev[4]) {
    ev[0] = ev[1] = ev[2] = ev[3] = 0;
    return ws_ift_impl(idt, input, shape, mdt, markers, strct, output, ev, NULL, NULL);
}

/* The same, plus WHERE: per-voxel event flags and per-level counters (tools/ift_defect_confinement.py). */
int orc_watershed_ift_trace(int idt, const void *input, const int64_t shape[3], int mdt, const void *markers,
                            const uint8_t *strct, void *output, int64_t ev[6], uint8_t *flags, int64_t *lvl) {
    ev[0] = ev[1] = ev[2] = ev[3] = ev[4] = ev[5] = 0;
    return ws_ift_impl(idt, input, shape, mdt, markers, strct, output, ev, flags, lvl);
}
