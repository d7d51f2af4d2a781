/*
 * ivx_oracle_wsz.c -- CPU ORACLE, part 2 for the IFT watershed (test infrastructure).
 *
 * orc_watershed_ift_clean: the algorithm scipy.ndimage.watershed_ift DOCUMENTS (ni_measure.c
 *   NI_WatershedIFT: bucket queue over the max-arc path cost |I(p)-I(v)|, positive labels pushed at the
 *   front of their bucket = LIFO, relabel on strictly smaller cost, neighbours taken by LINEAR index so
 *   that row / plane wrap-around neighbours are accepted) WITHOUT the linked-list defect of the C
 *   source: there `if (p->next || p->prev)` decides whether p sits in a queue, which is false for the
 *   only element of a bucket, so that element is re-queued without being unlinked and the two buckets'
 *   chains get spliced.  orc_watershed_ift (ivx_oracle_ws.c) reproduces the defect and equals live
 *   scipy bit for bit; this function is what the defect-free algorithm yields.  The two agree whenever
 *   orc_watershed_ift_events() reports no deferred pop.
 * Positive markers only (the reference passes 0 / 1 / 2: styles.py:2095-2103, 1950-1956).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_EINVAL (-1)
#define ORC_ENOMEM (-3)

static int ws_offsets(const int64_t dims[3], const uint8_t *strct, int64_t *offs) {
    const int64_t strides[3] = {dims[1] * dims[2], dims[2], 1};
    int n = 0;
    for (int kk = 0; kk < 27; kk++) {
        if (!strct[kk]) continue;
        int cz = kk / 9 - 1, cy = (kk / 3) % 3 - 1, cx = kk % 3 - 1;
        int64_t off = cz * strides[0] + cy * strides[1] + cx * strides[2];
        if (off != 0) offs[n++] = off;
    }
    return n;
}

/* cost_out: optional uint32[size] (0xFFFFFFFF = never reached). */
int orc_watershed_ift_clean(int idt, const void *input, const int64_t shape[3], int mdt, const void *markers,
                            const uint8_t *strct, void *output, uint32_t *cost_out) {
    const int64_t size = shape[0] * shape[1] * shape[2];
    if (size == 0) return ORC_OK;
#define IN(i) (idt == 0 ? (int32_t)((const uint8_t *)input)[i] : (int32_t)((const uint16_t *)input)[i])
#define MK(i) (mdt == 1 ? (int32_t)((const int16_t *)markers)[i] : (int32_t)((const int8_t *)markers)[i])
#define OUT(i) (mdt == 1 ? (int32_t)((int16_t *)output)[i] : (int32_t)((int8_t *)output)[i])
#define SETOUT(i, v) do { if (mdt == 1) ((int16_t *)output)[i] = (int16_t)(v); else ((int8_t *)output)[i] = (int8_t)(v); } while (0)
    int32_t maxval = 0;
    for (int64_t i = 0; i < size; i++) { int32_t v = IN(i); if (v > maxval) maxval = v; if (MK(i) < 0) return ORC_EINVAL; }
    int64_t offs[27];
    const int nn = ws_offsets(shape, strct, offs);
    /* per bucket: intrusive doubly linked stack, correctly unlinked */
    int32_t *cost = (int32_t *)malloc((size_t)size * sizeof(int32_t));
    int64_t *nxt = (int64_t *)malloc((size_t)size * sizeof(int64_t));
    int64_t *prv = (int64_t *)malloc((size_t)size * sizeof(int64_t));
    uint8_t *st = (uint8_t *)calloc((size_t)size, 1); /* 0 idle, 1 queued, 2 done */
    int64_t *head = (int64_t *)malloc(((size_t)maxval + 2) * sizeof(int64_t));
    if (!cost || !nxt || !prv || !st || !head) { free(cost); free(nxt); free(prv); free(st); free(head); return ORC_ENOMEM; }
    for (int32_t c = 0; c <= maxval + 1; c++) head[c] = -1;
#define PUSH(c, i) do { nxt[i] = head[c]; prv[i] = -1; if (head[c] >= 0) prv[head[c]] = (i); head[c] = (i); st[i] = 1; } while (0)
#define UNLINK(c, i) do { if (prv[i] >= 0) nxt[prv[i]] = nxt[i]; else head[c] = nxt[i]; if (nxt[i] >= 0) prv[nxt[i]] = prv[i]; } while (0)
    for (int64_t j = 0; j < size; j++) {
        int32_t l = MK(j);
        SETOUT(j, l);
        if (l != 0) { cost[j] = 0; PUSH(0, j); } else cost[j] = maxval + 1;
    }
    for (int32_t c = 0; c <= maxval; c++) {
        while (head[c] >= 0) {
            int64_t v = head[c];
            UNLINK(c, v);
            st[v] = 2;
            for (int h = 0; h < nn; h++) {
                int64_t p = v + offs[h];
                if (p < 0 || p >= size || st[p] == 2) continue;
                int32_t w = IN(p) - IN(v);
                if (w < 0) w = -w;
                int32_t m = cost[v] > w ? cost[v] : w;
                if (m < cost[p]) {
                    if (st[p] == 1) UNLINK(cost[p], p);
                    cost[p] = m;
                    SETOUT(p, OUT(v));
                    PUSH(m, p);
                }
            }
        }
    }
    if (cost_out)
        for (int64_t i = 0; i < size; i++) cost_out[i] = st[i] == 2 ? (uint32_t)cost[i] : 0xFFFFFFFFu;
    free(cost); free(nxt); free(prv); free(st); free(head);
    return ORC_OK;
#undef PUSH
#undef UNLINK
#undef IN
#undef MK
#undef OUT
#undef SETOUT
}
