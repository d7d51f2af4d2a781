/*
 * ivx_oracle_wssk.c -- CPU ORACLE, the scikit-image branch of the watershed (test infrastructure).
 *
 * orc_watershed_sk restates skimage.segmentation.watershed(image, markers, connectivity=bstruct) as the
 * reference calls it (invesalius/data/watershed_process.py:39,52 and styles.py:1958,1975: no mask, no
 * compactness, no watershed lines).  scikit-image is a third-party dependency (pinned 0.24.0,
 * pyproject.toml:35) that is not under /root/reference; the algorithm below is its published one
 * (segmentation/_watershed.py `watershed`, _watershed_cy.pyx `watershed_raveled`, _shared/heap_general.pxi,
 * morphology/_util.py `_offsets_to_raveled_neighbors`):
 *   - image -> float64, markers -> int32, everything padded by one cell per axis with mask = 0 on the pad;
 *   - neighbour offsets = the non-zero cells of the structure in raster order, sorted (stably) by L1
 *     distance from the centre, centre dropped;
 *   - every marker voxel is pushed in raster order with key (image value, age 0);
 *   - pop the smallest (value, age); each in-mask neighbour whose output is still 0 takes the popped
 *     voxel's label AT PUSH TIME and is pushed with key (its own image value, ++age);
 *   - the queue is a binary heap with strict comparisons; elements that compare equal -- only marker
 *     voxels of equal image value, all of age 0 -- leave it in an order that depends on the heap's array
 *     layout at that moment, i.e. on every push and pop before it.
 * tie_mode 0 reproduces that heap move for move (== the compiled kernel of scikit-image 0.18.3 run in
 * this container, tests/golden/watershed_sk.npz); tie_mode 1 breaks marker ties by raster index instead
 * (a total order: any correct priority queue then gives the same result, and so does the order-free statement
 * csrc/k_wssk.hip implements -- DESIGN.md section 6b).  The two differ only on inputs where equal-valued markers of
 * different labels compete.  tie_mode + 2: the neighbour list reversed (diagnostic: with raster ties the labels do not
 * depend on the neighbour order, with heap ties they do).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_EINVAL (-1)
#define ORC_ENOMEM (-3)

typedef struct {
    double value;
    int64_t age; /* tie_mode 1: markers carry (raster index - size) < 0 instead of 0 */
    int64_t index;
} SkItem;

static inline int sk_smaller(const SkItem *a, const SkItem *b) {
    if (a->value != b->value) return a->value < b->value;
    return a->age < b->age;
}

typedef struct {
    SkItem *d;
    int64_t n, cap;
} SkHeap;

static int sk_push(SkHeap *h, const SkItem *e) {
    if (h->n == h->cap) {
        int64_t nc = h->cap * 2;
        SkItem *nd = (SkItem *)realloc(h->d, (size_t)nc * sizeof(SkItem));
        if (!nd) return -1;
        h->d = nd;
        h->cap = nc;
    }
    int64_t child = h->n++;
    h->d[child] = *e;
    while (child > 0) {
        int64_t parent = (child + 1) / 2 - 1;
        if (!sk_smaller(&h->d[child], &h->d[parent])) break;
        SkItem t = h->d[child];
        h->d[child] = h->d[parent];
        h->d[parent] = t;
        child = parent;
    }
    return 0;
}

static void sk_pop(SkHeap *h, SkItem *dest) {
    *dest = h->d[0];
    h->n--;
    if (h->n == 0) return;
    h->d[0] = h->d[h->n]; /* the last leaf goes to the root (the popped one is dropped) */
    int64_t i = 0;
    for (;;) {
        int64_t smallest = i, l = 2 * i + 1, r = 2 * i + 2;
        if (l >= h->n) break;
        if (sk_smaller(&h->d[l], &h->d[i])) smallest = l;
        if (r < h->n && sk_smaller(&h->d[r], &h->d[smallest])) smallest = r;
        if (smallest == i) break;
        SkItem t = h->d[i];
        h->d[i] = h->d[smallest];
        h->d[smallest] = t;
        i = smallest;
    }
}

/* idt: 0 uint8, 3 uint16, 1 int16 (image); mdt: 1 int16, 4 int8, 2 int32 (markers).  output: int32[size].
 * stats (optional, int64[4]): pushes, pops, largest heap, pops of an age-0 element while another age-0
 * element of the same value was queued (the layout-dependent choices). */
int orc_watershed_sk(int idt, const void *input, const int64_t shape[3], int mdt, const void *markers,
                     const uint8_t *strct, int tie_mode, int32_t *output, int64_t *stats) {
    const int64_t dz = shape[0], dy = shape[1], dx = shape[2];
    const int64_t size = dz * dy * dx;
    if (stats) memset(stats, 0, 4 * sizeof(int64_t));
    if (size == 0) return ORC_OK;
    if (!(idt == 0 || idt == 3 || idt == 1) || !(mdt == 1 || mdt == 4 || mdt == 2)) return ORC_EINVAL;
    const int64_t pz = dz + 2, py = dy + 2, px = dx + 2, psize = pz * py * px;
    const int64_t ps[3] = {py * px, px, 1};

    /* neighbour list: raster order of the structure, stable by L1 distance */
    int64_t offs[27];
    int nn = 0;
    for (int dist = 1; dist <= 3; dist++)
        for (int kk = 0; kk < 27; kk++) {
            if (!strct[kk]) continue;
            int cz = kk / 9 - 1, cy = (kk / 3) % 3 - 1, cx = kk % 3 - 1;
            if (abs(cz) + abs(cy) + abs(cx) != dist) continue;
            offs[nn++] = cz * ps[0] + cy * ps[1] + cx * ps[2];
        }

    double *img = (double *)calloc((size_t)psize, sizeof(double));
    uint8_t *msk = (uint8_t *)calloc((size_t)psize, 1);
    int32_t *out = (int32_t *)calloc((size_t)psize, sizeof(int32_t));
    SkHeap h = {(SkItem *)malloc(1024 * sizeof(SkItem)), 0, 1024};
    if (!img || !msk || !out || !h.d) { free(img); free(msk); free(out); free(h.d); return ORC_ENOMEM; }
    for (int64_t z = 0; z < dz; z++)
        for (int64_t y = 0; y < dy; y++)
            for (int64_t x = 0; x < dx; x++) {
                const int64_t i = (z * dy + y) * dx + x, p = (z + 1) * ps[0] + (y + 1) * ps[1] + (x + 1);
                img[p] = idt == 0 ? (double)((const uint8_t *)input)[i]
                       : idt == 3 ? (double)((const uint16_t *)input)[i] : (double)((const int16_t *)input)[i];
                msk[p] = 1;
                out[p] = mdt == 1 ? (int32_t)((const int16_t *)markers)[i]
                       : mdt == 4 ? (int32_t)((const int8_t *)markers)[i] : ((const int32_t *)markers)[i];
            }
    if (tie_mode & 2) /* diagnostic: neighbour list reversed */
        for (int a = 0, b = nn - 1; a < b; a++, b--) { int64_t t = offs[a]; offs[a] = offs[b]; offs[b] = t; }
    tie_mode &= 1;
    int rc = ORC_OK;
    int64_t pushes = 0, pops = 0, peak = 0, tied = 0;
    /* tie census: how many age-0 elements of each value are queued (only needed for the statistics) */
    int64_t *zero_q = stats ? (int64_t *)calloc(65536 + 32768 + 1, sizeof(int64_t)) : NULL;
#define ZQ(v) zero_q[(int64_t)(v) + 32768]
    for (int64_t p = 0; p < psize; p++) {
        if (!out[p]) continue;
        SkItem e = {img[p], tie_mode ? p - psize : 0, p};
        if (sk_push(&h, &e)) { rc = ORC_ENOMEM; goto done; }
        pushes++;
        if (zero_q) ZQ(img[p])++;
    }
    int64_t age = 1;
    while (h.n > 0) {
        if (h.n > peak) peak = h.n;
        SkItem e;
        sk_pop(&h, &e);
        pops++;
        if (zero_q && e.age <= 0) { if (--ZQ(e.value) > 0) tied++; }
        for (int k = 0; k < nn; k++) {
            const int64_t q = e.index + offs[k];
            if (!msk[q] || out[q]) continue;
            age++;
            out[q] = out[e.index];
            SkItem ne = {img[q], age, q};
            if (sk_push(&h, &ne)) { rc = ORC_ENOMEM; goto done; }
            pushes++;
        }
    }
    for (int64_t z = 0; z < dz; z++)
        for (int64_t y = 0; y < dy; y++)
            memcpy(output + (z * dy + y) * dx, out + (z + 1) * ps[0] + (y + 1) * ps[1] + 1, (size_t)dx * sizeof(int32_t));
    if (stats) { stats[0] = pushes; stats[1] = pops; stats[2] = peak; stats[3] = tied; }
done:
    free(img); free(msk); free(out); free(h.d); free(zero_q);
    return rc;
}
