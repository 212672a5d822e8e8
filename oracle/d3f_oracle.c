/*
 * TEST INFRASTRUCTURE ONLY (oracle).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (d3feat_amd/) never does.
 *
 * Plain-C restatement of the reference's CPU preprocessing algorithms.  Every function cites the
 * reference lines it follows (paths relative to /root/reference).  Parity of THIS file is pinned by
 * tests/test_oracle_vs_ref.py, which compares it bit-for-bit with the reference's own C++ compiled
 * in place (oracle/_ref, see oracle/Makefile) and with the committed vectors under tests/golden/.
 *
 * Build: gcc -std=c11 -O2 -fPIC -shared -ffp-contract=off  (x86-64 baseline: no FMA, like the
 * reference's `g++ -O2` build, tf_custom_ops/compile_op.sh:8-13).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * libstdc++ std::unordered_map<size_t, T> iteration-order model.
 *
 * tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.cpp:47 keeps the voxels in an
 * unordered_map keyed by the voxel index and :81 emits them in iteration order, so the order of the
 * subsampled points IS libstdc++'s node order.  Model (GCC 11 bits/hashtable.h, hashtable_policy.h):
 *   - hash(size_t) = identity, bucket = key % bucket_count;
 *   - one singly linked list of all nodes; bucket[b] = node BEFORE the first node of bucket b;
 *   - _M_insert_bucket_begin: non-empty bucket -> link right after bucket[b]; empty bucket -> link
 *     at the list head and repoint the bucket of the old head to the new node;
 *   - _Prime_rehash_policy (max_load_factor 1): inserting element number bucket_count+1 rehashes
 *     first, to the next table prime >= 2*bucket_count; the first insert allocates 13 buckets;
 *   - _M_rehash_aux(unique): walk the old list in order, re-link each node with the same rule.
 * The growth chain below was produced by oracle/tools/probe_libstdcxx_chain.cpp
 * (std::__detail::_Prime_rehash_policy::_M_next_bkt) and cross-checked against a live map.
 * ------------------------------------------------------------------------------------------------ */
static const uint64_t ORC_CHAIN[] = {13ull, 29ull, 59ull, 127ull, 257ull, 541ull, 1109ull, 2357ull, 5087ull,
    10273ull, 20753ull, 42043ull, 85229ull, 172933ull, 351061ull, 712697ull, 1447153ull, 2938679ull,
    5967347ull, 12117689ull, 24607243ull, 49969847ull, 101473717ull, 206062531ull, 418451333ull,
    849749479ull, 1725587117ull, 3504151727ull};
#define ORC_NCHAIN ((int)(sizeof(ORC_CHAIN) / sizeof(ORC_CHAIN[0])))

typedef struct {
    uint64_t nb;      /* bucket count (0 = nothing allocated yet) */
    int chain_i;      /* index into ORC_CHAIN of nb */
    int64_t* bucket;  /* node index BEFORE first node of bucket; -2 = empty, -1 = before_begin */
    int64_t* next;    /* per node */
    const uint64_t* key;
    int64_t head;     /* before_begin.next */
    int64_t size;
} orc_umap;

static void orc_umap_link(orc_umap* m, int64_t* bucket, uint64_t nb, int64_t n, int64_t* head) {
    uint64_t b = m->key[n] % nb;
    if (bucket[b] != -2) {
        int64_t before = bucket[b];
        if (before == -1) { m->next[n] = *head; *head = n; }
        else { m->next[n] = m->next[before]; m->next[before] = n; }
    } else {
        m->next[n] = *head;
        *head = n;
        if (m->next[n] >= 0) bucket[m->key[m->next[n]] % nb] = n;
        bucket[b] = -1;
    }
}

static void orc_umap_rehash(orc_umap* m, int chain_i) {
    uint64_t nb = ORC_CHAIN[chain_i];
    int64_t* nbk = (int64_t*)malloc(sizeof(int64_t) * nb);
    for (uint64_t i = 0; i < nb; i++) nbk[i] = -2;
    int64_t p = m->head, nhead = -1;
    while (p >= 0) {
        int64_t nx = m->next[p];
        orc_umap_link(m, nbk, nb, p, &nhead);
        p = nx;
    }
    free(m->bucket);
    m->bucket = nbk; m->nb = nb; m->chain_i = chain_i; m->head = nhead;
}

/* Insert node n (key[n] must be new). */
static void orc_umap_insert(orc_umap* m, int64_t n) {
    if (m->nb == 0) orc_umap_rehash(m, 0);
    else if ((uint64_t)m->size + 1 > m->nb) orc_umap_rehash(m, m->chain_i + 1);
    orc_umap_link(m, m->bucket, m->nb, n, &m->head);
    m->size++;
}

/* open-addressing lookup used only to find "is this voxel key already present" quickly;
 * it has no influence on the emitted order. */
typedef struct { uint64_t* k; int64_t* v; uint64_t cap; } orc_find;
static uint64_t orc_mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; return x; }
static int64_t orc_find_or_add(orc_find* f, uint64_t key, int64_t newv) {
    uint64_t h = orc_mix(key) & (f->cap - 1);
    while (f->v[h] >= 0) { if (f->k[h] == key) return f->v[h]; h = (h + 1) & (f->cap - 1); }
    f->k[h] = key; f->v[h] = newv;
    return -1;
}

/*
 * grid_subsampling  --  tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.cpp:5-97
 * (+ cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-105 for features/classes).
 *   pts  float[N*3];  feat float[N*fdim] or NULL;  cls int[N*ldim] or NULL
 *   out_p float[N*3], out_f float[N*fdim], out_c int[N*ldim] (caller-allocated for the worst case M == N)
 * Returns M.
 * Arithmetic, all fp32 exactly as the reference:
 *   min/max corner            cpp_utils/cloud/cloud.cpp:27-66
 *   origin = floor(min*(1/dl))*dl                                   grid_subsampling.cpp:26
 *   NX = (size_t)floor((max.x-origin.x)/dl)+1 (NY likewise)        :29-30
 *   iX = (size_t)floor((p.x-origin.x)/dl) ...; key = iX+NX*iY+NX*NY*iZ   :52-55
 *   point += p in input order (cloud.h:81-87); out = point * (float)(1.0/count)   :83, cloud.h:121-124
 *   features: in-order fp32 sum, then f / (float)count  (true division)     :86-92
 *   classes : per-column histogram; result = LARGEST label id present (max_element over
 *             unordered_map<int,int> compares the pair, key first)            :94 / cpp_wrappers :98-101
 */
int orc_grid_subsampling(const float* pts, int N, const float* feat, int fdim, const int* cls, int ldim,
                         float dl, float* out_p, float* out_f, int* out_c) {
    if (N <= 0) return 0;
    float mn[3] = {pts[0], pts[1], pts[2]}, mx[3] = {pts[0], pts[1], pts[2]};
    for (int i = 0; i < N; i++)
        for (int d = 0; d < 3; d++) {
            float v = pts[3 * i + d];
            if (v < mn[d]) mn[d] = v;
            if (v > mx[d]) mx[d] = v;
        }
    float inv = 1 / dl; /* `1/sampleDl` : int/float -> float division */
    float org[3];
    for (int d = 0; d < 3; d++) org[d] = floorf(mn[d] * inv) * dl;
    uint64_t NX = (uint64_t)floorf((mx[0] - org[0]) / dl) + 1;
    uint64_t NY = (uint64_t)floorf((mx[1] - org[1]) / dl) + 1;

    uint64_t cap = 16;
    while (cap < (uint64_t)N * 2) cap <<= 1;
    orc_find fd;
    fd.cap = cap;
    fd.k = (uint64_t*)malloc(sizeof(uint64_t) * cap);
    fd.v = (int64_t*)malloc(sizeof(int64_t) * cap);
    for (uint64_t i = 0; i < cap; i++) fd.v[i] = -1;

    uint64_t* vkey = (uint64_t*)malloc(sizeof(uint64_t) * N);
    float* acc = (float*)calloc((size_t)N * 3, sizeof(float));
    int* cnt = (int*)calloc(N, sizeof(int));
    float* facc = fdim > 0 ? (float*)calloc((size_t)N * fdim, sizeof(float)) : NULL;
    int* cmax = ldim > 0 ? (int*)malloc(sizeof(int) * (size_t)N * ldim) : NULL;

    orc_umap m;
    memset(&m, 0, sizeof(m));
    m.next = (int64_t*)malloc(sizeof(int64_t) * N);
    m.key = vkey;
    m.head = -1;

    int64_t M = 0;
    for (int i = 0; i < N; i++) {
        uint64_t iX = (uint64_t)floorf((pts[3 * i + 0] - org[0]) / dl);
        uint64_t iY = (uint64_t)floorf((pts[3 * i + 1] - org[1]) / dl);
        uint64_t iZ = (uint64_t)floorf((pts[3 * i + 2] - org[2]) / dl);
        uint64_t key = iX + NX * iY + NX * NY * iZ;
        int64_t v = orc_find_or_add(&fd, key, M);
        if (v < 0) {
            v = M++;
            vkey[v] = key;
            orc_umap_insert(&m, v);
            for (int l = 0; l < ldim; l++) cmax[(size_t)v * ldim + l] = cls[(size_t)i * ldim + l];
        }
        cnt[v] += 1;
        acc[3 * v + 0] += pts[3 * i + 0];
        acc[3 * v + 1] += pts[3 * i + 1];
        acc[3 * v + 2] += pts[3 * i + 2];
        for (int f = 0; f < fdim; f++) facc[(size_t)v * fdim + f] += feat[(size_t)i * fdim + f];
        for (int l = 0; l < ldim; l++) {
            int c = cls[(size_t)i * ldim + l];
            if (c > cmax[(size_t)v * ldim + l]) cmax[(size_t)v * ldim + l] = c;
        }
    }
    int64_t o = 0;
    for (int64_t p = m.head; p >= 0; p = m.next[p], o++) {
        float s = (float)(1.0 / cnt[p]);
        out_p[3 * o + 0] = acc[3 * p + 0] * s;
        out_p[3 * o + 1] = acc[3 * p + 1] * s;
        out_p[3 * o + 2] = acc[3 * p + 2] * s;
        float c = (float)cnt[p];
        for (int f = 0; f < fdim; f++) out_f[(size_t)o * fdim + f] = facc[(size_t)p * fdim + f] / c;
        for (int l = 0; l < ldim; l++) out_c[(size_t)o * ldim + l] = cmax[(size_t)p * ldim + l];
    }
    free(fd.k); free(fd.v); free(vkey); free(acc); free(cnt); free(facc); free(cmax);
    free(m.next); free(m.bucket);
    return (int)M;
}

/* batch_grid_subsampling -- tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.cpp:101-149:
 * independent per batch element, concatenated; out_b[b] = count of element b. */
int orc_batch_grid_subsampling(const float* pts, int N, const int* lens, int B, float dl, float* out_p, int* out_b) {
    int off = 0, M = 0;
    (void)N;
    for (int b = 0; b < B; b++) {
        int m = orc_grid_subsampling(pts + 3 * (size_t)off, lens[b], NULL, 0, NULL, 0, dl, out_p + 3 * (size_t)M, NULL, NULL);
        out_b[b] = m;
        M += m;
        off += lens[b];
    }
    return M;
}

/* ------------------------------------------------------------------------------------------------
 * Radius neighbours.
 * Follows batch_ordered_neighbors, tf_custom_ops/tf_neighbors/neighbors/neighbors.cpp:125-208 --
 * the reference's own deterministic variant (stable: equal d2 keep ascending support index, because
 * supports are visited in index order and inserted at upper_bound) -- which differs from the active
 * batch_nanoflann_neighbors (:211-332) only inside runs of exactly equal d2.
 *   r2 = radius*radius (fp32)                          :139
 *   d2 = (p0 - p).sq_norm() = (dx*dx + dy*dy) + dz*dz  cloud.h:71-74, dx = query - support
 *   keep iff d2 < r2 (strict)                          :169
 *   only supports of the query's own batch element; indices global (+sum_sb)      :157-177
 *   width = max count over all queries; pad = total number of supports            :193-203
 * `grid` != 0 uses a uniform cell grid (cell edge > radius) to visit candidates; the result is the
 * same set in the same order (key = (d2, index)), only faster.  Returns Kmax, *out malloc'ed.
 * ------------------------------------------------------------------------------------------------ */
typedef struct { float d2; int idx; } orc_hit;
static int orc_hit_cmp(const void* a, const void* b) {
    const orc_hit* x = (const orc_hit*)a; const orc_hit* y = (const orc_hit*)b;
    if (x->d2 < y->d2) return -1;
    if (x->d2 > y->d2) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

int orc_batch_neighbors(const float* q, int Nq, const float* s, int Ns, const int* qb, const int* sb, int B,
                        float radius, int grid, int** out) {
    float r2 = radius * radius;
    int* cnt = (int*)calloc(Nq > 0 ? Nq : 1, sizeof(int));
    orc_hit** rows = (orc_hit**)calloc(Nq > 0 ? Nq : 1, sizeof(orc_hit*));
    int kmax = 0, qoff = 0, soff = 0;
    for (int b = 0; b < B; b++) {
        int nq = qb[b], ns = sb[b];
        const float* S = s + 3 * (size_t)soff;
        /* optional cell grid over this element's supports */
        int* cell_start = NULL; int* order = NULL;
        double mn[3] = {0, 0, 0}, h = (double)radius * 1.001; long dims[3] = {1, 1, 1};
        if (grid && ns > 0) {
            double mx[3];
            for (int d = 0; d < 3; d++) mn[d] = mx[d] = S[d];
            for (int i = 0; i < ns; i++) for (int d = 0; d < 3; d++) {
                if (S[3 * i + d] < mn[d]) mn[d] = S[3 * i + d];
                if (S[3 * i + d] > mx[d]) mx[d] = S[3 * i + d];
            }
            for (;;) {
                for (int d = 0; d < 3; d++) dims[d] = (long)floor((mx[d] - mn[d]) / h) + 1;
                if ((double)dims[0] * dims[1] * dims[2] <= 1e7) break;
                h *= 2;
            }
            long nc = dims[0] * dims[1] * dims[2];
            cell_start = (int*)calloc(nc + 1, sizeof(int));
            order = (int*)malloc(sizeof(int) * ns);
            int* cid = (int*)malloc(sizeof(int) * ns);
            for (int i = 0; i < ns; i++) {
                long c[3];
                for (int d = 0; d < 3; d++) c[d] = (long)floor((S[3 * i + d] - mn[d]) / h);
                cid[i] = (int)(c[0] + dims[0] * (c[1] + dims[1] * c[2]));
                cell_start[cid[i] + 1]++;
            }
            for (long c = 0; c < nc; c++) cell_start[c + 1] += cell_start[c];
            int* cur = (int*)malloc(sizeof(int) * nc);
            memcpy(cur, cell_start, sizeof(int) * nc);
            for (int i = 0; i < ns; i++) order[cur[cid[i]]++] = i;
            free(cur); free(cid);
        }
        for (int i = 0; i < nq; i++) {
            const float* p0 = q + 3 * (size_t)(qoff + i);
            int cap = 64, n = 0;
            orc_hit* hits = (orc_hit*)malloc(sizeof(orc_hit) * cap);
#define ORC_TEST(j)                                                                  \
            do {                                                                     \
                float dx = p0[0] - S[3 * (j) + 0], dy = p0[1] - S[3 * (j) + 1], dz = p0[2] - S[3 * (j) + 2]; \
                float d2 = dx * dx + dy * dy + dz * dz;                              \
                if (d2 < r2) {                                                       \
                    if (n == cap) { cap *= 2; hits = (orc_hit*)realloc(hits, sizeof(orc_hit) * cap); } \
                    hits[n].d2 = d2; hits[n].idx = soff + (j); n++;                  \
                }                                                                    \
            } while (0)
            if (cell_start) {
                long c[3];
                for (int d = 0; d < 3; d++) c[d] = (long)floor(((double)p0[d] - mn[d]) / h);
                for (long z = c[2] - 1; z <= c[2] + 1; z++) {
                    if (z < 0 || z >= dims[2]) continue;
                    for (long y = c[1] - 1; y <= c[1] + 1; y++) {
                        if (y < 0 || y >= dims[1]) continue;
                        long x0 = c[0] - 1 < 0 ? 0 : c[0] - 1, x1 = c[0] + 1 >= dims[0] ? dims[0] - 1 : c[0] + 1;
                        if (x0 > x1) continue;
                        long base = dims[0] * (y + dims[1] * z);
                        for (int t = cell_start[base + x0]; t < cell_start[base + x1 + 1]; t++) ORC_TEST(order[t]);
                    }
                }
            } else {
                for (int j = 0; j < ns; j++) ORC_TEST(j);
            }
#undef ORC_TEST
            qsort(hits, n, sizeof(orc_hit), orc_hit_cmp);
            rows[qoff + i] = hits; cnt[qoff + i] = n;
            if (n > kmax) kmax = n;
        }
        free(cell_start); free(order);
        qoff += nq; soff += ns;
    }
    int* res = (int*)malloc(sizeof(int) * ((size_t)Nq * kmax + 1));
    for (int i = 0; i < Nq; i++) {
        for (int j = 0; j < kmax; j++) res[(size_t)i * kmax + j] = j < cnt[i] ? rows[i][j].idx : Ns;
        free(rows[i]);
    }
    free(rows); free(cnt);
    *out = res;
    return kmax;
}

void orc_free(void* p) { free(p); }
