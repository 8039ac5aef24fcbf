/*
 * vdb_oracle.c -- CPU restatement of the reference's distance / top-k path.
 * TEST INFRASTRUCTURE ONLY (see vdb_oracle.h).  Build: see oracle/Makefile
 * (-O2 -fno-fast-math -ffp-contract=off: Rust never reassociates or contracts
 * f32 arithmetic, so neither may this file).
 *
 * Citations are file:line under /root/reference/src.
 */
#include "vdb_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ===================================================================== *
 * distance/mod.rs
 * ===================================================================== */

/* distance/mod.rs:72-74 -- a.iter().zip(b).map(|(x,y)| x*y).sum(): strict left fold,
 * product rounded to f32, then the add rounded to f32. */
float orc_dot(const float *a, const float *b, size_t n) {
    float acc = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float p = a[i] * b[i];
        acc = acc + p;
    }
    return acc;
}

/* distance/mod.rs:75-77 */
float orc_l2(const float *a, const float *b, size_t n) {
    float acc = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float df = a[i] - b[i];
        float sq = df * df;
        acc = acc + sq;
    }
    return acc;
}

/* distance/mod.rs:46-48 */
float orc_norm(const float *a, size_t n) { return sqrtf(orc_dot(a, a, n)); }

static inline float f32_max(float a, float b) {
    /* f32::max: returns the non-NaN operand if one is NaN */
    if (isnan(a)) return b;
    if (isnan(b)) return a;
    return a > b ? a : b;
}

/* distance/mod.rs:66-69 */
static inline float cosine_cached(const float *a, const float *b, size_t n, float na, float nb) {
    float den = f32_max(na * nb, 1e-10f);
    float q = orc_dot(a, b, n) / den;
    return 1.0f - q;
}

/* distance/mod.rs:60-64 */
float orc_cosine(const float *a, const float *b, size_t n) {
    float na = orc_norm(a, n);
    float nb = orc_norm(b, n);
    return cosine_cached(a, b, n, na, nb);
}

/* distance/mod.rs:106-113 */
float orc_dist(int dist, const float *a, const float *b, size_t n) {
    return dist == ORC_L2SQR ? orc_l2(a, b, n) : orc_cosine(a, b, n);
}

/* distance/mod.rs:31-36 */
float orc_dist_cache(int dist, const float *a, size_t n) {
    return dist == ORC_L2SQR ? orc_dot(a, a, n) : orc_norm(a, n);
}

/* distance/mod.rs:54-57 (ip_a + ip_b - 2.0*dot, evaluated left to right) and :120-129 */
float orc_dist_cached(int dist, const float *a, const float *b, size_t n, float ca, float cb) {
    if (dist == ORC_L2SQR) {
        float s = ca + cb;
        float t = 2.0f * orc_dot(a, b, n);
        return s - t;
    }
    return cosine_cached(a, b, n, ca, cb);
}

/* distance/mod.rs:79-94 (u8 scalar: each element cast to f32 first) */
float orc_dot_u8(const uint8_t *a, const uint8_t *b, size_t n) {
    float acc = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float p = (float)a[i] * (float)b[i];
        acc = acc + p;
    }
    return acc;
}
float orc_l2_u8(const uint8_t *a, const uint8_t *b, size_t n) {
    float acc = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float df = (float)a[i] - (float)b[i];
        float sq = df * df;
        acc = acc + sq;
    }
    return acc;
}
float orc_dist_u8(int dist, const uint8_t *a, const uint8_t *b, size_t n) {
    if (dist == ORC_L2SQR) return orc_l2_u8(a, b, n);
    float na = sqrtf(orc_dot_u8(a, a, n));
    float nb = sqrtf(orc_dot_u8(b, b, n));
    float den = f32_max(na * nb, 1e-10f);
    return 1.0f - orc_dot_u8(a, b, n) / den;
}

/* ===================================================================== *
 * index_algorithm/candidate_pair.rs
 * ===================================================================== */

/* ordered-float 4.2.2 total order: NaN greatest, all NaN equal, -0 == +0 */
static inline int f32_total_cmp(float a, float b) {
    int an = isnan(a), bn = isnan(b);
    if (an || bn) return an - bn;
    return (a < b) ? -1 : (a > b) ? 1 : 0;
}

/* candidate_pair.rs:36-41 */
int orc_pair_cmp(float da, uint64_t ia, float db, uint64_t ib) {
    int c = f32_total_cmp(da, db);
    if (c) return c;
    return (ia < ib) ? -1 : (ia > ib) ? 1 : 0;
}

typedef struct {
    float d;
    uint64_t i;
} pair_t;

static inline int pcmp(pair_t a, pair_t b) { return orc_pair_cmp(a.d, a.i, b.d, b.i); }

/* ResultSet: BTreeSet<CandidatePair> bounded by k (candidate_pair.rs:43-53),
 * held as an ascending array. */
typedef struct {
    size_t k, n, cap;
    pair_t *v;
} rset;

static void rset_init(rset *r, size_t k) {
    r->k = k;
    r->n = 0;
    r->cap = k < 16 ? 16 : (k < 4096 ? k : 4096);
    r->v = (pair_t *)malloc(r->cap * sizeof(pair_t));
}
static void rset_free(rset *r) {
    free(r->v);
    r->v = NULL;
    r->n = 0;
}
/* BTreeSet::insert: no-op when an equal element exists */
static void rset_insert_sorted(rset *r, pair_t p) {
    size_t lo = 0, hi = r->n;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (pcmp(r->v[mid], p) < 0)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo < r->n && pcmp(r->v[lo], p) == 0) return;
    if (r->n == r->cap) {
        r->cap *= 2;
        r->v = (pair_t *)realloc(r->v, r->cap * sizeof(pair_t));
    }
    memmove(r->v + lo + 1, r->v + lo, (r->n - lo) * sizeof(pair_t));
    r->v[lo] = p;
    r->n++;
}
/* candidate_pair.rs:61-74 */
static int rset_add(rset *r, pair_t p) {
    if (r->n < r->k) {
        rset_insert_sorted(r, p);
        return 1;
    }
    if (r->n > 0) {
        pair_t last = r->v[r->n - 1];
        if (f32_total_cmp(p.d, last.d) < 0) { /* strict, distance only */
            r->n--;                           /* pop_last */
            rset_insert_sorted(r, p);
            return 1;
        }
    }
    return 0;
}
/* candidate_pair.rs:55-57 */
static int rset_check_candidate(const rset *r, pair_t p) {
    return r->n < r->k || pcmp(p, r->v[r->n - 1]) < 0;
}

/* candidate_pair.rs:127-140 */
float orc_recall(const uint64_t *gt, size_t n_gt, const uint64_t *pred, size_t n_pred) {
    size_t rec = 0;
    for (size_t i = 0; i < n_gt; i++)
        for (size_t j = 0; j < n_pred; j++)
            if (pred[j] == gt[i]) {
                rec++;
                break;
            }
    return (float)rec / (float)n_gt;
}

/* ===================================================================== *
 * index_algorithm/flat_index.rs
 * ===================================================================== */

/* flat_index.rs:48-57 */
size_t orc_flat_knn(const float *base, size_t n, size_t dim, int dist, const float *query, size_t k,
                    uint64_t *out_idx, float *out_dist) {
    rset r;
    rset_init(&r, k);
    for (size_t i = 0; i < n; i++) {
        pair_t p = {orc_dist(dist, query, base + i * dim, dim), i};
        rset_add(&r, p);
    }
    size_t cnt = r.n;
    for (size_t j = 0; j < cnt; j++) {
        out_idx[j] = r.v[j].i;
        out_dist[j] = r.v[j].d;
    }
    rset_free(&r);
    return cnt;
}

typedef struct {
    const float *base;
    size_t n, dim;
    int dist;
    const float *queries;
    size_t nq, k;
    uint64_t *out_idx;
    float *out_dist;
    uint64_t *out_count;
    size_t *next;
    pthread_mutex_t *mu;
} flat_job;

static void *flat_worker(void *arg) {
    flat_job *j = (flat_job *)arg;
    for (;;) {
        pthread_mutex_lock(j->mu);
        size_t q = (*j->next)++;
        pthread_mutex_unlock(j->mu);
        if (q >= j->nq) break;
        size_t c = orc_flat_knn(j->base, j->n, j->dim, j->dist, j->queries + q * j->dim, j->k,
                                j->out_idx + q * j->k, j->out_dist + q * j->k);
        if (j->out_count) j->out_count[q] = c;
    }
    return NULL;
}

void orc_flat_knn_batch(const float *base, size_t n, size_t dim, int dist, const float *queries,
                        size_t nq, size_t k, uint64_t *out_idx, float *out_dist,
                        uint64_t *out_count, int nthreads) {
    size_t next = 0;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    flat_job j = {base, n, dim, dist, queries, nq, k, out_idx, out_dist, out_count, &next, &mu};
    if (nthreads <= 1) {
        flat_worker(&j);
        return;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, flat_worker, &j);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
}

/* ===================================================================== *
 * RNG (own stream; parity unpinned vs the reference's ChaCha12 StdRng)
 * ===================================================================== */
uint64_t orc_splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* uniform in (0,1), 24 bits */
float orc_uniform_open01(uint64_t *s) {
    uint32_t r = (uint32_t)(orc_splitmix64(s) >> 40); /* 24 bits */
    return ((float)r + 0.5f) * (1.0f / 16777216.0f);
}
static size_t rng_below(uint64_t *s, size_t n) { return (size_t)(orc_splitmix64(s) % (uint64_t)n); }

/* ===================================================================== *
 * distance/k_means.rs
 * ===================================================================== */

/* k_means.rs:40-57: min over CandidatePair(i, dist.d(v, c)) */
static size_t find_nearest_base(const float *v, const float *cents, size_t k, size_t gd, int dist) {
    size_t best = 0;
    float bd = 0.0f;
    for (size_t c = 0; c < k; c++) {
        float d = orc_dist(dist, v, cents + c * gd, gd);
        if (c == 0 || orc_pair_cmp(d, c, bd, best) < 0) {
            bd = d;
            best = c;
        }
    }
    return best;
}

void orc_kmeans(const float *rows, size_t n, size_t dim, size_t c0, size_t c1, size_t k,
                size_t max_iter, float tol, int dist, uint64_t *rng, float *out) {
    size_t gd = c1 - c0;
    float *sel = (float *)malloc(n * gd * sizeof(float));
    for (size_t i = 0; i < n; i++) memcpy(sel + i * gd, rows + i * dim + c0, gd * sizeof(float));
    float *cent = out;
    /* k_means_init :61-87 */
    size_t first = rng_below(rng, n);
    memcpy(cent, sel + first * gd, gd * sizeof(float));
    float *w = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; i++) w[i] = INFINITY;
    for (size_t idx = 1; idx < k; idx++) {
        const float *prev = cent + (idx - 1) * gd;
        int bad = 0;
        float total = 0.0f;
        for (size_t i = 0; i < n; i++) {
            float d = orc_dist(dist, prev, sel + i * gd, gd);
            /* f32::min: non-NaN operand wins */
            if (isnan(w[i]))
                w[i] = d;
            else if (!isnan(d) && d < w[i])
                w[i] = d;
            if (!(w[i] >= 0.0f) || isinf(w[i])) bad = 1; /* WeightedIndex: InvalidWeight */
            total += w[i];
        }
        size_t c;
        if (bad || !(total > 0.0f) || isinf(total)) {
            c = rng_below(rng, n); /* :80-82 fallback */
        } else {
            float u = orc_uniform_open01(rng) * total;
            float cum = 0.0f;
            c = n - 1;
            for (size_t i = 0; i < n; i++) {
                cum += w[i];
                if (cum > u) {
                    c = i;
                    break;
                }
            }
        }
        memcpy(cent + idx * gd, sel + c * gd, gd * sizeof(float));
    }
    /* Lloyd :95-162 */
    float *sums = (float *)malloc(k * gd * sizeof(float));
    size_t *assign = (size_t *)malloc(n * sizeof(size_t));
    size_t *cnt = (size_t *)malloc(k * sizeof(size_t));
    for (size_t it = 0; it < max_iter; it++) {
        for (size_t i = 0; i < n; i++) assign[i] = find_nearest_base(sel + i * gd, cent, k, gd, dist);
        memset(cnt, 0, k * sizeof(size_t));
        for (size_t c = 0; c < k * gd; c++) sums[c] = 0.0f;
        for (size_t i = 0; i < n; i++) { /* point order within each cluster == ascending i */
            size_t c = assign[i];
            cnt[c]++;
            for (size_t j = 0; j < gd; j++) sums[c * gd + j] += sel[i * gd + j];
        }
        for (size_t c = 0; c < k; c++) {
            if (cnt[c] == 0) { /* :131-137 empty cluster keeps its centroid */
                memcpy(sums + c * gd, cent + c * gd, gd * sizeof(float));
            } else {
                float fn = (float)cnt[c];
                for (size_t j = 0; j < gd; j++) sums[c * gd + j] /= fn;
            }
        }
        float max_diff = -INFINITY;
        for (size_t c = 0; c < k; c++) max_diff = f32_max(max_diff, orc_l2(cent + c * gd, sums + c * gd, gd));
        memcpy(cent, sums, k * gd * sizeof(float));
        if (max_diff < tol) break;
    }
    free(sums);
    free(assign);
    free(cnt);
    free(w);
    free(sel);
}

/* k_means.rs:166-170 (find_nearest, no column selection) for every row: the cluster assignment of
 * IVFIndex::from_vec_set (ivf_index.rs:92-100) */
void orc_assign_nearest(const float *base, size_t n, size_t dim, int dist, const float *cents, size_t k, uint64_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = find_nearest_base(base + i * dim, cents, k, dim, dist);
}

/* ===================================================================== *
 * index_algorithm/ivf_index.rs
 * ===================================================================== */
/* IVFIndex::knn_with_ef (ivf_index.rs:143-154): find_n_nearest (k_means.rs:174-190) picks the probes, the members of
 * the probed clusters are offered to ResultSet::add cluster by cluster in ascending id (:95-100 builds them so).
 * offsets: k_clusters+1, members: n ids grouped by cluster. */
size_t orc_ivf_knn(const float *base, size_t dim, int dist, const float *cents, size_t k_clusters,
                   const uint64_t *offsets, const uint64_t *members, const float *query, size_t k, size_t n_probes,
                   uint64_t *out_idx, float *out_dist) {
    rset probes;
    rset_init(&probes, n_probes);
    for (size_t c = 0; c < k_clusters; c++) {
        pair_t p = {orc_dist(dist, query, cents + c * dim, dim), c};
        rset_add(&probes, p);
    }
    rset r;
    rset_init(&r, k);
    for (size_t j = 0; j < probes.n; j++) {
        size_t c = (size_t)probes.v[j].i;
        for (uint64_t t = offsets[c]; t < offsets[c + 1]; t++) {
            uint64_t i = members[t];
            pair_t p = {orc_dist(dist, base + i * dim, query, dim), i};
            rset_add(&r, p);
        }
    }
    size_t cnt = r.n;
    for (size_t j = 0; j < cnt; j++) {
        out_idx[j] = r.v[j].i;
        out_dist[j] = r.v[j].d;
    }
    rset_free(&r);
    rset_free(&probes);
    return cnt;
}

/* ===================================================================== *
 * distance/pq_table.rs
 * ===================================================================== */

/* pq_table.rs:38-53 */
size_t orc_pq_groups(size_t dim, size_t m, uint64_t *gstart) {
    size_t cur = 0, g = 0;
    gstart[0] = 0;
    while (cur < dim) {
        size_t rem_g = m - g;
        size_t gs = (dim - cur + rem_g - 1) / rem_g;
        cur += gs;
        g++;
        gstart[g] = cur;
    }
    return g;
}

orc_pq *orc_pq_new(size_t dim, size_t m, size_t n_bits, int dist, const float *centroids) {
    orc_pq *pq = (orc_pq *)calloc(1, sizeof(orc_pq));
    pq->dim = dim;
    pq->m = m;
    pq->n_bits = n_bits;
    pq->k = (uint64_t)1 << n_bits;                     /* :150 */
    pq->enc_dim = n_bits == 4 ? (m + 1) / 2 : m;       /* :169-173 */
    pq->dist = dist;
    pq->gstart = (uint64_t *)malloc((m + 1) * sizeof(uint64_t));
    orc_pq_groups(dim, m, pq->gstart);
    pq->centroids = (float *)malloc(pq->k * dim * sizeof(float));
    if (centroids) memcpy(pq->centroids, centroids, pq->k * dim * sizeof(float));
    pq->cent_cache = (float *)calloc(m * pq->k, sizeof(float));
    if (centroids && dist == ORC_COSINE) { /* :160-165 */
        for (size_t g = 0; g < m; g++) {
            size_t gd = pq->gstart[g + 1] - pq->gstart[g];
            const float *cg = pq->centroids + pq->k * pq->gstart[g];
            for (size_t c = 0; c < pq->k; c++) pq->cent_cache[g * pq->k + c] = orc_dot(cg + c * gd, cg + c * gd, gd);
        }
    }
    return pq;
}
void orc_pq_free(orc_pq *pq) {
    if (!pq) return;
    free(pq->gstart);
    free(pq->centroids);
    free(pq->cent_cache);
    free(pq->codes);
    free(pq);
}
static size_t pq_find_nearest(const orc_pq *pq, size_t g, const float *v) { /* k_means.rs:166-170 */
    size_t gd = pq->gstart[g + 1] - pq->gstart[g];
    return find_nearest_base(v + pq->gstart[g], pq->centroids + pq->k * pq->gstart[g], pq->k, gd, pq->dist);
}
/* pq_table.rs:66-91 */
void orc_pq_encode_row(const orc_pq *pq, const float *v, uint8_t *out) {
    size_t m = pq->m;
    if (pq->n_bits == 4) {
        memset(out, 0, (m + 1) / 2);
        for (size_t i = 0; i < m / 2; i++) {
            size_t v0 = pq_find_nearest(pq, 2 * i, v);
            size_t v1 = pq_find_nearest(pq, 2 * i + 1, v);
            out[i] = (uint8_t)(v0 | (v1 << 4));
        }
        if (m % 2 == 1) out[m / 2] = (uint8_t)pq_find_nearest(pq, m - 1, v);
    } else {
        for (size_t g = 0; g < m; g++) out[g] = (uint8_t)pq_find_nearest(pq, g, v);
    }
}
void orc_pq_encode_all(orc_pq *pq, const float *base, size_t n) {
    free(pq->codes);
    pq->codes = (uint8_t *)malloc(n * pq->enc_dim + 1);
    pq->n = n;
    for (size_t i = 0; i < n; i++) orc_pq_encode_row(pq, base + i * pq->dim, pq->codes + i * pq->enc_dim);
}
void orc_pq_set_codes(orc_pq *pq, const uint8_t *codes, size_t n) {
    free(pq->codes);
    pq->codes = (uint8_t *)malloc(n * pq->enc_dim + 1);
    memcpy(pq->codes, codes, n * pq->enc_dim);
    pq->n = n;
}
/* pq_table.rs:195-224 */
float orc_pq_lookup(const orc_pq *pq, const float *query, float *lut) {
    for (size_t g = 0; g < pq->m; g++) {
        size_t gd = pq->gstart[g + 1] - pq->gstart[g];
        const float *vs = query + pq->gstart[g];
        const float *cg = pq->centroids + pq->k * pq->gstart[g];
        for (size_t c = 0; c < pq->k; c++)
            lut[g * pq->k + c] = pq->dist == ORC_L2SQR ? orc_l2(vs, cg + c * gd, gd) : orc_dot(vs, cg + c * gd, gd);
    }
    return pq->dist == ORC_L2SQR ? 0.0f : orc_norm(query, pq->dim);
}
/* pq_table.rs:239-301 */
float orc_pq_adc(const orc_pq *pq, const uint8_t *code, const float *lut, float q_cache) {
    float sum = 0.0f, cdp = 0.0f;
    size_t m = pq->m, k = pq->k;
    int cosine = pq->dist == ORC_COSINE;
    if (pq->n_bits == 4) {
        size_t i = 0;
        for (size_t b = 0; b < pq->enc_dim; b++) {
            uint8_t u = code[b];
            if (i < m) {
                sum += lut[i * k + (u & 0xf)];
                if (cosine) cdp += pq->cent_cache[i * k + (u & 0xf)];
            }
            i++;
            if (i < m) {
                sum += lut[i * k + (u >> 4)];
                if (cosine) cdp += pq->cent_cache[i * k + (u >> 4)];
            }
            i++;
        }
    } else {
        for (size_t i = 0; i < m; i++) {
            sum += lut[i * k + code[i]];
            if (cosine) cdp += pq->cent_cache[i * k + code[i]];
        }
    }
    if (!cosine) return sum;
    float norm0 = sqrtf(cdp);
    float den = f32_max(norm0 * q_cache, 1e-10f);
    return 1.0f - sum / den;
}

/* candidate_pair.rs:102-108 */
typedef float (*idx_dist_fn)(void *ctx, uint64_t idx);
static size_t pq_resort(const rset *src, size_t k, idx_dist_fn f, void *ctx, uint64_t *out_idx, float *out_dist) {
    rset r;
    rset_init(&r, k);
    for (size_t j = 0; j < src->n; j++) {
        pair_t p = {f(ctx, src->v[j].i), src->v[j].i};
        rset_add(&r, p);
    }
    size_t cnt = r.n;
    for (size_t j = 0; j < cnt; j++) {
        out_idx[j] = r.v[j].i;
        out_dist[j] = r.v[j].d;
    }
    rset_free(&r);
    return cnt;
}

typedef struct {
    const float *base;
    size_t dim;
    int dist;
    const float *query;
} flat_ctx;
static float flat_exact(void *c, uint64_t idx) {
    flat_ctx *f = (flat_ctx *)c;
    return orc_dist(f->dist, f->query, f->base + idx * f->dim, f->dim);
}

/* ADC distance of every encoded row (the operand stream of flat_index.rs:97-100), for shard-merge tests */
void orc_pq_adc_all(const orc_pq *pq, size_t n, const float *query, float *out) {
    float *lut = (float *)malloc(pq->m * pq->k * sizeof(float));
    float qc = orc_pq_lookup(pq, query, lut);
    for (size_t i = 0; i < n; i++) out[i] = orc_pq_adc(pq, pq->codes + i * pq->enc_dim, lut, qc);
    free(lut);
}

/* flat_index.rs:84-104 */
size_t orc_flat_knn_pq(const float *base, size_t n, size_t dim, int dist, const orc_pq *pq,
                       const float *query, size_t k, size_t ef, uint64_t *out_idx, float *out_dist) {
    rset r;
    rset_init(&r, ef > k ? ef : k);
    float *lut = (float *)malloc(pq->m * pq->k * sizeof(float));
    float qc = orc_pq_lookup(pq, query, lut);
    for (size_t i = 0; i < n; i++) {
        pair_t p = {orc_pq_adc(pq, pq->codes + i * pq->enc_dim, lut, qc), i};
        rset_add(&r, p);
    }
    flat_ctx fc = {base, dim, dist, query};
    size_t cnt = pq_resort(&r, k, flat_exact, &fc, out_idx, out_dist);
    free(lut);
    rset_free(&r);
    return cnt;
}

/* pq_table.rs:141-191 */
orc_pq *orc_pq_train(const float *base, size_t n, size_t dim, size_t m, size_t n_bits, int dist,
                     size_t k_means_size, size_t max_iter, float tol, uint64_t seed) {
    uint64_t rng = seed;
    orc_pq *pq = orc_pq_new(dim, m, n_bits, dist, NULL);
    const float *train = base;
    size_t nt = n;
    float *sample = NULL;
    if (k_means_size && k_means_size < n) { /* vec_set.rs:154-163 random_sample */
        size_t *perm = (size_t *)malloc(n * sizeof(size_t));
        for (size_t i = 0; i < n; i++) perm[i] = i;
        sample = (float *)malloc(k_means_size * dim * sizeof(float));
        for (size_t i = 0; i < k_means_size; i++) {
            size_t j = i + rng_below(&rng, n - i);
            size_t t = perm[i];
            perm[i] = perm[j];
            perm[j] = t;
            memcpy(sample + i * dim, base + perm[i] * dim, dim * sizeof(float));
        }
        free(perm);
        train = sample;
        nt = k_means_size;
    }
    for (size_t g = 0; g < m; g++) {
        size_t c0 = pq->gstart[g], c1 = pq->gstart[g + 1];
        float *cg = pq->centroids + pq->k * c0;
        orc_kmeans(train, nt, dim, c0, c1, pq->k, max_iter, tol, dist, &rng, cg);
        for (size_t c = 0; c < pq->k; c++)
            pq->cent_cache[g * pq->k + c] =
                dist == ORC_L2SQR ? 0.0f : orc_dot(cg + c * (c1 - c0), cg + c * (c1 - c0), c1 - c0);
    }
    free(sample);
    orc_pq_encode_all(pq, base, n);
    return pq;
}

/* ===================================================================== *
 * index_algorithm/hnsw_index.rs
 * ===================================================================== */

/* hnsw_index.rs:493-536 */
orc_hnsw *orc_hnsw_new(size_t dim, int dist, size_t M, size_t ef_construction) {
    orc_hnsw *h = (orc_hnsw *)calloc(1, sizeof(orc_hnsw));
    h->dim = dim;
    h->dist = dist;
    h->m = M < 10000 ? M : 10000;
    h->max_m0 = h->m * 2;
    h->ef_construction = ef_construction > h->max_m0 ? ef_construction : h->max_m0;
    h->default_ef = h->ef_construction / 2;
    h->inv_log_m = 1.0f / logf((float)h->m);
    return h;
}
void orc_hnsw_free(orc_hnsw *h) {
    if (!h) return;
    free(h->rows);
    free(h->cache);
    free(h->level0);
    free(h->len0);
    free(h->vec_level);
    free(h->upper_off);
    free(h->upper);
    free(h->upper_len);
    free(h);
}
/* hnsw_index.rs:144-147 */
uint64_t orc_hnsw_level_from_uniform(const orc_hnsw *h, float u) {
    float l = floorf(-logf(u) * h->inv_log_m);
    return (uint64_t)l;
}

static void hnsw_reserve(orc_hnsw *h, size_t want) {
    if (want <= h->cap) return;
    size_t nc = h->cap ? h->cap : 1024;
    while (nc < want) nc *= 2;
    h->rows = (float *)realloc(h->rows, nc * h->dim * sizeof(float));
    h->cache = (float *)realloc(h->cache, nc * sizeof(float));
    h->level0 = (uint32_t *)realloc(h->level0, nc * h->max_m0 * sizeof(uint32_t));
    h->len0 = (uint64_t *)realloc(h->len0, nc * sizeof(uint64_t));
    h->vec_level = (uint64_t *)realloc(h->vec_level, nc * sizeof(uint64_t));
    h->upper_off = (uint64_t *)realloc(h->upper_off, (nc + 1) * sizeof(uint64_t));
    h->cap = nc;
}
/* hnsw_index.rs:244-256 */
static uint64_t hnsw_push_init(orc_hnsw *h, const float *vec, uint64_t level) {
    hnsw_reserve(h, h->n + 1);
    uint64_t idx = h->n++;
    memcpy(h->rows + idx * h->dim, vec, h->dim * sizeof(float));
    memset(h->level0 + idx * h->max_m0, 0, h->max_m0 * sizeof(uint32_t));
    h->len0[idx] = 0;
    h->vec_level[idx] = level;
    h->upper_off[idx] = h->upper_total;
    if (h->upper_total + level > h->upper_cap) {
        size_t nc = h->upper_cap ? h->upper_cap : 1024;
        while (nc < h->upper_total + level) nc *= 2;
        h->upper = (uint32_t *)realloc(h->upper, nc * h->m * sizeof(uint32_t));
        h->upper_len = (uint64_t *)realloc(h->upper_len, nc * sizeof(uint64_t));
        h->upper_cap = nc;
    }
    for (uint64_t l = 0; l < level; l++) {
        memset(h->upper + (h->upper_total + l) * h->m, 0, h->m * sizeof(uint32_t));
        h->upper_len[h->upper_total + l] = 0;
    }
    h->upper_total += level;
    h->upper_off[h->n] = h->upper_total;
    h->cache[idx] = orc_dist_cache(h->dist, vec, h->dim);
    return idx;
}
/* hnsw_index.rs:173-183 */
static inline const uint32_t *hnsw_links(const orc_hnsw *h, uint64_t v, uint64_t level, size_t *len) {
    if (level == 0) {
        *len = h->len0[v];
        return h->level0 + v * h->max_m0;
    }
    uint64_t slot = h->upper_off[v] + level - 1;
    *len = h->upper_len[slot];
    return h->upper + slot * h->m;
}
static inline uint32_t *hnsw_links_mut(orc_hnsw *h, uint64_t v, uint64_t level) {
    if (level == 0) return h->level0 + v * h->max_m0;
    return h->upper + (h->upper_off[v] + level - 1) * h->m;
}
/* hnsw_index.rs:197-201 */
static void hnsw_put_links(orc_hnsw *h, uint64_t v, uint64_t level, const uint32_t *links, size_t len) {
    if (level == 0)
        h->len0[v] = len;
    else
        h->upper_len[h->upper_off[v] + level - 1] = len;
    memcpy(hnsw_links_mut(h, v, level), links, len * sizeof(uint32_t));
}

typedef struct {
    orc_hnsw *h;
    const float *query;
    float qcache;
} hq_ctx;
/* hnsw_index.rs:351-355 */
static float hq_dist(void *c, uint64_t idx) {
    hq_ctx *q = (hq_ctx *)c;
    q->h->stat_n_dist++;
    return orc_dist_cached(q->h->dist, q->h->rows + idx * q->h->dim, q->query, q->h->dim, q->h->cache[idx], q->qcache);
}
/* hnsw_index.rs:356-358 */
static float hnsw_inner(const orc_hnsw *h, uint64_t a, uint64_t b) {
    return orc_dist_cached(h->dist, h->rows + a * h->dim, h->rows + b * h->dim, h->dim, h->cache[a], h->cache[b]);
}

/* min-heap standing in for the BTreeSet queue (pop_first == min under the same total order) */
typedef struct {
    pair_t *v;
    size_t n, cap;
} heap_t;
static void heap_push(heap_t *hp, pair_t p) {
    if (hp->n == hp->cap) {
        hp->cap = hp->cap ? hp->cap * 2 : 256;
        hp->v = (pair_t *)realloc(hp->v, hp->cap * sizeof(pair_t));
    }
    size_t i = hp->n++;
    while (i > 0) {
        size_t par = (i - 1) / 2;
        if (pcmp(hp->v[par], p) <= 0) break;
        hp->v[i] = hp->v[par];
        i = par;
    }
    hp->v[i] = p;
}
static pair_t heap_pop(heap_t *hp) {
    pair_t top = hp->v[0];
    pair_t last = hp->v[--hp->n];
    size_t i = 0;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, s = i;
        pair_t sv = last;
        if (l < hp->n && pcmp(hp->v[l], sv) < 0) {
            s = l;
            sv = hp->v[l];
        }
        if (r < hp->n && pcmp(hp->v[r], sv) < 0) {
            s = r;
            sv = hp->v[r];
        }
        if (s == i) break;
        hp->v[i] = hp->v[s];
        i = s;
    }
    if (hp->n) hp->v[i] = last;
    return top;
}

typedef struct {
    uint32_t *stamp;
    uint32_t epoch;
    size_t n;
    heap_t heap;
} scratch_t;
static void scratch_prepare(scratch_t *s, size_t n) {
    if (s->n < n) {
        free(s->stamp);
        s->stamp = (uint32_t *)calloc(n, sizeof(uint32_t));
        s->n = n;
        s->epoch = 0;
    }
    s->epoch++;
    if (s->epoch == 0) {
        memset(s->stamp, 0, s->n * sizeof(uint32_t));
        s->epoch = 1;
    }
    s->heap.n = 0;
}
static void scratch_free(scratch_t *s) {
    free(s->stamp);
    free(s->heap.v);
}

/* hnsw_index.rs:258-291 */
static void search_on_level_fn(orc_hnsw *h, uint64_t ep, uint64_t level, size_t ef, idx_dist_fn f, void *ctx,
                               scratch_t *s, rset *result) {
    scratch_prepare(s, h->n);
    rset_init(result, ef);
    s->stamp[ep] = s->epoch;
    pair_t e = {f(ctx, ep), ep};
    rset_add(result, e);
    heap_push(&s->heap, e);
    while (s->heap.n) {
        pair_t p = heap_pop(&s->heap);
        if (!rset_check_candidate(result, p)) break;
        h->stat_n_expanded++;
        size_t len;
        const uint32_t *lk = hnsw_links(h, p.i, level, &len);
        for (size_t j = 0; j < len; j++) {
            uint64_t nb = lk[j];
            if (s->stamp[nb] == s->epoch) continue;
            s->stamp[nb] = s->epoch;
            pair_t np = {f(ctx, nb), nb};
            rset_add(result, np);
            heap_push(&s->heap, np);
        }
    }
}
/* hnsw_index.rs:306-330 */
static uint64_t greedy_on_level_fn(orc_hnsw *h, uint64_t level, uint64_t ep, idx_dist_fn f, void *ctx) {
    uint64_t cur_p = ep;
    float cur_d = f(ctx, cur_p);
    for (;;) {
        int flag = 0;
        size_t len;
        const uint32_t *lk = hnsw_links(h, cur_p, level, &len); /* slice of the sweep's starting node */
        for (size_t j = 0; j < len; j++) {
            uint64_t nb = lk[j];
            float nd = f(ctx, nb);
            if (nd < cur_d) {
                cur_d = nd;
                cur_p = nb;
                flag = 1;
            }
        }
        if (!flag) break;
    }
    return cur_p;
}
/* hnsw_index.rs:336-350 */
static uint64_t greedy_until_level_fn(orc_hnsw *h, uint64_t target, idx_dist_fn f, void *ctx) {
    uint64_t level = h->enter_level, cur = h->enter_point;
    while (level > target) {
        cur = greedy_on_level_fn(h, level, cur, f, ctx);
        level--;
    }
    return cur;
}

/* candidate_pair.rs:85-99 */
static size_t heuristic(const orc_hnsw *h, const rset *set, size_t m, uint32_t *out) {
    size_t cnt = 0;
    for (size_t j = 0; j < set->n && cnt < m; j++) {
        float d = set->v[j].d;
        uint64_t v = set->v[j].i;
        int ok = 1;
        for (size_t t = 0; t < cnt; t++)
            if (!(hnsw_inner(h, v, out[t]) >= d)) {
                ok = 0;
                break;
            }
        if (ok) out[cnt++] = (uint32_t)v;
    }
    return cnt;
}
/* hnsw_index.rs:204-224 */
static void arrange_links(orc_hnsw *h, uint64_t v, uint64_t level, uint64_t newv) {
    size_t limit = level == 0 ? h->max_m0 : h->m;
    size_t len;
    const uint32_t *lk = hnsw_links(h, v, level, &len);
    uint32_t *links = (uint32_t *)malloc((len + 1) * sizeof(uint32_t));
    memcpy(links, lk, len * sizeof(uint32_t));
    links[len++] = (uint32_t)newv;
    if (len <= limit) {
        hnsw_put_links(h, v, level, links, len);
        free(links);
        return;
    }
    rset set;
    rset_init(&set, limit + 1);
    for (size_t j = 0; j < len; j++) {
        pair_t p = {hnsw_inner(h, v, links[j]), links[j]};
        rset_add(&set, p);
    }
    size_t nl = heuristic(h, &set, limit, links);
    hnsw_put_links(h, v, level, links, nl);
    rset_free(&set);
    free(links);
}
/* hnsw_index.rs:226-239 */
static void connect_new_links(orc_hnsw *h, uint64_t v, uint64_t level, const rset *cand) {
    uint32_t *nb = (uint32_t *)malloc((h->m + 1) * sizeof(uint32_t));
    size_t nn = heuristic(h, cand, h->m, nb); /* M, not max_m0, even on level 0 */
    hnsw_put_links(h, v, level, nb, nn);
    for (size_t j = 0; j < nn; j++) arrange_links(h, nb[j], level, v);
    free(nb);
}

/* hnsw_index.rs:538-572 */
uint64_t orc_hnsw_add(orc_hnsw *h, const float *vec, uint64_t level) {
    uint64_t idx = hnsw_push_init(h, vec, level);
    if (!h->has_enter) {
        h->has_enter = 1;
        h->enter_level = level;
        h->enter_point = idx;
        return idx;
    }
    const float *row = h->rows + idx * h->dim; /* rows may have been realloc'ed */
    hq_ctx q = {h, row, orc_dist_cache(h->dist, row, h->dim)};
    uint64_t enter_level = h->enter_level;
    uint64_t cur_p = level < enter_level ? greedy_until_level_fn(h, level, hq_dist, &q) : h->enter_point;
    scratch_t s = {0};
    uint64_t top = level < enter_level ? level : enter_level;
    for (uint64_t l = top + 1; l-- > 0;) {
        rset cand;
        search_on_level_fn(h, cur_p, l, h->ef_construction, hq_dist, &q, &s, &cand);
        cur_p = cand.v[0].i;
        connect_new_links(h, idx, l, &cand);
        rset_free(&cand);
    }
    scratch_free(&s);
    if (level > enter_level) {
        h->enter_level = level;
        h->enter_point = idx;
    }
    return idx;
}

/* hnsw_index.rs:399-457, executed serially (candidate phase reads only the pre-batch graph) */
void orc_hnsw_add_batch(orc_hnsw *h, const float *vecs, size_t nb, const uint64_t *levels) {
    if (h->n < 1000 /* start_batch_since :506 */ || nb == 1) {
        for (size_t i = 0; i < nb; i++) orc_hnsw_add(h, vecs + i * h->dim, levels[i]);
        return;
    }
    uint64_t first = h->n;
    for (size_t i = 0; i < nb; i++) hnsw_push_init(h, vecs + i * h->dim, levels[i]);
    uint64_t enter_point = h->enter_point, enter_level = h->enter_level;
    rset **cands = (rset **)calloc(nb, sizeof(rset *));
    size_t *ncand = (size_t *)calloc(nb, sizeof(size_t));
    scratch_t s = {0};
    for (size_t i = 0; i < nb; i++) {
        uint64_t idx = first + i;
        uint64_t level = h->vec_level[idx];
        const float *row = h->rows + idx * h->dim;
        hq_ctx q = {h, row, orc_dist_cache(h->dist, row, h->dim)};
        uint64_t cur_p = level < enter_level ? greedy_until_level_fn(h, level, hq_dist, &q) : enter_point;
        uint64_t top = level < enter_level ? level : enter_level;
        cands[i] = (rset *)calloc(top + 1, sizeof(rset));
        ncand[i] = top + 1;
        size_t slot = 0;
        for (uint64_t l = top + 1; l-- > 0;) {
            rset *c = &cands[i][slot++];
            search_on_level_fn(h, cur_p, l, h->ef_construction, hq_dist, &q, &s, c);
            cur_p = c->v[0].i;
            for (size_t r = 0; r < i; r++) { /* rhs_idx < idx && vec_level[rhs] >= level :431-437 */
                uint64_t rhs = first + r;
                if (h->vec_level[rhs] >= l) {
                    pair_t p = {hnsw_inner(h, idx, rhs), rhs};
                    rset_add(c, p);
                }
            }
        }
    }
    scratch_free(&s);
    for (size_t i = 0; i < nb; i++) {
        uint64_t idx = first + i;
        uint64_t level = h->vec_level[idx];
        uint64_t top = level < enter_level ? level : enter_level;
        size_t slot = 0;
        for (uint64_t l = top + 1; l-- > 0;) {
            connect_new_links(h, idx, l, &cands[i][slot]);
            rset_free(&cands[i][slot]);
            slot++;
        }
        free(cands[i]);
    }
    free(cands);
    free(ncand);
    for (size_t i = 0; i < nb; i++) {
        uint64_t idx = first + i;
        if (h->vec_level[idx] > h->enter_level) {
            h->enter_level = h->vec_level[idx];
            h->enter_point = idx;
        }
    }
}

/* hnsw_index.rs:595-611 + :459-475 with an explicit batch size */
orc_hnsw *orc_hnsw_build(const float *base, size_t n, size_t dim, int dist, size_t M, size_t ef_construction,
                         uint64_t seed, size_t batch) {
    orc_hnsw *h = orc_hnsw_new(dim, dist, M, ef_construction);
    uint64_t rng = seed;
    uint64_t *levels = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    for (size_t i = 0; i < n; i++) levels[i] = orc_hnsw_level_from_uniform(h, orc_uniform_open01(&rng));
    size_t cur = 0;
    while (cur < n) {
        size_t bs = 1;
        if (h->n >= 1000) { /* next_batch_size :391-397 with rayon threads*4 replaced by `batch` */
            bs = batch < h->n / h->m ? batch : h->n / h->m;
            if (bs < 1) bs = 1;
        }
        size_t next = cur + bs < n ? cur + bs : n;
        orc_hnsw_add_batch(h, base + cur * dim, next - cur, levels + cur);
        cur = next;
    }
    free(levels);
    return h;
}

orc_hnsw *orc_hnsw_from_graph(const float *base, size_t n, size_t dim, int dist, size_t M, size_t ef_construction,
                              const uint32_t *level0, const uint64_t *len0, const uint64_t *vec_level,
                              const uint32_t *upper, const uint64_t *upper_len, int has_enter,
                              uint64_t enter_point, uint64_t enter_level) {
    orc_hnsw *h = orc_hnsw_new(dim, dist, M, ef_construction);
    hnsw_reserve(h, n ? n : 1);
    h->n = n;
    memcpy(h->rows, base, n * dim * sizeof(float));
    for (size_t i = 0; i < n; i++) h->cache[i] = orc_dist_cache(dist, base + i * dim, dim);
    memcpy(h->level0, level0, n * h->max_m0 * sizeof(uint32_t));
    memcpy(h->len0, len0, n * sizeof(uint64_t));
    memcpy(h->vec_level, vec_level, n * sizeof(uint64_t));
    uint64_t tot = 0;
    for (size_t i = 0; i < n; i++) {
        h->upper_off[i] = tot;
        tot += vec_level[i];
    }
    h->upper_off[n] = tot;
    h->upper_total = h->upper_cap = tot;
    h->upper = (uint32_t *)malloc((tot ? tot : 1) * h->m * sizeof(uint32_t));
    h->upper_len = (uint64_t *)malloc((tot ? tot : 1) * sizeof(uint64_t));
    if (tot) {
        memcpy(h->upper, upper, tot * h->m * sizeof(uint32_t));
        memcpy(h->upper_len, upper_len, tot * sizeof(uint64_t));
    }
    h->has_enter = has_enter;
    h->enter_point = enter_point;
    h->enter_level = enter_level;
    return h;
}

/* hnsw_index.rs:619-634 */
static size_t hnsw_knn_scratch(orc_hnsw *h, const float *query, size_t k, size_t ef, uint64_t *out_idx,
                               float *out_dist, scratch_t *s) {
    if (h->n == 0) return 0;
    if (ef < k) ef = k;
    hq_ctx q = {h, query, orc_dist_cache(h->dist, query, h->dim)};
    uint64_t ep = greedy_until_level_fn(h, 0, hq_dist, &q);
    rset r;
    search_on_level_fn(h, ep, 0, ef, hq_dist, &q, s, &r);
    size_t cnt = r.n < k ? r.n : k;
    for (size_t j = 0; j < cnt; j++) {
        out_idx[j] = r.v[j].i;
        out_dist[j] = r.v[j].d;
    }
    rset_free(&r);
    return cnt;
}
size_t orc_hnsw_knn(orc_hnsw *h, const float *query, size_t k, size_t ef, uint64_t *out_idx, float *out_dist) {
    scratch_t s = {0};
    size_t c = hnsw_knn_scratch(h, query, k, ef, out_idx, out_dist, &s);
    scratch_free(&s);
    return c;
}

typedef struct {
    const orc_pq *pq;
    const float *lut;
    float qc;
    orc_hnsw *h;
} adc_ctx;
static float adc_dist(void *c, uint64_t idx) {
    adc_ctx *a = (adc_ctx *)c;
    a->h->stat_n_dist++;
    return orc_pq_adc(a->pq, a->pq->codes + idx * a->pq->enc_dim, a->lut, a->qc);
}
static float hq_dist_nostat(void *c, uint64_t idx) {
    hq_ctx *q = (hq_ctx *)c;
    return orc_dist_cached(q->h->dist, q->h->rows + idx * q->h->dim, q->query, q->h->dim, q->h->cache[idx], q->qcache);
}
/* hnsw_index.rs:672-697 */
size_t orc_hnsw_knn_pq(orc_hnsw *h, const orc_pq *pq, const float *query, size_t k, size_t ef, uint64_t *out_idx,
                       float *out_dist) {
    if (h->n == 0) return 0;
    float *lut = (float *)malloc(pq->m * pq->k * sizeof(float));
    float qc = orc_pq_lookup(pq, query, lut);
    adc_ctx a = {pq, lut, qc, h};
    if (ef < k) ef = k;
    uint64_t ep = greedy_until_level_fn(h, 0, adc_dist, &a);
    scratch_t s = {0};
    rset r;
    search_on_level_fn(h, ep, 0, ef, adc_dist, &a, &s, &r);
    hq_ctx q = {h, query, orc_dist_cache(h->dist, query, h->dim)};
    size_t cnt = pq_resort(&r, k, hq_dist_nostat, &q, out_idx, out_dist);
    rset_free(&r);
    scratch_free(&s);
    free(lut);
    return cnt;
}

typedef struct {
    orc_hnsw *h;
    const float *queries;
    size_t nq, k, ef;
    uint64_t *out_idx;
    float *out_dist;
    uint64_t *out_count;
    size_t *next;
    pthread_mutex_t *mu;
    uint64_t n_dist, n_exp;
} hnsw_job;

static void *hnsw_worker(void *arg) {
    hnsw_job *j = (hnsw_job *)arg;
    /* private shallow copy so the stat counters do not race */
    orc_hnsw local = *j->h;
    local.stat_n_dist = local.stat_n_expanded = 0;
    scratch_t s = {0};
    for (;;) {
        pthread_mutex_lock(j->mu);
        size_t q = (*j->next)++;
        pthread_mutex_unlock(j->mu);
        if (q >= j->nq) break;
        size_t c = hnsw_knn_scratch(&local, j->queries + q * local.dim, j->k, j->ef, j->out_idx + q * j->k,
                                    j->out_dist + q * j->k, &s);
        if (j->out_count) j->out_count[q] = c;
    }
    scratch_free(&s);
    pthread_mutex_lock(j->mu);
    j->n_dist += local.stat_n_dist;
    j->n_exp += local.stat_n_expanded;
    pthread_mutex_unlock(j->mu);
    return NULL;
}

void orc_hnsw_knn_batch(orc_hnsw *h, const float *queries, size_t nq, size_t k, size_t ef, uint64_t *out_idx,
                        float *out_dist, uint64_t *out_count, int nthreads, uint64_t *stat_n_dist,
                        uint64_t *stat_n_expanded) {
    size_t next = 0;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    hnsw_job j = {h, queries, nq, k, ef, out_idx, out_dist, out_count, &next, &mu, 0, 0};
    if (nthreads <= 1) {
        hnsw_worker(&j);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, hnsw_worker, &j);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        free(th);
    }
    if (stat_n_dist) *stat_n_dist = j.n_dist;
    if (stat_n_expanded) *stat_n_expanded = j.n_exp;
}
