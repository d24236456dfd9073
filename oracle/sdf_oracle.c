/*
 * sdf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference hot path
 *   sdf_tools/include/sdf_tools/sdf_generation.hpp
 * written from scratch in plain C for use as the parity oracle and as the
 * timed "cpu_baseline" (kind="port").  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (libsdfgpu.so) never links or calls it.
 *
 * Pinning status: the reference header cannot be compiled in this image
 * (needs Eigen, arc_utilities, ROS message headers -- all absent), so
 * oracle/_ref is "unbuildable here".  The oracle is pinned instead against
 * the reference's only assertion-bearing test (test/test_bindings.py:24-33)
 * and the known-answer scenes the reference ships (tutorial, convex-segments);
 * see tests/test_oracle_golden.py and tests/golden/.
 *
 * Every function cites the reference lines it restates.  The algorithm is
 * the reference's *inexact* bucket-queue propagation, reproduced literally
 * (same seeding order, same neighbourhood tables, same bucket order, queue
 * entries are snapshots taken at push time, no stale-entry check).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- sdf_generation.hpp:19-25  struct bucket_cell ------------------------
 * The reference stores distance_square as double (always an exact integer or
 * +inf).  We keep the same field set; +inf is represented by DSQ_INF. */
typedef struct {
    double distance_square;
    int32_t update_direction;
    uint32_t location[3];
    uint32_t closest_point[3];
} bucket_cell;

typedef struct {
    bucket_cell *items;
    size_t len, cap;
} bucket;

static int bucket_push(bucket *b, const bucket_cell *c) {
    if (b->len == b->cap) {
        size_t ncap = b->cap ? b->cap * 2 : 8;
        bucket_cell *p = (bucket_cell *)realloc(b->items, ncap * sizeof(bucket_cell));
        if (!p) return -1;
        b->items = p;
        b->cap = ncap;
    }
    b->items[b->len++] = *c;
    return 0;
}

/* sdf_generation.hpp:29-32 */
static int get_direction_number(int dx, int dy, int dz) {
    return ((dx + 1) * 9) + ((dy + 1) * 3) + (dz + 1);
}

/* sdf_generation.hpp:34-85  MakeNeighborhoods.
 * nbh[n][dir] = list of (tdx,tdy,tdz); n=0: all 26; n=1: axis steps that do
 * not oppose the source direction.  Enumeration order (tdx, tdy, tdz nested,
 * -1..1) is kept because it fixes FIFO tie-breaking inside buckets. */
typedef struct {
    int count;
    int8_t d[26][3];
} nbh_list;

static void make_neighborhoods(nbh_list nbh[2][27]) {
    for (int n = 0; n < 2; n++) {
        for (int dx = -1; dx <= 1; dx++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dz = -1; dz <= 1; dz++) {
                    nbh_list *l = &nbh[n][get_direction_number(dx, dy, dz)];
                    l->count = 0;
                    for (int tdx = -1; tdx <= 1; tdx++)
                        for (int tdy = -1; tdy <= 1; tdy++)
                            for (int tdz = -1; tdz <= 1; tdz++) {
                                if (tdx == 0 && tdy == 0 && tdz == 0) continue;
                                if (n >= 1) {
                                    if ((abs(tdx) + abs(tdy) + abs(tdz)) != 1) continue;
                                    if ((dx * tdx) < 0 || (dy * tdy) < 0 || (dz * tdz) < 0) continue;
                                }
                                l->d[l->count][0] = (int8_t)tdx;
                                l->d[l->count][1] = (int8_t)tdy;
                                l->d[l->count][2] = (int8_t)tdz;
                                l->count++;
                            }
                }
    }
}

/* sdf_generation.hpp:87-93 (int32 arithmetic, widened to double, truncated
 * back to int at the call site :179). */
static int compute_distance_squared(int32_t x1, int32_t y1, int32_t z1,
                                    int32_t x2, int32_t y2, int32_t z2) {
    int32_t dx = x1 - x2, dy = y1 - y2, dz = z1 - z2;
    return (int)((double)((dx * dx) + (dy * dy) + (dz * dz)));
}

/* sdf_generation.hpp:95-207  BuildDistanceField.
 * `seed` is a byte per voxel (z fastest: idx = x*ny*nz + y*nz + z), nonzero
 * where the voxel belongs to the seed set; seeds are enqueued in x->y->z
 * order exactly as the `points` vector is built at :219-240.
 * out_dsq[idx] receives distance_square as double (INFINITY if unreached).
 * Returns 0, or -1 on allocation failure. */
static int build_distance_field(int64_t nx, int64_t ny, int64_t nz,
                                const uint8_t *seed, int seed_value,
                                double *out_dsq) {
    const int64_t n = nx * ny * nz;
    bucket_cell *field = (bucket_cell *)malloc((size_t)(n > 0 ? n : 1) * sizeof(bucket_cell));
    if (!field) return -1;
    for (int64_t i = 0; i < n; i++) {           /* :103-105 default cell */
        field[i].distance_square = INFINITY;
        field[i].update_direction = 0;
        memset(field[i].location, 0, sizeof field[i].location);
        memset(field[i].closest_point, 0, sizeof field[i].closest_point);
    }
    const long max_distance_square = (long)(nx * nx + ny * ny + nz * nz);   /* :107 */
    bucket *queue = (bucket *)calloc((size_t)max_distance_square + 1, sizeof(bucket));  /* :109 */
    if (!queue) { free(field); return -1; }
    const int initial_dir = get_direction_number(0, 0, 0);                 /* :112 */
    int rc = 0;
    /* :114-135 seed bucket 0 */
    for (int64_t x = 0; x < nx && !rc; x++)
        for (int64_t y = 0; y < ny && !rc; y++)
            for (int64_t z = 0; z < nz; z++) {
                const int64_t idx = (x * ny + y) * nz + z;
                if ((seed[idx] != 0) != (seed_value != 0)) continue;
                bucket_cell *c = &field[idx];
                c->location[0] = (uint32_t)x; c->location[1] = (uint32_t)y; c->location[2] = (uint32_t)z;
                c->closest_point[0] = (uint32_t)x; c->closest_point[1] = (uint32_t)y; c->closest_point[2] = (uint32_t)z;
                c->distance_square = 0.0;
                c->update_direction = initial_dir;
                if (bucket_push(&queue[0], c)) { rc = -1; break; }
            }
    nbh_list nbh[2][27];
    make_neighborhoods(nbh);                                                /* :137 */
    /* :138-205 process buckets in increasing d^2 */
    for (long b = 0; b <= max_distance_square && !rc; b++) {
        bucket *bq = &queue[b];
        /* Pushes made while iterating bucket b never land in bucket b
         * (nd2 - b is odd for axis steps; 1..3 for the 26-neighbourhood), so
         * indexing by position is equivalent to the reference's iterator. */
        for (size_t qi = 0; qi < bq->len; qi++) {
            const bucket_cell cur = bq->items[qi];       /* snapshot copy, :144 */
            const double x = cur.location[0], y = cur.location[1], z = cur.location[2];
            int D = (int)b;                               /* :149-153 */
            if (D > 1) D = 1;
            if (cur.update_direction < 0 || cur.update_direction > 26) continue;   /* :155-159 */
            const nbh_list *l = &nbh[D][cur.update_direction];
            for (int k = 0; k < l->count; k++) {
                const int dx = l->d[k][0], dy = l->d[k][1], dz = l->d[k][2];
                const int nxi = (int)(x + dx), nyi = (int)(y + dy), nzi = (int)(z + dz);  /* :169-171 */
                if (nxi < 0 || nyi < 0 || nzi < 0 || nxi >= nx || nyi >= ny || nzi >= nz) continue;  /* :172-177 */
                bucket_cell *nb = &field[((int64_t)nxi * ny + nyi) * nz + nzi];
                const int nd2 = compute_distance_squared(nxi, nyi, nzi,
                                                         (int32_t)cur.closest_point[0],
                                                         (int32_t)cur.closest_point[1],
                                                         (int32_t)cur.closest_point[2]);  /* :179 */
                if (nd2 > max_distance_square) continue;                    /* :180-184 */
                if (nd2 < nb->distance_square) {                            /* :185 */
                    nb->distance_square = nd2;
                    nb->closest_point[0] = cur.closest_point[0];
                    nb->closest_point[1] = cur.closest_point[1];
                    nb->closest_point[2] = cur.closest_point[2];
                    nb->location[0] = (uint32_t)nxi; nb->location[1] = (uint32_t)nyi; nb->location[2] = (uint32_t)nzi;
                    nb->update_direction = get_direction_number(dx, dy, dz);
                    if (bucket_push(&queue[nd2], nb)) { rc = -1; break; }   /* :197 */
                }
            }
            if (rc) break;
        }
        free(bq->items);                                                    /* :204 clear */
        bq->items = NULL; bq->len = bq->cap = 0;
    }
    for (long b = 0; b <= max_distance_square; b++) free(queue[b].items);
    free(queue);
    if (!rc) for (int64_t i = 0; i < n; i++) out_dsq[i] = field[i].distance_square;
    free(field);
    return rc;
}

/* sdf_generation.hpp:209-271  ExtractSignedDistanceField (dims + predicate core).
 * filled: byte mask (nonzero = is_filled_fn true), z fastest.
 * out_sdf: N floats.  out_extrema = {max, min} of the un-narrowed doubles.
 * out_dsq_filled / out_dsq_free (optional, may be NULL) receive the two
 * BuildDistanceField results so tests can compare integer d^2 directly. */
int sdf_oracle_extract(const uint8_t *filled, int64_t nx, int64_t ny, int64_t nz,
                       double resolution, float *out_sdf, double *out_extrema,
                       double *out_dsq_filled, double *out_dsq_free) {
    const int64_t n = nx * ny * nz;
    double *df = out_dsq_filled ? out_dsq_filled : (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    double *de = out_dsq_free ? out_dsq_free : (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    int rc = (!df || !de) ? -1 : 0;
    if (!rc) rc = build_distance_field(nx, ny, nz, filled, 1, df);   /* :242 distance to filled */
    if (!rc) rc = build_distance_field(nx, ny, nz, filled, 0, de);   /* :243 distance to free   */
    if (!rc) {
        double max_distance = -INFINITY, min_distance = INFINITY;   /* :246-247 */
        for (int64_t i = 0; i < n; i++) {                           /* :248-268, same x->y->z order */
            const double distance1 = sqrt(df[i]) * resolution;
            const double distance2 = sqrt(de[i]) * resolution;
            const double distance = distance1 - distance2;
            if (distance > max_distance) max_distance = distance;
            if (distance < min_distance) min_distance = distance;
            out_sdf[i] = (float)distance;                           /* SetValue(float) :265 */
        }
        out_extrema[0] = max_distance;
        out_extrema[1] = min_distance;
    }
    if (!out_dsq_filled) free(df);
    if (!out_dsq_free) free(de);
    return rc;
}

/* sdf_generation.hpp:273-420  grid overload with add_virtual_border.
 * vb == 0 forwards to the core (:282-286).  vb != 0 pads every axis with
 * n > 1 by one layer per side, runs the core twice (border counted as filled,
 * then as free) and combines (:385-418). */
int sdf_oracle_extract_vb(const uint8_t *filled, int64_t nx, int64_t ny, int64_t nz,
                          double resolution, int add_virtual_border,
                          float *out_sdf, double *out_extrema) {
    if (!add_virtual_border)
        return sdf_oracle_extract(filled, nx, ny, nz, resolution, out_sdf, out_extrema, NULL, NULL);
    const int64_t xo = nx > 1 ? 1 : 0, yo = ny > 1 ? 1 : 0, zo = nz > 1 ? 1 : 0;  /* :289-294 */
    const int64_t px = nx + 2 * xo, py = ny + 2 * yo, pz = nz + 2 * zo;          /* :296-298 */
    const int64_t pn = px * py * pz;
    uint8_t *m_free = (uint8_t *)malloc((size_t)pn);    /* border = filled (:301-339) */
    uint8_t *m_fill = (uint8_t *)malloc((size_t)pn);    /* border = free   (:341-379) */
    float *s_free = (float *)malloc((size_t)pn * sizeof(float));
    float *s_fill = (float *)malloc((size_t)pn * sizeof(float));
    int rc = (!m_free || !m_fill || !s_free || !s_fill) ? -1 : 0;
    double ex_free[2], ex_fill[2];
    if (!rc) {
        for (int64_t x = 0; x < px; x++)
            for (int64_t y = 0; y < py; y++)
                for (int64_t z = 0; z < pz; z++) {
                    const int64_t pi = (x * py + y) * pz + z;
                    const int border = (xo && (x == 0 || x == px - 1)) ||
                                       (yo && (y == 0 || y == py - 1)) ||
                                       (zo && (z == 0 || z == pz - 1));
                    if (border) { m_free[pi] = 1; m_fill[pi] = 0; }
                    else {
                        const uint8_t v = filled[((x - xo) * ny + (y - yo)) * nz + (z - zo)] ? 1 : 0;
                        m_free[pi] = v; m_fill[pi] = v;
                    }
                }
        rc = sdf_oracle_extract(m_free, px, py, pz, resolution, s_free, ex_free, NULL, NULL);   /* :381 */
        if (!rc) rc = sdf_oracle_extract(m_fill, px, py, pz, resolution, s_fill, ex_fill, NULL, NULL);  /* :382 */
    }
    if (!rc) {
        for (int64_t x = 0; x < nx; x++)                 /* :385-414 */
            for (int64_t y = 0; y < ny; y++)
                for (int64_t z = 0; z < nz; z++) {
                    const int64_t pi = ((x + xo) * py + (y + yo)) * pz + (z + zo);
                    const float fv = s_free[pi], gv = s_fill[pi];
                    float r;
                    if (fv >= 0.0) r = fv;
                    else if (gv <= -0.0) r = gv;
                    else r = 0.0f;
                    out_sdf[(x * ny + y) * nz + z] = r;
                }
        out_extrema[0] = ex_free[0];                     /* :416-417 (free.max, filled.min) */
        out_extrema[1] = ex_fill[1];
    }
    free(m_free); free(m_fill); free(s_free); free(s_fill);
    return rc;
}

/* collision_map.hpp:680-712  CollisionMapGrid::ExtractSignedDistanceField predicate,
 * applied to an array of COLLISION_CELL {float occupancy; uint32 component}
 * (collision_map.hpp:20-32).  Writes the byte mask used by the functions above. */
void sdf_oracle_classify_cells(const void *cells, int64_t n, int unknown_is_filled, uint8_t *out_mask) {
    const uint8_t *p = (const uint8_t *)cells;
    for (int64_t i = 0; i < n; i++) {
        float occ;
        memcpy(&occ, p + 8 * i, sizeof occ);
        int f = 0;
        if (occ > 0.5) f = 1;
        else if (unknown_is_filled && (occ == 0.5)) f = 1;
        out_mask[i] = (uint8_t)f;
    }
}

/* ------------------------------------------------------------------------
 * Exact squared EDT (second opinion / tie-breaker).
 * Independent of the propagation above: three separable passes with the
 * Felzenszwalb-Huttenlocher lower envelope in integer arithmetic (Meijster
 * style separator), giving for every voxel the exact squared distance to the
 * nearest voxel v with (seed[v] != 0) == (seed_value != 0), or -1 if none.
 * ---------------------------------------------------------------------- */
#define EDT_INF ((int64_t)1 << 40)

static void edt_line(const int64_t *f, int64_t *d, int64_t n, int64_t *v, int64_t *zb) {
    /* Lower envelope of the parabolas p -> f[q] + (p-q)^2 over sites q with
     * finite f[q].  v[0..k] = sites on the envelope, zb[j] = first integer
     * position where v[j] is strictly better than v[j-1].  For q > vk, q beats
     * vk at p iff 2p(q-vk) > f[q]-f[vk]+q^2-vk^2, all in exact integers. */
    int64_t k = -1;
    for (int64_t q = 0; q < n; q++) {
        if (f[q] >= EDT_INF) continue;
        for (;;) {
            if (k < 0) { k = 0; v[0] = q; zb[0] = INT64_MIN / 4; break; }
            const int64_t vk = v[k];
            const int64_t num = f[q] - f[vk] + q * q - vk * vk;
            const int64_t den = 2 * (q - vk);
            const int64_t fl = num >= 0 ? num / den : -((-num + den - 1) / den);  /* floor */
            const int64_t s = fl + 1;
            if (s <= zb[k]) { k--; continue; }
            k++; v[k] = q; zb[k] = s;
            break;
        }
    }
    if (k < 0) { for (int64_t p = 0; p < n; p++) d[p] = EDT_INF; return; }
    int64_t j = 0;
    for (int64_t p = 0; p < n; p++) {
        while (j < k && zb[j + 1] <= p) j++;
        const int64_t q = v[j];
        d[p] = f[q] + (p - q) * (p - q);
    }
}

int sdf_oracle_exact_edt(const uint8_t *seed, int seed_value, int64_t nx, int64_t ny, int64_t nz,
                         int64_t *out_dsq) {
    const int64_t n = nx * ny * nz;
    int64_t maxd = nx > ny ? nx : ny; if (nz > maxd) maxd = nz;
    int64_t *f = (int64_t *)malloc((size_t)maxd * sizeof(int64_t));
    int64_t *d = (int64_t *)malloc((size_t)maxd * sizeof(int64_t));
    int64_t *v = (int64_t *)malloc((size_t)maxd * sizeof(int64_t));
    int64_t *zb = (int64_t *)malloc((size_t)(maxd + 1) * sizeof(int64_t));
    if (!f || !d || !v || !zb) { free(f); free(d); free(v); free(zb); return -1; }
    for (int64_t i = 0; i < n; i++)
        out_dsq[i] = ((seed[i] != 0) == (seed_value != 0)) ? 0 : EDT_INF;
    /* z lines */
    for (int64_t x = 0; x < nx; x++) for (int64_t y = 0; y < ny; y++) {
        int64_t *row = out_dsq + (x * ny + y) * nz;
        for (int64_t z = 0; z < nz; z++) f[z] = row[z];
        edt_line(f, d, nz, v, zb);
        for (int64_t z = 0; z < nz; z++) row[z] = d[z];
    }
    /* y lines */
    for (int64_t x = 0; x < nx; x++) for (int64_t z = 0; z < nz; z++) {
        for (int64_t y = 0; y < ny; y++) f[y] = out_dsq[(x * ny + y) * nz + z];
        edt_line(f, d, ny, v, zb);
        for (int64_t y = 0; y < ny; y++) out_dsq[(x * ny + y) * nz + z] = d[y];
    }
    /* x lines */
    for (int64_t y = 0; y < ny; y++) for (int64_t z = 0; z < nz; z++) {
        for (int64_t x = 0; x < nx; x++) f[x] = out_dsq[(x * ny + y) * nz + z];
        edt_line(f, d, nx, v, zb);
        for (int64_t x = 0; x < nx; x++) out_dsq[(x * ny + y) * nz + z] = d[x];
    }
    for (int64_t i = 0; i < n; i++) if (out_dsq[i] >= EDT_INF) out_dsq[i] = -1;
    free(f); free(d); free(v); free(zb);
    return 0;
}

/* O(N * sites) brute force, for tiny grids only: validates the exact EDT. */
int sdf_oracle_brute_edt(const uint8_t *seed, int seed_value, int64_t nx, int64_t ny, int64_t nz,
                         int64_t *out_dsq) {
    const int64_t n = nx * ny * nz;
    for (int64_t i = 0; i < n; i++) {
        const int64_t x = i / (ny * nz), y = (i / nz) % ny, z = i % nz;
        int64_t best = -1;
        for (int64_t j = 0; j < n; j++) {
            if ((seed[j] != 0) != (seed_value != 0)) continue;
            const int64_t dx = x - j / (ny * nz), dy = y - (j / nz) % ny, dz = z - j % nz;
            const int64_t d2 = dx * dx + dy * dy + dz * dz;
            if (best < 0 || d2 < best) best = d2;
        }
        out_dsq[i] = best;
    }
    return 0;
}

/* Exact signed field with the reference's merge arithmetic (:254-265) applied
 * to exact d^2:  sdf = filled ? -sqrt(D_free)*res : +sqrt(D_filled)*res, with
 * the optional virtual-border clamp D = min(D, b^2) (net effect of :287-419).
 * out_dsq (optional) receives the signed integer d^2 (+ for free voxels,
 * - for filled; INT64_MAX magnitude for "no opposite voxel"). */
int sdf_oracle_exact_sdf(const uint8_t *filled, int64_t nx, int64_t ny, int64_t nz,
                         double resolution, int add_virtual_border,
                         float *out_sdf, double *out_extrema, int64_t *out_dsq) {
    const int64_t n = nx * ny * nz;
    int64_t *dfil = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    int64_t *dfre = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    if (!dfil || !dfre) { free(dfil); free(dfre); return -1; }
    int rc = sdf_oracle_exact_edt(filled, 1, nx, ny, nz, dfil);
    if (!rc) rc = sdf_oracle_exact_edt(filled, 0, nx, ny, nz, dfre);
    if (!rc) {
        double mx = -INFINITY, mn = INFINITY;
        int any_free = 0, any_filled = 0;
        for (int64_t i = 0; i < n; i++) {
            const int64_t x = i / (ny * nz), y = (i / nz) % ny, z = i % nz;
            const int is_filled = filled[i] != 0;
            int64_t D = is_filled ? dfre[i] : dfil[i];      /* -1 = none */
            if (add_virtual_border) {
                int64_t b = -1;
                if (nx > 1) { int64_t t = x + 1 < nx - x ? x + 1 : nx - x; if (b < 0 || t < b) b = t; }
                if (ny > 1) { int64_t t = y + 1 < ny - y ? y + 1 : ny - y; if (b < 0 || t < b) b = t; }
                if (nz > 1) { int64_t t = z + 1 < nz - z ? z + 1 : nz - z; if (b < 0 || t < b) b = t; }
                if (b >= 0 && (D < 0 || b * b < D)) D = b * b;
            }
            const double dist = (D < 0) ? INFINITY : sqrt((double)D) * resolution;
            const double sd = is_filled ? (0.0 - dist) : (dist - 0.0);
            out_sdf[i] = (float)sd;
            if (out_dsq) out_dsq[i] = (D < 0) ? (is_filled ? -INT64_MAX : INT64_MAX) : (is_filled ? -D : D);
            if (is_filled) { any_filled = 1; if (sd < mn) mn = sd; }
            else { any_free = 1; if (sd > mx) mx = sd; }
        }
        /* Extrema semantics of the reference loops (:246-269, :416-418):
         * max comes from a free voxel, min from a filled one; a class with no
         * voxels leaves max = -inf (all-filled) / min = +inf (all-free). */
        if (!any_free) mx = -INFINITY;
        if (!any_filled) mn = INFINITY;
        out_extrema[0] = mx;
        out_extrema[1] = mn;
    }
    free(dfil); free(dfre);
    return rc;
}
