/*
 * cat_oracle.c -- TEST INFRASTRUCTURE ONLY (see cat_oracle.h; PARITY UNPINNED vs Pymunk).
 *
 * CPU restatement of the reference env tick.  Citations:
 *   [REF  file:line]  the reference's own Python under /root/reference/src
 *   [CP   name]       Chipmunk2D 7.0.x function whose published algorithm is restated
 *                     (third-party, not in /root/reference)
 * Compile with -ffp-contract=off: every multiply/add below rounds separately, as in a
 * non-FMA x86-64 build of Chipmunk, and as the HIP kernels are built.
 *
 * Stated deviations from Chipmunk (all far below the 1e-5 position tolerance):
 *   D1  bodies do not rotate: angular velocity terms (O(1e-12)) are dropped.
 *   D2  the spatial index is a linear list: shapes are visited in index order (walls, then
 *       agents), which fixes tie-breaks and solver order.  The BBTree's pruning rule is kept
 *       (visit iff the bb entry value is below the best alpha so far); its visiting ORDER (nearer child first) is
 *       available as a diagnostic (cato_set_index_order(0)): see segment_query_first.
 *   D3  bb slab test multiplies by 1/delta instead of dividing (gate decision only).
 *   D4  centre-inside-hull contacts use least-penetration instead of EPA (unreachable in
 *       play: needs > 5 px penetration).
 *   D5  spawn sampling draws from Philox4x32-10 counter streams.
 *   D6  Chipmunk's SubtreeSegmentQuery does not gate a BBTree root that is itself a leaf: with ONE
 *       static shape in the space that shape is queried even when the thin segment misses its bb.
 *       The linear index here gates every wall alike (bb_gate below).  None of the five maps has a
 *       single wall; tests/test_oracle_known_answers.py::test_single_wall_map_... shows the case.
 */
#include "cat_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOB_MAGIC 0x31544143
#define K_WALL CATO_WALL_CACHE
#define MAX_CONTACTS (CATO_MAX_AGENTS * K_WALL + CATO_MAX_PAIRS)

typedef struct {
    int S, P, A, n_cops, n_thieves, n_regions;
    double win_w, win_h;
    double *bb;       /* [S][4] */
    double *planes;   /* [P][8] n.x n.y v0.x v0.y vn dtMin dtMax pad */
    double *start;    /* [A][2] */
    double *regions;  /* [Rg][4] */
    int32_t *first, *count, *region_off;
    struct bbt_node *tree;   /* diagnostic (cato_set_index_order(2)): Chipmunk's static BBTree over the walls, restated */
    int tree_root;
} cato_map;

struct cato_sim {
    cato_config cfg;
    int N, A, R, NP, n_maps;
    cato_map *maps;
    int32_t *slot_map;
    double *ray_dx, *ray_dy;
    float *cop_lut, *thief_lut;
    /* state, same layout as cato_state */
    double *pos, *vel, *vbias, *tc, *leaf_bb, *wall_jn, *pair_jn;
    int32_t *wall_shape, *wall_age, *pair_age, *step_count, *reset_count;
};

static char g_err[256];
static int g_threads = 1;
const char *cato_last_error(void) { return g_err; }
void cato_set_threads(int n) { g_threads = n < 1 ? 1 : n; }

/* ---------------------------------------------------------------- float16 ---------- */
/* round-to-nearest-even double -> half, the conversion NumPy applies for
   np.array(points, dtype=np.float16) [REF entity.py:206] and for casting origin_x [REF :208] */
uint16_t cato_f64_to_f16(double x)
{
    uint64_t b;
    memcpy(&b, &x, 8);
    uint16_t sign = (uint16_t)((b >> 48) & 0x8000u);
    uint64_t a = b & 0x7FFFFFFFFFFFFFFFull;
    if (a >= 0x7FF0000000000000ull)
        return (uint16_t)(sign | (a > 0x7FF0000000000000ull ? 0x7E00u : 0x7C00u));
    if (a == 0) return sign;
    int e = (int)(a >> 52) - 1023;
    if (e > 15) return (uint16_t)(sign | 0x7C00u);
    uint64_t M = (a & 0xFFFFFFFFFFFFFull) | (1ull << 52);
    int shift = 42;
    if (e < -14) shift += (-14 - e);
    if (shift > 54) return sign;
    uint64_t q = M >> shift;
    uint64_t rem = M & ((1ull << shift) - 1);
    uint64_t half = 1ull << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) q++;
    uint32_t bits = (e >= -14) ? (uint32_t)(((uint32_t)(e + 14) << 10) + q) : (uint32_t)q;
    if (bits >= 0x7C00u) bits = 0x7C00u;
    return (uint16_t)(sign | bits);
}

double cato_f16_to_f64(uint16_t h)
{
    int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    double v;
    if (e == 0) v = ldexp((double)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexp((double)(m | 1024), e - 25);
    return s ? -v : v;
}

static inline float h2f(uint16_t h) { return (float)cato_f16_to_f64(h); }
static inline uint16_t f2h(float f) { return cato_f64_to_f16((double)f); }

/* [REF entity.py:206-210] + SURVEY quirk Q3: points -> f16, origin -> f16 (weak python float),
   f16 - f16 via f32, np.hypot on f16 = hypotf on f32, result -> f16 */
uint16_t cato_obs_distance_f16(double px, double py, double ox, double oy)
{
    float dx32 = h2f(cato_f64_to_f16(px)) - h2f(cato_f64_to_f16(ox));
    float dy32 = h2f(cato_f64_to_f16(py)) - h2f(cato_f64_to_f16(oy));
    float dx = h2f(f2h(dx32)), dy = h2f(f2h(dy32));
    float hyp = (float)sqrt((double)dx * (double)dx + (double)dy * (double)dy);
    return f2h(hyp);
}

/* ---------------------------------------------------------------- Philox4x32-10 ----- */
void cato_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void philox_env(const cato_sim *s, int env, uint32_t c1, uint32_t c2, uint32_t c3,
                       uint32_t out[4])
{
    uint64_t gid = (uint64_t)(s->cfg.env_id_offset + env);
    uint32_t ctr[4] = {(uint32_t)gid, c1, c2, c3 ^ ((uint32_t)(gid >> 32) << 24)};
    uint32_t key[2] = {(uint32_t)s->cfg.seed, (uint32_t)(s->cfg.seed >> 32)};
    cato_philox4x32(ctr, key, out);
}

/* 53-bit uniform in [0,1), the construction of Python's random.random() [REF map_utils.py:9] */
static inline double u53(uint32_t a, uint32_t b)
{
    return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
}

/* ---------------------------------------------------------------- geometry ---------- */
static inline double fmax2(double a, double b) { return (a > b) ? a : b; } /* [CP cpfmax] */
static inline double fmin2(double a, double b) { return (a < b) ? a : b; } /* [CP cpfmin] */

/* [CP cpBBSegmentQuery] with deviation D3. Returns INFINITY when the thin segment misses. */
static double bb_segment_query(const double *bb, double ax, double ay, double dx, double dy,
                               double idx, double idy)
{
    double tmin = -INFINITY, tmax = INFINITY;
    if (dx == 0.0) {
        if (ax < bb[0] || bb[2] < ax) return INFINITY;
    } else {
        double t1 = (bb[0] - ax) * idx, t2 = (bb[2] - ax) * idx;
        tmin = fmax2(tmin, fmin2(t1, t2));
        tmax = fmin2(tmax, fmax2(t1, t2));
    }
    if (dy == 0.0) {
        if (ay < bb[1] || bb[3] < ay) return INFINITY;
    } else {
        double t1 = (bb[1] - ay) * idy, t2 = (bb[3] - ay) * idy;
        tmin = fmax2(tmin, fmin2(t1, t2));
        tmax = fmin2(tmax, fmax2(t1, t2));
    }
    if (tmin <= tmax && 0.0 <= tmax && tmin <= 1.0) return fmax2(tmin, 0.0);
    return INFINITY;
}

typedef struct { int hit; double alpha, px, py; } seg_info;

/* [CP CircleSegmentQuery] */
static void circle_segment_query(double cx, double cy, double r1, double ax, double ay,
                                 double bx, double by, double r2, seg_info *info)
{
    double dax = ax - cx, day = ay - cy, dbx = bx - cx, dby = by - cy;
    double rsum = r1 + r2;
    double dada = dax * dax + day * day, dadb = dax * dbx + day * dby, dbdb = dbx * dbx + dby * dby;
    double qa = dada - 2.0 * dadb + dbdb;
    double qb = dadb - dada;
    double det = qb * qb - qa * (dada - rsum * rsum);
    if (det >= 0.0) {
        double t = (-qb - sqrt(det)) / qa;
        if (0.0 <= t && t <= 1.0) {
            /* n = cpvnormalize(cpvlerp(da, db, t)) */
            double lx = dax * (1.0 - t) + dbx * t, ly = day * (1.0 - t) + dby * t;
            double inv = 1.0 / (sqrt(lx * lx + ly * ly) + DBL_MIN);
            double nx = lx * inv, ny = ly * inv;
            info->hit = 1;
            info->alpha = t;
            info->px = (ax * (1.0 - t) + bx * t) - nx * r2;
            info->py = (ay * (1.0 - t) + by * t) - ny * r2;
        }
    }
}

/* [CP cpPolyShapeSegmentQuery]; vn/dtMin/dtMax pre-folded by the map compiler */
static void poly_segment_query(const cato_map *m, int sh, double r, double ax, double ay,
                               double bx, double by, double r2, seg_info *info)
{
    int first = m->first[sh], count = m->count[sh];
    double rsum = r + r2;
    for (int i = 0; i < count; i++) {
        const double *pl = m->planes + 8 * (size_t)(first + i);
        double nx = pl[0], ny = pl[1];
        double an = ax * nx + ay * ny;
        double d = an - pl[4] - rsum;
        if (d < 0.0) continue;
        double bn = bx * nx + by * ny;
        double t = d / fmax2(an - bn, DBL_MIN);
        if (t < 0.0 || 1.0 < t) continue;
        double ptx = ax * (1.0 - t) + bx * t, pty = ay * (1.0 - t) + by * t;
        double dtv = nx * pty - ny * ptx;
        if (pl[5] <= dtv && dtv <= pl[6]) {
            info->hit = 1;
            info->alpha = t;
            info->px = ptx - nx * r2;
            info->py = pty - ny * r2;
        }
    }
    if (rsum > 0.0) {
        for (int i = 0; i < count; i++) {
            const double *pl = m->planes + 8 * (size_t)(first + i);
            seg_info ci = {0, 1.0, bx, by};
            circle_segment_query(pl[2], pl[3], r, ax, ay, bx, by, r2, &ci);
            if (ci.alpha < info->alpha) *info = ci;
        }
    }
}

/* [CP cpPolyShapePointQuery] -> signed distance to the rounded surface */
static double poly_point_distance(const cato_map *m, int sh, double r, double px, double py)
{
    int first = m->first[sh], count = m->count[sh];
    const double *last = m->planes + 8 * (size_t)(first + count - 1);
    double v0x = last[2], v0y = last[3];
    double minDist = INFINITY;
    int outside = 0;
    for (int i = 0; i < count; i++) {
        const double *pl = m->planes + 8 * (size_t)(first + i);
        double v1x = pl[2], v1y = pl[3];
        outside = outside || (pl[0] * (px - v1x) + pl[1] * (py - v1y) > 0.0);
        /* cpClosetPointOnSegment(p, v0, v1) */
        double dx = v0x - v1x, dy = v0y - v1y;
        double t = (dx * (px - v1x) + dy * (py - v1y)) / (dx * dx + dy * dy);
        t = fmax2(0.0, fmin2(t, 1.0)); /* [CP cpfclamp01] */
        double cx = v1x + dx * t, cy = v1y + dy * t;
        double ex = px - cx, ey = py - cy;
        double dist = sqrt(ex * ex + ey * ey);
        if (dist < minDist) minDist = dist;
        v0x = v1x; v0y = v1y;
    }
    double dist = outside ? minDist : -minDist;
    return dist - r;
}

/* ---------------------------------------------------------------- queries ----------- */
typedef struct {
    const cato_sim *s;
    const cato_map *m;
    int env;
} env_ctx;

static inline const double *TC(const cato_sim *s, int env, int j) { return s->tc + 2 * ((size_t)env * s->A + j); }
static inline const double *LEAF(const cato_sim *s, int env, int j) { return s->leaf_bb + 4 * ((size_t)env * s->A + j); }

/* ---- DIAGNOSTIC: Chipmunk's static spatial index restated ([CP cpBBTree.c]; Chipmunk2D 7.0.x as Pymunk 6 vendors it).  Not what the product or the
   default oracle do (D2: index order) -- cato_set_index_order(2) makes the wall part of a segment query descend THIS tree the way
   [CP SubtreeSegmentQuery] does, so that tools/query_order_diff.py can count the observations on which the two orders differ.  The tree is what
   [CP cpBBTreeInsert] builds when the walls are added in index order ([REF map.py populate_space] adds them in file order): [CP SubtreeInsert]
   descends into the child with the smaller cost  area(other child) + merged area(this child, leaf)  ([CP cpBBProximity] on equal costs; B only if
   strictly cheaper), a new inner node takes the NEW leaf as A and the subtree it met as B, and every node on the way is merged with the leaf's bb.
   Static leaves carry the plain shape bb (the static index has no velocity function).  Nothing reorders the static tree afterwards. */
struct bbt_node { double bb[4]; int a, b, obj; };   /* obj >= 0: a leaf (wall id) */
static double bbt_area(const double *x) { return (x[2] - x[0]) * (x[3] - x[1]); }
static double bbt_merged_area(const double *x, const double *y)
{
    return (fmax2(x[2], y[2]) - fmin2(x[0], y[0])) * (fmax2(x[3], y[3]) - fmin2(x[1], y[1]));
}
static double bbt_proximity(const double *x, const double *y) { return fabs(x[0] + x[2] - y[0] - y[2]) + fabs(x[1] + x[3] - y[1] - y[3]); }
static void bbt_merge(double *d, const double *x, const double *y)
{
    d[0] = fmin2(x[0], y[0]); d[1] = fmin2(x[1], y[1]); d[2] = fmax2(x[2], y[2]); d[3] = fmax2(x[3], y[3]);
}
static int bbt_insert(struct bbt_node *n, int *count, int subtree, int leaf)
{
    if (subtree < 0) return leaf;
    if (n[subtree].obj >= 0) {   /* [CP NodeNew(tree, leaf, subtree)] */
        const int k = (*count)++;
        n[k].a = leaf; n[k].b = subtree; n[k].obj = -1;
        bbt_merge(n[k].bb, n[leaf].bb, n[subtree].bb);
        return k;
    }
    const double *A = n[n[subtree].a].bb, *B = n[n[subtree].b].bb, *Lf = n[leaf].bb;
    double cost_a = bbt_area(B) + bbt_merged_area(A, Lf), cost_b = bbt_area(A) + bbt_merged_area(B, Lf);
    if (cost_a == cost_b) { cost_a = bbt_proximity(A, Lf); cost_b = bbt_proximity(B, Lf); }
    if (cost_b < cost_a) n[subtree].b = bbt_insert(n, count, n[subtree].b, leaf);
    else n[subtree].a = bbt_insert(n, count, n[subtree].a, leaf);
    double merged[4];
    bbt_merge(merged, n[subtree].bb, Lf);
    memcpy(n[subtree].bb, merged, sizeof merged);
    return subtree;
}
static void bbt_build(cato_map *m)
{
    m->tree = (struct bbt_node *)calloc((size_t)(2 * m->S + 1), sizeof(struct bbt_node));
    int count = m->S, root = -1;
    for (int s = 0; s < m->S; s++) { memcpy(m->tree[s].bb, m->bb + 4 * (size_t)s, 32); m->tree[s].a = m->tree[s].b = -1; m->tree[s].obj = s; }
    for (int s = 0; s < m->S; s++) root = bbt_insert(m->tree, &count, root, s);
    m->tree_root = root;
}

/* one candidate of a segment query: its spatial-index gate value and its place in the shape list */
typedef struct { double tbb; int id; } seg_cand;

static int g_index_order = 1;   /* 1 = visit in index order (D2, what the HIP kernels do); diagnostics: 0 = nearest-bb-first, 2 = the walls by Chipmunk's own tree descent */
void cato_set_index_order(int on) { g_index_order = on; }

/* ascending (tbb, id): insertion sort, the lists are short and nearly sorted */
static void sort_cands(seg_cand *c, int n)
{
    for (int i = 1; i < n; i++) {
        seg_cand x = c[i];
        int j = i - 1;
        while (j >= 0 && (c[j].tbb > x.tbb || (c[j].tbb == x.tbb && c[j].id > x.id))) { c[j + 1] = c[j]; j--; }
        c[j + 1] = x;
    }
}

/* [CP cpSpaceSegmentQueryFirst]: the static index is queried first (t_exit = 1), then the dynamic index with
   t_exit = the best alpha so far; the ray filter shares the agent's group, so only its own circle is rejected
   [REF entity.py:118-123]; los: the mask excludes both agent categories [REF base_env.py:536-538], so walls only.
   A shape is visited iff the value t_bb at which the thin segment enters its bb is below the best alpha so far, and
   strict '<' keeps the first of equal alphas.  ORDER of the visits (D2): index order.  Chipmunk's BBTree
   ([CP SubtreeSegmentQuery]) descends into the child whose bb the segment enters first, which for sibling leaves is
   ascending t_bb; cato_set_index_order(0) switches to that order (ascending t_bb, index on equal values) so that
   tools/query_order_diff.py can count the observations that depend on it (0.003 - 0.2 % of the rays, by map: two walls
   whose hits lie less than the ray radius apart, DESIGN D2). */
/* Diagnostic (tests of the product's candidate tables): when set, cato_segment_query visits only the walls whose byte is non-zero,
   in index order as always.  Never set by a stepping sim; single-threaded use. */
static const uint8_t *g_wall_subset = NULL;
void cato_set_wall_subset(const uint8_t *mask) { g_wall_subset = mask; }

/* one wall's [CP SegmentQueryFirst]: the shape query, and the result kept if its alpha is below the best so far */
typedef struct { const cato_map *m; const cato_config *c; double ax, ay, bx, by, dx, dy, idx, idy, r2; seg_info *out; int best; } seg_ctx;
static double visit_wall(seg_ctx *x, int sh)
{
    seg_info info = {0, 1.0, x->bx, x->by};
    /* [CP cpShapeSegmentQuery]: start point within `radius` of the shape -> alpha 0, point stays at the segment end */
    if (poly_point_distance(x->m, sh, x->c->wall_radius, x->ax, x->ay) <= x->r2) { info.hit = 1; info.alpha = 0.0; }
    else poly_segment_query(x->m, sh, x->c->wall_radius, x->ax, x->ay, x->bx, x->by, x->r2, &info);
    if (info.hit && info.alpha < x->out->alpha) { *x->out = info; x->best = sh; }
    return x->out->alpha;
}
/* DIAGNOSTIC [CP SubtreeSegmentQuery]: the child whose bb the segment enters first is descended first; a child is entered only if its entry lies
   before the best alpha so far (the gate of D3's slab test, as everywhere in this file) */
static double bbt_query(seg_ctx *x, int node, double t_exit)
{
    const struct bbt_node *n = x->m->tree + node;
    if (n->obj >= 0) return visit_wall(x, n->obj);
    const double t_a = bb_segment_query(x->m->tree[n->a].bb, x->ax, x->ay, x->dx, x->dy, x->idx, x->idy);
    const double t_b = bb_segment_query(x->m->tree[n->b].bb, x->ax, x->ay, x->dx, x->dy, x->idx, x->idy);
    if (t_a < t_b) {
        if (t_a < t_exit) t_exit = fmin2(t_exit, bbt_query(x, n->a, t_exit));
        if (t_b < t_exit) t_exit = fmin2(t_exit, bbt_query(x, n->b, t_exit));
    } else {
        if (t_b < t_exit) t_exit = fmin2(t_exit, bbt_query(x, n->b, t_exit));
        if (t_a < t_exit) t_exit = fmin2(t_exit, bbt_query(x, n->a, t_exit));
    }
    return t_exit;
}

/* Diagnostic (tools/query_order_diff.py): how many queries have a result that can depend on the ORDER of the visits at all, whatever tree Chipmunk
   builds.  With t the gate value and a the hit alpha of every candidate (t < 1) of one index, the candidate of the smallest alpha is visited under
   EVERY order iff its t is below the second-smallest alpha (the best alpha so far can never be lower than that when its turn comes); otherwise the
   order "second-smallest first" gates it out and "smallest first" does not.  Two equal smallest alphas make the winner's identity depend on the
   order.  Counted per index (walls with best = 1 at the start; agents with best = the walls' result).  Relaxed atomics: the oracle steps on threads. */
static int g_count_order = 0;
static long long g_order_counts[4];   /* queries, queries with an order-dependent wall result, with an order-dependent agent result, ties among them */
void cato_count_order_dependence(int on) { g_count_order = on; if (on) for (int i = 0; i < 4; i++) g_order_counts[i] = 0; }
void cato_order_dependence(long long out[4]) { for (int i = 0; i < 4; i++) out[i] = __atomic_load_n(&g_order_counts[i], __ATOMIC_RELAXED); }
/* alphas[i] (>= start = "no hit that could win") and gates[i] of the n candidates of one index: 1 = order-dependent, 2 = by a tie */
static int order_dependent(const double *alphas, const double *gates, int n, double start)
{
    int arg = -1;
    double a1 = start, a2 = start;
    for (int i = 0; i < n; i++) {
        if (alphas[i] < a1) { a2 = a1; a1 = alphas[i]; arg = i; }
        else if (alphas[i] < a2) a2 = alphas[i];
    }
    if (arg < 0) return 0;                         /* nothing can win: the result is "no hit" (or the walls' hit) under every order */
    int ties = 0;
    for (int i = 0; i < n; i++) ties += alphas[i] == a1;
    if (ties > 1) return 2;
    return (a2 < start && !(gates[arg] < a2)) ? 1 : 0;
}

static int segment_query_first(const cato_sim *s, int env, int self, double ax, double ay,
                               double bx, double by, double r2, int los, seg_info *out)
{
    const cato_map *m = &s->maps[s->slot_map[env]];
    const cato_config *c = &s->cfg;
    int best = -1;
    out->hit = 0; out->alpha = 1.0; out->px = bx; out->py = by;
    double dx = bx - ax, dy = by - ay;
    double idx = 1.0 / dx, idy = 1.0 / dy;
    double t_exit = 1.0;
    seg_cand cand[256 + CATO_MAX_AGENTS];
    int n = 0;
    for (int sh = 0; sh < m->S; sh++) {
        if (g_wall_subset && !g_wall_subset[sh]) continue;   /* diagnostic: the walls a caller's candidate table lists */
        double tbb = c->bbtree_gate ? bb_segment_query(m->bb + 4 * (size_t)sh, ax, ay, dx, dy, idx, idy) : 0.0;
        if (tbb < 1.0 || !c->bbtree_gate) { cand[n].tbb = tbb; cand[n].id = sh; n++; }
    }
    if (!g_index_order) sort_cands(cand, n);
    if (g_count_order && !los && c->bbtree_gate) {   /* diagnostic: every candidate's alpha, gated or not */
        double al[256], gt[256];
        for (int q = 0; q < n; q++) {
            seg_info info = {0, 1.0, bx, by};
            if (poly_point_distance(m, cand[q].id, c->wall_radius, ax, ay) <= r2) { info.hit = 1; info.alpha = 0.0; }
            else poly_segment_query(m, cand[q].id, c->wall_radius, ax, ay, bx, by, r2, &info);
            al[q] = info.hit ? info.alpha : 2.0; gt[q] = cand[q].tbb;
        }
        const int dep = order_dependent(al, gt, n, 1.0);
        __atomic_fetch_add(&g_order_counts[0], 1, __ATOMIC_RELAXED);
        if (dep) __atomic_fetch_add(&g_order_counts[1], 1, __ATOMIC_RELAXED);
        if (dep == 2) __atomic_fetch_add(&g_order_counts[3], 1, __ATOMIC_RELAXED);
    }
    seg_ctx ctx = {m, c, ax, ay, bx, by, dx, dy, idx, idy, r2, out, -1};
    if (g_index_order == 2 && c->bbtree_gate && !g_wall_subset && m->tree_root >= 0) {   /* diagnostic: Chipmunk's own descent of its static tree */
        t_exit = fmin2(t_exit, bbt_query(&ctx, m->tree_root, t_exit));
        n = 0;
    }
    for (int q = 0; q < n; q++) {
        const int sh = cand[q].id;
        if (c->bbtree_gate && !(cand[q].tbb < t_exit)) { if (g_index_order) continue; else break; }
        t_exit = fmin2(t_exit, visit_wall(&ctx, sh));
    }
    best = ctx.best;
    if (!los) {
        n = 0;
        for (int j = 0; j < s->A; j++) {
            if (j == self) continue;
            double tbb = c->bbtree_gate ? bb_segment_query(LEAF(s, env, j), ax, ay, dx, dy, idx, idy) : 0.0;
            if (tbb < 1.0 || !c->bbtree_gate) { cand[n].tbb = tbb; cand[n].id = j; n++; }
        }
        if (!g_index_order) sort_cands(cand, n);
        if (g_count_order && c->bbtree_gate && n > 1) {   /* diagnostic: the agents' index, entered with best = the walls' result */
            double al[CATO_MAX_AGENTS], gt[CATO_MAX_AGENTS];
            for (int q = 0; q < n; q++) {
                const double *tc = TC(s, env, cand[q].id);
                seg_info info = {0, 1.0, bx, by};
                double ex = ax - tc[0], ey = ay - tc[1];
                if (sqrt(ex * ex + ey * ey) - c->agent_radius <= r2) { info.hit = 1; info.alpha = 0.0; }
                else circle_segment_query(tc[0], tc[1], c->agent_radius, ax, ay, bx, by, r2, &info);
                al[q] = info.hit ? info.alpha : 2.0; gt[q] = cand[q].tbb;
            }
            const int dep = order_dependent(al, gt, n, out->alpha);
            if (dep) __atomic_fetch_add(&g_order_counts[2], 1, __ATOMIC_RELAXED);
            if (dep == 2) __atomic_fetch_add(&g_order_counts[3], 1, __ATOMIC_RELAXED);
        }
        for (int q = 0; q < n; q++) {
            const int j = cand[q].id;
            if (c->bbtree_gate && !(cand[q].tbb < t_exit)) { if (g_index_order) continue; else break; }
            const double *tc = TC(s, env, j);
            seg_info info = {0, 1.0, bx, by};
            double ex = ax - tc[0], ey = ay - tc[1];
            if (sqrt(ex * ex + ey * ey) - c->agent_radius <= r2) { /* [CP cpCircleShapePointQuery] */
                info.hit = 1; info.alpha = 0.0;
            } else {
                circle_segment_query(tc[0], tc[1], c->agent_radius, ax, ay, bx, by, r2, &info);
            }
            if (info.hit && info.alpha < out->alpha) { *out = info; best = m->S + j; }
            t_exit = fmin2(t_exit, out->alpha);
        }
    }
    return best;
}

int cato_segment_query(cato_sim *s, int env, int self, double ax, double ay, double bx, double by,
                       double r2, int los, double *alpha, double *point_xy)
{
    seg_info o;
    int sh = segment_query_first(s, env, self, ax, ay, bx, by, r2, los, &o);
    if (alpha) *alpha = o.alpha;
    if (point_xy) { point_xy[0] = o.px; point_xy[1] = o.py; }
    return sh;
}

/* [CP cpSpacePointQueryNearest] reduced to "is anything nearer than maxd" — all the reference
   uses [REF base_env.py:154-157]; cached circle centres, strict '<' */
static int point_query_any(const cato_sim *s, int env, int self, double px, double py, double maxd)
{
    const cato_map *m = &s->maps[s->slot_map[env]];
    for (int j = 0; j < s->A; j++) {
        if (j == self) continue;
        const double *tc = TC(s, env, j);
        double ex = px - tc[0], ey = py - tc[1];
        if (sqrt(ex * ex + ey * ey) - s->cfg.agent_radius < maxd) return 1;
    }
    for (int sh = 0; sh < m->S; sh++)
        if (poly_point_distance(m, sh, s->cfg.wall_radius, px, py) < maxd) return 1;
    return 0;
}

int cato_point_query_any(cato_sim *s, int env, int self, double px, double py, double maxd)
{
    return point_query_any(s, env, self, px, py, maxd);
}

/* ---------------------------------------------------------------- observations ------ */
/* [REF entity.py:159-220 get_observation] + [REF entity.py:222-241 _query_body] */
static void observe_env(const cato_sim *s, int env, uint16_t *dist, uint8_t *type, int32_t *shape)
{
    const int A = s->A, R = s->R;
    const cato_map *m = &s->maps[s->slot_map[env]];
    for (int i = 0; i < A; i++) {
        const double *p = s->pos + 2 * ((size_t)env * A + i);
        double ox = p[0], oy = p[1];
        for (int k = 0; k < R; k++) {
            double bx = ox + s->ray_dx[k], by = oy + s->ray_dy[k]; /* :191-193 */
            seg_info o;
            int sh = segment_query_first(s, env, i, ox, oy, bx, by, s->cfg.ray_radius, 0, &o);
            uint16_t d16 = 0x5E40; /* 400.0 for the default length; recomputed below */
            uint8_t ty = CATO_EMPTY;
            if (sh < 0) {
                d16 = cato_f64_to_f16(s->cfg.ray_length); /* :200 */
            } else {
                d16 = cato_obs_distance_f16(o.px, o.py, ox, oy);
                if (sh < m->S) ty = CATO_WALL;
                else ty = (sh - m->S) >= s->cfg.n_cops ? CATO_THIEF : CATO_COP;
            }
            dist[i * R + k] = d16;
            type[i * R + k] = ty;
            if (shape) shape[i * R + k] = sh;
        }
    }
}

/* [REF observation_spaces.py:67-131 get_shared_observations]: effective rule:
   per team and ray index the first member (roster order) with a non-EMPTY ray supplies
   (type, distance); otherwise (EMPTY, the last member's distance = ray length). */
static void shared_env(const cato_sim *s, const uint16_t *dist, const uint8_t *type,
                       uint16_t *sdist, uint8_t *stype)
{
    const int R = s->R;
    int lo[2] = {0, s->cfg.n_cops}, hi[2] = {s->cfg.n_cops, s->A};
    for (int team = 0; team < 2; team++) {
        for (int k = 0; k < R; k++) {
            uint8_t ty = CATO_EMPTY;
            uint16_t d = 0; /* np.zeros_like :102 — stays 0 only for an empty team */
            for (int i = lo[team]; i < hi[team]; i++) {
                if (ty == CATO_EMPTY) { ty = type[i * R + k]; d = dist[i * R + k]; }
            }
            stype[team * R + k] = ty;
            sdist[team * R + k] = d;
        }
    }
}

/* [REF cop.py:49-75] / [REF thief.py:48-69]; non-terminal values are NumPy float16 scalars
   under NumPy 2 — taken from host-built LUTs indexed by the f16 min distance. */
static void rewards_env(const cato_sim *s, const uint16_t *dist, const uint8_t *type, int captured,
                        int timeout, float *rew)
{
    const int R = s->R;
    for (int i = 0; i < s->A; i++) {
        int is_cop = i < s->cfg.n_cops;
        float r;
        if (captured) r = is_cop ? 1.0f : -1.0f;
        else if (timeout) r = is_cop ? -1.0f : 1.0f;
        else {
            uint8_t want = is_cop ? CATO_THIEF : CATO_COP;
            int found = 0;
            uint16_t dmin = 0xFFFF;
            for (int k = 0; k < R; k++)
                if (type[i * R + k] == want) { /* non-negative f16: bit order == value order */
                    if (!found || dist[i * R + k] < dmin) dmin = dist[i * R + k];
                    found = 1;
                }
            if (found) r = (is_cop ? s->cop_lut : s->thief_lut)[dmin & 0x7FFF];
            else r = is_cop ? (float)(-0.02 - 0.02) : (float)0.15;
        }
        rew[i] = r;
    }
}

/* ---------------------------------------------------------------- termination ------- */
/* [REF base_env.py:521-554] evaluated on fresh body positions, walls-only LOS with radius 0 */
static void termination_env(const cato_sim *s, int env, int *captured, int *timeout)
{
    const int A = s->A, nc = s->cfg.n_cops;
    *captured = 0;
    for (int t = nc; t < A && !*captured; t++) {
        for (int c = 0; c < nc; c++) {
            const double *pt = s->pos + 2 * ((size_t)env * A + t);
            const double *pc = s->pos + 2 * ((size_t)env * A + c);
            seg_info o;
            int hit = segment_query_first(s, env, -1, pt[0], pt[1], pc[0], pc[1], 0.0, 1, &o);
            if (hit < 0) {
                double dx = pt[0] - pc[0], dy = pt[1] - pc[1]; /* Vec2d.get_distance */
                if (sqrt(dx * dx + dy * dy) < s->cfg.termination_radius) { *captured = 1; break; }
            }
        }
    }
    *timeout = (!*captured && s->step_count[env] >= s->cfg.max_step_count);
}

/* ---------------------------------------------------------------- physics ----------- */
typedef struct {
    int a, b;          /* body indices; b = -1 -> static */
    double nx, ny, r1x, r1y, r2x, r2y;
    double nMass, bias, jBias, jnAcc, bounce;
    int first;
    double *cache_jn;  /* where jnAcc is persisted */
} contact_t;

static inline int pair_index(int A, int i, int j) /* i < j, lexicographic */
{
    return i * A - i * (i + 1) / 2 + (j - i - 1);
}

/* closest-feature search + [CP ClosestPointsNew] on the winning hull edge; returns 1 and fills
   n (circle -> wall), contact points when d <= r_c + r_p  [CP CircleToPoly] */
static int circle_poly_contact(const cato_map *m, int sh, double rp, double cx, double cy, double rc,
                               double *nx, double *ny, double *p1x, double *p1y, double *p2x,
                               double *p2y)
{
    int first = m->first[sh], count = m->count[sh];
    int best = -1;
    double bestd = INFINITY, bt = 0, bpx = 0, bpy = 0;
    int inside = 1;
    double maxsep = -INFINITY;
    int sepi = 0;
    for (int i = 0; i < count; i++) {
        const double *pl = m->planes + 8 * (size_t)(first + i);
        const double *pv = m->planes + 8 * (size_t)(first + (i - 1 + count) % count);
        double sep = pl[0] * (cx - pl[2]) + pl[1] * (cy - pl[3]);
        if (sep > 0.0) inside = 0;
        if (sep > maxsep) { maxsep = sep; sepi = i; }
        /* Minkowski points (poly vertex - circle centre); GJK's final ordering for a CCW hull:
           v0 = vert[i], v1 = vert[i-1] */
        double ax_ = pl[2] - cx, ay_ = pl[3] - cy, bx_ = pv[2] - cx, by_ = pv[3] - cy;
        double dx = bx_ - ax_, dy = by_ - ay_;
        /* [CP ClosestT] */
        double t = -fmin2(fmax2((dx * (ax_ + bx_) + dy * (ay_ + by_)) / (dx * dx + dy * dy), -1.0), 1.0);
        double ht = 0.5 * t; /* [CP LerpT] */
        double px = ax_ * (0.5 - ht) + bx_ * (0.5 + ht), py = ay_ * (0.5 - ht) + by_ * (0.5 + ht);
        double dd = px * px + py * py;
        /* GJK only terminates on an edge the origin lies in front of ([CP GJKRecurse] flips / leaves an
           edge with the origin behind it): at a vertex shared with such an edge the tie goes to the
           other one, whose normal gives d > 0 and hence the vertex/vertex branch below. */
        if (sep > 0.0 && dd < bestd) { bestd = dd; best = i; bt = t; bpx = px; bpy = py; }
    }
    if (inside) { /* D4 */
        const double *pl = m->planes + 8 * (size_t)(first + sepi);
        double d = maxsep; /* <= 0 */
        if (!(d <= rc + rp)) return 0;
        *nx = -pl[0]; *ny = -pl[1];
        *p1x = cx + *nx * rc; *p1y = cy + *ny * rc;
        double qx = cx - pl[0] * d, qy = cy - pl[1] * d;
        *p2x = qx + *nx * (-rp); *p2y = qy + *ny * (-rp);
        return 1;
    }
    const double *pl = m->planes + 8 * (size_t)(first + best);
    const double *pv = m->planes + 8 * (size_t)(first + (best - 1 + count) % count);
    double ax_ = pl[2] - cx, ay_ = pl[3] - cy, bx_ = pv[2] - cx, by_ = pv[3] - cy;
    double t = bt, ht = 0.5 * t;
    /* pa = LerpT(tc, tc, t), pb = LerpT(vert[i], vert[i-1], t) */
    double pax = cx * (0.5 - ht) + cx * (0.5 + ht), pay = cy * (0.5 - ht) + cy * (0.5 + ht);
    double pbx = pl[2] * (0.5 - ht) + pv[2] * (0.5 + ht), pby = pl[3] * (0.5 - ht) + pv[3] * (0.5 + ht);
    double dx = bx_ - ax_, dy = by_ - ay_;
    double rx = dy, ry = -dx; /* cpvrperp(delta) */
    double inv = 1.0 / (sqrt(rx * rx + ry * ry) + DBL_MIN);
    double n_x = rx * inv, n_y = ry * inv;
    double d = n_x * bpx + n_y * bpy;
    if (!(d <= 0.0 || (-1.0 < t && t < 1.0))) { /* vertex/vertex */
        double d2 = sqrt(bpx * bpx + bpy * bpy);
        double inv2 = 1.0 / (d2 + DBL_MIN);
        n_x = bpx * inv2; n_y = bpy * inv2;
        d = d2;
    }
    if (!(d <= rc + rp)) return 0;
    *nx = n_x; *ny = n_y;
    *p1x = pax + n_x * rc; *p1y = pay + n_y * rc;
    *p2x = pbx + n_x * (-rp); *p2y = pby + n_y * (-rp);
    return 1;
}

/* [CP cpSpaceStep] for one env */
static void physics_env(cato_sim *s, int env)
{
    const cato_config *c = &s->cfg;
    const int A = s->A;
    const cato_map *m = &s->maps[s->slot_map[env]];
    double *pos = s->pos + 2 * (size_t)env * A, *vel = s->vel + 2 * (size_t)env * A;
    double *vb = s->vbias + 2 * (size_t)env * A, *tc = s->tc + 2 * (size_t)env * A;
    double *leaf = s->leaf_bb + 4 * (size_t)env * A;
    int32_t *wsh = s->wall_shape + (size_t)env * A * K_WALL, *wag = s->wall_age + (size_t)env * A * K_WALL;
    double *wjn = s->wall_jn + (size_t)env * A * K_WALL;
    int32_t *pag = s->pair_age + (size_t)env * s->NP;
    double *pjn = s->pair_jn + (size_t)env * s->NP;
    const double dt = c->dt, rc = c->agent_radius;
    double bb[CATO_MAX_AGENTS][4];

    for (int i = 0; i < A; i++) {
        /* [CP cpBodyUpdatePosition] */
        pos[2 * i] = pos[2 * i] + (vel[2 * i] + vb[2 * i]) * dt;
        pos[2 * i + 1] = pos[2 * i + 1] + (vel[2 * i + 1] + vb[2 * i + 1]) * dt;
        vb[2 * i] = 0.0; vb[2 * i + 1] = 0.0;
        /* [CP cpCircleShapeCacheData] */
        tc[2 * i] = pos[2 * i]; tc[2 * i + 1] = pos[2 * i + 1];
        bb[i][0] = tc[2 * i] - rc; bb[i][1] = tc[2 * i + 1] - rc;
        bb[i][2] = tc[2 * i] + rc; bb[i][3] = tc[2 * i + 1] + rc;
        /* [CP LeafUpdate] + [CP GetBB] with the shape's body velocity */
        double *lf = leaf + 4 * i;
        if (!(lf[0] <= bb[i][0] && lf[2] >= bb[i][2] && lf[1] <= bb[i][1] && lf[3] >= bb[i][3])) {
            double x = (bb[i][2] - bb[i][0]) * 0.1, y = (bb[i][3] - bb[i][1]) * 0.1;
            double vx = vel[2 * i] * 0.1, vy = vel[2 * i + 1] * 0.1;
            lf[0] = bb[i][0] + fmin2(-x, vx); lf[1] = bb[i][1] + fmin2(-y, vy);
            lf[2] = bb[i][2] + fmax2(x, vx); lf[3] = bb[i][3] + fmax2(y, vy);
        }
    }

    contact_t con[MAX_CONTACTS];
    int nc = 0;
    int seen_w[CATO_MAX_AGENTS][K_WALL];
    int seen_p[CATO_MAX_PAIRS];
    memset(seen_w, 0, sizeof seen_w);
    memset(seen_p, 0, sizeof seen_p);

    /* collide: agent-vs-walls in (agent, shape) order, then agent pairs (D2) */
    for (int i = 0; i < A; i++) {
        for (int sh = 0; sh < m->S; sh++) {
            const double *sb = m->bb + 4 * (size_t)sh;
            /* [CP cpBBIntersects] */
            if (!(bb[i][0] <= sb[2] && sb[0] <= bb[i][2] && bb[i][1] <= sb[3] && sb[1] <= bb[i][3])) continue;
            double nx, ny, p1x, p1y, p2x, p2y;
            if (!circle_poly_contact(m, sh, c->wall_radius, tc[2 * i], tc[2 * i + 1], rc, &nx, &ny,
                                     &p1x, &p1y, &p2x, &p2y)) continue;
            /* arbiter cache lookup [CP cpSpaceCollideShapes / cpArbiterUpdate] */
            int slot = -1;
            for (int k = 0; k < K_WALL; k++) if (wsh[i * K_WALL + k] == sh) { slot = k; break; }
            int first = 0;
            if (slot < 0) {
                first = 1;
                for (int k = 0; k < K_WALL; k++) if (wsh[i * K_WALL + k] < 0) { slot = k; break; }
                if (slot < 0) { /* evict the oldest entry not seen this step */
                    int oldest = -1;
                    for (int k = 0; k < K_WALL; k++)
                        if (!seen_w[i][k] && (oldest < 0 || wag[i * K_WALL + k] > wag[i * K_WALL + oldest])) oldest = k;
                    if (oldest < 0) continue; /* table full of live contacts: drop */
                    slot = oldest;
                }
                wsh[i * K_WALL + slot] = sh; wjn[i * K_WALL + slot] = 0.0; wag[i * K_WALL + slot] = 0;
            } else {
                first = wag[i * K_WALL + slot] > 0; /* was CACHED -> FIRST_COLLISION */
            }
            seen_w[i][slot] = 1;
            contact_t *k = &con[nc++];
            k->a = i; k->b = -1; k->nx = nx; k->ny = ny;
            k->r1x = p1x - pos[2 * i]; k->r1y = p1y - pos[2 * i + 1];
            k->r2x = p2x - 0.0; k->r2y = p2y - 0.0; /* static body at the origin */
            k->first = first; k->cache_jn = &wjn[i * K_WALL + slot]; k->jnAcc = *k->cache_jn;
        }
    }
    for (int i = 0; i < A; i++) {
        for (int j = i + 1; j < A; j++) {
            /* [CP CircleToCircle] */
            double mindist = rc + rc;
            double dx = tc[2 * j] - tc[2 * i], dy = tc[2 * j + 1] - tc[2 * i + 1];
            double distsq = dx * dx + dy * dy;
            if (!(distsq < mindist * mindist)) continue;
            double dist = sqrt(distsq);
            double nx = 1.0, ny = 0.0;
            if (dist != 0.0) { double inv = 1.0 / dist; nx = dx * inv; ny = dy * inv; }
            int pi = pair_index(A, i, j);
            int first;
            if (pag[pi] < 0) { first = 1; pjn[pi] = 0.0; }
            else first = pag[pi] > 0;
            pag[pi] = 0; seen_p[pi] = 1;
            contact_t *k = &con[nc++];
            k->a = i; k->b = j; k->nx = nx; k->ny = ny;
            double p1x = tc[2 * i] + nx * rc, p1y = tc[2 * i + 1] + ny * rc;
            double p2x = tc[2 * j] + nx * (-rc), p2y = tc[2 * j + 1] + ny * (-rc);
            k->r1x = p1x - pos[2 * i]; k->r1y = p1y - pos[2 * i + 1];
            k->r2x = p2x - pos[2 * j]; k->r2y = p2y - pos[2 * j + 1];
            k->first = first; k->cache_jn = &pjn[pi]; k->jnAcc = *k->cache_jn;
        }
    }
    /* age / expire cached arbiters [CP cpSpaceArbiterSetFilter] */
    for (int i = 0; i < A; i++)
        for (int k = 0; k < K_WALL; k++) {
            int q = i * K_WALL + k;
            if (wsh[q] < 0) continue;
            if (seen_w[i][k]) wag[q] = 0;
            else if (++wag[q] >= c->persistence) { wsh[q] = -1; wag[q] = 0; wjn[q] = 0.0; }
        }
    for (int q = 0; q < s->NP; q++) {
        if (pag[q] < 0 || seen_p[q]) continue;
        if (++pag[q] >= c->persistence) { pag[q] = -1; pjn[q] = 0.0; }
    }

    /* [CP cpArbiterPreStep] */
    const double m_inv = 1.0 / c->agent_mass;
    for (int q = 0; q < nc; q++) {
        contact_t *k = &con[q];
        double mia = m_inv, mib = (k->b < 0) ? 0.0 : m_inv;
        k->nMass = 1.0 / (mia + mib); /* D1: i_inv*cross(r,n)^2 ~ 1e-27, dropped */
        double bpx = (k->b < 0) ? 0.0 : pos[2 * k->b], bpy = (k->b < 0) ? 0.0 : pos[2 * k->b + 1];
        double bdx = bpx - pos[2 * k->a], bdy = bpy - pos[2 * k->a + 1];
        double dist = ((k->r2x - k->r1x) + bdx) * k->nx + ((k->r2y - k->r1y) + bdy) * k->ny;
        k->bias = -c->bias_coef * fmin2(0.0, dist + c->slop) / dt;
        k->jBias = 0.0;
        double vbx = (k->b < 0) ? 0.0 : vel[2 * k->b], vby = (k->b < 0) ? 0.0 : vel[2 * k->b + 1];
        k->bounce = ((vbx - vel[2 * k->a]) * k->nx + (vby - vel[2 * k->a + 1]) * k->ny) * 0.0; /* e = 0 */
    }
    /* [CP cpBodyUpdateVelocity]: gravity 0, damping 1, no forces -> identity */
    /* [CP cpArbiterApplyCachedImpulse], dt_coef = dt/prev_dt = 1 (first-ever step has only
       first-collision arbiters, which skip this) */
    for (int q = 0; q < nc; q++) {
        contact_t *k = &con[q];
        if (k->first) continue;
        double jx = (k->nx * k->jnAcc - k->ny * 0.0) * 1.0, jy = (k->nx * 0.0 + k->ny * k->jnAcc) * 1.0;
        vel[2 * k->a] = vel[2 * k->a] + (-jx) * m_inv; vel[2 * k->a + 1] = vel[2 * k->a + 1] + (-jy) * m_inv;
        if (k->b >= 0) { vel[2 * k->b] = vel[2 * k->b] + jx * m_inv; vel[2 * k->b + 1] = vel[2 * k->b + 1] + jy * m_inv; }
    }
    /* [CP cpArbiterApplyImpulse] x iterations, arbiters in list order */
    for (int it = 0; it < c->iterations; it++) {
        for (int q = 0; q < nc; q++) {
            contact_t *k = &con[q];
            int a = k->a, b = k->b;
            double vbbx = (b < 0) ? 0.0 : vb[2 * b], vbby = (b < 0) ? 0.0 : vb[2 * b + 1];
            double vvbx = (b < 0) ? 0.0 : vel[2 * b], vvby = (b < 0) ? 0.0 : vel[2 * b + 1];
            double vbn = (vbbx - vb[2 * a]) * k->nx + (vbby - vb[2 * a + 1]) * k->ny;
            double vrn = (vvbx - vel[2 * a]) * k->nx + (vvby - vel[2 * a + 1]) * k->ny;
            double jbn = (k->bias - vbn) * k->nMass;
            double jbnOld = k->jBias;
            k->jBias = fmax2(jbnOld + jbn, 0.0);
            double jn = -(k->bounce + vrn) * k->nMass;
            double jnOld = k->jnAcc;
            k->jnAcc = fmax2(jnOld + jn, 0.0);
            double jbx = k->nx * (k->jBias - jbnOld), jby = k->ny * (k->jBias - jbnOld);
            double dj = k->jnAcc - jnOld;
            double jx = k->nx * dj - k->ny * 0.0, jy = k->nx * 0.0 + k->ny * dj; /* cpvrotate, jt = 0 */
            vb[2 * a] = vb[2 * a] + (-jbx) * m_inv; vb[2 * a + 1] = vb[2 * a + 1] + (-jby) * m_inv;
            vel[2 * a] = vel[2 * a] + (-jx) * m_inv; vel[2 * a + 1] = vel[2 * a + 1] + (-jy) * m_inv;
            if (b >= 0) {
                vb[2 * b] = vb[2 * b] + jbx * m_inv; vb[2 * b + 1] = vb[2 * b + 1] + jby * m_inv;
                vel[2 * b] = vel[2 * b] + jx * m_inv; vel[2 * b + 1] = vel[2 * b + 1] + jy * m_inv;
            }
        }
    }
    for (int q = 0; q < nc; q++) *con[q].cache_jn = con[q].jnAcc;
}

/* ---------------------------------------------------------------- step / reset ------ */
static void write_obs(const cato_sim *s, int env, const cato_outputs *out, const uint16_t *dist,
                      const uint8_t *type, const int32_t *shape)
{
    const int A = s->A, R = s->R;
    size_t o = (size_t)env * A * R;
    if (out->obs_distance) memcpy(out->obs_distance + o, dist, sizeof(uint16_t) * A * R);
    if (out->obs_type) memcpy(out->obs_type + o, type, (size_t)A * R);
    if (out->hit_shape) memcpy(out->hit_shape + o, shape, sizeof(int32_t) * A * R);
    uint16_t sd[2 * 512]; uint8_t st[2 * 512];
    shared_env(s, dist, type, sd, st);
    if (out->shared_distance) memcpy(out->shared_distance + (size_t)env * 2 * R, sd, sizeof(uint16_t) * 2 * R);
    if (out->shared_type) memcpy(out->shared_type + (size_t)env * 2 * R, st, 2 * (size_t)R);
    if (out->team_positions) /* [REF observation_spaces.py:92-95] fresh body.position -> f16 */
        for (int i = 0; i < 2 * A; i++)
            out->team_positions[(size_t)env * A * 2 + i] = cato_f64_to_f16(s->pos[(size_t)env * A * 2 + i]);
}

/* [REF base_env.py:354-413 BaseEnv.step] */
static void step_env(cato_sim *s, int env, const int32_t *actions, const cato_outputs *out)
{
    const int A = s->A, R = s->R;
    const cato_config *c = &s->cfg;
    double *vel = s->vel + 2 * (size_t)env * A;
    s->step_count[env] += 1;                                          /* :372 */
    int captured, timeout;
    termination_env(s, env, &captured, &timeout);                     /* :378 */
    for (int i = 0; i < A; i++) {                                     /* [REF entity.py:126-134] */
        int act = actions[(size_t)env * A + i];
        double jx = 0.0, jy = 0.0;                                    /* :77-82 force mappings */
        if (act == 0) jx = -c->impulse; else if (act == 1) jy = c->impulse;
        else if (act == 2) jx = c->impulse; else if (act == 3) jy = -c->impulse;
        double m_inv = 1.0 / c->agent_mass;
        double vx = vel[2 * i] + jx * m_inv, vy = vel[2 * i + 1] + jy * m_inv; /* [CP cpBodyApplyImpulseAtWorldPoint] */
        double len = sqrt(vx * vx + vy * vy);                         /* Vec2d.__abs__ */
        if (len > c->max_speed) { vx = vx / len * c->max_speed; vy = vy / len * c->max_speed; } /* normalized()*max */
        vel[2 * i] = vx; vel[2 * i + 1] = vy;
    }
    uint16_t dist[CATO_MAX_AGENTS * 512]; uint8_t type[CATO_MAX_AGENTS * 512];
    int32_t shape[CATO_MAX_AGENTS * 512];
    observe_env(s, env, dist, type, shape);                           /* entity.py:143 */
    float rew[CATO_MAX_AGENTS];
    rewards_env(s, dist, type, captured, timeout, rew);               /* entity.py:144 */
    write_obs(s, env, out, dist, type, shape);                        /* :388-390 */
    if (out->reward) memcpy(out->reward + (size_t)env * A, rew, sizeof(float) * A);
    physics_env(s, env);                                              /* :392 */
    if (out->terminated) out->terminated[env] = (uint8_t)(captured || timeout); /* entity.py:146 */
    if (out->truncated) out->truncated[env] = (uint8_t)timeout;       /* :397 */
    if (out->winner) out->winner[env] = (int8_t)(captured ? 0 : (timeout ? 1 : -1)); /* :399-406 */
    (void)R;
}

/* [REF base_env.py:286-352 reset] + [REF :123-166 _get_non_colliding_position] +
   [REF entity.py:148-157 Entity.reset]; spawn RNG is Philox here */
static void reset_env(cato_sim *s, int env, const double *positions, const cato_outputs *out)
{
    const int A = s->A;
    const cato_map *m = &s->maps[s->slot_map[env]];
    const cato_config *c = &s->cfg;
    double *pos = s->pos + 2 * (size_t)env * A, *vel = s->vel + 2 * (size_t)env * A;
    s->reset_count[env] += 1;
    uint32_t rc = (uint32_t)s->reset_count[env];
    double np[CATO_MAX_AGENTS][2];
    for (int i = 0; i < A; i++) {
        if (positions) {
            np[i][0] = positions[((size_t)env * A + i) * 2]; np[i][1] = positions[((size_t)env * A + i) * 2 + 1];
            continue;
        }
        int r0 = m->region_off[i], nr = m->region_off[i + 1] - r0;
        if (nr <= 0) { /* no spawn regions: Entity.reset() -> remembered initial position :323-332 */
            np[i][0] = m->start[2 * i]; np[i][1] = m->start[2 * i + 1];
            continue;
        }
        uint32_t rnd[4];
        philox_env(s, env, rc, (uint32_t)i, 0x100u, rnd);
        const double *rg = m->regions + 4 * (size_t)(r0 + (int)(rnd[0] % (uint32_t)nr)); /* :144-145 */
        int ok = 0;
        for (int att = 0; att < 20 && !ok; att++) {                   /* :151 */
            philox_env(s, env, rc, (uint32_t)i, 0x200u + (uint32_t)att, rnd);
            /* random.uniform(a, b) = a + (b - a) * random()  [REF map_utils.py:9-10] */
            double x = rg[0] + ((rg[0] + rg[2]) - rg[0]) * u53(rnd[0], rnd[1]);
            double y = rg[1] + ((rg[1] + rg[3]) - rg[1]) * u53(rnd[2], rnd[3]);
            if (!point_query_any(s, env, i, x, y, c->agent_radius)) { np[i][0] = x; np[i][1] = y; ok = 1; } /* :154-158 */
        }
        if (!ok) { np[i][0] = rg[0] + rg[2] / 2; np[i][1] = rg[1] + rg[3] / 2; } /* :163-166 */
    }
    for (int i = 0; i < A; i++) { /* body.position = pos; body.velocity = 0; caches stay stale (Q1) */
        pos[2 * i] = np[i][0]; pos[2 * i + 1] = np[i][1];
        vel[2 * i] = 0.0; vel[2 * i + 1] = 0.0;
    }
    uint16_t dist[CATO_MAX_AGENTS * 512]; uint8_t type[CATO_MAX_AGENTS * 512];
    int32_t shape[CATO_MAX_AGENTS * 512];
    observe_env(s, env, dist, type, shape);                           /* :334-339 */
    write_obs(s, env, out, dist, type, shape);                        /* :342-344 */
    s->step_count[env] = 0;                                           /* :350 */
}

int cato_step(cato_sim *s, const int32_t *actions, const cato_outputs *out)
{
    cato_outputs none;
    memset(&none, 0, sizeof none);
    if (!out) out = &none;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
#endif
    for (int e = 0; e < s->N; e++) step_env(s, e, actions, out);
    return 0;
}

int cato_reset(cato_sim *s, const uint8_t *mask, const double *positions, const cato_outputs *out)
{
    cato_outputs none;
    memset(&none, 0, sizeof none);
    if (!out) out = &none;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
#endif
    for (int e = 0; e < s->N; e++)
        if (!mask || mask[e]) reset_env(s, e, positions, out);
    return 0;
}

int cato_random_actions(cato_sim *s, uint64_t tick, int32_t *actions)
{
    for (int e = 0; e < s->N; e++)
        for (int i = 0; i < s->A; i++) {
            uint32_t rnd[4];
            philox_env(s, e, (uint32_t)tick, (uint32_t)i, 0xAC710u, rnd);
            actions[(size_t)e * s->A + i] = (int32_t)(rnd[0] & 3u);
        }
    return 0;
}

/* ---------------------------------------------------------------- create / state ---- */
static int parse_blob(const void *blob, size_t size, cato_map *m)
{
    if (size < 64) return -1;
    const int32_t *h = (const int32_t *)blob;
    if (h[0] != BLOB_MAGIC || h[1] != 1) return -1;
    m->S = h[2]; m->P = h[3]; m->A = h[4]; m->n_cops = h[5]; m->n_thieves = h[6]; m->n_regions = h[7];
    size_t nf = 2 + 4 * (size_t)m->S + 8 * (size_t)m->P + 2 * (size_t)m->A + 4 * (size_t)m->n_regions;
    size_t ni = 2 * (size_t)m->S + (size_t)m->A + 1;
    if (size != 64 + nf * 8 + ni * 4) return -1;
    double *f = (double *)malloc(nf * 8);
    int32_t *iv = (int32_t *)malloc(ni * 4);
    memcpy(f, (const char *)blob + 64, nf * 8);
    memcpy(iv, (const char *)blob + 64 + nf * 8, ni * 4);
    m->win_w = f[0]; m->win_h = f[1];
    m->bb = f + 2; m->planes = m->bb + 4 * (size_t)m->S; m->start = m->planes + 8 * (size_t)m->P;
    m->regions = m->start + 2 * (size_t)m->A;
    m->first = iv; m->count = iv + m->S; m->region_off = iv + 2 * (size_t)m->S;
    bbt_build(m);
    return 0;
}

int cato_create(const cato_config *cfg, const cato_tables *tab, const void *const *map_blobs,
                const size_t *blob_sizes, int n_maps, const int32_t *slot_map_ids, cato_sim **out)
{
    int A = cfg->n_cops + cfg->n_thieves;
    if (A < 1 || A > CATO_MAX_AGENTS || cfg->n_rays < 1 || cfg->n_rays > 512 || cfg->n_envs < 1) {
        snprintf(g_err, sizeof g_err, "bad config: agents=%d rays=%d envs=%d", A, cfg->n_rays, cfg->n_envs);
        return -1;
    }
    cato_sim *s = (cato_sim *)calloc(1, sizeof *s);
    s->cfg = *cfg; s->N = cfg->n_envs; s->A = A; s->R = cfg->n_rays; s->NP = A * (A - 1) / 2;
    s->n_maps = n_maps;
    s->maps = (cato_map *)calloc((size_t)n_maps, sizeof(cato_map));
    for (int i = 0; i < n_maps; i++) {
        if (parse_blob(map_blobs[i], blob_sizes[i], &s->maps[i]) != 0 || s->maps[i].A != A ||
            s->maps[i].n_cops != cfg->n_cops) {
            snprintf(g_err, sizeof g_err, "map blob %d invalid or roster mismatch", i);
            return -2;
        }
        /* the arbiter cache holds K_WALL wall contacts per agent: a map on which the bb of one agent circle can overlap more wall bbs
           than that at once is refused here, as the HIP library refuses it (deepest point of the grown bbs: a left and a bottom edge) */
        const cato_map *mp = &s->maps[i];
        const double rc = cfg->agent_radius;
        int depth = 0;
        for (int a = 0; a < mp->S; a++)
            for (int b = 0; b < mp->S; b++) {
                const double x = mp->bb[4 * a] - rc, y = mp->bb[4 * b + 1] - rc;
                int n = 0;
                for (int q = 0; q < mp->S; q++)
                    n += (mp->bb[4 * q] - rc <= x && x <= mp->bb[4 * q + 2] + rc && mp->bb[4 * q + 1] - rc <= y && y <= mp->bb[4 * q + 3] + rc);
                if (n > depth) depth = n;
            }
        if (depth > K_WALL) {
            snprintf(g_err, sizeof g_err, "map blob %d: an agent can touch the bounding boxes of %d walls at once; %d wall contacts are cached per agent", i, depth, K_WALL);
            return -2;
        }
    }
    size_t N = (size_t)s->N;
    s->slot_map = (int32_t *)calloc(N, 4);
    for (size_t e = 0; e < N; e++) {
        s->slot_map[e] = slot_map_ids ? slot_map_ids[e] : 0;
        if (s->slot_map[e] < 0 || s->slot_map[e] >= n_maps) { snprintf(g_err, sizeof g_err, "slot_map_ids[%zu] out of range", e); return -3; }
    }
    s->ray_dx = (double *)malloc(8 * (size_t)s->R); s->ray_dy = (double *)malloc(8 * (size_t)s->R);
    memcpy(s->ray_dx, tab->ray_dx, 8 * (size_t)s->R); memcpy(s->ray_dy, tab->ray_dy, 8 * (size_t)s->R);
    s->cop_lut = (float *)malloc(4 * 32768); s->thief_lut = (float *)malloc(4 * 32768);
    memcpy(s->cop_lut, tab->cop_reward_lut, 4 * 32768); memcpy(s->thief_lut, tab->thief_reward_lut, 4 * 32768);
    s->pos = (double *)calloc(N * A * 2, 8); s->vel = (double *)calloc(N * A * 2, 8);
    s->vbias = (double *)calloc(N * A * 2, 8); s->tc = (double *)calloc(N * A * 2, 8);
    s->leaf_bb = (double *)calloc(N * A * 4, 8);
    s->wall_shape = (int32_t *)calloc(N * A * K_WALL, 4); s->wall_age = (int32_t *)calloc(N * A * K_WALL, 4);
    s->wall_jn = (double *)calloc(N * A * K_WALL, 8);
    s->pair_age = (int32_t *)calloc(N * (s->NP ? s->NP : 1), 4); s->pair_jn = (double *)calloc(N * (s->NP ? s->NP : 1), 8);
    s->step_count = (int32_t *)calloc(N, 4); s->reset_count = (int32_t *)calloc(N, 4);
    double rc = cfg->agent_radius;
    for (size_t e = 0; e < N; e++) {
        const cato_map *m = &s->maps[s->slot_map[e]];
        for (int i = 0; i < A; i++) {
            /* Entity.__init__: body.position = start; space.add -> caches + BBTree leaf at the
               start position with zero velocity [REF entity.py:109-124] [CP cpSpaceAddShape] */
            size_t q = e * A + i;
            double x = m->start[2 * i], y = m->start[2 * i + 1];
            s->pos[2 * q] = x; s->pos[2 * q + 1] = y; s->tc[2 * q] = x; s->tc[2 * q + 1] = y;
            double l = x - rc, b = y - rc, r = x + rc, t = y + rc;
            double mx = (r - l) * 0.1, my = (t - b) * 0.1;
            s->leaf_bb[4 * q] = l + fmin2(-mx, 0.0 * 0.1); s->leaf_bb[4 * q + 1] = b + fmin2(-my, 0.0 * 0.1);
            s->leaf_bb[4 * q + 2] = r + fmax2(mx, 0.0 * 0.1); s->leaf_bb[4 * q + 3] = t + fmax2(my, 0.0 * 0.1);
            for (int k = 0; k < K_WALL; k++) s->wall_shape[q * K_WALL + k] = -1;
        }
        for (int p = 0; p < s->NP; p++) s->pair_age[e * s->NP + p] = -1;
    }
    *out = s;
    return 0;
}

void cato_destroy(cato_sim *s)
{
    if (!s) return;
    for (int i = 0; i < s->n_maps; i++) { free(s->maps[i].bb - 2); free(s->maps[i].first); free(s->maps[i].tree); }
    free(s->maps); free(s->slot_map); free(s->ray_dx); free(s->ray_dy); free(s->cop_lut); free(s->thief_lut);
    free(s->pos); free(s->vel); free(s->vbias); free(s->tc); free(s->leaf_bb); free(s->wall_shape);
    free(s->wall_age); free(s->wall_jn); free(s->pair_age); free(s->pair_jn); free(s->step_count);
    free(s->reset_count); free(s);
}

#define COPY(dst, src, n) do { if (dst) memcpy((dst), (src), (n)); } while (0)
int cato_get_state(cato_sim *s, const cato_state *d)
{
    size_t N = (size_t)s->N, A = (size_t)s->A;
    COPY(d->pos, s->pos, N * A * 16); COPY(d->vel, s->vel, N * A * 16); COPY(d->vbias, s->vbias, N * A * 16);
    COPY(d->tc, s->tc, N * A * 16); COPY(d->leaf_bb, s->leaf_bb, N * A * 32);
    COPY(d->wall_shape, s->wall_shape, N * A * K_WALL * 4); COPY(d->wall_age, s->wall_age, N * A * K_WALL * 4);
    COPY(d->wall_jn, s->wall_jn, N * A * K_WALL * 8);
    COPY(d->pair_age, s->pair_age, N * s->NP * 4); COPY(d->pair_jn, s->pair_jn, N * s->NP * 8);
    COPY(d->step_count, s->step_count, N * 4); COPY(d->reset_count, s->reset_count, N * 4);
    return 0;
}
#define COPYIN(dst, src, n) do { if (src) memcpy((dst), (src), (n)); } while (0)
int cato_set_state(cato_sim *s, const cato_state *d)
{
    size_t N = (size_t)s->N, A = (size_t)s->A;
    COPYIN(s->pos, d->pos, N * A * 16); COPYIN(s->vel, d->vel, N * A * 16); COPYIN(s->vbias, d->vbias, N * A * 16);
    COPYIN(s->tc, d->tc, N * A * 16); COPYIN(s->leaf_bb, d->leaf_bb, N * A * 32);
    COPYIN(s->wall_shape, d->wall_shape, N * A * K_WALL * 4); COPYIN(s->wall_age, d->wall_age, N * A * K_WALL * 4);
    COPYIN(s->wall_jn, d->wall_jn, N * A * K_WALL * 8);
    COPYIN(s->pair_age, d->pair_age, N * s->NP * 4); COPYIN(s->pair_jn, d->pair_jn, N * s->NP * 8);
    COPYIN(s->step_count, d->step_count, N * 4); COPYIN(s->reset_count, d->reset_count, N * 4);
    return 0;
}
