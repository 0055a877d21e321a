// cat_sim_fan.h -- part of the env core's single translation unit (included by cat_sim.hip, in this order; not a stand-alone header):
// Entity.get_observation: per-agent setup, the ray fan in its chunk / slot / group forms, rewards, shared observations, the termination criterion.
// ------------------------------------------------------------------ ray fan -------------------
// Broadphase = spatial hash (GridDesc): (cell of the agent, ray index) -> ascending candidate wall ids,
// looked up in a table built once per map; the other agents' circles are added per ray from the cone
// their (leaf) bb subtends.  Per pass over <= kPassJ candidate positions: the ray's own lane computes the
// BBTree gate value t_bb of its candidate and drops it when t_bb >= the ray's best alpha so far (best only
// decreases, so that candidate could never be visited); the surviving (ray, candidate) pairs are packed
// j-major into a dense item list (ballot + mbcnt, no scan), every lane evaluates one item -- the shape's
// own segment query (alpha + which face/vertex was hit) -- and each ray then walks ITS items in index
// order with [CP cpSpaceSegmentQueryFirst]'s sequential rule "visit iff t_bb < best alpha so far, accept
// iff alpha < best": identical to visiting every shape one after the other.
constexpr int kFeatNear = 63;       // alpha = 0 hit ([CP cpShapeSegmentQuery] start-inside rule)
constexpr int kItemCap = 128;       // live items per pass: two FULL 64-lane rounds of shape queries (160: a third round of 32; agh-map 110.0 -> 107.1 us;
                                    // 96 and 64 are slower again: 112)
constexpr int kPassJ = 8;           // candidate positions per ray per pass
constexpr int kFanBytes = 2 * kItemCap * 8 + kItemCap * 2 + kPassJ * kLanes * 2;   // itbb, ialpha, itm, itemidx
constexpr int kGroupRays = 256;     // fan_group: most rays of one agent (R) it is built for; a group holds <= 4 chunks

// [CP cpPolyShapeSegmentQuery] returning (alpha, feature): plane i -> i, bevel of vertex i -> count + i.
// Planes overwrite unconditionally, bevels replace on strictly smaller alpha; tracking both separately
// and merging afterwards is the same "min, earlier wins ties".
// Shaped for a wave whose lanes hold unrelated (ray, wall) pairs: a branch-free sweep over the hull
// classifies every edge (can the segment cross its face line / can it touch its corner circle), then
// the exact face test and the exact corner test each run once per surviving candidate (ascending edge
// order, so "a later face overwrites" is kept) instead of being entered from inside every edge iteration.
// An agent's circle goes through the same code as a hull with no edges and one "corner" (its cached centre, radius
// r = the agent radius): the lanes of a round hold walls and agents side by side, and a separate circle path would be
// executed for the whole wave whenever one lane needs it.  cx, cy: that centre (ignored for walls).
__device__ __forceinline__ void poly_query_feat(const Lds &L, float cmax, bool wall, int sh, double r, double cx, double cy, double ax, double ay,
                                                double bx, double by, double r2, double &alpha, int &feat)
{
    const int fc = wall ? L.fc[sh] : 0, first = fc & 0xFFFF, count = fc >> 16;
    const double rsum = r + r2, rr = rsum * rsum;
    // Conservative f32 pre-classification of every hull edge (from the f32 copy of the plane records, which holds
    // c = dot(v0, n) + rsum for THIS rsum = wall radius + ray radius): a face stays a candidate unless the f32 evaluation,
    // widened by a bound on its error, excludes one of the exact conditions 0 <= d <= den and dtMin <= dt <= dtMax; a
    // corner stays a candidate unless its centre is farther than rsum from the ray's LINE or projects outside the
    // segment by more than rsum.  Everything the exact tests below would accept is kept, so the results are those of
    // evaluating every edge exactly; what changes is that the exact f64 tests (a divide / a square root and a divide)
    // mostly run for the one face or corner that is really hit.
    const float axf = (float)ax, ayf = (float)ay;
    const float dxf = (float)(bx - ax), dyf = (float)(by - ay);
    const float len2 = dxf * dxf + dyf * dyf, len = sqrtf(len2);
    const float e1 = 1e-6f * (fabsf(axf) + fabsf(ayf) + cmax + 512.0f);   // >= 3x the error of d and den evaluated in f32
    const float thr = ((float)rsum + 0.01f) * len * 1.00001f + 0.25f + 64.0f * e1;
    const float s_lo = -((float)rsum + 1.0f) * len, s_hi = len2 + ((float)rsum + 1.0f) * len;
    const bool bevels = rsum > 0.0;
    unsigned pm = 0u, vm = wall ? 0u : 1u;
    const double *pl0 = L.planes + 8 * first;
    {   // two edges per iteration on packed f32 arithmetic (the pair records interleave the two edges' components)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const float *q = L.p32 + kPairF * (wall ? L.fp[sh] : 0);
        const f32x2 ax2 = {axf, axf}, ay2 = {ayf, ayf}, dx2 = {dxf, dxf}, dy2 = {dyf, dyf};
        const f32x2 len3 = {3.0f * len, 3.0f * len}, four = {4.0f, 4.0f};
        QCOUNT(24);
        for (int i = 0; i < count; i += 2, q += kPairF) {
            QCOUNT(26);
            const float4 r0 = *reinterpret_cast<const float4 *>(q);        // n.x n.x' n.y n.y'
            const float4 r1 = *reinterpret_cast<const float4 *>(q + 4);    // c c' dtMin dtMin'
            const float4 r2 = *reinterpret_cast<const float4 *>(q + 8);    // dtMax dtMax' v0.x v0.x'
            const float2 r3 = *reinterpret_cast<const float2 *>(q + 12);   // v0.y v0.y'
            const f32x2 nx = {r0.x, r0.y}, ny = {r0.z, r0.w}, cc = {r1.x, r1.y}, vx = {r2.z, r2.w}, vy = {r3.x, r3.y};
            const f32x2 d = __builtin_elementwise_fma(ay2, ny, ax2 * nx) - cc;
            const f32x2 den = -__builtin_elementwise_fma(dy2, ny, dx2 * nx);
            // where the crossing point falls along the face (skipped for a ray almost parallel to it: ill-conditioned)
            const f32x2 ri = {__builtin_amdgcn_rcpf(fmaxf(den.x, 0.25f)), __builtin_amdgcn_rcpf(fmaxf(den.y, 0.25f))};
            const f32x2 t = d * ri;
            const f32x2 ptx = __builtin_elementwise_fma(t, dx2, ax2), pty = __builtin_elementwise_fma(t, dy2, ay2);
            const f32x2 dt = __builtin_elementwise_fma(nx, pty, -(ny * ptx));
            const f32x2 e2 = e1 * __builtin_elementwise_fma(len3, ri, four);
            const bool f0 = (d.x >= -e1) && (d.x <= den.x + e1) && ((den.x < 0.25f) || ((dt.x >= r1.z - e2.x) && (dt.x <= r2.x + e2.x)));
            const bool f1 = (d.y >= -e1) && (d.y <= den.y + e1) && ((den.y < 0.25f) || ((dt.y >= r1.w - e2.y) && (dt.y <= r2.y + e2.y)));
            pm |= ((unsigned)f0 | ((unsigned)f1 << 1)) << i;
            const f32x2 ex = vx - ax2, ey = vy - ay2;
            const f32x2 cr = __builtin_elementwise_fma(dx2, ey, -(dy2 * ex)), sp = __builtin_elementwise_fma(dx2, ex, dy2 * ey);
            const bool v0 = bevels && !(fabsf(cr.x) > thr) && (sp.x >= s_lo) && (sp.x <= s_hi);
            const bool v1 = bevels && !(fabsf(cr.y) > thr) && (sp.y >= s_lo) && (sp.y <= s_hi);
            vm |= ((unsigned)v0 | ((unsigned)v1 << 1)) << i;
        }
    }
    double pa = 1.0, va = 1.0;
    int pf = -1, vf = -1;
    while (pm) {   // exact face test
        QCOUNT(28);
        const int i = __builtin_ctz(pm);
        pm &= pm - 1;
        const double *pl = pl0 + 8 * i;
        const double2 n = *reinterpret_cast<const double2 *>(pl);
        const double2 e0 = *reinterpret_cast<const double2 *>(pl + 4);  // vn, dtMin
        const double an = ax * n.x + ay * n.y;
        const double d = an - e0.x - rsum;
        const double bn = bx * n.x + by * n.y;
        const double den = fmax2(an - bn, DBL_MIN);
        const double t = d / den;
        if (!(t < 0.0 || 1.0 < t)) {
            const double ptx = ax * (1.0 - t) + bx * t, pty = ay * (1.0 - t) + by * t;
            const double dtv = n.x * pty - n.y * ptx;
            if (e0.y <= dtv && dtv <= pl[6]) { pa = t; pf = i; }
        }
    }
    while (vm) {   // [CP CircleSegmentQuery] on the corner circle
        QCOUNT(30);
        const int i = __builtin_ctz(vm);
        vm &= vm - 1;
        double2 v = *reinterpret_cast<const double2 *>(pl0 + 8 * i + 2);
        if (!wall) { v.x = cx; v.y = cy; }
        const double dax = ax - v.x, day = ay - v.y, dbx = bx - v.x, dby = by - v.y;
        const double dada = dax * dax + day * day, dadb = dax * dbx + day * dby, dbdb = dbx * dbx + dby * dby;
        const double qa = dada - 2.0 * dadb + dbdb;
        const double qb = dadb - dada;
        const double det = qb * qb - qa * (dada - rr);
        if (det >= 0.0) {
            const double t = (-qb - sqrt(det)) / qa;
            if (0.0 <= t && t <= 1.0 && t < va) { va = t; vf = count + i; }
        }
    }
    alpha = 2.0; feat = -1;
    if (pf >= 0) { alpha = pa; feat = pf; }
    if (vf >= 0 && va < (pf >= 0 ? pa : 1.0)) { alpha = va; feat = vf; }
}

// hit point of [CP CircleSegmentQuery]: lerp(a,b,t) - normalize(lerp(da,db,t)) * r2
__device__ __forceinline__ void circle_hit_point(double cx, double cy, double ax, double ay, double bx, double by,
                                                 double t, double r2, double &px, double &py)
{
    double dax = ax - cx, day = ay - cy, dbx = bx - cx, dby = by - cy;
    double lx = dax * (1.0 - t) + dbx * t, ly = day * (1.0 - t) + dby * t;
    double inv = 1.0 / (sqrt(lx * lx + ly * ly) + DBL_MIN);
    double nx = lx * inv, ny = ly * inv;
    px = (ax * (1.0 - t) + bx * t) - nx * r2;
    py = (ay * (1.0 - t) + by * t) - ny * r2;
}

// atan2 good to ~2e-4 rad (only used for a conservative cone, never for results)
__device__ __forceinline__ float fast_atan2(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mn * __builtin_amdgcn_rcpf(fmaxf(mx, 1e-30f));
    const float s = a * a;
    float r = ((-0.0464964749f * s + 0.15931422f) * s - 0.327622764f) * s * a + a;
    if (ay > ax) r = 1.57079637f - r;
    if (x < 0.0f) r = 3.14159274f - r;
    return y < 0.0f ? -r : r;
}

// contiguous ray-index range [k0, k0+cnt) (mod R) whose directions can enter the box; any
// superset is correct, the exact decision is the slab test of the visit.  Seen from a point
// outside an axis-aligned box the cone is bounded by two silhouette corners that depend only on
// which side(s) of the box the point lies: start corner (counter-clockwise first) and end corner.
__device__ __forceinline__ void ray_cone(const Params &p, double ax, double ay, double l, double b, double r,
                                         double t, int R, int &k0, int &cnt)
{
    const int sx = ax < l ? 0 : (ax > r ? 2 : 1), sy = ay < b ? 0 : (ay > t ? 2 : 1);
    if (!p.ang_ok || (sx == 1 && sy == 1)) { k0 = 0; cnt = R; return; }
    const float x0 = (float)(l - ax), x1 = (float)(r - ax), y0 = (float)(b - ay), y1 = (float)(t - ay);
    // angles grow from +x toward +y.  start = silhouette corner with the smallest angle, end = the
    // one with the largest (x0 < x1, y0 < y1 are the box sides relative to the point):
    //   box above (sy 0):  left-of-box (sx 0): (x1,y0)->(x0,y1)   inside: (x1,y0)->(x0,y0)   right: (x1,y1)->(x0,y0)
    //   box level (sy 1):  sx 0: (x0,y0)->(x0,y1)                                              sx 2: (x1,y1)->(x1,y0) (wraps)
    //   box below (sy 2):  sx 0: (x0,y0)->(x1,y1)               inside: (x0,y1)->(x1,y1)     sx 2: (x0,y1)->(x1,y0)
    float sxx, syy, exx, eyy;
    if (sy == 0) {
        sxx = x1; syy = (sx == 2) ? y1 : y0;
        exx = x0; eyy = (sx == 0) ? y1 : y0;
    } else if (sy == 2) {
        sxx = x0; syy = (sx == 0) ? y0 : y1;
        exx = x1; eyy = (sx == 2) ? y0 : y1;
    } else if (sx == 0) {
        sxx = x0; syy = y0; exx = x0; eyy = y1;
    } else {
        sxx = x1; syy = y1; exx = x1; eyy = y0;
    }
    float th0 = fast_atan2(syy, sxx), th1 = fast_atan2(eyy, exx);
    if (th1 < th0) th1 += 6.28318548f;
    const float eps = 1.5e-3f;
    const float a0 = (th0 - eps - p.ang0) * p.inv_step, a1 = (th1 + eps - p.ang0) * p.inv_step;
    const int ka = (int)ceilf(a0), kb = (int)floorf(a1);
    int c = kb - ka + 1;
    if (c <= 0) { k0 = 0; cnt = 0; return; }
    if (c >= R) { k0 = 0; cnt = R; return; }
    int m = ka % R;
    k0 = m < 0 ? m + R : m;
    cnt = c;
}

// Entity.get_observation for every agent of the env (entity.py:159-220) is split into a per-env setup
// (agent_setup), independent 64-ray chunks (fan_chunk: any wave of the workgroup may run one) and the
// rewards (rewards_and_positions).  All three read the tick-start snapshot L.fpos / L.ftc / L.fleaf.
struct LateOut { float reward; unsigned tp16; };   // per-lane values stored at the very end of the kernel

// Per-agent setup, published in the env area: grid cell, walls the origin is "inside" (alpha = 0 rule),
// cones of the other agents' circles; resets the per-agent minimum wanted-class distance.
// lane i < A: grid cell of agent i (at positions pos[2A]) and the cell's packed contact row -- ONE global round trip for all agents.  The front requests it
// before its termination check and actions (neither moves the tick-start positions the fan reads), so that the row has arrived when the setup looks at it.
struct SetupRow { int cell; unsigned long long crow; };
__device__ __forceinline__ SetupRow setup_row(const double *pos, const Params &p, const GridDesc &gd, int lane, int A)
{
    SetupRow r = {-1, 0ull};
    if (lane < A) {
        const double ax = pos[2 * lane], ay = pos[2 * lane + 1];
        const int cx = (int)floor((ax - gd.x0) * gd.inv_cell), cy = (int)floor((ay - gd.y0) * gd.inv_cell);
        if (cx >= 0 && cy >= 0 && cx < gd.nx && cy < gd.ny) {
            r.cell = cy * gd.nx + cx;
            r.crow = G(p.cgrid_rows)[gd.crow_base + r.cell];
        }
    }
    return r;
}

template <class D>
__device__ void agent_setup(const Lds &L, const Params &p, const GridDesc &gd, int lane, const SetupRow &row)
{
    const int A = D::A(p), R = D::R(p);
    const double r2 = p.ray_radius;
    const double reach = p.ray_length + r2 + 1e-6;
    const double cone_m = p.gate ? 1e-6 : r2 + 1e-6;
    int my_cell = row.cell, my_near0 = -1, my_near1 = -1, my_dk0 = 0, my_dcnt = 0;
    const unsigned long long crow = row.crow;
    {   // lane = 8 i + q: is agent i's origin within the ray radius of the cell's q-th candidate wall?
        const int pi = lane >> 3, pq = lane & 7;
        const unsigned lo = (unsigned)__shfl((int)(unsigned)crow, pi), hi = (unsigned)__shfl((int)(unsigned)(crow >> 32), pi);
        const unsigned long long row = ((unsigned long long)hi << 32) | lo;
        const int n_i = (int)(row & 0xFF);
        bool near = false;
        int sh = 0;
        if (pi < A && pq < 7 && pq < n_i) {
            sh = (int)((row >> (8 * (pq + 1))) & 0xFF);
            const double ax = L.fpos[2 * pi], ay = L.fpos[2 * pi + 1];
            const double *bb = L.bb + kBB * sh;
            const double m = r2 + 1e-6;
            if ((bb[0] - m <= ax) && (ax <= bb[2] + m) && (bb[1] - m <= ay) && (ay <= bb[3] + m))
                near = poly_point_within(L, sh, p.wall_r, ax, ay, r2);  // [CP cpShapeSegmentQuery] alpha = 0 rule
        }
        const unsigned long long m = __ballot(near);
        // ascending wall ids; more than two such walls cannot matter: the first visited wins at alpha 0
        unsigned mi = lane < A ? (unsigned)((m >> (8 * lane)) & 0x7Full) : 0u;
        const int b0 = mi ? __builtin_ctz(mi) : 0;
        const unsigned mi2 = mi & (mi - 1u);
        const int b1 = mi2 ? __builtin_ctz(mi2) : 0;
        const int id0 = __shfl(sh, (8 * lane + b0) & 63), id1 = __shfl(sh, (8 * lane + b1) & 63);
        if (mi) my_near0 = id0;
        if (mi2) my_near1 = id1;
    }
    // cells with more than 7 contact candidates (dense maps): that agent's list is walked from the CSR arrays
    unsigned long long longm = __ballot(lane < A && (int)(crow & 0xFF) > 7);
    while (longm) {
        const int i = __builtin_ctzll(longm);
        longm &= longm - 1;
        const double ax = L.fpos[2 * i], ay = L.fpos[2 * i + 1];
        const int cellid = __builtin_amdgcn_readlane(my_cell, i);
        int near0 = -1, near1 = -1;
        const int c0 = uni(G(p.cgrid_off)[gd.coff_base + cellid]), c1 = uni(G(p.cgrid_off)[gd.coff_base + cellid + 1]);
        for (int base = c0; base < c1; base += kLanes) {   // lanes stride the cell's contact candidates
            const int e = base + lane;
            bool near = false;
            int sh = 0;
            if (e < c1) {
                sh = G(p.cgrid_ent)[gd.cent_base + e];
                const double *bb = L.bb + kBB * sh;
                const double m = r2 + 1e-6;
                if ((bb[0] - m <= ax) && (ax <= bb[2] + m) && (bb[1] - m <= ay) && (ay <= bb[3] + m))
                    near = poly_point_within(L, sh, p.wall_r, ax, ay, r2);
            }
            unsigned long long m = __ballot(near);
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                const int id = __builtin_amdgcn_readlane(sh, src);
                if (near0 < 0) near0 = id; else if (near1 < 0) near1 = id;
            }
        }
        if (lane == i) { my_near0 = near0; my_near1 = near1; }
    }
    if (lane < A * A) {   // lane = (i, j): cone of agent j's circle seen from agent i
        const int i = lane / A, j = lane % A;
        int k0 = 0, cnt = 0, near = 0;
        if (i != j) {
            const double ax = L.fpos[2 * i], ay = L.fpos[2 * i + 1];
            const double tcx = L.ftc[2 * j], tcy = L.ftc[2 * j + 1];
            double l, b, r, t;
            if (p.gate) { l = L.fleaf[4 * j]; b = L.fleaf[4 * j + 1]; r = L.fleaf[4 * j + 2]; t = L.fleaf[4 * j + 3]; }
            else { l = tcx - p.rc; b = tcy - p.rc; r = tcx + p.rc; t = tcy + p.rc; }
            if ((l <= ax + reach) && (ax - reach <= r) && (b <= ay + reach) && (ay - reach <= t)) {
                const double ex = ax - tcx, ey = ay - tcy;
                near = sqrt(ex * ex + ey * ey) - p.rc <= r2;  // [CP cpCircleShapePointQuery]
                ray_cone(p, ax, ay, l - cone_m, b - cone_m, r + cone_m, t + cone_m, R, k0, cnt);
            }
        }
        my_dk0 = k0; my_dcnt = cnt | (near << 16);
    }
    {
        const unsigned long long nearbits = __ballot((my_dcnt >> 16) & 1);   // lane i * A + j
        if (lane < A) L.adn[lane] = (int)((nearbits >> (lane * A)) & ((1ull << A) - 1ull));
    }
    if (lane < A) { L.acell[lane] = my_cell; L.anear[2 * lane] = my_near0; L.anear[2 * lane + 1] = my_near1; L.dmin[lane] = 0x10000u; }
    if (lane < A * A) { L.dk0[lane] = my_dk0; L.dcnt[lane] = my_dcnt; }
    wave_sync();
}

// One 64-ray chunk c (agent c / cpa, rays (c % cpa) * 64 ...) of the env whose env area is in L; the scratch
// union of L is the calling wave's.  Writes the chunk's observations to the env's output staging.
template <class D>
__device__ void fan_chunk(const Lds &L, const Params &p, const LaunchArgs &la, const GridDesc &gd, long long env, int lane,
                          int S, float cmax, int rew_mode, int c, PhaseClock &pc)
{
    const int A = D::A(p), R = D::R(p);
    const double r2 = p.ray_radius;
    const unsigned d_empty = f64_to_f16(p.ray_length);  // np.full(R, ray_length, float16) entity.py:200
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int cpa = (R + kLanes - 1) / kLanes;   // chunks per agent
    const int rw = uni(p.row_words), row_cap = 8 * rw - 1;
    const int idb = launder(uni(p.row_id_bits)), cmul = launder(uni(p.row_cnt_mul));   // != 0: one word of fields (wall id + 1), finalize_rows
    const int gate = launder(uni(p.gate)), n_cops = launder(uni(D::n_cops(p)));
    const double wall_r = launder(p.wall_r), rc = launder(p.rc);
    // the setup of agent_setup, back into registers (lane i / lane i*A+j), broadcast with readlane below
    const int my_cell = lane < A ? L.acell[lane] : -1, my_near0 = lane < A ? L.anear[2 * lane] : -1,
              my_near1 = lane < A ? L.anear[2 * lane + 1] : -1;
    const int my_dk0 = lane < A * A ? L.dk0[lane] : 0, my_dcnt = lane < A * A ? L.dcnt[lane] : 0;
    const int i = c / cpa, kb = (c % cpa) * kLanes;
    // packed spatial-hash row of (agent cell, ray): every lane loads a valid address (clamped), validity is
    // applied when the row is consumed
    unsigned long long w0, w1 = 0ull, w2 = 0ull, w3 = 0ull;
    {
        const int ck = kb + lane;
        const int cell = __builtin_amdgcn_readlane(my_cell, i);
        const size_t r = (cell < 0 || ck >= R) ? 0 : (size_t)cell * R + ck;
        GAS const unsigned long long *ptr = G(p.grid_rows) + (gd.row_base + r) * rw;
        w0 = ptr[0];
        if (rw > 1) w1 = ptr[1];
        if (rw > 2) { w2 = ptr[2]; w3 = ptr[3]; }
    }
    auto row_byte = [&](int b) -> int {   // b is wave-uniform
        unsigned long long w = w0;
        if (b >= 8) w = b < 16 ? w1 : (b < 24 ? w2 : w3);
        return (int)((w >> (8 * (b & 7))) & 0xFF);
    };
    const double ax = L.fpos[2 * i], ay = L.fpos[2 * i + 1];  // fresh body.position (entity.py:186)
    const int cellid = __builtin_amdgcn_readlane(my_cell, i), near0 = __builtin_amdgcn_readlane(my_near0, i),
              near1 = __builtin_amdgcn_readlane(my_near1, i);
    unsigned dnear_mask = 0;   // other agents whose circle the origin is "inside" (alpha = 0 rule)
    for (int j = 0; j < A; j++) dnear_mask |= (unsigned)((__builtin_amdgcn_readlane(my_dcnt, i * A + j) >> 16) & 1) << j;
    const bool is_cop = i < n_cops;
    const unsigned want = is_cop ? CAT_THIEF : CAT_COP;
    unsigned dmin = 0x10000u;
    {
        const int k = kb + lane;
        const bool active = k < R;
        const int kk = active ? k : 0;
        const double bx = ax + L.rayd[2 * kk], by = ay + L.rayd[2 * kk + 1];  // entity.py:191-193
        const double rdx = bx - ax, rdy = by - ay, rix = 1.0 / rdx, riy = 1.0 / rdy;
        // ---- candidates of this ray: walls from the spatial hash (ascending ids), then the other agents
        int cnt_w = (active && cellid >= 0) ? (int)(w0 & 0xFF) : 0;
        if (idb) cnt_w = (active && cellid >= 0 && w0 != 0ull) ? (((63 - __builtin_clzll(w0)) * cmul) >> 16) + 1 : 0;
        else if (__ballot(cnt_w == 255) != 0ull) {   // saturated count byte (a map with >= 255 walls along one ray)
            if (cnt_w == 255) {
                const size_t r0 = (size_t)cellid * R + k;
                cnt_w = G(p.grid_off)[gd.off_base + r0 + 1] - G(p.grid_off)[gd.off_base + r0];
                asm volatile("" : "+v"(cnt_w));   // consume the loads inside this branch
            }
        }
        unsigned dynmask = 0;
        if (active)
            for (int j = 0; j < A; j++) {
                if (j == i) continue;
                const int dc = __builtin_amdgcn_readlane(my_dcnt, i * A + j) & 0xFFFF, dk = __builtin_amdgcn_readlane(my_dk0, i * A + j);
                int rel = k - dk; if (rel < 0) rel += R;
                if (rel < dc) dynmask |= 1u << j;
            }
        const int cnt = cnt_w + __popc(dynmask);
        PHASE(pc, 20);
        double best_a = 1.0;
        int best_fi = -1;   // id << 6 | feature of the accepted item
        int jj0 = 0;
        while (__ballot(cnt > jj0) != 0ull) {
            // ---- pack the items (ray, jj) for jj in [jj0, jj1) j-major
            int n_items = 0, jj = jj0;
            for (; jj < jj0 + kPassJ; jj++) {
                const bool has = cnt > jj;
                if (__ballot(has) == 0ull) break;
                int id = 0;
                double tbb = 0.0;
                if (has) {
                    if (jj < cnt_w) {
                        if (idb) id = (int)((w0 >> (idb * jj)) & ((1ull << idb) - 1ull)) - 1;
                        else if (jj < row_cap) id = row_byte(jj + 1);
                        else {   // more than 31 candidate walls on one ray: the rest of the list, from the CSR arrays
                            const size_t r0 = (size_t)cellid * R + k;
                            id = G(p.grid_ent)[gd.ent_base + G(p.grid_off)[gd.off_base + r0] + jj];
                            asm volatile("" : "+v"(id));   // consume the load inside this branch
                        }
                    } else {
                        unsigned dj = dynmask;
                        for (int q = jj - cnt_w; q > 0; q--) dj &= dj - 1;
                        id = S + __builtin_ctz(dj);
                    }
                    // the BBTree gate value, by the ray's own lane.  A candidate whose t_bb is not below the
                    // ray's best alpha NOW can never be visited (best only decreases): it gets no item.
                    if (gate) tbb = bb_segment_query((id < S) ? (L.bb + kBB * id) : (L.fleaf + 4 * (id - S)), ax, ay, rdx, rdy, rix, riy);
                }
                const bool live = has && tbb < best_a;
                const unsigned long long m = __ballot(live);
                const int c = __popcll(m);
                if (n_items + c > kItemCap) break;
                int t = 0xFFFF;
                if (live) {
                    t = n_items + __popcll(m & lt_mask);
                    L.itm[t] = (unsigned short)(lane | (id << 6));
                    L.itbb[t] = tbb;
                }
                L.itemidx[(jj - jj0) * kLanes + lane] = (unsigned short)t;
                n_items += c;
            }
            const int jj1 = jj;
            wave_sync();
            PHASE(pc, 5);
            // ---- one item per lane
            for (int t0 = 0; t0 < n_items; t0 += kLanes) {
                const int t = t0 + lane;
                if (t < n_items) {
                    const int d = L.itm[t];
                    const int il = d & 63, id = d >> 6;
                    const int k2 = kb + il;
                    const double cbx = ax + L.rayd[2 * k2], cby = ay + L.rayd[2 * k2 + 1];
                    double alpha = 2.0;   // 2.0 = no hit (never below a best alpha <= 1)
                    int feat = 0;
                    {
                        const bool wall = id < S;
                        const int j = wall ? 0 : id - S;
                        const bool inside = wall ? (id == near0 || id == near1) : (((dnear_mask >> j) & 1u) != 0u);
                        if (inside) { alpha = 0.0; feat = kFeatNear; }
                        else {   // an accepted circle hit at alpha == 1 could never beat the initial best of 1: "t < 1" is equivalent
                            int f;
                            poly_query_feat(L, cmax, wall, wall ? id : 0, wall ? wall_r : rc, L.ftc[2 * j], L.ftc[2 * j + 1], ax, ay, cbx, cby, r2, alpha, f);
                            feat = f < 0 ? 0 : f;
                        }
                    }
                    L.ialpha[t] = alpha; L.itm[t] = (unsigned short)((id << 6) | feat);
                }
            }
            wave_sync();
            PHASE(pc, 6);
            // ---- each ray walks its own items in index order
            for (int q = jj0; q < jj1; q++) {
                const int t = cnt > q ? (int)L.itemidx[(q - jj0) * kLanes + lane] : 0xFFFF;
                if (t != 0xFFFF) {
                    const double al = L.ialpha[t];
                    if (al < best_a && L.itbb[t] < best_a) { best_a = al; best_fi = L.itm[t]; }   // t_exit == best alpha
                }
            }
            wave_sync();
            PHASE(pc, 7);
            jj0 = jj1;
        }
        // ---- hit point -> f16 distance and class (entity.py:200-215, :222-241)
        unsigned d16 = d_empty, ty = CAT_EMPTY;
        int best = -1;
        if (best_fi >= 0) {
            best = best_fi >> 6;
            const int f = best_fi & 63;
            const double t = best_a;
            double px = bx, py = by;  // alpha = 0 hits keep the segment end as their point
            if (f != kFeatNear) {
                const bool wall = best < S;
                const int fc = wall ? L.fc[best] : 0, first = fc & 0xFFFF, count = fc >> 16;
                if (wall && f < count) {
                    const double2 n = *reinterpret_cast<const double2 *>(L.planes + 8 * (first + f));
                    px = (ax * (1.0 - t) + bx * t) - n.x * r2;
                    py = (ay * (1.0 - t) + by * t) - n.y * r2;
                } else {   // corner circle of the hull or the agent's circle: the same formula around a different centre
                    double2 v = *reinterpret_cast<const double2 *>(L.planes + 8 * (first + (wall ? f - count : 0)) + 2);
                    if (!wall) { v.x = L.ftc[2 * (best - S)]; v.y = L.ftc[2 * (best - S) + 1]; }
                    circle_hit_point(v.x, v.y, ax, ay, bx, by, t, r2, px, py);
                }
            }
            d16 = obs_distance_f16(px, py, ax, ay);
            ty = (best < S) ? CAT_WALL : ((best - S) >= n_cops ? CAT_THIEF : CAT_COP);
        }
        if (active) {  // observations go to LDS; one coalesced burst to HBM after the agent loop
            const int q = i * R + k;
            L.od[q] = (unsigned short)d16;
            L.ot[q] = (unsigned char)ty;
            if (la.out.hit_shape) la.out.hit_shape[(size_t)env * A * R + q] = best;  // parity/debug only
            if (ty == want && d16 < dmin) dmin = d16;  // non-negative f16: bit order = value order
        }
        PHASE(pc, 8);
    }
    if (rew_mode) {  // min over the wave, then into the agent's slot (other chunks of the agent may run on other waves)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            unsigned o2 = (unsigned)__shfl_xor((int)dmin, off);
            dmin = o2 < dmin ? o2 : dmin;
        }
        if (lane == 0 && dmin < 0x10000u) __hip_atomic_fetch_min(&L.dmin[i], dmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// Several 64-ray chunks of one slot (chunks c0 .. c0 + nq - 1, nq <= kSlotChunks) as ONE work unit with ONE item list (dense maps, chunk form).  Chunk by
// chunk the item rounds of fan_chunk run 64 + ~20 lanes wide (agh-map: 81 items per chunk, 5.9 rounds of 41 items per env-step): here every chunk's rays are
// gated and packed first, each by its own lane as in fan_chunk, then the shape queries of ALL chunks run in rounds of 64 items, then every chunk's rays walk
// their items and finish.  What a ray keeps between packing and walk is its candidate count and the offset of its items (kSlotPos bits per chunk in one
// register).  The per-ray arithmetic and the visiting rule are fan_chunk's (every candidate gets an item with its gate value; the shape query runs where
// the gate value is below the initial best alpha 1.0; the walk applies the sequential rule), so the results are identical.  The list holds Params::item_cap items;
// chunks that do not fit together are traced in a second turn (a bound on a chunk's items -- the sum of its rays' candidate counts -- is known before it is
// packed), and a chunk that alone exceeds the list is handed back to the caller for fan_chunk.
// Requires (cat_create): one-word rows of id fields (row_id_bits != 0, row_words == 1), shape ids S + A <= 127, at most kSlotPos candidates per ray.
constexpr int kSlotChunks = 3, kSlotPos = 16;   // (kSlotPos bits per chunk in a 64-bit mask; a chunk's first item in 10 bits of a 32-bit word: item_cap < 1024)
template <class D>
__device__ __forceinline__ unsigned fan_slot(const Lds &L, const Params &p, const LaunchArgs &la, const GridDesc &gd, long long env, int lane,
                                             int S, float cmax, int rew_mode, int c0, int nq, PhaseClock &pc)
{
    const int A = D::A(p), R = D::R(p);
    const double r2 = p.ray_radius;
    const unsigned d_empty = f64_to_f16(p.ray_length);  // np.full(R, ray_length, float16) entity.py:200
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int cpa = (R + kLanes - 1) / kLanes;   // chunks per agent
    const int idb = launder(uni(p.row_id_bits)), cmul = launder(uni(p.row_cnt_mul));
    const int gate = launder(uni(p.gate)), n_cops = launder(uni(D::n_cops(p)));
    const int cap = launder(uni(p.item_cap));
    const double wall_r = launder(p.wall_r), rc = launder(p.rc);
    double *const itbb = L.itbb, *const ialpha = itbb + cap;                       // [cap] each: BBTree gate value, hit alpha (2.0 = none)
    unsigned short *const itm = reinterpret_cast<unsigned short *>(ialpha + cap);  // [cap] in: ray lane | id << 6 | chunk << 13   out: id << 6 | feature
    const int my_cell = lane < A ? L.acell[lane] : -1;
    const int my_dk0 = lane < A * A ? L.dk0[lane] : 0, my_dcnt = lane < A * A ? L.dcnt[lane] : 0;
    auto row_count = [&](unsigned long long w) -> int { return w ? (((63 - __builtin_clzll(w)) * cmul) >> 16) + 1 : 0; };
    auto request_row = [&](int q) -> unsigned long long {   // the packed candidate row of (chunk q, this lane's ray); clamped address, validity applied by the consumer
        const int c = c0 + q, i = c / cpa, k = (c - i * cpa) * kLanes + lane;
        const int cell = __builtin_amdgcn_readlane(my_cell, i);
        const size_t r = (cell < 0 || k >= R) ? 0 : (size_t)cell * R + k;
        return G(p.grid_rows)[gd.row_base + r];
    };
    unsigned long long live_bits = 0ull;   // kSlotPos bits per chunk q: the ray's candidate count (4 bits) | the offset of its further candidates' items (10 bits)
    unsigned starts = 0u;                   // bits 10 * q ...: the first item of chunk q (wave-uniform)
    unsigned pending = (1u << nq) - 1u, legacy = 0u;
    while (pending) {   // (one turn, unless the chunks do not fit the list together)
        int n_items = 0;
        unsigned batch = 0u;
        // ---- gate and pack, chunk by chunk, lane = ray (the next chunk's row is requested before this chunk is packed)
        unsigned long long w_next = request_row(__builtin_ctz(pending));
        for (int q = __builtin_ctz(pending); q < nq; q++) {
            const unsigned long long w_got = w_next;
            if (q + 1 < nq) w_next = request_row(q + 1);
            if (!((pending >> q) & 1u)) continue;
            const int c = c0 + q, i = c / cpa, kb = (c - i * cpa) * kLanes, k = kb + lane;
            const int cell = __builtin_amdgcn_readlane(my_cell, i);
            const unsigned long long w = (cell < 0 || k >= R) ? 0ull : w_got;
            unsigned dynmask = 0;   // cone mask of the other agents
            if (k < R)
                for (int j = 0; j < A; j++) {
                    if (j == i) continue;
                    const int dc = __builtin_amdgcn_readlane(my_dcnt, i * A + j) & 0xFFFF, dk = __builtin_amdgcn_readlane(my_dk0, i * A + j);
                    int rel = k - dk; if (rel < 0) rel += R;
                    if (rel < dc) dynmask |= 1u << j;
                }
            const int cnt_w = row_count(w), cnt = cnt_w + __popc(dynmask);
            int bound = 0;   // the chunk's items at most: the sum of its rays' candidate counts (each at most kSlotPos)
#pragma unroll
            for (int b = 0; b < 5; b++) bound += __popcll(__ballot((cnt >> b) & 1)) << b;
            if (n_items + bound > cap) {
                if (batch != 0u) break;                                          // the next turn
                legacy |= 1u << q; pending &= ~(1u << q); continue;              // larger than the list by itself: fan_chunk (the caller)
            }
            const double ax = L.fpos[2 * i], ay = L.fpos[2 * i + 1];  // fresh body.position (entity.py:186)
            const int kk = k < R ? k : 0;
            const double bx = ax + L.rayd[2 * kk], by = ay + L.rayd[2 * kk + 1];  // entity.py:191-193
            const double rdx = bx - ax, rdy = by - ay, rix = 1.0 / rdx, riy = 1.0 / rdy;
            starts = (starts & ~(1023u << (10 * q))) | ((unsigned)n_items << (10 * q));
            // Candidate (ray, position) -> item, every candidate (the gate value goes with it; the shape query is skipped where it says "never visited"):
            // position 0 by the ray's own lane -- every ray of a dense map has one --, the positions from 1 on FLATTENED over the lanes.  Position by position
            // on the ray's own lane the loop runs to the longest list of the chunk (agh-map: five turns, 100 / 25 / 8 / 3 / 1 % of the lanes busy with the f64
            // slab test); flattened, the ~22 further candidates of a chunk take one turn.
            auto candidate_id = [&](unsigned long long row, unsigned dyn, int nw, int jj) -> int {
                if (jj < nw) return (int)((row >> (idb * jj)) & ((1ull << idb) - 1ull)) - 1;
                unsigned dj = dyn;
                for (int z = jj - nw; z > 0; z--) dj &= dj - 1;
                return S + __builtin_ctz(dj);
            };
            const unsigned long long m0 = __ballot(cnt > 0);
            if (cnt > 0) {
                const int id = candidate_id(w, dynmask, cnt_w, 0);
                const int t = n_items + __popcll(m0 & lt_mask);
                itm[t] = (unsigned short)(lane | (id << 6) | (q << 13));
                // the BBTree gate value: a candidate whose t_bb is not below the initial best alpha can never be visited
                itbb[t] = gate ? bb_segment_query((id < S) ? (L.bb + kBB * id) : (L.fleaf + 4 * (id - S)), ax, ay, rdx, rdy, rix, riy) : 0.0;
            }
            n_items += __popcll(m0);
            const int ex = cnt > 1 ? cnt - 1 : 0;   // further candidates of this lane's ray
            int pre = 0, n_ex = 0;                  // exclusive prefix over the lanes, total
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const unsigned long long mb = __ballot((ex >> b) & 1);
                pre += __popcll(mb & lt_mask) << b; n_ex += __popcll(mb) << b;
            }
            if (n_ex) {
                unsigned char *const own = reinterpret_cast<unsigned char *>(ialpha);   // [n_ex] flat index -> ray lane (the alpha array is idle until the shape queries)
                for (int j = 0; __ballot(j < ex) != 0ull; j++)
                    if (j < ex) own[pre + j] = (unsigned char)lane;
                wave_sync();
                const int rix_lo = (int)__double_as_longlong(rix), rix_hi = (int)(__double_as_longlong(rix) >> 32);
                const int riy_lo = (int)__double_as_longlong(riy), riy_hi = (int)(__double_as_longlong(riy) >> 32);
                for (int f0 = 0; f0 < n_ex; f0 += kLanes) {
                    const int f = f0 + lane;
                    const int r = f < n_ex ? (int)own[f] : 0;             // the ray (lane) this candidate belongs to; its row, cone mask and reciprocals by lane permute
                    const int a4 = 4 * r;
                    const int jj = f - __builtin_amdgcn_ds_bpermute(a4, pre) + 1;
                    const unsigned long long rw = ((unsigned long long)(unsigned)__builtin_amdgcn_ds_bpermute(a4, (int)(w >> 32)) << 32) | (unsigned)__builtin_amdgcn_ds_bpermute(a4, (int)w);
                    const unsigned rdyn = (unsigned)__builtin_amdgcn_ds_bpermute(a4, (int)dynmask);
                    const double fix = __longlong_as_double(((long long)__builtin_amdgcn_ds_bpermute(a4, rix_hi) << 32) | (unsigned)__builtin_amdgcn_ds_bpermute(a4, rix_lo));
                    const double fiy = __longlong_as_double(((long long)__builtin_amdgcn_ds_bpermute(a4, riy_hi) << 32) | (unsigned)__builtin_amdgcn_ds_bpermute(a4, riy_lo));
                    if (f < n_ex) {
                        const int kr = kb + r;   // (a ray with candidates lies inside the agent's R rays)
                        const double fbx = ax + L.rayd[2 * kr], fby = ay + L.rayd[2 * kr + 1];
                        const double fdx = fbx - ax, fdy = fby - ay;   // the same operands as on the ray's own lane: bit-identical
                        const int id = candidate_id(rw, rdyn, row_count(rw), jj);
                        const int t = n_items + f;
                        itm[t] = (unsigned short)(r | (id << 6) | (q << 13));
                        itbb[t] = gate ? bb_segment_query((id < S) ? (L.bb + kBB * id) : (L.fleaf + 4 * (id - S)), ax, ay, fdx, fdy, fix, fiy) : 0.0;
                    }
                }
                n_items += n_ex;
                wave_sync();   // the next chunk's `own` table goes over this one
            }
            // what the ray keeps for its walk: its candidate count and where its further candidates' items start
            live_bits = (live_bits & ~(0xFFFFull << (kSlotPos * q))) | ((unsigned long long)((unsigned)cnt | ((unsigned)pre << 4)) << (kSlotPos * q));
            batch |= 1u << q;
        }
        wave_sync();
        PHASE(pc, 5);
        // ---- one item per lane, whatever chunk it comes from
        for (int t0 = 0; t0 < n_items; t0 += kLanes) {
            const int t = t0 + lane;
            if (t < n_items) {
                const int d = itm[t];
                const int il = d & 63, id = (d >> 6) & 127, c = c0 + (d >> 13);
                const int ia = c / cpa, k2 = (c - ia * cpa) * kLanes + il;
                const double2 o2 = *reinterpret_cast<const double2 *>(L.fpos + 2 * ia);
                const double cbx = o2.x + L.rayd[2 * k2], cby = o2.y + L.rayd[2 * k2 + 1];
                double alpha = 2.0;   // 2.0 = no hit (never below a best alpha <= 1)
                int feat = 0;
                if (itbb[t] < 1.0) {   // (else: never visited, whatever the ray has found by then)
                    const bool wall = id < S;
                    const int j = wall ? 0 : id - S;
                    const bool inside = wall ? (id == L.anear[2 * ia] || id == L.anear[2 * ia + 1]) : ((((unsigned)L.adn[ia] >> j) & 1u) != 0u);
                    if (inside) { alpha = 0.0; feat = kFeatNear; }
                    else {   // an accepted circle hit at alpha == 1 could never beat the initial best of 1: "t < 1" is equivalent
                        int f;
                        poly_query_feat(L, cmax, wall, wall ? id : 0, wall ? wall_r : rc, L.ftc[2 * j], L.ftc[2 * j + 1], o2.x, o2.y, cbx, cby, r2, alpha, f);
                        feat = f < 0 ? 0 : f;
                    }
                }
                ialpha[t] = alpha; itm[t] = (unsigned short)((id << 6) | feat);
            }
        }
        wave_sync();
        PHASE(pc, 6);
        // ---- chunk by chunk: each ray walks its own items in index order, then the hit point -> f16 distance and class (entity.py:200-215, :222-241)
        for (int q = 0; q < nq; q++) {
            if (!((batch >> q) & 1u)) continue;
            const int c = c0 + q, i = c / cpa, kb = (c - i * cpa) * kLanes, k = kb + lane;
            const bool active = k < R;
            const unsigned kept = (unsigned)(live_bits >> (kSlotPos * q)) & 0xFFFFu;
            const int cnt = (int)(kept & 15u), pre = (int)(kept >> 4);
            double best_a = 1.0;
            int best_fi = -1;   // id << 6 | feature of the accepted item
            const int base = (int)((starts >> (10 * q)) & 1023u);
            const unsigned long long m0 = __ballot(cnt > 0);
            const int t_first = base + __popcll(m0 & lt_mask), t_more = base + __popcll(m0) + pre - 1;   // the item of position 0; of position jj >= 1: t_more + jj
            for (int jj = 0; __ballot(jj < cnt) != 0ull; jj++) {
                if (jj < cnt) {
                    const int t = jj == 0 ? t_first : t_more + jj;
                    const double al = ialpha[t];
                    if (al < best_a && itbb[t] < best_a) { best_a = al; best_fi = itm[t]; }   // t_exit == best alpha
                }
            }
            PHASE(pc, 7);
            const double ax = L.fpos[2 * i], ay = L.fpos[2 * i + 1];
            const int kk = active ? k : 0;
            const double bx = ax + L.rayd[2 * kk], by = ay + L.rayd[2 * kk + 1];
            unsigned d16 = d_empty, ty = CAT_EMPTY;
            int best = -1;
            if (best_fi >= 0) {
                best = best_fi >> 6;
                const int f = best_fi & 63;
                const double t = best_a;
                double px = bx, py = by;  // alpha = 0 hits keep the segment end as their point
                if (f != kFeatNear) {
                    const bool wall = best < S;
                    const int fc = wall ? L.fc[best] : 0, first = fc & 0xFFFF, count = fc >> 16;
                    if (wall && f < count) {
                        const double2 n = *reinterpret_cast<const double2 *>(L.planes + 8 * (first + f));
                        px = (ax * (1.0 - t) + bx * t) - n.x * r2;
                        py = (ay * (1.0 - t) + by * t) - n.y * r2;
                    } else {   // corner circle of the hull or the agent's circle: the same formula around a different centre
                        double2 v = *reinterpret_cast<const double2 *>(L.planes + 8 * (first + (wall ? f - count : 0)) + 2);
                        if (!wall) { v.x = L.ftc[2 * (best - S)]; v.y = L.ftc[2 * (best - S) + 1]; }
                        circle_hit_point(v.x, v.y, ax, ay, bx, by, t, r2, px, py);
                    }
                }
                d16 = obs_distance_f16(px, py, ax, ay);
                ty = (best < S) ? CAT_WALL : ((best - S) >= n_cops ? CAT_THIEF : CAT_COP);
            }
            unsigned dmin = 0x10000u;
            if (active) {  // observations go to LDS; one coalesced burst to HBM at the write-back
                const int o = i * R + k;
                L.od[o] = (unsigned short)d16;
                L.ot[o] = (unsigned char)ty;
                if (la.out.hit_shape) la.out.hit_shape[(size_t)env * A * R + o] = best;  // parity/debug only
                const unsigned want = i < n_cops ? CAT_THIEF : CAT_COP;
                if (ty == want) dmin = d16;  // non-negative f16: bit order = value order
            }
            if (rew_mode) {  // min over the wave, then into the agent's slot (other chunks of the agent may run elsewhere)
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const unsigned o2 = (unsigned)__shfl_xor((int)dmin, off);
                    dmin = o2 < dmin ? o2 : dmin;
                }
                if (lane == 0 && dmin < 0x10000u) __hip_atomic_fetch_min(&L.dmin[i], dmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            PHASE(pc, 8);
        }
        wave_sync();   // the next turn reuses the item list
        pending &= ~batch;
    }
    return legacy;   // chunks (bit q: chunk c0 + q) left to fan_chunk: their candidate counts alone exceed the list (never on the maps shipped)
}

// The ray fan of an agent GROUP (work unit g: agents g * gsz ..., at most four 64-ray chunks in all), for maps whose rays meet few
// walls -- on the labyrinth 38 of the 64 rays of a chunk have no candidate wall at all, and chunk by chunk every phase still runs
// over all 64 lanes.  Here the rays are first sorted out with lane = ray (packed row loaded, candidate count, cone mask of the other
// agents): a ray without a candidate gets its EMPTY observation at once, the others go into a compact list; then rounds of 64
// ACTIVE rays run the position-major fan of fan_chunk with the origin, the "inside" walls and the roster side per lane.
// Requires (cat_create): every candidate list fits a four-byte row (fields of wall id + 1), shape ids S + A fit 6 bits, R <= kGroupRays.
template <class D>
__device__ void fan_group(const Lds &L, const Params &p, const LaunchArgs &la, const GridDesc &gd, long long env, int lane,
                          int S, float cmax, int rew_mode, int g, int gsz, PhaseClock &pc)
{
    const int A = D::A(p), R = D::R(p);
    const double r2 = p.ray_radius;
    const unsigned d_empty = f64_to_f16(p.ray_length);  // np.full(R, ray_length, float16) entity.py:200
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int cpa = (R + kLanes - 1) / kLanes;   // chunks per agent
    const int i0 = g * gsz, i1 = (i0 + gsz < A) ? i0 + gsz : A;
    const int gate = launder(uni(p.gate)), n_cops = launder(uni(D::n_cops(p)));
    const double wall_r = launder(p.wall_r), rc = launder(p.rc);
    const int my_cell = lane < A ? L.acell[lane] : -1;
    const int my_dk0 = lane < A * A ? L.dk0[lane] : 0, my_dcnt = lane < A * A ? L.dcnt[lane] : 0;
    // ---- lane = ray: the packed rows of the group's chunks (all requested before the first is looked at), then the sorting
    const int nslots = (i1 - i0) * cpa;          // <= 4
    const int idb = launder(uni(p.row_id_bits)), cmul = launder(uni(p.row_cnt_mul));   // four-byte rows: fields of idb bits = id + 1 (finalize_rows)
    auto row_count = [&](unsigned w) -> int { return w ? (((31 - __builtin_clz(w)) * cmul) >> 16) + 1 : 0; };
    unsigned wrow[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int sl = 0; sl < 4; sl++) {
        if (sl < nslots) {
            const int i = i0 + sl / cpa, k = (sl % cpa) * kLanes + lane;
            const int cell = __builtin_amdgcn_readlane(my_cell, i);
            const size_t r = (cell < 0 || k >= R) ? 0 : (size_t)cell * R + k;
            wrow[sl] = ((GAS const unsigned *)G(p.grid_rows))[gd.row_base + r];
        }
    }
    int n_act = 0;
#pragma unroll
    for (int sl = 0; sl < 4; sl++) {
        if (sl < nslots) {
            const int i = i0 + sl / cpa, k = (sl % cpa) * kLanes + lane;
            const int cell = __builtin_amdgcn_readlane(my_cell, i);
            const bool in = k < R;
            const unsigned rowv = (in && cell >= 0) ? wrow[sl] : 0u;   // non-zero: the ray has candidate walls
            unsigned dynmask = 0;
            if (in)
                for (int j = 0; j < A; j++) {
                    if (j == i) continue;
                    const int dc = __builtin_amdgcn_readlane(my_dcnt, i * A + j) & 0xFFFF, dk = __builtin_amdgcn_readlane(my_dk0, i * A + j);
                    int rel = k - dk; if (rel < 0) rel += R;
                    if (rel < dc) dynmask |= 1u << j;
                }
            const bool act = rowv != 0u || dynmask != 0u;
            const unsigned long long m = __ballot(act);
            if (act) {
                const int a = n_act + __popcll(m & lt_mask);
                L.alist[a] = (unsigned char)((sl << 6) | lane);
                L.arow[a] = rowv;
                L.adyn[a] = (unsigned char)dynmask;
            } else if (in) {   // nothing along this ray: its observation is final
                const int q = i * R + k;
                L.od[q] = (unsigned short)d_empty;
                L.ot[q] = (unsigned char)CAT_EMPTY;
                if (la.out.hit_shape) la.out.hit_shape[(size_t)env * A * R + q] = -1;  // parity/debug only
            }
            n_act += __popcll(m);
        }
    }
    n_act = uni(n_act);
    wave_sync();
    PHASE(pc, 20);
    // ---- rounds of 64 active rays
    for (int r0 = 0; r0 < n_act; r0 += kLanes) {
        const bool on = r0 + lane < n_act;
        const int gr = on ? (int)L.alist[r0 + lane] : 0;
        const int sl = gr >> 6;
        const int i = i0 + (cpa == 1 ? sl : (cpa == 2 ? (sl >> 1) : 0));          // this lane's agent
        const int k = (sl - (i - i0) * cpa) * kLanes + (gr & 63);                    // ... and ray
        const double2 org = *reinterpret_cast<const double2 *>(L.fpos + 2 * i);     // fresh body.position (entity.py:186)
        const double ax = org.x, ay = org.y;
        const int near0 = L.anear[2 * i], near1 = L.anear[2 * i + 1];
        const unsigned dnear_mask = (unsigned)L.adn[i];
        const unsigned w0 = on ? L.arow[r0 + lane] : 0u;
        const unsigned dynmask = on ? (unsigned)L.adyn[r0 + lane] : 0u;
        const int cnt_w = row_count(w0);
        const int cnt = cnt_w + __popc(dynmask);
        double rdx, rdy, rix, riy;
        {
            const double bx = ax + L.rayd[2 * k], by = ay + L.rayd[2 * k + 1];  // entity.py:191-193
            rdx = bx - ax; rdy = by - ay; rix = 1.0 / rdx; riy = 1.0 / rdy;
        }
        double best_a = 1.0;
        int best_fi = -1;   // id << 6 | feature of the accepted item
        int jj0 = 0;
        while (__ballot(cnt > jj0) != 0ull) {
            // ---- pack the items (ray, jj) for jj in [jj0, jj1) j-major
            int n_items = 0, jj = jj0;
            for (; jj < jj0 + kPassJ; jj++) {
                const bool has = cnt > jj;
                if (__ballot(has) == 0ull) break;
                int id = 0;
                double tbb = 0.0;
                if (has) {
                    if (jj < cnt_w) id = (int)((w0 >> (idb * jj)) & ((1u << idb) - 1u)) - 1;      // the row holds the whole list
                    else {
                        unsigned dj = dynmask;
                        for (int q = jj - cnt_w; q > 0; q--) dj &= dj - 1;
                        id = S + __builtin_ctz(dj);
                    }
                    // the BBTree gate value, by the ray's own lane.  A candidate whose t_bb is not below the
                    // ray's best alpha NOW can never be visited (best only decreases): it gets no item.
                    if (gate) tbb = bb_segment_query((id < S) ? (L.bb + kBB * id) : (L.fleaf + 4 * (id - S)), ax, ay, rdx, rdy, rix, riy);
                }
                const bool live = has && tbb < best_a;
                const unsigned long long m = __ballot(live);
                const int c = __popcll(m);
                if (n_items + c > kItemCap) break;
                int t = 0xFFFF;
                if (live) {
                    t = n_items + __popcll(m & lt_mask);
                    L.itm[t] = (unsigned short)(lane | (id << 6) | ((i - i0) << 12));
                    L.itbb[t] = tbb;
                }
                L.itemidx[(jj - jj0) * kLanes + lane] = (unsigned short)t;
                n_items += c;
            }
            const int jj1 = jj;
            wave_sync();
            PHASE(pc, 5);
            // ---- one item per lane
            for (int t0 = 0; t0 < n_items; t0 += kLanes) {
                const int t = t0 + lane;
                if (t < n_items) {
                    const int d = L.itm[t];
                    const int il = d & 63, id = (d >> 6) & 63, ia = i0 + (d >> 12);
                    const int g2 = L.alist[r0 + il];
                    const int k2 = ((g2 >> 6) - (ia - i0) * cpa) * kLanes + (g2 & 63);
                    const double2 o2 = *reinterpret_cast<const double2 *>(L.fpos + 2 * ia);
                    const double cbx = o2.x + L.rayd[2 * k2], cby = o2.y + L.rayd[2 * k2 + 1];
                    double alpha = 2.0;   // 2.0 = no hit (never below a best alpha <= 1)
                    int feat = 0;
                    {
                        const bool wall = id < S;
                        const int j = wall ? 0 : id - S;
                        const bool inside = wall ? (id == L.anear[2 * ia] || id == L.anear[2 * ia + 1]) : ((((unsigned)L.adn[ia] >> j) & 1u) != 0u);
                        if (inside) { alpha = 0.0; feat = kFeatNear; }
                        else {   // an accepted circle hit at alpha == 1 could never beat the initial best of 1: "t < 1" is equivalent
                            int f;
                            poly_query_feat(L, cmax, wall, wall ? id : 0, wall ? wall_r : rc, L.ftc[2 * j], L.ftc[2 * j + 1], o2.x, o2.y, cbx, cby, r2, alpha, f);
                            feat = f < 0 ? 0 : f;
                        }
                    }
                    L.ialpha[t] = alpha; L.itm[t] = (unsigned short)((id << 6) | feat);
                }
            }
            wave_sync();
            PHASE(pc, 6);
            // ---- each ray walks its own items in index order
            for (int q = jj0; q < jj1; q++) {
                const int t = cnt > q ? (int)L.itemidx[(q - jj0) * kLanes + lane] : 0xFFFF;
                if (t != 0xFFFF) {
                    const double al = L.ialpha[t];
                    if (al < best_a && L.itbb[t] < best_a) { best_a = al; best_fi = L.itm[t]; }   // t_exit == best alpha
                }
            }
            wave_sync();
            PHASE(pc, 7);
            jj0 = jj1;
        }
        // ---- hit point -> f16 distance and class (entity.py:200-215, :222-241)
        unsigned d16 = d_empty, ty = CAT_EMPTY;
        int best = -1;
        if (best_fi >= 0) {
            const double bx = ax + L.rayd[2 * k], by = ay + L.rayd[2 * k + 1];
            best = best_fi >> 6;
            const int f = best_fi & 63;
            const double t = best_a;
            double px = bx, py = by;  // alpha = 0 hits keep the segment end as their point
            if (f != kFeatNear) {
                const bool wall = best < S;
                const int fc = wall ? L.fc[best] : 0, first = fc & 0xFFFF, count = fc >> 16;
                if (wall && f < count) {
                    const double2 n = *reinterpret_cast<const double2 *>(L.planes + 8 * (first + f));
                    px = (ax * (1.0 - t) + bx * t) - n.x * r2;
                    py = (ay * (1.0 - t) + by * t) - n.y * r2;
                } else {   // corner circle of the hull or the agent's circle: the same formula around a different centre
                    double2 v = *reinterpret_cast<const double2 *>(L.planes + 8 * (first + (wall ? f - count : 0)) + 2);
                    if (!wall) { v.x = L.ftc[2 * (best - S)]; v.y = L.ftc[2 * (best - S) + 1]; }
                    circle_hit_point(v.x, v.y, ax, ay, bx, by, t, r2, px, py);
                }
            }
            d16 = obs_distance_f16(px, py, ax, ay);
            ty = (best < S) ? CAT_WALL : ((best - S) >= n_cops ? CAT_THIEF : CAT_COP);
        }
        if (on) {  // observations go to LDS; one coalesced burst to HBM at the write-back
            const int q = i * R + k;
            L.od[q] = (unsigned short)d16;
            L.ot[q] = (unsigned char)ty;
            if (la.out.hit_shape) la.out.hit_shape[(size_t)env * A * R + q] = best;  // parity/debug only
            const unsigned want = i < n_cops ? CAT_THIEF : CAT_COP;
            // min over the agent's rays (other groups / rounds add theirs); non-negative f16: bit order = value order
            if (rew_mode && ty == want) __hip_atomic_fetch_min(&L.dmin[i], d16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        PHASE(pc, 8);
    }
}

// Cop.reward / Thief.reward (cop.py:49-75, thief.py:48-69; lane = agent) from the per-agent minimum the
// chunks left in L.dmin, and the f16 team positions (observation_spaces.py:92-95: positions BEFORE Space.step).
template <class D>
__device__ void rewards_and_positions(const Lds &L, const Params &p, const LaunchArgs &la, int lane, int rew_mode,
                                      int captured, int timeout, GAS const float *cop_lut, GAS const float *thief_lut, LateOut &late)
{
    const int A = D::A(p);
    late.reward = 0.0f; late.tp16 = 0;
    if (rew_mode && lane < A && la.out.reward) {
        const unsigned my_dmin = L.dmin[lane];
        const bool is_cop = lane < D::n_cops(p);
        float r;
        if (captured) r = is_cop ? 1.0f : -1.0f;
        else if (timeout) r = is_cop ? -1.0f : 1.0f;
        else if (my_dmin < 0x10000u) r = (is_cop ? cop_lut : thief_lut)[my_dmin & 0x7FFFu];
        else r = is_cop ? (float)(-0.02 - 0.02) : (float)0.15;
        late.reward = r;
    }
    if (lane < 2 * A) late.tp16 = f64_to_f16(L.fpos[lane]);
}
// Wait for the LUT value of rewards_and_positions, while no store is in flight: a load still pending when the write-back stores
// start makes the compiler's (in-order) vmcnt waits sit on the acknowledgement of every store issued before
// them -- 18 k cycles per slot in the write-back before this wait existed.  (Called after the LDS-only staging of the shared
// observations, which the lookup's round trip then overlaps.)
__device__ __forceinline__ void await_reward(LateOut &late) { asm volatile("" : "+v"(late.reward)); }

// LDS -> HBM copy of n bytes with the widest store both sides allow (LDS side is 16-byte aligned)
__device__ __forceinline__ void wide_store(void *gdst, const void *lsrc, int n, int lane)
{
    const unsigned long long ga = (unsigned long long)gdst;
    if (((ga | (unsigned)n) & 15u) == 0) {
        for (int o = lane; o < n / 16; o += kLanes) ((GAS u32x4 *)gdst)[o] = reinterpret_cast<const u32x4 *>(lsrc)[o];
    } else if (((ga | (unsigned)n) & 3u) == 0) {
        for (int o = lane; o < n / 4; o += kLanes) ((GAS unsigned *)gdst)[o] = reinterpret_cast<const unsigned *>(lsrc)[o];
    } else if (((ga | (unsigned)n) & 1u) == 0) {
        for (int o = lane; o < n / 2; o += kLanes) ((GAS unsigned short *)gdst)[o] = reinterpret_cast<const unsigned short *>(lsrc)[o];
    } else {
        for (int o = lane; o < n; o += kLanes) ((GAS unsigned char *)gdst)[o] = reinterpret_cast<const unsigned char *>(lsrc)[o];
    }
}

// get_shared_observations (observation_spaces.py:98-129) into the slot's staging (LDS only): first team member, roster order, with a non-EMPTY ray
// supplies (type, distance); else EMPTY with the last member's distance.  Runs while the write-back's reward lookup is in flight.
template <class D>
__device__ __forceinline__ void stage_shared_observations(const Lds &L, const Params &p, int lane)
{
    const int A = D::A(p), R = D::R(p);
    for (int k = lane; k < R; k += kLanes) {
        for (int team = 0; team < 2; team++) {
            const int lo = team ? D::n_cops(p) : 0, hi = team ? A : D::n_cops(p);
            unsigned ty = CAT_EMPTY, d = 0;
            for (int i = lo; i < hi; i++)
                if (ty == CAT_EMPTY) { ty = L.ot[i * R + k]; d = L.od[i * R + k]; }
            L.st[team * R + k] = (unsigned char)ty;
            L.sd[team * R + k] = (unsigned short)d;
        }
    }
    wave_sync();
}

// All output stores, issued at the very end of the kernel: the compiler's s_waitcnt vmcnt(0) (in-order
// with stores, and forced by every flat access) would otherwise stall mid-kernel on HBM write latency.
template <class D>
__device__ __forceinline__ void emit_observations(const Lds &L, const Params &p, const LaunchArgs &la, long long env, int lane,
                                                  int rew_mode, const LateOut &late)
{
    const int A = D::A(p), R = D::R(p);
    if (rew_mode && lane < A && la.out.reward) la.out.reward[(size_t)env * A + lane] = late.reward;
    if (lane < 2 * A && la.out.team_positions) la.out.team_positions[(size_t)env * A * 2 + lane] = (unsigned short)late.tp16;
    const size_t g0 = (size_t)env * A * R;   // Entity.get_observation outputs: [A*R] contiguous per env
    if (la.out.obs_distance) wide_store(la.out.obs_distance + g0, L.od, A * R * 2, lane);
    if (la.out.obs_type) wide_store(la.out.obs_type + g0, L.ot, A * R, lane);
    if (la.out.shared_distance) wide_store(la.out.shared_distance + (size_t)env * 2 * R, L.sd, 2 * R * 2, lane);
    if (la.out.shared_type) wide_store(la.out.shared_type + (size_t)env * 2 * R, L.st, 2 * R, lane);
}

// ------------------------------------------------------------------ termination ---------------
// BaseEnv._termination_criterion (base_env.py:521-554).  The wall-only LOS query is only consulted
// for pairs inside the capture radius, so it is evaluated only there; lanes stride the walls.
template <class D>
__device__ int termination_captured(const Lds &L, const Params &p, int S, int lane)
{
    const int A = D::A(p), nc = D::n_cops(p), npairs = (A - nc) * nc;   // <= 16
    // lane = pair (thief-major, as the reference's nested loops): inside the capture radius?
    bool within = false;
    if (lane < npairs) {
        const int t = nc + lane / nc, c = lane % nc;
        const double ddx = L.pos[2 * t] - L.pos[2 * c], ddy = L.pos[2 * t + 1] - L.pos[2 * c + 1];  // Vec2d.get_distance
        within = sqrt(ddx * ddx + ddy * ddy) < p.term_radius;
    }
    unsigned long long cand = __ballot(within);
    while (cand) {   // in pair order; the first pair with a clear line of sight captures
        {
            const int pair = __builtin_ctzll(cand);
            cand &= cand - 1;
            const int t = nc + pair / nc, c = pair % nc;
            const double ax = L.pos[2 * t], ay = L.pos[2 * t + 1], bx = L.pos[2 * c], by = L.pos[2 * c + 1];
            const double dx = bx - ax, dy = by - ay, idx = 1.0 / dx, idy = 1.0 / dy;
            bool any = false;
            for (int base = 0; base < S; base += kLanes) {
                const int s = base + lane;
                bool hit = false;
                if (s < S) {
                    bool visit = true;
                    if (p.gate) visit = bb_segment_query(L.bb + kBB * s, ax, ay, dx, dy, idx, idy) < 1.0;
                    if (visit) {
                        SegInfo info = {0, 1.0, bx, by};
                        if (poly_point_distance(L, s, p.wall_r, ax, ay) <= 0.0) { info.hit = 1; info.alpha = 0.0; }
                        else poly_segment_query(L, s, p.wall_r, ax, ay, bx, by, 0.0, info);
                        hit = info.hit && info.alpha < 1.0;
                    }
                }
                any = any || (__ballot(hit) != 0ull);
            }
            if (!any) return 1;
        }
    }
    return 0;
}
