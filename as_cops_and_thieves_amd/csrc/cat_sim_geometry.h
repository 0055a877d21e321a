// cat_sim_geometry.h -- part of the env core's single translation unit (included by cat_sim.hip, in this order; not a stand-alone header):
// segment / point queries against boxes, circles and rounded hulls ([CP cpBBSegmentQuery], [CP CircleSegmentQuery], [CP cpPolyShapeSegmentQuery], [CP cpPolyShapePointQuery]).
// ------------------------------------------------------------------ geometry ------------------
// [CP cpBBSegmentQuery]; slab test multiplies by 1/delta (DESIGN.md deviation D3)
__device__ __forceinline__ double bb_segment_query(const double *bb, double ax, double ay, double dx,
                                                   double dy, double idx, double idy)
{
    // [CP cpfmax / cpfmin] as v_max_f64 / v_min_f64 here: the operands are never NaN (a zero delta takes the other
    // branch, every other delta is at least an ulp of a coordinate, so 1/delta is finite) and the sign of a zero result
    // is immaterial -- the value is only ever compared.  One instruction instead of a compare and two selects.
    const double2 lo = *reinterpret_cast<const double2 *>(bb);
    const double2 hi = *reinterpret_cast<const double2 *>(bb + 2);
    double tmin = -INFINITY, tmax = INFINITY;
    if (dx == 0.0) {
        if (ax < lo.x || hi.x < ax) return INFINITY;
    } else {
        double t1 = (lo.x - ax) * idx, t2 = (hi.x - ax) * idx;
        tmin = CAT_FMAX(tmin, CAT_FMIN(t1, t2));
        tmax = CAT_FMIN(tmax, CAT_FMAX(t1, t2));
    }
    if (dy == 0.0) {
        if (ay < lo.y || hi.y < ay) return INFINITY;
    } else {
        double t1 = (lo.y - ay) * idy, t2 = (hi.y - ay) * idy;
        tmin = CAT_FMAX(tmin, CAT_FMIN(t1, t2));
        tmax = CAT_FMIN(tmax, CAT_FMAX(t1, t2));
    }
    if (tmin <= tmax && 0.0 <= tmax && tmin <= 1.0) return CAT_FMAX(tmin, 0.0);
    return INFINITY;
}

struct SegInfo { int hit; double alpha, px, py; };

// [CP CircleSegmentQuery]
__device__ __forceinline__ void circle_segment_query(double cx, double cy, double r1, double ax, double ay,
                                                     double bx, double by, double r2, SegInfo &info)
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
            double lx = dax * (1.0 - t) + dbx * t, ly = day * (1.0 - t) + dby * t;
            double inv = 1.0 / (sqrt(lx * lx + ly * ly) + DBL_MIN);
            double nx = lx * inv, ny = ly * inv;
            info.hit = 1;
            info.alpha = t;
            info.px = (ax * (1.0 - t) + bx * t) - nx * r2;
            info.py = (ay * (1.0 - t) + by * t) - ny * r2;
        }
    }
}

// [CP cpPolyShapeSegmentQuery]; plane record = n.x n.y v0.x v0.y dot(v0,n) dtMin dtMax pad.
// Chipmunk runs all face planes first (a passing plane overwrites the result unconditionally) and
// then the bevel circles (strictly smaller alpha replaces).  The two passes only interact through
// "min, earlier wins ties", so one loop over the records that tracks the plane result and the best
// bevel result separately and merges them afterwards gives the identical answer.
__device__ __forceinline__ void poly_segment_query(const Lds &L, int sh, double r, double ax, double ay,
                                                   double bx, double by, double r2, SegInfo &info)
{
    const int fc = L.fc[sh], first = fc & 0xFFFF, count = fc >> 16;
    const double rsum = r + r2;
    // Conservative f32 pre-test for the bevels: a circle whose centre lies farther than rsum + 0.01 from
    // the ray's line cannot be hit (the exact f64 discriminant is then negative by a margin ~1e3 that
    // dwarfs its ~1e-4 rounding error), so the exact test is skipped for it.
    const float dxf = (float)(bx - ax), dyf = (float)(by - ay);
    const float thr = ((float)rsum + 0.01f) * sqrtf(dxf * dxf + dyf * dyf) * 1.00001f + 0.25f;
    const bool bevels = rsum > 0.0;
    SegInfo ci = {0, 1.0, bx, by};  // best bevel hit so far
    const double *pl = L.planes + 8 * first;
    for (int i = 0; i < count; i++, pl += 8) {
        const double2 n = *reinterpret_cast<const double2 *>(pl);
        const double2 v = *reinterpret_cast<const double2 *>(pl + 2);
        const double2 e0 = *reinterpret_cast<const double2 *>(pl + 4);  // vn, dtMin
        double an = ax * n.x + ay * n.y;
        double d = an - e0.x - rsum;
        if (!(d < 0.0)) {
            double bn = bx * n.x + by * n.y;
            double den = fmax2(an - bn, DBL_MIN);
            if (!(d > den)) {  // d > den <=> fl(d/den) > 1: exact pre-reject before the division
                double t = d / den;
                if (!(t < 0.0 || 1.0 < t)) {
                    double ptx = ax * (1.0 - t) + bx * t, pty = ay * (1.0 - t) + by * t;
                    double dtv = n.x * pty - n.y * ptx;
                    if (e0.y <= dtv && dtv <= pl[6]) {
                        info.hit = 1;
                        info.alpha = t;
                        info.px = ptx - n.x * r2;
                        info.py = pty - n.y * r2;
                    }
                }
            }
        }
        if (bevels) {
            const float ex = (float)(v.x - ax), ey = (float)(v.y - ay);
            if (!(fabsf(dxf * ey - dyf * ex) > thr)) {
                SegInfo c2 = {0, 1.0, bx, by};
                circle_segment_query(v.x, v.y, r, ax, ay, bx, by, r2, c2);
                if (c2.alpha < ci.alpha) ci = c2;
            }
        }
    }
    if (ci.alpha < info.alpha) info = ci;
}

// [CP cpPolyShapePointQuery] -> signed distance to the rounded surface
__device__ __forceinline__ double poly_point_distance(const Lds &L, int sh, double r, double px, double py)
{
    const int fc = L.fc[sh], first = fc & 0xFFFF, count = fc >> 16;
    const double *last = L.planes + 8 * (first + count - 1);
    double v0x = last[2], v0y = last[3];
    double minDist = INFINITY;
    bool outside = false;
    for (int i = 0; i < count; i++) {
        const double *pl = L.planes + 8 * (first + i);
        const double2 n = *reinterpret_cast<const double2 *>(pl);
        const double2 v1 = *reinterpret_cast<const double2 *>(pl + 2);
        outside = outside || (n.x * (px - v1.x) + n.y * (py - v1.y) > 0.0);
        double dx = v0x - v1.x, dy = v0y - v1.y;  // [CP cpClosetPointOnSegment]
        double t = (dx * (px - v1.x) + dy * (py - v1.y)) / (dx * dx + dy * dy);
        t = fmax2(0.0, fmin2(t, 1.0));
        double cx = v1.x + dx * t, cy = v1.y + dy * t;
        double ex = px - cx, ey = py - cy;
        double dist = sqrt(ex * ex + ey * ey);
        if (dist < minDist) minDist = dist;
        v0x = v1.x; v0y = v1.y;
    }
    double dist = outside ? minDist : -minDist;
    return dist - r;
}

// poly_point_distance(L, sh, r, px, py) <= lim, evaluated exactly behind a reject that no rounding can fool: a point outside one face plane by more
// than r + lim + 1e-6 is farther than that from the hull (which lies behind every plane), and the distance's own error is ~1e-12.  The setup's
// "origin inside the query radius" test runs on walls whose inflated bb holds the origin: on a map of slanted footprints (agh-map) that is often a
// 20-edge hull many pixels away, and the full distance costs a divide and a square root per edge (10.5 k cycles of an agh-map front before this).
__device__ __forceinline__ bool poly_point_within(const Lds &L, int sh, double r, double px, double py, double lim)
{
    const int fc = L.fc[sh], first = fc & 0xFFFF, count = fc >> 16;
    const double far = r + lim + 1e-6;
    bool out = false;
    for (int i = 0; i < count; i++) {
        const double *pl = L.planes + 8 * (first + i);
        const double2 n = *reinterpret_cast<const double2 *>(pl);
        const double2 v1 = *reinterpret_cast<const double2 *>(pl + 2);
        out = out || (n.x * (px - v1.x) + n.y * (py - v1.y) > far);
    }
    if (out) return false;
    return poly_point_distance(L, sh, r, px, py) <= lim;
}
