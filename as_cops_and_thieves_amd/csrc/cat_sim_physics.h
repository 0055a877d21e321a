// cat_sim_physics.h -- part of the env core's single translation unit (included by cat_sim.hip, in this order; not a stand-alone header):
// pymunk Space.step -> [CP cpSpaceStep] for one env.
// ------------------------------------------------------------------ physics -------------------

// closest hull feature + [CP ClosestPointsNew] -> contact of [CP CircleToPoly].
// Called wave-uniformly; lane i evaluates hull edge i (ClosestT / LerpT of its Minkowski edge), the
// closest edge is then found by a scalar scan over the per-lane results (lowest index wins ties, as
// in the sequential loop), and every lane finishes the winning edge identically.
__device__ int circle_poly_contact(const Lds &L, int sh, double rp, double cx, double cy, double rc, int lane,
                                   double &nx, double &ny, double &p1x, double &p1y, double &p2x, double &p2y)
{
    const int fc = uni(L.fc[sh]), first = fc & 0xFFFF, count = fc >> 16;
    double sep = -INFINITY, dd = INFINITY, tt = 0.0, ppx = 0.0, ppy = 0.0;
    if (lane < count) {
        const double *pl = L.planes + 8 * (first + lane);
        sep = pl[0] * (cx - pl[2]) + pl[1] * (cy - pl[3]);
    }
    // a plane farther than rc + rp from the centre separates: no contact (hull lies behind every plane).  Checked before the per-edge closest
    // points (a divide each): on a map of slanted footprints most bb overlaps end here
    if (__ballot(sep > rc + rp + 1e-9) != 0ull) return 0;
    if (lane < count) {
        const int i = lane;
        const double *pl = L.planes + 8 * (first + i);
        const double *pv = L.planes + 8 * (first + (i - 1 + count) % count);
        // Minkowski points (poly vertex - circle centre); GJK's final ordering for a CCW hull: v0 = vert[i], v1 = vert[i-1]
        double ax_ = pl[2] - cx, ay_ = pl[3] - cy, bx_ = pv[2] - cx, by_ = pv[3] - cy;
        double dx = bx_ - ax_, dy = by_ - ay_;
        double t = -fmin2(fmax2((dx * (ax_ + bx_) + dy * (ay_ + by_)) / (dx * dx + dy * dy), -1.0), 1.0);  // [CP ClosestT]
        double ht = 0.5 * t;                                                                                  // [CP LerpT]
        ppx = ax_ * (0.5 - ht) + bx_ * (0.5 + ht); ppy = ay_ * (0.5 - ht) + by_ * (0.5 + ht);
        dd = ppx * ppx + ppy * ppy;
        tt = t;
    }
    const bool inside = __ballot(sep > 0.0) == 0ull;
    int best = 0, sepi = 0;
    double bestd = INFINITY, maxsep = -INFINITY;
    for (int i = 0; i < count; i++) {   // scalar scan, index order
        const double di = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(dd) >> 32), i) << 32) |
                                               (unsigned)__builtin_amdgcn_readlane((int)__double_as_longlong(dd), i));
        const double si = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(sep) >> 32), i) << 32) |
                                               (unsigned)__builtin_amdgcn_readlane((int)__double_as_longlong(sep), i));
        // GJK only terminates on an edge the origin lies in front of: at a vertex shared with an edge the
        // centre is behind, the tie goes to the other edge (whose normal gives d > 0: vertex/vertex branch)
        if (si > 0.0 && di < bestd) { bestd = di; best = i; }
        if (si > maxsep) { maxsep = si; sepi = i; }
    }
    if (inside) {  // centre inside the hull: least-penetration plane (deviation D4)
        const double *pl = L.planes + 8 * (first + sepi);
        double d = maxsep;
        if (!(d <= rc + rp)) return 0;
        nx = -pl[0]; ny = -pl[1];
        p1x = cx + nx * rc; p1y = cy + ny * rc;
        double qx = cx - pl[0] * d, qy = cy - pl[1] * d;
        p2x = qx + nx * (-rp); p2y = qy + ny * (-rp);
        return 1;
    }
    auto bcast = [&](double v) -> double {
        return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), best) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane((int)__double_as_longlong(v), best));
    };
    const double bt = bcast(tt), bpx = bcast(ppx), bpy = bcast(ppy);
    const double *pl = L.planes + 8 * (first + best);
    const double *pv = L.planes + 8 * (first + (best - 1 + count) % count);
    double ax_ = pl[2] - cx, ay_ = pl[3] - cy, bx_ = pv[2] - cx, by_ = pv[3] - cy;
    double t = bt, ht = 0.5 * t;
    double pax = cx * (0.5 - ht) + cx * (0.5 + ht), pay = cy * (0.5 - ht) + cy * (0.5 + ht);
    double pbx = pl[2] * (0.5 - ht) + pv[2] * (0.5 + ht), pby = pl[3] * (0.5 - ht) + pv[3] * (0.5 + ht);
    double dx = bx_ - ax_, dy = by_ - ay_;
    double rx = dy, ry = -dx;
    double inv = 1.0 / (sqrt(rx * rx + ry * ry) + DBL_MIN);
    double n_x = rx * inv, n_y = ry * inv;
    double d = n_x * bpx + n_y * bpy;
    if (!(d <= 0.0 || (-1.0 < t && t < 1.0))) {
        double d2 = sqrt(bpx * bpx + bpy * bpy);
        double inv2 = 1.0 / (d2 + DBL_MIN);
        n_x = bpx * inv2; n_y = bpy * inv2;
        d = d2;
    }
    if (!(d <= rc + rp)) return 0;
    nx = n_x; ny = n_y;
    p1x = pax + n_x * rc; p1y = pay + n_y * rc;
    p2x = pbx + n_x * (-rp); p2y = pby + n_y * (-rp);
    return 1;
}

// contact record q in LDS, kConD doubles at conf + kConD * q: nx ny r1x r1y r2x r2y nMass bias jBias jnAcc bounce - | (ints) a b first cache_index
// (wall: i*K+slot, pair: 1<<20 | pi).  One record per contact, so the carve needs no contact count; Params::maxc (what cat_create proves possible for
// the sim's maps: agents x the deepest overlap of wall bbs an agent's bb can reach, + agent pairs) sizes the array, and a contact beyond it -- never
// on a map cat_create accepted -- is dropped and flagged (CAT_DEVERR_CONTACT_DROPPED) instead of written.
constexpr int kConD = 14;
// [CP cpSpaceStep] for one env.  Executed wave-uniformly (every lane computes the same values and
// stores them to the same LDS words) except the bb-overlap test, where lanes stride the walls.
template <class D>
__device__ void physics_env(const Lds &L, const Params &p, int S, int lane, PhaseClock &pc)
{
    const int A = D::A(p);
    const double dt = p.dt, rc = p.rc;
    if (lane < A) {   // lane = agent
        const int i = lane;
        // [CP cpBodyUpdatePosition]
        double px = L.pos[2 * i] + (L.vel[2 * i] + L.vb[2 * i]) * dt;
        double py = L.pos[2 * i + 1] + (L.vel[2 * i + 1] + L.vb[2 * i + 1]) * dt;
        L.pos[2 * i] = px; L.pos[2 * i + 1] = py;
        L.vb[2 * i] = 0.0; L.vb[2 * i + 1] = 0.0;
        L.tc[2 * i] = px; L.tc[2 * i + 1] = py;  // [CP cpCircleShapeCacheData]
        double bl = px - rc, bb_ = py - rc, br = px + rc, bt = py + rc;
        double *lf = L.leaf + 4 * i;             // [CP LeafUpdate] / [CP GetBB]
        if (!(lf[0] <= bl && lf[2] >= br && lf[1] <= bb_ && lf[3] >= bt)) {
            double x = (br - bl) * 0.1, y = (bt - bb_) * 0.1;
            double vx = L.vel[2 * i] * 0.1, vy = L.vel[2 * i + 1] * 0.1;
            lf[0] = bl + fmin2(-x, vx); lf[1] = bb_ + fmin2(-y, vy);
            lf[2] = br + fmax2(x, vx); lf[3] = bt + fmax2(y, vy);
        }
    }
    wave_sync();
    PHASE(pc, 12);
    int nc = 0;
    unsigned long long seen_w = 0ull;  // bit i*K+slot (A*K <= 64)
    unsigned seen_p = 0u;
    // arbiter cache snapshot: lane q = (agent, slot)
    const int my_wsh = lane < A * kK ? L.wsh[lane] : -1;
    // lane = wall: its bb against every agent's circle bb, all walls in one LDS round; bit i of ovm[q]: wall 64 q + lane
    // overlaps agent i.  The contacts are then created agent by agent, walls ascending (the order fixes the solver's).
    unsigned ovm[CAT_MAX_SHAPES / kLanes] = {0u, 0u, 0u, 0u};
    bool any_ov = false;
#pragma unroll
    for (int q = 0; q < CAT_MAX_SHAPES / kLanes; q++) {
        const int s = q * kLanes + lane;
        if (s < S) {
            const double *sb = L.bb + kBB * s;  // [CP cpBBIntersects]
            const double s0 = sb[0], s1 = sb[1], s2 = sb[2], s3 = sb[3];
            for (int i = 0; i < A; i++) {
                const double cx = L.tc[2 * i], cy = L.tc[2 * i + 1];
                const double bl = cx - rc, bb_ = cy - rc, br = cx + rc, bt = cy + rc;
                ovm[q] |= (unsigned)(bl <= s2 && s0 <= br && bb_ <= s3 && s1 <= bt) << i;
            }
            any_ov = any_ov || ovm[q] != 0u;
        }
    }
    const bool some = __ballot(any_ov) != 0ull;
    for (int i = 0; some && i < A; i++) {
        const double cx = L.tc[2 * i], cy = L.tc[2 * i + 1];
#pragma unroll
        for (int q = 0; q < CAT_MAX_SHAPES / kLanes; q++) {
            const int base = q * kLanes;
            if (base >= S) break;
            unsigned long long m = __ballot((ovm[q] >> i) & 1u);
            while (m) {
                const int sh = base + __builtin_ctzll(m);
                m &= m - 1;
                double nx, ny, p1x, p1y, p2x, p2y;
                if (!circle_poly_contact(L, sh, p.wall_r, cx, cy, rc, lane, nx, ny, p1x, p1y, p2x, p2y)) continue;
                if (nc >= p.maxc) { if (lane == 0) atomicOr(p.err_word, CAT_DEVERR_CONTACT_DROPPED); continue; }   // (the contact array is full: see kConD)
                // arbiter cache lookup [CP cpSpaceCollideShapes / cpArbiterUpdate]: lanes = the agent's slots
                const bool mine = lane >= i * kK && lane < (i + 1) * kK;
                const int cur = mine ? L.wsh[lane] : -2;
                unsigned long long hit = __ballot(cur == sh), freem = __ballot(cur == -1);
                int slot, first;
                if (hit) { slot = __builtin_ctzll(hit) - i * kK; first = uni(L.wag[i * kK + slot]) > 0; }
                else {
                    first = 1;
                    if (freem) slot = __builtin_ctzll(freem) - i * kK;
                    else {   // evict the oldest entry not seen this step (table full of live contacts: drop)
                        int oldest = -1, oldage = -1;
                        for (int k = 0; k < kK; k++) {
                            const int ag = uni(L.wag[i * kK + k]);
                            if (!((seen_w >> (i * kK + k)) & 1ull) && ag > oldage) { oldest = k; oldage = ag; }
                        }
                        if (oldest < 0) {   // all CAT_WALL_CACHE slots hold contacts of THIS step: the contact gets no constraint
                            if (lane == 0) atomicOr(p.err_word, CAT_DEVERR_CONTACT_DROPPED);
                            continue;
                        }
                        slot = oldest;
                    }
                    L.wsh[i * kK + slot] = sh; L.wjn[i * kK + slot] = 0.0; L.wag[i * kK + slot] = 0;
                }
                seen_w |= 1ull << (i * kK + slot);
                double *cf = L.conf + kConD * nc;
                int *ci = reinterpret_cast<int *>(cf + 12);
                cf[0] = nx; cf[1] = ny;
                cf[2] = p1x - L.pos[2 * i]; cf[3] = p1y - L.pos[2 * i + 1];
                cf[4] = p2x - 0.0; cf[5] = p2y - 0.0;
                cf[9] = L.wjn[i * kK + slot];
                ci[0] = i; ci[1] = -1; ci[2] = first; ci[3] = i * kK + slot;
                nc++;
            }
        }
    }
    PHASE(pc, 13);
    {   // [CP CircleToCircle] candidates: lane = pair index, then the (rare) overlapping pairs in order
        bool touch = false;
        if (lane < D::NP(p)) {
            int i = 0, rem = lane;
            while (rem >= A - 1 - i) { rem -= A - 1 - i; i++; }
            const int j = i + 1 + rem;
            const double dx = L.tc[2 * j] - L.tc[2 * i], dy = L.tc[2 * j + 1] - L.tc[2 * i + 1];
            const double mindist = rc + rc;
            touch = dx * dx + dy * dy < mindist * mindist;
        }
        unsigned long long pm = __ballot(touch);
        while (pm) {
            const int pi = __builtin_ctzll(pm);
            pm &= pm - 1;
            int i = 0, rem = pi;
            while (rem >= A - 1 - i) { rem -= A - 1 - i; i++; }
            const int j = i + 1 + rem;
            double mindist = rc + rc;
            double dx = L.tc[2 * j] - L.tc[2 * i], dy = L.tc[2 * j + 1] - L.tc[2 * i + 1];
            double distsq = dx * dx + dy * dy;
            if (!(distsq < mindist * mindist)) continue;
            if (nc >= p.maxc) { if (lane == 0) atomicOr(p.err_word, CAT_DEVERR_CONTACT_DROPPED); continue; }
            double dist = sqrt(distsq);
            double nx = 1.0, ny = 0.0;
            if (dist != 0.0) { double inv = 1.0 / dist; nx = dx * inv; ny = dy * inv; }
            int first;
            const int page = uni(L.pag[pi]);
            if (page < 0) { first = 1; L.pjn[pi] = 0.0; }
            else first = page > 0;
            L.pag[pi] = 0; seen_p |= 1u << pi;
            double *cf = L.conf + kConD * nc;
            int *ci = reinterpret_cast<int *>(cf + 12);
            double p1x = L.tc[2 * i] + nx * rc, p1y = L.tc[2 * i + 1] + ny * rc;
            double p2x = L.tc[2 * j] + nx * (-rc), p2y = L.tc[2 * j + 1] + ny * (-rc);
            cf[0] = nx; cf[1] = ny;
            cf[2] = p1x - L.pos[2 * i]; cf[3] = p1y - L.pos[2 * i + 1];
            cf[4] = p2x - L.pos[2 * j]; cf[5] = p2y - L.pos[2 * j + 1];
            cf[9] = L.pjn[pi];
            ci[0] = i; ci[1] = j; ci[2] = first; ci[3] = (1 << 20) | pi;
            nc++;
        }
    }
    (void)my_wsh;
    wave_sync();
    PHASE(pc, 14);
    // [CP cpSpaceArbiterSetFilter]: age / expire, lane = cache entry
    if (lane < A * kK && L.wsh[lane] >= 0) {
        if ((seen_w >> lane) & 1ull) L.wag[lane] = 0;
        else {
            const int a = L.wag[lane] + 1;
            if (a >= p.persistence) { L.wsh[lane] = -1; L.wag[lane] = 0; L.wjn[lane] = 0.0; }
            else L.wag[lane] = a;
        }
    }
    if (lane < D::NP(p)) {
        const int page = L.pag[lane];
        if (page >= 0 && !((seen_p >> lane) & 1u)) {
            const int a = page + 1;
            if (a >= p.persistence) { L.pag[lane] = -1; L.pjn[lane] = 0.0; }
            else L.pag[lane] = a;
        }
    }
    wave_sync();
    PHASE(pc, 15);
    if (nc == 0) return;
    wave_sync();   // contact records were written by every lane identically; make them visible per lane
    // Solver: lane q owns contact q and keeps its constants and accumulators in registers; the bodies stay
    // in LDS.  Arbiters are processed strictly in list order (only lane q is active in step q), which is
    // what makes the result equal to the sequential Gauss-Seidel of Chipmunk bit for bit.
    const double m_inv = 1.0 / p.mass;
    const int q = lane;
    const bool own = q < nc;
    int ca = 0, cb = -1, cfirst = 1, cidx = 0;
    double nx = 0, ny = 0, nMass = 0, bias = 0, jBiasAcc = 0.0, jnAcc = 0, bounce = 0;
    if (own) {   // [CP cpArbiterPreStep]
        const double *cf = L.conf + kConD * q;
        const int *ci = reinterpret_cast<const int *>(cf + 12);
        ca = ci[0]; cb = ci[1]; cfirst = ci[2]; cidx = ci[3];
        nx = cf[0]; ny = cf[1]; jnAcc = cf[9];
        const double mib = (cb < 0) ? 0.0 : m_inv;
        nMass = 1.0 / (m_inv + mib);
        const double bpx = (cb < 0) ? 0.0 : L.pos[2 * cb], bpy = (cb < 0) ? 0.0 : L.pos[2 * cb + 1];
        const double bdx = bpx - L.pos[2 * ca], bdy = bpy - L.pos[2 * ca + 1];
        const double dist = ((cf[4] - cf[2]) + bdx) * nx + ((cf[5] - cf[3]) + bdy) * ny;
        bias = -p.bias_coef * fmin2(0.0, dist + p.slop) / dt;
        const double vbx = (cb < 0) ? 0.0 : L.vel[2 * cb], vby = (cb < 0) ? 0.0 : L.vel[2 * cb + 1];
        bounce = ((vbx - L.vel[2 * ca]) * nx + (vby - L.vel[2 * ca + 1]) * ny) * 0.0;   // e = 0
    }
    wave_sync();
    for (int step = 0; step < nc; step++) {  // [CP cpArbiterApplyCachedImpulse], dt_coef = 1
        if (q == step && !cfirst) {
            const double jx = (nx * jnAcc - ny * 0.0) * 1.0, jy = (nx * 0.0 + ny * jnAcc) * 1.0;
            L.vel[2 * ca] = L.vel[2 * ca] + (-jx) * m_inv; L.vel[2 * ca + 1] = L.vel[2 * ca + 1] + (-jy) * m_inv;
            if (cb >= 0) { L.vel[2 * cb] = L.vel[2 * cb] + jx * m_inv; L.vel[2 * cb + 1] = L.vel[2 * cb + 1] + jy * m_inv; }
        }
    }
    for (int it = 0; it < p.iterations; it++) {  // [CP cpArbiterApplyImpulse]
        for (int step = 0; step < nc; step++) {
            if (q == step) {
                const double2 va = *reinterpret_cast<const double2 *>(L.vel + 2 * ca);
                const double2 vba = *reinterpret_cast<const double2 *>(L.vb + 2 * ca);
                double2 vb2 = {0.0, 0.0}, vbb2 = {0.0, 0.0};
                if (cb >= 0) { vb2 = *reinterpret_cast<const double2 *>(L.vel + 2 * cb); vbb2 = *reinterpret_cast<const double2 *>(L.vb + 2 * cb); }
                const double vbn = (vbb2.x - vba.x) * nx + (vbb2.y - vba.y) * ny;
                const double vrn = (vb2.x - va.x) * nx + (vb2.y - va.y) * ny;
                const double jbn = (bias - vbn) * nMass;
                const double jbnOld = jBiasAcc;
                jBiasAcc = fmax2(jbnOld + jbn, 0.0);
                const double jn = -(bounce + vrn) * nMass;
                const double jnOld = jnAcc;
                jnAcc = fmax2(jnOld + jn, 0.0);
                const double jbx = nx * (jBiasAcc - jbnOld), jby = ny * (jBiasAcc - jbnOld);
                const double dj = jnAcc - jnOld;
                const double jx = nx * dj - ny * 0.0, jy = nx * 0.0 + ny * dj;   // cpvrotate, jt = 0
                L.vb[2 * ca] = vba.x + (-jbx) * m_inv; L.vb[2 * ca + 1] = vba.y + (-jby) * m_inv;
                L.vel[2 * ca] = va.x + (-jx) * m_inv; L.vel[2 * ca + 1] = va.y + (-jy) * m_inv;
                if (cb >= 0) {
                    L.vb[2 * cb] = vbb2.x + jbx * m_inv; L.vb[2 * cb + 1] = vbb2.y + jby * m_inv;
                    L.vel[2 * cb] = vb2.x + jx * m_inv; L.vel[2 * cb + 1] = vb2.y + jy * m_inv;
                }
            }
        }
    }
    if (own) {
        if (cidx & (1 << 20)) L.pjn[cidx & 0xFFFFF] = jnAcc;
        else L.wjn[cidx] = jnAcc;
    }
    wave_sync();
}
