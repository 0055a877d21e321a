// cat_sim_host.h -- part of the env core's single translation unit (included by cat_sim.hip, in this order; not a stand-alone header):
// host side: spatial-hash tables, kernel selection, LDS sizing, cat_create and every other C-ABI entry.
// ====================================================================== host side ============
// ---------------------------------------------------------------------- spatial-hash grids ----
struct GridHost {
    std::vector<GridDesc> desc;
    std::vector<unsigned long long> rows;   // per (cell, ray): count | first ids (row_words 8-byte words), see finalize_rows
    std::vector<int> rows_of;               // rows per map
    std::vector<unsigned long long> crows;  // per cell: count | first 7 contact candidates
    int max_row = 0, row_words = 1;
    int id_bits = 0;                        // > 0: four-byte rows (finalize_rows)
    std::vector<int> off, coff;
    std::vector<unsigned char> ent, cent;
};

// One map.  bb: [S][4] wall bbs (already inflated by the wall radius); the walls' plane records follow them (bb + 4 S:
// n.x n.y v0.x v0.y dot(v0, n) ...; hull_first / hull_count index them).  Three rules decide whether a wall is listed for
// (cell, ray k); all three keep [CP cpSpaceSegmentQueryFirst]'s result for every origin in the cell exactly as the full wall list gives it.
//  1. The visit (always): the wall's bb, grown by 1e-6 (gate off: by the ray radius), meets the region swept by the thin segment origin -> origin + d_k over
//     all origins of the cell: conv(cell, cell + d_k), a hexagon whose edge normals are x, y and perp(d_k) -- a separating-axis
//     test on those three axes is exact.
//  2. The hit (CAT_GRID_HULLS=0 turns it off): some ray of the cell can come within rsum = wall radius + ray radius of the HULL
//     (separating axes: perp(d_k) with the hull's own vertices, every face normal).  A wall whose shape query cannot return a hit
//     leaves no trace whether it is visited or not (agh-map: 21 % fewer entries -- triangles, slanted and merged blocks).
//  3. Occlusion (CAT_GRID_OCCLUSION=0 turns it off): if every ray of the cell is certain to cross the hull of some listed wall W
//     no later than T (in units of the ray), then after W's turn the best alpha is <= T whatever came before; a wall whose bb,
//     grown by the ray radius, is entered later than T by every ray of the cell (so t_bb > T and alpha > T) can then be taken out
//     of the sequence: while the best alpha is above T such a wall can only replace it by another value above T, and every wall
//     that stays is visited or not regardless of such values PROVIDED its own t_bb never exceeds T -- so T is first raised past
//     the latest thin-bb entry of every remaining wall whose range of entries straddles it.  (Agents come after the walls in
//     the visiting order: by then the best alpha is the same with and without the walls taken out.)
struct RotPoly {   // a convex polygon seen from one ray direction: per vertex its depth along the ray in units of the ray, and its offset across it
    int n;
    double al[CAT_MAX_HULL_EDGES + 1], si[CAT_MAX_HULL_EDGES + 1], smin, smax;
    void close() { smin = 1e300; smax = -1e300; for (int i = 0; i < n; i++) { smin = std::fmin(smin, si[i]); smax = std::fmax(smax, si[i]); } }
    // smallest depth among the polygon's points at offset s (s within [smin, smax])
    double entry(double s) const
    {
        double best = 1e300;
        for (int i = 0; i < n; i++) {
            const int j = i + 1 < n ? i + 1 : 0;
            const double s0 = si[i], s1 = si[j];
            if ((s0 <= s && s <= s1) || (s1 <= s && s <= s0))
                best = std::fmin(best, s0 == s1 ? std::fmin(al[i], al[j]) : al[i] + (al[j] - al[i]) * ((s - s0) / (s1 - s0)));
        }
        return best;
    }
    // bounds of entry() over the offsets [a, b] clipped to the polygon: false if they do not meet
    bool entry_range(double a, double b, double &emin, double &emax) const
    {
        const double lo = std::fmax(a, smin), hi = std::fmin(b, smax);
        if (lo > hi) return false;
        const double e0 = entry(lo), e1 = entry(hi);
        emax = std::fmax(e0, e1);            // entry() is convex in s: its maximum over an interval is at an end
        emin = std::fmin(e0, e1);            // its minimum is at an end or at a vertex in between (any vertex there bounds it from below)
        for (int i = 0; i < n; i++) if (lo <= si[i] && si[i] <= hi) emin = std::fmin(emin, al[i]);
        return true;
    }
};

struct GridRowOut { std::vector<int> off, coff; std::vector<unsigned char> ent, cent; int max_row = 0; };

static void build_grids(const double *bb, int S, int R, const double *rdx, const double *rdy, double reach,
                        bool gate, double m_contact, double cell, GridHost &g, const int *hull_first, const int *hull_count,
                        double rsum, double ray_radius)
{
    const double m_ray = gate ? 1e-6 : ray_radius + 1e-6;   // gate off: every wall the fat ray can touch counts as visited
    const double *planes = bb + 4 * (size_t)S;
    bool by_hull = hull_first != nullptr, occlusion = hull_first != nullptr;
    if (const char *e = getenv("CAT_GRID_HULLS")) { if (atoi(e) == 0) by_hull = false; }
    if (const char *e = getenv("CAT_GRID_OCCLUSION")) { if (atoi(e) == 0) occlusion = false; }
    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
    for (int s = 0; s < S; s++) {
        lo[0] = std::fmin(lo[0], bb[4 * s]); lo[1] = std::fmin(lo[1], bb[4 * s + 1]);
        hi[0] = std::fmax(hi[0], bb[4 * s + 2]); hi[1] = std::fmax(hi[1], bb[4 * s + 3]);
    }
    GridDesc d{};
    d.x0 = std::floor(lo[0] - reach - cell); d.y0 = std::floor(lo[1] - reach - cell);
    d.nx = (int)std::ceil((hi[0] + reach + cell - d.x0) / cell); d.ny = (int)std::ceil((hi[1] + reach + cell - d.y0) / cell);
    d.inv_cell = 1.0 / cell;
    d.off_base = (int)g.off.size(); d.ent_base = (int)g.ent.size();
    d.coff_base = (int)g.coff.size(); d.cent_base = (int)g.cent.size();
    d.row_base = 0;   // set by finalize_rows
    const double eps = 1e-6;   // cell membership is decided in floating point on the device
    const double mt = 1e-7;    // occlusion: slack on every bound, in units of the ray (4e-5 px of a 400-px ray)

    // one row of cells: its part of the CSR arrays, offsets relative to the row
    auto do_row = [&](int cy, GridRowOut &o) {
        std::vector<int> near, list;     // walls within reach of the cell (prefilter); the walls listed for (cell, ray)
        std::vector<double> f_lo, b_hi, cuts;
        std::vector<RotPoly> hulls;
        for (int cx = 0; cx < d.nx; cx++) {
            const double X0 = d.x0 + cx * cell - eps, X1 = d.x0 + (cx + 1) * cell + eps;
            const double Y0 = d.y0 + cy * cell - eps, Y1 = d.y0 + (cy + 1) * cell + eps;
            near.clear();
            o.coff.push_back((int)o.cent.size());
            for (int s = 0; s < S; s++) {
                const double l = bb[4 * s], b = bb[4 * s + 1], r = bb[4 * s + 2], t = bb[4 * s + 3];
                if (l - m_contact <= X1 && X0 <= r + m_contact && b - m_contact <= Y1 && Y0 <= t + m_contact)
                    o.cent.push_back((unsigned char)s);
                if (l - reach <= X1 && X0 <= r + reach && b - reach <= Y1 && Y0 <= t + reach) near.push_back(s);
            }
            // the walls listed for ray k and the origins of the rectangle [X0, X1] x [Y0, Y1] (rules 1 - 3), ascending, into `list`
            auto list_for = [&](int k, double X0, double X1, double Y0, double Y1) {
                const double dx = rdx[k], dy = rdy[k];
                const double hx0 = X0 + std::fmin(0.0, dx) - eps, hx1 = X1 + std::fmax(0.0, dx) + eps;
                const double hy0 = Y0 + std::fmin(0.0, dy) - eps, hy1 = Y1 + std::fmax(0.0, dy) + eps;
                // projections of the cell on n = (-dy, dx)
                const double c0 = -dy * X0 + dx * Y0, c1 = -dy * X1 + dx * Y0, c2 = -dy * X0 + dx * Y1, c3 = -dy * X1 + dx * Y1;
                const double nscale = std::fabs(dx) + std::fabs(dy);
                const double pmin = std::fmin(std::fmin(c0, c1), std::fmin(c2, c3)) - eps * nscale;
                const double pmax = std::fmax(std::fmax(c0, c1), std::fmax(c2, c3)) + eps * nscale;
                list.clear();
                for (int s : near) {
                    const double l = bb[4 * s] - m_ray, b = bb[4 * s + 1] - m_ray, r = bb[4 * s + 2] + m_ray, t = bb[4 * s + 3] + m_ray;
                    if (!(l <= hx1 && hx0 <= r && b <= hy1 && hy0 <= t)) continue;
                    const double q0 = -dy * l + dx * b, q1 = -dy * r + dx * b, q2 = -dy * l + dx * t, q3 = -dy * r + dx * t;
                    const double qmin = std::fmin(std::fmin(q0, q1), std::fmin(q2, q3)), qmax = std::fmax(std::fmax(q0, q1), std::fmax(q2, q3));
                    if (!(qmin <= pmax && pmin <= qmax)) continue;
                    if (by_hull) {
                        const double *pl = planes + 8 * (size_t)hull_first[s];
                        const int ne = hull_count[s];
                        const double grow = (rsum + eps) * std::sqrt(dx * dx + dy * dy);
                        double vmin = 1e300, vmax = -1e300;
                        for (int e = 0; e < ne; e++) {
                            const double v = -dy * pl[8 * e + 2] + dx * pl[8 * e + 3];
                            vmin = std::fmin(vmin, v); vmax = std::fmax(vmax, v);
                        }
                        bool apart = vmin - grow > pmax || vmax + grow < pmin;
                        for (int e = 0; e < ne && !apart; e++) {
                            const double nx = pl[8 * e], ny = pl[8 * e + 1];
                            const double lowest = std::fmin(nx * X0, nx * X1) + std::fmin(ny * Y0, ny * Y1) + std::fmin(0.0, nx * dx + ny * dy);
                            apart = lowest > pl[8 * e + 4] + rsum + eps;
                        }
                        if (apart) continue;
                    }
                    list.push_back(s);
                }
                if (occlusion && list.size() > 1) {
                    const double dd = dx * dx + dy * dy;
                    const double a0 = (X0 * dx + Y0 * dy) / dd, a1 = (X1 * dx + Y0 * dy) / dd, a2 = (X0 * dx + Y1 * dy) / dd, a3 = (X1 * dx + Y1 * dy) / dd;
                    const double amin = std::fmin(std::fmin(a0, a1), std::fmin(a2, a3)), amax = std::fmax(std::fmax(a0, a1), std::fmax(a2, a3));
                    // a margin across the ray, in the units of pmin / pmax: a ray passing that far outside a hull's end vertex still meets the
                    // ROUNDED shape -- which needs a wall radius well above the margin.  With wall_radius ~ 0 the bb coincides with the hull,
                    // a thin ray in that sliver misses the bb and the wall is never visited: no slack then.
                    const double ms = (rsum - ray_radius) > 1e3 * eps ? eps * nscale : 0.0;
                    const size_t n = list.size();
                    f_lo.assign(n, 1e300); b_hi.assign(n, -1e300);
                    hulls.resize(n);
                    cuts.clear();
                    cuts.push_back(pmin); cuts.push_back(pmax);
                    for (size_t q = 0; q < n; q++) {
                        const int s = list[q];
                        RotPoly &H = hulls[q];
                        RotPoly B, F;
                        const double *pl = planes + 8 * (size_t)hull_first[s];
                        H.n = hull_count[s];
                        for (int e = 0; e < H.n; e++) {
                            const double vx = pl[8 * e + 2], vy = pl[8 * e + 3];
                            H.al[e] = (vx * dx + vy * dy) / dd; H.si[e] = -dy * vx + dx * vy;
                            if (pmin < H.si[e] && H.si[e] < pmax) cuts.push_back(H.si[e]);
                        }
                        H.close();
                        const double l = bb[4 * s], b = bb[4 * s + 1], r = bb[4 * s + 2], t = bb[4 * s + 3];
                        const double cxs[4] = {l, r, r, l}, cys[4] = {b, b, t, t};
                        B.n = F.n = 4;
                        for (int e = 0; e < 4; e++) {
                            B.al[e] = (cxs[e] * dx + cys[e] * dy) / dd; B.si[e] = -dy * cxs[e] + dx * cys[e];
                            const double fx = cxs[e] + ((e == 1 || e == 2) ? ray_radius : -ray_radius), fy = cys[e] + (e >= 2 ? ray_radius : -ray_radius);
                            F.al[e] = (fx * dx + fy * dy) / dd; F.si[e] = -dy * fx + dx * fy;
                        }
                        B.close(); F.close();
                        double emin, emax;
                        if (F.entry_range(pmin, pmax, emin, emax)) f_lo[q] = std::fmax(0.0, emin - amax) - mt;
                        if (!gate) b_hi[q] = 0.0;   // every listed wall counts as entered at once
                        else if (B.entry_range(pmin, pmax, emin, emax)) b_hi[q] = std::fmax(0.0, emax - amin) + mt;
                    }
                    // T: by when every ray of the cell has certainly crossed the hull of SOME listed wall.  Between two neighbouring
                    // cuts (the cell's span across the ray, cut at the hull vertices inside it) every hull's entry depth is linear; a
                    // hull counts there if it spans the piece (a ray passing a hair -- ms -- outside its end vertex still meets the
                    // rounded shape before that vertex's depth) and every origin of the cell lies before it; the piece's bound is the
                    // smallest of the hulls' larger end values (max-min <= min-max), T the largest bound of any piece.
                    std::sort(cuts.begin(), cuts.end());
                    double T = -1e300;
                    for (size_t c = 0; c + 1 < cuts.size() && T < 1e299; c++) {
                        const double lft = cuts[c], rgt = cuts[c + 1];
                        if (!(lft < rgt)) continue;
                        double best = 1e300;
                        for (size_t q = 0; q < n; q++) {
                            const RotPoly &H = hulls[q];
                            if (H.n < 3 || !(H.smin - ms <= lft && rgt <= H.smax + ms)) continue;
                            const double e0 = H.entry(std::fmin(std::fmax(lft, H.smin), H.smax)), e1 = H.entry(std::fmin(std::fmax(rgt, H.smin), H.smax));
                            if (!(amax <= std::fmin(e0, e1) - mt)) continue;
                            best = std::fmin(best, std::fmax(e0, e1));
                        }
                        T = std::fmax(T, best);
                    }
                    T = (T > -1e299 && T < 1e299 && T - amin <= 1.0 - 2.0 * mt) ? T - amin + mt : 1e300;
                    if (T < 1e299) {
                        for (bool again = true; again;) {
                            again = false;
                            for (size_t q = 0; q < n; q++)
                                if (f_lo[q] <= T && T < b_hi[q]) { T = b_hi[q]; again = true; }
                        }
                        size_t w = 0;
                        for (size_t q = 0; q < n; q++) if (f_lo[q] <= T) list[w++] = list[q];
                        list.resize(w);
                    }
                }
            };
            for (int k = 0; k < R; k++) {
                o.off.push_back((int)o.ent.size());
                list_for(k, X0, X1, Y0, Y1);
                for (int s : list) o.ent.push_back((unsigned char)s);
                if ((int)list.size() > o.max_row) o.max_row = (int)list.size();
            }
        }
    };
    std::vector<GridRowOut> rows((size_t)d.ny);
    {
        unsigned nt = std::thread::hardware_concurrency();
        {   // the CPUs this process may run on (a container's share), not the machine's
            cpu_set_t set;
            if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0 && (unsigned)CPU_COUNT(&set) < nt) nt = (unsigned)CPU_COUNT(&set);
        }
        if (nt > 16) nt = 16;
        if (nt < 1) nt = 1;
        if ((int)nt > d.ny) nt = (unsigned)d.ny;
        std::atomic<int> next{0};
        auto worker = [&]() { for (int cy = next.fetch_add(1); cy < d.ny; cy = next.fetch_add(1)) do_row(cy, rows[(size_t)cy]); };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; t++) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
    }
    for (int cy = 0; cy < d.ny; cy++) {
        const GridRowOut &o = rows[(size_t)cy];
        const int e0 = (int)g.ent.size() - d.ent_base, c0 = (int)g.cent.size() - d.cent_base;
        for (int v : o.off) g.off.push_back(e0 + v);
        for (int v : o.coff) g.coff.push_back(c0 + v);
        g.ent.insert(g.ent.end(), o.ent.begin(), o.ent.end());
        g.cent.insert(g.cent.end(), o.cent.begin(), o.cent.end());
        if (o.max_row > g.max_row) g.max_row = o.max_row;
    }
    rows.clear();
    g.off.push_back((int)g.ent.size() - d.ent_base);
    g.coff.push_back((int)g.cent.size() - d.cent_base);
    d.crow_base = (int)g.crows.size();
    for (int c = 0; c < d.nx * d.ny; c++) {
        const int o0 = g.coff[d.coff_base + c] + d.cent_base, n = g.coff[d.coff_base + c + 1] + d.cent_base - o0;
        unsigned long long w = (unsigned long long)(n > 255 ? 255 : n);
        for (int q = 0; q < n && q < 7; q++) w |= (unsigned long long)g.cent[o0 + q] << (8 * (q + 1));
        g.crows.push_back(w);
    }
    while (g.ent.size() & 3) g.ent.push_back(0);
    while (g.cent.size() & 3) g.cent.push_back(0);
    g.desc.push_back(d);
    g.rows_of.push_back(d.nx * d.ny * R);
}

// The most wall bounding boxes the bb of ONE agent circle can overlap at once, anywhere on the map: an upper bound on the wall
// arbiters an agent can hold in one step ([CP cpSpaceCollideShapes] makes one only on a real contact), to be held against the
// CAT_WALL_CACHE slots of the state record at cat_create instead of being discovered at run time (CAT_DEVERR_CONTACT_DROPPED).
// The circle's bb overlaps a wall's iff its centre lies in the wall's bb grown by the radius (closed rectangles): the deepest point
// of such an arrangement is the left edge of one rectangle and the bottom edge of one.
static int max_wall_bb_depth(const double *bb, int S, double rc)
{
    int best = 0;
    for (int a = 0; a < S; a++) {
        const double x = bb[4 * a] - rc;
        for (int b = 0; b < S; b++) {
            const double y = bb[4 * b + 1] - rc;
            int n = 0;
            for (int s = 0; s < S; s++)
                n += (bb[4 * s] - rc <= x && x <= bb[4 * s + 2] + rc && bb[4 * s + 1] - rc <= y && y <= bb[4 * s + 3] + rc) ? 1 : 0;
            best = n > best ? n : best;
        }
    }
    return best;
}

extern "C" int cat_map_wall_bb_depth_host(const void *blob, size_t size, double agent_radius)
{
    if (!blob || size < 64) return CAT_ERR_BAD_ARG;
    int32_t h[16];
    memcpy(h, blob, 64);
    const int S = h[2];
    if ((unsigned)h[0] != kBlobMagic || S < 1 || S > CAT_MAX_SHAPES || size < 64 + (2 + 4 * (size_t)S) * 8) return CAT_ERR_BAD_MAP;
    std::vector<double> bb(4 * (size_t)S);
    memcpy(bb.data(), static_cast<const unsigned char *>(blob) + 64 + 16, bb.size() * 8);
    return max_wall_bb_depth(bb.data(), S, agent_radius);
}

// Packed rows, one per (cell, ray): byte 0 = count (saturating at 255), then the first 8*row_words - 1
// candidate ids; row_words (1, 2 or 4 eight-byte words) is the smallest that holds the longest list of
// any map of the sim, lists beyond 31 ids continue in the CSR arrays (slow path on the device).
// id_bits > 0 (fan_group sims: S <= 2^id_bits - 1 and every list has at most 32 / id_bits walls): FOUR-byte rows, two per word of
// `rows` -- field q (id_bits bits) = the list's q-th wall id + 1, zero beyond the list, so the count is the highest non-zero field's
// index + 1; d.row_base counts rows either way.
static void finalize_rows(GridHost &g, int id_bits = 0, bool wide = false)
{
    g.row_words = g.max_row <= 7 ? 1 : (g.max_row <= 15 ? 2 : 4);
    g.id_bits = id_bits;
    const int cap = 8 * g.row_words - 1;
    g.rows.clear();
    if (id_bits > 0 && wide) {   // the same fields in ONE eight-byte word per row (fan_chunk sims whose lists fit 64 / id_bits walls)
        g.row_words = 1;
        for (size_t m = 0; m < g.desc.size(); m++) {
            GridDesc &d = g.desc[m];
            d.row_base = (int)g.rows.size();
            for (int r = 0; r < g.rows_of[m]; r++) {
                const int o0 = g.off[d.off_base + r] + d.ent_base, n = g.off[d.off_base + r + 1] + d.ent_base - o0;
                unsigned long long w = 0ull;
                for (int q = 0; q < n; q++) w |= ((unsigned long long)g.ent[o0 + q] + 1ull) << (id_bits * q);
                g.rows.push_back(w);
            }
        }
        return;
    }
    if (id_bits > 0) {
        std::vector<unsigned> r32;
        for (size_t m = 0; m < g.desc.size(); m++) {
            GridDesc &d = g.desc[m];
            d.row_base = (int)r32.size();
            for (int r = 0; r < g.rows_of[m]; r++) {
                const int o0 = g.off[d.off_base + r] + d.ent_base, n = g.off[d.off_base + r + 1] + d.ent_base - o0;
                unsigned w = 0u;                                // n <= max_row <= 32 / id_bits
                for (int q = 0; q < n; q++) w |= ((unsigned)g.ent[o0 + q] + 1u) << (id_bits * q);
                r32.push_back(w);
            }
        }
        if (r32.size() & 1) r32.push_back(0u);
        g.rows.resize(r32.size() / 2);
        memcpy(g.rows.data(), r32.data(), r32.size() * 4);
        return;
    }
    for (size_t m = 0; m < g.desc.size(); m++) {
        GridDesc &d = g.desc[m];
        d.row_base = (int)(g.rows.size() / g.row_words);
        for (int r = 0; r < g.rows_of[m]; r++) {
            const int o0 = g.off[d.off_base + r] + d.ent_base, n = g.off[d.off_base + r + 1] + d.ent_base - o0;
            unsigned long long w[4] = {(unsigned long long)(n > 255 ? 255 : n), 0ull, 0ull, 0ull};
            for (int q = 0; q < n && q < cap; q++) {
                const int byte = q + 1;
                w[byte >> 3] |= (unsigned long long)g.ent[o0 + q] << (8 * (byte & 7));
            }
            for (int q = 0; q < g.row_words; q++) g.rows.push_back(w[q]);
        }
    }
}

// Kernel instantiations: fixed dimensions for the rosters / ray counts of the BASELINE configurations and of the
// reference's defaults, the generic one for everything else (CAT_GENERIC_KERNEL=1 forces it: A/B tests).
using KernelFn = void (*)(const Params *, const LaunchArgs, const Prologue);
template <class D> static void kernels_of(int fan, KernelFn &reset, KernelFn &rollout, KernelFn &step)
{
    if (fan == 1) { reset = reset_kernel<WithFan<D, 1>>; rollout = rollout_kernel<WithFan<D, 1>>; step = step_kernel<WithFan<D, 1>>; }
    else { reset = reset_kernel<WithFan<D, 0>>; rollout = rollout_kernel<WithFan<D, 0>>; step = step_kernel<WithFan<D, 0>>; }
}
// fan: 0 = chunk by chunk, 1 = agent groups with compacted rays (cat_create decides from the maps; CAT_FAN=chunks forces 0)
// pool_roll / pool_step: the sim's rays fit a workgroup ring (cat_create) and the resident / the one-tick entry runs the pooled fan; exact: the ring's capacity is not a
// power of two.  Instantiated for 2v1 / 64 and 1v1 / 90 (power-of-two rings), 2v1 / 90 (exact) and generically in both forms.
template <class D, bool kExact> static void pooled_of(bool pool_roll, bool pool_step, KernelFn &rollout, KernelFn &step)
{
    if (pool_roll) rollout = rollout_kernel_pooled<WithFan<D, 1>, kExact>;
    if (pool_step) step = step_kernel_pooled<WithFan<D, 1>, kExact>;
}
static const char *select_kernels(int A, int R, int n_cops, int fan, bool pool_roll, bool pool_step, bool exact, KernelFn &reset, KernelFn &rollout, KernelFn &step)
{
    const char *e = getenv("CAT_GENERIC_KERNEL");
    const bool generic = e && atoi(e) != 0;
    if ((pool_roll || pool_step) && fan == 1) {
        if (!generic && A == 3 && n_cops == 2 && R == 64 && !exact) {
            kernels_of<FixDims<3, 64, 2>>(fan, reset, rollout, step);
            pooled_of<FixDims<3, 64, 2>, false>(pool_roll, pool_step, rollout, step);
            return "3 agents (2 cops), 64 rays, pooled fan";
        }
#ifndef CAT_QUICK_BUILD
        if (!generic && A == 3 && n_cops == 2 && R == 90 && exact) {
            kernels_of<FixDims<3, 90, 2>>(fan, reset, rollout, step);
            pooled_of<FixDims<3, 90, 2>, true>(pool_roll, pool_step, rollout, step);
            return "3 agents (2 cops), 90 rays, pooled fan";
        }
        if (!generic && A == 5 && n_cops == 3 && R == 64 && exact) {    // BASELINE configs[3]: the ring fits since the contact arrays are sized by the map
            kernels_of<FixDims<5, 64, 3>>(fan, reset, rollout, step);
            pooled_of<FixDims<5, 64, 3>, true>(pool_roll, pool_step, rollout, step);
            return "5 agents (3 cops), 64 rays, pooled fan";
        }
        if (!generic && A == 2 && n_cops == 1 && R == 90 && !exact) {   // the reference's own defaults: 1v1 (simple_env.py), 90 rays (entity.py:86)
            kernels_of<FixDims<2, 90, 1>>(fan, reset, rollout, step);
            pooled_of<FixDims<2, 90, 1>, false>(pool_roll, pool_step, rollout, step);
            return "2 agents (1 cop), 90 rays, pooled fan";
        }
#endif
        kernels_of<DynDims>(fan, reset, rollout, step);
        if (exact) pooled_of<DynDims, true>(pool_roll, pool_step, rollout, step);
        else pooled_of<DynDims, false>(pool_roll, pool_step, rollout, step);
        return "generic, pooled fan";
    }
    if (!generic && A == 3 && n_cops == 2 && R == 64) { kernels_of<FixDims<3, 64, 2>>(fan, reset, rollout, step); return "3 agents (2 cops), 64 rays"; }
#ifndef CAT_QUICK_BUILD   // diagnostic builds (tools/build_variant.sh -DCAT_QUICK_BUILD): the headline instantiation + the generic one only
    if (!generic && A == 3 && n_cops == 2 && R == 90) { kernels_of<FixDims<3, 90, 2>>(fan, reset, rollout, step); return "3 agents (2 cops), 90 rays"; }
    if (!generic && A == 2 && n_cops == 1 && R == 90) { kernels_of<FixDims<2, 90, 1>>(fan, reset, rollout, step); return "2 agents (1 cop), 90 rays"; }
    if (!generic && A == 5 && n_cops == 3 && R == 64) { kernels_of<FixDims<5, 64, 3>>(fan, reset, rollout, step); return "5 agents (3 cops), 64 rays"; }
#endif
    kernels_of<DynDims>(fan, reset, rollout, step);
    return "generic";
}

// LDS carve sizes (must match carve())
struct LdsSizes {
    int map, env, uni;
    size_t total(int wpb) const { return (size_t)map + (size_t)ctrl_bytes(wpb) + (size_t)wpb * ((size_t)env + (size_t)uni); }
};
static LdsSizes lds_sizes(int A, int R, int maxS, int maxP, int maxPP, bool group_fan, int maxc, int grp_rays = 4 * 64)
{
    auto up = [](int x, int a) { return (x + a - 1) / a * a; };
    const int NP = A * (A - 1) / 2, NPs = NP > 0 ? NP : 1;
    LdsSizes z;
    const int rest = 8 * maxP + (kPairF / 2) * maxPP;
    z.map = up((kBB * maxS + rest) * 8 + 2 * maxS * 4, 16) + 16 * R;
    const int phys_bytes = kConD * 8 * maxc;   // contact records (physics_env)
    const int cpa = (R + 63) / 64;
    const int fan_bytes = kFanBytes + (group_fan ? grp_rays * (4 + 1 + 1) : 0);   // the rays of an agent group (four chunks unless the ray pool needs the LDS): arow, alist, adyn
    (void)cpa;
    z.uni = up(phys_bytes > fan_bytes ? phys_bytes : fan_bytes, 16);
    const int rec_bytes = 96 * A + 16 + ((A * kK + NPs) * 8 + (2 * A * kK + NPs) * 4 + 15) / 16 * 16;
    int eb = rec_bytes + 8 * A * 8;                                // record, spawn/snapshot
    eb += (3 * A + 2 * A * A + A + A + 4) * 4;                     // acell, anear, dk0, dcnt, adn, dmin, flags
    eb = up(eb, 16) + up(A * R * 2, 16) + up(A * R, 16) + up(2 * R * 2, 16) + up(2 * R, 16);   // output staging
    z.env = up(eb, 16);
    // The pooled rounds address env areas PER LANE (a round's rays come from several slots: origin, cached circles, "inside" walls, output staging of the
    // lane's slot).  An area of a multiple of 256 bytes -- 2v1 at 64 rays: exactly 2048 -- puts the same field of every slot on the same LDS banks; with
    // 16 A bytes more (the step between two agents' 16-byte fields) neighbouring slots' fields of any two agents fall on different banks.  CAT_ENV_PAD=0: off.
    {
        const char *e = getenv("CAT_ENV_PAD");
        const int want = (16 * A) % 256;
        if (!(e && atoi(e) == 0) && want != 0 && z.env % 256 != want) z.env += (want - z.env % 256 + 256) % 256;
    }
    return z;
}

// The env slots of a sim whose maps take one form of the ray fan: their candidate tables, parameter block, LDS carve, work list and kernels -- one
// dispatch per entry.  A sim has one part, or two when its maps want both forms (cat_create).
struct Part {
    GridHost grid;                 // grid.desc[k] belongs to map map_ids[k] of the sim
    std::vector<int> map_ids;
    Params p;
    Params *dev_p = nullptr;
    Prologue pro;
    int n_blocks = 0, wpb = 0, n_envs = 0;
    size_t lds_bytes = 0;
    bool pool_step = false;   // the one-tick entry runs the pooled kernel (the resident one does whenever the ring exists: p.pool_mask)
    KernelFn reset_fn = nullptr, rollout_fn = nullptr, step_fn = nullptr;   // the instantiations matching (agents, rays, cops): cat_reset*, cat_rollout_fused, cat_step*
    const char *kernel_variant = "";
};

struct cat_sim {
    std::vector<Part> parts;
    int device;
    hipStream_t side = nullptr;                       // two parts: the second one's stream ...
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;  // ... forked from and joined to the caller's
    hipEvent_t t_start = nullptr, t_stop = nullptr;   // cat_arm_kernel_timing
    std::vector<MapDesc> maps;
    std::vector<void *> allocs;
    std::string one_tick_name, rollout_name;
    char err[256];
};

// the tables of one map (a GridHost of its own, as build_grids leaves it) appended to a part's
static void append_grid(GridHost &g, const GridHost &m)
{
    GridDesc d = m.desc[0];
    d.off_base = (int)g.off.size(); d.ent_base = (int)g.ent.size();
    d.coff_base = (int)g.coff.size(); d.cent_base = (int)g.cent.size();
    d.crow_base = (int)g.crows.size();
    g.off.insert(g.off.end(), m.off.begin(), m.off.end());
    g.ent.insert(g.ent.end(), m.ent.begin(), m.ent.end());
    g.coff.insert(g.coff.end(), m.coff.begin(), m.coff.end());
    g.cent.insert(g.cent.end(), m.cent.begin(), m.cent.end());
    g.crows.insert(g.crows.end(), m.crows.begin(), m.crows.end());
    g.desc.push_back(d);
    g.rows_of.push_back(m.rows_of[0]);
    if (m.max_row > g.max_row) g.max_row = m.max_row;
}

#define HIP_TRY(sim, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            snprintf((sim)->err, sizeof((sim)->err), "%s failed: %s", #expr, hipGetErrorString(e_)); \
            return CAT_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

template <typename T>
static int dev_alloc(cat_sim *s, T **ptr, size_t count, const void *init)
{
    void *d = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    HIP_TRY(s, hipMalloc(&d, bytes));
    s->allocs.push_back(d);
    if (init) HIP_TRY(s, hipMemcpy(d, init, count * sizeof(T), hipMemcpyHostToDevice));
    else HIP_TRY(s, hipMemset(d, 0, bytes));
    *ptr = static_cast<T *>(d);
    return CAT_OK;
}

extern "C" int cat_abi_version(void) { return CAT_ABI_VERSION; }
// (a sim of two parts: both names, "+"-joined, the group-form part first)
extern "C" const char *cat_one_tick_kernel(const cat_sim *sim) { return sim ? sim->one_tick_name.c_str() : ""; }
extern "C" const char *cat_rollout_kernel(const cat_sim *sim) { return sim ? sim->rollout_name.c_str() : ""; }
extern "C" const char *cat_last_error(const cat_sim *sim) { return sim ? sim->err : g_create_err; }
extern "C" int cat_chunks_per_unit(const cat_sim *sim, int resident)
{
    if (!sim) return CAT_ERR_BAD_ARG;
    const Part &pt = sim->parts.back();   // (two parts: the chunk-form part is the second)
    int best = 1;                          // (several maps: the largest figure)
    for (const GridDesc &d : pt.grid.desc) best = std::max(best, resident ? d.span : d.span_tick);
    return best;
}
extern "C" int cat_num_agents(const cat_sim *sim) { return sim ? sim->parts[0].p.A : CAT_ERR_BAD_ARG; }
extern "C" int cat_num_shapes(const cat_sim *sim, int m)
{
    if (!sim || m < 0 || m >= (int)sim->maps.size()) return CAT_ERR_BAD_ARG;
    return sim->maps[m].S;
}

extern "C" int cat_create(const cat_config *cfg, const cat_tables *tab, const void *const *blobs,
                          const size_t *sizes, int n_maps, const int32_t *slot_map_ids, int device,
                          cat_sim **out)
{
    if (!cfg || !tab || !blobs || !sizes || !out || n_maps < 1) {
        snprintf(g_create_err, sizeof g_create_err, "null argument");
        return CAT_ERR_BAD_ARG;
    }
    const int A = cfg->n_cops + cfg->n_thieves;
    if (A < 1 || A > CAT_MAX_AGENTS || cfg->n_cops < 0 || cfg->n_thieves < 0 || cfg->n_rays < 1 ||
        cfg->n_rays > CAT_MAX_RAYS || cfg->n_envs < 1 || A * kK > 64) {
        snprintf(g_create_err, sizeof g_create_err, "bad config: agents=%d rays=%d envs=%d", A, cfg->n_rays, cfg->n_envs);
        return CAT_ERR_BAD_CONFIG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) {
        snprintf(g_create_err, sizeof g_create_err, "no usable HIP device (count=%d, requested %d): this library has no CPU path", ndev, device);
        return CAT_ERR_NO_DEVICE;
    }
    // ---- parse blobs into one packed geometry buffer
    std::vector<MapDesc> descs((size_t)n_maps);
    std::vector<int> wall_depth((size_t)n_maps, 0);   // per map: the most wall bbs one agent's bb can overlap at once (max_wall_bb_depth)
    std::vector<double> geo_f;
    std::vector<int> geo_i;
    for (int m = 0; m < n_maps; m++) {
        const unsigned char *b = static_cast<const unsigned char *>(blobs[m]);
        int32_t h[16];
        if (sizes[m] < 64) { snprintf(g_create_err, sizeof g_create_err, "map blob %d too small", m); return CAT_ERR_BAD_MAP; }
        memcpy(h, b, 64);
        MapDesc d{};
        d.S = h[2]; d.P = h[3]; d.A = h[4]; d.n_regions = h[7];
        const size_t nf = 2 + 4 * (size_t)d.S + 8 * (size_t)d.P + 2 * (size_t)d.A + 4 * (size_t)d.n_regions;
        const size_t ni = 2 * (size_t)d.S + (size_t)d.A + 1;
        if ((unsigned)h[0] != kBlobMagic || h[1] != 1 || sizes[m] != 64 + nf * 8 + ni * 4 || d.A != A ||
            h[5] != cfg->n_cops || d.S < 1 || d.S > CAT_MAX_SHAPES) {
            snprintf(g_create_err, sizeof g_create_err, "map blob %d invalid, roster mismatch or more than %d shapes", m, CAT_MAX_SHAPES);
            return CAT_ERR_BAD_MAP;
        }
        std::vector<double> f(nf);
        std::vector<int> iv(ni);
        memcpy(f.data(), b + 64, nf * 8);
        memcpy(iv.data(), b + 64 + nf * 8, ni * 4);
        for (int sidx = 0; sidx < d.S; sidx++)   // feature codes (edge, count + corner) must stay below kFeatNear; circle_poly_contact: lane = edge
            if (iv[d.S + sidx] < 1 || iv[d.S + sidx] > CAT_MAX_HULL_EDGES) {
                snprintf(g_create_err, sizeof g_create_err, "map blob %d: wall %d has %d hull edges (1..%d supported)", m, sidx, iv[d.S + sidx], CAT_MAX_HULL_EDGES);
                return CAT_ERR_BAD_MAP;
            }
        {
            // bb overlap is a loose bound on simultaneous CONTACTS (slanted or star-shaped walls overlap boxes without touching):
            // CAT_ALLOW_DEEP_WALL_OVERLAP=1 accepts such a map, with CAT_DEVERR_CONTACT_DROPPED as the run-time check
            const int depth = max_wall_bb_depth(f.data() + 2, d.S, cfg->agent_radius);
            wall_depth[(size_t)m] = depth;
            const char *allow = getenv("CAT_ALLOW_DEEP_WALL_OVERLAP");
            if (depth > CAT_WALL_CACHE && !(allow && atoi(allow) != 0)) {
                snprintf(g_create_err, sizeof g_create_err, "map blob %d: an agent can touch the bounding boxes of %d walls at once; the state record "
                         "caches %d wall contacts per agent (CAT_WALL_CACHE); CAT_ALLOW_DEEP_WALL_OVERLAP=1 accepts the map", m, depth, CAT_WALL_CACHE);
                return CAT_ERR_BAD_MAP;
            }
        }
        d.f64_off = (int)geo_f.size();
        const size_t n_geo = 4 * (size_t)d.S + 8 * (size_t)d.P;
        geo_f.insert(geo_f.end(), f.begin() + 2, f.begin() + 2 + n_geo);  // drop window w,h: [bb][planes]
        std::vector<int> first_pair((size_t)d.S, 0);
        {   // f32 copy of the plane records for the ray fan's conservative pre-classification (poly_query_feat)
            const double rsum = cfg->wall_radius + cfg->ray_radius;
            const double *pl = f.data() + 2 + 4 * (size_t)d.S;
            float cmax = 0.0f;
            auto rec8 = [&](int q, float *o) {   // n.x n.y c dtMin | dtMax v0.x v0.y -
                const double *r = pl + 8 * (size_t)q;   // n.x n.y v0.x v0.y dot(v0,n) dtMin dtMax -
                o[0] = (float)r[0]; o[1] = (float)r[1]; o[2] = (float)(r[4] + rsum); o[3] = (float)r[5];
                o[4] = (float)r[6]; o[5] = (float)r[2]; o[6] = (float)r[3];
                cmax = std::fmax(cmax, std::fabs(o[2]));
            };
            std::vector<float> prs;
            int pp = 0;
            for (int sidx = 0; sidx < d.S; sidx++) {
                const int first = iv[sidx], count = iv[d.S + sidx];
                first_pair[sidx] = pp;
                for (int e = 0; e < count; e += 2, pp++) {
                    float a[8] = {0}, b[8] = {0.f, 0.f, 1e30f, 1e30f, -1e30f, 1e18f, 1e18f, 0.f};   // b: an edge nothing can reach
                    rec8(first + e, a);
                    if (e + 1 < count) rec8(first + e + 1, b);
                    const float rec[kPairF] = {a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], a[6], b[6], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    prs.insert(prs.end(), rec, rec + kPairF);
                }
            }
            d.PP = pp;
            const size_t at = geo_f.size();
            geo_f.resize(at + (kPairF / 2) * (size_t)pp);
            memcpy(geo_f.data() + at, prs.data(), prs.size() * sizeof(float));
            d.cmax = cmax;
        }
        geo_f.insert(geo_f.end(), f.begin() + 2 + n_geo, f.end());        // [start][regions]
        d.i32_off = (int)geo_i.size();
        geo_i.insert(geo_i.end(), iv.begin(), iv.end());
        geo_i.insert(geo_i.end(), first_pair.begin(), first_pair.end());   // [first S][count S][region_off A+1][first pair S]
        if (geo_f.size() & 1) geo_f.push_back(0.0);  // keep 16-byte alignment of each map's base
        descs[m] = d;
    }
    const int N = cfg->n_envs;
    std::vector<int> slot((size_t)N, 0);
    for (int e = 0; e < N; e++) {
        if (slot_map_ids) slot[e] = slot_map_ids[e];
        if (slot[e] < 0 || slot[e] >= n_maps) {
            snprintf(g_create_err, sizeof g_create_err, "slot_map_ids[%d]=%d out of range", e, slot[e]);
            return CAT_ERR_BAD_SLOT_MAP;
        }
    }
    // ---- spatial-hash grids, one GridHost per map first (cell size: CAT_GRID_CELL px, default 4 ... while the table fits); a map's longest candidate
    //      list decides which form of the ray fan can serve it
    std::vector<GridHost> map_grid((size_t)n_maps);
    std::vector<int> map_grid_max_row((size_t)n_maps, 0);   // per map: its longest candidate list
    {
        // Cell size: the smaller the cell, the tighter the three listing rules (agh-map, entries per ray: 16 px 2.1, 8 px 1.56, 4 px 1.35,
        // 2 px: kernel 64.6 -> 63.2 us for four times the table) and the larger the table (rows of 4 - 8 B per cell and ray: labyrinth
        // 48 MB, agh-map 98 MB at 4 px).  4 px while a map's table stays under 384 MB, else 8, 16 ...; CAT_GRID_CELL fixes it.
        double forced_cell = 0.0;
        if (const char *e = getenv("CAT_GRID_CELL")) { double v = atof(e); if (v >= 2.0 && v <= 512.0) forced_cell = v; }
        const double reach = cfg->ray_length + cfg->ray_radius + 1e-3;
        for (int m = 0; m < n_maps; m++) {
            double cell = forced_cell;
            if (cell == 0.0) {
                const double *bbm = geo_f.data() + descs[m].f64_off;
                double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
                for (int sidx = 0; sidx < descs[m].S; sidx++) {
                    lo[0] = std::fmin(lo[0], bbm[4 * sidx]); lo[1] = std::fmin(lo[1], bbm[4 * sidx + 1]);
                    hi[0] = std::fmax(hi[0], bbm[4 * sidx + 2]); hi[1] = std::fmax(hi[1], bbm[4 * sidx + 3]);
                }
                for (cell = 4.0; cell < 256.0; cell *= 2.0) {
                    const double rows = std::ceil((hi[0] - lo[0] + 2.0 * reach) / cell + 2.0) * std::ceil((hi[1] - lo[1] + 2.0 * reach) / cell + 2.0) * cfg->n_rays;
                    // eight bytes per row unless a map's lists need the wide byte format; a sim of several maps shares the budget of two
                    if (rows * 8.0 <= 384e6 * std::fmin(1.0, 2.0 / n_maps)) break;
                }
            }
            build_grids(geo_f.data() + descs[m].f64_off, descs[m].S, cfg->n_rays, tab->ray_dx, tab->ray_dy, reach, cfg->bbtree_gate != 0,
                        cfg->ray_radius + 2e-6, cell, map_grid[(size_t)m], geo_i.data() + descs[m].i32_off, geo_i.data() + descs[m].i32_off + descs[m].S,
                        cfg->wall_radius + cfg->ray_radius, cfg->ray_radius);
            map_grid_max_row[(size_t)m] = map_grid[(size_t)m].max_row;
        }
    }
    // ---- parts: the maps whose rays meet few walls (every candidate list fits a four-byte row -- the labyrinth's six walls of 5 bits --; shape ids and agents
    //      fit 6 bits; an agent's rays fit four chunks) take the GROUP form of the ray fan (and its pooled kernels where the ring fits the LDS), every other map
    //      the CHUNK form.  A sim whose maps want both runs in ONE part on the chunk form by default; CAT_SPLIT=1 cuts it in two parts -- each with its own
    //      candidate tables, LDS carve, workgroup size, work list and kernels -- that every entry launches side by side on two streams.  Measured (round 5, five
    //      maps x16384, us per tick one part -> two parts): one launch per tick 167.8 -> 185.6 (the two kernels do overlap -- 130 and 170 us inside a 182 us period by
    //      the trace -- but a mixed batch costs the SUM of its workgroups' times either way, the pooled one-tick kernel gains + 1 - 2 % on three of the four box maps and
    //      loses 4 % on lbirinth, and the fork / join events add 14 us between launches); resident T = 64 128.7 -> 127.8.  So the split is kept as a tested option, not
    //      the default.  CAT_FAN=chunks forces the chunk form everywhere.
    std::vector<std::vector<int>> part_maps;
    std::vector<int> part_fan;
    {
        bool chunks_only = false, split = false;
        if (const char *e = getenv("CAT_FAN")) chunks_only = !strcmp(e, "chunks");
        if (const char *e = getenv("CAT_SPLIT")) split = atoi(e) != 0;
        std::vector<int> light, dense;
        for (int m = 0; m < n_maps; m++) (map_grid[(size_t)m].max_row <= 7 && cfg->n_rays <= kGroupRays && !chunks_only ? light : dense).push_back(m);
        if (!light.empty()) {   // the four-byte row must hold the longest list of the part in fields of the part's id width
            int S_l = 0, row_l = 0, idb = 1;
            for (int m : light) { S_l = std::max(S_l, descs[m].S); row_l = std::max(row_l, map_grid[(size_t)m].max_row); }
            while ((1 << idb) <= S_l) idb++;
            if (!(row_l * idb <= 32 && S_l + A <= 63)) { dense.insert(dense.end(), light.begin(), light.end()); light.clear(); }
        }
        if (!light.empty() && !dense.empty() && !split) { dense.insert(dense.end(), light.begin(), light.end()); light.clear(); }
        std::sort(dense.begin(), dense.end());
        if (!light.empty()) { part_maps.push_back(light); part_fan.push_back(1); }
        if (!dense.empty()) { part_maps.push_back(dense); part_fan.push_back(0); }
    }

    cat_sim *s = new cat_sim();
    s->err[0] = 0;
    s->device = device;
    s->maps = descs;
    if (hipSetDevice(device) != hipSuccess) {
        snprintf(g_create_err, sizeof g_create_err, "hipSetDevice(%d) failed", device);
        delete s;
        return CAT_ERR_NO_DEVICE;
    }
    int rc = CAT_OK;
    auto fail = [&](int code) { strncpy(g_create_err, s->err, sizeof g_create_err - 1); cat_destroy(s); return code; };
#define TRY_ALLOC(call) do { rc = (call); if (rc != CAT_OK) return fail(rc); } while (0)
    // ---- what every part shares: the configuration, the ray table and reward LUTs, the geometry of all maps, the env state records, the error word
    Params base;
    memset(&base, 0, sizeof base);
    base.N = N; base.A = A; base.n_cops = cfg->n_cops; base.R = cfg->n_rays; base.max_step = cfg->max_step_count;
    base.iterations = cfg->iterations; base.persistence = cfg->persistence; base.gate = cfg->bbtree_gate;
    base.NP = A * (A - 1) / 2;
    base.env_id_offset = cfg->env_id_offset; base.seed = cfg->seed;
    base.dt = cfg->dt; base.bias_coef = cfg->bias_coef; base.slop = cfg->slop; base.ray_length = cfg->ray_length;
    base.ray_radius = cfg->ray_radius; base.rc = cfg->agent_radius; base.mass = cfg->agent_mass; base.impulse = cfg->impulse;
    base.max_speed = cfg->max_speed; base.term_radius = cfg->termination_radius; base.wall_r = cfg->wall_radius;
    const int NPs_rec = base.NP > 0 ? base.NP : 1;
    base.hot_bytes = 96 * A + 16;
    base.rec_bytes = base.hot_bytes + ((A * kK + NPs_rec) * 8 + (2 * A * kK + NPs_rec) * 4 + 15) / 16 * 16;
    if (base.hot_bytes > kLanes * 16) {   // StateRegs
        snprintf(s->err, sizeof s->err, "state record of %d bytes exceeds the kernels' register staging", base.hot_bytes);
        return fail(CAT_ERR_BAD_CONFIG);
    }
    {   // env state records (layout documented at Params::state)
        std::vector<char> rec0((size_t)N * base.rec_bytes, 0);
        for (int e = 0; e < N; e++) {
            const MapDesc &d = descs[slot[e]];
            const double *start = geo_f.data() + d.f64_off + 4 * d.S + geo_rest_doubles(d);
            double *rd = reinterpret_cast<double *>(rec0.data() + (size_t)e * base.rec_bytes);
            int *ri = reinterpret_cast<int *>(rec0.data() + (size_t)e * base.rec_bytes + base.hot_bytes + (A * kK + NPs_rec) * 8);   // cold ints
            for (int i = 0; i < A; i++) {
                // Entity.__init__ + space.add: caches and BBTree leaf at the start position, v = 0
                const double x = start[2 * i], y = start[2 * i + 1], r = cfg->agent_radius;
                rd[2 * i] = x; rd[2 * i + 1] = y;                       // pos
                rd[6 * A + 2 * i] = x; rd[6 * A + 2 * i + 1] = y;       // tc
                const double l = x - r, b = y - r, rr = x + r, t = y + r;
                const double mx = (rr - l) * 0.1, my = (t - b) * 0.1;
                double *lf = rd + 8 * A + 4 * i;
                lf[0] = l + (-mx < 0.0 ? -mx : 0.0); lf[1] = b + (-my < 0.0 ? -my : 0.0);
                lf[2] = rr + (mx > 0.0 ? mx : 0.0); lf[3] = t + (my > 0.0 ? my : 0.0);
            }
            for (int q = 0; q < A * kK; q++) ri[q] = -1;                // wall_shape: free slots
            for (int q = 0; q < NPs_rec; q++) ri[2 * A * kK + q] = -1;  // pair_age: none
        }
        TRY_ALLOC(dev_alloc(s, &base.state, rec0.size(), rec0.data()));
    }
    TRY_ALLOC(dev_alloc(s, const_cast<double **>(&base.ray_dx), (size_t)base.R, tab->ray_dx));
    TRY_ALLOC(dev_alloc(s, const_cast<double **>(&base.ray_dy), (size_t)base.R, tab->ray_dy));
    TRY_ALLOC(dev_alloc(s, const_cast<float **>(&base.cop_lut), 32768, tab->cop_reward_lut));
    TRY_ALLOC(dev_alloc(s, const_cast<float **>(&base.thief_lut), 32768, tab->thief_reward_lut));
    TRY_ALLOC(dev_alloc(s, const_cast<MapDesc **>(&base.maps), descs.size(), descs.data()));
    TRY_ALLOC(dev_alloc(s, const_cast<double **>(&base.geo_f64), geo_f.size(), geo_f.data()));
    TRY_ALLOC(dev_alloc(s, const_cast<int **>(&base.geo_i32), geo_i.size(), geo_i.data()));
    TRY_ALLOC(dev_alloc(s, &base.err_word, 1, nullptr));
    {   // ray-direction cone parameters: valid when the table is a uniform full circle
        const double two_pi = 6.283185307179586;
        const double a0 = atan2(tab->ray_dy[0], tab->ray_dx[0]);
        const double step = two_pi / base.R;
        bool ok = base.R >= 4;
        for (int k = 0; k < base.R && ok; k++) {
            double d = atan2(tab->ray_dy[k], tab->ray_dx[k]) - (a0 + k * step);
            d -= two_pi * floor(d / two_pi + 0.5);
            if (fabs(d) > 1e-6) ok = false;
        }
        base.ang_ok = ok ? 1 : 0;  // otherwise every shape is paired with every ray (still exact)
        base.ang0 = (float)a0;
        base.inv_step = (float)(1.0 / step);
    }

    s->parts.resize(part_maps.size());
    for (size_t pi = 0; pi < part_maps.size(); pi++) {
        Part &pt = s->parts[pi];
        pt.map_ids = part_maps[pi];
        const int fan = part_fan[pi];
        int maxS = 0, maxP = 0, maxPP = 0, depth = 0, n_part_envs = 0;
        std::vector<int> local_of((size_t)n_maps, -1);   // map -> its index among the part's grids
        for (size_t k = 0; k < pt.map_ids.size(); k++) {
            const int m = pt.map_ids[k];
            local_of[(size_t)m] = (int)k;
            maxS = std::max(maxS, descs[m].S); maxP = std::max(maxP, descs[m].P); maxPP = std::max(maxPP, descs[m].PP);
            depth = std::max(depth, wall_depth[(size_t)m]);
            append_grid(pt.grid, map_grid[(size_t)m]);
            map_grid[(size_t)m] = GridHost();   // the part owns the tables now
        }
        for (int e = 0; e < N; e++) n_part_envs += local_of[(size_t)slot[e]] >= 0;
        pt.n_envs = n_part_envs;
        GridHost &grid_host = pt.grid;
        int id_bits = 1;   // bits of a wall id + 1
        while ((1 << id_bits) <= maxS) id_bits++;
        // the group form reads four-byte rows; the chunk form eight-byte rows of the same fields where the longest list fits
        // (agh-map: 9 walls of 7 bits), else byte rows of 8 / 16 / 32 bytes with the CSR continuation (CAT_GRID_FIELDS=0 forces those)
        bool wide = fan == 0 && grid_host.max_row * id_bits <= 64;
        if (const char *e = getenv("CAT_GRID_FIELDS")) { if (atoi(e) == 0) wide = false; }
        finalize_rows(grid_host, (fan == 1 || wide) ? id_bits : 0, wide);
        if (getenv("CAT_VERBOSE"))
            fprintf(stderr, "[cat_sim] part %zu of %zu: %zu map(s), %d env slots; ray fan: %s form; longest candidate list %d; rows of %d bytes (%s); table %.1f MB\n", pi + 1, part_maps.size(),
                    pt.map_ids.size(), n_part_envs, fan ? "group" : "chunk", grid_host.max_row, fan ? 4 : 8 * grid_host.row_words,
                    (fan || wide) ? "fields of wall id + 1" : "count byte + id bytes, CSR beyond", grid_host.rows.size() * 8 / 1e6);
        // the contact array of a scratch union: what the part's maps make possible (an agent's bb overlaps at most `depth` wall bbs at once -- 2 on the box
        // maps, 5 on agh-map -- and holds at most CAT_WALL_CACHE arbiters), + every agent pair; lane q solves contact q, so never more than a wave's lanes
        const int maxc = std::min(kLanes, A * std::min(depth, kK) + base.NP);
        // ---- LDS carve sizes (must match carve()) and the workgroup size
        LdsSizes ls = lds_sizes(A, cfg->n_rays, maxS, maxP, maxPP, fan == 1, maxc);
        int wpb = 0;
        {   // waves (= env slots) per workgroup.  Most resident waves per CU first (cap 16 = 4 per SIMD at <= 128 VGPRs);
            // among equals a launch of at most two rounds takes the LARGEST workgroup (its waves share ray chunks, which
            // removes the lone-wave tail of a single round), a longer launch the SMALLEST (workgroups of a CU overlap
            // each other's drain).  CAT_WAVES_PER_BLOCK overrides (tuning).
            const int cu = 256;
            int forced = 0;
            if (const char *e = getenv("CAT_WAVES_PER_BLOCK")) forced = atoi(e);
            int best_score = -1, best_w = 0;
            if (forced >= 1 && forced <= kMaxWaves && ls.total(forced) <= 160 * 1024) wpb = forced;   // any size, also not a power of two
            for (int w = 1; w <= kMaxWaves && wpb == 0; w *= 2) {
                const size_t bytes = ls.total(w);
                if (bytes > 160 * 1024) continue;
                int resident = (int)((160 * 1024) / bytes) * w;
                if (resident > 16) resident = 16;
                const bool small_launch = (long long)N <= 2LL * 16 * cu;
                if (resident > best_score || (resident == best_score && small_launch)) { best_score = resident; best_w = w; }
            }
            if (wpb == 0) wpb = best_w;
            if (wpb == 0) {
                snprintf(s->err, sizeof s->err, "LDS budget exceeded: %zu bytes for one env slot", ls.total(1));
                return fail(CAT_ERR_BAD_CONFIG);
            }
        }
        // ---- chunk form: several chunks of a slot per work unit with one item list (fan_slot), where the rows are one word of id fields, ids fit seven bits and
        //      the LDS left beside the env areas gives every wave's scratch union a list of at least 160 items (agh-map 2v1: 240 = the 17 KB that are free)
        int item_cap = 0;
        for (GridDesc &d : grid_host.desc) d.span = d.span_tick = 1;
        {
            const int nch = A * ((cfg->n_rays + kLanes - 1) / kLanes);
            const bool fixed_roster = !(getenv("CAT_GENERIC_KERNEL") && atoi(getenv("CAT_GENERIC_KERNEL")) != 0) && cfg->n_rays == 64 && ((A == 3 && cfg->n_cops == 2) || (A == 5 && cfg->n_cops == 3));
            // (fan_slot is carried by the chunk-form kernels of fixed dimensions only -- select_kernels: 2v1 and 3v2 at 64 rays; 2v1 / 1v1 at 90 rays and the generic
            //  kernels keep one chunk per unit: with run-time dimensions the unit needs scratch)
            if (fan == 0 && fixed_roster && wide && grid_host.row_words == 1 && maxS + A <= 127 && nch >= 2 && grid_host.max_row + A - 1 <= 15) {
#ifdef CAT_PHASE_TIMING
                const size_t budget = 160 * 1024 - 4096;   // (the diagnostic build keeps its cycle accumulators in static LDS)
#else
                const size_t budget = 160 * 1024;
#endif
                const size_t spare = ls.total(wpb) < budget ? (budget - ls.total(wpb)) / (size_t)wpb : 0;
                int cap = (int)(((size_t)ls.uni + spare) / 18) & ~7;
                if (cap > 768) cap = 768;
                if (const char *e = getenv("CAT_ITEM_CAP")) { const int v = atoi(e); if (v >= 16 && v <= cap) cap = v & ~7; }   // tests: lists too short for the chunks of a slot
                if (cap >= 160 || getenv("CAT_ITEM_CAP")) {
                    item_cap = cap;
                    const int need = (18 * cap + 15) / 16 * 16;
                    if (need > ls.uni) ls.uni = need;
                }
            }
            // How many chunks share a unit, per map: as many as (mostly) fit the list in one turn.  Measured on the dense map (agh-map 2v1 x4096, 86 candidates
            // per chunk before the gate, list of 240; us per tick one-tick / resident): 1 chunk per unit 63.8 / 47.3, 2 chunks 61.2 / 45.2, 3 chunks 65.3 / 49.0 -- three
            // chunks do not fit together, the third is traced in a second turn and the unit is one long serial job; the box maps' rays meet few walls (three chunks
            // = ~120 items): five maps x16384 172.5 / 132.1 -> 164.3 / 126.6 with three.  CAT_SLOT_CHUNKS / CAT_SLOT_CHUNKS_TICK override (every map).
            for (size_t k = 0; k < grid_host.desc.size() && item_cap; k++) {
                const int want = map_grid_max_row[(size_t)pt.map_ids[k]] <= 7 ? 3 : 2;
                int v = want < nch ? want : nch, vt = v;
                if (const char *e = getenv("CAT_SLOT_CHUNKS")) { const int u = atoi(e); if (u >= 1 && u <= kSlotChunks) v = u < nch ? u : nch; }
                if (const char *e = getenv("CAT_SLOT_CHUNKS_TICK")) { const int u = atoi(e); if (u >= 1 && u <= kSlotChunks) vt = u < nch ? u : nch; }
                grid_host.desc[k].span = v; grid_host.desc[k].span_tick = vt;
                if (getenv("CAT_VERBOSE"))
                    fprintf(stderr, "[cat_sim] chunk form, map %d: %d chunk(s) per work unit in the resident launch, %d in the one-tick launch; item list of %d\n", pt.map_ids[k], v, vt, item_cap);
            }
        }
        // ---- the workgroup's ray pool (step_kernel_pooled / rollout_kernel_pooled): a ring of wpb * A * R eight-byte entries beside the env areas, where it fits
        // Where the ring fits, the RESIDENT launch always runs pooled (whole runs from the reset, tools/pool_soak.py, M env-steps/s unit -> pooled: labyrinth 201 -> 229,
        // labyrinth-inside 148 -> 159, squarinth 156 -> 169, grandbyrinth 154 -> 169, lbirinth 123.0 -> 123.6).  The ONE-TICK launch pays the sorting pass in its serial
        // chain and loses on a map whose rays all meet a wall (lbirinth 90.5 -> 87.0; labyrinth 132.5 -> 139.1, the others + 1 - 2 %): it runs pooled unless practically no
        // (cell, ray) row sampled around the spawn points is empty (lbirinth 0.008; labyrinth-inside 0.06, squarinth 0.22, grandbyrinth 0.27, labyrinth 0.41).
        // CAT_POOL=1 / 0 forces both / neither.
        int pool_cap = 0, grp_rays = 4 * kLanes;
        double empty_rows = 0.0;
        {   // ... sampled where episodes start: five points of every spawn region (the JSON start position of an agent without regions), every ray
            size_t n_rows = 0, n_empty = 0;
            for (size_t k = 0; k < grid_host.desc.size(); k++) {
                const MapDesc &md = descs[pt.map_ids[k]];
                const GridDesc &gd = grid_host.desc[k];
                const double *start = geo_f.data() + md.f64_off + 4 * md.S + geo_rest_doubles(md), *regions = start + 2 * md.A;
                const int *region_off = geo_i.data() + md.i32_off + 2 * md.S;
                auto sample = [&](double x, double y) {
                    const int cx = (int)floor((x - gd.x0) * gd.inv_cell), cy = (int)floor((y - gd.y0) * gd.inv_cell);
                    if (cx < 0 || cy < 0 || cx >= gd.nx || cy >= gd.ny) return;
                    const size_t r0 = (size_t)gd.off_base + ((size_t)cy * gd.nx + cx) * cfg->n_rays;
                    for (int k2 = 0; k2 < cfg->n_rays; k2++) { n_rows++; n_empty += grid_host.off[r0 + k2 + 1] == grid_host.off[r0 + k2]; }
                };
                for (int i = 0; i < md.A; i++) {
                    const int r0 = region_off[i], nr = region_off[i + 1] - r0;
                    if (nr <= 0) { sample(start[2 * i], start[2 * i + 1]); continue; }
                    for (int q = 0; q < nr; q++) {
                        const double *rg = regions + 4 * (r0 + q);
                        sample(rg[0] + rg[2] / 2, rg[1] + rg[3] / 2);
                        for (int c = 0; c < 4; c++) sample(rg[0] + rg[2] * ((c & 1) ? 0.75 : 0.25), rg[1] + rg[3] * ((c & 2) ? 0.75 : 0.25));
                    }
                }
            }
            empty_rows = n_rows ? (double)n_empty / (double)n_rows : 0.0;
        }
        // ... and on rosters of at most four agents: the pooled one-tick launch keeps the whole sorting pass in the slot's serial front, which grows with the roster
        // (3v2 at 64 rays x8192, round 5: 97.4 us unit form, 99.0 pooled; resident 77.6 -> 74.8 us per tick: the resident launch takes the ring whenever it exists)
        bool want_ring = true, pool_step = empty_rows >= kPoolEmptyRows && A <= 4;
        if (const char *e = getenv("CAT_POOL")) want_ring = pool_step = atoi(e) != 0;
        if (fan == 1 && want_ring) {
            // capacity: the next power of two (ring position by a mask), else wpb * A * R + 64 entries exactly (position by an invariant division); group
            // arrays of the scratch unions: what group_agents() holds at once (two agents up to 128 rays each), else one agent's chunks
            int cap2 = 64;
            while (cap2 < wpb * A * cfg->n_rays) cap2 *= 2;
            const int cap_x = (wpb * A * cfg->n_rays + 64 + 1) / 2 * 2;
            const int cpa = (cfg->n_rays + 63) / 64, gsz = cpa <= 2 ? 2 : 1;
            const int g_full = kLanes * std::min(4, std::min(A, gsz) * cpa), g_one = kLanes * std::min(4, cpa);
            const bool ok_dims = A <= 8 && cfg->n_rays <= 256 && wpb <= 16 && cpa <= 4;
            // While the one-tick launch stays on the unit form, the ring may not take the group arrays below what that kernel's units want (rosters whose rays
            // fill more than four chunks: as many agents as fill four -- group_agents_resident).  3v2 at 64 rays x8192, round 5: with the ring (units of
            // 2 + 2 + 1 agents in the one-tick launch instead of 4 + 1) 102.5 us one-tick / 75.2 resident, without 95.9 - 97.4 / 76.1 - 77.6: the one-tick launch
            // is the one this library is measured on, so the ring stays out (CAT_POOL=1 brings it in: both entries pooled).
            const bool forced = getenv("CAT_POOL") != nullptr;
            const int g_want = (!pool_step && !forced && A * cpa > 4) ? kLanes * 4 : 0;
            for (int attempt = 0; ok_dims && attempt < 3 && !pool_cap; attempt++) {
                const int cap = attempt == 0 ? cap2 : cap_x, g2 = std::max(attempt < 2 ? g_full : g_one, g_want);
                const LdsSizes l2 = lds_sizes(A, cfg->n_rays, maxS, maxP, maxPP, true, maxc, g2);
                if (l2.total(wpb) + 16 + (size_t)cap * 8 <= 160 * 1024) { pool_cap = cap; grp_rays = g2; ls = l2; }
            }
        }
        if (getenv("CAT_VERBOSE")) {   // contact-candidate rows (agent_setup): how many cells overflow the packed row of seven
            size_t n = 0, n0 = 0, n7 = 0, n15 = 0; int mx = 0;
            for (unsigned long long w : grid_host.crows) { const int c = (int)(w & 0xFF); n++; n0 += c > 0; n7 += c > 7; n15 += c > 15; if (c > mx) mx = c; }
            fprintf(stderr, "[cat_sim] contact rows: %zu cells, %.3f with a candidate, %.4f with more than 7 (CSR walk), %.4f with more than 15; longest %d; contact array of %d (wall bb depth %d)\n", n,
                    n ? (double)n0 / n : 0.0, n ? (double)n7 / n : 0.0, n ? (double)n15 / n : 0.0, mx, maxc, depth);
        }
        if (!pool_cap) pool_step = false;
        if (getenv("CAT_VERBOSE"))
            fprintf(stderr, "[cat_sim] ray pool: %d entries, group arrays for %d rays (resident launch: %s, one-tick launch: %s); rows without a candidate around the spawn points: %.3f; "
                    "LDS %zu bytes per workgroup of %d waves\n", pool_cap, grp_rays, pool_cap ? "pooled" : "unit form", pool_step ? "pooled" : "unit form", empty_rows,
                    ls.total(wpb) + (pool_cap ? 16 + (size_t)pool_cap * 8 : 0), wpb);
        // ---- work list: workgroups are map-homogeneous; env slots grouped by map, padded with -1
        std::vector<int> work, block_map;
        int helpers = 0;   // CAT_HELPERS (diagnostic): that many waves of every workgroup own no env slot and only take work units
        if (const char *e = getenv("CAT_HELPERS")) { helpers = atoi(e); if (helpers < 0 || helpers >= wpb) helpers = 0; }
        const int epb = wpb - helpers;
        for (int m : pt.map_ids) {
            int cnt = 0;
            for (int e = 0; e < N; e++)
                if (slot[e] == m) {
                    if (cnt % wpb == 0) block_map.push_back(m);
                    work.push_back(e);
                    cnt++;
                    if (cnt % wpb == epb) for (int h = 0; h < helpers; h++) { work.push_back(-1); cnt++; }
                }
            while (cnt % wpb) { work.push_back(-1); cnt++; }
        }
        pt.n_blocks = (int)block_map.size();
        Params &p = pt.p;
        p = base;
        p.maxc = maxc;
        {
            p.row_words = grid_host.row_words;
            p.row_id_bits = grid_host.id_bits;
            if (p.row_id_bits) {
                p.row_cnt_mul = (65536 + p.row_id_bits - 1) / p.row_id_bits;
                for (int b = 0; b < 64; b++)
                    if (((b * p.row_cnt_mul) >> 16) != b / p.row_id_bits) { snprintf(s->err, sizeof s->err, "row field divider"); return fail(CAT_ERR_BAD_CONFIG); }
            }
            TRY_ALLOC(dev_alloc(s, const_cast<GridDesc **>(&p.grids), grid_host.desc.size(), grid_host.desc.data()));
            TRY_ALLOC(dev_alloc(s, const_cast<unsigned long long **>(&p.grid_rows), grid_host.rows.size(), grid_host.rows.data()));
            // the CSR arrays of the ray grid are only read for lists beyond a row's capacity: not uploaded when no list is that long
            const bool csr = !grid_host.id_bits && grid_host.max_row > 8 * grid_host.row_words - 1;
            TRY_ALLOC(dev_alloc(s, const_cast<int **>(&p.grid_off), csr ? grid_host.off.size() : 1, csr ? grid_host.off.data() : nullptr));
            TRY_ALLOC(dev_alloc(s, const_cast<unsigned char **>(&p.grid_ent), csr ? grid_host.ent.size() : 1, csr ? grid_host.ent.data() : nullptr));
            TRY_ALLOC(dev_alloc(s, const_cast<int **>(&p.cgrid_off), grid_host.coff.size(), grid_host.coff.data()));
            TRY_ALLOC(dev_alloc(s, const_cast<unsigned char **>(&p.cgrid_ent), grid_host.cent.size(), grid_host.cent.data()));
            TRY_ALLOC(dev_alloc(s, const_cast<unsigned long long **>(&p.cgrid_rows), grid_host.crows.size(), grid_host.crows.data()));
        }
        TRY_ALLOC(dev_alloc(s, const_cast<int **>(&p.work_env), work.size(), work.data()));
        TRY_ALLOC(dev_alloc(s, const_cast<int **>(&p.block_map), block_map.size(), block_map.data()));
        {
            std::vector<BlockDesc> bd(block_map.size());
            for (size_t b = 0; b < block_map.size(); b++) {
                bd[b].md = descs[block_map[b]];
                bd[b].gd = grid_host.desc[(size_t)local_of[(size_t)block_map[b]]];
            }
            TRY_ALLOC(dev_alloc(s, const_cast<BlockDesc **>(&p.block_desc), bd.size(), bd.data()));
        }
        p.maxE = maxS + A;
        p.lds_map_bytes = ls.map; p.lds_env_bytes = ls.env; p.lds_union_bytes = ls.uni; p.wpb = wpb;
        p.grp_rays = grp_rays;
        p.item_cap = item_cap;
        p.lds_pool_off = pool_cap ? (int)((ls.total(wpb) + 15) / 16 * 16) : 0;
        p.pool_mask = pool_cap ? pool_cap - 1 : 0;
        p.pool_magic = 0u; p.pool_shift = -1;
        if (pool_cap && (pool_cap & (pool_cap - 1))) {   // not a power of two: floor(i / cap) = (t + ((i - t) >> 1)) >> shift with t = mulhi(magic, i)  [Granlund & Montgomery]
            int l = 0;
            while ((1u << l) < (unsigned)pool_cap) l++;
            p.pool_magic = (unsigned)((((unsigned long long)1 << 32) * ((1ull << l) - (unsigned long long)pool_cap)) / (unsigned long long)pool_cap + 1ull);
            p.pool_shift = l - 1;
        }
        pt.wpb = wpb;
        pt.lds_bytes = pool_cap ? (size_t)p.lds_pool_off + (size_t)pool_cap * 8 : ls.total(wpb);
        pt.kernel_variant = select_kernels(A, p.R, p.n_cops, fan, pool_cap != 0, pool_step, p.pool_shift >= 0, pt.reset_fn, pt.rollout_fn, pt.step_fn);
        pt.pool_step = pool_step;
        if (pt.lds_bytes > 64 * 1024) {
            hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(pt.step_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pt.lds_bytes);
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(pt.reset_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pt.lds_bytes);
            hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void *>(pt.rollout_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pt.lds_bytes);
            if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
                snprintf(s->err, sizeof s->err, "cannot raise dynamic LDS limit to %zu", pt.lds_bytes);
                return fail(CAT_ERR_HIP);
            }
        }
        {
            Params *dp = nullptr;
            rc = dev_alloc(s, &dp, 1, &p);
            if (rc != CAT_OK) return fail(rc);
            pt.dev_p = dp;
        }
        {   // the prologue's copy (third kernel argument)
            Prologue &q = pt.pro;
            memset(&q, 0, sizeof q);
            q.lds_map_bytes = p.lds_map_bytes; q.lds_env_bytes = p.lds_env_bytes; q.lds_union_bytes = p.lds_union_bytes; q.wpb = p.wpb;
            q.A = p.A; q.R = p.R; q.NP = p.NP; q.maxc = p.maxc; q.n_cops = p.n_cops; q.rec_bytes = p.rec_bytes; q.hot_bytes = p.hot_bytes; q.N = p.N;
            q.lds_pool_off = p.lds_pool_off; q.pool_mask = p.pool_mask; q.grp_rays = p.grp_rays;
            q.work_env = p.work_env; q.block_desc = p.block_desc; q.state = p.state; q.geo_f64 = p.geo_f64; q.geo_i32 = p.geo_i32;
            q.ray_dx = p.ray_dx; q.ray_dy = p.ray_dy; q.cop_lut = p.cop_lut; q.thief_lut = p.thief_lut;
            bool ident = n_maps == 1 && (int)work.size() == pt.n_blocks * wpb;
            for (size_t k = 0; ident && k < work.size(); k++) ident = work[k] == ((int)k < N ? (int)k : -1);
            q.uniform = ident ? 1 : 0;
            q.bd.md = descs[block_map[0]];
            q.bd.gd = grid_host.desc[(size_t)local_of[(size_t)block_map[0]]];
        }
    }
#undef TRY_ALLOC
    if (s->parts.size() > 1) {   // the second part's launches run on a stream of the handle, forked from and joined to the caller's (launch_parts)
        if (hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) != hipSuccess) {
            snprintf(s->err, sizeof s->err, "cannot create the second part's stream / events");
            return fail(CAT_ERR_HIP);
        }
    }
    for (size_t pi = 0; pi < s->parts.size(); pi++) {
        const Part &pt = s->parts[pi];
        s->one_tick_name += (pi ? "+" : "") + std::string(pt.pool_step ? "step_kernel_pooled" : "step_kernel");
        s->rollout_name += (pi ? "+" : "") + std::string(pt.p.pool_mask ? "rollout_kernel_pooled" : "rollout_kernel");
    }
    *out = s;
    return CAT_OK;
}

extern "C" int cat_destroy(cat_sim *s)
{
    if (!s) return CAT_ERR_BAD_ARG;
    (void)hipSetDevice(s->device);
    if (s->side) { (void)hipStreamSynchronize(s->side); (void)hipStreamDestroy(s->side); }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    for (void *d : s->allocs) (void)hipFree(d);
    delete s;
    return CAT_OK;
}

// One dispatch per part.  A sim of two parts launches the second on the handle's own stream, forked from the caller's stream by an event and joined back
// to it by another: for the caller the entry stays one stream-ordered operation, and the two kernels share the device.  An armed pair of timing events
// (cat_arm_kernel_timing) is attached to the dispatch itself when there is one, else recorded on the caller's stream around the fork and the join.
enum { kFnReset, kFnStep, kFnRollout };
static int launch_parts(cat_sim *s, const LaunchArgs &la, void *stream, int which)
{
    hipStream_t user = static_cast<hipStream_t>(stream);
    const bool two = s->parts.size() > 1;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (which != kFnReset && s->t_start && s->t_stop) { t0 = s->t_start; t1 = s->t_stop; s->t_start = s->t_stop = nullptr; }
    auto launch = [&](const Part &pt, hipStream_t st, bool events) {
        const KernelFn fn = which == kFnReset ? pt.reset_fn : (which == kFnStep ? pt.step_fn : pt.rollout_fn);
        const dim3 grid(pt.n_blocks), block(pt.wpb * kLanes);
        if (events) hipExtLaunchKernelGGL(fn, grid, block, pt.lds_bytes, st, t0, t1, 0, pt.dev_p, la, pt.pro);
        else hipLaunchKernelGGL(fn, grid, block, pt.lds_bytes, st, pt.dev_p, la, pt.pro);
    };
    if (!two) { launch(s->parts[0], user, t0 != nullptr); return CAT_OK; }
    if (t0) HIP_TRY(s, hipEventRecord(t0, user));
    HIP_TRY(s, hipEventRecord(s->ev_fork, user));
    HIP_TRY(s, hipStreamWaitEvent(s->side, s->ev_fork, 0));
    launch(s->parts[1], s->side, false);   // the chunk-form part first: its workgroups are the long ones
    HIP_TRY(s, hipEventRecord(s->ev_join, s->side));
    launch(s->parts[0], user, false);
    HIP_TRY(s, hipStreamWaitEvent(user, s->ev_join, 0));
    if (t1) HIP_TRY(s, hipEventRecord(t1, user));
    return CAT_OK;
}

static int launch_reset(cat_sim *s, const uint8_t *mask, const double *positions, const cat_outputs *out,
                        int use_done, void *stream)
{
    if (!s) return CAT_ERR_BAD_ARG;
    HIP_TRY(s, hipSetDevice(s->device));
    LaunchArgs la;
    memset(&la, 0, sizeof la);
    if (out) la.out = *out;
    la.mask = mask; la.positions = positions; la.use_done_mask = use_done;
    const int rc = launch_parts(s, la, stream, kFnReset);
    if (rc != CAT_OK) return rc;
    HIP_TRY(s, hipGetLastError());
    return CAT_OK;
}

extern "C" int cat_reset(cat_sim *s, const uint8_t *mask, const double *positions, const cat_outputs *out, void *stream)
{
    return launch_reset(s, mask, positions, out, 0, stream);
}

extern "C" int cat_reset_done(cat_sim *s, const cat_outputs *out, void *stream)
{
    return launch_reset(s, nullptr, nullptr, out, 1, stream);
}

extern "C" int cat_arm_kernel_timing(cat_sim *s, void *start_event, void *stop_event)
{
    if (!s || !start_event || !stop_event) return CAT_ERR_BAD_ARG;
    s->t_start = static_cast<hipEvent_t>(start_event); s->t_stop = static_cast<hipEvent_t>(stop_event);
    return CAT_OK;
}

extern "C" int cat_step(cat_sim *s, const int32_t *actions, const cat_outputs *out, void *stream)
{
    if (!s || !actions) { if (s) snprintf(s->err, sizeof s->err, "cat_step: actions is NULL"); return CAT_ERR_BAD_ARG; }
    HIP_TRY(s, hipSetDevice(s->device));
    LaunchArgs la;
    memset(&la, 0, sizeof la);
    if (out) la.out = *out;
    la.actions = actions;
    const int rc = launch_parts(s, la, stream, kFnStep);
    if (rc != CAT_OK) return rc;
    HIP_TRY(s, hipGetLastError());
    return CAT_OK;
}

extern "C" int cat_step_fused(cat_sim *s, const int32_t *actions, uint64_t synth_tick, int auto_reset,
                              const cat_outputs *out, void *stream)
{
    if (!s) return CAT_ERR_BAD_ARG;
    HIP_TRY(s, hipSetDevice(s->device));
    LaunchArgs la;
    memset(&la, 0, sizeof la);
    if (out) la.out = *out;
    la.actions = actions; la.synth_tick = synth_tick;
    la.auto_reset = auto_reset ? 1 : 0;   // finished episodes are reset inside the same launch (no second kernel)
    const int rc = launch_parts(s, la, stream, kFnStep);
    if (rc != CAT_OK) return rc;
    HIP_TRY(s, hipGetLastError());
    return CAT_OK;
}

extern "C" int cat_rollout_fused(cat_sim *s, int T, const int32_t *actions, uint64_t synth_tick0, int auto_reset,
                                 const cat_outputs *out, void *stream)
{
    if (!s) return CAT_ERR_BAD_ARG;
    if (T < 1 || T > CAT_MAX_ROLLOUT_TICKS) { snprintf(s->err, sizeof s->err, "cat_rollout_fused: T = %d outside 1..%d", T, CAT_MAX_ROLLOUT_TICKS); return CAT_ERR_BAD_ARG; }
    HIP_TRY(s, hipSetDevice(s->device));
    LaunchArgs la;
    memset(&la, 0, sizeof la);
    if (out) la.out = *out;
    la.actions = actions; la.synth_tick = synth_tick0; la.T = T;
    la.auto_reset = auto_reset ? 1 : 0;
    const int rc = launch_parts(s, la, stream, kFnRollout);
    if (rc != CAT_OK) return rc;
    HIP_TRY(s, hipGetLastError());
    return CAT_OK;
}

extern "C" int cat_random_actions(cat_sim *s, uint64_t tick, int32_t *actions, void *stream)
{
    if (!s || !actions) return CAT_ERR_BAD_ARG;
    HIP_TRY(s, hipSetDevice(s->device));
    const int n = s->parts[0].p.N * s->parts[0].p.A;
    hipLaunchKernelGGL(random_actions_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       s->parts[0].dev_p, (unsigned long long)tick, actions);
    HIP_TRY(s, hipGetLastError());
    return CAT_OK;
}

extern "C" int cat_device_errors(cat_sim *s, uint32_t *flags, int clear, void *stream)
{
    if (!s || !flags) return CAT_ERR_BAD_ARG;
    HIP_TRY(s, hipSetDevice(s->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(s, hipMemcpyAsync(flags, s->parts[0].p.err_word, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (clear) HIP_TRY(s, hipMemsetAsync(s->parts[0].p.err_word, 0, sizeof(uint32_t), st));
    HIP_TRY(s, hipStreamSynchronize(st));
    if (*flags) snprintf(s->err, sizeof s->err, "device-side error flags 0x%x:%s%s%s", *flags,
                         (*flags & CAT_DEVERR_BAD_ACTION) ? " an action outside 0..3 (applied as no impulse)" : "",
                         (*flags & CAT_DEVERR_CONTACT_DROPPED) ? " a contact was dropped (more simultaneous contacts than the cache / contact array holds)" : "",
                         (*flags & CAT_DEVERR_SCHEDULER) ? " a work item of the pooled ray fan never arrived (results of that launch are invalid)" : "");
    return CAT_OK;
}

extern "C" int cat_set_seed(cat_sim *s, uint64_t seed, void *stream)
{
    if (!s) return CAT_ERR_BAD_ARG;
    HIP_TRY(s, hipSetDevice(s->device));
    for (Part &pt : s->parts) {
        pt.p.seed = seed;
        // the 8-byte source lives in the handle, which outlives the async copy
        HIP_TRY(s, hipMemcpyAsync(&pt.dev_p->seed, &pt.p.seed, sizeof(pt.p.seed), hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    }
    return CAT_OK;
}

static int copy_state(cat_sim *s, const cat_state *v, bool get, void *stream)
{
    if (!s || !v) return CAT_ERR_BAD_ARG;
    HIP_TRY(s, hipSetDevice(s->device));
    const Params &p = s->parts[0].p;   // (the record layout and the state pointer are the same in every part)
    const int A = p.A, NPs = p.NP > 0 ? p.NP : 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // field <-> strided slice of the per-env records
    auto cp = [&](void *user, size_t rec_off, size_t width) -> hipError_t {
        if (!user || width == 0) return hipSuccess;
        char *recp = p.state + rec_off;
        return get ? hipMemcpy2DAsync(user, width, recp, (size_t)p.rec_bytes, width, (size_t)p.N, hipMemcpyDeviceToDevice, st)
                   : hipMemcpy2DAsync(recp, (size_t)p.rec_bytes, user, width, width, (size_t)p.N, hipMemcpyDeviceToDevice, st);
    };
    const size_t D = 8, I = 4, hot = (size_t)p.hot_bytes, ci = hot + ((size_t)A * kK + NPs) * D;   // cold f64 at `hot`, cold i32 at `ci`
    const bool cold_touched = v->wall_jn || v->pair_jn || v->wall_shape || v->wall_age || v->pair_age;
    // a slot whose cache_live flag is 0 keeps STALE bytes in the cold part of its record: make them say "empty" before
    // they are read out, and raise the flag of every slot after cold fields were written from outside
    if (get) hipLaunchKernelGGL(cold_fixup_kernel, dim3((p.N + 255) / 256), dim3(256), 0, st, s->parts[0].dev_p, 0);
    HIP_TRY(s, cp(v->pos, 0, 2 * A * D));
    HIP_TRY(s, cp(v->vel, 2 * A * D, 2 * A * D));
    HIP_TRY(s, cp(v->vbias, 4 * A * D, 2 * A * D));
    HIP_TRY(s, cp(v->tc, 6 * A * D, 2 * A * D));
    HIP_TRY(s, cp(v->leaf_bb, 8 * A * D, 4 * A * D));
    HIP_TRY(s, cp(v->step_count, 12 * A * D, I));
    HIP_TRY(s, cp(v->reset_count, 12 * A * D + I, I));
    HIP_TRY(s, cp(v->wall_jn, hot, (size_t)A * kK * D));
    if (p.NP > 0) HIP_TRY(s, cp(v->pair_jn, hot + (size_t)A * kK * D, (size_t)p.NP * D));
    HIP_TRY(s, cp(v->wall_shape, ci, (size_t)A * kK * I));
    HIP_TRY(s, cp(v->wall_age, ci + (size_t)A * kK * I, (size_t)A * kK * I));
    if (p.NP > 0) HIP_TRY(s, cp(v->pair_age, ci + 2 * (size_t)A * kK * I, (size_t)p.NP * I));
    if (!get && cold_touched) hipLaunchKernelGGL(cold_fixup_kernel, dim3((p.N + 255) / 256), dim3(256), 0, st, s->parts[0].dev_p, 1);
    HIP_TRY(s, hipGetLastError());
    return CAT_OK;
}

extern "C" int cat_get_state(cat_sim *s, const cat_state *dst, void *stream) { return copy_state(s, dst, true, stream); }
extern "C" int cat_set_state(cat_sim *s, const cat_state *src, void *stream) { return copy_state(s, src, false, stream); }

#ifdef CAT_PHASE_TIMING
static unsigned long long g_last_counts[8];
// event counters as of the last cat_debug_phase_cycles call: shape-query rounds / their lanes, classification iterations / lanes,
// exact face iterations / lanes, exact corner iterations / lanes (the counting distorts the cycle marks of the same run)
extern "C" void cat_debug_counts(unsigned long long *out8) { for (int i = 0; i < 8; i++) out8[i] = g_last_counts[i]; }
extern "C" int cat_debug_phase_cycles(unsigned long long *out24, int reset)
{
    unsigned long long h[32];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_cycles), sizeof h) != hipSuccess) return CAT_ERR_HIP;
    for (int i = 0; i < 24; i++) out24[i] = h[i];
    for (int i = 24; i < 32; i++) g_last_counts[i - 24] = h[i];
    if (reset) { memset(h, 0, sizeof h); if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), h, sizeof h) != hipSuccess) return CAT_ERR_HIP; }
    return CAT_OK;
}
#endif

static int grid_lookup(const GridHost &g, int map_index, int R, double x, double y, int k, int *out, int max_out)
{
    const GridDesc &d = g.desc[map_index];
    const int cx = (int)std::floor((x - d.x0) * d.inv_cell), cy = (int)std::floor((y - d.y0) * d.inv_cell);
    if (cx < 0 || cy < 0 || cx >= d.nx || cy >= d.ny) return 0;
    const int cellid = cy * d.nx + cx;
    int o0, o1;
    const unsigned char *ent;
    if (k >= 0) {
        const int row = cellid * R + k;
        o0 = g.off[d.off_base + row]; o1 = g.off[d.off_base + row + 1];
        ent = g.ent.data() + d.ent_base;
    } else {
        o0 = g.coff[d.coff_base + cellid]; o1 = g.coff[d.coff_base + cellid + 1];
        ent = g.cent.data() + d.cent_base;
    }
    int n = 0;
    for (int i = o0; i < o1 && n < max_out; i++) out[n++] = ent[i];
    return o1 - o0;
}

// Host-only construction of the tables of ONE map (no device needed): used by the CPU tests that
// check the tables are supersets of the exact gate.
struct cat_grid_host { GridHost g; int R; };

extern "C" int cat_grid_build_host(const cat_config *cfg, const cat_tables *tab, const void *blob, size_t size,
                                   double cell, cat_grid_host **out)
{
    if (!cfg || !tab || !blob || !out || size < 64) return CAT_ERR_BAD_ARG;
    int32_t h[16];
    memcpy(h, blob, 64);
    const int S = h[2], P = h[3], A = h[4], Rg = h[7];
    const size_t nf = 2 + 4 * (size_t)S + 8 * (size_t)P + 2 * (size_t)A + 4 * (size_t)Rg;
    if ((unsigned)h[0] != kBlobMagic || size < 64 + nf * 8 + 2 * (size_t)S * 4 || S < 1 || S > CAT_MAX_SHAPES) return CAT_ERR_BAD_MAP;
    std::vector<double> f(nf);
    std::vector<int> iv(2 * (size_t)S);   // [first plane S][plane count S]
    memcpy(f.data(), static_cast<const unsigned char *>(blob) + 64, nf * 8);
    memcpy(iv.data(), static_cast<const unsigned char *>(blob) + 64 + nf * 8, iv.size() * 4);
    cat_grid_host *gh = new cat_grid_host();
    gh->R = cfg->n_rays;
    build_grids(f.data() + 2, S, cfg->n_rays, tab->ray_dx, tab->ray_dy, cfg->ray_length + cfg->ray_radius + 1e-3,
                cfg->bbtree_gate != 0, cfg->ray_radius + 2e-6, cell > 0 ? cell : 8.0, gh->g,
                iv.data(), iv.data() + S, cfg->wall_radius + cfg->ray_radius, cfg->ray_radius);
    finalize_rows(gh->g);
    *out = gh;
    return CAT_OK;
}

extern "C" int cat_grid_lookup_host(const cat_grid_host *gh, double x, double y, int k, int *out, int max_out)
{
    if (!gh || k >= gh->R) return CAT_ERR_BAD_ARG;
    return grid_lookup(gh->g, 0, gh->R, x, y, k, out, max_out);
}

extern "C" long long cat_grid_bytes_host(const cat_grid_host *gh)
{
    return gh ? (long long)(gh->g.rows.size() * 8 + gh->g.off.size() * 4 + gh->g.ent.size() + gh->g.coff.size() * 4 + gh->g.cent.size()) : 0;
}

extern "C" void cat_grid_free_host(cat_grid_host *gh) { delete gh; }

// Host copy of the spatial-hash tables, for tests: candidate walls of ray k (k >= 0) or contact
// candidates (k < 0) for an origin at (x, y) on map `map_index`.  Returns the count (ids in out).
extern "C" int cat_debug_grid_lookup(const cat_sim *s, int map_index, double x, double y, int k, int *out, int max_out)
{
    if (!s || map_index < 0 || map_index >= (int)s->maps.size() || k >= s->parts[0].p.R) return CAT_ERR_BAD_ARG;
    for (const Part &pt : s->parts)
        for (size_t q = 0; q < pt.map_ids.size(); q++)
            if (pt.map_ids[q] == map_index) return grid_lookup(pt.grid, (int)q, pt.p.R, x, y, k, out, max_out);
    return CAT_ERR_BAD_ARG;
}

extern "C" int cat_selftest_arith(int op, const double *a, const double *b, double *out, int n, int device, void *stream)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) {
        snprintf(g_create_err, sizeof g_create_err, "no usable HIP device");
        return CAT_ERR_NO_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) return CAT_ERR_NO_DEVICE;
    hipLaunchKernelGGL(selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), op, a, b, out, n);
    return hipGetLastError() == hipSuccess ? CAT_OK : CAT_ERR_HIP;
}
