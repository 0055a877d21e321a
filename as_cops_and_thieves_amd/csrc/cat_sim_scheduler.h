// cat_sim_scheduler.h -- part of the env core's single translation unit (included by cat_sim.hip, in this order; not a stand-alone header):
// LDS carve and map staging, state records, the work-unit schedulers (reset kernel; one-tick / resident kernels in the unit and the pooled form), spawn sampling, small utility kernels.
// ------------------------------------------------------------------ kernel plumbing -----------
// LDS of a workgroup: [map | ctrl | wpb env areas | wpb scratch unions].  An env area holds the state
// record, the snapshot, the per-agent ray-fan setup and the output staging of ONE env slot; a scratch
// union belongs to ONE wave (contact arrays / ray-fan items: disjoint phases).  A wave working on
// another slot's ray chunks combines that slot's env area with its own scratch.
template <class D>
__device__ __forceinline__ Lds carve(const Params &p, char *smem, const MapDesc &md, int slot, int wave)
{
    Lds L;
    const int S = md.S, P = md.P, W = p.wpb;
    L.bb = reinterpret_cast<const double *>(smem);
    L.planes = L.bb + kBB * S;
    L.p32 = reinterpret_cast<const float *>(L.planes + 8 * P);
    L.fc = reinterpret_cast<const int *>(L.planes + geo_rest_doubles(md));
    L.fp = L.fc + S;
    L.rayd = reinterpret_cast<const double *>(smem + p.lds_map_bytes - 16 * D::R(p));
    L.ctrl = reinterpret_cast<int *>(smem + p.lds_map_bytes);
    char *w = smem + p.lds_map_bytes + ctrl_bytes(W) + slot * p.lds_env_bytes;
    const int A = D::A(p), R = D::R(p), NPs = D::NP(p) > 0 ? D::NP(p) : 1;
    L.rec = w;
    double *d = reinterpret_cast<double *>(w);
    L.pos = d; d += 2 * A; L.vel = d; d += 2 * A; L.vb = d; d += 2 * A; L.tc = d; d += 2 * A;
    L.leaf = d; d += 4 * A;
    L.cnt = reinterpret_cast<int *>(d);                       // step_count reset_count done cache_live: end of the hot part
    d = reinterpret_cast<double *>(w + D::hot_bytes(p));      // the cold part: arbiter caches
    L.wjn = d; d += A * kK; L.pjn = d; d += NPs;
    {
        int *ri = reinterpret_cast<int *>(d);
        L.wsh = ri; ri += A * kK; L.wag = ri; ri += A * kK; L.pag = ri;
    }
    d = reinterpret_cast<double *>(w + D::rec_bytes(p));
    L.spawn = d; L.fpos = d; L.ftc = d + 2 * A; L.fleaf = d + 4 * A; d += 8 * A;
    int *iv = reinterpret_cast<int *>(d);
    L.acell = iv; iv += A; L.anear = iv; iv += 2 * A; L.dk0 = iv; iv += A * A; L.dcnt = iv; iv += A * A; L.adn = iv; iv += A;
    L.dmin = reinterpret_cast<unsigned *>(iv); iv += A;
    L.flags = iv; iv += 4;
    {   // output staging, every array 16-byte aligned
        char *o = reinterpret_cast<char *>(iv);
        o = w + align_up((int)(o - w), 16);
        L.od = reinterpret_cast<unsigned short *>(o); o += align_up(A * R * 2, 16);
        L.ot = reinterpret_cast<unsigned char *>(o); o += align_up(A * R, 16);
        L.sd = reinterpret_cast<unsigned short *>(o); o += align_up(2 * R * 2, 16);
        L.st = reinterpret_cast<unsigned char *>(o);
    }
    // union: contact arrays (physics) / ray-fan scratch
    char *u = smem + p.lds_map_bytes + ctrl_bytes(W) + W * p.lds_env_bytes + wave * p.lds_union_bytes;
    L.conf = reinterpret_cast<double *>(u);
    L.itbb = reinterpret_cast<double *>(u);
    L.ialpha = L.itbb + kItemCap;
    L.itm = reinterpret_cast<unsigned short *>(L.ialpha + kItemCap);
    L.itemidx = L.itm + kItemCap;
    L.arow = reinterpret_cast<unsigned *>(u + kFanBytes);
    const int grays = p.grp_rays;   // most rays of one agent group (four chunks, fewer where the workgroup's ray pool needs the LDS): what the arrays are sized for
    L.alist = reinterpret_cast<unsigned char *>(L.arow + grays);
    L.adyn = L.alist + grays;
    return L;
}

template <class D>
__device__ __forceinline__ void stage_map(const Params &p, char *smem, const MapDesc &md, const BlockDesc *desc = nullptr, BlockDesc *desc_dst = nullptr)
{
    const int nrest = geo_rest_doubles(md), nf = kBB * md.S + nrest;      // doubles of geometry in LDS
    double *dst = reinterpret_cast<double *>(smem);
    GAS const double *src = G(p.geo_f64) + md.f64_off;
    {   // wall bbs: 32-byte records in memory, kBB doubles apart in LDS (16-byte copies)
        GAS const f64x2 *s2 = (GAS const f64x2 *)src;
        for (int i = threadIdx.x; i < 2 * md.S; i += blockDim.x)
            *reinterpret_cast<f64x2 *>(dst + kBB * (i >> 1) + 2 * (i & 1)) = s2[i];
    }
    {   // the rest as it lies: 16-byte copies, four in flight per thread (every map base is 16-byte aligned, sizes even)
        GAS const f64x2 *s2 = (GAS const f64x2 *)(src + 4 * md.S);
        f64x2 *d2 = reinterpret_cast<f64x2 *>(dst + kBB * md.S);
        const int n2 = nrest / 2, T = blockDim.x;
        for (int i = threadIdx.x; i < n2; i += 4 * T) {
            f64x2 v0 = s2[i], v1, v2, v3;
            const bool h1 = i + T < n2, h2 = i + 2 * T < n2, h3 = i + 3 * T < n2;
            if (h1) v1 = s2[i + T];
            if (h2) v2 = s2[i + 2 * T];
            if (h3) v3 = s2[i + 3 * T];
            d2[i] = v0;
            if (h1) d2[i + T] = v1;
            if (h2) d2[i + 2 * T] = v2;
            if (h3) d2[i + 3 * T] = v3;
        }
    }
    int *di = reinterpret_cast<int *>(dst + nf);
    GAS const int *si = G(p.geo_i32) + md.i32_off;
    for (int i = threadIdx.x; i < md.S; i += blockDim.x) {
        di[i] = si[i] | (si[md.S + i] << 16);
        di[md.S + i] = si[2 * md.S + md.A + 1 + i];        // first edge-pair record
    }
    double *rd = reinterpret_cast<double *>(smem + p.lds_map_bytes - 16 * D::R(p));
    for (int i = threadIdx.x; i < D::R(p); i += blockDim.x) { rd[2 * i] = G(p.ray_dx)[i]; rd[2 * i + 1] = G(p.ray_dy)[i]; }
    // the workgroup's BlockDesc -> LDS, from the registers the caller loaded it into (no second trip to memory); member by member:
    // a struct copy would put the source on the stack
    if (desc_dst && threadIdx.x == 0) {
        const MapDesc &m = desc->md;
        const GridDesc &g = desc->gd;
        MapDesc &dm = desc_dst->md;
        GridDesc &dg = desc_dst->gd;
        dm.S = m.S; dm.P = m.P; dm.A = m.A; dm.n_regions = m.n_regions; dm.f64_off = m.f64_off; dm.i32_off = m.i32_off; dm.cmax = m.cmax; dm.PP = m.PP;
        dg.x0 = g.x0; dg.y0 = g.y0; dg.inv_cell = g.inv_cell; dg.nx = g.nx; dg.ny = g.ny; dg.off_base = g.off_base; dg.ent_base = g.ent_base;
        dg.coff_base = g.coff_base; dg.cent_base = g.cent_base; dg.crow_base = g.crow_base; dg.span = g.span; dg.row_base = g.row_base; dg.span_tick = g.span_tick;
    }
    __syncthreads();
}

// The cold part of a slot's record (the arbiter caches): fetched from HBM when the hot part says it holds something,
// else set to "no cached arbiter" in LDS.  Called once the hot part is in LDS.
template <class D>
__device__ __forceinline__ void load_cold(const Lds &L, const Params &p, int env, int lane)
{
    const int A = D::A(p), NPs = D::NP(p) > 0 ? D::NP(p) : 1;
    const int hot16 = D::hot_bytes(p) / 16, cold16 = (D::rec_bytes(p) - D::hot_bytes(p)) / 16;
    u32x4 *dst = reinterpret_cast<u32x4 *>(L.rec) + hot16;
    if (uni(L.cnt[3]) != 0) {
        GAS const u32x4 *src = (GAS const u32x4 *)(G(p.state) + (size_t)env * D::rec_bytes(p)) + hot16;
        for (int o = lane; o < cold16; o += kLanes) dst[o] = src[o];
    } else {
        const int nd2 = 2 * (A * kK + NPs);   // dwords of the f64 fields; then wall_shape (-1), wall_age (0), pair_age (-1)
        for (int o = lane; o < cold16; o += kLanes) {
            u32x4 v;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int j = 4 * o + q - nd2;
                v[q] = (j >= 0 && (j < A * kK || (j >= 2 * A * kK && j < 2 * A * kK + NPs))) ? 0xFFFFFFFFu : 0u;
            }
            dst[o] = v;
        }
    }
    wave_sync();
}

template <class D>
__device__ __forceinline__ void load_state(const Lds &L, const Params &p, int env, int lane)
{
    GAS const u32x4 *src = (GAS const u32x4 *)(G(p.state) + (size_t)env * D::rec_bytes(p));
    u32x4 *dst = reinterpret_cast<u32x4 *>(L.rec);
    for (int o = lane; o < D::hot_bytes(p) / 16; o += kLanes) dst[o] = src[o];
    wave_sync();
    load_cold<D>(L, p, env, lane);
}

// The hot part in two halves, so that the HBM round trip overlaps the map staging: fetch into registers
// before stage_map (whose barrier keeps the loads in front of it), write to LDS after it.
struct StateRegs { u32x4 v; };   // 64 lanes x 16 B = 1 KB >= the largest hot part (A = 8: 784 B)
template <class D>
__device__ __forceinline__ void fetch_state(StateRegs &r, const Params &p, int env, int lane)
{
    GAS const u32x4 *src = (GAS const u32x4 *)(G(p.state) + (size_t)(env < 0 ? 0 : env) * D::rec_bytes(p));
    if (lane < D::hot_bytes(p) / 16) r.v = src[lane];
}
template <class D>
__device__ __forceinline__ void commit_state(const Lds &L, const StateRegs &r, const Params &p, int lane)
{
    u32x4 *dst = reinterpret_cast<u32x4 *>(L.rec);
    if (lane < D::hot_bytes(p) / 16) dst[lane] = r.v;
    wave_sync();
}

// LDS -> HBM.  cache_live (cnt[3]) is recomputed: the cold part goes out only while some arbiter is cached.
template <class D>
__device__ __forceinline__ void store_state(const Lds &L, const Params &p, int env, int lane)
{
    wave_sync();
    const int A = D::A(p);
    const bool mine = (lane < A * kK && L.wsh[lane] >= 0) || (lane < D::NP(p) && L.pag[lane] >= 0);
    const bool live = __ballot(mine) != 0ull;
    if (lane == 0) L.cnt[3] = live ? 1 : 0;
    wave_sync();
    GAS u32x4 *dst = (GAS u32x4 *)(G(p.state) + (size_t)env * D::rec_bytes(p));
    const u32x4 *src = reinterpret_cast<const u32x4 *>(L.rec);
    const int n16 = (live ? D::rec_bytes(p) : D::hot_bytes(p)) / 16;
    for (int o = lane; o < n16; o += kLanes) dst[o] = src[o];
}

#ifdef CAT_WAVE_SPREAD
// Diagnostic build only (-DCAT_WAVE_SPREAD, tools/wave_spread.py): the timeline of the last step_kernel launch on the 100 MHz realtime counter (one
// domain for the whole device).  g_wave_t, per wave: start, after the staging barrier, own front published, scheduler exit; shader clock at
// start / exit.  g_slot_t, per env slot: front start, publish, unit u start / end (2 + 2u, 3 + 2u; u < 5), write-back start / end (12, 13).
__device__ unsigned long long g_wave_t[8 * 65536];
__device__ unsigned long long g_slot_t[16 * 65536];
extern "C" int cat_debug_spread(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_t), sizeof(unsigned long long) * 8 * n) == hipSuccess ? 0 : -1;
}
extern "C" int cat_debug_slot_times(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slot_t), sizeof(unsigned long long) * 16 * n) == hipSuccess ? 0 : -1;
}
#define SSPREAD(slot_, i) do { if (kOneTick && lane0 == 0 && (i) < 14) g_slot_t[16 * (blockIdx.x * (blockDim.x / kLanes) + (slot_)) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define WSPREAD(i) do { if (kOneTick && lane0 == 0) g_wave_t[8 * (blockIdx.x * (blockDim.x / kLanes) + wave) + (i)] = ((i) < 4 || (i) > 5) ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter(); } while (0)
#else
#define WSPREAD(i) do {} while (0)
#define SSPREAD(slot_, i) do {} while (0)
#endif
// Workgroup control words (LDS, L.ctrl): lane 0 operates, the result is broadcast.  Relaxed accesses; the
// callers place the workgroup-scope release / acquire fences where data is handed over.
__device__ __forceinline__ int ctrl_add(int *w, int lane)   // fetch-and-increment
{
    int v = 0;
    if (lane == 0) v = __hip_atomic_fetch_add(w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return uni(v);
}
__device__ __forceinline__ void copy_snapshot(const Lds &L, int A, int lane)
{   // record order is pos vel vb tc leaf: the snapshot keeps pos[2A] tc[2A] leaf[4A]
    if (lane < 8 * A) L.spawn[lane] = L.pos[lane + (lane < 2 * A ? 0 : 4 * A)];
    wave_sync();
}

// ctrl word 2 of a slot: 0 = not published yet, else the number of its work units (ray chunks [+ Space.step])
__device__ __forceinline__ void publish_slot(const Lds &L, int wave, int lane, int n_units)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(&L.ctrl[4 * wave + 2], n_units, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Write-back of one finished slot tick: rewards (cop.py / thief.py), the counters of the state record, the record itself
// (store_rec: the one-tick kernels store it every tick, the resident rollout kernel only after its last tick) and every
// output.  env_out indexes the output buffers: the env slot, or row t * N + env of buffers with a leading T.
template <class D>
__device__ __forceinline__ void slot_writeback(const Lds &Ls, const Params &p, const LaunchArgs &la, int e_s, long long env_out, int lane,
                                               int tick, bool store_rec, int step2, int captured2, int timeout2, int rcount,
                                               GAS const float *cop_lut, GAS const float *thief_lut, PhaseClock &pc)
{
    LateOut late;
    rewards_and_positions<D>(Ls, p, la, lane, tick, captured2, timeout2, cop_lut, thief_lut, late);
    stage_shared_observations<D>(Ls, p, lane);
    await_reward(late);
    PHASE(pc, 17);
    const unsigned char term = (unsigned char)(captured2 || timeout2);
    if (lane == 0) {   // a slot that went through a reset (rcount >= 0) starts its new episode: base_env.py:350
        Ls.cnt[0] = step2; Ls.cnt[2] = rcount >= 0 ? 0 : term;
        if (rcount >= 0) Ls.cnt[1] = rcount;
    }
    if (store_rec) store_state<D>(Ls, p, e_s, lane);
    PHASE(pc, 18);
    emit_observations<D>(Ls, p, la, env_out, lane, tick, late);
    PHASE(pc, 19);
    if (tick && lane == 0) {
        if (la.out.terminated) la.out.terminated[env_out] = term;       // entity.py:146
        if (la.out.truncated) la.out.truncated[env_out] = (unsigned char)timeout2;  // :397
        if (la.out.winner) la.out.winner[env_out] = (signed char)(captured2 ? 0 : (timeout2 ? 1 : -1));  // :399-406
    }
}

// Large kernels whose inlined phases share one loop (run_units, the resident rollout's scheduler): whatever is invariant across
// the loop -- lane-derived LDS addresses, output pointers plus lane offsets, compare masks, fields of Params and of the launch
// arguments -- the compiler hoists in front of it and then keeps alive through every phase (first build of the rollout kernel:
// 141 spilled VGPRs, 760 B of scratch per lane; the one-tick kernel: 64 - 134 SGPRs spilled, scratch in the generic
// instantiation).  So each phase starts from opaque copies of its roots (lane id, parameter pointer, kernarg pointer) and
// re-derives what it needs, the workgroup's map / grid descriptors are re-read from an LDS copy by the phase that needs them
// (BlockDesc behind the control words), and nothing but the scheduler's own few scalars lives across phases.
// (opaque_v is only ever given a lane id: the range is handed back to the compiler, which otherwise unrolls every lane-strided
// loop -- the wide stores of the write-back -- for an unknown start: 283 global stores in the one-tick kernel instead of 27)
__device__ __forceinline__ int opaque_v(int v) { asm volatile("" : "+v"(v)); __builtin_assume((unsigned)v < (unsigned)kLanes); return v; }
typedef const LaunchArgs __attribute__((address_space(4))) *LaunchArgsK;   // the by-value launch arguments, in the kernarg segment
typedef const Params __attribute__((address_space(4))) *ParamsK;   // the parameter block is never written while a kernel runs: constant address space -> scalar loads
// kernarg layout of the three env kernels: [const Params *][LaunchArgs] (8-byte aligned)
__device__ __forceinline__ LaunchArgsK kernarg_launch_args()
{
    return (LaunchArgsK)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + 8);
}
// the workgroup's descriptors in LDS (written by stage_map before its barrier)
__device__ __forceinline__ const BlockDesc *block_desc_lds(char *smem, const Params &p, int W)
{
    return reinterpret_cast<const BlockDesc *>(smem + p.lds_map_bytes + 16 * W);
}
// What a kernel's prologue needs of the parameter block (env id, descriptors, state record, map staging, LDS carve), requested
// in ONE burst of scalar loads and pinned: left to itself the compiler loads each field where it is first used -- behind the
// prologue's branches -- and the launch starts with a chain of four dependent round trips to a cold scalar cache instead of two
// (measured: + 1 600 cycles in front of the map staging).  The copy lives in registers only (every field access is resolved at
// compile time); fields that are not listed here must not be read through it.  (Pinning the WHOLE block, so that the serial front
// after the barrier reads registers too, was built: 75 spilled SGPRs in the one-tick kernel.)
typedef const Prologue __attribute__((address_space(4))) *PrologueK;
__device__ __forceinline__ PrologueK kernarg_prologue()
{
    return (PrologueK)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + 8 + sizeof(LaunchArgs));
}
__device__ __forceinline__ Params prologue_params(PrologueK pk, int &uniform)
{
    Params q;
    q.lds_map_bytes = pk->lds_map_bytes; q.lds_env_bytes = pk->lds_env_bytes; q.lds_union_bytes = pk->lds_union_bytes; q.wpb = pk->wpb;
    q.A = pk->A; q.R = pk->R; q.NP = pk->NP; q.maxc = pk->maxc; q.n_cops = pk->n_cops; q.rec_bytes = pk->rec_bytes; q.hot_bytes = pk->hot_bytes;
    q.N = pk->N; uniform = pk->uniform; q.lds_pool_off = pk->lds_pool_off; q.pool_mask = pk->pool_mask; q.grp_rays = pk->grp_rays;
    q.work_env = pk->work_env; q.block_desc = pk->block_desc; q.state = pk->state;
    q.geo_f64 = pk->geo_f64; q.geo_i32 = pk->geo_i32; q.ray_dx = pk->ray_dx; q.ray_dy = pk->ray_dy;
    q.cop_lut = pk->cop_lut; q.thief_lut = pk->thief_lut;
    // ONE pin for all of them: the loads above are issued together and waited for once
    asm volatile("" : "+s"(q.lds_map_bytes), "+s"(q.lds_env_bytes), "+s"(q.lds_union_bytes), "+s"(q.wpb), "+s"(q.A), "+s"(q.R), "+s"(q.NP),
                      "+s"(q.n_cops), "+s"(q.rec_bytes), "+s"(q.hot_bytes), "+s"(q.N), "+s"(uniform), "+s"(q.lds_pool_off), "+s"(q.pool_mask), "+s"(q.grp_rays), "+s"(q.work_env), "+s"(q.block_desc), "+s"(q.state),
                      "+s"(q.geo_f64), "+s"(q.geo_i32), "+s"(q.ray_dx), "+s"(q.ray_dy), "+s"(q.cop_lut), "+s"(q.thief_lut));
    return q;
}
// member by member (a struct copy would go through the stack)
__device__ __forceinline__ void copy_desc(BlockDesc &d, const BlockDesc &s)
{
    d.md.S = s.md.S; d.md.P = s.md.P; d.md.A = s.md.A; d.md.n_regions = s.md.n_regions; d.md.f64_off = s.md.f64_off; d.md.i32_off = s.md.i32_off;
    d.md.cmax = s.md.cmax; d.md.PP = s.md.PP;
    d.gd.x0 = s.gd.x0; d.gd.y0 = s.gd.y0; d.gd.inv_cell = s.gd.inv_cell; d.gd.nx = s.gd.nx; d.gd.ny = s.gd.ny; d.gd.off_base = s.gd.off_base;
    d.gd.ent_base = s.gd.ent_base; d.gd.coff_base = s.gd.coff_base; d.gd.cent_base = s.gd.cent_base; d.gd.crow_base = s.gd.crow_base; d.gd.span = s.gd.span;
    d.gd.row_base = s.gd.row_base; d.gd.span_tick = s.gd.span_tick;
}
// The workgroup's env id (wave's slot) and descriptor: computed / read from the kernarg copy where the sim is uniform, else loaded.
// The descriptor's vector load from the kernarg segment is issued at once: it depends on nothing that is loaded.
__device__ __forceinline__ void prologue_env_desc(const Params &q, int uniform, int W, int wave, int &env, BlockDesc &bd0)
{
    copy_desc(bd0, *(const BlockDesc *)(const void *)&kernarg_prologue()->bd);
    const int e = blockIdx.x * W + wave;
    env = e < q.N ? e : -1;
    if (!uniform) {
        env = uni(q.work_env[e]);
        copy_desc(bd0, q.block_desc[blockIdx.x]);
    }
}

// The shared part of both kernels.  Units of a published slot, claimed in order by any wave of the workgroup:
// ray chunks 0 .. nchunks-1, then (tick only) Space.step.  A wave starts with its own slot.  The wave that completes
// a slot's last unit writes that slot back: rewards, state record and outputs to HBM.
// L.flags of a slot = {step_count to store, captured, timeout, reset_count to store or -1}.
template <class D>
__device__ __forceinline__ void run_units(const Params *pp0, LaunchArgsK lap0, char *smem, int W, int wave, int lane0, int tick, PhaseClock &pc)
{
    unsigned fin_mask = 0u;
    bool own_first = true;
    int *const ctrl0 = reinterpret_cast<int *>(smem + launder((ParamsK)pp0)->lds_map_bytes);   // one scalar, kept across the loop
    // fetched now, used at every write-back: the reward lookup then costs one global round trip, not two
    GAS const float *cop_lut = launder(G(launder((ParamsK)pp0)->cop_lut)), *thief_lut = launder(G(launder((ParamsK)pp0)->thief_lut));
    for (;;) {
        int slot, e_s, last_unit;
        {   // one LDS round trip for the whole workgroup: lane s < W reads the control words of slot s
            const int lane = opaque_v(lane0);
            int *const ctrl = ctrl0;
            int e_l = -1, nu_l = 0, cl_l = 0;
            if (lane < W) {
                e_l = ctrl[4 * lane + 3];   // written before the workgroup barrier
                nu_l = __hip_atomic_load(&ctrl[4 * lane + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                cl_l = __hip_atomic_load(&ctrl[4 * lane + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            const unsigned open = (unsigned)__ballot(e_l >= 0 && nu_l > 0 && cl_l < nu_l);      // published, units left to claim
            const unsigned unpublished = (unsigned)__ballot(e_l >= 0 && nu_l == 0);
            if (open == 0u) {
                if (unpublished == 0u) break;
                __builtin_amdgcn_s_sleep(8);   // an owner is still in its serial part
                continue;
            }
            // the own slot first, then the next open slot after the own index (spreads the helpers over the slots)
            if (own_first && ((open >> wave) & 1u)) slot = wave;
            else {
                const unsigned rot = wave == 0 ? open : ((open >> wave) | (open << (32 - wave)));
                slot = (wave + __builtin_ctz(rot)) & 31;   // bits >= W are never set (W <= 16)
            }
            own_first = false;
            slot = uni(slot);
            e_s = __builtin_amdgcn_readlane(e_l, slot);              // the scan already holds them
            last_unit = __builtin_amdgcn_readlane(nu_l, slot) - 1;
            if (slot != wave) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // another wave's slot
        }
        int c = ctrl_add(&ctrl0[4 * slot + 0], opaque_v(lane0));
        while (c <= last_unit) {
            const int lane = opaque_v(lane0);
            const Params &p = *(const Params *)launder((ParamsK)pp0);
            const LaunchArgs &la = *(const LaunchArgs *)launder(lap0);
            int *const ctrl = reinterpret_cast<int *>(smem + p.lds_map_bytes);
            const BlockDesc *const K = block_desc_lds(smem, p, W);
            const Lds Ls = carve<D>(p, smem, K->md, slot, wave);
            const int unit = c;
            if (unit < fan_units<D>(p)) {   // entity.py:143-144, base_env.py:388-390 / :334-344
                if constexpr (D::kFan == 1) fan_group<D>(Ls, p, la, K->gd, e_s, lane, uni(K->md.S), K->md.cmax, tick, unit, group_agents<D>(p), pc);
                else fan_chunk<D>(Ls, p, la, K->gd, e_s, lane, uni(K->md.S), K->md.cmax, tick, unit, pc);
            }
            else {
                PHASE(pc, 9);
                physics_env<D>(Ls, p, uni(K->md.S), lane, pc);                     // base_env.py:392
                PHASE(pc, 10);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the unit's LDS writes, before it counts as done
            // "done" and the next claim in one LDS round trip (if this was the slot's last unit the claim returns past the end)
            int d = 0;
            if (lane == 0) {
                d = __hip_atomic_fetch_add(&ctrl[4 * slot + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                c = __hip_atomic_fetch_add(&ctrl[4 * slot + 0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            d = uni(d); c = uni(c);
            if (d == last_unit) fin_mask |= 1u << slot;   // this wave completed the slot
        }
    }
    PHASE(pc, 16);
    if (fin_mask) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // the other waves' units of those slots
    PHASE(pc, 21);
    // ---- write-backs of the slots this wave completed: all output stores at the very end of the kernel
#ifdef CAT_PHASE_TIMING
    bool wb_first = true;
#endif
    while (fin_mask) {
        const int lane = opaque_v(lane0);
        const Params &p = *(const Params *)launder((ParamsK)pp0);
        const LaunchArgs &la = *(const LaunchArgs *)launder(lap0);
        int *const ctrl = reinterpret_cast<int *>(smem + p.lds_map_bytes);
        const BlockDesc *const K = block_desc_lds(smem, p, W);
        const int slot = uni(__builtin_ctz(fin_mask));
        fin_mask &= fin_mask - 1;
        const Lds Ls = carve<D>(p, smem, K->md, slot, wave);
        const int e_s = uni(ctrl[4 * slot + 3]);
        const int step2 = uni(Ls.flags[0]), captured2 = uni(Ls.flags[1]), timeout2 = uni(Ls.flags[2]), rcount = uni(Ls.flags[3]);
        PHASE(pc, 22);
#ifdef CAT_PHASE_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // diagnostic: what the wave still has in flight when it starts a write-back
        if (wb_first) PHASE(pc, 23); else PHASE(pc, 3);       // its first one / a further one (the stores of the one before)
        wb_first = false;
#endif
        slot_writeback<D>(Ls, p, la, e_s, (long long)e_s, lane, tick, true, step2, captured2, timeout2, rcount, cop_lut, thief_lut, pc);
    }
}

template <class D>
__device__ __forceinline__ void spawn_and_reset(const Lds &L, const Params &p, const LaunchArgs &la, const MapDesc &md,
                                                int env, unsigned rc, int lane);

// The serial front of one slot's tick (BaseEnv.step up to the observations, base_env.py:372-383): step count, termination on
// last tick's positions, the tick-start snapshot, Entity._perform_action with lane = agent, the in-kernel auto-reset of an
// episode that ends with this tick, and the per-agent ray-fan setup.  Leaves L.flags for the write-back and returns the
// number of work units to publish (ray-fan units [+ Space.step]).  act_pref: lane i's action when la.actions is set.
template <class D>
__device__ __forceinline__ int slot_front(const Lds &L, ParamsK pk, const LaunchArgs &la, const MapDesc &md, const GridDesc &gd,
                                          int env, int lane, int act_pref, unsigned long long synth_tick, int n_fan, PhaseClock &pc)
{
    // three sub-phases, each from a freshly laundered parameter pointer: what one has loaded does not stay alive through the next
    // (the rare auto-reset path inlines Space.step and the spawn sampling between the two common ones)
    int captured, timeout, step;
    SetupRow row;
    {
        const Params &p = *(const Params *)launder(pk);
        const int S = md.S, A = D::A(p);
        row = setup_row(L.pos, p, gd, lane, A);                             // the setup's global round trip, under the termination check and the actions
        step = uni(L.cnt[0]) + 1;                                           // :372
        captured = termination_captured<D>(L, p, S, lane);                     // :378
        timeout = (!captured && step >= p.max_step) ? 1 : 0;
        copy_snapshot(L, A, lane);

        // Entity._perform_action (entity.py:126-134), lane = agent.  (As a wave-uniform loop over the agents -- the
        // synthetic-action Philox rounds and the sqrt/divide chain of each agent one after the other, on the scalar
        // unit -- this was 6 us of the tick, in the part of the kernel every wave of the launch executes in step.)
        if (lane < A) {
            const int i = lane;
            const double m_inv = 1.0 / p.mass;
            int act;
            if (la.actions) act = act_pref;
            else { unsigned rnd[4]; philox_env(p, env, (unsigned)synth_tick, (unsigned)i, 0xAC710u, rnd); act = (int)(rnd[0] & 3u); }
            if ((unsigned)act > 3u) atomicOr(p.err_word, CAT_DEVERR_BAD_ACTION);   // applied as "no impulse", and flagged
            double jx = 0.0, jy = 0.0;
            if (act == 0) jx = -p.impulse; else if (act == 1) jy = p.impulse;
            else if (act == 2) jx = p.impulse; else if (act == 3) jy = -p.impulse;
            double vx = L.vel[2 * i] + jx * m_inv, vy = L.vel[2 * i + 1] + jy * m_inv;
            double len = sqrt(vx * vx + vy * vy);
            if (len > p.max_speed) { vx = vx / len * p.max_speed; vy = vy / len * p.max_speed; }
            L.vel[2 * i] = vx; L.vel[2 * i + 1] = vy;
        }
        wave_sync();
        PHASE(pc, 2);
    }
    int n_units, rcount = -1, step_store = step;
    {
        const Params &p = *(const Params *)launder(pk);
        n_units = n_fan + 1;
        if (la.auto_reset && (captured || timeout)) {
            // The episode ends with this tick and the caller wants the slot reset in the same call: the terminal
            // observations would be overwritten by the reset's (the rewards of a terminal tick are constants), so the
            // slot's ray chunks are those of the NEW episode.  The reset needs the stepped state (stale circle caches
            // and leaf bbs, quirk Q1), so the owner runs Space.step here instead of queueing it.
            physics_env<D>(L, p, uni(md.S), lane, pc);                // :392
            rcount = uni(L.cnt[1]) + 1;
            spawn_and_reset<D>(L, p, la, md, env, (unsigned)rcount, lane);   // base_env.py:286-352
            wave_sync();
            copy_snapshot(L, D::A(p), lane);
            n_units -= 1; step_store = 0;
            row = setup_row(L.fpos, p, gd, lane, D::A(p));           // the agents have moved to their spawn points
        }
    }
    {
        const Params &p = *(const Params *)launder(pk);
        agent_setup<D>(L, p, gd, lane, row);                         // entity.py:143-144, :388-390 (setup part)
        PHASE(pc, 4);
    }
    if (lane == 0) { L.flags[0] = step_store; L.flags[1] = captured; L.flags[2] = timeout; L.flags[3] = rcount; }
    return n_units;
}

// ------------------------------------------------------------------ resident rollout ----------
// T consecutive ticks of BaseEnv.step in ONE launch (the random-action phases of the reference's loops: src/driver.py:65-69,
// random_timesteps of src/configs/mappo_config.py:9; with an action tape: any fixed-policy replay).  The map is staged once, a
// slot's state record stays in its LDS env area for all T ticks (HBM sees it after the last one), and EVERY tick's outputs go to
// row t of caller buffers with a leading T.  Per tick the arithmetic is the one-tick step's (slot_front, the same work units, the same
// write-back), so the results equal T calls of cat_step_fused bit for bit.
//
// Scheduling.  The one-tick kernel pays, per launch, the dispatch floor, the map staging, the state record both ways and one slot's
// front -> fan -> write-back chain during which most waves of the workgroup wait (DESIGN: ~16 of 31 us).  Here the slots of a
// workgroup advance INDEPENDENTLY -- tick t + 1 of a slot starts as soon as its own tick t is written back, whatever the other
// slots are doing -- so in steady state every wave always finds a unit and only the last ticks of the launch have a tail.
// One control word per slot, W = epoch << 14 | units << 7 | next (epoch = tick + 1; 0 = nothing published yet; all ones =
// the slot has finished its T ticks): a wave claims unit `next` with a compare-and-swap on the whole word, so a claim made
// on a stale view (another epoch, another unit count) simply fails and the wave rescans.  The wave that completes a slot's
// last unit writes the tick back and runs the slot's NEXT front itself, then publishes the new epoch.
constexpr unsigned kRwFinished = 0xFFFFFFFFu;
__device__ __forceinline__ unsigned rw_make(int epoch, int n_units) { return ((unsigned)epoch << 14) | ((unsigned)n_units << 7); }
__device__ __forceinline__ int rw_next(unsigned w) { return (int)(w & 127u); }
__device__ __forceinline__ int rw_units(unsigned w) { return (int)((w >> 7) & 127u); }
__device__ __forceinline__ int rw_epoch(unsigned w) { return (int)(w >> 14); }

// LDS-only workgroup fences: the units hand LDS data from wave to wave; global stores of an earlier write-back that are still
// in flight need not be waited for (a fence over every address space would sit on their acknowledgements at every unit).
__device__ __forceinline__ void lds_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); }
__device__ __forceinline__ void lds_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }

template <class D, bool kOneTick>
__device__ __forceinline__ void rollout_body(const Params *__restrict__ pp0, const LaunchArgs &la0)
{
    extern __shared__ __align__(16) char smem[];
    const int lane0 = threadIdx.x % kLanes;
    const int wave = uni(threadIdx.x / kLanes);
    const LaunchArgsK lap0 = kernarg_launch_args();
    PhaseClock pc;
    WSPREAD(0); WSPREAD(4);
    int env, T, W;
    GAS const float *lut_c, *lut_t;
    {   // ---- prologue: descriptors -> LDS, control words, state record -> LDS, map staging
        int uniform;
        const Params q = prologue_params(kernarg_prologue(), uniform);   // the pre-barrier part reads this register copy
        const Params &p = *(const Params *)(ParamsK)pp0;
        lut_c = G(q.cop_lut); lut_t = G(q.thief_lut);          // four scalars kept for the write-backs: the reward lookup is then one round trip
        const int lane = lane0;
        W = uni((int)(blockDim.x / kLanes));
        T = kOneTick ? 1 : la0.T;
        WSPREAD(6);   // the parameter burst has arrived
        int *const ctrl = reinterpret_cast<int *>(smem + q.lds_map_bytes);
        BlockDesc bd0;
        prologue_env_desc(q, uniform, W, wave, env, bd0);
        const MapDesc &md0 = bd0.md;
#ifdef CAT_WAVE_SPREAD
        { int e_ = env, s_ = md0.S; asm volatile("" : "+s"(e_), "+v"(s_)); WSPREAD(7); }   // env id and descriptor have arrived
#endif
        // control words of slot `wave`: claim word (above), units done in this epoch, -, env id
        if (lane < 4) ctrl[4 * wave + lane] = lane == 3 ? env : ((lane == 0 && env < 0) ? (int)kRwFinished : 0);
        StateRegs sregs;
        fetch_state<D>(sregs, q, env, lane);
        stage_map<D>(q, smem, md0, &bd0, const_cast<BlockDesc *>(block_desc_lds(smem, q, W)));   // ends with the workgroup barrier
        PHASE(pc, 0);
        WSPREAD(1);
        if (env >= 0) {
            const Lds L = carve<D>(q, smem, md0, wave, wave);
            commit_state<D>(L, sregs, q, lane);
            load_cold<D>(L, p, env, lane);
            PHASE(pc, 1);
        }
    }
    int pend = env >= 0 ? wave : -1, pend_t = 0;   // the slot whose next front this wave is to run, and its tick
    int hint = wave;                                // the slot this wave worked on last: looked at first
    for (;;) {
        if (pend >= 0) {   // ---- the serial front of (slot pend, tick pend_t), then its units are published
            const int lane = opaque_v(lane0);
            const Params &p = *(const Params *)launder((ParamsK)pp0);
            const LaunchArgs &la = *(const LaunchArgs *)launder(lap0);
            int *const ctrl = reinterpret_cast<int *>(smem + p.lds_map_bytes);
            const BlockDesc *const K = block_desc_lds(smem, p, W);   // the workgroup's descriptors, in LDS
            const int slot = pend, t = pend_t;
            pend = -1;
            SSPREAD(slot, 0);
            const int e_s = uni(ctrl[4 * slot + 3]);
            const Lds Ls = carve<D>(p, smem, K->md, slot, wave);
            int ap = 0;
            if (la.actions && lane < D::A(p)) ap = la.actions[((size_t)t * p.N + e_s) * D::A(p) + lane];
            if (lane == 0) ctrl[4 * slot + 1] = 0;
            const int n_fan = fan_units<D>(p, unit_span<D, kOneTick>(p, K->gd));
#ifndef CAT_ABL_NOFRONT
            const int n2 = slot_front<D>(Ls, (ParamsK)pp0, la, K->md, K->gd, e_s, lane, ap, la.synth_tick + (unsigned long long)t, n_fan, pc);
#else
            const int n2 = n_fan + 1;
            if (lane == 0) { Ls.flags[0] = 1; Ls.flags[1] = 0; Ls.flags[2] = 0; Ls.flags[3] = -1; }
            (void)ap;
#endif
            lds_release();
            // (publishing with unit 0 already claimed for this wave, and re-claiming the slot tick's next unit without a scan, were
            // built: labyrinth T = 64 26.7 us per tick against 20.5 -- waves then stay on their slots and "help the hindmost" is gone)
            if (lane == 0) __hip_atomic_store((unsigned *)&ctrl[4 * slot], rw_make(t + 1, n2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            hint = slot;
            PHASE(pc, 3);
            WSPREAD(2); SSPREAD(slot, 1);
        }
        int slot, unit, n_units, t;
        {   // ---- look for an open unit and claim it
            const int lane = opaque_v(lane0);
            int *const ctrl = reinterpret_cast<int *>(smem + launder((ParamsK)pp0)->lds_map_bytes);
            unsigned w_l = kRwFinished;
            if (lane < W) w_l = __hip_atomic_load((unsigned *)&ctrl[4 * lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned open = (unsigned)__ballot(rw_next(w_l) < rw_units(w_l));
            if (open == 0u) {
                // leave when nothing can be published any more: every slot has finished its T ticks -- with one tick per launch, when
                // every slot HAS published (its units are all claimed; the waves running them write it back): a wave that stayed would
                // only spin on the control words beside waves that still compute
                if (__ballot(kOneTick ? (w_l == 0u) : (w_l != kRwFinished)) == 0ull) break;
                __builtin_amdgcn_s_sleep(4);                        // fronts / write-backs under way on other waves
                PHASE(pc, 21);
                continue;
            }
            // Which open slot: the one FURTHEST BEHIND (lowest epoch), ties going to the slot this wave worked on last and then round
            // the ring from it.  With "own slot first" every slot advances at its own pace -- envs differ in work per tick -- and
            // over T ticks the slots of a workgroup drift apart: the launch then ends on its slowest slots, three units wide, while
            // the other waves idle (10 % of all wave time at T = 64).  Helping the hindmost keeps the slots together.
            unsigned key = 0xFFFFFFFFu;
            if (rw_next(w_l) < rw_units(w_l)) key = ((unsigned)rw_epoch(w_l) << 5) | (unsigned)((lane - hint) & 31);
#define CAT_ROW_MIN(SH) { const unsigned o_ = (unsigned)__builtin_amdgcn_update_dpp((int)key, (int)key, 0x120 + SH, 0xF, 0xF, false); key = o_ < key ? o_ : key; }
            CAT_ROW_MIN(8) CAT_ROW_MIN(4) CAT_ROW_MIN(2) CAT_ROW_MIN(1)   // minimum over the 16 lanes of the row: row_ror by 8, 4, 2, 1
#undef CAT_ROW_MIN
            slot = uni((hint + (int)(key & 31u)) & 31);   // lane 0's row holds slots 0 .. 15
            const unsigned wv = (unsigned)__builtin_amdgcn_readlane((int)w_l, slot);
            unsigned seen = wv;
            if (lane == 0)
                __hip_atomic_compare_exchange_strong((unsigned *)&ctrl[4 * slot], &seen, wv + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((unsigned)uni((int)seen) != wv) continue;   // someone else took it (or the epoch moved on): look again
            lds_acquire();
            hint = slot;
            unit = rw_next(wv); n_units = rw_units(wv); t = rw_epoch(wv) - 1;
            PHASE(pc, 22);
        }
        bool fin;
        {   // ---- the unit: a part of the slot's ray fan (entity.py:143-144, base_env.py:388-390) or its Space.step (base_env.py:392)
            const int lane = opaque_v(lane0);
            const Params &p = *(const Params *)launder((ParamsK)pp0);
            const LaunchArgs &la = *(const LaunchArgs *)launder(lap0);
            int *const ctrl = reinterpret_cast<int *>(smem + p.lds_map_bytes);
            const BlockDesc *const K = block_desc_lds(smem, p, W);
            const int e_s = uni(ctrl[4 * slot + 3]);
            const Lds Ls = carve<D>(p, smem, K->md, slot, wave);
            const long long eo = (long long)t * p.N + e_s;   // row of the [T, N, ...] output buffers
            const int gsz = unit_span<D, kOneTick>(p, K->gd);
            SSPREAD(slot, 2 + 2 * unit);
            if (unit < fan_units<D>(p, gsz)) {
#ifndef CAT_ABL_NOFAN      // diagnostic builds: a phase compiled out, for instruction counts by difference (tools/ablate_rollout.sh)
                if constexpr (D::kFan == 1) fan_group<D>(Ls, p, la, K->gd, eo, lane, uni(K->md.S), K->md.cmax, 1, unit, gsz, pc);
                else {
                    const int nch = D::A(p) * ((D::R(p) + kLanes - 1) / kLanes), c0 = unit * gsz;
                    unsigned chunks = 1u;   // one chunk per unit: fan_chunk; several: fan_slot, which hands back what its item list cannot hold
                    if (D::kFixed && gsz > 1) chunks = fan_slot<D>(Ls, p, la, K->gd, eo, lane, uni(K->md.S), K->md.cmax, 1, c0, nch - c0 < gsz ? nch - c0 : gsz, pc);
                    while (chunks) {
                        const int q = uni(__builtin_ctz(chunks));
                        chunks &= chunks - 1;
                        fan_chunk<D>(Ls, p, la, K->gd, eo, lane, uni(K->md.S), K->md.cmax, 1, c0 + q, pc);
                    }
                }
#endif
            } else {
                PHASE(pc, 9);
#ifndef CAT_ABL_NOPHYS
                physics_env<D>(Ls, p, uni(K->md.S), lane, pc);
#endif
                PHASE(pc, 10);
            }
            lds_release();   // the unit's LDS writes, before it counts as done
            SSPREAD(slot, 3 + 2 * unit);
            int d = 0;
            if (lane == 0) d = __hip_atomic_fetch_add(&ctrl[4 * slot + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            fin = uni(d) == n_units - 1;
            PHASE(pc, 23);
        }
        if (!fin) continue;
        {   // ---- this wave completed the slot's tick t: write it back; the slot's next front is this wave's next job
            const int lane = opaque_v(lane0);
            const Params &p = *(const Params *)launder((ParamsK)pp0);
            const LaunchArgs &la = *(const LaunchArgs *)launder(lap0);
            int *const ctrl = reinterpret_cast<int *>(smem + p.lds_map_bytes);
            const BlockDesc *const K = block_desc_lds(smem, p, W);
            lds_acquire();
            PHASE(pc, 16);
            SSPREAD(slot, 12);
            const int e_s = uni(ctrl[4 * slot + 3]);
            const Lds Ls = carve<D>(p, smem, K->md, slot, wave);
            const long long eo = (long long)t * p.N + e_s;
            const int step2 = uni(Ls.flags[0]), captured2 = uni(Ls.flags[1]), timeout2 = uni(Ls.flags[2]), rcount = uni(Ls.flags[3]);
            const bool last = t + 1 >= T;
#ifndef CAT_ABL_NOWB
            slot_writeback<D>(Ls, p, la, e_s, eo, lane, 1, last, step2, captured2, timeout2, rcount, lut_c, lut_t, pc);
#else
            (void)eo; (void)step2; (void)captured2; (void)timeout2; (void)rcount; (void)la;
#endif
            wave_sync();   // the write-back has read the slot's staging and flags; the next front overwrites them
            SSPREAD(slot, 13);
            if (!last) { pend = slot; pend_t = t + 1; }
            else if (lane == 0) __hip_atomic_store((unsigned *)&ctrl[4 * slot], kRwFinished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    PHASE(pc, 11);
    WSPREAD(3); WSPREAD(5);
    pc.flush(lane0);
}

template <class D>
__global__ __launch_bounds__(kMaxWaves *kLanes) void rollout_kernel(const Params *__restrict__ pp0, const LaunchArgs la0, const Prologue)
{
    rollout_body<D, false>(pp0, la0);
}

// BaseEnv.step (base_env.py:354-413), ONE tick per launch: what cat_step / cat_step_fused launch.  The same scheduler with T fixed
// at 1 at compile time (the launch arguments are cat_step's).  Rounds 1 - 3 had a kernel of its own for this (tick_kernel: wave w
// owned slot w for the front, units claimed own-slot-first, every write-back after the unit loop); rebuilt on this round's
// per-phase roots it measured 1 - 3 % behind this one on every BASELINE shape (labyrinth x4096 33.0 against 32.0 us, agh-map 65.2 /
// 64.7, 3v2 x8192 101.4 / 101.5, five maps x16384 173.8 / 169.8, 90 rays 41.8 / 41.1; the round-3 binary: 31.9 / 64.9 / 100.4 / 171.7 /
// 41.6) and was removed.
template <class D>
__global__ __launch_bounds__(kMaxWaves *kLanes) void step_kernel(const Params *__restrict__ pp0, const LaunchArgs la0, const Prologue)
{
    rollout_body<D, true>(pp0, la0);
}

#ifdef CAT_WB_COUNTS
__device__ unsigned long long g_wb_counts[8];
extern "C" int cat_debug_wb_counts(unsigned long long *out8, int reset)
{
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wb_counts), sizeof h) != hipSuccess) return -1;
    for (int i = 0; i < 8; i++) out8[i] = h[i];
    if (reset) { memset(h, 0, sizeof h); if (hipMemcpyToSymbol(HIP_SYMBOL(g_wb_counts), h, sizeof h) != hipSuccess) return -1; }
    return 0;
}
#endif
// ------------------------------------------------------------------ pooled ray fan -----------
template <class T> __device__ __forceinline__ T *slot_ptr(T *p0, int sl, int envb) { return (T *)((char *)const_cast<typename std::remove_const<T>::type *>(p0) + sl * envb); }
// step_kernel_pooled / rollout_kernel_pooled (light maps whose rays fit the pool: wpb * A * R <= 4096).  In the unit form above a slot's
// fan runs as rounds of ITS OWN active rays -- the labyrinth's two units hold 52 and 26 rays: rounds cost the same at 26 lanes as at
// 64 (tools/wave_spread.py: the thief's fan 5.4 us, the two cops' 6.8).  Here a slot's front sorts its rays itself (pool_sort: a ray with
// no candidate gets EMPTY at once, the others become 8-byte entries -- row word | slot, agent, ray, cone mask -- of ONE ring of LDS per
// workgroup), and any wave takes the next 64 entries whatever slots they come from (pool_round: the round body of fan_group with the
// slot per lane).  The per-ray arithmetic is fan_group's, so the results are bit-identical.  A slot's tick is complete when its
// A * R rays and its Space.step have been counted (ctrl word 1); the wave that counts the last writes it back.
constexpr unsigned kPoolValid = 0x80000000u;
constexpr double kPoolEmptyRows = 0.05;   // cat_create: the pooled one-tick kernel serves a sim whose candidate rows around the spawn points are empty at least this often
#ifndef CAT_POOL_ROUND
#define CAT_POOL_ROUND 60
#endif
#ifndef CAT_POOL_MIN_PARTIAL
#define CAT_POOL_MIN_PARTIAL 40
#endif
#ifndef CAT_POOL_PATIENCE
#define CAT_POOL_PATIENCE 3
#endif
constexpr int kPoolRound = CAT_POOL_ROUND;            // rays of a full round
constexpr int kPoolMinPartial = CAT_POOL_MIN_PARTIAL; // a wave with nothing else to do takes fewer than a full round only from this many on ...
constexpr int kPoolPatience = CAT_POOL_PATIENCE;      // ... or after this many idle looks (fronts under way will add to the ring; the end of a launch drains it)
#ifdef CAT_FAULT_INJECT
constexpr int kSpinLimit = 1 << 14;   // the fault-injection build reaches its limits quickly
#else
constexpr int kSpinLimit = 1 << 22;   // a ring entry that never arrives / a lost wake-up: leave with CAT_DEVERR_SCHEDULER instead of hanging the device
#endif
__device__ __forceinline__ int *pool_ctl(char *smem, const Params &p, int W) { return reinterpret_cast<int *>(smem + p.lds_map_bytes + 16 * W + kWgConstBytes - 8); }   // head, tail
static_assert(sizeof(BlockDesc) <= kWgConstBytes - 8, "the pool counters live behind the BlockDesc");

// entry index (free-running 32-bit counter) -> position in the ring.  kExact: the capacity is not a power of two (a compile-time property of the
// kernel instantiation: with both paths behind a run-time test the headline shape lost 1 %)
template <bool kExact>
__device__ __forceinline__ int ring_pos(const Params &p, unsigned i)
{
    if constexpr (!kExact) return (int)(i & (unsigned)p.pool_mask);
    const unsigned t = __umulhi(p.pool_magic, i);
    const unsigned q = (t + ((i - t) >> 1)) >> p.pool_shift;
    return (int)(i - q * (unsigned)(p.pool_mask + 1));
}

// The rays of one slot (its front just ran agent_setup): EMPTY observations for the candidate-less ones, ring entries for the others.
// Returns the number of rays resolved here.  env: the slot's row of the output buffers (hit_shape only).
template <class D, bool kExact>
__device__ __forceinline__ int pool_sort(const Lds &L, const Params &p, const LaunchArgs &la, const GridDesc &gd, long long env, int slot, int lane,
                                         int *pctl, unsigned long long *pool)
{
    const int A = D::A(p), R = D::R(p);
    const unsigned d_empty = f64_to_f16(p.ray_length);  // np.full(R, ray_length, float16) entity.py:200
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int cpa = (R + kLanes - 1) / kLanes, nch = A * cpa;
    const int my_cell = lane < A ? L.acell[lane] : -1;
    const int my_dk0 = lane < A * A ? L.dk0[lane] : 0, my_dcnt = lane < A * A ? L.dcnt[lane] : 0;
    int n_res = 0;
    for (int c0 = 0; c0 < nch; c0 += 4) {   // four chunks at a time: their packed rows are requested together
        unsigned wrow[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (c0 + q < nch) {
                const int i = (c0 + q) / cpa, k = ((c0 + q) - i * cpa) * kLanes + lane;
                const int cell = __builtin_amdgcn_readlane(my_cell, i);
                const size_t r = (cell < 0 || k >= R) ? 0 : (size_t)cell * R + k;
                wrow[q] = ((GAS const unsigned *)G(p.grid_rows))[gd.row_base + r];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (c0 + q < nch) {
                const int i = (c0 + q) / cpa, k = ((c0 + q) - i * cpa) * kLanes + lane;
                const int cell = __builtin_amdgcn_readlane(my_cell, i);
                const bool in = k < R;
                const unsigned rowv = (in && cell >= 0) ? wrow[q] : 0u;   // non-zero: the ray has candidate walls
                unsigned dynmask = 0;
                if (in)
                    for (int j = 0; j < A; j++) {
                        if (j == i) continue;
                        const int dc = __builtin_amdgcn_readlane(my_dcnt, i * A + j) & 0xFFFF, dk = __builtin_amdgcn_readlane(my_dk0, i * A + j);
                        int rel = k - dk; if (rel < 0) rel += R;
                        if (rel < dc) dynmask |= 1u << j;
                    }
                bool act = rowv != 0u || dynmask != 0u;
                const unsigned long long m = __ballot(act);
                const int n = __popcll(m);
                int base = 0;
                if (n) {
                    if (lane == 0) base = __hip_atomic_fetch_add(&pctl[1], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    base = uni(base);
                }
                unsigned long long *const e = &pool[act ? ring_pos<kExact>(p, (unsigned)(base + __popcll(m & lt_mask))) : 0];
#ifdef CAT_FAULT_INJECT   // diagnostic build only (tests/test_gpu_fault_injection.py): the first entry of slot 0 is reserved and never written
                if (slot == 0 && c0 + q == 0 && n && lane == (int)__builtin_ctzll(m)) act = false;
#endif
                // The ring holds at most wpb * A * R entries that are not yet counted, but an entry counts as read only once the wave that claimed its
                // round has loaded it: the position must read 0 (cleared by that reader) before a new entry goes there.  True at the first look in every
                // run observed; the wait makes it an invariant instead of a matter of timing (bounded like every wait of the scheduler).
                for (int spins = 0; __ballot(act && __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0ull) != 0ull;)
                    if (++spins >= kSpinLimit) { if (lane == 0) atomicOr(p.err_word, CAT_DEVERR_SCHEDULER); break; }
                if (act) {
                    const unsigned meta = kPoolValid | ((unsigned)slot << 24) | ((unsigned)i << 16) | (dynmask << 8) | (unsigned)k;
                    __hip_atomic_store(e, ((unsigned long long)meta << 32) | rowv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if (in && (rowv == 0u && dynmask == 0u)) {   // nothing along this ray: its observation is final
                    const int o = i * R + k;
                    L.od[o] = (unsigned short)d_empty;
                    L.ot[o] = (unsigned char)CAT_EMPTY;
                    if (la.out.hit_shape) la.out.hit_shape[(size_t)env * A * R + o] = -1;  // parity/debug only
                }
                n_res += __popcll(__ballot(in && (rowv == 0u && dynmask == 0u)));
            }
        }
    }
    return n_res;
}

// One round: entries [base, base + n) of the ring, n <= 64, lane = entry.  L0: slot 0's view with the calling wave's scratch union.
// Returns the mask of the slots whose tick this round completed.
template <class D, bool kExact>
__device__ __forceinline__ unsigned pool_round(const Lds &L0, const Params &p, const LaunchArgs &la, int S, float cmax, int base, int n, int lane, int *ctrl,
                               unsigned long long *pool, PhaseClock &pc)
{
    const int A = D::A(p), R = D::R(p), envb = p.lds_env_bytes;
    const double r2 = p.ray_radius;
    const unsigned d_empty = f64_to_f16(p.ray_length);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int gate = launder(uni(p.gate)), n_cops = launder(uni(D::n_cops(p)));
    const double wall_r = launder(p.wall_r), rc = launder(p.rc);
    const int idb = launder(uni(p.row_id_bits)), cmul = launder(uni(p.row_cnt_mul));   // four-byte rows: fields of idb bits = id + 1 (finalize_rows)
    auto row_count = [&](unsigned w) -> int { return w ? (((31 - __builtin_clz(w)) * cmul) >> 16) + 1 : 0; };
    bool on = lane < n;
    unsigned w0 = 0u, meta = 0u;
    if (on) {   // the entry may still be on its way from the front that reserved it
        unsigned long long *e = &pool[ring_pos<kExact>(p, (unsigned)(base + lane))];
        unsigned long long v = 0ull;
        int spins = 0;
        do { v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (!(v >> 63) && ++spins < kSpinLimit);
        if (!(v >> 63)) { atomicOr(p.err_word, CAT_DEVERR_SCHEDULER); on = false; }
        __hip_atomic_store(e, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        w0 = (unsigned)v; meta = (unsigned)(v >> 32);
    }
    lds_acquire();
    const int s = (int)((meta >> 24) & 15u), i = (int)((meta >> 16) & 7u), k = (int)(meta & 255u);
    const unsigned dynmask = on ? ((meta >> 8) & 255u) : 0u;
    if (!on) w0 = 0u;
    unsigned *const rlist = L0.arow;   // [64] this round's entries (slot, agent, ray), read back by the item stage
    rlist[lane] = meta;
    const double *const fpos = slot_ptr(L0.fpos, s, envb), *const ftc = slot_ptr(L0.ftc, s, envb), *const fleaf = slot_ptr(L0.fleaf, s, envb);
    const double2 org = *reinterpret_cast<const double2 *>(fpos + 2 * i);     // fresh body.position (entity.py:186)
    const double ax = org.x, ay = org.y;
    const int cnt_w = row_count(w0);
    const int cnt = cnt_w + __popc(dynmask);
    double rdx, rdy, rix, riy;
    {
        const double bx = ax + L0.rayd[2 * k], by = ay + L0.rayd[2 * k + 1];  // entity.py:191-193
        rdx = bx - ax; rdy = by - ay; rix = 1.0 / rdx; riy = 1.0 / rdy;
    }
    double best_a = 1.0;
    int best_fi = -1;   // id << 6 | feature of the accepted item
    int jj0 = 0;
    wave_sync();
    PHASE(pc, 20);
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
                if (gate) tbb = bb_segment_query((id < S) ? (L0.bb + kBB * id) : (fleaf + 4 * (id - S)), ax, ay, rdx, rdy, rix, riy);
            }
            const bool live = has && tbb < best_a;
            const unsigned long long m = __ballot(live);
            const int c = __popcll(m);
            if (n_items + c > kItemCap) break;
            int t = 0xFFFF;
            if (live) {
                t = n_items + __popcll(m & lt_mask);
                L0.itm[t] = (unsigned short)(lane | (id << 6));
                L0.itbb[t] = tbb;
            }
            L0.itemidx[(jj - jj0) * kLanes + lane] = (unsigned short)t;
            n_items += c;
        }
        const int jj1 = jj;
        wave_sync();
        PHASE(pc, 5);
        // ---- one item per lane
        for (int t0 = 0; t0 < n_items; t0 += kLanes) {
            const int t = t0 + lane;
            if (t < n_items) {
                const int d = L0.itm[t];
                const int il = d & 63, id = (d >> 6) & 63;
                const unsigned m2 = rlist[il];
                const int s2 = (int)((m2 >> 24) & 15u), ia = (int)((m2 >> 16) & 7u), k2 = (int)(m2 & 255u);
                const double *const fpos2 = slot_ptr(L0.fpos, s2, envb), *const ftc2 = slot_ptr(L0.ftc, s2, envb);
                const double2 o2 = *reinterpret_cast<const double2 *>(fpos2 + 2 * ia);
                const double cbx = o2.x + L0.rayd[2 * k2], cby = o2.y + L0.rayd[2 * k2 + 1];
                double alpha = 2.0;   // 2.0 = no hit (never below a best alpha <= 1)
                int feat = 0;
                {
                    const bool wall = id < S;
                    const int j = wall ? 0 : id - S;
                    const int *const an2 = slot_ptr(L0.anear, s2, envb);
                    const bool inside = wall ? (id == an2[2 * ia] || id == an2[2 * ia + 1]) : ((((unsigned)slot_ptr(L0.adn, s2, envb)[ia] >> j) & 1u) != 0u);
                    if (inside) { alpha = 0.0; feat = kFeatNear; }
                    else {   // an accepted circle hit at alpha == 1 could never beat the initial best of 1: "t < 1" is equivalent
                        int f;
                        poly_query_feat(L0, cmax, wall, wall ? id : 0, wall ? wall_r : rc, ftc2[2 * j], ftc2[2 * j + 1], o2.x, o2.y, cbx, cby, r2, alpha, f);
                        feat = f < 0 ? 0 : f;
                    }
                }
                L0.ialpha[t] = alpha; L0.itm[t] = (unsigned short)((id << 6) | feat);
            }
        }
        wave_sync();
        PHASE(pc, 6);
        // ---- each ray walks its own items in index order
        for (int q = jj0; q < jj1; q++) {
            const int t = cnt > q ? (int)L0.itemidx[(q - jj0) * kLanes + lane] : 0xFFFF;
            if (t != 0xFFFF) {
                const double al = L0.ialpha[t];
                if (al < best_a && L0.itbb[t] < best_a) { best_a = al; best_fi = L0.itm[t]; }   // t_exit == best alpha
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
        const double bx = ax + L0.rayd[2 * k], by = ay + L0.rayd[2 * k + 1];
        best = best_fi >> 6;
        const int f = best_fi & 63;
        const double t = best_a;
        double px = bx, py = by;  // alpha = 0 hits keep the segment end as their point
        if (f != kFeatNear) {
            const bool wall = best < S;
            const int fc = wall ? L0.fc[best] : 0, first = fc & 0xFFFF, count = fc >> 16;
            if (wall && f < count) {
                const double2 nn = *reinterpret_cast<const double2 *>(L0.planes + 8 * (first + f));
                px = (ax * (1.0 - t) + bx * t) - nn.x * r2;
                py = (ay * (1.0 - t) + by * t) - nn.y * r2;
            } else {   // corner circle of the hull or the agent's circle: the same formula around a different centre
                double2 v = *reinterpret_cast<const double2 *>(L0.planes + 8 * (first + (wall ? f - count : 0)) + 2);
                if (!wall) { v.x = ftc[2 * (best - S)]; v.y = ftc[2 * (best - S) + 1]; }
                circle_hit_point(v.x, v.y, ax, ay, bx, by, t, r2, px, py);
            }
        }
        d16 = obs_distance_f16(px, py, ax, ay);
        ty = (best < S) ? CAT_WALL : ((best - S) >= n_cops ? CAT_THIEF : CAT_COP);
    }
    if (on) {  // observations go to the slot's staging in LDS; one coalesced burst to HBM at its write-back
        const int o = i * R + k;
        slot_ptr(L0.od, s, envb)[o] = (unsigned short)d16;
        slot_ptr(L0.ot, s, envb)[o] = (unsigned char)ty;
        if (la.out.hit_shape) {   // parity/debug only: row (tick, env) of the slot
            const long long eo = (long long)(rw_epoch((unsigned)ctrl[4 * s]) - 1) * p.N + ctrl[4 * s + 3];
            la.out.hit_shape[(size_t)eo * A * R + o] = best;
        }
        const unsigned want = i < n_cops ? CAT_THIEF : CAT_COP;
        // min over the agent's rays (other rounds add theirs); non-negative f16: bit order = value order
        if (ty == want) __hip_atomic_fetch_min(&slot_ptr(L0.dmin, s, envb)[i], d16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    PHASE(pc, 8);
    lds_release();   // this round's observations, before its rays count as done
    bool fin = false;
    if (on) fin = __hip_atomic_fetch_add(&ctrl[4 * s + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1 == A * R + 1;
    unsigned long long fm = __ballot(fin);
    unsigned done = 0u;
    while (fm) {
        const int l = __builtin_ctzll(fm);
        fm &= fm - 1;
        done |= 1u << __builtin_amdgcn_readlane(s, l);
    }
    return done;
}

template <class D, bool kOneTick, bool kExact>
__device__ __forceinline__ void rollout_body_pool(const Params *__restrict__ pp0, const LaunchArgs &la0)
{
    extern __shared__ __align__(16) char smem[];
    const int lane0 = threadIdx.x % kLanes;
    const int wave = uni(threadIdx.x / kLanes);
    const LaunchArgsK lap0 = kernarg_launch_args();
    PhaseClock pc;
    WSPREAD(0); WSPREAD(4);
    int env, T, W;
    GAS const float *lut_c, *lut_t;
    {   // ---- prologue: descriptors -> LDS, control words, empty ring, state record -> LDS, map staging
        int uniform;
        const Params q = prologue_params(kernarg_prologue(), uniform);   // the pre-barrier part reads this register copy
        const Params &p = *(const Params *)(ParamsK)pp0;
        lut_c = G(q.cop_lut); lut_t = G(q.thief_lut);
        const int lane = lane0;
        W = uni((int)(blockDim.x / kLanes));
        T = kOneTick ? 1 : la0.T;
        WSPREAD(6);   // the parameter burst has arrived
        int *const ctrl = reinterpret_cast<int *>(smem + q.lds_map_bytes);
        BlockDesc bd0;
        prologue_env_desc(q, uniform, W, wave, env, bd0);
        const MapDesc &md0 = bd0.md;
#ifdef CAT_WAVE_SPREAD
        { int e_ = env, s_ = md0.S; asm volatile("" : "+s"(e_), "+v"(s_)); WSPREAD(7); }   // env id and descriptor have arrived
#endif
        // control words of slot `wave`: claim word (epoch << 14 | units << 7 | next; the only claimable unit is Space.step), rays + units counted
        // in this tick, the tick its next front runs, env id
        if (lane < 4) ctrl[4 * wave + lane] = lane == 3 ? env : ((lane == 0 && env < 0) ? (int)kRwFinished : 0);
        {
            u32x4 *pz = reinterpret_cast<u32x4 *>(smem + q.lds_pool_off);
            const u32x4 z = {0u, 0u, 0u, 0u};
            for (int o = threadIdx.x; o < (q.pool_mask + 1) / 2; o += blockDim.x) pz[o] = z;
            if (threadIdx.x == 0) { int *pc2 = pool_ctl(smem, q, W); pc2[0] = 0; pc2[1] = 0; }
        }
        StateRegs sregs;
        fetch_state<D>(sregs, q, env, lane);
        stage_map<D>(q, smem, md0, &bd0, const_cast<BlockDesc *>(block_desc_lds(smem, q, W)));   // ends with the workgroup barrier
        PHASE(pc, 0);
        WSPREAD(1);
        if (env >= 0) {
            const Lds L = carve<D>(q, smem, md0, wave, wave);
            commit_state<D>(L, sregs, q, lane);
            load_cold<D>(L, p, env, lane);
            PHASE(pc, 1);
        }
    }
    unsigned todo = env >= 0 ? 1u << wave : 0u;   // slots whose next front this wave is to run (their tick: ctrl word 2)
    unsigned wbm = 0u;                             // slots whose tick this wave completed: it writes them back
    int hint = wave, idle = 0;
    // watchdog of the idle loop: looks in a row during which NOTHING in the workgroup moved (ring head and tail, every slot's claim word).  A resident
    // launch may legitimately run for seconds (T up to 65536); a wave with nothing to take is stuck only if nobody else makes progress either.
    int stall = 0, moved_sig = 0;
    unsigned w_seen = 0u;
    for (;;) {
        while (todo) {   // ---- the serial front of a slot, its rays into the ring, its Space.step published
            const int lane = opaque_v(lane0);
            const Params &p = *(const Params *)launder((ParamsK)pp0);
            const LaunchArgs &la = *(const LaunchArgs *)launder(lap0);
            int *const ctrl = reinterpret_cast<int *>(smem + p.lds_map_bytes);
            const BlockDesc *const K = block_desc_lds(smem, p, W);
            const int slot = uni(__builtin_ctz(todo));
            todo &= todo - 1;
            SSPREAD(slot, 0);
            const int e_s = uni(ctrl[4 * slot + 3]), t = uni(ctrl[4 * slot + 2]);
            const Lds Ls = carve<D>(p, smem, K->md, slot, wave);
            int ap = 0;
            if (la.actions && lane < D::A(p)) ap = la.actions[((size_t)t * p.N + e_s) * D::A(p) + lane];
            if (lane == 0) ctrl[4 * slot + 1] = 0;
            const int n2 = slot_front<D>(Ls, (ParamsK)pp0, la, K->md, K->gd, e_s, lane, ap, la.synth_tick + (unsigned long long)t, 0, pc);   // 1: Space.step to come; 0: it ran inside (reset)
            lds_release();
            if (lane == 0) __hip_atomic_store((unsigned *)&ctrl[4 * slot], rw_make(t + 1, n2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            PHASE(pc, 3);
            const int n_res = pool_sort<D, kExact>(Ls, p, la, K->gd, (long long)t * p.N + e_s, slot, lane, pool_ctl(smem, p, W), reinterpret_cast<unsigned long long *>(smem + p.lds_pool_off));
            lds_release();
            const int add = n_res + (n2 == 0 ? 1 : 0);
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(&ctrl[4 * slot + 1], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (add > 0 && uni(old) + add == D::A(p) * D::R(p) + 1) wbm |= 1u << slot;   // only a party that counted something can complete the tick (add == 0: the
            hint = slot;                                                                     // wave that counted the last ray or the Space.step has seen the total already)
            WSPREAD(2); SSPREAD(slot, 1);
        }
#ifdef CAT_WB_COUNTS   // diagnostic build: how many slot ticks a wave completes at once (cat_debug_wb_counts)
        if (wbm && lane0 == 0) { const int k_ = __popc(wbm); atomicAdd(&g_wb_counts[k_ > 7 ? 7 : k_], 1ull); }
#endif
        while (wbm) {   // ---- this wave completed these slots' ticks: write them back; their next fronts are this wave's next job
            const int lane = opaque_v(lane0);
            const Params &p = *(const Params *)launder((ParamsK)pp0);
            const LaunchArgs &la = *(const LaunchArgs *)launder(lap0);
            int *const ctrl = reinterpret_cast<int *>(smem + p.lds_map_bytes);
            const BlockDesc *const K = block_desc_lds(smem, p, W);
            const int slot = uni(__builtin_ctz(wbm));
            wbm &= wbm - 1;
            lds_acquire();
            PHASE(pc, 16);
            SSPREAD(slot, 12);
            const int e_s = uni(ctrl[4 * slot + 3]), t = uni(ctrl[4 * slot + 2]);
            const Lds Ls = carve<D>(p, smem, K->md, slot, wave);
            const long long eo = (long long)t * p.N + e_s;
            const int step2 = uni(Ls.flags[0]), captured2 = uni(Ls.flags[1]), timeout2 = uni(Ls.flags[2]), rcount = uni(Ls.flags[3]);
            const bool last = t + 1 >= T;
            slot_writeback<D>(Ls, p, la, e_s, eo, lane, 1, last, step2, captured2, timeout2, rcount, lut_c, lut_t, pc);
            wave_sync();   // the write-back has read the slot's staging and flags; the next front overwrites them
            SSPREAD(slot, 13);
            if (!last) { if (lane == 0) ctrl[4 * slot + 2] = t + 1; todo |= 1u << slot; }
            else if (lane == 0) __hip_atomic_store((unsigned *)&ctrl[4 * slot], kRwFinished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (todo) continue;
        // ---- look for work: a full round of the ring, else an open Space.step, else what the ring holds
        int base, n, slot = -1;
        {
            const int lane = opaque_v(lane0);
            const Params &p = *(const Params *)launder((ParamsK)pp0);
            int *const ctrl = reinterpret_cast<int *>(smem + p.lds_map_bytes);
            int *const pctl = pool_ctl(smem, p, W);
            unsigned w_l = kRwFinished;
            if (lane < W) w_l = __hip_atomic_load((unsigned *)&ctrl[4 * lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int hd = 0, tl = 0;
            if (lane == 0) { hd = __hip_atomic_load(&pctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); tl = __hip_atomic_load(&pctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            hd = uni(hd); tl = uni(tl);
            const int avail = tl - hd;
            const unsigned open = (unsigned)__ballot(rw_next(w_l) < rw_units(w_l));
            if (avail >= kPoolRound || (open == 0u && (avail >= kPoolMinPartial || (avail > 0 && idle >= kPoolPatience)))) {
                base = hd; n = avail < kPoolRound ? avail : kPoolRound;
                int seen = hd;
                if (lane == 0) __hip_atomic_compare_exchange_strong(&pctl[0], &seen, hd + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (uni(seen) != hd) continue;   // another wave took them: look again
            } else if (open != 0u) {
                // the Space.step of the slot FURTHEST BEHIND (lowest epoch), ties round the ring from the slot this wave worked on last
                n = 0; base = 0;
                unsigned key = 0xFFFFFFFFu;
                if (rw_next(w_l) < rw_units(w_l)) key = ((unsigned)rw_epoch(w_l) << 5) | (unsigned)((lane - hint) & 31);
#define CAT_ROW_MIN(SH) { const unsigned o_ = (unsigned)__builtin_amdgcn_update_dpp((int)key, (int)key, 0x120 + SH, 0xF, 0xF, false); key = o_ < key ? o_ : key; }
                CAT_ROW_MIN(8) CAT_ROW_MIN(4) CAT_ROW_MIN(2) CAT_ROW_MIN(1)
#undef CAT_ROW_MIN
                slot = uni((hint + (int)(key & 31u)) & 31);
                const unsigned wv = (unsigned)__builtin_amdgcn_readlane((int)w_l, slot);
                unsigned seen = wv;
                if (lane == 0)
                    __hip_atomic_compare_exchange_strong((unsigned *)&ctrl[4 * slot], &seen, wv + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((unsigned)uni((int)seen) != wv) continue;
                hint = slot;
            } else {
                // nothing to take: leave when nothing can come any more -- every slot has finished its T ticks; with one tick per launch, when
                // every slot HAS published (its rays are in rounds under way on other waves, which write it back)
                if (avail == 0 && __ballot(kOneTick ? (w_l == 0u) : (w_l != kRwFinished)) == 0ull) break;   // (a remainder below kPoolMinPartial is taken after kPoolPatience looks)
                ++idle;
                {
                    const bool moved = __ballot(w_l != w_seen) != 0ull || hd + tl != moved_sig;   // head and tail only grow: their sum changes with either
                    w_seen = w_l; moved_sig = hd + tl;
                    stall = moved ? 0 : stall + 1;
                }
                if (stall > kSpinLimit) { if (lane == 0) atomicOr(p.err_word, CAT_DEVERR_SCHEDULER); break; }   // never in a correct run
                __builtin_amdgcn_s_sleep(4);
                PHASE(pc, 21);
                continue;
            }
            idle = 0;
            lds_acquire();
            PHASE(pc, 22);
        }
        {   // ---- a round of rays (entity.py:143-144, base_env.py:388-390) or a slot's Space.step (base_env.py:392)
            const int lane = opaque_v(lane0);
            const Params &p = *(const Params *)launder((ParamsK)pp0);
            const LaunchArgs &la = *(const LaunchArgs *)launder(lap0);
            int *const ctrl = reinterpret_cast<int *>(smem + p.lds_map_bytes);
            const BlockDesc *const K = block_desc_lds(smem, p, W);
            if (n > 0) {
                const Lds L0 = carve<D>(p, smem, K->md, 0, wave);
                wbm |= pool_round<D, kExact>(L0, p, la, uni(K->md.S), K->md.cmax, base, n, lane, ctrl, reinterpret_cast<unsigned long long *>(smem + p.lds_pool_off), pc);
            } else {
                const Lds Ls = carve<D>(p, smem, K->md, slot, wave);
                SSPREAD(slot, 2);
                PHASE(pc, 9);
                physics_env<D>(Ls, p, uni(K->md.S), lane, pc);
                PHASE(pc, 10);
                lds_release();   // the unit's LDS writes, before it counts as done
                SSPREAD(slot, 3);
                int old = 0;
                if (lane == 0) old = __hip_atomic_fetch_add(&ctrl[4 * slot + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (uni(old) + 1 == D::A(p) * D::R(p) + 1) wbm |= 1u << slot;
            }
            PHASE(pc, 23);
        }
    }
    PHASE(pc, 11);
    WSPREAD(3); WSPREAD(5);
    pc.flush(lane0);
}

template <class D, bool kExact = false>
__global__ __launch_bounds__(kMaxWaves *kLanes) void rollout_kernel_pooled(const Params *__restrict__ pp0, const LaunchArgs la0, const Prologue)
{
    rollout_body_pool<D, false, kExact>(pp0, la0);
}
template <class D, bool kExact = false>
__global__ __launch_bounds__(kMaxWaves *kLanes) void step_kernel_pooled(const Params *__restrict__ pp0, const LaunchArgs la0, const Prologue)
{
    rollout_body_pool<D, true, kExact>(pp0, la0);
}

template <class D>
__device__ __forceinline__ void reset_slot(const Lds &L, const Params &p, const LaunchArgs &la, const MapDesc &md,
                                           const GridDesc &gd, int env, int wave, int lane);

// BaseEnv.reset (base_env.py:286-352) for masked envs
template <class D>
__global__ __launch_bounds__(kMaxWaves *kLanes) void reset_kernel(const Params *__restrict__ pp0, const LaunchArgs la0, const Prologue)
{
    extern __shared__ __align__(16) char smem[];
    const int lane0 = threadIdx.x % kLanes;
    const int wave = uni(threadIdx.x / kLanes);
    const LaunchArgsK lap0 = kernarg_launch_args();
    PhaseClock pc;
    int W;
    {
        const Params &p = *pp0;
        int uniform;
        const Params q = prologue_params(kernarg_prologue(), uniform);   // what comes before the barrier reads this register copy
        const int lane = lane0;
        W = uni((int)(blockDim.x / kLanes));
        int env;
        BlockDesc bd0;
        prologue_env_desc(q, uniform, W, wave, env, bd0);
        bool need = env >= 0;
        if (need) {
            if (la0.use_done_mask)
                need = ((GAS const int *)(G(q.state) + (size_t)env * D::rec_bytes(q) + 96 * D::A(q)))[2] != 0;   // the hot part's `done`
            else if (la0.mask) need = la0.mask[env] != 0;
        }
        if (!__syncthreads_or(need ? 1 : 0)) return;  // nothing to reset in this workgroup: skip the map staging
        const MapDesc &md0 = bd0.md;
        const GridDesc &gd0 = bd0.gd;
        int *const ctrl = reinterpret_cast<int *>(smem + q.lds_map_bytes);
        if (lane < 4) ctrl[4 * wave + lane] = lane == 3 ? (need ? env : -1) : 0;   // claimed, done, published, env id
        stage_map<D>(q, smem, md0, &bd0, const_cast<BlockDesc *>(block_desc_lds(smem, q, W)));   // ends with the workgroup barrier
        if (need) {
            const Lds L = carve<D>(p, smem, md0, wave, wave);
            reset_slot<D>(L, p, la0, md0, gd0, env, wave, lane);
        }
    }
    run_units<D>(pp0, lap0, smem, W, wave, lane0, 0, pc);   // waves with nothing to reset help with the others' ray chunks
}

// Spawn sampling + Entity.reset of one env, then its ray-fan setup is published (reset_kernel).
template <class D>
__device__ __forceinline__ void reset_slot(const Lds &L, const Params &p, const LaunchArgs &la, const MapDesc &md,
                                           const GridDesc &gd, int env, int wave, int lane)
{
    const int A = D::A(p);
    load_state<D>(L, p, env, lane);
    const unsigned rc = (unsigned)(uni(L.cnt[1]) + 1);
    spawn_and_reset<D>(L, p, la, md, env, rc, lane);
    wave_sync();
    copy_snapshot(L, A, lane);   // fresh positions, stale circle caches and leaf bbs (Q1); overwrites the spawn points
    agent_setup<D>(L, p, gd, lane, setup_row(L.fpos, p, gd, lane, A));   // :334-344 (setup part)
    if (lane == 0) { L.flags[0] = 0; L.flags[1] = 0; L.flags[2] = 0; L.flags[3] = (int)rc; }  // :350
    publish_slot(L, wave, lane, fan_units<D>(p));
}

// _get_non_colliding_position + Entity.reset for every agent of the env in L (base_env.py:313-332, 123-166;
// entity.py:148-157): new positions, zero velocities; the circle caches and leaf bbs stay stale (quirk Q1).
template <class D>
__device__ __forceinline__ void spawn_and_reset(const Lds &L, const Params &p, const LaunchArgs &la, const MapDesc &md,
                                                int env, unsigned rc, int lane)
{
    const int S = md.S, A = D::A(p);
    GAS const double *start = G(p.geo_f64) + md.f64_off + 4 * md.S + geo_rest_doubles(md);
    GAS const double *regions = start + 2 * md.A;
    GAS const int *region_off = G(p.geo_i32) + md.i32_off + 2 * md.S;

    for (int i = 0; i < A; i++) {
        double sx, sy;
        if (la.positions) {
            sx = la.positions[((size_t)env * A + i) * 2]; sy = la.positions[((size_t)env * A + i) * 2 + 1];
        } else {
            const int r0 = region_off[i], nr = region_off[i + 1] - r0;
            if (nr <= 0) { sx = start[2 * i]; sy = start[2 * i + 1]; }      // :323-332 Entity.reset()
            else {
                unsigned rnd[4];
                philox_env_v(p, env, rc, (unsigned)i, 0x100u, rnd);
                GAS const double *rg = regions + 4 * (r0 + (int)(rnd[0] % (unsigned)nr));  // :144-145
                const double rgx = rg[0], rgy = rg[1], rgw = rg[2], rgh = rg[3];
                bool ok = false;
                sx = rgx + rgw / 2; sy = rgy + rgh / 2;                     // :163-166 fallback
                for (int att = 0; att < 20 && !ok; att++) {                 // :151
                    philox_env_v(p, env, rc, (unsigned)i, 0x200u + (unsigned)att, rnd);
                    const double x = rgx + ((rgx + rgw) - rgx) * u53(rnd[0], rnd[1]);  // map_utils.py:9-10
                    const double y = rgy + ((rgy + rgh) - rgy) * u53(rnd[2], rnd[3]);
                    // Space.point_query_nearest(pos, radius, ray_filter) is None  (:154-157)
                    bool any = false;
                    for (int j = 0; j < A; j++) {
                        if (j == i) continue;
                        double ex = x - L.tc[2 * j], ey = y - L.tc[2 * j + 1];
                        if (sqrt(ex * ex + ey * ey) - p.rc < p.rc) any = true;
                    }
                    for (int base = 0; base < S && !any; base += kLanes) {
                        const int s = base + lane;
                        bool hit = false;
                        if (s < S) {
                            const double *bb = L.bb + kBB * s;
                            const double m = p.rc + 1e-6;
                            if ((bb[0] - m <= x) && (x <= bb[2] + m) && (bb[1] - m <= y) && (y <= bb[3] + m))
                                hit = poly_point_distance(L, s, p.wall_r, x, y) < p.rc;
                        }
                        any = (__ballot(hit) != 0ull);
                    }
                    if (!any) { sx = x; sy = y; ok = true; }
                }
            }
        }
        L.spawn[2 * i] = sx; L.spawn[2 * i + 1] = sy;
    }
    for (int i = 0; i < A; i++) {  // Entity.reset (entity.py:148-157); shape caches stay stale (Q1)
        L.pos[2 * i] = L.spawn[2 * i]; L.pos[2 * i + 1] = L.spawn[2 * i + 1];
        L.vel[2 * i] = 0.0; L.vel[2 * i + 1] = 0.0;
    }
}

__global__ void random_actions_kernel(const Params *__restrict__ pp, unsigned long long tick, int *actions)
{
    using D = DynDims;
    const Params &p = *pp;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.N * D::A(p)) return;
    const int env = idx / D::A(p), i = idx % D::A(p);
    unsigned rnd[4];
    philox_env(p, env, (unsigned)tick, (unsigned)i, 0xAC710u, rnd);
    actions[idx] = (int)(rnd[0] & 3u);
}

// cat_get_state / cat_set_state see the records field by field (strided copies); the cold part of a slot whose
// cache_live flag is 0 holds stale bytes.  mode 0 (before a read-out): such slots get the "no cached arbiter" pattern;
// mode 1 (after cold fields were written from outside): every slot's flag is raised, so the kernels read what was set.
__global__ void cold_fixup_kernel(const Params *__restrict__ pp, int mode)
{
    const Params &p = *pp;
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= p.N) return;
    char *rec = p.state + (size_t)env * p.rec_bytes;
    int *cnt = reinterpret_cast<int *>(rec + 96 * p.A);
    if (mode == 1) { cnt[3] = 1; return; }
    if (cnt[3] != 0) return;
    const int A = p.A, NPs = p.NP > 0 ? p.NP : 1;
    double *cd = reinterpret_cast<double *>(rec + p.hot_bytes);
    for (int q = 0; q < A * kK + NPs; q++) cd[q] = 0.0;
    int *ci = reinterpret_cast<int *>(cd + A * kK + NPs);
    for (int q = 0; q < A * kK; q++) { ci[q] = -1; ci[A * kK + q] = 0; }
    for (int q = 0; q < NPs; q++) ci[2 * A * kK + q] = -1;
}

__global__ void selftest_kernel(int op, const double *a, const double *b, double *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r = 0.0;
    if (op == 0) r = sqrt(a[i]);
    else if (op == 1) r = a[i] / b[i];
    else if (op == 2) r = (double)f64_to_f16(a[i]);
    else if (op == 3) r = (double)obs_distance_f16(a[i], b[i], 0.0, 0.0);
    out[i] = r;
}

thread_local char g_create_err[256] = "";
