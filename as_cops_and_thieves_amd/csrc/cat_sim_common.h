// cat_sim_common.h -- part of the env core's single translation unit (included by cat_sim.hip, in this order; not a stand-alone header):
// constants, the parameter block, problem dimensions, diagnostics hooks, small device helpers, the LDS view.

constexpr int kMaxWaves = 16;       // waves (= env slots) per workgroup: Params::wpb in {1, 2, 4, 8, 16}
constexpr int kLanes = 64;          // gfx950 wavefront
constexpr int kK = CAT_WALL_CACHE;
constexpr int kBB = 6;              // doubles per wall bb record in LDS: 4 used (l b r t); 6 (48 B) spreads 16 lanes' b128 reads over all 64 banks
// Edge-pair records: the f32 records of TWO consecutive hull edges interleaved component by component (pre-classification on packed
// f32 arithmetic, v_pk_*_f32: one instruction for both edges).  20 floats (80 B) per pair: 14 used, and 16 lanes reading 16
// different records with ds_read_b128 still cover all 64 banks.  [nx0 nx1 ny0 ny1 | c0 c1 dtMin0 dtMin1 | dtMax0 dtMax1 vx0 vx1 | vy0 vy1 - - | pad]
constexpr int kPairF = 20;
constexpr unsigned kBlobMagic = 0x31544143u;

struct MapDesc {
    int S, P, A, n_regions;
    int f64_off;   // into geo_f64: [bb 4S][planes 8P][edge pairs (kPairF/2) PP][start 2A][regions 4Rg]
    int i32_off;   // into geo_i32: [first S][count S][region_off A+1]
    float cmax;    // max over the planes of |dot(v0, n) + wall radius + ray radius| (error bound of the f32 pre-classification)
    int PP;        // edge-pair records (sum over the walls of ceil(edges / 2))
};

// Spatial-hash grids, built once per (map, ray table) on the host (build_grids):
//   ray grid     : (origin cell, ray index) -> ascending ids of the walls that can matter to that ray's query from ANY origin
//                  inside the cell: visitable (bb), hittable (hull), not occluded by a certain earlier hit -- see build_grids;
//                  the exact [CP cpBBSegmentQuery] gate and shape query are still evaluated per listed wall
//   contact grid : cell -> ascending wall ids whose bb comes within the ray radius of the cell (the "origin within the
//                  query radius of the shape" rule of the ray fan's setup)
struct GridDesc {
    double x0, y0, inv_cell;
    int nx, ny;
    int off_base, ent_base;     // into grid_off / grid_ent ; rows = nx*ny*R (+1); row_base == off_base - map index
    int coff_base, cent_base;   // into cgrid_off / cgrid_ent ; rows = nx*ny (+1)
    int crow_base, span;        // into cgrid_rows (one packed 8-byte row per cell); chunk form: 64-ray chunks of a slot per work unit in the resident launch ...
    int row_base, span_tick;    // into grid_rows, in rows of Params::row_words words; ... and in the one-tick launch (1: fan_chunk, one chunk per unit)
};

// doubles of geometry after the wall bbs: the f64 plane records, then the f32 records
__host__ __device__ inline int geo_rest_doubles(const MapDesc &md) { return 8 * md.P + (kPairF / 2) * md.PP; }

struct BlockDesc { MapDesc md; GridDesc gd; };   // per workgroup: one load instead of block_map -> maps / grids
constexpr int kWgConstBytes = 128;   // LDS copy of the workgroup's BlockDesc (resident rollout kernel: descriptors are re-read from LDS
                                     // by the phase that needs them instead of living in registers across the scheduler loop)
static_assert(sizeof(BlockDesc) <= kWgConstBytes, "BlockDesc must fit its LDS block");
__host__ __device__ inline int ctrl_bytes(int W) { return 16 * W + kWgConstBytes; }   // [W][4] control words, then the constant block

struct Params {
    int N, A, n_cops, R, max_step, iterations, persistence, gate, NP, maxc;
    long long env_id_offset;
    unsigned long long seed;
    double dt, bias_coef, slop, ray_length, ray_radius, rc, mass, impulse, max_speed, term_radius, wall_r;
    const double *ray_dx, *ray_dy;
    const float *cop_lut, *thief_lut;
    const MapDesc *maps;
    const double *geo_f64;
    const int *geo_i32;
    const GridDesc *grids;  // [n_maps]
    const unsigned long long *grid_rows;
    const int *grid_off, *cgrid_off;
    const unsigned char *grid_ent, *cgrid_ent;
    const unsigned long long *cgrid_rows;   // per cell: count | first 7 contact-candidate wall ids (longer lists: the CSR arrays)
    unsigned *err_word;     // device-side error flags (CAT_DEVERR_*), read back by cat_device_errors
    const int *work_env;    // [n_blocks*wpb] env slot or -1
    const int *block_map;   // [n_blocks]
    const BlockDesc *block_desc;   // [n_blocks]: maps[block_map[b]], grids[block_map[b]]
    // Env state: one contiguous record per env slot (so a wave moves it with 16-byte lanes), in two parts:
    //   HOT  (always moved)   f64  pos[2A] vel[2A] vbias[2A] tc[2A] leaf[4A]   i32  step_count reset_count done cache_live
    //   COLD (arbiter caches) f64  wall_jn[8A] pair_jn[NPs]                    i32  wall_shape[8A] wall_age[8A] pair_age[NPs]
    // The cold part is only read when cache_live says it holds something, and only written while it does: an agent
    // in free space has no cached arbiter, and its slot then moves 96 A + 16 bytes per tick instead of the whole record.
    char *state;
    int rec_bytes, hot_bytes;
    int maxE, ang_ok, row_words, row_id_bits, row_cnt_mul;   // four-byte rows: field width; (bit index * row_cnt_mul) >> 16 = field index
    float ang0, inv_step;
    // LDS carve (bytes)
    int lds_map_bytes, lds_env_bytes, lds_union_bytes;
    int wpb;                // waves per workgroup (= blockDim.x / 64)
    int lds_pool_off, pool_mask;   // ray pool of the *_pooled kernels: byte offset in LDS, capacity - 1 (capacity >= wpb * A * R, even); 0: no pool
    int grp_rays;           // rays the arow / alist / adyn arrays of a scratch union hold (fan_group)
    int item_cap;           // fan_slot: items its list holds (itbb / ialpha / itm arrays of a scratch union); the chunks per unit are per map: GridDesc::span
    unsigned pool_magic; int pool_shift;   // ring position of entry i: i & pool_mask where the capacity is a power of two (pool_shift < 0), else i - capacity * (i / capacity)
                                           // with the quotient by multiply-high and shifts (unsigned division by an invariant, computed by cat_create)
};

// Problem dimensions as seen by the device code: either read from the parameter block (DynDims) or compile-time
// constants for the rosters / ray counts that are instantiated (FixDims): constant LDS offsets and unrolled agent loops.
// Round 4, as the compiler reports them (tools/regs.sh, profiles/r04_registers.txt): step_kernel / rollout_kernel 115 - 121 VGPRs and 9 - 17 spilled SGPRs fixed (23 - 41 generic),
// the pooled pair 113 - 115 and 20 - 21 (37 - 38), reset_kernel 103 - 107 VGPRs and 6 - 22; no scratch in any instantiation (rounds 1 - 3, tick_kernel: 127 - 128 VGPRs, 64 - 78
// spilled SGPRs fixed, 107 - 134 + 64 B of scratch generic -- see "opaque roots" below).
struct DynDims {
    static constexpr bool kFixed = false;
    static __device__ __forceinline__ int A(const Params &p) { return p.A; }
    static __device__ __forceinline__ int R(const Params &p) { return p.R; }
    static __device__ __forceinline__ int n_cops(const Params &p) { return p.n_cops; }
    static __device__ __forceinline__ int NP(const Params &p) { return p.NP; }
    static __device__ __forceinline__ int rec_bytes(const Params &p) { return p.rec_bytes; }
    static __device__ __forceinline__ int hot_bytes(const Params &p) { return p.hot_bytes; }
};
template <int TA, int TR, int TC> struct FixDims {
    static constexpr bool kFixed = true;
    static constexpr int kNP = TA * (TA - 1) / 2, kNPs = kNP > 0 ? kNP : 1;
    static constexpr int kHotBytes = 96 * TA + 16;
    static constexpr int kColdBytes = ((TA * CAT_WALL_CACHE + kNPs) * 8 + (2 * TA * CAT_WALL_CACHE + kNPs) * 4 + 15) / 16 * 16;
    static __device__ __forceinline__ constexpr int A(const Params &) { return TA; }
    static __device__ __forceinline__ constexpr int R(const Params &) { return TR; }
    static __device__ __forceinline__ constexpr int n_cops(const Params &) { return TC; }
    static __device__ __forceinline__ constexpr int NP(const Params &) { return kNP; }
    static __device__ __forceinline__ constexpr int rec_bytes(const Params &) { return kHotBytes + kColdBytes; }
    static __device__ __forceinline__ constexpr int hot_bytes(const Params &) { return kHotBytes; }
};

// Which form of the ray fan an instantiation carries: 0 = fan_chunk (one 64-ray chunk of one agent per work unit: any map),
// 1 = fan_group (the agents of a group per unit, only the rays that have a candidate on the lanes: maps whose rays meet few walls).
template <class Base, int F> struct WithFan : Base { static constexpr int kFan = F; };
// fan_group: agents per work unit
template <class D> __device__ __forceinline__ int group_agents(const Params &p)
{
    // Two agents per unit when their rays fill at most four chunks: compaction across the pair (two cops in the open: ~40 active
    // rays of 128 -> one round) while a slot still has several units for the waves of the workgroup to share.  (All three agents of
    // a 2v1 roster in ONE unit: 40.3 us against 37.9 on the labyrinth x4096, 32.4 against 31.6 with the round-3 candidate table -- and
    // 106.9 against 110.4 at 16384 envs, where the launch is several workgroup rounds long; one agent per unit: 42.0.)
    const int cpa = (D::R(p) + 63) / 64;
    int g = cpa <= 2 ? 2 : 1;
    if (g * cpa * kLanes > p.grp_rays) g = p.grp_rays / (cpa * kLanes);   // (a sim whose ring leaves the scratch unions room for one agent's chunks only)
    return g;
}
// ... and in the scheduler kernels (step_kernel, rollout_kernel), for rosters whose rays fill more than four chunks: as many agents as fill four (3v2 at 64 rays:
// units of 4 + 1 agents instead of 2 + 2 + 1).  With T ticks per launch the slots of a workgroup run out of phase and a slot needs
// less parallelism inside itself; fewer units pay fewer prologues and pack their rounds fuller: 3v2 x8192 79.7 -> 76.8 us per tick.
// (2v1 in ONE unit: 21.0 us against 20.5 with two, and 24.8 against 22.4 at T = 16: kept at two agents per unit.)
template <class D> __device__ __forceinline__ int group_agents_resident(const Params &p)
{
    const int cpa = (D::R(p) + 63) / 64, most = cpa <= 4 ? 4 / cpa : 1;
    return D::A(p) * cpa > 4 && most > group_agents<D>(p) && most * cpa * kLanes <= p.grp_rays ? most : group_agents<D>(p);
}
// ray-fan work units of an env slot (span: agents per unit of the group form, 64-ray chunks per unit of the chunk form)
template <class D> __device__ __forceinline__ int fan_units(const Params &p, int span)
{
    if constexpr (D::kFan == 1) return (D::A(p) + span - 1) / span;
    else return (D::A(p) * ((D::R(p) + 63) / 64) + span - 1) / span;
}
template <class D> __device__ __forceinline__ int fan_units(const Params &p) { return fan_units<D>(p, D::kFan == 1 ? group_agents<D>(p) : 1); }   // (reset_kernel: one chunk per unit)
// ... of the scheduler kernels: what one work unit spans
template <class D, bool kOneTick> __device__ __forceinline__ int unit_span(const Params &p, const GridDesc &gd)
{
    if constexpr (D::kFan == 1) return group_agents_resident<D>(p);   // also with one tick per launch: 3v2 x8192 98.8 against 100.5 us
    else return kOneTick ? gd.span_tick : gd.span;                    // (per map: cat_create)
}

// Diagnostic build only (-DCAT_PHASE_TIMING): per-phase shader-clock totals, summed over waves
// into a debug buffer no other kernel code reads.  The shipped library is built without it.
#ifdef CAT_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[32];
struct PhaseClock {   // accumulators live in LDS (one row per wave) to keep register pressure unchanged
    unsigned long long prev;
    unsigned long long *acc;
    __device__ PhaseClock()
    {
        __shared__ unsigned long long rows[kMaxWaves][24];
        acc = rows[threadIdx.x / 64];
        if (threadIdx.x % 64 < 24) acc[threadIdx.x % 64] = 0;
        prev = __builtin_readcyclecounter();
    }
    __device__ __forceinline__ void mark(int id)
    {
        unsigned long long t = __builtin_readcyclecounter();
        if (threadIdx.x % 64 == 0) acc[id] += t - prev;
        prev = t;
    }
    __device__ void flush(int lane) { if (lane < 24 && acc[lane]) atomicAdd(&g_phase_cycles[lane], acc[lane]); }
};
#define PHASE(pc, id) (pc).mark(id)
// event counters (slots 24..31; -DCAT_EVENT_COUNTS on top: the atomics distort the cycle marks): the active lanes' leader adds
// (1, number of active lanes) to slots (id, id + 1)
#ifdef CAT_EVENT_COUNTS
#define QCOUNT(id) do { const unsigned long long m_ = __ballot(true); if ((int)__builtin_ctzll(m_) == (int)(threadIdx.x % 64)) { \
    atomicAdd(&g_phase_cycles[id], 1ull); atomicAdd(&g_phase_cycles[(id) + 1], (unsigned long long)__popcll(m_)); } } while (0)
#else
#define QCOUNT(id) do {} while (0)
#endif
#else
#define QCOUNT(id) do {} while (0)
struct PhaseClock { __device__ __forceinline__ void flush(int) {} };
#define PHASE(pc, id) do {} while (0)
#endif

// Per-launch arguments travel by value (kernarg); the static Params live in device memory and are
// read with scalar loads where needed, which keeps them out of long-lived SGPRs.
struct LaunchArgs {
    cat_outputs out;
    const int *actions;
    const unsigned char *mask;
    const double *positions;
    int use_done_mask;
    int auto_reset;                 // step / rollout: episodes that end with a tick are reset inside the same launch
    unsigned long long synth_tick;  // step / rollout with actions == NULL: Philox actions of this tick (rollout: of the first tick)
    int T;                          // rollout_kernel: ticks per launch (outputs and actions carry a leading T)
};

// What a kernel's prologue needs of the parameter block (env id, descriptors, state record, map staging, LDS carve), as a THIRD
// kernel argument by value: it then arrives with the kernarg segment's first scalar loads instead of behind two more dependent round
// trips (kernarg -> Params pointer -> Params fields -> work list / descriptor -> geometry: the staging barrier stood 2.4 us after a
// wave's start, 0.64 of them for the parameter fields and 0.68 for env id + descriptor; tools/wave_spread.py).  `uniform`: every
// workgroup of the sim has the same map and the work list is the identity (one map, no helper waves): env id and descriptor then need
// no load at all (bd).  Filled once by cat_create.
struct Prologue {
    int lds_map_bytes, lds_env_bytes, lds_union_bytes, wpb, A, R, NP, maxc, n_cops, rec_bytes, hot_bytes, N;
    int uniform, lds_pool_off, pool_mask, grp_rays;
    const int *work_env;
    const BlockDesc *block_desc;
    char *state;
    const double *geo_f64;
    const int *geo_i32;
    const double *ray_dx, *ray_dy;
    const float *cop_lut, *thief_lut;
    BlockDesc bd;
};
static_assert(sizeof(LaunchArgs) % 8 == 0 && alignof(Prologue) == 8, "kernarg layout: [const Params *][LaunchArgs][Prologue]");

// ------------------------------------------------------------------ small helpers -----------
// Pointers read out of the Params block are generic to the compiler, which then emits FLAT accesses:
// those count on vmcnt AND lgkmcnt, so every LDS wait also drains them (no prefetch survives).  An
// explicit cast to the global address space turns them into global_load/global_store.
#define GAS __attribute__((address_space(1)))
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // 16-byte moves that work across address spaces
typedef double f64x2 __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ GAS T *G(T *p) { return (GAS T *)p; }

// value known to be the same in every lane -> SGPR (lets the compiler keep loop control scalar)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ double fmax2(double a, double b) { return (a > b) ? a : b; }  // [CP cpfmax]
__device__ __forceinline__ double fmin2(double a, double b) { return (a < b) ? a : b; }  // [CP cpfmin]

// (bb_segment_query: v_max_f64 / v_min_f64 -- see there)
#define CAT_FMAX(a, b) __builtin_fmax((a), (b))
#define CAT_FMIN(a, b) __builtin_fmin((a), (b))

// wave-local ordering of LDS traffic between lanes (one wave owns its scratch; no s_barrier)
__device__ __forceinline__ void wave_sync()
{
    // LDS-only ("local") fences: ordering global stores here would make every sync wait for HBM
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// round-to-nearest-even f64 -> f16 bits: NumPy's cast for np.array(points, dtype=np.float16)
// (entity.py:206) and for the weak python-float origin (entity.py:208).  f64 -> f32 with
// round-to-odd (truncate, then OR the sticky bit into the lsb) followed by the hardware's RNE
// f32 -> f16 is a correctly rounded single step: 24 bits >= 11 + 2.  Checked bit for bit against
// NumPy by tests/test_gpu_parity.py::test_device_arithmetic_is_ieee_exact.
__device__ __forceinline__ unsigned f32_to_f16(float f)
{
    _Float16 h = (_Float16)f;  // v_cvt_f16_f32: RNE, f16 denormals enabled
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}

__device__ __forceinline__ unsigned f64_to_f16(double x)
{
    float r = (float)x;  // v_cvt_f32_f64, RNE
    const double back = (double)r;
    if (back != x && !(x != x)) {
        int bits = __float_as_int(r);
        // |r| > |x|: step one ulp toward zero to get the truncated value (same sign, r != 0 here)
        if (fabs(back) > fabs(x)) bits -= 1;
        bits |= 1;  // inexact -> odd
        r = __int_as_float(bits);
    }
    return f32_to_f16(r);
}

__device__ __forceinline__ float f16_to_f32(unsigned h)
{
    return (float)__builtin_bit_cast(_Float16, (unsigned short)h);  // v_cvt_f32_f16, exact
}

// entity.py:206-210: f16(point) - f16(origin) in f32 -> f16; np.hypot on f16 = hypotf -> f16
__device__ __forceinline__ unsigned obs_distance_f16(double px, double py, double ox, double oy)
{
    float dx32 = f16_to_f32(f64_to_f16(px)) - f16_to_f32(f64_to_f16(ox));
    float dy32 = f16_to_f32(f64_to_f16(py)) - f16_to_f32(f64_to_f16(oy));
    float dx = f16_to_f32(f32_to_f16(dx32)), dy = f16_to_f32(f32_to_f16(dy32));
    float hyp = (float)sqrt((double)dx * (double)dx + (double)dy * (double)dy);
    return f32_to_f16(hyp);
}

__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                           unsigned k0, unsigned k1, unsigned out[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void philox_env(const Params &p, int env, unsigned c1, unsigned c2,
                                           unsigned c3, unsigned out[4])
{
    unsigned long long gid = (unsigned long long)(p.env_id_offset + env);
    philox4x32((unsigned)gid, c1, c2, c3 ^ ((unsigned)(gid >> 32) << 24), (unsigned)p.seed,
               (unsigned)(p.seed >> 32), out);
}

// The same stream for the spawn sampling (rare path): the key in VGPRs, so that its ten round keys are not precomputed as twenty
// wave-uniform scalars that the surrounding code then has to spill
__device__ __forceinline__ void philox_env_v(const Params &p, int env, unsigned c1, unsigned c2, unsigned c3, unsigned out[4])
{
    unsigned long long gid = (unsigned long long)(p.env_id_offset + env);
    unsigned k0 = (unsigned)p.seed, k1 = (unsigned)(p.seed >> 32);
    asm volatile("" : "+v"(k0), "+v"(k1));
    philox4x32((unsigned)gid, c1, c2, c3 ^ ((unsigned)(gid >> 32) << 24), k0, k1, out);
}

__device__ __forceinline__ double u53(unsigned a, unsigned b)
{
    return (double)(((unsigned long long)(a >> 5) << 26) | (unsigned long long)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// ------------------------------------------------------------------ LDS view ------------------
struct Lds {
    const double *bb;      // [S][4]            workgroup-shared
    const double *planes;  // [P][8]
    const float *p32;      // [PP][kPairF] f32 edge-pair records for the conservative pre-classification (layout: kPairF)
    const int *fc;         // [S] first plane | plane count << 16
    const int *fp;         // [S] first edge-pair record of the wall
    // per env slot
    double *pos, *vel, *vb, *tc, *leaf;  // [A][2] x4, [A][4]
    const double *fpos, *ftc, *fleaf;    // what the ray fan reads: the tick-start snapshot of pos / tc / leaf
    unsigned *dmin;   // [A] minimum wanted-class distance (f16 bits), 0x10000 = none seen
    int *flags;       // step, captured, timeout, -
    int *ctrl;        // workgroup: [wpb][4] = chunks claimed, chunks done, published, env id
    double *wjn, *pjn;
    int *wsh, *wag, *pag;
    int *cnt;       // step_count, reset_count, done, pad (tail of the state record)
    char *rec;      // the env's state record (same layout as in HBM)
    unsigned short *sd;  // [2R] team-shared distance (staged for wide stores)
    unsigned char *st;   // [2R] team-shared type
    double *conf;   // [maxc] contact records of kConD doubles: 12 doubles, then four ints (physics_env)
    const double *rayd;  // [R][2]  ray offsets (workgroup-shared)
    int *acell, *anear;     // [A], [A][2]  grid cell and "origin inside" wall ids per agent
    int *dk0, *dcnt;        // [A*A]  ray cone (start, count | near << 16) of agent j seen from agent i
    int *adn;               // [A]    bit j: the origin of agent i lies within the ray radius of agent j's cached circle
    // ray-fan scratch (overlays the contact arrays: disjoint phases)
    double *itbb, *ialpha;  // [kItemCap] per item: BBTree gate value, hit alpha (2.0 = none)
    unsigned short *itm;    // [kItemCap] in: ray lane | id << 6   out: id << 6 | feature
    unsigned short *itemidx;  // [kPassJ][64] item index of (candidate position, ray lane)
    // fan_group only (light maps): the rays of an agent group that have any candidate, compacted
    unsigned *arow;           // [rays of a group] the active ray's packed candidate row (four bytes: finalize_rows)
    unsigned char *alist;     // [rays of a group] the active ray: chunk slot of the group << 6 | lane
    unsigned char *adyn;      // [rays of a group] its cone mask of the other agents
    unsigned short *od;  // [A*R]
    unsigned char *ot;   // [A*R]
    double *spawn;  // [8A] reset: spawn points [2A]; every kernel: pre-step pos[2A] tc[2A] leaf[4A] snapshot
};

__device__ __forceinline__ int align_up(int x, int a) { return (x + a - 1) / a * a; }

// Pin a wave-uniform value in scalar registers: a kernel parameter read through the Params pointer is
// otherwise re-loaded (s_load + wait) at every use inside the hot loops instead of being kept.
template <class T> __device__ __forceinline__ T launder(T v) { asm volatile("" : "+s"(v)); return v; }
