// cat_sim.hip -- MI355X (gfx950 / CDNA4) batched Cops-and-Thieves env core + its C ABI.
//
// A workgroup of 1..16 wavefronts advances as many envs that share one map, whose geometry (hull planes +
// shape bbs) is staged once per workgroup in LDS.  A wavefront owns one env for the serial part of the tick;
// the ray fan -- cut into 64-ray chunks (fan_chunk), into agent groups whose candidate-less rays are sorted out first
// (fan_group: maps whose rays meet few walls), or pooled over all slots of the workgroup (pool_sort / pool_round: the *_pooled kernels) --
// and the physics step are work units any wave of the workgroup may claim.
// All body state and ray arithmetic is FP64 with contraction off, so results are bit-comparable with a non-FMA
// CPU evaluation of the same formulas.  No MFMA: the path is geometry/indexing, bounded by FP64 VALU issue and
// LDS latency (DESIGN.md "Kernels").
//
// What each device function reproduces (paths relative to the reference repo; [CP x] = the
// Chipmunk2D 7.0.x function of that name, a third-party dependency of the reference whose
// published algorithm is followed -- SURVEY.md appendix A):
//   step_kernel / rollout_kernel (and their _pooled forms)   BaseEnv.step (one tick / T ticks per launch)   src/environments/base_env.py:354-413
//   reset_kernel  BaseEnv.reset                      src/environments/base_env.py:286-352
//   agent_setup, fan_chunk, fan_group, pool_sort, pool_round   Entity.get_observation/_query_body   src/agents/entity.py:159-241
//   rewards_and_positions    Cop.reward / Thief.reward            src/agents/cop.py:49-75, thief.py:48-69
//   emit_observations        get_shared_observations              src/environments/observation_spaces.py:67-131
//   physics_env              pymunk Space.step -> [CP cpSpaceStep]  (call site base_env.py:392)
//   termination_captured     BaseEnv._termination_criterion       src/environments/base_env.py:521-554
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <thread>
#include <type_traits>
#include <sched.h>
#include <string>
#include <vector>

#include "cat_sim.h"

namespace {

constexpr int kMaxWaves = 16;       // waves (= env slots) per workgroup: Params::wpb in {1, 2, 4, 8, 16}
constexpr int kLanes = 64;          // gfx950 wavefront
constexpr int kK = CAT_WALL_CACHE;
#ifndef CAT_P32_FLOATS
#define CAT_P32_FLOATS 12
#endif
#ifndef CAT_BB_DOUBLES
#define CAT_BB_DOUBLES 6
#endif
constexpr int kBB = CAT_BB_DOUBLES;     // doubles per wall bb record in LDS: 4 used (l b r t); 6 (48 B) for the same reason
constexpr int kP32F = CAT_P32_FLOATS;   // floats per f32 plane record: 8 used; 12 (48 B) spreads 16 lanes' b128 reads over all 64 banks
constexpr int kGeoPerPlane = 8 + kP32F / 2;   // doubles of LDS geometry per plane: the f64 record + the f32 record (CAT_EDGE_PAIRS == 0)
#ifndef CAT_EDGE_PAIRS
#define CAT_EDGE_PAIRS 1
#endif
// CAT_EDGE_PAIRS: the f32 records of TWO consecutive hull edges interleaved component by component (pre-classification on packed
// f32 arithmetic, v_pk_*_f32: one instruction for both edges).  20 floats (80 B) per pair: 14 used, and 16 lanes reading 16
// different records with ds_read_b128 still cover all 64 banks.  [nx0 nx1 ny0 ny1 | c0 c1 dtMin0 dtMin1 | dtMax0 dtMax1 vx0 vx1 | vy0 vy1 - - | pad]
constexpr int kPairF = 20;
constexpr unsigned kBlobMagic = 0x31544143u;

struct MapDesc {
    int S, P, A, n_regions;
    int f64_off;   // into geo_f64: [bb 4S][planes 8P][planes32 (kP32F/2) P][start 2A][regions 4Rg]
    int i32_off;   // into geo_i32: [first S][count S][region_off A+1]
    float cmax;    // max over the planes of |dot(v0, n) + wall radius + ray radius| (error bound of the f32 pre-classification)
    int PP;        // edge-pair records (sum over the walls of ceil(edges / 2)); 0 when CAT_EDGE_PAIRS == 0
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
    int crow_base, pad2;        // into cgrid_rows (one packed 8-byte row per cell)
    int row_base, pad1;         // into grid_rows, in rows of Params::row_words words
};

// doubles of geometry after the wall bbs: the f64 plane records, then the f32 records
__host__ __device__ inline int geo_rest_doubles(const MapDesc &md) { return CAT_EDGE_PAIRS ? 8 * md.P + (kPairF / 2) * md.PP : kGeoPerPlane * md.P; }

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
    unsigned pool_magic; int pool_shift;   // ring position of entry i: i & pool_mask where the capacity is a power of two (pool_shift < 0), else i - capacity * (i / capacity)
                                           // with the quotient by multiply-high and shifts (unsigned division by an invariant, computed by cat_create)
};

// Problem dimensions as seen by the device code: either read from the parameter block (DynDims) or compile-time
// constants for the rosters / ray counts that are instantiated (FixDims): constant LDS offsets and unrolled agent loops.
// Round 4, as the compiler reports them (tools/regs.sh, profiles/r04_registers.txt): step_kernel / rollout_kernel 115 - 121 VGPRs and 9 - 17 spilled SGPRs fixed (23 - 41 generic),
// the pooled pair 113 - 115 and 20 - 21 (37 - 38), reset_kernel 103 - 107 VGPRs and 6 - 22; no scratch in any instantiation (rounds 1 - 3, tick_kernel: 127 - 128 VGPRs, 64 - 78
// spilled SGPRs fixed, 107 - 134 + 64 B of scratch generic -- see "opaque roots" below).
struct DynDims {
    static __device__ __forceinline__ int A(const Params &p) { return p.A; }
    static __device__ __forceinline__ int R(const Params &p) { return p.R; }
    static __device__ __forceinline__ int n_cops(const Params &p) { return p.n_cops; }
    static __device__ __forceinline__ int NP(const Params &p) { return p.NP; }
    static __device__ __forceinline__ int rec_bytes(const Params &p) { return p.rec_bytes; }
    static __device__ __forceinline__ int hot_bytes(const Params &p) { return p.hot_bytes; }
};
template <int TA, int TR, int TC> struct FixDims {
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
// ray-fan work units of an env slot (gsz: agents per unit of the group form)
template <class D> __device__ __forceinline__ int fan_units(const Params &p, int gsz)
{
    if constexpr (D::kFan == 1) return (D::A(p) + gsz - 1) / gsz;
    else return D::A(p) * ((D::R(p) + 63) / 64);
}
template <class D> __device__ __forceinline__ int fan_units(const Params &p) { return fan_units<D>(p, group_agents<D>(p)); }

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

#ifndef CAT_HW_MINMAX
#define CAT_HW_MINMAX 1
#endif
#if CAT_HW_MINMAX
#define CAT_FMAX(a, b) __builtin_fmax((a), (b))
#define CAT_FMIN(a, b) __builtin_fmin((a), (b))
#else
#define CAT_FMAX(a, b) fmax2((a), (b))
#define CAT_FMIN(a, b) fmin2((a), (b))
#endif

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
    const float *p32;      // [P][kP32F] f32 copy for the conservative pre-classification: n.x n.y c dtMin | dtMax v0.x v0.y - | pad
    const int *fc;         // [S] first plane | plane count << 16
    const int *fp;         // [S] first edge-pair record of the wall (CAT_EDGE_PAIRS)
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
#if CAT_EDGE_PAIRS
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
#else
    {
        const float *q = L.p32 + kP32F * first;
        for (int i = 0; i < count; i++, q += kP32F) {
            const float4 q0 = *reinterpret_cast<const float4 *>(q);       // n.x n.y c dtMin
            const float4 q1 = *reinterpret_cast<const float4 *>(q + 4);   // dtMax v0.x v0.y -
            const float d = __builtin_fmaf(ayf, q0.y, axf * q0.x) - q0.z;
            const float den = -__builtin_fmaf(dyf, q0.y, dxf * q0.x);
            bool face = (d >= -e1) && (d <= den + e1);
            {   // where the crossing point falls along the face (skipped for a ray almost parallel to it: ill-conditioned)
                const float ri = __builtin_amdgcn_rcpf(fmaxf(den, 0.25f));
                const float t = d * ri;
                const float ptx = __builtin_fmaf(t, dxf, axf), pty = __builtin_fmaf(t, dyf, ayf);
                const float dt = __builtin_fmaf(q0.x, pty, -(q0.y * ptx));
                const float e2 = e1 * __builtin_fmaf(3.0f * len, ri, 4.0f);
                face = face && ((den < 0.25f) || ((dt >= q0.w - e2) && (dt <= q1.x + e2)));
            }
            pm |= (unsigned)face << i;
            const float ex = q1.y - axf, ey = q1.z - ayf;
            const float cr = __builtin_fmaf(dxf, ey, -(dyf * ex)), sp = __builtin_fmaf(dxf, ex, dyf * ey);
            vm |= (unsigned)(bevels && !(fabsf(cr) > thr) && (sp >= s_lo) && (sp <= s_hi)) << i;
        }
    }
#endif
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
template <class D>
__device__ void agent_setup(const Lds &L, const Params &p, const GridDesc &gd, int lane)
{
    const int A = D::A(p), R = D::R(p);
    const double r2 = p.ray_radius;
    const double reach = p.ray_length + r2 + 1e-6;
    const double cone_m = p.gate ? 1e-6 : r2 + 1e-6;
    int my_cell = -1, my_near0 = -1, my_near1 = -1, my_dk0 = 0, my_dcnt = 0;
    // lane i < A: grid cell of agent i and the cell's packed contact row -- ONE global round trip for all agents
    unsigned long long crow = 0ull;
    if (lane < A) {
        const double ax = L.fpos[2 * lane], ay = L.fpos[2 * lane + 1];
        const int cx = (int)floor((ax - gd.x0) * gd.inv_cell), cy = (int)floor((ay - gd.y0) * gd.inv_cell);
        if (cx >= 0 && cy >= 0 && cx < gd.nx && cy < gd.ny) {
            my_cell = cy * gd.nx + cx;
            crow = G(p.cgrid_rows)[gd.crow_base + my_cell];
        }
    }
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
    // Wait for the LUT value HERE, while no store is in flight: a load still pending when the write-back stores
    // start makes the compiler's (in-order) vmcnt waits sit on the acknowledgement of every store issued before
    // them -- 18 k cycles per slot in the write-back before this line.
    asm volatile("" : "+v"(late.reward));
}

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

// All output stores, issued at the very end of the kernel: the compiler's s_waitcnt vmcnt(0) (in-order
// with stores, and forced by every flat access) would otherwise stall mid-kernel on HBM write latency.
template <class D>
__device__ __forceinline__ void emit_observations(const Lds &L, const Params &p, const LaunchArgs &la, long long env, int lane,
                                                  int rew_mode, const LateOut &late)
{
    const int A = D::A(p), R = D::R(p);
    if (rew_mode && lane < A && la.out.reward) la.out.reward[(size_t)env * A + lane] = late.reward;
    if (lane < 2 * A && la.out.team_positions) la.out.team_positions[(size_t)env * A * 2 + lane] = (unsigned short)late.tp16;
    // get_shared_observations (observation_spaces.py:98-129): first team member, roster order,
    // with a non-EMPTY ray supplies (type, distance); else EMPTY with the last member's distance
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
        dg.coff_base = g.coff_base; dg.cent_base = g.cent_base; dg.crow_base = g.crow_base; dg.pad2 = g.pad2; dg.row_base = g.row_base; dg.pad1 = g.pad1;
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
    d.gd.ent_base = s.gd.ent_base; d.gd.coff_base = s.gd.coff_base; d.gd.cent_base = s.gd.cent_base; d.gd.crow_base = s.gd.crow_base; d.gd.pad2 = s.gd.pad2;
    d.gd.row_base = s.gd.row_base; d.gd.pad1 = s.gd.pad1;
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
    {
        const Params &p = *(const Params *)launder(pk);
        const int S = md.S, A = D::A(p);
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
        }
    }
    {
        const Params &p = *(const Params *)launder(pk);
        agent_setup<D>(L, p, gd, lane);                              // entity.py:143-144, :388-390 (setup part)
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
            const int n_fan = fan_units<D>(p, group_agents_resident<D>(p));
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
            const int gsz = group_agents_resident<D>(p);   // also with one tick per launch: 3v2 x8192 98.8 against 100.5 us
            SSPREAD(slot, 2 + 2 * unit);
            if (unit < fan_units<D>(p, gsz)) {
#ifndef CAT_ABL_NOFAN      // diagnostic builds: a phase compiled out, for instruction counts by difference (tools/ablate_rollout.sh)
                if constexpr (D::kFan == 1) fan_group<D>(Ls, p, la, K->gd, eo, lane, uni(K->md.S), K->md.cmax, 1, unit, gsz, pc);
                else fan_chunk<D>(Ls, p, la, K->gd, eo, lane, uni(K->md.S), K->md.cmax, 1, unit, pc);
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
    agent_setup<D>(L, p, gd, lane);                                  // :334-344 (setup part)
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

}  // namespace

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
    const int rest = CAT_EDGE_PAIRS ? 8 * maxP + (kPairF / 2) * maxPP : kGeoPerPlane * maxP;
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
#if CAT_EDGE_PAIRS
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
#else
            std::vector<float> p32(kP32F * (size_t)d.P, 0.0f);
            for (int q = 0; q < d.P; q++) rec8(q, p32.data() + kP32F * (size_t)q);
            d.PP = 0;
            const size_t at = geo_f.size();
            geo_f.resize(at + (kP32F / 2) * (size_t)d.P);
            memcpy(geo_f.data() + at, p32.data(), p32.size() * sizeof(float));
#endif
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
            for (int attempt = 0; ok_dims && attempt < 3 && !pool_cap; attempt++) {
                const int cap = attempt == 0 ? cap2 : cap_x, g2 = attempt < 2 ? g_full : g_one;
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
