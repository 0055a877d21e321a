/*
 * cat_sim.h -- C ABI of libcat_sim.so, the MI355X (gfx950) batched Cops-and-Thieves env core.
 *
 * The reference has no FFI: its hot path sits behind the PettingZoo ParallelEnv Python API
 * (SURVEY.md section 8b).  This ABI is what a binding for that path binds instead of Pymunk;
 * each entry point names the reference interface it replaces (paths relative to
 * /root/reference).  Conventions: opaque handle, int status (0 ok, <0 error; message from
 * cat_last_error), no exceptions across the boundary, CALLER-OWNED DEVICE buffers, explicit
 * hipStream_t passed as void* (NULL = default stream), one host thread per handle, handles are
 * independent (one per GPU / per process).  There is no CPU fallback: every compute entry fails
 * with CAT_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef CAT_SIM_H
#define CAT_SIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAT_ABI_VERSION 1
#define CAT_MAX_AGENTS 8
#define CAT_MAX_RAYS 512
#define CAT_MAX_SHAPES 256
#define CAT_MAX_HULL_EDGES 31   /* edges of one convex wall */
#define CAT_MAX_ROLLOUT_TICKS 65536   /* ticks of one cat_rollout_fused launch */
#define CAT_WALL_CACHE 8        /* cached wall arbiters per agent; cat_create refuses a map on which an agent could need more */

enum {
    CAT_OK = 0,
    CAT_ERR_BAD_CONFIG = -1,
    CAT_ERR_BAD_MAP = -2,
    CAT_ERR_BAD_SLOT_MAP = -3,
    CAT_ERR_NO_DEVICE = -4,
    CAT_ERR_HIP = -5,
    CAT_ERR_BAD_ARG = -6
};

/* ObjectType -- src/utils/object_types.py:4-9 */
enum { CAT_WALL = 0, CAT_COP = 1, CAT_THIEF = 2, CAT_MOVABLE = 3, CAT_EMPTY = 4 };

/* Replaces the constructor arguments of SimpleEnv/BaseEnv (src/environments/simple_env.py:14-38,
   base_env.py:51-58), the constants of pyproject.toml:12-19, the sensor literals of
   src/agents/entity.py:84-86,196 and Chipmunk2D's cpSpace defaults (base_env.py:77). */
typedef struct cat_config {
    int32_t n_envs;
    int32_t n_cops;
    int32_t n_thieves;
    int32_t n_rays;
    int32_t max_step_count;
    int32_t iterations;
    int32_t persistence;
    int32_t bbtree_gate;          /* 1 = Chipmunk BBTree visiting rule for segment queries */
    int64_t env_id_offset;        /* global id of env slot 0 (env sharding across GPUs) */
    uint64_t seed;                /* Philox4x32-10 key */
    double dt;
    double bias_coef;             /* 1 - pow(collisionBias, dt), computed by the host */
    double slop;
    double ray_length;
    double ray_radius;
    double agent_radius;
    double agent_mass;
    double impulse;
    double max_speed;
    double termination_radius;
    double wall_radius;
} cat_config;

/* HOST pointers, copied at create.  ray_dx/dy[k] = ray_length * cos/sin(2*pi*k/R) built with
   NumPy as entity.py:182-193 builds them; reward LUTs indexed by float16 bits (cop.py:69-74,
   thief.py:63-69 evaluated by NumPy per float16 value). */
typedef struct cat_tables {
    const double *ray_dx;         /* [R] */
    const double *ray_dy;         /* [R] */
    const float *cop_reward_lut;  /* [32768] */
    const float *thief_reward_lut;/* [32768] */
} cat_tables;

/* DEVICE pointers; any may be NULL (that output is skipped). */
typedef struct cat_outputs {
    uint16_t *obs_distance;       /* [N,A,R] float16 bits -- Entity.get_observation "distance" */
    uint8_t *obs_type;            /* [N,A,R]              -- "object_type" */
    int32_t *hit_shape;           /* [N,A,R] -1 none, s wall, S+j agent j (parity/debug) */
    uint16_t *shared_distance;    /* [N,2,R] team 0 cops / 1 thieves -- distance_shared */
    uint8_t *shared_type;         /* [N,2,R]                          -- object_type_shared */
    uint16_t *team_positions;     /* [N,A,2] float16 bits             -- team_positions */
    float *reward;                /* [N,A] */
    uint8_t *terminated;          /* [N] capture or timeout (entity.py:146) */
    uint8_t *truncated;           /* [N] timeout only (base_env.py:397) */
    int8_t *winner;               /* [N] -1 none, 0 cop, 1 thief (base_env.py:399-406) */
} cat_outputs;

/* DEVICE pointers for state get/set (env checkpointing, parity tests); any may be NULL. */
typedef struct cat_state {
    double *pos;                  /* [N,A,2] body.position */
    double *vel;                  /* [N,A,2] body.velocity */
    double *vbias;                /* [N,A,2] Chipmunk v_bias */
    double *tc;                   /* [N,A,2] cached circle centre (stale after reset) */
    double *leaf_bb;              /* [N,A,4] BBTree leaf bb */
    int32_t *wall_shape;          /* [N,A,CAT_WALL_CACHE] */
    int32_t *wall_age;            /* [N,A,CAT_WALL_CACHE] */
    double *wall_jn;              /* [N,A,CAT_WALL_CACHE] */
    int32_t *pair_age;            /* [N,A(A-1)/2] */
    double *pair_jn;              /* [N,A(A-1)/2] */
    int32_t *step_count;          /* [N] */
    int32_t *reset_count;         /* [N] */
} cat_state;

typedef struct cat_sim cat_sim;

/* Replaces BaseEnv.__init__ (base_env.py:51-121): pymunk.Space(), Map.populate_space
   (src/maps/map.py:119-128, via compiled map blobs), _init_cops/_init_thieves (:168-214) and
   Entity.__init__ (entity.py:41-124).  map_blobs are HOST pointers to blobs produced by
   as_cops_and_thieves_amd.maps.CompiledMap.to_blob(); slot_map_ids[N] (HOST, NULL = all 0)
   selects the map of each env slot.  device = HIP device ordinal. */
int cat_create(const cat_config *cfg, const cat_tables *tables, const void *const *map_blobs,
               const size_t *blob_sizes, int n_maps, const int32_t *slot_map_ids, int device,
               cat_sim **out);
int cat_destroy(cat_sim *sim);
/* sim may be NULL: message of the last failed cat_create on this thread's library. */
const char *cat_last_error(const cat_sim *sim);

/* Replaces BaseEnv.reset (base_env.py:286-352) incl. _get_non_colliding_position (:123-166),
   Entity.reset (entity.py:148-157) and Space.point_query_nearest (call site :154).  mask
   (DEVICE, [N] u8, NULL = all) selects envs; positions (DEVICE, [N,A,2] f64, NULL = Philox
   spawn sampling) injects spawn positions.  Writes the post-reset observations. */
int cat_reset(cat_sim *sim, const uint8_t *mask, const double *positions, const cat_outputs *out,
              void *stream);
/* Same, with mask = "terminated flag set by the previous cat_step" (kept on device, no host
   sync): the auto-reset of the batched env. */
int cat_reset_done(cat_sim *sim, const cat_outputs *out, void *stream);

/* Replaces BaseEnv.step (base_env.py:354-413): _termination_criterion (:521-554),
   Entity.step/_perform_action/get_observation/_query_body (entity.py:126-241), Cop.reward /
   Thief.reward (cop.py:49-75, thief.py:48-69), get_shared_observations
   (src/environments/observation_spaces.py:67-131) and pymunk Space.step (call site :392).
   actions: DEVICE [N,A] int32 in {0,1,2,3}.  The launch is asynchronous, so an action outside 0..3 (the reference
   raises on it: entity.py:126-134 indexes its impulse table) cannot fail the call: it is applied as "no impulse" and
   raises CAT_DEVERR_BAD_ACTION in the handle's device-side error word -- read it with cat_device_errors. */
int cat_step(cat_sim *sim, const int32_t *actions, const cat_outputs *out, void *stream);

/* Device-side error flags raised by the kernels since the last clear (synchronises `stream`); flags != 0 also sets
   cat_last_error.  CAT_DEVERR_CONTACT_DROPPED: an agent touched more than CAT_WALL_CACHE walls in one step, or an env had
   more simultaneous contacts than cat_create proved possible for its maps (agents x the deepest overlap of wall bounding
   boxes one agent can reach, + agent pairs), and a contact got no constraint (Chipmunk's arbiter hash has no such limit).
   The asynchronous entries cannot fail on these: a caller reads the word where it needs the results to be valid (the
   reference raises at once: entity.py:126-134). */
#define CAT_DEVERR_BAD_ACTION 1u
#define CAT_DEVERR_CONTACT_DROPPED 2u
#define CAT_DEVERR_SCHEDULER 4u   /* internal: a work item of the pooled ray fan never arrived (the launch left instead of hanging); results of that launch are invalid */
int cat_device_errors(cat_sim *sim, uint32_t *flags, int clear, void *stream);

/* One-LAUNCH form of the rollout tick: cat_step, the auto-reset of cat_reset_done (auto_reset != 0) and,
   when actions == NULL, the synthetic Philox actions of cat_random_actions for tick `synth_tick`, all inside
   the step kernel.  Results are identical to the three separate calls: for an episode that ends with this
   tick, reward / terminated / truncated / winner are the terminal tick's and the observation buffers hold
   the first observations of the new episode (the terminal observations, which the separate calls compute
   and then overwrite, are not computed). */
int cat_step_fused(cat_sim *sim, const int32_t *actions, uint64_t synth_tick, int auto_reset,
                   const cat_outputs *out, void *stream);

/* RESIDENT ROLLOUT: T consecutive ticks of cat_step_fused in ONE launch -- the random-action phases of the reference's loops
   (src/driver.py:65-69 steps the env with sampled actions tick after tick; skrl's trainer does so for random_timesteps = 10000
   ticks, src/configs/mappo_config.py:9) and, with an action tape, any replay of fixed actions.  The map geometry is staged into
   LDS once, every env slot's state record stays in LDS for the T ticks and reaches HBM after the last one, and EVERY tick's
   outputs are written: each non-NULL pointer of `out` addresses a buffer with a leading T -- obs_distance [T,N,A,R], reward
   [T,N,A], terminated [T,N] ... -- whose row t holds exactly what cat_step_fused(actions_t, synth_tick0 + t, auto_reset)
   leaves in its [N,...] buffers after tick t.  actions: DEVICE [T,N,A] int32, or NULL = the synthetic Philox actions of
   cat_random_actions for ticks synth_tick0 .. synth_tick0 + T - 1.  1 <= T <= CAT_MAX_ROLLOUT_TICKS.  Bit-identical to T calls
   of cat_step_fused (outputs of every tick and the state afterwards). */
int cat_rollout_fused(cat_sim *sim, int T, const int32_t *actions, uint64_t synth_tick0, int auto_reset,
                      const cat_outputs *out, void *stream);

/* Env-state access (the reference cannot checkpoint env state). D2D copies. */
int cat_get_state(cat_sim *sim, const cat_state *dst, void *stream);
int cat_set_state(cat_sim *sim, const cat_state *src, void *stream);

/* Synthetic uniform actions (driver.py:68 samples action_space(agent) per agent):
   Philox(key = seed, counter = (env_global, tick, agent, 0xAC710)) & 3 -> DEVICE [N,A] int32. */
int cat_random_actions(cat_sim *sim, uint64_t tick, int32_t *actions, void *stream);

/* gymnasium-style reseeding (BaseEnv.reset(seed=...), base_env.py:307-311): replaces the Philox key
   used by later spawn sampling / synthetic actions. */
int cat_set_seed(cat_sim *sim, uint64_t seed, void *stream);

/* Measurement aid (bench.py's roofline leg): the NEXT cat_step / cat_step_fused launches its tick kernel with
 * this pair of hipEvent_t attached to the dispatch itself, so hipEventElapsedTime(start, stop) is the kernel's own
 * duration -- the figure rocprofv3 --kernel-trace reports -- without the inter-kernel dispatch gap that events
 * recorded around the call include.  One shot.  No reference counterpart (the reference has no timers on the path). */
int cat_arm_kernel_timing(cat_sim *sim, void *start_event, void *stop_event);

/* Introspection */
int cat_abi_version(void);
/* Names of the kernels cat_step / cat_step_fused and cat_rollout_fused launch ("step_kernel" / "rollout_kernel": one tick / T ticks through the resident
   scheduler; "step_kernel_pooled" / "rollout_kernel_pooled" where cat_create chose the pooled ray fan for that entry): what a profile lists.  No reference
   counterpart. */
const char *cat_one_tick_kernel(const cat_sim *sim);
const char *cat_rollout_kernel(const cat_sim *sim);
/* Chunk form of the ray fan (dense maps): how many 64-ray chunks of an env slot one work unit traces with one shared item list, in the resident launch
   (resident != 0) or the one-tick launch; 1 = chunk by chunk.  A sim of two parts: the chunk-form part's figure.  No reference counterpart. */
int cat_chunks_per_unit(const cat_sim *sim, int resident);
int cat_num_agents(const cat_sim *sim);
int cat_num_shapes(const cat_sim *sim, int map_index);
/* Timing hook for bench.py: seconds spent in the last n recorded step launches are measured by
   the caller with HIP events on `stream`; this returns the stream the handle would use for
   NULL (always the device's default stream). */

/* Test hook: the host copy of the spatial-hash broadphase tables built at cat_create.  k >= 0:
   candidate wall ids of ray k for an origin at (x, y); k < 0: contact candidates of that cell.
   Returns the list length and writes up to max_out ids. */
int cat_debug_grid_lookup(const cat_sim *sim, int map_index, double x, double y, int k, int *out, int max_out);

/* Host-only construction/lookup of one map's tables (no device): lets the CPU test-suite verify that
   the tables are supersets of the exact bb gate.  cell <= 0 selects the default cell size. */
typedef struct cat_grid_host cat_grid_host;
int cat_grid_build_host(const cat_config *cfg, const cat_tables *tables, const void *map_blob, size_t blob_size,
                        double cell, cat_grid_host **out);
int cat_grid_lookup_host(const cat_grid_host *grid, double x, double y, int k, int *out, int max_out);
long long cat_grid_bytes_host(const cat_grid_host *grid);
void cat_grid_free_host(cat_grid_host *grid);

/* The most wall bounding boxes the bb of one agent circle (radius agent_radius) can overlap at once on this map (host only): the
   bound cat_create holds against CAT_WALL_CACHE -- a map where it is larger is refused (CAT_ERR_BAD_MAP) instead of dropping a
   contact at run time (CAT_DEVERR_CONTACT_DROPPED stays as the backstop).  Negative: an error code. */
int cat_map_wall_bb_depth_host(const void *map_blob, size_t blob_size, double agent_radius);

/* Device arithmetic self-test used by tests: out[i] = op(in_a[i], in_b[i]) evaluated on the GPU
   with the same primitives the kernels use (op 0 sqrt(a), 1 a/b, 2 f64->f16 bits of a,
   3 obs-distance f16 bits of (a,b) relative to the origin (0,0)).  DEVICE pointers. */
int cat_selftest_arith(int op, const double *in_a, const double *in_b, double *out, int n, int device,
                       void *stream);

#ifdef __cplusplus
}
#endif
#endif
