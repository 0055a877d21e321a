/*
 * cat_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar, double-precision CPU restatement of the Cops-and-Thieves env hot path, used as the
 * checker for the HIP library (libcat_sim.so) and as bench.py's "cpu_baseline" leg.  Nothing
 * in the product package may include, link or call this.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in Pymunk/Chipmunk2D, a third-party
 * dependency that is NOT under /root/reference (requirements.txt:4, unpinned) and is not
 * installable here.  The Chipmunk side is restated from its published algorithm
 * (marked [CHIPMUNK-RECALL]); the reference's own Python side
 * (src/environments/base_env.py, src/agents/{entity,cop,thief}.py, src/environments/observation_spaces.py)
 * is restated line by line with file:line citations in cat_oracle.c.  What IS pinned: the
 * float16 observation/reward arithmetic against NumPy, the physical constants against the
 * reference's pyproject.toml, Philox against the Random123 known-answer vectors, and analytic
 * known-answer geometry cases (tests/).
 */
#ifndef CAT_ORACLE_H
#define CAT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CATO_MAX_AGENTS 8
#define CATO_WALL_CACHE 8     /* cached wall arbiters per agent */
#define CATO_MAX_PAIRS 28     /* A*(A-1)/2 at A = 8 */

/* ObjectType, reference src/utils/object_types.py:4-9 */
enum { CATO_WALL = 0, CATO_COP = 1, CATO_THIEF = 2, CATO_MOVABLE = 3, CATO_EMPTY = 4 };

typedef struct cato_config {
    int32_t n_envs;
    int32_t n_cops;
    int32_t n_thieves;
    int32_t n_rays;
    int32_t max_step_count;       /* base_env.py:56 / simple_env.py:19 */
    int32_t iterations;           /* cpSpace default 10 */
    int32_t persistence;          /* cpSpace collisionPersistence default 3 */
    int32_t bbtree_gate;          /* 1: segment queries visit a shape only if the THIN segment
                                     enters its spatial-index bb before the current best hit
                                     (Chipmunk BBTree behaviour); 0: visit every shape */
    int64_t env_id_offset;        /* global id of env slot 0 (multi-GPU sharding) */
    uint64_t seed;                /* Philox key */
    double dt;                    /* simple_env.py:20 -> 1/60 */
    double bias_coef;             /* 1 - pow(collisionBias, dt), computed by the host */
    double slop;                  /* collisionSlop 0.1 */
    double ray_length;            /* entity.py:84 */
    double ray_radius;            /* entity.py:196 */
    double agent_radius;          /* pyproject.toml unit_size */
    double agent_mass;            /* unit_mass */
    double impulse;               /* unit_velocity */
    double max_speed;             /* max_speed */
    double termination_radius;    /* termination_radius */
    double wall_radius;           /* map.py:127 radius=1 */
} cato_config;

/* Per-ray direction table built by the host with NumPy exactly as entity.py:182-193 does:
   lc[k] = ray_length*cos(angle_k), ls[k] = ray_length*sin(angle_k). */
typedef struct cato_tables {
    const double *ray_dx;         /* [R] */
    const double *ray_dy;         /* [R] */
    const float *cop_reward_lut;  /* [32768] indexed by the f16 bits of min THIEF distance */
    const float *thief_reward_lut;/* [32768] indexed by the f16 bits of min COP distance */
} cato_tables;

typedef struct cato_outputs {     /* any pointer may be NULL */
    uint16_t *obs_distance;       /* [N,A,R] f16 bits */
    uint8_t *obs_type;            /* [N,A,R] */
    int32_t *hit_shape;           /* [N,A,R] -1 none, s static, S+j agent j (debug/parity) */
    uint16_t *shared_distance;    /* [N,2,R] team 0 = cops, 1 = thieves */
    uint8_t *shared_type;         /* [N,2,R] */
    uint16_t *team_positions;     /* [N,A,2] f16 bits */
    float *reward;                /* [N,A] */
    uint8_t *terminated;          /* [N] capture OR timeout (entity.py:146) */
    uint8_t *truncated;           /* [N] timeout only (base_env.py:397) */
    int8_t *winner;               /* [N] -1 none, 0 cop, 1 thief (base_env.py:399-406) */
} cato_outputs;

typedef struct cato_state {       /* views for get/set; any pointer may be NULL */
    double *pos;                  /* [N,A,2] body.position */
    double *vel;                  /* [N,A,2] */
    double *vbias;                /* [N,A,2] */
    double *tc;                   /* [N,A,2] cached circle centre (stale after reset, quirk Q1) */
    double *leaf_bb;              /* [N,A,4] BBTree leaf bb l,b,r,t */
    int32_t *wall_shape;          /* [N,A,K] -1 = free slot */
    int32_t *wall_age;            /* [N,A,K] steps since last seen */
    double *wall_jn;              /* [N,A,K] cached jnAcc */
    int32_t *pair_age;            /* [N,NP] -1 = none */
    double *pair_jn;              /* [N,NP] */
    int32_t *step_count;          /* [N] */
    int32_t *reset_count;         /* [N] */
} cato_state;

typedef struct cato_sim cato_sim;

int cato_create(const cato_config *cfg, const cato_tables *tab, const void *const *map_blobs,
                const size_t *blob_sizes, int n_maps, const int32_t *slot_map_ids,
                cato_sim **out);
void cato_destroy(cato_sim *s);
const char *cato_last_error(void);

/* reset masked envs (mask NULL = all). positions NULL = Philox spawn sampling with rejection
   (base_env.py:123-166); else [N,A,2] injected spawn positions. */
int cato_reset(cato_sim *s, const uint8_t *mask, const double *positions, const cato_outputs *out);
int cato_step(cato_sim *s, const int32_t *actions, const cato_outputs *out);
int cato_get_state(cato_sim *s, const cato_state *dst);
int cato_set_state(cato_sim *s, const cato_state *src);
/* synthetic uniform actions in {0..3}: Philox(key=seed, ctr=(env_global, tick, agent, 0xAC710)) */
int cato_random_actions(cato_sim *s, uint64_t tick, int32_t *actions);
void cato_set_threads(int n);     /* OpenMP threads over envs; 1 = scalar port */
void cato_set_index_order(int on);/* 1 (default): segment queries visit shapes in index order (D2); 0: nearest-bb-first (diagnostic) */
void cato_count_order_dependence(int on);   /* diagnostic: start (and zero) / stop counting the queries whose result can depend on the visiting order at all */
void cato_order_dependence(long long out[4]);/* queries, order-dependent among the walls, among the agents, of those: by a tie of the two smallest alphas */

/* ---- elementary pieces, exported so tests can pin them individually ---- */
uint16_t cato_f64_to_f16(double x);
double cato_f16_to_f64(uint16_t h);
uint16_t cato_obs_distance_f16(double px, double py, double ox, double oy);
void cato_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* one segment query in env e as agent `self` (-1: no agent excluded). los != 0 -> walls only.
   returns hit shape (-1 none), writes alpha and point. */
int cato_segment_query(cato_sim *s, int env, int self, double ax, double ay, double bx, double by,
                       double r2, int los, double *alpha, double *point_xy);
/* diagnostic: cato_segment_query visits only the walls whose byte in mask[S] is non-zero (NULL: all walls, the default) */
void cato_set_wall_subset(const uint8_t *mask);
/* point_query_nearest(p, maxd) != None, as agent `self` */
int cato_point_query_any(cato_sim *s, int env, int self, double px, double py, double maxd);

#ifdef __cplusplus
}
#endif
#endif
