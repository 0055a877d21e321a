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
// One translation unit: this file includes cat_sim_{common,geometry,fan,physics,scheduler}.h inside its anonymous namespace, then cat_sim_host.h.
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
#include "cat_sim_common.h"
#include "cat_sim_geometry.h"
#include "cat_sim_fan.h"
#include "cat_sim_physics.h"
#include "cat_sim_scheduler.h"
}  // namespace

#include "cat_sim_host.h"
