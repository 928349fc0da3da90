/* mate_engine.h -- C ABI of the MI355X-native batched MultiAgentTracking step engine.
 *
 * The upstream reference (XuehaiPan/mate) has no FFI: the boundary its callers use is the
 * Python class mate.environment.MultiAgentTracking (mate/environment.py:288) reached through
 * mate.make (mate/__init__.py:24-46).  This header is the C-ABI a binding for that class
 * sits on; every entry point names the reference method it stands in for.  Plain pointers
 * and sizes only: all `*_dev` pointers are DEVICE pointers (e.g. torch `.data_ptr()`),
 * caller-owned, never freed by the engine.  `stream` is a hipStream_t passed as void*
 * (NULL = the default stream); calls on one handle must be serialised by the caller.
 * Every function returns 0 on success or a negative MATE_E* code; the message is available
 * from mate_engine_last_error().  One handle drives N independent environments on one GPU.
 *
 * Synchronisation.  Launching entry points only enqueue on `stream`.  The host-side accessors without a stream
 * argument (seed, lut_read / lut_write(_outer), set_obs_transform, set_obs_mode, set_action_grids,
 * enable_outer_boundary, idle_steps, kernel_time) wait for the stream of the handle's MOST RECENT launch
 * (hipStreamSynchronize), never for the whole device: other streams of the caller keep running.  Exception: a caller that
 * launched through this handle on MORE than one stream since the last such wait, or that has destroyed that stream, gets
 * hipDeviceSynchronize instead -- an accessor never touches engine memory under a launch in flight.
 *
 * Environment switches.  All are read ONCE, in mate_engine_create(), and fixed for the life of the handle; none
 * changes results (each selects between implementations the tests hold bit-identical, except MATE_ZOOM_ITERATE,
 * see below).  They exist for tests and measurements:
 *   MATE_GENERIC=1           run the generic kernels even for a (cameras, targets, obstacles) shape with a compiled
 *                            specialisation (mate_layout.specialised reports which one runs).  Compiled: the shapes of all
 *                            seventeen scenarios the reference ships (csrc/shape_groups.hpp); a generic fused rollout runs
 *                            at less than half the rate of a specialised one
 *   MATE_FLOW_GENERIC=1      run the kernel that reads every launch switch at run time instead of the ones compiled
 *                            for the common flows (mate_engine_last_flow)
 *   MATE_STAGGER=<digits>    wave priorities of the single-step kernel at its five phase boundaries, one decimal digit
 *                            (0-3) each, e.g. 33210; default 0 = off (round 4: they cost 3 % once the rows leave early)
 *   MATE_ROLLOUT_ROTATE=0    no per-step rotation of the wave priorities in the fused rollout kernels (default 1; 8..20: the
 *                            turn follows the shader clock >> n instead of the wave's step count -- measured, no gain)
 *   MATE_RESET_MONOLITHIC=1  whole-batch / masked / batched resets as ONE launch instead of placement, per-camera
 *                            occlusion tables and first view as separate launches
 *   MATE_LUT_SMALL_CAP=<n>   ray capacity of the small-LDS occlusion-table launch (default: half of the worst case);
 *                            tables with more rays are built by the full-size launch behind it
 *   MATE_NO_IMAGE=1          the fused rollouts pack observations through the descriptor table even for a shape with a
 *                            row-image compilation (same rows; tests compare the two)
 *   MATE_POLICY_SPLIT=1      mate_engine_step_greedy / _step_versus_greedy as two launches (the agents' kernel, then the step
 *                            kernel) even where the fused one-launch form applies (same results; tests compare the two)
 *   MATE_STEP_GREEDY_ROLLOUT=1  the one-launch form of mate_engine_step_greedy / _step_versus_greedy on the fused rollout kernel with a
 *                            single step (rounds 3-4) instead of step_greedy_kernel (policy_kernels.hpp: the per-step kernel's own
 *                            sequence with the agents in front; same results, tests compare the two)
 *   MATE_STEP_SPLIT=0|1      the per-step kernel of the folded flows (step_random / step with real-valued actions, f32 observations)
 *                            as one wave per environment (0) or two (1: cameras, sector tests and goals on one wave, targets
 *                            and range tests on the other; engine_kernels.hpp: step_split_kernel); default: by what measured
 *                            faster at the batch size (profiles/HISTORY.md 3.1d); the same bytes either way (tests compare the two)
 *   MATE_ZOOM_ITERATE=1      the on-device GreedyCameraAgent runs the reference's 20-iteration zoom solve
 *                            (mate/agents/greedy.py:139-145) instead of reading its tabulation; the two differ by
 *                            <= 1.5e-13 degrees in the viewing angle (parity runs that want the iteration itself)
 *   MATE_BLOCK_FREE_RANGE=1  mate_engine_block_free also gives the block's virtual address range back (hipMemAddressFree) instead
 *                            of keeping it reserved for the life of the process (read once, at the first block_free).  Unsafe on
 *                            this driver: a range handed out again lost stores of the next kernel (tools/va_reuse.hip)
 *   MATE_BLOCK_DEAD_GIB=g    address space [GiB] that freed blocks may keep reserved per process (default 4096 of the 131072 a process
 *                            has): beyond it mate_engine_block_alloc fails with MATE_ENOMEM and the caller uses plain device memory
 *                            (read once, at the first block_alloc)
 *   MATE_PIPELINED_SERIAL=1  pipelined restarts (MATE_RESET_PIPELINED) with the resets on the caller's stream: the reference form
 *                            the tests compare the concurrent one with; MATE_PIPELINED_PRIORITY=0: the side stream at the default
 *                            priority instead of the device's lowest (both read when the mode is first entered)
 *   MATE_SUBWAVE=0|1         environments per wave of the fused rollouts of the small scenarios (mate_engine_set_sub_wave below): 0 = always
 *                            one, 1 = the shape's number (four; two for MATE-4v8-0's Greedy flows) in every such launch; default: where it
 *                            measured faster.  MATE_STEP_SUBWAVE=0: the per-step mate_engine_step / _step_random / _step_greedy /
 *                            _step_versus_greedy keep the per-step kernels where the fused flows run sub-wave groups (default: they follow,
 *                            as one-step launches of the same kernels).  Same bytes either way.
 * Read by the Python host: MATE_ENGINE_LIB=<path> (mate_amd/_native.py: another build of this library, e.g. the profiling build
 * lib/libmate_engine_prof.so); MATE_BUILD_JOBS=<n> (mate_amd/build.py: parallel hipcc processes, default one per translation
 * unit up to the CPU count); and (mate_amd/engine.py), once, when an Engine object is built -- they steer where
 * Engine.reserve_rollout puts the [steps][N][...] observation blocks of the fused rollouts, never what is written there:
 *   MATE_PLAIN_BLOCKS=1      blocks from torch.zeros instead of mate_engine_block_alloc
 *   MATE_BLOCK_CANDIDATES=n  at most n candidates probed per block (default 3, allocated side by side; 6 and -- for the target block --
 *                            as many as the memory and time bounds allow in a deep search; n <= 1: one block, unprobed)
 *   MATE_BLOCK_DEEP=1        the deep search (candidates separated by unmapped 12 GB spacers: a walk through the device's memory)
 *                            for the target block even when reserve_rollout is not asked for it (search='deep': bench.py does;
 *                            off by default since round 5 -- a co-resident learner should not see 45 % of the HBM vanish for seconds)
 *   MATE_BLOCK_SECONDS=s     wall-time bound of the search (default 0.3; deep: 3)
 *   MATE_BLOCK_GIB=g         bound of the deep search's transient footprint in GiB (default 96; always at most 45 % of the free memory)
 *   MATE_STORE_FORM=0|1      force the form of the row stores (mate_engine_set_store_form) instead of choosing by the probed rate
 */
#ifndef MATE_ENGINE_H
#define MATE_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MATE_ABI_VERSION 1

enum {
    MATE_OK = 0,
    MATE_EINVAL = -1,   /* bad argument / configuration (reference: ValueError / AssertionError) */
    MATE_EHIP = -2,     /* a HIP runtime call failed */
    MATE_ENOMEM = -3,
    MATE_ESTATE = -4    /* call sequence error (e.g. step before reset) */
};

enum { MATE_OBS_F32 = 0, MATE_OBS_F64 = 1 };
enum { MATE_ACT_F32 = 0, MATE_ACT_F64 = 1,
       /* OR-ed into act_dtype: that team's joint action is int32 grid indices [N][agents] instead of [N][agents][2]
        * reals (DiscreteCamera / DiscreteTarget, mate/wrappers/discrete_action_spaces.py:59-74, 165-180);
        * the grids come from mate_engine_set_action_grids */
       MATE_ACT_CAMERA_DISCRETE = 0x100, MATE_ACT_TARGET_DISCRETE = 0x200 };

/* Scenario description = the validated YAML/JSON/dict configuration of the reference
 * (mate/environment.py:113-269: read_config + validate_config), flattened. */
typedef struct mate_config {
    int32_t num_cameras, num_targets, num_obstacles;
    int32_t max_episode_steps;            /* environment.py:199-203 */
    int32_t sparse_reward;                /* reward_type == 'sparse', environment.py:526 */
    int32_t num_cargoes_per_target;       /* environment.py:223-229 */
    int32_t shuffle_entities;             /* environment.py:253-256 */
    int32_t targets_start_with_cargoes;   /* environment.py:240-243 */
    double high_capacity_target_split;    /* environment.py:231-238 */
    double bounty_factor;                 /* environment.py:245-251 */
    double transmittance;                 /* obstacle/transmittance, environment.py:1470-1476 */
    double camera_radius, camera_min_viewing_angle, camera_max_sight_range;
    double camera_rotation_step, camera_zooming_step;   /* entities.py:248-254 */
    double target_step_size, target_sight_range;        /* entities.py:563-566 */
    double obstacle_radius_range[2];      /* radius_random_range (lo, hi) */
    const double *camera_location_ranges;   /* [num_cameras][4]  = x_lo, x_hi, y_lo, y_hi (host memory) */
    const double *target_location_ranges;   /* [num_targets][4] */
    const double *obstacle_location_ranges; /* [num_obstacles][4] */
    int32_t obs_dtype;                    /* MATE_OBS_F32 (default product path) or MATE_OBS_F64 */
} mate_config;

/* Sizes and the bit layout of the packed visibility masks, for the binding. */
typedef struct mate_layout {
    int32_t camera_obs_dim, target_obs_dim, state_dim;   /* constants.py:267-300, environment.py:450-466 */
    int32_t mask_words;          /* u32 words per environment in the packed mask export */
    int32_t bit_camera_target;   /* bit(c,t)  = bit_camera_target + c*Nt + t       (environment.py:1364-1367) */
    int32_t bit_camera_camera;   /* bit(c,c') = bit_camera_camera + c*Nc + c'      (environment.py:1381-1386) */
    int32_t bit_target_row;      /* bit(t,j)  = bit_target_row + t*(Nc+No+Nt) + j, j: cameras, obstacles, targets */
    int32_t bit_camera_obstacle; /* bit(c,o)  = bit_camera_obstacle + c*64 + o      (environment.py:752-755) */
    int32_t export_width;        /* doubles per environment in mate_engine_export_state */
    int32_t lut_capacity;        /* max knots per camera occlusion table */
    int32_t scalars_per_env;     /* floats per environment in the step scalar record (8) */
    int32_t specialised;         /* 1 = step kernels compiled for exactly this (Nc, Nt, No) run; 0 = generic kernels */
} mate_layout;

/* Per-step outputs (any pointer may be NULL to skip that output).
 *   scalars_dev: [N][8] f32 = camera_team_reward, target_team_reward, done, coverage_rate,
 *                real_coverage_rate, mean_transport_rate, num_delivered_cargoes,
 *                normalized target_team_reward   (environment.py:618-661)
 *   masks_dev:   [N][mask_words] u32 packed view masks + tracked bits (environment.py:1356-1388) */
typedef struct mate_step_io {
    const void *camera_actions_dev;   /* [N][Nc][2]  f32 or f64 (act_dtype) */
    const void *target_actions_dev;   /* [N][Nt][2] */
    int32_t act_dtype;
    const double *tape_camera_target_dev; /* [N][Nc][Nt] uniforms for the see-through draw, or NULL = Philox */
    const double *tape_goal_dev;          /* [N][Nt] uniforms for the goal choice, or NULL = Philox */
    void *camera_obs_dev;             /* [N][Nc][Dc] f32/f64 (obs_dtype) */
    void *target_obs_dev;             /* [N][Nt][Dt] */
    float *scalars_dev;
    uint32_t *masks_dev;
} mate_step_io;

typedef struct mate_engine mate_engine;

const char *mate_engine_last_error(void);
int mate_engine_abi_version(void);

/* MultiAgentTracking.__init__ (environment.py:330-562) for N environments on HIP device `device`.
 * RNG streams are keyed by (seed, first_env_index + i), so results do not depend on how a
 * global batch is sharded over GPUs.
 * Limits (MATE_EINVAL beyond them, never a silent fallback): at most 16 cameras, 16 targets, 64 obstacles (bit widths of
 * the packed records).  The occlusion-table build sorts 360 + 185 * obstacles rays per camera: in the 160 KiB LDS up
 * to 20 obstacles, in an HBM scratch slice per workgroup beyond (same code, slower resets; the reference's scenarios
 * use 0 or 9 obstacles with cameras, MATE-Navigation has 32 and no camera). */
int mate_engine_create(const mate_config *config, int64_t num_envs, int32_t device, uint64_t seed,
                       uint64_t first_env_index, mate_engine **out);
int mate_engine_destroy(mate_engine *engine);                               /* close(), environment.py:1192 */
int mate_engine_get_layout(const mate_engine *engine, mate_layout *out);
/* seed() (environment.py:1203-1227): installs the Philox key AND rewinds every counter that enters a counter word
 * (per-environment episode number, step tick), as the reference re-creates its RandomStates: after seed(s) the next
 * reset() and everything that follows depend on (s, global environment index) only, whatever ran before. */
int mate_engine_seed(mate_engine *engine, uint64_t seed);

/* reset() (environment.py:679-834) of every environment (env_mask_dev == NULL) or of those with a
 * non-zero byte in env_mask_dev[N].  Writes the initial observations/masks like a step does. */
int mate_engine_reset(mate_engine *engine, const uint8_t *env_mask_dev, const mate_step_io *io, void *stream);

/* reset() with every random draw of environment.py:679-834 on a tape (parity runs against the reference):
 * `tape_dev` [N][tape_len] uniforms in the reference's call order -- shuffles of the three entity lists, the
 * high-capacity choice, per placement attempt (camera: radius box, x, y, orientation index, viewing angle; obstacle:
 * radius, x, y; target: x, y), cargo pairs, goal choices of targets that spawn inside a warehouse, and per still
 * unloaded target a warehouse permutation + goal choice; how a uniform becomes a shuffle / choice / integer is stated
 * in oracle/mate_oracle.c above reset_impl (Fisher-Yates, a + int(u * n)).  io->tape_camera_target_dev, if set,
 * supplies the see-through uniforms of the first _update_view (environment.py:766).  `draws_used_dev` (optional,
 * [N] int32) receives the number of uniforms each environment consumed, -1 if its tape ran out.  tests/golden/
 * reset_*.npz hold such tapes recorded from the reference's own reset() together with the state it produced. */
int mate_engine_reset_tape(mate_engine *engine, const uint8_t *env_mask_dev, const mate_step_io *io,
                           const double *tape_dev, int32_t tape_len, int32_t *draws_used_dev, void *stream);

/* step() (environment.py:590-676).  auto_reset == 1: environments whose episode ended are reset in
 * the same call and their observation rows hold the first observation of the new episode (rewards/done
 * in `scalars_dev` still describe the finished step).  auto_reset == k > 1: batched resets -- a
 * finished environment idles (its scalar record reads done = 2, no new observation) until every k-th
 * call restarts all finished environments together, which amortises the reset latency (occlusion-table
 * build) when episodes end every step somewhere in the batch.  auto_reset == 0: the caller resets. */
int mate_engine_step(mate_engine *engine, const mate_step_io *io, int32_t auto_reset, void *stream);

/* Episode statistics for logging: `stats_dev` = 5 doubles in caller-owned device memory (or NULL to stop), to which
 * every later step / rollout launch adds, for each episode that ends in it: 1, the target team's episode reward
 * (environment.py:624), the episode length (:629), the final coverage_rate (:966) and num_delivered_cargoes.  This is
 * the record the sharded job all-gathers over RCCL; the engine only accumulates it, with a
 * handful of atomics per finished episode.  The caller zeroes / reads the buffer on its own streams. */
int mate_engine_set_episode_stats(mate_engine *engine, double *stats_dev);
/* ... and a snapshot of those 5 doubles into `dst_dev`, ordered on `stream` behind everything enqueued on it so far (one tiny launch):
 * the buffer a sharded job all-gathers while later launches go on accumulating (bench.py, StatsGather).  No reference counterpart:
 * the reference's trainers log episode returns on the host (SURVEY.md section 8e defines the record). */
int mate_engine_snapshot_episode_stats(mate_engine *engine, double *dst_dev, void *stream);

/* Graph-replayable stepping (the learner-in-the-loop flow: policy kernels write the joint actions into caller
 * buffers, step() consumes them, K such iterations are captured once in a HIP graph and replayed).  A step() launch
 * normally carries the step counter (the Philox tick) and the ping-pong index of the finished-episode lists as launch
 * arguments, which change at every step.  enable = k >= 1 moves both to device memory: the step kernels read them there
 * and the auto-reset launch that closes every interval of k steps advances them, so a whole interval (k step launches +
 * one auto-reset launch; k = 1: a (step, auto-reset) pair) has identical arguments every time and any number of
 * intervals may be captured (hipStreamBeginCapture on `stream`, or torch.cuda.graph) and replayed.  Results are
 * bit-identical to the host-counted flow with the same auto_reset.  While enabled only step() / step_random() /
 * step_greedy() / step_versus_greedy() with auto_reset = k, observe() and reset() are accepted (MATE_ESTATE otherwise) -- and
 * mate_engine_rollout_versus_greedy (FrameSkip in a graph): there k counts LAUNCHES, every launch of an interval has the same
 * number of frames K, and the auto-reset launch behind the k-th advances the counter by k * K.  enable == 0 (at an interval boundary)
 * drains `stream` and takes the counter back to the host.  The reference has no counterpart (environment.py:590 runs one
 * Python call per step); this is how its `for t in range(T): env.step(policy(obs))` loop is enqueued on a GPU. */
int mate_engine_device_tick(mate_engine *engine, int32_t enable, void *stream);

/* step() with the uniform random policy of SURVEY.md section 8d generated on-device
 * (camera U[-rot,rot] x U[-zoom,zoom], target U[-v,v]^2, Philox keyed by seed/env/tick);
 * io->*_actions_dev are ignored. */
int mate_engine_step_random(mate_engine *engine, const mate_step_io *io, int32_t auto_reset, void *stream);

/* `steps` consecutive step() calls under the on-device random policy fused into ONE launch (rollout
 * collection, the `for _ in range(T): env.step(...)` loop of examples/random.py).  Every io output
 * buffer is rollout-shaped: [steps][N][...] (row r*N + i = step r of environment i).  An environment
 * whose episode ends at step r stops there: its scalar rows of later steps carry done = 2 and its
 * observation rows are left untouched; with auto_reset it starts a new episode before the next call.
 * Those skipped slots are added to mate_engine_idle_steps.  auto_reset == k > 1 batches the restarts as step() does: a
 * finished environment idles through the following launches until every k-th call restarts all finished environments in
 * one reset launch (episodes that end every few steps somewhere in the batch, e.g. Greedy vs Greedy, otherwise pay one
 * latency-bound reset launch per rollout launch).  This is the fastest way to step: the records stay in LDS
 * for the whole launch and the waves of a SIMD take turns in issue priority (MATE_ROLLOUT_ROTATE=0 turns that off). */
int mate_engine_rollout_random(mate_engine *engine, const mate_step_io *io, int32_t steps, int32_t auto_reset, void *stream);

/* On-device rule-based policies: the reference's GreedyCameraAgent / GreedyTargetAgent
 * (mate/agents/greedy.py:13-227, 229-365) for every agent of every environment, acting on the same
 * partial observations (own state, opponents gated by the view masks of the previous step, teammates'
 * messages).  mate_engine_policy_enable() must precede the reset()/step() whose view they first act on.  Any scenario the
 * engine takes (up to 16 cameras and 16 targets): the camera agents' message exchange runs one lane per sender-recipient pair,
 * in as many rounds of 64 pairs as the cameras need; the fused rollouts need the workgroup's four environments -- step
 * records + agents' memory -- to fit the 160 KiB LDS (MATE_EINVAL otherwise, mate_engine_step_greedy still works).
 * mate_engine_step_greedy() = group_step of both teams (observe, two-phase message exchange, act;
 * mate/wrappers/single_team.py:79-92) + step().  `tape` (device arrays, any member NULL = Philox):
 * recorded draws of the agents, for parity runs.  (mate_engine_policy_enable also tabulates the camera agents' 20-iteration
 * zoom solve, greedy.py:139-145 -- a function of one scalar -- with the reference's own iteration: 225 KB, read with cubic
 * interpolation to 1.5e-13 degrees; the agents' memory, joint actions and a copy of the view masks are allocated there too.) */
typedef struct mate_policy_tape {
    const double *camera_resample_u_dev;     /* [N][Nc]      Bernoulli(0.1) uniform when no target is remembered (greedy.py:93) */
    const double *camera_sample_u_dev;       /* [N][Nc][2]   action_space.sample() uniforms (greedy.py:94) */
    const int32_t *camera_delay_dev;         /* [N][Nc][Nc]  randint(6, 50) per sent message sender->recipient (greedy.py:184) */
    const double *target_choice_u_dev;       /* [N][Nt]      choice among non-empty warehouses (greedy.py:298) */
    const double *target_resample_u_dev;     /* [N][Nt]      Bernoulli(prob) uniform (greedy.py:316) */
    const double *target_sample_u_dev;       /* [N][Nt][2]   noise sample uniforms (greedy.py:317) */
    const double *target_reset_sample_u_dev; /* [N][Nt][2]   initial noise at agent.reset (greedy.py:277) */
} mate_policy_tape;
int mate_engine_policy_enable(mate_engine *engine);
int mate_engine_step_greedy(mate_engine *engine, const mate_step_io *io, const mate_policy_tape *tape,
                            int32_t auto_reset, void *stream);
/* MultiCamera / MultiTarget (mate/wrappers/single_team.py:180-306, SingleTeamMultiAgent.step :245-264): the caller -- the
 * learner of examples/ippo|mappo|qmix|... -- plays ONE team, the on-device greedy agents play the other:
 * `group_step(env, opponent_agents, ...)` + `env.step((action, opponent_joint_action))` -- ONE launch (step_greedy_kernel: the opponents'
 * agents, then the step) unless a policy tape, a step tape, a fused observation transform / team mode, f64 observations or a missing output
 * asks for the two-launch form.  Only the OPPONENTS' agents act, as in the reference's wrapper (which holds no agents for the learner's team):
 * the caller's team's agent memory is left as it is, and a team's agent.reset(observation) runs at its own first acting call of an episode.  `team` is the
 * CALLER's team; io->camera_actions_dev (MATE_TEAM_CAMERA) or io->target_actions_dev (MATE_TEAM_TARGET) holds its joint
 * action in the encoding io->act_dtype names (f32 / f64 pairs, or grid indices with that team's *_DISCRETE bit); the other
 * action pointer is ignored.  The opponents observe, exchange messages and act exactly as in mate_engine_step_greedy
 * (their draws come from the same Philox streams / `tape`), so feeding the greedy agents' own recorded joint action for
 * `team` reproduces the step_greedy trajectory bit for bit (tests/test_gpu_policies.py). */
enum { MATE_TEAM_CAMERA = 0, MATE_TEAM_TARGET = 1 };           /* mate/utils.py Team */
int mate_engine_step_versus_greedy(mate_engine *engine, int32_t team, const mate_step_io *io, const mate_policy_tape *tape,
                                   int32_t auto_reset, void *stream);
/* `steps` consecutive (agents act, environment steps) iterations of GreedyCameraAgent vs GreedyTargetAgent fused into ONE
 * launch: mate.group_step + env.step of the evaluation loop (mate/evaluate.py:104-139, agents/greedy.py), with the
 * environment records, the agents' memory and the view masks resident in LDS.  Outputs are rollout-shaped like
 * mate_engine_rollout_random ([steps][N][...]); observations and scalars are required.  An environment whose episode ends
 * stops there (done = 2 in its later scalar rows, counted by mate_engine_idle_steps) and, with auto_reset, starts a new
 * episode before the next call.  Philox draws only (no policy tape).  Bit-identical to `steps` calls of
 * mate_engine_step_greedy. */
/* auto_reset == MATE_RESET_PIPELINED: restarts taken off the critical path.  Greedy-vs-Greedy episodes last ~1.2 k steps, so a few
 * hundred of 8192 environments finish in every launch and their reset -- placement, one occlusion table per camera, first view:
 * three latency-bound launches, ~0.25 ms -- otherwise sits between two rollout launches.  Here the reset of what launch n
 * finished is enqueued on a stream of the engine's own behind launch n and runs UNDER launch n + 1; a restarted environment joins
 * launch n + 2 (its `done` word carries a hand-over tag: no launch ever steps or stores an environment a concurrent reset owns,
 * so results do not depend on timing -- the same calls with MATE_PIPELINED_SERIAL=1, which runs the resets on the caller's
 * stream, give the same bytes).  A finished environment idles through the rest of its launch and all of the next one (scalar
 * rows done = 2, counted by mate_engine_idle_steps).  Any other entry point of the handle first waits for the resets in flight
 * and turns the tags back into plain live environments.
 * auto_reset = -m (m > 1): ONE such restart launch behind every m-th rollout launch (short launches -- a learner's FrameSkip actions --
 * whose restart group is longer than a launch): what finishes in interval i of m launches idles through i + 1 and is live again in
 * the first launch of i + 2; leaving the mode inside an interval restarts what it had listed.  (Measured on FrameSkip(5) launches of
 * MATE-4v8-9 x 4096 / 16 384: 65.5 against 66.3 us per launch and 1.3 % more idle slots -- no gain: the launch fills its registers'
 * worth of every SIMD and the restart's workgroups wait for its tail either way.  tools/archive/frameskip_pipelined_probe.py) */
#define MATE_RESET_PIPELINED (-1)
int mate_engine_rollout_greedy(mate_engine *engine, const mate_step_io *io, int32_t steps, int32_t auto_reset, void *stream);
/* FrameSkip(frame_skip = steps) over MultiCamera / MultiTarget (examples/utils/wrappers.py:301-323: the same action for
 * `frame_skip` env.step calls; every example trainer's make_env applies it last) in ONE launch: the caller's `team` repeats
 * the joint action of io (as in mate_engine_step_versus_greedy) for `steps` frames, the greedy opponents act anew on every
 * frame.  Outputs are rollout-shaped; the wrapper's summed reward is the column sum of the scalar rows (rows with
 * done = 2 are zero), its observation the last row with done != 2.  Bit-identical to `steps` calls of
 * mate_engine_step_versus_greedy with the same action.  Graph-replayable under mate_engine_device_tick (see there). */
int mate_engine_rollout_versus_greedy(mate_engine *engine, int32_t team, const mate_step_io *io, int32_t steps,
                                      int32_t auto_reset, void *stream);

/* copies the joint actions of the last step_greedy into caller buffers [N][Nc][2] / [N][Nt][2] f64 (either may be NULL) */
int mate_engine_policy_actions(mate_engine *engine, double *camera_actions_dev, double *target_actions_dev, void *stream);

/* joint_observation() (environment.py:908-983) without advancing the simulation: recomputes
 * the view masks from the current state (see-through draws from io tape or Philox) and packs. */
int mate_engine_observe(mate_engine *engine, const mate_step_io *io, void *stream);

/* Fuse the reference's observation post-processing wrappers into the packer (no second pass over the
 * observation bytes): relative != 0 = RelativeCoordinates (mate/agents/utils.py:40-94: coordinates of
 * warehouses and of visible entities minus the observing agent's own location); scale/bias per column
 * ([camera_obs_dim] / [target_obs_dim], host arrays, NULL = identity) = any affine per-column map, e.g.
 * RescaledObservation (utils.py:97-137).  out = ((value - own) if visible else 0) * scale + bias.
 * All NULL / 0 restores the plain observations. */
int mate_engine_set_obs_transform(mate_engine *engine, int32_t relative, const double *camera_scale,
                                  const double *camera_bias, const double *target_scale, const double *target_bias);

/* Team observation modes, fused into the packer as well: 0 plain; 1 = EnhancedObservation
 * (mate/wrappers/enhanced_observation.py:72-126: every opponent / obstacle / teammate block visible, targets see
 * the true emptiness of all warehouses); 2 = SharedFieldOfView (mate/wrappers/shared_field_of_view.py:72-148:
 * an opponent or obstacle is visible to the whole team when any member sees it, teammates always visible,
 * targets share their empty-warehouse knowledge).  The exported view masks are not affected, like in the
 * reference (the wrappers only rewrite observations). */
#define MATE_OBS_PLAIN 0
#define MATE_OBS_ENHANCED 1
#define MATE_OBS_SHARED 2
int mate_engine_set_obs_mode(mate_engine *engine, int32_t camera_team_mode, int32_t target_team_mode);

/* Normalised action grids for discrete joint actions (host arrays [n][2], copied; n = 0 / NULL removes one):
 * continuous action = action_high * grid[index], with action_high = (rotation_step, zooming_step) for a camera
 * and the target's own step size for a target (discrete_action_spaces.py:71-73, 161-163, 177-179).  Indices
 * outside [0, n) are clamped on the device; the Python boundary raises like the reference's assert. */
int mate_engine_set_action_grids(mate_engine *engine, const double *camera_grid, int32_t num_camera_actions,
                                 const double *target_grid, int32_t num_target_actions);

/* Canonical f64 export / import of the whole simulation state, [N][export_width] doubles
 * (layout documented in DESIGN.md; used by state(), the attribute views and the parity tests). */
int mate_engine_export_state(mate_engine *engine, double *dst_dev, void *stream);
int mate_engine_import_state(mate_engine *engine, const double *src_dev, void *stream);

/* Occlusion table of one camera (Camera.sight_range_func, entities.py:457-479): host buffers. */
int mate_engine_lut_read(mate_engine *engine, int64_t env, int32_t camera, double *phis_host,
                         double *rhos_host, int32_t capacity, int32_t *count);

/* The outer occlusion boundary, Camera.boundary_outer / sight_range_outer_func (entities.py:419-448, 479), read by
 * boundary_between(outer=True) (entities.py:513-543: AuxiliaryCameraRewards' soft coverage score and the renderer).
 * Off by default; once enabled every reset / rebuild_luts builds it next to the inner table (`*capacity` = knots
 * per camera to provide to lut_read_outer).  360 + 223 * obstacles rays per table: sorted in the LDS up to 16 obstacles, in HBM scratch beyond. */
int mate_engine_enable_outer_boundary(mate_engine *engine, int32_t *capacity);
int mate_engine_lut_read_outer(mate_engine *engine, int64_t env, int32_t camera, double *phis, double *rhos,
                               int32_t capacity, int32_t *count);
/* Install a recorded outer table (what sight_range_outer_func.x / .y hold, closing knot included). */
int mate_engine_lut_write_outer(mate_engine *engine, int64_t env, int32_t camera, const double *phis_host,
                                const double *rhos_host, int32_t count);
/* AuxiliaryCameraRewards' soft coverage score of the current state (wrappers/auxiliary_camera_rewards.py:181-239
 * compute_soft_coverage_scores, :128-139 per-camera reduction): `matrix_dev` [N][Nc][Nt] f64 (or NULL) receives
 * +-distance(target, nearest point of the camera's sector outline) / radius of the sector's inscribed circle, signed by
 * camera_target_view_mask; `scores_dev` [N][Nc] f64 (or NULL) the sum over the targets a camera tracks, or tanh(max)
 * when it tracks none.  `masks_dev` = the packed masks of the step / reset this follows ([N][mask_words] uint32, as
 * mate_step_io.masks_dev).  Needs the outer boundary (enable + a reset or rebuild_luts since); environments whose outer
 * table was never built yield NaN. */
int mate_engine_soft_coverage(mate_engine *engine, const uint32_t *masks_dev, double *matrix_dev, double *scores_dev, void *stream);
int mate_engine_lut_write(mate_engine *engine, int64_t env, int32_t camera, const double *phis_host,
                          const double *rhos_host, int32_t count);
/* Rebuild the occlusion tables of all environments from the current static geometry
 * (Camera.add_obstacles, entities.py:362-479) -- used after mate_engine_import_state. */
int mate_engine_rebuild_luts(mate_engine *engine, void *stream);

/* Number of (environment, step) slots spent idle waiting for a batched reset (auto_reset > 1) since creation:
 * executed env-steps = N * calls - idle. */
int mate_engine_idle_steps(mate_engine *engine, int64_t *total);

/* Average duration (ms) of the dominant kernel (step_kernel) over the launches timed since the
 * previous call, measured with HIP event pairs on the launch stream (bench.py roofline).
 * `enable` = k > 0 arms the timer for every k-th step launch from now on, 0 disarms it. */
int mate_engine_kernel_time(mate_engine *engine, int32_t enable, double *avg_ms, int64_t *launches);

/* Which compilation of the step kernel the last step()/step_random() launch ran: 0 = the generic flow (every launch
 * switch read on the device), 1 = the on-device random policy flow, 2 = the f32-continuous-actions flow.  1 and 2 are
 * the same code with the switches folded at compile time (no tapes, no discrete actions, plain observations, all four
 * outputs present, immediate auto-reset); results are bit-identical.  MATE_FLOW_GENERIC=1 forces 0.
 * After a launch with the on-device agents: 3 = rollout_greedy_kernel (the fused rollouts; a per-step call under
 * MATE_STEP_GREEDY_ROLLOUT=1), 4 = step_greedy_kernel (the one-launch form of step_greedy / step_versus_greedy). */
int mate_engine_last_flow(const mate_engine *engine);

/* Device memory for the [steps][N][...] observation blocks of the fused rollouts (mate_step_io.camera_obs_dev /
 * target_obs_dev of mate_engine_rollout_*), laid out for the way those kernels write: every environment-wave streams its
 * own 2-4 KB rows, thousands of rows at a time, and how fast HBM takes that depends on where the pages lie -- blocks from
 * hipMalloc measured 4.3-5.6 TB/s from one allocation to the next on the same GPU, physically contiguous ones
 * (hipDeviceMallocContiguous) 2.9-3.2 TB/s, the same block built from 2 MiB physical chunks mapped in a SHUFFLED order
 * 5.4-5.8 TB/s (tools/store_vmm.hip).  block_alloc builds such a block (hipMemCreate / hipMemMap, the virtual range is
 * contiguous); any device pointer works in mate_step_io -- this one is only faster to write.  `bytes` is rounded up to
 * 2 MiB.  block_free unmaps the block chunk by chunk (one hipMemUnmap per hipMemMap) and releases its memory; the virtual
 * range stays reserved (address space only; MATE_BLOCK_FREE_RANGE above).  The pointer must come from block_alloc; the CALLER
 * has waited for every launch that reads or writes the block -- the library does not synchronise here.  A failed block_alloc / block_probe / block_free reports through its return code
 * only: HIP's sticky last-error is cleared, so the caller's next launch check does not see it. */
/* The form of the row stores of the row-image rollouts (MATE-4v8-9, MATE-4v8-0, MATE-4v2-9, MATE-Navigation under mate_engine_rollout_random): 0 (default) the
 * rows' 16-byte chunks as they lie -- 1.3-2 % faster where the blocks take the rows fast, i.e. where the arithmetic bounds a launch
 * --, 1 every store instruction an aligned kilobyte -- 3 % faster where they do not (mate_engine_block_probe of the target block
 * below ~4.8 TB/s: the stores bound the launch).  Same rows either way.  Engine.reserve_rollout sets it from what it probed. */
int mate_engine_set_store_form(mate_engine *engine, int32_t shifted);
int mate_engine_block_alloc(int32_t device, int64_t bytes, void **ptr_out);
int mate_engine_block_free(void *ptr);
/* How fast THIS block takes the fused rollouts' stores: a store-only launch with their shape (one wave per environment, four
 * per workgroup, `rows_per_step` environments, `row_bytes` per environment and step -- a multiple of 16 --, a kilobyte of
 * non-temporal 16-byte stores per instruction) over the whole block, best of three, in GB/s.  Shuffled chunks
 * make a slow block unlikely, not impossible (one in four on some GPUs of the pool): a caller that has memory to spare allocates
 * a few candidates, keeps the fastest and frees the rest -- Engine.reserve_rollout does.  The block is left filled with zeros. */
int mate_engine_block_probe(int32_t device, void *block, int64_t bytes, int32_t rows_per_step, int32_t row_bytes, void *stream,
                            double *gbytes_per_s);
/* Environments per wave of the fused rollouts (mate_engine_rollout_random / _rollout_greedy / _rollout_versus_greedy) and of mate_engine_step /
 * _step_random / _step_greedy / _step_versus_greedy (which run the same kernels with one step where the choice below says so).  The engine maps ONE
 * environment onto one 64-lane wave; the small scenarios (at most four cameras and four targets: MATE-{1v1,1v2,2v2,2v4,4v2,4v4}-{0,9}, e.g. the
 * MATE-2v4-0 of the reference's target trainers, examples/ippo/target/config.py:63-66) fill a quarter of one, so their fused rollouts
 * can step FOUR environments per wave, sixteen lanes each -- same results, bit for bit.  `enable`: 0 = one per wave; 1 = the shape's
 * own number in every fused launch; 2 (the default; MATE_SUBWAVE=0 / MATE_SUBWAVE=1 in the environment make 0 / 1 the default) = where it measured faster:
 * batches of at least 32 environments per compute unit, and under the random policy every such shape but MATE-4v4-* (whose
 * one-per-wave rollout, carried by the register-resident row image, is as fast); negative = leave it as it is.  `*in_use` (may be NULL) receives
 * the number the Greedy rollouts of this engine now run with (1 for a shape without such kernels).  Takes effect from the next launch
 * on; no state changes. */
int mate_engine_set_sub_wave(mate_engine *engine, int32_t enable, int32_t *in_use);
/* The HBM rates of THIS GPU as this library's own streaming kernels see them (the yardsticks beside the vendor peak in bench.py's
 * roofline object): `mode` 0 = read `src` and write `dst` (read + write bytes counted), 1 = write `dst` only (non-temporal stores, as the row
 * writers), 2 = read `src` only, 3 = write `dst` only with plain stores (a memset);
 * 16 bytes per lane, grid-stride over `bytes` (use >= 1 GiB), non-temporal, median of five launches timed with HIP events on `stream`.
 * No reference counterpart: measurement support (SURVEY.md section 8d "confirm on the box"). */
int mate_engine_hbm_probe(int32_t device, const void *src, void *dst, int64_t bytes, int32_t mode, void *stream, double *gbytes_per_s);
/* Physical device memory taken and held without being mapped (hipMemCreate in 256 MiB pieces), and given back: what a search
 * for a fast block puts between two candidates so that the next one comes from further into the device's memory -- the blocks'
 * chunks come out of the same pool, in order; memory from hipMalloc does not move that pool's cursor. */
int mate_engine_memory_hold(int32_t device, int64_t bytes, void **token_out);
int mate_engine_memory_release(void *token);

#ifdef __cplusplus
}
#endif
#endif /* MATE_ENGINE_H */
