/* mate_oracle.h -- CPU restatement of the MultiAgentTracking step path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the CPU baseline:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (mate_amd + libmate_engine.so) never links, imports
 * or calls anything in oracle/.
 *
 * Parity status: PINNED against golden vectors recorded from the upstream
 * Python reference run in the build container (the .npz files in tests/golden, produced by
 * tests/golden/make_golden.py).  The reference ships no tests of its own.
 *
 * All file:line citations are into the upstream reference tree
 * (mate/environment.py, mate/entities.py, mate/utils.py, mate/constants.py).
 */
#ifndef MATE_ORACLE_H
#define MATE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MO_MAXC 16   /* cameras   */
#define MO_MAXT 16   /* targets   */
#define MO_MAXO 64   /* obstacles */
#define MO_NW 4      /* warehouses (constants.py:70-76) */

typedef struct mo_env mo_env;

/* ---- scalar known-answer entry points (F3 KATs) ------------------------ */
double mo_normalize_angle(double angle);                           /* utils.py:155-158 */
void mo_clamp_step(double ax, double ay, double step_size, double out[2]); /* entities.py:648-650 + utils.py:223-229 */
void mo_obstruct(double ox, double oy, double vx, double vy, double cx, double cy, double r,
                 int keep_tangential, int outer, double out[2]);   /* entities.py:158-184 */
void mo_camera_simulate(double phi, double theta, double dphi, double dtheta, double theta_min,
                        double rmax, double rot_step, double zoom_step, double out[3]); /* entities.py:347-360 */
double mo_interp(const double *xp, const double *fp, int n, double x); /* scipy interp1d(linear) == np.interp */
int mo_build_lut_raw(double cx, double cy, double rmax, const double *obstacles_xyr, int nobs,
                     double tau, int outer, double *phis, double *rhos, int cap); /* entities.py:362-479 */
int mo_camera_perceive(double cx, double cy, double phi, double theta, double sight, double px,
                       double py, double u, double tau, const double *lut_phi, const double *lut_rho,
                       int lut_n);                                  /* entities.py:491-511 */

/* ---- one environment --------------------------------------------------- */
mo_env *mo_create(int num_cameras, int num_targets, int num_obstacles);
void mo_destroy(mo_env *env);
/* Named field access; every value travels as double.  Returns the number of
 * elements copied, or -1 for an unknown field / size mismatch. */
int mo_set(mo_env *env, const char *field, const double *data, int n);
int mo_get(const mo_env *env, const char *field, double *data, int n);
int mo_camera_obs_dim(const mo_env *env);
int mo_target_obs_dim(const mo_env *env);
int mo_state_dim(const mo_env *env);

void mo_build_luts(mo_env *env);                      /* Camera.add_obstacles for every camera (a10) */
int mo_get_lut(const mo_env *env, int camera, int outer, double *phis, double *rhos, int cap);
int mo_set_lut(mo_env *env, int camera, int outer, const double *phis, const double *rhos, int n);
void mo_update_view(mo_env *env, const double *tape_ct); /* environment.py:1356-1388 (a6); tape [Nc*Nt] or NULL */
/* One env.step() (environment.py:590-676).  tape_ct [Nc*Nt] uniforms for the
 * see-through draw and goal_u [Nt] uniforms for the goal choice; NULL = use the
 * counter-based Philox streams shared with the HIP engine. */
void mo_step(mo_env *env, const double *cam_act, const double *tgt_act, const double *tape_ct,
             const double *goal_u);
void mo_observe(const mo_env *env, double *cam_obs, double *tgt_obs); /* environment.py:908-983 (a9) */
/* ... followed by the observation wrappers per team: 0 plain, 1 EnhancedObservation, 2 SharedFieldOfView */
void mo_observe_mode(const mo_env *env, int cam_mode, int tgt_mode, double *cam_obs, double *tgt_obs);
/* DiscreteCamera / DiscreteTarget action decode (wrappers/discrete_action_spaces.py:59-74, 165-180) */
void mo_decode_discrete(const mo_env *env, const int *cam_idx, const double *cam_grid, const int *tgt_idx,
                        const double *tgt_grid, double *cam_act, double *tgt_act);
void mo_soft_coverage(const mo_env *env, double *matrix, double *scores);  /* wrappers/auxiliary_camera_rewards.py:128-139,181-239 */
void mo_state(const mo_env *env, double *out);                        /* environment.py:894-906 */
void mo_reset(mo_env *env);   /* environment.py:679-834 with the engine's own Philox reset stream */
/* ... and with the uniforms recorded from the reference (tests/golden/reset_*.npz): `tape` [n] in call order,
 * `tape_ct` [Nc*Nt] see-through uniforms of the first view.  Returns the number of draws consumed, -1 if the tape
 * ran out. */
int mo_reset_tape(mo_env *env, const double *tape, int n, const double *tape_ct);
void mo_update_view_reset(mo_env *env);   /* recompute the first view of the episode (reset-view stream) */

/* ---- batch (cpu_baseline and GPU-vs-CPU rollouts) ---------------------- */
typedef struct mo_batch mo_batch;
mo_batch *mo_batch_create(const mo_env *prototype, int n, uint64_t seed, uint64_t first_env_index);
void mo_batch_destroy(mo_batch *b);
mo_env *mo_batch_env(mo_batch *b, int i);
int mo_batch_get(const mo_batch *b, const char *field, double *out, int n);   /* mo_get of every environment, [envs][n] */
void mo_batch_reset(mo_batch *b, int threads);
/* Steps every env once with the on-the-fly uniform random policy,
 * auto-resetting finished episodes.  Actions may be NULL (= Philox policy). */
void mo_batch_step(mo_batch *b, const float *cam_act, const float *tgt_act, int auto_reset, int threads);
void mo_batch_observe(mo_batch *b, float *cam_obs, float *tgt_obs, int threads);
void mo_random_actions(uint64_t seed, uint64_t env_index, uint64_t tick, int Nc, int Nt,
                       double rot_step, double zoom_step, double step_size, float *cam_act, float *tgt_act);

/* counter-based RNG shared (by specification) with the HIP engine */
/* Rule-based agents of the reference (mate/agents/greedy.py) for one environment; all random draws on a tape:
 * cam_binom_u[Nc], cam_sample_u[Nc][2], cam_delay[Nc][Nc] (value drawn for sender -> recipient),
 * tgt_choice_u[Nt], tgt_binom_u[Nt], tgt_sample_u[Nt][2], tgt_reset_sample_u[Nt][2] (first call of an episode). */
typedef struct mo_policy mo_policy;
typedef struct mo_policy_tape {
    const double *cam_binom_u, *cam_sample_u;
    const int *cam_delay;
    const double *tgt_choice_u, *tgt_binom_u, *tgt_sample_u, *tgt_reset_sample_u;
} mo_policy_tape;
mo_policy *mo_policy_create(void);
void mo_policy_destroy(mo_policy *policy);
void mo_policy_act(mo_policy *policy, const mo_env *env, const mo_policy_tape *tape, double *cam_act, double *tgt_act);

void mo_philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
