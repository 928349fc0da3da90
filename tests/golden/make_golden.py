#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container).

Imports the upstream reference read-only from /root/reference under the
build-owned `gymshim` package, drives `MultiAgentTracking.reset()/step()` with
recorded actions and *recorded random draws*, and dumps small `.npz` fixtures
into this directory.  Nothing from the reference is copied: the fixtures are
data (inputs and expected outputs).

    python tests/golden/make_golden.py            # regenerate everything

Fixture families:
  trace_<cfg>_<policy>_s<seed>.npz   F1 reset snapshot + F4 LUT knots + F2 step trace
  kat_obstruct.npz                   F3 ray/circle known answers (entities.py:158-184)
  kat_scalar.npz                     F3 normalize_angle / polar clamp (utils.py:155,223-229)
  kat_perceive.npz                   F3 Camera.perceive on crafted geometry (entities.py:491-511)
  reset_<cfg>_s<seed>.npz            F1 reset() with EVERY random draw on a tape (environment.py:679-834)
  chain_<cfg>_s<seed>.npz            the example trainers' whole wrapper chain for a camera learner (examples/ippo/camera/config.py:19-51)

Random draws are captured by replacing each RandomState with a recording proxy
(the reference source is not modified): every in-sector `binomial(1, tau)` draw
becomes a logged uniform U (result = U > 1 - tau for tau <= 0.5, identical to
numpy's legacy inversion sampler, asserted below), and every goal `choice`
becomes a logged (k candidates, picked j).
"""

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get('MATE_REFERENCE', '/root/reference')
sys.path.insert(0, os.path.join(HERE, 'gymshim'))
sys.path.insert(0, REFERENCE)

import gym  # noqa: E402,F401  (the shim)
import mate  # noqa: E402  (the reference, read-only)
from mate.agents import GreedyCameraAgent, GreedyTargetAgent  # noqa: E402
from mate.entities import Camera, Obstacle, Target  # noqa: E402
from mate.utils import Vector2D, normalize_angle  # noqa: E402


class RecordingRNG:
    """Pass-through proxy for a numpy RandomState that logs the draws the step
    path makes (`binomial` in Camera.perceive, `choice` in _assign_goals)."""

    def __init__(self, real, log, owner):
        self._real = real
        self._log = log
        self._owner = owner

    def binomial(self, n, p, size=None):
        assert n == 1 and size is None
        frame = sys._getframe(1)
        other = frame.f_locals.get('other', None)
        u = float(self._real.random_sample())
        out = int(u > 1.0 - p) if p <= 0.5 else int(u <= p)
        self._log.append(('binomial', self._owner, other, float(p), u, out))
        return out

    def choice(self, a, size=None, replace=True, p=None):
        if size is not None or p is not None or np.isscalar(a):
            return self._real.choice(a, size=size, replace=replace, p=p)
        a = np.asarray(a)
        frame = sys._getframe(1)
        t = frame.f_locals.get('t', None)
        j = int(self._real.randint(0, len(a)))
        self._log.append(('choice', self._owner, t, len(a), j, int(a[j])))
        return a[j]

    def __getattr__(self, name):
        return getattr(self._real, name)


def check_binomial_model():
    """numpy legacy binomial(1,p) == (U > 1-p) on the same stream, p <= 0.5."""
    a = np.random.RandomState(1234)
    b = np.random.RandomState(1234)
    for p in (0.1, 0.25, 0.5):
        for _ in range(2000):
            assert a.binomial(1, p) == int(b.random_sample() > 1.0 - p)


def install_proxies(env, log):
    for c, camera in enumerate(env.cameras):
        box = camera.location_random_range
        if not isinstance(box._np_random, RecordingRNG):
            box._np_random = RecordingRNG(box.np_random, log, camera)
    if not isinstance(env._np_random, RecordingRNG):
        env._np_random = RecordingRNG(env.np_random, log, 'env')


def pad_knots(funcs, width):
    n = len(funcs)
    phis = np.full((n, width), np.nan)
    rhos = np.full((n, width), np.nan)
    counts = np.zeros(n, dtype=np.int64)
    for i, f in enumerate(funcs):
        k = len(f.x)
        counts[i] = k
        phis[i, :k] = f.x
        rhos[i, :k] = f.y
    return phis, rhos, counts


def snapshot_static(env):
    out = {}
    Nc, Nt, No = env.num_cameras, env.num_targets, env.num_obstacles
    out['cam_xy'] = np.array([c.location for c in env.cameras]).reshape(Nc, 2)
    out['cam_radius'] = np.array([c.radius for c in env.cameras], dtype=np.float64)
    out['cam_min_viewing_angle'] = np.array([c.min_viewing_angle for c in env.cameras], dtype=np.float64)
    out['cam_max_sight_range'] = np.array([c.max_sight_range for c in env.cameras], dtype=np.float64)
    out['cam_rotation_step'] = np.array([c.rotation_step for c in env.cameras], dtype=np.float64)
    out['cam_zooming_step'] = np.array([c.zooming_step for c in env.cameras], dtype=np.float64)
    out['obs_xyr'] = np.array([np.append(o.location, o.radius) for o in env.obstacles]).reshape(No, 3)
    out['tgt_capacity'] = np.array(env.target_capacities, dtype=np.int64)
    out['tgt_step_size'] = np.array([t.step_size for t in env.targets], dtype=np.float64)
    out['tgt_sight_range'] = np.array([t.sight_range for t in env.targets], dtype=np.float64)
    out['camera_obstacle_view_mask'] = env.camera_obstacle_view_mask.copy()
    if Nc > 0:
        width = max(max(len(c.sight_range_func.x), len(c.sight_range_outer_func.x)) for c in env.cameras)
        p, r, k = pad_knots([c.sight_range_func for c in env.cameras], width)
        out['lut_phis'], out['lut_rhos'], out['lut_count'] = p, r, k
        p, r, k = pad_knots([c.sight_range_outer_func for c in env.cameras], width)
        out['lut_outer_phis'], out['lut_outer_rhos'], out['lut_outer_count'] = p, r, k
    return out


def snapshot_dynamic(env):
    Nc, Nt = env.num_cameras, env.num_targets
    d = {}
    d['cam_phi'] = np.array([c.orientation for c in env.cameras], dtype=np.float64)
    d['cam_theta'] = np.array([c.viewing_angle for c in env.cameras], dtype=np.float64)
    d['cam_sight'] = np.array([c.sight_range for c in env.cameras], dtype=np.float64)
    d['tgt_xy'] = np.array([t.location for t in env.targets]).reshape(Nt, 2)
    d['tgt_colliding'] = np.array([bool(t.is_colliding) for t in env.targets])
    d['tgt_empty_bits'] = np.array([t.empty_bits for t in env.targets]).astype(bool).reshape(Nt, 4)
    d['tgt_goal_bits'] = env.target_goal_bits.astype(np.int64).copy()
    d['tgt_goals'] = env.target_goals.astype(np.int64).copy()
    d['freights'] = env.freights.astype(np.int64).copy()
    d['bounties'] = env.bounties.astype(np.int64).copy()
    d['target_steps'] = env.target_steps.astype(np.int64).copy()
    d['tracked_steps'] = env.tracked_steps.astype(np.int64).copy()
    d['remaining_cargoes'] = env.remaining_cargoes.astype(np.int64).copy()
    d['awaiting_cargo_counts'] = env.awaiting_cargo_counts.astype(np.int64).copy()
    d['num_delivered_cargoes'] = np.int64(env.num_delivered_cargoes)
    d['episode_reward'] = np.float64(env.target_team_episode_reward)
    d['delayed_episode_reward'] = np.float64(env.delayed_target_team_episode_reward)
    d['episode_step'] = np.int64(env.episode_step)
    d['camera_target_view_mask'] = env.camera_target_view_mask.copy()
    d['target_camera_view_mask'] = env.target_camera_view_mask.copy()
    d['target_obstacle_view_mask'] = env.target_obstacle_view_mask.copy()
    d['target_target_view_mask'] = env.target_target_view_mask.copy()
    d['camera_camera_view_mask'] = env.camera_camera_view_mask.copy()
    d['tracked_bits'] = np.asarray(env.tracked_bits).astype(bool).copy()
    d['target_dones'] = np.asarray(env.target_dones).astype(bool).copy()
    d['target_warehouse_distances'] = env.target_warehouse_distances.copy()
    d['coverage_rate'] = np.float64(env.coverage_rate)
    d['real_coverage_rate'] = np.float64(env.real_coverage_rate)
    d['mean_transport_rate'] = np.float64(env.mean_transport_rate)
    d['state'] = env.state()
    return d


def drain_log(env, log):
    """Turn the draw log of one step into dense tapes."""
    Nc, Nt = env.num_cameras, env.num_targets
    tape_ct = np.full((Nc, Nt), np.nan)
    tape_cc = np.full((Nc, Nc), np.nan)
    goal_u = np.full(Nt, np.nan)
    goal_k = np.zeros(Nt, dtype=np.int64)
    goal_j = np.full(Nt, -1, dtype=np.int64)
    for item in log:
        if item[0] == 'binomial':
            _, cam, other, p, u, out = item
            c = env.cameras.index(cam)
            if isinstance(other, Camera):
                tape_cc[c, env.cameras.index(other)] = u
            else:
                t = env.targets.index(other)
                assert np.isnan(tape_ct[c, t])
                tape_ct[c, t] = u
        else:
            _, _, t, k, j, value = item
            assert t is not None and goal_j[t] < 0
            goal_k[t], goal_j[t] = k, j
            goal_u[t] = (j + 0.5) / k
    log.clear()
    return tape_ct, tape_cc, goal_u, goal_k, goal_j


def random_actions(env, rng, step):
    Nc, Nt = env.num_cameras, env.num_targets
    cam = rng.uniform(-1.5, 1.5, size=(Nc, 2)) * np.array([env.camera_rotation_step, env.camera_zooming_step]) if Nc else np.zeros((0, 2))
    tgt = rng.uniform(-1.5, 1.5, size=(Nt, 2)) * env.target_step_size
    if step % 7 == 3:  # exact zero actions exercise the zero-length ray branch
        tgt[step % Nt] = 0.0
        if Nc:
            cam[step % Nc] = 0.0
    if step % 11 == 5:  # axis-aligned full-speed moves
        tgt[(step // 11) % Nt] = [env.target_step_size, 0.0]
    return cam, tgt


class AgentRNG:
    """Recording proxy for a rule-based agent's RandomState (agents/greedy.py draws)."""

    def __init__(self, real, log, who):
        self._real, self._log, self._who = real, log, who

    def binomial(self, n, p, size=None):
        assert n == 1 and size is None
        u = float(self._real.random_sample())
        out = int(u > 1.0 - p) if p <= 0.5 else int(u <= p)
        self._log.append(('binom', self._who, float(p), u))
        return out

    def randint(self, low, high=None, size=None, dtype=int):
        if high is None or size is not None or high - low > 1000:
            return self._real.randint(low, high, size=size)
        value = int(self._real.randint(low, high))
        recipient = sys._getframe(1).f_locals.get('c', None)
        self._log.append(('randint', self._who, int(low), int(high), value, recipient))
        return value

    def choice(self, a, size=None, replace=True, p=None):
        assert size is None and p is None
        a = list(a)
        j = int(self._real.randint(0, len(a)))
        self._log.append(('choice', self._who, len(a), j))
        return a[j]

    def __getattr__(self, name):
        return getattr(self._real, name)


AGENT_LOG = []
_ORIG_BOX_SAMPLE = gym.spaces.Box.sample


def _recording_box_sample(self):
    """Box.sample of the (build-owned) gym stand-in with the uniforms logged: low + (high - low) * U."""
    if not np.all(np.isfinite(self.low)) or not np.all(np.isfinite(self.high)):
        return _ORIG_BOX_SAMPLE(self)
    u = self.np_random.random_sample(self.shape)
    AGENT_LOG.append(('sample', id(self), np.array(u, dtype=np.float64)))
    return (self.low + (self.high - self.low) * u).astype(self.dtype)


def drain_agent_log(cam_agents, tgt_agents):
    Nc, Nt = len(cam_agents), len(tgt_agents)
    spaces_of = {id(a.action_space): ('cam', a.index) for a in cam_agents}
    spaces_of.update({id(a.action_space): ('tgt', a.index) for a in tgt_agents})
    out = {
        'cam_binom_u': np.full(Nc, np.nan), 'cam_sample_u': np.full((Nc, 2), np.nan), 'cam_delay': np.full((Nc, Nc), -1, dtype=np.int64),
        'tgt_choice_u': np.full(Nt, np.nan), 'tgt_binom_u': np.full(Nt, np.nan), 'tgt_sample_u': np.full((Nt, 2), np.nan),
    }
    for item in AGENT_LOG:
        if item[0] == 'sample':
            if item[1] not in spaces_of:
                continue
            team, idx = spaces_of[item[1]]
            out[team + '_sample_u'][idx] = item[2]
        else:
            team, idx = item[1]
            if item[0] == 'binom':
                out[team + '_binom_u'][idx] = item[3]
            elif item[0] == 'choice':
                assert team == 'tgt'
                out['tgt_choice_u'][idx] = (item[3] + 0.5) / item[2]
            elif item[0] == 'randint':
                assert team == 'cam' and item[5] is not None
                out['cam_delay'][idx, item[5]] = item[4]
    AGENT_LOG.clear()
    return out


def make_trace(name, config, seed, policy, steps, overrides=None, tweak=None, f64_obs_steps=None, record_agents=False,
               extra_factory=None, discrete_levels=None, aux_rewards=None, aux_target_rewards=None, env=None, first_obs=None):
    """`env` / `first_obs`: continue on an environment object that already exists, from the observations its last reset() returned (the
    second episode of two_episode_fixture); otherwise the environment is made, seeded and reset here.  Returns the environment."""
    if env is None:
        env = mate.make('MultiAgentTracking-v0', config=config, **(overrides or {}))
        env.seed(seed)
        cam_obs, tgt_obs = env.reset()
        if tweak is not None:
            tweak(env)
            cam_obs, tgt_obs = env.joint_observation()
    else:
        cam_obs, tgt_obs = first_obs
    log = []
    install_proxies(env, log)

    Nc, Nt, No = env.num_cameras, env.num_targets, env.num_obstacles
    out = {
        'config_file': np.str_(config),
        'seed': np.int64(seed),
        'policy': np.str_(policy),
        'num_cameras': np.int64(Nc),
        'num_targets': np.int64(Nt),
        'num_obstacles': np.int64(No),
        'transmittance': np.float64(env.obstacle_transmittance),
        'max_episode_steps': np.int64(env.max_episode_steps),
        'sparse_reward': np.bool_(env._sparse_reward),
        'freight_scale': np.float64(env.freight_scale),
        'bounty_scale': np.float64(env.bounty_scale),
        'reward_scale': np.float64(env.reward_scale),
        'max_target_team_episode_reward': np.float64(env.max_target_team_episode_reward),
        'target_step_size': np.float64(env.target_step_size),
    }
    for k, v in snapshot_static(env).items():
        out['static/' + k] = v
    for k, v in snapshot_dynamic(env).items():
        out['reset/' + k] = v
    out['reset/cam_obs'] = cam_obs
    out['reset/tgt_obs'] = tgt_obs
    extra = extra_factory(env) if extra_factory is not None else None
    if extra is not None:
        for k, v in extra(cam_obs, tgt_obs).items():
            out['reset/' + k] = v
    if discrete_levels is not None:     # DiscreteCamera / DiscreteTarget of the reference decode the indices
        disc_cam = mate.DiscreteCamera(env, levels=discrete_levels[0]) if Nc_of(env) else None
        disc_tgt = mate.DiscreteTarget(env, levels=discrete_levels[1])
        for t, target in enumerate(env.targets):     # what DiscreteTarget.reset() does (discrete_action_spaces.py:156-163)
            disc_tgt.action_high[t] = target.step_size
        if disc_cam is not None:
            out['camera_action_grid'] = disc_cam.normalized_action_grid
        out['target_action_grid'] = disc_tgt.normalized_action_grid
        out['discrete_levels'] = np.asarray(discrete_levels, dtype=np.int64)

    aux = None
    if aux_rewards is not None:         # the reference's AuxiliaryCameraRewards shapes the camera rewards of this trace
        aux = mate.AuxiliaryCameraRewards(mate.RepeatedRewardIndividualDone(env), coefficients=aux_rewards[0], reduction=aux_rewards[1])
        out['aux_keys'] = np.asarray(list(aux_rewards[0].keys()))
        out['aux_coefficients'] = np.asarray(list(aux_rewards[0].values()), dtype=np.float64)
        out['aux_reduction'] = np.str_(aux_rewards[1])
    auxt = None
    if aux_target_rewards is not None:  # ... and AuxiliaryTargetRewards the target rewards (wrappers/auxiliary_target_rewards.py)
        auxt = mate.AuxiliaryTargetRewards(mate.RepeatedRewardIndividualDone(env), coefficients=aux_target_rewards[0],
                                           reduction=aux_target_rewards[1])
        out['auxt_keys'] = np.asarray(list(aux_target_rewards[0].keys()))
        out['auxt_coefficients'] = np.asarray(list(aux_target_rewards[0].values()), dtype=np.float64)
        out['auxt_reduction'] = np.str_(aux_target_rewards[1])
        assert aux is None

    rng = np.random.RandomState(seed + 1000)
    if policy == 'greedy':
        cam_agents = GreedyCameraAgent(seed=seed + 1).spawn(Nc) if Nc else []
        tgt_agents = GreedyTargetAgent(seed=seed + 2).spawn(Nt)
        if record_agents:
            gym.spaces.Box.sample = _recording_box_sample
            for i, agent in enumerate(cam_agents):
                agent._np_random = AgentRNG(agent.np_random, AGENT_LOG, None)
            for i, agent in enumerate(tgt_agents):
                agent._np_random = AgentRNG(agent.np_random, AGENT_LOG, None)
        AGENT_LOG.clear()
        mate.group_reset(cam_agents, cam_obs)
        mate.group_reset(tgt_agents, tgt_obs)
        if record_agents:
            for agent in cam_agents:
                agent._np_random._who = ('cam', agent.index)
            for agent in tgt_agents:
                agent._np_random._who = ('tgt', agent.index)
            reset_draws = drain_agent_log(cam_agents, tgt_agents)
            out['agent/tgt_reset_sample_u'] = reset_draws['tgt_sample_u']
        cam_infos = tgt_infos = None

    per_step = {}

    def push(key, value):
        per_step.setdefault(key, []).append(np.asarray(value))

    n_done = 0
    for step in range(steps):
        if policy == 'greedy':
            cam_act = np.asarray(mate.group_step(env, cam_agents, cam_obs, cam_infos), dtype=np.float64).reshape(Nc, 2)
            tgt_act = np.asarray(mate.group_step(env, tgt_agents, tgt_obs, tgt_infos), dtype=np.float64).reshape(Nt, 2)
            if record_agents:
                for k, v in drain_agent_log(cam_agents, tgt_agents).items():
                    push('agent_' + k, v)
        elif policy == 'discrete':
            cam_idx = rng.randint(0, discrete_levels[0] ** 2, size=Nc)
            tgt_idx = rng.randint(0, discrete_levels[1] ** 2, size=Nt)
            cam_act = np.zeros((0, 2))
            if disc_cam is not None:
                cam_act = np.asarray(disc_cam.action((cam_idx, None))[0], dtype=np.float64).reshape(Nc, 2)
            tgt_act = np.asarray(disc_tgt.action((None, tgt_idx))[1], dtype=np.float64).reshape(Nt, 2)
            push('cam_idx', cam_idx)
            push('tgt_idx', tgt_idx)
        else:
            cam_act, tgt_act = random_actions(env, rng, step)
        log.clear()
        if auxt is not None:
            (cam_obs, tgt_obs), (r_cams, shaped), (_, tgt_dones), (cam_infos, tgt_infos) = auxt.step((cam_act, tgt_act))
            r_tgt, done = tgt_infos[0]['raw_reward'], tgt_dones[0]
            r_cam = r_cams[0] if len(r_cams) else -r_tgt
            push('aux_reward_tgt', np.asarray(shaped, dtype=np.float64))
            for key in mate.AuxiliaryTargetRewards.ACCEPTABLE_KEYS:       # every term, whatever its coefficient
                if key in aux_target_rewards[0]:
                    push('auxt_' + key, np.asarray([info['auxiliary_reward_' + key] for info in tgt_infos], dtype=np.float64))
        elif aux is None:
            (cam_obs, tgt_obs), (r_cam, r_tgt), done, (cam_infos, tgt_infos) = env.step((cam_act, tgt_act))
        else:
            (cam_obs, tgt_obs), (shaped, r_tgts), (cam_dones, _), (cam_infos, tgt_infos) = aux.step((cam_act, tgt_act))
            r_cam, r_tgt, done = cam_infos[0]['raw_reward'], r_tgts[0], cam_dones[0]
            push('aux_reward_cam', np.asarray(shaped, dtype=np.float64))
            push('soft_coverage_matrix', np.asarray(aux.soft_coverage_score_matrix, dtype=np.float64))
            push('soft_coverage_score', np.asarray([info['auxiliary_reward_soft_coverage_score'] for info in cam_infos], dtype=np.float64))
        tape_ct, tape_cc, goal_u, goal_k, goal_j = drain_log(env, log)

        push('cam_act', cam_act)
        push('tgt_act', tgt_act)
        push('tape_ct', tape_ct)
        push('goal_u', goal_u)
        push('goal_k', goal_k)
        push('goal_j', goal_j)
        push('cam_obs', cam_obs)
        push('tgt_obs', tgt_obs)
        push('reward_cam', r_cam)
        push('reward_tgt', r_tgt)
        push('normalized_reward_tgt', tgt_infos[0]['normalized_raw_reward'])
        push('done', done)
        if extra is not None:
            for k, v in extra(cam_obs, tgt_obs).items():
                push(k, v)
        for k, v in snapshot_dynamic(env).items():
            push(k, v)
        if done:
            n_done += 1
            if n_done >= 2:  # keep one extra step after done, then stop
                break

    gym.spaces.Box.sample = _ORIG_BOX_SAMPLE
    for k, v in per_step.items():
        arr = np.stack(v)
        out['step/' + k] = arr
    if record_agents or auxt is not None:   # observations are not needed to check a policy / a reward: keep these fixtures small
        for k in ('step/cam_obs', 'step/tgt_obs', 'step/state'):
            out.pop(k, None)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    nsteps = len(per_step['done'])
    ndel = int(out['step/num_delivered_cargoes'][-1])
    ncol = int(out['step/tgt_colliding'].sum())
    nsee = int(np.isfinite(out['step/tape_ct']).sum()) if Nc else 0
    print(f'{name}: {nsteps} steps, delivered={ndel}, collisions={ncol}, in-sector draws={nsee}, '
          f'done={bool(out["step/done"][-1])}, {os.path.getsize(path) / 1024:.0f} KiB')
    return env


def Nc_of(env):
    return env.num_cameras


def observation_mode_extras(teams):
    """Outputs of the reference's EnhancedObservation / SharedFieldOfView wrappers on the observations of a
    trace (wrappers/enhanced_observation.py:72-126, wrappers/shared_field_of_view.py:72-148), f32."""
    def factory(env):
        wrappers = {}
        for team in teams:
            wrappers['enhanced_' + team] = mate.EnhancedObservation(env, team=team)
            wrappers['shared_' + team] = mate.SharedFieldOfView(env, team=team)

        def extra(cam_obs, tgt_obs):
            res = {}
            for key, w in wrappers.items():
                c, t = w.observation((np.array(cam_obs, dtype=np.float64), np.array(tgt_obs, dtype=np.float64)))
                res['cam_obs_' + key] = np.asarray(c, dtype=np.float32)
                res['tgt_obs_' + key] = np.asarray(t, dtype=np.float32)
            return res
        return extra
    return factory


def tweak_few_cargoes(env):
    """Leave only the cargoes in transit so the episode ends within the trace
    (state injection on the reference object; reference code untouched)."""
    env.remaining_cargoes.fill(0)
    env.awaiting_cargo_counts[:] = 0
    for t in range(env.num_targets):
        g = env.target_goals[t]
        if g >= 0:
            env.awaiting_cargo_counts[g] += env.target_goal_bits[t, g]
    env._state = None


# --------------------------------------------------------------------------- KATs
def kat_obstruct():
    rng = np.random.RandomState(7)
    rows = []

    def run(origin, vec, center, radius, keep_tangential, outer):
        ray = Vector2D(vector=np.array(vec, dtype=np.float64), origin=np.array(origin, dtype=np.float64))
        obstacle = Obstacle(location=np.array(center, dtype=np.float64), radius=float(radius))
        obstacle.location = np.array(center, dtype=np.float64)  # undo the terrain clip of Entity.reset
        res = obstacle.obstruct(ray, keep_tangential=keep_tangential, outer=outer)
        v = np.array(res.vector, dtype=np.float64)
        rows.append(list(origin) + list(vec) + list(center) + [radius, float(keep_tangential), float(outer)] + list(v))

    for _ in range(1500):
        origin = rng.uniform(-300, 300, 2)
        center = origin + rng.uniform(-150, 150, 2)
        radius = rng.uniform(5, 100)
        vec = rng.uniform(-60, 60, 2)
        for kt in (False, True):
            for outer in (False, True):
                run(origin, vec, center, radius, kt, outer)
    # crafted edge cases
    for kt in (False, True):
        for outer in (False, True):
            run([0, 0], [0, 0], [10, 0], 5, kt, outer)            # zero-length ray
            run([0, 0], [10, 0], [3, 0], 5, kt, outer)            # origin inside the circle
            run([0, 0], [10, 0], [20, 0], 5, kt, outer)           # far: norm + r <= dist
            run([0, 0], [15, 0], [20, 0], 5, kt, outer)           # exact touch: dist == norm + r
            run([0, 0], [30, 0], [20, 5], 5, kt, outer)           # tangent: perpendicular == r
            run([0, 0], [30, 0], [20, 0], 5, kt, outer)           # head-on
            run([0, 0], [-30, 0], [20, 0], 5, kt, outer)          # pointing away (inner < 0)
            run([0, 0], [30, 0], [20, 4.999], 5, kt, outer)       # grazing
            run([0, 0], [0, 25], [0, 20], 5, kt, outer)           # vertical
            run([5, 5], [20, 20], [25, 25], 10, kt, outer)        # diagonal through centre
            run([0, 0], [100, 0], [20, 0], 5, kt, outer)          # passes fully through
    np.savez_compressed(os.path.join(HERE, 'kat_obstruct.npz'), rows=np.array(rows, dtype=np.float64),
                        columns=np.str_('ox oy vx vy cx cy r keep_tangential outer outx outy'))
    print(f'kat_obstruct: {len(rows)} rows')


def kat_scalar():
    rng = np.random.RandomState(11)
    angles = np.concatenate([
        np.array([-540.0, -360.0, -180.0, -179.99999999999997, -0.0, 0.0, 1e-300, -1e-300, 179.99999999999997,
                  180.0, 180.00000000000003, 360.0, 539.9999, 540.0, 720.0, 1e6 + 0.5, -1e6 - 0.5]),
        rng.uniform(-1000, 1000, 500),
    ])
    normalized = np.array([normalize_angle(float(a)) for a in angles])

    # target step clamp: Vector2D(vector=a); if norm > v: norm = v   (entities.py:648-650)
    acts = np.concatenate([rng.uniform(-40, 40, (600, 2)),
                           np.array([[0, 0], [20, 0], [0, 20], [-20, 0], [0, -20], [20, 20], [1e-12, 0], [14.142135623730951, 14.142135623730951]])])
    vs = np.concatenate([np.full(304, 20.0), np.full(304, 10.0)])
    clamped = []
    for a, v in zip(acts, vs):
        step = Vector2D(vector=np.array(a, dtype=np.float64), origin=np.zeros(2))
        if step.norm > v:
            step.norm = v
        clamped.append(np.array(step.vector, dtype=np.float64))

    # Camera.simulate clamps (entities.py:347-360)
    cam_rows = []
    for _ in range(600):
        cam = Camera(location=np.array([0.0, 0.0]), min_viewing_angle=30.0, max_sight_range=1500.0,
                     rotation_step=5.0, zooming_step=2.5, radius=40.0)
        phi0 = float(rng.choice([rng.uniform(-180, 180), -180.0, 177.5, -177.5, 179.0]))
        th0 = float(rng.choice([rng.uniform(30, 180), 30.0, 180.0, 31.0, 179.0]))
        cam.orientation = phi0
        cam.viewing_angle = th0
        act = rng.uniform(-8, 8, 2)
        phi0n = cam.orientation
        cam.simulate(act)
        cam_rows.append([phi0n, th0, act[0], act[1], cam.orientation, cam.viewing_angle, cam.sight_range])
    np.savez_compressed(os.path.join(HERE, 'kat_scalar.npz'), angles=angles, normalized=normalized,
                        clamp_action=acts, clamp_step=vs, clamp_out=np.array(clamped),
                        cam_sim=np.array(cam_rows), cam_sim_columns=np.str_('phi0 theta0 dphi dtheta phi1 theta1 sight1'))
    print(f'kat_scalar: {len(angles)} angles, {len(acts)} clamps, {len(cam_rows)} camera steps')


class TapeRNG:
    """Feeds a prescribed uniform to Camera.perceive's binomial draw."""

    def __init__(self):
        self.u = 0.0

    def binomial(self, n, p):
        return int(self.u > 1.0 - p) if p <= 0.5 else int(self.u <= p)


def kat_perceive():
    rng = np.random.RandomState(21)
    cases = []
    luts = []
    for case in range(12):
        cam = Camera(location=np.array([0.0, 0.0]), min_viewing_angle=30.0, max_sight_range=1500.0 if case % 2 == 0 else 700.0,
                     rotation_step=5.0, zooming_step=2.5, radius=40.0)
        cam.location = rng.uniform(-600, 600, 2)
        cam.reset.__func__  # noqa: B018  (documenting that reset() is what builds the 360-ray boundary)
        loc = cam.location.copy()
        cam.location_random_range = gym.spaces.Box(low=loc, high=loc, dtype=np.float64)
        cam.reset()
        nobs = [0, 1, 3, 6, 9, 9, 2, 4, 9, 5, 7, 1][case]
        obstacles = []
        for _ in range(nobs):
            while True:
                c = cam.location + rng.uniform(-900, 900, 2)
                r = rng.uniform(25, 100)
                if np.linalg.norm(c - cam.location) > r + 60:
                    break
            ob = Obstacle(location=c, radius=float(r))
            ob.location = np.asarray(c, dtype=np.float64)
            ob.radius = float(r)
            obstacles.append(ob)
        cam.clear_obstacles()
        cam.add_obstacles(*obstacles)
        tape = TapeRNG()
        cam.location_random_range._np_random = tape
        obs_arr = np.array([np.append(o.location, o.radius) for o in obstacles]).reshape(nobs, 3)
        luts.append((cam.sight_range_func.x.copy(), cam.sight_range_func.y.copy(), obs_arr, cam.location.copy(), cam.max_sight_range))
        for _ in range(400):
            cam.orientation = float(rng.choice([rng.uniform(-180, 180), -180.0, 179.5, 0.0]))
            cam.viewing_angle = float(rng.uniform(30, 180))
            cam.sight_range = np.sqrt(cam.area_product / cam.viewing_angle)
            mode = rng.randint(0, 5)
            if mode == 0 and nobs > 0:      # just behind / beside an obstacle
                ob = obstacles[rng.randint(nobs)]
                d = ob.location - cam.location
                p = cam.location + d * rng.uniform(0.8, 1.6) + rng.uniform(-1, 1, 2) * ob.radius * 1.3
            elif mode == 1:                 # exactly on a knot angle
                k = rng.randint(len(cam.sight_range_func.x) - 1)
                ang = np.deg2rad(cam.sight_range_func.x[k])
                p = cam.location + rng.uniform(10, cam.sight_range * 1.1) * np.array([np.cos(ang), np.sin(ang)])
            elif mode == 2:                 # near the sector edge
                edge = cam.orientation + rng.choice([-0.5, 0.5]) * cam.viewing_angle + rng.uniform(-0.2, 0.2)
                ang = np.deg2rad(edge)
                p = cam.location + rng.uniform(10, cam.sight_range) * np.array([np.cos(ang), np.sin(ang)])
            elif mode == 3:                 # near the range limit, centre of the sector
                ang = np.deg2rad(cam.orientation + rng.uniform(-0.4, 0.4) * cam.viewing_angle)
                p = cam.location + cam.sight_range * rng.uniform(0.98, 1.02) * np.array([np.cos(ang), np.sin(ang)])
            else:
                p = cam.location + rng.uniform(-1, 1, 2) * cam.sight_range
            tgt = Target(location=np.array([0.0, 0.0]))
            tgt.location = np.asarray(p, dtype=np.float64)
            tape.u = float(rng.uniform(0, 1))
            tau = float(rng.choice([0.0, 0.1, 0.1]))
            seen = bool(cam.perceive(tgt, transmittance=tau))
            cases.append([case, cam.orientation, cam.viewing_angle, cam.sight_range, p[0], p[1], tape.u, tau, float(seen)])
    width = max(len(l[0]) for l in luts)
    phis = np.full((len(luts), width), np.nan)
    rhos = np.full((len(luts), width), np.nan)
    counts = np.zeros(len(luts), dtype=np.int64)
    obs = np.full((len(luts), 9, 3), np.nan)
    nobs = np.zeros(len(luts), dtype=np.int64)
    cam_xy = np.zeros((len(luts), 2))
    rmax = np.zeros(len(luts))
    for i, (x, y, o, loc, rm) in enumerate(luts):
        counts[i] = len(x)
        phis[i, :len(x)] = x
        rhos[i, :len(x)] = y
        nobs[i] = len(o)
        obs[i, :len(o)] = o
        cam_xy[i] = loc
        rmax[i] = rm
    np.savez_compressed(os.path.join(HERE, 'kat_perceive.npz'), cases=np.array(cases),
                        columns=np.str_('case phi theta sight px py u tau seen'),
                        lut_phis=phis, lut_rhos=rhos, lut_count=counts, obstacles=obs, num_obstacles=nobs,
                        cam_xy=cam_xy, cam_max_sight_range=rmax)
    print(f'kat_perceive: {len(cases)} cases, seen={int(np.array(cases)[:, -1].sum())}')


# --------------------------------------------------------------------------- reset() on a tape
RESET_TAPE = []


class TapeDrivenRNG:
    """Stands in for a numpy RandomState while the reference's reset() runs: every draw is ONE uniform taken from the
    wrapped generator, appended to RESET_TAPE in call order, and turned into the requested quantity by a fixed rule
    (the rules are restated in oracle/mate_oracle.c above reset_impl and in reset_kernels.hpp).  The reference's
    source is untouched; it simply consumes this object through the RandomState API.  The see-through Bernoulli draws
    of the first _update_view go to a per-pair log instead (they are keyed by (camera, target), not by order)."""

    def __init__(self, real, pair_log, owner):
        self._real, self._pair_log, self._owner = real, pair_log, owner

    def _u(self):
        u = float(self._real.random_sample())
        RESET_TAPE.append(u)
        return u

    def shuffle(self, x):                              # Fisher-Yates
        for i in range(len(x) - 1, 0, -1):
            j = int(self._u() * (i + 1))
            x[i], x[j] = x[j], x[i]

    def permutation(self, n):
        arr = np.arange(int(n))
        self.shuffle(arr)
        return arr

    def choice(self, a, size=None, replace=True, p=None):
        assert p is None
        if np.isscalar(a):                             # choice(n, size=k, replace=False): partial Fisher-Yates
            assert size is not None and not replace
            idx = list(range(int(a)))
            for i in range(int(size)):
                j = i + int(self._u() * (len(idx) - i))
                idx[i], idx[j] = idx[j], idx[i]
            return np.asarray(idx[:int(size)])
        a = np.asarray(a)
        assert size is None
        return a[int(self._u() * len(a))]

    def randint(self, low, high=None, size=None, dtype=int):
        assert size is None
        if high is None:
            low, high = 0, low
        return int(low) + int(self._u() * (int(high) - int(low)))

    def uniform(self, low=0.0, high=1.0, size=None):
        assert size is None
        return low + (high - low) * self._u()

    def random(self, size=None):                       # target headings for the renderer only (environment.py:821): not on the tape
        return self._real.random_sample(size)

    def binomial(self, n, p, size=None):
        assert n == 1 and size is None
        other = sys._getframe(1).f_locals.get('other', None)
        u = float(self._real.random_sample())
        self._pair_log.append((self._owner, other, u))
        return int(u > 1.0 - p) if p <= 0.5 else int(u <= p)


def _tape_box_sample(self):
    """Box.sample of the build-owned gym stand-in with its uniforms on RESET_TAPE: low + (high - low) * U per element
    (what gym's own Box.sample computes for a bounded box through RandomState.uniform)."""
    rng = self.np_random
    real = rng._real if isinstance(rng, TapeDrivenRNG) else rng
    u = np.asarray(real.random_sample(self.shape), dtype=np.float64)
    # reset() builds four stand-in obstacles for the warehouses (environment.py:724-727); constructing an entity seeds its
    # degenerate boxes with the constant 0 and samples them (entities.py:54-55): not randomness of reset(), not on the tape
    frame, constructing = sys._getframe(1), False
    for _ in range(6):
        if frame is None:
            break
        constructing = constructing or frame.f_code.co_name == '__init__'
        frame = frame.f_back
    if not constructing:
        RESET_TAPE.extend(float(v) for v in u.ravel())
    return (self.low + (self.high - self.low) * u).astype(self.dtype)


def _underlying(rng):
    """The numpy RandomState behind any stack of this script's proxies."""
    while isinstance(rng, (RecordingRNG, TapeDrivenRNG)):
        rng = rng._real
    return rng


def reset_fixture(name, config, seed, overrides=None, env=None):
    """reset() of the reference with every RandomState replaced by a TapeDrivenRNG: the tape + the state it produced.
    `env`: an environment object that has already run an episode (two_episode_fixture) -- its generators go on where that episode
    left them; returns (environment, observations of the reset)."""
    if env is None:
        env = mate.make('MultiAgentTracking-v0', config=config, **(overrides or {}))
        env.seed(seed)
    pair_log = []
    for entity in list(env.cameras_ordered) + list(env.targets_ordered) + list(env.obstacles_ordered):
        box = entity.location_random_range
        box._np_random = TapeDrivenRNG(_underlying(box.np_random), pair_log, entity)
    env._np_random = TapeDrivenRNG(_underlying(env.np_random), pair_log, 'env')
    RESET_TAPE.clear()
    gym.spaces.Box.sample = _tape_box_sample
    try:
        cam_obs, tgt_obs = env.reset()
    finally:
        gym.spaces.Box.sample = _ORIG_BOX_SAMPLE
    tape = np.asarray(RESET_TAPE, dtype=np.float64)
    Nc, Nt, No = env.num_cameras, env.num_targets, env.num_obstacles
    tape_ct = np.full((Nc, Nt), np.nan)
    for cam, other, u in pair_log:
        if isinstance(other, Target):
            c, t = env.cameras.index(cam), env.targets.index(other)
            assert np.isnan(tape_ct[c, t])
            tape_ct[c, t] = u
    out = {
        'config_file': np.str_(config), 'seed': np.int64(seed),
        'overrides': np.str_(json.dumps(overrides or {}, sort_keys=True)),
        'num_cameras': np.int64(Nc), 'num_targets': np.int64(Nt), 'num_obstacles': np.int64(No),
        'transmittance': np.float64(env.obstacle_transmittance), 'max_episode_steps': np.int64(env.max_episode_steps),
        'sparse_reward': np.bool_(env._sparse_reward), 'freight_scale': np.float64(env.freight_scale),
        'bounty_scale': np.float64(env.bounty_scale), 'reward_scale': np.float64(env.reward_scale),
        'max_target_team_episode_reward': np.float64(env.max_target_team_episode_reward),
        'target_step_size': np.float64(env.target_step_size),
        'tape': tape, 'tape_ct': tape_ct,
    }
    for k, v in snapshot_static(env).items():
        if not k.startswith('lut_outer'):
            out['static/' + k] = v
    for k, v in snapshot_dynamic(env).items():
        out['reset/' + k] = v
    out['reset/cam_obs'] = cam_obs
    out['reset/tgt_obs'] = tgt_obs
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    minimal = ((Nc + Nt + No - 3 if env.shuffle_entities else 0) + (env.num_high_capacity_targets if env.shuffle_entities else 0)
               + 4 * Nc + 3 * No + 2 * Nt + 2 * env.num_cargoes_per_target * Nt)
    print(f'{name}: {len(tape)} draws ({len(tape) - minimal} beyond the retry-free minimum without goal draws), '
          f'{int(np.isfinite(tape_ct).sum())} in-sector pairs, zero-radius obstacles={int((out["static/obs_xyr"][:, 2] == 0).sum()) if No else 0}, '
          f'{os.path.getsize(path) / 1024:.0f} KiB')
    # the plain generators back in place: whoever steps this environment next installs its own proxies
    for entity in list(env.cameras_ordered) + list(env.targets_ordered) + list(env.obstacles_ordered):
        entity.location_random_range._np_random = _underlying(entity.location_random_range.np_random)
    env._np_random = _underlying(env.np_random)
    return env, (cam_obs, tgt_obs)


def two_episode_fixture(tag, config, seed, steps1, steps2, policy='greedy'):
    """ONE reference environment through an episode end (environment.py:629-632) and the reset() behind it (shuffle_entities on: the
    shipped scenarios' default), as three fixtures in the standard formats:
      trace_<tag>_<policy>_ep1_s<seed>   an episode that ends inside the trace (only the cargoes in transit are left: tweak_few_cargoes)
      reset_<tag>_ep2_s<seed>            the reset() of that SAME object behind it, on a tape (its generators go on where episode 1 left them)
      trace_<tag>_<policy>_ep2_s<seed>   the steps of the second episode
    tests/test_gpu_parity.py::test_two_episodes_through_a_recorded_reset replays all three on one engine, in order."""
    env = make_trace(f'trace_{tag}_{policy}_ep1_s{seed}', config, seed, policy, steps1, None, tweak_few_cargoes)
    assert env.episode_step > 0
    env, first_obs = reset_fixture(f'reset_{tag}_ep2_s{seed}', config, seed, env=env)
    make_trace(f'trace_{tag}_{policy}_ep2_s{seed}', config, seed + 100, policy, steps2, env=env, first_obs=first_obs)


def xform_fixture(trace_name, steps):
    """Reference outputs of the observation post-processing wrappers (RelativeCoordinates =
    agents/utils.py:40-94 `convert_coordinates`, RescaledObservation = :97-137 `rescale_observation`) applied
    to the observations already recorded in a trace fixture.  Stored as f32."""
    from mate.agents.utils import convert_coordinates, rescale_observation
    from mate.utils import Team
    fx = np.load(os.path.join(HERE, trace_name + '.npz'))
    Nc, Nt, No = int(fx['num_cameras']), int(fx['num_targets']), int(fx['num_obstacles'])
    out = {'trace': np.str_(trace_name), 'steps': np.int64(steps)}
    for team, key in ((Team.CAMERA, 'cam_obs'), (Team.TARGET, 'tgt_obs')):
        if team is Team.CAMERA and Nc == 0:
            continue
        obs = fx['step/' + key][:steps]
        rel = np.stack([convert_coordinates(o, team, Nc, Nt, No) for o in obs])
        res = np.stack([rescale_observation(o, team, Nc, Nt, No) for o in obs])
        relres = np.stack([rescale_observation(o, team, Nc, Nt, No) for o in rel])
        out[key + '_relative'] = rel.astype(np.float32)
        out[key + '_rescaled'] = res.astype(np.float32)
        out[key + '_relative_rescaled'] = relres.astype(np.float32)
    path = os.path.join(HERE, 'xform_' + trace_name[6:] + '.npz')
    np.savez_compressed(path, **out)
    print(f'xform_{trace_name[6:]}: {steps} steps, {os.path.getsize(path) / 1024:.0f} KiB')


def chain_fixture(name, config, seed, learner_steps, frame_skip=5, levels=5, coefficients=None, reduction='mean', team='camera'):
    """The wrapper chain every example trainer's make_env builds for a camera learner (examples/ippo/camera/config.py:19-51, and its
    qmix / mappo / ... siblings): base -> DiscreteCamera(levels) -> MultiCamera(GreedyTargetAgent(seed=0)) -> RelativeCoordinates ->
    RescaledObservation -> RepeatedRewardIndividualDone -> AuxiliaryCameraRewards(coverage_rate, 'mean') -> FrameSkip(5) -- or, with
    team='target', for a target learner (examples/ippo/target/config.py:20-51: DiscreteTarget, MultiTarget(GreedyCameraAgent(seed=0)),
    AuxiliaryTargetRewards, FrameSkip(10) on MATE-2v4-0) -- all of them the REFERENCE's own classes from mate.wrappers, except
    FrameSkip, which lives in examples/utils/wrappers.py:254-323 behind `ray` imports: it is restated here as what it does (the same
    action for `frame_skip` env.step calls, rewards summed, stop when all(dones)).  Recorded per FRAME: the learner's grid indices, the
    opponents' joint action and every random draw (environment and agents), the chain's observations of the learner's team (relative
    + rescaled), the shaped and raw rewards, dones, masks, state; per LEARNER STEP: the FrameSkip sums and the observation it
    returns.  Keys carry the learner's team: cam_idx / chain_cam_obs / chain_reward_cam, or tgt_idx / chain_tgt_obs / chain_reward_tgt."""
    import mate.wrappers.single_team as single_team
    me, opp = ('cam', 'tgt') if team == 'camera' else ('tgt', 'cam')
    base = mate.make('MultiAgentTracking-v0', config=config, reward_type='dense')
    if team == 'camera':
        coefficients = coefficients or {'coverage_rate': 1.0}
        disc = mate.DiscreteCamera(base, levels=levels)
        multi = mate.MultiCamera(disc, target_agent=GreedyTargetAgent(seed=0))
        shaper = mate.AuxiliaryCameraRewards
    else:
        disc = mate.DiscreteTarget(base, levels=levels)
        multi = mate.MultiTarget(disc, camera_agent=GreedyCameraAgent(seed=0))
        shaper = mate.AuxiliaryTargetRewards
    chain = shaper(mate.RepeatedRewardIndividualDone(mate.RescaledObservation(mate.RelativeCoordinates(multi))),
                   coefficients=coefficients, reduction=reduction)
    chain.seed(seed)
    opp_agents = multi.opponent_agents_ordered
    cam_agents, tgt_agents = ([], opp_agents) if team == 'camera' else (opp_agents, [])
    gym.spaces.Box.sample = _recording_box_sample
    for agent in opp_agents:
        agent._np_random = AgentRNG(agent.np_random, AGENT_LOG, None)
    AGENT_LOG.clear()
    opponent_actions = []
    real_group_step = single_team.group_step

    def recording_group_step(env, agents, observation, infos=None, **kwargs):
        action = real_group_step(env, agents, observation, infos, **kwargs)
        opponent_actions.append(np.asarray(action, dtype=np.float64))
        return action

    single_team.group_step = recording_group_step
    try:
        my_obs = chain.reset()
        for agent in opp_agents:
            agent._np_random._who = (opp, agent.index)
        reset_draws = drain_agent_log(cam_agents, tgt_agents)
        log = []
        install_proxies(base, log)
        Nc, Nt, No = base.num_cameras, base.num_targets, base.num_obstacles
        n_me, n_opp = (Nc, Nt) if team == 'camera' else (Nt, Nc)
        out = {
            'config_file': np.str_(config), 'seed': np.int64(seed), 'policy': np.str_('chain'), 'learner_team': np.str_(team),
            'num_cameras': np.int64(Nc), 'num_targets': np.int64(Nt), 'num_obstacles': np.int64(No),
            'transmittance': np.float64(base.obstacle_transmittance), 'max_episode_steps': np.int64(base.max_episode_steps),
            'sparse_reward': np.bool_(base._sparse_reward), 'freight_scale': np.float64(base.freight_scale),
            'bounty_scale': np.float64(base.bounty_scale), 'reward_scale': np.float64(base.reward_scale),
            'max_target_team_episode_reward': np.float64(base.max_target_team_episode_reward),
            'target_step_size': np.float64(base.target_step_size),
            'frame_skip': np.int64(frame_skip), 'discrete_levels': np.int64(levels),
            ('camera' if team == 'camera' else 'target') + '_action_grid': disc.normalized_action_grid,
            'aux_keys': np.asarray(list(coefficients.keys())), 'aux_coefficients': np.asarray(list(coefficients.values()), dtype=np.float64),
            'aux_reduction': np.str_(reduction),
            'agent/tgt_reset_sample_u': reset_draws['tgt_sample_u'],
        }
        for k, v in snapshot_static(base).items():
            out['static/' + k] = v
        for k, v in snapshot_dynamic(base).items():
            out['reset/' + k] = v
        out[f'reset/chain_{me}_obs'] = np.asarray(my_obs, dtype=np.float64)
        per_step, per_skip = {}, {}

        def push(store, key, value):
            store.setdefault(key, []).append(np.asarray(value))

        rng = np.random.RandomState(seed + 1000)
        finished = False
        for ls in range(learner_steps):
            my_idx = rng.randint(0, levels ** 2, size=n_me)
            if team == 'target':                             # a learner that mostly heads for its goal: the chain's deliveries get exercised
                import mate.constants as consts
                grid = np.asarray(disc.normalized_action_grid, dtype=np.float64)
                for i, target in enumerate(base.targets):
                    if target.goal_bits.any() and rng.random_sample() < 0.8:
                        heading = consts.WAREHOUSES[int(np.argmax(target.goal_bits))] - np.asarray(target.location, dtype=np.float64)
                        heading /= max(np.abs(heading).max(), 1e-9)
                        my_idx[i] = int(np.argmin(((grid - heading) ** 2).sum(axis=1)))
            fragment_rewards, frames = [], 0
            for f in range(frame_skip):                      # FrameSkip.step (examples/utils/wrappers.py:301-323)
                log.clear()
                observations, rewards, dones, infos = chain.step(my_idx)
                fragment_rewards.append(rewards)
                frames += 1
                tape_ct, _, goal_u, goal_k, goal_j = drain_log(base, log)
                for k, v in drain_agent_log(cam_agents, tgt_agents).items():
                    if k.startswith(opp + '_'):
                        push(per_step, 'agent_' + k, v)
                push(per_step, me + '_idx', my_idx)
                decoded = disc.action((my_idx, None))[0] if team == 'camera' else disc.action((None, my_idx))[1]
                push(per_step, me + '_act', np.asarray(decoded, dtype=np.float64).reshape(n_me, 2))
                push(per_step, opp + '_act', opponent_actions.pop().reshape(n_opp, 2))
                assert not opponent_actions
                push(per_step, 'tape_ct', tape_ct)
                push(per_step, 'goal_u', goal_u)
                push(per_step, 'goal_k', goal_k)
                push(per_step, 'goal_j', goal_j)
                push(per_step, f'chain_{me}_obs', np.asarray(observations, dtype=np.float64))
                push(per_step, f'chain_reward_{me}', np.asarray(rewards, dtype=np.float64))
                push(per_step, 'reward_' + me, infos[0]['raw_reward'])
                push(per_step, 'info_coverage_rate', infos[0]['coverage_rate'])
                if team == 'target':                         # AuxiliaryTargetRewards' per-target terms (auxiliary_target_rewards.py:118-216)
                    for key in coefficients:
                        push(per_step, 'aux_' + key, np.asarray([info['auxiliary_reward_' + key] for info in infos], dtype=np.float64))
                push(per_step, 'done', bool(dones[0]))
                push(per_step, 'learner_step', ls)
                for k, v in snapshot_dynamic(base).items():
                    push(per_step, k, v)
                if all(dones):
                    finished = True
                    break
            push(per_skip, me + '_idx', my_idx)
            push(per_skip, 'frames', frames)
            push(per_skip, 'reward_' + me, np.sum(fragment_rewards, axis=0))
            push(per_skip, f'chain_{me}_obs', np.asarray(observations, dtype=np.float64))
            push(per_skip, 'done', bool(dones[0]))
            if finished:
                break
    finally:
        single_team.group_step = real_group_step
        gym.spaces.Box.sample = _ORIG_BOX_SAMPLE
    for k, v in per_step.items():
        out['step/' + k] = np.stack(v)
    for k, v in per_skip.items():
        out['skip/' + k] = np.stack(v)
    out.pop('step/state', None)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    nsteps = len(per_step['done'])
    print(f'{name}: {len(per_skip["done"])} learner steps = {nsteps} frames, delivered={int(out["step/num_delivered_cargoes"][-1])}, '
          f'in-sector draws={int(np.isfinite(out["step/tape_ct"]).sum())}, mean shaped reward={float(out[f"step/chain_reward_{me}"].mean()):.3f}, '
          f'{os.path.getsize(path) / 1024:.0f} KiB')


def main():
    check_binomial_model()
    if sys.argv[1:] == ['chain']:
        chain_fixture('chain_4v8-9_s15', 'MATE-4v8-9.yaml', 15, learner_steps=13)
        chain_fixture('chain_target_2v4-0_s16', 'MATE-2v4-0.yaml', 16, learner_steps=24, frame_skip=10, team='target', reduction='none',
                      coefficients={'raw_reward': 1.0, 'normalized_goal_distance': -0.5, 'is_tracked': -0.25, 'is_colliding': -1.0,
                                    'sparse_delivery': 5.0})
        return
    if sys.argv[1:] == ['agents']:
        make_trace('greedy_4v8-9_s5', 'MATE-4v8-9.yaml', 5, 'greedy', 300, record_agents=True)
        make_trace('greedy_8v8-9_s6', 'MATE-8v8-9.yaml', 6, 'greedy', 200, record_agents=True)
        make_trace('greedy_4v2-9_s7', 'MATE-4v2-9.yaml', 7, 'greedy', 200, record_agents=True)
        return
    if sys.argv[1:] == ['discrete']:
        make_trace('discrete_4v8-9_s6', 'MATE-4v8-9.yaml', 6, 'discrete', 64, discrete_levels=(5, 5))
        make_trace('discrete_4v2-9_s7', 'MATE-4v2-9.yaml', 7, 'discrete', 48, discrete_levels=(3, 9))
        return
    if sys.argv[1:] == ['wrappers']:
        make_trace('obsmode_4v8-9_s4', 'MATE-4v8-9.yaml', 4, 'greedy', 72, extra_factory=observation_mode_extras(('both', 'camera', 'target')))
        make_trace('obsmode_4v8-9_fewcargo', 'MATE-4v8-9.yaml', 3, 'greedy', 160, tweak=tweak_few_cargoes,
                   extra_factory=observation_mode_extras(('both',)))
        make_trace('obsmode_nav_s2', 'MATE-Navigation.yaml', 2, 'greedy', 48, extra_factory=observation_mode_extras(('target',)))
        make_trace('discrete_4v8-9_s6', 'MATE-4v8-9.yaml', 6, 'discrete', 64, discrete_levels=(5, 5))
        make_trace('discrete_4v2-9_s7', 'MATE-4v2-9.yaml', 7, 'discrete', 48, discrete_levels=(3, 9))
        return
    if sys.argv[1:] == ['softcov']:
        make_trace('softcov_4v8-9_s8', 'MATE-4v8-9.yaml', 8, 'greedy', 64,
                   aux_rewards=({'raw_reward': 1.0, 'soft_coverage_score': 0.5, 'num_tracked': 0.25, 'coverage_rate': 2.0}, 'none'))
        make_trace('softcov_8v8-9_s9', 'MATE-8v8-9.yaml', 9, 'greedy', 48,
                   aux_rewards=({'soft_coverage_score': 1.0, 'real_coverage_rate': 1.0, 'baseline': -0.5}, 'mean'))
        make_trace('softcov_4v2-9_s10', 'MATE-4v2-9.yaml', 10, 'random', 48,
                   aux_rewards=({'soft_coverage_score': 1.0, 'mean_transport_rate': 3.0}, 'max'))
        return
    if sys.argv[1:] == ['auxtarget']:
        every = {'raw_reward': 1.0, 'coverage_rate': -0.5, 'real_coverage_rate': -0.25, 'mean_transport_rate': 2.0,
                 'normalized_goal_distance': -1.5, 'sparse_delivery': 100.0, 'soft_coverage_score': -0.75, 'is_tracked': -0.125,
                 'is_colliding': -3.0, 'baseline': 0.0625}
        make_trace('auxtgt_4v8-9_s11', 'MATE-4v8-9.yaml', 11, 'greedy', 120, aux_target_rewards=(every, 'none'))
        make_trace('auxtgt_8v8-9_s12', 'MATE-8v8-9.yaml', 12, 'greedy', 64, aux_target_rewards=(every, 'mean'))
        make_trace('auxtgt_4v2-9_s13', 'MATE-4v2-9.yaml', 13, 'random', 64,
                   aux_target_rewards=({k: every[k] for k in ('raw_reward', 'normalized_goal_distance', 'is_colliding', 'soft_coverage_score')}, 'max'))
        make_trace('auxtgt_nav_s14', 'MATE-Navigation.yaml', 14, 'greedy', 48,
                   aux_target_rewards=({k: every[k] for k in ('raw_reward', 'normalized_goal_distance', 'sparse_delivery', 'is_colliding')}, 'sum'))
        return
    if sys.argv[1:] == ['reset']:
        for cfg_name, tag in (('MATE-4v2-9.yaml', '4v2-9'), ('MATE-4v8-9.yaml', '4v8-9'), ('MATE-8v8-9.yaml', '8v8-9'),
                              ('MATE-4v8-0.yaml', '4v8-0'), ('MATE-Navigation.yaml', 'nav')):
            for seed in (0, 1, 2):
                reset_fixture(f'reset_{tag}_s{seed}', cfg_name, seed)
        # the branches the shipped scenarios do not take
        reset_fixture('reset_4v8-9_noshuffle_s3', 'MATE-4v8-9.yaml', 3, {'shuffle_entities': False})
        reset_fixture('reset_4v8-9_nocargo_s4', 'MATE-4v8-9.yaml', 4, {'targets_start_with_cargoes': False, 'high_capacity_target_split': 0.25})
        reset_fixture('reset_8v8-9_crowded_s5', 'MATE-8v8-9.yaml', 5, {'num_cargoes_per_target': 4, 'high_capacity_target_split': 1.0,
                                                                       'obstacle': {'radius_random_range': [90.0, 100.0]}})
        reset_fixture('reset_2v4-9_s6', 'MATE-2v4-9.yaml', 6, {'high_capacity_target_split': 0.0})
        # two obstacles pinned to the same spot: the second exhausts its 500 retries and gets radius 0 (environment.py:734-736)
        reset_fixture('reset_4v2-3_stacked_s7', 'MATE-4v2-9.yaml', 7,
                      {'obstacle': {'location_random_range': [[0, 0, 0, 0], [0, 0, 0, 0], [300, 320, 300, 320]], 'radius_random_range': [50.0, 60.0]}})
        return
    if sys.argv[1:] == ['xform']:
        xform_fixture('trace_4v8-9_greedy_s2', 48)
        xform_fixture('trace_nav_greedy_s1', 32)
        return
    if sys.argv[1:] == ['episodes']:
        two_episode_fixture('8v8-9', 'MATE-8v8-9.yaml', 21, 500, 96)
        two_episode_fixture('4v2-9', 'MATE-4v2-9.yaml', 22, 500, 64)
        return
    if not sys.argv[1:]:
        kat_obstruct()
        kat_scalar()
        kat_perceive()
    plan = [
        # name,                      config,                 seed, policy,  steps, overrides, tweak
        ('trace_4v2-9_random_s0',    'MATE-4v2-9.yaml',        0, 'random',   96, None, None),
        ('trace_4v2-9_greedy_s1',    'MATE-4v2-9.yaml',        1, 'greedy',  192, None, None),
        ('trace_4v8-9_random_s0',    'MATE-4v8-9.yaml',        0, 'random',   96, None, None),
        ('trace_4v8-9_random_s1',    'MATE-4v8-9.yaml',        1, 'random',   64, None, None),
        ('trace_4v8-9_greedy_s2',    'MATE-4v8-9.yaml',        2, 'greedy',  256, None, None),
        ('trace_4v8-9_greedy_done',  'MATE-4v8-9.yaml',        3, 'greedy',  400, None, tweak_few_cargoes),
        ('trace_4v8-9_timelimit',    'MATE-4v8-9.yaml',        4, 'random',   16, {'max_episode_steps': 8}, None),
        ('trace_8v8-9_random_s0',    'MATE-8v8-9.yaml',        0, 'random',   64, None, None),
        ('trace_8v8-9_greedy_s1',    'MATE-8v8-9.yaml',        1, 'greedy',  256, None, None),
        ('trace_4v8-0_random_s0',    'MATE-4v8-0.yaml',        0, 'random',   64, None, None),
        ('trace_4v8-0_greedy_s1',    'MATE-4v8-0.yaml',        1, 'greedy',  192, None, None),
        ('trace_nav_random_s0',      'MATE-Navigation.yaml',   0, 'random',   96, None, None),
        ('trace_nav_greedy_s1',      'MATE-Navigation.yaml',   1, 'greedy',  256, None, None),
        # third seeds of configurations 1, 3, 4, 5 (>= 3 F2 traces per configuration; configuration 2 has five)
        ('trace_4v2-9_random_s3',    'MATE-4v2-9.yaml',        3, 'random',   96, None, None),
        ('trace_8v8-9_greedy_s3',    'MATE-8v8-9.yaml',        3, 'greedy',  160, None, None),
        ('trace_4v8-0_greedy_s3',    'MATE-4v8-0.yaml',        3, 'greedy',  160, None, None),
        ('trace_nav_random_s3',      'MATE-Navigation.yaml',   3, 'random',  128, None, None),
        # the small scenarios whose kernels step four environments per wave (round 6): the target trainers' MATE-2v4-0 and two more shapes
        ('trace_2v4-0_greedy_s5',    'MATE-2v4-0.yaml',        5, 'greedy',  160, None, None),
        ('trace_2v2-9_random_s6',    'MATE-2v2-9.yaml',        6, 'random',   96, None, None),
        ('trace_1v1-9_greedy_s7',    'MATE-1v1-9.yaml',        7, 'greedy',  128, None, None),
    ]
    only = sys.argv[1:]
    for name, config, seed, policy, steps, overrides, tweak in plan:
        if only and not any(o in name for o in only):
            continue
        make_trace(name, config, seed, policy, steps, overrides, tweak)


if __name__ == '__main__':
    main()
