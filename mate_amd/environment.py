"""The environment boundary: `MultiAgentTracking` (one environment, NumPy in/out -- the drop-in for
the reference class mate/environment.py:288) and `BatchedMultiAgentTracking` (N environments,
torch tensors resident in HBM).  Both are thin hosts over the HIP engine (`mate_amd.engine.Engine`);
there is no CPU simulation path in this package.
"""
import copy
from collections import OrderedDict, defaultdict, deque

import numpy as np
import torch

from mate_amd import constants as consts
from mate_amd import spaces
from mate_amd.config import DEFAULT_CONFIG_FILE, read_config
from mate_amd.engine import Engine
from mate_amd.utils import Message, Team, polar2cartesian

__all__ = ['MultiAgentTracking', 'BatchedMultiAgentTracking', 'EnvMeta']

try:  # reference wrappers are gym.Wrapper subclasses; inherit from gym.Env when gym exists
    import gym as _gym
    _EnvBase = _gym.Env
except Exception:  # pragma: no cover
    _gym = None

    class _EnvBase:
        metadata = {'render.modes': []}
        reward_range = (-float('inf'), float('inf'))
        spec = None

        @property
        def unwrapped(self):
            return self

        def __str__(self):
            return f'<{type(self).__name__}<{getattr(self.spec, "id", "MultiAgentTracking-v0")}>>'


class EnvMeta(type(_EnvBase)):
    """isinstance(wrapped_env, MultiAgentTracking) looks through wrapper chains (environment.py:272-284)."""

    def __instancecheck__(cls, instance):
        if super().__instancecheck__(instance):
            return True
        while hasattr(instance, 'env') and not super().__instancecheck__(instance):
            instance = instance.env
        return super().__instancecheck__(instance)


# --------------------------------------------------------------------------------------------- entity views
class _EntityView:
    """Read-only window onto one entity of one environment (the reference exposes live objects)."""

    def __init__(self, env, index):
        self._env, self.index = env, index

    def _f(self, name):
        return self._env._fields()[name]

    @property
    def x(self):
        return self.location[0]

    @property
    def y(self):
        return self.location[1]

    def distance(self, other):
        other = other.location if isinstance(other, _EntityView) else np.asarray(other, dtype=np.float64)
        return float(np.linalg.norm(self.location - other))

    def __sub__(self, other):
        return self.location - other.location


class ObstacleView(_EntityView):
    @property
    def location(self):
        return np.array([self._f('obs_x')[self.index], self._f('obs_y')[self.index]])

    @property
    def radius(self):
        return float(self._f('obs_radius')[self.index])

    @property
    def transmittance(self):
        return self._env.obstacle_transmittance

    def state(self, private=False):
        return np.append(self.location, self.radius).astype(np.float64)


class CameraView(_EntityView):
    @property
    def location(self):
        return np.array([self._f('cam_x')[self.index], self._f('cam_y')[self.index]])

    @property
    def radius(self):
        return float(self._env.config['camera']['radius'])

    @property
    def orientation(self):
        return float(self._f('cam_phi')[self.index])

    @property
    def viewing_angle(self):
        return float(self._f('cam_theta')[self.index])

    @property
    def min_viewing_angle(self):
        return float(self._env.camera_min_viewing_angle)

    @property
    def max_sight_range(self):
        return float(self._env.camera_max_sight_range)

    @property
    def rotation_step(self):
        return float(self._env.camera_rotation_step)

    @property
    def zooming_step(self):
        return float(self._env.camera_zooming_step)

    @property
    def area_product(self):
        return self.min_viewing_angle * self.max_sight_range ** 2

    @property
    def sight_range(self):
        return float(np.sqrt(self.area_product / self.viewing_angle))

    def state(self, private=False):
        out = np.concatenate([self.location, [self.radius], polar2cartesian(self.sight_range, self.orientation), [self.viewing_angle]])
        if private:
            out = np.append(out, [self.max_sight_range, self.rotation_step, self.zooming_step])
        return out.astype(np.float64)

    def _table(self, outer=False):
        engine = self._env.engine
        if outer and not getattr(engine, 'outer_capacity', 0):
            engine.enable_outer_boundary()      # built from now on at every reset; build it once for the running episode
            engine.rebuild_luts()
        return engine.lut_read(0, self.index, outer=outer)

    def sight_range_at(self, angle, outer=False):
        """Camera.sight_range_at (entities.py:507-511): linear interpolation of the (outer) occlusion boundary."""
        phis, rhos = self._table(outer)
        return float(np.interp((angle + 180.0) % 360.0 - 180.0, phis, rhos))

    def boundary_between(self, angle_left, angle_right, outer=False):
        """Knots of the occlusion boundary inside a sector (entities.py:513-543); like the reference, the two end
        points are interpolated on the INNER boundary whichever table the knots come from."""
        assert 0.0 < angle_right - angle_left <= 360.0
        phis_all, rhos_all = self._table(outer)
        left = (angle_left + 180.0) % 360.0 - 180.0
        right = left + (angle_right - angle_left)
        if right <= 180.0:
            pick = (left < phis_all) & (phis_all < right)
            phis, rhos = phis_all[pick], rhos_all[pick]
        else:
            a = (left < phis_all) & (phis_all <= 180.0)
            b = (phis_all > -180.0) & (phis_all < right - 360.0)
            phis, rhos = np.concatenate([phis_all[a], phis_all[b]]), np.concatenate([rhos_all[a], rhos_all[b]])
        phis = np.concatenate([[left], phis, [right]])
        rhos = np.concatenate([[self.sight_range_at(left)], rhos, [self.sight_range_at(right)]])
        return phis.astype(np.float64), rhos.astype(np.float64)


class TargetView(_EntityView):
    @property
    def location(self):
        return np.array([self._f('tgt_x')[self.index], self._f('tgt_y')[self.index]])

    radius = consts.TARGET_RADIUS

    @property
    def capacity(self):
        return int(self._f('tgt_capacity')[self.index])

    @property
    def step_size(self):
        return self._env.target_step_size / self.capacity

    @property
    def sight_range(self):
        return float(self._env.target_sight_range)

    @property
    def is_colliding(self):
        return bool(self._f('tgt_colliding')[self.index])

    @property
    def goal_bits(self):
        return self._f('tgt_goal_bits')[self.index].astype(np.int64)

    @property
    def empty_bits(self):
        return self._f('tgt_empty_bits')[self.index].astype(bool)

    @property
    def is_loaded(self):
        return bool(self.goal_bits.any())

    def state(self, private=False):
        out = np.append(self.location, [self.sight_range, self.is_loaded])
        if private:
            out = np.concatenate([out, [self.step_size, self.capacity], self.goal_bits, self.empty_bits])
        return out.astype(np.float64)


# --------------------------------------------------------------------------------------------- shared host logic
class _ScenarioMixin:
    """Configuration-derived attributes shared by the single and the batched environment."""

    def _setup_scenario(self, config, kwargs):
        if config is None:
            config = {} if len(kwargs) > 0 else DEFAULT_CONFIG_FILE
        self.config = read_config(config, **kwargs)

    # properties with the reference's names (environment.py:1396-1560)
    name = property(lambda self: self.config['name'])
    max_episode_steps = property(lambda self: self.config['max_episode_steps'])
    camera_min_viewing_angle = property(lambda self: self.config['camera']['min_viewing_angle'])
    camera_max_sight_range = property(lambda self: self.config['camera']['max_sight_range'])
    camera_rotation_step = property(lambda self: self.config['camera']['rotation_step'])
    camera_zooming_step = property(lambda self: self.config['camera']['zooming_step'])
    target_step_size = property(lambda self: self.config['target']['step_size'])
    target_sight_range = property(lambda self: self.config['target']['sight_range'])
    num_cargoes_per_target = property(lambda self: self.config['num_cargoes_per_target'])
    targets_start_with_cargoes = property(lambda self: self.config.get('targets_start_with_cargoes', True))
    bounty_factor = property(lambda self: max(0.0, self.config.get('bounty_factor', 1.0)))
    obstacle_transmittance = property(lambda self: min(max(0.0, self.config.get('obstacle', {}).get('transmittance', 0.0)), 1.0))
    shuffle_entities = property(lambda self: self.config.get('shuffle_entities', True))
    num_warehouses = property(lambda self: consts.NUM_WAREHOUSES)
    high_capacity_target_split = property(lambda self: min(max(0.0, self.config.get('high_capacity_target_split', 0.5)), 1.0))
    num_high_capacity_targets = property(lambda self: int(self.num_targets * self.high_capacity_target_split))
    num_low_capacity_targets = property(lambda self: self.num_targets - self.num_high_capacity_targets)
    camera_observation_dim = property(lambda self: self.camera_observation_space.shape[-1])
    target_observation_dim = property(lambda self: self.target_observation_space.shape[-1])

    def _setup_spaces(self):
        Nc, Nt, No = self.num_cameras, self.num_targets, self.num_obstacles

        def box(lo, hi):
            return spaces.Box(low=np.asarray(lo, dtype=np.float64), high=np.asarray(hi, dtype=np.float64), dtype=np.float64)

        if Nc > 0:
            rot, zoom = self.camera_rotation_step, self.camera_zooming_step
            self.camera_action_space = box([-rot, -zoom], [rot, zoom])
        else:
            self.camera_action_space = box([0.0, 0.0], [0.0, 0.0])
        # the reference merges per-target boxes with an element-wise min (environment.py:369-378):
        # with mixed capacities the joint bound is the full step size of a capacity-1 target
        step = self.target_step_size
        self.target_action_space = box([-step, -step], [step, step])
        self.camera_joint_action_space = spaces.Tuple((self.camera_action_space,) * Nc)
        self.target_joint_action_space = spaces.Tuple((self.target_action_space,) * Nt)
        self.action_space = spaces.Tuple((self.camera_joint_action_space, self.target_joint_action_space))
        self.camera_observation_space = consts.camera_observation_space_of(Nc, Nt, No)
        self.target_observation_space = consts.target_observation_space_of(Nc, Nt, No)
        self.camera_joint_observation_space = spaces.Tuple((self.camera_observation_space,) * Nc)
        self.target_joint_observation_space = spaces.Tuple((self.target_observation_space,) * Nt)
        self.observation_space = spaces.Tuple((self.camera_joint_observation_space, self.target_joint_observation_space))
        self.camera_state_space_public, self.camera_state_space_private = consts.CAMERA_STATE_SPACE_PUBLIC, consts.CAMERA_STATE_SPACE_PRIVATE
        self.target_state_space_public, self.target_state_space_private = consts.TARGET_STATE_SPACE_PUBLIC, consts.TARGET_STATE_SPACE_PRIVATE
        self.obstacle_state_space = consts.OBSTACLE_STATE_SPACE
        tail = 2 * Nt + consts.NUM_WAREHOUSES ** 2
        low = np.concatenate([consts.PRESERVED_SPACE.low] + [consts.CAMERA_STATE_SPACE_PRIVATE.low] * Nc + [consts.TARGET_STATE_SPACE_PRIVATE.low] * Nt
                             + [consts.OBSTACLE_STATE_SPACE.low] * No + [np.zeros(tail)])
        high = np.concatenate([consts.PRESERVED_SPACE.high] + [consts.CAMERA_STATE_SPACE_PRIVATE.high] * Nc + [consts.TARGET_STATE_SPACE_PRIVATE.high] * Nt
                              + [consts.OBSTACLE_STATE_SPACE.high] * No + [np.full(tail, np.inf)])
        self.state_space = box(low, high)
        self.freight_scale = float(np.ceil(consts.TERRAIN_WIDTH / self.target_step_size))
        self.bounty_scale = float(np.ceil(self.freight_scale * self.bounty_factor))
        self.reward_scale = self.freight_scale + self.bounty_scale
        self.max_target_team_episode_reward = self.reward_scale * self.num_cargoes_per_target * Nt


class MultiAgentTracking(_ScenarioMixin, _EnvBase, metaclass=EnvMeta):
    """One Multi-Agent Tracking environment on the MI355X engine, with the reference's Python API:
    `seed / reset / step / state / joint_observation / send_messages / receive_messages / load_config /
    close` and the attribute set its wrappers and harnesses read."""

    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 60, 'video.output_frames_per_second': 60}
    DEFAULT_CONFIG_FILE = DEFAULT_CONFIG_FILE

    def __init__(self, config=None, device=0, obs_dtype=torch.float64, **kwargs):
        self._setup_scenario(config, kwargs)
        self._device_index, self._obs_dtype = device, obs_dtype
        self._seed_value = 0
        self.engine = Engine(self.config, 1, device=device, seed=0, obs_dtype=obs_dtype)
        self.engine.stage_outputs()          # one device-to-host copy per step (Engine.fetch_host)
        Na = self.engine.num_cameras * consts.CAMERA_ACTION_DIM + self.engine.num_targets * consts.TARGET_ACTION_DIM
        self._act_dev = torch.zeros(Na, dtype=torch.float64, device=self.engine.device)      # ... and one host-to-device copy of the joint action,
        self._act_host = torch.zeros(Na, dtype=torch.float64, pin_memory=True)               # from page-locked memory
        self.num_cameras, self.num_targets, self.num_obstacles = self.engine.num_cameras, self.engine.num_targets, self.engine.num_obstacles
        self._setup_spaces()
        Nc, Nt, No = self.num_cameras, self.num_targets, self.num_obstacles
        self.cameras = [CameraView(self, c) for c in range(Nc)]
        self.targets = [TargetView(self, t) for t in range(Nt)]
        self.obstacles = [ObstacleView(self, o) for o in range(No)]
        self.cameras_ordered, self.targets_ordered, self.obstacles_ordered = list(self.cameras), list(self.targets), list(self.obstacles)
        self.preserved_data = np.concatenate([[Nc, Nt, No], [0], consts.WAREHOUSES.ravel(), [consts.WAREHOUSE_RADIUS]]).astype(np.float64)
        self._sparse_reward = self.config['reward_type'] == 'sparse'
        self.viewer = None
        self.render_callbacks = OrderedDict()
        self.camera_message_buffer, self.target_message_buffer = defaultdict(list), defaultdict(list)
        self.message_buffers = (self.camera_message_buffer, self.target_message_buffer)
        self.camera_message_queue, self.target_message_queue = defaultdict(deque), defaultdict(deque)
        self.message_queues = (self.camera_message_queue, self.target_message_queue)
        self.camera_communication_edges = np.zeros((Nc, Nc), dtype=np.int64)
        self.target_communication_edges = np.zeros((Nt, Nt), dtype=np.int64)
        self.camera_total_communication_edges = self.camera_communication_edges.copy()
        self.target_total_communication_edges = self.target_communication_edges.copy()
        self.communication_edges = (self.camera_communication_edges, self.target_communication_edges)
        self.coverage_rate = self.real_coverage_rate = self.mean_transport_rate = 0.0
        self.num_delivered_cargoes = 0
        self.episode_step = 0
        self._cache = None
        self._masks = None
        self._np_random = None
        self.seed(0)

    # ------------------------------------------------------------------ state access
    def _fields(self):
        if self._cache is None:
            self._cache = {k: v[0] for k, v in self.engine.state_dict().items()}
        return self._cache

    def _mask(self, name):
        if self._masks is None:
            self._masks = {k: v[0] for k, v in self.engine.unpack_masks().items()}
        return self._masks[name]

    camera_target_view_mask = property(lambda self: self._mask('camera_target_view_mask'))
    target_camera_view_mask = property(lambda self: self._mask('target_camera_view_mask'))
    target_obstacle_view_mask = property(lambda self: self._mask('target_obstacle_view_mask'))
    target_target_view_mask = property(lambda self: self._mask('target_target_view_mask'))
    camera_camera_view_mask = property(lambda self: self._mask('camera_camera_view_mask'))
    camera_obstacle_view_mask = property(lambda self: self._mask('camera_obstacle_view_mask'))
    tracked_bits = property(lambda self: self._mask('tracked_bits'))
    remaining_cargoes = property(lambda self: self._fields()['remaining_cargoes'].astype(np.int64))
    awaiting_cargo_counts = property(lambda self: self._fields()['awaiting_cargo_counts'].astype(np.int64))
    target_goals = property(lambda self: self._fields()['tgt_goals'].astype(np.int64))
    target_goal_bits = property(lambda self: self._fields()['tgt_goal_bits'].astype(np.int64))
    target_capacities = property(lambda self: self._fields()['tgt_capacity'].astype(np.int64))
    target_steps = property(lambda self: self._fields()['target_steps'].astype(np.int64))
    tracked_steps = property(lambda self: self._fields()['tracked_steps'].astype(np.int64))
    freights = property(lambda self: self._fields()['freights'].astype(np.int64))
    bounties = property(lambda self: self._fields()['bounties'].astype(np.int64))
    target_team_episode_reward = property(lambda self: float(self._fields()['episode_reward']))
    delayed_target_team_episode_reward = property(lambda self: float(self._fields()['delayed_episode_reward']))

    @property
    def obstacle_states(self):
        f = self._fields()
        return np.stack([f['obs_x'], f['obs_y'], f['obs_radius']], axis=-1).reshape(self.num_obstacles, 3)

    @property
    def obstacle_states_flagged(self):
        return np.hstack([self.obstacle_states, np.ones((self.num_obstacles, 1))])

    @property
    def target_warehouse_distances(self):
        f = self._fields()
        xy = np.stack([f['tgt_x'], f['tgt_y']], axis=-1)
        return np.linalg.norm(xy[:, None, :] - consts.WAREHOUSES[None], axis=-1)

    @property
    def np_random(self):
        if self._np_random is None:
            self.seed()
        return self._np_random

    # ------------------------------------------------------------------ gym API
    def seed(self, seed=None):
        """Seed the host RNG handed to callers and the engine's counter-based streams (environment.py:1203-1227).

        Returns the main seed followed by one seed per entity (cameras, targets, obstacles), drawn from the fresh main
        generator exactly as the reference draws them (`np_random.randint(int_max)` per entity), so `len(env.seed())`
        and the position of `env.np_random` afterwards match.  The engine itself needs one key: its streams are
        Philox counters keyed by (seed, environment, episode, tick), and seeding rewinds those counters, so the same
        seed always produces the same episodes -- like the reference, whose `__init__` also ends with `seed(0)`."""
        if seed is None:
            seed = int(np.random.SeedSequence().entropy % (2 ** 31))
        if not (isinstance(seed, (int, np.integer)) and seed >= 0):
            raise ValueError(f'Seed must be a non-negative integer or omitted, not {seed}')
        self._seed_value = int(seed)
        self._np_random = np.random.RandomState(self._seed_value % (2 ** 32))
        self.engine.seed(self._seed_value)
        int_max = np.iinfo(int).max
        entities = self.num_cameras + self.num_targets + self.num_obstacles
        return [self._seed_value] + [int(self._np_random.randint(int_max)) for _ in range(entities)]

    def _collect(self):
        """Fresh observations + the metrics of joint_observation (environment.py:966-979), computed here in f64 from the
        exported state and masks as the reference computes them (the engine's own scalar record is f32: the batched
        product dtype)."""
        host = self.engine.fetch_host()
        self._cache = {k: v[0] for k, v in self.engine.state_dict_from(host['state']).items()}
        self._masks = {k: v[0] for k, v in self.engine.unpack_masks(words_host=host['masks']).items()}
        cam = host['camera_obs'][0].astype(np.float64) if self.num_cameras else np.zeros((0, self.camera_observation_dim))
        tgt = host['target_obs'][0].astype(np.float64)
        scalars = host['scalars'][0].copy()
        f = self._fields()
        tracked = self.tracked_bits.astype(bool)
        with_bounty = f['bounties'] > 0
        self.coverage_rate = tracked.sum() / self.num_targets
        self.real_coverage_rate = float(np.logical_and(tracked, with_bounty).sum() / max(1, with_bounty.sum())) if with_bounty.any() else 0.0
        self.coverage_rate = float(self.coverage_rate)
        self.num_delivered_cargoes = int(f['num_delivered_cargoes'])
        self.mean_transport_rate = (float(f['delayed_episode_reward']) / (self.reward_scale * self.num_delivered_cargoes)
                                    if self.num_delivered_cargoes > 0 else 0.0)
        return cam, tgt, scalars

    def reset(self, *, seed=None):
        self._clear_messages(totals=True)
        if seed is not None:
            self.seed(seed)
        self.engine.reset()
        cam, tgt, _ = self._collect()
        self.target_dones = np.zeros(self.num_targets, dtype=bool)
        self._last_goals = self.target_goals.copy()
        self._last_episode_rewards = (self.target_team_episode_reward, self.delayed_target_team_episode_reward)
        self.episode_step = 0
        return cam, tgt

    #: parity hook: {'camera_target': [Nc, Nt] uniforms, 'goal': [Nt] uniforms} consumed by the NEXT step() instead of the
    #: engine's Philox draws (how the golden traces recorded from the reference are replayed through this API)
    step_tape = None

    def _tapes(self):
        tape, self.step_tape = self.step_tape, None
        if not tape:
            return None, None
        dev = self.engine.device
        ct = tape.get('camera_target')
        goal = tape.get('goal')
        ct = None if ct is None else torch.from_numpy(np.nan_to_num(np.asarray(ct, dtype=np.float64), nan=0.0)[None].copy()).to(dev)
        goal = None if goal is None else torch.from_numpy(np.nan_to_num(np.asarray(goal, dtype=np.float64), nan=0.0)[None].copy()).to(dev)
        return ct, goal

    def enable_greedy_policies(self):
        """Let the engine play GreedyCameraAgent vs GreedyTargetAgent itself (`step_greedy`); call before the reset()
        whose observations the agents first act on."""
        self.engine.enable_policies()
        self._greedy = True

    def step_greedy(self):
        """`env.step(mate.group_step(...))` of both teams with the reference's Greedy agents computed on the device
        (mate/agents/greedy.py through Engine.step_greedy); same return value as step()."""
        if not getattr(self, '_greedy', False):
            raise RuntimeError('enable_greedy_policies() must precede the reset() the agents first act on')
        tape_ct, tape_goal = self._tapes()
        self.engine.step_greedy(tape_ct=tape_ct, tape_goal=tape_goal, auto_reset=False)
        return self._finish_step()

    def step_versus_greedy(self, team, joint_action):
        """`mate.MultiCamera(env, target_agent=GreedyTargetAgent()).step(joint_action)` for team = 'camera' (or mate_amd.Team.CAMERA),
        `mate.MultiTarget(env, camera_agent=GreedyCameraAgent()).step(...)` for 'target' (mate/wrappers/single_team.py:245-264): the
        caller's team acts, the greedy opponents act on the device.  Same return value as step() (both teams' halves)."""
        if not getattr(self, '_greedy', False):
            raise RuntimeError('enable_greedy_policies() must precede the reset() the agents first act on')
        team = getattr(team, 'name', team)
        team = team.lower() if isinstance(team, str) else ('camera', 'target')[int(team)]
        assert team in ('camera', 'target'), f'Invalid team {team!r}.'
        agents, dim = (self.num_cameras, consts.CAMERA_ACTION_DIM) if team == 'camera' else (self.num_targets, consts.TARGET_ACTION_DIM)
        act = np.asarray(joint_action, dtype=np.float64).reshape(agents, dim)
        assert np.isfinite(act).all(), f'Got unexpected joint action {act}.'
        tape_ct, tape_goal = self._tapes()
        self.engine.step_versus_greedy(team, torch.from_numpy(act[None]).to(self.engine.device), tape_ct=tape_ct, tape_goal=tape_goal,
                                       auto_reset=False)
        return self._finish_step()

    def step(self, action):
        camera_joint_action, target_joint_action = action
        cam_act = np.asarray(camera_joint_action, dtype=np.float64).reshape(self.num_cameras, consts.CAMERA_ACTION_DIM)
        tgt_act = np.asarray(target_joint_action, dtype=np.float64).reshape(self.num_targets, consts.TARGET_ACTION_DIM)
        assert np.isfinite(cam_act).all(), f'Got unexpected joint action {cam_act}.'
        assert np.isfinite(tgt_act).all(), f'Got unexpected joint action {tgt_act}.'
        tape_ct, tape_goal = self._tapes()
        nc = cam_act.size
        act_np = self._act_host.numpy()
        act_np[:nc] = cam_act.ravel()
        act_np[nc:] = tgt_act.ravel()
        self._act_dev.copy_(self._act_host, non_blocking=True)      # (stream-ordered before the step; the next call rewrites the host buffer only after fetch_host's synchronise)
        self.engine.step(self._act_dev[:nc].view(1, self.num_cameras, consts.CAMERA_ACTION_DIM), self._act_dev[nc:].view(1, self.num_targets, consts.TARGET_ACTION_DIM),
                         tape_ct=tape_ct, tape_goal=tape_goal, auto_reset=False)
        return self._finish_step()

    def _finish_step(self):
        cam, tgt, scalars = self._collect()
        goals = self.target_goals
        self.target_dones = (goals != self._last_goals) & (self._last_goals >= 0)
        self._last_goals = goals.copy()
        self.episode_step += 1
        # the step's team reward in f64: the increment of the episode sums (integers: exact), environment.py:618-624
        sums = (self.target_team_episode_reward, self.delayed_target_team_episode_reward)
        r_tgt = float(sums[1] - self._last_episode_rewards[1]) if self._sparse_reward else float(sums[0] - self._last_episode_rewards[0])
        self._last_step_rewards = (float(sums[0] - self._last_episode_rewards[0]), float(sums[1] - self._last_episode_rewards[1]))
        self._last_episode_rewards = sums
        assert np.float32(r_tgt) == scalars[1]
        r_cam = -r_tgt
        done = bool(scalars[2])
        norm = r_tgt / self.max_target_team_episode_reward
        common = {'coverage_rate': self.coverage_rate, 'real_coverage_rate': self.real_coverage_rate,
                  'mean_transport_rate': self.mean_transport_rate, 'num_delivered_cargoes': self.num_delivered_cargoes}
        camera_infos = [dict(raw_reward=r_cam, normalized_raw_reward=-norm, messages=self.camera_message_buffer[c],
                             out_communication_edges=self.camera_communication_edges[c, :].sum(),
                             in_communication_edges=self.camera_communication_edges[:, c].sum(), **common)
                        for c in range(self.num_cameras)]
        target_infos = [dict(raw_reward=r_tgt, normalized_raw_reward=norm, messages=self.target_message_buffer[t],
                             out_communication_edges=self.target_communication_edges[t, :].sum(),
                             in_communication_edges=self.target_communication_edges[:, t].sum(), **common)
                        for t in range(self.num_targets)]
        self._clear_messages(totals=False)
        return (cam, tgt), (r_cam, r_tgt), done, (camera_infos, target_infos)

    def snapshot(self):
        """Everything a caller of the reference class can read after reset() / step(), as one dict of NumPy values keyed
        like the golden fixtures (tests/golden): the backend protocol of mate_amd.reference_adapter."""
        f = self._fields()
        Nc, Nt, No = self.num_cameras, self.num_targets, self.num_obstacles
        area = self.config['camera']['min_viewing_angle'] * self.config['camera']['max_sight_range'] ** 2 if Nc else 0.0
        last = getattr(self, '_last_step_rewards', (0.0, 0.0))
        snap = {
            'cam_xy': np.stack([f['cam_x'], f['cam_y']], axis=-1).reshape(Nc, 2), 'cam_phi': f['cam_phi'].copy(), 'cam_theta': f['cam_theta'].copy(),
            'cam_sight': np.sqrt(area / f['cam_theta']) if Nc else np.zeros(0),
            'obs_xyr': np.stack([f['obs_x'], f['obs_y'], f['obs_radius']], axis=-1).reshape(No, 3),
            'tgt_xy': np.stack([f['tgt_x'], f['tgt_y']], axis=-1).reshape(Nt, 2), 'tgt_capacity': f['tgt_capacity'].astype(np.int64),
            'tgt_colliding': f['tgt_colliding'].astype(bool), 'tgt_empty_bits': f['tgt_empty_bits'].astype(bool),
            'tgt_goal_bits': f['tgt_goal_bits'].astype(np.int64), 'tgt_goals': self.target_goals, 'freights': self.freights, 'bounties': self.bounties,
            'target_steps': self.target_steps, 'tracked_steps': self.tracked_steps, 'remaining_cargoes': self.remaining_cargoes,
            'awaiting_cargo_counts': self.awaiting_cargo_counts, 'num_delivered_cargoes': int(f['num_delivered_cargoes']),
            'target_warehouse_distances': self.target_warehouse_distances, 'target_dones': np.asarray(self.target_dones, dtype=bool),
            'coverage_rate': self.coverage_rate, 'real_coverage_rate': self.real_coverage_rate, 'mean_transport_rate': self.mean_transport_rate,
            'reward_dense': last[0], 'reward_delayed': last[1],
            'luts': [self.engine.lut_read(0, c) for c in range(Nc)],
        }
        for name in ('camera_target_view_mask', 'target_camera_view_mask', 'target_obstacle_view_mask', 'target_target_view_mask',
                     'camera_camera_view_mask', 'camera_obstacle_view_mask', 'tracked_bits'):
            snap[name] = self._mask(name).copy()
        return snap

    def joint_observation(self):
        self.engine.observe()
        cam, tgt, _ = self._collect()
        return cam, tgt

    def state(self):
        f = self._fields()
        parts = [self.preserved_data] + [c.state(private=True) for c in self.cameras] + [t.state(private=True) for t in self.targets] \
            + [o.state() for o in self.obstacles] + [f['freights'], f['bounties'], f['remaining_cargoes'].ravel()]
        return np.concatenate(parts).astype(np.float64)

    def load_config(self, config=None):
        seed = self.np_random.randint(np.iinfo(np.int32).max)
        self.engine.close()
        self.__init__(config=config, device=self._device_index, obs_dtype=self._obs_dtype)
        self.seed(seed)

    def render(self, mode='human', **kwargs):
        raise NotImplementedError('rendering is out of scope of the MI355X step engine (no display on the GPU box)')

    def add_render_callback(self, name, callback):
        self.render_callbacks[name] = callback

    def close(self):
        self.engine.close()

    def __str__(self):
        def plural(n, word):
            return f'{n} {word}{"s" if n > 1 else ""}'
        return f'{_EnvBase.__str__(self)}({plural(self.num_cameras, "camera")}, {plural(self.num_targets, "target")}, {plural(self.num_obstacles, "obstacle")})'

    # ------------------------------------------------------------------ intra-team messaging (host-side mailbox)
    def _clear_messages(self, totals):
        if totals:
            self.camera_total_communication_edges.fill(0)
            self.target_total_communication_edges.fill(0)
        else:
            self.camera_total_communication_edges += self.camera_communication_edges
            self.target_total_communication_edges += self.target_communication_edges
        self.camera_communication_edges.fill(0)
        self.target_communication_edges.fill(0)
        for store in (self.camera_message_buffer, self.target_message_buffer, self.camera_message_queue, self.target_message_queue):
            store.clear()

    def route_messages(self, messages):
        """Expand broadcasts into one message per teammate (environment.py:1249-1269)."""
        routed = []
        for message in messages:
            if message.recipient is None:
                for recipient in range((self.num_cameras, self.num_targets)[message.team.value]):
                    routed.append(Message(sender=message.sender, recipient=recipient, content=copy.deepcopy(message.content),
                                          team=message.team, broadcasting=True))
            else:
                routed.append(message)
        return routed

    def send_messages(self, messages):
        if isinstance(messages, Message):
            messages = (messages,)
        messages = list(messages)
        assert len({m.team for m in messages}) <= 1, f'All messages must be from the same team. Got messages = {messages}.'
        for message in self.route_messages(messages):
            team = message.team.value
            self.message_queues[team][message.recipient].append(message)
            self.message_buffers[team][message.recipient].append(message)
            self.communication_edges[team][message.sender, message.recipient] += 1

    def receive_messages(self, agent_id=None, agent=None):
        if agent_id is None and agent is None:
            out = ([list(self.camera_message_queue[c]) for c in range(self.num_cameras)],
                   [list(self.target_message_queue[t]) for t in range(self.num_targets)])
            self.camera_message_queue.clear()
            self.target_message_queue.clear()
            return out
        if agent_id is not None and not isinstance(agent_id, tuple) and agent is None:
            agent_id, agent = None, agent_id
        team, index = (agent.TEAM, agent.index) if agent is not None else agent_id
        out = list(self.message_queues[team.value][index])
        del self.message_queues[team.value][index]
        return out


class BatchedMultiAgentTracking(_ScenarioMixin):
    """N independent environments stepped by one kernel launch; every array is a torch tensor on the GPU.

        env = BatchedMultiAgentTracking('MATE-4v8-9.yaml', num_envs=4096)
        cam_obs, tgt_obs = env.reset()
        (cam_obs, tgt_obs), (r_cam, r_tgt), done, info = env.step((cam_act, tgt_act))

    `first_env_index` makes the RNG streams of a shard equal to those of the same environments in a
    larger single-GPU batch (sharding invariance)."""

    def __init__(self, config=None, num_envs=1, device=0, seed=0, first_env_index=0, obs_dtype=torch.float32, auto_reset=True,
                 relative_coordinates=False, rescaled_observation=False, enhanced_observation=None, shared_field_of_view=None,
                 discrete_camera_levels=None, discrete_target_levels=None, **kwargs):
        self._setup_scenario(config, kwargs)
        self.num_envs, self.auto_reset = int(num_envs), int(auto_reset)   # 0 / False: never; 1 / True: immediately; k > 1: batched, every k-th call
        self.engine = Engine(self.config, self.num_envs, device=device, seed=seed, first_env_index=first_env_index, obs_dtype=obs_dtype)
        self.num_cameras, self.num_targets, self.num_obstacles = self.engine.num_cameras, self.engine.num_targets, self.engine.num_obstacles
        self._setup_spaces()
        self.device = self.engine.device
        if relative_coordinates or rescaled_observation:   # the reference's RelativeCoordinates / RescaledObservation wrappers, fused
            self.engine.set_obs_transform(relative_coordinates, rescaled_observation)
        # EnhancedObservation(team=...) / SharedFieldOfView(team=...) of the reference: 'both', 'camera', 'target' or None
        modes = {'camera': 'plain', 'target': 'plain'}
        for name, team in (('shared', shared_field_of_view), ('enhanced', enhanced_observation)):
            if team in (None, False, 'none'):
                continue
            assert team in ('both', 'camera', 'target'), f'Invalid argument team {team!r}. Expect one of ("both", "camera", "target", "none").'
            for side in (('camera', 'target') if team == 'both' else (team,)):
                modes[side] = name
        if modes['camera'] != 'plain' or modes['target'] != 'plain':
            self.engine.set_obs_mode(**modes)
        # DiscreteCamera(levels) / DiscreteTarget(levels): integer action tensors are grid indices
        if discrete_camera_levels or discrete_target_levels:
            self.engine.set_action_grids(discrete_camera_levels, discrete_target_levels)

    def seed(self, seed):
        self.engine.seed(int(seed))
        return [int(seed)]

    def reset(self, env_mask=None):
        out = self.engine.reset(env_mask)
        for shaper in self.__dict__.get('_target_shapers', {}).values():
            shaper.observe_reset()
        return out

    def _result(self):
        s = self.engine.scalars
        info = {'coverage_rate': s[:, 3], 'real_coverage_rate': s[:, 4], 'mean_transport_rate': s[:, 5],
                'num_delivered_cargoes': s[:, 6], 'normalized_raw_reward': s[:, 7]}
        return (self.engine.camera_obs, self.engine.target_obs), (s[:, 0], s[:, 1]), s[:, 2] > 0, info

    def step(self, action):
        cam_act, tgt_act = action
        self.engine.step(cam_act, tgt_act, auto_reset=self.auto_reset)
        return self._result()

    def step_random(self):
        """One step under the on-device uniform random policy (no action tensors)."""
        self.engine.step_random(auto_reset=self.auto_reset)
        return self._result()

    def _rollout_result(self, out):
        cam, tgt, s = out
        info = {'coverage_rate': s[..., 3], 'real_coverage_rate': s[..., 4], 'mean_transport_rate': s[..., 5],
                'num_delivered_cargoes': s[..., 6], 'normalized_raw_reward': s[..., 7],
                'skipped': s[..., 2] == 2}          # slots after the end of an episode inside the launch (no step, stale rows)
        return (cam, tgt), (s[..., 0], s[..., 1]), s[..., 2] == 1, info

    def rollout_random(self, steps):
        """`steps` env.step(random action) iterations in ONE launch (the fastest flow, profiles/HISTORY.md 3.1b): every tensor of
        step()'s result with a leading [steps] axis.  Finished episodes restart after the launch."""
        return self._rollout_result(self.engine.rollout_random(steps, auto_reset=int(self.auto_reset)))

    def rollout_greedy(self, steps):
        """`steps` iterations of mate.group_step with the reference's Greedy camera / target agents + env.step in ONE
        launch (agents on the device, profiles/HISTORY.md 3.1c)."""
        if not getattr(self, '_policies_on', False):
            raise RuntimeError('enable_greedy_policies() must precede the reset() the agents first act on')
        return self._rollout_result(self.engine.rollout_greedy(steps, auto_reset=int(self.auto_reset)))

    def enable_greedy_policies(self):
        self.engine.enable_policies()
        self._policies_on = True

    def step_versus_greedy(self, team, joint_action):
        """MultiCamera(env, target_agent=GreedyTargetAgent()) for team = 'camera', MultiTarget(env,
        camera_agent=GreedyCameraAgent()) for team = 'target' (mate/wrappers/single_team.py:281-306): the caller acts for
        `team`, the greedy agents of the other team act on the device.  step()'s full result (both teams' observations
        and rewards; a single-team learner reads its own half)."""
        if not getattr(self, '_policies_on', False):
            raise RuntimeError('enable_greedy_policies() must precede the reset() the agents first act on')
        self.engine.step_versus_greedy(team, joint_action, auto_reset=self.auto_reset)
        return self._result()

    def rollout_versus_greedy(self, team, joint_action, frame_skip):
        """FrameSkip(MultiCamera | MultiTarget, frame_skip) in one launch (examples/utils/wrappers.py:301-323): rollout-shaped
        result; `rewards.sum(0)` is the wrapper's reward, `done.any(0)` its done."""
        if not getattr(self, '_policies_on', False):
            raise RuntimeError('enable_greedy_policies() must precede the reset() the agents first act on')
        return self._rollout_result(self.engine.rollout_versus_greedy(team, joint_action, frame_skip, auto_reset=int(self.auto_reset)))

    def masks(self):
        return self.engine.unpack_masks()

    AUXILIARY_REWARD_KEYS = ('raw_reward', 'coverage_rate', 'real_coverage_rate', 'mean_transport_rate', 'soft_coverage_score',
                             'num_tracked', 'baseline')

    def auxiliary_camera_rewards(self, coefficients, reduction='none'):
        """Per-camera shaped rewards of the last step, the reference's AuxiliaryCameraRewards wrapper
        (wrappers/auxiliary_camera_rewards.py:110-176) with constant coefficients: a weighted sum of the team reward,
        the coverage / transport metrics of the step record, the number of targets each camera tracks and a
        baseline; `reduction` in ('none', 'mean', 'sum', 'max', 'min') shares one value among the cameras.
        Returns a [num_envs, num_cameras] tensor on the GPU (a handful of elementwise ops on the [N, 8] step record
        and the packed masks: nothing here touches the observation bytes).  `soft_coverage_score` (:181-239) is one more
        small kernel over the outer occlusion boundary, which the engine builds from the first request on."""
        assert set(self.AUXILIARY_REWARD_KEYS).issuperset(coefficients.keys()), (
            f'The coefficient mapping only accepts keys in {self.AUXILIARY_REWARD_KEYS}. Got list(coefficients.keys()) = {list(coefficients.keys())}.')
        assert reduction in ('mean', 'sum', 'max', 'min', 'none'), f'Invalid reduction method {reduction}.'
        s = self.engine.scalars.double()
        N, Nc, Nt = self.num_envs, self.num_cameras, self.num_targets
        bits = torch.arange(Nc * Nt, device=self.device)
        words = self.engine.masks.long()[:, bits // 32]                       # camera_target_view_mask lives in the first words
        seen = ((words >> (bits % 32)) & 1).view(N, Nc, Nt)
        terms = {'raw_reward': s[:, 0:1].expand(N, Nc), 'coverage_rate': s[:, 3:4].expand(N, Nc),
                 'real_coverage_rate': s[:, 4:5].expand(N, Nc), 'mean_transport_rate': s[:, 5:6].expand(N, Nc),
                 'num_tracked': seen.sum(dim=2).double(), 'baseline': torch.ones((N, Nc), dtype=torch.float64, device=self.device)}
        if 'soft_coverage_score' in coefficients:
            if not getattr(self.engine, 'outer_capacity', 0):
                self.engine.enable_outer_boundary()     # built at every reset from now on; once now for the running episodes
                self.engine.rebuild_luts()
            terms['soft_coverage_score'] = self.engine.soft_coverage()[1]
        reward = torch.zeros((N, Nc), dtype=torch.float64, device=self.device)
        for key, coefficient in coefficients.items():
            assert isinstance(coefficient, (int, float)), 'only constant coefficients are supported on the batched path'
            reward = reward + float(coefficient) * terms[key]
        if reduction != 'none':
            shared = {'mean': reward.mean(dim=1), 'sum': reward.sum(dim=1), 'max': reward.max(dim=1).values, 'min': reward.min(dim=1).values}[reduction]
            reward = shared[:, None].expand(N, Nc)
        return reward

    def auxiliary_target_rewards(self, coefficients, reduction='none'):
        """Per-target shaped rewards of the last step, the reference's AuxiliaryTargetRewards wrapper
        (wrappers/auxiliary_target_rewards.py:118-216) with constant coefficients: [num_envs, num_targets] on the GPU.  Call
        it after every step (its `sparse_delivery` term compares the goals with those of the previous call; see
        mate_amd/auxiliary_rewards.py)."""
        key = (tuple(sorted(coefficients.items())), reduction)
        shapers = self.__dict__.setdefault('_target_shapers', {})
        if key not in shapers:
            from mate_amd.auxiliary_rewards import AuxiliaryTargetRewards
            shapers[key] = AuxiliaryTargetRewards(self.engine, coefficients, reduction)
        return shapers[key]()

    def state_dict(self):
        return self.engine.state_dict()

    def close(self):
        self.engine.close()
