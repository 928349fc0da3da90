"""The reference-side binding: the upstream class `mate.environment.MultiAgentTracking` with its step engine swapped.

The reference's wrappers assert `isinstance(env.unwrapped, mate.environment.MultiAgentTracking)`
(mate/wrappers/typing.py:59-66), read ~55 attributes of that object (masks, cargo arrays, entity
objects with `.location / .orientation / .is_colliding / .boundary_between(...)`, metrics, mailboxes) and call its
`reset / step / joint_observation / state / send_messages / receive_messages`.  A drop-in therefore has to BE that class.
`hip_environment_class(mate.environment)` derives the subclass a maintainer of the reference would add
(INTEGRATION.md section 2): construction, spaces, mailboxes, `state()`, the info dictionaries and the bookkeeping of
`step()` (environment.py:590-676) stay the reference's own code; only the three internals that ARE the hot path --
`_simulate` (:1326), `_assign_goals` (:1271), `joint_observation` (:908) -- and the episode boundary `reset` (:679) are
replaced by calls into a backend, after which the backend's state is mirrored into the reference's own attributes and
entity objects so that every wrapper sees what it expects.

The backend is duck-typed: `reset(seed=None)`, `step((camera_joint_action, target_joint_action))`, `snapshot()`
(see `mate_amd.environment.MultiAgentTracking.snapshot`).  The product backend is the HIP engine
(`mate_amd.MultiAgentTracking`, N = 1); the build container has no GPU, so tests/test_reference_adapter.py drives the
same subclass with a backend that replays a golden trace -- the reference's wrappers then run, unmodified, on top.

This module imports nothing from the reference: the caller passes the reference's `environment` module in.
"""
import numpy as np

__all__ = ['hip_environment_class']


def hip_environment_class(reference_environment):
    """`reference_environment` = the imported module `mate.environment`.  Returns the subclass."""
    Base = reference_environment.MultiAgentTracking

    class MultiAgentTrackingHIP(Base):
        """mate.environment.MultiAgentTracking stepping on the MI355X engine (or any backend with its protocol)."""

        def __init__(self, config=None, backend=None, **kwargs):
            self._backend = backend
            self._backend_kwargs = dict(kwargs)
            self._step_record = None
            super().__init__(config, **kwargs)           # the reference's own constructor (spaces, entity objects, mailboxes)

        # ---------------------------------------------------------------- backend
        def _engine(self):
            if self._backend is None:
                from mate_amd.environment import MultiAgentTracking as HipEnvironment
                self._backend = HipEnvironment(_plain_config(self.config))
            return self._backend

        def _mirror(self, snap):
            """Backend state -> the attributes and entity objects the reference's callers read."""
            Nc, Nt, No = self.num_cameras, self.num_targets, self.num_obstacles
            for c, camera in enumerate(self.cameras):
                camera.location = np.asarray(snap['cam_xy'][c], dtype=np.float64).copy()
                camera._orientation = float(snap['cam_phi'][c])          # already normalised (utils.py:155-158)
                camera.viewing_angle = float(snap['cam_theta'][c])
                camera.sight_range = float(snap['cam_sight'][c])
            for o, obstacle in enumerate(self.obstacles):
                obstacle.location = np.asarray(snap['obs_xyr'][o, :2], dtype=np.float64).copy()
                obstacle.radius = float(snap['obs_xyr'][o, 2])
            for t, target in enumerate(self.targets):
                target.location = np.asarray(snap['tgt_xy'][t], dtype=np.float64).copy()
                target.capacity = int(snap['tgt_capacity'][t])           # setter derives step_size (entities.py:612-615)
                target.goal_bits[:] = snap['tgt_goal_bits'][t]
                target.empty_bits[:] = snap['tgt_empty_bits'][t]
                target.is_colliding = bool(snap['tgt_colliding'][t])
            self.target_capacities[:] = snap['tgt_capacity']
            for name in ('camera_target_view_mask', 'target_camera_view_mask', 'target_obstacle_view_mask',
                         'target_target_view_mask', 'camera_camera_view_mask'):
                getattr(self, name)[...] = snap[name]                     # live views, mutated in place like the reference does
            self.tracked_bits = np.asarray(snap['tracked_bits'], dtype=bool).copy()
            self.target_goals[:] = snap['tgt_goals']
            self.target_goal_bits[...] = snap['tgt_goal_bits']
            self.freights[:] = snap['freights']
            self.bounties = np.asarray(snap['bounties'], dtype=np.int64).copy()
            self.remaining_cargoes[...] = snap['remaining_cargoes']
            self.awaiting_cargo_counts = np.asarray(snap['awaiting_cargo_counts'], dtype=np.int64).copy()
            self.num_delivered_cargoes = int(snap['num_delivered_cargoes'])
            self.target_warehouse_distances[...] = snap['target_warehouse_distances']
            self.target_dones = np.asarray(snap['target_dones'], dtype=bool).copy()
            # step() itself adds this step's increments after _assign_goals (environment.py:626-627): hand it the counters
            # as they stood before, so that they end the step at the engine's values (zeroed at a pick-up / delivery)
            self.target_steps[:] = np.asarray(snap['target_steps'], dtype=np.int64) - 1
            self.tracked_steps[:] = np.asarray(snap['tracked_steps'], dtype=np.int64) - self.tracked_bits.astype(np.int64)
            self._state = None

        def _mirror_episode(self, snap):
            """What only changes at reset(): obstacle blocks, the camera->obstacle mask, occlusion tables."""
            from scipy.interpolate import interp1d
            No = self.num_obstacles
            if No > 0:
                self.obstacle_states = np.asarray(snap['obs_xyr'], dtype=np.float64).reshape(No, 3).copy()
                self.obstacle_states_flagged = np.hstack([self.obstacle_states, np.ones((No, 1))])
                self.camera_obstacle_view_mask[...] = snap['camera_obstacle_view_mask']
                if self.num_cameras > 0:                                  # environment.py:757-764
                    self.camera_obstacle_observations = np.vstack([
                        np.where(self.camera_obstacle_view_mask[c, :, np.newaxis], self.obstacle_states_flagged, 0.0).ravel()
                        for c in range(self.num_cameras)])
            for c, table in enumerate(snap.get('luts') or []):            # Camera.sight_range_func (entities.py:476)
                self.cameras[c].sight_range_func = interp1d(np.asarray(table[0]), np.asarray(table[1]))
            for c, table in enumerate(snap.get('luts_outer') or []):
                self.cameras[c].sight_range_outer_func = interp1d(np.asarray(table[0]), np.asarray(table[1]))

        # ---------------------------------------------------------------- the swapped internals
        def reset(self, *, seed=None):                                    # environment.py:679-834
            self._destroy()
            self._step_record = None                                      # the previous episode's last step says nothing about the new one
            if seed is not None:
                self.seed(seed)
            self.cameras, self.targets, self.obstacles = list(self.cameras_ordered), list(self.targets_ordered), list(self.obstacles_ordered)
            self._observations = self._engine().reset(seed=seed)
            snap = self._backend.snapshot()
            self._mirror_episode(snap)
            self._mirror(snap)
            self._metrics(snap)
            self.target_steps.fill(0)
            self.tracked_steps.fill(0)
            self.target_dones = np.zeros(self.num_targets, dtype=bool)
            self.target_team_episode_reward = 0.0
            self.delayed_target_team_episode_reward = 0.0
            self.target_orientations.fill(0.0)
            for edges in (self.camera_total_communication_edges, self.target_total_communication_edges,
                          self.camera_communication_edges, self.target_communication_edges):
                edges.fill(0)
            for box in (self.camera_message_buffer, self.target_message_buffer, self.camera_message_queue, self.target_message_queue):
                box.clear()
            self.episode_step = 0
            return self.joint_observation()

        def _simulate(self, action):                                      # environment.py:1326-1354
            camera_joint_action, target_joint_action = action
            camera_joint_action = np.asarray(camera_joint_action, dtype=np.float64).reshape(self.num_cameras, 2)
            target_joint_action = np.asarray(target_joint_action, dtype=np.float64).reshape(self.num_targets, 2)
            assert np.isfinite(camera_joint_action).all(), f'Got unexpected joint action {camera_joint_action}.'
            assert np.isfinite(target_joint_action).all(), f'Got unexpected joint action {target_joint_action}.'
            self._observations, _, _, _ = self._engine().step((camera_joint_action, target_joint_action))
            self._step_record = self._backend.snapshot()
            self._mirror(self._step_record)

        def _update_view(self):                                           # the engine's step already did (environment.py:1356-1388)
            return None

        def _assign_goals(self):                                          # environment.py:1271-1324: the engine's step already did
            snap = self._step_record
            return float(snap['reward_dense']), float(snap['reward_delayed'])

        def _metrics(self, snap):
            self.coverage_rate = float(snap['coverage_rate'])
            self.real_coverage_rate = float(snap['real_coverage_rate'])
            self.mean_transport_rate = float(snap['mean_transport_rate'])

        def joint_observation(self):                                      # environment.py:908-983
            self._metrics(self._backend.snapshot() if self._step_record is None else self._step_record)
            camera_obs, target_obs = self._observations
            return np.array(camera_obs, dtype=np.float64), np.array(target_obs, dtype=np.float64)

        def close(self):
            if self._backend is not None and hasattr(self._backend, 'close'):
                self._backend.close()
            super().close()

    MultiAgentTrackingHIP.__qualname__ = 'MultiAgentTrackingHIP'
    return MultiAgentTrackingHIP


def _plain_config(config):
    """The reference keeps random ranges as gym Boxes after read_config (environment.py:162-193): back to plain lists."""
    import copy
    plain = {}
    for key, value in config.items():
        if isinstance(value, dict):
            sub = {}
            for k, v in value.items():
                if k == 'location_random_range':
                    sub[k] = [[float(b.low[0]), float(b.high[0]), float(b.low[1]), float(b.high[1])] if hasattr(b, 'low') else list(b) for b in v]
                elif k == 'radius_random_range':
                    sub[k] = [float(np.asarray(v.low).ravel()[0]), float(np.asarray(v.high).ravel()[0])] if hasattr(v, 'low') else list(v)
                elif k == 'location':
                    sub[k] = [[float(x) for x in loc] for loc in v]
                else:
                    sub[k] = copy.deepcopy(v)
            plain[key] = sub
        else:
            plain[key] = copy.deepcopy(value)
    return plain
