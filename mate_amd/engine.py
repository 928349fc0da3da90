"""`Engine`: the batched step engine as torch tensors over the C ABI.

PyTorch is plumbing here (device memory, streams); all simulation work happens
in the HIP kernels behind ``include/mate_engine.h``.
"""
import ctypes
import os

import numpy as np
import torch

from mate_amd import _native
from mate_amd._native import MateConfig, MateLayout, MatePolicyTape, MateStepIO, check

__all__ = ['Engine', 'EngineGroups', 'Stepper', 'export_layout', 'SCALAR_NAMES']

SCALAR_NAMES = ('camera_team_reward', 'target_team_reward', 'done', 'coverage_rate', 'real_coverage_rate',
                'mean_transport_rate', 'num_delivered_cargoes', 'normalized_target_team_reward')


def export_layout(Nc, Nt, No):
    """name -> (offset, shape) inside one row of `Engine.export_state()` ("state export")."""
    fields = [
        ('cam_x', (Nc,)), ('cam_y', (Nc,)), ('obs_x', (No,)), ('obs_y', (No,)), ('obs_radius', (No,)),
        ('tgt_capacity', (Nt,)), ('camera_obstacle_view_mask', (Nc, No)),
        ('cam_phi', (Nc,)), ('cam_theta', (Nc,)), ('tgt_x', (Nt,)), ('tgt_y', (Nt,)),
        ('tgt_colliding', (Nt,)), ('tgt_empty_bits', (Nt, 4)), ('tgt_goal_bits', (Nt, 4)), ('tgt_goals', (Nt,)),
        ('freights', (Nt,)), ('bounties', (Nt,)), ('target_steps', (Nt,)), ('tracked_steps', (Nt,)),
        ('remaining_cargoes', (4, 4)), ('awaiting_cargo_counts', (4,)), ('num_delivered_cargoes', ()),
        ('episode_reward', ()), ('delayed_episode_reward', ()), ('episode_step', ()), ('tick', ()), ('episode', ()),
        ('done', ()),
    ]
    layout, off = {}, 0
    for name, shape in fields:
        layout[name] = (off, shape)
        off += int(np.prod(shape)) if shape else 1
    return layout, off


class Engine:
    """N environments of one scenario on one GPU."""

    def __init__(self, config, num_envs, device=0, seed=0, first_env_index=0, obs_dtype=torch.float32):
        """`config` is a validated scenario mapping (mate_amd.config.read_config)."""
        self.lib = _native.load()
        self._block_switches = self._read_block_switches()
        self.config = config
        self.num_envs = int(num_envs)
        self.device_index = int(device)
        self.device = torch.device('cuda', self.device_index)
        self.obs_dtype = obs_dtype
        cam, tgt, obs = config.get('camera', {}), config['target'], config.get('obstacle', {})

        def ranges(sub):
            rows = [[x, x, y, y] for x, y in sub.get('location', [])] + [list(r) for r in sub.get('location_random_range', [])]
            return np.ascontiguousarray(np.asarray(rows, dtype=np.float64).reshape(-1, 4))

        self._ranges = (ranges(cam), ranges(tgt), ranges(obs))
        self.num_cameras, self.num_targets, self.num_obstacles = (len(r) for r in self._ranges)
        c = MateConfig()
        c.num_cameras, c.num_targets, c.num_obstacles = self.num_cameras, self.num_targets, self.num_obstacles
        c.max_episode_steps = int(config['max_episode_steps'])
        c.sparse_reward = int(config['reward_type'] == 'sparse')
        c.num_cargoes_per_target = int(config['num_cargoes_per_target'])
        c.shuffle_entities = int(bool(config['shuffle_entities']))
        c.targets_start_with_cargoes = int(bool(config['targets_start_with_cargoes']))
        c.high_capacity_target_split = float(config['high_capacity_target_split'])
        c.bounty_factor = float(config['bounty_factor'])
        c.transmittance = float(obs.get('transmittance', 0.0))
        c.camera_radius = float(cam.get('radius', 40.0))
        c.camera_min_viewing_angle = float(cam.get('min_viewing_angle', 90.0))
        c.camera_max_sight_range = float(cam.get('max_sight_range', 500.0))
        c.camera_rotation_step = float(cam.get('rotation_step', 5.0))
        c.camera_zooming_step = float(cam.get('zooming_step', 2.5))
        c.target_step_size = float(tgt['step_size'])
        c.target_sight_range = float(tgt['sight_range'])
        if 'radius_random_range' in obs:
            rr = list(obs['radius_random_range'])
        else:
            rr = [float(obs.get('radius', 0.0))] * 2
        c.obstacle_radius_range[0], c.obstacle_radius_range[1] = float(rr[0]), float(rr[1])
        dp = ctypes.POINTER(ctypes.c_double)
        c.camera_location_ranges = self._ranges[0].ctypes.data_as(dp)
        c.target_location_ranges = self._ranges[1].ctypes.data_as(dp)
        c.obstacle_location_ranges = self._ranges[2].ctypes.data_as(dp)
        c.obs_dtype = 1 if obs_dtype == torch.float64 else 0
        self._cfg = c
        handle = ctypes.c_void_p()
        check(self.lib.mate_engine_create(ctypes.byref(c), self.num_envs, self.device_index, int(seed), int(first_env_index), ctypes.byref(handle)))
        self._h = handle
        layout = MateLayout()
        check(self.lib.mate_engine_get_layout(self._h, ctypes.byref(layout)))
        self.layout = layout
        self.camera_obs_dim, self.target_obs_dim = layout.camera_obs_dim, layout.target_obs_dim
        self.specialised = bool(layout.specialised)   # shape-specialised step kernels in use (MATE_GENERIC=1 forces generic)
        self.export_fields, width = export_layout(self.num_cameras, self.num_targets, self.num_obstacles)
        assert width == layout.export_width, (width, layout.export_width)
        N, Nc, Nt = self.num_envs, self.num_cameras, self.num_targets
        with torch.cuda.device(self.device):
            self.camera_obs = torch.zeros((N, Nc, layout.camera_obs_dim), dtype=obs_dtype, device=self.device)
            self.target_obs = torch.zeros((N, Nt, layout.target_obs_dim), dtype=obs_dtype, device=self.device)
            self.scalars = torch.zeros((N, 8), dtype=torch.float32, device=self.device)
            self.masks = torch.zeros((N, layout.mask_words), dtype=torch.int32, device=self.device)
            # running sums over finished episodes (mate_amd.distributed.EpisodeStats.FIELDS): count, return, length, coverage, delivered
            self.episode_stats = torch.zeros(5, dtype=torch.float64, device=self.device)
        check(self.lib.mate_engine_set_episode_stats(self._h, ctypes.c_void_p(self.episode_stats.data_ptr())))

    def close(self):
        if getattr(self, '_h', None):
            self.lib.mate_engine_destroy(self._h)
            self._h = None

    __del__ = close

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _io(self, cam_act=None, tgt_act=None, tape_ct=None, tape_goal=None, want_masks=True):
        io = MateStepIO()
        keep = []
        if tgt_act is not None or cam_act is not None:
            # integer tensors are grid indices (set_action_grids): int32 [N, agents]; reals are [N, agents, 2].
            # One team alone = the caller's team of step_versus_greedy.
            ints = (torch.int32, torch.int64, torch.int16, torch.uint8)
            if self.num_cameras == 0:
                cam_act = None
            teams = [(name, act, agents) for name, act, agents in (('camera', cam_act, self.num_cameras), ('target', tgt_act, self.num_targets))
                     if act is not None]
            reals = [act.dtype for _, act, _ in teams if act.dtype not in ints]
            act_dtype = reals[0] if reals else torch.float64
            assert act_dtype in (torch.float32, torch.float64)
            io.act_dtype = 1 if act_dtype == torch.float64 else 0
            for name, act, agents in teams:
                if act.dtype in ints:
                    assert getattr(self, name + '_action_grid', None) is not None, f'call set_action_grids({name}_levels=...) first'
                    act = act.to(torch.int32).contiguous()
                    assert act.numel() == self.num_envs * agents and act.device == self.device
                    io.act_dtype |= 0x100 if name == 'camera' else 0x200
                else:
                    act = act.to(act_dtype).contiguous()
                    assert act.numel() == self.num_envs * agents * 2 and act.device == self.device
                setattr(io, name + '_actions_dev', act.data_ptr())
                keep.append(act)
        if tape_ct is not None and self.num_cameras:
            tape_ct = tape_ct.to(torch.float64).contiguous()
            assert tape_ct.numel() == self.num_envs * self.num_cameras * self.num_targets
            io.tape_camera_target_dev = tape_ct.data_ptr()
            keep.append(tape_ct)
        if tape_goal is not None:
            tape_goal = tape_goal.to(torch.float64).contiguous()
            assert tape_goal.numel() == self.num_envs * self.num_targets
            io.tape_goal_dev = tape_goal.data_ptr()
            keep.append(tape_goal)
        io.camera_obs_dev = self.camera_obs.data_ptr() if self.num_cameras else None
        io.target_obs_dev = self.target_obs.data_ptr()
        io.scalars_dev = self.scalars.data_ptr()
        io.masks_dev = self.masks.data_ptr() if want_masks else None
        return io, keep

    # ---------------------------------------------------------------------- API
    def set_obs_transform(self, relative_coordinates=False, rescaled_observation=False):
        """Fuse RelativeCoordinates / RescaledObservation (the reference's observation wrappers) into the
        kernel's packer.  The affine map of the rescale comes from the observation-space bounds
        (mate_amd.constants), exactly as `rescale_observation` derives it."""
        from mate_amd import constants as consts

        def affine(space):
            low, high = np.asarray(space.low, dtype=np.float64), np.asarray(space.high, dtype=np.float64)
            scale, bias = np.ones_like(low), np.zeros_like(low)
            below = np.isfinite(low)
            both = below & np.isfinite(high) & (high > low)
            bias[below] = -low[below]                      # rescaled[bounded_below] -= low
            span = np.where(both, high - low, 1.0)
            scale[both] = 2.0 / span[both]                 # rescaled[mask] = 2 * rescaled / (high - low) - 1
            bias[both] = -2.0 * low[both] / span[both] - 1.0
            return np.ascontiguousarray(scale), np.ascontiguousarray(bias)

        nums = (self.num_cameras, self.num_targets, self.num_obstacles)
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        if rescaled_observation:
            cs, cb = affine(consts.camera_observation_space_of(*nums))
            ts, tb = affine(consts.target_observation_space_of(*nums))
            check(self.lib.mate_engine_set_obs_transform(self._h, int(relative_coordinates), ptr(cs), ptr(cb), ptr(ts), ptr(tb)))
        else:
            check(self.lib.mate_engine_set_obs_transform(self._h, int(relative_coordinates), None, None, None, None))

    OBS_MODES = {None: 0, False: 0, 'none': 0, 'plain': 0, 'enhanced': 1, 'shared': 2}

    def set_obs_mode(self, camera='plain', target='plain'):
        """Per-team observation mode fused into the packer: 'plain', 'enhanced' (the reference's
        EnhancedObservation wrapper) or 'shared' (SharedFieldOfView)."""
        check(self.lib.mate_engine_set_obs_mode(self._h, self.OBS_MODES[camera], self.OBS_MODES[target]))

    def set_action_grids(self, camera_levels=None, target_levels=None):
        """Discrete joint actions (the reference's DiscreteCamera / DiscreteTarget wrappers,
        wrappers/discrete_action_spaces.py): actions passed as int32 indices into `levels**2` grids are decoded
        in the step kernel.  The normalised grids are computed here with the reference's own NumPy formulas
        and handed to the engine as tables."""
        from mate_amd.spaces import camera_action_grid, target_action_grid
        cg = np.ascontiguousarray(camera_action_grid(camera_levels), dtype=np.float64) if camera_levels else None
        tg = np.ascontiguousarray(target_action_grid(target_levels), dtype=np.float64) if target_levels else None
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None  # noqa: E731
        check(self.lib.mate_engine_set_action_grids(self._h, ptr(cg), 0 if cg is None else len(cg), ptr(tg), 0 if tg is None else len(tg)))
        self.camera_action_grid, self.target_action_grid = cg, tg

    def seed(self, seed):
        check(self.lib.mate_engine_seed(self._h, int(seed)))

    def reset(self, env_mask=None):
        io, keep = self._io()
        mask_ptr = None
        if env_mask is not None:
            env_mask = env_mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mask_ptr = ctypes.c_void_p(env_mask.data_ptr())
        check(self.lib.mate_engine_reset(self._h, mask_ptr, ctypes.byref(io), self._stream()))
        return self.camera_obs, self.target_obs

    def reset_tape(self, tape, tape_ct=None, env_mask=None):
        """reset() with the random draws taken from `tape` ([N, L] uniforms in the reference's call order, see
        mate_engine_reset_tape) and the see-through draws of the first view from `tape_ct` ([N, Nc, Nt]).  Returns
        (camera_obs, target_obs, draws_used [N] int32)."""
        tape = tape.to(device=self.device, dtype=torch.float64).contiguous()
        assert tape.dim() == 2 and tape.shape[0] == self.num_envs
        io, keep = self._io(tape_ct=tape_ct)
        used = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        mask_ptr = None
        if env_mask is not None:
            env_mask = env_mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mask_ptr = ctypes.c_void_p(env_mask.data_ptr())
        check(self.lib.mate_engine_reset_tape(self._h, mask_ptr, ctypes.byref(io), ctypes.c_void_p(tape.data_ptr()), int(tape.shape[1]),
                                              ctypes.c_void_p(used.data_ptr()), self._stream()))
        return self.camera_obs, self.target_obs, used

    def step(self, cam_act, tgt_act, tape_ct=None, tape_goal=None, auto_reset=False):
        io, keep = self._io(cam_act, tgt_act, tape_ct, tape_goal)
        check(self.lib.mate_engine_step(self._h, ctypes.byref(io), int(auto_reset), self._stream()))
        return self.camera_obs, self.target_obs, self.scalars

    def step_random(self, auto_reset=True, want_masks=False):
        # the argument structure of this call never changes: built once per (masks or not), the host cost per step
        # matters when the step kernel is a dozen microseconds
        cache = self.__dict__.setdefault('_random_io', {})
        ref = cache.get(want_masks)
        if ref is None:
            io, _ = self._io(want_masks=want_masks)
            ref = cache[want_masks] = (io, ctypes.byref(io))
        status = self.lib.mate_engine_step_random(self._h, ref[1], int(auto_reset), self._stream())
        if status != 0:
            check(status)
        return self.camera_obs, self.target_obs, self.scalars

    RESET_PIPELINED = -1     # MATE_RESET_PIPELINED (include/mate_engine.h)

    @classmethod
    def _auto_reset_code(cls, auto_reset):
        """'pipelined' -> MATE_RESET_PIPELINED; ('pipelined', m) -> -m: one restart launch (on the engine's side stream) behind every
        m-th rollout launch; everything else as it is."""
        if auto_reset == 'pipelined':
            return cls.RESET_PIPELINED
        if isinstance(auto_reset, tuple):
            if len(auto_reset) != 2 or auto_reset[0] != 'pipelined' or not 1 <= int(auto_reset[1]) <= 1 << 16:
                raise ValueError(f"auto_reset = {auto_reset!r}: ('pipelined', m) takes 1 <= m <= 65536")
            return -int(auto_reset[1])
        if int(auto_reset) < 0 and int(auto_reset) != cls.RESET_PIPELINED:      # (a typo such as -5 would silently defer restarts five launches)
            raise ValueError(f"auto_reset = {auto_reset!r}: negative values other than Engine.RESET_PIPELINED are spelt ('pipelined', m)")
        return int(auto_reset)

    def rollout_greedy(self, steps, auto_reset=True, want_masks=False):
        """`steps` fused (agents act, environment steps) iterations of the on-device Greedy policies (enable_policies()
        first).  Same rollout-shaped tensors as rollout_random.  auto_reset = 'pipelined' (Engine.RESET_PIPELINED): the reset of
        what a launch finishes runs on the engine's side stream under the next launch, restarted environments join the one after."""
        auto_reset = self._auto_reset_code(auto_reset)
        return self._run_rollout(self.lib.mate_engine_rollout_greedy, steps, auto_reset, want_masks)

    def rollout_versus_greedy(self, team, joint_action, steps, auto_reset=True, want_masks=False):
        """FrameSkip(frame_skip=steps) over MultiCamera / MultiTarget (examples/utils/wrappers.py:301-323 over
        mate/wrappers/single_team.py:245-264) in ONE launch: the caller's `team` repeats `joint_action` for `steps` frames while
        the on-device greedy opponents act anew on every frame.  Rollout-shaped tensors; the caller sums the reward rows."""
        team = {'camera': 0, 'target': 1}.get(team, team)
        assert team in (0, 1)
        auto_reset = self._auto_reset_code(auto_reset)
        steps = int(steps)
        buf = self.reserve_rollout(steps, want_masks)
        io, keep = self._io(cam_act=joint_action if team == 0 else None, tgt_act=joint_action if team == 1 else None)
        io.camera_obs_dev = buf['camera_obs'].data_ptr() if self.num_cameras else None
        io.target_obs_dev = buf['target_obs'].data_ptr()
        io.scalars_dev = buf['scalars'].data_ptr()
        io.masks_dev = buf['masks'].data_ptr() if want_masks else None
        check(self.lib.mate_engine_rollout_versus_greedy(self._h, team, ctypes.byref(io), steps, int(auto_reset), self._stream()))
        return buf['camera_obs'][:steps], buf['target_obs'][:steps], buf['scalars'][:steps]

    def rollout_random(self, steps, auto_reset=True, want_masks=False):
        """`steps` fused steps under the on-device random policy.  Returns rollout-shaped tensors
        (camera_obs [T,N,Nc,Dc], target_obs [T,N,Nt,Dt], scalars [T,N,8]); scalars[..., 2] == 2 marks
        steps skipped because the episode had already ended inside this rollout."""
        return self._run_rollout(self.lib.mate_engine_rollout_random, steps, auto_reset, want_masks)

    # Host-side switches of the observation-block search (read ONCE, when the Engine is built; include/mate_engine.h lists them)
    @staticmethod
    def _read_block_switches():
        env = os.environ
        return {
            'plain': env.get('MATE_PLAIN_BLOCKS') == '1',
            # (0 or less: no search at all -- one block, unprobed)
            'candidates': max(1, int(env['MATE_BLOCK_CANDIDATES'])) if 'MATE_BLOCK_CANDIDATES' in env else None,
            # the deep search (candidates separated by 12 GB spacers, a transient footprint of up to 45 % of the free memory) is
            # OPT-IN: reserve_rollout(search='deep') -- bench.py asks for it -- or MATE_BLOCK_DEEP=1; by default a few candidates side
            # by side within 0.3 s
            'deep': env.get('MATE_BLOCK_DEEP', '0') != '0',
            'seconds': float(env['MATE_BLOCK_SECONDS']) if 'MATE_BLOCK_SECONDS' in env else None,
            'deep_gib': float(env.get('MATE_BLOCK_GIB', '96')),
            'store_form': env.get('MATE_STORE_FORM', 'auto'),
        }

    def _observation_block(self, shape, deep=False):
        """A zeroed [steps][N][...] observation block and the store rates [GB/s] of the candidates probed for it.  Blocks of
        64 MiB and more come from ``mate_engine_block_alloc`` (2 MiB physical chunks in a shuffled order): the fused rollouts
        store 10-25 % faster into them than into what hipMalloc / the caching allocator hands out (include/mate_engine.h).
        MATE_PLAIN_BLOCKS=1: torch.zeros."""
        sw = self._block_switches
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=self.obs_dtype).element_size()
        if nbytes < (64 << 20) or sw['plain']:
            return torch.zeros(shape, dtype=self.obs_dtype, device=self.device), []
        # Shuffled chunks make a slow block unlikely, not impossible (tools/store_vmm.hip): blocks of 128 MiB and more -- the ones
        # a launch is bounded by -- are the fastest of up to MATE_BLOCK_CANDIDATES (default 6) candidates in the kernels' own
        # store pattern, where the device has the memory to hold them side by side; the search ends at the first candidate
        # that is a class (28 %) faster than another.
        # `deep` (the target block, which decides a launch's store rate): the fast blocks of a device lie in ZONES of its memory, 20-35 GB
        # wide and at a different depth on every GPU (tools/depth_probe.py: forty 4.4 GB blocks allocated in a row and held -- five to
        # twelve consecutive ones fast, the rest slow).  The deep search walks through the memory in allocation order -- a
        # candidate, a 12 GB spacer that is not probed, a candidate ... -- until one is fast (5.35 TB/s, or 28 % above the
        # slowest seen: the classes lie at ~4.2, ~5.0 and 5.4-6.1).  Its transient footprint is bounded three ways: 45 % of what is
        # free (two ranks that share a GPU in a test must both fit), MATE_BLOCK_GIB (default 96) GiB in absolute terms, and
        # MATE_BLOCK_SECONDS (default 3) of wall time -- a learner whose model already holds most of the HBM gets a short search,
        # not an out-of-memory error.  The wall-time bound covers BOTH blocks of a reserve_rollout (the camera block's candidates
        # take what the target block's search left, at least one).  Everything but the winner is freed at the end
        # (mate_engine_block_free: the physical memory comes back, the address range stays reserved).
        row_bytes = nbytes // (shape[0] * shape[1])
        search = getattr(self, '_block_search', None) or ('deep' if sw['deep'] else 'shallow')
        tries = (sw['candidates'] or (6 if search == 'deep' else 3)) if nbytes >= (128 << 20) and row_bytes % 16 == 0 and search != 'none' else 1
        free = torch.cuda.mem_get_info(self.device)[0]
        deep = deep and tries > 1 and search == 'deep'
        seconds = sw['seconds'] if sw['seconds'] is not None else (3.0 if search == 'deep' else 0.3)
        spacer_bytes = 12 << 30
        budget = min(0.45 * free, sw['deep_gib'] * (1 << 30))
        if deep and sw['candidates'] is None:      # (an explicit count bounds the deep search too)
            tries = max(tries, int(budget // (nbytes + spacer_bytes)))
        tries = max(1, min(tries, int(free // (2 * nbytes))))
        import time
        t0 = getattr(self, '_reserve_t0', None) or time.perf_counter()      # (reserve_rollout's start: one time budget for both blocks)
        best, rates, held, spacers = None, [], [], []
        for _ in range(tries):
            try:
                block = _native.ScatteredBlock(self.device_index, nbytes)
            except _native.EngineError as err:      # no virtual-memory management on this driver, or out of memory: plain memory works as well
                if best is None:
                    import warnings
                    warnings.warn(f'mate_engine_block_alloc failed ({err}); the rollout block comes from torch.zeros')
                    return torch.zeros(shape, dtype=self.obs_dtype, device=self.device), rates
                break
            if tries == 1:
                best = (0.0, block)
                break
            rate = block.store_rate(shape[1], row_bytes, self._stream())
            rates.append(rate)
            held.append(block)          # (kept until the search ends: a freed candidate would be handed out again)
            if best is None or rate > best[0]:
                best = (rate, block)
            if len(rates) > 1 and best[0] >= 1.28 * min(rates):
                break
            if time.perf_counter() - t0 > seconds:
                break
            if deep:
                if rate >= (5350.0 if nbytes >= (1 << 30) else 5100.0):      # (a short block's probe is a short launch: its ramp weighs more)
                    break
                try:
                    spacers.append(_native.HeldMemory(self.device_index, spacer_bytes))      # (moves the allocation on; never mapped, never touched)
                except _native.EngineError:
                    break
        block = best[1]
        del held, best
        del spacers
        return block.tensor(self.obs_dtype, shape).zero_(), rates

    def reserve_rollout(self, steps, want_masks=False, search=None):
        """Allocate the rollout-shaped output buffers ([steps][N][...]) now, so that a later rollout of up to `steps`
        steps allocates nothing (a training loop or a timed region calls this once up front).  Rows a launch does not
        write -- the observation rows of an environment that had finished earlier in the launch -- keep whatever they
        held; its scalar rows say done = 2.  `Engine.reserve_seconds` says how long the last (re)allocation took, candidate
        search included; `Engine.block_rates` = [target block's candidates, camera block's candidates] in GB/s.
        `search`: where the observation blocks come from -- 'shallow' (default: the fastest of up to three candidates allocated
        side by side, at most 0.3 s), 'deep' (the walk through the device's memory described in _observation_block: seconds,
        and a transient footprint of up to 45 % of the free HBM -- for a process that owns the GPU, e.g. bench.py), 'none' (one
        block, unprobed).  Every candidate that loses keeps its address range reserved (mate_engine_block_free), bounded per
        process by MATE_BLOCK_DEAD_GIB."""
        steps = int(steps)
        if search not in (None, 'shallow', 'deep', 'none'):
            raise ValueError(f"reserve_rollout(search={search!r}): 'shallow', 'deep', 'none' or None (= the last explicit choice)")
        if search is not None:                       # an explicit choice stays for the (re)allocations that follow, the internal ones included
            self._block_search = search
        buf = getattr(self, '_rollout', None)
        depth = {'none': 0, 'shallow': 1, 'deep': 2}
        if buf is not None and search is not None and depth[search] > depth.get(buf.get('search'), 0) and buf['steps'] >= steps:
            import warnings                          # (buffers that exist are kept: say that the deeper search was not run)
            warnings.warn(f"reserve_rollout(search={search!r}): the rollout buffers exist (searched {buf.get('search')!r}); release them "
                          '(Engine.close() or a longer reservation) to search again', RuntimeWarning, stacklevel=2)
        if buf is None or buf['steps'] < steps or (want_masks and buf['masks'] is None):     # a shorter rollout fills a prefix
            import time
            t0 = self._reserve_t0 = time.perf_counter()
            N, Nc, Nt, L = self.num_envs, self.num_cameras, self.num_targets, self.layout
            self._rollout = None
            with torch.cuda.device(self.device):
                target_block, target_rates = self._observation_block((steps, N, Nt, L.target_obs_dim), deep=True)      # (first: its search holds the most memory)
                camera_block, camera_rates = self._observation_block((steps, N, Nc, L.camera_obs_dim))
                buf = {
                    'steps': steps, 'search': getattr(self, '_block_search', None) or ('deep' if self._block_switches['deep'] else 'shallow'),
                    'camera_obs': camera_block,
                    'target_obs': target_block,
                    'scalars': torch.zeros((steps, N, 8), dtype=torch.float32, device=self.device),
                    'masks': torch.zeros((steps, N, L.mask_words), dtype=torch.int32, device=self.device) if want_masks else None,
                }
            self._rollout = buf
            self.block_rates = [target_rates, camera_rates]
            # where even the best candidate takes the rows slowly the stores bound a launch, and the line-aligned form of the row
            # stores wins 3 %; elsewhere it costs 1.3-2 % (include/mate_engine.h).  MATE_STORE_FORM=0 / 1 forces a form.
            form = self._block_switches['store_form']
            shifted = form == '1' or (form == 'auto' and bool(target_rates) and max(target_rates) < 4800.0)
            self.store_form = int(shifted)
            check(self.lib.mate_engine_set_store_form(self._h, int(shifted)))
            torch.cuda.synchronize(self.device)
            self.reserve_seconds = time.perf_counter() - t0
            self._reserve_t0 = None
        return buf

    def _run_rollout(self, entry_point, steps, auto_reset, want_masks):
        steps = int(steps)
        buf = self.reserve_rollout(steps, want_masks)
        # the argument block and the returned views of a (launch length, masks) combination never change while the buffers
        # live: built once -- a 20-step launch lasts 0.18 ms, and this call is what the GPU waits for before it starts
        cache = buf.setdefault('_calls', {})
        call = cache.get((steps, bool(want_masks)))
        if call is None:
            io = MateStepIO()
            io.camera_obs_dev = buf['camera_obs'].data_ptr() if self.num_cameras else None
            io.target_obs_dev = buf['target_obs'].data_ptr()
            io.scalars_dev = buf['scalars'].data_ptr()
            io.masks_dev = buf['masks'].data_ptr() if want_masks else None
            call = cache[(steps, bool(want_masks))] = (io, ctypes.byref(io), (buf['camera_obs'][:steps], buf['target_obs'][:steps], buf['scalars'][:steps]))
        status = entry_point(self._h, call[1], steps, int(auto_reset), self._stream())
        if status != 0:
            check(status)
        return call[2]

    def device_tick(self, enable=True):
        """Keep the step counter on the device (mate_engine_device_tick): step()/step_random() launches with
        auto_reset = `enable` (True = 1: immediate; k > 1: batched, one reset launch per k steps) then carry identical
        arguments in every reset interval and can be captured in a HIP graph.  False / 0 gives the counter back."""
        check(self.lib.mate_engine_device_tick(self._h, int(enable), self._stream()))

    def make_stepper(self, cam_act, tgt_act, auto_reset=True, graph_steps=0, between=None, versus=None, frame_skip=1):
        """A replayable `for _ in range(n): between(); step((cam_act, tgt_act))` loop over caller-owned action tensors
        (see Stepper); versus = 'camera' / 'target': the caller plays that team only, the greedy agents the other;
        frame_skip = K > 1 on top of `versus`: every step is one K-frame launch (FrameSkip, the example trainers' flow).
        With graph_steps > 0 the constructor runs `auto_reset` REAL steps before it captures (Stepper.warmup_steps)."""
        return Stepper(self, cam_act, tgt_act, auto_reset, graph_steps, between, versus, frame_skip)

    def enable_policies(self):
        """Allocate the on-device policy state (call before the reset whose observations the agents act on)."""
        check(self.lib.mate_engine_policy_enable(self._h))

    def _policy_tape(self, policy_tape, keep):
        if policy_tape is None:
            return None
        tape = MatePolicyTape()
        for name, dtype in (('camera_resample_u', torch.float64), ('camera_sample_u', torch.float64), ('camera_delay', torch.int32),
                            ('target_choice_u', torch.float64), ('target_resample_u', torch.float64), ('target_sample_u', torch.float64),
                            ('target_reset_sample_u', torch.float64)):
            value = policy_tape.get(name)
            if value is not None:
                value = value.to(device=self.device, dtype=dtype).contiguous()
                keep.append(value)
                setattr(tape, name + '_dev', value.data_ptr())
        return ctypes.byref(tape)

    def step_greedy(self, policy_tape=None, tape_ct=None, tape_goal=None, auto_reset=True):
        """One step with GreedyCameraAgent vs GreedyTargetAgent computed on the device.  `policy_tape`:
        dict of recorded agent draws (tensors) for parity runs.  Returns (camera_obs, target_obs, scalars)."""
        io, keep = self._io(tape_ct=tape_ct, tape_goal=tape_goal)
        check(self.lib.mate_engine_step_greedy(self._h, ctypes.byref(io), self._policy_tape(policy_tape, keep), int(auto_reset), self._stream()))
        return self.camera_obs, self.target_obs, self.scalars

    def step_versus_greedy(self, team, joint_action, policy_tape=None, tape_ct=None, tape_goal=None, auto_reset=True):
        """MultiCamera (team = 'camera' / 0) or MultiTarget (team = 'target' / 1), mate/wrappers/single_team.py:245-264:
        `joint_action` is the caller's team ([N, agents, 2] reals or [N, agents] grid indices), the opponents are the
        on-device greedy agents.  Returns (camera_obs, target_obs, scalars)."""
        team = {'camera': 0, 'target': 1}.get(team, team)
        assert team in (0, 1)
        io, keep = self._io(cam_act=joint_action if team == 0 else None, tgt_act=joint_action if team == 1 else None,
                            tape_ct=tape_ct, tape_goal=tape_goal)
        check(self.lib.mate_engine_step_versus_greedy(self._h, team, ctypes.byref(io), self._policy_tape(policy_tape, keep),
                                                      int(auto_reset), self._stream()))
        return self.camera_obs, self.target_obs, self.scalars

    def policy_actions(self):
        """(camera_actions [N,Nc,2], target_actions [N,Nt,2]) f64: the joint actions of the last step_greedy."""
        N, Nc, Nt = self.num_envs, self.num_cameras, self.num_targets
        cam = torch.zeros((N, Nc, 2), dtype=torch.float64, device=self.device)
        tgt = torch.zeros((N, Nt, 2), dtype=torch.float64, device=self.device)
        check(self.lib.mate_engine_policy_actions(self._h, ctypes.c_void_p(cam.data_ptr()) if Nc else None,
                                                  ctypes.c_void_p(tgt.data_ptr()), self._stream()))
        return cam, tgt

    def observe(self, tape_ct=None):
        io, keep = self._io(tape_ct=tape_ct)
        check(self.lib.mate_engine_observe(self._h, ctypes.byref(io), self._stream()))
        return self.camera_obs, self.target_obs

    def export_state(self, out=None):
        if out is None:
            out = torch.empty((self.num_envs, self.layout.export_width), dtype=torch.float64, device=self.device)
        check(self.lib.mate_engine_export_state(self._h, ctypes.c_void_p(out.data_ptr()), self._stream()))
        return out

    # One copy per step for the N = 1 NumPy API (mate_amd.environment): the output tensors and an export_state buffer become views of
    # ONE device allocation, and fetch_host() brings a step's results over in a single transfer (five blocking copies of a few KB each
    # were a third of that API's 250 us per step).
    def stage_outputs(self):
        """Re-home camera_obs / target_obs / scalars / masks (and a state-export buffer) in one flat device buffer.  Call before
        the first reset; the tensors keep their names, shapes and dtypes."""
        parts = [('camera_obs', self.camera_obs), ('target_obs', self.target_obs), ('scalars', self.scalars), ('masks', self.masks),
                 ('state', torch.empty((self.num_envs, self.layout.export_width), dtype=torch.float64, device=self.device))]
        offsets, total = [], 0
        for _, t in parts:
            offsets.append(total)
            total += -(-t.numel() * t.element_size() // 256) * 256
        flat = torch.zeros(total, dtype=torch.uint8, device=self.device)
        self._staged = {'flat': flat, 'views': {}}
        for (name, t), off in zip(parts, offsets):
            nbytes = t.numel() * t.element_size()
            view = flat[off:off + nbytes].view(t.dtype).view(t.shape)
            self._staged['views'][name] = (off, nbytes, t.dtype, tuple(t.shape))
            if name == 'state':
                self._staged['state'] = view
            else:
                setattr(self, name, view)
        self.__dict__.pop('_random_io', None)

    _NP = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32}

    def fetch_host(self):
        """(after stage_outputs) the current state exported + ONE device-to-host copy: dict of NumPy arrays camera_obs, target_obs,
        scalars, masks, state -- views of one host buffer, valid until the next call."""
        self.export_state(out=self._staged['state'])
        pinned = self._staged.get('pinned')
        if pinned is None:
            pinned = self._staged['pinned'] = torch.empty(self._staged['flat'].shape, dtype=torch.uint8, pin_memory=True)
        pinned.copy_(self._staged['flat'], non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        host = pinned.numpy().copy()          # (the caller keeps views of it; the pinned buffer is rewritten by the next call)
        return {name: host[off:off + nbytes].view(self._NP[dtype]).reshape(shape)
                for name, (off, nbytes, dtype, shape) in self._staged['views'].items()}

    def state_dict_from(self, flat):
        """state_dict() of an already fetched [N, export_width] f64 array."""
        if getattr(self, '_export_slices', None) is None:
            self._export_slices = [(name, off, int(np.prod(shape)) if shape else 1, (self.num_envs,) + tuple(shape))
                                   for name, (off, shape) in self.export_fields.items()]
        return {name: flat[:, off:off + n].reshape(shape) for name, off, n, shape in self._export_slices}

    def import_state(self, flat):
        flat = flat.to(device=self.device, dtype=torch.float64).contiguous()
        assert flat.shape == (self.num_envs, self.layout.export_width)
        check(self.lib.mate_engine_import_state(self._h, ctypes.c_void_p(flat.data_ptr()), self._stream()))
        torch.cuda.synchronize(self.device)

    def state_dict(self):
        """All state fields as numpy arrays [N, ...] (host copy)."""
        return self.state_dict_from(self.export_state().cpu().numpy())

    def load_state_dict(self, fields):
        flat = self.export_state().cpu().numpy()
        for name, value in fields.items():
            off, shape = self.export_fields[name]
            n = int(np.prod(shape)) if shape else 1
            flat[:, off:off + n] = np.asarray(value, dtype=np.float64).reshape(self.num_envs, n)
        self.import_state(torch.from_numpy(flat))

    def rebuild_luts(self):
        check(self.lib.mate_engine_rebuild_luts(self._h, self._stream()))

    def enable_outer_boundary(self):
        """Also build Camera.boundary_outer at every later reset / rebuild_luts (entities.py:419-448); needed only by
        boundary_between(outer=True)."""
        cap = ctypes.c_int32()
        check(self.lib.mate_engine_enable_outer_boundary(self._h, ctypes.byref(cap)))
        self.outer_capacity = cap.value

    def lut_read(self, env, camera, outer=False):
        cap = self.outer_capacity if outer else self.layout.lut_capacity
        phis, rhos = np.zeros(cap), np.zeros(cap)
        n = ctypes.c_int32()
        fn = self.lib.mate_engine_lut_read_outer if outer else self.lib.mate_engine_lut_read
        check(fn(self._h, int(env), int(camera), phis.ctypes.data_as(ctypes.c_void_p), rhos.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(n)))
        return phis[:n.value].copy(), rhos[:n.value].copy()

    def lut_write(self, env, camera, phis, rhos, outer=False):
        phis = np.ascontiguousarray(phis, dtype=np.float64)
        rhos = np.ascontiguousarray(rhos, dtype=np.float64)
        fn = self.lib.mate_engine_lut_write_outer if outer else self.lib.mate_engine_lut_write
        check(fn(self._h, int(env), int(camera), phis.ctypes.data_as(ctypes.c_void_p), rhos.ctypes.data_as(ctypes.c_void_p), len(phis)))

    def soft_coverage(self, masks=None):
        """AuxiliaryCameraRewards' soft coverage score of the current state (wrappers/auxiliary_camera_rewards.py:128-139,
        181-239): (score matrix [N, Nc, Nt], per-camera scores [N, Nc]) f64 on the GPU.  `masks` = packed masks of the
        step this follows (default: the engine's own output buffer).  Needs enable_outer_boundary() + a reset /
        rebuild_luts since."""
        masks = self.masks if masks is None else masks
        assert masks.dtype == torch.int32 and masks.is_contiguous() and masks.shape == (self.num_envs, self.layout.mask_words)
        if getattr(self, '_softcov', None) is None:
            self._softcov = (torch.empty((self.num_envs, self.num_cameras, self.num_targets), dtype=torch.float64, device=self.device),
                             torch.empty((self.num_envs, self.num_cameras), dtype=torch.float64, device=self.device))
        matrix, scores = self._softcov
        check(self.lib.mate_engine_soft_coverage(self._h, masks.data_ptr(), matrix.data_ptr(), scores.data_ptr(), self._stream()))
        return matrix, scores

    def idle_steps(self):
        """(environment, step) slots spent idle waiting for a batched reset since creation."""
        total = ctypes.c_int64()
        check(self.lib.mate_engine_idle_steps(self._h, ctypes.byref(total)))
        return total.value

    def snapshot_episode_stats(self, out):
        """`out` (5 f64 on the device) <- the episode-statistics accumulators, ordered on the current stream behind the launches enqueued
        so far (mate_engine_snapshot_episode_stats: one tiny launch)."""
        check(self.lib.mate_engine_snapshot_episode_stats(self._h, ctypes.c_void_p(out.data_ptr()), self._stream()))
        return out

    def set_sub_wave(self, enable='auto'):
        """Environments per wave of the fused rollouts (mate_engine_set_sub_wave): the small scenarios (at most four cameras and four
        targets) can step four environments per wave.  'auto' (the default of a new engine) = where that measured faster (batches of at
        least 32 environments per compute unit; under the random policy every such shape but MATE-4v4-*), True = in every fused
        launch of such a shape, False = one per wave.  Returns the number the Greedy rollouts now run with."""
        in_use = ctypes.c_int32()
        check(self.lib.mate_engine_set_sub_wave(self._h, 2 if enable == 'auto' else int(bool(enable)), ctypes.byref(in_use)))
        return in_use.value

    @property
    def sub_wave(self):
        """Environments per wave the fused Greedy rollouts run with (1, or 4 for the small scenarios)."""
        in_use = ctypes.c_int32()
        check(self.lib.mate_engine_set_sub_wave(self._h, -1, ctypes.byref(in_use)))
        return in_use.value

    def kernel_time(self, enable=1):
        """(avg ms, launches) of the step kernel since the last call; `enable` = k arms the HIP-event timer
        for every k-th launch (0 disarms)."""
        avg, n = ctypes.c_double(), ctypes.c_int64()
        check(self.lib.mate_engine_kernel_time(self._h, int(enable), ctypes.byref(avg), ctypes.byref(n)))
        return avg.value, n.value

    @property
    def last_flow(self):
        """Compilation of the step kernel the last launch ran: 0 generic, 1 random-policy flow, 2 f32-actions flow."""
        return int(self.lib.mate_engine_last_flow(self._h))

    # decoded masks ------------------------------------------------------------
    def unpack_masks(self, masks=None, words_host=None):
        """Packed u32 words -> dict of boolean numpy arrays [N, ...] (environment.py:475-494 names).  `words_host`: the words as a
        NumPy array already on the host (fetch_host)."""
        words = (words_host if words_host is not None else (self.masks if masks is None else masks).cpu().numpy()).astype(np.uint32)
        N, Nc, Nt, No = self.num_envs, self.num_cameras, self.num_targets, self.num_obstacles
        bits = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool).reshape(N, -1)
        L = self.layout
        NJ = Nc + No + Nt
        ct = bits[:, L.bit_camera_target:L.bit_camera_target + Nc * Nt].reshape(N, Nc, Nt)
        cc = bits[:, L.bit_camera_camera:L.bit_camera_camera + Nc * Nc].reshape(N, Nc, Nc)
        rows = bits[:, L.bit_target_row:L.bit_target_row + Nt * NJ].reshape(N, Nt, NJ)
        co = np.zeros((N, Nc, No), dtype=bool)
        for c in range(Nc):
            co[:, c] = bits[:, L.bit_camera_obstacle + 64 * c:L.bit_camera_obstacle + 64 * c + No]
        return {
            'camera_target_view_mask': ct, 'camera_camera_view_mask': cc,
            'target_camera_view_mask': rows[:, :, :Nc], 'target_obstacle_view_mask': rows[:, :, Nc:Nc + No],
            'target_target_view_mask': rows[:, :, Nc + No:], 'camera_obstacle_view_mask': co,
            'tracked_bits': ct.any(axis=1),
        }


class Stepper:
    """The learner-in-the-loop stepping flow: `between()` (the caller's policy: any torch code that rewrites the joint
    action tensors in place) then `step((cam_act, tgt_act))`, with outputs in the engine's own observation / scalar /
    mask tensors.  With `graph_steps` = K > 0 the step counter moves to the device (Engine.device_tick) and K
    iterations are captured once in a HIP graph (torch.cuda.CUDAGraph: the policy's kernels and the engine's launches
    in one graph); `run(n)` replays it n // K times and launches the remainder directly.  The host then spends one
    graph launch per K steps instead of two kernel launches through ctypes per step.  Bit-identical to calling
    Engine.step in a loop (tested).  `close()` gives the step counter back to the host; an open reset interval (run()
    stopped between two reset launches) is closed there: what had finished in it restarts at once.

    NOTE -- building a graph Stepper ADVANCES the environments: before the capture, one whole reset interval
    (`auto_reset` real steps: `between()`, the greedy opponents of `versus`, the step) runs directly, so that every code
    object is loaded and the device-resident counter sits on an interval boundary.  Those `warmup_steps` transitions are
    real (outputs in the engine's tensors, episode statistics counted) and a learner that must see every transition
    reads them like any other step; "bit-identical to Engine.step in a loop" means a loop that includes them."""

    def __init__(self, eng, cam_act, tgt_act, auto_reset=True, graph_steps=0, between=None, versus=None, frame_skip=1):
        self.eng, self.between, self.graph_steps = eng, between, int(graph_steps)
        # frame_skip = K > 1 (with `versus`): FrameSkip(K) over MultiCamera / MultiTarget, the example trainers' flow -- every "step" of
        # this stepper is ONE K-frame launch (Engine.rollout_versus_greedy: the caller's action repeated, the greedy opponents acting
        # anew on every frame), `auto_reset` and `graph_steps` count launches, and run() returns the rollout-shaped tensors
        # ([K, N, ...]: the caller sums the reward rows, reads the last frame's observation) instead of the engine's per-step ones
        self.frame_skip = int(frame_skip)
        assert self.frame_skip >= 1 and (self.frame_skip == 1 or versus is not None), 'frame_skip belongs to the learner-versus-greedy flow'
        self.auto_reset = int(auto_reset)        # True / 1: immediate; k > 1: batched (finished environments idle up to k - 1 steps)
        # versus = 'camera' / 'target': the caller's team (MultiCamera / MultiTarget); the other team is played by the on-device
        # greedy agents (Engine.step_versus_greedy) and its tensor argument is ignored
        self.versus = {'camera': 0, 'target': 1, None: None}.get(versus, versus)
        assert self.versus in (None, 0, 1)
        if self.versus == 0:
            tgt_act = None
        elif self.versus == 1:
            cam_act = None
        self.io, self.keep = eng._io(cam_act, tgt_act)
        # _io may have made contiguous copies: the stepper must read the caller's own storage
        assert (tgt_act is None or self.io.target_actions_dev == tgt_act.data_ptr()) and \
            (cam_act is None or eng.num_cameras == 0 or self.io.camera_actions_dev == cam_act.data_ptr()), \
            'action tensors must be contiguous f32/f64 (or int32 grid indices) on the engine device'
        self.outputs = None
        if self.frame_skip > 1:
            buf = eng.reserve_rollout(self.frame_skip)
            self.io.camera_obs_dev = buf['camera_obs'].data_ptr() if eng.num_cameras else None
            self.io.target_obs_dev = buf['target_obs'].data_ptr()
            self.io.scalars_dev = buf['scalars'].data_ptr()
            self.io.masks_dev = None
            self.outputs = (buf['camera_obs'][:self.frame_skip], buf['target_obs'][:self.frame_skip], buf['scalars'][:self.frame_skip])
            self.keep = (self.keep, buf)
        self.ref = ctypes.byref(self.io)
        self.graph = None
        self._phase = 0                                   # steps into the current reset interval (graphs hold whole intervals)
        self.warmup_steps = self.auto_reset if self.graph_steps > 0 else 0      # real steps the constructor runs (see the class note)
        if self.graph_steps > 0:
            assert self.auto_reset >= 1 and self.graph_steps % self.auto_reset == 0, \
                'graph replay needs auto_reset >= 1 (the auto-reset launch advances the device step counter) and whole reset intervals per graph'
            eng.device_tick(self.auto_reset)
            for _ in range(self.auto_reset):
                self._one()                               # code objects loaded before the capture (one whole reset interval)
            torch.cuda.synchronize(eng.device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                for _ in range(self.graph_steps):
                    self._one()

    def _one(self):
        if self.between is not None:
            self.between()
        eng = self.eng
        if self.versus is None:
            status = eng.lib.mate_engine_step(eng._h, self.ref, self.auto_reset, eng._stream())
        elif self.frame_skip > 1:
            status = eng.lib.mate_engine_rollout_versus_greedy(eng._h, self.versus, self.ref, self.frame_skip, self.auto_reset, eng._stream())
        else:
            status = eng.lib.mate_engine_step_versus_greedy(eng._h, self.versus, self.ref, None, self.auto_reset, eng._stream())
        if status != 0:
            check(status)

    def run(self, steps):
        steps = int(steps)
        if self.graph is not None:
            while steps > 0 and self._phase != 0:         # finish a reset interval a previous call stopped in, launch by launch
                self._one()
                self._phase = (self._phase + 1) % self.auto_reset
                steps -= 1
            while steps >= self.graph_steps:              # whole graphs (whole reset intervals)
                self.graph.replay()
                steps -= self.graph_steps
            self._phase = steps % self.auto_reset
        for _ in range(steps):
            self._one()
        if self.outputs is not None:
            return self.outputs
        eng = self.eng
        return eng.camera_obs, eng.target_obs, eng.scalars

    def close(self):
        if self.graph is not None:
            torch.cuda.synchronize(self.eng.device)
            self.graph = None
            self._phase = 0
            self.eng.device_tick(False)                   # (closes an open reset interval: mate_engine_device_tick)

    def __del__(self):
        try:
            self.close()
        except Exception as exc:      # an engine left in device-tick mode refuses rollouts / seed / import_state: say so
            import warnings
            warnings.warn(f'Stepper.close() failed while the stepper was collected: {exc!r}; the engine may still count steps on the device', RuntimeWarning)


class EngineGroups:
    """One batch of N environments as G groups of N / G on G streams, for a learner that interleaves its groups (double-buffered
    sampling: while it computes the actions of one group, the other group steps).

    A one-launch-per-step flow leaves the GPU idle at every kernel boundary -- the tail of step t's slowest waves, the ramp of step
    t + 1, the caller's policy kernel in between, and the two dependency gaps around it: at 4096 x MATE-4v8-9 5 of a step's 14.5 us.
    Two independent half-batches on two streams fill each other's gaps: 14.5 -> 13.2 us per step of the whole batch at 4096
    environments, 33.7 -> 27.6 at 16 384 (+20 %; 0.456 -> 0.55 of the HBM roofline), 53.6 -> 43.8 for the learner-versus-greedy step;
    three groups do no better, four worse (profiles/r05_groups_probe.txt).  The groups ARE the batch: group g holds the global
    environment indices [first_env_index + g N / G, ...), so every environment's episode is, bit for bit, what the single engine of N
    steps (the random streams are keyed by the global index; tests/test_gpu_groups.py).

    `streams[0]` is the stream current at construction, the others come from `pick_streams()`: HIP maps streams onto a handful of
    hardware queues, and two streams that share a queue run their work one after the other (23 instead of 13 us per step when
    that happens) -- a short trial of a few fresh streams with the caller's own loop body picks ones that overlap."""

    def __init__(self, config, num_envs, groups=2, device=0, seed=0, first_env_index=0, obs_dtype=torch.float32, policies=False):
        assert groups >= 1 and num_envs % groups == 0, 'the groups share the batch evenly'
        self.groups, self.per_group = int(groups), int(num_envs) // int(groups)
        self.device = torch.device('cuda', int(device))
        self.streams = [torch.cuda.current_stream(self.device)] + [torch.cuda.Stream(device=self.device) for _ in range(self.groups - 1)]
        self.engines = []
        for g in range(self.groups):
            with torch.cuda.stream(self.streams[g]):
                eng = Engine(config, self.per_group, device=device, seed=seed, first_env_index=first_env_index + g * self.per_group, obs_dtype=obs_dtype)
                if policies:
                    eng.enable_policies()
                self.engines.append(eng)

    def each(self, fn):
        """fn(group index, engine) for every group, with the group's stream current; returns the results."""
        out = []
        for g, eng in enumerate(self.engines):
            with torch.cuda.stream(self.streams[g]):
                out.append(fn(g, eng))
        return out

    def reset(self):
        return self.each(lambda g, eng: eng.reset())

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def pick_streams(self, body, candidates=3, warm=2, timed=4):
        """Choose the side streams by trial.  `body(g, engine)` enqueues one slice of the caller's loop for group g (e.g. a Stepper
        replay); for every side stream, `candidates` fresh streams are tried with `warm + timed` rounds of all groups and the
        fastest is kept.  Returns the trial times [s] of the last group's candidates.

        The trial STEPS the environments (`candidates * (warm + timed)` slices per group): call it before the episodes that matter,
        or reset afterwards.  A group's engine moves to a stream that knows nothing of the work queued on its previous one (a
        reset, an import), so the device is synchronised before the first trial and at every change of stream; the engine's buffers
        were allocated on the first stream and live as long as the engine, so the caching allocator never hands them out again."""
        import time
        times = []
        torch.cuda.synchronize(self.device)           # everything queued on the groups' present streams is complete before one of them changes
        for g in range(1, self.groups):
            times = []
            pool = [torch.cuda.Stream(device=self.device) for _ in range(candidates)]
            for cand in pool:
                torch.cuda.synchronize(self.device)
                self.streams[g] = cand
                for _ in range(warm):
                    self.each(body)
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                for _ in range(timed):
                    self.each(body)
                torch.cuda.synchronize(self.device)
                times.append(time.perf_counter() - t0)
            torch.cuda.synchronize(self.device)
            self.streams[g] = pool[times.index(min(times))]
        return times

    def idle_steps(self):
        return sum(eng.idle_steps() for eng in self.engines)

    def close(self):
        for eng in self.engines:
            eng.close()
