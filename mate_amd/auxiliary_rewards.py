"""Per-target shaped rewards on the batch: the reference's AuxiliaryTargetRewards wrapper
(mate/wrappers/auxiliary_target_rewards.py:26-216) with constant coefficients, over N environments at once.

The wrapper sits on the caller's side of step() (examples/*/target/config.py apply it right after MultiTarget): a weighted
sum, per target, of the team reward, the coverage / transport metrics of the step record, and five per-target terms read
off the environment's state after the step -- the normalised distance to the destination (or to the nearest non-empty
warehouse), whether the target just delivered, the soft coverage score against every camera's outline, whether it is
tracked, whether it is colliding.  Everything here is a handful of elementwise torch ops on the engine's own device
tensors ([N, 8] step record, packed masks, one state export, the soft-coverage matrix kernel); nothing touches the
observation bytes.

`sparse_delivery` is `target_dones` (environment.py:1320-1322: the goal changed and there was one), which needs the
goals BEFORE the step: the shaper keeps the goals it saw at its previous call, so call it once after every step (the
wrapper's own calling pattern).  It notices restarted environments by their episode counter: the call that follows
an auto-reset describes the NEW episode's first state (as every output of an auto-resetting step does) with
sparse_delivery = 0; pass auto_reset = 0 / k > 1 (finished environments keep their final state until the batched
restart) where the terminal step's shaped reward matters.
"""
import numpy as np
import torch

from mate_amd import constants as consts

__all__ = ['AuxiliaryTargetRewards']


class AuxiliaryTargetRewards:
    ACCEPTABLE_KEYS = ('raw_reward', 'coverage_rate', 'real_coverage_rate', 'mean_transport_rate', 'normalized_goal_distance',
                       'sparse_delivery', 'soft_coverage_score', 'is_tracked', 'is_colliding', 'baseline')

    def __init__(self, engine, coefficients, reduction='none'):
        assert reduction in ('mean', 'sum', 'max', 'none'), (          # auxiliary_target_rewards.py:84-88 (no 'min' there)
            f'Invalid reduction method {reduction}. The reduction method should be one of ("mean", "sum", "max") (for shared reward), '
            f'or "none" for no reduction (for individual reward).')
        assert set(self.ACCEPTABLE_KEYS).issuperset(coefficients.keys()), (
            f'The coefficient mapping only accepts keys in {self.ACCEPTABLE_KEYS}. Got list(coefficients.keys()) = {list(coefficients.keys())}.')
        for key, coefficient in coefficients.items():
            assert isinstance(coefficient, (int, float)), f'only constant coefficients are supported on the batched path (got {key!r}: {coefficient!r})'
        self.engine, self.coefficients, self.reduction = engine, {k: float(v) for k, v in coefficients.items()}, reduction
        if 'soft_coverage_score' in self.coefficients:
            assert engine.num_cameras > 0, 'soft_coverage_score needs cameras (the reference takes a max over them)'
            if not getattr(engine, 'outer_capacity', 0):
                engine.enable_outer_boundary()       # built at every reset from now on; once now for the running episodes
                engine.rebuild_luts()
        self._warehouses = torch.as_tensor(consts.WAREHOUSES, dtype=torch.float64, device=engine.device)
        self.terms = None
        self.observe_reset()

    def _field(self, flat, name):
        off, shape = self.engine.export_fields[name]
        n = int(np.prod(shape)) if shape else 1
        return flat[:, off:off + n].reshape((flat.shape[0],) + tuple(shape))

    def observe_reset(self):
        """Remember the goals of the current state (call after an explicit reset(); construction does)."""
        flat = self.engine.export_state()
        self._goals = self._field(flat, 'tgt_goals').clone()
        self._episode = self._field(flat, 'episode').clone()

    def __call__(self, masks=None):
        """Shaped rewards [N, Nt] f64 of the step that just ran.  `masks` = that step's packed masks (default: the
        engine's own buffer).  `self.terms` keeps every term ([N, Nt]) for the info dictionaries."""
        eng = self.engine
        N, Nc, Nt = eng.num_envs, eng.num_cameras, eng.num_targets
        dev = eng.device
        s = eng.scalars.double()
        flat = eng.export_state()
        goals = self._field(flat, 'tgt_goals')
        episode = self._field(flat, 'episode')
        same_episode = (episode == self._episode)[:, None]
        delivered = (goals != self._goals) & (self._goals >= 0) & same_episode           # environment.py:1320-1322
        self._goals, self._episode = goals.clone(), episode.clone()

        # :131-143 -- distance to the goal warehouse's rim, else to the nearest non-empty one, else half the terrain
        tx, ty = self._field(flat, 'tgt_x'), self._field(flat, 'tgt_y')
        dx = tx[:, :, None] - self._warehouses[None, None, :, 0]
        dy = ty[:, :, None] - self._warehouses[None, None, :, 1]
        rim = torch.clamp(torch.sqrt(dx * dx + dy * dy) - consts.WAREHOUSE_RADIUS, min=0.0)
        nonempty = self._field(flat, 'tgt_empty_bits') == 0
        to_goal = torch.gather(rim, 2, goals.clamp(min=0).long()[:, :, None])[:, :, 0]
        nearest = torch.where(nonempty, rim, torch.full_like(rim, float('inf'))).min(dim=2).values
        goal_distance = torch.where(goals >= 0, to_goal,
                                    torch.where(nonempty.any(dim=2), nearest, torch.full_like(nearest, consts.TERRAIN_WIDTH / 2.0)))

        words = (eng.masks if masks is None else masks).long()
        if Nc:
            bits = torch.arange(Nc * Nt, device=dev)
            seen = (((words[:, bits // 32] >> (bits % 32)) & 1) != 0).view(N, Nc, Nt)       # camera_target_view_mask: the first words
        else:
            seen = torch.zeros((N, 0, Nt), dtype=torch.bool, device=dev)
        tracked = seen.any(dim=1)

        def shared(column):
            return s[:, column:column + 1].expand(N, Nt)

        terms = {'raw_reward': shared(1), 'coverage_rate': shared(3), 'real_coverage_rate': shared(4), 'mean_transport_rate': shared(5),
                 'normalized_goal_distance': goal_distance / consts.TERRAIN_WIDTH, 'sparse_delivery': delivered.double(),
                 'is_tracked': tracked.double(), 'is_colliding': (self._field(flat, 'tgt_colliding') != 0).double(),
                 'baseline': torch.ones((N, Nt), dtype=torch.float64, device=dev)}
        if 'soft_coverage_score' in self.coefficients:          # :146-158: sum over the cameras that see it, else tanh(max)
            matrix = eng.soft_coverage(masks)[0]
            summed = torch.where(seen, matrix, torch.zeros_like(matrix)).sum(dim=1)
            terms['soft_coverage_score'] = torch.where(tracked, summed, torch.tanh(matrix.max(dim=1).values))
        reward = torch.zeros((N, Nt), dtype=torch.float64, device=dev)
        for key, coefficient in self.coefficients.items():      # same summation order as the wrapper (:175-181)
            reward = reward + coefficient * terms[key]
        if self.reduction != 'none':
            one = {'mean': reward.mean(dim=1), 'sum': reward.sum(dim=1), 'max': reward.max(dim=1).values}[self.reduction]
            reward = one[:, None].expand(N, Nt)
        self.terms = terms
        return reward
