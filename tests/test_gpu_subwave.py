"""GPU (-m gpu): FOUR ENVIRONMENTS PER WAVE (engine_kernels.hpp, Ctx<ObsT, L>; include/mate_engine.h, mate_engine_set_sub_wave).

The engine maps one environment onto one 64-lane wave.  The scenarios of the reference's target trainers and its other small scenarios
(at most four cameras and four targets: MATE-2v4-0 of examples/*/target/config.py, MATE-4v2-9 of BASELINE config 1, ...) fill a quarter
of a wave, so their fused rollouts run sub-wave groups of sixteen lanes, one environment each.  Everything here holds the two mappings
to the same bits: rows, scalars, masks, the records, the agents' memory, across episode ends and restarts, at batch sizes that are no
multiple of sixteen, in every fused flow (random policy, Greedy vs Greedy, a learner's team against the greedy opponents with frame skip,
pipelined restarts, the fused observation transforms of the generic flow)."""
import numpy as np
import pytest
import torch

from mate_amd.config import read_config
from mate_amd.engine import Engine

pytestmark = pytest.mark.gpu

SMALL = ['MATE-1v1-0', 'MATE-1v1-9', 'MATE-1v2-0', 'MATE-1v2-9', 'MATE-2v2-0', 'MATE-2v2-9', 'MATE-2v4-0', 'MATE-2v4-9', 'MATE-4v2-0', 'MATE-4v2-9',
         'MATE-4v4-0', 'MATE-4v4-9']


def same(a, b):
    return torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8))


def _pair(name, n, **kw):
    cfg = read_config(name + '.yaml', **kw)
    engines = []
    per_wave = 2 if name.startswith('MATE-4v8') else 4
    for on in (True, False):
        eng = Engine(cfg, n, seed=41, first_env_index=3)
        assert eng.set_sub_wave(on) == (per_wave if on else 1) and eng.sub_wave == (per_wave if on else 1)
        eng.enable_policies()
        eng.reset()
        engines.append(eng)
    return engines


@pytest.mark.parametrize('name', SMALL + ['MATE-4v8-0'])      # (MATE-4v8-0: two per wave, thirty-two lanes each)
def test_four_environments_per_wave_equal_one_per_wave(name):
    n = 53                                                   # (no multiple of 16: the last wave's groups past the batch idle)
    sub, one = _pair(name, n, max_episode_steps=11)
    assert same(sub.export_state(), one.export_state())
    for steps, auto_reset in ((5, True), (9, True), (4, 2), (4, 2), (7, False)):
        out = []
        for eng in (sub, one):
            cam, tgt, sc = eng.rollout_random(steps, auto_reset=auto_reset, want_masks=True)
            out.append((cam.clone(), tgt.clone(), sc.clone(), eng._rollout['masks'][:steps].clone(), eng.export_state().clone()))
        for x, y in zip(*out):
            assert same(x, y), (name, 'random', steps, auto_reset)
    for eng in (sub, one):
        eng.reset()
    for steps, auto_reset in ((6, True), (8, True), (3, 2), (3, 2), (5, 'pipelined'), (5, 'pipelined'), (5, 'pipelined'), (4, True)):
        out = []
        for eng in (sub, one):
            cam, tgt, sc = eng.rollout_greedy(steps, auto_reset=auto_reset, want_masks=True)
            torch.cuda.synchronize()
            out.append((cam.clone(), tgt.clone(), sc.clone(), eng._rollout['masks'][:steps].clone()))
        for x, y in zip(*out):
            assert same(x, y), (name, 'greedy', steps, auto_reset)
    for eng in (sub, one):
        eng.rollout_greedy(1, auto_reset=True)               # (leaves the pipelined mode: restarts what it finished)
    assert same(sub.export_state(), one.export_state()) and same(sub.policy_actions()[1], one.policy_actions()[1])
    assert sub.idle_steps() == one.idle_steps() and same(sub.episode_stats, one.episode_stats)
    assert float(sub.episode_stats[0]) >= n                  # episodes ended (time limit 11) and restarted inside all this


@pytest.mark.parametrize('name,team,frames', [('MATE-2v4-0', 'target', 10), ('MATE-2v4-0', 'camera', 5), ('MATE-4v2-9', 'target', 10), ('MATE-4v4-9', 'camera', 5),
                                              ('MATE-1v1-9', 'target', 3), ('MATE-2v2-0', 'camera', 4), ('MATE-4v8-0', 'camera', 5)])
def test_frame_skip_against_the_greedy_opponents(name, team, frames):
    """FrameSkip(K) over MultiTarget(GreedyCameraAgent) / MultiCamera(GreedyTargetAgent) (examples/utils/wrappers.py:301-323 over
    mate/wrappers/single_team.py:245-306; the target trainers run MATE-2v4-0 with K = 10): one launch per learner action, f32 / f64 /
    discrete joint actions of the learner's team, restarts every other launch."""
    n = 40
    sub, one = _pair(name, n, max_episode_steps=23)
    gen = torch.Generator(device='cuda').manual_seed(9)
    k = sub.num_targets if team == 'target' else sub.num_cameras
    scale = torch.tensor([20.0, 20.0] if team == 'target' else [5.0, 2.5], device='cuda')
    for it in range(9):
        act = (torch.rand((n, k, 2), device='cuda', generator=gen) * 2 - 1) * scale * 1.2
        if it % 3 == 1:
            act = act.double()
        out = []
        for eng in (sub, one):
            cam, tgt, sc = eng.rollout_versus_greedy(team, act, frames, auto_reset=2, want_masks=True)
            out.append((cam.clone(), tgt.clone(), sc.clone(), eng._rollout['masks'][:frames].clone(), eng.export_state().clone(),
                        eng.policy_actions()[0 if team == 'target' else 1].clone()))
        for x, y in zip(*out):
            assert same(x, y), (name, team, it)
    assert float(sub.episode_stats[0]) > 0 and same(sub.episode_stats, one.episode_stats)


def test_generic_flow_with_fused_transforms():
    """rollout_random under the FLOW_ANY compilation (a fused RelativeCoordinates + RescaledObservation transform and the
    EnhancedObservation team mode switch the folded flow off): the element-wise packer and the team-wide flags in sixteen-lane groups."""
    name, n = 'MATE-4v2-9', 37
    sub, one = _pair(name, n, max_episode_steps=9)
    out = []
    for eng in (sub, one):
        eng.set_obs_transform(relative_coordinates=True, rescaled_observation=True)
        eng.set_obs_mode(camera='enhanced', target='shared')
        rec = []
        for steps in (6, 7):
            cam, tgt, sc = eng.rollout_random(steps, auto_reset=True, want_masks=True)
            rec += [cam.clone(), tgt.clone(), sc.clone(), eng._rollout['masks'][:steps].clone()]
        assert eng.last_flow == 0
        out.append(rec)
    for x, y in zip(*out):
        assert same(x, y)


def test_sub_wave_is_the_default_only_where_it_is_compiled_and_measured_faster():
    for name, want in (('MATE-2v4-0', 4), ('MATE-4v2-9', 4), ('MATE-4v8-0', 2), ('MATE-4v8-9', 1), ('MATE-8v8-9', 1), ('MATE-Navigation', 1)):
        eng = Engine(read_config(name + '.yaml'), 8, seed=1)
        assert eng.sub_wave == 1                                  # 'auto' at a batch of 8: one per wave everywhere
        assert eng.set_sub_wave(True) == want and eng.set_sub_wave(False) == 1 and eng.set_sub_wave('auto') == 1
    big = Engine(read_config('MATE-2v4-0.yaml'), 16384, seed=1)
    assert big.sub_wave == 4 and big.set_sub_wave(False) == 1 and big.set_sub_wave('auto') == 4


@pytest.mark.parametrize('team,name,skip,interval', [('target', 'MATE-2v4-0', 10, 2), ('camera', 'MATE-2v4-0', 5, 3)])
def test_graph_replayed_frame_skip_with_four_environments_per_wave(team, name, skip, interval):
    """The target trainers' loop as bench.py and a learner run it -- FrameSkip(K) over MultiTarget(GreedyCameraAgent), one K-frame launch
    per learner action, `graph_steps` of them replayed from ONE HIP graph with the step counter on the device -- on the four-per-wave
    kernels: bit for bit the same launches made one by one with one environment per wave and the host counting, across episode ends."""
    cfg = read_config(name + '.yaml', max_episode_steps=23)
    n = 70
    outs = []
    for sub, graph_steps in ((False, 0), (True, 2 * interval)):
        eng = Engine(cfg, n, seed=5)
        assert eng.set_sub_wave(sub) == (4 if sub else 1)
        eng.enable_policies()
        eng.reset()
        agents = eng.num_cameras if team == 'camera' else eng.num_targets
        gen = torch.Generator(device='cuda').manual_seed(9)
        mine = (torch.rand((n, agents, 2), device='cuda', generator=gen) * 2 - 1) * (6 if team == 'camera' else 25)
        stepper = eng.make_stepper(mine if team == 'camera' else None, mine if team == 'target' else None, auto_reset=interval,
                                   graph_steps=graph_steps, between=lambda m=mine: m.mul_(-1.0).add_(0.125), versus=team, frame_skip=skip)
        rec = []
        if not graph_steps:
            stepper.run(interval)             # what the constructor's warm-up (one reset interval) did on the graph side
        for chunk in (4 * interval, interval + 1, 2 * interval - 1):
            co, to, sc = stepper.run(chunk)
            torch.cuda.synchronize()
            rec.append([t.clone() for t in (co, to, sc) if t.numel()])
        stepper.close()
        rec.append([eng.export_state().clone(), eng.policy_actions()[0 if team == 'target' else 1].clone()])
        outs.append(rec)
        del stepper, eng
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert same(x, y)
    assert float(outs[0][-1][0][:, -2].min()) >= 2          # (export column `episode`) every environment restarted at least once


@pytest.mark.parametrize('workload', ['MATE-2v4-0.yaml', 'MATE-4v2-9.yaml', 'MATE-1v2-9.yaml', 'MATE-4v4-9.yaml'])
def test_four_per_wave_census_against_the_oracle(workload, oracle_lib, monkeypatch):
    """... and directly against the CPU oracle on the same Philox streams (not only through the one-per-wave kernels): native reset +
    fused random-policy rollouts with four environments per wave forced (MATE_SUBWAVE=1) -- no environment may differ in a mask bit or
    an integer at any step, positions 1e-9, both teams' f32 rows 1e-5 relative and the reward at EVERY step."""
    from test_gpu_stepper import _census
    monkeypatch.setenv('MATE_SUBWAVE', '1')
    diverged, worst, obs_ok, rew_ok = _census(workload, 176, 24, 8, False, oracle_lib, seed=7, first=300)
    assert diverged == 0 and worst < 1e-9 and obs_ok and rew_ok, (diverged, worst, obs_ok, rew_ok)
    eng = Engine(read_config(workload), 8, seed=1)
    assert eng.sub_wave == 4                                   # the switch was in force (engines read it when they are created)


@pytest.mark.parametrize('n', [1, 3, 17])
def test_tiny_batches(n):
    """Fewer environments than one wave's groups: the groups past the end of the batch idle, the others are what they are one per wave."""
    sub, one = _pair('MATE-2v4-9', n, max_episode_steps=6)
    for steps in (4, 5):
        out = []
        for eng in (sub, one):
            rec = []
            cam, tgt, sc = eng.rollout_random(steps, auto_reset=True, want_masks=True)
            rec += [cam.clone(), tgt.clone(), sc.clone()]
            cam, tgt, sc = eng.rollout_greedy(steps, auto_reset=True, want_masks=True)
            rec += [cam.clone(), tgt.clone(), sc.clone(), eng.export_state().clone()]
            out.append(rec)
        for x, y in zip(*out):
            assert same(x, y), (n, steps)


@pytest.mark.parametrize('name,team', [('MATE-2v4-0', 'target'), ('MATE-4v2-9', 'camera'), ('MATE-1v2-9', 'target'), ('MATE-4v4-0', 'camera')])
def test_per_step_greedy_flows_on_the_sub_wave_kernel(name, team):
    """step_versus_greedy / step_greedy -- MultiTarget / MultiCamera WITHOUT a frame skip, one launch per step -- run the one-step form of the
    four-per-wave rollout kernel where the fused flows run it (FLOW_GREEDY instead of step_greedy_kernel's FLOW_STEP_GREEDY): the caller's
    [N][...] buffers, immediate and batched restarts that write the restarted environments' first rows, and the agents' memory, bit for bit."""
    n = 70
    sub, one = _pair(name, n, max_episode_steps=9)
    gen = torch.Generator(device='cuda').manual_seed(2)
    k = sub.num_targets if team == 'target' else sub.num_cameras
    for it in range(30):
        act = (torch.rand((n, k, 2), device='cuda', generator=gen) * 2 - 1) * (25 if team == 'target' else 6)
        out = []
        for eng in (sub, one):
            auto_reset = 4 if it >= 15 else True
            if it % 3 == 2:
                eng.step_greedy(auto_reset=auto_reset)
            else:
                eng.step_versus_greedy(team, act, auto_reset=auto_reset)
            out.append([eng.camera_obs.clone(), eng.target_obs.clone(), eng.scalars.clone(), eng.export_state().clone(), eng.policy_actions()[0].clone()])
        for x, y in zip(*out):
            assert same(x, y), (name, it)
    # (MATE-1v2-* has no step_greedy_kernel: its one-per-wave per-step flows run the one-step rollout form too)
    assert sub.last_flow == 3 and one.last_flow == (3 if name.startswith('MATE-1v2') else 4) and float(sub.episode_stats[0]) >= n


@pytest.mark.parametrize('name', ['MATE-2v4-0', 'MATE-4v2-9', 'MATE-2v2-9', 'MATE-1v1-0'])
def test_per_step_flows_with_external_and_random_actions(name):
    """step(actions) / step_random -- one launch per step, the learner-in-the-loop flows -- on the one-step form of the four-per-wave rollout
    kernel (Ptrs::per_step): step()'s semantics, bit for bit.  f32 and f64 joint actions, immediate and batched restarts, a stretch of
    auto_reset = 0 in which finished environments go on stepping (as in step_kernel) before a restarting call meets them, the observe() that
    never takes this path, and a fused observation transform (the FLOW_ANY compilation)."""
    n = 70
    sub, one = _pair(name, n, max_episode_steps=7)
    gen = torch.Generator(device='cuda').manual_seed(4)
    Nc, Nt = sub.num_cameras, sub.num_targets
    for it in range(40):
        cam = (torch.rand((n, Nc, 2), device='cuda', generator=gen) * 2 - 1) * 6
        tgt = (torch.rand((n, Nt, 2), device='cuda', generator=gen) * 2 - 1) * 25
        if it % 4 == 1:
            cam, tgt = cam.double(), tgt.double()
        auto_reset = True if it < 12 else (0 if it < 18 else (3 if it < 30 else True))
        out = []
        for eng in (sub, one):
            if it == 32:
                eng.set_obs_transform(relative_coordinates=True, rescaled_observation=True)
            if it % 5 == 4:
                eng.step_random(auto_reset=auto_reset, want_masks=True)
            else:
                eng.step(cam, tgt, auto_reset=auto_reset)
            out.append([eng.camera_obs.clone(), eng.target_obs.clone(), eng.scalars.clone(), eng.masks.clone(), eng.export_state().clone()])
            if it % 7 == 6:
                co, to = eng.observe()
                out[-1] += [co.clone(), to.clone()]
        for x, y in zip(*out):
            assert same(x, y), (name, it)
    assert float(sub.episode_stats[0]) >= 3 * n and same(sub.episode_stats, one.episode_stats) and sub.idle_steps() == one.idle_steps()


def test_graph_replayed_external_actions_with_four_environments_per_wave():
    """Engine.make_stepper (the device-resident step counter, K (policy kernel, step) pairs per HIP graph, one restart launch per interval) on
    the four-per-wave kernels against direct one-per-wave launches with the host counting."""
    cfg = read_config('MATE-2v4-0.yaml', max_episode_steps=11)
    n, interval = 70, 4
    outs = []
    for sub, graph_steps in ((False, 0), (True, 2 * interval)):
        eng = Engine(cfg, n, seed=5)
        assert eng.set_sub_wave(sub) == (4 if sub else 1)
        eng.reset()
        gen = torch.Generator(device='cuda').manual_seed(9)
        flat = (torch.rand(n * 6 * 2, device='cuda', generator=gen) * 2 - 1) * 9
        cam, tgt = flat[:n * 2 * 2].view(n, 2, 2), flat[n * 2 * 2:].view(n, 4, 2)
        stepper = eng.make_stepper(cam, tgt, auto_reset=interval, graph_steps=graph_steps, between=lambda f=flat: f.mul_(-1.0).add_(0.25))
        rec = []
        if not graph_steps:
            stepper.run(interval)             # what the constructor's warm-up (one reset interval) did on the graph side
        for chunk in (4 * interval, interval + 1, 2 * interval - 1):
            stepper.run(chunk)
            torch.cuda.synchronize()
            rec.append([eng.camera_obs.clone(), eng.target_obs.clone(), eng.scalars.clone()])
        stepper.close()
        rec.append([eng.export_state().clone()])
        outs.append(rec)
        del stepper, eng
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert same(x, y)


def _small_traces():
    import glob
    import os
    import golden_util as G
    out = []
    for path in G.trace_files():
        tag = os.path.basename(path)[6:].split('_')[0]
        if tag in ('4v2-9', '2v4-0', '2v2-9', '1v1-9'):
            out.append(path)
    return out


@pytest.mark.parametrize('path', _small_traces(), ids=lambda p: __import__('os').path.basename(p)[6:-4])
def test_reference_traces_on_the_sub_wave_kernels(path):
    """The REFERENCE's recorded traces of the small scenarios (tests/golden/trace_*: its actions, its see-through and goal draws on tapes, its
    masks / integers / rewards / observations at every step) replayed through step() with four environments per wave forced -- the tape-driven
    FLOW_ANY compilation of the sub-wave kernel -- not only through the one-per-wave kernels the other parity tests run at their batch sizes."""
    import golden_util as G
    import gpu_util as U
    from test_gpu_parity import _check_step_against_trace, _replay
    fx = G.load(path)
    N = 6                                        # (two groups of the second wave idle)
    eng = U.engine_from_fixture(fx, N, obs_dtype=torch.float32)
    assert eng.set_sub_wave(True) == 4
    for s in range(len(fx['step/done'])):
        co, to, sc = _replay(eng, fx, s, N)
        _check_step_against_trace(eng, fx, s, N, co, to, sc, torch.float32)


def test_frame_skip_on_grid_indices_is_the_sum_of_one_per_wave_steps():
    """The target trainers' chain as tests/test_gpu_chain.py pins it to the reference (DiscreteTarget(5) -> MultiTarget(GreedyCameraAgent) ->
    FrameSkip(10) on MATE-2v4-0): the ten-frame launch on the FOUR-per-wave kernel, joint actions as grid indices, is bit for bit ten per-step
    calls of the one-per-wave step_greedy_kernel -- the calls that test replays against the reference's recorded chain."""
    cfg = read_config('MATE-2v4-0.yaml')
    n, skip = 128, 10
    a, b = (Engine(cfg, n, seed=31) for _ in range(2))
    assert a.set_sub_wave(True) == 4 and b.set_sub_wave(False) == 1
    for e in (a, b):
        e.set_action_grids(target_levels=5)
        e.enable_policies()
        e.reset()
    gen = torch.Generator(device='cuda').manual_seed(3)
    for ls in range(6):
        idx = torch.randint(0, 25, (n, 4), device='cuda', generator=gen, dtype=torch.int32)
        cam_r, tgt_r, sc_r = a.rollout_versus_greedy('target', idx, skip, auto_reset=False)
        for f in range(skip):
            b.step_versus_greedy('target', idx, auto_reset=False)
            assert same(sc_r[f], b.scalars) and same(cam_r[f], b.camera_obs) and same(tgt_r[f], b.target_obs), (ls, f)
    assert same(a.export_state(), b.export_state()) and b.last_flow == 4
