"""GPU (-m gpu): the graph-replayed learner-in-the-loop flow, fused rollouts over already finished environments,
and the batch-scale census against the oracle (oracle-built occlusion tables on the oracle's side)."""
import os

import numpy as np
import pytest
import torch

import gpu_util as U

pytestmark = pytest.mark.gpu

MASKS = ['camera_target_view_mask', 'target_camera_view_mask', 'target_obstacle_view_mask', 'target_target_view_mask', 'camera_camera_view_mask']
INTS = ['tgt_colliding', 'tgt_goals', 'freights', 'bounties', 'remaining_cargoes', 'awaiting_cargo_counts', 'num_delivered_cargoes', 'episode_step']


@pytest.mark.parametrize('workload,n,interval', [('MATE-4v8-9.yaml', 130, 1), ('MATE-Navigation.yaml', 64, 1), ('MATE-4v8-9.yaml', 67, 3)])
def test_graph_replayed_steps_equal_direct_steps(workload, n, interval):
    """K (policy kernel, step, idle/real auto-reset) iterations captured in ONE HIP graph with the step counter on the
    device (mate_engine_device_tick) == the same iterations launched one by one with the host counting: every output
    and the whole state bit for bit, across episode ends (time limit 7 steps, so resets happen inside the graph).
    `interval` > 1: batched resets, one reset launch per `interval` steps (the graph then holds whole intervals)."""
    from mate_amd._native import EngineError
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config(workload, max_episode_steps=7)
    outs = []
    for graph_steps in (0, 6):
        eng = Engine(cfg, n, seed=21)
        eng.reset()
        eng.step_random(auto_reset=True)      # an odd host tick before the counter moves to the device
        gen = torch.Generator(device='cuda').manual_seed(3)
        cam = (torch.rand((n, eng.num_cameras, 2), device='cuda', generator=gen) * 2 - 1) * 6
        tgt = (torch.rand((n, eng.num_targets, 2), device='cuda', generator=gen) * 2 - 1) * 25

        def policy():
            cam.mul_(-1.0).add_(0.125)
            tgt.mul_(-1.0).add_(0.25)

        stepper = eng.make_stepper(cam, tgt, auto_reset=interval, graph_steps=graph_steps, between=policy)
        rec = []
        if not graph_steps:
            stepper.run(interval)             # what the constructor's warm-up (one reset interval) did on the graph side
        for chunk in (12, 5, 7):              # full graph replays, remainders of direct launches, a call that starts inside an interval
            stepper.run(chunk)
            torch.cuda.synchronize()
            rec.append([t.clone() for t in (eng.camera_obs, eng.target_obs, eng.scalars, eng.masks)])
        if graph_steps:
            with pytest.raises(EngineError):
                eng.rollout_random(4)          # host-counted flows are refused while the counter is on the device
        stepper.close()
        eng.step_random(auto_reset=True)      # and the host-counted flow continues seamlessly afterwards
        ro = eng.rollout_random(5, auto_reset=True)
        rec.append([t.clone() for t in ro if t.numel()])
        rec.append([eng.export_state().clone()])
        outs.append(rec)
        del stepper, eng
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))


@pytest.mark.parametrize('policy', ['random', 'greedy'])
def test_rollout_restarts_environment_finished_before_the_launch(policy):
    """An episode that ended under auto_reset=False (or arrived with done = 1 through import_state) is skipped by every
    step of a later fused rollout; the launch must still list it for the reset behind it."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=5)
    eng = Engine(cfg, 9, seed=4)
    if policy == 'greedy':
        eng.enable_policies()
    rollout = eng.rollout_greedy if policy == 'greedy' else eng.rollout_random
    eng.reset()
    rollout(8, auto_reset=False)                   # time limit: every episode ends at step 6 and stays finished
    sd = eng.state_dict()
    assert (sd['done'] == 1).all() and (sd['episode'] == 1).all()
    idle0 = eng.idle_steps()
    _, _, sc = rollout(4, auto_reset=True)         # nothing steps, but the finished environments are listed and reset
    assert (sc[:, :, 2] == 2).all() and eng.idle_steps() - idle0 == 4 * 9
    sd = eng.state_dict()
    assert (sd['done'] == 0).all() and (sd['episode'] == 2).all() and (sd['episode_step'] == 0).all()
    _, _, sc = rollout(3, auto_reset=True)
    assert (sc[:, :, 2] == 0).all() and (eng.state_dict()['episode_step'] == 3).all()


def _census(workload, n, steps, fused, own_tables, oracle_lib, seed=99, first=12345):
    """Native reset + Philox random-policy rollout next to the oracle on the same streams; returns the environments
    that ever differed in a mask bit or an integer, the worst position error of the others, and the final f32
    observation error / reward equality of the others."""
    O = oracle_lib
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    threads = min(32, len(os.sched_getaffinity(0)))
    cfg = read_config(workload)
    eng = Engine(cfg, n, seed=seed, first_env_index=first, obs_dtype=torch.float32)
    eng.reset()
    torch.cuda.synchronize()
    batch = O.OracleBatch(U.oracle_proto_from_config(cfg, O), n, seed=seed, first_env_index=first)
    batch.reset(threads=threads)
    sd = eng.state_dict()
    for k in U.STATE_KEYS:
        ref = batch.gather(k)
        assert np.array_equal(sd[k].reshape(ref.shape), ref), ('reset', k)
    if not own_tables:
        for e in range(n):
            for c in range(eng.num_cameras):
                batch.env(e).set_lut(c, *eng.lut_read(e, c))
    bad = np.zeros(n, dtype=bool)
    worst = 0.0
    rows = None
    obs_bad = np.zeros(n, dtype=bool)          # an f32 row element off by more than 1e-5 relative, at ANY step, in EITHER team's rows
    rew_bad = np.zeros(n, dtype=bool)          # a step reward that is not the oracle's, at any step
    for s in range(steps):
        if fused > 1:
            if s % fused == 0:
                rows = eng.rollout_random(min(fused, steps - s), auto_reset=False, want_masks=True)
            masks = eng.unpack_masks(eng._rollout['masks'][s % fused])
        else:
            eng.step_random(auto_reset=False, want_masks=True)
            masks = eng.unpack_masks()
        batch.step(auto_reset=False, threads=threads)
        for m in MASKS:
            ref = batch.gather(m) != 0
            bad |= (masks[m].reshape(ref.shape) != ref).reshape(n, -1).any(axis=1)
        # both teams' rows and the reward of EVERY step (inside a fused launch: its row r), not only the last one's target rows
        oc, ot = batch.observe(threads=threads)
        co = (rows[0][s % fused] if fused > 1 else eng.camera_obs).cpu().numpy()
        to = (rows[1][s % fused] if fused > 1 else eng.target_obs).cpu().numpy()
        sc = (rows[2][s % fused] if fused > 1 else eng.scalars).cpu().numpy()
        if eng.num_cameras:
            obs_bad |= (np.abs(co - oc) > 1e-5 * np.maximum(1.0, np.abs(oc))).reshape(n, -1).any(axis=1)
        obs_bad |= (np.abs(to - ot) > 1e-5 * np.maximum(1.0, np.abs(ot))).reshape(n, -1).any(axis=1)
        rew_bad |= sc[:, 1] != batch.gather('reward_tgt').astype(np.float32)
        if (fused > 1 and (s % fused == fused - 1 or s == steps - 1)) or fused == 1:
            sdg = eng.state_dict()
            for k in INTS:
                ref = batch.gather(k)
                bad |= (sdg[k].reshape(ref.shape) != ref).reshape(n, -1).any(axis=1)
            good = ~bad
            worst = max(worst, float(np.abs(sdg['tgt_x'] - batch.gather('tgt_x'))[good].max()), float(np.abs(sdg['tgt_y'] - batch.gather('tgt_y'))[good].max()))
    good = ~bad
    obs_ok = not bool(obs_bad[good].any())
    rew_ok = not bool(rew_bad[good].any())
    return int(bad.sum()), worst, obs_ok, rew_ok


@pytest.mark.parametrize('workload,n,steps,fused', [('MATE-4v8-9.yaml', 4096, 32, 16), ('MATE-8v8-9.yaml', 1024, 24, 1), ('MATE-Navigation.yaml', 2048, 32, 32),
                                                    ('MATE-4v8-0.yaml', 2048, 32, 8), ('MATE-4v2-9.yaml', 1024, 32, 1)])
def test_batch_scale_census_vs_oracle(workload, n, steps, fused, oracle_lib):
    """The soak of tests/soak_vs_oracle.py at a driver-runnable size: with identical occlusion tables on both sides no
    environment may ever differ from the oracle in a mask bit or an integer; positions to 1e-9, f32 observations to
    1e-5 relative, rewards equal."""
    diverged, worst, obs_ok, rew_ok = _census(workload, n, steps, fused, False, oracle_lib)
    assert diverged == 0 and worst < 1e-9 and obs_ok and rew_ok, (diverged, worst, obs_ok, rew_ok)


@pytest.mark.parametrize('workload,n', [('MATE-4v8-9.yaml', 1024), ('MATE-8v8-9.yaml', 512)])
def test_step_masks_with_independently_built_tables(workload, n, oracle_lib):
    """The same census with the oracle keeping the occlusion tables IT built (nothing copied from the device), so that a
    table divergence would surface as a mask disagreement.  The two builders differ only where the reference itself is
    a coin flip -- rays exactly tangent to an obstacle, clipped or not by the last bit of asin / atan2 (DESIGN.md
    section 5), at most two knots per obstacle and table -- and a target has to stand within a hundredth of a degree
    of such a ray, behind the obstacle, inside the sector, to notice: at most 1 environment in 256 may diverge, and all
    the others must agree with the oracle exactly."""
    diverged, worst, obs_ok, rew_ok = _census(workload, n, 40, 1, True, oracle_lib)
    assert diverged <= max(1, n // 256) and worst < 1e-9 and obs_ok and rew_ok, (diverged, worst, obs_ok, rew_ok)


@pytest.mark.parametrize('team,interval', [('camera', 1), ('target', 2)])
def test_graph_replayed_learner_versus_greedy(team, interval):
    """The single-team training loop -- the learner's policy rewrites ITS team's joint action, the greedy opponents act on the
    device, the environment steps (Engine.step_versus_greedy = MultiCamera / MultiTarget) -- captured in one HIP graph with the step
    counter on the device == the same iterations launched one by one: outputs, state and the agents' joint actions bit for bit,
    across episode ends (time limit 9)."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=9)
    n = 70
    outs = []
    for graph_steps in (0, 6):
        eng = Engine(cfg, n, seed=5)
        eng.enable_policies()
        eng.reset()
        agents = eng.num_cameras if team == 'camera' else eng.num_targets
        gen = torch.Generator(device='cuda').manual_seed(9)
        mine = (torch.rand((n, agents, 2), device='cuda', generator=gen) * 2 - 1) * (6 if team == 'camera' else 25)

        def policy():
            mine.mul_(-1.0).add_(0.125)

        stepper = eng.make_stepper(mine if team == 'camera' else None, mine if team == 'target' else None, auto_reset=interval,
                                   graph_steps=graph_steps, between=policy, versus=team)
        rec = []
        if not graph_steps:
            stepper.run(interval)
        for chunk in (12, 5, 7):
            stepper.run(chunk)
            torch.cuda.synchronize()
            rec.append([t.clone() for t in (eng.camera_obs, eng.target_obs, eng.scalars, eng.masks) + tuple(eng.policy_actions())])
        stepper.close()
        eng.step_greedy(auto_reset=True)      # the host-counted flow continues seamlessly afterwards
        rec.append([eng.export_state().clone(), eng.scalars.clone()])
        outs.append(rec)
        del stepper, eng
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))
    assert float(outs[0][-1][0][:, -2].min()) >= 2          # (export column `episode`) every environment restarted at least once


@pytest.mark.parametrize('team,config,skip,interval', [('camera', 'MATE-4v8-9.yaml', 5, 2), ('target', 'MATE-2v4-0.yaml', 10, 1),
                                                       ('camera', 'MATE-4v8-9.yaml', 3, 4)])
def test_graph_replayed_frame_skip_versus_greedy(team, config, skip, interval):
    """FrameSkip(K) over MultiCamera / MultiTarget, the example trainers' loop, in ONE HIP graph: every learner action is one K-frame
    launch (rollout_versus_greedy), the step counter on the device advances by interval * K at the restart launch behind every
    `interval`-th launch == the same launches made one by one with the host counting: the rollout-shaped outputs, the state and the
    agents' memory bit for bit, across episode ends (time limit 23 frames), through a call that stops inside an interval, and on
    into the host-counted flow after close()."""
    from mate_amd._native import EngineError
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config(config, max_episode_steps=23)
    n = 70
    outs = []
    for graph_steps in (0, 2 * interval):
        eng = Engine(cfg, n, seed=5)
        eng.enable_policies()
        eng.reset()
        eng.step_greedy(auto_reset=True)      # an odd host tick before the counter moves to the device
        agents = eng.num_cameras if team == 'camera' else eng.num_targets
        gen = torch.Generator(device='cuda').manual_seed(9)
        mine = (torch.rand((n, agents, 2), device='cuda', generator=gen) * 2 - 1) * (6 if team == 'camera' else 25)

        def policy():
            mine.mul_(-1.0).add_(0.125)

        stepper = eng.make_stepper(mine if team == 'camera' else None, mine if team == 'target' else None, auto_reset=interval,
                                   graph_steps=graph_steps, between=policy, versus=team, frame_skip=skip)
        rec = []
        if not graph_steps:
            stepper.run(interval)             # what the constructor's warm-up (one reset interval) did on the graph side
        for chunk in (4 * interval, interval + 1, 2 * interval + 1):
            co, to, sc = stepper.run(chunk)
            torch.cuda.synchronize()
            assert sc.shape[0] == skip and to.shape[0] == skip
            rec.append([t.clone() for t in (co, to, sc) if t.numel()])
        if graph_steps and (7 * interval + 2) % interval:      # the last call stopped inside a reset interval
            with pytest.raises(EngineError):
                eng.step_versus_greedy(team, mine, auto_reset=interval)      # a one-frame step inside an interval of K-frame launches
        stepper.close()
        eng.step_greedy(auto_reset=True)      # the host-counted flow continues seamlessly afterwards
        rec.append([eng.export_state().clone(), eng.scalars.clone(), eng.camera_obs.clone(), eng.target_obs.clone()])
        outs.append(rec)
        del stepper, eng
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))
    assert float(outs[0][-1][0][:, -2].min()) >= 2          # (export column `episode`) every environment restarted at least once


@pytest.mark.parametrize('policy', ['random', 'greedy'])
def test_rollouts_with_batched_resets(policy):
    """auto_reset = k > 1 on the fused rollouts: finished environments idle through the following launches and all
    restart together after every k-th call; executed steps + idle slots account for every slot, nothing is lost."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=9)
    n = 21
    eng = Engine(cfg, n, seed=8)
    if policy == 'greedy':
        eng.enable_policies()
    rollout = eng.rollout_greedy if policy == 'greedy' else eng.rollout_random
    eng.reset()
    episodes = []
    for call in range(9):
        _, _, sc = rollout(4, auto_reset=3)
        sd = eng.state_dict()
        episodes.append(sd['episode'].copy())
        if call % 3 != 2:           # between batched resets a finished environment stays finished
            assert ((sd['done'] != 0) == (sd['episode_step'] >= 10)).all()
        else:                       # the third call restarted everything that had finished
            assert (sd['done'] == 0).all()
    # time limit 9 -> done on the 10th step: every environment finishes in call 2 (steps 9-12), idles, restarts after call 2, ...
    assert (episodes[1] == 1).all() and (episodes[2] == 2).all() and (episodes[5] == 3).all() and (episodes[8] == 4).all()
    assert eng.idle_steps() == n * 3 * 2          # per cycle of 12 slots: 10 executed steps, 2 idle


def test_batched_step_resets_list_once_and_survive_mode_changes():
    """auto_reset = k on step(): a finished environment idles (done = 2 rows) and is listed exactly once for the reset
    launch that closes the interval, whatever happened before: episodes that ended under auto_reset = 0, a change of k or
    of the flow inside an interval (the pending interval is flushed), and the specialised flows run it
    (no fall-back to the generic kernel)."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=6)
    n = 33
    eng = Engine(cfg, n, seed=12)
    eng.reset()
    for _ in range(8):
        eng.step_random(auto_reset=False)                 # every episode ends at step 7 and stays finished, unlisted
    assert (eng.state_dict()['done'] == 1).all()
    eng.step_random(auto_reset=4)                         # 1st of the interval: all idle, all listed now
    assert eng.last_flow == 1 and (eng.scalars[:, 2] == 2).all()
    assert (eng.state_dict()['done'] == 3).all()
    for _ in range(3):
        eng.step_random(auto_reset=4)                     # ... 4th: the interval's reset launch restarts all of them once
    sd = eng.state_dict()
    assert (sd['done'] == 0).all() and (sd['episode'] == 2).all() and (sd['episode_step'] == 0).all()
    for _ in range(9):                                    # 7 steps to the time limit, then idle; interval boundaries at 4, 8
        eng.step_random(auto_reset=4)
    sd = eng.state_dict()
    assert (sd['episode'] == 3).all() and (sd['episode_step'] == 1).all()
    # a change of mode inside an interval restarts what has finished so far and forgets the lists
    eng2 = Engine(cfg, n, seed=12)
    eng2.reset()
    for _ in range(7):
        eng2.step_random(auto_reset=8)                    # finished at the 7th step, listed, interval not over
    assert (eng2.state_dict()['done'] == 3).all()
    eng2.rollout_random(3, auto_reset=True)               # flushes: everything restarted before the rollout runs
    sd = eng2.state_dict()
    assert (sd['done'] == 0).all() and (sd['episode'] == 2).all() and (sd['episode_step'] == 3).all()
    eng2.step_random(auto_reset=True)
    assert (eng2.state_dict()['episode_step'] == 4).all()


@pytest.mark.parametrize('greedy', [False, True])
def test_listed_environment_restarted_by_mask_is_not_restarted_again(greedy):
    """auto_reset = k on step(): an environment on the interval's finished-episode list that the caller restarts itself
    (reset(env_mask)) before the interval ends is in a NEW episode when the interval's reset launch runs -- that launch
    must leave it alone (no second restart in mid-episode, no rewritten view), and still restart the others.  `greedy`:
    the list-driven reset split into placement / table / view launches (the flows with the on-device agents).
    A twin engine whose interval (k = 16) does not end at the 8th step is the reference for the restarted ones."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=4)
    n = 19
    mask = torch.zeros(n, dtype=torch.uint8)
    mask[::3] = 1
    picked = mask.numpy().astype(bool)
    runs = []
    for k in (8, 16):
        eng = Engine(cfg, n, seed=5)
        if greedy:
            eng.enable_policies()
        step = (lambda: eng.step_greedy(auto_reset=k)) if greedy else (lambda: eng.step_random(auto_reset=k))
        eng.reset()
        for _ in range(5):
            step()                                        # time limit 4 -> done on the 5th step: all finished and listed
        assert (eng.state_dict()['done'] == 3).all()
        eng.reset(env_mask=mask)                          # the caller restarts a third of them inside the interval
        sd = eng.state_dict()
        assert (sd['done'][picked] == 0).all() and (sd['episode'][picked] == 2).all() and (sd['done'][~picked] == 3).all()
        for _ in range(3):
            step()                                        # 6th .. 8th: the restarted ones advance; k = 8: then the interval's reset launch
        torch.cuda.synchronize()
        runs.append((eng.state_dict(), eng.target_obs.clone(), eng.camera_obs.clone(), eng.scalars.clone(), eng))
    sd, sd16 = runs[0][0], runs[1][0]
    assert (sd['episode'][picked] == 2).all() and (sd['episode_step'][picked] == 3).all() and (sd['done'][picked] == 0).all()
    assert (sd['episode'][~picked] == 2).all() and (sd['episode_step'][~picked] == 0).all() and (sd['done'][~picked] == 0).all()
    assert (sd16['episode'][~picked] == 1).all() and (sd16['done'][~picked] == 3).all()          # the twin's interval is still open
    for key in sd:                                        # state, last observations and step record of the restarted ones: untouched
        assert np.array_equal(sd[key][picked], sd16[key][picked]), key
    sel = torch.from_numpy(picked).cuda()
    for a, b in zip(runs[0][1:4], runs[1][1:4]):
        assert torch.equal(a[sel].view(torch.uint8), b[sel].view(torch.uint8))


@pytest.mark.parametrize('policy', ['random', 'greedy'])
def test_idle_rows_of_a_fused_rollout_are_marked_and_never_written(policy):
    """The stale-row contract of the fused rollouts (Engine.reserve_rollout): a slot behind the end of an episode inside
    the launch carries done = 2 and zeros in its scalar row and is NOT written in the observation / mask buffers -- those
    rows keep what they held (here: a NaN / -1 sentinel), every executed row is fully written, and the batched API's
    `info['skipped']` is exactly the done == 2 slots.  The reference never returns such rows (its loop stops at `done`):
    a consumer drops them by that flag."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=5)
    n, K = 13, 9
    eng = Engine(cfg, n, seed=4)
    if policy == 'greedy':
        eng.enable_policies()
    eng.reset()
    buf = eng.reserve_rollout(K, want_masks=True)
    buf['camera_obs'].fill_(float('nan')); buf['target_obs'].fill_(float('nan')); buf['masks'].fill_(-1)
    fn = eng.rollout_greedy if policy == 'greedy' else eng.rollout_random
    cam, tgt, sc = fn(K, auto_reset=True, want_masks=True)
    torch.cuda.synchronize()
    idle = sc[..., 2] == 2
    assert idle[6:].all() and not idle[:5].any()             # time limit 5: the 6th step ends every episode still running
    assert (sc[idle][:, [0, 1, 3, 4, 5, 6, 7]] == 0).all()
    assert torch.isnan(cam[idle]).all() and torch.isnan(tgt[idle]).all() and (buf['masks'][:K][idle] == -1).all()
    assert torch.isfinite(cam[~idle]).all() and torch.isfinite(tgt[~idle]).all() and (buf['masks'][:K][~idle][:, -1] == 1).all()
    assert eng.idle_steps() == int(idle.sum())
    from mate_amd.environment import BatchedMultiAgentTracking
    env = BatchedMultiAgentTracking(cfg, n, seed=4)
    if policy == 'greedy':
        env.enable_greedy_policies()
    env.reset()
    _, _, done, info = env.rollout_greedy(K) if policy == 'greedy' else env.rollout_random(K)
    assert torch.equal(info['skipped'], idle) and torch.equal(done, sc[..., 2] == 1)


@pytest.mark.parametrize('workload,n', [('MATE-4v8-9.yaml', 301), ('MATE-4v8-0.yaml', 130), ('MATE-4v2-9.yaml', 67), ('MATE-Navigation.yaml', 131)])
def test_row_image_rollout_equals_descriptor_rollout(workload, n):
    """The fused random-policy rollout of the shapes with a row-image compilation (observation rows resident in LDS, every
    visibility lane writes its own block) against the same kernel packing through the descriptor table (MATE_NO_IMAGE=1,
    read at create): every output row, the masks and the state bit for bit, across launches, restarts and idle slots."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config(workload, max_episode_steps=11)
    runs = []
    for no_image in ('0', '1'):
        os.environ['MATE_NO_IMAGE'] = no_image
        try:
            eng = Engine(cfg, n, seed=31, first_env_index=7)
        finally:
            os.environ.pop('MATE_NO_IMAGE', None)
        eng.reset()
        rec = []
        for steps in (5, 9, 1, 16):
            buf = eng.reserve_rollout(16, want_masks=True)
            for key in ('camera_obs', 'target_obs'):
                buf[key].fill_(-7.0)
            cam, tgt, sc = eng.rollout_random(steps, auto_reset=True, want_masks=True)
            torch.cuda.synchronize()
            rec.append([cam.clone(), tgt.clone(), sc.clone(), buf['masks'][:steps].clone(), eng.export_state().clone()])
        runs.append(rec)
        assert eng.last_flow == 1
    for a, b in zip(*runs):
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))
    assert (runs[0][-1][2][..., 2] == 2).any() and (runs[0][1][2][..., 2] == 1).any()      # episodes ended and slots idled inside the launches


@pytest.mark.parametrize('workload,n', [('MATE-4v8-9.yaml', 517), ('MATE-8v8-9.yaml', 130), ('MATE-4v2-9.yaml', 66), ('MATE-4v8-0.yaml', 129),
                                        ('MATE-2v4-any', 33)])
def test_two_wave_step_equals_one_wave_step(workload, n):
    """step_split_kernel (an environment split over two waves: cameras / sector tests / goals on one, targets / range tests on the
    other, two workgroup barriers; MATE_STEP_SPLIT=1) against step_kernel (MATE_STEP_SPLIT=0) in the folded flows it is compiled
    for -- the on-device random policy with immediate and with batched resets, caller-supplied f32 and f64 joint actions, the
    graph-replayed stepper -- every output and the whole state bit for bit, across episode ends (time limit 9).  The last case is
    a shape without a compiled specialisation (2 cameras, 4 targets, 3 obstacles: the generic kernels)."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    if workload == 'MATE-2v4-any':
        cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=9)
        cfg['camera']['location_random_range'] = list(cfg['camera']['location_random_range'])[:2]
        cfg['target']['location_random_range'] = list(cfg['target']['location_random_range'])[:4]
        cfg['obstacle']['location_random_range'] = list(cfg['obstacle']['location_random_range'])[:3]
    else:
        cfg = read_config(workload, max_episode_steps=9)
    outs = []
    for split in ('0', '1'):                  # one wave per environment; two (one 128-thread workgroup per environment)
        os.environ['MATE_STEP_SPLIT'] = split
        try:
            eng = Engine(cfg, n, seed=31, first_env_index=7)
        finally:
            os.environ.pop('MATE_STEP_SPLIT', None)
        eng.reset()
        rec = []

        def snap():
            rec.append([t.clone() for t in (eng.camera_obs, eng.target_obs, eng.scalars, eng.masks)])

        for s in range(14):
            eng.step_random(auto_reset=True, want_masks=True)
            assert eng.last_flow == 1
            snap()
        for s in range(17):
            eng.step_random(auto_reset=5, want_masks=True)
            snap()
        gen = torch.Generator(device='cuda').manual_seed(11)
        for dtype in (torch.float32, torch.float64):
            cam = ((torch.rand((n, eng.num_cameras, 2), device='cuda', generator=gen) * 2 - 1) * 7).to(dtype)
            tgt = ((torch.rand((n, eng.num_targets, 2), device='cuda', generator=gen) * 2 - 1) * 30).to(dtype)
            for s in range(12):
                eng.step(cam, tgt, auto_reset=True)
                assert eng.last_flow == 2
                snap()
                cam.mul_(-1.0).add_(0.25); tgt.mul_(-1.0).add_(0.5)
        cam = (torch.rand((n, eng.num_cameras, 2), device='cuda', generator=gen) * 2 - 1) * 7
        tgt = (torch.rand((n, eng.num_targets, 2), device='cuda', generator=gen) * 2 - 1) * 30

        def policy():
            cam.mul_(-1.0).add_(0.125)
            tgt.mul_(-1.0).add_(0.25)

        stepper = eng.make_stepper(cam, tgt, auto_reset=4, graph_steps=8, between=policy)
        for _ in range(3):
            stepper.run(8)
            snap()
        stepper.close()
        rec.append([eng.export_state().clone()])
        outs.append(rec)
        assert (eng.state_dict()['episode'] >= 3).all()
        del stepper, eng
    for other in outs[1:]:
        for i, (a, b) in enumerate(zip(outs[0], other)):
            for x, y in zip(a, b):
                assert torch.equal(x.view(torch.uint8), y.view(torch.uint8)), i


SHIPPED = ['MATE-1v1-0', 'MATE-1v1-9', 'MATE-1v2-0', 'MATE-1v2-9', 'MATE-2v2-0', 'MATE-2v2-9', 'MATE-2v4-0', 'MATE-2v4-9', 'MATE-4v2-0', 'MATE-4v2-9',
           'MATE-4v4-0', 'MATE-4v4-9', 'MATE-4v8-0', 'MATE-4v8-9', 'MATE-8v8-0', 'MATE-8v8-9', 'MATE-Navigation']


@pytest.mark.parametrize('name', SHIPPED)
def test_every_shipped_scenario_runs_compiled_kernels_that_equal_the_generic_ones_and_the_oracle(name, oracle_lib):
    """All seventeen scenarios the reference ships (mate/assets/*.yaml) have compiled kernel specialisations since round 4
    (csrc/shape_groups.hpp; twelve of them ran the generic kernels before, at less than half the fused rate).  For each: the
    specialised build reports itself; fused rollouts (row-image path where the shape has one), per-step launches of both folded
    flows and Greedy-vs-Greedy rollouts equal the generic kernels (MATE_GENERIC=1) bit for bit across episode ends; and a native
    reset + Philox rollout stays with the oracle (masks and integers exact, positions 1e-9)."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    workload = name + '.yaml'
    n = 70
    cfg = read_config(workload, max_episode_steps=12)
    runs = []
    for generic in ('0', '1'):
        os.environ['MATE_GENERIC'] = generic
        try:
            eng = Engine(cfg, n, seed=77, first_env_index=5)
        finally:
            os.environ.pop('MATE_GENERIC', None)
        assert eng.specialised == (generic == '0')
        eng.enable_policies()
        eng.reset()
        rec = []
        for steps in (7, 9):
            cam, tgt, sc = eng.rollout_random(steps, auto_reset=True, want_masks=True)
            rec.append([cam.clone(), tgt.clone(), sc.clone(), eng._rollout['masks'][:steps].clone()])
        gen = torch.Generator(device='cuda').manual_seed(2)
        cam_a = (torch.rand((n, eng.num_cameras, 2), device='cuda', generator=gen) * 2 - 1) * 6
        tgt_a = (torch.rand((n, eng.num_targets, 2), device='cuda', generator=gen) * 2 - 1) * 25
        for s in range(10):
            eng.step_random(auto_reset=True, want_masks=True)
            rec.append([eng.camera_obs.clone(), eng.target_obs.clone(), eng.scalars.clone(), eng.masks.clone()])
            eng.step(cam_a, tgt_a, auto_reset=3)
            rec.append([eng.camera_obs.clone(), eng.target_obs.clone(), eng.scalars.clone()])
        cam, tgt, sc = eng.rollout_greedy(8, auto_reset=True)
        rec.append([cam.clone(), tgt.clone(), sc.clone()])
        rec.append([eng.export_state().clone()])
        runs.append(rec)
        del eng
    for i, (a, b) in enumerate(zip(*runs)):
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8)), i
    diverged, worst, obs_ok, rew_ok = _census(workload, 96, 16, 8, False, oracle_lib, seed=5, first=900)
    assert diverged == 0 and worst < 1e-9 and obs_ok and rew_ok, (diverged, worst, obs_ok, rew_ok)
