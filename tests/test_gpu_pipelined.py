"""Pipelined restarts of the fused Greedy rollouts (mate_engine_rollout_greedy, auto_reset = MATE_RESET_PIPELINED): the reset of what
launch n finishes runs on the engine's side stream under launch n + 1, restarted environments join launch n + 2."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(cfg, n, K, launches, serial, seed=9, mode='pipelined'):
    from mate_amd.engine import Engine
    os.environ['MATE_PIPELINED_SERIAL'] = '1' if serial else '0'
    try:
        eng = Engine(cfg, n, seed=seed, first_env_index=3)
        eng.enable_policies()
        eng.reset()
        rec = []
        for _ in range(launches):
            cam, tgt, sc = eng.rollout_greedy(K, auto_reset=mode, want_masks=True)
            rec.append((cam.clone(), tgt.clone(), sc.clone(), eng._rollout['masks'][:K].clone()))
        idle = eng.idle_steps()
        state = eng.export_state().clone()           # (any other entry point: waits for the resets in flight, clears the hand-over tags)
    finally:
        os.environ.pop('MATE_PIPELINED_SERIAL', None)
    return eng, rec, state, idle


@pytest.mark.parametrize('workload,n', [('MATE-8v8-9.yaml', 203), ('MATE-4v8-9.yaml', 130)])
def test_pipelined_restarts_equal_their_serial_form_and_join_two_launches_later(workload, n):
    """The concurrent form (resets on the side stream, under the next launch) against the same protocol with the resets on the
    caller's stream (MATE_PIPELINED_SERIAL=1): every row and the final state bit for bit -- what a launch does never depends on how
    far the concurrent reset has come.  And the protocol itself: an environment whose episode ends in launch n idles through the
    rest of n and all of n + 1 and steps again from the first row of n + 2, in a fresh episode."""
    from mate_amd.config import read_config
    cfg = read_config(workload, max_episode_steps=17)
    K, launches = 8, 14
    _, conc, state_c, idle_c = _run(cfg, n, K, launches, serial=False)
    eng, ser, state_s, idle_s = _run(cfg, n, K, launches, serial=True)
    for a, b in zip(conc, ser):
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))
    assert torch.equal(state_c.view(torch.uint8), state_s.view(torch.uint8)) and idle_c == idle_s
    done = torch.stack([r[2][:, :, 2] for r in conc]).cpu().numpy()          # [launch][step][env]: 0 running, 1 the step that ended the episode, 2 idle
    finished = 0
    for ln in range(launches - 2):
        ended = (done[ln] == 1).any(axis=0)
        finished += int(ended.sum())
        assert (done[ln + 1][:, ended] == 2).all()                           # the reset owns them for the whole next launch
        assert (done[ln + 2][0, ended] != 2).all()                           # ... and they are live again in the one after
        for env in np.nonzero(ended)[0][:8]:
            r = int(np.argmax(done[ln][:, env] == 1))
            assert (done[ln][r + 1:, env] == 2).all()
    assert finished >= 2 * n                                                 # the time limit (17) ended every episode, more than once
    sd = {k: state_s[:, off:off + (int(np.prod(shape)) if shape else 1)].cpu().numpy() for k, (off, shape) in eng.export_fields.items()}
    assert set(np.unique(sd['done'])) <= {0.0, 1.0, 3.0}                     # no hand-over tag survives the mode
    assert (sd['episode'] >= 3).all()
    # the ordinary flows carry on from there
    cam, tgt, sc = eng.rollout_greedy(K, auto_reset=True)
    torch.cuda.synchronize()
    assert torch.isfinite(tgt).all() and (sc[0, :, 2] != 2).sum() >= n // 2


@pytest.mark.parametrize('every', [3, 4])
def test_pipelined_restarts_behind_every_mth_launch(every):
    """auto_reset = ('pipelined', m) (the C ABI's -m): ONE restart launch, on the side stream, behind every m-th rollout launch --
    short launches (a learner's FrameSkip(5) actions) whose restart group, four latency-bound launches, is longer than a launch.
    The concurrent form equals the serial one bit for bit; an environment whose episode ends in interval i (launches i m ... i m + m - 1)
    idles through the rest of i and all of i + 1 and is live again in the first launch of interval i + 2; leaving the mode inside an
    interval (any other entry point) restarts what that interval had listed."""
    from mate_amd.config import read_config
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=29)
    n, K, launches = 130, 5, 8 * every + 2                  # (stops inside an interval)
    mode = ('pipelined', every)
    _, conc, state_c, idle_c = _run(cfg, n, K, launches, serial=False, mode=mode)
    eng, ser, state_s, idle_s = _run(cfg, n, K, launches, serial=True, mode=mode)
    for a, b in zip(conc, ser):
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))
    assert torch.equal(state_c.view(torch.uint8), state_s.view(torch.uint8)) and idle_c == idle_s
    done = torch.stack([r[2][:, :, 2] for r in conc]).cpu().numpy()          # [launch][frame][env]
    intervals = launches // every
    per = done[:intervals * every].reshape(intervals, every * K, n)           # [interval][frame of the interval][env]
    finished = 0
    for i in range(intervals - 2):
        ended = (per[i] == 1).any(axis=0)
        finished += int(ended.sum())
        assert (per[i + 1][:, ended] == 2).all()                             # the restart owns them for the whole next interval
        assert (per[i + 2][0, ended] != 2).all()                             # ... and they are live again in the one after
        for env in np.nonzero(ended)[0][:8]:
            r = int(np.argmax(per[i][:, env] == 1))
            assert (per[i][r + 1:, env] == 2).all()
    assert finished >= n
    sd = {k: state_s[:, off:off + (int(np.prod(shape)) if shape else 1)].cpu().numpy() for k, (off, shape) in eng.export_fields.items()}
    assert set(np.unique(sd['done'])) <= {0.0}                               # the open interval's finished environments restarted on the way out
    cam, tgt, sc = eng.rollout_greedy(K, auto_reset=True)
    torch.cuda.synchronize()
    assert torch.isfinite(tgt).all() and (sc[0, :, 2] != 2).all()


def test_pipelined_mode_changes_nothing_while_no_episode_ends():
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-4v8-9.yaml')
    outs = []
    for mode in (True, 'pipelined'):
        eng = Engine(cfg, 96, seed=4)
        eng.enable_policies()
        eng.reset()
        rec = []
        for _ in range(4):
            rec.append([t.clone() for t in eng.rollout_greedy(6, auto_reset=mode)])
        rec.append([eng.export_state().clone()])
        outs.append(rec)
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))


@pytest.mark.parametrize('form', ['two_launch_tape', 'one_launch', 'versus_rollout'])
def test_per_step_flows_behind_pipelined_rollouts_wait_for_the_resets_in_flight(form):
    """Every other entry point first waits for the resets in flight on the side stream (include/mate_engine.h).  The two-launch
    form of step_greedy -- recorded agent draws -- used to launch the AGENTS' kernel before anything had left the pipelined mode
    (round 4's advisor finding): its agents read records, masks and `done` tags the side-stream reset was still writing.  The
    concurrent form must equal the serial one (MATE_PIPELINED_SERIAL=1: resets on the caller's stream), bit for bit, for the
    two-launch form, the one-launch form and the FrameSkip rollout with auto_reset = 'pipelined' spelled as the C enum's name."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=11)
    n, outs = 160, []
    for serial in (False, True):
        os.environ['MATE_PIPELINED_SERIAL'] = '1' if serial else '0'
        try:
            eng = Engine(cfg, n, seed=21)
            eng.enable_policies()
            eng.reset()
            rec = []
            gen = torch.Generator(device='cuda')
            gen.manual_seed(77)
            for _ in range(6):
                eng.rollout_greedy(6, auto_reset='pipelined')
                if form == 'two_launch_tape':
                    tape = {'camera_resample_u': torch.rand((n, 4), device='cuda', generator=gen, dtype=torch.float64)}
                    eng.step_greedy(policy_tape=tape, auto_reset=True)
                elif form == 'one_launch':
                    eng.step_greedy(auto_reset=True)
                else:
                    mine = torch.full((n, 4, 2), 1.5, device='cuda')
                    eng.rollout_versus_greedy('camera', mine, 3, auto_reset='pipelined')
                    eng.step_greedy(auto_reset=True)
                rec.append((eng.scalars.clone(), eng.masks.clone(), eng.target_obs.clone(), eng.export_state().clone()))
        finally:
            os.environ.pop('MATE_PIPELINED_SERIAL', None)
        outs.append(rec)
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))
    assert (outs[0][-1][3][:, -2] >= 2).all()          # episodes ended and restarted under the launches
