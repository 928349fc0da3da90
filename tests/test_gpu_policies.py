"""GPU parity of the on-device rule-based policies (row f1): GreedyCameraAgent vs GreedyTargetAgent run
closed-loop on the device with the reference agents' recorded draws must reproduce the reference's joint
actions and the resulting environment trace (fixtures greedy_*.npz, recorded by tests/golden/make_golden.py agents)."""
import os

import numpy as np
import pytest
import torch

import golden_util as G
import gpu_util as U

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['greedy_4v8-9_s5', 'greedy_8v8-9_s6', 'greedy_4v2-9_s7'])
def test_greedy_policies_closed_loop(name):
    fx = G.load(name + '.npz')
    N = 2
    eng = U.engine_from_fixture(fx, N, obs_dtype=torch.float64)
    eng.enable_policies()
    Nc, Nt = eng.num_cameras, eng.num_targets
    dev = eng.device
    # the agents first act on the reset observation: reproduce the reference's reset view masks
    # (see-through draws of reset() are not on tape; re-derive the masks from the recorded ones)
    tape0 = torch.from_numpy(np.where(fx['reset/camera_target_view_mask'], 1.0, 0.0)[None].repeat(N, 0)).to(dev)
    eng.observe(tape_ct=tape0)     # u = 1 sees through, u = 0 goes to the occlusion test
    m0 = eng.unpack_masks()
    assert np.array_equal(m0['camera_target_view_mask'][0], fx['reset/camera_target_view_mask'])

    def bc(a, dtype=np.float64):
        a = np.asarray(a)
        return torch.from_numpy(np.broadcast_to(a, (N,) + a.shape).astype(dtype).copy()).to(dev)

    T = len(fx['step/done'])
    for s in range(T):
        tape = {
            'camera_resample_u': bc(np.nan_to_num(fx['step/agent_cam_binom_u'][s], nan=0.0)),
            'camera_sample_u': bc(np.nan_to_num(fx['step/agent_cam_sample_u'][s], nan=0.0)),
            'camera_delay': bc(fx['step/agent_cam_delay'][s], np.int32),
            'target_choice_u': bc(np.nan_to_num(fx['step/agent_tgt_choice_u'][s], nan=0.0)),
            'target_resample_u': bc(np.nan_to_num(fx['step/agent_tgt_binom_u'][s], nan=0.0)),
            'target_sample_u': bc(np.nan_to_num(fx['step/agent_tgt_sample_u'][s], nan=0.0)),
            'target_reset_sample_u': bc(fx['agent/tgt_reset_sample_u']),
        }
        env_tape = bc(np.nan_to_num(fx['step/tape_ct'][s], nan=0.0))
        goal_tape = bc(np.nan_to_num(fx['step/goal_u'][s], nan=0.0))
        eng.step_greedy(policy_tape=tape, tape_ct=env_tape, tape_goal=goal_tape, auto_reset=False)
        cam_act, tgt_act = eng.policy_actions()
        for e in range(N):
            assert np.abs(tgt_act[e].cpu().numpy() - fx['step/tgt_act'][s]).max() < 1e-8, ('target action', s)
            if Nc:
                assert np.abs(cam_act[e].cpu().numpy() - fx['step/cam_act'][s]).max() < 1e-8, ('camera action', s)
        masks = eng.unpack_masks()
        assert np.array_equal(masks['camera_target_view_mask'][0], fx['step/camera_target_view_mask'][s]), s
        sd = eng.state_dict()
        assert np.abs(sd['tgt_x'][0] - fx['step/tgt_xy'][s][:, 0]).max() < 1e-8
        assert np.array_equal(sd['tgt_goals'][0], fx['step/tgt_goals'][s].astype(np.float64)), s
        assert np.array_equal(sd['bounties'][1], fx['step/bounties'][s].astype(np.float64)), s
        assert sd['episode_reward'][0] == fx['step/episode_reward'][s]


@pytest.mark.parametrize('team', ['camera', 'target'])
@pytest.mark.parametrize('name,act_dtype', [('greedy_4v8-9_s5', torch.float64), ('greedy_8v8-9_s6', torch.float64),
                                            ('greedy_4v8-9_s5', torch.float32)])
def test_single_team_versus_greedy_closed_loop(name, act_dtype, team):
    """MultiCamera / MultiTarget (mate/wrappers/single_team.py:245-264): the caller's team replays the joint action the
    reference's own agents took (fixture), the opponents are the on-device agents on the recorded draws -> the
    reference's trace.  With f32 caller actions the f64 trace is only reproduced up to the rounding of those actions."""
    fx = G.load(name + '.npz')
    N = 2
    eng = U.engine_from_fixture(fx, N, obs_dtype=torch.float64)
    eng.enable_policies()
    dev = eng.device
    tape0 = torch.from_numpy(np.where(fx['reset/camera_target_view_mask'], 1.0, 0.0)[None].repeat(N, 0)).to(dev)
    eng.observe(tape_ct=tape0)

    def bc(a, dtype=np.float64):
        a = np.asarray(a)
        return torch.from_numpy(np.broadcast_to(a, (N,) + a.shape).astype(dtype).copy()).to(dev)

    exact = act_dtype == torch.float64
    tol = 1e-8 if exact else 2e-3
    T = len(fx['step/done']) if exact else 25          # rounded actions: compare while the trajectories cannot have forked yet
    for s in range(T):
        tape = {
            'camera_resample_u': bc(np.nan_to_num(fx['step/agent_cam_binom_u'][s], nan=0.0)),
            'camera_sample_u': bc(np.nan_to_num(fx['step/agent_cam_sample_u'][s], nan=0.0)),
            'camera_delay': bc(fx['step/agent_cam_delay'][s], np.int32),
            'target_choice_u': bc(np.nan_to_num(fx['step/agent_tgt_choice_u'][s], nan=0.0)),
            'target_resample_u': bc(np.nan_to_num(fx['step/agent_tgt_binom_u'][s], nan=0.0)),
            'target_sample_u': bc(np.nan_to_num(fx['step/agent_tgt_sample_u'][s], nan=0.0)),
            'target_reset_sample_u': bc(fx['agent/tgt_reset_sample_u']),
        }
        mine = bc(fx['step/cam_act' if team == 'camera' else 'step/tgt_act'][s]).to(act_dtype)
        eng.step_versus_greedy(team, mine, policy_tape=tape, tape_ct=bc(np.nan_to_num(fx['step/tape_ct'][s], nan=0.0)),
                               tape_goal=bc(np.nan_to_num(fx['step/goal_u'][s], nan=0.0)), auto_reset=False)
        assert eng.last_flow == 0                      # mixed encodings / tapes: the generic kernel
        cam_act, tgt_act = eng.policy_actions()        # the opponents' joint action is the recorded one
        theirs, key = (tgt_act, 'step/tgt_act') if team == 'camera' else (cam_act, 'step/cam_act')
        assert np.abs(theirs[1].cpu().numpy() - fx[key][s]).max() < tol, (key, s)
        sd = eng.state_dict()
        assert np.abs(sd['tgt_x'][0] - fx['step/tgt_xy'][s][:, 0]).max() < tol, s
        assert np.abs(sd['cam_phi'][1] - fx['step/cam_phi'][s]).max() < tol, s
        if exact:
            masks = eng.unpack_masks()
            assert np.array_equal(masks['camera_target_view_mask'][0], fx['step/camera_target_view_mask'][s]), s
            assert np.array_equal(sd['tgt_goals'][0], fx['step/tgt_goals'][s].astype(np.float64)), s
            assert np.array_equal(sd['bounties'][1], fx['step/bounties'][s].astype(np.float64)), s
            assert sd['episode_reward'][0] == fx['step/episode_reward'][s]


def test_single_team_versus_greedy_matches_step_greedy_on_philox():
    """On Philox draws: engine A steps both greedy teams; engine B is fed A's target joint action as the caller's team (f64)
    and A's camera joint action as grid indices is not possible, so: B1 = caller plays targets, B2 = caller plays cameras.
    All three must stay bit-identical, auto-resets included."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=60)
    engines = [Engine(cfg, 96, seed=11) for _ in range(3)]
    for e in engines:
        e.enable_policies()
        e.reset()
    a, b1, b2 = engines
    for s in range(150):
        a.step_greedy(auto_reset=True)
        cam_act, tgt_act = a.policy_actions()
        b1.step_versus_greedy('target', tgt_act, auto_reset=True)
        b2.step_versus_greedy('camera', cam_act, auto_reset=True)
        for b in (b1, b2):
            assert torch.equal(a.scalars, b.scalars), s
            assert torch.equal(a.target_obs, b.target_obs), s
            assert torch.equal(a.camera_obs, b.camera_obs), s
    assert (a.state_dict()['episode'] >= 2).all()


@pytest.mark.parametrize('team,encoding', [('camera', 'f32'), ('target', 'f64'), ('camera', 'grid'), ('target', 'grid')])
def test_frame_skip_rollout_versus_greedy_is_the_per_step_flow(team, encoding):
    """FrameSkip over MultiCamera / MultiTarget fused into one launch (mate_engine_rollout_versus_greedy) against
    frame_skip calls of mate_engine_step_versus_greedy with the same action: bit-identical rows, states, restarts."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=50)
    N, K = 70, 5
    a, b = (Engine(cfg, N, seed=23, obs_dtype=torch.float32) for _ in range(2))
    for e in (a, b):
        e.enable_policies()
        e.set_action_grids(5, 5)
        e.reset()
    agents = a.num_cameras if team == 'camera' else a.num_targets
    gen = torch.Generator(device='cpu').manual_seed(3)
    for it in range(24):
        if encoding == 'grid':
            act = torch.randint(0, 25, (N, agents), generator=gen, dtype=torch.int32).to(a.device)
        else:
            scale = 5.0 if team == 'camera' else 1000.0     # beyond the action box: clipping / step-size normalisation is exercised
            act = ((torch.rand((N, agents, 2), generator=gen, dtype=torch.float64) * 2 - 1) * scale).to(a.device)
            act = act.to(torch.float32 if encoding == 'f32' else torch.float64)
        cam, tgt, sc = a.rollout_versus_greedy(team, act, K, auto_reset=True)
        alive = torch.ones(N, dtype=torch.bool, device=a.device)
        for r in range(K):
            # batched auto-reset every K calls = the rollout's semantics: a finished environment idles (done = 2) until the
            # K-th call restarts all finished environments together
            b.step_versus_greedy(team, act, auto_reset=K)
            assert torch.equal(sc[r][:, :3], b.scalars[:, :3]), (it, r)       # reward, reward, done (2 = idle) of every environment
            assert torch.equal(sc[r][alive], b.scalars[alive]), (it, r)        # (an idle row keeps its last metrics in the per-step buffer)
            assert torch.equal(tgt[r][alive], b.target_obs[alive]), (it, r)
            assert torch.equal(cam[r][alive], b.camera_obs[alive]), (it, r)
            alive &= b.scalars[:, 2] == 0
        sa, sb = a.state_dict(), b.state_dict()
        for key in ('tgt_x', 'tgt_y', 'cam_phi', 'cam_theta', 'episode', 'episode_reward'):
            assert np.array_equal(sa[key], sb[key]), (it, key)
    assert (a.state_dict()['episode'] >= 2).any()


def test_greedy_rollout_on_philox_streams():
    """C3-style workload: 8v8 Greedy vs Greedy stays on the device; episodes finish and auto-reset."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    eng = Engine(read_config('MATE-8v8-9.yaml', max_episode_steps=400), 64, seed=4)
    eng.enable_policies()
    eng.reset()
    delivered = 0.0
    for _ in range(450):
        eng.step_greedy(auto_reset=True)
        delivered = max(delivered, float(eng.scalars[:, 6].max()))
    sd = eng.state_dict()
    assert delivered >= 4                      # greedy targets do deliver cargo
    assert (sd['episode'] >= 2).all()          # every environment finished an episode (time limit) and restarted
    assert float(eng.scalars[:, 3].mean()) > 0.2   # greedy cameras keep a sizeable coverage


@pytest.mark.parametrize('config,n,steps', [('MATE-8v8-9.yaml', 256, 200), ('MATE-4v8-9.yaml', 64, 120), ('MATE-Navigation.yaml', 32, 80)])
def test_greedy_policies_batch_vs_oracle(config, n, steps, oracle_lib):
    """The on-device policies against the oracle's restatement of the reference agents, closed loop, on a batch of
    independently reset environments with random draw tapes (every agent branch fires somewhere in the batch):
    joint actions to 1e-8, view masks / goals / bounties exact."""
    O = oracle_lib
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config(config)
    eng = Engine(cfg, n, seed=21, first_env_index=500, obs_dtype=torch.float32)
    eng.enable_policies()
    eng.reset()
    torch.cuda.synchronize()
    batch = O.OracleBatch(U.oracle_proto_from_config(cfg, O), n, seed=21, first_env_index=500)
    batch.reset(threads=4)
    Nc, Nt = eng.num_cameras, eng.num_targets
    envs = [batch.env(e) for e in range(n)]
    for e in range(n):
        for c in range(Nc):
            envs[e].set_lut(c, *eng.lut_read(e, c))
    agents = [O.GreedyPolicies() for _ in range(n)]
    rng = np.random.RandomState(5)
    reset_u = rng.random_sample((n, Nt, 2))
    dev = eng.device
    worst = 0.0
    for s in range(steps):
        t = {'camera_resample_u': rng.random_sample((n, max(Nc, 1))), 'camera_sample_u': rng.random_sample((n, max(Nc, 1), 2)),
             'camera_delay': rng.randint(6, 50, size=(n, max(Nc, 1), max(Nc, 1))).astype(np.int32),
             'target_choice_u': rng.random_sample((n, Nt)), 'target_resample_u': rng.random_sample((n, Nt)),
             'target_sample_u': rng.random_sample((n, Nt, 2)), 'target_reset_sample_u': reset_u}
        tape_ct, goal_u = rng.random_sample((n, max(Nc, 1), Nt)), rng.random_sample((n, Nt))
        eng.step_greedy(policy_tape={k: torch.from_numpy(v).to(dev) for k, v in t.items()}, tape_ct=torch.from_numpy(tape_ct).to(dev),
                        tape_goal=torch.from_numpy(goal_u).to(dev), auto_reset=False)
        cam_act, tgt_act = (a.cpu().numpy() for a in eng.policy_actions())
        for e in range(n):
            ca, ta = agents[e].act(envs[e], t['camera_resample_u'][e, :Nc], t['camera_sample_u'][e, :Nc], t['camera_delay'][e, :Nc, :Nc],
                                   t['target_choice_u'][e], t['target_resample_u'][e], t['target_sample_u'][e], reset_u[e])
            worst = max(worst, float(np.abs(ta - tgt_act[e]).max()), float(np.abs(ca - cam_act[e]).max()) if Nc else 0.0)
            envs[e].step(ca, ta, tape_ct[e, :Nc] if Nc else None, goal_u[e])
        assert worst < 1e-8, (s, worst)
        masks = eng.unpack_masks()
        if Nc:
            assert np.array_equal(masks['camera_target_view_mask'], batch.gather('camera_target_view_mask').reshape(n, Nc, Nt) != 0), s
        sd = eng.state_dict()
        for k in ('tgt_goals', 'bounties', 'freights', 'num_delivered_cargoes'):
            ref = batch.gather(k)
            assert np.array_equal(sd[k].reshape(ref.shape), ref), (k, s)
        assert np.abs(sd['tgt_x'] - batch.gather('tgt_x')).max() < 1e-8


@pytest.mark.parametrize('config,n,kw', [('MATE-4v8-9.yaml', 37, {}), ('MATE-8v8-9.yaml', 22, {}), ('MATE-4v2-9.yaml', 16, {}),
                                         ('MATE-4v8-9.yaml', 24, {'max_episode_steps': 20})])
def test_fused_greedy_rollout_equals_single_steps(config, n, kw):
    """rollout_greedy(K) == K x step_greedy(): observations, scalars, masks, joint actions, final state and agent memory,
    bit for bit -- including a batch size that is not a multiple of the workgroup's four environments, and episodes
    that end inside the rollout (time limit 20): those environments idle until the reset after the launch, exactly like
    the batched auto-reset of the single-step flow."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config(config, **kw)
    K = 12
    a = Engine(cfg, n, seed=9, first_env_index=11)
    b = Engine(cfg, n, seed=9, first_env_index=11)
    for e in (a, b):
        e.enable_policies()
        e.reset()
    rounds = 3 if kw else 2
    idled = False
    for rnd in range(rounds):
        cam_r, tgt_r, sc_r = a.rollout_greedy(K, auto_reset=True, want_masks=True)
        idle = (sc_r[:, :, 2] == 2)
        for r in range(K):
            b.step_greedy(auto_reset=10 ** 6)     # batched mode: finished environments idle (the reset comes below)
            live = ~idle[r]
            assert torch.equal(sc_r[r][live], b.scalars[live]), (rnd, r)
            assert torch.equal(tgt_r[r][live], b.target_obs[live]), (rnd, r)
            if a.num_cameras:
                assert torch.equal(cam_r[r][live], b.camera_obs[live]), (rnd, r)
            assert torch.equal(a._rollout['masks'][r][live], b.masks[live]), (rnd, r)
            assert bool((b.scalars[idle[r]][:, 2] == 2).all())
        idled = idled or bool(idle.any())
        # the launch after the rollout restarted the finished environments; do the same on the single-step engine
        done = torch.from_numpy(b.state_dict()['done'] != 0).to(b.device)
        if bool(done.any()):
            b.reset(env_mask=done)
        sa, sb = a.state_dict(), b.state_dict()
        for k in sa:
            assert np.array_equal(sa[k], sb[k]), (rnd, k)
    assert idled == bool(kw)         # the time-limit case did end episodes inside a rollout


def _twelve_cameras(**overrides):
    """MATE-8v8-9 with four more cameras (12: three rounds of (sender, recipient) pairs, three of sector pairs)."""
    from mate_amd.config import read_config
    cfg = read_config('MATE-8v8-9.yaml', **overrides)
    cfg['camera']['location_random_range'] = list(cfg['camera']['location_random_range']) + [
        [300.0, 400.0, 300.0, 400.0], [300.0, 400.0, -400.0, -300.0], [-400.0, -300.0, -400.0, -300.0], [-400.0, -300.0, 300.0, 400.0]]
    return cfg


def test_greedy_agents_with_more_than_eight_cameras_vs_oracle(oracle_lib):
    """Twelve camera agents (the message exchange then takes three rounds of (sender, recipient) pairs; rounds 1-3 lifted the engine's
    16-camera limit to the agents in round 4): closed loop against the oracle's restatement of the reference agents on recorded
    draws -- joint actions 1e-8, masks / goals / bounties exact -- and, on Philox draws, the fused rollout against single steps."""
    O = oracle_lib
    from mate_amd.engine import Engine
    cfg = _twelve_cameras()
    n, steps = 40, 90
    eng = Engine(cfg, n, seed=23, first_env_index=77, obs_dtype=torch.float32)
    assert eng.num_cameras == 12 and not eng.specialised
    eng.enable_policies()
    eng.reset()
    torch.cuda.synchronize()
    batch = O.OracleBatch(U.oracle_proto_from_config(cfg, O), n, seed=23, first_env_index=77)
    batch.reset(threads=4)
    Nc, Nt = eng.num_cameras, eng.num_targets
    envs = [batch.env(e) for e in range(n)]
    for e in range(n):
        for c in range(Nc):
            envs[e].set_lut(c, *eng.lut_read(e, c))
    agents = [O.GreedyPolicies() for _ in range(n)]
    rng = np.random.RandomState(6)
    reset_u = rng.random_sample((n, Nt, 2))
    dev = eng.device
    worst, messages = 0.0, 0
    for s in range(steps):
        t = {'camera_resample_u': rng.random_sample((n, Nc)), 'camera_sample_u': rng.random_sample((n, Nc, 2)),
             'camera_delay': rng.randint(6, 50, size=(n, Nc, Nc)).astype(np.int32),
             'target_choice_u': rng.random_sample((n, Nt)), 'target_resample_u': rng.random_sample((n, Nt)),
             'target_sample_u': rng.random_sample((n, Nt, 2)), 'target_reset_sample_u': reset_u}
        tape_ct, goal_u = rng.random_sample((n, Nc, Nt)), rng.random_sample((n, Nt))
        eng.step_greedy(policy_tape={k: torch.from_numpy(v).to(dev) for k, v in t.items()}, tape_ct=torch.from_numpy(tape_ct).to(dev),
                        tape_goal=torch.from_numpy(goal_u).to(dev), auto_reset=False)
        cam_act, tgt_act = (a.cpu().numpy() for a in eng.policy_actions())
        for e in range(n):
            ca, ta = agents[e].act(envs[e], t['camera_resample_u'][e], t['camera_sample_u'][e], t['camera_delay'][e],
                                   t['target_choice_u'][e], t['target_resample_u'][e], t['target_sample_u'][e], reset_u[e])
            worst = max(worst, float(np.abs(ta - tgt_act[e]).max()), float(np.abs(ca - cam_act[e]).max()))
            envs[e].step(ca, ta, tape_ct[e], goal_u[e])
        assert worst < 1e-8, (s, worst)
        masks = eng.unpack_masks()
        assert np.array_equal(masks['camera_target_view_mask'], batch.gather('camera_target_view_mask').reshape(n, Nc, Nt) != 0), s
        sd = eng.state_dict()
        for k in ('tgt_goals', 'bounties', 'freights', 'num_delivered_cargoes'):
            ref = batch.gather(k)
            assert np.array_equal(sd[k].reshape(ref.shape), ref), (k, s)
        assert np.abs(sd['tgt_x'] - batch.gather('tgt_x')).max() < 1e-8
    # Philox draws (a pair beyond the 64th takes its message delay from a block keyed by the pair's index): fused == single steps
    a, b = Engine(_twelve_cameras(), 22, seed=9), Engine(_twelve_cameras(), 22, seed=9)
    for e in (a, b):
        e.enable_policies()
        e.reset()
    for rnd in range(2):
        cam, tgt, sc = a.rollout_greedy(12, auto_reset=False)
        for r in range(12):
            b.step_greedy(auto_reset=False)
            assert torch.equal(cam[r], b.camera_obs) and torch.equal(tgt[r], b.target_obs) and torch.equal(sc[r], b.scalars), (rnd, r)
        assert torch.equal(a.export_state(), b.export_state())
        ca, ta = a.policy_actions()
        cb, tb = b.policy_actions()
        assert torch.equal(ca, cb) and torch.equal(ta, tb) and float(ca.abs().sum()) > 0.0


@pytest.mark.parametrize('switch,auto_reset', [('MATE_ZOOM_ITERATE', 1), ('MATE_POLICY_SPLIT', 1), ('MATE_POLICY_SPLIT', 4),
                                               ('MATE_STEP_GREEDY_ROLLOUT', 1), ('MATE_STEP_GREEDY_ROLLOUT', 4)])
def test_policy_implementation_switches_give_the_same_episodes(switch, auto_reset):
    """Two implementation choices of the on-device Greedy agents, selected at create by an environment switch:
    MATE_ZOOM_ITERATE=1 runs the reference's 20-iteration zoom solve (agents/greedy.py:139-145) instead of its tabulation --
    joint actions within 1e-10 degrees, the episodes (masks, goals, rewards) identical; MATE_POLICY_SPLIT=1 runs
    step_greedy / step_versus_greedy as two launches (agents' kernel, step kernel) instead of the fused one -- bit for bit,
    also with batched resets (auto_reset = 4: the idle rows of finished environments included); MATE_STEP_GREEDY_ROLLOUT=1 runs
    the one-launch form on the fused rollout kernel with a single step (rounds 3-4) instead of step_greedy_kernel (round 5):
    bit for bit as well."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-8v8-9.yaml', max_episode_steps=40)
    n = 192
    runs = []
    for value in ('0', '1'):
        os.environ[switch] = value
        try:
            eng = Engine(cfg, n, seed=17)
        finally:
            os.environ.pop(switch, None)
        eng.enable_policies()
        eng.reset()
        rec = []
        mine = torch.zeros((n, eng.num_targets, 2), device='cuda')
        idle_rows = 0
        for s in range(60):
            if s % 2:
                eng.step_greedy(auto_reset=auto_reset)
            else:
                mine.fill_(float(3 * (s % 5) - 6))
                eng.step_versus_greedy('target', mine, auto_reset=auto_reset)
            cam, tgt = eng.policy_actions()
            idle_rows += int((eng.scalars[:, 2] == 2).sum())
            rec.append((cam.clone(), eng.scalars.clone(), eng.masks.clone(), eng.target_obs.clone()))
        rec.append((eng.export_state().clone(),))
        runs.append(rec)
        assert (idle_rows > 0) == (auto_reset > 1)
    exact = switch != 'MATE_ZOOM_ITERATE'
    for a, b in zip(*runs):
        if exact:
            assert all(torch.equal(x.view(torch.uint8), y.view(torch.uint8)) for x, y in zip(a, b))
        else:
            assert (a[0] - b[0]).abs().max() <= 1e-10            # the camera agents' joint action
            for x, y in zip(a[1:], b[1:]):
                assert torch.allclose(x.double(), y.double(), rtol=0, atol=1e-6) and (x.dtype.is_floating_point or torch.equal(x, y))
    assert (runs[0][-1][0][:, -2] >= 2).all()                     # every environment went through an episode end


def test_policy_split_switch_under_the_graph_stepper_versus_greedy():
    """The learner-versus-greedy loop replayed from a HIP graph (Engine.make_stepper(versus=...)) with the one-launch and with
    the two-launch form of a step (MATE_POLICY_SPLIT), batched resets inside the graph: the same bytes."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=25)
    n, runs = 96, []
    for value in ('0', '1'):
        os.environ['MATE_POLICY_SPLIT'] = value
        try:
            eng = Engine(cfg, n, seed=5)
        finally:
            os.environ.pop('MATE_POLICY_SPLIT', None)
        eng.enable_policies()
        eng.reset()
        mine = torch.zeros((n, eng.num_cameras, 2), device='cuda')
        counter = torch.zeros((), device='cuda')

        def policy():
            counter.add_(1.0)
            mine.copy_((torch.sin(counter) * 4.0).expand_as(mine))

        stepper = eng.make_stepper(mine, None, auto_reset=4, graph_steps=8, between=policy, versus='camera')
        rec = []
        for _ in range(9):
            stepper.run(8)
            rec.append((eng.scalars.clone(), eng.masks.clone(), eng.camera_obs.clone(), eng.target_obs.clone()))
        stepper.close()
        rec.append((eng.export_state().clone(),))
        runs.append(rec)
    for a, b in zip(*runs):
        assert all(torch.equal(x.view(torch.uint8), y.view(torch.uint8)) for x, y in zip(a, b))
    assert (runs[0][-1][0][:, -2] >= 2).all()
