"""GPU parity of the on-device rule-based policies (row f1): GreedyCameraAgent vs GreedyTargetAgent run
closed-loop on the device with the reference agents' recorded draws must reproduce the reference's joint
actions and the resulting environment trace (fixtures greedy_*.npz, recorded by tests/golden/make_golden.py agents)."""
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


@pytest.mark.parametrize('config,n,steps', [('MATE-8v8-9.yaml', 48, 120), ('MATE-4v8-9.yaml', 64, 120), ('MATE-Navigation.yaml', 32, 80)])
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
