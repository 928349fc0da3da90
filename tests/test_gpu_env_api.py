"""GPU tests of the Python boundary: the single-environment NumPy API (drop-in for the reference
class) and the batched torch API."""
import numpy as np
import pytest
import torch

import golden_util as G
import gpu_util as U

pytestmark = pytest.mark.gpu


def test_single_env_reference_style_loop():
    """config[0] plumbing: MATE-4v2-9, one environment, random actions, evaluate-style loop."""
    import mate_amd
    env = mate_amd.make('MATE-4v2-9-v0')
    assert str(env).endswith('(4 cameras, 2 targets, 9 obstacles)')
    seeds = env.seed(3)
    assert seeds[0] == 3 and len(seeds) == 1 + 4 + 2 + 9     # main seed + one per entity (environment.py:1221-1227)
    cam_obs, tgt_obs = env.reset()
    assert cam_obs.shape == (4, 96) and tgt_obs.shape == (2, 101) and cam_obs.dtype == np.float64
    assert env.camera_observation_space.contains(cam_obs[0]) and env.target_observation_space.contains(tgt_obs[1])
    assert env.state().shape == env.state_space.shape == (13 + 9 * 4 + 14 * 2 + 3 * 9 + 2 * 2 + 16,)
    assert np.array_equal(cam_obs[:, 3], np.arange(4)) and np.array_equal(tgt_obs[:, 3], np.arange(2))
    assert (env.target_goals >= 0).all() and env.awaiting_cargo_counts.sum() == 8 * 2
    rng = np.random.RandomState(0)
    total = 0.0
    for step in range(60):
        action = (rng.uniform(-1, 1, (4, 2)) * [5.0, 2.5], rng.uniform(-20, 20, (2, 2)))
        (cam_obs, tgt_obs), (r_cam, r_tgt), done, (cam_infos, tgt_infos) = env.step(action)
        assert r_cam == -r_tgt and isinstance(done, bool) and len(cam_infos) == 4 and len(tgt_infos) == 2
        assert set(cam_infos[0]) >= {'raw_reward', 'normalized_raw_reward', 'messages', 'coverage_rate', 'real_coverage_rate',
                                     'mean_transport_rate', 'num_delivered_cargoes', 'out_communication_edges', 'in_communication_edges'}
        assert env.episode_step == step + 1
        total += r_tgt
        # the observation is the packed view of the state the attributes expose
        t0 = env.targets[0]
        assert np.allclose(tgt_obs[0, 13:15], t0.location) and tgt_obs[0, 16] == float(t0.is_loaded)
        assert np.allclose(cam_obs[1, 13:22], env.cameras[1].state(private=True))
        assert np.array_equal(env.camera_target_view_mask.any(axis=0), env.tracked_bits)
        assert env.coverage_rate == pytest.approx(env.tracked_bits.mean())
    assert env.target_team_episode_reward == pytest.approx(total)
    with pytest.raises(AssertionError):
        env.step((np.full((4, 2), np.nan), np.zeros((2, 2))))
    env.close()


def test_single_env_replays_golden_trace_without_obstacles():
    """MATE-4v8-0 has no obstacles and transmittance 0: no random draw on the step path, so the plain
    `env.step()` API must reproduce the reference trace from the injected reset state."""
    import mate_amd
    fx = G.load('trace_4v8-0_random_s0.npz')
    env = mate_amd.make('MATE-4v8-0-v0')
    env.reset()
    state = {k: v[None] for k, v in U.fixture_state(fx).items()}
    state.update(tick=np.zeros(1), episode=np.ones(1), done=np.zeros(1))
    env.engine.load_state_dict(state)
    env.engine.rebuild_luts()
    env._cache = env._masks = None
    env._last_goals = env.target_goals.copy()
    for s in range(len(fx['step/done'])):
        (cam_obs, tgt_obs), (r_cam, r_tgt), done, _ = env.step((fx['step/cam_act'][s], fx['step/tgt_act'][s]))
        assert np.allclose(cam_obs, fx['step/cam_obs'][s], rtol=0, atol=1e-9)
        assert np.allclose(tgt_obs, fx['step/tgt_obs'][s], rtol=0, atol=1e-9)
        assert r_tgt == fx['step/reward_tgt'][s] and done == bool(fx['step/done'][s])
        assert np.array_equal(env.camera_target_view_mask, fx['step/camera_target_view_mask'][s])
        assert np.array_equal(env.target_dones, fx['step/target_dones'][s])
        assert np.allclose(env.state(), fx['step/state'][s], rtol=0, atol=1e-9)
        assert np.allclose(env.target_warehouse_distances, fx['step/target_warehouse_distances'][s], rtol=0, atol=1e-9)
        if s % 16 == 0:      # the backend protocol of mate_amd.reference_adapter: one dict with everything, keyed like the fixtures
            snap = env.snapshot()
            for key in ('cam_phi', 'cam_theta', 'cam_sight', 'tgt_xy', 'target_warehouse_distances'):
                assert np.allclose(snap[key], fx['step/' + key][s], rtol=0, atol=1e-9), key
            for key in ('tgt_colliding', 'tgt_empty_bits', 'tgt_goal_bits', 'tgt_goals', 'freights', 'bounties', 'target_steps', 'tracked_steps',
                        'remaining_cargoes', 'awaiting_cargo_counts', 'num_delivered_cargoes', 'target_dones', 'camera_target_view_mask',
                        'target_camera_view_mask', 'target_target_view_mask', 'camera_camera_view_mask', 'tracked_bits'):
                assert np.array_equal(np.asarray(snap[key]), np.asarray(fx['step/' + key][s])), key
            assert snap['coverage_rate'] == fx['step/coverage_rate'][s] and snap['mean_transport_rate'] == fx['step/mean_transport_rate'][s]
            prev = fx['step/episode_reward'][s - 1] if s else 0.0
            assert snap['reward_dense'] == fx['step/episode_reward'][s] - prev and len(snap['luts']) == 4


def test_messaging_is_a_host_side_mailbox():
    import mate_amd
    from mate_amd import Message, Team
    env = mate_amd.make('MATE-4v2-9-v0')
    env.reset()
    env.send_messages(Message(sender=0, recipient=None, content={'k': 1}, team=Team.CAMERA))
    env.send_messages([Message(sender=1, recipient=0, content='x', team=Team.TARGET)])
    assert env.camera_communication_edges[0].sum() == 4 and env.target_communication_edges[1, 0] == 1
    got = env.receive_messages(agent_id=(Team.CAMERA, 2))
    assert len(got) == 1 and got[0].broadcasting and got[0].content == {'k': 1}
    cams, tgts = env.receive_messages()
    assert [len(q) for q in cams] == [1, 1, 0, 1] and [len(q) for q in tgts] == [1, 0]
    _, _, _, (cam_infos, tgt_infos) = env.step((np.zeros((4, 2)), np.zeros((2, 2))))
    assert len(cam_infos[3]['messages']) == 1 and cam_infos[0]['out_communication_edges'] == 4
    assert env.camera_communication_edges.sum() == 0 and env.camera_total_communication_edges.sum() == 4


def test_batched_env_and_sharding_invariance():
    """A shard (first_env_index = k) reproduces environments k.. of the full batch bit for bit."""
    from mate_amd.environment import BatchedMultiAgentTracking
    full = BatchedMultiAgentTracking('MATE-4v8-9.yaml', num_envs=24, seed=5)
    part = BatchedMultiAgentTracking('MATE-4v8-9.yaml', num_envs=8, seed=5, first_env_index=16)
    full.reset(); part.reset()
    for _ in range(12):
        (co_f, to_f), (rc_f, rt_f), done_f, info_f = full.step_random()
        (co_p, to_p), (rc_p, rt_p), done_p, info_p = part.step_random()
    assert torch.equal(co_f[16:], co_p) and torch.equal(to_f[16:], to_p) and torch.equal(rt_f[16:], rt_p)
    assert co_f.dtype == torch.float32 and co_f.shape == (24, 4, 126) and to_f.shape == (24, 8, 131)
    # explicit actions: zero actions leave the targets where they are
    before = full.state_dict()
    cam_act = torch.zeros((24, 4, 2), device=full.device)
    tgt_act = torch.zeros((24, 8, 2), device=full.device)
    full.step((cam_act, tgt_act))
    after = full.state_dict()
    assert np.array_equal(before['tgt_x'], after['tgt_x']) and np.array_equal(before['cam_phi'], after['cam_phi'])
    assert (after['episode_step'] == before['episode_step'] + 1).all()


def test_fused_rollout_equals_single_steps():
    """rollout_random(T) == T x step_random(): same observations, scalars, masks and final state."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-4v8-9.yaml')
    a = Engine(cfg, 37, seed=9)
    b = Engine(cfg, 37, seed=9)
    a.reset(); b.reset()
    T = 12
    cam_r, tgt_r, sc_r = a.rollout_random(T, auto_reset=False, want_masks=True)
    for r in range(T):
        b.step_random(auto_reset=False, want_masks=True)
        assert torch.equal(cam_r[r], b.camera_obs) and torch.equal(tgt_r[r], b.target_obs), r
        assert torch.equal(sc_r[r], b.scalars), r
        assert torch.equal(a._rollout['masks'][r], b.masks), r
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    # continue stepping after a rollout: ticks line up
    a.step_random(auto_reset=False); b.step_random(auto_reset=False)
    assert torch.equal(a.target_obs, b.target_obs)


def test_rollout_stops_at_episode_end_and_resets():
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    eng = Engine(read_config('MATE-4v8-9.yaml', max_episode_steps=5), 9, seed=2)
    eng.reset()
    _, _, sc = eng.rollout_random(10, auto_reset=True)
    done = sc[:, :, 2].cpu().numpy()
    assert (done[:5] == 0).all() and (done[5] == 1).all() and (done[6:] == 2).all()
    sd = eng.state_dict()
    assert (sd['episode'] == 2).all() and (sd['episode_step'] == 0).all() and (sd['done'] == 0).all()
    _, _, sc = eng.rollout_random(3, auto_reset=True)
    assert (sc[:, :, 2].cpu().numpy() == 0).all()


def test_auxiliary_camera_rewards_and_wrapper_kwargs():
    """Batched counterparts of the reference's wrapper stack: AuxiliaryCameraRewards terms from the step record and
    the packed masks; EnhancedObservation / SharedFieldOfView / Discrete* through constructor keywords."""
    import torch
    from mate_amd.environment import BatchedMultiAgentTracking
    env = BatchedMultiAgentTracking('MATE-4v8-9.yaml', num_envs=32, seed=3, enhanced_observation='target', shared_field_of_view='camera',
                                    discrete_camera_levels=5, discrete_target_levels=5)
    env.reset()
    g = torch.Generator(device='cpu').manual_seed(0)
    for _ in range(20):
        ci = torch.randint(0, 25, (32, 4), generator=g).cuda()
        ti = torch.randint(0, 25, (32, 8), generator=g).cuda()
        (co, to), (rc, rt), done, info = env.step((ci, ti))
    assert co.shape == (32, 4, 126) and to.shape == (32, 8, 131)
    # shared camera view: every camera row carries the same opponent block; enhanced targets see every camera
    opp = co[:, :, 22:22 + 40]
    assert torch.equal(opp, opp[:, :1].expand_as(opp))
    assert bool((to[:, :, 27:27 + 28].view(32, 8, 4, 7)[..., 6] == 1).all())
    m = env.masks()
    coef = {'raw_reward': 1.0, 'coverage_rate': 2.0, 'real_coverage_rate': 0.5, 'mean_transport_rate': -1.0, 'num_tracked': 0.25, 'baseline': 1}
    got = env.auxiliary_camera_rewards(coef).cpu().numpy()
    s = env.engine.scalars.double().cpu().numpy()
    want = (s[:, 0:1] + 2.0 * s[:, 3:4] + 0.5 * s[:, 4:5] - s[:, 5:6] + 1.0) + 0.25 * m['camera_target_view_mask'].sum(axis=2)
    assert np.allclose(got, want, rtol=0, atol=1e-12)
    shared = env.auxiliary_camera_rewards(coef, reduction='mean').cpu().numpy()
    assert np.allclose(shared, want.mean(axis=1, keepdims=True).repeat(4, axis=1), atol=1e-12)
    soft = env.auxiliary_camera_rewards({'soft_coverage_score': 2.0, 'baseline': 1.0}).cpu().numpy()   # builds the outer tables on first use
    matrix, scores = (x.cpu().numpy() for x in env.engine.soft_coverage())
    assert np.allclose(soft, 2.0 * scores + 1.0, atol=1e-12) and np.isfinite(matrix).all()
    seen = m['camera_target_view_mask']
    assert ((matrix > 0) == seen)[np.abs(matrix) > 1e-12].all()             # signed by the view mask
    assert (scores <= 8 + 1e-9).all() and (scores >= -1 - 1e-9).all()         # documented range [-1, Nt]
    with pytest.raises(AssertionError):
        env.auxiliary_camera_rewards({'bogus': 1.0})
    env.close()


def test_single_env_versus_greedy_follows_step_greedy():
    """N = 1 API: MultiTarget / MultiCamera with the greedy opponent on the device.  Two instances on the same seed, one stepping
    both greedy teams, the other fed the first one's joint action for its own team: identical trajectories."""
    import mate_amd
    a, b, c = (mate_amd.make('MATE-4v8-9-v0') for _ in range(3))
    for env in (a, b, c):
        env.enable_greedy_policies()
        env.reset(seed=21)
    for _ in range(40):
        (ca, ta), (rc, rt), done, _ = a.step_greedy()
        cam_act, tgt_act = (x[0].cpu().numpy() for x in a.engine.policy_actions())
        (cb, tb), (rcb, rtb), doneb, _ = b.step_versus_greedy('target', tgt_act)
        (cc, tc), (rcc, rtc), donec, _ = c.step_versus_greedy(mate_amd.Team.CAMERA, cam_act)
        for other_c, other_t, r in ((cb, tb, rtb), (cc, tc, rtc)):
            assert np.array_equal(np.asarray(ca), np.asarray(other_c)) and np.array_equal(np.asarray(ta), np.asarray(other_t))
            assert r == rt
    with pytest.raises(RuntimeError):
        mate_amd.make('MATE-4v8-9-v0').step_versus_greedy('camera', np.zeros((4, 2)))
    for env in (a, b, c):
        env.close()


def test_single_team_training_flow_of_the_examples():
    """The call pattern of examples/ippo/target/config.py on the batch: MultiTarget(env, GreedyCameraAgent) ->
    AuxiliaryTargetRewards -> FrameSkip, i.e. the learner plays the targets (grid indices here: DiscreteTarget), the greedy
    cameras run on the device, shaped per-target rewards come from the state, and k frames fuse into one launch."""
    import torch
    from mate_amd.environment import BatchedMultiAgentTracking
    N = 48
    env = BatchedMultiAgentTracking('MATE-4v8-9.yaml', num_envs=N, seed=5, discrete_target_levels=5, auto_reset=4, max_episode_steps=40)
    env.enable_greedy_policies()
    env.reset()
    coef = {'raw_reward': 1.0, 'normalized_goal_distance': -1.0, 'sparse_delivery': 10.0, 'is_tracked': -0.5, 'is_colliding': -1.0,
            'soft_coverage_score': -0.25}
    g = torch.Generator(device='cpu').manual_seed(1)
    episodes_before = env.state_dict()['episode'].copy()
    for it in range(60):
        act = torch.randint(0, 25, (N, 8), generator=g, dtype=torch.int32).cuda()
        (co, to), (rc, rt), done, info = env.step_versus_greedy('target', act)
        shaped = env.auxiliary_target_rewards(coef)
        assert shaped.shape == (N, 8) and bool(torch.isfinite(shaped).all())
        terms = env._target_shapers[next(iter(env._target_shapers))].terms
        assert bool(((terms['normalized_goal_distance'] >= 0) & (terms['normalized_goal_distance'] <= 2 ** 0.5)).all())
        assert bool(((terms['soft_coverage_score'] >= -1 - 1e-9) & (terms['soft_coverage_score'] <= 4 + 1e-9)).all())   # [-1, Nc]
        tracked = torch.from_numpy(env.masks()['tracked_bits']).cuda()
        assert torch.equal(terms['is_tracked'] != 0, tracked)
    assert (env.state_dict()['episode'] > episodes_before).all()          # time limit 40: every environment restarted (batched, every 4th call)
    (co, to), (rc, rt), done, info = env.rollout_versus_greedy('target', act, 4)     # FrameSkip(4)
    assert rt.shape == (4, N) and to.shape[:2] == (4, N)
    with pytest.raises(RuntimeError):
        BatchedMultiAgentTracking('MATE-4v8-9.yaml', num_envs=4).step_versus_greedy('camera', torch.zeros(4, 4, 2).cuda())
    env.close()


def test_boundary_between_inner_and_outer():
    """Camera.boundary_between of the N=1 API (entities.py:513-543) on the reference's geometry: the knots inside a
    sector from the device-built tables, inner and (lazily enabled) outer, follow the reference's own tables."""
    import mate_amd
    env = mate_amd.make('MATE-4v8-9-v0')
    env.seed(3)
    env.reset()
    cam = env.cameras[0]
    left, right = cam.orientation - cam.viewing_angle / 2.0, cam.orientation + cam.viewing_angle / 2.0
    for outer in (False, True):
        phis, rhos = cam.boundary_between(left, right, outer=outer)
        table_p, table_r = env.engine.lut_read(0, 0, outer=outer)
        assert phis[0] < phis[-1] and len(phis) >= 2 and np.all(np.diff(phis[1:-1]) > 0)
        inside = (table_p > phis[0]) & (table_p < phis[-1]) if phis[-1] <= 180.0 else None
        if inside is not None:
            assert np.array_equal(phis[1:-1], table_p[inside]) and np.array_equal(rhos[1:-1], table_r[inside])
        assert rhos[0] == cam.sight_range_at(phis[0]) and rhos[-1] == cam.sight_range_at(phis[-1])   # inner table at the ends
        assert np.all(rhos >= 0.0) and np.all(rhos <= cam.max_sight_range + 1e-9)
    outer_r = np.interp(np.linspace(-180, 179, 360), *env.engine.lut_read(0, 0, outer=True))
    inner_r = np.interp(np.linspace(-180, 179, 360), *env.engine.lut_read(0, 0))
    assert np.all(outer_r >= inner_r - 1e-9)          # the far side of an obstacle is never nearer than its near side
    env.close()


def test_batched_env_rollouts():
    """BatchedMultiAgentTracking.rollout_random / rollout_greedy: step()'s result with a leading [steps] axis."""
    import torch
    from mate_amd.environment import BatchedMultiAgentTracking
    env = BatchedMultiAgentTracking('MATE-4v8-9.yaml', num_envs=48, seed=2)
    env.enable_greedy_policies()
    env.reset()
    (co, to), (rc, rt), done, info = env.rollout_random(6)
    assert co.shape == (6, 48, 4, 126) and to.shape == (6, 48, 8, 131) and rc.shape == (6, 48) and done.shape == (6, 48)
    assert torch.equal(rc, -rt) and not bool(done.any()) and not bool(info['skipped'].any())
    (co, to), (rc, rt), done, info = env.rollout_greedy(6)
    assert co.shape == (6, 48, 4, 126) and bool(torch.isfinite(co).all()) and float(info['coverage_rate'].mean()) > 0.0
    other = BatchedMultiAgentTracking('MATE-4v8-9.yaml', num_envs=48, seed=2)
    other.reset()
    with pytest.raises(RuntimeError):
        other.rollout_greedy(2)
    env.close(); other.close()


def test_seed_makes_episodes_reproducible():
    """environment.py:1203-1227: seeding re-creates the generators, so the same seed gives the same episode whatever ran
    before -- `reset(seed=s)` twice on one instance, and on a fresh instance; a different seed gives another episode."""
    import mate_amd
    env = mate_amd.make('MATE-4v8-9-v0')

    def episode(e, seed, steps=12):
        rng = np.random.RandomState(9)
        obs = [e.reset(seed=seed)]
        for _ in range(steps):
            action = (rng.uniform(-1, 1, (4, 2)) * [5.0, 2.5], rng.uniform(-20, 20, (8, 2)))
            (c, t), (_, r), done, _ = e.step(action)
            obs.append((c, t))
        return obs, e.state()

    a, sa = episode(env, 7)
    env.reset()                       # an unseeded episode in between advances every counter
    env.step((np.zeros((4, 2)), np.zeros((8, 2))))
    b, sb = episode(env, 7)
    for (c0, t0), (c1, t1) in zip(a, b):
        assert np.array_equal(c0, c1) and np.array_equal(t0, t1)
    assert np.array_equal(sa, sb)
    fresh = mate_amd.make('MATE-4v8-9-v0')
    c, sc = episode(fresh, 7)
    assert all(np.array_equal(x[1], y[1]) for x, y in zip(a, c)) and np.array_equal(sa, sc)
    d, sd = episode(fresh, 8)
    assert not np.array_equal(a[0][1], d[0][1])
    # the batched API: seed() rewinds the whole batch
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    eng = Engine(read_config('MATE-4v8-9.yaml'), 32, seed=5)
    eng.reset(); eng.rollout_random(9, auto_reset=True)
    first = eng.export_state().clone()
    eng.step_random(auto_reset=True)
    eng.seed(5)
    eng.reset(); eng.rollout_random(9, auto_reset=True)
    assert torch.equal(first, eng.export_state())


@pytest.mark.parametrize('name', ['trace_4v8-9_greedy_s2', 'trace_4v2-9_greedy_s1', 'trace_4v8-0_greedy_s1'])
def test_evaluate_harness_reproduces_a_reference_trace(name):
    """mate_amd.evaluate.evaluate (the counterpart of mate/evaluate.py:85-167) over a golden trace: with the recorded
    joint actions and draws it must report, step by step, the reference's coverage rate, mean transport rate, delivered
    cargoes, step reward and episode reward, in f64, with the reference's status keys."""
    import mate_amd
    from mate_amd.evaluate import COLUMNS, evaluate
    fx = G.load(name + '.npz')
    env = mate_amd.MultiAgentTracking(str(fx['config_file']))
    real_reset = env.reset

    def reset_into_fixture():
        real_reset()
        state = {k: v[None] for k, v in U.fixture_state(fx).items()}
        state.update(tick=np.zeros(1), episode=np.ones(1), done=np.zeros(1))
        env.engine.load_state_dict(state)
        for c, (phis, rhos) in enumerate(G.luts_of(fx)):
            env.engine.lut_write(0, c, phis, rhos)
        env._cache = env._masks = None
        env._last_goals = env.target_goals.copy()
        env._last_episode_rewards = (env.target_team_episode_reward, env.delayed_target_team_episode_reward)
        return env.joint_observation()

    env.reset = reset_into_fixture
    T = len(fx['step/done'])
    cursor = iter(range(T))

    def recorded_policy(e, observations, infos):
        s = next(cursor)
        e.step_tape = {'camera_target': fx['step/tape_ct'][s] if e.num_cameras else None, 'goal': fx['step/goal_u'][s]}
        return fx['step/cam_act'][s], fx['step/tgt_act'][s]

    env.config['max_episode_steps'] = T          # the trace is a prefix of an episode
    history = []
    status = evaluate(env, recorded_policy, history=history)
    assert len(history) == T and list(history[0]) == list(COLUMNS)
    episode_reward = np.cumsum(fx['step/reward_tgt'])
    for s, row in enumerate(history):
        assert row['Step'] == s + 1 and row['Cargo'] == int(fx['step/num_delivered_cargoes'][s])
        assert row['Reward'] == fx['step/reward_tgt'][s] and row['Target Episode Reward'] == episode_reward[s]
        assert row['Mean Transport Rate'] == fx['step/mean_transport_rate'][s]          # f64, exact
        assert row['Mean Coverage Rate'] == pytest.approx(np.mean(fx['step/coverage_rate'][:s + 1]), abs=1e-15)
        assert row['Normalized Target Episode Reward'] == episode_reward[s] / float(fx['max_target_team_episode_reward'])
        assert env.real_coverage_rate is not None
    delivered = fx['step/num_delivered_cargoes']
    if delivered[-1] > 0:
        assert status['Cargo'] == int(delivered[-1]) and status['Step'] == T
        assert status['Step / Cargo'] == T / delivered[-1]
    assert status == {} or set(status) == set(COLUMNS)


def test_evaluate_config1_with_on_device_greedy_agents():
    """BASELINE config 1 (MATE-4v2-9, one environment): an evaluate-style episode with both teams played by the on-device
    Greedy agents; the status row carries the reference's columns and is reproducible from the seed."""
    import mate_amd
    from mate_amd.evaluate import COLUMNS, evaluate, random_policy
    rows = []
    for _ in range(2):
        env = mate_amd.MultiAgentTracking('MATE-4v2-9.yaml', max_episode_steps=400)
        env.enable_greedy_policies()
        env.seed(0)
        history = []
        status = evaluate(env, history=history)
        rows.append([(r['Cargo'], r['Reward'], r['Mean Coverage Rate']) for r in history])
        assert set(status) == set(COLUMNS) and status['Step'] == len(history) <= 400 and status['FPS'] > 0
        assert status['Cargo'] >= 1                      # Greedy targets deliver within 400 steps
        assert status['Target Episode Reward'] == pytest.approx(sum(r['Reward'] for r in history))
        env.close()
    assert rows[0] == rows[1]
    env = mate_amd.MultiAgentTracking('MATE-4v2-9.yaml', max_episode_steps=50)
    history = []
    status = evaluate(env, random_policy(3), history=history)
    # the harness stops at max_episode_steps itself (evaluate.py:117), one call before the environment's own time-limit
    # `done` (environment.py:629-632); without a delivery the reference's status stays empty
    assert len(history) == 50 and env.episode_step == 50
    assert status == {} or status['Cargo'] > 0
