"""Pins the CPU oracle (oracle/mate_oracle.c) to golden vectors recorded from the
upstream Python reference (tests/golden/make_golden.py).  CPU only.

Tolerances: integer state, masks, rewards, done: exact.  f64 quantities: 1e-9
absolute on O(1e3) coordinates (observed <= 3e-12; the residue is libm vs numpy
SIMD sin/cos/atan2 last-place differences accumulated along a trace).
"""
import os

import numpy as np
import pytest

import golden_util as G
from oracle import oracle as O

F64_TOL = 1e-9


def test_kat_obstruct():
    rows = G.load('kat_obstruct.npz')['rows']
    exact = 0
    for r in rows:
        out = O.obstruct(r[0:2], r[2:4], r[4:6], r[6], bool(r[7]), bool(r[8]))
        np.testing.assert_allclose(out, r[9:11], rtol=0, atol=1e-12)
        exact += np.array_equal(out, r[9:11])
    assert exact >= 0.99 * len(rows)  # the rest differ in the last place through atan2/sincos


def test_kat_scalar():
    k = G.load('kat_scalar.npz')
    got = np.array([O.normalize_angle(a) for a in k['angles']])
    assert np.array_equal(got, k['normalized'])  # utils.py:155-158, pure arithmetic: bit exact
    cl = np.array([O.clamp_step(a[0], a[1], v) for a, v in zip(k['clamp_action'], k['clamp_step'])])
    np.testing.assert_allclose(cl, k['clamp_out'], rtol=0, atol=1e-12)
    cs = np.array([O.camera_simulate(r[0], r[1], r[2], r[3], 30.0, 1500.0, 5.0, 2.5) for r in k['cam_sim']])
    assert np.array_equal(cs, k['cam_sim'][:, 4:7])  # entities.py:347-360: bit exact


def test_kat_lut_builder_and_perceive():
    k = G.load('kat_perceive.npz')
    luts = []
    for i in range(len(k['lut_count'])):
        n, no = int(k['lut_count'][i]), int(k['num_obstacles'][i])
        phis, rhos = O.build_lut(k['cam_xy'][i], k['cam_max_sight_range'][i], k['obstacles'][i, :no], tau=0.0)
        assert len(phis) == n
        np.testing.assert_allclose(phis, k['lut_phis'][i, :n], rtol=0, atol=1e-10)
        # A ray exactly tangent to an obstacle is clipped or not depending on the last
        # bit of asin/atan2 (entities.py:170: `radius > perpendicular`): at most the two
        # tangent knots per obstacle may land on the other side of that coin flip.
        # ... and ONLY those: every other knot agrees to 1e-8, a differing one sits within 0.011 degrees of a tangent direction
        G.assert_only_tangent_flips(phis, rhos, k['lut_rhos'][i, :n], k['cam_xy'][i], float(k['cam_max_sight_range'][i]), k['obstacles'][i, :no], 1e-8, ('kat', i))
        luts.append((phis, rhos))
    for r in k['cases']:
        i = int(r[0])
        n = int(k['lut_count'][i])
        seen = O.camera_perceive(k['cam_xy'][i], r[1], r[2], r[3], r[4:6], r[6], r[7], k['lut_phis'][i, :n], k['lut_rhos'][i, :n])
        assert seen == bool(r[8])


@pytest.mark.parametrize('path', G.trace_files(), ids=lambda p: os.path.basename(p)[6:-4])
def test_trace_parity(path):
    fx = G.load(path)
    env = G.oracle_from_fixture(fx)
    co, to = env.observe()
    if co.size:
        np.testing.assert_allclose(co, fx['reset/cam_obs'], rtol=0, atol=F64_TOL)
    np.testing.assert_allclose(to, fx['reset/tgt_obs'], rtol=0, atol=F64_TOL)
    np.testing.assert_allclose(env.state(), fx['reset/state'], rtol=0, atol=F64_TOL)
    exact_keys = ['tgt_colliding', 'tgt_empty_bits', 'tgt_goal_bits', 'tgt_goals', 'freights', 'bounties', 'target_steps',
                  'tracked_steps', 'remaining_cargoes', 'awaiting_cargo_counts', 'num_delivered_cargoes', 'episode_step']
    fields = dict(G.DYNAMIC_FIELDS)
    for s in range(len(fx['step/done'])):
        env.step(fx['step/cam_act'][s], fx['step/tgt_act'][s], fx['step/tape_ct'][s], fx['step/goal_u'][s])
        for m in G.MASK_FIELDS + ['target_dones']:
            got = np.asarray(env.get(m)) != 0
            assert np.array_equal(got, fx['step/' + m][s].astype(bool)), (m, s)
        for key in exact_keys:
            got = np.asarray(env.get(fields[key]), dtype=np.float64)
            assert np.array_equal(got, np.asarray(fx['step/' + key][s], dtype=np.float64)), (key, s)
        assert env.get('reward_cam') == fx['step/reward_cam'][s]
        assert env.get('reward_tgt') == fx['step/reward_tgt'][s]
        assert env.get('normalized_reward_tgt') == fx['step/normalized_reward_tgt'][s]
        assert bool(env.get('done')) == bool(fx['step/done'][s])
        assert env.get('episode_reward') == fx['step/episode_reward'][s]
        assert env.get('delayed_episode_reward') == fx['step/delayed_episode_reward'][s]
        for k in ('coverage_rate', 'real_coverage_rate', 'mean_transport_rate'):
            assert env.get(k) == fx['step/' + k][s], (k, s)
        co, to = env.observe()
        if co.size:
            np.testing.assert_allclose(co, fx['step/cam_obs'][s], rtol=0, atol=F64_TOL)
        np.testing.assert_allclose(to, fx['step/tgt_obs'][s], rtol=0, atol=F64_TOL)
        np.testing.assert_allclose(env.state(), fx['step/state'][s], rtol=0, atol=F64_TOL)
        np.testing.assert_allclose(env.get('target_warehouse_distances'), fx['step/target_warehouse_distances'][s], rtol=0, atol=F64_TOL)
        for key in ('cam_phi', 'cam_theta', 'cam_sight'):
            np.testing.assert_allclose(env.get(key), fx['step/' + key][s], rtol=0, atol=F64_TOL)


@pytest.mark.parametrize('path', G.reset_files(), ids=lambda p: os.path.basename(p)[6:-4])
def test_reset_tape_parity(path):
    """reset() (environment.py:679-834) pinned to the reference: the oracle consumes the uniforms the reference's RNG
    proxies logged (shuffles, capacities, rejection-sampled placement, cargo matrix, initial goals) and must arrive
    at the state the reference arrived at -- placements and integers exact, occlusion tables knot for knot."""
    import gpu_util as U
    fx = G.load(path)
    cfg = G.config_of_reset_fixture(fx)
    env = U.oracle_proto_from_config(cfg, O)
    used = env.reset_tape(fx['tape'], fx['tape_ct'])
    assert used == len(fx['tape']), (used, len(fx['tape']))
    want = G.reset_expectation(fx)
    for key, ref in want.items():
        got = np.asarray(env.get(key), dtype=np.float64)
        assert np.array_equal(got.reshape(ref.shape), ref), (key, got, ref)
    Nc, No = int(fx['num_cameras']), int(fx['num_obstacles'])
    if Nc:
        assert np.array_equal(env.get('cam_sight'), fx['reset/cam_sight'])
    flips = 0
    for c, (phis, rhos) in enumerate(G.luts_of(fx)):
        p2, r2 = env.get_lut(c)
        assert len(p2) == len(phis), (c, len(p2), len(phis))
        np.testing.assert_allclose(p2, phis, rtol=0, atol=1e-10)
        # tangent-ray coin flips of the reference itself: at a tangent direction or nowhere
        flips += G.assert_only_tangent_flips(phis, r2, rhos, fx['static/cam_xy'][c], float(fx['static/cam_max_sight_range'][c]), fx['static/obs_xyr'], 1e-8, c)
    for m in G.MASK_FIELDS:
        assert np.array_equal(np.asarray(env.get(m)) != 0, fx['reset/' + m].astype(bool)), m
    co, to = env.observe()
    if co.size:
        np.testing.assert_allclose(co, fx['reset/cam_obs'], rtol=0, atol=F64_TOL)
    np.testing.assert_allclose(to, fx['reset/tgt_obs'], rtol=0, atol=F64_TOL)
    np.testing.assert_allclose(env.state(), fx['reset/state'], rtol=0, atol=F64_TOL)
    for k in ('coverage_rate', 'real_coverage_rate', 'mean_transport_rate'):
        assert env.get(k) == fx['reset/' + k], k
    # a tape that is one draw short is reported, not read past
    if len(fx['tape']) > 1:
        assert U.oracle_proto_from_config(cfg, O).reset_tape(fx['tape'][:-1], fx['tape_ct']) == -1
        assert U.oracle_proto_from_config(cfg, O).reset_tape(fx['tape'][:30], fx['tape_ct']) == -1      # and terminates


def test_trace_parity_with_own_lut():
    """Same replay, but the occlusion LUT comes from the oracle's own builder (a10)."""
    fx = G.load('trace_4v8-9_greedy_s2.npz')
    env = G.oracle_from_fixture(fx, use_golden_lut=False)
    for c, (phis, rhos) in enumerate(G.luts_of(fx)):
        p2, r2 = env.get_lut(c)
        assert len(p2) == len(phis)
        G.assert_only_tangent_flips(phis, r2, rhos, fx['static/cam_xy'][c], float(fx['static/cam_max_sight_range'][c]), fx['static/obs_xyr'], 1e-8, c)
    mism = 0
    for s in range(len(fx['step/done'])):
        env.step(fx['step/cam_act'][s], fx['step/tgt_act'][s], fx['step/tape_ct'][s], fx['step/goal_u'][s])
        mism += int(not np.array_equal(env.get('camera_target_view_mask') != 0, fx['step/camera_target_view_mask'][s]))
    assert mism == 0


def test_philox_known_answers():
    """The RNG shared by the oracle and the HIP engine is the published Philox-4x32-10 (Salmon et al., SC'11):
    the three known-answer vectors of Random123's kat_vectors file."""
    kat = [((0, 0), (0, 0, 0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff, 0xffffffff), (0xffffffff,) * 4, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0xa4093822, 0x299f31d0), (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for key, ctr, want in kat:
        assert tuple(O.philox(*key, *ctr)) == want


@pytest.mark.parametrize('name,teams', [('obsmode_4v8-9_s4', ('both', 'camera', 'target')), ('obsmode_4v8-9_fewcargo', ('both',)),
                                        ('obsmode_nav_s2', ('target',))])
def test_observation_wrappers(name, teams):
    """EnhancedObservation / SharedFieldOfView restated in the oracle == the reference's wrappers (f32 fixtures)."""
    fx = G.load(name + '.npz')
    env = G.oracle_from_fixture(fx)

    def check(prefix, s=None):
        for mode in ('enhanced', 'shared'):
            for team in teams:
                kw = {side: mode for side in (('camera', 'target') if team == 'both' else (team,))}
                co, to = env.observe_mode(**kw)
                ref_c, ref_t = fx[prefix + 'cam_obs_' + mode + '_' + team], fx[prefix + 'tgt_obs_' + mode + '_' + team]
                if s is not None:
                    ref_c, ref_t = ref_c[s], ref_t[s]
                if co.size:
                    assert np.array_equal(co.astype(np.float32), ref_c), (mode, team, s)
                assert np.array_equal(to.astype(np.float32), ref_t), (mode, team, s)

    check('reset/')
    for s in range(len(fx['step/done'])):
        env.step(fx['step/cam_act'][s], fx['step/tgt_act'][s], fx['step/tape_ct'][s], fx['step/goal_u'][s])
        check('step/', s)


@pytest.mark.parametrize('name', ['discrete_4v8-9_s6', 'discrete_4v2-9_s7'])
def test_discrete_action_decode(name):
    """Grids as the host builds them == the reference's; decode == the wrappers' continuous actions, bit for bit."""
    from mate_amd.spaces import camera_action_grid, target_action_grid
    fx = G.load(name + '.npz')
    lc, lt = (int(v) for v in fx['discrete_levels'])
    assert np.array_equal(camera_action_grid(lc), fx['camera_action_grid'])
    assert np.array_equal(target_action_grid(lt), fx['target_action_grid'])
    with pytest.raises(AssertionError):
        camera_action_grid(4)
    env = G.oracle_from_fixture(fx)
    for s in range(len(fx['step/done'])):
        ca, ta = env.decode_discrete(fx['step/cam_idx'][s], fx['camera_action_grid'], fx['step/tgt_idx'][s], fx['target_action_grid'])
        assert np.array_equal(ca, fx['step/cam_act'][s]) and np.array_equal(ta, fx['step/tgt_act'][s]), s
        env.step(ca, ta, fx['step/tape_ct'][s], fx['step/goal_u'][s])
        co, to = env.observe()
        np.testing.assert_allclose(co, fx['step/cam_obs'][s], rtol=0, atol=F64_TOL)
        np.testing.assert_allclose(to, fx['step/tgt_obs'][s], rtol=0, atol=F64_TOL)


@pytest.mark.parametrize('name', ['greedy_4v8-9_s5', 'greedy_8v8-9_s6', 'greedy_4v2-9_s7'])
def test_greedy_agents_closed_loop(name):
    """GreedyCameraAgent / GreedyTargetAgent restated in the oracle, run closed-loop with the reference agents'
    recorded draws: the reference's joint actions (1e-9) and its whole environment trace come back."""
    fx = G.load(name + '.npz')
    env = G.oracle_from_fixture(fx)
    env.set('camera_target_view_mask', fx['reset/camera_target_view_mask'].astype(np.float64))   # the view reset() left
    agents = O.GreedyPolicies()
    for s in range(len(fx['step/done'])):
        ca, ta = agents.act(env, fx['step/agent_cam_binom_u'][s], fx['step/agent_cam_sample_u'][s], fx['step/agent_cam_delay'][s],
                            fx['step/agent_tgt_choice_u'][s], fx['step/agent_tgt_binom_u'][s], fx['step/agent_tgt_sample_u'][s],
                            fx['agent/tgt_reset_sample_u'])
        if ca.size:
            assert np.abs(ca - fx['step/cam_act'][s]).max() < 1e-9, ('camera action', s)
        assert np.abs(ta - fx['step/tgt_act'][s]).max() < 1e-9, ('target action', s)
        env.step(ca, ta, fx['step/tape_ct'][s], fx['step/goal_u'][s])
        for m in G.MASK_FIELDS:
            assert np.array_equal(np.asarray(env.get(m)) != 0, fx['step/' + m][s].astype(bool)), (m, s)
        assert np.array_equal(np.asarray(env.get('tgt_goals'), dtype=np.float64), np.asarray(fx['step/tgt_goals'][s], dtype=np.float64)), s
        assert env.get('episode_reward') == fx['step/episode_reward'][s]


@pytest.mark.parametrize('name', ['trace_4v8-9_greedy_s2', 'trace_8v8-9_random_s0', 'trace_4v2-9_random_s0', 'trace_4v8-0_random_s0'])
def test_outer_boundary_builder(name):
    """Camera.boundary_outer / sight_range_outer_func (entities.py:419-448, 479) from the oracle's own builder against
    the tables the reference built for the same geometry (same tolerance as the inner table: tangent-ray coin flips)."""
    fx = G.load(name + '.npz')
    env = G.oracle_from_fixture(fx, use_golden_lut=False)
    No = int(fx['num_obstacles'])
    for c, (phis, rhos) in enumerate(G.luts_of(fx, outer=True)):
        p2, r2 = env.get_lut(c, outer=True)
        assert len(p2) == len(phis), (c, len(p2), len(phis))
        np.testing.assert_allclose(p2, phis, rtol=0, atol=1e-9)
        assert (np.abs(r2 - rhos) > 1e-8).sum() <= 2 * No, (c, (np.abs(r2 - rhos) > 1e-8).sum())


def aux_reward_terms(fx, s, soft):
    """The terms of AuxiliaryCameraRewards.step (wrappers/auxiliary_camera_rewards.py:141-150) from a fixture step."""
    seen = fx['step/camera_target_view_mask'][s].astype(bool)
    Nc = seen.shape[0]
    return {'raw_reward': np.full(Nc, float(fx['step/reward_cam'][s])), 'coverage_rate': np.full(Nc, float(fx['step/coverage_rate'][s])),
            'real_coverage_rate': np.full(Nc, float(fx['step/real_coverage_rate'][s])),
            'mean_transport_rate': np.full(Nc, float(fx['step/mean_transport_rate'][s])),
            'soft_coverage_score': np.asarray(soft, dtype=np.float64), 'num_tracked': seen.sum(axis=1).astype(np.float64), 'baseline': np.ones(Nc)}


@pytest.mark.parametrize('name', ['softcov_4v8-9_s8', 'softcov_8v8-9_s9', 'softcov_4v2-9_s10'])
def test_soft_coverage_score(name):
    """AuxiliaryCameraRewards' soft coverage score restated in the oracle == the reference wrapper's matrix, per-camera
    scores and shaped rewards on a recorded trace (1e-9: libm vs NumPy sin/cos/hypot last places)."""
    fx = G.load(name + '.npz')
    env = G.oracle_from_fixture(fx)
    keys, coef, reduction = [str(k) for k in fx['aux_keys']], fx['aux_coefficients'], str(fx['aux_reduction'])
    for s in range(len(fx['step/done'])):
        env.step(fx['step/cam_act'][s], fx['step/tgt_act'][s], fx['step/tape_ct'][s], fx['step/goal_u'][s])
        matrix, scores = env.soft_coverage()
        np.testing.assert_allclose(matrix, fx['step/soft_coverage_matrix'][s], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(scores, fx['step/soft_coverage_score'][s], rtol=1e-9, atol=1e-9)
        terms = aux_reward_terms(fx, s, scores)
        shaped = sum(c * terms[k] for k, c in zip(keys, coef))
        if reduction != 'none':
            shaped = np.full_like(shaped, {'mean': np.mean, 'sum': np.sum, 'max': np.max, 'min': np.min}[reduction](shaped))
        np.testing.assert_allclose(shaped, fx['step/aux_reward_cam'][s], rtol=1e-9, atol=1e-9)


def _chain_rows(team, rows, Nc, Nt, No):
    """RelativeCoordinates then RescaledObservation of one team's rows (mate/agents/utils.py:40-137), restated in NumPy for this test:
    coordinates of the warehouses and of VISIBLE entities minus the row owner's location; then (v - low) on every column bounded
    below and 2 (v - low) / (high - low) - 1 on every column bounded on both sides."""
    from mate_amd import constants as consts
    rows = np.array(rows, dtype=np.float64)
    if team == 'camera':
        space, cmask = consts.camera_observation_space_of(Nc, Nt, No), consts.camera_coordinate_mask_of(Nc, Nt, No)
        first, blocks = consts.PRESERVED_DIM + consts.CAMERA_STATE_DIM_PRIVATE, ((Nt, 5), (No, 4), (Nc, 7))
    else:
        space, cmask = consts.target_observation_space_of(Nc, Nt, No), consts.target_coordinate_mask_of(Nc, Nt, No)
        first, blocks = consts.PRESERVED_DIM + consts.TARGET_STATE_DIM_PRIVATE, ((Nc, 7), (No, 4), (Nt, 5))
    cmask = np.array(cmask, dtype=bool)
    low, high = np.asarray(space.low, dtype=np.float64), np.asarray(space.high, dtype=np.float64)
    out = rows.copy()
    for i in range(rows.shape[0]):
        visible = np.ones(rows.shape[1], dtype=bool)
        col = first
        for count, width in blocks:
            for _ in range(count):
                visible[col:col + width] = rows[i, col + width - 1] != 0.0
                col += width
        assert col == rows.shape[1]
        sel = cmask & visible
        origin = rows[i, consts.PRESERVED_DIM:consts.PRESERVED_DIM + 2]
        out[i, sel] -= np.tile(origin, sel.sum() // 2)
    below = np.isfinite(low)
    both = below & np.isfinite(high) & (high > low)
    out[:, below] -= low[below]
    out[:, both] = 2.0 * out[:, both] / (high - low)[both] - 1.0
    return out


@pytest.mark.parametrize('name', ['chain_4v8-9_s15', 'chain_target_2v4-0_s16'])
def test_training_chain_fixture_through_the_oracle(name):
    """The example trainers' wrapper chains (examples/ippo/camera/config.py:19-51 and examples/ippo/target/config.py:20-51; fixtures
    recorded from the reference's own wrappers) on the CPU oracle: DiscreteCamera's / DiscreteTarget's decode, the greedy opponents on
    their recorded draws, the environment, RelativeCoordinates + RescaledObservation of the learner's rows, the auxiliary reward terms
    and FrameSkip's sums -- joint actions 1e-9, masks / goals / episode reward exact, chain observations 1e-9, shaped rewards 1e-12."""
    from mate_amd.spaces import camera_action_grid, target_action_grid
    fx = G.load(name + '.npz')
    team = str(fx['learner_team'])
    me, opp = ('cam', 'tgt') if team == 'camera' else ('tgt', 'cam')
    Nc, Nt, No = (int(fx[k]) for k in ('num_cameras', 'num_targets', 'num_obstacles'))
    grid = fx[team + '_action_grid']
    assert np.array_equal((camera_action_grid if team == 'camera' else target_action_grid)(int(fx['discrete_levels'])), grid)
    env = G.oracle_from_fixture(fx)
    env.set('camera_target_view_mask', fx['reset/camera_target_view_mask'].astype(np.float64))
    mine = lambda pair: pair[0] if team == 'camera' else pair[1]  # noqa: E731
    np.testing.assert_allclose(_chain_rows(team, mine(env.observe()), Nc, Nt, No), fx[f'reset/chain_{me}_obs'], rtol=0, atol=1e-9)
    agents = O.GreedyPolicies()
    T = len(fx['step/done'])
    n_me = Nc if team == 'camera' else Nt
    shaped = np.zeros((T, n_me))
    keys, coef = [str(k) for k in fx['aux_keys']], fx['aux_coefficients']
    zeros_cam = (np.zeros(Nc), np.zeros((Nc, 2)), np.full((Nc, Nc), -1))
    zeros_tgt = (np.zeros(Nt), np.zeros(Nt), np.zeros((Nt, 2)))
    for s in range(T):
        idx = fx[f'step/{me}_idx'][s]
        decoded = mine(env.decode_discrete(idx, grid, None, None) if team == 'camera' else env.decode_discrete(None, None, idx, grid))
        assert np.array_equal(decoded, fx[f'step/{me}_act'][s]), s
        if team == 'camera':
            _, theirs = agents.act(env, *zeros_cam, fx['step/agent_tgt_choice_u'][s], fx['step/agent_tgt_binom_u'][s],
                                   fx['step/agent_tgt_sample_u'][s], fx['agent/tgt_reset_sample_u'])
            joint = (decoded, theirs)
        else:
            theirs, _ = agents.act(env, fx['step/agent_cam_binom_u'][s], fx['step/agent_cam_sample_u'][s], fx['step/agent_cam_delay'][s],
                                   *zeros_tgt, fx['agent/tgt_reset_sample_u'])
            joint = (theirs, decoded)
        assert np.abs(theirs - fx[f'step/{opp}_act'][s]).max() < 1e-9, ('the opponents joint action', s)
        before = env.get('episode_reward')
        env.step(joint[0], joint[1], fx['step/tape_ct'][s], fx['step/goal_u'][s])
        for m in G.MASK_FIELDS:
            assert np.array_equal(np.asarray(env.get(m)) != 0, fx['step/' + m][s].astype(bool)), (m, s)
        assert np.array_equal(np.asarray(env.get('tgt_goals'), dtype=np.float64), np.asarray(fx['step/tgt_goals'][s], dtype=np.float64)), s
        assert env.get('episode_reward') == fx['step/episode_reward'][s]
        np.testing.assert_allclose(_chain_rows(team, mine(env.observe()), Nc, Nt, No), fx[f'step/chain_{me}_obs'][s], rtol=0, atol=1e-9,
                                   err_msg=str(s))
        coverage = float(np.mean(np.asarray(env.get('tracked_bits')) != 0))      # environment.py:966
        assert abs(coverage - float(fx['step/info_coverage_rate'][s])) < 1e-12 and abs(coverage - float(fx['step/coverage_rate'][s])) < 1e-12
        if team == 'camera':
            shaped[s] = coverage                               # {'coverage_rate': 1.0}, reduction 'mean' of equal values
        else:                                                  # reduction 'none': the sum of the terms (auxiliary_target_rewards.py:118-216)
            raw = env.get('episode_reward') - before           # the target team's reward is the step's change of the episode sum
            assert abs(raw - float(fx['step/reward_tgt'][s])) < 1e-9, s
            terms = {
                'raw_reward': np.full(Nt, float(fx['step/reward_tgt'][s])),
                'is_tracked': (np.asarray(env.get('tracked_bits')) != 0).astype(np.float64),
                'is_colliding': (np.asarray(env.get('tgt_colliding')) != 0).astype(np.float64),
            }
            for key in terms:
                assert np.array_equal(terms[key], fx['step/aux_' + key][s]), (key, s)
            shaped[s] = sum(c * (terms[k] if k in terms else fx['step/aux_' + k][s]) for k, c in zip(keys, coef))
        np.testing.assert_allclose(shaped[s], fx[f'step/chain_reward_{me}'][s], rtol=0, atol=1e-12 if team == 'camera' else 1e-9)
    for ls in range(len(fx['skip/done'])):                      # FrameSkip: rewards summed over the action's frames
        frames = np.nonzero(fx['step/learner_step'] == ls)[0]
        np.testing.assert_allclose(shaped[frames].sum(axis=0), fx[f'skip/reward_{me}'][ls], rtol=0, atol=1e-9)
