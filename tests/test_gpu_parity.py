"""GPU parity tests (run with -m gpu on an MI355X): the HIP engine, driven through the C ABI,
against (a) the golden vectors recorded from the upstream reference and (b) the CPU oracle.

Bars (BASELINE.json north_star): tracked-by / view masks and all integer state bit-exact;
observations and rewards within 1e-5 relative in f32 (the product dtype).  The f64 observation
build of the same kernels is additionally held to 1e-9 absolute, which localises any
divergence to last-place differences of the device libm.
"""
import os

import numpy as np
import pytest
import torch

import golden_util as G
import gpu_util as U

pytestmark = pytest.mark.gpu

INT_KEYS = ['tgt_colliding', 'tgt_empty_bits', 'tgt_goal_bits', 'tgt_goals', 'freights', 'bounties', 'target_steps',
            'tracked_steps', 'remaining_cargoes', 'awaiting_cargo_counts', 'num_delivered_cargoes', 'episode_step']
MASKS = ['camera_target_view_mask', 'target_camera_view_mask', 'target_obstacle_view_mask', 'target_target_view_mask',
         'camera_camera_view_mask', 'tracked_bits']


def rel_close(got, ref, rtol):
    return np.all(np.abs(got - ref) <= rtol * np.maximum(1.0, np.abs(ref)))


@pytest.mark.parametrize('path', G.trace_files(), ids=lambda p: os.path.basename(p)[6:-4])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64obs', 'f32obs'])
def test_trace_parity(path, dtype):
    fx = G.load(path)
    N = 3
    eng = U.engine_from_fixture(fx, N, obs_dtype=dtype)
    Nc, Nt = eng.num_cameras, eng.num_targets
    dev = eng.device
    # initial observation of the injected state == the reference's reset() observation
    tape0 = torch.zeros((N, max(Nc, 1), Nt), dtype=torch.float64, device=dev)  # u=0 never sees through
    co, to = eng.observe(tape_ct=tape0)
    atol64, rtol32 = 1e-9, 1e-5
    def check_obs(co, to, ref_c, ref_t, where):
        for e in range(N):
            if Nc:
                got = co[e].double().cpu().numpy()
                ok = np.allclose(got, ref_c, rtol=0, atol=atol64) if dtype == torch.float64 else rel_close(got, ref_c, rtol32)
                assert ok, (where, 'camera obs', e, np.abs(got - ref_c).max())
            got = to[e].double().cpu().numpy()
            ok = np.allclose(got, ref_t, rtol=0, atol=atol64) if dtype == torch.float64 else rel_close(got, ref_t, rtol32)
            assert ok, (where, 'target obs', e, np.abs(got - ref_t).max())
    m0 = eng.unpack_masks()
    for m in ('target_camera_view_mask', 'target_obstacle_view_mask', 'target_target_view_mask', 'camera_obstacle_view_mask'):
        ref = fx['static/' + m] if m.startswith('camera_obstacle') else fx['reset/' + m]
        assert np.array_equal(m0[m][0], ref.astype(bool)), m
    T = len(fx['step/done'])
    for s in range(T):
        ca = torch.from_numpy(np.broadcast_to(fx['step/cam_act'][s], (N, Nc, 2)).copy()).to(dev)
        ta = torch.from_numpy(np.broadcast_to(fx['step/tgt_act'][s], (N, Nt, 2)).copy()).to(dev)
        tape = torch.from_numpy(np.broadcast_to(np.nan_to_num(fx['step/tape_ct'][s], nan=0.0), (N, Nc, Nt)).copy()).to(dev)
        goal = torch.from_numpy(np.broadcast_to(np.nan_to_num(fx['step/goal_u'][s], nan=0.0), (N, Nt)).copy()).to(dev)
        co, to, sc = eng.step(ca, ta, tape_ct=tape, tape_goal=goal, auto_reset=False)
        masks = eng.unpack_masks()
        for m in MASKS:
            for e in range(N):
                assert np.array_equal(masks[m][e], fx['step/' + m][s].astype(bool)), (m, s, e)
        sd = eng.state_dict()
        for k in INT_KEYS:
            for e in range(N):
                assert np.array_equal(sd[k][e], np.asarray(fx['step/' + k][s], dtype=np.float64)), (k, s, e)
        sc = sc.cpu().numpy()
        for e in range(N):
            assert sc[e, 0] == np.float32(fx['step/reward_cam'][s]) and sc[e, 1] == np.float32(fx['step/reward_tgt'][s]), (s, sc[e])
            assert bool(sc[e, 2]) == bool(fx['step/done'][s]), s
            assert abs(sc[e, 3] - fx['step/coverage_rate'][s]) < 1e-6
            assert abs(sc[e, 4] - fx['step/real_coverage_rate'][s]) < 1e-6
            assert abs(sc[e, 5] - fx['step/mean_transport_rate'][s]) < 1e-6
            assert sc[e, 6] == fx['step/num_delivered_cargoes'][s]
            assert abs(sc[e, 7] - fx['step/normalized_reward_tgt'][s]) < 1e-7
            assert sd['episode_reward'][e] == fx['step/episode_reward'][s]
            assert sd['delayed_episode_reward'][e] == fx['step/delayed_episode_reward'][s]
        xy = fx['step/tgt_xy'][s]
        assert np.abs(sd['tgt_x'] - xy[:, 0]).max() < 1e-9 and np.abs(sd['tgt_y'] - xy[:, 1]).max() < 1e-9, s
        if Nc:
            assert np.abs(sd['cam_phi'] - fx['step/cam_phi'][s]).max() < 1e-9
            assert np.abs(sd['cam_theta'] - fx['step/cam_theta'][s]).max() < 1e-9
        check_obs(co, to, fx['step/cam_obs'][s], fx['step/tgt_obs'][s], ('step', s))


def _check_step_against_trace(eng, fx, s, N, co, to, sc, dtype):
    """One replayed step of a reference trace: masks and integer state exact, rewards exact, positions 1e-9, observations 1e-9 (f64) / 1e-5 rel (f32)."""
    Nc = eng.num_cameras
    masks = eng.unpack_masks()
    for m in MASKS:
        for e in range(N):
            assert np.array_equal(masks[m][e], fx['step/' + m][s].astype(bool)), (m, s, e)
    sd = eng.state_dict()
    for k in INT_KEYS:
        for e in range(N):
            assert np.array_equal(sd[k][e], np.asarray(fx['step/' + k][s], dtype=np.float64)), (k, s, e)
    sc = sc.cpu().numpy()
    for e in range(N):
        assert sc[e, 0] == np.float32(fx['step/reward_cam'][s]) and sc[e, 1] == np.float32(fx['step/reward_tgt'][s]), (s, sc[e])
        assert bool(sc[e, 2]) == bool(fx['step/done'][s]), s
        assert sd['episode_reward'][e] == fx['step/episode_reward'][s]
    xy = fx['step/tgt_xy'][s]
    assert np.abs(sd['tgt_x'] - xy[:, 0]).max() < 1e-9 and np.abs(sd['tgt_y'] - xy[:, 1]).max() < 1e-9, s
    for e in range(N):
        for got, ref in (((co[e], fx['step/cam_obs'][s]),) if Nc else ()) + ((to[e], fx['step/tgt_obs'][s]),):
            got = got.double().cpu().numpy()
            assert np.allclose(got, ref, rtol=0, atol=1e-9) if dtype == torch.float64 else rel_close(got, ref, 1e-5), (s, e, np.abs(got - ref).max())


@pytest.mark.parametrize('tag,seed', [('8v8-9', 21), ('4v2-9', 22)])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64obs', 'f32obs'])
def test_two_episodes_through_a_recorded_reset(tag, seed, dtype):
    """ONE reference environment object through an episode end and the reset() behind it (shuffle_entities on), replayed on ONE engine:
    the first episode's steps until `done` (environment.py:629-632), then -- on the same engine, mid-run -- the reset the reference made
    next, consuming its recorded draws (mate_engine_reset_tape; environment.py:679-834: placements, shuffles, cargo matrix, goals, the
    occlusion tables BUILT ON THE DEVICE from that placement), then the second episode's steps on that state.  Nothing is injected
    between the two episodes: what the second episode's masks, rewards and observations are checked against is the reference's own
    continuation (tests/golden/make_golden.py two_episode_fixture)."""
    ep1, rs, ep2 = (G.load(f'trace_{tag}_greedy_ep1_s{seed}.npz'), G.load(f'reset_{tag}_ep2_s{seed}.npz'), G.load(f'trace_{tag}_greedy_ep2_s{seed}.npz'))
    N = 3
    eng = U.engine_from_fixture(ep1, N, obs_dtype=dtype)
    Nc, Nt = eng.num_cameras, eng.num_targets
    T1 = len(ep1['step/done'])
    assert bool(ep1['step/done'][T1 - 2]) and not bool(ep1['step/done'][T1 - 3])      # the episode ends inside the trace (one more step is recorded behind it)
    for s in range(T1 - 1):                                                            # up to and including the step that reports done
        co, to, sc = _replay(eng, ep1, s, N)
        _check_step_against_trace(eng, ep1, s, N, co, to, sc, dtype)
    tape = torch.from_numpy(np.broadcast_to(rs['tape'], (N, len(rs['tape']))).copy())
    tape_ct = torch.from_numpy(np.broadcast_to(np.nan_to_num(rs['tape_ct'], nan=0.0), (N, Nc, Nt)).copy()).cuda()
    co, to, used = eng.reset_tape(tape, tape_ct)
    assert used.cpu().tolist() == [len(rs['tape'])] * N
    sd = eng.state_dict()
    for key, ref in G.reset_expectation(rs).items():
        for e in range(N):
            assert np.array_equal(sd[key][e].reshape(ref.shape), ref), (key, e)
    # the reset fixture and the second trace describe the same moment of the same object
    assert np.array_equal(rs['reset/tgt_xy'], ep2['reset/tgt_xy']) and np.array_equal(rs['static/obs_xyr'], ep2['static/obs_xyr'])
    for e in range(N):
        got = to[e].double().cpu().numpy()
        assert np.allclose(got, ep2['reset/tgt_obs'], rtol=0, atol=1e-9) if dtype == torch.float64 else rel_close(got, ep2['reset/tgt_obs'], 1e-5)
    for s in range(len(ep2['step/done'])):
        co, to, sc = _replay(eng, ep2, s, N)
        _check_step_against_trace(eng, ep2, s, N, co, to, sc, dtype)
    assert int(ep2['step/num_delivered_cargoes'][-1]) > 0 or tag == '4v2-9'


@pytest.mark.parametrize('name', ['auxtgt_4v8-9_s11', 'auxtgt_8v8-9_s12', 'auxtgt_4v2-9_s13', 'auxtgt_nav_s14'])
def test_auxiliary_target_rewards_fixtures(name):
    """mate_amd.auxiliary_rewards.AuxiliaryTargetRewards on traces the reference's AuxiliaryTargetRewards wrapper shaped
    (wrappers/auxiliary_target_rewards.py:118-216): every term and the shaped reward of every target and step.  The
    per-target terms come from the f64 state (1e-9); the shared ones from the f32 step record (1e-6)."""
    from mate_amd.auxiliary_rewards import AuxiliaryTargetRewards
    fx = G.load(name + '.npz')
    N = 3
    eng = U.engine_from_fixture(fx, N, obs_dtype=torch.float32)
    keys, coef, reduction = [str(k) for k in fx['auxt_keys']], [float(c) for c in fx['auxt_coefficients']], str(fx['auxt_reduction'])
    if 'soft_coverage_score' in keys:
        eng.enable_outer_boundary()
        for c, (phis, rhos) in enumerate(G.luts_of(fx, outer=True)):
            for e in range(N):
                eng.lut_write(e, c, phis, rhos, outer=True)
    shaper = AuxiliaryTargetRewards(eng, dict(zip(keys, coef)), reduction)
    seen_delivery = False
    for s in range(len(fx['step/done'])):
        _replay(eng, fx, s, N)
        shaped = shaper().cpu().numpy()
        for key in keys:
            term = shaper.terms[key].cpu().numpy()
            loose = key in ('raw_reward', 'coverage_rate', 'real_coverage_rate', 'mean_transport_rate')
            for e in range(N):
                np.testing.assert_allclose(term[e], fx['step/auxt_' + key][s], rtol=1e-6 if loose else 1e-9, atol=1e-6 if loose else 1e-9,
                                           err_msg=f'{key} step {s}')
        for e in range(N):
            np.testing.assert_allclose(shaped[e], fx['step/aux_reward_tgt'][s], rtol=1e-6, atol=1e-5, err_msg=str(s))
        seen_delivery |= bool(fx['step/auxt_sparse_delivery'][s].any()) if 'sparse_delivery' in keys else False
    assert seen_delivery == (name in ('auxtgt_4v8-9_s11', 'auxtgt_8v8-9_s12'))     # the delivery term is exercised


@pytest.mark.parametrize('config,n', [('MATE-4v8-9.yaml', 256), ('MATE-8v8-9.yaml', 65), ('MATE-4v8-0.yaml', 64),
                                       ('MATE-Navigation.yaml', 64), ('MATE-4v2-9.yaml', 33), ('MATE-2v4-9.yaml', 16),
                                       ('MATE-1v1-0.yaml', 7)])
def test_reset_and_rollout_vs_oracle(config, n, oracle_lib):
    """Native GPU reset + Philox random-policy rollout against the CPU oracle on the same streams."""
    _reset_and_rollout_vs_oracle(config, n, oracle_lib)


def test_largest_supported_scenario_vs_oracle(oracle_lib):
    """A scenario at the limit of the LDS sort (16 cameras, 16 targets, 20 obstacles: 8 sector rounds, 13 range rounds,
    4k-knot occlusion tables in a 160 KiB-LDS sort, generic kernels) goes through the same reset + rollout parity, and so
    does one beyond it (40 obstacles: 7.8k-knot tables sorted in an HBM scratch slice per workgroup); more entities than
    the packed records can index are refused loudly."""
    from mate_amd._native import EngineError
    from mate_amd.engine import Engine
    from mate_amd.config import read_config
    cfg = read_config('MATE-8v8-9.yaml')
    cfg['name'] = 'MultiAgentTracking(16v16, 40)'
    cam = cfg['camera']['location_random_range']
    cfg['camera']['location_random_range'] = cam + [[-x1, -x0, y0, y1] if i % 2 else [x0, x1, -y1, -y0] for i, (x0, x1, y0, y1) in enumerate(cam)]
    cfg['camera']['location_random_range'] = [[float(v) for v in box] for box in cfg['camera']['location_random_range']]
    cfg['target']['location_random_range'] = [[-300.0, 300.0, -300.0, 300.0]] * 16
    obs = cfg['obstacle']['location_random_range']
    cfg['obstacle']['location_random_range'] = (obs * 5)[:20]
    cfg['obstacle']['radius_random_range'] = [10.0, 40.0]
    _reset_and_rollout_vs_oracle(cfg, 9, oracle_lib, steps=12)
    import copy
    bigger = copy.deepcopy(cfg)
    bigger['name'] = 'MultiAgentTracking(16v16, 40)'
    bigger['obstacle']['location_random_range'] = (obs * 5)[:40]
    _reset_and_rollout_vs_oracle(bigger, 5, oracle_lib, steps=8)
    too_big = copy.deepcopy(cfg)
    too_big['obstacle']['location_random_range'] = (obs * 8)[:65]
    with pytest.raises(EngineError, match='unsupported entity counts'):
        Engine(too_big, 4)


def _reset_and_rollout_vs_oracle(config, n, oracle_lib, steps=40):
    O = oracle_lib
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config(config) if isinstance(config, str) else config
    seed, first = 1234, 1000
    eng = Engine(cfg, n, seed=seed, first_env_index=first, obs_dtype=torch.float64)
    eng.reset()
    torch.cuda.synchronize()
    sd = eng.state_dict()
    proto = U.oracle_proto_from_config(cfg, O)
    batch = O.OracleBatch(proto, n, seed=seed, first_env_index=first)
    batch.reset(threads=4)
    Nc, Nt, No = eng.num_cameras, eng.num_targets, eng.num_obstacles
    # (1) reset parity: placement / cargo / goals come from the same Philox stream -> exact
    for k in U.STATE_KEYS:
        ref = batch.gather(k)
        assert np.array_equal(sd[k].reshape(ref.shape), ref), ('reset', k)
    # (2) occlusion tables: knot-for-knot up to libm last-place noise and tangent-ray coin flips
    flips = 0
    for e in range(min(n, 24)):
        oe = batch.env(e)
        for c in range(Nc):
            gp, gr = eng.lut_read(e, c)
            op, orr = oe.get_lut(c)
            assert len(gp) == len(op), (e, c, len(gp), len(op))
            assert np.abs(gp - op).max() < 1e-9
            obstacles = np.stack([sd['obs_x'][e], sd['obs_y'][e], sd['obs_radius'][e]], axis=-1)
            flips += G.assert_only_tangent_flips(gp, gr, orr, (sd['cam_x'][e][c], sd['cam_y'][e][c]), float(cfg['camera']['max_sight_range']), obstacles, 1e-6, (e, c))
            oe.set_lut(c, gp, gr)   # continue with identical tables on both sides
    for e in range(min(n, 24), n):
        oe = batch.env(e)
        for c in range(Nc):
            gp, gr = eng.lut_read(e, c)
            oe.set_lut(c, gp, gr)
    # reset observations: the first observation of the episode (view drawn from the reset-view stream) on both sides
    batch.update_view_reset()
    oc, ot = batch.observe()
    co, to = eng.camera_obs.cpu().numpy(), eng.target_obs.cpu().numpy()
    assert rel_close(to, ot, 1e-5), ('reset target obs', np.abs(to - ot).max())
    if Nc:
        assert rel_close(co, oc, 1e-5), ('reset camera obs', np.abs(co - oc).max())
    # (3) rollout
    for s in range(steps):
        eng.step_random(auto_reset=False, want_masks=True)
        batch.step(auto_reset=False, threads=4)
        masks = eng.unpack_masks()
        sdg = eng.state_dict()
        for m, field in [('camera_target_view_mask', 'camera_target_view_mask'), ('target_camera_view_mask', 'target_camera_view_mask'),
                         ('target_obstacle_view_mask', 'target_obstacle_view_mask'), ('target_target_view_mask', 'target_target_view_mask'),
                         ('camera_camera_view_mask', 'camera_camera_view_mask')]:
            ref = batch.gather(field) != 0
            assert np.array_equal(masks[m].reshape(ref.shape), ref), (m, s, np.argwhere(masks[m].reshape(ref.shape) != ref)[:4])
        for k in INT_KEYS:
            ref = batch.gather(k)
            assert np.array_equal(sdg[k].reshape(ref.shape), ref), (k, s)
        assert np.abs(sdg['tgt_x'] - batch.gather('tgt_x')).max() < 1e-9, s
        assert np.abs(sdg['tgt_y'] - batch.gather('tgt_y')).max() < 1e-9, s
    oc, ot = batch.observe()
    co, to = eng.camera_obs.cpu().numpy(), eng.target_obs.cpu().numpy()
    assert rel_close(to, ot, 1e-5), np.abs(to - ot).max()   # oracle batch observations are f32: 1e-5 relative, the north-star bar
    if Nc:
        assert rel_close(co, oc, 1e-5), np.abs(co - oc).max()
    sc = eng.scalars.cpu().numpy()
    assert np.array_equal(sc[:, 1], batch.gather('reward_tgt').astype(np.float32))


def test_auto_reset_and_done():
    """Episodes end (time limit), finished environments restart inside the same call."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=5)
    eng = Engine(cfg, 10, seed=7)
    eng.reset()
    dones = []
    for s in range(14):
        eng.step_random(auto_reset=True)
        dones.append(eng.scalars[:, 2].cpu().numpy().copy())
        sd = eng.state_dict()
        if dones[-1].all():
            assert (sd['episode_step'] == 0).all() and (sd['episode'] == sd['episode'][0]).all()
    dones = np.array(dones)
    # done fires on call max_steps + 1 (environment.py:629-632), then every 6 calls
    assert dones[:5].sum() == 0 and dones[5].all() and dones[6:11].sum() == 0 and dones[11].all()
    assert (eng.state_dict()['episode'] == 3).all()


def test_masked_reset_only_touches_selected():
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    eng = Engine(read_config('MATE-4v8-9.yaml'), 8, seed=3)
    eng.reset()
    before = eng.state_dict()
    mask = torch.tensor([1, 0, 0, 1, 0, 0, 0, 1], dtype=torch.uint8)
    eng.reset(env_mask=mask)
    after = eng.state_dict()
    changed = (before['tgt_x'] != after['tgt_x']).any(axis=1)
    assert np.array_equal(changed, mask.numpy().astype(bool))
    assert np.array_equal(after['episode'], 1 + mask.numpy())


def test_errors_are_loud():
    from mate_amd import _native
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    eng = Engine(read_config('MATE-4v8-9.yaml'), 4)
    with pytest.raises(_native.EngineError):
        eng.step_random()      # before reset
    with pytest.raises(ValueError):
        read_config('MATE-4v8-9.yaml', num_cargoes_per_target=2)


@pytest.mark.parametrize('name', ['4v8-9_greedy_s2', 'nav_greedy_s1'])
@pytest.mark.parametrize('mode', ['relative', 'rescaled', 'relative_rescaled'])
def test_fused_observation_transforms(name, mode):
    """RelativeCoordinates / RescaledObservation fused into the packer == the reference's wrapper functions
    applied to the reference's observations (fixtures xform_*.npz)."""
    fx = G.load('trace_' + name + '.npz')
    xf = G.load('xform_' + name + '.npz')
    N = 2
    eng = U.engine_from_fixture(fx, N, obs_dtype=torch.float32)
    eng.set_obs_transform(relative_coordinates='relative' in mode, rescaled_observation='rescaled' in mode)
    Nc, Nt = eng.num_cameras, eng.num_targets
    dev = eng.device
    for s in range(int(xf['steps'])):
        ca = torch.from_numpy(np.broadcast_to(fx['step/cam_act'][s], (N, Nc, 2)).copy()).to(dev)
        ta = torch.from_numpy(np.broadcast_to(fx['step/tgt_act'][s], (N, Nt, 2)).copy()).to(dev)
        tape = torch.from_numpy(np.broadcast_to(np.nan_to_num(fx['step/tape_ct'][s], nan=0.0), (N, Nc, Nt)).copy()).to(dev)
        goal = torch.from_numpy(np.broadcast_to(np.nan_to_num(fx['step/goal_u'][s], nan=0.0), (N, Nt)).copy()).to(dev)
        co, to, _ = eng.step(ca, ta, tape_ct=tape, tape_goal=goal)
        ref_t = xf['tgt_obs_' + mode][s].astype(np.float64)
        got_t = to[1].double().cpu().numpy()
        # bound: 1e-5 of the column scale (rescaled columns live in [-1, 1]; raw coordinates in +-2000)
        tol = 1e-5 * np.maximum(1.0, np.abs(ref_t)) if 'rescaled' not in mode else 1e-5
        assert np.all(np.abs(got_t - ref_t) <= tol + 1e-4 * ('rescaled' not in mode)), (s, np.abs(got_t - ref_t).max())
        if Nc:
            ref_c = xf['cam_obs_' + mode][s].astype(np.float64)
            got_c = co[1].double().cpu().numpy()
            tolc = 1e-5 * np.maximum(1.0, np.abs(ref_c)) if 'rescaled' not in mode else 1e-5
            assert np.all(np.abs(got_c - ref_c) <= tolc + 1e-4 * ('rescaled' not in mode)), (s, np.abs(got_c - ref_c).max())


@pytest.mark.parametrize('workload', ['MATE-4v8-9.yaml', 'MATE-4v2-9.yaml', 'MATE-8v8-9.yaml', 'MATE-4v8-0.yaml', 'MATE-Navigation.yaml'])
def test_shape_specialised_kernel_equals_generic(workload, monkeypatch):
    """The step kernels compiled for the shipped scenario shapes (FixedShape) and the generic kernel (AnyShape,
    MATE_GENERIC=1) are the same code with different constant folding: every output must be bit-identical."""
    import torch
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    outs = []
    for generic in ('0', '1'):
        monkeypatch.setenv('MATE_GENERIC', generic)
        eng = Engine(read_config(workload), 64, seed=11)
        assert eng.specialised == (generic == '0')
        eng.reset()
        rec = []
        for _ in range(40):
            eng.step_random(auto_reset=True)
            rec.append([t.clone() for t in (getattr(eng, 'camera_obs', None), eng.target_obs, eng.scalars, eng.masks) if t is not None])
        rec.append([eng.export_state().clone()])
        outs.append(rec)
        del eng
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))


@pytest.mark.parametrize('workload,generic_shape', [('MATE-4v8-9.yaml', '0'), ('MATE-4v8-9.yaml', '1'), ('MATE-Navigation.yaml', '0'), ('MATE-8v8-9.yaml', '0')])
@pytest.mark.parametrize('policy', ['random', 'actions'])
def test_flow_specialised_kernel_equals_generic(workload, generic_shape, policy, monkeypatch):
    """The step kernels compiled with the launch switches folded (Flow: on-device random policy / f32 joint actions)
    and the generic flow (MATE_FLOW_GENERIC=1) are the same code: every output and the state must be bit-identical,
    and a launch outside the folded switches (a tape, masks-only outputs, f64 actions) must fall back to the generic flow."""
    import torch
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    monkeypatch.setenv('MATE_GENERIC', generic_shape)
    outs = []
    for generic in ('0', '1'):
        monkeypatch.setenv('MATE_FLOW_GENERIC', generic)
        eng = Engine(read_config(workload), 64, seed=13)
        eng.reset()
        gen = torch.Generator(device='cpu').manual_seed(5)
        rec = []
        for _ in range(40):
            if policy == 'random':
                eng.step_random(auto_reset=True, want_masks=True)
            else:
                ca = ((torch.rand((64, eng.num_cameras, 2), generator=gen) * 2 - 1) * 6).cuda()
                ta = ((torch.rand((64, eng.num_targets, 2), generator=gen) * 2 - 1) * 25).cuda()
                eng.step(ca, ta, auto_reset=True)
            assert eng.last_flow == (0 if generic == '1' else (1 if policy == 'random' else 2))
            rec.append([t.clone() for t in (getattr(eng, 'camera_obs', None), eng.target_obs, eng.scalars, eng.masks) if t is not None])
        for _ in range(8):      # batched auto-reset (finished environments idle, one reset launch per 4 steps) runs the folded flows too
            if policy == 'random':
                eng.step_random(auto_reset=4, want_masks=True)
            else:
                eng.step(ca, ta, auto_reset=4)
            assert eng.last_flow == (0 if generic == '1' else (1 if policy == 'random' else 2))
            rec.append([t.clone() for t in (getattr(eng, 'camera_obs', None), eng.target_obs, eng.scalars, eng.masks) if t is not None])
        if policy == 'random':  # the fused K-step rollout has the same two compilations
            ro = eng.rollout_random(6, auto_reset=True)
            assert eng.last_flow == (0 if generic == '1' else 1)
            rec.append([t.clone() for t in ro if t is not None and t.numel()])
        rec.append([eng.export_state().clone()])
        outs.append(rec)
        if generic == '0':      # launches the folded flows do not cover
            ta = torch.zeros((64, eng.num_targets, 2), dtype=torch.float64, device='cuda')
            ca = torch.zeros((64, eng.num_cameras, 2), dtype=torch.float64, device='cuda')
            eng.step(ca, ta, auto_reset=False)
            assert eng.last_flow == 2          # f64 (or per-team mixed) joint actions run the folded kernel too: the encoding is a launch argument
            eng.step(ca.float(), ta.float(), tape_goal=torch.zeros((64, eng.num_targets), dtype=torch.float64, device='cuda'), auto_reset=False)
            assert eng.last_flow == 0
        del eng
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))


def _replay(eng, fx, s, N, cam=None, tgt=None):
    Nc, Nt, dev = eng.num_cameras, eng.num_targets, eng.device
    ca = cam if cam is not None else torch.from_numpy(np.broadcast_to(fx['step/cam_act'][s], (N, Nc, 2)).copy()).to(dev)
    ta = tgt if tgt is not None else torch.from_numpy(np.broadcast_to(fx['step/tgt_act'][s], (N, Nt, 2)).copy()).to(dev)
    tape = torch.from_numpy(np.broadcast_to(np.nan_to_num(fx['step/tape_ct'][s], nan=0.0), (N, Nc, Nt)).copy()).to(dev)
    goal = torch.from_numpy(np.broadcast_to(np.nan_to_num(fx['step/goal_u'][s], nan=0.0), (N, Nt)).copy()).to(dev)
    return eng.step(ca, ta, tape_ct=tape, tape_goal=goal, auto_reset=False)


@pytest.mark.parametrize('name,mode,team', [
    ('obsmode_4v8-9_s4', 'enhanced', 'both'), ('obsmode_4v8-9_s4', 'enhanced', 'camera'), ('obsmode_4v8-9_s4', 'enhanced', 'target'),
    ('obsmode_4v8-9_s4', 'shared', 'both'), ('obsmode_4v8-9_s4', 'shared', 'camera'), ('obsmode_4v8-9_s4', 'shared', 'target'),
    ('obsmode_4v8-9_fewcargo', 'enhanced', 'both'), ('obsmode_4v8-9_fewcargo', 'shared', 'both'),
    ('obsmode_nav_s2', 'enhanced', 'target'), ('obsmode_nav_s2', 'shared', 'target')])
def test_observation_modes(name, mode, team):
    """EnhancedObservation / SharedFieldOfView fused into the packer == the reference's wrappers applied to the
    reference's observations on the same trace (fixtures obsmode_*.npz); the view masks stay the plain ones."""
    fx = G.load(name + '.npz')
    N = 2
    eng = U.engine_from_fixture(fx, N, obs_dtype=torch.float32)
    sides = ('camera', 'target') if team == 'both' else (team,)
    eng.set_obs_mode(**{side: mode for side in sides})
    Nc = eng.num_cameras
    key = mode + '_' + team
    tape0 = torch.zeros((N, max(Nc, 1), eng.num_targets), dtype=torch.float64, device=eng.device)
    co, to = eng.observe(tape_ct=tape0)
    if Nc:
        assert rel_close(co[1].double().cpu().numpy(), fx['reset/cam_obs_' + key].astype(np.float64), 1e-5), 'reset camera rows'
    assert rel_close(to[1].double().cpu().numpy(), fx['reset/tgt_obs_' + key].astype(np.float64), 1e-5), 'reset target rows'
    for s in range(len(fx['step/done'])):
        co, to, _ = _replay(eng, fx, s, N)
        masks = eng.unpack_masks()
        for m in MASKS:
            assert np.array_equal(masks[m][1], fx['step/' + m][s].astype(bool)), (m, s)
        if Nc:
            got, ref = co[1].double().cpu().numpy(), fx['step/cam_obs_' + key][s].astype(np.float64)
            assert rel_close(got, ref, 1e-5), (s, 'camera', np.abs(got - ref).max())
        got, ref = to[1].double().cpu().numpy(), fx['step/tgt_obs_' + key][s].astype(np.float64)
        assert rel_close(got, ref, 1e-5), (s, 'target', np.abs(got - ref).max())


@pytest.mark.parametrize('name', ['discrete_4v8-9_s6', 'discrete_4v2-9_s7'])
def test_discrete_actions(name):
    """Joint actions given as grid indices (DiscreteCamera / DiscreteTarget of the reference) are decoded in the
    kernel: the trace the reference produced from the wrappers' continuous actions is reproduced from the indices."""
    fx = G.load(name + '.npz')
    N = 2
    eng = U.engine_from_fixture(fx, N, obs_dtype=torch.float32)
    lc, lt = (int(v) for v in fx['discrete_levels'])
    eng.set_action_grids(camera_levels=lc, target_levels=lt)
    assert np.array_equal(eng.camera_action_grid, fx['camera_action_grid']) and np.array_equal(eng.target_action_grid, fx['target_action_grid'])
    dev = eng.device
    for s in range(len(fx['step/done'])):
        ci = torch.from_numpy(np.broadcast_to(fx['step/cam_idx'][s], (N, eng.num_cameras)).copy()).to(dev)
        ti = torch.from_numpy(np.broadcast_to(fx['step/tgt_idx'][s], (N, eng.num_targets)).copy()).to(dev)
        co, to, sc = _replay(eng, fx, s, N, cam=ci, tgt=ti)
        masks = eng.unpack_masks()
        for m in MASKS:
            assert np.array_equal(masks[m][1], fx['step/' + m][s].astype(bool)), (m, s)
        sd = eng.state_dict()
        xy = fx['step/tgt_xy'][s]
        assert np.abs(sd['tgt_x'][1] - xy[:, 0]).max() < 1e-9 and np.abs(sd['tgt_y'][1] - xy[:, 1]).max() < 1e-9, s
        assert np.abs(sd['cam_phi'][1] - fx['step/cam_phi'][s]).max() < 1e-9 and np.abs(sd['cam_theta'][1] - fx['step/cam_theta'][s]).max() < 1e-9
        assert rel_close(co[1].double().cpu().numpy(), fx['step/cam_obs'][s], 1e-5) and rel_close(to[1].double().cpu().numpy(), fx['step/tgt_obs'][s], 1e-5)
    # mixed: discrete cameras with continuous targets (the common RLlib configuration)
    eng2 = U.engine_from_fixture(fx, N, obs_dtype=torch.float32)
    eng2.set_action_grids(camera_levels=lc)
    ci = torch.from_numpy(np.broadcast_to(fx['step/cam_idx'][0], (N, eng2.num_cameras)).copy()).to(dev)
    co, to, _ = _replay(eng2, fx, 0, N, cam=ci)
    assert rel_close(co[1].double().cpu().numpy(), fx['step/cam_obs'][0], 1e-5) and rel_close(to[1].double().cpu().numpy(), fx['step/tgt_obs'][0], 1e-5)


@pytest.mark.parametrize('config,n', [('MATE-4v8-9.yaml', 24), ('MATE-8v8-9.yaml', 12)])
def test_outer_boundary_vs_oracle(config, n, oracle_lib):
    """Camera.boundary_outer built by the reset kernel (mate_engine_enable_outer_boundary) against the oracle's builder
    on the same natively reset geometry: knot for knot, up to the tangent-ray coin flips of the reference."""
    O = oracle_lib
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config(config)
    eng = Engine(cfg, n, seed=77, first_env_index=40)
    eng.enable_outer_boundary()
    eng.reset()
    sd = eng.state_dict()
    batch = O.OracleBatch(U.oracle_proto_from_config(cfg, O), n, seed=77, first_env_index=40)
    batch.reset(threads=4)
    No = eng.num_obstacles
    grid = np.linspace(-180.0, 180.0, 7201)
    for e in range(n):
        oe = batch.env(e)
        for c in range(eng.num_cameras):
            gp, gr = eng.lut_read(e, c)
            op, orr = oe.get_lut(c)
            assert len(gp) == len(op) and np.abs(gp - op).max() < 1e-9, (e, c)
            obstacles = np.stack([sd['obs_x'][e], sd['obs_y'][e], sd['obs_radius'][e]], axis=-1)
            G.assert_only_tangent_flips(gp, gr, orr, (sd['cam_x'][e][c], sd['cam_y'][e][c]), float(cfg['camera']['max_sight_range']), obstacles, 1e-6, (e, c))
            # outer: the flank points next to a tangent direction may or may not merge with the arc's end ray in the
            # reference (an atan2(sin, cos) round trip decides), so the piecewise-linear FUNCTIONS are compared
            gp, gr = eng.lut_read(e, c, outer=True)
            op, orr = oe.get_lut(c, outer=True)
            assert abs(len(gp) - len(op)) <= 2 * No, (e, c, len(gp), len(op))
            diff = np.abs(np.interp(grid, gp, gr) - np.interp(grid, op, orr))
            assert (diff > 1e-6).mean() < 0.002 * max(No, 1), (e, c, (diff > 1e-6).sum(), diff.max())


def test_outer_boundary_of_reference_geometry():
    """... and against the tables the reference itself built (fixture geometry imported, tables rebuilt on the device);
    boundary_between(outer=True) of the N=1 API then returns the reference's knots."""
    fx = G.load('trace_4v8-9_greedy_s2.npz')
    eng = U.engine_from_fixture(fx, 2)
    eng.enable_outer_boundary()
    eng.rebuild_luts()
    No = eng.num_obstacles
    grid = np.linspace(-180.0, 180.0, 7201)
    for c, (phis, rhos) in enumerate(G.luts_of(fx, outer=True)):
        gp, gr = eng.lut_read(1, c, outer=True)
        assert abs(len(gp) - len(phis)) <= 2 * No
        diff = np.abs(np.interp(grid, gp, gr) - np.interp(grid, phis, rhos))
        assert (diff > 1e-6).mean() < 0.002 * No, (c, (diff > 1e-6).sum(), diff.max())
    for c, (phis, rhos) in enumerate(G.luts_of(fx)):
        gp, gr = eng.lut_read(1, c)
        assert len(gp) == len(phis) and np.abs(gp - phis).max() < 1e-9
        G.assert_only_tangent_flips(phis, gr, rhos, fx['static/cam_xy'][c], float(fx['static/cam_max_sight_range'][c]), fx['static/obs_xyr'], 1e-6, c)


@pytest.mark.parametrize('shape', [(1, 1, 0), (1, 3, 2), (2, 5, 1), (3, 2, 7), (5, 7, 4), (6, 3, 12), (7, 16, 9), (10, 4, 15), (16, 1, 20), (0, 5, 3), (0, 16, 64)],
                         ids=lambda s: '%dv%d-%d' % s)
def test_arbitrary_shapes_vs_oracle(shape, oracle_lib):
    """The generic kernels on scenario shapes no specialisation exists for (including odd, prime and extreme counts):
    native reset + rollout against the oracle on the same streams."""
    from mate_amd.config import read_config
    nc, nt, no = shape
    base = read_config('MATE-8v8-9.yaml')
    rng = np.random.RandomState(nc * 1000 + nt * 50 + no)

    def boxes(n, lo, hi, size):
        out = []
        for _ in range(n):
            x, y = rng.uniform(lo, hi, size=2)
            out.append([float(x), float(x + size), float(y), float(y + size)])
        return out
    cfg = {k: v for k, v in base.items() if k not in ('camera', 'target', 'obstacle')}
    cfg['name'] = 'MultiAgentTracking(%dv%d, %d)' % shape
    cfg['high_capacity_target_split'] = 0.37
    if nc:
        cfg['camera'] = dict(base['camera'], location_random_range=boxes(nc, -850.0, 750.0, 100.0))
    cfg['target'] = dict(base['target'], location_random_range=boxes(nt, -400.0, 300.0, 100.0))
    if no:
        cfg['obstacle'] = dict(base['obstacle'], location_random_range=boxes(no, -800.0, 700.0, 100.0), radius_random_range=[10.0, 45.0])
    _reset_and_rollout_vs_oracle(cfg, 5, oracle_lib, steps=15)


@pytest.mark.parametrize('config,cam_mode,tgt_mode', [('MATE-8v8-9.yaml', 'shared', 'enhanced'), ('MATE-4v8-9.yaml', 'enhanced', 'shared'),
                                                      ('MATE-4v2-9.yaml', 'shared', 'shared'), ('MATE-Navigation.yaml', 'plain', 'shared'),
                                                      ('MATE-4v8-0.yaml', 'enhanced', 'plain')])
def test_observation_modes_vs_oracle_rollout(config, cam_mode, tgt_mode, oracle_lib):
    """The fused EnhancedObservation / SharedFieldOfView modes on natively reset batches, mixed per team, against the
    oracle's restatement of the wrappers (which the reference fixtures pin) after a random rollout."""
    O = oracle_lib
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config(config)
    n = 24
    eng = Engine(cfg, n, seed=31, first_env_index=7, obs_dtype=torch.float64)
    eng.set_obs_mode(camera=cam_mode, target=tgt_mode)
    eng.reset()
    batch = O.OracleBatch(U.oracle_proto_from_config(cfg, O), n, seed=31, first_env_index=7)
    batch.reset(threads=4)
    for e in range(n):
        for c in range(eng.num_cameras):
            batch.env(e).set_lut(c, *eng.lut_read(e, c))
    for s in range(30):
        eng.step_random(auto_reset=False, want_masks=True)
        batch.step(auto_reset=False, threads=4)
        if s % 10 == 9:
            co, to = eng.camera_obs.cpu().numpy(), eng.target_obs.cpu().numpy()
            for e in range(n):
                oc, ot = batch.env(e).observe_mode(camera=cam_mode, target=tgt_mode)
                assert np.abs(to[e] - ot).max() < 1e-9, (s, e, 'target rows')
                if eng.num_cameras:
                    assert np.abs(co[e] - oc).max() < 1e-9, (s, e, 'camera rows')


@pytest.mark.parametrize('name', ['softcov_4v8-9_s8', 'softcov_8v8-9_s9', 'softcov_4v2-9_s10'])
def test_soft_coverage_score_fixtures(name):
    """AuxiliaryCameraRewards' soft coverage score (mate_engine_soft_coverage) on traces the reference wrapper shaped:
    score matrix, per-camera scores and shaped rewards, with the reference's own inner and outer tables installed.
    f64 arithmetic throughout; tolerance 1e-9 relative (last places of sin/cos and hypot vs sqrt)."""
    fx = G.load(name + '.npz')
    N = 3
    eng = U.engine_from_fixture(fx, N, obs_dtype=torch.float32)
    eng.enable_outer_boundary()
    for c, (phis, rhos) in enumerate(G.luts_of(fx, outer=True)):
        for e in range(N):
            eng.lut_write(e, c, phis, rhos, outer=True)
    keys, coef, reduction = [str(k) for k in fx['aux_keys']], fx['aux_coefficients'], str(fx['aux_reduction'])
    for s in range(len(fx['step/done'])):
        _replay(eng, fx, s, N)
        matrix, scores = (x.cpu().numpy() for x in eng.soft_coverage())
        for e in range(N):
            np.testing.assert_allclose(matrix[e], fx['step/soft_coverage_matrix'][s], rtol=1e-9, atol=1e-9, err_msg=str(s))
            np.testing.assert_allclose(scores[e], fx['step/soft_coverage_score'][s], rtol=1e-9, atol=1e-9, err_msg=str(s))
        seen = eng.unpack_masks()['camera_target_view_mask'][1]
        sc = eng.scalars.double().cpu().numpy()[1]
        terms = {'raw_reward': sc[0], 'coverage_rate': sc[3], 'real_coverage_rate': sc[4], 'mean_transport_rate': sc[5],
                 'soft_coverage_score': scores[1], 'num_tracked': seen.sum(axis=1).astype(np.float64), 'baseline': 1.0}
        shaped = sum(c * terms[k] for k, c in zip(keys, coef)) * np.ones(eng.num_cameras)
        if reduction != 'none':
            shaped = np.full_like(shaped, {'mean': np.mean, 'sum': np.sum, 'max': np.max, 'min': np.min}[reduction](shaped))
        np.testing.assert_allclose(shaped, fx['step/aux_reward_cam'][s], rtol=1e-6, atol=1e-6, err_msg=str(s))   # f32 step record


@pytest.mark.parametrize('config,n', [('MATE-4v8-9.yaml', 20), ('MATE-8v8-9.yaml', 10)])
def test_soft_coverage_vs_oracle_rollout(config, n, oracle_lib):
    """... and on natively reset batches with the device-built tables, against the oracle's restatement on the same
    state, view masks and tables, along a random rollout (wide and narrow sectors, sectors across +-180 degrees)."""
    O = oracle_lib
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config(config)
    eng = Engine(cfg, n, seed=5, first_env_index=3)
    eng.enable_outer_boundary()
    eng.reset()
    proto = U.oracle_proto_from_config(cfg, O)
    envs = [O.OracleEnv(eng.num_cameras, eng.num_targets, eng.num_obstacles) for _ in range(n)]
    tables = [[(eng.lut_read(e, c), eng.lut_read(e, c, outer=True)) for c in range(eng.num_cameras)] for e in range(n)]
    del proto
    wrapped = 0
    for s in range(40):
        eng.step_random(auto_reset=False, want_masks=True)
        if s % 8 != 7:
            continue
        matrix, scores = (x.cpu().numpy() for x in eng.soft_coverage())
        sd, masks = eng.state_dict(), eng.unpack_masks()
        for e in range(n):
            U.oracle_load_engine_state(envs[e], sd, e, cfg)
            for c in range(eng.num_cameras):
                envs[e].set_lut(c, *tables[e][c][0])
                envs[e].set_lut(c, *tables[e][c][1], outer=True)
            envs[e].set('camera_target_view_mask', masks['camera_target_view_mask'][e].astype(np.float64))
            om, osc = envs[e].soft_coverage()
            np.testing.assert_allclose(matrix[e], om, rtol=1e-9, atol=1e-9, err_msg=str((s, e)))
            np.testing.assert_allclose(scores[e], osc, rtol=1e-9, atol=1e-9, err_msg=str((s, e)))
            left = (sd['cam_phi'][e] - sd['cam_theta'][e] / 2 + 180.0) % 360.0 - 180.0
            wrapped += int((left + sd['cam_theta'][e] > 180.0).sum())
    assert wrapped > 0        # sectors crossing the +-180 degree seam were part of the check
