"""GPU (-m gpu) parity of the episode boundary and of the scalar edge cases, through the C ABI:

* reset() pinned to the reference: the HIP reset kernel consumes the uniforms the reference's own reset() drew
  (tests/golden/reset_*.npz, recorded by make_golden.py `reset_fixture`) and must produce the reference's state;
* the F3 known-answer vectors (kat_perceive / kat_obstruct / kat_scalar), which the CPU suite runs on the oracle, driven
  through the device's own `Camera.perceive`, `Obstacle.obstruct`, `normalize_angle`, step clamp and `Camera.simulate`
  (entities.py:491-511, 158-184, 347-360; utils.py:155-158, 223-229).
"""
import os

import numpy as np
import pytest
import torch

import golden_util as G

pytestmark = pytest.mark.gpu


def rel_close(got, ref, rtol):
    return np.all(np.abs(got - ref) <= rtol * np.maximum(1.0, np.abs(ref)))


@pytest.mark.parametrize('path', G.reset_files(), ids=lambda p: os.path.basename(p)[6:-4])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64obs', 'f32obs'])
def test_reset_tape_parity(path, dtype):
    """environment.py:679-834 on the device with the reference's recorded draws: placements, capacities, cargo matrix,
    goals and the camera->obstacle mask exact; occlusion tables knot for knot; first view and observations."""
    from mate_amd.engine import Engine
    fx = G.load(path)
    cfg = G.config_of_reset_fixture(fx)
    N = 3
    eng = Engine(cfg, N, seed=5, obs_dtype=dtype)
    Nc, Nt, No = eng.num_cameras, eng.num_targets, eng.num_obstacles
    tape = torch.from_numpy(np.broadcast_to(fx['tape'], (N, len(fx['tape']))).copy())
    tape_ct = torch.from_numpy(np.broadcast_to(np.nan_to_num(fx['tape_ct'], nan=0.0), (N, Nc, Nt)).copy()).cuda() if Nc else None
    co, to, used = eng.reset_tape(tape, tape_ct)
    torch.cuda.synchronize()
    assert used.cpu().tolist() == [len(fx['tape'])] * N
    sd = eng.state_dict()
    for key, ref in G.reset_expectation(fx).items():
        for e in range(N):
            assert np.array_equal(sd[key][e].reshape(ref.shape), ref), (key, e, sd[key][e], ref)
    for c, (phis, rhos) in enumerate(G.luts_of(fx)):
        gp, gr = eng.lut_read(1, c)
        assert len(gp) == len(phis), (c, len(gp), len(phis))
        assert np.abs(gp - phis).max() < 1e-9
        # ... except the reference's own tangent-ray coin flips: only knots AT a tangent direction of an obstacle in range may differ
        G.assert_only_tangent_flips(phis, gr, rhos, fx['static/cam_xy'][c], float(fx['static/cam_max_sight_range'][c]), fx['static/obs_xyr'], 1e-6, (path, c))
    masks = eng.unpack_masks()
    for m in G.MASK_FIELDS:
        for e in range(N):
            assert np.array_equal(masks[m][e], fx['reset/' + m].astype(bool)), (m, e)
    assert np.array_equal(masks['camera_obstacle_view_mask'][0], fx['static/camera_obstacle_view_mask'].astype(bool))
    for e in range(N):
        got_t = to[e].double().cpu().numpy()
        ok = np.allclose(got_t, fx['reset/tgt_obs'], rtol=0, atol=1e-9) if dtype == torch.float64 else rel_close(got_t, fx['reset/tgt_obs'], 1e-5)
        assert ok, ('target obs', e, np.abs(got_t - fx['reset/tgt_obs']).max())
        if Nc:
            got_c = co[e].double().cpu().numpy()
            ok = np.allclose(got_c, fx['reset/cam_obs'], rtol=0, atol=1e-9) if dtype == torch.float64 else rel_close(got_c, fx['reset/cam_obs'], 1e-5)
            assert ok, ('camera obs', e, np.abs(got_c - fx['reset/cam_obs']).max())
    sc = eng.scalars.cpu().numpy()
    assert np.all(np.abs(sc[:, 3] - fx['reset/coverage_rate']) < 1e-6) and np.all(np.abs(sc[:, 4] - fx['reset/real_coverage_rate']) < 1e-6)
    # a masked tape reset touches the selected environments only, a tape that is too short is reported
    if path.endswith('4v8-9_s0.npz'):
        eng2 = Engine(cfg, N, seed=5, obs_dtype=dtype)
        eng2.reset()
        before = eng2.state_dict()
        _, _, used = eng2.reset_tape(tape, tape_ct, env_mask=torch.tensor([0, 1, 0], dtype=torch.uint8))
        after = eng2.state_dict()
        assert np.array_equal(after['tgt_x'][1], sd['tgt_x'][1]) and np.array_equal(after['tgt_x'][[0, 2]], before['tgt_x'][[0, 2]])
        _, _, used = eng2.reset_tape(tape[:, :40], tape_ct)
        assert used.cpu().tolist() == [-1] * N


def _single_camera_engine(rmax, n):
    """One camera at the origin, one target, nine obstacles (room for the 545-knot tables of the KAT), tau = 0.1."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-1v1-9.yaml', camera={'max_sight_range': float(rmax)}, obstacle={'transmittance': 0.1})
    eng = Engine(cfg, n, seed=1, obs_dtype=torch.float64)
    eng.reset()
    return eng


def test_kat_perceive_on_device():
    """Camera.perceive + sight_range_at (entities.py:491-511) on crafted geometry: +-180 degree wrap, targets exactly on
    a knot angle, on the sector edge, at the range limit, behind / beside obstacles, see-through draws.  Every case is
    one environment (one camera, one target) holding the recorded camera pose, target position and occlusion table."""
    k = G.load('kat_perceive.npz')
    cases = k['cases']
    total = wrong = 0
    for case in range(len(k['lut_count'])):
        rows = cases[cases[:, 0] == case]
        n = len(rows)
        eng = _single_camera_engine(k['cam_max_sight_range'][case], n)
        cnt = int(k['lut_count'][case])
        phis, rhos = k['lut_phis'][case, :cnt], k['lut_rhos'][case, :cnt]
        for e in range(n):
            eng.lut_write(e, 0, phis, rhos)
        eng.load_state_dict({
            'cam_x': np.full((n, 1), k['cam_xy'][case, 0]), 'cam_y': np.full((n, 1), k['cam_xy'][case, 1]),
            'cam_phi': rows[:, 1:2], 'cam_theta': rows[:, 2:3], 'tgt_x': rows[:, 4:5], 'tgt_y': rows[:, 5:6]})
        # the engine's transmittance is 0.1; a case recorded with tau = 0 never sees through: u = 0
        u = np.where(rows[:, 7] > 0, rows[:, 6], 0.0)
        eng.observe(tape_ct=torch.from_numpy(u.reshape(n, 1, 1).copy()).cuda())
        got = eng.unpack_masks()['camera_target_view_mask'][:, 0, 0]
        # the recorded sight range is sqrt(area / theta), which is what the device derives from theta
        assert np.allclose(np.sqrt(30.0 * k['cam_max_sight_range'][case] ** 2 / rows[:, 2]), rows[:, 3], rtol=1e-15)
        bad = got != rows[:, 8].astype(bool)
        wrong += int(bad.sum())
        total += n
        assert not bad.any(), (case, rows[bad][:4])
    assert total == len(cases) and wrong == 0


def test_kat_obstruct_on_device():
    """Obstacle.obstruct(ray, keep_tangential=True) (entities.py:158-184) as Target.simulate applies it: every recorded
    row (miss, tangent, grazing, head-on, origin inside, zero-length ray, exact touch, ...) is one environment with one
    target at the ray's origin, one obstacle, and the ray as the target's action (step size 100 > every |ray|)."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    rows = G.load('kat_obstruct.npz')['rows']
    rows = rows[(rows[:, 7] == 1.0) & (rows[:, 8] == 0.0)]       # keep_tangential, inner crossing: the step path
    n = len(rows)
    assert n > 1500
    cfg = read_config('MATE-Navigation.yaml', target={'location_random_range': [[-10.0, 10.0, -10.0, 10.0]], 'step_size': 100.0},
                      obstacle={'location_random_range': [[500.0, 600.0, 500.0, 600.0]], 'radius_random_range': [10.0, 20.0]},
                      high_capacity_target_split=0.0)
    eng = Engine(cfg, n, seed=2, obs_dtype=torch.float64)
    assert (eng.num_cameras, eng.num_targets, eng.num_obstacles) == (0, 1, 1)
    eng.reset()
    eng.load_state_dict({'tgt_x': rows[:, 0:1], 'tgt_y': rows[:, 1:2], 'obs_x': rows[:, 4:5], 'obs_y': rows[:, 5:6], 'obs_radius': rows[:, 6:7],
                         'tgt_capacity': np.ones((n, 1))})
    act = torch.from_numpy(rows[:, 2:4].reshape(n, 1, 2).copy()).cuda()
    eng.step(None, act, auto_reset=False)
    sd = eng.state_dict()
    want = np.clip(rows[:, 0:2] + rows[:, 9:11], -1000.0, 1000.0)
    got = np.concatenate([sd['tgt_x'], sd['tgt_y']], axis=1)
    err = np.abs(got - want).max(axis=1)
    assert err.max() < 1e-9, (err.max(), rows[err.argmax()])
    desired = rows[:, 0:2] + rows[:, 2:4]
    colliding = (np.abs(want - desired) > 1e-6).any(axis=1)
    margin = np.abs(np.abs(want - desired) - 1e-6).min(axis=1) > 1e-9     # rows whose verdict is not a rounding coin flip
    assert np.array_equal(sd['tgt_colliding'][:, 0].astype(bool)[margin], colliding[margin])


def test_kat_scalar_on_device():
    """normalize_angle (utils.py:155-158), the step clamp through the polar form (entities.py:648-650,
    utils.py:223-229) and Camera.simulate's clamps (entities.py:347-360) on the recorded vectors."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    k = G.load('kat_scalar.npz')
    # Camera.simulate: phi0, theta0, (dphi, dtheta) -> phi1, theta1, sight1
    rows = k['cam_sim']
    n = len(rows)
    eng = _single_camera_engine(1500.0, n)
    eng.load_state_dict({'cam_phi': rows[:, 0:1], 'cam_theta': rows[:, 1:2]})
    cam_act = torch.from_numpy(rows[:, 2:4].reshape(n, 1, 2).copy()).cuda()
    tgt_act = torch.zeros((n, 1, 2), dtype=torch.float64, device='cuda')
    co, _, _ = eng.step(cam_act, tgt_act, auto_reset=False)
    sd = eng.state_dict()
    assert np.array_equal(sd['cam_phi'][:, 0], rows[:, 4]) and np.array_equal(sd['cam_theta'][:, 0], rows[:, 5])
    # sight range through the observation: Camera.state = [x, y, r, Rs cos(phi), Rs sin(phi), theta, ...] at columns 13..
    obs = co[:, 0].cpu().numpy()
    sight = np.hypot(obs[:, 16], obs[:, 17])
    assert np.abs(sight - rows[:, 6]).max() < 1e-9
    # normalize_angle: an un-normalised orientation + a zero action comes back normalised
    angles = k['angles']
    n = len(angles)
    eng = _single_camera_engine(1500.0, n)
    eng.load_state_dict({'cam_phi': angles.reshape(n, 1)})
    eng.step(torch.zeros((n, 1, 2), dtype=torch.float64, device='cuda'), torch.zeros((n, 1, 2), dtype=torch.float64, device='cuda'), auto_reset=False)
    got = eng.state_dict()['cam_phi'][:, 0]
    assert np.array_equal(got, k['normalized']), np.argwhere(got != k['normalized'])[:5]
    # step clamp: |a| > v -> v * (cos, sin)(atan2(a)); no obstacle in reach
    acts, vs, want = k['clamp_action'], k['clamp_step'], k['clamp_out']
    n = len(acts)
    cfg = read_config('MATE-Navigation.yaml', target={'location_random_range': [[-10.0, 10.0, -10.0, 10.0]]},
                      obstacle={'location_random_range': [[800.0, 900.0, 800.0, 900.0]], 'radius_random_range': [10.0, 20.0]})
    eng = Engine(cfg, n, seed=3, obs_dtype=torch.float64)
    eng.reset()
    eng.load_state_dict({'tgt_x': np.zeros((n, 1)), 'tgt_y': np.zeros((n, 1)), 'tgt_capacity': (20.0 / vs).reshape(n, 1)})
    eng.step(None, torch.from_numpy(acts.reshape(n, 1, 2).copy()).cuda(), auto_reset=False)
    sd = eng.state_dict()
    got = np.concatenate([sd['tgt_x'], sd['tgt_y']], axis=1)
    assert np.abs(got - want).max() < 1e-12, np.abs(got - want).max()


def test_two_tier_table_launches_build_the_same_tables(monkeypatch):
    """The per-camera table launch runs with half-size sort arrays and defers larger tables to a full-size launch behind it
    (mate_engine.hip: setup_two_tier).  MATE_LUT_SMALL_CAP=512 makes most tables of a 9-obstacle scenario take the deferred path
    (they have ~550 rays, and the closing knot needs slot nr): tables, static masks and the first observations must be bit-identical
    to the default build's."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config('MATE-8v8-9.yaml')
    n = 300
    a = Engine(cfg, n, seed=41, first_env_index=9)
    a.enable_outer_boundary()                # the outer tables (up to 59 rays per obstacle more) go through the same launches
    monkeypatch.setenv('MATE_LUT_SMALL_CAP', '512')      # read when the reset layout is (re)computed: at creation and here
    b = Engine(cfg, n, seed=41, first_env_index=9)
    b.enable_outer_boundary()
    monkeypatch.delenv('MATE_LUT_SMALL_CAP')
    for eng in (a, b):
        eng.reset()
    torch.cuda.synchronize()
    assert torch.equal(a.camera_obs, b.camera_obs) and torch.equal(a.target_obs, b.target_obs) and torch.equal(a.masks, b.masks)
    deferred = 0
    for e in range(0, n, 7):
        for c in range(a.num_cameras):
            pa, ra = a.lut_read(e, c)
            pb, rb = b.lut_read(e, c)
            assert np.array_equal(pa, pb) and np.array_equal(ra, rb), (e, c)
            deferred += len(pa) >= 500
            qa, sa = a.lut_read(e, c, outer=True)
            qb, sb = b.lut_read(e, c, outer=True)
            assert np.array_equal(qa, qb) and np.array_equal(sa, sb), (e, c, 'outer')
    assert deferred > 20          # the deferred path was exercised
    sa, sb = a.state_dict(), b.state_dict()
    assert np.array_equal(sa['camera_obstacle_view_mask'], sb['camera_obstacle_view_mask'])


@pytest.mark.parametrize('rmax', [150.0, 1500.0])
def test_kat_obstruct_geometries_as_one_obstacle_tables(rmax, oracle_lib):
    """The NON-tangential variants of Obstacle.obstruct (entities.py:158-184: near crossing = the inner table, far crossing
    = `outer`) are what the occlusion-table builder applies to every ray (entities.py:450-455).  Every distinct (origin,
    circle) geometry of the non-tangential kat_obstruct rows -- miss, tangent, grazing, head-on, exact touch -- becomes one
    environment (one camera at the origin, that one obstacle): the device's tables against the oracle's builder, whose
    `obstruct` the CPU suite pins to those rows' recorded outputs.  Inner table knot for knot (only a knot AT a tangent
    direction may sit on the other side of the reference's coin flip); outer table as a piecewise-linear function (its
    flank points may merge with the arc's end ray, as in test_outer_boundary_*)."""
    O = oracle_lib
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    rows = G.load('kat_obstruct.npz')['rows']
    geo = np.unique(rows[rows[:, 7] == 0][:, [0, 1, 4, 5, 6]], axis=0)
    dist = np.hypot(geo[:, 2] - geo[:, 0], geo[:, 3] - geo[:, 1])
    geo = geo[dist > geo[:, 4]]                     # (a camera never stands inside an obstacle: Camera.overlap, entities.py:484-489)
    n = len(geo)
    assert n > 1200
    cfg = read_config('MATE-1v1-9.yaml', camera={'max_sight_range': float(rmax)}, obstacle={'location_random_range': [[-100, 100, -100, 100]]})
    eng = Engine(cfg, n, seed=2, obs_dtype=torch.float64)
    assert (eng.num_cameras, eng.num_targets, eng.num_obstacles) == (1, 1, 1)
    eng.enable_outer_boundary()
    eng.reset()
    eng.load_state_dict({'cam_x': geo[:, 0:1], 'cam_y': geo[:, 1:2], 'obs_x': geo[:, 2:3], 'obs_y': geo[:, 3:4], 'obs_radius': geo[:, 4:5]})
    eng.rebuild_luts()
    torch.cuda.synchronize()
    in_range = np.hypot(geo[:, 2] - geo[:, 0], geo[:, 3] - geo[:, 1]) < rmax + geo[:, 4]
    assert np.array_equal(eng.state_dict()['camera_obstacle_view_mask'][:, 0, 0] != 0, in_range)        # entities.py:365, strict
    grid = np.linspace(-180.0, 180.0, 7201)
    flips = clipped = 0
    outer_bad = []
    for e in range(n):
        obstacle = geo[e, 2:5].reshape(1, 3)
        gp, gr = eng.lut_read(e, 0)
        op, orr = O.build_lut(geo[e, 0:2], rmax, obstacle)
        assert len(gp) == len(op) and np.abs(gp - op).max() < 1e-9, (e, len(gp), len(op))
        flips += G.assert_only_tangent_flips(gp, gr, orr, geo[e, 0:2], rmax, obstacle, 1e-7, e)
        clipped += int((orr < rmax * (1 - 1e-9)).any())
        gp, gr = eng.lut_read(e, 0, outer=True)
        op, orr = O.build_lut(geo[e, 0:2], rmax, obstacle, outer=True)
        assert abs(len(gp) - len(op)) <= 2, (e, len(gp), len(op))
        diff = np.abs(np.interp(grid, gp, gr) - np.interp(grid, op, orr))
        # (one obstacle a few radii away subtends tens of degrees: a flank point merged on one side and not on the other moves
        # the function over ~1 degree of it; test_outer_boundary_* allow 0.2 % of the circle per obstacle on real scenarios)
        outer_bad.append(float((diff > 1e-6).mean()))
        assert outer_bad[-1] < 0.006, (e, (diff > 1e-6).sum(), diff.max())
    assert clipped >= 0.9 * in_range.sum()        # (`flips`: a tangent ray is a coin flip between two libms; a quarter of the tables has one)
    assert np.mean(np.asarray(outer_bad) > 0.002) < 0.2, np.sort(outer_bad)[-10:]       # ... and that is the minority (11 % observed)
