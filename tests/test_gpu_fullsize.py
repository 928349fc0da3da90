"""Properties at the full BASELINE.json sizes (one GPU's share; the oracle is too slow to follow here, so these are
size-independent invariants of the domain): sharding invariance, determinism, export -> import round trips,
mask <-> observation consistency, physical invariants, cargo conservation, fused rollout == single steps, sorted
occlusion tables."""
import numpy as np
import pytest
import torch

from mate_amd.config import read_config
from mate_amd.engine import Engine

pytestmark = pytest.mark.gpu

FULL = [('MATE-4v8-9.yaml', 4096), ('MATE-8v8-9.yaml', 8192), ('MATE-4v8-0.yaml', 65536), ('MATE-Navigation.yaml', 32768)]
IDS = ['C2-4v8-9x4096', 'C3-8v8-9x8192', 'C4-4v8-0x65536', 'C5-navx32768']


def outputs(eng):
    parts = [eng.target_obs, eng.scalars, eng.masks] + ([eng.camera_obs] if eng.num_cameras else [])
    return [p.clone() for p in parts]


def same(a, b):
    return all(torch.equal(x.view(torch.uint8), y.view(torch.uint8)) for x, y in zip(a, b))


@pytest.mark.parametrize('workload,n', FULL, ids=IDS)
def test_sharding_and_determinism(workload, n):
    cfg = read_config(workload)
    whole = Engine(cfg, n, seed=5)
    again = Engine(cfg, n, seed=5)
    half = n // 2
    shards = [Engine(cfg, half, seed=5, first_env_index=0), Engine(cfg, n - half, seed=5, first_env_index=half)]
    for e in [whole, again] + shards:
        e.reset()
    for _ in range(6):
        for e in [whole, again] + shards:
            e.step_random(auto_reset=True, want_masks=True)
    ref = outputs(whole)
    assert same(ref, outputs(again))                                       # same seed, same bits
    parts = [outputs(s) for s in shards]
    joined = [torch.cat([a, b], dim=0) for a, b in zip(*parts)]
    assert same(ref, joined)                                               # a sharded batch is the same batch
    assert torch.equal(whole.export_state(), torch.cat([s.export_state() for s in shards], dim=0))


@pytest.mark.parametrize('workload,n', FULL, ids=IDS)
def test_invariants_conservation_and_consistency(workload, n):
    cfg = read_config(workload)
    eng = Engine(cfg, n, seed=9)
    eng.reset()
    Nc, Nt, No = eng.num_cameras, eng.num_targets, eng.num_obstacles
    total_cargo = cfg['num_cargoes_per_target'] * Nt
    sd0 = eng.state_dict()
    for _ in range(25):
        eng.step_random(auto_reset=False, want_masks=True)
    sd = eng.state_dict()
    # physics: terrain, angles, no target inside an obstacle or a camera body
    assert np.abs(sd['tgt_x']).max() <= 1000.0 and np.abs(sd['tgt_y']).max() <= 1000.0
    if Nc:
        assert sd['cam_phi'].min() >= -180.0 and sd['cam_phi'].max() < 180.0
        assert sd['cam_theta'].min() >= cfg['camera']['min_viewing_angle'] and sd['cam_theta'].max() <= 180.0
        d = np.hypot(sd['tgt_x'][:, :, None] - sd['cam_x'][:, None, :], sd['tgt_y'][:, :, None] - sd['cam_y'][:, None, :])
        assert (d >= cfg['camera']['radius'] - 1e-6).all()
    if No:
        d = np.hypot(sd['tgt_x'][:, :, None] - sd['obs_x'][:, None, :], sd['tgt_y'][:, :, None] - sd['obs_y'][:, None, :])
        assert (d >= sd['obs_radius'][:, None, :] - 1e-6).all()
    assert np.array_equal(sd['cam_x'], sd0['cam_x']) and np.array_equal(sd['obs_x'], sd0['obs_x'])     # static geometry untouched
    # cargo conservation (environment.py:768-775, 1286-1315): in warehouses + in transit + delivered == all cargo
    carried = sd['tgt_goal_bits'].sum(axis=(1, 2))
    assert np.array_equal(sd['remaining_cargoes'].sum(axis=(1, 2)) + carried + sd['num_delivered_cargoes'], np.full(n, float(total_cargo)))
    assert np.array_equal(sd['awaiting_cargo_counts'].sum(axis=1) + sd['num_delivered_cargoes'], np.full(n, float(total_cargo)))
    assert (sd['bounties'] >= 0).all() and (sd['episode_step'] == 25).all()
    # masks <-> observation rows: every visibility flag column of the packed rows is the mask bit
    m = eng.unpack_masks()
    to = eng.target_obs.cpu().numpy()
    off = 27
    if Nc:
        assert np.array_equal(to[:, :, off:off + 7 * Nc].reshape(n, Nt, Nc, 7)[..., 6] != 0, m['target_camera_view_mask'])
    off += 7 * Nc
    if No:
        assert np.array_equal(to[:, :, off:off + 4 * No].reshape(n, Nt, No, 4)[..., 3] != 0, m['target_obstacle_view_mask'])
    off += 4 * No
    assert np.array_equal(to[:, :, off:off + 5 * Nt].reshape(n, Nt, Nt, 5)[..., 4] != 0, m['target_target_view_mask'])
    assert m['target_target_view_mask'][:, np.arange(Nt), np.arange(Nt)].all()
    if Nc:
        co = eng.camera_obs.cpu().numpy()
        assert np.array_equal(co[:, :, 22:22 + 5 * Nt].reshape(n, Nc, Nt, 5)[..., 4] != 0, m['camera_target_view_mask'])
        assert np.array_equal(m['tracked_bits'], m['camera_target_view_mask'].any(axis=1))
        hidden = ~m['camera_target_view_mask']
        assert (co[:, :, 22:22 + 5 * Nt].reshape(n, Nc, Nt, 5)[hidden] == 0).all()                      # hidden blocks are exact zeros
    sc = eng.scalars.cpu().numpy()
    assert np.allclose(sc[:, 3], m['tracked_bits'].mean(axis=1) if Nc else 0.0, atol=1e-6)                # coverage_rate
    assert np.array_equal(sc[:, 0], -sc[:, 1])                                                           # zero-sum team rewards


@pytest.mark.parametrize('workload,n', FULL[:2] + FULL[3:], ids=IDS[:2] + IDS[3:])
def test_export_import_round_trip_and_fused_rollout(workload, n):
    cfg = read_config(workload)
    a = Engine(cfg, n, seed=13)
    a.reset()
    for _ in range(3):
        a.step_random(auto_reset=False)
    b = Engine(cfg, n, seed=13)
    b.import_state(a.export_state())
    if a.num_cameras:
        b.rebuild_luts()                      # tables follow from the imported geometry
    for _ in range(4):
        a.step_random(auto_reset=False, want_masks=True)
        b.step_random(auto_reset=False, want_masks=True)
    assert same(outputs(a), outputs(b))
    # K fused steps in one launch == K launches
    c = Engine(cfg, n, seed=13)
    c.import_state(a.export_state())
    if a.num_cameras:
        c.rebuild_luts()
    K = 5
    rows = c.rollout_random(K, auto_reset=False)
    for _ in range(K):
        a.step_random(auto_reset=False)
    assert torch.equal(rows[1][K - 1], a.target_obs) and torch.equal(c.export_state(), a.export_state())


def test_occlusion_tables_are_sorted_and_closed():
    cfg = read_config('MATE-8v8-9.yaml')
    eng = Engine(cfg, 8192, seed=3)
    eng.reset()
    rng = np.random.RandomState(0)
    for e in rng.randint(0, 8192, size=24):
        for c in range(eng.num_cameras):
            phis, rhos = eng.lut_read(int(e), c)
            assert phis[0] == -180.0 and phis[-1] == phis[0] + 360.0 and rhos[-1] == rhos[0]            # entities.py:470-471
            assert np.all(np.diff(phis) > 0) and 361 <= len(phis) <= eng.layout.lut_capacity
            assert rhos.min() >= 0.0 and rhos.max() <= cfg['camera']['max_sight_range']


def test_greedy_workload_sharding_and_progress():
    """BASELINE config 3 at full size: the on-device Greedy teams give the same batch whether it runs whole or as two
    shards (their Philox streams are keyed by the global environment index), targets do deliver cargo, cameras do
    track, and the batched auto-reset accounts for every idle slot."""
    cfg = read_config('MATE-8v8-9.yaml', max_episode_steps=300)
    n = 8192
    engines = [Engine(cfg, n, seed=17), Engine(cfg, n // 2, seed=17, first_env_index=0), Engine(cfg, n // 2, seed=17, first_env_index=n // 2)]
    for e in engines:
        e.enable_policies()
        e.reset()
    for _ in range(40):
        for e in engines:
            e.step_greedy(auto_reset=False)
    whole, a, b = (outputs(e) for e in engines)
    assert same(whole, [torch.cat([x, y], dim=0) for x, y in zip(a, b)])
    eng = engines[0]
    idle0 = eng.idle_steps()
    steps = 330
    for _ in range(steps):
        eng.step_greedy(auto_reset=32)
    sd = eng.state_dict()
    assert (sd['episode'] >= 2).all()                                  # the time limit ended every first episode
    assert float(eng.scalars[:, 3].mean()) > 0.2                       # coverage the greedy cameras keep
    idle = eng.idle_steps() - idle0
    assert 0 < idle < 0.12 * n * steps                                 # finished environments waited for the next batched reset


@pytest.mark.parametrize('K,launches,resets,store_form', [(32, 2, 1, 'auto'), (20, 7, 6, '0'), (20, 7, 6, '1')])
def test_benchmark_flow_equals_single_steps_at_the_headline_size(K, launches, resets, store_form):
    """bench.py's flows at MATE-4v8-9 x 4096 -- fused 32-step launches with a restart launch behind each, and the DRIVER's own
    launch shape: `--steps 20` = one 20-step launch per region, finished environments restarted after every 6th launch, in either
    form of the row stores (Engine.reserve_rollout picks one per box from the probed store rate; MATE_STORE_FORM forces it) --
    against single-step launches: every row of both observation blocks, the scalar records and the masks, then the state, bit
    for bit."""
    import os
    cfg = read_config('MATE-4v8-9.yaml')
    n = 4096
    os.environ['MATE_STORE_FORM'] = store_form
    try:
        a = Engine(cfg, n, seed=21)
    finally:
        os.environ.pop('MATE_STORE_FORM', None)
    b = Engine(cfg, n, seed=21)
    a.reset(); b.reset()
    for launch in range(launches):
        cam, tgt, sc = a.rollout_random(K, auto_reset=resets, want_masks=True)
        assert a.last_flow == 1 and (store_form == 'auto' or a.store_form == int(store_form))
        for r in range(K):
            b.step_random(auto_reset=False, want_masks=True)
            assert torch.equal(cam[r], b.camera_obs) and torch.equal(tgt[r], b.target_obs), (launch, r)
            assert torch.equal(sc[r], b.scalars) and torch.equal(a._rollout['masks'][r], b.masks), (launch, r)
        assert not bool((sc[:, :, 2] != 0).any())           # nothing ends this early: no idle slot, the restart launches find nothing
        assert torch.equal(a.export_state(), b.export_state())


def test_fused_greedy_rollout_equals_single_steps_at_config3_size():
    """BASELINE config 3 (MATE-8v8-9 x 8192, Greedy vs Greedy): one fused 8-step launch == 8 x (policy launch + step launch)."""
    cfg = read_config('MATE-8v8-9.yaml')
    n, K = 8192, 8
    a = Engine(cfg, n, seed=5)
    b = Engine(cfg, n, seed=5)
    for e in (a, b):
        e.enable_policies()
        e.reset()
    cam, tgt, sc = a.rollout_greedy(K, auto_reset=True)
    for r in range(K):
        b.step_greedy(auto_reset=10 ** 6)
        assert torch.equal(cam[r], b.camera_obs) and torch.equal(tgt[r], b.target_obs) and torch.equal(sc[r], b.scalars), r
    assert torch.equal(a.export_state(), b.export_state())
