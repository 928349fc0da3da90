"""CPU-only tests of the host side: scenario data, configuration reader, observation layout,
C-ABI library symbols, sharding helper.  Comparisons against the upstream reference run only where
/root/reference exists (the build container); they are skipped elsewhere."""
import ctypes
import glob
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = '/root/reference'
HAVE_REF = os.path.isdir(os.path.join(REFERENCE, 'mate'))


@pytest.fixture(scope='module')
def ref_mate():
    if not HAVE_REF:
        pytest.skip('upstream reference not available here')
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden', 'gymshim'))
    sys.path.insert(0, REFERENCE)
    import gym  # noqa: F401  (the build-owned shim)
    import mate
    return mate


def test_scenarios_match_reference_assets():
    if not HAVE_REF:
        pytest.skip('upstream reference not available here')
    import yaml
    from mate_amd import scenarios
    files = sorted(glob.glob(os.path.join(REFERENCE, 'mate', 'assets', '*.yaml')))
    assert len(files) == len(scenarios.SCENARIOS) == 18
    for f in files:
        with open(f) as fh:
            assert yaml.safe_load(fh) == scenarios.scenario(os.path.basename(f)), f


def test_read_config_defaults_and_errors():
    from mate_amd.config import read_config
    cfg = read_config('MATE-1v1-0.yaml')
    assert cfg['shuffle_entities'] is True and cfg['high_capacity_target_split'] == 0.5
    assert cfg['camera']['location'] == [[0.0, 0.0]]
    cfg = read_config('MATE-Navigation.yaml')
    assert cfg['bounty_factor'] == 1.0 and cfg['reward_type'] == 'sparse' and len(cfg['obstacle']['location_random_range']) == 32
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=12, camera={'radius': 30.0})
    assert cfg['max_episode_steps'] == 12 and cfg['camera']['radius'] == 30.0 and cfg['camera']['max_sight_range'] == 1500.0
    with pytest.raises(ValueError, match='Did you mean'):
        read_config('MATE-4v8-8.yaml')
    with pytest.raises(ValueError, match='num_cargoes_per_target'):
        read_config('MATE-4v8-9.yaml', num_cargoes_per_target=3)
    with pytest.raises(ValueError, match='at least one target'):
        read_config({'num_cargoes_per_target': 8, 'target': {}})
    with pytest.raises(ValueError, match='max_episode_steps'):
        read_config('MATE-4v8-9.yaml', max_episode_steps=0)
    with pytest.raises(ValueError, match='reward type'):
        read_config('MATE-4v8-9.yaml', reward_type='shaped')


def test_read_config_matches_reference(ref_mate):
    from mate_amd.config import read_config
    from mate.environment import read_config as ref_read
    for name in ('MATE-4v2-9.yaml', 'MATE-4v8-9.yaml', 'MATE-8v8-9.yaml', 'MATE-4v8-0.yaml', 'MATE-Navigation.yaml', 'MATE-1v1-0.yaml'):
        mine, ref = read_config(name), ref_read(name)
        for key in ('max_episode_steps', 'reward_type', 'num_cargoes_per_target', 'high_capacity_target_split',
                    'targets_start_with_cargoes', 'bounty_factor', 'shuffle_entities'):
            assert mine[key] == ref[key], (name, key)
        for ent in ('camera', 'target', 'obstacle'):
            boxes = ref[ent].get('location_random_range', [])
            got = mine[ent].get('location_random_range', [])
            assert len(boxes) == len(got)
            for b, g in zip(boxes, got):
                assert [b.low[0], b.high[0], b.low[1], b.high[1]] == g
            for k, v in ref[ent].items():
                if k not in ('location', 'location_random_range', 'radius_random_range'):
                    assert mine[ent][k] == v, (name, ent, k)


def test_observation_layout_matches_reference(ref_mate):
    from mate import constants as R
    from mate_amd import constants as C
    for nums in ((4, 2, 9), (4, 8, 9), (8, 8, 9), (4, 8, 0), (0, 8, 32), (1, 1, 0)):
        for mine, ref in ((C.camera_observation_space_of(*nums), R.camera_observation_space_of(*nums)),
                          (C.target_observation_space_of(*nums), R.target_observation_space_of(*nums))):
            assert np.array_equal(mine.low, ref.low) and np.array_equal(mine.high, ref.high)
        assert np.array_equal(C.camera_observation_indices_of(*nums), R.camera_observation_indices_of(*nums))
        assert np.array_equal(C.target_observation_indices_of(*nums), R.target_observation_indices_of(*nums))
        assert C.camera_observation_slices_of(*nums) == R.camera_observation_slices_of(*nums)
        assert C.target_observation_slices_of(*nums) == R.target_observation_slices_of(*nums)
        assert np.array_equal(C.camera_coordinate_mask_of(*nums), R.camera_coordinate_mask_of(*nums))
        assert np.array_equal(C.target_coordinate_mask_of(*nums), R.target_coordinate_mask_of(*nums))
    assert np.array_equal(C.WAREHOUSES, R.WAREHOUSES) and C.WAREHOUSE_RADIUS == R.WAREHOUSE_RADIUS
    assert np.array_equal(C.PRESERVED_SPACE.high, R.PRESERVED_SPACE.high)


def test_c_abi_library_exports_every_declared_symbol():
    """The in-tree HIP library loads without a GPU and exports exactly what include/mate_engine.h declares."""
    from mate_amd import _native
    assert os.path.exists(_native.LIB_PATH), 'run __graft_entry__.build() first'
    with open(os.path.join(ROOT, 'include', 'mate_engine.h')) as fh:
        header = fh.read()
    declared = set(re.findall(r'\b(mate_engine_[a-z_]+)\s*\(', header))
    assert declared == set(_native.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.mate_engine_abi_version.restype = ctypes.c_int
    assert lib.mate_engine_abi_version() == 1


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from mate_amd import _native
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    with pytest.raises((_native.EngineError, RuntimeError, AssertionError)):
        Engine(read_config('MATE-4v8-9.yaml'), 4)


def test_shard_of_covers_the_batch():
    from mate_amd.distributed import shard_of
    for total, world in ((65536, 8), (32768, 8), (4096, 3), (10, 4), (7, 8)):
        spans = [shard_of(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1


def test_export_layout_is_dense():
    from mate_amd.engine import export_layout
    layout, width = export_layout(4, 8, 9)
    offs = sorted((off, int(np.prod(shape)) if shape else 1) for off, shape in layout.values())
    pos = 0
    for off, n in offs:
        assert off == pos
        pos += n
    assert pos == width == 242


def test_kernels_need_no_scratch_and_keep_full_occupancy():
    """The compiler's resource report of every kernel in the library (written by mate_amd/build.py): no private scratch
    memory anywhere (a kernel with scratch costs ~5 us more per launch, a third of the headline step), and the step
    kernels stay inside the register budget of 8 waves per SIMD (<= 96 SGPRs, <= 64 VGPRs, no spills)."""
    import json
    from mate_amd import build
    build.build_engine()
    if not os.path.exists(build.RESOURCES):      # library built before the report existed
        build.build_engine(force=True)
    with open(build.RESOURCES) as f:
        kernels = json.load(f)
    step = {k: r for k, r in kernels.items() if 'step_kernel' in k}
    assert len(step) >= 24 and any('soft_coverage' in k for k in kernels) and any('greedy_policy' in k for k in kernels)
    for name, r in kernels.items():
        assert r['ScratchSize'] == 0 and r['Dynamic Stack'] == 'False', name
    # SGPRs parked in VGPR lanes, by kernel: none anywhere but in the two folded flows of MATE-8v8-9, which sit at the SGPR limit
    # (9 in either flow today -- v_writelane / v_readlane pairs, no memory; two more would be a regression worth looking at)
    spill_allowance = {'FixedShapeILi8ELi8ELi9ELb0ELb0EEELi1E': 10, 'FixedShapeILi8ELi8ELi9ELb0ELb0EEELi2E': 10}
    for name, r in step.items():
        allowed = next((v for k, v in spill_allowance.items() if k in name), 0)
        assert r['TotalSGPRs'] <= 96 and r['VGPRs'] <= 64 and r['SGPRs Spill'] <= allowed and r['VGPRs Spill'] == 0 and r['Occupancy'] == 8, (name, r)
    # step_greedy_kernel (the per-step flows with the on-device agents): full occupancy for every compiled shape but MATE-8v8-9
    fused = {k: r for k, r in kernels.items() if 'step_greedy_kernel' in k}
    assert len(fused) >= 15
    for name, r in fused.items():
        assert r['VGPRs Spill'] == 0 and r['Occupancy'] >= (7 if ('Li8ELi8ELi9E' in name or 'AnyShape' in name) else 8), (name, r)


def test_auxiliary_target_rewards_on_a_replayed_trace():
    """mate_amd.auxiliary_rewards.AuxiliaryTargetRewards (the batched counterpart of the reference wrapper) on CPU tensors: a stand-in
    engine serves the state, step record and masks of a trace the reference's AuxiliaryTargetRewards shaped; every term but the
    soft coverage score (a device kernel, checked under -m gpu) and the shaped reward must match the wrapper's own numbers."""
    import numpy as np
    import torch
    import golden_util as G
    from mate_amd.auxiliary_rewards import AuxiliaryTargetRewards
    from mate_amd.engine import export_layout

    fx = G.load('auxtgt_4v8-9_s11.npz')
    Nc, Nt, No = int(fx['num_cameras']), int(fx['num_targets']), int(fx['num_obstacles'])
    N = 2

    class Replay:
        device = torch.device('cpu')
        num_envs, num_cameras, num_targets = N, Nc, Nt

        def __init__(self):
            self.export_fields, self.width = export_layout(Nc, Nt, No)
            self.step = -1

        def _row(self):
            row = np.zeros(self.width)
            pre = 'reset/' if self.step < 0 else 'step/'
            pick = (lambda k: fx[pre + k]) if self.step < 0 else (lambda k: fx[pre + k][self.step])

            def put(name, value):
                off, shape = self.export_fields[name]
                n = int(np.prod(shape)) if shape else 1
                row[off:off + n] = np.asarray(value, dtype=np.float64).reshape(n)
            put('tgt_x', pick('tgt_xy')[:, 0]); put('tgt_y', pick('tgt_xy')[:, 1])
            put('tgt_goals', pick('tgt_goals')); put('tgt_empty_bits', pick('tgt_empty_bits')); put('tgt_colliding', pick('tgt_colliding'))
            put('episode', 1)
            return row

        def export_state(self):
            return torch.from_numpy(np.broadcast_to(self._row(), (N, self.width)).copy())

        @property
        def scalars(self):
            s = self.step
            rec = [-fx['step/reward_tgt'][s], fx['step/reward_tgt'][s], 0.0, fx['step/coverage_rate'][s], fx['step/real_coverage_rate'][s],
                   fx['step/mean_transport_rate'][s], fx['step/num_delivered_cargoes'][s], 0.0]
            return torch.tensor([rec] * N, dtype=torch.float32)

        @property
        def masks(self):
            bits = fx['step/camera_target_view_mask'][self.step].reshape(-1)
            words = np.zeros((bits.size + 31) // 32 + 1, dtype=np.int64)
            for b in np.flatnonzero(bits):
                words[b >> 5] |= 1 << (b & 31)
            return torch.from_numpy(np.broadcast_to(words.astype(np.uint32).view(np.int32), (N, words.size)).copy())

    eng = Replay()
    keys = [str(k) for k in fx['auxt_keys'] if str(k) != 'soft_coverage_score']
    coef = dict(zip([str(k) for k in fx['auxt_keys']], [float(c) for c in fx['auxt_coefficients']]))
    shaper = AuxiliaryTargetRewards(eng, {k: coef[k] for k in keys}, 'none')
    soft = coef.get('soft_coverage_score', 0.0)
    delivered = 0
    for s in range(len(fx['step/done'])):
        eng.step = s
        shaped = shaper().numpy()
        for key in keys:
            loose = key in ('raw_reward', 'coverage_rate', 'real_coverage_rate', 'mean_transport_rate')
            np.testing.assert_allclose(shaper.terms[key].numpy()[1], fx['step/auxt_' + key][s], rtol=1e-6 if loose else 1e-12,
                                       atol=1e-6 if loose else 1e-12, err_msg=f'{key} step {s}')
        want = fx['step/aux_reward_tgt'][s] - soft * fx['step/auxt_soft_coverage_score'][s]
        np.testing.assert_allclose(shaped[0], want, rtol=1e-6, atol=1e-5, err_msg=str(s))
        delivered += int(fx['step/auxt_sparse_delivery'][s].sum())
    assert delivered == 4          # the trace delivers four cargoes: the delivery term was exercised
    with pytest.raises(AssertionError):
        AuxiliaryTargetRewards(eng, {'bogus': 1.0})
    with pytest.raises(AssertionError):
        AuxiliaryTargetRewards(eng, {'raw_reward': 1.0}, reduction='min')      # the reference's target wrapper has no 'min'


def test_zoom_table_interpolation_matches_the_iteration_over_the_whole_range():
    """The on-device GreedyCameraAgent reads the zoom solve of greedy.py:139-145 -- b <- K / (1 + sin(b / 2))^2, twenty times
    from 180 -- from a table over K = area_product / distance^2 at steps of 1 / 40 with cubic (Lagrange) interpolation
    (policy_enable in mate_engine.hip builds it, zoom_lookup in policy_kernels.hpp reads it; MATE_ZOOM_ITERATE=1 selects
    the iteration itself).  The same construction in NumPy against the iteration, densely over (0.025, 720) and right
    below the upper end of the solving branch: the interpolation error stays at the iteration's own rounding."""
    import numpy as np

    def iterate(K):
        b = np.full_like(K, 180.0)
        for _ in range(20):
            half = np.minimum(b * 0.5, 90.0)
            b = K / (1.0 + np.sin(np.deg2rad(half))) ** 2
        return b

    inv_h, n = 40.0, int(720.0 * 40.0) + 1            # nodes K = 0, 1/40, ..., 720: none beyond the kink of the 90-degree clamp
    table = iterate(np.arange(n) / inv_h)
    rng = np.random.default_rng(3)
    K = np.concatenate([rng.uniform(0.025, 720.0, 400000), np.linspace(719.0, 720.0 - 1e-9, 4001), np.linspace(0.025, 0.2, 2001)])
    x = K * inv_h
    i = x.astype(np.int64)
    inside = (i >= 1) & (i <= n - 3)                  # zoom_lookup iterates outside: K < 1/40, and the last cell below 720
    assert inside.mean() > 0.999 and (~inside).sum() > 0
    K, x, i = K[inside], x[inside], i[inside]
    t = x - i
    fm1, f0, f1, f2 = table[i - 1], table[i], table[i + 1], table[i + 2]
    tp1, tm1, tm2 = t + 1.0, t - 1.0, t - 2.0
    got = (t * tm1 * tm2 * (-1.0 / 6.0)) * fm1 + (tp1 * tm1 * tm2 * 0.5) * f0 + (tp1 * t * tm2 * (-0.5)) * f1 + (tp1 * t * tm1 * (1.0 / 6.0)) * f2
    err = np.abs(got - iterate(K))
    assert err.max() < 1e-11, err.max()               # (4e-12 for K < 10, where the function bends most; 1.5e-13 above K = 100)
    assert np.median(err) < 2e-13 and err[K > 100.0].max() < 5e-13


def test_every_environment_switch_is_documented_in_the_header():
    """Every MATE_* environment variable the library (getenv in csrc/) or the Python host (mate_amd/*.py) reads is listed in
    include/mate_engine.h -- the one place an integrator looks."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, 'include', 'mate_engine.h')).read()
    names = set()
    for name in os.listdir(os.path.join(root, 'mate_amd', 'csrc')):
        if not name.endswith(('.hip', '.hpp')):
            continue
        text = open(os.path.join(root, 'mate_amd', 'csrc', name)).read()
        names |= set(re.findall(r'getenv\("(MATE_[A-Z0-9_]+)"\)', text))
        names |= set(re.findall(r'flag\("(MATE_[A-Z0-9_]+)"\)', text))
    for name in ('engine.py', '_native.py', 'build.py', 'environment.py'):
        text = open(os.path.join(root, 'mate_amd', name)).read()
        names |= set(re.findall(r"""(?:environ|env)(?:\.get\(|\[)['"](MATE_[A-Z0-9_]+)['"]""", text))
        names |= set(re.findall(r"""['"](MATE_[A-Z0-9_]+)['"] in env""", text))
    assert len(names) >= 15, names
    missing = sorted(n for n in names if n not in header)
    assert not missing, missing


def test_auto_reset_spellings_of_the_python_mirror():
    """Engine._auto_reset_code: what the rollout entry points hand to the C ABI -- 'pipelined' is MATE_RESET_PIPELINED (-1),
    ('pipelined', m) is -m (one restart launch on the side stream behind every m-th rollout launch), everything else an int."""
    from mate_amd.engine import Engine
    assert Engine._auto_reset_code('pipelined') == Engine.RESET_PIPELINED == -1
    assert Engine._auto_reset_code(('pipelined', 12)) == -12 and Engine._auto_reset_code(('pipelined', 1)) == -1
    assert Engine._auto_reset_code(True) == 1 and Engine._auto_reset_code(False) == 0 and Engine._auto_reset_code(7) == 7
    # what is not one of those spellings is refused, not passed on: a typo such as -5 would silently defer restarts five launches
    for bad in (('pipelined', 0), ('pipelined', (1 << 16) + 1), ('piplined', 2), ('pipelined',), -5, -(1 << 31)):
        with pytest.raises(ValueError):
            Engine._auto_reset_code(bad)
