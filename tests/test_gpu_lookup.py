"""The occlusion lookup's fast paths against np.interp, probe by probe (round 4: pivot records, two record cells per degree).

Camera.perceive (entities.py:491-511) ends in `norm <= sight_range_at(angle) * (1 + 1e-6)` with sight_range_at = np.interp on the
camera's knots.  The engine answers it from per-cell records (normal cells: up to four segments; overflowing cells: pivot angles + a
bracket of knots; beyond 37 knots: narrowing levels; else the general search).  Here targets are PLACED around a camera -- at random
angles of its sector, at and beside knots of dense cells, at cell boundaries -- just inside and just outside the boundary the
camera's own table (read back from the device) draws, and the mask bit of every probe is compared with the oracle's Camera.perceive
on those knots."""
import numpy as np
import pytest
import torch

from mate_amd.config import read_config
from mate_amd.engine import Engine
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def dense_cells(phis, width=0.5, slots=4):
    """start angles of the cells of `width` degrees holding more knots than a record takes"""
    starts = np.arange(-180.0, 180.0, width)
    lo = np.searchsorted(phis, starts, side='left')
    hi = np.searchsorted(phis, starts + width, side='left')
    n = hi - lo + (phis[np.minimum(lo, len(phis) - 1)] != starts)
    return starts[n > slots]


@pytest.mark.parametrize('workload,envs', [('MATE-4v8-9.yaml', 192), ('MATE-8v8-9.yaml', 96), ('MATE-4v2-9.yaml', 128)])
def test_occlusion_lookups_equal_np_interp_probe_by_probe(workload, envs):
    cfg = read_config(workload)
    eng = Engine(cfg, envs, seed=11)
    eng.reset()
    Nc, Nt = eng.num_cameras, eng.num_targets
    cam_cfg = cfg['camera']
    area = cam_cfg['min_viewing_angle'] * cam_cfg['max_sight_range'] ** 2
    tau = float(cfg.get('obstacle', {}).get('transmittance', 0.0))
    rng = np.random.RandomState(7)
    sd = eng.state_dict()
    tables = [[eng.lut_read(e, c) for c in range(Nc)] for e in range(envs)]
    probes = aimed = in_dense = seen_total = 0
    for rnd in range(6):
        cam = rnd % Nc                      # the camera this round's probes are aimed at
        tx, ty = sd['tgt_x'].copy(), sd['tgt_y'].copy()
        for e in range(envs):
            cx, cy, phi, theta = sd['cam_x'][e, cam], sd['cam_y'][e, cam], sd['cam_phi'][e, cam], sd['cam_theta'][e, cam]
            sight = np.sqrt(area / theta)
            phis, rhos = tables[e][cam]
            half = 0.5 * theta - 0.6        # stay off the sector's edges: the verdict there is the angle test's, not the lookup's
            dense = dense_cells(phis)
            off = np.abs((dense + 0.25 - phi + 180.0) % 360.0 - 180.0)
            dense = dense[off < half]       # ... and the dense cells inside the sector
            for t in range(Nt):
                kind = rng.randint(4)
                if kind == 0 or len(dense) == 0:
                    a = phi + rng.uniform(-half, half)
                elif kind == 1:             # at / beside a knot of a dense cell
                    start = dense[rng.randint(len(dense))]
                    ks = np.nonzero((phis >= start) & (phis < start + 0.5))[0]
                    a = phis[ks[rng.randint(len(ks))]] + rng.choice([0.0, 1e-9, -1e-9, 1e-4, -1e-4])
                elif kind == 2:             # anywhere in a dense cell
                    a = dense[rng.randint(len(dense))] + rng.uniform(0.0, 0.5)
                else:                       # at a cell boundary
                    a = np.floor((phi + rng.uniform(-half, half)) * 2.0) / 2.0 + rng.choice([0.0, 1e-9, -1e-9])
                a = (a + 180.0) % 360.0 - 180.0
                rel = abs(phi - a); rel = min(rel, 360.0 - rel)
                if 2.0 * rel > theta - 0.03:
                    a = phi
                limit = float(np.interp(a, phis, rhos))
                d = min(limit, sight * 0.999) * rng.choice([0.999, 1.001, rng.uniform(0.3, 1.3)])
                x, y = cx + d * np.cos(np.radians(a)), cy + d * np.sin(np.radians(a))
                if abs(x) > 999.0 or abs(y) > 999.0 or d < 1.0:      # off the terrain: a probe well inside instead
                    d = min(limit, sight) * 0.5
                    x, y = cx + d * np.cos(np.radians(a)), cy + d * np.sin(np.radians(a))
                tx[e, t], ty[e, t] = x, y
        eng.load_state_dict({'tgt_x': tx, 'tgt_y': ty})
        eng.observe(tape_ct=torch.zeros((envs, Nc, Nt), dtype=torch.float64, device=eng.device))
        got = eng.unpack_masks()['camera_target_view_mask']
        now = eng.state_dict()
        assert np.array_equal(now['tgt_x'], tx) and np.array_equal(now['tgt_y'], ty)
        for e in range(envs):
            for c in range(Nc):
                cxy = (now['cam_x'][e, c], now['cam_y'][e, c])
                phi, theta = now['cam_phi'][e, c], now['cam_theta'][e, c]
                phis, rhos = tables[e][c]
                dense = dense_cells(phis)
                for t in range(Nt):
                    point = (tx[e, t], ty[e, t])
                    want = O.camera_perceive(cxy, phi, theta, np.sqrt(area / theta), point, 0.0, tau, phis, rhos)
                    assert bool(got[e, c, t]) == want, (workload, rnd, e, c, t, point)
                    probes += 1
                    if c == cam:
                        aimed += 1
                        seen_total += want
                        ang = np.degrees(np.arctan2(point[1] - cxy[1], point[0] - cxy[0]))
                        in_dense += bool(len(dense)) and bool(np.any((dense <= ang) & (ang < dense + 0.5)))
    # the probes did go where the slow paths are, and both verdicts occur
    assert in_dense > aimed // 10 and aimed // 10 < seen_total < aimed - aimed // 10, (probes, aimed, in_dense, seen_total)
