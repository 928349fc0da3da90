"""Helpers shared by the parity tests: load golden fixtures into an oracle env."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# (fixture key under reset/ or step/, oracle field)
DYNAMIC_FIELDS = [
    ('cam_phi', 'cam_phi'), ('cam_theta', 'cam_theta'), ('cam_sight', 'cam_sight'),
    ('tgt_colliding', 'tgt_colliding'), ('tgt_empty_bits', 'tgt_empty_bits'), ('tgt_goal_bits', 'tgt_goal_bits'),
    ('tgt_goals', 'tgt_goals'), ('freights', 'freights'), ('bounties', 'bounties'),
    ('target_steps', 'target_steps'), ('tracked_steps', 'tracked_steps'),
    ('remaining_cargoes', 'remaining_cargoes'), ('awaiting_cargo_counts', 'awaiting_cargo_counts'),
    ('num_delivered_cargoes', 'num_delivered_cargoes'), ('episode_reward', 'episode_reward'),
    ('delayed_episode_reward', 'delayed_episode_reward'), ('episode_step', 'episode_step'),
]
MASK_FIELDS = ['camera_target_view_mask', 'target_camera_view_mask', 'target_obstacle_view_mask',
               'target_target_view_mask', 'camera_camera_view_mask', 'tracked_bits']


def trace_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, 'trace_*.npz')))


def reset_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, 'reset_*.npz')))


def config_of_reset_fixture(fx):
    """The validated scenario mapping a reset fixture was recorded with (scenario file + keyword overrides)."""
    import json
    from mate_amd.config import read_config
    return read_config(str(fx['config_file']), **json.loads(str(fx['overrides'])))


# state a reset() determines, by the names of Engine.state_dict() / the oracle's fields: exact in tape mode
RESET_EXACT = ['cam_x', 'cam_y', 'obs_x', 'obs_y', 'obs_radius', 'tgt_capacity', 'cam_phi', 'cam_theta', 'tgt_x', 'tgt_y',
               'tgt_colliding', 'tgt_empty_bits', 'tgt_goal_bits', 'tgt_goals', 'freights', 'bounties', 'target_steps',
               'tracked_steps', 'remaining_cargoes', 'awaiting_cargo_counts', 'num_delivered_cargoes', 'episode_reward',
               'delayed_episode_reward', 'episode_step', 'camera_obstacle_view_mask']


def reset_expectation(fx):
    """name -> array the reference's reset() produced (fixture reset_*.npz)."""
    st, dyn = static_of(fx), dynamic_of(fx)
    out = {k: np.asarray(st[k], dtype=np.float64) for k in RESET_EXACT if k in st}
    out.update({k: np.asarray(dyn[k], dtype=np.float64) for k in RESET_EXACT if k in dyn})
    return out


def load(name):
    path = name if os.path.isabs(name) else os.path.join(GOLDEN_DIR, name)
    return dict(np.load(path))


def static_of(fx):
    """Static per-episode geometry of a trace fixture as plain arrays."""
    Nc, Nt, No = int(fx['num_cameras']), int(fx['num_targets']), int(fx['num_obstacles'])
    st = {
        'Nc': Nc, 'Nt': Nt, 'No': No,
        'transmittance': float(fx['transmittance']), 'max_episode_steps': int(fx['max_episode_steps']),
        'sparse_reward': int(bool(fx['sparse_reward'])), 'freight_scale': float(fx['freight_scale']),
        'bounty_scale': float(fx['bounty_scale']), 'reward_scale': float(fx['reward_scale']),
        'max_target_team_episode_reward': float(fx['max_target_team_episode_reward']),
        'target_step_size': float(fx['target_step_size']),
        'cam_x': fx['static/cam_xy'][:, 0], 'cam_y': fx['static/cam_xy'][:, 1], 'cam_radius': fx['static/cam_radius'],
        'cam_min_viewing_angle': fx['static/cam_min_viewing_angle'], 'cam_max_sight_range': fx['static/cam_max_sight_range'],
        'cam_rotation_step': fx['static/cam_rotation_step'], 'cam_zooming_step': fx['static/cam_zooming_step'],
        'obs_x': fx['static/obs_xyr'][:, 0], 'obs_y': fx['static/obs_xyr'][:, 1], 'obs_radius': fx['static/obs_xyr'][:, 2],
        'tgt_capacity': fx['static/tgt_capacity'], 'tgt_step_size': fx['static/tgt_step_size'],
        'tgt_sight_range': fx['static/tgt_sight_range'],
        'camera_obstacle_view_mask': fx['static/camera_obstacle_view_mask'],
    }
    return st


def dynamic_of(fx, prefix='reset/', index=None):
    def pick(key):
        v = fx[prefix + key]
        return v if index is None else v[index]
    dyn = {field: pick(key) for key, field in DYNAMIC_FIELDS}
    xy = pick('tgt_xy')
    dyn['tgt_x'], dyn['tgt_y'] = xy[:, 0], xy[:, 1]
    for m in MASK_FIELDS + ['target_dones', 'target_warehouse_distances']:
        dyn[m] = pick(m)
    return dyn


def luts_of(fx, outer=False):
    Nc = int(fx['num_cameras'])
    tag = 'lut_outer_' if outer else 'lut_'
    out = []
    for c in range(Nc):
        n = int(fx['static/' + tag + 'count'][c])
        out.append((fx['static/' + tag + 'phis'][c, :n].copy(), fx['static/' + tag + 'rhos'][c, :n].copy()))
    return out


SCALARS = ['transmittance', 'max_episode_steps', 'sparse_reward', 'freight_scale', 'bounty_scale', 'reward_scale',
           'max_target_team_episode_reward', 'target_step_size']


def oracle_from_fixture(fx, use_golden_lut=True):
    from oracle import oracle as O
    st = static_of(fx)
    env = O.OracleEnv(st['Nc'], st['Nt'], st['No'])
    for k in SCALARS:
        env.set(k, st[k])
    for k, v in st.items():
        if k in ('Nc', 'Nt', 'No') or k in SCALARS:
            continue
        env.set(k, v)
    if use_golden_lut:
        for c, (phis, rhos) in enumerate(luts_of(fx)):
            env.set_lut(c, phis, rhos)
        if 'static/lut_outer_count' in fx:
            for c, (phis, rhos) in enumerate(luts_of(fx, outer=True)):
                env.set_lut(c, phis, rhos, outer=True)
    else:
        env.build_luts()
    for k, v in dynamic_of(fx).items():
        env.set(k, v)
    return env


def assert_only_tangent_flips(phis, got, ref, cam_xy, rmax, obstacles_xyr, tol, where=None):
    """The occlusion-table tolerance, as narrow as the reference's own indeterminacy.  A ray EXACTLY tangent to an obstacle
    is clipped or not depending on the last bit of asin / atan2 (`radius > perpendicular`, entities.py:170), so a knot
    of `Camera.sight_range_func` may differ from the reference's -- but only a knot whose angle is a tangent direction
    `atan2(rel) +- asin(r / |rel|)` of an obstacle in range of that camera (entities.py:389-417: the arc's first / last
    ray, with the +-0.01 degree edge rays beside it), and at most two of them per obstacle.  Everything else must agree
    to `tol`."""
    phis, got, ref = np.asarray(phis, dtype=np.float64), np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    bad = np.flatnonzero(np.abs(got - ref) > tol)
    obstacles = np.asarray(obstacles_xyr, dtype=np.float64).reshape(-1, 3)
    assert len(bad) <= 2 * len(obstacles), (where, 'mismatching knots', len(bad))
    if not len(bad):
        return 0
    rel = obstacles[:, :2] - np.asarray(cam_xy, dtype=np.float64).reshape(1, 2)
    dist = np.hypot(rel[:, 0], rel[:, 1])
    in_range = (dist < rmax + obstacles[:, 2]) & (dist > obstacles[:, 2])          # Camera.add_obstacles, entities.py:365
    centre = np.degrees(np.arctan2(rel[in_range, 1], rel[in_range, 0]))
    half = np.degrees(np.arcsin(np.minimum(1.0, obstacles[in_range, 2] / dist[in_range])))
    tangents = np.concatenate([centre - half, centre + half])
    per_obstacle = np.zeros(int(in_range.sum()), dtype=int)
    for i in bad:
        off = np.abs((phis[i] - tangents + 180.0) % 360.0 - 180.0)
        j = int(np.argmin(off)) if len(off) else -1
        assert j >= 0 and off[j] <= 0.011, (where, 'knot', int(i), 'angle', float(phis[i]), 'differs by', float(abs(got[i] - ref[i])),
                                            'and is', float(off[j]) if j >= 0 else None, 'degrees from the nearest tangent direction')
        per_obstacle[j % len(per_obstacle)] += 1
    assert per_obstacle.max() <= 2 + 2, (where, per_obstacle)      # the tangent ray and its +-0.01 degree edge rays, both sides
    return len(bad)
