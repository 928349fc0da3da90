"""Helpers for the `-m gpu` parity tests: golden fixture / oracle <-> HIP engine."""
import numpy as np
import torch

import golden_util as G
from mate_amd.config import read_config
from mate_amd.engine import Engine

STATE_KEYS = ['cam_x', 'cam_y', 'obs_x', 'obs_y', 'obs_radius', 'tgt_capacity', 'camera_obstacle_view_mask', 'cam_phi',
              'cam_theta', 'tgt_x', 'tgt_y', 'tgt_colliding', 'tgt_empty_bits', 'tgt_goal_bits', 'tgt_goals', 'freights',
              'bounties', 'target_steps', 'tracked_steps', 'remaining_cargoes', 'awaiting_cargo_counts',
              'num_delivered_cargoes', 'episode_reward', 'delayed_episode_reward', 'episode_step']


def config_of_fixture(fx):
    return read_config(str(fx['config_file']), max_episode_steps=int(fx['max_episode_steps']))


def fixture_state(fx, prefix='reset/', index=None):
    st = G.static_of(fx)
    dyn = G.dynamic_of(fx, prefix, index)
    out = {k: np.asarray(st[k], dtype=np.float64) for k in STATE_KEYS if k in st}
    out.update({k: np.asarray(dyn[k], dtype=np.float64) for k in STATE_KEYS if k in dyn})
    return out


def engine_from_fixture(fx, num_envs=3, obs_dtype=torch.float64, seed=0):
    """Engine whose every environment holds the fixture's post-reset state and occlusion tables."""
    return load_fixture_state(Engine(config_of_fixture(fx), num_envs, seed=seed, obs_dtype=obs_dtype), fx)


def load_fixture_state(eng, fx):
    """Every environment of `eng` takes the fixture's post-reset state and occlusion tables."""
    num_envs = eng.num_envs
    st = fixture_state(fx)
    fields = {k: np.broadcast_to(v, (num_envs,) + v.shape) for k, v in st.items()}
    fields['tick'] = np.zeros(num_envs)
    fields['episode'] = np.ones(num_envs)
    fields['done'] = np.zeros(num_envs)
    # start from zeros: export of a fresh engine is all-zero state
    eng.was_loaded = True
    eng.import_state(torch.zeros((num_envs, eng.layout.export_width), dtype=torch.float64))
    eng.load_state_dict(fields)
    for c, (phis, rhos) in enumerate(G.luts_of(fx)):
        for e in range(num_envs):
            eng.lut_write(e, c, phis, rhos)
    return eng


def oracle_proto_from_config(cfg, O):
    """Oracle prototype env carrying the scenario (ranges, reward scales, ...)."""
    cam, tgt, obs = cfg.get('camera', {}), cfg['target'], cfg.get('obstacle', {})

    def ranges(sub):
        rows = [[x, x, y, y] for x, y in sub.get('location', [])] + [list(r) for r in sub.get('location_random_range', [])]
        return np.asarray(rows, dtype=np.float64).reshape(-1, 4)

    rc, rt, ro = ranges(cam), ranges(tgt), ranges(obs)
    env = O.OracleEnv(len(rc), len(rt), len(ro))
    env.set('cam_range', rc); env.set('tgt_range', rt); env.set('obs_range', ro)
    rr = obs.get('radius_random_range', [obs.get('radius', 0.0)] * 2)
    env.set('obs_radius_range', rr)
    env.set('transmittance', obs.get('transmittance', 0.0))
    env.set('max_episode_steps', cfg['max_episode_steps'])
    env.set('sparse_reward', cfg['reward_type'] == 'sparse')
    env.set('num_cargoes_per_target', cfg['num_cargoes_per_target'])
    env.set('shuffle_entities', cfg['shuffle_entities'])
    env.set('targets_start_with_cargoes', cfg['targets_start_with_cargoes'])
    env.set('high_capacity_target_split', cfg['high_capacity_target_split'])
    env.set('target_step_size', tgt['step_size']); env.set('target_sight_range', tgt['sight_range'])
    freight = np.ceil(2000.0 / tgt['step_size'])
    bounty = np.ceil(freight * max(0.0, cfg['bounty_factor']))
    env.set('freight_scale', freight); env.set('bounty_scale', bounty); env.set('reward_scale', freight + bounty)
    env.set('max_target_team_episode_reward', (freight + bounty) * cfg['num_cargoes_per_target'] * len(rt))
    env.set('cfg_cam_radius', cam.get('radius', 40.0)); env.set('cfg_cam_min_viewing_angle', cam.get('min_viewing_angle', 90.0))
    env.set('cfg_cam_max_sight_range', cam.get('max_sight_range', 500.0)); env.set('cfg_cam_rotation_step', cam.get('rotation_step', 5.0))
    env.set('cfg_cam_zooming_step', cam.get('zooming_step', 2.5))
    return env


def oracle_load_engine_state(oenv, sd, i, cfg):
    """Copy environment i of an Engine.state_dict() into an oracle env."""
    cam = cfg.get('camera', {})
    Nc = oenv.Nc
    for k in STATE_KEYS:
        oenv.set(k, sd[k][i])
    oenv.set('cam_radius', np.full(Nc, cam.get('radius', 40.0)))
    oenv.set('cam_min_viewing_angle', np.full(Nc, cam.get('min_viewing_angle', 90.0)))
    oenv.set('cam_max_sight_range', np.full(Nc, cam.get('max_sight_range', 500.0)))
    oenv.set('cam_rotation_step', np.full(Nc, cam.get('rotation_step', 5.0)))
    oenv.set('cam_zooming_step', np.full(Nc, cam.get('zooming_step', 2.5)))
    oenv.set('tgt_step_size', cfg['target']['step_size'] / sd['tgt_capacity'][i])
    oenv.set('tgt_sight_range', np.full(oenv.Nt, cfg['target']['sight_range']))
    theta_min, rmax = cam.get('min_viewing_angle', 90.0), cam.get('max_sight_range', 500.0)
    if Nc:
        oenv.set('cam_sight', np.sqrt(theta_min * rmax * rmax / sd['cam_theta'][i]))
    oenv.set('tick', sd['tick'][i]); oenv.set('episode', sd['episode'][i])
