"""Scenario configuration: the host-side counterpart of `read_config` / `validate_config`
(reference mate/environment.py:59-269).  Accepts the same inputs (None, a mapping, a path to
a YAML/JSON file, the bare name of a built-in asset) and applies the same defaults and
validation errors; random ranges are kept as plain `[x_lo, x_hi, y_lo, y_hi]` lists.
"""
import copy
import json
import os
import warnings
from collections.abc import Mapping
from pathlib import Path

from mate_amd import scenarios

__all__ = ['ASSETS_DIR', 'DEFAULT_CONFIG_FILE', 'read_config', 'validate_config']

ASSETS_DIR = Path(__file__).absolute().parent / 'assets'
DEFAULT_CONFIG_FILE = ASSETS_DIR / 'MATE-4v8-9.yaml'   # environment.py:38

NUM_WAREHOUSES = 4
CAMERA_DEFAULTS = {'radius': 40.0, 'min_viewing_angle': 90.0, 'max_sight_range': 500.0,
                   'rotation_step': 5.0, 'zooming_step': 2.5}          # entities.py:248-254
TARGET_DEFAULTS = {'sight_range': 500.0, 'step_size': 10.0}           # entities.py:563-566
QUIET = True


def _warn(message):
    if not QUIET:
        warnings.warn(message)


def _edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j - 1] + (ca != cb), prev[j] + 1, cur[j - 1] + 1))
        prev = cur
    return prev[-1]


def _did_you_mean(path):
    """Closest known configuration file name (environment.py:59-96)."""
    names = set(scenarios.scenario_names())
    for pattern in ('*.yaml', '*.yml', '*.json'):
        names.update(p.name for p in Path(os.getcwd()).glob(pattern))
    return sorted(names, key=lambda n: (_edit_distance(n, os.path.basename(str(path))), n))


def _deep_update(base, update, prefix=''):
    base, update = copy.deepcopy(base), copy.deepcopy(update)
    for key, value in update.items():
        if isinstance(base.get(key), dict) and isinstance(value, dict):
            value = _deep_update(base[key], value, prefix=f'{key}/')
        elif key in base:
            _warn(f'Override configuration "{prefix}{key}" with `{value!r}`.')
        else:
            _warn(f'Set configuration "{prefix}{key}" with `{value!r}`.')
        base[key] = value
    return base


def _load_file(path):
    ext = os.path.splitext(path)[1].lower()
    if ext not in ('.json', '.yaml', '.yml'):
        return None
    with open(path, encoding='UTF-8') as file:
        if ext == '.json':
            return json.load(file)
        import yaml
        return yaml.load(file, yaml.SafeLoader)


def _as_range(random_range):
    """dict(low, high) / flat [x_lo, x_hi, y_lo, y_hi] / objects with .low/.high -> flat list."""
    if hasattr(random_range, 'low') and hasattr(random_range, 'high'):
        low, high = list(random_range.low), list(random_range.high)
    elif isinstance(random_range, dict):
        low, high = list(random_range['low']), list(random_range['high'])
    else:
        flat = [float(v) for v in random_range]
        return flat
    out = []
    for lo, hi in zip(low, high):
        out += [float(lo), float(hi)]
    return out


def read_config(config_or_path=None, **kwargs):
    """Load a scenario from a mapping, a JSON/YAML file or a built-in asset name."""
    config = None
    if config_or_path is None:
        config = {}
    elif isinstance(config_or_path, Mapping):
        config = copy.deepcopy(dict(config_or_path))
    else:
        path = os.fspath(config_or_path) if isinstance(config_or_path, os.PathLike) else config_or_path
        if not isinstance(path, str):
            raise ValueError(f'The configuration should be a dictionary mapping or a path to a readable JSON/YAML file. Got {config_or_path!r}.')
        if os.path.exists(path):
            config = _load_file(path)
        else:
            for candidate in (Path(os.getcwd()) / path, ASSETS_DIR / path):
                if candidate.is_file():
                    config = _load_file(str(candidate))
                    break
            else:
                base = os.path.basename(path)
                if base in scenarios.SCENARIOS:      # built-in assets need no file on disk
                    config = scenarios.scenario(base)
                else:
                    candidates = _did_you_mean(path)
                    raise ValueError(f'Cannot found the configuration file "{path}". Did you mean: "{candidates[0]}"?')
        if config is None:
            raise ValueError(f'The configuration should be a dictionary mapping or a path to a readable JSON/YAML file. Got {config_or_path!r}.')

    config = _deep_update(config, kwargs)
    validate_config(config)
    for entity in ('camera', 'obstacle', 'target'):
        sub = config.setdefault(entity, {})
        if 'location' in sub:
            sub['location'] = [[float(v) for v in loc] for loc in sub['location']]
        if 'location_random_range' in sub:
            sub['location_random_range'] = [_as_range(r) for r in sub['location_random_range']]
        if 'radius_random_range' in sub:
            sub['radius_random_range'] = _as_range(sub['radius_random_range'])
    return config


def validate_config(config):
    """Defaults + validation, same rules and messages as environment.py:196-269."""
    if 'max_episode_steps' not in config:
        _warn('Missing key "max_episode_steps", set to 10000.')
        config['max_episode_steps'] = 10000
    if config['max_episode_steps'] <= 0:
        raise ValueError('`max_episode_steps` must be a positive integer.')
    if 'reward_type' not in config:
        _warn('Missing key "reward_type", set to "dense".')
        config['reward_type'] = 'dense'
    if config['reward_type'] not in ('dense', 'sparse'):
        raise ValueError(f'Invalid reward type {config["reward_type"]}. Expect one of {("dense", "sparse")}')
    if 'target' not in config:
        raise ValueError('Missing key "target". There must be at least one target in the environment.')
    target = config['target']
    if len(target.get('location', [])) + len(target.get('location_random_range', [])) == 0:
        raise ValueError('There must be at least one target in the environment.')
    if 'num_cargoes_per_target' not in config:
        raise ValueError('Missing key "num_cargoes_per_target".')
    if config['num_cargoes_per_target'] < NUM_WAREHOUSES:
        raise ValueError(f'`num_cargoes_per_target` should be no less than {NUM_WAREHOUSES}. Got {config["num_cargoes_per_target"]}.')
    if 'high_capacity_target_split' not in config:
        _warn('Missing key "high_capacity_target_split", set to 0.5.')
        config['high_capacity_target_split'] = 0.5
    if not 0.0 <= config['high_capacity_target_split'] <= 1.0:
        raise ValueError(f'`high_capacity_target_split` must be between 0 and 1. Got {config["high_capacity_target_split"]}.')
    if 'targets_start_with_cargoes' not in config:
        _warn('Missing key "targets_start_with_cargoes", set to True.')
        config['targets_start_with_cargoes'] = True
    config['targets_start_with_cargoes'] = bool(config['targets_start_with_cargoes'])
    if 'bounty_factor' not in config:
        _warn('Missing key "bounty_factor", set to 1.0.')
        config['bounty_factor'] = 1.0
    if not config['bounty_factor'] >= 0.0:
        raise ValueError(f'`bounty_factor` must be a non-negative number. Got {config["bounty_factor"]}.')
    if 'shuffle_entities' not in config:
        _warn('Missing key "shuffle_entities", set to True.')
        config['shuffle_entities'] = True
    config['shuffle_entities'] = bool(config['shuffle_entities'])
    for entity, defaults in (('camera', CAMERA_DEFAULTS), ('target', TARGET_DEFAULTS)):
        if entity in config:
            for key, default in defaults.items():
                if key not in config[entity]:
                    _warn(f'Missing key "{entity}/{key}", set to {default}.')
                    config[entity][key] = default
                if not config[entity][key] > 0.0:
                    raise ValueError(f'`{entity}/{key}` must be a positive number. Got {config[entity][key]}.')
