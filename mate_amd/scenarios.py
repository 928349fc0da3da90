"""Built-in scenarios (the data of the reference's `mate/assets/MATE-*.yaml` files).

The scenario numbers are restated programmatically: quadrant / edge / centre
location ranges are generated from their symmetry instead of being listed, and
`scenario(name)` returns the same nested mapping `yaml.safe_load` gives for the
corresponding upstream file (checked in tests/test_host_logic.py::test_scenarios_match_reference_assets against the
reference when it is present).
"""
import copy

__all__ = ['SCENARIOS', 'scenario', 'scenario_names']


def _quadrants(lo, hi):
    """(+,+), (+,-), (-,-), (-,+) boxes [x_lo, x_hi, y_lo, y_hi]."""
    return [[lo, hi, lo, hi], [lo, hi, -hi, -lo], [-hi, -lo, -hi, -lo], [-hi, -lo, lo, hi]]


def _edges(at, half):
    """Boxes pinned to the four sides of the terrain: east, north, west, south."""
    return [[at, at, -half, half], [-half, half, at, at], [-at, -at, -half, half], [-half, half, -at, -at]]


def _axis_boxes(lo, hi, half):
    """Boxes on the +x, +y, -x, -y axes."""
    return [[lo, hi, -half, half], [-half, half, lo, hi], [-hi, -lo, -half, half], [-half, half, -hi, -lo]]


def _centre(n, half=200):
    return [[-half, half, -half, half] for _ in range(n)]


_CAMERA_COMMON = {'min_viewing_angle': 30.0, 'max_sight_range': 1500.0, 'rotation_step': 5.0, 'zooming_step': 2.5, 'radius': 40.0}
_NINE_OBSTACLES = {
    'location_random_range': _quadrants(200, 800) + _edges(900, 500) + _centre(1),
    'radius_random_range': [25.0, 100.0],
    'transmittance': 0.1,
}


def _cameras(n):
    if n == 1:
        return dict(location=[[0, 0]], **_CAMERA_COMMON)
    if n == 2:
        return dict(location=[[-300, -300], [300, 300]], **_CAMERA_COMMON)
    if n == 4:
        return dict(location_random_range=_quadrants(500, 800), **_CAMERA_COMMON)
    if n == 8:
        cfg = dict(location_random_range=_quadrants(700, 850) + _axis_boxes(500, 600, 100), **_CAMERA_COMMON)
        cfg['max_sight_range'] = 1000.0
        return cfg
    raise ValueError(n)


def _tracking(nc, nt, no, explicit_defaults=True):
    cfg = {
        'name': f'MultiAgentTracking({nc}v{nt}, {no})',
        'max_episode_steps': 10000,
        'num_cargoes_per_target': 8,
        'high_capacity_target_split': 0.5,
        'targets_start_with_cargoes': True,
        'bounty_factor': 1.0,
        'shuffle_entities': True,
        'reward_type': 'dense',
        'camera': _cameras(nc),
        'target': {'location_random_range': _centre(nt), 'step_size': 20.0, 'sight_range': 500.0},
    }
    if no:
        cfg['obstacle'] = copy.deepcopy(_NINE_OBSTACLES)
    if not explicit_defaults:  # the two 1v1 files leave some keys to validate_config's defaults
        del cfg['high_capacity_target_split']
        if not no:
            del cfg['shuffle_entities']
    return cfg


def _navigation():
    return {
        'name': 'MultiAgentTracking(0v8, 32)',
        'max_episode_steps': 10000,
        'num_cargoes_per_target': 8,
        'high_capacity_target_split': 0.5,
        'targets_start_with_cargoes': False,
        'shuffle_entities': True,
        'reward_type': 'sparse',
        'target': {'location_random_range': _centre(8), 'step_size': 20.0, 'sight_range': 500.0},
        'obstacle': {
            'location_random_range': _quadrants(200, 800) * 2 + _edges(900, 500) * 2 + _centre(8) + _centre(8, half=900),
            'radius_random_range': [25.0, 100.0],
            'transmittance': 0.1,
        },
    }


def _build():
    table = {}
    for nc, nt in ((1, 1), (1, 2), (2, 2), (2, 4), (4, 2), (4, 4), (4, 8), (8, 8)):
        for no in (0, 9):
            table[f'MATE-{nc}v{nt}-{no}.yaml'] = _tracking(nc, nt, no, explicit_defaults=(nc, nt) != (1, 1))
    table['MATE-Navigation.yaml'] = _navigation()
    table['MATE.yaml'] = _tracking(4, 8, 9)
    return table


SCENARIOS = _build()


def scenario_names():
    return sorted(SCENARIOS)


def scenario(name):
    """Deep copy of a built-in scenario; `name` is the upstream asset file name."""
    key = str(name)
    key = key.rsplit('/', 1)[-1]
    if key not in SCENARIOS and key + '.yaml' in SCENARIOS:
        key += '.yaml'
    return copy.deepcopy(SCENARIOS[key])
