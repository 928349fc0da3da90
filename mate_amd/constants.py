"""World constants and the observation layout (public names of the reference's mate/constants.py).

Observation rows (constants.py:194-300 of the reference, restated):
  camera: [preserved 13 | own private state 9 | Nt x (target public 4 + flag) | No x (obstacle 3 + flag) | Nc x (camera public 6 + flag)]
  target: [preserved 13 | own private state 14 | Nc x (camera public 6 + flag) | No x (obstacle 3 + flag) | Nt x (target public 4 + flag)]
"""
import functools

import numpy as np

from mate_amd import spaces
from mate_amd.utils import Team

TERRAIN_SIZE = 1000.0
TERRAIN_WIDTH = 2.0 * TERRAIN_SIZE
WAREHOUSE_RADIUS = 0.075 * TERRAIN_SIZE
WAREHOUSES = (TERRAIN_SIZE - WAREHOUSE_RADIUS) * np.array([[+1.0, +1.0], [-1.0, +1.0], [-1.0, -1.0], [+1.0, -1.0]])
NUM_WAREHOUSES = len(WAREHOUSES)
MAX_CAMERA_VIEWING_ANGLE = 180.0
TARGET_RADIUS = 0.0

PRESERVED_DIM = 3 + 1 + 2 * NUM_WAREHOUSES + 1
OBSERVATION_OFFSET = PRESERVED_DIM
CAMERA_STATE_DIM_PUBLIC, CAMERA_STATE_DIM_PRIVATE = 6, 9
TARGET_STATE_DIM_PUBLIC, TARGET_STATE_DIM_PRIVATE = 4, 6 + 2 * NUM_WAREHOUSES
OBSTACLE_STATE_DIM = 3
CAMERA_ACTION_DIM = TARGET_ACTION_DIM = 2
CAMERA_DEFAULT_ACTION = np.zeros(2)
TARGET_DEFAULT_ACTION = np.zeros(2)

_W2, _INF = TERRAIN_WIDTH, np.inf


def _box(low, high):
    return spaces.Box(low=np.asarray(low, dtype=np.float64), high=np.asarray(high, dtype=np.float64), dtype=np.float64)


# bounds per block: (low, high) lists
_TERRAIN_LO, _TERRAIN_HI = [-TERRAIN_SIZE] * 2, [TERRAIN_SIZE] * 2
_PRESERVED = ([0.0] * 4 + [-_W2] * (2 * NUM_WAREHOUSES) + [0.0], [_INF] * 4 + [_W2] * (2 * NUM_WAREHOUSES) + [TERRAIN_SIZE])
_CAM_PUB = ([-_W2, -_W2, 0.0, -_W2, -_W2, 0.0], [_W2, _W2, TERRAIN_SIZE, _W2, _W2, MAX_CAMERA_VIEWING_ANGLE])
_CAM_PRIV = (_CAM_PUB[0] + [0.0] * 3, _CAM_PUB[1] + [_W2, MAX_CAMERA_VIEWING_ANGLE, MAX_CAMERA_VIEWING_ANGLE])
_TGT_PUB = ([-_W2, -_W2, 0.0, -1.0], [_W2, _W2, _W2, 1.0])
_TGT_PRIV = (_TGT_PUB[0] + [0.0, 1.0] + [0.0] * NUM_WAREHOUSES + [-1.0] * NUM_WAREHOUSES,
             _TGT_PUB[1] + [_W2, 2.0] + [_INF] * NUM_WAREHOUSES + [1.0] * NUM_WAREHOUSES)
_OBS = ([-_W2, -_W2, 0.0], [_W2, _W2, TERRAIN_SIZE])

TERRAIN_SPACE = _box(_TERRAIN_LO, _TERRAIN_HI)
PRESERVED_SPACE = _box(*_PRESERVED)
CAMERA_STATE_SPACE_PUBLIC, CAMERA_STATE_SPACE_PRIVATE = _box(*_CAM_PUB), _box(*_CAM_PRIV)
TARGET_STATE_SPACE_PUBLIC, TARGET_STATE_SPACE_PRIVATE = _box(*_TGT_PUB), _box(*_TGT_PRIV)
OBSTACLE_STATE_SPACE = _box(*_OBS)


def _flagged(bounds, reps):
    return [(bounds[0] + [-1.0]) * reps, (bounds[1] + [1.0]) * reps]


def _row_space(private, blocks):
    low, high = list(_PRESERVED[0]) + list(private[0]), list(_PRESERVED[1]) + list(private[1])
    for bounds, reps in blocks:
        lo, hi = _flagged(bounds, reps)
        low += lo
        high += hi
    return _box(low, high)


@functools.lru_cache(maxsize=None)
def camera_observation_space_of(num_cameras, num_targets, num_obstacles):
    return _row_space(_CAM_PRIV, [(_TGT_PUB, num_targets), (_OBS, num_obstacles), (_CAM_PUB, num_cameras)])


@functools.lru_cache(maxsize=None)
def target_observation_space_of(num_cameras, num_targets, num_obstacles):
    return _row_space(_TGT_PRIV, [(_CAM_PUB, num_cameras), (_OBS, num_obstacles), (_TGT_PUB, num_targets)])


def observation_space_of(team, num_cameras, num_targets, num_obstacles):
    fn = (camera_observation_space_of, target_observation_space_of)[team.value]
    return fn(num_cameras, num_targets, num_obstacles)


def _block_widths(team, nc, nt, no):
    cam, tgt, obs = nc * (CAMERA_STATE_DIM_PUBLIC + 1), nt * (TARGET_STATE_DIM_PUBLIC + 1), no * (OBSTACLE_STATE_DIM + 1)
    if team is Team.CAMERA:
        return [PRESERVED_DIM, CAMERA_STATE_DIM_PRIVATE, tgt, obs, cam], (TARGET_STATE_DIM_PUBLIC, CAMERA_STATE_DIM_PUBLIC)
    return [PRESERVED_DIM, TARGET_STATE_DIM_PRIVATE, cam, obs, tgt], (CAMERA_STATE_DIM_PUBLIC, TARGET_STATE_DIM_PUBLIC)


@functools.lru_cache(maxsize=None)
def observation_indices_of(team, num_cameras, num_targets, num_obstacles):
    widths, _ = _block_widths(team, num_cameras, num_targets, num_obstacles)
    return np.cumsum([0] + widths)


def camera_observation_indices_of(num_cameras, num_targets, num_obstacles):
    return observation_indices_of(Team.CAMERA, num_cameras, num_targets, num_obstacles)


def target_observation_indices_of(num_cameras, num_targets, num_obstacles):
    return observation_indices_of(Team.TARGET, num_cameras, num_targets, num_obstacles)


@functools.lru_cache(maxsize=None)
def observation_slices_of(team, num_cameras, num_targets, num_obstacles):
    idx = observation_indices_of(team, num_cameras, num_targets, num_obstacles)
    _, (opp_dim, mate_dim) = _block_widths(team, num_cameras, num_targets, num_obstacles)
    return {
        'preserved_data': slice(idx[0], idx[1]),
        'self_state': slice(idx[1], idx[2]),
        'opponent_states_with_mask': slice(idx[2], idx[3]),
        'opponent_mask': slice(idx[2] + opp_dim, idx[3], opp_dim + 1),
        'obstacle_states_with_mask': slice(idx[3], idx[4]),
        'obstacle_mask': slice(idx[3] + OBSTACLE_STATE_DIM, idx[4], OBSTACLE_STATE_DIM + 1),
        'teammate_states_with_mask': slice(idx[4], idx[5]),
        'teammate_mask': slice(idx[4] + mate_dim, idx[5], mate_dim + 1),
    }


def camera_observation_slices_of(num_cameras, num_targets, num_obstacles):
    return observation_slices_of(Team.CAMERA, num_cameras, num_targets, num_obstacles)


def target_observation_slices_of(num_cameras, num_targets, num_obstacles):
    return observation_slices_of(Team.TARGET, num_cameras, num_targets, num_obstacles)


@functools.lru_cache(maxsize=None)
def coordinate_mask_of(team, num_cameras, num_targets, num_obstacles):
    """True where an observation entry is an (x, y) coordinate of another entity or a warehouse."""
    widths, _ = _block_widths(team, num_cameras, num_targets, num_obstacles)
    mask = np.zeros(sum(widths), dtype=bool)
    mask[PRESERVED_DIM - 1 - 2 * NUM_WAREHOUSES:PRESERVED_DIM - 1] = True
    start = widths[0] + widths[1]
    strides = ((CAMERA_STATE_DIM_PUBLIC + 1, TARGET_STATE_DIM_PUBLIC + 1)[team is Team.CAMERA], OBSTACLE_STATE_DIM + 1,
               (TARGET_STATE_DIM_PUBLIC + 1, CAMERA_STATE_DIM_PUBLIC + 1)[team is Team.CAMERA])
    for width, stride in zip(widths[2:], strides):
        for base in range(start, start + width, stride):
            mask[base:base + 2] = True
        start += width
    return mask


def camera_coordinate_mask_of(num_cameras, num_targets, num_obstacles):
    return coordinate_mask_of(Team.CAMERA, num_cameras, num_targets, num_obstacles)


def target_coordinate_mask_of(num_cameras, num_targets, num_obstacles):
    return coordinate_mask_of(Team.TARGET, num_cameras, num_targets, num_obstacles)
