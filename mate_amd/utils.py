"""Small host-side helpers with the reference's public names (mate/utils.py): degree trigonometry,
`Team`, `Message`.  Only what the environment boundary needs."""
import enum
from dataclasses import dataclass
from typing import Any, Optional

import numpy as np

__all__ = ['RAD2DEG', 'DEG2RAD', 'sin_deg', 'cos_deg', 'arctan2_deg', 'polar2cartesian', 'normalize_angle', 'Team', 'Message']

RAD2DEG = 180.0 / np.pi
DEG2RAD = np.pi / 180.0


def sin_deg(x):
    return np.sin(np.deg2rad(x))


def cos_deg(x):
    return np.cos(np.deg2rad(x))


def arctan2_deg(y, x):
    return np.rad2deg(np.arctan2(y, x))


def polar2cartesian(rho, phi):
    """(rho, phi in degrees) -> (x, y)."""
    rad = np.deg2rad(phi)
    return rho * np.array([np.cos(rad), np.sin(rad)])


def normalize_angle(angle):
    """Map an angle in degrees onto [-180, 180)."""
    return (angle + 180.0) % 360.0 - 180.0


class Team(enum.Enum):
    CAMERA = 0
    TARGET = 1


@dataclass
class Message:
    """Intra-team message (sender/recipient are agent indices; recipient None = broadcast)."""
    sender: int
    recipient: Optional[int]
    content: Any
    team: Team
    broadcasting: bool = False

    def __contains__(self, name):
        return name in self.content

    def __getitem__(self, name):
        return self.content[name]

    def __setitem__(self, name, value):
        self.content[name] = value
