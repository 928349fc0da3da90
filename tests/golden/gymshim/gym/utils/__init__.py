from gym.utils import seeding
from gym.utils.ezpickle import EzPickle


def colorize(string, color=None, bold=False, highlight=False):
    return string
