"""mate_amd: MI355X-native batched MultiAgentTracking step engine -- a drop-in for the step path
of XuehaiPan/mate (`make / reset / step / seed` over HIP kernels).

`import mate_amd` never touches the GPU.  The HIP engine (mate_amd/lib/libmate_engine.so, C ABI in
include/mate_engine.h) is loaded when the first environment is created; there is no CPU fallback.
"""
from mate_amd import constants, utils
from mate_amd.config import ASSETS_DIR, DEFAULT_CONFIG_FILE, read_config, validate_config
from mate_amd.constants import *  # noqa: F401,F403
from mate_amd.utils import Message, Team, normalize_angle

__version__ = '0.1.0'

__all__ = ['make', 'make_environment', 'register', 'MultiAgentTracking', 'BatchedMultiAgentTracking', 'read_config',
           'validate_config', 'ASSETS_DIR', 'DEFAULT_CONFIG_FILE', 'Team', 'Message', 'normalize_angle']

_REGISTRY = {}


def register(id, entry_point, kwargs=None):  # noqa: A002  (gym's spelling)
    _REGISTRY[id] = (entry_point, dict(kwargs or {}))


def make_environment(config=None, wrappers=(), num_envs=None, **kwargs):
    """Create a (wrapped) environment: `num_envs=None` -> the NumPy single-environment API of the
    reference; an integer -> `BatchedMultiAgentTracking` with torch tensors."""
    from mate_amd.environment import BatchedMultiAgentTracking, MultiAgentTracking
    if num_envs is None:
        env = MultiAgentTracking(config, **kwargs)
    else:
        env = BatchedMultiAgentTracking(config, num_envs=num_envs, **kwargs)
    for wrapper in wrappers:
        assert callable(wrapper), f'You should provide a wrapper class or a callable. Got wrapper = {wrapper!r}.'
        env = wrapper(env)
    return env


def make(id, **kwargs):  # noqa: A002
    """`mate.make(id, config=..., wrappers=..., **overrides)` (mate/__init__.py:24-101)."""
    if id not in _REGISTRY:
        raise KeyError(f'No registered environment with id {id!r}; known ids: {sorted(_REGISTRY)}')
    entry_point, defaults = _REGISTRY[id]
    merged = dict(defaults)
    merged.update(kwargs)
    return entry_point(**merged)


register('MultiAgentTracking-v0', make_environment)
register('MATE-v0', make_environment)
for _name in ('4v2-9', '4v2-0', '4v4-9', '4v4-0', '4v8-9', '4v8-0', '8v8-9', '8v8-0', 'Navigation'):
    register(f'MATE-{_name}-v0', make_environment, {'config': f'MATE-{_name}.yaml'})
del _name


def __getattr__(name):  # lazy: importing the package must not need torch / the GPU
    if name in ('MultiAgentTracking', 'BatchedMultiAgentTracking', 'EnvMeta'):
        from mate_amd import environment
        return getattr(environment, name)
    raise AttributeError(name)
