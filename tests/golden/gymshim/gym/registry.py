import importlib


class EnvSpec:
    def __init__(self, id, entry_point, kwargs=None):
        self.id = id
        self.entry_point = entry_point
        self.kwargs = dict(kwargs or {})

    def make(self, **kwargs):
        merged = dict(self.kwargs)
        merged.update(kwargs)
        entry = self.entry_point
        if isinstance(entry, str):
            mod, attr = entry.split(':')
            entry = getattr(importlib.import_module(mod), attr)
        env = entry(**merged)
        try:
            env.unwrapped.spec = self
        except AttributeError:
            pass
        return env


registry = {}


def register(id, entry_point=None, kwargs=None, **_):
    registry[id] = EnvSpec(id, entry_point, kwargs)


def make(id, **kwargs):
    return registry[id].make(**kwargs)
