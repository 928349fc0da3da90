import numpy as np


def np_random(seed=None):
    """Legacy-style: returns (RandomState, seed).  The seed->stream mapping of a
    real gym install is NOT reproduced (unpinned)."""
    if seed is not None and not (isinstance(seed, (int, np.integer)) and seed >= 0):
        raise ValueError(f'Seed must be a non-negative integer or omitted, not {seed}')
    if seed is None:
        seed = int(np.random.SeedSequence().entropy % (2**31))
    seed = int(seed)
    rng = np.random.RandomState(seed % (2**32))
    return rng, seed
