"""gym compatibility: use `gym` (old API, what the reference was written against) or
`gymnasium` when importable, else a minimal in-tree stand-in with the few names the
environment needs (Env, spaces.Discrete, spaces.Box, seeding.np_random, register)."""
import hashlib
import os
import struct

import numpy as np

try:  # pragma: no cover - depends on the host image
    import gym as _gym
    from gym import spaces
    Env = _gym.Env
    try:
        from gym.envs.registration import register
    except Exception:  # noqa: BLE001
        register = None
    HAVE_GYM = "gym"
except Exception:  # noqa: BLE001
    try:  # pragma: no cover
        import gymnasium as _gym
        from gymnasium import spaces
        Env = _gym.Env
        from gymnasium.envs.registration import register
        HAVE_GYM = "gymnasium"
    except Exception:  # noqa: BLE001
        HAVE_GYM = None
        register = None

        class Env:
            metadata = {}
            reward_range = (-float("inf"), float("inf"))
            action_space = None
            observation_space = None

        class _Space:
            def __init__(self, shape, dtype):
                self.shape, self.dtype = tuple(shape), np.dtype(dtype)
                self._rng = np.random.RandomState()

            def seed(self, seed=None):
                self._rng = np.random.RandomState(seed)
                return [seed]

        class Discrete(_Space):
            def __init__(self, n):
                super().__init__((), np.int64)
                self.n = int(n)

            def sample(self):
                return int(self._rng.randint(self.n))

            def contains(self, x):
                if isinstance(x, (int, np.integer)):
                    v = int(x)
                elif isinstance(x, np.ndarray) and x.dtype.kind in "iu" and x.shape == ():
                    v = int(x)
                else:
                    return False
                return 0 <= v < self.n

            def __repr__(self):
                return "Discrete(%d)" % self.n

        class Box(_Space):
            def __init__(self, low, high, shape=None, dtype=np.float32):
                low, high = np.asarray(low), np.asarray(high)
                super().__init__(low.shape if shape is None else shape, dtype)
                self.low, self.high = low.astype(dtype), high.astype(dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

            def sample(self):
                return self._rng.normal(size=self.shape).astype(self.dtype)

            def __repr__(self):
                return "Box%s" % (self.shape,)

        class spaces:  # noqa: N801 - mirrors the module name
            Discrete = Discrete
            Box = Box


# ---- old-gym seeding (gym <= 0.21 gym.utils.seeding.np_random): SHA-512 hash_seed ->
# numpy RandomState.  The reference calls it at ssa_tasker_simple_2.py:189 and then uses
# .randint / .normal, which newer gym / gymnasium Generators do not offer.
def _bigint_from_bytes(b):
    sizeof_int = 4
    padding = sizeof_int - len(b) % sizeof_int
    b += b"\0" * padding
    int_count = int(len(b) / sizeof_int)
    unpacked = struct.unpack("{}I".format(int_count), b)
    accum = 0
    for i, val in enumerate(unpacked):
        accum += 2 ** (sizeof_int * 8 * i) * val
    return accum


def _int_list_from_bigint(bigint):
    if bigint < 0:
        raise ValueError("Seed must be non-negative, not {}".format(bigint))
    if bigint == 0:
        return [0]
    ints = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        ints.append(mod)
    return ints


def create_seed(a=None, max_bytes=8):
    if a is None:
        a = _bigint_from_bytes(os.urandom(max_bytes))
    elif isinstance(a, str):
        a = a.encode("utf8")
        a += hashlib.sha512(a).digest()
        a = _bigint_from_bytes(a[:max_bytes])
    elif isinstance(a, (int, np.integer)):
        a = int(a) % 2 ** (8 * max_bytes)
    else:
        raise ValueError("Invalid type for seed: {} ({})".format(type(a), a))
    return a


def hash_seed(seed=None, max_bytes=8):
    if seed is None:
        seed = create_seed(max_bytes=max_bytes)
    h = hashlib.sha512(str(seed).encode("utf8")).digest()
    return _bigint_from_bytes(h[:max_bytes])


def np_random(seed=None):
    """-> (numpy.random.RandomState, seed) exactly like gym<=0.21's seeding.np_random."""
    if seed is not None and not (isinstance(seed, (int, np.integer)) and 0 <= seed):
        raise ValueError("Seed must be a non-negative integer or omitted, not {}".format(seed))
    seed = create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng, seed
