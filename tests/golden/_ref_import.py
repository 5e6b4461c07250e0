"""Test-infrastructure only: import the *unmodified* GOPS reference from /root/reference.

The reference needs `gym`, `gymnasium` and `tensorboard`, none of which are installed in this
image (and there is no network).  This module fabricates the tiny slice of those packages the
reference touches at import time for the ADP hot path and puts
/root/reference on sys.path.  It is used ONLY by `make_golden.py` to generate the committed
fixtures in this directory; nothing in the product (`gops_amd/`), `bench.py` or the `-m gpu`
tests imports it, and it cannot work on the GPU box (no /root/reference there).
"""
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package so `import a.b` works
    sys.modules[name] = m
    return m


def _build_gym(prefix):
    class Space:
        def __init__(self, shape=None, dtype=None, seed=None):
            self.shape = None if shape is None else tuple(shape)
            self.dtype = None if dtype is None else np.dtype(dtype)
            self._rng = np.random.RandomState(seed)

        def seed(self, seed=None):
            self._rng = np.random.RandomState(seed)
            return [seed]

        def sample(self):
            raise NotImplementedError

        def contains(self, x):
            return True

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            super().__init__(shape, dtype, seed)
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi).astype(self.dtype)

    class Discrete(Space):
        def __init__(self, n, seed=None):
            super().__init__((), np.int64, seed)
            self.n = n

        def sample(self):
            return int(self._rng.randint(self.n))

    class Dict(Space, dict):
        def __init__(self, spaces=None, **kw):
            Space.__init__(self)
            dict.__init__(self, spaces or {}, **kw)
            self.spaces = self

    class Tuple(Space):
        def __init__(self, spaces):
            super().__init__()
            self.spaces = tuple(spaces)

    class MultiBinary(Space):
        def __init__(self, n):
            super().__init__((n,), np.int8)

    class MultiDiscrete(Space):
        def __init__(self, nvec):
            super().__init__(np.asarray(nvec).shape, np.int64)
            self.nvec = np.asarray(nvec)

    class Env:
        metadata = {}
        reward_range = (-float("inf"), float("inf"))
        spec = None
        _np_random = None

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random = np.random.RandomState()
            return self._np_random

        @np_random.setter
        def np_random(self, v):
            self._np_random = v

        def seed(self, seed=None):
            self._np_random = np.random.RandomState(seed)
            return [seed]

        @property
        def unwrapped(self):
            return self

        def close(self):
            pass

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)
            return getattr(self.env, name)

        def step(self, action):
            return self.env.step(action)

        def reset(self, **kw):
            return self.env.reset(**kw)

        def seed(self, seed=None):
            return self.env.seed(seed)

        @property
        def unwrapped(self):
            return self.env.unwrapped

    class ActionWrapper(Wrapper):
        def step(self, action):
            return self.env.step(self.action(action))

    class ObservationWrapper(Wrapper):
        pass

    class RewardWrapper(Wrapper):
        pass

    class TimeLimit(Wrapper):
        def __init__(self, env, max_episode_steps=None):
            super().__init__(env)
            self._max_episode_steps = max_episode_steps

    def np_random(seed=None):
        return np.random.RandomState(seed), seed

    spaces = _mod(prefix + ".spaces", Space=Space, Box=Box, Discrete=Discrete, Dict=Dict,
                  Tuple=Tuple, MultiBinary=MultiBinary, MultiDiscrete=MultiDiscrete)
    core = _mod(prefix + ".core", Env=Env, Wrapper=Wrapper, ActionWrapper=ActionWrapper,
                ObservationWrapper=ObservationWrapper, RewardWrapper=RewardWrapper,
                ObsType=object, ActType=object)
    logger = _mod(prefix + ".logger", ERROR=40, setLevel=lambda *_: None, warn=lambda *_a, **_k: None)
    seeding = _mod(prefix + ".utils.seeding", np_random=np_random,
                   RandomNumberGenerator=np.random.RandomState)
    utils = _mod(prefix + ".utils", seeding=seeding)
    tl = _mod(prefix + ".wrappers.time_limit", TimeLimit=TimeLimit)
    wrappers = _mod(prefix + ".wrappers", time_limit=tl, TimeLimit=TimeLimit)
    error = _mod(prefix + ".error", **{n: type(n, (Exception,), {}) for n in (
        "AlreadyPendingCallError", "ClosedEnvironmentError", "CustomSpaceError", "NoAsyncCallError")})
    placeholder = lambda *a, **k: None  # noqa: E731
    vspaces = _mod(prefix + ".vector.utils.spaces", batch_space=placeholder)
    vutils = _mod(prefix + ".vector.utils", spaces=vspaces, **{n: placeholder for n in (
        "CloudpickleWrapper", "clear_mpi_env_vars", "concatenate", "create_empty_array",
        "create_shared_memory", "iterate", "read_from_shared_memory", "write_to_shared_memory",
        "batch_space")})
    vector = _mod(prefix + ".vector", utils=vutils)
    _mod(prefix, spaces=spaces, core=core, logger=logger, utils=utils, wrappers=wrappers,
         error=error, vector=vector, Env=Env, Wrapper=Wrapper, ActionWrapper=ActionWrapper,
         ObservationWrapper=ObservationWrapper, RewardWrapper=RewardWrapper, Space=Space)


def install():
    """Idempotently make `import gops...` resolve to the reference tree."""
    if "gops" in sys.modules and getattr(sys.modules["gops"], "__file__", "").startswith(REFERENCE_ROOT):
        return
    for prefix in ("gym", "gymnasium"):
        if prefix not in sys.modules:
            _build_gym(prefix)
    if "tensorboard" not in sys.modules:
        import logging
        app = _mod("tensorboard.backend.application", logger=logging.getLogger("tb-stub"))
        backend = _mod("tensorboard.backend", application=app)
        _mod("tensorboard", backend=backend)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
