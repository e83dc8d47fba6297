"""Env-id registry (reference: tactile_gym/rl_envs/__init__.py:3-41).

`register`/`make` work without gym; when gym is importable the same ids are registered there as well so
`gym.make("edge_follow-v0", ...)` resolves to this package's classes.
"""
import importlib

_REGISTRY = {}


def register(id, entry_point):  # noqa: A002 - gym's keyword
    _REGISTRY[id] = entry_point
    try:  # pragma: no cover - gym absent in the build image
        from gym.envs.registration import register as gym_register
        gym_register(id=id, entry_point=entry_point)
    except Exception:  # noqa: BLE001
        pass


def _resolve(entry_point):
    mod, _, name = entry_point.partition(":")
    try:
        return getattr(importlib.import_module(mod), name)
    except (ImportError, AttributeError) as e:
        raise ImportError(f"env entry point {entry_point!r} cannot be resolved: {e}") from e


def spec(id):  # noqa: A002
    if id not in _REGISTRY:
        raise KeyError(f"No registered env with id: {id}")
    return _REGISTRY[id]


def make(id, **kwargs):  # noqa: A002
    """gym.make equivalent: one environment with the reference's constructor kwargs."""
    return _resolve(spec(id))(**kwargs)


def make_vec(id, num_envs, **kwargs):  # noqa: A002
    """Device-resident vectorised env (SB3 VecEnv API) — the MI355X-native replacement for
    make_vec_env(..., vec_env_cls=SubprocVecEnv) in sb3_helpers/rl_utils.py:17-30."""
    cls = _resolve(spec(id))
    if not hasattr(cls, "make_vec"):
        raise NotImplementedError(f"{id} has no vectorised implementation yet")
    return cls.make_vec(num_envs=num_envs, **kwargs)


def registered_ids():
    return sorted(_REGISTRY)
