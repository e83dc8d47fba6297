"""Importing this module registers the env ids, like `import tactile_gym.rl_envs` does in the reference
(tactile_gym/rl_envs/__init__.py:3-41; imported for side effect at sb3_helpers/train_agent.py:13).

`edge_follow_aotu-v0` (reference :8-11) points at a class that does not exist upstream (EdgeFollowAutoEnv); it is
registered here too, so that `make` fails with the same kind of import error instead of an unknown-id error.
"""
from ..registry import register

register(id="edge_follow-v0", entry_point="tactile_gym_amd.rl_envs.edge_follow:EdgeFollowEnv")
register(id="edge_follow_aotu-v0", entry_point="tactile_gym_amd.rl_envs.edge_follow:EdgeFollowAutoEnv")
register(id="surface_follow-v0", entry_point="tactile_gym_amd.rl_envs.surface_follow:SurfaceFollowAutoEnv")
register(id="surface_follow-v1", entry_point="tactile_gym_amd.rl_envs.surface_follow:SurfaceFollowGoalEnv")
register(id="surface_follow-v2", entry_point="tactile_gym_amd.rl_envs.surface_follow:SurfaceFollowVertEnv")
register(id="object_roll-v0", entry_point="tactile_gym_amd.rl_envs.object_roll:ObjectRollEnv")
register(id="object_push-v0", entry_point="tactile_gym_amd.rl_envs.object_push:ObjectPushEnv")
register(id="object_balance-v0", entry_point="tactile_gym_amd.rl_envs.object_balance:ObjectBalanceEnv")
