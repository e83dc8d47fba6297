"""tactile_gym_amd — MI355X-native vectorised tactile-env step (drop-in for the hot path of ac-93/tactile_gym).

    import tactile_gym_amd as tg
    env = tg.make("edge_follow-v0", max_steps=200, image_size=[128, 128], env_modes={...})           # gym.Env surface
    venv = tg.make_vec("edge_follow-v0", num_envs=1024, max_steps=200, image_size=[128, 128], env_modes={...})  # VecEnv surface

The per-step arithmetic (rigid-body tick x24, tactile depth raster) runs in hand-written HIP kernels behind the C ABI
declared in include/tactile_gym_hip.h; importing the package never touches the GPU, constructing an env does and fails
loudly when the HIP library or a GPU is missing.
"""
from . import rl_envs  # noqa: F401  (registers the env ids)
from .registry import make, make_vec, register, registered_ids  # noqa: F401
from .vec_env import HipVecEnv  # noqa: F401  (vec_env_cls for stable_baselines3's make_vec_env)

__version__ = "0.1.0"
