from .ppo import PPO  # noqa: F401
from . import gail  # noqa: F401
