"""simgan_amd -- SimGAN's GAIL+PPO update path on MI355X (gfx950) behind the reference's own
class surface.  The compute lives in libsimgan_hip.so (hand-written HIP, C ABI in
include/simgan_hip.h); these modules are the thin host-side mirror of
third_party/a2c_ppo_acktr/{model,model_split,storage,algo/ppo,algo/gail}.py.
"""
from . import _lib  # noqa: F401
from .model import Policy  # noqa: F401
from .model_split import SplitPolicy  # noqa: F401
from .storage import RolloutStorage  # noqa: F401
from . import algo  # noqa: F401
from .utils import RunningMeanStd, update_linear_schedule  # noqa: F401

__all__ = ["Policy", "SplitPolicy", "RolloutStorage", "algo", "RunningMeanStd",
           "update_linear_schedule"]
