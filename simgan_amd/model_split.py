"""SplitPolicy -- host mirror of a2c/model_split.py:39-95 (SplitPolicyBaseNew :157-198 +
StateDiagGaussianNew :201-238): contact / actuator / critic trunks, state-dependent logstd."""
import numpy as np

from . import _lib
from .model import _PolicyBase
from .utils import derive_seed, orthogonal


class SplitPolicy(_PolicyBase):
    KIND = _lib.POLICY_SPLIT

    def __init__(self, obs_shape, action_space, base_kwargs=None, ctx=None, seed=0):
        if base_kwargs is None:
            base_kwargs = {}
        hidden = base_kwargs.get("hidden_size", 64)
        num_feet = base_kwargs.get("num_feet", 1)
        num_outputs = action_space.shape[0]
        assert num_outputs == (4 + 3) * num_feet  # contact 4, act 3   (a2c/model_split.py:205)
        self._create(obs_shape[0], num_outputs, hidden, num_feet, ctx)
        self.seed = derive_seed(seed, 0x5EED, per_instance=True)
        self._init_params(np.random.default_rng(seed))

    def param_shapes(self):
        O, Hh, f = self.obs_dim, self.hidden_size, self.num_feet
        out = []
        for trunk in ("actor_contact", "actor_actuator", "critic_full"):
            out += [(f"base.{trunk}.0.weight", (Hh, O)), (f"base.{trunk}.0.bias", (Hh,)),
                    (f"base.{trunk}.2.weight", (Hh, Hh)), (f"base.{trunk}.2.bias", (Hh,))]
        out += [("base.critic_full.4.weight", (1, Hh)), ("base.critic_full.4.bias", (1,)),
                ("dist.contact_mean.weight", (4 * f, Hh)), ("dist.contact_mean.bias", (4 * f,)),
                ("dist.actuator_mean.weight", (3 * f, Hh)), ("dist.actuator_mean.bias", (3 * f,)),
                ("dist.contact_logstd.weight", (4 * f, Hh)), ("dist.contact_logstd.bias", (4 * f,)),
                ("dist.actuator_logstd.weight", (3 * f, Hh)), ("dist.actuator_logstd.bias", (3 * f,))]
        return out

    def _init_params(self, rng):
        """a2c/model_split.py:168-183 (trunks gain sqrt2, critic head gain 1), :209-222 (mean heads
        gain 0.02, logstd heads gain 1 with bias -0.5)."""
        sd = {}
        for name, shape in self.param_shapes():
            if name.endswith("bias"):
                sd[name] = np.full(shape, -0.5 if "logstd" in name else 0.0, np.float32)
            elif "_mean" in name:
                sd[name] = orthogonal(rng, *shape, gain=0.02)
            elif "logstd" in name or name == "base.critic_full.4.weight":
                sd[name] = orthogonal(rng, *shape, gain=1.0)
            else:
                sd[name] = orthogonal(rng, *shape, gain=np.sqrt(2))
        self.load_state_dict(sd)
