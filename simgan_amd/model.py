"""Policy -- host mirror of a2c/model.py:37-114 (MLPBase + DiagGaussian, Box actions,
non-recurrent).  Parameters live in HBM inside libsimgan_hip.so; this class keeps the
reference's constructor and method signatures and hands host tensors across the C ABI."""
import ctypes as C

import numpy as np

from . import _lib
from .utils import derive_seed, orthogonal, to_host_tensor


# Context for policies that build their device twin lazily (a bare torch.load of a reference checkpoint): None = the
# process default (`Context.default()`, which `ctx.make_default()` re-points).  A process that holds several contexts
# (loopback ranks on threads) assigns `simgan_amd.model.MATERIALISE_CTX = ctx` around its torch.load, or uses
# `checkpoint.load_policy(path, ctx=ctx)`, which builds the policy on `ctx` directly and never comes through here.
MATERIALISE_CTX = None


class _PolicyBase(object):
    KIND = None

    def _create(self, obs_dim, act_dim, hidden, num_feet, ctx, critic_hidden=None):
        self.ctx = ctx or _lib.Context.default()
        self.lib = self.ctx.lib
        self.obs_dim, self.act_dim = int(obs_dim), int(act_dim)
        self.hidden_size, self.num_feet = int(hidden), int(num_feet)
        # width of the critic trunk: the actors' unless reset_critic rebuilt it (a2c/model.py:80-87: always 64 units)
        self.critic_hidden = int(critic_hidden) if critic_hidden else self.hidden_size
        h = _lib.H()
        _lib.check(self.lib.sg_policy_create2(self.ctx.h, self.KIND, self.obs_dim, self.act_dim, self.hidden_size, self.num_feet,
                                              0 if self.critic_hidden == self.hidden_size else self.critic_hidden, C.byref(h)))
        self.h = h
        n = C.c_int64(0)
        _lib.check(self.lib.sg_policy_num_params(self.h, C.byref(n)))
        self.num_params = n.value
        self.seed = derive_seed(0, 0x5EED, per_instance=True)   # action-noise stream; re-derived from the constructor seed below
        self._act_calls = 0

    def _register_handle_user(self, obj):
        """Objects that keep this policy's device handle (PPO, PolicyEnsemble) announce themselves: reset_critic refuses to
        replace the handle while one of them is alive."""
        import weakref
        self.__dict__.setdefault("_handle_users", []).append(weakref.ref(obj))

    def __del__(self):
        try:
            h = self.__dict__.get("h")
            if h:
                self.__dict__["lib"].sg_policy_destroy(h)
                self.__dict__["h"] = None
        except Exception:
            pass

    def __getattr__(self, name):
        # Only reached for attributes that are not set: an object torch.load is still assembling (see __setstate__)
        # builds its device twin on first use.
        if not name.startswith("__") and self.__dict__.get("_pending") is not None:
            self._materialise()
            return getattr(self, name)
        raise AttributeError(name)

    def _materialise(self):
        from .checkpoint import policy_from_module_state
        st = self.__dict__.pop("_pending")
        try:
            dims, sd = policy_from_module_state(type(self).__name__, st)
            # the context an unpickled policy lands on: `simgan_amd.model.MATERIALISE_CTX` when the caller set one,
            # the process default otherwise (see the module-level comment)
            self._create(dims["obs_dim"], dims["act_dim"], dims["hidden"], dims["num_feet"], MATERIALISE_CTX,
                         critic_hidden=dims.get("critic_hidden"))
            self.seed = derive_seed(0, 0x5EED, per_instance=True)
            self.load_state_dict(sd)
        except BaseException:
            # keep the pickled state: the next attribute access retries (and raises the real cause again) instead of a
            # bare AttributeError with the state lost
            for k in ("h", "ctx", "lib"):
                self.__dict__.pop(k, None)
            self.__dict__["_pending"] = st
            raise

    # ---- nn.Module-ish surface the reference mains touch
    @property
    def is_recurrent(self):
        return False

    @property
    def recurrent_hidden_state_size(self):
        """Size of rnn_hx."""
        return 1

    def to(self, device):
        return self

    def train(self, mode=True):
        return self

    def eval(self):
        return self

    def parameters(self):
        """One flat host tensor (a copy): enough for `next(p.parameters())`-style probes."""
        yield to_host_tensor(self.get_flat_params())

    def forward(self, inputs, rnn_hxs, masks):
        raise NotImplementedError

    # ---- flat parameter access (torch state_dict order, see oracle/sg_oracle.c header)
    def get_flat_params(self):
        out = np.empty(self.num_params, np.float32)
        _lib.check(self.lib.sg_policy_get_params(self.h, _lib.fptr(out), out.size))
        return out

    def set_flat_params(self, flat):
        flat = _lib.as_f32(flat).reshape(-1)
        _lib.check(self.lib.sg_policy_set_params(self.h, _lib.fptr(flat), flat.size))

    def state_dict(self):
        flat, out, off = self.get_flat_params(), {}, 0
        for name, shape in self.param_shapes():
            n = int(np.prod(shape))
            out[name] = to_host_tensor(flat[off:off + n].reshape(shape).copy())
            off += n
        return out

    def load_state_dict(self, sd):
        parts = []
        for name, shape in self.param_shapes():
            a = _lib.as_f32(sd[name])
            assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
            parts.append(a.reshape(-1))
        self.set_flat_params(np.concatenate(parts))

    def __getstate__(self):
        return {"obs_dim": self.obs_dim, "act_dim": self.act_dim, "hidden": self.hidden_size,
                "num_feet": self.num_feet, "critic_hidden": self.critic_hidden, "flat": self.get_flat_params()}

    def __reduce_ex__(self, protocol):
        """A policy built through the reference's import path (`third_party.a2c_ppo_acktr.model[_split]`, i.e. by the
        unchanged mains) pickles in the REFERENCE's object layout: what their `torch.save([actor_critic, ob_rms], path)`
        writes is then a file the reference itself loads -- on the CPU, without this package (the stage-2 environment
        workers do, my_pybullet_envs/utils.py:24-57).  Loading it here goes through `__setstate__`'s "_modules" branch like
        any reference checkpoint.  The package's own classes keep their compact native pickle."""
        if type(self).__module__.startswith("third_party."):
            import copyreg
            from .checkpoint import reference_module_state
            return (copyreg.__newobj__, (type(self),), reference_module_state("mlp" if self.KIND == _lib.POLICY_MLP else "split", self.state_dict()))
        return object.__reduce_ex__(self, protocol)

    def __setstate__(self, st):
        if "_modules" in st:   # a reference whole-module pickle (a2c/main.py:81-83) resolved to this class by the alias modules.
            # torch's legacy container fills the tensors' storages only AFTER the whole object graph is unpickled, so the
            # weights are read when the policy is first used, not here.
            self.__dict__["_pending"] = st
            return
        self._create(st["obs_dim"], st["act_dim"], st["hidden"], st["num_feet"], None, critic_hidden=st.get("critic_hidden"))
        self.seed = derive_seed(0, 0x5EED, per_instance=True)
        self.set_flat_params(st["flat"])

    # ---- the three calls on the hot path
    def act(self, inputs, rnn_hxs, masks, deterministic=False, noise=None):
        """a2c/model.py:89-101 -> (value [n,1], action [n,A], action_log_probs [n,1], rnn_hxs).
        `noise` ([n,A] standard normal) injects the sampling draw; default = library RNG."""
        obs = _lib.as_f32(inputs).reshape(-1, self.obs_dim)
        n = obs.shape[0]
        value = np.empty((n, 1), np.float32)
        action = np.empty((n, self.act_dim), np.float32)
        logp = np.empty((n, 1), np.float32)
        nz = None if noise is None else _lib.as_f32(noise).reshape(n, self.act_dim)
        self._act_calls += 1
        _lib.check(self.lib.sg_policy_act(self.h, _lib.fptr(obs), n,
                                          None if nz is None else _lib.fptr(nz), (self.seed + self._act_calls) & (2 ** 64 - 1),
                                          1 if deterministic else 0, _lib.fptr(value),
                                          _lib.fptr(action), _lib.fptr(logp)))
        return to_host_tensor(value), to_host_tensor(action), to_host_tensor(logp), rnn_hxs

    def get_value(self, inputs, rnn_hxs, masks):
        """a2c/model.py:103-105"""
        obs = _lib.as_f32(inputs).reshape(-1, self.obs_dim)
        value = np.empty((obs.shape[0], 1), np.float32)
        _lib.check(self.lib.sg_policy_get_value(self.h, _lib.fptr(obs), obs.shape[0], _lib.fptr(value)))
        return to_host_tensor(value)

    def evaluate_actions(self, inputs, rnn_hxs, masks, action):
        """a2c/model.py:107-114 -> (value, action_log_probs, dist_entropy, rnn_hxs)"""
        obs = _lib.as_f32(inputs).reshape(-1, self.obs_dim)
        n = obs.shape[0]
        act = _lib.as_f32(action).reshape(n, self.act_dim)
        value = np.empty((n, 1), np.float32)
        logp = np.empty((n, 1), np.float32)
        ent = C.c_float(0)
        _lib.check(self.lib.sg_policy_evaluate(self.h, _lib.fptr(obs), _lib.fptr(act), n,
                                               _lib.fptr(value), _lib.fptr(logp), C.byref(ent)))
        ent_t = to_host_tensor(np.array(ent.value, np.float32))
        return to_host_tensor(value), to_host_tensor(logp), ent_t, rnn_hxs


class Policy(_PolicyBase):
    KIND = _lib.POLICY_MLP

    def __init__(self, obs_shape, action_space, base=None, base_kwargs=None, ctx=None, seed=0, critic_hidden=None):
        if base_kwargs is None:
            base_kwargs = {}
        if base is not None or len(obs_shape) != 1:
            raise NotImplementedError("only the MLP base on 1-D observations is built (SURVEY.md section 2, row 3)")
        if base_kwargs.get("recurrent", False):
            raise NotImplementedError("recurrent policies are not used by any shipped SimGAN config")
        if action_space.__class__.__name__ != "Box":
            raise NotImplementedError("only Box action spaces (a2c/model.py:55-57)")
        hidden = base_kwargs.get("hidden_size", 64)
        self._create(obs_shape[0], action_space.shape[0], hidden, 1, ctx, critic_hidden=critic_hidden)
        self.seed = derive_seed(seed, 0x5EED, per_instance=True)
        self._init_params(np.random.default_rng(seed))

    def param_shapes(self):
        O, A, Hh, Hc = self.obs_dim, self.act_dim, self.hidden_size, self.critic_hidden
        return [("base.actor.0.weight", (Hh, O)), ("base.actor.0.bias", (Hh,)),
                ("base.actor.2.weight", (Hh, Hh)), ("base.actor.2.bias", (Hh,)),
                ("base.critic.0.weight", (Hc, O)), ("base.critic.0.bias", (Hc,)),
                ("base.critic.2.weight", (Hc, Hc)), ("base.critic.2.bias", (Hc,)),
                ("base.critic_linear.weight", (1, Hc)), ("base.critic_linear.bias", (1,)),
                ("dist.fc_mean.weight", (A, Hh)), ("dist.fc_mean.bias", (A,)),
                ("dist.logstd._bias", (A, 1))]

    def _init_params(self, rng):
        """a2c/model.py:240-251 (orthogonal, gain sqrt2, zero bias), a2c/distributions.py:95-104
        (fc_mean gain 1 then /50, logstd -0.5)."""
        sd = {}
        for name, shape in self.param_shapes():
            if name.endswith("logstd._bias"):
                sd[name] = np.full(shape, -0.5, np.float32)
            elif name.endswith("bias"):
                sd[name] = np.zeros(shape, np.float32)
            elif name.startswith("dist.fc_mean"):
                sd[name] = orthogonal(rng, *shape, gain=1.0) / 50.0
            else:
                sd[name] = orthogonal(rng, *shape, gain=np.sqrt(2))
        self.load_state_dict(sd)

    def reset_variance(self, action_space, log_std):
        """a2c/model.py:76-78"""
        sd = self.state_dict()
        sd["dist.logstd._bias"] = np.full((action_space.shape[0], 1), log_std, np.float32)
        self.load_state_dict(sd)

    def reset_critic(self, obs_shape, seed=1):
        """a2c/model.py:80-87: a fresh critic -- Linear(obs, 64)-Tanh-Linear(64, 64)-Tanh + Linear(64, 1), orthogonal with gain
        sqrt 2, zero biases -- whatever the actor's width ("64" is hard-coded in the reference; a2c/main.py:85 calls this on
        every warm start).  When the actor is not 64 wide the policy is rebuilt on the device with a critic trunk of its own
        width (sg_policy_create2); actor, mean head and log-std keep their values.  Call it BEFORE constructing PPO on this
        policy, as the reference's main does (:85 then :149): an agent built earlier holds the old device handle."""
        if int(obs_shape[0]) != self.obs_dim:
            raise ValueError(f"reset_critic: obs_shape {tuple(obs_shape)} != the policy's observation size {self.obs_dim}")
        rng = np.random.default_rng(seed)
        sd = self.state_dict()
        if self.critic_hidden != 64:
            users = [u for u in self.__dict__.get("_handle_users", ()) if u() is not None]
            if users:   # they hold the device handle that is about to be replaced: refuse instead of freeing it under them
                raise RuntimeError(f"reset_critic: {len(users)} object(s) built on this policy ({', '.join(type(u()).__name__ for u in users)}) "
                                   "hold its device handle, which a critic of another width replaces; call reset_critic first "
                                   "(a2c/main.py:85 precedes :149) or rebuild them afterwards")
            old, seed_, calls_ = self.h, self.seed, self._act_calls
            self._create(self.obs_dim, self.act_dim, self.hidden_size, self.num_feet, self.ctx, critic_hidden=64)
            self.seed, self._act_calls = seed_, calls_     # the action-noise stream goes on where it was
            self.lib.sg_policy_destroy(old)
        for name, shape in self.param_shapes():
            if name.startswith("base.critic"):
                sd[name] = (np.zeros(shape, np.float32) if name.endswith("bias")
                            else orthogonal(rng, *shape, gain=np.sqrt(2)))
        self.load_state_dict(sd)
