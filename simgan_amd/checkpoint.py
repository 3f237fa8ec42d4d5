"""Checkpoint interop with the reference's whole-module pickles.

The reference saves `torch.save([actor_critic, ob_rms], "<env>.pt")` and `torch.save(discr, "<env>_D.pt")`
(a2c/main.py:260-269, a2c/main_gail_dyn_ppo.py:308-320) and reloads them with a bare `torch.load`
(my_pybullet_envs/utils.py:24-82, a2c/main.py:78-88): the files pickle the module OBJECTS, by reference to classes
under `third_party.a2c_ppo_acktr.*`.  This module

  * reads such files without the reference on the import path and without executing anything from the file: the
    unpickler resolves ONLY an explicit allowlist of (module, name) pairs -- the tensor / storage / parameter
    rebuilders, the three torch.nn layer classes and the Adam optimizer the reference's modules contain, the numpy
    array rebuilders and the plain containers; classes under `third_party.` (the reference's own) become inert
    stand-ins that only receive their `__dict__`; every other global (builtins.eval, os.system, torch.hub, ...) raises
    `pickle.UnpicklingError`.  Parameters are collected by walking `_parameters` / `_modules` (== `state_dict()` order);
  * writes files the reference's `torch.load` accepts: real torch.nn layers inside objects whose classes carry the
    reference's module paths and attribute names (`base.actor`, `dist.fc_mean`, `dist.logstd._bias`, ...), in the
    legacy (non-zip) container the shipped `trained_models_*/ppo/*.pt` use.

Pure host code (torch CPU + numpy); `load_policy` / `save_policy` are the thin device-facing wrappers.
"""
import collections
import pickle
import sys
import types

import numpy as np

REF_PKG = "third_party.a2c_ppo_acktr"
# The globals a reference checkpoint may name.  Nothing else is ever resolved (a pickle REDUCE on an allowed rebuilder
# only constructs tensors / arrays / containers; a prefix rule such as "anything under builtins or torch" would let a
# crafted file call builtins.eval or torch.hub.load).
_ALLOWED = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("builtins", "dict"), ("builtins", "list"),
    ("builtins", "set"), ("builtins", "tuple"), ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"),
    ("builtins", "complex"), ("builtins", "slice"), ("__builtin__", "dict"), ("__builtin__", "list"), ("__builtin__", "set"),
    ("_codecs", "encode"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch.nn.parameter", "Parameter"), ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"),
    ("torch.serialization", "_get_layout"), ("torch.storage", "UntypedStorage"),
    ("torch.storage", "TypedStorage"),
    ("torch.nn.modules.linear", "Linear"), ("torch.nn.modules.activation", "Tanh"),
    ("torch.nn.modules.container", "Sequential"), ("torch.optim.adam", "Adam"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy", "float64"), ("numpy", "float32"), ("numpy", "int64"),
    ("types", "SimpleNamespace"),
}
_STUB_ROOTS = ("third_party", "a2c_ppo_acktr")   # the reference's own classes (also when imported via sys.path.append("third_party"))
_TORCH_STORAGES = ("FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage",
                   "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage")
_TORCH_DTYPES = ("float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool")


class _RefStub(object):
    """Stand-in for a reference class: keeps whatever state the pickle carries, runs no reference code."""

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state


_STUBS = {}


def _stub_class(module, name):
    key = (module, name)
    if key not in _STUBS:
        _STUBS[key] = type(name, (_RefStub,), {"__module__": module, "_ref_path": f"{module}.{name}"})
    return _STUBS[key]


class _StubUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED or (module == "torch" and (name in _TORCH_STORAGES or name in _TORCH_DTYPES)):
            return super().find_class(module, name)
        if module.split(".")[0] in _STUB_ROOTS:
            return _stub_class(module, name)
        raise pickle.UnpicklingError(f"checkpoint names the global {module}.{name}, which a SimGAN checkpoint has no use "
                                     "for: refusing to resolve it")


def _stub_pickle_module():
    m = types.ModuleType("simgan_amd._stub_pickle")
    m.__dict__.update({k: getattr(pickle, k) for k in dir(pickle) if not k.startswith("__")})
    m.Unpickler = _StubUnpickler

    def load(f, **kw):
        return _StubUnpickler(f, **kw).load()

    m.load = load
    return m


def _torch_load(path):
    import warnings

    import torch
    with warnings.catch_warnings():   # legacy containers carry the classes' source for a check that cannot apply to stand-ins
        warnings.simplefilter("ignore")
        return torch.load(path, map_location="cpu", pickle_module=_stub_pickle_module(), weights_only=False)


def _walk(obj, prefix, out):
    for k, p in (getattr(obj, "_parameters", None) or {}).items():
        if p is not None:
            out[prefix + k] = p.detach().cpu().numpy().astype(np.float32)
    for k, m in (getattr(obj, "_modules", None) or {}).items():
        if m is not None:
            _walk(m, prefix + k + ".", out)
    return out


def _rms_state(rms):
    if rms is None:
        return None
    return {"mean": np.asarray(rms.mean, np.float64), "var": np.asarray(rms.var, np.float64), "count": float(rms.count)}


def _policy_dims(class_name, sd):
    if class_name == "Policy":
        H, O = sd["base.actor.0.weight"].shape
        A = sd["dist.fc_mean.weight"].shape[0]
        Hc = sd["base.critic.0.weight"].shape[0]     # 64 beside any actor width once reset_critic has run (a2c/model.py:80-87)
        assert sd["base.critic.0.weight"].shape == (Hc, O) and sd["base.critic.2.weight"].shape == (Hc, Hc) and \
            sd["base.critic_linear.weight"].shape == (1, Hc), "inconsistent critic shapes in the checkpoint"
        return dict(kind="mlp", obs_dim=int(O), act_dim=int(A), hidden=int(H), num_feet=1, critic_hidden=int(Hc))
    if class_name == "SplitPolicy":
        H, O = sd["base.actor_contact.0.weight"].shape
        f = sd["dist.contact_mean.weight"].shape[0] // 4
        return dict(kind="split", obs_dim=int(O), act_dim=int(7 * f), hidden=int(H), num_feet=int(f))
    raise ValueError(f"unsupported policy class in checkpoint: {class_name}")


def policy_from_module_state(class_name, state):
    """The `__dict__` torch pickles for a reference `Policy` / `SplitPolicy` module (what `__setstate__` receives when
    a bare `torch.load` of a reference checkpoint resolves the class to this package's shim through the
    `third_party.a2c_ppo_acktr` alias modules) -> (dims dict, state_dict)."""
    holder = _RefStub()
    holder.__setstate__(state)
    sd = _walk(holder, "", collections.OrderedDict())
    return _policy_dims(class_name, sd), sd


def read_reference_checkpoint(path):
    """`[actor_critic, ob_rms]` file -> dict(kind, obs_dim, act_dim, hidden, num_feet, class_name, state_dict, ob_rms).
    state_dict is an OrderedDict name -> float32 array in the reference's `state_dict()` order."""
    obj = _torch_load(path)
    actor_critic, ob_rms = (obj[0], obj[1]) if isinstance(obj, (list, tuple)) else (obj, None)
    sd = _walk(actor_critic, "", collections.OrderedDict())
    name = type(actor_critic).__name__
    out = _policy_dims(name, sd)
    out.update(class_name=name, state_dict=sd, ob_rms=_rms_state(ob_rms))
    return out


def read_reference_discriminator(path):
    """`<env>_D.pt` -> dict(input_dim, hidden_dim, state_dict, adam (m, v, step flat in state_dict order, or None),
    returns, ret_rms)."""
    d = _torch_load(path)
    sd = _walk(d, "", collections.OrderedDict())
    Hd, F = sd["trunk.0.weight"].shape
    adam = None
    opt = getattr(d, "optimizer", None)
    if opt is not None and getattr(opt, "state", None):
        params = [p for k, p in _named_params(d)]
        ms, vs, steps = [], [], []
        for p in params:
            st = opt.state.get(p)
            if st is None:
                ms, vs = None, None
                break
            ms.append(st["exp_avg"].detach().cpu().numpy().reshape(-1))
            vs.append(st["exp_avg_sq"].detach().cpu().numpy().reshape(-1))
            steps.append(int(st["step"]))
        if ms is not None:
            adam = (np.concatenate(ms).astype(np.float32), np.concatenate(vs).astype(np.float32), steps[0])
    rets = getattr(d, "returns", None)
    return dict(input_dim=int(F), hidden_dim=int(Hd), state_dict=sd, adam=adam,
                returns=None if rets is None else rets.detach().cpu().numpy(), ret_rms=_rms_state(getattr(d, "ret_rms", None)))


def _named_params(obj, prefix=""):
    for k, p in (getattr(obj, "_parameters", None) or {}).items():
        if p is not None:
            yield prefix + k, p
    for k, m in (getattr(obj, "_modules", None) or {}).items():
        if m is not None:
            yield from _named_params(m, prefix + k + ".")


# ------------------------------------------------------------------------------------------ writing
class _RefModules(object):
    """While active, `third_party.a2c_ppo_acktr.{model,model_split,distributions,utils}` resolve to torch.nn.Module
    classes with the reference's names, so pickle can store the objects by reference.  If the real reference is already
    imported its own classes are used; otherwise (nothing imported, or this repository's alias package whose `Policy`
    is the device-backed shim) empty torch.nn.Module subclasses are installed for the duration of the dump and the
    previous attributes restored afterwards."""

    NAMES = {"model": ["Policy", "MLPBase"], "model_split": ["SplitPolicy", "SplitPolicyBaseNew", "StateDiagGaussianNew"],
             "distributions": ["DiagGaussian"], "utils": ["AddBias"]}

    def __enter__(self):
        import torch
        self.added, self.cls, self.swapped = [], {}, []
        for pkg in ("third_party", REF_PKG):
            if pkg not in sys.modules:
                sys.modules[pkg] = types.ModuleType(pkg)
                self.added.append(pkg)
        for sub, names in self.NAMES.items():
            full = f"{REF_PKG}.{sub}"
            mod = sys.modules.get(full)
            if mod is None:
                mod = types.ModuleType(full)
                sys.modules[full] = mod
                self.added.append(full)
            for n in names:
                cur = getattr(mod, n, None)
                if not (isinstance(cur, type) and issubclass(cur, torch.nn.Module)):
                    self.swapped.append((mod, n, cur))
                    setattr(mod, n, type(n, (torch.nn.Module,), {"__module__": full}))
                self.cls[n] = getattr(mod, n)
        return self

    def __exit__(self, *exc):
        for mod, n, cur in self.swapped:
            if cur is None:
                delattr(mod, n)
            else:
                setattr(mod, n, cur)
        for name in self.added:
            sys.modules.pop(name, None)
        return False

    def new(self, name):
        import torch
        obj = self.cls[name].__new__(self.cls[name])
        torch.nn.Module.__init__(obj)
        return obj


def _linear(w, b):
    import torch
    lin = torch.nn.Linear(w.shape[1], w.shape[0])
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(np.ascontiguousarray(w, np.float32)))
        lin.bias.copy_(torch.from_numpy(np.ascontiguousarray(b, np.float32)))
    return lin


def _trunk(sd, prefix, head=False):
    import torch
    layers = [_linear(sd[f"{prefix}.0.weight"], sd[f"{prefix}.0.bias"]), torch.nn.Tanh(),
              _linear(sd[f"{prefix}.2.weight"], sd[f"{prefix}.2.bias"]), torch.nn.Tanh()]
    if head:
        layers.append(_linear(sd[f"{prefix}.4.weight"], sd[f"{prefix}.4.bias"]))
    return torch.nn.Sequential(*layers)


def save_reference_checkpoint(path, kind, state_dict, ob_rms=None):
    """Write `[actor_critic, ob_rms]` as the reference's `torch.load` expects it.  `state_dict`: name -> array with the
    reference's parameter names (what `Policy.state_dict()` / `SplitPolicy.state_dict()` of this package return)."""
    import torch
    sd = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, np.float32) for k, v in state_dict.items()}
    with _RefModules() as ref:
        if kind == "mlp":
            base = ref.new("MLPBase")
            base._hidden_size = int(sd["base.actor.0.weight"].shape[0])
            base._recurrent = False
            base.actor = _trunk(sd, "base.actor")
            base.critic = _trunk(sd, "base.critic")
            base.critic_linear = _linear(sd["base.critic_linear.weight"], sd["base.critic_linear.bias"])
            dist = ref.new("DiagGaussian")
            dist.fc_mean = _linear(sd["dist.fc_mean.weight"], sd["dist.fc_mean.bias"])
            bias = ref.new("AddBias")
            bias._bias = torch.nn.Parameter(torch.from_numpy(sd["dist.logstd._bias"].reshape(-1, 1).copy()))
            dist.logstd = bias
            pol = ref.new("Policy")
        elif kind == "split":
            base = ref.new("SplitPolicyBaseNew")
            base.actor_contact = _trunk(sd, "base.actor_contact")
            base.actor_actuator = _trunk(sd, "base.actor_actuator")
            base.critic_full = _trunk(sd, "base.critic_full", head=True)
            dist = ref.new("StateDiagGaussianNew")
            dist.hidden_size = int(sd["base.actor_contact.0.weight"].shape[0])     # a2c/model_split.py:204
            for head in ("contact_mean", "actuator_mean", "contact_logstd", "actuator_logstd"):
                setattr(dist, head, _linear(sd[f"dist.{head}.weight"], sd[f"dist.{head}.bias"]))
            pol = ref.new("SplitPolicy")
        else:
            raise ValueError(kind)
        pol.base, pol.dist = base, dist
        pol.train()
        rms = None
        if ob_rms is not None:   # plain namespace with the three fields the reference reads (envs.py VecNormalize.ob_rms)
            rms = types.SimpleNamespace(mean=np.asarray(ob_rms["mean"]), var=np.asarray(ob_rms["var"]), count=ob_rms["count"])
        import warnings
        with warnings.catch_warnings():   # "couldn't retrieve source code": the stand-in classes have none to embed
            warnings.simplefilter("ignore")
            torch.save([pol, rms], path, _use_new_zipfile_serialization=False)


# ------------------------------------------------------------ torch.save of a shim policy, as the unchanged mains call it
def _module_state(**extra):
    """A fresh torch.nn.Module's `__dict__` (whatever this torch version keeps in it) with private containers."""
    import torch
    d = {}
    for k, v in torch.nn.Module().__dict__.items():
        d[k] = type(v)() if isinstance(v, (dict, set, list)) else v
    d.update(extra)
    return d


def _holder(module, name, modules=(), params=(), **attrs):
    """An instance of the alias package's holder class `third_party.a2c_ppo_acktr.<module>.<name>` carrying an
    nn.Module-shaped `__dict__`: pickled by reference to that path, it is what the REFERENCE's class of the same name
    receives in `__setstate__` on the loading side."""
    import importlib
    cls = getattr(importlib.import_module(f"{REF_PKG}.{module}"), name)
    h = cls.__new__(cls)
    st = _module_state(**attrs)
    for k, v in modules:
        st["_modules"][k] = v
    for k, v in params:
        st["_parameters"][k] = v
    h.__dict__.update(st)
    return h


def reference_module_state(kind, sd):
    """The `__dict__` the reference's `Policy` / `SplitPolicy` module would pickle for the parameters `sd` (name -> array,
    the reference's parameter names): `_modules = {base, dist}` with real torch.nn layers inside holder objects whose
    classes live at the reference's import paths.  `simgan_amd.model._PolicyBase.__reduce_ex__` returns it when the policy
    was built through `third_party.a2c_ppo_acktr.model[_split]` -- i.e. by the unchanged mains -- so the file their
    `torch.save([actor_critic, ob_rms], path)` writes (a2c/main_gail_dyn_ppo.py:307-316, a2c/main.py:260-269) is one the
    REFERENCE's `torch.load` turns back into its own modules: the stage-2 environment workers load five of them on the
    CPU (my_pybullet_envs/utils.py:24-57, hopper_env_combined_policy.py:113-140), with no GPU and without this package."""
    import torch
    sd = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, np.float32) for k, v in sd.items()}
    if kind == "mlp":
        base = _holder("model", "MLPBase", modules=[("actor", _trunk(sd, "base.actor")), ("critic", _trunk(sd, "base.critic")),
                                                    ("critic_linear", _linear(sd["base.critic_linear.weight"], sd["base.critic_linear.bias"]))],
                       _hidden_size=int(sd["base.actor.0.weight"].shape[0]), _recurrent=False)
        bias = _holder("utils", "AddBias", params=[("_bias", torch.nn.Parameter(torch.from_numpy(sd["dist.logstd._bias"].reshape(-1, 1).copy())))])
        dist = _holder("distributions", "DiagGaussian", modules=[("fc_mean", _linear(sd["dist.fc_mean.weight"], sd["dist.fc_mean.bias"])),
                                                                 ("logstd", bias)])
    elif kind == "split":
        base = _holder("model_split", "SplitPolicyBaseNew", modules=[("actor_contact", _trunk(sd, "base.actor_contact")),
                                                                     ("actor_actuator", _trunk(sd, "base.actor_actuator")),
                                                                     ("critic_full", _trunk(sd, "base.critic_full", head=True))])
        dist = _holder("model_split", "StateDiagGaussianNew",
                       modules=[(h, _linear(sd[f"dist.{h}.weight"], sd[f"dist.{h}.bias"])) for h in
                                ("contact_mean", "actuator_mean", "contact_logstd", "actuator_logstd")],
                       hidden_size=int(sd["base.actor_contact.0.weight"].shape[0]))      # a2c/model_split.py:204
    else:
        raise ValueError(kind)
    st = _module_state()
    st["_modules"]["base"] = base
    st["_modules"]["dist"] = dist
    return st


# ---------------------------------------------------------------------------------- device wrappers
class _Box(object):
    def __init__(self, n):
        self.shape = (int(n),)


_Box.__name__ = "Box"


def load_policy(path, ctx=None):
    """my_pybullet_envs/utils.py:24-57 / a2c/main.py:78-84: -> (Policy | SplitPolicy on the GPU, ob_rms dict or None)."""
    from .model import Policy
    from .model_split import SplitPolicy
    ck = read_reference_checkpoint(path)
    if ck["kind"] == "mlp":
        pol = Policy((ck["obs_dim"],), _Box(ck["act_dim"]), base_kwargs={"recurrent": False, "hidden_size": ck["hidden"]}, ctx=ctx,
                     critic_hidden=ck.get("critic_hidden"))
    else:
        pol = SplitPolicy((ck["obs_dim"],), _Box(ck["act_dim"]),
                          base_kwargs={"hidden_size": ck["hidden"], "num_feet": ck["num_feet"]}, ctx=ctx)
    pol.load_state_dict(ck["state_dict"])
    return pol, ck["ob_rms"]


def save_policy(path, policy, ob_rms=None):
    """a2c/main.py:260-269: a file the reference's `torch.load(path)` turns back into `[actor_critic, ob_rms]`."""
    save_reference_checkpoint(path, "mlp" if policy.KIND == 0 else "split", policy.state_dict(), ob_rms)


def load_discriminator(path, ctx=None):
    """my_pybullet_envs/utils.py:60-82: weights, Adam state, running returns of a saved discriminator."""
    from .algo.gail import Discriminator
    ck = read_reference_discriminator(path)
    d = Discriminator(ck["input_dim"], ck["hidden_dim"], None, ctx=ctx)
    d.set_flat_params(np.concatenate([v.reshape(-1) for v in ck["state_dict"].values()]))
    if ck["adam"] is not None:
        d.set_adam(*ck["adam"])
    if ck["returns"] is not None:
        d.returns = ck["returns"]
    return d, ck["ret_rms"]
