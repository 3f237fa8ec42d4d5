"""Import the SimGAN reference hot-path modules in THIS container (dev-only tool).

The reference (/root/reference, read-only) needs gym / pybullet / pybullet_data /
pybullet_utils at import time although the hot path never calls them (import
chain: a2c/model.py:28-29 -> a2c/utils.py:29 -> a2c/envs.py:25-28 and
a2c/algo/ppo.py:26 -> my_pybullet_envs/__init__.py:15-22).  Those four
top-level names are absent here, so we register empty stand-in *modules* in
sys.modules (no reference code is copied or modified) and then import the
reference's own classes.

This file is only used by tools/gen_golden.py to produce the committed
fixtures under tests/golden/.  Nothing under tests/, bench.py or the product
package imports it: /root/reference does not exist on the GPU box.
"""
import importlib.abc
import importlib.machinery
import sys
import types

REFERENCE_ROOT = "/root/reference"


class _Anything:
    """Attribute sink: any attribute access / call returns another sink."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


class Box:
    """Duck-typed gym.spaces.Box: the hot path reads only __class__.__name__ and .shape
    (a2c/model.py:55-57, a2c/storage.py:42-45)."""

    def __init__(self, low=None, high=None, shape=None, dtype=None):
        if shape is None:
            import numpy as np
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.low = low
        self.high = high


class _StubBase:
    """Real class so the reference's `class X(gym.Wrapper)` statements evaluate."""

    def __init__(self, *a, **k):
        pass


def _stub_getattr(modname):
    def _getattr(name):  # module-level __getattr__ (PEP 562)
        if name.startswith("__"):
            raise AttributeError(name)
        if name[:1].isupper():
            return type(name, (_StubBase,), {})
        return _Anything()
    return _getattr


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Serves empty stand-in modules for the four absent top-level packages."""
    ROOTS = ("gym", "pybullet", "pybullet_data", "pybullet_utils")

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        m.__path__ = []
        m._simgan_stub = True
        m.__getattr__ = _stub_getattr(spec.name)
        if spec.name in ("gym.spaces", "gym.spaces.box"):
            m.Box = Box
        if spec.name == "gym.envs.registration":
            m.registry = types.SimpleNamespace(env_specs={})
            m.register = lambda *a, **k: None
        if spec.name == "pybullet_data":
            m.getDataPath = lambda: "/nonexistent"
        return m

    def exec_module(self, module):
        pass


def install_stubs():
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.append(_StubFinder())


def import_reference():
    """Returns a namespace with the reference's hot-path classes."""
    import torch  # noqa: F401  (import before the stubs so torch never sees them)
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from third_party.a2c_ppo_acktr import algo  # noqa
    from third_party.a2c_ppo_acktr.algo import gail  # noqa
    from third_party.a2c_ppo_acktr.model import Policy  # noqa
    from third_party.a2c_ppo_acktr.model_split import SplitPolicy  # noqa
    from third_party.a2c_ppo_acktr.storage import RolloutStorage  # noqa
    from third_party.a2c_ppo_acktr.baselines.common.running_mean_std import RunningMeanStd  # noqa
    from third_party.a2c_ppo_acktr import utils as a2c_utils  # noqa
    from my_pybullet_envs import utils as gan_utils  # noqa  (expert trajectory helpers; pybullet itself is a stand-in)
    ns = types.SimpleNamespace(PPO=algo.PPO, Discriminator=gail.Discriminator, Policy=Policy,
                               SplitPolicy=SplitPolicy, RolloutStorage=RolloutStorage,
                               RunningMeanStd=RunningMeanStd, Box=Box, a2c_utils=a2c_utils, gan_utils=gan_utils)
    return ns


if __name__ == "__main__":
    ns = import_reference()
    print("reference imported:", [k for k in vars(ns)])
