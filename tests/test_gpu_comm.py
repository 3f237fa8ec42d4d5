"""GPU, one rank: the RCCL data path of the library (communicator creation, gradient / statistics
all-reduces enqueued on the library's stream between the gradient and optimizer kernels).  With
SG_COMM_ALWAYS=1 the collectives stay in the launch sequence for world_size 1, where they are the
identity, so the results must equal the golden fixtures exactly as in the non-collective path.
Multi-rank equivalence of the decomposition itself is covered on CPU by tests/test_dp_design.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, "tests")
import numpy as np
import simgan_amd as sg
from simgan_amd import _lib
from helpers import assert_close, load
from test_gpu_parity import Box, Loader, make_policy, fill_rollout

ctx = _lib.Context.default()
ctx.comm_init(_lib.comm_unique_id(), 0, 1)          # ncclCommInitRank(world=1) via dlopen'd RCCL
g = load("ppo_mlp_northstar"); m = g["meta"]
p = make_policy(sg, m); p.set_flat_params(g["params0"])
ro = sg.RolloutStorage(m["T"], m["N"], (m["O"],), Box((m["A"],)), 1, g["obs_feat"].shape[-1])
fill_rollout(ro, g)
agent = sg.algo.PPO(p, m["clip_param"], m["ppo_epoch"], m["num_mini_batch"], m["value_loss_coef"], m["entropy_coef"],
                    lr=m["lr"], eps=m["eps"], max_grad_norm=m["max_grad_norm"])
assert_close(agent.update(ro, perms=g["perms"]), g["losses"], what="ppo losses (collective path)")
assert_close(p.get_flat_params(), g["params1"], what="params (collective path)")
g = load("disc_northstar"); m = g["meta"]
D = sg.algo.gail.Discriminator(m["F"], m["Hd"], None); D.set_flat_params(g["params0"])
ro = sg.RolloutStorage(m["T"], m["N"], (3,), Box((2,)), 1, m["F"])
ro.obs_feat.copy_(ro.obs_feat.new_tensor(g["obs_feat"]))
for ep in range(m["epochs"]):
    losses = D.update_gail_dyn(Loader(g["expert"], m["B"]), ro, expert_perm=g[f"expert_perm{ep}"],
                               policy_perm=g[f"policy_perm{ep}"], alpha=g[f"alpha{ep}"])
    assert_close(losses, g[f"losses{ep}"], what="D losses (collective path)")
    assert_close(D.get_flat_params(), g[f"params_after{ep}"], what="D params (collective path)")
g = load("relabel_northstar"); m = g["meta"]
D = sg.algo.gail.Discriminator(m["F"], m["Hd"], None); D.set_flat_params(g["params"])
rms = sg.RunningMeanStd(shape=())
for call in range(2):
    ro = sg.RolloutStorage(m["T"], m["N"], (3,), Box((2,)), 1, m["F"])
    ro.obs_feat.copy_(ro.obs_feat.new_tensor(g[f"obs_feat{call}"])); ro.masks.copy_(ro.masks.new_tensor(g[f"masks{call}"]))
    D.relabel_rewards(ro, m["gamma"], float(g[f"offset{call}"]), rms)
    assert_close(ro.rewards.numpy(), g[f"rewards{call}"], what="rewards (collective path)")
print("COMM-PATH-OK")
'''


@pytest.mark.parametrize("disc_mode", ["replicated", "sharded"])
def test_rccl_path_single_rank(disc_mode):
    """replicated: all-gather of next_obs_feat + local D steps; sharded: per-step gradient all-reduce."""
    env = dict(os.environ, SG_COMM_ALWAYS="1", SG_DISC_DP=disc_mode, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "COMM-PATH-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


GRAPH_CHILD = r'''
import ctypes as C, json, os, sys
import numpy as np
import simgan_amd as sg
from simgan_amd import _lib

class Box:
    def __init__(self, shape): self.shape = tuple(shape)
class Loader:
    def __init__(self, expert, batch_size): self.expert, self.batch_size = expert, batch_size

ctx = _lib.Context.default()
ctx.comm_init(_lib.comm_unique_id(), 0, 1)
assert ctx.comm_info() == (0, 1)                     # as RCCL itself reports it
lib = _lib.load()
T, N, O, A, F = 16, 64, 47, 12, 86
pol = sg.Policy((O,), Box((A,)), base_kwargs={"recurrent": False, "hidden_size": 64}, seed=1)
disc = sg.algo.gail.Discriminator(F, 100, None, seed=2)
agent = sg.algo.PPO(pol, 0.2, 3, 4, 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
agent.seed, disc.seed = 5, 6
ro = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, F)
ro.device_resident = True
_lib.check(lib.sg_rollout_fill_synthetic(ro.h, pol.h, 7, 0.02))
_lib.check(lib.sg_rollout_compute_returns_policy(ro.h, pol.h, 1, 0.99, 0.95, 1))
expert = np.random.default_rng(0).standard_normal((640, F)).astype(np.float32)
out = []
for _ in range(2):                                    # second round replays the instantiated graphs
    out.append(disc.update_gail_dyn(Loader(expert, 128), ro))
    out.append(agent.update(ro))
fn = _lib.load_test().sg_test_graph_state
st = (C.c_int * 2)()
_lib.check_test(fn(agent.h, disc.h, st))
np.savez(sys.argv[1], losses=np.array(out, np.float64), pi=pol.get_flat_params(), d=disc.get_flat_params(), state=np.array(list(st)))
print("GRAPH-CHILD-OK")
'''


@pytest.mark.parametrize("disc_mode", ["replicated", "sharded"])
def test_collectives_inside_the_graphs_equal_direct_launches(disc_mode, tmp_path):
    """With a communicator the RCCL all-reduces are captured into the PPO graph (and, in sharded mode, into the
    discriminator epoch graph).  One rank, SG_COMM_ALWAYS=1: the captured path must be taken and must give results
    bit-identical to issuing the same kernels and collectives directly."""
    import numpy as np
    res = {}
    for tag, extra in (("graph", {}), ("direct", {"SG_PPO_GRAPH_COMM": "0", "SG_DISC_GRAPH_COMM": "0"})):
        env = dict(os.environ, SG_COMM_ALWAYS="1", SG_DISC_DP=disc_mode, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, **extra)
        out = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, "-c", GRAPH_CHILD, out], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "GRAPH-CHILD-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
        res[tag] = np.load(out)
    g, e = res["graph"], res["direct"]
    assert list(g["state"]) == [1, 1 if disc_mode == "sharded" else 1], list(g["state"])   # both replayed graphs
    assert list(e["state"]) == [0, 0 if disc_mode == "sharded" else 1], list(e["state"])   # replicated D has no per-step collective
    assert np.array_equal(g["losses"], e["losses"]) and np.array_equal(g["pi"], e["pi"]) and np.array_equal(g["d"], e["d"])
