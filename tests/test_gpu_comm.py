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
