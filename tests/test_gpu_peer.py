"""GPU: the peer-mesh all-reduce (SG_COMM_PEER=1, csrc/sg_comm.cpp) -- the per-step float32 gradient all-reduce as ONE kernel
over peer-mapped device memory -- on one device: ranks as threads of one process (plain pointers) and as processes
(hipIpc handles, through bench.py --loopback).  It adds the ranks' vectors in rank order, the loopback transport's order, so
everything downstream must be BIT-identical to the same run without it, and equal to the world-1 oracle as before.
Across GPUs (xGMI) it has never run: a gpurun box has one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_world as tw  # noqa: E402

pytestmark = pytest.mark.gpu


def _ppo_world(world, name, peer, monkeypatch):
    import simgan_amd as sg
    monkeypatch.setenv("SG_COMM_PEER", "1" if peer else "0")
    hp, g = tw.prepare(world, name)
    next_value = np.random.default_rng(5).standard_normal(g["Ng"]).astype(np.float32)

    def rank_fn(rank, ctx):
        assert ctx.comm_peer() == peer and ctx.comm_kind() == "loopback"
        pol, agent, _, ro = tw.build_rank(sg, g, rank, ctx, hp)
        ro.compute_returns(next_value[rank * hp["N_loc"]:(rank + 1) * hp["N_loc"]], True, tw.GAMMA, tw.LAM, True)
        losses = [agent.update(ro, perms=g["ppo_perms"]) for _ in range(2)]
        return dict(losses=losses, pi=pol.get_flat_params())

    return hp, g, next_value, tw.run_ranks(world, rank_fn)


@pytest.mark.parametrize("world,name", [(2, "mlp_small"), (4, "northstar"), (3, "mlp_small"), (4, "split")])
def test_ppo_over_the_peer_mesh_is_bit_identical_to_the_loopback_sum_and_equals_the_oracle(world, name, monkeypatch):
    from oracle import oracle as orc
    from helpers import assert_close
    hp, g, next_value, res = _ppo_world(world, name, True, monkeypatch)
    _, _, _, ref = _ppo_world(world, name, False, monkeypatch)
    for r in res:
        assert np.array_equal(r["pi"], res[0]["pi"]) and r["losses"] == res[0]["losses"], "replicas diverged"
    assert np.array_equal(res[0]["pi"], ref[0]["pi"]) and res[0]["losses"] == ref[0]["losses"], "peer sum differs from the loopback sum"
    ret, vp = tw.oracle_returns(g, next_value)
    pi, adam = g["pi"].copy(), orc.AdamState(g["pi"].size)
    cfg = orc.ppo_cfg(hp["clip"], hp["E"], hp["M"], 0.5, hp["ecoef"], 3e-4, 1e-5, 0.5, True)
    want = [orc.ppo_update(g["d"], pi, adam, cfg, g["obs"], g["actions"], vp, ret, g["logp"], g["ppo_perms"]) for _ in range(2)]
    assert_close(res[0]["losses"][0], want[0], what="PPO losses, first update")
    assert_close(res[0]["pi"], pi, rtol=2e-4, atol=2e-5, what="policy after two updates")


@pytest.mark.parametrize("world,name", [(2, "northstar"), (4, "split")])
def test_sharded_discriminator_over_the_peer_mesh(world, name, monkeypatch):
    """One gradient all-reduce per discriminator step (sharded mode): 8+ collectives back to back exercise the double-buffered
    slots and the collective counter."""
    import simgan_amd as sg

    def run(peer):
        monkeypatch.setenv("SG_COMM_PEER", "1" if peer else "0")
        hp, g = tw.prepare(world, name)

        def rank_fn(rank, ctx):
            assert ctx.comm_peer() == peer
            ctx.set_disc_dp(True)
            _, _, disc, ro = tw.build_rank(sg, g, rank, ctx, hp)
            out = [disc.update_gail_dyn(tw.Loader(g["expert"], hp["B"]), ro, expert_perm=g["d_eperm"], policy_perm=g["d_pperm"], alpha=g["alpha"])
                   for _ in range(3)]
            return dict(losses=out, dpar=disc.get_flat_params(), n=disc.last_n_steps)
        return tw.run_ranks(world, rank_fn)

    res, ref = run(True), run(False)
    assert 3 * res[0]["n"] >= 8
    for r in res:
        assert np.array_equal(r["dpar"], res[0]["dpar"]), "replicas diverged"
    assert np.array_equal(res[0]["dpar"], ref[0]["dpar"]) and res[0]["losses"] == ref[0]["losses"]


def _bench(extra_env, gpus=2):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SG_LOOPBACK_TIMEOUT_S="120", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(gpus), "--loopback", "--steps", "2", "--warmup", "1", "--workload", "refine",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_ranks_in_separate_processes_map_each_other_through_hipipc():
    """bench.py --gpus 2 --loopback: two PROCESSES on the device, their slot buffers opened through hipIpc handles that travel
    over the base communicator's all-gather; 16 PPO steps per update, each with one peer all-reduce.  Same losses as without."""
    a = _bench({"SG_COMM_PEER": "1"})
    b = _bench({"SG_COMM_PEER": "0"})
    assert a["comm"]["peer_allreduce"] is True and b["comm"]["peer_allreduce"] is False
    assert a["replica_check"]["ok"] is True
    assert a["last_losses"] == b["last_losses"], (a["last_losses"], b["last_losses"])
    assert a["replica_check"]["weights_sha256_rank0"] == b["replica_check"]["weights_sha256_rank0"]


def test_more_ranks_in_one_process_than_hardware_queues_are_refused(monkeypatch):
    """Contexts of ONE process share its hardware queues (4 by default): a kernel that waits for the kernel of a stream queued
    behind it would never finish, so the mesh refuses that set-up with an explanation instead of timing out."""
    monkeypatch.setenv("SG_COMM_PEER", "1")
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    with pytest.raises(AssertionError, match="hardware queues"):
        tw.run_ranks(8, lambda rank, ctx: None, timeout_s=120)
