"""GPU: the peer-mesh all-reduce (SG_COMM_PEER=1, csrc/sg_comm.cpp) -- the per-step float32 gradient all-reduce as ONE kernel
over peer-mapped device memory -- on one device: ranks as threads of one process (plain pointers) and as processes
(hipIpc handles, through bench.py --loopback).  It adds the ranks' vectors in rank order, the loopback transport's order, so
everything downstream must be BIT-identical to the same run without it, and equal to the world-1 oracle as before.
Across GPUs (xGMI) it has never run: a gpurun box has one GPU.

The thread-rank cases run in a FRESH interpreter with GPU_MAX_HW_QUEUES=8: a collective kernel waits for the other ranks'
kernels, and streams of one process that share a hardware queue (4 by default, assigned by the runtime as it sees fit -- the
pytest process holds streams of earlier tests) cannot wait for each other.  `python tests/test_gpu_peer.py` runs them directly."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _ppo_world(tw, world, name, peer):
    import simgan_amd as sg
    os.environ["SG_COMM_PEER"] = "1" if peer else "0"
    hp, g = tw.prepare(world, name)
    next_value = np.random.default_rng(5).standard_normal(g["Ng"]).astype(np.float32)

    def rank_fn(rank, ctx):
        assert ctx.comm_peer() == peer and ctx.comm_kind() == "loopback"
        pol, agent, _, ro = tw.build_rank(sg, g, rank, ctx, hp)
        ro.compute_returns(next_value[rank * hp["N_loc"]:(rank + 1) * hp["N_loc"]], True, tw.GAMMA, tw.LAM, True)
        losses = [agent.update(ro, perms=g["ppo_perms"]) for _ in range(2)]
        return dict(losses=losses, pi=pol.get_flat_params())

    return hp, g, next_value, tw.run_ranks(world, rank_fn)


def case_ppo(tw, world, name):
    """PPO over the mesh: replicas identical, bit-identical to the loopback sum, equal to the world-1 oracle."""
    from oracle import oracle as orc
    from helpers import assert_close
    hp, g, next_value, res = _ppo_world(tw, world, name, True)
    _, _, _, ref = _ppo_world(tw, world, name, False)
    for r in res:
        assert np.array_equal(r["pi"], res[0]["pi"]) and r["losses"] == res[0]["losses"], "replicas diverged"
    assert np.array_equal(res[0]["pi"], ref[0]["pi"]) and res[0]["losses"] == ref[0]["losses"], "peer sum differs from the loopback sum"
    ret, vp = tw.oracle_returns(g, next_value)
    pi, adam = g["pi"].copy(), orc.AdamState(g["pi"].size)
    cfg = orc.ppo_cfg(hp["clip"], hp["E"], hp["M"], 0.5, hp["ecoef"], 3e-4, 1e-5, 0.5, True)
    want = [orc.ppo_update(g["d"], pi, adam, cfg, g["obs"], g["actions"], vp, ret, g["logp"], g["ppo_perms"]) for _ in range(2)]
    assert_close(res[0]["losses"][0], want[0], what="PPO losses, first update")
    assert_close(res[0]["pi"], pi, rtol=2e-4, atol=2e-5, what="policy after two updates")


def case_disc_sharded(tw, world, name):
    """One gradient all-reduce per discriminator step (sharded mode): 12+ collectives back to back exercise the double-buffered
    slots and the collective counter."""
    import simgan_amd as sg

    def run(peer):
        os.environ["SG_COMM_PEER"] = "1" if peer else "0"
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


def case_refused(tw):
    """More contexts of one process than it has hardware queues: refused at set-up with an explanation, not a time-out."""
    os.environ["SG_COMM_PEER"] = "1"
    os.environ["GPU_MAX_HW_QUEUES_SAVED"] = os.environ.get("GPU_MAX_HW_QUEUES", "")
    os.environ["GPU_MAX_HW_QUEUES"] = "4"   # (read by the library's check at set-up; the runtime took its own copy at start-up)
    try:
        tw.run_ranks(8, lambda rank, ctx: None, timeout_s=120)
    except AssertionError as exc:
        assert "hardware queues" in str(exc), str(exc)
    else:
        raise AssertionError("8 same-process ranks on 4 hardware queues were accepted")
    finally:
        os.environ["GPU_MAX_HW_QUEUES"] = os.environ.pop("GPU_MAX_HW_QUEUES_SAVED") or "8"


def case_setup_failure_is_collective(tw):
    """One rank cannot open a peer's buffer (injected): EVERY rank's set-up returns an error -- the failing rank its own, the
    others "1 of 2 ranks could not ..." -- and nobody is left running the mesh kernel against a rank that is not in it.  The same
    communicators then take the mesh through sg_ctx_comm_set_peer, run a PPO update over it, drop it again, and agree with the
    run that never had one."""
    from simgan_amd import _lib
    os.environ["SG_COMM_PEER"] = "1"
    os.environ["SG_COMM_PEER_FAIL_RANK"] = "1"
    seen = {}

    def rank_fn_init(rank, ctx):
        return None

    # (a) at sg_ctx_comm_init: run_ranks' own comm_init raises on every rank
    import threading
    uid = _lib.comm_loopback_id()
    errs = [None, None]

    def body(rank):
        try:
            _lib.Context(0).comm_init(uid, rank, 2)
        except _lib.SimganHipError as exc:
            errs[rank] = str(exc)

    th = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(2)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not any(t.is_alive() for t in th), "a rank hangs in the set-up of a mesh another rank failed to join"
    assert errs[1] and "injected by SG_COMM_PEER_FAIL_RANK" in errs[1] and "no rank uses the mesh" in errs[1], errs
    assert errs[0] and "1 of 2 ranks could not set the mesh up" in errs[0], errs
    # (b) toggled on a live communicator
    os.environ["SG_COMM_PEER"] = "0"
    os.environ.pop("SG_COMM_PEER_FAIL_RANK")

    def rank_fn(rank, ctx):
        assert not ctx.comm_peer()
        os.environ["SG_COMM_PEER_FAIL_RANK"] = "0"        # (process-wide: both ranks read it; only rank 0 matches)
        try:
            ctx.comm_set_peer(True)
        except _lib.SimganHipError as exc:
            seen[rank] = str(exc)
        assert not ctx.comm_peer()
        return None

    tw.run_ranks(2, rank_fn, timeout_s=120)
    assert "injected" in seen.get(0, "") and "1 of 2 ranks" in seen.get(1, ""), seen
    os.environ.pop("SG_COMM_PEER_FAIL_RANK")

    def rank_fn2(rank, ctx):
        ctx.comm_set_peer(True)
        assert ctx.comm_peer()
        ctx.comm_set_peer(False)
        assert not ctx.comm_peer()
        ctx.comm_set_peer(True)
        return ctx.comm_peer()

    assert tw.run_ranks(2, rank_fn2, timeout_s=120) == [True, True]


THREAD_CASES = [("setup_failure",), ("ppo", 2, "mlp_small"), ("ppo", 3, "mlp_small"), ("ppo", 4, "northstar"), ("ppo", 4, "split"), ("ppo", 8, "northstar"),
                ("disc_sharded", 2, "northstar"), ("disc_sharded", 4, "split"), ("refused",)]


def main():
    import test_gpu_world as tw
    for case in THREAD_CASES:
        {"ppo": case_ppo, "disc_sharded": case_disc_sharded, "refused": case_refused, "setup_failure": case_setup_failure_is_collective}[case[0]](tw, *case[1:])
        print("OK", *case, flush=True)


def test_thread_ranks_over_the_peer_mesh_equal_the_loopback_sums_and_the_oracle():
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", HSA_ENABLE_IPC_MODE_LEGACY="0", SG_LOOPBACK_TIMEOUT_S="120")
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    done = [ln for ln in r.stdout.splitlines() if ln.startswith("OK ")]
    assert r.returncode == 0 and len(done) == len(THREAD_CASES), (done, r.stdout[-1500:], r.stderr[-4000:])


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


@pytest.mark.parametrize("gpus", [2, 4])
def test_ranks_in_separate_processes_map_each_other_through_hipipc(gpus):
    """bench.py --gpus N --loopback: N PROCESSES on the device, their slot buffers opened through hipIpc handles that travel
    over the base communicator's all-gather; 80 PPO steps per update, each with one peer all-reduce.  Same losses as without."""
    a = _bench({"SG_COMM_PEER": "1"}, gpus)
    b = _bench({"SG_COMM_PEER": "0"}, gpus)
    assert a["comm"]["peer_allreduce"] is True and b["comm"]["peer_allreduce"] is False
    assert a["replica_check"]["ok"] is True
    assert a["last_losses"] == b["last_losses"], (a["last_losses"], b["last_losses"])
    assert a["replica_check"]["weights_sha256_rank0"] == b["replica_check"]["weights_sha256_rank0"]


if __name__ == "__main__":
    main()
