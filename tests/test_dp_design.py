"""CPU, world_size 2 on gloo: the data-parallel decomposition used by libsimgan_hip.so for N > 1.

The library shards by environment column: every rank holds N/world columns
of the rollout and full replicas of the policy / discriminator / Adam state / expert matrix, and
issues, per optimizer step, ONE sum-all-reduce of its partial gradient (+ 3 loss sums), with the
loss mean taken over the GLOBAL minibatch.  Advantage statistics and the per-step return statistics
of the reward relabel are merged with two tiny all-reduces (sum, then squares about the global mean).

These tests replay exactly that sequence of collectives -- same order, same buffers as
sg_ppo_update / sg_disc_update_gail_dyn / sg_disc_relabel_rewards in simgan_amd/csrc -- with the CPU
oracle doing the per-rank math and gloo doing the reductions, and check that the result equals the
single-process oracle on the concatenated rollout (i.e. the reference's semantics at
num_processes = world * N) up to fp32 summation order.  simgan_amd/dist.py (the launcher plumbing
bench.py uses) is exercised on the way.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORLD = 2
T, N_LOC, O, A, F, H, HD = 6, 8, 11, 3, 9, 16, 16
M, E_P, B = 2, 2, 8


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_global(seed=0):
    """The world=1 problem: rollout with world*N_LOC columns, weights, expert, RNG artefacts."""
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    n_tot = WORLD * N_LOC
    f32 = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    g = dict(obs=f32(T + 1, n_tot, O), obs_feat=f32(T + 1, n_tot, F), actions=f32(T, n_tot, A),
             value_preds=f32(T + 1, n_tot), returns=f32(T + 1, n_tot), logp=f32(T, n_tot) * 0.1 - 3.0,
             masks=(rng.random((T + 1, n_tot)) > 0.15).astype(np.float32), expert=f32(5 * B, F))
    d = orc.dims(orc.KIND_MLP, O, A, H, 1)
    g["pi"] = (0.2 * f32(orc.policy_num_params(d))).astype(np.float32)
    g["dpar"] = (0.2 * f32(orc.disc_num_params(F, HD))).astype(np.float32)
    # per-rank local permutations (what each rank's generator would draw) + global expert perm / alpha
    tn_loc = T * N_LOC
    g["ppo_perms"] = np.stack([[rng.permutation(tn_loc) for _ in range(E_P)] for _ in range(WORLD)]).astype(np.int64)
    g["d_pperm"] = np.stack([rng.permutation(tn_loc) for _ in range(WORLD)]).astype(np.int64)
    g["d_eperm"] = rng.permutation(5 * B).astype(np.int64)
    g["alpha"] = rng.random(5 * B).astype(np.float32)
    return g


def to_global_rows(local_rows, rank):
    t, n = local_rows // N_LOC, local_rows % N_LOC
    return t * (WORLD * N_LOC) + rank * N_LOC + n


def shard(a, rank):
    return np.ascontiguousarray(a[:, rank * N_LOC:(rank + 1) * N_LOC])


def allreduce(x):
    t = torch.from_numpy(np.ascontiguousarray(x))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()


def worker(rank, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import oracle as orc
    from simgan_amd.dist import ProcessGroup   # imports no HIP code
    pg = ProcessGroup()
    assert (pg.rank, pg.world) == (rank, WORLD)
    uid = pg.broadcast_bytes(bytes(range(128)) if rank == 0 else None)
    assert uid == bytes(range(128))
    assert pg.max(float(rank)) == WORLD - 1 and pg.sum(1.0) == WORLD

    g = make_global()
    d = orc.dims(orc.KIND_MLP, O, A, H, 1)
    cfg = orc.ppo_cfg(0.2, E_P, M, 0.5, 0.01, 3e-4, 1e-5, 0.5, True)
    loc = {k: shard(g[k], rank) for k in ("obs", "obs_feat", "actions", "value_preds", "returns", "logp", "masks")}
    tn_loc = T * N_LOC

    # ---------------- PPO.update, data parallel (mirrors sg_ppo_update, world > 1)
    adv = (loc["returns"][:-1] - loc["value_preds"][:-1]).reshape(-1).astype(np.float32)
    stats = allreduce(np.array([adv.astype(np.float64).sum(), 0.0, float(adv.size)]))
    mean = np.float32(stats[0] / stats[2])
    sq = allreduce(np.array([((adv.astype(np.float64) - float(mean)) ** 2).sum()]))[0]
    std = np.float32(np.sqrt(sq / (stats[2] - 1.0)))
    adv = ((adv - mean) / (std + np.float32(1e-5))).astype(np.float32)
    pi, adam = g["pi"].copy(), orc.AdamState(g["pi"].size)
    mb_loc = tn_loc // M
    inv_B = 1.0 / (mb_loc * WORLD)
    loss_acc = np.zeros(3)
    for e in range(E_P):
        for k in range(M):
            rows = g["ppo_perms"][rank, e, k * mb_loc:(k + 1) * mb_loc]
            G, sums = orc.ppo_grad_rows(d, pi, cfg, loc["obs"], loc["actions"], loc["value_preds"], loc["returns"],
                                        loc["logp"], adv, rows, inv_B)
            buf = allreduce(np.concatenate([G, sums.astype(np.float32)]))      # ONE collective per step
            G = np.ascontiguousarray(buf[:-3])
            loss_acc += buf[-3:].astype(np.float64) * inv_B
            orc.ppo_apply(pi, G, adam, cfg)
    ppo_losses = loss_acc / (E_P * M)

    # ---------------- Discriminator.update_gail_dyn, data parallel (mirrors sg_disc_update_gail_dyn)
    dpar, dadam = g["dpar"].copy(), orc.AdamState(g["dpar"].size)
    b_loc = B // WORLD
    n_d = min(g["expert"].shape[0] // B, tn_loc // b_loc)
    next_feat = loc["obs_feat"][1:].reshape(-1, F)
    d_tot = np.zeros(3)
    for k in range(n_d):
        sl = slice(k * B + rank * b_loc, k * B + (rank + 1) * b_loc)     # this rank's slice of the GLOBAL batch
        e_rows = g["expert"][g["d_eperm"][sl]]
        p_rows = next_feat[g["d_pperm"][rank, k * b_loc:(k + 1) * b_loc]]
        G, sums = orc.disc_grad_rows(F, HD, dpar, e_rows, p_rows, g["alpha"][sl], 1.0 / B)
        buf = allreduce(np.concatenate([G, sums.astype(np.float32)]))
        orc.adam_step(dpar, np.ascontiguousarray(buf[:-3]), dadam, 1e-3, 1e-8)
        el, pl, gp = buf[-3] / B, buf[-2] / B, 10.0 * buf[-1] / B
        d_tot += np.array([el + pl + gp, el, pl])
    d_losses = d_tot / n_d

    # ---------------- reward relabel statistics, data parallel (mirrors sg_disc_relabel_rewards)
    n_glob = float(N_LOC * WORLD)
    raw = np.stack([orc.disc_predict_reward(F, HD, dpar, loc["obs_feat"][t + 1], 0.99, loc["masks"][t], -0.3)[0][:, 0]
                    for t in range(T)])
    rets = np.zeros((T, N_LOC), np.float32)
    ret = raw[0].copy()
    for t in range(T):
        ret = raw[t] if t == 0 else (ret * np.float32(0.99) * loc["masks"][t] + raw[t]).astype(np.float32)
        rets[t] = ret
    sums = allreduce(rets.astype(np.float64).sum(axis=1))
    means = (sums / n_glob).astype(np.float32)
    sqs = allreduce(((rets - means[:, None]).astype(np.float32).astype(np.float64) ** 2).sum(axis=1))
    rms = [0.0, 1.0, 1e-4]
    rewards = np.zeros_like(raw)
    for t in range(T):
        bmean, bvar = float(means[t]), float(np.float32(sqs[t] / n_glob))
        delta, tot = bmean - rms[0], rms[2] + n_glob
        m2 = rms[1] * rms[2] + bvar * n_glob + delta * delta * rms[2] * n_glob / tot
        rms = [rms[0] + delta * n_glob / tot, m2 / tot, tot]
        rewards[t] = np.clip(raw[t] / np.float32(np.sqrt(rms[1] + 1e-7)), -10, 10)
    all_rewards = [None] * WORLD
    dist.all_gather_object(all_rewards, rewards)

    # ---------------- replicated-D mode (the library's default for world > 1): all-gather the ranks'
    # next_obs_feat rows once, then every rank runs the same full-batch steps on the global row set
    feats = [None] * WORLD
    dist.all_gather_object(feats, np.ascontiguousarray(loc["obs_feat"][1:].reshape(-1, F)))
    feat_all = np.concatenate(feats, axis=0)                       # [world*TN_loc, F], rank-major
    rng_r = np.random.default_rng(77)                              # same seed on every rank
    pperm_r = rng_r.permutation(WORLD * tn_loc).astype(np.int64)
    dpar_r, dadam_r = g["dpar"].copy(), orc.AdamState(g["dpar"].size)
    n_d_r = min(g["expert"].shape[0] // B, WORLD * tn_loc // B)
    for k in range(n_d_r):
        sl = slice(k * B, (k + 1) * B)
        G, _ = orc.disc_grad_rows(F, HD, dpar_r, g["expert"][g["d_eperm"][sl]], feat_all[pperm_r[sl]], g["alpha"][sl], 1.0 / B)
        orc.adam_step(dpar_r, G, dadam_r, 1e-3, 1e-8)
    all_dpar_r = [None] * WORLD
    dist.all_gather_object(all_dpar_r, dpar_r)

    if rank == 0:
        out.update(pi=pi, ppo_losses=ppo_losses, dpar=dpar, d_losses=d_losses, n_d=n_d,
                   rewards=np.concatenate(all_rewards, axis=1), rms=rms, dpar_repl=all_dpar_r, pperm_repl=pperm_r)
    pg.shutdown()


def run_world():
    mgr = mp.Manager()
    out = mgr.dict()
    port = free_port()
    mp.spawn(worker, args=(port, out), nprocs=WORLD, join=True)
    return dict(out)


@pytest.fixture(scope="module")
def dp():
    return run_world()


def test_ppo_update_sharded_equals_single_process(dp):
    from oracle import oracle as orc
    from helpers import assert_close
    g = make_global()
    d = orc.dims(orc.KIND_MLP, O, A, H, 1)
    cfg = orc.ppo_cfg(0.2, E_P, M, 0.5, 0.01, 3e-4, 1e-5, 0.5, True)
    tn_loc, mb_loc = T * N_LOC, T * N_LOC // M
    # the single-process permutation that draws the same minibatches
    perms = np.zeros((E_P, WORLD * tn_loc), np.int64)
    for e in range(E_P):
        used = []
        for k in range(M):
            for r in range(WORLD):
                used.append(to_global_rows(g["ppo_perms"][r, e, k * mb_loc:(k + 1) * mb_loc], r))
        used = np.concatenate(used)
        rest = np.setdiff1d(np.arange(WORLD * tn_loc), used)
        perms[e] = np.concatenate([used, rest])
    pi, adam = g["pi"].copy(), orc.AdamState(g["pi"].size)
    losses = orc.ppo_update(d, pi, adam, cfg, g["obs"], g["actions"], g["value_preds"], g["returns"], g["logp"], perms)
    assert_close(dp["ppo_losses"], losses, rtol=1e-5, what="DP ppo losses")
    assert_close(dp["pi"], pi, rtol=1e-4, atol=2e-6, what="DP policy params")
    assert np.max(np.abs(pi - g["pi"])) > 1e-4


def test_disc_update_sharded_equals_single_process(dp):
    from oracle import oracle as orc
    from helpers import assert_close
    g = make_global()
    tn_loc, b_loc = T * N_LOC, B // WORLD
    n_d = dp["n_d"]
    pperm = np.zeros(WORLD * tn_loc, np.int64)
    for k in range(n_d):
        for r in range(WORLD):
            pperm[k * B + r * b_loc:k * B + (r + 1) * b_loc] = to_global_rows(g["d_pperm"][r, k * b_loc:(k + 1) * b_loc], r)
    dpar, adam = g["dpar"].copy(), orc.AdamState(g["dpar"].size)
    # truncate the expert set so the single-process run stops after the same n_d steps
    losses, n1 = orc.disc_update(F, HD, dpar, adam, g["expert"], g["obs_feat"], B, g["d_eperm"], pperm, g["alpha"])
    assert n1 == n_d
    assert_close(dp["d_losses"], losses, rtol=1e-5, what="DP D losses")
    assert_close(dp["dpar"], dpar, rtol=1e-4, atol=2e-6, what="DP D params")


def test_relabel_sharded_equals_single_process(dp):
    from oracle import oracle as orc
    from helpers import assert_close
    g = make_global()
    rewards, _, rms = orc.relabel(F, HD, dp["dpar"], g["obs_feat"], g["masks"], 0.99, -0.3, None, [0.0, 1.0, 1e-4])
    assert_close(dp["rewards"], rewards, rtol=1e-5, what="DP relabelled rewards")
    assert_close(dp["rms"], rms, rtol=1e-6, what="DP ret_rms")


def test_disc_update_replicated_equals_single_process(dp):
    """Replicated-D mode: replicas stay bit-identical across ranks and equal the single-process update
    on the concatenated rollout (row ids mapped rank-major -> (t, n) of the world-size rollout)."""
    from oracle import oracle as orc
    from helpers import assert_close
    g = make_global()
    tn_loc = T * N_LOC
    for other in dp["dpar_repl"][1:]:
        assert np.array_equal(dp["dpar_repl"][0], other), "replicas diverged"
    # rank-major global index j = rank*TN_loc + local_row  ->  flattened (t, n_global) row of the full rollout
    j = dp["pperm_repl"]
    pperm = to_global_rows(j % tn_loc, j // tn_loc)
    dpar, adam = g["dpar"].copy(), orc.AdamState(g["dpar"].size)
    orc.disc_update(F, HD, dpar, adam, g["expert"], g["obs_feat"], B, g["d_eperm"], pperm, g["alpha"])
    assert_close(dp["dpar_repl"][0], dpar, rtol=1e-6, atol=1e-7, what="replicated D params")
