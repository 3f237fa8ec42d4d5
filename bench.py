#!/usr/bin/env python
"""bench.py -- env-steps/s of the GAIL-dyn PPO update on MI355X.

One "step" = one full update of the reference's outer loop body on an already-collected synthetic
rollout (a2c/main_gail_dyn_ppo.py:255-302, a2c/ = third_party/a2c_ppo_acktr/):
    gail_epoch x Discriminator.update_gail_dyn  ->  alive-bonus offset  ->  T x predict_reward
    (+ return normalisation)  ->  compute_returns (GAE)  ->  PPO.update
Inputs (rollout, expert matrix, weights, optimizer state) are resident in HBM before the timed
region starts; nothing is skipped inside it.

    python bench.py --gpus N --steps K --warmup W [--workload northstar|hopper|laikago|refine|hopper_ppo]

`--workload refine` is the plain-PPO caller (a2c/main.py:199-257, BASELINE.json configs[4]: Laikago policy refinement,
2048 envs -> 256 per GPU, obs 111, 8 minibatches, clip 0.1, lr 1.5e-4 with linear decay): no discriminator, one step =
get_value + GAE + PPO.update + after_update.

N > 1: one process per GPU; control plane (barrier, id broadcast, max-over-ranks) on gloo, data plane (gradient
all-reduce) on RCCL inside the library.  Either launched by torch.distributed.run (RANK / WORLD_SIZE in the environment),
or plainly as `python bench.py --gpus N`: it then starts the N ranks itself (the same torch.distributed.run command),
forwards rank 0's line and returns non-zero if any rank failed.  Prints ONE JSON line on rank 0.

The line is COMPACT (about 1.7 KB, `compact_line`): every key of the driver's contract, the dominant kernel's roofline (HIP-event
and rocprofv3 clocks, counter traffic, SQ MFMA-busy), the whole update's MFMA / HBM fractions, the CPU baseline, one
[ms, env-steps/s, frac, MFMA-busy] quadruple per other BASELINE.json configuration, the drop-in legs, and for N > 1 the comm block
(which discriminator mode ran, the other mode's value, both all-reduce forms).  The FULL record -- per-kernel tables, spreads,
sample descriptions, per-rank lists -- is written to gpurun_out/bench_full_<workload>_n<N>.json (named in the line as
`full_record`) and to stderr; `--full-line` prints it on stdout instead.

`--loopback` (self-test of the N > 1 path on a box with fewer than N GPUs): the ranks share the visible device(s) and use
the library's shared-memory loopback communicator instead of RCCL; the line says so and is not a scaling measurement.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json north_star synthetic shape; BASELINE.md section 4 prices this one
    "northstar": dict(kind="mlp", T=128, N=512, O=47, A=12, F=86, H=64, feet=1, Hd=100, E_p=10, M=16, E_d=5,
                      B=128, Ne=100000, clip=0.2),
    # BASELINE.json configs[1]: HopperCombinedEnv-v1 GAIL-dyn, num_processes=256, SplitPolicy h100 as shipped
    "hopper": dict(kind="split", T=128, N=256, O=14, A=7, F=25, H=100, feet=1, Hd=100, E_p=10, M=16, E_d=5,
                   B=128, Ne=100000, clip=0.2),
    # BASELINE.json configs[2] real shapes: LaikagoCombinedEnv-v1, SplitPolicy h100 nf=4
    "laikago": dict(kind="split", T=128, N=512, O=64, A=28, F=86, H=100, feet=4, Hd=100, E_p=10, M=16, E_d=5,
                    B=128, Ne=100000, clip=0.2),
    # BASELINE.json configs[4]: Laikago policy refinement (train_laika_power.sh:7), main.py caller, no discriminator
    "refine": dict(kind="mlp", T=128, N=256, O=111, A=12, F=111, H=64, feet=1, Hd=100, E_p=10, M=8, E_d=0,
                   B=128, Ne=0, clip=0.1, lr=1.5e-4),
    # BASELINE.json configs[0]: HopperURDFEnv-v3 plain PPO at the reference's own CPU-runnable geometry (a2c/main.py with the
    # defaults of a2c/arguments.py: 8 processes x 128 steps, 32 minibatches -> 32-ROW optimizer steps, 320 per update, entropy
    # coefficient 0.01, no LR decay): the smallest grid k_ppo_bwd ever runs on (one workgroup per trunk)
    "hopper_ppo": dict(kind="mlp", T=128, N=8, O=11, A=3, F=11, H=64, feet=1, Hd=100, E_p=10, M=32, E_d=0,
                       B=128, Ne=0, clip=0.2, lr=3e-4, ecoef=0.01, lr_decay=False),
}
GAMMA, LAM = 0.99, 0.95
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0


class Box:
    def __init__(self, shape):
        self.shape = tuple(shape)


class Loader:
    def __init__(self, expert, batch_size):
        self.expert, self.batch_size = expert, batch_size


def algorithmic_work(w, world, disc_sharded=False):
    """Per-update algorithmic FLOPs and HBM bytes for ONE rank (SURVEY.md 8(d) formulas).  world > 1: the PPO rows are
    always sharded; the discriminator's are only in sharded mode -- in the default replicated mode every rank runs the
    full-batch steps."""
    T, N, O, A, F, H, Hd, B = w["T"], w["N"], w["O"], w["A"], w["F"], w["H"], w["Hd"], w["B"]
    TN = T * N
    n_d = min(w["Ne"] // B, TN * world // B) if w["E_d"] else 0
    Wd = F * Hd + Hd * Hd + Hd
    d_flops_triple = 2 * (8 * Wd + F * Hd + 4 * Hd * Hd + Hd)
    if w["kind"] == "mlp":
        fwd = 2 * (O * H + H * H) + H + H * A
        n_tr = 2
    else:
        fwd = 3 * (O * H + H * H) + H + 2 * H * A
        n_tr = 3
    ppo_flops_row = 2 * (3 * fwd - n_tr * O * H)
    mb = TN // w["M"]
    B_rank = B // world if disc_sharded else B
    d_step_flops = B_rank * d_flops_triple
    d_wgrad_flops = B_rank * 2 * 4 * (F * Hd + Hd * Hd)   # dW1, dW2 over expert + policy + 2 mixup terms
    d_chain_flops = d_step_flops - d_wgrad_flops                 # forward + activation backward + double backward
    ppo_step_flops = mb * ppo_flops_row
    relabel_flops = TN * 2 * Wd if w["E_d"] else 0
    flops = w["E_d"] * n_d * d_step_flops + w["E_p"] * w["M"] * ppo_step_flops + relabel_flops
    bytes_ = (w["E_d"] * n_d * 2 * B_rank * F * 4 + (TN * (F + 2) * 4 if w["E_d"] else 0) + TN * 20 + TN * 12 +
              w["E_p"] * w["M"] * mb * (O + A + 4) * 4)
    # per-launch algorithmic work of every profiled kernel: (kernel name, bound, FLOPs, compulsory HBM bytes)
    P_pi = (2 * (O * H + H + H * H + H) + H + 1 + H * A + A + A) if w["kind"] == "mlp" else \
           (3 * (O * H + H + H * H + H) + H + 1 + 2 * (H * 4 * w["feet"] + 4 * w["feet"]) + 2 * (H * 3 * w["feet"] + 3 * w["feet"]))
    fused = w["kind"] == "mlp"    # Policy: k_ppo_bwd recomputes the forward, no k_ppo_fwd launch
    fwd_flops = mb * 2 * fwd
    # SplitPolicy with more (32-row group, trunk) workgroups than the 256 CUs: the critic's whole fused forward + backward
    # rides in the forward launch and the backward launch covers the two actor trunks (csrc/sg_ppo.hip: crit_first)
    crit_first = (not fused) and ((mb + 31) // 32) * 3 > 256
    crit_fwd = O * H + H * H + H
    crit_bwd_flops = mb * 2 * (2 * crit_fwd - O * H) if crit_first else 0
    # SplitPolicy whose 3 G workgroups are resident together: ONE launch per step, every trunk fused, the actor pairs exchanging
    # their head outputs inside it (csrc/sg_ppo.hip: pair; k_ppo_pair) -- no forward launch
    pair = (not fused) and (not crit_first) and ((mb + 31) // 32) * 3 <= 256 and os.environ.get("SG_PPO_PAIR") != "0"
    kernels = {
        "disc_chain": ("k_disc_chain4", "mfma", d_chain_flops, 2 * B_rank * F * 4),
        "disc_wgrad": ("k_disc_wgrad", "mfma", d_wgrad_flops, 0),
        # one launch per optimizer step (csrc/sg_disc_step4.hpp): the chain and the weight-gradient workgroups side by side
        "disc_step": ("k_disc_step4", "mfma", d_step_flops, 2 * B_rank * F * 4),
        "ppo_fwd": ("k_ppo_fwd_critic" if crit_first else "k_ppo_fwd", "mfma", 0 if fused or pair else fwd_flops + crit_bwd_flops,
                    0 if fused or pair else mb * (O + (3 if crit_first else 0)) * 4),
        "ppo_bwd": ("k_ppo_pair" if pair else "k_ppo_bwd", "mfma", ppo_step_flops - (0 if fused or pair else fwd_flops) - crit_bwd_flops,
                    mb * (O + A + 4) * 4),
        "ppo_reduce": ("k_ppo_reduce", "hbm", 0, P_pi * 4),          # the gradient vector; the slabs it sums are an implementation artefact
        "ppo_adam": ("k_ppo_adam", "hbm", 0, P_pi * 4 * 7),          # grad + params / m / v read and written
        "relabel_fwd": ("k_disc_forward", "mfma", relabel_flops, TN * (F + 1) * 4 if w["E_d"] else 0),
    }
    return dict(n_d=n_d, d_step_flops=d_step_flops, d_chain_flops=d_chain_flops, d_wgrad_flops=d_wgrad_flops, ppo_step_flops=ppo_step_flops, flops=flops, bytes=bytes_,
                d_steps=w["E_d"] * n_d, ppo_steps=w["E_p"] * w["M"], kernels=kernels)


def build_problem(sg, w, seed):
    from simgan_amd.driver import ExpertLoader, GailDynLearner, PpoLearner
    rng = np.random.default_rng(seed)
    lr = w.get("lr", 3e-4)
    if w["kind"] == "mlp":
        pol = sg.Policy((w["O"],), Box((w["A"],)), base_kwargs={"recurrent": False, "hidden_size": w["H"]}, seed=seed)
    else:
        pol = sg.SplitPolicy((w["O"],), Box((w["A"],)), base_kwargs={"hidden_size": w["H"], "num_feet": w["feet"]}, seed=seed)
    agent = sg.algo.PPO(pol, w["clip"], w["E_p"], w["M"], 0.5, w.get("ecoef", 0.0), lr=lr, eps=1e-5, max_grad_norm=0.5)
    ro = sg.RolloutStorage(w["T"], w["N"], (w["O"],), Box((w["A"],)), 1, w["F"])
    ro.device_resident = True
    if not w["E_d"]:   # a2c/main.py: PPO only, linear LR decay over the run's updates (train_laika_power.sh:7)
        learner = PpoLearner(pol, agent, ro, gamma=GAMMA, gae_lambda=LAM, use_linear_lr_decay=w.get("lr_decay", True), lr=lr, num_updates=1000)
        return pol, None, agent, ro, None, None, learner
    disc = sg.algo.gail.Discriminator(w["F"], w["Hd"], None, seed=seed)
    expert = rng.standard_normal((w["Ne"], w["F"])).astype(np.float32)   # identical on every rank (same seed)
    loader = ExpertLoader(expert, w["B"])
    disc._bind_loader(loader)          # expert matrix resident in HBM before anything is timed
    learner = GailDynLearner(pol, agent, disc, ro, loader, gail_batch_size=w["B"], gail_epoch=w["E_d"], gamma=GAMMA,
                             gae_lambda=LAM, gail_tar_length=500.0)
    return pol, disc, agent, ro, loader, expert, learner


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(w, expert, budget_s):
    """The update on the host's cores, timed on a bounded sample of the same workload: real discriminator steps at the gail
    batch size and real PPO steps on full-size minibatches; each leg's per-step times are multiplied by the update's step
    counts.  Two implementations, both this repository's own code (kind "port"), neither part of the product path:
      * `value`: oracle/sg_cpu_fast.c -- the whole minibatch through blocked, vectorised GEMMs (what a CPU implementation
        meant to be fast looks like; the reference's PyTorch-CPU path is of this class), built -march=native on this host and
        parity-checked against the oracle (tests/test_oracle_golden.py).  One thread, as the reference learner runs
        (torch.set_num_threads(1), a2c/main.py:65), plus a sweep of OpenMP team sizes.
      * `port_value`: oracle/sg_oracle.c -- the scalar, row-by-row parity oracle (rounds 1-3 reported this one; it is about
        half as fast as the reference's own path and flatters the GPU)."""
    from oracle import oracle as orc
    rng = np.random.default_rng(1)
    T, N, O, A, F, H, Hd, B = w["T"], w["N"], w["O"], w["A"], w["F"], w["H"], w["Hd"], w["B"]
    work = algorithmic_work(w, 1)
    cores = os.cpu_count() or 1
    kind = orc.KIND_MLP if w["kind"] == "mlp" else orc.KIND_SPLIT
    d = orc.dims(kind, O, A, H, w["feet"])
    mb = T * N // w["M"]
    obs = rng.standard_normal((mb, O)).astype(np.float32)
    act = rng.standard_normal((mb, A)).astype(np.float32)
    z = rng.standard_normal((4, mb)).astype(np.float32)
    cfg = orc.ppo_cfg(w["clip"], 1, 1, 0.5, w.get("ecoef", 0.0), w.get("lr", 3e-4), 1e-5, 0.5, True)
    rows = np.arange(mb)
    x = rng.standard_normal((2048, F)).astype(np.float32)
    if w["E_d"]:
        dpar = (rng.standard_normal(orc.disc_num_params(F, Hd)) * 0.1).astype(np.float32)
        e, p, al = expert[:B], rng.standard_normal((B, F)).astype(np.float32), rng.random(B).astype(np.float32)
    fast_build = orc.fast_lib(native=True)[1]     # "native": compiled for this host just now; "v3": the in-tree AVX2 build

    def leg(threads, budget, fast):
        t_d, n_dsteps = 0.0, 0
        if w["E_d"]:
            par, adam = dpar.copy(), orc.AdamState(dpar.size)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < budget * 0.55 or n_dsteps < 2:
                if fast:
                    G, _ = orc.disc_grad_rows_fast(F, Hd, par, e, p, al, 1.0 / B, native=True, n_threads=min(threads, B))
                else:
                    G, _ = (orc.disc_grad_rows(F, Hd, par, e, p, al, 1.0 / B) if threads == 1 else
                            orc.disc_grad_rows_mt(F, Hd, par, e, p, al, 1.0 / B, min(threads, B)))
                orc.adam_step(par, G, adam, 1e-3, 1e-8)
                n_dsteps += 1
            t_d = (time.perf_counter() - t0) / n_dsteps
        ppar = (rng.standard_normal(orc.policy_num_params(d)) * 0.1).astype(np.float32)
        padam = orc.AdamState(ppar.size)
        n_p, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget * (0.4 if w["E_d"] else 0.95) or n_p < 1:
            a_ = (d, ppar, cfg, obs, act, z[0], z[1], z[2] - 20.0, z[3], rows, 1.0 / mb)
            if fast:
                G, _ = orc.ppo_grad_rows_fast(*a_, native=True, n_threads=threads)
            else:
                G, _ = orc.ppo_grad_rows(*a_) if threads == 1 else orc.ppo_grad_rows_mt(*a_, threads)
            orc.ppo_apply(ppar, G, padam, cfg)
            n_p += 1
        t_p = (time.perf_counter() - t0) / n_p
        t_r = 0.0
        if w["E_d"]:   # relabel forward on a 2048-row sample (single-threaded, scalar, in every leg: < 1 % of the update)
            t0 = time.perf_counter()
            orc.disc_predict_reward(F, Hd, dpar, x, GAMMA, np.ones(2048, np.float32), 0.0)
            t_r = (time.perf_counter() - t0) * (T * N / 2048)
        est = work["d_steps"] * t_d + work["ppo_steps"] * t_p + t_r
        return dict(value=round(T * N / est, 1), cores=threads,
                    sample=(f"{n_dsteps} discriminator steps (batch {B}) + {n_p} PPO steps on {mb}-row minibatches"
                            f"{' + a 2048-row relabel forward' if w['E_d'] else ''}, scaled to {work['d_steps']} + {work['ppo_steps']} "
                            f"steps per update; {1e3 * t_d:.2f} ms/D-step, {1e3 * t_p:.1f} ms/PPO-step"))

    one = leg(1, budget_s * 0.35, True)
    port = leg(1, budget_s * 0.25, False)
    out = dict(value=one["value"], unit="env-steps/s", cores=1, kind="port", sample=one["sample"], cpu_model=_cpu_model(),
               implementation=f"oracle/sg_cpu_fast.c (batched, vectorised GEMMs; gcc -O3 -ffast-math, build: {fast_build}), 1 thread",
               port_value=port["value"], port_sample=port["sample"],
               port_implementation="oracle/sg_oracle.c (the scalar parity oracle), 1 thread",
               note=("both are this repository's C restatements of the update, timed on this host; the reference's own PyTorch-CPU path, "
                     "1 thread, measured 4,627 env-steps/s on the north-star shape in the development container "
                     "(BASELINE.md section 2) -- it cannot be re-timed here because the reference does not travel"),
               reference_pytorch_cpu_env_steps_s=4627.0)
    if cores > 1:
        # the rows of one step (128 row triples / 4096 rows) do not feed hundreds of threads: try a few team sizes up to
        # every host core and report the fastest beside the table of what was tried
        sizes = sorted({c for c in (8, 16, 32, 64, cores) if c <= cores})
        tried = [leg(c, budget_s * 0.4 / len(sizes), True) for c in sizes]
        best = max(tried, key=lambda r: r["value"])
        out["all_cores"] = dict(value=best["value"], unit="env-steps/s", cores=best["cores"], host_cores=cores, sample=best["sample"],
                                tried={str(r["cores"]): r["value"] for r in tried}, implementation="oracle/sg_cpu_fast.c, OpenMP team over the rows of a step")
    return out


PROF_SLOTS = ["disc_chain", "disc_wgrad", "ppo_fwd", "ppo_bwd", "ppo_reduce", "relabel_fwd", "ppo_adam", "disc_step", "comm_f32"]


def load_tdoc(workload):
    """The committed counter / kernel-trace summary of `workload` (tools/profile_workload.sh -> tools/make_traffic.py ->
    profiles/traffic.json), or {}."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f)["workloads"].get(workload, {})
    except (OSError, KeyError, ValueError):
        return {}


def load_cdoc(workload):
    """The committed SQ / TCC counter summary of `workload` (tools/profile_counters.sh -> tools/make_counters.py ->
    profiles/counters.json), or {}."""
    try:
        with open(os.path.join(ROOT, "profiles", "counters.json")) as f:
            return json.load(f)["workloads"].get(workload, {})
    except (OSError, KeyError, ValueError):
        return {}


def load_ctag():
    try:
        with open(os.path.join(ROOT, "profiles", "counters.json")) as f:
            return json.load(f).get("tag")
    except (OSError, ValueError):
        return None


def kernel_report(w, work, prof, tdoc, cdoc=None):
    """Per-kernel roofline entries from the HIP-event pass `prof` (slot -> (total ms, launches)) and the dominant kernel:
    the discriminator step -- one launch (k_disc_step4) where the library runs it so, else its chain kernel -- or, without
    a discriminator, the PPO backward."""
    def tk(kname, key):
        try:
            return tdoc["kernels"][kname][key]
        except KeyError:
            return None

    def ck(kname, key):
        try:
            return cdoc["kernels"][kname]["derived"][key]
        except (KeyError, TypeError):
            return None

    kmap = {}
    for slot, (kname, bound, kflops, kbytes) in work["kernels"].items():
        ms_, n_ = prof[slot]
        if not n_:
            continue
        avg_s = ms_ / n_ * 1e-3
        if bound == "mfma":
            ach, peak, unit = kflops / avg_s / 1e12, PEAK_F32_MFMA_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = kbytes / avg_s / 1e9, PEAK_HBM_GBS, "GB/s"
        pus = tk(kname, "rocprof_avg_us")
        kmap[kname] = {"bound": bound, "avg_us": round(avg_s * 1e6, 2), "launches": n_, "algorithmic_flops": kflops,
                       "algorithmic_bytes": kbytes, "achieved": round(ach, 3), "peak": peak, "unit": unit, "frac": round(ach / peak, 5),
                       "rocprof_avg_us": pus, "frac_profiled": round(ach / peak * (avg_s * 1e6) / pus, 5) if pus else None,
                       "traffic": tk(kname, "hbm_bytes_per_launch")}
        if bound == "mfma":
            kmap[kname].update({"mfma_busy": ck(kname, "mfma_busy"), "mfma_busy_resident": ck(kname, "mfma_busy_resident"),
                                "wave_wait": ck(kname, "wave_wait"), "l2_hit": ck(kname, "l2_hit")})
    one_launch = bool(w["E_d"]) and prof["disc_step"][1] > 0
    slot = ("disc_step" if one_launch else "disc_chain") if w["E_d"] else "ppo_bwd"
    kname = work["kernels"][slot][0]
    flops = (work["d_step_flops"] if one_launch else work["d_chain_flops"]) if w["E_d"] else work["ppo_step_flops"]
    ms_, n_ = prof[slot]
    avg_s = (ms_ / max(n_, 1)) * 1e-3
    achieved = flops / avg_s / 1e12 if avg_s > 0 else 0.0
    pus = tk(kname, "rocprof_avg_us")
    dom = {"kernel": kname, "flops": flops, "avg_s": avg_s, "launches": n_, "achieved": achieved,
           "frac": achieved / PEAK_F32_MFMA_TFLOPS, "rocprof_avg_us": pus,
           "frac_profiled": (flops / (pus * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS) if pus else None,
           "traffic": tk(kname, "hbm_bytes_per_launch"),
           "counters": {k: ck(kname, k) for k in ("mfma_busy", "mfma_busy_resident", "mfma_busy_grbm", "mfma_flops", "wave_wait", "wave_issue_stall",
                                                  "wave_active", "lds_conflict", "l2_hit")}}
    return kmap, dom


def spread_of(ms):
    ms = sorted(ms)
    return {"min_ms": round(ms[0], 3), "median_ms": round(ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2]), 3),
            "max_ms": round(ms[-1], 3), "n": len(ms), "clock": "one HIP event between consecutive updates on the library's stream (sg_ctx_mark), read after the loop"}


def run_updates(ctx, learner, steps):
    """`steps` updates back to back with a device timestamp between them -> (wall seconds incl. the final synchronise,
    per-update milliseconds on the device's clock, the last update's losses)."""
    ctx.marks_reset()
    marks = [ctx.mark()]
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = learner.update()
        marks.append(ctx.mark())
    ctx.synchronize()
    wall = time.perf_counter() - t0
    per = [ctx.mark_elapsed(marks[i], marks[i + 1]) for i in range(steps)]
    ctx.marks_reset()
    return wall, per, last


def profile_pass(ctx, learner):
    """Per-kernel durations with HIP events on the library's stream (separate, untimed update)."""
    ctx.profile_reset()
    ctx.profile(True)
    learner.update()
    ctx.profile(False)
    return {name: ctx.profile_read(i) for i, name in enumerate(PROF_SLOTS)}


def brief_workload(sg, _lib, ctx, name, steps=10, warmup=5):
    """One of BASELINE.json's other configurations, timed like the headline but briefly (device-resident rollout, `warmup`
    untimed + `steps` timed updates, one HIP-event pass) -- so that the driver's own bench run carries a number for every
    configuration BASELINE.json names, not only the one `value` is quoted on."""
    w = WORKLOADS[name]
    pol, disc, agent, ro, loader, expert, learner = build_problem(sg, w, seed=0)
    _lib.check(ctx.lib.sg_rollout_fill_synthetic(ro.h, pol.h, 1234, 0.01))
    for _ in range(warmup):
        learner.update()
    ctx.synchronize()
    wall, per, last = run_updates(ctx, learner, steps)
    prof = profile_pass(ctx, learner)
    work = algorithmic_work(w, 1)
    kmap, dom = kernel_report(w, work, prof, load_tdoc(name), load_cdoc(name))
    ms = 1e3 * wall / steps
    out = {"metric": "env-steps/sec of GAIL-dyn PPO update" if w["E_d"] else "env-steps/sec of PPO update (a2c/main.py)",
           "value": round(w["T"] * w["N"] * steps / wall, 1), "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup,
           "spread": spread_of(per), "optimizer_steps_per_update": work["d_steps"] + work["ppo_steps"],
           "us_per_optimizer_step": round(1e3 * ms / (work["d_steps"] + work["ppo_steps"]), 2),
           "shape": f"T={w['T']} N={w['N']} obs={w['O']} act={w['A']} D-in={w['F'] if w['E_d'] else '-'} policy={w['kind']} h{w['H']} num_mini_batch={w['M']}",
           "roofline": {"kernel": dom["kernel"], "bound": "mfma", "avg_launch_us": round(dom["avg_s"] * 1e6, 2), "launches": dom["launches"],
                        "achieved": round(dom["achieved"], 3), "unit": "TFLOP/s", "frac": round(dom["frac"], 5),
                        "frac_profiled": round(dom["frac_profiled"], 5) if dom["frac_profiled"] else None,
                        "mfma_busy": dom["counters"]["mfma_busy"]},
           "kernel_us": {k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in prof.items() if v[1]},
           "last_losses": dict(last)}
    del learner, agent, disc, ro, pol, loader, last
    import gc
    gc.collect()
    return out


def dropin_leg(sg, _lib, ctx, w, updates=5):
    """What a user of the reference's UNCHANGED main pays (never `value`): the rollout handed over as HOST tensors, every
    call of a2c/main_gail_dyn_ppo.py:255-304 made literally through the alias classes of third_party/a2c_ppo_acktr/ --
    gail_epoch x update_gail_dyn (losses read back), T x [predict_reward_combined + host ret_rms.update + clip],
    compute_returns, agent.update, after_update -- plus, beside it, the same update through GailDynLearner on host buffers
    (fused on-device relabel instead of the T-step loop) and the bare upload of one whole rollout."""
    import torch
    from third_party.a2c_ppo_acktr import algo                      # the names the main imports (:30-38)
    from third_party.a2c_ppo_acktr.storage import RolloutStorage    # noqa: F401  (build_problem constructs the same class)
    from simgan_amd.driver import alive_bonus_offset
    from simgan_amd.utils import RunningMeanStd
    assert algo.PPO is sg.algo.PPO and RolloutStorage is sg.RolloutStorage
    torch.set_num_threads(1)     # a2c/main_gail_dyn_ppo.py:64 (with a thread per core of a 100+-core host, torch's pool turns every
    T, N = w["T"], w["N"]        # small host-side tensor op of the sequence into a 10-50 ms stall: measured)
    res = {"updates_timed": updates, "note": "host tensors in, host tensors out; PCIe and every read-back included; not the metric's `value`"}

    def fresh():
        pol, disc, agent, ro, loader, expert, learner = build_problem(sg, w, seed=0)
        _lib.check(ctx.lib.sg_rollout_fill_synthetic(ro.h, pol.h, 1234, 0.01))
        ro.sync_from_device()            # the host tensors now hold the synthetic rollout
        ro.device_resident = False
        return pol, disc, agent, ro, loader, learner

    def refill(ro):                      # what the next rollout's insert() calls mean for the device copy: every field is new
        ro.mark_host_written()

    # (1) the literal main
    pol, disc, agent, ro, loader, learner = fresh()
    ret_rms = RunningMeanStd(shape=())

    split = {"d_epochs": 0.0, "relabel_loop": 0.0, "relabel_loop_library": 0.0, "returns_ppo_after": 0.0, "n": 0}
    real_prc = disc.predict_reward_combined

    def timed_prc(*a_, **k_):       # the library's share of the relabel loop (the rest of the loop is the main's own numpy / torch code)
        t_ = time.perf_counter()
        out_ = real_prc(*a_, **k_)
        split["relabel_loop_library"] += time.perf_counter() - t_
        return out_

    disc.predict_reward_combined = timed_prc

    def main_iteration():
        t_a = time.perf_counter()
        with torch.no_grad():
            next_value = pol.get_value(ro.obs[-1], ro.recurrent_hidden_states[-1], ro.masks[-1]).detach()
        for _ in range(w["E_d"]):
            gail_loss, gail_loss_e, gail_loss_p = disc.update_gail_dyn(loader, ro)
        num_of_dones = float((1.0 - ro.masks).sum().cpu().numpy())
        r_sa = alive_bonus_offset(num_of_dones, T, N, 500.0)
        t_b = time.perf_counter()
        for step in range(T):
            ro.rewards[step], returns = disc.predict_reward_combined(ro.obs_feat[step + 1], GAMMA, ro.masks[step], offset=-r_sa)
            ret_rms.update(returns.view(-1).cpu().numpy())
            rews = ro.rewards[step].view(-1).cpu().numpy()
            rews = np.clip(rews / np.sqrt(ret_rms.var + 1e-7), -10.0, 10.0)
            ro.rewards[step] = torch.Tensor(rews).view(-1, 1)
        t_c = time.perf_counter()
        ro.compute_returns(next_value, True, GAMMA, LAM, True)
        out = agent.update(ro)
        ro.after_update()
        t_d = time.perf_counter()
        split["d_epochs"] += t_b - t_a
        split["relabel_loop"] += t_c - t_b
        split["returns_ppo_after"] += t_d - t_c
        split["n"] += 1
        return out

    def timed_host(fn, prewarmed=False):
        if not prewarmed:
            fn()
            fn()
        ctx.synchronize()
        per, b0 = [], ro.bytes_uploaded
        for _ in range(updates):
            t0 = time.perf_counter()
            refill(ro)
            fn()
            ctx.synchronize()
            per.append(1e3 * (time.perf_counter() - t0))
        return sum(per) / len(per) * 1e-3, per, (ro.bytes_uploaded - b0) // updates

    def timed_host_main():
        main_iteration()
        main_iteration()
        for k_ in split:
            split[k_] = 0.0 if k_ != "n" else 0
        return timed_host(main_iteration, prewarmed=True)

    dt, per, up = timed_host_main()
    res["literal_main"] = {"ms_per_step": round(1e3 * dt, 3), "env_steps_s": round(T * N / dt, 1), "min_ms": round(min(per), 3), "max_ms": round(max(per), 3),
                           "bytes_uploaded_per_update": up,
                           # where the host's wall clock goes (ms per update): 5 synchronous D epochs (+ get_value, done count) | the T-step
                           # relabel loop, of which inside the library (ONE fused launch on step 0, T - 1 calls served from its result) --
                           # the remainder is the main's own per-step numpy / torch code | compute_returns + agent.update + after_update
                           "split_ms": {k_: round(1e3 * v_ / max(split["n"], 1), 3) for k_, v_ in split.items() if k_ != "n"},
                           "sequence": f"{w['E_d']} x update_gail_dyn, {T} x (predict_reward_combined + ret_rms.update + clip), compute_returns, agent.update, after_update"}
    # the bare upload of one rollout (every field once)
    ctx.synchronize()
    b1, t0 = ro.bytes_uploaded, time.perf_counter()
    for _ in range(3):
        ro.sync_to_device()
    ctx.synchronize()
    res["rollout_upload_ms"] = round(1e3 * (time.perf_counter() - t0) / 3, 3)
    res["rollout_bytes"] = (ro.bytes_uploaded - b1) // 3
    del learner, agent, disc, ro, pol, loader
    import gc
    gc.collect()
    # (2) GailDynLearner.update() on host buffers
    pol, disc, agent, ro, loader, learner = fresh()
    dt, per, up = timed_host(learner.update)
    res["learner_host_buffers"] = {"ms_per_step": round(1e3 * dt, 3), "env_steps_s": round(T * N / dt, 1), "min_ms": round(min(per), 3),
                                   "max_ms": round(max(per), 3), "bytes_uploaded_per_update": up}
    del learner, agent, disc, ro, pol, loader
    gc.collect()
    return res


def _sel(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(full, full_path=None):
    """The ONE stdout line: every key of the driver's contract plus the numbers a reader of a 2 KB tail needs -- the
    dominant kernel's roofline (both clocks, counter traffic, MFMA-busy from the SQ counters), the CPU baseline, and one
    {ms, env_steps_s, frac} triple per other BASELINE.json configuration and for the unchanged main's call sequence.
    Everything else (per-kernel tables, spreads, samples' descriptions, the comm legs' per-rank lists) is in the FULL record,
    written to `full_path` (named in the line) and to stderr."""
    c = _sel(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data", "error", "stage", "elapsed_s", "error_after_headline", "ranks", "us_per_optimizer_step"))
    if full.get("n_gpus", 1) > 1 and "per_rank_ms_per_step" in full:
        c["per_rank_ms_per_step"] = full["per_rank_ms_per_step"]
    c.setdefault("vs_baseline", None)
    c.setdefault("value", full.get("value"))
    cfg = dict(full.get("config", {}))
    if len(str(cfg.get("workload", ""))) > 120:       # the driver's record keeps 120 characters of a string
        cfg["workload"] = cfg["workload"][:117] + "..."
    c["config"] = cfg
    r = full.get("roofline")
    if r:
        c["roofline"] = _sel(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_profiled", "mfma_busy", "mfma_busy_resident_cus", "traffic",
                                 "avg_launch_us", "traffic_stale", "counters_stale"))
        c["roofline"]["traffic"] = r.get("traffic")
        c["roofline"]["lib_sha16"] = (r.get("lib_sha256") or "")[:16]
        c["roofline"]["update_frac_mfma"] = r.get("whole_update", {}).get("frac_mfma")
        c["roofline"]["update_frac_hbm"] = r.get("whole_update", {}).get("frac_hbm")
        kk = {name[2:]: [k.get("avg_us"), k.get("frac"), k.get("mfma_busy")] for name, k in (r.get("kernels") or {}).items()
              if name != r.get("kernel") and k.get("bound") == "mfma"}
        if kk:
            c["roofline"]["others_us_frac_busy"] = kk
    if "last_losses" in full and full.get("n_gpus", 1) > 1:
        c["last_losses"] = {k: (round(v, 6) if full.get("n_gpus", 1) > 1 else float(f"{v:.4g}")) for k, v in full["last_losses"].items()}
    ow = full.get("other_workloads")
    if ow:
        c["other_workloads_ms_envsteps_frac_busy"] = {n: (str(v["error"])[:80] if "error" in v else
                                                      [v.get("ms_per_step"), round(v.get("value", 0.0)), v.get("roofline", {}).get("frac"), v.get("roofline", {}).get("mfma_busy")]) for n, v in ow.items()}
        cfg["others_ms"] = " ".join(f"{n}={v.get('ms_per_step', 'err')}" for n, v in ow.items())
    dr = full.get("dropin")
    if dr:
        c["dropin_ms"] = _sel(dr, ("error",)) or {k: dr.get(k, {}).get("ms_per_step") for k in ("literal_main", "learner_host_buffers")}
        if "literal_main" in dr:
            cfg["literal_main_ms"] = dr["literal_main"].get("ms_per_step")
    cb = full.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = _sel(cb, ("value", "unit", "cores", "kind", "port_value", "host_cores_available"))
        c["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:72]
        if isinstance(cb.get("all_cores"), dict):
            c["cpu_baseline"]["all_cores_value"] = cb["all_cores"].get("value")
            c["cpu_baseline"]["all_cores_threads"] = cb["all_cores"].get("cores")
    cm = full.get("comm")
    if cm:
        c["comm"] = _sel(cm, ("kind", "nranks_reported_by_rccl", "disc_mode", "peer_allreduce", "allreduce_us", "allreduce_count_per_update", "allreduce_form"))
        for leg in ("disc_other_mode", "peer", "base"):
            if leg in cm:
                c["comm"][leg] = _sel(cm[leg], ("mode", "form", "value", "ms_per_step", "allreduce_us", "error", "restore_error"))
    if "replica_check" in full:
        c["replica_check"] = full["replica_check"]
    if full_path:
        c["full_record"] = full_path
    return c


def write_full(full, workload, world):
    """The full record beside the compact line: gpurun_out/bench_full_<workload>_n<N>.json under the repository (merged back
    by gpurun), else the temp directory.  Returns the path written (None when nothing could be written)."""
    import tempfile
    for d in (os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out"), tempfile.gettempdir()):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, f"bench_full_{workload}_n{world}.json")
            with open(path, "w") as f:
                json.dump(full, f)
            return os.path.relpath(path, os.path.dirname(os.path.abspath(__file__))) if d.endswith("gpurun_out") else path
        except OSError:
            continue
    return None


KEEPER_CODE = r"""
import json, os, sys
fd, last, final = int(sys.argv[1]), None, False
for line in sys.stdin:                      # until the bench process closes the pipe -- or dies
    line = line.rstrip("\n")
    if line[:2] == "F ":
        last, final = line[2:], True
    elif line[:2] == "S ":
        last = line[2:]
if last is not None:
    if not final:
        try:
            d = json.loads(last)
            d["error_after_headline"] = ("the bench process ended before it printed its line (killed or crashed in a leg that runs after the "
                                         "headline measurement); this is the line as it stood")
            last = json.dumps(d)
        except Exception:
            pass
    os.write(fd, (last + "\n").encode())
"""


class LineKeeper(object):
    """From the moment the headline measurement is complete, the ONE result line is printed by a small helper process that
    holds the latest complete version of it: rank 0 hands it every update of the line and, at the end, the final one.  If the
    bench process is killed or aborts in a later leg (a GPU fault in the first multi-GPU run of the peer mesh, an
    out-of-memory kill during the CPU baseline), the helper sees the pipe close and prints the line as it stood, marked
    `error_after_headline` -- the measured headline is never lost with the process.  The helper sits in a session of its own
    (a launcher that kills the ranks' process group does not take it along) and never touches the GPU."""

    def __init__(self, result_fd):
        self.p = subprocess.Popen([sys.executable, "-c", KEEPER_CODE, str(result_fd)], stdin=subprocess.PIPE, pass_fds=(result_fd,),
                                  start_new_session=True, text=True)

    def store(self, obj):
        try:
            self.p.stdin.write("S " + json.dumps(obj) + "\n")
            self.p.stdin.flush()
        except (OSError, ValueError):
            pass

    def finish(self, obj):
        """True when the helper printed the final line."""
        try:
            self.p.stdin.write("F " + json.dumps(obj) + "\n")
            self.p.stdin.close()
            return self.p.wait(timeout=20) == 0
        except (OSError, ValueError, subprocess.TimeoutExpired):
            return False


class Watchdog(object):
    """First-run insurance for the N-GPU launch: whatever happens -- a hung ncclCommInitRank, a first collective that never
    completes, a peer that dies and takes the launcher's SIGTERM with it -- rank 0 still prints ONE JSON line, carrying
    `error`, the `stage` it happened in and every rank's last reported stage, and the process exits non-zero well inside the
    driver's own timeout.  Every rank records its stage in a small file under /tmp (no collective needed to read them).
    The thread never touches HIP or torch; ctypes and gloo release the GIL while they block, so it runs."""

    def __init__(self, rank, world, emit, base_line):
        import signal
        import tempfile
        import threading
        self.rank, self.world, self.emit, self.base = rank, world, emit, base_line
        self.dir = os.path.join(tempfile.gettempdir(), f"sg_bench_status_{os.environ.get('MASTER_PORT', 'single')}_{os.getppid() if world > 1 else os.getpid()}")
        os.makedirs(self.dir, exist_ok=True)
        self.lock = threading.Lock()
        self.name, self.deadline, self.done = "start", time.time() + 600.0, False
        self.final = None        # set once the headline line is complete (legs reported beside it may still fail)
        self.t0 = time.time()
        self._write()
        # a SIGTERM (torch.distributed.run ends the surviving ranks when one rank fails) must not lose the line either: the
        # C-level handler writes the signal number into this pipe at once, whatever the main thread is blocked in
        self.r_fd, w_fd = os.pipe()
        os.set_blocking(w_fd, False)
        try:
            signal.signal(signal.SIGTERM, lambda *_: None)
            signal.set_wakeup_fd(w_fd, warn_on_full_buffer=False)
        except ValueError:      # not the main thread (bench.main() called from a test harness thread)
            pass
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _write(self):
        try:
            with open(os.path.join(self.dir, f"rank{self.rank}.json"), "w") as f:
                json.dump({"rank": self.rank, "stage": self.name, "since_start_s": round(time.time() - self.t0, 1), "pid": os.getpid()}, f)
        except OSError:
            pass

    def stage(self, name, budget_s):
        with self.lock:
            self.name, self.deadline = name, time.time() + budget_s
        self._write()
        sys.stderr.write(f"[bench] rank {self.rank}: {name}\n")

    def finish(self):
        self.done = True
        self.stage("done", 1e9)

    def ranks(self):
        out = {}
        for r in range(self.world):
            try:
                with open(os.path.join(self.dir, f"rank{r}.json")) as f:
                    d = json.load(f)
                out[str(r)] = f"{d['stage']} (reported {d['since_start_s']} s after its start)"
            except (OSError, ValueError, KeyError):
                out[str(r)] = "no status file: the rank never got as far as creating its watchdog"
        return out

    def fail(self, why, code=3):
        if self.done:
            return
        self.done = True
        with self.lock:
            stage = self.name
        self._write()
        sys.stderr.write(f"[bench] rank {self.rank}: {why} in stage '{stage}'\n")
        if self.rank == 0 and self.final is not None:
            # the headline measurement is complete and only a leg reported beside it went wrong: the line goes out as it stands
            line = dict(self.final)
            line.update({"error_after_headline": why, "stage": stage, "elapsed_s": round(time.time() - self.t0, 1)})
            self.emit(line)
            os._exit(0)
        if self.rank == 0:
            time.sleep(1.0)      # let the other ranks' watchdogs record their final stage
            line = dict(self.base)
            line.update({"value": None, "ms_per_step": None, "error": why, "stage": stage, "ranks": self.ranks(),
                         "elapsed_s": round(time.time() - self.t0, 1)})
            self.emit(line)
        else:
            time.sleep(4.0)      # rank 0 prints first; then this rank's exit lets the launcher end the run
        os._exit(code)

    def _run(self):
        import select
        while not self.done:
            with self.lock:
                left = self.deadline - time.time()
            if left <= 0:
                self.fail(f"watchdog: no progress within the stage's budget ({self.name})")
            ready, _, _ = select.select([self.r_fd], [], [], min(max(left, 0.05), 2.0))
            if ready:
                sig = os.read(self.r_fd, 16)
                if 15 in sig:    # SIGTERM
                    self.fail("SIGTERM from the launcher (a peer rank failed or the run was cancelled)", code=143)


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: run the N ranks under torch.distributed.run (what the
    driver's own command line does), hand rank 0's JSON line through, fail if any rank fails or no line appears."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in lines[-1:]:
        sys.stdout.write(ln + "\n")
    sys.stdout.flush()
    if proc.returncode != 0:
        raise SystemExit(proc.returncode)
    if not lines:
        raise SystemExit("bench.py: the ranks exited cleanly but rank 0 printed no result line")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="northstar", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the brief runs of BASELINE.json's other configurations")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in legs (the unchanged main's call sequence on host tensors)")
    ap.add_argument("--no-line-keeper", action="store_true", help="print the line from this process only (no helper process holding it after the headline)")
    ap.add_argument("--full-line", action="store_true", help="print the FULL record on stdout instead of the compact line (the default writes it to gpurun_out/bench_full_<workload>_n<N>.json and stderr)")
    ap.add_argument("--headline-only", action="store_true", help="only the headline measurement (+ cpu baseline unless --no-cpu-baseline)")
    ap.add_argument("--no-other-disc-mode", action="store_true", help="N > 1: skip timing the non-default discriminator mode")
    ap.add_argument("--no-other-allreduce", action="store_true", help="N > 1: skip timing the other form of the per-step all-reduce (peer mesh / base communicator)")
    ap.add_argument("--init-timeout", type=float, default=480.0, help="watchdog budget (s) for start-up: imports, context, communicator, problem build")
    ap.add_argument("--stage-timeout", type=float, default=420.0, help="watchdog budget (s) for each later stage (warm-up, timed region, profile pass, ...)")
    ap.add_argument("--no-dp-check", action="store_true", help="N > 1: skip the replica-consistency check after the warm-up")
    ap.add_argument("--loopback", action="store_true",
                    help="self-test: the N ranks share the visible device(s) and use the shared-memory loopback communicator")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)

    # Exactly ONE line on stdout: RCCL (and anything else underneath) writes banners to the C-level stdout, which would land
    # before or after the JSON line depending on buffering.  Everything written to fd 1 from here on goes to stderr; the
    # result line is written to a private duplicate of the original stdout.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    keeper = {"k": None}

    def shape(obj):
        """stdout gets the compact line (--full-line: the whole record); the whole record goes to a file and to stderr."""
        if args.full_line:
            return obj
        path = write_full(obj, args.workload, int(os.environ.get("WORLD_SIZE", "1")))
        return compact_line(obj, path)

    def emit(obj):
        k, keeper["k"] = keeper["k"], None
        line = shape(obj)
        if not args.full_line:
            sys.stderr.write("[bench] full record: " + json.dumps(obj) + "\n")
        if k is not None and k.finish(line):
            return
        os.write(result_fd, (json.dumps(line) + "\n").encode())

    w = WORKLOADS[args.workload]
    rank_env, world_env = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    base_line = {"metric": "env-steps/sec of GAIL-dyn PPO update" if w["E_d"] else "env-steps/sec of PPO update (a2c/main.py)",
                 "value": None, "unit": "env-steps/s", "n_gpus": world_env, "steps": args.steps, "warmup": args.warmup,
                 "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                 "config": {"workload": args.workload}}
    dog = Watchdog(rank_env, world_env, emit, base_line)
    dog.stage("init: process group (gloo)", args.init_timeout)
    if world_env > 1:
        os.environ.setdefault("NCCL_DEBUG", "WARN")   # RCCL's warnings go to (what is now) stderr; INFO would flood it
    if os.environ.get("SG_BENCH_HANG_RANK") == str(rank_env):   # test hook: a rank that never arrives (tests/test_gpu_world.py)
        dog.stage("TEST HOOK: this rank sleeps instead of joining (SG_BENCH_HANG_RANK)", 1e9)
        time.sleep(1e6)
    from simgan_amd.dist import ProcessGroup
    pg = ProcessGroup()
    rank, world = pg.rank, pg.world
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run "
                         "(one process per GPU)")

    dog.stage("init: library context", args.init_timeout)
    import simgan_amd as sg
    from simgan_amd import _lib
    if args.loopback:   # ranks share the visible device(s): LOCAL_RANK modulo their number (SG_LOOPBACK_DEVICES, default 1)
        ctx = _lib.Context(pg.local_rank % max(1, int(os.environ.get("SG_LOOPBACK_DEVICES", "1")))).make_default()
    else:
        ctx = _lib.Context.default()   # device = LOCAL_RANK
    lib = ctx.lib
    dog.stage("init: communicator (ncclCommInitRank + first all-reduce)" if world > 1 else "init: build problem", args.init_timeout)
    pg.init_device_comm(ctx, _lib.comm_loopback_id if args.loopback else _lib.comm_unique_id)
    if world == 1 and os.environ.get("SG_COMM_ALWAYS") == "1":   # one-rank communicator: the collectives run as identities
        ctx.comm_init(_lib.comm_unique_id(), 0, 1)

    dbg = (lambda tag: sys.stderr.write(f"[bench dbg] rank {rank} {tag}: comm {ctx.comm_kind()} {ctx.comm_info()}\n")) if os.environ.get("SG_BENCH_DEBUG") else (lambda tag: None)
    dbg("after comm init")
    if os.environ.get("SG_BENCH_FAIL_RANK") == str(rank):   # test hook: a rank that dies (tests/test_gpu_world.py)
        raise SystemExit(f"rank {rank}: SG_BENCH_FAIL_RANK set")
    dog.stage("init: build problem", args.init_timeout)
    pol, disc, agent, ro, loader, expert, learner = build_problem(sg, w, seed=0)
    if disc is not None:   # the discriminator's draw streams are global: every rank must hold the same seed
        seeds = pg.gather_object(int(disc.seed))
        if len(set(seeds)) != 1:
            raise SystemExit(f"discriminator seeds differ across ranks ({seeds}): expert permutation / alpha would diverge")
    dbg("after build_problem")
    assert pol.ctx is ctx and ro.ctx is ctx and agent.ctx is ctx, "the bench objects must live on the context that holds the communicator"
    _lib.check(lib.sg_rollout_fill_synthetic(ro.h, pol.h, 1234, 0.01))

    def barrier():
        ctx.synchronize()
        pg.barrier()

    def replica_digest():
        import hashlib
        h = hashlib.sha256(pol.get_flat_params().tobytes())
        if disc is not None:
            h.update(disc.get_flat_params().tobytes())
        return h.hexdigest()

    replica = {"checked": False}

    def timed(steps, warmup, label=""):
        dog.stage(f"{label}warm-up ({warmup} updates; graph capture, first collectives)", args.stage_timeout)
        for _ in range(warmup):
            learner.update()
        barrier()
        if world > 1 and not args.no_dp_check and not replica["checked"]:
            # data parallelism replicates pi and D: after the warm-up's optimizer steps every rank must hold the SAME bits
            dog.stage("replica check (hash of pi and D weights on every rank)", args.stage_timeout)
            digests = pg.gather_object(replica_digest())
            replica.update(checked=True, digests=digests, ok=len(set(digests)) == 1)
            if not replica["ok"]:
                dog.fail(f"replicas diverged after the warm-up: weight hashes per rank {[d[:12] for d in digests]}", code=4)
        dog.stage(f"{label}timed region ({steps} updates)", args.stage_timeout)
        mine, per_update, last_ = run_updates(ctx, learner, steps)
        barrier()
        spreads.append(per_update)
        return pg.max(mine), pg.gather(mine), last_

    spreads = []
    elapsed, per_rank, last = timed(args.steps, args.warmup)
    # self-test of the N > 1 reporting path on one GPU (with SG_COMM_ALWAYS=1 the collectives stay in the launch sequence)
    force_alt = world == 1 and os.environ.get("SG_BENCH_FORCE_ALT") == "1" and ctx.comm_info()[1] == 1 and os.environ.get("SG_COMM_ALWAYS") == "1"
    # per-kernel durations with HIP events on the library's stream (separate, untimed pass)
    dog.stage("per-kernel HIP-event pass", args.stage_timeout)
    prof = profile_pass(ctx, learner)

    if rank == 0:
        work = algorithmic_work(w, world, ctx.disc_sharded)
        env_steps = w["T"] * w["N"] * world
        ms_per_step = 1e3 * elapsed / args.steps
        value = env_steps * args.steps / elapsed
        name, num_cu, hbm = ctx.device_info()
        # PMC passes and rocprofv3's kernel trace cannot run inside the timed process: counter traffic and the profiler's
        # own per-kernel durations come from the committed summary of the last profile run of THIS workload
        # (tools/profile_workload.sh -> tools/make_traffic.py -> profiles/traffic.json), which records the sha256 of the
        # library it profiled; a summary taken with another build is flagged, not silently reused.
        import hashlib
        with open(_lib.LIB_PATH, "rb") as f:
            lib_sha = hashlib.sha256(f.read()).hexdigest()
        tdoc = load_tdoc(args.workload)
        traffic_sha = tdoc.get("lib_sha256")
        traffic_stale = bool(tdoc) and traffic_sha != lib_sha
        cdoc = load_cdoc(args.workload)
        kmap, dom = kernel_report(w, work, prof, tdoc, cdoc)
        dom_kernel, dom_flops, dg_avg_s, dg_n, achieved, traffic = dom["kernel"], dom["flops"], dom["avg_s"], dom["launches"], dom["achieved"], dom["traffic"]

        def profiled_us(kname):
            try:
                return tdoc["kernels"][kname]["rocprof_avg_us"]
            except KeyError:
                return None

        out = {
            "metric": "env-steps/sec of GAIL-dyn PPO update" if w["E_d"] else ("env-steps/sec of PPO update (policy refinement, a2c/main.py)" if args.workload == "refine" else "env-steps/sec of PPO update (a2c/main.py)"), "value": round(value, 1), "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "spread": spread_of(spreads[0]),     # this rank's device clock, update by update, over the timed region
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.workload}: GAIL-dyn update, T={w['T']} N={w['N']}/GPU obs={w['O']} act={w['A']} "
                                    f"D-in={w['F']} policy={w['kind']} h{w['H']} D h{w['Hd']} ppo_epoch={w['E_p']} "
                                    f"num_mini_batch={w['M']} gail_epoch={w['E_d']} gail_batch={w['B']} expert_rows={w['Ne']}") if w["E_d"] else
                                   (f"{args.workload}: PPO update (a2c/main.py caller), T={w['T']} N={w['N']}/GPU obs={w['O']} act={w['A']} "
                                    f"policy={w['kind']} h{w['H']} ppo_epoch={w['E_p']} num_mini_batch={w['M']} clip={w['clip']} "
                                    f"entropy_coef={w.get('ecoef', 0.0)} lr={w['lr']}{' linear decay' if w.get('lr_decay', True) else ''}"),
                       "optimizer_steps_per_update": work["d_steps"] + work["ppo_steps"],
                       "parallelism": f"dp{world} (env columns sharded, RCCL grad all-reduce)" if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "kernel": dom_kernel, "achieved": round(achieved, 3),
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 5),
                         # two clocks for the same launches: HIP events around every launch in THIS process (graphs bypassed while
                         # they are recorded) and rocprofv3's kernel trace of the committed profile run, whose interception sits on
                         # every dispatch and reads 10-20 % longer on 4-7 us kernels.  `frac` is the HIP-event one (measured live, as
                         # the contract asks); `frac_profiled` is what a reader recomputes from profiles/.
                         "frac_basis": "hip_events", "frac_events": round(achieved / PEAK_F32_MFMA_TFLOPS, 5),
                         "frac_profiled": (round(dom_flops / (profiled_us(dom_kernel) * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 5)
                                           if profiled_us(dom_kernel) else None),
                         "rocprof_avg_launch_us": profiled_us(dom_kernel),
                         "rocprof_source": tdoc.get("kernel_trace", "no kernel trace of this workload committed"),
                         "lib_sha256": lib_sha, "traffic_lib_sha256": traffic_sha, "traffic_stale": traffic_stale,
                         # SQ counters of the committed counter passes (tools/profile_counters.sh -> profiles/counters.json): the matrix
                         # pipes' busy cycles over (the cycles the kernel keeps a shader engine busy x the chip's 1024 SIMDs), and the
                         # same over the CU-cycles in which a CU holds a wave of the kernel
                         "mfma_busy": dom["counters"]["mfma_busy"], "mfma_busy_resident_cus": dom["counters"]["mfma_busy_resident"],
                         "counters": dom["counters"], "counters_lib_sha256": cdoc.get("lib_sha256"),
                         "counters_stale": bool(cdoc) and cdoc.get("lib_sha256") != lib_sha,
                         "counters_source": f"profiles/counters.json (tag {load_ctag()}, passes {cdoc.get('passes')})" if cdoc else "no counter passes of this workload committed",
                         "traffic": traffic, "traffic_unit": "HBM-side bytes per launch: rocprofv3 PMC FETCH_SIZE and WRITE_SIZE (separate passes) with the per-access-width calibration named in profiles/traffic.json; " + str(tdoc.get("source", "no profile of this workload committed")),
                         "avg_launch_us": round(dg_avg_s * 1e6, 2), "launches": dg_n,
                         "kernels": kmap,
                         "algorithmic_flops_per_launch": dom_flops,
                         "note": ("serial chain of 7 dependent GEMM phases on 96 four-row workgroups per 128-row step: latency-bound, see DESIGN.md section 4"
                                  if w["E_d"] else "fused forward + loss + backward + weight gradients of one minibatch, one row group per workgroup"),
                         "whole_update": {"TFLOP/s": round(work["flops"] * args.steps / elapsed / 1e12, 3),
                                          "GB/s": round(work["bytes"] * args.steps / elapsed / 1e9, 2),
                                          "frac_mfma": round(work["flops"] * args.steps / elapsed / 1e12 / PEAK_F32_MFMA_TFLOPS, 5),
                                          "frac_hbm": round(work["bytes"] * args.steps / elapsed / 1e9 / PEAK_HBM_GBS, 6)}},
            "kernel_us": {k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in prof.items()},
            "kernel_launches": {k: v[1] for k, v in prof.items()},
            "us_per_optimizer_step": round(1e3 * ms_per_step / (work["d_steps"] + work["ppo_steps"]), 2),
            "last_losses": dict(last),
            "device": name, "num_cu": num_cu,
            "per_rank_ms_per_step": [round(1e3 * x / args.steps, 3) for x in per_rank],
        }
        if world > 1 or force_alt:
            kind = ctx.comm_kind()
            out["comm"] = {"backend": ("RCCL (dlopen) on the library stream, captured into the update's hipGraphs" if kind == "rccl" else
                                       "LOOPBACK self-test (shared-memory transport, ranks share a device): exercises the N > 1 code "
                                       "path, NOT a scaling measurement"),
                           "kind": kind, "nranks_reported_by_rccl": ctx.comm_info()[1], "rank0_reported_by_rccl": ctx.comm_info()[0],
                           "disc_mode": "sharded" if ctx.disc_sharded else "replicated",
                           "peer_allreduce": ctx.comm_peer(),   # SG_COMM_PEER=1: the per-step gradient all-reduce as one peer-write kernel
                           # the per-step float32 gradient all-reduce of the headline run, bracketed by HIP events in the separate
                           # profile pass (device time from reaching the collective to leaving it: the wait for the slowest rank included)
                           "allreduce_us": round(1e3 * prof["comm_f32"][0] / max(prof["comm_f32"][1], 1), 2),
                           "allreduce_count_per_update": prof["comm_f32"][1],
                           "allreduce_form": "peer mesh" if ctx.comm_peer() else kind}
    else:
        out = None
    # N > 1: the discriminator has two data-parallel modes.  `value` is the default (replicated:
    # every rank runs the full-batch steps on the union of the ranks' rows, no per-step collective); the sharded mode
    # (batch/world rows per rank, one gradient all-reduce per step) is timed as well and reported beside it.  The headline
    # line is complete at this point: a watchdog on every rank prints it and ends the run if the extra measurement stalls.
    if world > 1 and rank == 0:
        out["replica_check"] = ({"ok": replica.get("ok"), "weights_sha256_rank0": replica["digests"][0][:16], "ranks": world}
                                if replica["checked"] else "skipped (--no-dp-check)")
    # The headline is measured.  Everything below is reported beside it; from here on the line is in the keeping of a helper
    # process (LineKeeper), so that whatever happens to THIS process in a later leg, the line still comes out.
    def keep():
        if keeper["k"] is not None:
            keeper["k"].store(shape(out))

    if rank == 0:
        dog.final = out      # a watchdog time-out or the launcher's SIGTERM from here on prints THIS line (with what went wrong), not an empty one
    if rank == 0 and not args.no_line_keeper:
        try:
            keeper["k"] = LineKeeper(result_fd)
            keep()
        except OSError:
            keeper["k"] = None
    if os.environ.get("SG_BENCH_DIE_AFTER_HEADLINE") == str(rank):   # test hook: the process is killed outright in a later leg
        import signal
        time.sleep(0.5)
        os.kill(os.getpid(), signal.SIGKILL)
    dog.stage("other discriminator mode / cpu baseline / output", 1e9)   # the other-mode timing has its own watchdog below
    if (world > 1 or force_alt) and w["E_d"] and not args.no_other_disc_mode:
        import threading
        k2 = max(2, args.steps // 2)
        other = "replicated" if ctx.disc_sharded else "sharded"
        budget = max(90.0, 10.0 * elapsed * (k2 + 2) / args.steps)

        def give_up():
            if rank == 0:
                out["comm"]["disc_other_mode"] = dict(mode=other, error=f"no result within {budget:.0f} s; measurement abandoned")
                emit(out)
            os._exit(0)

        dog2 = threading.Timer(budget, give_up)
        dog2.daemon = True
        dog2.start()
        try:
            ctx.set_disc_dp(not ctx.disc_sharded)
            e2, pr2, _ = timed(k2, 2, label=f"[{other} D mode] ")
            dog.stage("other discriminator mode done / cpu baseline / output", 1e9)
            alt = dict(mode=other, steps=k2, ms_per_step=round(1e3 * e2 / k2, 3),
                       value=round(w["T"] * w["N"] * world * k2 / e2, 1), per_rank_ms_per_step=[round(1e3 * x / k2, 3) for x in pr2])
        except Exception as exc:   # the headline measurement above is already complete: report, do not lose the line
            alt = dict(mode=other, error=str(exc)[:300])
        dog2.cancel()
        ctx.set_disc_dp(not ctx.disc_sharded)
        if rank == 0:
            out["comm"]["disc_other_mode"] = alt
            keep()
    # N > 1: the OTHER form of the per-step all-reduce -- the peer mesh when the headline ran on the base communicator (RCCL),
    # the base communicator when SG_COMM_PEER=1 -- on the same communicator, toggled collectively (sg_ctx_comm_set_peer), so
    # that the first multi-GPU run comes back with both timed.  Own watchdog; a failure leaves an `error` entry.
    if (world > 1 or force_alt) and not args.no_other_allreduce:
        import threading
        k2 = max(2, args.steps // 2)
        was_peer = ctx.comm_peer()
        form = "base communicator" if was_peer else "peer mesh"
        field = "base" if was_peer else "peer"
        budget = max(120.0, 10.0 * elapsed * (k2 + 3) / args.steps)

        def give_up2():
            if rank == 0:
                out["comm"][field] = dict(form=form, error=f"no result within {budget:.0f} s; measurement abandoned")
                emit(out)
            os._exit(0)

        dog3 = threading.Timer(budget, give_up2)
        dog3.daemon = True
        dog3.start()
        try:
            dog.stage(f"[{form}] switching the all-reduce form (collective set-up)", 1e9)
            ctx.comm_set_peer(not was_peer)
            e3, pr3, _ = timed(k2, 2, label=f"[{form}] ")
            dog.stage(f"[{form}] per-kernel HIP-event pass", 1e9)
            prof3 = profile_pass(ctx, learner)
            leg = dict(form=form, steps=k2, ms_per_step=round(1e3 * e3 / k2, 3), value=round(w["T"] * w["N"] * world * k2 / e3, 1),
                       allreduce_us=round(1e3 * prof3["comm_f32"][0] / max(prof3["comm_f32"][1], 1), 2),
                       per_rank_ms_per_step=[round(1e3 * x / k2, 3) for x in pr3])
        except Exception as exc:   # every rank gets the set-up's error (it is collective): the headline stands
            leg = dict(form=form, error=str(exc)[:400])
        dog3.cancel()
        try:
            if ctx.comm_peer() != was_peer:
                ctx.comm_set_peer(was_peer)
        except Exception as exc:
            leg["restore_error"] = str(exc)[:200]
        dog.stage("all-reduce forms done / cpu baseline / output", 1e9)
        if rank == 0:
            out["comm"][field] = leg
            keep()
    if world == 1 and not force_alt and not args.headline_only:
        # Everything below is reported BESIDE `value`, never as it; the headline is complete.  A leg that fails or runs out
        # of its budget leaves an `error` entry: the line is not lost (the stage watchdog prints what it has and ends the run).
        import gc
        del learner, agent, disc, ro, pol, loader
        gc.collect()    # the handles are freed HERE (hipFree waits for the device), not by a collection inside a later timed loop
        if args.workload == "northstar" and not args.no_other_workloads:
            out["other_workloads"] = {}
            for other_name in ("hopper", "laikago", "refine", "hopper_ppo"):   # BASELINE.json configs[1], [2], [4], [0]
                dog.stage(f"other workload: {other_name}", 120.0)
                try:
                    out["other_workloads"][other_name] = brief_workload(sg, _lib, ctx, other_name)
                except Exception as exc:
                    out["other_workloads"][other_name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
                keep()
        if w["E_d"] and not args.no_dropin:
            dog.stage("drop-in legs (host tensors through the alias classes)", 180.0)
            try:
                out["dropin"] = dropin_leg(sg, _lib, ctx, w)
            except Exception as exc:
                out["dropin"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            keep()
        dog.stage("cpu baseline / output", 1e9)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w, expert, args.cpu_seconds)
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        emit(out)
    dog.finish()
    pg.shutdown()


if __name__ == "__main__":
    main()
