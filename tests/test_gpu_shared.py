"""Two PROCESSES training on one GPU at the same time (a shared box, a profiler's second process): each sees the other's
per-device advisory lock (csrc/sg_ctx.cpp), leaves the launches that wait inside themselves (k_disc_step4, k_ppo_pair) for
their multi-launch forms without any environment variable, nobody times out, and -- the forms being bit-identical --
each ends with exactly the weights of a solo run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
import simgan_amd as sg
from simgan_amd import _lib
tag, n_peers, rdv, updates = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
w = dict(bench.WORKLOADS["hopper"], E_d=2)
ctx = _lib.Context.default()
pol, disc, agent, ro, loader, expert, learner = bench.build_problem(sg, w, seed=0)
_lib.check(ctx.lib.sg_rollout_fill_synthetic(ro.h, pol.h, 1234, 0.01))
slots = ["disc_chain", "disc_wgrad", "ppo_fwd", "ppo_bwd", "ppo_reduce", "relabel_fwd", "ppo_adam", "disc_step"]
def launches():
    ctx.profile_reset(); ctx.profile(True); out = learner.update().resolve(); ctx.profile(False)
    return {n: ctx.profile_read(i)[1] for i, n in enumerate(slots)}, out
before, _ = launches()                     # (possibly alone on the device at this point)
open(os.path.join(rdv, f"ready_{tag}"), "w").close()
t0 = time.time()
while len([f for f in os.listdir(rdv) if f.startswith("ready_")]) < n_peers:
    if time.time() - t0 > 240: raise SystemExit("peer never became ready")
    time.sleep(0.01)
losses = [dict(learner.update().resolve()) for _ in range(updates)]     # every update read: a time-out would raise here
during, last = launches()
ctx.synchronize()
h = hashlib.sha256(pol.get_flat_params().tobytes() + disc.get_flat_params().tobytes()).hexdigest()
open(os.path.join(rdv, f"done_{tag}"), "w").close()
print("RESULT " + json.dumps({"tag": tag, "sha": h, "before": before, "during": during, "finite": bool(all(np.isfinite(v) for l in losses for v in l.values())),
                              "last": {k: float(v) for k, v in last.items()}}))
"""


def _run(tags, rdv, updates=40):
    # a lock directory of their own: the processes of this test see EACH OTHER, not the pytest process (which may still hold
    # learner objects of earlier tests, and with them its own lock on the device)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SG_LOCK_DIR=rdv)
    env.pop("SG_DISC_FUSED", None)
    env.pop("SG_PPO_PAIR", None)
    procs = [subprocess.Popen([sys.executable, "-c", CHILD, t, str(len(tags)), rdv, str(updates)], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for t in tags]
    out = {}
    for t, p in zip(tags, procs):
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, f"process {t} failed:\n{se[-3000:]}"
        line = [ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1]
        out[t] = json.loads(line[7:])
    return out


def test_two_processes_share_one_gpu_without_time_outs(tmp_path):
    solo_dir, pair_dir = tmp_path / "solo", tmp_path / "pair"
    solo_dir.mkdir()
    pair_dir.mkdir()
    solo = _run(["s"], str(solo_dir))["s"]
    assert solo["finite"]
    if not solo["during"]["disc_step"]:
        pytest.skip("the solo run does not use the one-launch discriminator step here (device already shared?)")
    pair = _run(["a", "b"], str(pair_dir))
    for t in ("a", "b"):
        r = pair[t]
        assert r["finite"], r
        assert r["sha"] == solo["sha"], f"process {t} ended with other weights than the solo run"
    # while both were training, at least the later-finishing one ran whole updates beside the other: the measurement pass right
    # after the timed updates must have seen the multi-launch forms in at least one of them, and never a time-out in either
    assert any(pair[t]["during"]["disc_step"] == 0 and pair[t]["during"]["disc_chain"] > 0 for t in ("a", "b")), pair
