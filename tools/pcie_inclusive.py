"""The update with the rollout handed over as HOST buffers (drop-in mode: every device call uploads the fields it reads and
downloads what it wrote, simgan_amd/storage.py) beside the device-resident update bench.py times -- the PCIe-inclusive rate
DESIGN.md section 5 quotes (never bench.py's `value`).  GPU box:  python tools/pcie_inclusive.py [workload] [updates]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import simgan_amd as sg  # noqa: E402
from simgan_amd import _lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "northstar"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = bench.WORKLOADS[wl]
ctx = _lib.Context.default()
out = {"workload": wl, "updates": n, "env_steps_per_update": w["T"] * w["N"]}
for mode in ("resident", "host"):
    pol, disc, agent, ro, loader, expert, learner = bench.build_problem(sg, w, seed=0)
    _lib.check(ctx.lib.sg_rollout_fill_synthetic(ro.h, pol.h, 1234, 0.01))
    ro.sync_from_device()                      # the host mirrors now hold the same synthetic rollout
    ro.device_resident = mode == "resident"
    for _ in range(3):
        learner.update()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        learner.update()
    ctx.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    out[mode] = {"ms_per_update": round(ms, 3), "env_steps_per_s": round(w["T"] * w["N"] / ms * 1e3, 1)}
    del learner, agent, disc, ro, pol
out["host_bytes_of_the_rollout"] = 4 * (w["T"] + 1) * w["N"] * (w["O"] + w["F"] + 5) + 4 * w["T"] * w["N"] * (w["A"] + 2)
print(json.dumps(out))
