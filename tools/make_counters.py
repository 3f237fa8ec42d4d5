"""profiles/<tag>_<workload>_pmc_<pass>.txt (tools/profile_counters.sh -> tools/rocpd_pmc.py) -> profiles/counters.json, read by
bench.py for roofline.mfma_busy and roofline.kernels[*].mfma_busy: the MFMA-utilisation side of the evidence
BASELINE.json:north_star asks for, from the SQ counters rather than from algorithmic FLOPs / time.

Per kernel and counter: `sum` (chip total per dispatch, mean over dispatches), `mean_inst` / `max_inst` (per hardware instance:
a shader engine for SQ_*, an XCD for GRBM_*).  Derived per kernel:
  mfma_busy        = SQ_VALU_MFMA_BUSY_CYCLES.sum / (SQ_BUSY_CYCLES.mean_inst x 1024 SIMDs)  -- the fraction of the chip's
                     matrix pipes' cycles, over the time the kernel keeps a shader engine busy, in which an MFMA executes
                     (AMD's MfmaUtil with the SQ's own busy clock in place of GRBM_GUI_ACTIVE: under the profiler's
                     per-dispatch serialisation GRBM_GUI_ACTIVE also counts the counter set-up around a 8 us kernel --
                     `mfma_busy_grbm` is that formula, for reference)
  mfma_busy_resident = SQ_VALU_MFMA_BUSY_CYCLES.sum / (SQ_BUSY_CU_CYCLES.sum x 4) -- the same over the CU-cycles in which
                     a CU holds at least one wave of the kernel (the kernel does not fill 256 CUs)
  mfma_flops       = SQ_INSTS_VALU_MFMA_MOPS_F32.sum x 512 (executed, padding included)
  wave_wait / wave_issue_stall / wave_active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
  lds_conflict     = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  l2_hit           = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
Usage: python tools/make_counters.py <tag>     (run from the repo root; <tag>_<workload>_build.json supplies the library hash)"""
import glob
import json
import os
import re
import sys

tag = sys.argv[1]
doc = {"tag": tag, "simd_count": 1024, "cu_count": 256, "workloads": {}}
for wl in ("northstar", "hopper", "laikago", "refine", "hopper_ppo"):
    kernels, passes = {}, []
    for f in sorted(glob.glob(f"profiles/{tag}_{wl}_pmc_*.txt")):
        p = re.match(rf"profiles/{tag}_{wl}_pmc_(\w+)\.txt", f).group(1)
        if p in ("fetch_size", "write_size"):
            continue
        passes.append(p)
        for line in open(f):
            m = re.match(r"(?:void )?(\w+)(?:<[^>]*>)?\s+(\w+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
            if m:
                k = kernels.setdefault(m.group(1), {})
                if m.group(2) == "GRBM_GUI_ACTIVE" and m.group(2) in k:
                    continue     # collected in every pass: keep the first
                k[m.group(2)] = {"sum": float(m.group(4)), "mean_inst": float(m.group(6)), "max_inst": float(m.group(7)), "dispatches": int(m.group(3))}
    if not kernels:
        continue
    out = {}
    for name, c in kernels.items():
        g = lambda n, f="sum": c[n][f] if n in c else None  # noqa: E731
        d = {}
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("SQ_BUSY_CYCLES", "mean_inst"):
            d["mfma_busy"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("SQ_BUSY_CYCLES", "mean_inst") * doc["simd_count"]), 5)
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("GRBM_GUI_ACTIVE", "max_inst"):
            d["mfma_busy_grbm"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE", "max_inst") * doc["simd_count"]), 5)
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("SQ_BUSY_CU_CYCLES"):
            d["mfma_busy_resident"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("SQ_BUSY_CU_CYCLES") * 4), 5)
        if g("SQ_INSTS_VALU_MFMA_MOPS_F32") is not None:
            d["mfma_flops"] = int(g("SQ_INSTS_VALU_MFMA_MOPS_F32") * 512)
        if g("SQ_WAVE_CYCLES"):
            for key, n in (("wave_wait", "SQ_WAIT_ANY"), ("wave_issue_stall", "SQ_WAIT_INST_ANY"), ("wave_active", "SQ_ACTIVE_INST_ANY")):
                if g(n) is not None:
                    d[key] = round(g(n) / g("SQ_WAVE_CYCLES"), 4)
        if g("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict"] = round((g("SQ_LDS_BANK_CONFLICT") or 0.0) / g("SQ_LDS_IDX_ACTIVE"), 4)
        if g("TCC_HIT_sum") is not None and (g("TCC_HIT_sum") + (g("TCC_MISS_sum") or 0.0)) > 0:
            d["l2_hit"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
        if g("SQ_BUSY_CYCLES", "mean_inst"):
            d["busy_cycles_per_se"] = round(g("SQ_BUSY_CYCLES", "mean_inst"), 1)
        out[name] = {"derived": d, "counters": c}
    build = {}
    try:
        build = json.load(open(f"profiles/{tag}_{wl}_build.json"))
    except (OSError, ValueError):
        pass
    doc["workloads"][wl] = {"passes": passes, "lib_sha256": build.get("lib_sha256"), "kernels": out}
json.dump(doc, open("profiles/counters.json", "w"), indent=1, sort_keys=True)
for wl, d in doc["workloads"].items():
    print(wl, d["passes"], (d["lib_sha256"] or "no build record")[:12])
    for k, v in d["kernels"].items():
        if k.startswith(("k_disc_step", "k_disc_chain", "k_disc_wgrad", "k_ppo_bwd", "k_ppo_pair", "k_ppo_fwd", "k_disc_forward")):
            print("   ", k, v["derived"])
