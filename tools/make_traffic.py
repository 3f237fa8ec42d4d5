"""profiles/<tag>_<workload>_pmc_{fetch,write}_size.txt -> profiles/traffic.json (read by bench.py for roofline.traffic and
roofline.kernels[*].traffic), keyed by workload.

HBM-side bytes per launch = FETCH_SIZE x c_f + WRITE_SIZE x c_w with the corrections MEASURED on this chip by
tools/profile_calibration.sh (profiles/<calib>_pmc_calibration.json; MI355X_MICROARCH.md section HBM prescribes calibrating
on a known byte count in one's own access pattern): coalesced reads of 4, 8 and 16 bytes per lane and range-checked 4-byte
buffer loads all report exactly HALF their bytes in FETCH_SIZE (c_f = 2), WRITE_SIZE reports stores of every width exactly
(c_w = 1).  Counter units are KB.

Each workload's entry also carries `lib_sha256` -- the sha256 of the libsimgan_hip.so the profile run loaded
(gpurun_out/<tag>_<workload>_build.json, written by tools/profile_workload.sh on the GPU box) -- so that bench.py can flag a
summary that belongs to another build (`roofline.traffic_stale`), and every kernel's `rocprof_avg_us` from the kernel trace of
the same run (`roofline.frac_profiled`).

A counter pass that is missing for <tag> (rocprofv3's PMC collection occasionally aborts or hangs on this pool) is taken from
<fallback tag> for the kernels that did not change between the two builds (STALE lists the ones that did); the entry's `source`
says so.

Usage: python tools/make_traffic.py <tag> [<calibration tag> [<fallback tag>]]    e.g.  python tools/make_traffic.py r03_v19 r03_v19 r03_v18"""
import json
import os
import re
import sys

tag = sys.argv[1]
calib_tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
fallback = sys.argv[3] if len(sys.argv) > 3 else None
STALE = tuple(os.environ.get("SG_TRAFFIC_STALE", "").split(",")) if os.environ.get("SG_TRAFFIC_STALE") else ()   # kernels changed since <fallback tag>
calib = json.load(open(f"profiles/{calib_tag}_pmc_calibration.json"))
reads = [v["counter_over_true"] for k, v in calib["FETCH_SIZE"].items() if k.startswith("k_calib_read")]
writes = [v["counter_over_true"] for k, v in calib["WRITE_SIZE"].items() if k.startswith("k_calib_write")]
assert max(reads) - min(reads) < 0.01 and max(writes) - min(writes) < 0.01, "the calibration differs by access width: extend this tool"
c_f, c_w = 1.0 / (sum(reads) / len(reads)), 1.0 / (sum(writes) / len(writes))
doc = {"calibration": {"source": f"profiles/{calib_tag}_pmc_calibration.json", "fetch_correction": round(c_f, 4), "write_correction": round(c_w, 4)},
       "workloads": {}}
for wl in ("northstar", "hopper", "laikago", "refine", "hopper_ppo"):
    files, stale = {}, set()
    for w in ("fetch", "write"):
        f = f"profiles/{tag}_{wl}_pmc_{w}_size.txt"
        if not (os.path.exists(f) and os.path.getsize(f)) and fallback:
            f = f"profiles/{fallback}_{wl}_pmc_{w}_size.txt"
            stale.add(w)
        files[w] = f
    if not all(os.path.exists(f) and os.path.getsize(f) for f in files.values()):
        continue
    out = {}
    for what in ("fetch", "write"):
        for line in open(files[what]):
            m = re.match(r"(?:void )?(\w+)(?:<[^>]*>)?\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)", line)
            if m and not (what in stale and m.group(1) in STALE):
                out.setdefault(m.group(1), {})[f"{what}_kb_raw"] = float(m.group(4))
    for k, v in out.items():
        complete = "fetch_kb_raw" in v and "write_kb_raw" in v
        v["hbm_bytes_per_launch"] = int(1024 * (v["fetch_kb_raw"] * c_f + v["write_kb_raw"] * c_w)) if complete else None
    note = "".join(f"; the {w.upper()}_SIZE pass of {tag} is missing (rocprofv3 aborted / hung): taken from {fallback}, kernels changed since ({', '.join(STALE)}) left without a total"
                   for w in sorted(stale))
    trace = f"profiles/{tag}_{wl}_kernel_trace.txt"
    if os.path.exists(trace):
        for line in open(trace):
            m = re.match(r"(?:void )?(\w+)(?:<[^>]*>)?\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+", line)
            if m:
                out.setdefault(m.group(1), {})["rocprof_avg_us"] = float(m.group(4))
                out[m.group(1)]["rocprof_calls"] = int(m.group(2))
    build = {}
    try:
        build = json.load(open(f"profiles/{tag}_{wl}_build.json"))
    except (OSError, ValueError):
        pass
    doc["workloads"][wl] = {"source": f"{files['fetch']}, {files['write']}{note}", "kernel_trace": trace if os.path.exists(trace) else None,
                            "lib_sha256": build.get("lib_sha256"), "kernels": out}
json.dump(doc, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
for wl, d in doc["workloads"].items():
    print(wl, {k: (v.get("hbm_bytes_per_launch"), v.get("rocprof_avg_us")) for k, v in d["kernels"].items() if k.startswith(("k_disc_chain", "k_disc_wgrad", "k_disc_step", "k_ppo_"))},
          (d["lib_sha256"] or "no build record")[:12], d["source"][-60:])
