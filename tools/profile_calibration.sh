#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/pmc_calibrate.py) -> gpurun_out/<tag>_pmc_calibration.json
set -u
tag=${1:-rXX}
mb=${2:-1024}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/calib_$c
    timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/calib_$c -o c --output-format rocpd -- python $root/tools/pmc_calibrate.py $mb > /tmp/calib_$c.out 2> /tmp/calib_$c.err
    python $root/tools/rocpd_pmc.py $(find /tmp/calib_$c -name "*.db" | head -1) > $out/${tag}_calib_$c.txt
done
python - <<PY
import json, re
mb = $mb
res = {"true_kb_per_dispatch": mb * 1024, "source": "tools/profile_calibration.sh: k_calib_* kernels move this many KB each, coalesced, buffers > Infinity Cache"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for line in open("$out/${tag}_calib_%s.txt" % c):
        m = re.match(r"(?:void )?(k_calib_\w+)(<\d+>)?\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)", line)
        if m:
            res.setdefault(c, {})[m.group(1) + (m.group(2) or "")] = {"counter_kb": float(m.group(5)), "counter_over_true": float(m.group(5)) / (mb * 1024)}
json.dump(res, open("$out/${tag}_pmc_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
