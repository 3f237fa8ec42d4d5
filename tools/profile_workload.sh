#!/bin/bash
# Kernel trace + FETCH_SIZE / WRITE_SIZE PMC passes (separate runs) + the bench line for ONE workload of bench.py.
# Usage (repo root, GPU box):  bash tools/profile_workload.sh <tag> <northstar|hopper|laikago|refine|hopper_ppo>
# Writes gpurun_out/<tag>_<workload>_{kernel_trace,pmc_fetch_size,pmc_write_size}.txt and <tag>_<workload>_bench.json;
# copy what should be judged into profiles/ and regenerate profiles/traffic.json with tools/make_traffic.py.
set -u
tag=${1:-rXX}
wl=${2:-northstar}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
python - <<PYEOF
import hashlib, json
json.dump({"lib_sha256": hashlib.sha256(open("$root/simgan_amd/libsimgan_hip.so", "rb").read()).hexdigest()}, open("$out/${tag}_${wl}_build.json", "w"))
PYEOF
cd /tmp && export TMPDIR=/tmp
run_prof() {  # name, rocprofv3 args...
    local name=$1; shift
    rm -rf /tmp/prof_${wl}_$name
    timeout 600 rocprofv3 "$@" -d /tmp/prof_${wl}_$name -o $name --output-format rocpd -- python $root/bench.py --workload $wl --steps ${PROF_STEPS:-2} --warmup ${PROF_WARMUP:-1} --no-cpu-baseline --headline-only > /tmp/prof_${wl}_$name.out 2> /tmp/prof_${wl}_$name.err
    find /tmp/prof_${wl}_$name -name "*.db" | head -1
}
if [ "${PMC_ONLY:-0}" != "1" ]; then
db=$(run_prof trace --kernel-trace)
python $root/tools/rocpd_stats.py $db > $out/${tag}_${wl}_kernel_trace.txt
fi
export SG_UPDATE_SYNC=1   # counter passes: one host wait per update (the profiler serialises every dispatch)
db=$(run_prof fetch --pmc FETCH_SIZE --kernel-trace)
python $root/tools/rocpd_pmc.py $db > $out/${tag}_${wl}_pmc_fetch_size.txt
db=$(run_prof write --pmc WRITE_SIZE --kernel-trace)
python $root/tools/rocpd_pmc.py $db > $out/${tag}_${wl}_pmc_write_size.txt
unset SG_UPDATE_SYNC
cd $root
[ "${PMC_ONLY:-0}" = "1" ] && exit 0
# (the north-star line is the driver's own command: it also carries the other workloads' brief runs and the drop-in legs)
extra=$([ "$wl" = "northstar" ] && echo "" || echo "--headline-only")
timeout 900 python bench.py --workload $wl --cpu-seconds 8 $extra > $out/${tag}_${wl}_bench.json 2> $out/${tag}_${wl}_bench.err
[ -f $out/bench_full_${wl}_n1.json ] && mv $out/bench_full_${wl}_n1.json $out/${tag}_${wl}_bench_full.json   # the full record beside the compact line
head -12 $out/${tag}_${wl}_kernel_trace.txt
head -6 $out/${tag}_${wl}_pmc_fetch_size.txt
head -6 $out/${tag}_${wl}_pmc_write_size.txt
cut -c1-200 $out/${tag}_${wl}_bench.json
