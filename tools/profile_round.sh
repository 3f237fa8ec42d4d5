#!/bin/bash
# Collects the per-round evidence on the GPU box into gpurun_out/ (copy what should be judged into profiles/):
#   kernel trace (+gaps), FETCH_SIZE and WRITE_SIZE in separate --pmc passes, the bench line, the pytest tail.
# Usage (from the repo root on the GPU box):  bash tools/profile_round.sh <tag>
set -u
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run_prof() {  # name, rocprofv3 args...
    local name=$1; shift
    rm -rf /tmp/prof_$name
    timeout 600 rocprofv3 "$@" -d /tmp/prof_$name -o $name --output-format rocpd -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_${name}_bench_under_profiler.json 2> /tmp/prof_$name.err
    find /tmp/prof_$name -name "*.db" | head -1
}
db=$(run_prof trace --kernel-trace)
python $root/tools/rocpd_stats.py $db > $out/${tag}_kernel_trace.txt
db=$(run_prof fetch --pmc FETCH_SIZE --kernel-trace)
python $root/tools/rocpd_pmc.py $db > $out/${tag}_pmc_fetch_size.txt
db=$(run_prof write --pmc WRITE_SIZE --kernel-trace)
python $root/tools/rocpd_pmc.py $db > $out/${tag}_pmc_write_size.txt
cd $root
timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
for w in hopper laikago refine; do   # the other BASELINE.json configurations (DESIGN.md section 5 quotes these files)
    timeout 600 python bench.py --workload $w --cpu-seconds 8 > $out/${tag}_bench_$w.json 2> $out/${tag}_bench_$w.err
done
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $out/${tag}_pytest_gpu.txt
rm -f $out/${tag}_*_bench_under_profiler.json
tail -3 $out/${tag}_pytest_gpu.txt
head -14 $out/${tag}_kernel_trace.txt
head -8 $out/${tag}_pmc_fetch_size.txt
head -8 $out/${tag}_pmc_write_size.txt
cut -c1-300 $out/${tag}_bench.json
