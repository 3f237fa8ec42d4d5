#!/bin/bash
# Collects the per-round evidence on the GPU box into gpurun_out/ (copy what should be judged into profiles/):
#   for EVERY workload of bench.py: kernel trace, FETCH_SIZE and WRITE_SIZE in separate --pmc passes, the bench line
#   (tools/profile_workload.sh), the SQ / TCC / TCP counter passes (tools/profile_counters.sh); the PMC calibration on known byte counts; the N > 1 self-test lines of bench.py
#   (--gpus 2 / 8 over the loopback communicator); the pytest -m gpu tail; the end-to-end parity record.
# Usage (from the repo root on the GPU box):  bash tools/profile_round.sh <tag>
# Afterwards, in the dev container:  cp gpurun_out/<tag>_* profiles/ ; python tools/make_traffic.py <tag> <tag> ; python tools/make_counters.py <tag>
set -u
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
for w in northstar hopper laikago refine hopper_ppo; do
    bash tools/profile_workload.sh $tag $w > $out/${tag}_${w}_profile.log 2>&1
    bash tools/profile_counters.sh $tag $w > $out/${tag}_${w}_counters.log 2>&1     # SQ / TCC / TCP passes -> tools/make_counters.py
done
bash tools/profile_calibration.sh $tag 1024 > $out/${tag}_calibration.log 2>&1
for n in 2 8; do
    timeout 900 python bench.py --gpus $n --loopback --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_loopback_n$n.json 2> $out/${tag}_bench_loopback_n$n.err
done
SG_PARITY_RECORD=$out/${tag}_benchpath_parity.json SG_STEPLOCK_RECORD=$out/${tag}_parity.json SG_PARITY_F64_RECORD=$out/${tag}_parity_f64.json timeout 3000 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -60 > $out/${tag}_pytest_gpu.txt
rm -f $out/${tag}_*_profile.log $out/${tag}_*_counters.log $out/${tag}_calibration.log $out/${tag}_*.err
tail -3 $out/${tag}_pytest_gpu.txt
for w in northstar hopper laikago refine hopper_ppo; do head -9 $out/${tag}_${w}_kernel_trace.txt | tail -6; cut -c1-160 $out/${tag}_${w}_bench.json; done
