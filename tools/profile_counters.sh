#!/bin/bash
# SQ / TCP / TCC counter passes (one rocprofv3 --pmc run per pass, never combined with a trace domain other than
# --kernel-trace) for ONE workload of bench.py: the MFMA-utilisation side of BASELINE.json:north_star's evidence.
# Usage (repo root, GPU box):  bash tools/profile_counters.sh <tag> <northstar|hopper|laikago|refine|hopper_ppo> [pass ...]
# Writes gpurun_out/<tag>_<workload>_pmc_<pass>.txt (kernel, counter, samples, mean per dispatch) per pass;
# tools/make_counters.py folds them into profiles/counters.json (read by bench.py for roofline.mfma_busy).
# gfx950 has 8 SQ slots, 4 TCC slots and 2 GRBM slots per pass (MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -u
tag=${1:-rXX}
wl=${2:-northstar}
shift 2 || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
declare -A PASS
PASS[sq_mfma]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
PASS[sq_lds]="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
PASS[sq_mem]="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
PASS[tcc_hit]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
PASS[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum"
passes=${*:-sq_mfma sq_lds tcc_hit tcp sq_mem}
cd /tmp && export TMPDIR=/tmp
export SG_UPDATE_SYNC=1   # counter passes: one host wait per update (the profiler serialises every dispatch)
for p in $passes; do
    d=/tmp/prof_${wl}_$p
    rm -rf $d
    timeout ${PASS_TIMEOUT:-420} rocprofv3 --pmc ${PASS[$p]} --kernel-trace -d $d -o $p --output-format rocpd -- \
        python $root/bench.py --workload $wl --steps ${PROF_STEPS:-2} --warmup ${PROF_WARMUP:-1} --no-cpu-baseline --headline-only > $d.out 2> $d.err
    rc=$?
    db=$(find $d -name "*.db" 2>/dev/null | head -1)
    if [ -n "$db" ]; then
        python $root/tools/rocpd_pmc.py $db > $out/${tag}_${wl}_pmc_$p.txt
        echo "== $p rc=$rc: $(wc -l < $out/${tag}_${wl}_pmc_$p.txt) lines"
    else
        echo "== $p rc=$rc: no database; stderr tail:"; tail -5 $d.err
        tail -20 $d.err > $out/${tag}_${wl}_pmc_$p.err
    fi
done
