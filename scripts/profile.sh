#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  bash scripts/profile.sh <tag> [workload]
# 1) kernel trace + stats of the bench command, 2) PMC passes (each in its own run, never combined with other trace
# domains), 3) HBM traffic passes (FETCH_SIZE / WRITE_SIZE separately).  workload: nsq24 (default) | rts96 | seq
TAG=${1:-r2}
WL=${2:-nsq24}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_$WL
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-time-to-cov --no-secondary --no-sustained"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace_bench.log 2>&1
PMC="python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-time-to-cov --no-secondary --no-sustained"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -f csv -d $OUT/pmc_sq -o pmc -- $PMC > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_lds -o pmc -- $PMC > $OUT/pmc_lds.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o pmc -- $PMC > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o pmc -- $PMC > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -40
