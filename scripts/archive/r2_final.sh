#!/bin/bash
# round-2 closing measurements: full GPU test suite, profiles of the three workloads, bench lines, auxiliary rates
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r2_final; mkdir -p $O; cd $R
python -m pytest tests/ -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log; tail -4 $O/pytest_all.log
bash scripts/profile.sh r2f nsq24 > $O/prof24.log 2>&1
bash scripts/profile.sh r2f rts96 > $O/prof96.log 2>&1
bash scripts/profile.sh r2f seq > $O/profseq.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python bench.py --workload rts96 > $O/bench_rts96.json 2> $O/bench_rts96.err; echo "bench96 rc $?"
python bench.py --workload seq > $O/bench_seq.json 2> $O/bench_seq.err; echo "benchseq rc $?"
python scripts/db_rate.py > $O/db_rate.log 2>&1
python scripts/pcie_rate.py > $O/pcie.log 2>&1
python scripts/converged.py > $O/converged.log 2>&1
head -c 600 $O/bench_default.json; echo; head -c 400 $O/bench_rts96.json; echo; head -c 400 $O/bench_seq.json; echo; tail -3 $O/pcie.log
