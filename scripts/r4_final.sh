#!/bin/bash
# round-4 closing measurements: full GPU test suite, profiles of the three workloads, bench lines, golden pin, instruction mix
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_final; mkdir -p $O; cd $R
python -m pytest tests/ -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
bash scripts/profile.sh r4f nsq24 > $O/prof24.log 2>&1
bash scripts/profile.sh r4f rts96 > $O/prof96.log 2>&1
bash scripts/profile.sh r4f seq > $O/profseq.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err; echo "bench (driver's flags) rc $?"
python bench.py --workload rts96 > $O/bench_rts96.json 2> $O/bench_rts96.err; echo "bench96 rc $?"
python bench.py --workload seq > $O/bench_seq.json 2> $O/bench_seq.err; echo "benchseq rc $?"
python tests/tools/golden_pin.py > $O/golden_pin.log 2>&1; echo "golden_pin rc $?"
python scripts/converged.py > $O/converged.log 2>&1
python scripts/pcie_rate.py > $O/pcie.log 2>&1
bash scripts/pmc.sh r4f > $O/pmc_mix.log 2>&1
head -c 500 $O/bench_default.json; echo; tail -3 $O/golden_pin.log
