#!/bin/bash
# round-6 closing measurements on the GPU box: full GPU test suite, rocprofv3 profiles of the three workloads (kernel trace + PMC passes), bench lines
# (default with the `screened` block, the driver's flags, the other workloads, the plain 2- and 8-rank commands on the one GPU), kernel trace of the
# screened paths, golden pins, the retried RTS-96 states' device record
#   bash scripts/r6_final.sh [notests|benchonly]      benchonly: the bench lines again, once profiles/ holds this code object's counters (counters_stale false)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6_final; mkdir -p $O; cd $R
# the pool's boxes are not all alike: one in a few runs every kernel 21 % slower (17.4 -> 21.2 ms per 1e6 RTS-24 scenarios, the same code object).  The
# committed profile is what bench.py's counters_stale compares launch times with, so it is taken on a box at the usual speed or not at all.
KMS=$(python scripts/one_launch.py 1000000 | awk '{print $2}'); echo "calibration launch: $KMS ms per 1e6 scenarios" | tee $O/box_speed.log
if python -c "import sys; sys.exit(0 if float('$KMS') > 18.6 else 1)"; then echo "slow box: not profiling here" | tee -a $O/box_speed.log; exit 9; fi
if [ "$1" != "notests" ] && [ "$1" != "benchonly" ]; then python -m pytest tests/ -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log; fi
if [ "$1" != "benchonly" ]; then
bash scripts/profile.sh r6f nsq24 > $O/prof24.log 2>&1
bash scripts/profile.sh r6f rts96 > $O/prof96.log 2>&1
bash scripts/profile.sh r6f seq > $O/profseq.log 2>&1
fi
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err; echo "bench (driver's flags) rc $?"
python bench.py --workload rts96 > $O/bench_rts96.json 2> $O/bench_rts96.err; echo "bench96 rc $?"
python bench.py --workload seq > $O/bench_seq.json 2> $O/bench_seq.err; echo "benchseq rc $?"
python bench.py --gpus 2 --share-device --comm host --steps 5 --warmup 1 > $O/bench_2rank_shared.json 2> $O/bench_2rank_shared.err; echo "bench 2 ranks rc $?"
python bench.py --gpus 8 --share-device --comm host --steps 5 --warmup 1 > $O/bench_8rank_shared.json 2> $O/bench_8rank_shared.err; echo "bench 8 ranks rc $?"
python bench.py --gpus 8 --share-device --comm host --scaling strong --total 100000000 --steps 1 --warmup 0 --no-time-to-cov > $O/bench_8rank_strong_1e8.json 2> $O/bench_8rank_strong_1e8.err; echo "bench 8 ranks 1e8 rc $?"
python bench.py --gpus 8 --share-device --comm host --workload seq --years 125 --steps 1 --warmup 0 > $O/bench_8rank_seq_1000y.json 2> $O/bench_8rank_seq_1000y.err; echo "bench 8 ranks seq rc $?"
python bench.py --gpus 8 --share-device --comm host --workload rts96 --batch 1250000 --steps 1 --warmup 0 > $O/bench_8rank_rts96_1e7.json 2> $O/bench_8rank_rts96_1e7.err; echo "bench 8 ranks rts96 rc $?"
if [ "$1" = "benchonly" ]; then head -c 300 $O/bench_default.json; echo; exit 0; fi
bash scripts/screen_profile.sh r6f > $O/screen_profile.log 2>&1
python tests/tools/golden_pin.py > $O/golden_pin.log 2>&1; echo "golden_pin rc $?"
python tests/tools/numfail96_device.py > $O/numfail96_device.json 2> $O/numfail96_device.err; echo "numfail96 rc $?"
python scripts/converged.py > $O/converged.log 2>&1
make -C powersystemsreliabilityassessment_amd/csrc ablate/librelmc_pt.so > $O/make_pt.log 2>&1     # the phase-timing build of THIS source (a stale one lacks the newer symbols)
python scripts/wave_tail.py > $O/wave_tail.log 2>&1
head -c 300 $O/bench_default.json; echo; tail -3 $O/golden_pin.log
