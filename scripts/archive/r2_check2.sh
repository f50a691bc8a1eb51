#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r2c2; mkdir -p $O; cd $R
python -m pytest tests/test_c_abi.py tests/test_host.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -15 $O/pytest.log
bash scripts/profile.sh r2a nsq24 > $O/prof24.log 2>&1
bash scripts/profile.sh r2a rts96 > $O/prof96.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json; tail -3 $O/bench_default.err
