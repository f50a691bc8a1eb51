#!/bin/bash
# closing run of the shipped binary: 1e10 / 1e9-sample soak, the three default bench lines (their roofline counters from the committed profile)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3_close; mkdir -p $O; cd $R
timeout 1500 python scripts/soak2.py 1e10 2>&1 | tee $O/soak.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python bench.py --workload rts96 > $O/bench_rts96.json 2> $O/bench_rts96.err; echo "bench96 rc $?"
python bench.py --workload seq > $O/bench_seq.json 2> $O/bench_seq.err; echo "benchseq rc $?"
head -c 300 $O/bench_default.json; echo
