#!/bin/bash
# kernel trace of the screened paths (pre-pass kernels + relmc_eval_kernel<7> / <2>):  bash scripts/screen_profile.sh <tag>  -> gpurun_out/screen_<tag>/
TAG=${1:-x}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/screen_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for w in rts24 rts96 seq; do
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/$w -o t -- python3 $R/scripts/screen_launch.py $w > $OUT/$w.log 2>&1
  f=$(find $OUT/$w -name "*kernel_stats.csv" | head -1); echo "== $w"; tail -1 $OUT/$w.log; [ -n "$f" ] && cut -d, -f1-7 $f | head -12 | cut -c1-200
done
