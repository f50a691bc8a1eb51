#!/bin/bash
# same-box A/B of library builds:  bash scripts/ab_libs.sh <workload> <lib relative to csrc> ...   (3 alternating repetitions)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
WL=$1; shift
for rep in 1 2 3; do for lib in "$@"; do
  RELMC_LIB_PATH=$R/powersystemsreliabilityassessment_amd/csrc/$lib python bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-time-to-cov --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$WL $lib kernel_ms %.3f ms_per_step %.3f n_nonconv %d edns %.9f' % (d['roofline']['kernel_ms_avg'], d['ms_per_step'], d['indices']['n_nonconverged'], d['indices']['edns_mw']))"
done; done
