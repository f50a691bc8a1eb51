#!/bin/bash
# RTS-96 under the shipped (tuned) primary order: the states of the first 1e8 samples (seed 1) on which it ends non-converged, retries off, + the C oracle on them
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c30; mkdir -p $O; cd $R
RELMC_NO_RETRY=1 python tests/tools/numfail96.py 1e8 1 > $O/numfail96_scan.json 2> $O/scan.err; tail -2 $O/scan.err; python -c "
import json; d = json.load(open('$O/numfail96_scan.json')); print(len(d['states']), 'entries;', len({tuple(x['failed']) for x in d['states']}), 'distinct states')"
