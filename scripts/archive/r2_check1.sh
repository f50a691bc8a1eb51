#!/bin/bash
# round-2 first GPU check: new database tests, base kernel timing, database rates
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r2c1; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_parity.py tests/test_rts96.py -m gpu -x -q -k "database or distinct" > $O/pytest_db.log 2>&1; echo "pytest rc $?" >> $O/pytest_db.log
tail -5 $O/pytest_db.log
python scripts/variant_check.py base > $O/variant.log 2>&1; cat $O/variant.log
python scripts/db_rate.py > $O/db_rate.log 2>&1; cat $O/db_rate.log
