#!/bin/bash
# One rocprofv3 --pmc pass over bench.py WITH its 100-step eager percentile pass (the combination that aborted in round 1
# when that pass queued 100 steps of launches without a host synchronisation; bench.py now bounds it to 8 steps in flight).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_verify
rm -rf $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-graph --no-overlap --no-cpu-baseline --steps 2 --warmup 1 --step-stats 100 > $OUT.log 2>&1
rc=$?
echo "rc=$rc"; grep -c . $(find $OUT -name "*counter_collection.csv" | head -1); tail -c 600 $OUT.log
exit $rc
