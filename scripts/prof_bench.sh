#!/bin/bash
# rocprofv3 kernel trace + stats of a bench.py command; the per-kernel summary lands in gpurun_out/<tag>_kernel_stats.csv and the
# bench line taken under the profiler in gpurun_out/<tag>_bench_under_rocprof.json.   bash scripts/prof_bench.sh <tag> [bench args]
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $ROOT/gpurun_out/${tag}_bench_under_rocprof.json 2> $ROOT/gpurun_out/${tag}_prof.err
rc=$?
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $ROOT/gpurun_out/${tag}_kernel_stats.csv
head -45 $ROOT/gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
exit $rc
