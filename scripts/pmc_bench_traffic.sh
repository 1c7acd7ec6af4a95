#!/bin/bash
# HBM bytes per launch of every kernel of a bench workload, from PMC counters as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes, kernel-trace only, KiB units, FETCH_SIZE doubled on gfx950 for wide
# coalesced streams.  Eager launches (PMC serialises kernels anyway; a HIP graph hides per-kernel attribution).
# Run on the GPU box from the repo root:  bash scripts/pmc_bench_traffic.sh [center|vid16|wholeim] [extra bench args]
#   -> gpurun_out/pmc_bench_traffic[_<workload>].json   (copy to profiles/rNN_pmc_bench_traffic[_<workload>].json)
set -e
wl=${1:-center}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_bench_$wl
rm -rf $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $ROOT/bench.py --workload $wl --no-graph --no-overlap --no-cpu-baseline --steps 2 --warmup 1 --step-stats 0 "$@" > $OUT.$c.log 2>&1
done
suffix=""; [ "$wl" != center ] && suffix="_$wl"
VF_PMC_WORKLOAD=$wl python3 $ROOT/scripts/pmc_bench_traffic.py $OUT $ROOT/gpurun_out/pmc_bench_traffic$suffix.json
