#!/bin/bash
# HBM traffic of the conv kernels from PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), kernel-trace only, units = KiB, and on gfx950
# FETCH_SIZE reports half the bytes of a wide coalesced stream (double it before comparing with a byte count).
# Run on the GPU box from the repo root:  bash scripts/pmc_traffic.sh
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
for c in FETCH_SIZE WRITE_SIZE; do
  ONLY=E2,E3,E6 REPS=2 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $GRAFT_REPO_ROOT/scripts/bench_conv.py 64 > $OUT.$c.log 2>&1
done
ls $OUT/*/*/
