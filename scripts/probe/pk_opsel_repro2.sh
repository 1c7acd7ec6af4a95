#!/bin/bash
# What the packed-FMA fault needs: (a) two copies of the reproducer itself on the GPU (other PROCESS, same kernel), (b) one busy
# trainer as the only neighbour.  bash scripts/probe/pk_opsel_repro2.sh [seconds]   (after pk_opsel_repro.sh built the binaries)
secs=${1:-15}
out=gpurun_out/pk_repro; mkdir -p $out
(cd scripts/probe && for v in 0 1; do [ -x pk_opsel_repro$v ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -DVARIANT=$v -fno-slp-vectorize pk_opsel_repro.hip -o pk_opsel_repro$v -ldl 2>/dev/null || exit 1; done) || exit 1
echo "== two copies of variant 0 side by side (no other neighbour)"
scripts/probe/pk_opsel_repro0 $secs > $out/pair_a.log 2>&1 & p=$!
scripts/probe/pk_opsel_repro0 $secs > $out/pair_b.log 2>&1; wait $p
head -1 $out/pair_a.log; head -1 $out/pair_b.log; tail -1 $out/pair_a.log
echo "== variant 0 beside ONE busy trainer process"
VF_PROBE_ITERS=100000 timeout -k 10 $((secs + 60)) python scripts/probe/multi_trainer_det.py > $out/noise1.log 2>&1 & n1=$!
sleep 12
scripts/probe/pk_opsel_repro0 $secs > $out/one_neighbour.log 2>&1; head -1 $out/one_neighbour.log; tail -1 $out/one_neighbour.log
kill $n1 2>/dev/null; wait
