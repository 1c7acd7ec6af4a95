#!/bin/bash
# The stand-alone packed-FMA reproducer (scripts/probe/pk_opsel_repro.hip) beside two busy neighbours (full trainers of the shipped
# library): variant 0 = the suspect operand form, variant 1 = the control.  bash scripts/probe/pk_opsel_repro.sh [seconds]
secs=${1:-25}
out=gpurun_out/pk_repro; mkdir -p $out
(cd scripts/probe && for v in 0 1; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -DVARIANT=$v -fno-slp-vectorize pk_opsel_repro.hip -o pk_opsel_repro$v -ldl 2>/dev/null || exit 1; done) || exit 1
echo "== alone on the GPU"
for v in 0 1; do timeout -k 10 $((secs + 30)) scripts/probe/pk_opsel_repro$v 8 > $out/alone$v.log 2>&1; head -3 $out/alone$v.log; done
echo "== beside two busy neighbours"
VF_PROBE_ITERS=100000 timeout -k 10 $((2 * secs + 60)) python scripts/probe/multi_trainer_det.py > $out/noise1.log 2>&1 & n1=$!
VF_PROBE_ITERS=100000 timeout -k 10 $((2 * secs + 60)) python scripts/probe/multi_trainer_det.py > $out/noise2.log 2>&1 & n2=$!
sleep 12
for v in 0 1; do timeout -k 10 $((secs + 30)) scripts/probe/pk_opsel_repro$v $secs > $out/variant$v.log 2>&1; head -16 $out/variant$v.log; done
kill $n1 $n2 2>/dev/null; wait
echo "neighbours: $(grep -c MISMATCH $out/noise1.log $out/noise2.log | tr '\n' ' ')"
