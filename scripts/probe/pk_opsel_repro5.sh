#!/bin/bash
# Which of the library's kernel families, running in ANOTHER process, triggers the packed-FMA fault in the reproducer?
# bash scripts/probe/pk_opsel_repro5.sh [seconds]
secs=${1:-8}
out=gpurun_out/pk_repro; mkdir -p $out
(cd scripts/probe && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -DVARIANT=0 -fno-slp-vectorize pk_opsel_repro.hip -o pk_opsel_repro0 -ldl 2>/dev/null) || exit 1
run() {   # name, command...
  name=$1; shift
  setsid bash -c 'while true; do "$@" > /dev/null 2>&1 || break; done' _ "$@" & nb=$!      # (its own process group: killed as a group below)
  sleep 12
  echo -n "$name: "; timeout -k 10 $((secs + 60)) scripts/probe/pk_opsel_repro0 $secs 0 | head -1 | sed 's/.*| //'
  kill -- -$nb 2>/dev/null; wait $nb 2>/dev/null
  sleep 2
}
run "planes GEMMs (bench_pconv.py: LDS-DMA + bf16 MFMA)" env NB=400 python scripts/bench_pconv.py 64
run "in-kernel-split GEMMs (bench_conv.py E2-E5)" env ONLY=E2,E3,E4,E5 python scripts/bench_conv.py 64
run "BatchNorm passes (bench_bn.py: streaming + LDS reductions)" python scripts/bench_bn.py 64
run "fused bottleneck update (bench_fused_adam.py: fp32 MFMA + streaming)" python scripts/bench_fused_adam.py
