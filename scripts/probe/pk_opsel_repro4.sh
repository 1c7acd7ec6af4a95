#!/bin/bash
# Is it the instruction stream?  The reproducer with its rounds unrolled into straight-line code several times the instruction
# cache, ALONE on the GPU (fetch stalls fall between the packed instructions without any neighbour).  bash scripts/probe/pk_opsel_repro4.sh
secs=${1:-10}
out=gpurun_out/pk_repro; mkdir -p $out
cd scripts/probe
for u in 1 512 2048; do
  for v in 0 1; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -DVARIANT=$v -DUNROLL=$u -fno-slp-vectorize pk_opsel_repro.hip -o /tmp/pk_u${u}_v$v -ldl 2>/dev/null || exit 1
    echo -n "UNROLL=$u "; timeout -k 10 $((secs + 60)) /tmp/pk_u${u}_v$v $secs 0 | head -1
  done
done
