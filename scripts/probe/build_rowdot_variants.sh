#!/bin/bash
# Builds video-filler_amd/lib/alt/libvf_hip_rd<tag>.so: the shipped objects with vf_smallm.o replaced by one variant of
# scripts/probe/rowdot_variants.hip (see its header).  Run from the repo root after video-filler_amd/build.py.
set -e
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
root=$PWD; obj=$root/video-filler_amd/lib/obj; alt=$root/video-filler_amd/lib/alt
mkdir -p $alt /tmp/rdvar
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -I$root/video-filler_amd/csrc"
objs=$(ls $obj/*.o | grep -v vf_smallm.o)
build() {  # tag RED LOADS DIAG [CHECK]
  $HIPCC $F -DRED=$2 -DLOADS=$3 -DDIAG=$4 -DCHECK=${5:-0} -c scripts/probe/rowdot_variants.hip -o /tmp/rdvar/rd_$1.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $alt/libvf_hip_rd$1.so $objs /tmp/rdvar/rd_$1.o -ldl
  if [ -n "$KEEP_ASM" ]; then $HIPCC $F -DRED=$2 -DLOADS=$3 -DDIAG=$4 -S --cuda-device-only scripts/probe/rowdot_variants.hip -o /tmp/rdvar/rd_$1.s; fi
}
build prefix 0 0 0 &      # the pre-fix kernel
build prefixdiag 0 0 1 &  # ... also storing the per-lane partials
build uncond 0 1 0 &      # butterfly kept, loads unconditional + sched_barrier
build lds 1 0 0 &         # LDS reduction, conditional loads kept
wait
build depth1 2 0 0 &      # butterfly, one value's ds_bpermute in flight at a time
build dpp 3 0 0 &         # no LDS-pipe instruction
build lanestore 4 0 0 &   # butterfly, one store per lane
build ldsalloc 5 0 0 &    # butterfly, lane-0 stores, but an LDS allocation
build prefixcheck 0 0 1 1 &  # pre-fix + per-launch checker kernel
wait
ls -la $alt
