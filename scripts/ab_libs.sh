#!/bin/bash
# Same-box A/B of library builds: scripts/ab_libs.sh OUT ROUNDS lib1 lib2 ...   ("cur" = the in-tree build)
out=$1; rounds=$2; shift 2
for i in $(seq 1 $rounds); do
  for lib in "$@"; do
    if [ "$lib" = cur ]; then unset VF_HIP_LIB; else export VF_HIP_LIB=$PWD/video-filler_amd/lib/alt/libvf_hip_$lib.so; fi
    timeout -k 10 120 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python scripts/ab_line.py "$lib" >> $out || exit 1
  done
done
