#!/bin/bash
# The isolated trigger of the packed-FMA fault, stand-alone: run from the repo root on the GPU box.
#   bash scripts/probe/pk_opsel_repro8.sh > gpurun_out/pk_opsel_repro8.txt
# Part 1 needs nothing but pk_opsel_repro.hip; parts 2 and 3 use the library as the neighbour (part 3: ablated builds, if present
# under video-filler_amd/lib/alt/ — they are made out of tree from patched copies of vf_conv.hip and not kept).
set -e
cd scripts/probe
hipcc --offload-arch=gfx950 -O2 -DVARIANT=0 -fno-slp-vectorize pk_opsel_repro.hip -o pk_opsel_repro0 -ldl 2>/dev/null
hipcc --offload-arch=gfx950 -O2 -DVARIANT=1 -fno-slp-vectorize pk_opsel_repro.hip -o pk_opsel_repro1 -ldl 2>/dev/null
cd ../..
short() { grep -v "  block" | sed "s/.*| //" | cut -c1-140; }
echo "# ---- part 1: two-instruction neighbour loops (300 + form: 0/1/2 bf16 MFMA 32x32x16 + ds_write_b32/b64/b128, 3 MFMA and ds_write_b64 in"
echo "#      different waves, 4 ds_write_b64 alone, 5 bf16 MFMA 16x16x32 + ds_write_b64, 6 fp32 MFMA 32x32x2 + ds_write_b64), 5 s each"
for w in 0 1 2 3 4 5 6; do echo "form $w:"; scripts/probe/pk_opsel_repro0 5 $((300 + w)) | short; done
echo "control (src0 / src1 exchanged) beside form 2 and form 3:"
scripts/probe/pk_opsel_repro1 5 302 | short
scripts/probe/pk_opsel_repro1 5 303 | short
echo "# ---- part 2: the library's E3 forward (vf_conv2d_fwd on a second stream) in each arithmetic mode, and mode 3 single-buffered"
for m in 0 1 3; do echo "mfma mode $m:"; NB_MFMA_MODE=$m scripts/probe/pk_opsel_repro0 6 5 | short; done
echo "mfma mode 3, VF_IGEMM_DB=0:"; VF_IGEMM_DB=0 scripts/probe/pk_opsel_repro0 6 5 | short
echo "# ---- part 3: ablations of the double-buffered kernel's K loop"
for v in noload nostore noboth nomfma; do
  lib=$PWD/video-filler_amd/lib/alt/libvf_hip_$v.so
  [ -f $lib ] || continue
  echo "$v:"; VF_HIP_LIB=$lib scripts/probe/pk_opsel_repro0 6 5 | short
done
