#!/bin/bash
# The in-place row-dot probe (scripts/probe/smallm_det.py) for ONE variant library, with two busy neighbours on the GPU
# (full trainers of the shipped library).  bash scripts/probe/rowdot_inplace_noisy.sh <variant> [N]
v=$1; N=${2:-40000}
out=gpurun_out/rowdot_noisy; mkdir -p $out
VF_PROBE_ITERS=400 timeout -k 10 300 python scripts/probe/multi_trainer_det.py > $out/noise1.log 2>&1 & n1=$!
VF_PROBE_ITERS=400 timeout -k 10 300 python scripts/probe/multi_trainer_det.py > $out/noise2.log 2>&1 & n2=$!
sleep 8
VF_HIP_LIB=$PWD/video-filler_amd/lib/alt/libvf_hip_rd$v.so N=$N timeout -k 10 280 python scripts/probe/smallm_det.py > $out/$v.log 2>&1
rc=$?
kill $n1 $n2 2>/dev/null; wait
echo "== $v (rc $rc): $(grep -h 'mismatching runs' $out/$v.log)"; grep -c "^run" $out/$v.log
