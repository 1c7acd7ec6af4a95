#!/bin/bash
# --overlap (dW beside dX on side streams, netG forward beside netD's real pass) is a mirror-host feature: same-box A/B on both workloads
tag=${1:-r5m}
out=gpurun_out/${tag}_overlap_ab.txt
: > $out
run() { label=$1; shift; v=$(timeout -k 10 200 python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --step-stats 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['config'].get('streams'))"); echo "$label: $v" | tee -a $out; }
for rep in 1 2; do
run "vid16 cabi" --workload vid16
run "vid16 mirror" --workload vid16 --host mirror
run "vid16 mirror --overlap" --workload vid16 --host mirror --overlap
run "center cabi" --workload center
run "center mirror" --workload center --host mirror
run "center mirror --overlap" --workload center --host mirror --overlap
done
