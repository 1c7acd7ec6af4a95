#!/bin/bash
# configs[2] (bench.py --workload vid16): same-box sweep of the routing knobs that decide how the batchSize-16 passes are tiled
tag=${1:-r5h}
mkdir -p gpurun_out
out=gpurun_out/${tag}_vid16_sweep.txt
: > $out
run() {   # label, env...
  label=$1; shift
  v=$(env "$@" timeout -k 10 200 python bench.py --workload vid16 --steps 40 --warmup 5 --no-cpu-baseline --step-stats 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])")
  echo "$label: $v" | tee -a $out
}
run "default" A=1
run "default (again)" A=1
run "VF_SPLIT_BLOCKS=256" VF_SPLIT_BLOCKS=256
run "VF_SPLIT_BLOCKS=384" VF_SPLIT_BLOCKS=384
run "VF_SPLIT_BLOCKS=768" VF_SPLIT_BLOCKS=768
run "VF_SPLIT_BLOCKS=1024" VF_SPLIT_BLOCKS=1024
run "VF_IGEMM_DB=0" VF_IGEMM_DB=0
run "VF_PCONV_MIN_GFLOP=1.0" VF_PCONV_MIN_GFLOP=1.0
run "VF_PCONV_MIN_GFLOP=0.5" VF_PCONV_MIN_GFLOP=0.5
run "VF_PCONV_MIN_GFLOP=1.0 VF_PG_SPLIT_BLOCKS=680" VF_PCONV_MIN_GFLOP=1.0 VF_PG_SPLIT_BLOCKS=680
run "VF_WGRAD_BLOCKS=256" VF_WGRAD_BLOCKS=256
run "VF_WGRAD_BLOCKS=1024" VF_WGRAD_BLOCKS=1024
run "VF_PWGRAD_BLOCKS=512" VF_PWGRAD_BLOCKS=512
run "default (end)" A=1
