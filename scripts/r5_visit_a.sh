#!/bin/bash
# round-5 GPU visit: new tests, the three workloads with the gather patch kernel on / off (same box), bf16 lines, vid16 --overlap
tag=${1:-r5g}
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_t7.py tests/test_gpu_pgemm.py -m gpu -q > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${tag}_tests.log
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["unit"], d["ms_per_step"], "ms |", r["kernel"], r["achieved"], "fam", r.get("family_weighted", {}).get("achieved"), "pipe", r.get("family_weighted", {}).get("frac_of_pipe_bound"), "worst", r.get("worst_symbol", {}).get("kernel"), r.get("worst_symbol", {}).get("frac_of_pipe_bound"))
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
}
for rep in 1 2; do
for g in 1 0; do
  VF_PG_GPATCH=$g timeout -k 10 300 python bench.py --workload center --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_center_gpatch${g}_$rep.json 2> gpurun_out/${tag}_center_gpatch${g}_$rep.err || exit 1
  line gpurun_out/${tag}_center_gpatch${g}_$rep.json
done
done
for wl in vid16 wholeim; do
  timeout -k 10 300 python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_$wl.json 2> gpurun_out/${tag}_$wl.err || exit 1
  line gpurun_out/${tag}_$wl.json
done
timeout -k 10 300 python bench.py --workload vid16 --overlap --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_vid16_overlap.json 2> gpurun_out/${tag}_vid16_overlap.err || exit 1
line gpurun_out/${tag}_vid16_overlap.json
for wl in center vid16 wholeim; do
  timeout -k 10 300 python bench.py --workload $wl --mfma bf16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_${wl}_mfma_bf16.json 2> gpurun_out/${tag}_${wl}_mfma_bf16.err || exit 1
  line gpurun_out/${tag}_${wl}_mfma_bf16.json
done
