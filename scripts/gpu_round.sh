#!/bin/bash
# One GPU-box visit: the -m gpu suite, then the three bench workloads.  Usage (from the repo root, through gpurun):
#   bash scripts/gpu_round.sh <tag> [pytest args...]
tag=${1:-x}; shift
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=15 "$@" > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -5 gpurun_out/${tag}_tests.log
if [ $rc -ge 124 ]; then echo "pytest killed (rc $rc): no further GPU step"; exit $rc; fi
for wl in center vid16 wholeim; do
  extra=""      # every committed line carries cpu_baseline: the oracle on a bounded sample, ~10-60 s
  timeout -k 10 420 python bench.py --workload $wl --steps 20 --warmup 5 $extra > gpurun_out/${tag}_bench_$wl.json 2> gpurun_out/${tag}_bench_$wl.err
  brc=$?
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${tag}_bench_$wl.json"))
    print("$wl", d["value"], d["unit"], d["ms_per_step"], "ms", d["roofline"]["kernel"], d["roofline"]["achieved"])
except Exception as e:
    print("$wl bench failed:", e)
PY
  if [ $brc -ge 124 ]; then echo "bench killed (rc $brc)"; exit $brc; fi
done
exit $rc
