#!/bin/bash
# The round's evidence in one visit: collect_evidence.sh (suite, PMC traffic, unprofiled lines, kernel statistics), the three lines in
# --mfma bf16 mode, smoke().      bash scripts/r5_visit_c.sh <tag> [round]
tag=${1:-r5z}; rnd=${2:-r05}
bash scripts/collect_evidence.sh $tag $rnd || exit 1
for wl in center vid16 wholeim; do
  timeout -k 10 300 python bench.py --workload $wl --mfma bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_${wl}_mfma_bf16.json 2> gpurun_out/${tag}_${wl}_mfma_bf16.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/${tag}_${wl}_mfma_bf16.json')); print('$wl bf16', d['value'], d['ms_per_step'], d['roofline'].get('mfma_view', d['roofline']).get('family_weighted'))"
done
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${tag}_smoke.log
