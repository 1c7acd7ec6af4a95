#!/bin/bash
# the three un-profiled bench lines once more (the committed PMC summaries now key the patch kernels as bench.py does: roofline.traffic)
tag=${1:-r5s2}
for wl in center vid16 wholeim; do
  timeout -k 10 420 python bench.py --workload $wl --steps 20 --warmup 5 > gpurun_out/${tag}_bench_$wl.json 2> gpurun_out/${tag}_bench_$wl.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/${tag}_bench_$wl.json')); r=d['roofline']; print('$wl', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r.get('traffic'), r.get('frac_of_pipe_bound'), d['cpu_baseline']['value'])"
done
