#!/bin/bash
# sweep of Adam launch variants on one box: NT x blocks; prints the kernel's GB/s from bench.py's table
for rep in 1 2; do
for nt in 0 1; do for bl in 1024 2048 4096 8192 16384; do
  VF_ADAM_NT=$nt VF_ADAM_BLOCKS=$bl timeout -k 10 120 python bench.py --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); a = d['kernels']['adam']
        print('NT=$nt blocks=$bl', 'adam_ms=%.4f' % a['ms_per_step'], 'GB/s=%s' % a['gbs'], 'step=%.4f' % d['ms_per_step'])
" || exit 1
done; done; done
