#!/bin/bash
# kernel trace (start / end timestamps) of two replayed iterations of a workload -> gpurun_out/<tag>_kernel_trace.csv
tag=${1:-r5l}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$tag -- python3 $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 --step-stats 0 "$@" > $ROOT/gpurun_out/${tag}_trace_bench.json 2> $ROOT/gpurun_out/${tag}_trace.err
f=$(find /tmp/trace_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" $ROOT/gpurun_out/${tag}_kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-600:]          # the last replayed iterations
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    prev_end = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write("%10.2f %8.2f %7.2f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:110]))
        prev_end = e
PY
echo done
