#!/bin/bash
# grid / workgroup sizes of every launch of one replayed iteration -> gpurun_out/<tag>_grids.txt (name, blocks, threads, us)
tag=${1:-r5g}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$tag -- python3 $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 --step-stats 0 "$@" > /dev/null 2> $ROOT/gpurun_out/${tag}_trace.err
f=$(find /tmp/trace_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" $ROOT/gpurun_out/${tag}_grids.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_adam_fused_multi" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
with open(sys.argv[2], "w") as f:
    for r in rows[a + 1:b + 1]:
        wg = int(r["Workgroup_Size"]) if "Workgroup_Size" in r else int(r["Workgroup_Size_X"])
        gs = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])
        f.write("%7d %4d %8.2f  %s\n" % (gs // wg, wg, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:100]))
PY
echo done
