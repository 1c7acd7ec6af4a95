"""Idle time between kernels inside the replayed HIP graph: rocprofv3 kernel trace of `bench.py`, dispatches of the timed region
(the densest run of back-to-back steps), sum of kernel durations against the span they cover.
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/gap -- python3 $ROOT/bench.py --no-cpu-baseline --step-stats 0 --steps 30
   python3 scripts/gap_analysis.py /tmp/gap"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]))
rows.sort()
# the timed region: the last 30 % of the dispatches are graph replays (warm-up and capture come first); take a window of them
n = len(rows)
w = rows[int(n * 0.55):int(n * 0.95)]
span = w[-1][1] - w[0][0]
busy = sum(e - s for s, e, _ in w)
# overlap-aware busy time (kernels of one stream do not overlap, but be safe)
cur_e, union = 0, 0
for s, e, _ in w:
    if s > cur_e:
        union += e - s
        cur_e = e
    elif e > cur_e:
        union += e - cur_e
        cur_e = e
gaps = sorted((w[i + 1][0] - w[i][1]) for i in range(len(w) - 1))
print("dispatches %d  span %.3f ms  sum of durations %.3f ms (%.1f %%)  union %.3f ms (%.1f %%)" % (
    len(w), span / 1e6, busy / 1e6, 100.0 * busy / span, union / 1e6, 100.0 * union / span))
print("gap between consecutive kernels: median %.2f us  p90 %.2f us  mean %.2f us" % (
    gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, sum(gaps) / len(gaps) / 1e3))
