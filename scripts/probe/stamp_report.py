"""Summarise the per-block stamps k_igemm writes under VF_IGEMM_STAMPS=<file> (slots 0-3: 100 MHz wall clock at block
start / first tile in LDS / loop end / stores retired; slots 4-5: shader cycle counter at start / end)."""
import sys
import numpy as np
launches, cur, name = [], [], None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        if cur: launches.append((name, np.array(cur, dtype=np.int64)))
        name, cur = line[1:].strip(), []
    else:
        cur.append([int(v) for v in line.split()[1:]])
if cur: launches.append((name, np.array(cur, dtype=np.int64)))
q = lambda x: "%.1f/%.1f/%.1f" % tuple(np.percentile(x, [10, 50, 90]))
for name, a in launches:
    t = (a[:, :4] - a[:, 0].min()) * 0.01          # us since the first block started
    start, pro, loop, end = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    print(name)
    print("   kernel span %.1f us | block start p10/50/90 %s max %.1f" % (end.max(), q(start), start.max()))
    print("   prologue %s | main loop %s | epilogue (incl. stores retired) %s" % (q(pro - start), q(loop - pro), q(end - loop)))
    if a.shape[1] >= 6:
        mhz = (a[:, 5] - a[:, 4]) / ((a[:, 3] - a[:, 0]) * 0.01)
        print("   shader cycle counter while the block ran: p10/50/90 %.0f/%.0f/%.0f MHz" % tuple(np.percentile(mhz, [10, 50, 90])))
