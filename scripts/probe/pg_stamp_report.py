"""Reader of VF_PG_STAMPS dumps (k_pconv_patch_g<0, true>, vf_pgemm.hip): per wave and step of the first 64 blocks four shader-clock
stamps — [0] after the step's barrier, [1] after the wave's DMA issue (and read-ahead), [2] after its last MFMA was issued, [3] after the
s_waitcnt in front of the NEXT barrier (written one step late) — as cycle counts per phase, medians over waves.
    python scripts/probe/pg_stamp_report.py <file>"""
import sys
import numpy as np

raw = open(sys.argv[1], "rb").read()
rec = 16 + 64 * 8 * 16 * 4 * 8
n = len(raw) // rec
for r in range(n):
    hdr = np.frombuffer(raw[r * rec:r * rec + 16], np.int32)
    st = np.frombuffer(raw[r * rec + 16:(r + 1) * rec], np.int64).reshape(64, 8, 16, 4)
    nblk = min(64, int(hdr[0]))
    st = st[:nblk]
    t0, t1, t2 = st[..., 0], st[..., 1], st[..., 2]
    t3 = st[:, :, 1:, 3]                       # slot 3 of step s + 1 holds "waited" of step s
    issue = (t1 - t0)[:, :, :15]
    compute = (t2 - t1)[:, :, :15]
    wait = t3 - t2[:, :, :15]
    barrier = t0[:, :, 1:] - t3
    step = t0[:, :, 1:] - t0[:, :, :15]
    print("dump %d: %d tiles, M %d N %d C %d; cycles per step, median over %d waves [p10 .. p90]" % (r, hdr[0], hdr[1], hdr[2], hdr[3], nblk * 8))
    for name, a in (("barrier -> DMAs issued", issue), ("reads + 24 MFMAs issued", compute), ("s_waitcnt (DMAs, reads)", wait), ("s_barrier", barrier), ("whole step", step)):
        per_t = [np.median(a[:, :, t::4]) for t in range(4)]
        print("  %-26s %7.0f  [%6.0f .. %6.0f]   by step of the unit (t = 0..3): %s" % (name, np.median(a), np.percentile(a, 10), np.percentile(a, 90), " ".join("%6.0f" % v for v in per_t)))
    lo, hi = (wait + barrier)[:, :4], (wait + barrier)[:, 4:]
    print("  wait + barrier, waves 0-3 %7.0f   waves 4-7 %7.0f" % (np.median(lo), np.median(hi)))
