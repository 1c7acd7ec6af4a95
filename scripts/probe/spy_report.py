"""Per-K-step phase stamps of one wave (block 7, thread 0) written by a -DVF_IGEMM_SPY build of vf_conv.hip
(hipcc ... -DVF_IGEMM_SPY -c csrc/vf_conv.hip, linked into an alternate .so selected with VF_HIP_LIB) under VF_IGEMM_STAMPS:
the dump's rows 256.. hold 32 steps x 8 slots of shader-clock stamps: 0 step top, 1 loads issued, 2 MFMAs issued,
3 past barrier 1, 4 split + LDS writes issued, 5 past barrier 2."""
import sys
import numpy as np
launches, cur, name, spy = [], [], None, {}
for line in open(sys.argv[1]):
    if line.startswith("#"):
        if cur: launches.append((name, cur))
        name, cur = line[1:].strip(), []
    else:
        if line.startswith("S"):
            spy.setdefault(name, []).append([int(v) for v in line.split()[1:]])
        else:
            cur.append([int(v) for v in line.split()[1:]])
if cur: launches.append((name, cur))
for name, rows in spy.items():
    a = np.array([r[:6] for r in rows[-32:]], dtype=np.int64)
    if (a == 0).any(): continue
    d = np.diff(a, axis=1)
    nxt = a[1:, 0] - a[:-1, 5]
    print(name)
    print("   per step (cycles, median over steps 2..30): loads-issue %d | frag reads + 12 MFMAs issued %d | barrier1 %d | wait loads + split + LDS writes %d | barrier2 %d | loop back %d | total %d"
          % tuple([int(np.median(d[2:30, i])) for i in range(5)] + [int(np.median(nxt[2:30])), int(np.median(a[3:31, 0] - a[2:30, 0]))]))
