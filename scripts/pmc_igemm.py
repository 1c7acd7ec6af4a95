"""Summarise scripts/pmc_igemm.sh: per kernel symbol, the mean of every collected SQ counter per launch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "igemm" not in k and "wgrad" not in k:
        continue
    print(k[:100])
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    for n in sorted(c):
        print("    %-32s %16.0f  (n=%d)" % (n, c[n], len(acc[k][n])))
    if "SQ_BUSY_CU_CYCLES" in c and c["SQ_BUSY_CU_CYCLES"]:
        b = c["SQ_BUSY_CU_CYCLES"]
        for n in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                  "SQ_ACTIVE_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
            if n in c:
                print("    %-32s / BUSY_CU_CYCLES = %.3f" % (n, c[n] / b))
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        w = c["SQ_WAVE_CYCLES"]
        for n in ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY"):
            if n in c:
                print("    %-32s / WAVE_CYCLES = %.3f" % (n, c[n] / w))
