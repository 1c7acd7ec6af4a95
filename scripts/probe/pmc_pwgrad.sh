#!/bin/bash
# SQ counters of k_pwgrad_group on the layers of scripts/bench_pwgrad.py (separate --pmc passes, kernel-trace only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export ONLY=${ONLY:-E3,C2@2B} NB=4
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_pwgrad
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/scripts/bench_pwgrad.py 64 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY > $ROOT/gpurun_out/pmc_pwgrad_summary.txt
import csv, glob, os, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join("$OUT", "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:60], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "wgrad" not in k[0]: continue
    print(k)
    m = {n: sum(v) / len(v) for n, v in acc[k].items()}
    for n in sorted(m): print("    %-34s %16.0f" % (n, m[n]))
    if m.get("SQ_BUSY_CU_CYCLES"): print("    => matrix pipe busy %.1f %% of CU-busy cycles" % (100.0 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * m["SQ_BUSY_CU_CYCLES"])))
    if m.get("SQ_LDS_IDX_ACTIVE"): print("    => LDS bank conflict cycles / LDS active cycles %.3f" % (m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]))
PY
cat $ROOT/gpurun_out/pmc_pwgrad_summary.txt
