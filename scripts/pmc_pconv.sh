#!/bin/bash
# Cache and SQ counters of the planes GEMM beside the in-kernel-split kernel on one layer (scripts/bench_pconv.py, ONLY=E3):
# separate --pmc passes, kernel-trace only.  Run on the GPU box from the repo root: bash scripts/pmc_pconv.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export ONLY=${ONLY:-E3}
export NB=${NB:-4}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_pconv
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/scripts/bench_pconv.py 64 > $OUT/p$i.log 2>&1 || echo "pass $i ($set) failed"
done
python3 - <<PY > $ROOT/gpurun_out/pmc_pconv_summary.txt
import csv, glob, os, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join("$OUT", "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:60], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "pconv" not in k[0]:
        continue
    print(k)
    for n in sorted(acc[k]):
        v = acc[k][n]
        print("    %-40s %16.0f  (n=%d)" % (n, sum(v) / len(v), len(v)))
    m = {n: sum(v) / len(v) for n, v in acc[k].items()}
    if m.get("SQ_BUSY_CU_CYCLES"):      # MFMA_BUSY counts per SIMD (4 per CU), BUSY_CU per CU
        print("    => matrix pipe busy %.1f %% of the CU-busy cycles" % (100.0 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * m["SQ_BUSY_CU_CYCLES"])))
    if m.get("TCC_HIT_sum") is not None and m.get("TCC_MISS_sum") is not None and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
        print("    => L2 hit rate %.1f %%; HBM read (FETCH_SIZE KiB x2, gfx950) %.1f MB" % (100.0 * m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 2 * m.get("FETCH_SIZE", 0) * 1024 / 1e6))
PY
cat $ROOT/gpurun_out/pmc_pconv_summary.txt
