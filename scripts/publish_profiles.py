"""Copy one GPU-box visit of scripts/collect_evidence.sh into profiles/ under this round's names and rebuild the PMC traffic
summary from the raw counter passes (tied to the digest of csrc/).   python scripts/publish_profiles.py <tag> [round]"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
names = {"center": ("a", "center_b64"), "vid16": ("b", "vid16_b16"), "wholeim": ("c", "wholeim_b4")}
for wl, (letter, sfx) in names.items():
    shutil.copy(os.path.join(G, "%s_%s_kernel_stats.csv" % (tag, wl)), os.path.join(P, "%s_%s_kernel_stats_%s.csv" % (rnd, letter, sfx)))
    shutil.copy(os.path.join(G, "%s_%s_bench_under_rocprof.json" % (tag, wl)), os.path.join(P, "%s_%s_bench_%s_under_rocprof.json" % (rnd, letter, wl)))
    shutil.copy(os.path.join(G, "%s_bench_%s.json" % (tag, wl)), os.path.join(P, "%s_%s_bench_%s_unprofiled.json" % (rnd, letter, wl)))
import glob
for f in glob.glob(os.path.join(G, "drift_report_*.json")):
    shutil.copy(f, os.path.join(P, "%s_e_%s" % (rnd, os.path.basename(f))))
if os.path.exists(os.path.join(G, "kinksync_stats.jsonl")):
    rows = [json.loads(l) for l in open(os.path.join(G, "kinksync_stats.jsonl"))]
    fr = sorted(r["frac"] for r in rows)
    json.dump(dict(runs=len(rows), rewritten_share=dict(min=fr[0], median=fr[len(fr) // 2], max=fr[-1]),
                   worst_difference_of_a_rewritten_element_rel_to_tensor_max=max(r["worst_near"] for r in rows),
                   note="tests/helpers.py KinkSync over one -m gpu run: share of activations the pin rewrote, and how far the rewritten "
                        "values were from the oracle's before the rewrite (guard: 1e-4)"),
              open(os.path.join(P, "%s_e_kinksync_stats.json" % rnd), "w"), indent=1)
head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()
for wl in names:
    suffix = "" if wl == "center" else "_" + wl
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "pmc_bench_traffic.py"), os.path.join(G, "pmc_bench_" + wl),
                           os.path.join(P, "%s_pmc_bench_traffic%s.json" % (rnd, suffix))],
                          env=dict(os.environ, VF_GIT_HEAD=head, VF_PMC_WORKLOAD=wl), stdout=subprocess.DEVNULL)
for wl in names:
    u = json.load(open(os.path.join(G, "%s_bench_%s.json" % (tag, wl))))
    p = json.load(open(os.path.join(G, "%s_%s_bench_under_rocprof.json" % (tag, wl))))
    r = u["roofline"]
    print("%-8s %9.1f %s  %.4f ms | under rocprof %9.1f  %.4f ms | %s  avg %.2f us (rocprof run %.2f)  %.1f %s  frac %.3f" % (
        wl, u["value"], u["unit"], u["ms_per_step"], p["value"], p["ms_per_step"], r["kernel"], r["avg_launch_us"],
        p["roofline"]["avg_launch_us"], r["achieved"], r["unit"], r["frac"]))
rows = list(csv.DictReader(open(os.path.join(G, "%s_center_kernel_stats.csv" % tag))))
for row in rows[:4]:
    print("   csv: %-60s calls %6s  avg %8.2f us  %5s %%" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"]))
t = json.load(open(os.path.join(P, "%s_pmc_bench_traffic.json" % rnd)))
print("traffic summary: git %s  csrc %s" % (t["git_head"], t["csrc_sha256"][:12]))
