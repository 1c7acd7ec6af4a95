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
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
names = {"center": ("a", "center_b64"), "vid16": ("b", "vid16_b16"), "wholeim": ("c", "wholeim_b4")}
for wl, (letter, sfx) in names.items():
    shutil.copy(os.path.join(G, "%s_%s_kernel_stats.csv" % (tag, wl)), os.path.join(P, "%s_%s_kernel_stats_%s.csv" % (rnd, letter, sfx)))
    shutil.copy(os.path.join(G, "%s_%s_bench_under_rocprof.json" % (tag, wl)), os.path.join(P, "%s_%s_bench_%s_under_rocprof.json" % (rnd, letter, wl)))
    shutil.copy(os.path.join(G, "%s_bench_%s.json" % (tag, wl)), os.path.join(P, "%s_%s_bench_%s_unprofiled.json" % (rnd, letter, wl)))
if os.path.exists(os.path.join(G, "drift_report.json")):
    shutil.copy(os.path.join(G, "drift_report.json"), os.path.join(P, "%s_e_drift_three_iterations.json" % rnd))
head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "pmc_bench_traffic.py"), os.path.join(G, "pmc_bench"),
                       os.path.join(P, "%s_pmc_bench_traffic.json" % rnd)], env=dict(os.environ, VF_GIT_HEAD=head), stdout=subprocess.DEVNULL)
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
