#!/bin/bash
# Everything the committed profiles/ of a round are made of, in ONE GPU-box visit (run through gpurun from the repo root):
#   bash scripts/collect_evidence.sh <tag> [round, default r04]
# -m gpu suite + the three un-profiled bench lines (gpu_round.sh), the two PMC passes for HBM traffic of each workload, the rocprofv3 kernel
# statistics of the three workloads.  scripts/publish_profiles.py <tag> then copies the results into profiles/ (run locally).
tag=${1:-x}
rnd=${2:-r04}
bash scripts/gpu_round.sh $tag
for wl in center vid16 wholeim; do
  timeout -k 10 420 bash scripts/pmc_bench_traffic.sh $wl > gpurun_out/${tag}_pmc_$wl.log 2>&1; echo "pmc $wl rc=$?"
done
# the un-profiled bench lines once more, now that THIS build's counter summaries exist (bench.py refuses a summary of another build of
# csrc/: the lines gpu_round.sh took above would carry "traffic": null whenever the kernels changed since the last visit)
for wl in center vid16 wholeim; do
  suffix=""; [ "$wl" != center ] && suffix="_$wl"
  [ -f gpurun_out/pmc_bench_traffic$suffix.json ] && cp gpurun_out/pmc_bench_traffic$suffix.json profiles/${rnd}_pmc_bench_traffic$suffix.json
  timeout -k 10 420 python bench.py --workload $wl --steps 20 --warmup 5 > gpurun_out/${tag}_bench_$wl.json 2> gpurun_out/${tag}_bench_$wl.err || exit 1
done
timeout -k 10 200 bash scripts/prof_bench.sh ${tag}_center > /dev/null || exit 1
timeout -k 10 200 bash scripts/prof_bench.sh ${tag}_vid16 --workload vid16 > /dev/null || exit 1
timeout -k 10 200 bash scripts/prof_bench.sh ${tag}_wholeim --workload wholeim > /dev/null || exit 1
echo "evidence collected under gpurun_out/${tag}_*"
