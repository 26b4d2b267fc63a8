#!/bin/bash
# Collects everything profiles/ keeps for one state of the code, on the GPU box:
#   profiles/collect.sh TAG      (e.g. r01_v3)
# 1. rocprofv3 --kernel-trace --stats of the bench step (kernels back to back, one stream)
# 2. the PMC passes of run_pmc.sh and run_pmc_sq.sh (each pass its own run)
# 3. HBM traffic per launch of the streaming kernels -> profiles/traffic.json
# Everything lands under gpurun_out/collect_TAG/ (the only directory a GPU run brings back); copy
# TAG_kernel_stats.csv, TAG_timeline.txt, TAG_pmc_summary.txt, TAG_stress_* and traffic.json into profiles/.
set -u
tag=$1
here=$(cd "$(dirname "$0")" && pwd)
root=$(dirname "$here")
cd "$root" || exit 1
export TMPDIR=/tmp
out=gpurun_out/collect_$tag
mkdir -p "$out"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --profile-only --steps 50 --warmup 10 > "$out/stats.log" 2>&1 || { echo "stats run failed"; tail -3 "$out/stats.log"; exit 1; }
cp "$(ls "$out"/stats/*/*kernel_stats.csv | head -1)" "$out/${tag}_kernel_stats.csv"
python3 "$here/timeline.py" "$out/stats" > "$out/${tag}_timeline.txt"
bash "$here/run_pmc.sh" "$out/pmc" || exit 1
bash "$here/run_pmc_sq.sh" "$out/pmc" || exit 1
python3 "$here/summarize_pmc.py" "$out/pmc" > "$out/${tag}_pmc_summary.txt"
python3 "$here/make_traffic.py" "$out/pmc" "$out/traffic.json"
# 4. the same for the stress batch (256 ground-truth boxes in every image): kernel stats and the SQ passes
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_stress" -- python3 bench.py --profile-only --stress --steps 50 --warmup 10 > "$out/stats_stress.log" 2>&1 || { echo "stress stats run failed"; tail -3 "$out/stats_stress.log"; exit 1; }
cp "$(ls "$out"/stats_stress/*/*kernel_stats.csv | head -1)" "$out/${tag}_stress_kernel_stats.csv"
EXTRA="--stress" bash "$here/run_pmc.sh" "$out/pmc_stress" || exit 1
python3 "$here/summarize_pmc.py" "$out/pmc_stress" > "$out/${tag}_stress_pmc_summary.txt"
echo collect-done
