#!/bin/bash
# One GPU call for a state of the code: the GPU tests, the default bench line and the profile
# summaries of profiles/collect.sh.   profiles/run_all.sh TAG
set -u
tag=$1
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/$tag
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/$tag/pytest.log 2>&1 || { echo tests failed; tail -40 gpurun_out/$tag/pytest.log; exit 1; }
tail -2 gpurun_out/$tag/pytest.log
timeout -k 10 600 python3 bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { echo bench failed; tail -5 gpurun_out/$tag/bench.err; exit 1; }
echo bench-done
bash profiles/collect.sh $tag || exit 1
