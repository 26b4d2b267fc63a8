set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3a/pytest.log 2>&1 || { echo tests failed; tail -30 gpurun_out/r3a/pytest.log; exit 1; }
tail -2 gpurun_out/r3a/pytest.log
timeout -k 10 400 python3 bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err || { echo bench failed; tail -5 gpurun_out/r3a/bench.err; exit 1; }
echo bench-done
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3a/tr4 -- python3 bench.py --profile-only --profile-half decode --profile-streams 4 --steps 400 --warmup 50 > gpurun_out/r3a/tr4.log 2>&1 || { echo trace4 failed; tail -3 gpurun_out/r3a/tr4.log; exit 1; }
python3 profiles/chain_trace.py gpurun_out/r3a/tr4 select > gpurun_out/r3a/chain4.txt
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3a/tr8 -- python3 bench.py --profile-only --profile-half decode --profile-streams 8 --steps 400 --warmup 50 > gpurun_out/r3a/tr8.log 2>&1 || { echo trace8 failed; tail -3 gpurun_out/r3a/tr8.log; exit 1; }
python3 profiles/chain_trace.py gpurun_out/r3a/tr8 select > gpurun_out/r3a/chain8.txt
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3a/trm -- python3 bench.py --profile-only --profile-half match --profile-streams 4 --steps 400 --warmup 50 > gpurun_out/r3a/trm.log 2>&1 || { echo tracem failed; exit 1; }
python3 profiles/chain_trace.py gpurun_out/r3a/trm match > gpurun_out/r3a/chainm.txt
cat gpurun_out/r3a/chain4.txt gpurun_out/r3a/chain8.txt gpurun_out/r3a/chainm.txt
