set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3b
INFLIGHT_DUMP=1 timeout -k 10 120 python3 profiles/inflight.py --half decode --streams 4 > gpurun_out/r3b/dec4.txt 2>&1 || { echo dec4 failed; tail -5 gpurun_out/r3b/dec4.txt; exit 1; }
timeout -k 10 120 python3 profiles/inflight.py --half decode --streams 8 --input-sets 8 > gpurun_out/r3b/dec8_q4.txt 2>&1 || { echo dec8 failed; exit 1; }
GPU_MAX_HW_QUEUES=8 timeout -k 10 120 python3 profiles/inflight.py --half decode --streams 8 --input-sets 8 > gpurun_out/r3b/dec8_q8.txt 2>&1 || { echo dec8q8 failed; exit 1; }
timeout -k 10 120 python3 profiles/inflight.py --half decode --streams 2 > gpurun_out/r3b/dec2.txt 2>&1 || { echo dec2 failed; exit 1; }
timeout -k 10 120 python3 profiles/inflight.py --half match --streams 4 > gpurun_out/r3b/match4.txt 2>&1 || { echo m4 failed; exit 1; }
timeout -k 10 120 python3 profiles/inflight.py --half step --streams 4 > gpurun_out/r3b/step4.txt 2>&1 || { echo s4 failed; exit 1; }
head -20 gpurun_out/r3b/dec4.txt; for f in dec8_q4 dec8_q8 dec2 match4 step4; do echo == $f; cat gpurun_out/r3b/$f.txt; done
