set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3d
export SSDBOX_LIB=$PWD/face_detection_amd/libssdbox_trace.so
INFLIGHT_DUMP=1 timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 4 > gpurun_out/r3d/dec4.txt 2>&1 || { echo dec4 failed; tail -5 gpurun_out/r3d/dec4.txt; exit 1; }
timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 8 --input-sets 8 > gpurun_out/r3d/dec8.txt 2>&1 || { echo dec8 failed; exit 1; }
timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 2 > gpurun_out/r3d/dec2.txt 2>&1 || { echo dec2 failed; exit 1; }
timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 1 > gpurun_out/r3d/dec1.txt 2>&1 || { echo dec1 failed; exit 1; }
timeout -k 10 120 python3 profiles/trace_inflight.py --half match --streams 4 > gpurun_out/r3d/match4.txt 2>&1 || { echo m4 failed; exit 1; }
timeout -k 10 120 python3 profiles/trace_inflight.py --half step --streams 4 > gpurun_out/r3d/step4.txt 2>&1 || { echo s4 failed; exit 1; }
for f in dec4 dec8 dec2 dec1 match4 step4; do echo == $f; grep -v amdgpu.ids gpurun_out/r3d/$f.txt; done
