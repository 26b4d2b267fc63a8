set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3g
SSDBOX_LIB=$PWD/face_detection_amd/libssdbox_stamps.so timeout -k 10 120 python3 profiles/fused_stamps.py 2>&1 | grep fused | tail -4
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3
export SSDBOX_LIB=$PWD/face_detection_amd/libssdbox_trace.so
timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 4 2>&1 | grep -v amdgpu.ids
timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 1 2>&1 | grep -v amdgpu.ids
