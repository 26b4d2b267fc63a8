set -u
cd "$GRAFT_REPO_ROOT"
export SSDBOX_FUSED_TAIL=0
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3
export SSDBOX_LIB=$PWD/face_detection_amd/libssdbox_trace.so
for cfg in "512" "256" "1024"; do
  echo "=== exclusive claim select, BLOCK $cfg"
  export SSDBOX_SEL_BLOCK=$cfg
  timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 4 2>&1 | grep -E "wall per batch|select  |topk  |nms_mask  |nms_scan  |streaming|gap|chain"
  timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 1 2>&1 | grep -E "wall per batch|select  "
  timeout -k 10 120 python3 profiles/trace_inflight.py --half step --streams 4 2>&1 | grep -E "wall per batch|select  |match  |streaming"
done
