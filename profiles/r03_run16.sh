set -u
cd "$GRAFT_REPO_ROOT"
export SSDBOX_FUSED_TAIL=0
for lib in trace trace4; do
export SSDBOX_LIB=$PWD/face_detection_amd/libssdbox_$lib.so
for cfg in "512" "256"; do
  echo "=== exclusive queue select, lib $lib BLOCK $cfg"
  export SSDBOX_SEL_BLOCK=$cfg
  timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 4 2>&1 | grep -E "wall per batch|select  |topk  |nms_mask  |nms_scan  |streaming|gap|chain"
  timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 1 2>&1 | grep -E "wall per batch|select  "
  timeout -k 10 120 python3 profiles/trace_inflight.py --half step --streams 4 2>&1 | grep -E "wall per batch|select  |match  |streaming"
done
done
