set -u
cd "$GRAFT_REPO_ROOT"
export SSDBOX_LIB=$PWD/face_detection_amd/libssdbox_trace.so
export SSDBOX_FUSED_TAIL=0
for cfg in "1 1 82432 512" "1 2 82432 512" "1 1 82432 1024" "1 3 82432 512" "1 2 82432 768"; do
  set -- $cfg
  echo "=== SEL_WGS $1 DEPTH $2 LDS $3 BLOCK $4"
  export SSDBOX_SEL_WGS=$1 SSDBOX_SEL_DEPTH=$2 SSDBOX_SEL_LDS=$3 SSDBOX_SEL_BLOCK=$4
  timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 4 2>&1 | grep -E "wall per batch|select  |topk  |nms_mask  |nms_scan  |streaming|gap|chain"
  timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 1 2>&1 | grep -E "wall per batch|select  "
  timeout -k 10 120 python3 profiles/trace_inflight.py --half step --streams 4 2>&1 | grep -E "wall per batch|select  |match  |streaming"
done
