set -u
cd "$GRAFT_REPO_ROOT"
export SSDBOX_FUSED_TAIL=0
export SSDBOX_SEL_EXCLUSIVE=0
export SSDBOX_LIB=$PWD/face_detection_amd/libssdbox_trace.so
for cfg in "2 1 82432 512" "3 1 82432 512" "4 1 82432 512" "2 1 82432 256" "4 1 82432 256" "8 1 82432 256" "0 1 82432 256" "0 1 82432 512"; do
  set -- $cfg
  echo "=== static SEL_WGS $1 DEPTH $2 LDS $3 BLOCK $4"
  export SSDBOX_SEL_WGS=$1 SSDBOX_SEL_DEPTH=$2 SSDBOX_SEL_LDS=$3 SSDBOX_SEL_BLOCK=$4
  timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 4 2>&1 | grep -E "wall per batch|select  |topk  |nms_mask  |nms_scan  |streaming|chain"
  timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 1 2>&1 | grep -E "select  "
done
