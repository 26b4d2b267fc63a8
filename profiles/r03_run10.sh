set -u
cd "$GRAFT_REPO_ROOT"
export SSDBOX_LIB=$PWD/face_detection_amd/libssdbox_trace.so
export SSDBOX_EXP_DIAG=1
for cfg in "8 -1,0" "8 -1,0,1" "6 -1,0" "12 -1,0,1" "8 -1,-1,-1,-1,0,0,0,0"; do
  set -- $cfg
  echo "=== streams $1 priorities $2"
  timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams $1 --input-sets $1 --stream-priority-list=$2 2>&1 | grep -E "wall per batch|select  |topk  |streaming|gap"
done
for q in 5 6; do
  echo "=== GPU_MAX_HW_QUEUES $q, streams $q"
  GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams $q --input-sets $q 2>&1 | grep -E "wall per batch|select  |topk  |streaming|gap"
done
