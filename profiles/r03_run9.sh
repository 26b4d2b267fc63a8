set -u
cd "$GRAFT_REPO_ROOT"
export SSDBOX_LIB=$PWD/face_detection_amd/libssdbox_trace.so
export SSDBOX_EXP_DIAG=1
python3 -c "import torch; print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
for pl in "-1,0" "-1,0,1" "-1,-1,0,0" "0,0,0,0" "-1,0,1,1" "-1,-1,-1,0"; do
  echo "=== priorities $pl"
  timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 4 --stream-priority-list=$pl 2>&1 | grep -E "wall per batch|select  |topk  |streaming"
done
echo "=== 3 streams"; timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 3 2>&1 | grep -E "wall per batch|select  |topk  |streaming"
echo "=== 3 streams prio"; timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 3 --stream-priority-list=-1,0,1 2>&1 | grep -E "wall per batch|select  |topk  |streaming"
echo "=== 5 streams"; timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 5 --input-sets 5 2>&1 | grep -E "wall per batch|select  |topk  |streaming"
echo "=== 6 streams"; timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 6 --input-sets 6 2>&1 | grep -E "wall per batch|select  |topk  |streaming"
unset SSDBOX_EXP_DIAG
export SSDBOX_FUSED_TAIL=0
for pl in "-1,0,1" "-1,-1,0,0" "-1,0,1,1"; do
  echo "=== unfused priorities $pl"
  timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 4 --stream-priority-list=$pl 2>&1 | grep -E "wall per batch|select  |topk  |nms_mask  |streaming"
done
