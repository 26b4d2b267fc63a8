set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3e
export SSDBOX_LIB=$PWD/face_detection_amd/libssdbox_trace.so
o=gpurun_out/r3e/sel.txt
: > $o
for cfg in "0 1" "2 2" "2 3" "3 2" "3 3" "4 1" "4 2" "4 3" "6 2" "8 1" "8 2"; do
  set -- $cfg
  echo "=== SEL_WGS $1 DEPTH $2" >> $o
  SSDBOX_SEL_WGS=$1 SSDBOX_SEL_DEPTH=$2 timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 4 2>&1 | grep -v amdgpu.ids >> $o || { echo "failed $cfg"; tail -5 $o; exit 1; }
  SSDBOX_SEL_WGS=$1 SSDBOX_SEL_DEPTH=$2 timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 1 2>&1 | grep -E "wall per batch|select " >> $o || { echo "failed $cfg"; exit 1; }
done
cat $o
