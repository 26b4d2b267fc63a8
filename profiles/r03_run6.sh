set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3f
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3f/pytest.log 2>&1 || { echo tests failed; tail -40 gpurun_out/r3f/pytest.log; exit 1; }
tail -2 gpurun_out/r3f/pytest.log
export SSDBOX_LIB=$PWD/face_detection_amd/libssdbox_trace.so
o=gpurun_out/r3f/fused.txt
: > $o
for f in 1 0; do
  echo "=== FUSED $f" >> $o
  SSDBOX_FUSED_TAIL=$f INFLIGHT_DUMP=1 timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 4 2>&1 | grep -v amdgpu.ids | head -40 >> $o || { echo "failed"; tail -5 $o; exit 1; }
  SSDBOX_FUSED_TAIL=$f timeout -k 10 120 python3 profiles/trace_inflight.py --half decode --streams 1 2>&1 | grep -v amdgpu.ids >> $o || { echo "failed"; exit 1; }
  SSDBOX_FUSED_TAIL=$f timeout -k 10 120 python3 profiles/trace_inflight.py --half step --streams 4 2>&1 | grep -v amdgpu.ids >> $o || { echo "failed"; exit 1; }
done
cat $o
