set -u
cd "$GRAFT_REPO_ROOT"
B=profiles/micro/chain_stamps
run() { timeout -k 5 60 "$@" 2>&1 | grep -v "reads running" || { echo "failed: $*"; exit 1; }; }
for mode in 0 1 2 3 4 6; do
  run $B 1 "" 300 1 512 83000 $mode
done
for mode in 0 2 4 6; do
  run $B 1 "" 300 1 256 83000 $mode
  run $B 1 "" 300 1 1024 83000 $mode
done
run $B 1 "" 300 0 256 0 1
run $B 1 "" 300 0 256 0 0
T="32,1024,38"
for mode in 0 2 4 6; do run $B 4 "$T" 300 1 512 83000 $mode; done
