set -u
cd "$GRAFT_REPO_ROOT"
B=profiles/micro/chain_stamps
run() { timeout -k 5 60 "$@" 2>&1 || { echo "failed: $*"; exit 1; }; }
for T in "32,1024,38" "32,1024,15;4352,64,4;32,1024,3" "32,1024,25"; do
  echo "--- tails $T"
  run $B 4 "$T"
  CHAIN_FIFO=1 run $B 4 "$T"
  CHAIN_FIFO=1 run $B 3 "$T"
  CHAIN_FIFO=1 run $B 6 "$T"
done
