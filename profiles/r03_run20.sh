set -u
cd "$GRAFT_REPO_ROOT"
B=profiles/micro/chain_stamps
run() { timeout -k 5 60 "$@" 2>&1 || { echo "failed: $*"; exit 1; }; }
CHAIN_TICKET=1 run $B 1 "" 300 1 256
CHAIN_TICKET=1 run $B 4 "" 300 1 256
for T in "32,1024,38" "32,1024,15;4352,64,4;32,1024,3" "32,1024,25" "32,1024,20;4352,64,10;32,1024,6"; do
  echo "--- tails $T"
  run $B 4 "$T"
  CHAIN_TICKET=1 run $B 4 "$T" 300 1 256
  CHAIN_TICKET=1 run $B 4 "$T" 300 2 256
  CHAIN_TICKET=1 run $B 4 "$T" 300 1 128
done
