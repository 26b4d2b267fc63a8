set -u
cd "$GRAFT_REPO_ROOT"
B=profiles/micro/chain_stamps
run() { timeout -k 5 60 "$@" 2>&1 || { echo "failed: $*"; exit 1; }; }
T="32,1024,38"
run $B 1 "" 300 1 512 83000
run $B 1 "" 300 1 1024 83000
run $B 4 "" 300 1 512 83000
run $B 4 "" 300 1 1024 83000
echo "--- tail 38, shared vs exclusive select"
run $B 4 "$T"
run $B 4 "$T" 300 1 512 83000
run $B 4 "$T" 300 1 1024 83000
run $B 4 "$T" 300 1 1024 0
echo "--- tail 15;4;3"
T="32,1024,15;4352,64,4;32,1024,3"
run $B 4 "$T"
run $B 4 "$T" 300 1 512 83000
run $B 4 "$T" 300 1 1024 83000
echo "--- tail 25"
T="32,1024,25"
run $B 4 "$T"
run $B 4 "$T" 300 1 1024 83000
