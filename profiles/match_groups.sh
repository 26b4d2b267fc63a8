set -u
cd "$GRAFT_REPO_ROOT"
for g in 1 2 3 4 5 6 8; do
  echo "=== SSDBOX_MATCH_GROUPS $g"
  SSDBOX_MATCH_GROUPS=$g timeout -k 10 200 python3 profiles/halves.py --steps 2000 --no-stress-leg 2>&1 | tail -1
  SSDBOX_MATCH_GROUPS=$g timeout -k 10 200 python3 profiles/kt.py 2>&1 | tail -2
done
