#!/bin/bash
# Collects rocprofv3 PMC counters for the bench step, one pass per counter group (the
# hardware cannot hold FETCH_SIZE and WRITE_SIZE together).  Usage: profiles/run_pmc.sh OUTDIR [bench args]
set -u
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --profile-only --steps 6 --warmup 2 $EXTRA > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$out/$name.log"; return 1; }
}
EXTRA="${EXTRA:-}"
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY && \
pass fetch FETCH_SIZE && \
pass write WRITE_SIZE && \
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum && echo pmc-done
