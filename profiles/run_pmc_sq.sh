#!/bin/bash
# Instruction-mix and wave-occupancy counters for the bench step, two SQ passes of at most
# eight counters each.  Usage: profiles/run_pmc_sq.sh OUTDIR   (EXTRA="--stress" for bench flags)
set -u
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
EXTRA="${EXTRA:-}"
pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --profile-only --steps 6 --warmup 2 $EXTRA > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$out/$name.log"; return 1; }
}
pass mix SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES && \
pass act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS && echo pmc-sq-done
