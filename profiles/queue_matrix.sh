#!/bin/bash
# Throughput of the halves and the step against the number of batches in flight (streams) and of
# hardware queues the HIP runtime may use:  profiles/queue_matrix.sh  (on the GPU box)
cd "$(dirname "$0")/.."
for q in 4 8 16; do
  for s in 2 4 6 8 12; do
    echo "queues=$q streams=$s"
    GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python profiles/halves.py --streams $s --steps 300 || exit 1
  done
done
