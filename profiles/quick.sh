#!/bin/bash
# Quick look at one state of the code on the GPU box: the decode / match parity tests, then the
# bench line without the CPU baseline at 4 and 8 batches in flight.  profiles/quick.sh TAG
tag=$1
out=gpurun_out/quick_$tag
mkdir -p "$out"
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q --timeout 120 --timeout-method thread > "$out/tests.txt" 2>&1 || { tail -30 "$out/tests.txt"; exit 1; }
tail -2 "$out/tests.txt"
for s in 4 8; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --seeds 1 --streams $s $2 > "$out/bench_s$s.json" 2> "$out/bench_s$s.err" || { tail -5 "$out/bench_s$s.err"; exit 1; }
  python3 - "$out/bench_s$s.json" $s <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("streams %s: step %.0f img/s  match %.0f  decode %.0f  thr0 %.0f  lat %.3f ms" % (sys.argv[2], d["value"], d["match_encode_images_per_s"], d["decode_nms_images_per_s"], d["decode_nms_thr0_images_per_s"], d["single_batch_latency_ms"]))
print("   kernel_us", d["kernel_us"])
PY
done
