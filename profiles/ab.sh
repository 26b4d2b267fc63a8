#!/bin/bash
# A/B of library builds on one box: profiles/ab.sh "libA.so libB.so" [rounds] [bench args]
# alternates the builds, prints step / match / decode images per second of every run.
libs=$1; rounds=${2:-3}; shift; shift
for r in $(seq $rounds); do
  for lib in $libs; do
    SSDBOX_LIB=$PWD/face_detection_amd/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --seeds 1 --streams 4 --steps 1000 "$@" > gpurun_out/tmp_b.json 2>gpurun_out/tmp_b.err || { tail -3 gpurun_out/tmp_b.err; exit 1; }
    python3 -c "
import json,sys; d=json.load(open('gpurun_out/tmp_b.json')); print('%-22s step %7.0f match %8.0f decode %8.0f ' % (sys.argv[1], d['value'], d['match_encode_images_per_s'], d['decode_nms_images_per_s']), {k: round(v,1) for k,v in d['kernel_us'].items()})" $lib
  done
done
