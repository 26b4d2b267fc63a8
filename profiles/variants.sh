#!/bin/bash
# Times the bench step with each experiment build in face_detection_amd/variants/ (SSDBOX_LIB
# picks the library).  Usage: profiles/variants.sh OUTFILE [bench args]
out=$1; shift
: > "$out"
for lib in face_detection_amd/libssdbox.so face_detection_amd/variants/*.so; do
  echo "== $lib" >> "$out"
  SSDBOX_LIB=$PWD/$lib timeout -k 10 120 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep -o "\"kernel_us\": {[^}]*}\|\"match_encode_images_per_s\": [0-9.]*" >> "$out" || exit 1
done
