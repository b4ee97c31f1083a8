#!/bin/bash
# A/B of two libplship builds on ONE box: the rank sweep around 128 with each (PLSHIP_LIBRARY selects the build)
for lib in "$PWD/tools/ab/libplship_nofence.so" ""; do
  echo "== PLSHIP_LIBRARY=$lib"
  PLSHIP_LIBRARY=$lib python tools/r2_probe.py ranks 2>&1 | grep -v amdgpu | awk 'NR==1 || $2==129 || $2==160 || $2==192 || $2==256'
done
