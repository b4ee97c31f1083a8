#!/bin/bash
# A/B of libplship builds on ONE box: the triangular-solve probe with each (PLSHIP_LIBRARY selects the build;
# tools/ab/libplship_<variant>.so are built by hand with -DPLS_STRIP_* flags, see csrc/chol.hip)
for lib in "" $PWD/tools/ab/libplship_*.so ""; do
  echo "== PLSHIP_LIBRARY=$lib"
  PLSHIP_LIBRARY=$lib python tools/r2_probe.py solve 2>&1 | grep -v amdgpu | awk 'NR==1 || ($1==1024 && $2==8192) || $1==4096 || ($1==1024 && $2==1024)'
done
