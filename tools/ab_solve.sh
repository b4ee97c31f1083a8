#!/bin/bash
# A/B of libplship builds on ONE box: the triangular-solve probe with each (PLSHIP_LIBRARY selects the build;
# tools/ab/libplship_<variant>.so are built by hand).  The -DPLS_STRIP_ABL_* / _AD / _INTERLEAVE ablation branches this was
# written for were removed from the product source in round 3; they live in csrc/chol.hip of commit 092b717 (round 2),
# whose measurements are profiles/r02_ab_solve_ablation.txt.
for lib in "" $PWD/tools/ab/libplship_*.so ""; do
  echo "== PLSHIP_LIBRARY=$lib"
  PLSHIP_LIBRARY=$lib python tools/r2_probe.py solve 2>&1 | grep -v amdgpu | awk 'NR==1 || ($1==1024 && $2==8192) || $1==4096 || ($1==1024 && $2==1024)'
done
