#!/bin/bash
# Soak run on the GPU box: the randomised parity tests with shifted seeds (tests/test_gpu_parity.py, PLS_FUZZ_SEED).
# usage: tools/soak.sh <first seed> <last seed>; stops at the first failing seed.
for s in $(seq ${1:-1} ${2:-4}); do
  echo "== PLS_FUZZ_SEED=$s"
  PLS_FUZZ_SEED=$s timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fuzz or random_shape" 2>&1 | tail -4 || exit 1
done
