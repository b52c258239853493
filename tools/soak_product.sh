#!/bin/bash
# Soak of the one-launch product's in-launch hand-off on the GPU box: the product test file N times over, then the product fuzz
# over many seeds.  Usage: bash tools/soak_product.sh <repeats> <first seed> <seeds>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for i in $(seq 1 ${1:-10}); do
  timeout -k 10 200 python -m pytest tests/test_product.py -m gpu -x -q 2>&1 | tail -1
done
timeout -k 10 900 python tools/soak_fuzz.py ${2:-1000} ${3:-1000} product 2>&1 | tail -3
