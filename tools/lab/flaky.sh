#!/bin/bash
# how often does a test fail under each environment setting?   tools/lab/flaky.sh <pytest -k expression> <runs> "ENV=.." ...
K=$1; N=$2; shift 2
for v in "$@"; do
  f=0
  for i in $(seq $N); do
    env $v python -m pytest tests/test_train_gpu.py -q -m gpu -k "$K" 2>&1 | grep -q "failed" && f=$((f+1))
  done
  echo "[$v] $f failures in $N runs"
done
