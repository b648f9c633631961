#!/bin/bash
# repeat the stream-equivalence test N times under the given environment; print the number of failures
N=${1:-10}; shift
f=0
for i in $(seq 1 $N); do
  env "$@" timeout 200 python -m pytest tests/test_gpu_streams_equivalence.py -x -q -m gpu -p no:cacheprovider > /tmp/se.log 2>&1 || { f=$((f+1)); grep -E "AssertionError|assert " /tmp/se.log | head -2; }
done
echo "env [$*]: $f failures of $N"
