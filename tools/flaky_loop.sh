#!/bin/bash
# repeats one GPU test under environment settings and counts failures: tools/flaky_loop.sh <pytest -k expression> <repeats> [VAR=value ...]
K="${1:-more_than_16k}"; N="${2:-12}"; shift 2 || true
fails=0
for i in $(seq 1 "$N"); do
  if ! env "$@" timeout 120 python -m pytest tests -m gpu -x -q -k "$K" > /tmp/flaky.log 2>&1; then fails=$((fails+1)); fi
done
echo "$K [$*]: $fails / $N failed"
