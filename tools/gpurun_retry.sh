#!/bin/bash
# retry wrapper around gpurun for "no box or slot free right now" (exit code 3: nothing charged)
#   tools/gpurun_retry.sh <timeout_s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@" > /tmp/gpurun_last.txt 2>&1
  rc=$?
  if grep -q "status=transient" /tmp/gpurun_last.txt || [ $rc -eq 3 ]; then sleep 90; continue; fi
  cat /tmp/gpurun_last.txt | tail -40
  exit $rc
done
echo "gave up after 40 attempts"; exit 3
