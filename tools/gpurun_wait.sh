#!/bin/bash
# Developer aid: call gpurun and, while it answers "no slot free" (exit code 3, nothing charged), wait and ask again.
# usage: tools/gpurun_wait.sh <timeout_s> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 75
done
exit 3
