#!/bin/bash
# usage: tools/gpu_step.sh <seconds> <log file> <command ...>
# One GPU step of a gpurun call: the command under `timeout -k 10`, stdout + stderr to the log, its exit code appended.  Returns 0
# for an ordinary failure (an assertion, a non-zero exit: the following steps of the call still tell something) and the
# command's code if it was killed at its limit or by a signal (>= 124): then NO further GPU step may run in the call, so chain
# steps with `&&`.
secs=$1; log=$2; shift 2
timeout -k 10 "$secs" "$@" > "$log" 2>&1
rc=$?
echo "[gpu_step] rc=$rc: $*" >> "$log"
echo "[gpu_step] rc=$rc: $* (log: $log)"
if [ $rc -ge 124 ]; then exit $rc; fi
exit 0
