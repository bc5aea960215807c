#!/bin/bash
# tools/run_guarded.sh <log> <timeout s> <command...>: runs one GPU step under `timeout -k 10`, output to the log;
# exits non-zero ONLY when the step was killed at its limit (124 / 137), so that `a && b` chains stop after a hang
# but go on after an ordinary test failure.
log=$1; lim=$2; shift 2
timeout -k 10 "$lim" "$@" > "$log" 2>&1
rc=$?
echo "[run_guarded] rc=$rc: $*" | tee -a "$log"
tail -n 3 "$log"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
exit 0
