#!/bin/bash
# tools/gpu_step.sh LIMIT_SECONDS LOGFILE command...
# Runs one GPU step under its own timeout, logs to gpurun_out/LOGFILE, prints
# the tail.  Exit status 0 unless the step timed out / was killed (so that
# later steps joined with && do not start after a hang); a plain failure of the
# step is reported but does not stop the chain.
limit=$1; log=gpurun_out/$2; shift 2
mkdir -p gpurun_out
echo "== $* (limit ${limit}s)" | tee "$log"
timeout -k 10 "$limit" "$@" >> "$log" 2>&1
rc=$?
tail -n 25 "$log"
echo "== exit $rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
exit 0
