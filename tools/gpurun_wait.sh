#!/bin/bash
# gpurun_wait.sh LOG TIMEOUT 'command'  — runs one gpurun call; when no box / slot is free (exit 3: nothing ran, nothing was
# charged) waits two minutes and asks again, up to 15 times.  Any other outcome (success or failure of the command) ends it.
LOG=$1; TO=$2; CMD=$3
for i in $(seq 1 15); do
  /usr/local/graft/bin/gpurun --timeout $TO -- "$CMD" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
