#!/bin/bash
# tools/gpu.sh with retries while no GPU slot / box is free (gpurun exit code 3: nothing charged).
for i in $(seq 1 30); do
  "$(dirname "$0")/gpu.sh" "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
