#!/bin/bash
# Retry ONE gpurun call until the pool accepts it (a refused call costs nothing), then exit: the caller runs this in
# the background and is woken when the call that got through has finished.  Runs the FROZEN tree (tools/freeze.sh).
#   bash tools/gpu_when_open.sh <timeout_s> <stages...>
T=${1:-2700}; shift
STAGES=${@:-tests smoke bench}
LOG=/tmp/gpu_when_open.log
while true; do
  /usr/local/graft/bin/gpurun --timeout $T -- "cd .frozen && rm -rf gpurun_out && ln -s \$GRAFT_REPO_ROOT/gpurun_out gpurun_out && cat FROZEN_HEAD && NSPARSE_TAG=${NSPARSE_TAG:-r06} bash tools/final_r05.sh $STAGES" > $LOG 2>&1
  if grep -qE "status=refused|no box|rc=3|exit code 3" $LOG && ! grep -q "#### stage" $LOG; then
    date +"%T refused" >> /tmp/gpu_when_open.hist
    sleep ${NSPARSE_PROBE_SLEEP:-240}
    continue
  fi
  break
done
tail -150 $LOG
