#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): a list of steps, each under its own timeout, logs under gpurun_out/<tag>/.
# A step that fails an assertion does not stop the list; a step that is KILLED (timeout: rc 124 / 137) does -- after a hung
# GPU step nothing else touches the card in this call.
#   gpurun --timeout 900 -- 'bash tools/gpu_steps.sh TAG "SECONDS|name|command" ...'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p "$O"
for step in "$@"; do
    IFS='|' read -r secs name cmd <<< "$step"
    echo "== $name"
    timeout -k 10 "$secs" bash -c "$cmd" > "$O/$name.log" 2>&1
    rc=$?
    echo "== $name rc=$rc"; tail -n 6 "$O/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit $rc; fi
done
exit 0
