#!/bin/bash
# Run GPU steps one after the other on the box, each under its own timeout, each logging to gpurun_out/<dir>/<name>.log.
# An ordinary failure (a red test, rc 1) does not stop the sequence; a step that TIMES OUT or is killed does -- no further
# GPU step is started after one (the card may be wedged).
#   tools/gpu_steps.sh <dir> <name> <seconds> <command...> -- <name> <seconds> <command...> -- ...
set -u
dir="gpurun_out/$1"; shift
mkdir -p "$dir"
while [ $# -gt 0 ]; do
    name="$1"; secs="$2"; shift 2
    cmd=()
    while [ $# -gt 0 ] && [ "$1" != "--" ]; do cmd+=("$1"); shift; done
    [ $# -gt 0 ] && shift
    echo "== $name: ${cmd[*]}"
    timeout -k 10 "$secs" "${cmd[@]}" > "$dir/$name.log" 2>&1
    rc=$?
    echo "rc=$rc" >> "$dir/$name.log"
    tail -n 12 "$dir/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "== $name timed out / was killed: stopping here"
        exit $rc
    fi
done
exit 0
