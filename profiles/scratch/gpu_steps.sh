#!/bin/bash
# Runs GPU steps one after another; a step that times out or is killed ends the call (no further GPU step after it),
# an ordinary failure (exit 1: a failed assertion) does not.  Usage: gpu_steps.sh OUTDIR "name|seconds|command" ...
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2> "$out/$name.err"
  rc=$?
  echo "== $name exit $rc"; tail -n 4 "$out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name timed out or was killed: stopping"; exit $rc; fi
done
