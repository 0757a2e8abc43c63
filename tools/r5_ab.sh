#!/bin/bash
# round 5 A/B helper: runs a list of (label, env, command) legs on one box and prints one line per leg.
#   tools/r5_ab.sh <out-file> -- "<label>|<env assignments>|<command>" ...
out=$1; shift; shift
: > "$out"
for leg in "$@"; do
  IFS='|' read -r label envs cmd <<< "$leg"
  res=$(env $envs bash -c "$cmd" 2>&1 | tail -n 3 | tr '\n' ' ')
  echo "$label | $envs | $res" | tee -a "$out"
done
