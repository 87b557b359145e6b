#!/bin/bash
# A/B of library builds / env knobs through tools/layer_table.py (development aid).
# usage: tools/ab_layers.sh <outdir> "<name>|<env assignments>" ...
out=$1; shift
mkdir -p $out
for spec in "$@"; do
  name=${spec%%|*}; envs=${spec#*|}
  echo "=== $name ($envs)"
  env $envs python tools/layer_table.py --B ${AB_B:-1000} --reps ${AB_REPS:-3} --json $out/$name.json > $out/$name.md 2>$out/$name.err
  tail -1 $out/$name.md
done
