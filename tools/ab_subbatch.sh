#!/bin/bash
# per-image launch times of stem .. layer2 entry at small batches (do the tensors between launches stay in the 256 MB Infinity Cache?)
R=$GRAFT_REPO_ROOT; cd $R
for v in "" _ab/bn_off.so; do for b in 32 48 64 96 128 1000; do
  echo "## lib=[$v] B=$b (us per image)"
  SSG_LIB_PATH=${v:+$R/$v} timeout 300 python tools/layer_table.py --B $b --reps 5 2>/dev/null | grep -E "^\| +[0-6] \|" | awk -F'|' -v b=$b '{printf "%s|%s| %.3f\n", $2, $3, $4*1000/b}'
done; done
