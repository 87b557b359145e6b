#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/ab_nt; mkdir -p $O
for e in "SSG_CONV_DMA=3" "SSG_CONV_DMA=3 SSG_SPLIT_BK16=0"; do
  echo "## $e"; env $e timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "conv or embedding or fused or bottleneck or stem" 2>&1 | tail -2
done
L=self-similarity-grouping_amd/libssg_hip.so
timeout 900 python tools/ab_inproc.py --reps 6 base=$L dma3=$L,SSG_CONV_DMA=3 dma3b=$L,SSG_CONV_DMA=3,SSG_SPLIT_BK16=0 2>&1 | grep -E "^launch|^ [0-9] |^1[0-2] |total"
