#!/bin/bash
# round 6: nt cache policy on the straight-line epilogue's residual loads / output stores (variant libraries under _ab/), and the pair kernel.
# The variant libraries are builds of the same source tree with -D switches (git-ignored, they travel to the GPU box with the snapshot), e.g.
#   cd self-similarity-grouping_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC ssg_hip.hip -ldl \
#      -Wl,-rpath,/opt/rocm/lib -DSSG_BN_NT_STORE=0 -DSSG_BN_NT_RES=0 -DSSG_DMA_RES_AUX=0 -DSSG_DMA_OUT_AUX_RES=0 -o ../../_ab/old.so
# and are compared interleaved in one process by tools/ab_inproc.py (name=path[,ENV=VAL...]); ab_nt2.sh / ab_nt3.sh / ab_dma3.sh are the later sweeps of
# profiles/r06_ab_nt_policy.txt (VARS lists the variants).
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/ab_nt; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "conv_pair or fast_epilogue" 2>&1 | tail -2 | tee $O/tests.txt
timeout 600 python tools/ab_inproc.py --reps ${REPS:-6} base=self-similarity-grouping_amd/libssg_hip.so $VARS > $O/ab.txt 2>&1; tail -45 $O/ab.txt
for r in 1 2; do for pr in 0 1; do
  SSG_CONV_PAIR=$pr timeout 300 python tools/layer_table.py --reps 3 2>/dev/null | grep -E "^\| (1[3-9]|2[0-7]) |total" | sed "s/^/PAIR=$pr /" | tee -a $O/pair.txt
done; done
