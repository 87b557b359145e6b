#!/bin/bash
# Round-4 build: matrix-pipe utilisation and wave states per kernel (one counter pass per command, kernel-trace + pmc only) -> gpurun_out/r04m/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04m; mkdir -p $O
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout 240 $R/tools/pmc_generic.sh r04m_group "$C" -- python $R/tools/time_stages.py --track hard --lam 0.3 --reps 1 > $O/pmc_mfma_group.txt 2>&1
timeout 300 $R/tools/pmc_generic.sh r04m_embed "$C" -- python $R/tools/time_embed.py --B 1000 --iters 1 > $O/pmc_mfma_embed.txt 2>&1
cd $R && python3 tools/pmc_mfma_summary.py $O/pmc_mfma_embed.txt $O/pmc_mfma_group.txt > $O/r04_pmc_mfma_table.md
cat $O/r04_pmc_mfma_table.md | cut -c1-170
