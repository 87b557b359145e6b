#!/bin/bash
# round-3 experiment batch 2 (run through gpurun): split introsort, 4-stage tall conv tiles, stream-grid sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/c2; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "introsort_rank or tile_shapes or rerank_stages or fused_stem or embedding_vs" > $O/t1.log 2>&1; tail -3 $O/t1.log
for t in 1024 4096; do SSG_INTRO_TAILN=$t python -m pytest tests/test_gpu_parity.py -x -q -k "introsort_rank and (heapsort or 5000 or 16000 or 40000)" > $O/t_tail$t.log 2>&1; tail -1 $O/t_tail$t.log; done
for t in 0 1024 2048 4096; do echo TAILN=$t; SSG_INTRO_TAILN=$t python tools/time_rank.py 16000 30000; done > $O/rank.log 2>&1
cat $O/rank.log
python tools/layer_table.py > $O/lt_ns4.md 2>&1
SSG_CONV_TALL_STAGES=3 python tools/layer_table.py > $O/lt_ns3.md 2>&1
SSG_CONV_TALL_RES=1 python tools/layer_table.py > $O/lt_ns4_res.md 2>&1
SSG_CONV_TALL_DUAL=1 python tools/layer_table.py > $O/lt_ns4_dual.md 2>&1
SSG_CONV_DMA=3 python tools/layer_table.py > $O/lt_ns4_dma3.md 2>&1
tail -1 $O/lt_*.md
for g in 1000 1280 2000 4000; do echo GRID=$g; SSG_STREAM_GRID=$g python tools/time_stages.py --reps 2 2>&1 | tail -24; done > $O/stages.log 2>&1
grep -E "GRID|region_query|compact_below|jaccard_rows|rep 1" $O/stages.log
