#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06t; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -6 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
