#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for sb in 0 4 6 8 12; do echo "sb $sb: $(SSG_I8_SB=$sb timeout 300 python tools/time_stages.py --track hard --lam 0.3 --reps 5 2>&1 | grep -i 'sqdist_self_i8')"; done
echo "N=30000: $(timeout 300 python tools/time_stages.py --N 30000 --track hard --lam 0.3 --reps 2 2>&1 | grep -i 'sqdist_self_i8')"
echo "N=30000 regs: $(SSG_I8_DMA=0 timeout 300 python tools/time_stages.py --N 30000 --track hard --lam 0.3 --reps 2 2>&1 | grep -i 'sqdist_self_i8')"
