#!/bin/bash
# tools/fuzz/run_jpeg_host_fuzz.sh [seconds=60]: build the harness with libFuzzer + ASan + UBSan (ROCm's clang, host only) and run it
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
W=${FUZZ_DIR:-/tmp/ssg_fuzz}
mkdir -p $W/corpus
python3 $HERE/make_seeds.py $W/corpus
/opt/rocm/lib/llvm/bin/clang++ -x hip --offload-host-only -O1 -g -fsanitize=fuzzer,address,undefined -fno-sanitize-recover=undefined \
  -I/opt/rocm/include -o $W/jpeg_host_fuzz $HERE/jpeg_host_fuzz.cpp -lpthread
cd $W && ./jpeg_host_fuzz -max_total_time=${1:-60} -max_len=8192 -print_final_stats=1 corpus 2>&1 | tail -15
