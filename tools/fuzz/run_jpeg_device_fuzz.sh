#!/bin/bash
# tools/fuzz/run_jpeg_device_fuzz.sh [seconds=60]: the three kernels of csrc/jpeg.hip executed on the host under libFuzzer + ASan + UBSan
# (shift-base / signed-overflow checks off: the integer IDCT shifts negative values and wraps on garbage coefficients exactly like jidctint.c)
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=$HERE/../../self-similarity-grouping_amd/csrc/jpeg.hip
W=${FUZZ_DIR:-/tmp/ssg_fuzz}
mkdir -p $W/corpus_dev $W/inc
python3 $HERE/make_seeds.py $W/corpus_dev
# the kernels' source text: from the first namespace line up to the host launcher; the include of ssg_common.h is dropped (the parser's
# translation unit has it already)
awk '/^namespace ssg \{/{on=1} /^using namespace ssg;/{exit} on{print}' $SRC > $W/inc/jpeg_kernels_cut.inc
grep -q colour_kernel $W/inc/jpeg_kernels_cut.inc
/opt/rocm/lib/llvm/bin/clang++ -x hip --offload-host-only -O1 -g -fsanitize=fuzzer,address,undefined -fno-sanitize-recover=undefined -fno-sanitize=shift-base,signed-integer-overflow \
  -I/opt/rocm/include -I$W/inc -o $W/jpeg_device_fuzz $HERE/jpeg_device_fuzz.cpp -lpthread 2>&1 | grep -v "option-ignored\|not currently supported" || true
cd $W && ./jpeg_device_fuzz -max_total_time=${1:-60} -max_len=8192 -timeout=10 -print_final_stats=1 corpus_dev 2>&1 | tail -12
