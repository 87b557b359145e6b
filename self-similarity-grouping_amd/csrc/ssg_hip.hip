// ssg_hip.hip -- single translation unit of libssg_hip.so (gfx950 only).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC ssg_hip.hip -o ../libssg_hip.so
#include "ssg_api.hip"
#include "pairwise.hip"
#include "gram_i8.hip"
#include "topk.hip"
#include "topk_intro.hip"
#include "krecip.hip"
#include "jaccard.hip"
#include "cluster.hip"
#include "conv.hip"
#include "rerank_init.hip"
#include "ranking.hip"
#include "rerank_plain.hip"
