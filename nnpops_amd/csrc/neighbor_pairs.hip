// neighbor_pairs.hip -- placeholder until the getNeighborPairs kernels land (same round).
#include "host_common.h"
using namespace nnpops;
#define NOT_YET return fail(NNPOPS_ERR_UNSUPPORTED, "%s: not built yet", __func__)
extern "C" {
int64_t nnpops_neighbor_pairs_workspace_bytes(int) { return 0; }
int nnpops_neighbor_pairs_forward(int, int, const void*, const void*, double, int64_t, int32_t*, void*, void*, int32_t*, void*, void*) { NOT_YET; }
int nnpops_neighbor_pairs_backward(int, int, int64_t, const int32_t*, const void*, const void*, const void*, const void*, void*, void*) { NOT_YET; }
}
