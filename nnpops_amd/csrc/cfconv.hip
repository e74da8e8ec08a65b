// cfconv.hip -- placeholder until the CFConv kernels land (same round); every entry point fails loudly.
#include "host_common.h"
using namespace nnpops;
#define NOT_YET return fail(NNPOPS_ERR_UNSUPPORTED, "%s: not built yet", __func__)
extern "C" {
int nnpops_cfconv_neighbors_create(nnpops_cfconv_neighbors_t*, int, float, int, int) { NOT_YET; }
int nnpops_cfconv_neighbors_destroy(nnpops_cfconv_neighbors_t) { NOT_YET; }
int nnpops_cfconv_neighbors_set_stream(nnpops_cfconv_neighbors_t, void*) { NOT_YET; }
int nnpops_cfconv_neighbors_build(nnpops_cfconv_neighbors_t, const float*, const float*) { NOT_YET; }
int nnpops_cfconv_neighbors_check(nnpops_cfconv_neighbors_t, int*) { NOT_YET; }
int nnpops_cfconv_neighbors_export(nnpops_cfconv_neighbors_t, int, int32_t*, float*) { NOT_YET; }
int nnpops_cfconv_create(nnpops_cfconv_t*, int, int, int, float, int, float, int, const float*, const float*, const float*, const float*, int) { NOT_YET; }
int nnpops_cfconv_destroy(nnpops_cfconv_t) { NOT_YET; }
int nnpops_cfconv_set_stream(nnpops_cfconv_t, void*) { NOT_YET; }
int nnpops_cfconv_compute(nnpops_cfconv_t, nnpops_cfconv_neighbors_t, const float*, const float*, const float*, float*) { NOT_YET; }
int nnpops_cfconv_backprop(nnpops_cfconv_t, nnpops_cfconv_neighbors_t, const float*, const float*, const float*, const float*, float*, float*) { NOT_YET; }
}
