// mfma4x4_layout.hip -- checks on the device the two hardware assumptions of ani_angular_mfma.h:
//   (1) v_mfma_f32_4x4x1_16b_f32:  D register r of lane l  +=  A(lane 4*(l/4) + r) * B(lane l)
//   (2) wave_max_nonneg (DPP row scan + row broadcasts) returns the wave maximum
// Build: hipcc --offload-arch=gfx950 -O3 -I nnpops_amd/csrc tools/ubench/mfma4x4_layout.hip -o tools/ubench/mfma4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "device_common.h"

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* a, const float* b, float* d, const int* v, int* vmax) {
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[64 + l], b[64 + l], c, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[4 * l + r] = c[r];
    const int m = nnpops::wave_max_nonneg(v[l]);
    if (l == 0) *vmax = m;
}

int main() {
    float ha[128], hb[128], hd[256];
    int hv[64], hmax = -1;
    int bad = 0;
    for (int trial = 0; trial < 20; trial++) {
        int ref = 0;
        for (int i = 0; i < 128; i++) { ha[i] = (float)(rand() % 17 - 8); hb[i] = (float)(rand() % 13 - 6); }
        for (int i = 0; i < 64; i++) { hv[i] = rand() % 1000; if (trial == 3) hv[i] = (i == 17) ? 5 : 0; if (hv[i] > ref) ref = hv[i]; }
        float *da, *db, *dd; int *dv, *dm;
        hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, sizeof(hd)); hipMalloc(&dv, sizeof(hv)); hipMalloc(&dm, 4);
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        hipMemcpy(dv, hv, sizeof(hv), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd, dv, dm);
        hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost); hipMemcpy(&hmax, dm, 4, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 4; r++) {
                const int al = 4 * (l / 4) + r;
                const float want = ha[al] * hb[l] + ha[64 + al] * hb[64 + l];
                if (hd[4 * l + r] != want) { if (bad < 5) printf("MFMA layout mismatch lane %d reg %d: %g vs %g\n", l, r, hd[4 * l + r], want); bad++; }
            }
        if (hmax != ref) { printf("wave_max mismatch: %d vs %d\n", hmax, ref); bad++; }
        hipFree(da); hipFree(db); hipFree(dd); hipFree(dv); hipFree(dm);
    }
    printf(bad ? "mfma4x4_layout: FAILED (%d)\n" : "mfma4x4_layout: OK\n", bad);
    return bad != 0;
}
