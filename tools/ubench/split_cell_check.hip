// split_cell_check.hip -- exhaustively compares celllist.h's division-free split_cell() with integer division over
// whole grids, including grids far larger than the float estimate alone can index (round-2 advisor finding: the
// uncorrected (c + 1/2) * rcp(n) floor is only safe below ~2.7 M cells).
//   hipcc --offload-arch=gfx950 -O3 -I nnpops_amd/csrc tools/ubench/split_cell_check.hip -o tools/ubench/split_cell_check
// prints "mismatches 0" and exits 0 when every cell of every grid splits exactly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "celllist.h"

__global__ void check(int nx, int ny, int nz, unsigned long long* bad) {
    nnpops::CellGrid g{};
    g.nx = nx; g.ny = ny; g.nz = nz; g.ncells = nx * ny * nz;
    for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < g.ncells; c += (long)gridDim.x * blockDim.x) {
        int cx, cy, cz;
        nnpops::split_cell(g, (int)c, cx, cy, cz);
        const int ez = (int)c / (nx * ny), rem = (int)c % (nx * ny), ey = rem / nx, ex = rem % nx;
        if (cx != ex || cy != ey || cz != ez) atomicAdd(bad, 1ull);
    }
}

int main() {
    const int grids[][3] = {{3, 3, 3}, {18, 18, 18}, {101, 103, 107}, {255, 257, 251}, {1, 1, 4000000}, {4000000, 1, 1}, {7, 4093, 577},
                            {2048, 2048, 4}, {3, 3, 1864135}, {1291, 1297, 10}, {16777215, 1, 1}, {1, 16777215, 1}, {4099, 4093, 1}};
    unsigned long long* bad;
    hipMalloc(&bad, sizeof(*bad));
    hipMemset(bad, 0, sizeof(*bad));
    for (auto& g : grids) hipLaunchKernelGGL(check, dim3(2048), dim3(256), 0, 0, g[0], g[1], g[2], bad);
    unsigned long long h = 0;
    hipMemcpy(&h, bad, sizeof(h), hipMemcpyDeviceToHost);
    printf("mismatches %llu\n", h);
    return h == 0 ? 0 : 1;
}
