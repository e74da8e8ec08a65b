// celllist.h -- device-side cell grid shared by the ANI neighbour rows, the CFConv half list and
// getNeighborPairs.  New design: the reference searches all O(N^2) pairs and says itself that a
// voxel algorithm is the fix (src/ani/CpuANISymmetryFunctions.cpp:114-116).
//
// Everything is decided on the device (the box lives in device memory and must not be read back):
//   grid_setup      1 block   box or bounding box -> CellGrid (dims, fractional transform); clears counts
//   assign_cells    1 thread/atom   cell id per atom + histogram (int atomics in L2)
//   scan_cells      1 block   exclusive prefix sum of the histogram
//   fill_cells      1 thread/atom   scatter atom ids into their cell segment (arrival order)
//   order_cells     1 thread/atom   rank inside the segment by atom id (deterministic order), emit
//                                   cell-ordered float4 {x, y, z, id}
// Consumers walk the 3x3x3 block of cells around an atom.  The stencil only prunes candidates: the
// displacement of every candidate is still computed with the reference's minimum-image rule
// (device_common.h: min_image), so results are identical to the all-pairs scan.  The stencil is
// valid when every periodic axis has >= 3 cells of perpendicular width >= cutoff; otherwise
// grid.ok = 0 and the caller falls back to the all-pairs kernels.
#pragma once

#include "device_common.h"

namespace nnpops {

constexpr int kTagShift = 24;                 // packed neighbour word: (tag << 24) | atom id
constexpr int kIdMask = (1 << kTagShift) - 1;

struct CellGrid {
    int nx, ny, nz, ncells;
    int periodic;
    int ok;                 // 0: the stencil would be invalid for this box (too few cells)
    // lattice coordinates: sz = (z-oz)*izz; sy = ((y-oy) - sz*cy)*iyy; sx = ((x-ox) - sy*bx - sz*cx)*ixx
    float ox, oy, oz;
    float ixx, iyy, izz;
    float bx, cx, cy;
};

__device__ __forceinline__ void cell_of(const CellGrid& g, float x, float y, float z, int& cx, int& cy, int& cz) {
    // lattice coordinates of p = sx*a + sy*b + sz*c for the lower-triangular cell a=(ax,0,0), b=(bx,by,0),
    // c=(cx,cy,cz); for the non-periodic bounding box bx = cx = cy = 0 and the origin is its corner
    float sz = (z - g.oz) * g.izz;
    float sy = ((y - g.oy) - sz * g.cy) * g.iyy;
    float sx = ((x - g.ox) - sy * g.bx - sz * g.cx) * g.ixx;
    if (g.periodic) {
        sx -= floorf(sx); sy -= floorf(sy); sz -= floorf(sz);
    }
    cx = min(max((int)(sx * g.nx), 0), g.nx - 1);
    cy = min(max((int)(sy * g.ny), 0), g.ny - 1);
    cz = min(max((int)(sz * g.nz), 0), g.nz - 1);
}

// One block of 256 threads.
static __global__ __launch_bounds__(256) void grid_setup(int N, const float* __restrict__ pos, const float* __restrict__ box,
                                                  int periodic, float cutoff, int max_cells, CellGrid* __restrict__ grid,
                                                  int* __restrict__ cell_count) {
    __shared__ float red[6][256];
    __shared__ CellGrid g;
    const int tid = threadIdx.x;
    if (!periodic) {
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (int i = tid; i < N; i += 256)
            for (int d = 0; d < 3; d++) {
                const float v = pos[3 * i + d];
                lo[d] = fminf(lo[d], v);
                hi[d] = fmaxf(hi[d], v);
            }
        for (int d = 0; d < 3; d++) { red[d][tid] = lo[d]; red[3 + d][tid] = hi[d]; }
        __syncthreads();
        for (int off = 128; off >= 1; off >>= 1) {
            if (tid < off)
                for (int d = 0; d < 3; d++) {
                    red[d][tid] = fminf(red[d][tid], red[d][tid + off]);
                    red[3 + d][tid] = fmaxf(red[3 + d][tid], red[3 + d][tid + off]);
                }
            __syncthreads();
        }
    }
    if (tid == 0) {
        g.periodic = periodic;
        g.ok = 1;
        float wx, wy, wz;      // perpendicular widths of the cell-able region
        if (periodic) {
            const float ax = box[0], bx = box[3], by = box[4], cx = box[6], cy = box[7], cz = box[8];
            g.ox = g.oy = g.oz = 0.f;
            g.ixx = 1.0f / ax; g.iyy = 1.0f / by; g.izz = 1.0f / cz;
            g.cy = cy; g.bx = bx; g.cx = cx;
            // perpendicular widths of the box (lower-triangular cell)
            wz = cz;
            wy = by * cz / sqrtf(cy * cy + cz * cz);
            const float nxv = by * cz, nyv = -bx * cz, nzv = bx * cy - by * cx;
            wx = ax * by * cz / sqrtf(nxv * nxv + nyv * nyv + nzv * nzv);
        } else {
            const float pad = 1e-3f;
            g.ox = red[0][0] - pad; g.oy = red[1][0] - pad; g.oz = red[2][0] - pad;
            wx = red[3][0] - red[0][0] + 2 * pad; wy = red[4][0] - red[1][0] + 2 * pad; wz = red[5][0] - red[2][0] + 2 * pad;
            g.ixx = 1.0f / wx; g.iyy = 1.0f / wy; g.izz = 1.0f / wz;
            g.bx = g.cx = g.cy = 0.f;
        }
        // largest dims with cell width >= cutoff (a hair of slack for rounding in cell_of)
        const float c = cutoff * 1.0001f;
        int nx = max(1, (int)floorf(wx / c)), ny = max(1, (int)floorf(wy / c)), nz = max(1, (int)floorf(wz / c));
        if (periodic && (nx < 3 || ny < 3 || nz < 3)) g.ok = 0;
        // cap the total cell count (sparse systems): coarser cells are always valid
        while ((long long)nx * ny * nz > max_cells) {
            if (nx >= ny && nx >= nz) nx = max(periodic ? 3 : 1, nx - (nx + 7) / 8);
            else if (ny >= nz) ny = max(periodic ? 3 : 1, ny - (ny + 7) / 8);
            else nz = max(periodic ? 3 : 1, nz - (nz + 7) / 8);
            if (periodic && nx == 3 && ny == 3 && nz == 3) break;
        }
        g.nx = nx; g.ny = ny; g.nz = nz;
        g.ncells = nx * ny * nz;
        if (g.ncells > max_cells) g.ok = 0;
        *grid = g;
    }
    __syncthreads();
    const int ncells = min(g.ncells, max_cells);
    for (int c = tid; c < ncells; c += 256) cell_count[c] = 0;
}

static __global__ void assign_cells(int N, const float* __restrict__ pos, const CellGrid* __restrict__ grid,
                             int* __restrict__ cell_count, int* __restrict__ atom_cell, int* __restrict__ atom_rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const CellGrid g = *grid;
    if (!g.ok) return;
    int cx, cy, cz;
    cell_of(g, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], cx, cy, cz);
    const int c = (cz * g.ny + cy) * g.nx + cx;
    atom_cell[i] = c;
    atom_rank[i] = atomicAdd(&cell_count[c], 1);
}

// Exclusive scan of cell_count[0..ncells) into cell_start[0..ncells]; one block of 1024 threads.
static __global__ __launch_bounds__(1024) void scan_cells(const CellGrid* __restrict__ grid, const int* __restrict__ cell_count,
                                                   int* __restrict__ cell_start) {
    __shared__ int wave_tot[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncells = grid->ok ? grid->ncells : 0;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < ncells; base += 1024) {
        const int c = base + tid;
        const int v = c < ncells ? cell_count[c] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int wave_off = 0;
        for (int w = 0; w < wave; w++) wave_off += wave_tot[w];
        const int excl = carry + wave_off + incl - v;
        if (c < ncells) cell_start[c] = excl;
        __syncthreads();
        if (tid == 1023) carry = excl + v;
        __syncthreads();
    }
    if (tid == 0) cell_start[ncells] = carry;
}

static __global__ void fill_cells(int N, const CellGrid* __restrict__ grid, const int* __restrict__ cell_start,
                           const int* __restrict__ atom_cell, const int* __restrict__ atom_rank,
                           int* __restrict__ sorted_atom) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || !grid->ok) return;
    sorted_atom[cell_start[atom_cell[i]] + atom_rank[i]] = i;
}

// One thread per atom: its final slot is the number of smaller atom ids in its cell segment (segments
// hold ~10-30 atoms), which makes the cell order deterministic without a serial per-cell sort; the
// same thread publishes {x,y,z,id} in cell order.
static __global__ void order_cells(int N, const float* __restrict__ pos, const CellGrid* __restrict__ grid,
                            const int* __restrict__ cell_start, const int* __restrict__ atom_cell,
                            const int* __restrict__ unsorted_atom, const int* __restrict__ tag,
                            int* __restrict__ sorted_atom, float4* __restrict__ sorted_pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || !grid->ok) return;
    const int c = atom_cell[i];
    const int lo = cell_start[c], hi = cell_start[c + 1];
    int rank = 0;
    for (int a = lo; a < hi; a++) rank += unsorted_atom[a] < i;
    sorted_atom[lo + rank] = i;
    // .w carries the atom id in its low 24 bits and an optional 8-bit tag (e.g. the species) above them
    const int packed = i | (tag ? (tag[i] << kTagShift) : 0);
    sorted_pos[lo + rank] = make_float4(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], __int_as_float(packed));
}

// Iterate the candidate ranges of the 3x3x3 stencil around cell (cx,cy,cz).  For every (dy,dz)
// the three x-neighbours are contiguous in memory except across the periodic seam, so a stencil is
// at most 18 [begin,end) ranges of sorted slots.  F(begin, end) is called wave-uniformly.
template <typename F>
__device__ __forceinline__ void for_each_stencil_range(const CellGrid& g, const int* __restrict__ cell_start, int cx,
                                                       int cy, int cz, F&& f) {
    for (int dz = -1; dz <= 1; dz++) {
        int z = cz + dz;
        if (g.periodic) z = (z + g.nz) % g.nz;
        else if (z < 0 || z >= g.nz) continue;
        for (int dy = -1; dy <= 1; dy++) {
            int y = cy + dy;
            if (g.periodic) y = (y + g.ny) % g.ny;
            else if (y < 0 || y >= g.ny) continue;
            const int rowbase = (z * g.ny + y) * g.nx;
            int x0 = cx - 1, x1 = cx + 1;
            if (!g.periodic) {
                x0 = max(x0, 0); x1 = min(x1, g.nx - 1);
                f(cell_start[rowbase + x0], cell_start[rowbase + x1 + 1]);
            } else if (x0 < 0) {                   // wraps on the low side: [nx-1] + [0..x1]
                f(cell_start[rowbase + g.nx - 1], cell_start[rowbase + g.nx]);
                f(cell_start[rowbase], cell_start[rowbase + x1 + 1]);
            } else if (x1 >= g.nx) {               // wraps on the high side: [x0..nx-1] + [0]
                f(cell_start[rowbase + x0], cell_start[rowbase + g.nx]);
                f(cell_start[rowbase], cell_start[rowbase + 1]);
            } else {
                f(cell_start[rowbase + x0], cell_start[rowbase + x1 + 1]);
            }
        }
    }
}

}  // namespace nnpops
