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
//   bin_atoms + order_binned   the same in two launches for stateful periodic callers (see below)
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
    int m;                  // stencil half-width in cells: cells are at least cutoff / m wide (1, or 2 for the fine grid)
    int ok;                 // 0: the stencil would be invalid for this box (too few cells), or a bin overflowed
    int bin_overflow;       // 1: ok was cleared because a cell holds more atoms than the bins of the two-kernel build
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

// Linear cell index -> (cx, cy, cz) without an integer division: the quotient is estimated with v_rcp_f32 (1 ulp) in
// float and corrected by one exact integer step.  The estimate alone is only a floor while c * 2e-7 < 1/2, i.e. below
// ~2.7 M cells (max_cells = N + 4096 allows more, N up to 2^24 - 1); the correction makes it exact for every grid an
// int can index: the estimate is off by at most one whenever the relative error times the QUOTIENT (<= nz, ny) is
// below one.  Checked against c / n over whole grids of up to 16 M cells by tools/ubench/split_cell_check.hip.
__device__ __forceinline__ void split_cell(const CellGrid& g, int c, int& cx, int& cy, int& cz) {
    const int nxy = g.nx * g.ny;
    cz = (int)(((float)c + 0.5f) * __builtin_amdgcn_rcpf((float)nxy));
    int rem = c - cz * nxy;
    if (rem < 0) { cz--; rem += nxy; } else if (rem >= nxy) { cz++; rem -= nxy; }
    cy = (int)(((float)rem + 0.5f) * __builtin_amdgcn_rcpf((float)g.nx));
    cx = rem - cy * g.nx;
    if (cx < 0) { cy--; cx += g.nx; } else if (cx >= g.nx) { cy++; cx -= g.nx; }
}

// The grid for a box (periodic) or a bounding box lo..hi (non-periodic): the largest dims whose cells are at
// least `cutoff` wide, capped at max_cells.  `fine`: prefer cells of half the cutoff (a 5x5x5 stencil holds 58 % of
// the volume of the 3x3x3 one of full-width cells, so a consumer tests 42 % fewer candidates) when such a grid has
// at most max_cells cells and, in a periodic box, at least 5 cells along every axis; g.m says which it is.
__device__ inline CellGrid decide_grid(int periodic, const float* __restrict__ box, const float* lo, const float* hi,
                                       float cutoff, int max_cells, int fine = 0) {
    CellGrid g;
    g.periodic = periodic;
    g.m = 1;
    g.ok = 1;
    g.bin_overflow = 0;
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
        g.ox = lo[0] - pad; g.oy = lo[1] - pad; g.oz = lo[2] - pad;
        wx = hi[0] - lo[0] + 2 * pad; wy = hi[1] - lo[1] + 2 * pad; wz = hi[2] - lo[2] + 2 * pad;
        g.ixx = 1.0f / wx; g.iyy = 1.0f / wy; g.izz = 1.0f / wz;
        g.bx = g.cx = g.cy = 0.f;
    }
    // largest dims with cell width >= cutoff (a hair of slack for rounding in cell_of)
    const float c = cutoff * 1.0001f;
    int nx = max(1, (int)floorf(wx / c)), ny = max(1, (int)floorf(wy / c)), nz = max(1, (int)floorf(wz / c));
    if (fine) {
        const float h = 0.5f * c;
        const int fx = max(1, (int)floorf(wx / h)), fy = max(1, (int)floorf(wy / h)), fz = max(1, (int)floorf(wz / h));
        if ((long long)fx * fy * fz <= max_cells && (!periodic || (fx >= 5 && fy >= 5 && fz >= 5))) {
            g.m = 2;
            g.nx = fx; g.ny = fy; g.nz = fz;
            g.ncells = fx * fy * fz;
            return g;
        }
    }
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
    return g;
}

// Blocks of 256 threads: one for a non-periodic system (bounding-box reduction), several for a periodic one.
static __global__ __launch_bounds__(256) void grid_setup(int N, const float* __restrict__ pos, const float* __restrict__ box,
                                                  int periodic, float cutoff, int max_cells, CellGrid* __restrict__ grid,
                                                  int* __restrict__ cell_count, int fine) {
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
        float lo3[3] = {red[0][0], red[1][0], red[2][0]}, hi3[3] = {red[3][0], red[4][0], red[5][0]};
        g = decide_grid(periodic, box, lo3, hi3, cutoff, max_cells, fine);
        if (blockIdx.x == 0) *grid = g;
    }
    __syncthreads();
    // (periodic: launched with several blocks, each decides the same grid from the box and clears a slice of the counts)
    const int ncells = min(g.ncells, max_cells);
    for (int c = blockIdx.x * 256 + tid; c < ncells; c += gridDim.x * 256) cell_count[c] = 0;
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

// Exclusive scan of cell_count[0..ncells) into cell_start[0..ncells].  A tile of 8192 cells goes through the LDS of one
// block of 1024 threads: coalesced loads, every thread scans a run of 8 in LDS, one block-wide scan, coalesced stores.
// Grids of one tile (every system of up to 4096 atoms) are done in ONE launch by one block looping over the tiles;
// larger grids take one block per tile plus a second launch that adds the totals of the tiles before (a 60 000-cell
// half-cutoff grid: 25 us as a loop of 8 tiles in one block, 46 us at 1024 cells per barrier round).
constexpr int kScanPer = 8, kScanTile = 1024 * kScanPer;

// tile_total == NULL: one block, loops over all tiles.  Otherwise block b scans tile b relative to its own start and
// leaves its total in tile_total[b].
static __global__ __launch_bounds__(1024) void scan_cells(const CellGrid* __restrict__ grid, const int* __restrict__ cell_count,
                                                   int* __restrict__ cell_start, int* __restrict__ tile_total) {
    constexpr int PER = kScanPer, TILE = kScanTile;
    __shared__ int tile[TILE];
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncells = grid->ok ? grid->ncells : 0;
    int carry = 0;                                            // the same in every thread
    const int first = tile_total ? blockIdx.x * TILE : 0, last = tile_total ? min(first + TILE, ncells) : ncells;
    for (int base = first; base < last; base += TILE) {
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int c = base + q * 1024 + tid;
            tile[q * 1024 + tid] = c < ncells ? cell_count[c] : 0;
        }
        __syncthreads();
        int v[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; q++) { v[q] = tile[tid * PER + q]; sum += v[q]; }
        const int incl = wave_prefix_sum(sum);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int run = carry + incl - sum, total = 0;
        for (int w = 0; w < 16; w++) { if (w < wave) run += wave_tot[w]; total += wave_tot[w]; }
#pragma unroll
        for (int q = 0; q < PER; q++) { tile[tid * PER + q] = run; run += v[q]; }
        carry += total;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int c = base + q * 1024 + tid;
            if (c < ncells) cell_start[c] = tile[q * 1024 + tid];
        }
        __syncthreads();
    }
    if (tile_total) { if (tid == 0) tile_total[blockIdx.x] = carry; }
    else if (tid == 0) cell_start[ncells] = carry;
}

// second launch of the tiled scan: block b adds the totals of tiles 0..b-1 to its tile; the last slot gets the grand total
static __global__ __launch_bounds__(1024) void add_tile_offsets(const CellGrid* __restrict__ grid, const int* __restrict__ tile_total,
                                                                int* __restrict__ cell_start) {
    const int ncells = grid->ok ? grid->ncells : 0;
    const int ntiles = (ncells + kScanTile - 1) / kScanTile;
    if ((int)blockIdx.x >= ntiles) return;
    int off = 0;
    for (int b = 0; b < (int)blockIdx.x; b++) off += tile_total[b];
    if (blockIdx.x > 0)
        for (int q = 0; q < kScanPer; q++) {
            const int c = blockIdx.x * kScanTile + q * 1024 + threadIdx.x;
            if (c < ncells) cell_start[c] += off;
        }
    if ((int)blockIdx.x == ntiles - 1 && threadIdx.x == 0) cell_start[ncells] = off + tile_total[blockIdx.x];
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
                            int* __restrict__ sorted_atom, float4* __restrict__ sorted_pos, int* __restrict__ sorted_cell) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || !grid->ok) return;
    const int c = atom_cell[i];
    const int lo = cell_start[c], hi = cell_start[c + 1];
    int rank = 0;
    for (int a = lo; a < hi; a++) rank += unsorted_atom[a] < i;
    sorted_atom[lo + rank] = i;
    if (sorted_cell) sorted_cell[lo + rank] = c;
    // .w carries the atom id in its low 24 bits and an optional 8-bit tag (e.g. the species) above them
    const int packed = i | (tag ? (tag[i] << kTagShift) : 0);
    sorted_pos[lo + rank] = make_float4(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], __int_as_float(packed));
}

// ---------------------------------------------------------------------------------------------
// Two-kernel build for handles that keep state between calls (periodic systems of up to kBinnedAtoms
// atoms).  Every kernel boundary costs ~4 us of dependent-launch latency on MI355X -- more than any of
// the five steps above -- so the steps are regrouped around the one true dependency (all ids of a cell
// must be known before an atom can be ranked inside it):
//   bin_atoms      1 thread/atom   grid from the box (recomputed per block: no setup kernel), cell id,
//                                  histogram, and the id dropped into a fixed-capacity bin of its cell
//   order_binned   1 thread/atom   every block scans the (<= 8192-cell) histogram in LDS on its own
//                                  (no scan kernel), ranks its atoms inside their bins by id and emits
//                                  the same cell-ordered arrays as order_cells
//   (consumer)                     the kernel that walks the grid next clears the histogram for the
//                                  following build (clear_cell_histogram): no memset node, and a captured
//                                  graph replays correctly
// A bin that overflows clears grid.ok; the owner grows the bins in its check() and rebuilds.
// ---------------------------------------------------------------------------------------------
constexpr int kBinnedAtoms = 65536;
constexpr int kPairsBinnedAtoms = 200000, kPairsBinCap = 128;    // the stateless getNeighborPairs op: <= 8 192 cells x 128 ids (4 MiB of workspace)
constexpr int kBinnedCells = 8192;
constexpr int kBinnedThreads = 256;

// hist: [kHistWords] ints: the cell histogram, then the overflow flag of that build.
// (Both phases in ONE launch, the blocks meeting at a device-wide arrival counter between them -- release fence, one
//  atomic per block, spin, acquire fence -- was tried: 10.9 us against 8.6 us for the two launches.  A device-scope fence
//  pair across eight XCDs costs more than the ~2.5 us of dispatch a launch adds.)
constexpr int kHistWords = kBinnedCells + 1;

// phase 1 of the binned build for atom i (any i; returns its cell, or -1)
__device__ __forceinline__ int bin_one_atom(int i, int N, const CellGrid& g, float x, float y, float z, int* __restrict__ hist,
                                            int* __restrict__ bins, int bin_cap) {
    if (i >= N || !g.ok) return -1;
    int cx, cy, cz;
    cell_of(g, x, y, z, cx, cy, cz);
    const int c = (cz * g.ny + cy) * g.nx + cx;
    const int r = atomicAdd(&hist[c], 1);
    if (r < bin_cap) bins[(size_t)c * bin_cap + r] = i;
    else hist[kBinnedCells] = 1;                              // benign race: everyone writes the same value
    return c;
}

// phase 2 (whole block): scan of the histogram in LDS, then atom i (cell c) is ranked inside its bin and published.
// Everything that does not depend on another load is requested up front -- the grid, the atom's cell, position and tag, the whole
// (8 192-word) histogram, then the first eight ids of the atom's bin as soon as the cell is there: two round trips to memory and
// the scan, where "grid, then cell and counts, then bin, then position" was four.
template <int T>
__device__ __forceinline__ void order_block(int N, CellGrid* __restrict__ grid, const float* __restrict__ pos,
                                            const int* __restrict__ tag, const int* __restrict__ hist, const int* __restrict__ bins,
                                            int bin_cap, const int* __restrict__ atom_cell, int* __restrict__ cell_start,
                                            int* __restrict__ sorted_atom, float4* __restrict__ sorted_pos,
                                            int* __restrict__ sorted_cell, int* s_start, int* wave_tot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * T + tid, ic = min(i, N - 1);
    const CellGrid g = *grid;
    const int c = atom_cell[ic];
    const float px = pos[3 * ic], py = pos[3 * ic + 1], pz = pos[3 * ic + 2];
    const int packed = ic | (tag ? (tag[ic] << kTagShift) : 0);
    const int overflowed = hist[kBinnedCells];
    constexpr int PER = kBinnedCells / T;
    int mine[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) mine[q] = hist[q * T + tid];
    // (cell ids of a grid that was not built are stale: stay inside the bins)
    const int4* bin4 = reinterpret_cast<const int4*>(bins + (size_t)min(max(c, 0), kBinnedCells - 1) * bin_cap);    // bin_cap is a multiple of 4
    const int4 first = bin4[0], second = bin_cap >= 8 ? bin4[1] : make_int4(0, 0, 0, 0);
    if (!g.ok) return;
    if (overflowed != 0) {                                    // a bin overflowed: this grid is unusable
        if (blockIdx.x == 0 && tid == 0) { grid->ok = 0; grid->bin_overflow = 1; }
        return;
    }
    const int ncells = g.ncells;
    // exclusive scan of the histogram, redundantly in every block: coalesced into LDS, then thread t owns a
    // contiguous run of cells
#pragma unroll
    for (int q = 0; q < PER; q++) s_start[q * T + tid] = mine[q];
    __syncthreads();
    const int per = (ncells + T - 1) / T;
    const int c0 = min(tid * per, ncells), c1 = min(c0 + per, ncells);
    int sum = 0;
    for (int q = c0; q < c1; q++) sum += s_start[q];
    const int incl = wave_prefix_sum(sum);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; w++) run += wave_tot[w];
    for (int q = c0; q < c1; q++) { const int v = s_start[q]; s_start[q] = run; run += v; }
    if (tid == T - 1) s_start[ncells] = run;
    __syncthreads();
    // the global copy of the offsets, a slice per block
    for (int q = blockIdx.x * T + tid; q <= ncells; q += gridDim.x * T) cell_start[q] = s_start[q];

    if (i >= N) return;
    const int lo = s_start[c], n = s_start[c + 1] - lo;
    auto below = [&](const int4& v, int k) { return (k < n && v.x < i) + (k + 1 < n && v.y < i) + (k + 2 < n && v.z < i) + (k + 3 < n && v.w < i); };
    int rank = below(first, 0) + below(second, 4);            // deterministic order inside the cell
    for (int k = 8; k < n; k += 4) rank += below(bin4[k >> 2], k);
    sorted_atom[lo + rank] = i;
    if (sorted_cell) sorted_cell[lo + rank] = c;              // (a consumer that walks the sorted order gets the cell without a dependent load)
    sorted_pos[lo + rank] = make_float4(px, py, pz, __int_as_float(packed));
}

static __global__ __launch_bounds__(kBinnedThreads) void bin_atoms(int N, const float* __restrict__ pos,
                                                                   const float* __restrict__ box, float cutoff, int max_cells,
                                                                   CellGrid* __restrict__ grid, int* __restrict__ hist,
                                                                   int* __restrict__ bins, int bin_cap,
                                                                   int* __restrict__ atom_cell, int fine) {
    __shared__ CellGrid g;
    // (the position is requested BEFORE the grid is decided: the box and the atom come back in one round trip, not two -- these
    //  two kernels are nothing but their chains of dependent loads)
    const int i = blockIdx.x * kBinnedThreads + threadIdx.x;
    const int ic = min(i, N - 1);
    const float x = pos[3 * ic], y = pos[3 * ic + 1], z = pos[3 * ic + 2];
    if (threadIdx.x == 0) {
        g = decide_grid(1, box, nullptr, nullptr, cutoff, min(max_cells, kBinnedCells), fine);
        if (blockIdx.x == 0) *grid = g;
    }
    __syncthreads();
    const int c = bin_one_atom(i, N, g, x, y, z, hist, bins, bin_cap);
    if (c >= 0) atom_cell[i] = c;
}

// (T threads per block.  Every block scans the whole histogram -- 23 cells per thread at 256 threads for the 5 832-cell grid of
//  the 10 000-atom frame -- but blocks of 1 024 threads, whose scan is four times shorter, measured SLOWER: the two grid kernels
//  together 8.9 -> 10.9 us (round 4, interleaved A/B): with 10 blocks instead of 40 the ranking loads of the atoms, the other half
//  of the kernel, run on 10 CUs.  Also built and measured in round 4: a second histogram of the ROWS of cells counted by bin_atoms,
//  so that this kernel scans 324 row totals instead of 5 832 cells and adds up the few cell counts in front of an atom's cell:
//  8.6 -> 11.3 us at 10 000 atoms and 12.7 -> 33.5 us at 40 000 -- the extra atomic of every atom lands on a few hundred hot
//  words, and same-address atomics serialise in the L2.)
template <int T>
static __global__ __launch_bounds__(T) void order_binned(int N, const float* __restrict__ pos,
                                                                      const int* __restrict__ tag, CellGrid* __restrict__ grid,
                                                                      const int* __restrict__ hist,
                                                                      const int* __restrict__ bins, int bin_cap,
                                                                      const int* __restrict__ atom_cell, int* __restrict__ cell_start,
                                                                      int* __restrict__ sorted_atom, float4* __restrict__ sorted_pos,
                                                                      int* __restrict__ sorted_cell) {
    __shared__ int s_start[kBinnedCells + 1];
    __shared__ int wave_tot[T / 64];
    order_block<T>(N, grid, pos, tag, hist, bins, bin_cap, atom_cell, cell_start, sorted_atom, sorted_pos, sorted_cell, s_start, wave_tot);
}

// The same stencil as ONE flat candidate index space: lane r < 18 looks up range r, a wave scan gives
// the offsets, and candidate k of the atom is sorted slot k + off[range of k].  A consumer then runs
// ceil(candidates / 64) full iterations with independent loads instead of one (mostly half-empty,
// latency-serialised) iteration per range.  Same candidate order as for_each_stencil_range.
constexpr int kStencilRanges = 18;
struct Stencil {
    int pre[kStencilRanges];      // first flat index of range r           (wave-uniform: SGPRs)
    int delta;                    // lane r: begin_r - pre[r]              (per lane: read back with ds_bpermute)
    int total;
};

__device__ __forceinline__ Stencil gather_stencil(const CellGrid& g, const int* __restrict__ cell_start, int cx, int cy, int cz) {
    const int lane = lane_id();
    int begin = 0, end = 0;
    if (lane < kStencilRanges) {
        const int pair = lane >> 1, sub = lane & 1;
        int z = cz + pair / 3 - 1, y = cy + pair % 3 - 1;
        bool live = true;
        if (g.periodic) { z = (z + g.nz) % g.nz; y = (y + g.ny) % g.ny; }
        else live = z >= 0 && z < g.nz && y >= 0 && y < g.ny;
        if (live) {
            const int rowbase = (z * g.ny + y) * g.nx;
            int x0 = cx - 1, x1 = cx + 1, a = 0, b = -1;              // cells [a, b] of this row
            if (!g.periodic) { if (sub == 0) { a = max(x0, 0); b = min(x1, g.nx - 1); } }
            else if (x0 < 0) { if (sub == 0) { a = b = g.nx - 1; } else { a = 0; b = x1; } }
            else if (x1 >= g.nx) { if (sub == 0) { a = x0; b = g.nx - 1; } else { a = b = 0; } }
            else if (sub == 0) { a = x0; b = x1; }
            if (b >= a) { begin = cell_start[rowbase + a]; end = cell_start[rowbase + b + 1]; }
        }
    }
    const int count = end - begin;
    int incl = count;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {                               // 18 live lanes: five steps
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    const int excl = incl - count;
    Stencil S;
#pragma unroll
    for (int r = 0; r < kStencilRanges; r++) S.pre[r] = __builtin_amdgcn_readlane(excl, r);
    S.delta = begin - excl;
    S.total = __builtin_amdgcn_readlane(incl, kStencilRanges - 1);
    return S;
}

// flat candidate index -> sorted slot.  The offsets are non-decreasing, so the range of k is the number of
// range starts <= k (empty ranges share a start with their successor, which is the one that counts).
// Must be called by ALL lanes of the wave (ds_bpermute returns 0 for a source lane that is masked off).
__device__ __forceinline__ int stencil_slot(const Stencil& S, int k) {
    int r = 0;
#pragma unroll
    for (int q = 1; q < kStencilRanges; q++) r += k >= S.pre[q] ? 1 : 0;
    return k + __builtin_amdgcn_ds_bpermute(r << 2, S.delta);
}

// The stencil of a grid of either width (g.m = 1: 3x3x3 cells, g.m = 2: 5x5x5 half-width cells) as one flat
// candidate space.  The ranges (rows of 2m+1 cells, split in two where they cross the periodic seam: at most 50)
// live one per lane; the range of a candidate is found per batch of 64 candidates by dropping "range r starts
// here" marks into a 64-int LDS strip and running a prefix maximum over the wave (the marks ascend), which costs
// the same ~20 instructions for 18 or 50 ranges -- the compare chain of stencil_slot costs 2 per range.
struct WideStencil {
    int pre;                      // lane r: first flat index of range r
    int count;                    // lane r: candidates in range r
    int delta;                    // lane r: begin_r - pre_r
    int total;                    // wave-uniform
};

__device__ __forceinline__ WideStencil gather_wide_stencil(const CellGrid& g, const int* __restrict__ cell_start, int cx, int cy,
                                                           int cz) {
    const int lane = lane_id();
    const int m = g.m, W = 2 * m + 1;
    int begin = 0, end = 0;
    if (lane < 2 * W * W) {
        const int row = lane >> 1, sub = lane & 1;
        const int rz = (row * (m == 1 ? 86 : 52)) >> 8;                // row / W for W = 3 (row < 9) or 5 (row < 25)
        int z = cz + rz - m, y = cy + (row - rz * W) - m;
        bool live = true;
        if (g.periodic) {                                              // every axis has at least W cells
            z += z < 0 ? g.nz : 0; z -= z >= g.nz ? g.nz : 0;
            y += y < 0 ? g.ny : 0; y -= y >= g.ny ? g.ny : 0;
        } else live = z >= 0 && z < g.nz && y >= 0 && y < g.ny;
        if (live) {
            const int rowbase = __mul24(__mul24(z, g.ny) + y, g.nx);      // (cells < 2^24)
            const int x0 = cx - m, x1 = cx + m;
            int a = 0, b = -1;                                         // cells [a, b] of this row
            if (sub == 0) { a = max(x0, 0); b = min(x1, g.nx - 1); }
            else if (g.periodic) {
                if (x0 < 0) { a = x0 + g.nx; b = g.nx - 1; }
                else if (x1 >= g.nx) { a = 0; b = x1 - g.nx; }
            }
            if (b >= a) { begin = cell_start[rowbase + a]; end = cell_start[rowbase + b + 1]; }
        }
    }
    WideStencil S;
    S.count = end - begin;
    const int incl = wave_prefix_sum(S.count);
    S.pre = incl - S.count;
    S.delta = begin - S.pre;
    S.total = __builtin_amdgcn_readlane(incl, 63);
    return S;
}

// Sorted slot of flat candidate base + lane (clamped to the last candidate).  `strip`: 64 ints of this wave's LDS;
// `carry`: wave-uniform state, 0 before the first batch; batches must be asked for in ascending order.  ALL lanes call.
__device__ __forceinline__ int wide_stencil_slot(const WideStencil& S, int base, int* strip, int& carry) {
    const int lane = lane_id();
    strip[lane] = 0;
    const int at = S.pre - base;
    if (S.count > 0 && at >= 0 && at < 64) strip[at] = lane + 1;       // LDS operations of a wave execute in order
    wave_fence();
    int x = strip[lane];
    wave_fence();
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));     // prefix maximum: row_shr 1, 2, 4, 8 ...
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));     // ... row_bcast:15 -> rows 1, 3
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));     // ... row_bcast:31 -> rows 2, 3
    x = max(x, carry);
    carry = __builtin_amdgcn_readlane(x, 63);
    const int k = min(base + lane, max(S.total - 1, 0));
    return k + __builtin_amdgcn_ds_bpermute(max(x - 1, 0) << 2, S.delta);
}

// Half-list variant: a consumer that wants only partners with a SMALLER atom id (getNeighborPairs: col < row)
// need not look at the others at all.  Inside a cell the sorted arrays ascend in atom id (order_cells /
// order_binned rank by id), so the partners of `row` in a cell are a PREFIX of that cell's run: 27 lanes
// binary-search their cell for the first id >= row, and the flat candidate space is the concatenation of
// the 27 prefixes -- half the candidates of the full stencil, none of them rejected for their id.
// (Round 4: counting the ids below `row` with independent loads of the whole cell instead of the four or five dependent loads of
//  the search was built and measured -- pairs_cells_stage 82 -> 142 us at 100 000 atoms: the kernel is bound by the number of
//  loads it issues, not by the length of that chain.)
constexpr int kStencilCells = 27;
struct PrefixStencil {
    int pre[kStencilCells];       // first flat index of cell r            (wave-uniform: SGPRs)
    int delta;                    // lane r: begin_r - pre[r]
    int total;
};

__device__ __forceinline__ PrefixStencil gather_prefix_stencil(const CellGrid& g, const int* __restrict__ cell_start,
                                                               const int* __restrict__ sorted_atom, int cx, int cy, int cz,
                                                               int row) {
    const int lane = lane_id();
    int begin = 0, count = 0;
    if (lane < kStencilCells) {
        int z = cz + lane / 9 - 1, y = cy + (lane / 3) % 3 - 1, x = cx + lane % 3 - 1;
        bool live = true;
        if (g.periodic) { z = (z + g.nz) % g.nz; y = (y + g.ny) % g.ny; x = (x + g.nx) % g.nx; }
        else live = z >= 0 && z < g.nz && y >= 0 && y < g.ny && x >= 0 && x < g.nx;
        if (live) {
            const int c = (z * g.ny + y) * g.nx + x;
            begin = cell_start[c];
            int lo = begin, hi = cell_start[c + 1];               // first slot in [lo, hi) whose id is >= row
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sorted_atom[mid] < row) lo = mid + 1;
                else hi = mid;
            }
            count = lo - begin;
        }
    }
    int incl = count;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {                               // 27 live lanes: five steps
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    const int excl = incl - count;
    PrefixStencil S;
#pragma unroll
    for (int r = 0; r < kStencilCells; r++) S.pre[r] = __builtin_amdgcn_readlane(excl, r);
    S.delta = begin - excl;
    S.total = __builtin_amdgcn_readlane(incl, kStencilCells - 1);
    return S;
}

// The same prefixes in the per-lane form of WideStencil: the range of a candidate is then found with wide_stencil_slot
// (LDS marks + prefix maximum, ~20 instructions) instead of the 26-step compare chain below, and nothing is read back
// lane by lane.  The wrap of the cell coordinates is a compare and an add, not three modulo operations by run-time values.
__device__ __forceinline__ WideStencil gather_prefix_stencil_wide(const CellGrid& g, const int* __restrict__ cell_start,
                                                                 const int* __restrict__ sorted_atom, int cx, int cy, int cz, int row) {
    const int lane = lane_id();
    int begin = 0, count = 0;
    if (lane < kStencilCells) {
        const int lz = lane / 9, ly = (lane / 3) % 3, lx = lane % 3;            // (constant divisors)
        int z = cz + lz - 1, y = cy + ly - 1, x = cx + lx - 1;
        bool live = true;
        if (g.periodic) {                                                      // every axis has at least 3 cells
            z += z < 0 ? g.nz : 0; z -= z >= g.nz ? g.nz : 0;
            y += y < 0 ? g.ny : 0; y -= y >= g.ny ? g.ny : 0;
            x += x < 0 ? g.nx : 0; x -= x >= g.nx ? g.nx : 0;
        } else live = z >= 0 && z < g.nz && y >= 0 && y < g.ny && x >= 0 && x < g.nx;
        if (live) {
            const int c = __mul24(__mul24(z, g.ny) + y, g.nx) + x;
            begin = cell_start[c];
            int lo = begin, hi = cell_start[c + 1];               // first slot in [lo, hi) whose id is >= row
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sorted_atom[mid] < row) lo = mid + 1;
                else hi = mid;
            }
            count = lo - begin;
        }
    }
    WideStencil S;
    S.count = count;
    const int incl = wave_prefix_sum(count);
    S.pre = incl - count;
    S.delta = begin - S.pre;
    S.total = __builtin_amdgcn_readlane(incl, 63);
    return S;
}

// (all lanes must call, like stencil_slot)
__device__ __forceinline__ int stencil_slot(const PrefixStencil& S, int k) {
    int r = 0;
#pragma unroll
    for (int q = 1; q < kStencilCells; q++) r += k >= S.pre[q] ? 1 : 0;
    return k + __builtin_amdgcn_ds_bpermute(r << 2, S.delta);
}

// Called by the kernel that consumes the grid (all of its threads, before any early exit): leaves the
// histogram of the two-kernel build zeroed for the next build.  `hist` may be NULL (five-kernel path).
__device__ __forceinline__ void clear_cell_histogram(int* __restrict__ hist) {
    if (hist == nullptr) return;
    const int stride = gridDim.x * blockDim.x;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < kHistWords; c += stride) hist[c] = 0;
}

// Host side: the buffers of one grid and the launch sequence.
struct CellBuffers {
    CellGrid* grid;
    int *cell_count, *cell_start;          // [max_cells], [max_cells + 1]
    int *atom_cell, *atom_rank;            // [N]
    int *unsorted_atom, *sorted_atom;      // [N]
    float4* sorted_pos;                    // [N]
    int max_cells;
    // two-kernel path (optional): a zero-initialised histogram of kHistWords ints that the consumer kernel
    // clears again after every build, and bins of kBinnedCells * bin_cap ints
    int* hist = nullptr;
    int* bins = nullptr;
    int bin_cap = 0;
    int fine = 0;                          // 1: prefer half-cutoff cells (decide_grid); only for consumers that read CellGrid::m
    int* sorted_cell = nullptr;            // [N] optional: cell of the atom in every sorted slot
    int* tile_total = nullptr;             // [max_cells / 8192 + 1] optional: lets grids of more than 8192 cells scan in parallel
};

static inline bool cell_build_is_binned(int N, bool periodic, const CellBuffers& b) {
    // (callers hand over hist / bins only for systems their bins are sized for: the stateful handles up to kBinnedAtoms atoms, with
    //  bins that grow in check(); getNeighborPairs up to kPairsBinnedAtoms with a fixed bin of 128 ids per cell)
    return periodic && b.hist != nullptr && b.bins != nullptr;
}

static inline void launch_cell_build(hipStream_t stream, int N, const float* pos, const float* box, bool periodic, float cutoff,
                                     const int* tag, const CellBuffers& b) {
    const int tb = 256, nb = (N + tb - 1) / tb;
    if (cell_build_is_binned(N, periodic, b)) {
        hipLaunchKernelGGL(bin_atoms, dim3(nb), dim3(kBinnedThreads), 0, stream, N, pos, box, cutoff, b.max_cells, b.grid, b.hist, b.bins,
                           b.bin_cap, b.atom_cell, b.fine);
        hipLaunchKernelGGL(order_binned<kBinnedThreads>, dim3(nb), dim3(kBinnedThreads), 0, stream, N, pos, tag, b.grid, b.hist, b.bins, b.bin_cap,
                           b.atom_cell, b.cell_start, b.sorted_atom, b.sorted_pos, b.sorted_cell);
        return;
    }
    // (the bounding box of a non-periodic system is reduced by ONE block; a periodic grid needs no reduction)
    hipLaunchKernelGGL(grid_setup, dim3(periodic ? 32 : 1), dim3(256), 0, stream, N, pos, box, (int)periodic, cutoff, b.max_cells, b.grid, b.cell_count, b.fine);
    hipLaunchKernelGGL(assign_cells, dim3(nb), dim3(tb), 0, stream, N, pos, b.grid, b.cell_count, b.atom_cell, b.atom_rank);
    if (b.max_cells <= kScanTile || b.tile_total == nullptr) {
        hipLaunchKernelGGL(scan_cells, dim3(1), dim3(1024), 0, stream, b.grid, b.cell_count, b.cell_start, (int*)nullptr);
    } else {
        const int ntiles = (b.max_cells + kScanTile - 1) / kScanTile;
        hipLaunchKernelGGL(scan_cells, dim3(ntiles), dim3(1024), 0, stream, b.grid, b.cell_count, b.cell_start, b.tile_total);
        hipLaunchKernelGGL(add_tile_offsets, dim3(ntiles), dim3(1024), 0, stream, b.grid, b.tile_total, b.cell_start);
    }
    hipLaunchKernelGGL(fill_cells, dim3(nb), dim3(tb), 0, stream, N, b.grid, b.cell_start, b.atom_cell, b.atom_rank, b.unsorted_atom);
    hipLaunchKernelGGL(order_cells, dim3(nb), dim3(tb), 0, stream, N, pos, b.grid, b.cell_start, b.atom_cell, b.unsorted_atom, tag,
                       b.sorted_atom, b.sorted_pos, b.sorted_cell);
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
