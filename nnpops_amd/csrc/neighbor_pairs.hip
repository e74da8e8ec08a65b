// neighbor_pairs.hip -- getNeighborPairs on gfx950 (C ABI: nnpops_neighbor_pairs_forward / _backward).
//
// Replaces the kernels behind the reference's `neighbors::getNeighborPairs` CUDA dispatch
// (src/pytorch/neighbors/getNeighborPairsCUDA.cu:31-101) and keeps their semantics:
//   * pair k of the lower triangle <-> (row, col < row); delta = pos[row] - pos[col]            (CUDA.cu:47-53)
//   * triclinic wrap: one round() per axis, z then y then x, dividing by the diagonal element   (CUDA.cu:54-63)
//   * a pair is kept when distance^2 <= cutoff^2                                               (CUDA.cu:66)
//   * every output slot is written: unused ones hold -1 / NaN / NaN                            (CUDA.cu:137-139)
//   * max_num_pairs == -1: slot index = pair index; > 0: compacted list, surplus pairs dropped, and
//     num_pairs reports the true number found either way                                       (CUDA.cu:68-78,163)
//
// What is different (design, not semantics):
//   * the compacted list is DETERMINISTIC.  A counting pass, an exclusive scan over rows and a fill pass
//     replace the reference's single pass with one global atomic per pair: pairs come out grouped by
//     `row` in ascending order.  Within a row the order is ascending `col` for the all-pairs search --
//     i.e. exactly the order of the reference's CPU implementation -- and stencil order for the cell search.
//   * for large systems the candidates come from the shared cell grid (celllist.h) instead of all
//     N(N-1)/2 pairs, which is what makes the 100 000-atom configuration possible at all (the
//     reference's int32 pair index overflows beyond ~65 000 atoms, CUDA.cu:129).
//   * one wave per row, lanes over the candidate columns, ballot compaction: coalesced position reads
//     and contiguous output writes.
#include <cmath>
#include <limits>

#include "celllist.h"
#include "host_common.h"

using namespace nnpops;

namespace {

template <typename T> struct Vec3 { T x, y, z; };

// rows are scanned in blocks of kScanBlock (scan_rows below): offset of a row = offset inside its block + block prefix
constexpr int kScanBlock = 1024;
__device__ __forceinline__ long long first_slot(const int* __restrict__ row_offset, const int* __restrict__ block_prefix, int row) {
    return (long long)row_offset[row] + block_prefix[row >> 10];
}

template <typename T>
__device__ __forceinline__ Vec3<T> wrapped_delta(const T* __restrict__ pos, int row, int col, const T* __restrict__ box,
                                                 bool periodic) {
    Vec3<T> d{pos[3 * row] - pos[3 * col], pos[3 * row + 1] - pos[3 * col + 1], pos[3 * row + 2] - pos[3 * col + 2]};
    if (periodic) {
        const T s3 = round(d.z / box[8]);
        d.x -= s3 * box[6]; d.y -= s3 * box[7]; d.z -= s3 * box[8];
        const T s2 = round(d.y / box[4]);
        d.x -= s2 * box[3]; d.y -= s2 * box[4];
        const T s1 = round(d.x / box[0]);
        d.x -= s1 * box[0];
    }
    return d;
}

template <typename T>
__global__ void fill_unused(long long num_slots, int32_t* __restrict__ neighbors, T* __restrict__ deltas,
                            T* __restrict__ distances, int32_t* __restrict__ num_pairs) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const T nan = std::numeric_limits<T>::quiet_NaN();
    if (k == 0) num_pairs[0] = 0;
    if (k >= num_slots) return;
    neighbors[k] = -1;
    neighbors[num_slots + k] = -1;
    deltas[3 * k] = nan; deltas[3 * k + 1] = nan; deltas[3 * k + 2] = nan;
    distances[k] = nan;
}

// ---- all-pairs search: one wave per row, lanes over columns < row --------------------------------
// PASS 0: count the row's pairs.  PASS 1: write them at row_offset[row] + rank (ascending col).
// ALL_SLOTS: write at the pair's own slot index instead (no offsets needed) and count.
template <typename T, int PASS, bool ALL_SLOTS>
__global__ __launch_bounds__(64) void pairs_allpairs(int N, const T* __restrict__ pos, const T* __restrict__ box,
                                                     int periodic, T cutoff2, long long num_slots,
                                                     int* __restrict__ row_count, const int* __restrict__ row_offset,
                                                     const int* __restrict__ block_prefix, int* __restrict__ ticket,
                                                     int32_t* __restrict__ neighbors, T* __restrict__ deltas,
                                                     T* __restrict__ distances, int32_t* __restrict__ num_pairs) {
    const int row = blockIdx.x;
    const int lane = lane_id();
    if (PASS == 0 && !ALL_SLOTS && row == 0 && lane == 0) *ticket = 0;     // the scan's ticket counter (workspace is not zeroed)
    long long base_slot = 0;
    if (ALL_SLOTS) base_slot = (long long)row * (row - 1) / 2;
    else if (PASS == 1) base_slot = first_slot(row_offset, block_prefix, row);
    int found = 0;
    for (int c0 = 0; c0 < row; c0 += 64) {
        const int col = c0 + lane;
        bool keep = false;
        Vec3<T> d{0, 0, 0};
        T d2 = 0;
        if (col < row) {
            d = wrapped_delta<T>(pos, row, col, box, periodic != 0);
            d2 = d.x * d.x + d.y * d.y + d.z * d.z;
            keep = !(d2 > cutoff2);
        }
        const unsigned long long m = __ballot(keep);
        if (keep && (PASS == 1 || ALL_SLOTS)) {
            const long long slot = ALL_SLOTS ? base_slot + col : base_slot + found + prefix_popc(m);
            if (slot < num_slots) {
                neighbors[slot] = row;
                neighbors[num_slots + slot] = col;
                deltas[3 * slot] = d.x; deltas[3 * slot + 1] = d.y; deltas[3 * slot + 2] = d.z;
                distances[slot] = sqrt(d2);
            }
        }
        found += __popcll(m);
    }
    if (lane == 0) {
        if (PASS == 0 && !ALL_SLOTS) row_count[row] = found;
        if (ALL_SLOTS && found) atomicAdd(num_pairs, found);
    }
}

// ---- cell-grid search: one wave per atom ("row"), candidates = stencil atoms with a smaller id --------
// STAGE pass: walks the stencil once (flat candidate space, celllist.h), counts the row's pairs and parks them
// in a per-row staging area {col, dx, dy, dz, dist}.  EMIT pass (after the scan over rows): copies the staged
// row to its final offset -- no second distance computation.  A row with more than kStageCap pairs, or a box
// too small for the stencil, is only counted by STAGE and recomputed by EMIT (walk_row with MODE = kEmit).
constexpr int kStageCap = 64;
constexpr int kPairsFineBinnedAtoms = 16384;    // with half-width cells the two-launch grid build (<= 8 192 cells) serves systems up to here
constexpr int kCellThreshold = 8192;        // below this the N^2/2 scan is cheaper than building a grid
enum { kStage = 0, kEmit = 1 };

template <typename T> struct Staged { T dx, dy, dz, dist; };
template <typename T> __device__ constexpr T kTieTol() { return sizeof(T) == 4 ? T(1e-6) : T(1e-14); }

template <typename T, int MODE>
__device__ __forceinline__ int walk_row(int row, const T* __restrict__ pos, const T* __restrict__ box, int periodic, T cutoff2,
                                        long long num_slots, long long base_slot, const CellGrid& g,
                                        const int* __restrict__ cell_start, const int* __restrict__ atom_cell,
                                        const int* __restrict__ sorted_atom, const float4* __restrict__ sorted_pos,
                                        int* __restrict__ st_col, Staged<T>* __restrict__ st_rec,
                                        int32_t* __restrict__ neighbors, T* __restrict__ deltas, T* __restrict__ distances) {
    const int lane = lane_id();
    int found = 0;
    const T inv_x = periodic ? T(1) / box[0] : T(0), inv_y = periodic ? T(1) / box[4] : T(0), inv_z = periodic ? T(1) / box[8] : T(0);
    // cutoff < 0.49 of the shortest box edge (a margin of 1 % where the tie test's is 1e-6): see visit()
    bool no_ties = false;
    if (periodic && !(periodic & 4)) {
        const T edge = fmin(fmin(fabs(box[0]), fabs(box[4])), fabs(box[8]));
        no_ties = cutoff2 < T(0.49 * 0.49) * edge * edge;
    }
    auto visit = [&](bool have, int col, Vec3<T> d) {
        bool keep = false;
        T d2 = 0;
        const bool active = have && col < row;
        if (periodic) {
            // round(v / b) of the reference (CUDA.cu:54-63), bit for bit, at the price of a multiply and v_rndne: the quotient
            // by reciprocal is within 2e-7 |q| (fp32) of the divided one, so unless it lies that close to a half-integer both
            // round to the same integer; a wave that sees such a candidate redoes the batch with the division (rare: uniform
            // branch).  Three IEEE divisions and three round() were 51 of the ~150 instructions per batch.
            // (round 5) No candidate that ends up INSIDE the cutoff can sit near a tie when the cutoff is safely below half of every
            // box edge: its reduced component is at most cutoff / edge < 1/2 - margin away from the integer it rounds to, on every
            // axis (the sequential z, y, x reduction of a triclinic cell included: the component left after each step is a component
            // of the final displacement).  A candidate that does sit near a tie is therefore outside the cutoff with either image --
            // rejected both ways, its displacement never written.  `no_ties` (wave-uniform, from the box) drops the test: 12 of the
            // ~100 vector instructions per batch of 64 candidates.
            const Vec3<T> d0 = d;
            bool near = false;
            auto rq = [&](T v, T inv) {
                const T q = v * inv, s = rint(q);
                if (!no_ties) near = near || (fabs(q - s) > T(0.5) - kTieTol<T>() * (fabs(q) + T(1)));
                return s;
            };
            const T s3 = rq(d.z, inv_z);
            d.x -= s3 * box[6]; d.y -= s3 * box[7]; d.z -= s3 * box[8];
            const T s2 = rq(d.y, inv_y);
            d.x -= s2 * box[3]; d.y -= s2 * box[4];
            const T s1 = rq(d.x, inv_x);
            d.x -= s1 * box[0];
            if ((periodic & 2) || (!no_ties && __any(active && near))) {           // (bit 1: $NNPOPS_PAIRS_DIVIDE=1, the division for every candidate)
                d = d0;
                const T e3 = round(d.z / box[8]);
                d.x -= e3 * box[6]; d.y -= e3 * box[7]; d.z -= e3 * box[8];
                const T e2 = round(d.y / box[4]);
                d.x -= e2 * box[3]; d.y -= e2 * box[4];
                const T e1 = round(d.x / box[0]);
                d.x -= e1 * box[0];
            }
        }
        if (active) {
            d2 = d.x * d.x + d.y * d.y + d.z * d.z;
            keep = !(d2 > cutoff2);
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int rank = found + prefix_popc(m);
            if (MODE == kStage) {
                if (rank < kStageCap) {
                    st_col[rank] = col;
                    st_rec[rank] = Staged<T>{d.x, d.y, d.z, (T)sqrt(d2)};
                }
            } else {
                const long long slot = base_slot + rank;
                if (slot < num_slots) {
                    neighbors[slot] = row;
                    neighbors[num_slots + slot] = col;
                    deltas[3 * slot] = d.x; deltas[3 * slot + 1] = d.y; deltas[3 * slot + 2] = d.z;
                    distances[slot] = sqrt(d2);
                }
            }
        }
        found += __popcll(m);
    };
    const T xr = pos[3 * row], yr = pos[3 * row + 1], zr = pos[3 * row + 2];
    auto from_col = [&](bool have, int col) {
        Vec3<T> d{0, 0, 0};
        if (have && col < row) d = Vec3<T>{xr - pos[3 * col], yr - pos[3 * col + 1], zr - pos[3 * col + 2]};
        visit(have, col, d);
    };
    if (!g.ok) {
        // the box is too small for the 27-cell stencil (fewer than 3 cells on an axis): scan every column
        for (int c0 = 0; c0 < row; c0 += 64) from_col(c0 + lane < row, c0 + lane);
        return found;
    }
    const int c = atom_cell[row];
    int cx, cy, cz;
    split_cell(g, c, cx, cy, cz);                          // (no integer division; exact: celllist.h)
    // Full-width cells (3 x 3 x 3): only partners with a smaller id -- the prefix of every stencil cell (celllist.h), half the
    // candidates of the full walk, at the price of 27 binary searches: four or five DEPENDENT loads in a kernel whose waves do
    // little else than wait for loads (12 occupancy rounds of ~6.5 us at 100 000 atoms: position -> cell -> cell starts -> search
    // -> candidates -> staged stores).  Half-width cells (5 x 5 x 5, round 5): the stencil holds 58 % of the volume, so walking ALL
    // of it and dropping the candidates with a larger id (`active` above) tests about as many as the prefixes of the coarse grid
    // did -- with no search: the chain is four round trips.
    const WideStencil st = g.m == 2 ? gather_wide_stencil(g, cell_start, cx, cy, cz)
                                    : gather_prefix_stencil_wide(g, cell_start, sorted_atom, cx, cy, cz, row);
    __shared__ int strips[4][64];                                          // (256-thread blocks: one strip per wave)
    int* strip = strips[threadIdx.x >> 6];
    int carry = 0;
    for (int base = 0; base < st.total; base += 64) {
        const int k = base + lane;
        const int slot = wide_stencil_slot(st, base, strip, carry);       // all lanes
        const bool have = k < st.total;
        if (sizeof(T) == 4) {
            // fp32: the grid's cell-ordered copy {x, y, z, id} is the same numbers, read coalesced
            const float4 pj = sorted_pos[slot];
            const int col = __float_as_int(pj.w) & kIdMask;
            visit(have, col, Vec3<T>{xr - (T)pj.x, yr - (T)pj.y, zr - (T)pj.z});
        } else {
            from_col(have, sorted_atom[slot]);
        }
    }
    return found;
}

template <typename T>
__global__ __launch_bounds__(256) void pairs_cells_stage(int N, const T* __restrict__ pos, const T* __restrict__ box, int periodic,
                                                         T cutoff2, const CellGrid* __restrict__ grid,
                                                         const int* __restrict__ cell_start, const int* __restrict__ atom_cell,
                                                         const int* __restrict__ sorted_atom, const float4* __restrict__ sorted_pos,
                                                         int* __restrict__ st_col, Staged<T>* __restrict__ st_rec,
                                                         int* __restrict__ row_count, int* __restrict__ ticket) {
    const int row = __builtin_amdgcn_readfirstlane(wave_global_id());      // (wave-uniform: the row header through the scalar cache)
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticket = 0;                  // the scan's ticket counter (workspace is not zeroed)
    if (row >= N) return;
    const CellGrid g = *grid;
    const int found = walk_row<T, kStage>(row, pos, box, periodic, cutoff2, 0, 0, g, cell_start, atom_cell, sorted_atom, sorted_pos,
                                          st_col + (size_t)row * kStageCap, st_rec + (size_t)row * kStageCap, nullptr, nullptr,
                                          nullptr);
    if (lane_id() == 0) row_count[row] = found;
}

template <typename T>
__global__ __launch_bounds__(256) void pairs_cells_emit(int N, const T* __restrict__ pos, const T* __restrict__ box, int periodic,
                                                        T cutoff2, long long num_slots, const CellGrid* __restrict__ grid,
                                                        const int* __restrict__ cell_start, const int* __restrict__ atom_cell,
                                                        const int* __restrict__ sorted_atom, const float4* __restrict__ sorted_pos,
                                                        const int* __restrict__ st_col, const Staged<T>* __restrict__ st_rec,
                                                        const int* __restrict__ row_count, const int* __restrict__ row_offset,
                                                        const int* __restrict__ block_prefix,
                                                        int32_t* __restrict__ neighbors, T* __restrict__ deltas,
                                                        T* __restrict__ distances) {
    // unused tail of the output: -1 / NaN (CUDA.cu:137-139), written once instead of pre-filling every slot
    {
        const long long found = block_prefix[(N + kScanBlock - 1) / kScanBlock];
        const T nan = std::numeric_limits<T>::quiet_NaN();
        const long long stride = (long long)gridDim.x * blockDim.x;
        for (long long k = found + (long long)blockIdx.x * blockDim.x + threadIdx.x; k < num_slots; k += stride) {
            neighbors[k] = -1;
            neighbors[num_slots + k] = -1;
            deltas[3 * k] = nan; deltas[3 * k + 1] = nan; deltas[3 * k + 2] = nan;
            distances[k] = nan;
        }
    }
    const int row = __builtin_amdgcn_readfirstlane(wave_global_id());      // (wave-uniform: the row header through the scalar cache)
    if (row >= N) return;
    const int lane = lane_id();
    const int n = row_count[row];
    const long long base_slot = first_slot(row_offset, block_prefix, row);
    const CellGrid g = *grid;
    if (n > kStageCap || !g.ok) {
        walk_row<T, kEmit>(row, pos, box, periodic, cutoff2, num_slots, base_slot, g, cell_start, atom_cell, sorted_atom, sorted_pos,
                           nullptr, nullptr, neighbors, deltas, distances);
        return;
    }
    if (lane < n) {
        const long long slot = base_slot + lane;
        if (slot < num_slots) {
            const Staged<T> r = st_rec[(size_t)row * kStageCap + lane];
            neighbors[slot] = row;
            neighbors[num_slots + slot] = st_col[(size_t)row * kStageCap + lane];
            deltas[3 * slot] = r.dx; deltas[3 * slot + 1] = r.dy; deltas[3 * slot + 2] = r.dz;
            distances[slot] = r.dist;
        }
    }
}

// Exclusive scan of row_count[0..N) in one launch of ceil(N/1024) blocks: every block scans its 1024 rows
// (row_offset = offset inside the block) and publishes its total; the block that finishes LAST (ticket counter,
// nobody waits for anybody) scans the block totals into block_prefix[0..nb] and sets num_pairs.  The offset of
// a row is row_offset[row] + block_prefix[row >> 10]  (first_slot() below).
__global__ __launch_bounds__(kScanBlock) void scan_rows(int N, const int* __restrict__ row_count, int* __restrict__ row_offset,
                                                        int* __restrict__ block_prefix, int* __restrict__ ticket,
                                                        int32_t* __restrict__ num_pairs) {
    __shared__ int wave_tot[kScanBlock / 64];
    __shared__ bool last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x * kScanBlock + tid;
    const int v = r < N ? row_count[r] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < kScanBlock / 64; w++) {
        before += w < wave ? wave_tot[w] : 0;
        total += wave_tot[w];
    }
    if (r < N) row_offset[r] = before + incl - v;
    if (tid == 0) {
        __hip_atomic_store(&block_prefix[blockIdx.x], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // the last block: exclusive scan of the block totals, in place (nb <= a few thousand: tiles of 1024)
    const int nb = gridDim.x;
    __shared__ int carry;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += kScanBlock) {
        const int bq = base + tid;
        const int t = bq < nb ? __hip_atomic_load(&block_prefix[bq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int inc = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(inc, off, 64);
            if (lane >= off) inc += up;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        int wb = 0;
        for (int w = 0; w < wave; w++) wb += wave_tot[w];
        const int excl = carry + wb + inc - t;
        if (bq < nb) block_prefix[bq] = excl;
        __syncthreads();
        if (tid == kScanBlock - 1) carry = excl + t;
        __syncthreads();
    }
    if (tid == 0) { block_prefix[nb] = carry; num_pairs[0] = carry; }
}

// (a launch of our own instead of hipMemsetAsync: the runtime's fill path costs the host ~20 us per call on this stack, a kernel ~3)
__global__ __launch_bounds__(256) void zero_words(long long n, int* __restrict__ p) {
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256) p[k] = 0;
}

template <typename T>
__global__ void to_float_positions(int n3, const T* __restrict__ in, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n3) out[k] = (float)in[k];
}

// ---- backward: forces of the pairs added up WITHOUT float atomics --------------------------------------------------
// The reference scatters six floating-point atomicAdds per pair (getNeighborPairsCUDA.cu:93-100): the sums depend on the order
// in which the hardware happens to serve them.  The op receives nothing but the four tensors -- any list, in any order, possibly
// edited by the caller -- so there is no row structure to rely on and no owner to gather.  Instead every contribution is turned
// into a FIXED-POINT number on one scale for the whole call and added with integer atomics: integer addition is associative, the
// result is the same bit pattern whatever the order (round 4; tests/test_neighbor_pairs_gpu.py::test_backward_bitwise_reproducible).
// Two 64-bit words per component (round 5): the first counts units of 2^-40 of the largest contribution of the call,
// the second units of 2^-80 (the exact remainder of the first rounding, rounded once more).  A float64 contribution therefore keeps
// all of its 53 bits unless it is more than 2^27 times smaller than the largest one of the call (float32: never loses any); with
// the single word of round 4 a pair in near contact coarsened every other atom's gradient to 1e-12 of ITS force.  2^22 terms fit
// either word.  A NaN / infinite contribution cannot enter an integer sum: it marks its two atoms, whose gradients come out NaN --
// the atoms the reference's atomicAdds poison (getNeighborPairsCUDA.cu:96-100) -- and nobody else's.  Launches: the largest |g| of
// the call (one integer atomicMax per block), the accumulation, the conversion back.
template <typename T>
__device__ __forceinline__ void pair_gradient(long long k, long long num_slots, const int32_t* __restrict__ neighbors,
                                              const T* __restrict__ deltas, const T* __restrict__ distances,
                                              const T* __restrict__ grad_deltas, const T* __restrict__ grad_distances, int& a, int& b,
                                              T (&g)[3]) {
    a = neighbors[k];
    b = -1;
    g[0] = g[1] = g[2] = T(0);
    if (a < 0) return;                                                   // CUDA.cu:93-94
    b = neighbors[num_slots + k];
    const T gd = grad_distances[k] / distances[k];
#pragma unroll
    for (int c = 0; c < 3; c++) g[c] = grad_deltas[3 * k + c] + deltas[3 * k + c] * gd;     // CUDA.cu:96-99
}

// scratch: [0] the bit pattern of the largest |g| (non-negative IEEE numbers order like integers), [1] unused,
// [2 .. 2 + 3N) the accumulators in units of 2^-40 of the scale, [2 + 3N .. 2 + 6N) in units of 2^-80, [2 + 6N .. 2 + 7N) the atoms
// that received a NaN / infinite contribution
template <typename T>
__global__ __launch_bounds__(256) void pairs_backward_max(long long num_slots, const int32_t* __restrict__ neighbors,
                                                          const T* __restrict__ deltas, const T* __restrict__ distances,
                                                          const T* __restrict__ grad_deltas, const T* __restrict__ grad_distances,
                                                          unsigned long long* __restrict__ scratch) {
    __shared__ double red[256 / 64];
    double m = 0.0;
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < num_slots; k += (long long)gridDim.x * 256) {
        int a, b;
        T g[3];
        pair_gradient(k, num_slots, neighbors, deltas, distances, grad_deltas, grad_distances, a, b, g);
        const double v = fmax(fabs((double)g[0]), fmax(fabs((double)g[1]), fabs((double)g[2])));
        if (v == v && v <= 1.7e308) m = fmax(m, v);                      // (NaN / inf contributions do not set the scale)
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / 64; w++) m = fmax(m, red[w]);
        atomicMax(&scratch[0], (unsigned long long)__double_as_longlong(m));
    }
}

__device__ __forceinline__ double pairs_fixed_scale(const unsigned long long* scratch) {
    const double m = __longlong_as_double((long long)scratch[0]);
    if (!(m > 0.0)) return 1.0;
    int e;
    frexp(m, &e);                                                        // m = f 2^e, 1/2 <= f < 1
    return ldexp(1.0, 40 - e);                                           // |g| scale < 2^40: 2^22 such terms fit an int64
}

template <typename T>
__global__ __launch_bounds__(256) void pairs_backward_accumulate(long long num_slots, const int32_t* __restrict__ neighbors,
                                                                 const T* __restrict__ deltas, const T* __restrict__ distances,
                                                                 const T* __restrict__ grad_deltas, const T* __restrict__ grad_distances,
                                                                 unsigned long long* __restrict__ scratch, int N) {
    // Lists as the forward op emits them are sorted by neighbors[0]: a wave of 64 consecutive slots holds two or three runs of one
    // atom each.  The lanes of a run add their fixed-point terms up across the wave (integers: still exact, still independent of
    // any order) and the run's first lane issues ONE atomic per word for that atom -- the other atom of every pair takes its own.
    // A list in any other order is runs of one: the same code, the same result.  (100 000 atoms, 2.6 M pairs, float32: 850 -> 400 us
    // per call, float64 1 600 -> 740: the pass is bound by the number of 64-bit atomics.)
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int a = -1, b = -1;
    T g[3] = {T(0), T(0), T(0)};
    if (k < num_slots) pair_gradient(k, num_slots, neighbors, deltas, distances, grad_deltas, grad_distances, a, b, g);
    const bool live = a >= 0;
    const double scale = pairs_fixed_scale(scratch);
    unsigned long long* acc = scratch + 2;
    unsigned long long* fine = acc + 3 * (size_t)N;
    unsigned long long* marked = fine + 3 * (size_t)N;
    long long q[3] = {0, 0, 0}, q2[3] = {0, 0, 0};
    if (live) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double v = (double)g[c] * scale;                       // |v| < 2^40 for every finite g
            if (v == v && fabs(v) < 9.0e18) {
                q[c] = __double2ll_rn(v);
                q2[c] = __double2ll_rn((v - (double)q[c]) * 1099511627776.0);      // (the remainder is exact; |q2| <= 2^39)
            } else {
                marked[a] = 1; marked[b] = 1;                            // (benign race: everyone writes the same value)
            }
        }
        // the other atom of the pair: its own atomics
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (q[c] != 0) atomicAdd(&acc[3 * (size_t)b + c], (unsigned long long)(-q[c]));
            if (q2[c] != 0) atomicAdd(&fine[3 * (size_t)b + c], (unsigned long long)(-q2[c]));
        }
    }
    // runs of equal neighbors[0] inside the wave: [lane, end) is what is left of this lane's run
    const int before = __shfl_up(a, 1, 64);
    const unsigned long long heads = __ballot(lane == 0 || a != before);
    const unsigned long long later = lane == 63 ? 0ull : heads >> (lane + 1);
    const int end = later ? lane + 1 + __builtin_ctzll(later) : 64;
    const bool any_fine = __ballot(q2[0] != 0 || q2[1] != 0 || q2[2] != 0) != 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const bool take = lane + off < end;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const long long o = __shfl_down(q[c], off, 64);
            if (take) q[c] += o;
        }
        if (any_fine) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const long long o = __shfl_down(q2[c], off, 64);
                if (take) q2[c] += o;
            }
        }
    }
    if (live && (heads >> lane & 1)) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (q[c] != 0) atomicAdd(&acc[3 * (size_t)a + c], (unsigned long long)q[c]);
            if (q2[c] != 0) atomicAdd(&fine[3 * (size_t)a + c], (unsigned long long)q2[c]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pairs_backward_finish(int N, const unsigned long long* __restrict__ scratch, T* __restrict__ grad_positions) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= 3 * N) return;
    const unsigned long long* acc = scratch + 2;
    const unsigned long long* fine = acc + 3 * (size_t)N;
    const bool poisoned = fine[3 * (size_t)N + k / 3] != 0;
    const double inv = 1.0 / pairs_fixed_scale(scratch);
    const double v = ((double)(long long)acc[k] + (double)(long long)fine[k] * (1.0 / 1099511627776.0)) * inv;
    grad_positions[k] = poisoned ? (T)NAN : (T)v;
}

// workspace layout (bytes): row_count[N+1] | row_offset[N+1] | cell grid arrays | float positions | row staging
struct Workspace {
    int* row_count;
    int* row_offset;
    int* block_prefix;    // [ceil(N/1024) + 1] + the scan's ticket counter behind it
    CellGrid* grid;
    int* cell_count;
    int* cell_start;
    int* atom_cell;
    int* atom_rank;
    int* unsorted_atom;
    int* sorted_atom;
    float4* sorted_pos;
    float* fpos;
    int* st_col;          // [N][kStageCap]
    void* st_rec;         // [N][kStageCap] Staged<T> (sized for double)
    int* hist;            // [kHistWords] two-launch grid build (periodic systems of up to kPairsBinnedAtoms atoms), zeroed by every call
    int* bins;            // [kBinnedCells][kPairsBinCap]
    int* tile_total;      // [max_cells / kScanTile + 2] partial sums of the tiled cell scan (grids of more than 8 192 cells)
    int max_cells;
};

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

size_t carve(Workspace* w, char* base, int N) {
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    const int max_cells = N + 4096;
    Workspace tmp;
    tmp.max_cells = max_cells;
    tmp.row_count = (int*)take(sizeof(int) * ((size_t)N + 1));
    tmp.row_offset = (int*)take(sizeof(int) * ((size_t)N + 1));
    tmp.block_prefix = (int*)take(sizeof(int) * ((size_t)N / 1024 + 4));
    tmp.grid = (CellGrid*)take(sizeof(CellGrid));
    tmp.cell_count = (int*)take(sizeof(int) * (size_t)max_cells);
    tmp.cell_start = (int*)take(sizeof(int) * ((size_t)max_cells + 1));
    tmp.atom_cell = (int*)take(sizeof(int) * (size_t)N);
    tmp.atom_rank = (int*)take(sizeof(int) * (size_t)N);
    tmp.unsorted_atom = (int*)take(sizeof(int) * (size_t)N);
    tmp.sorted_atom = (int*)take(sizeof(int) * (size_t)N);
    tmp.sorted_pos = (float4*)take(sizeof(float4) * (size_t)N);
    tmp.fpos = (float*)take(sizeof(float) * 3 * (size_t)N);
    const bool staged = N >= kCellThreshold;          // only the cell-grid path stages rows
    tmp.st_col = (int*)take(staged ? sizeof(int) * (size_t)N * kStageCap : 0);
    tmp.st_rec = (void*)take(staged ? sizeof(Staged<double>) * (size_t)N * kStageCap : 0);
    const bool binned = staged && N <= kPairsBinnedAtoms;
    tmp.hist = (int*)take(binned ? sizeof(int) * kHistWords : 0);
    tmp.bins = (int*)take(binned ? sizeof(int) * (size_t)kBinnedCells * kPairsBinCap : 0);
    tmp.tile_total = (int*)take(sizeof(int) * ((size_t)max_cells / kScanTile + 2));
    if (w) *w = tmp;
    return off;
}

template <typename T>
int forward_impl(int N, const T* pos, const T* box, double cutoff, long long max_num_pairs, int32_t* neighbors, T* deltas,
                 T* distances, int32_t* num_pairs, void* workspace, hipStream_t stream) {
    const bool all_slots = max_num_pairs == -1;
    const long long num_slots = all_slots ? (long long)N * (N - 1) / 2 : max_num_pairs;
    const int periodic = box != nullptr;
    const T c = (T)cutoff;
    const T cutoff2 = c * c;
    Workspace w;
    carve(&w, (char*)workspace, N);
    const int nscan = div_up(N, kScanBlock);
    int* ticket = w.block_prefix + nscan + 1;
    // The cell grid pays off once the N^2/2 scan is the bigger cost.  A periodic box must be at least 3
    // cells wide per axis for the stencil, which the caller's contract (box >= 2*cutoff) does not
    // guarantee: the device checks, and rows fall back to scanning every column when it is not.
    const bool use_cells = !all_slots && N >= kCellThreshold;
    if (num_slots > 0 && !use_cells) {        // (the cell path writes the unused tail itself, once)
        const int tb = 256;
        hipLaunchKernelGGL(fill_unused<T>, dim3(div_up(num_slots, tb)), dim3(tb), 0, stream, num_slots, neighbors, deltas,
                           distances, num_pairs);
    } else if (!use_cells) {
        NNPOPS_HIP_TRY(hipMemsetAsync(num_pairs, 0, sizeof(int32_t), stream));
    }
    if (N < 2) return NNPOPS_OK;
    if (all_slots) {
        hipLaunchKernelGGL((pairs_allpairs<T, 1, true>), dim3(N), dim3(64), 0, stream, N, pos, box, periodic, cutoff2, num_slots,
                           w.row_count, w.row_offset, w.block_prefix, ticket, neighbors, deltas, distances, num_pairs);
        NNPOPS_HIP_TRY(hipGetLastError());
        return NNPOPS_OK;
    }
    if (use_cells) {
        const int tb = 256;
        const float* fpos = (const float*)pos;
        if (sizeof(T) == 8) {
            hipLaunchKernelGGL(to_float_positions<T>, dim3(div_up(3 * N, tb)), dim3(tb), 0, stream, 3 * N, pos, w.fpos);
            fpos = w.fpos;
        }
        const float* fbox = nullptr;
        float* fbox_dev = (float*)w.sorted_pos;      // reuse: sorted_pos is not needed by the pair kernels
        if (periodic) {
            if (sizeof(T) == 8) {
                hipLaunchKernelGGL(to_float_positions<T>, dim3(1), dim3(64), 0, stream, 9, box, fbox_dev);
                fbox = fbox_dev;
            } else {
                fbox = (const float*)box;
            }
        }
        // (the grid also emits cell-ordered positions; they land in scratch this op does not otherwise use)
        CellBuffers cb{w.grid, w.cell_count, w.cell_start, w.atom_cell, w.atom_rank, w.unsorted_atom, w.sorted_atom,
                       w.sorted_pos, w.max_cells};
        cb.tile_total = w.tile_total;
        // A periodic system of up to kPairsBinnedAtoms atoms takes the two-launch grid of the stateful handles (celllist.h:
        // bin_atoms + order_binned) behind ONE memset of its 32 KiB histogram -- three launches where grid_setup / assign_cells /
        // scan_cells / fill_cells / order_cells are five (round 4; the workspace is the caller's and arrives dirty, so the
        // histogram cannot be left clean by the previous call as the handles do).  A cell with more than kPairsBinCap atoms
        // (nine times liquid density at the usual cell size) clears grid.ok: every row then scans all columns -- correct, slow.
        // Half-width cells where they fit (walk_row: no binary searches).  The two-launch build holds at most kBinnedCells cells:
        // at liquid density a half-cutoff grid of that size is ~14 000 atoms, so larger systems take the five-launch build (its
        // scan is tiled) -- the grid costs ~6 us more there and pairs_cells_stage ~35 us less (100 000 atoms).
        const bool fine = !(std::getenv("NNPOPS_PAIRS_FINE_GRID") && std::atoi(std::getenv("NNPOPS_PAIRS_FINE_GRID")) == 0);
        cb.fine = fine ? 1 : 0;
        if (periodic && N <= (fine ? kPairsFineBinnedAtoms : kPairsBinnedAtoms)) {
            hipLaunchKernelGGL(zero_words, dim3(div_up(kHistWords, 256)), dim3(256), 0, stream, (long long)kHistWords, w.hist);
            cb.hist = w.hist; cb.bins = w.bins; cb.bin_cap = kPairsBinCap;
        }
        launch_cell_build(stream, N, fpos, fbox, periodic != 0, (float)cutoff, nullptr, cb);
        const dim3 rgrid(div_up(N, 4)), rblock(256);       // one wave per row
        Staged<T>* st_rec = (Staged<T>*)w.st_rec;
        const int periodic_flags = periodic | ((periodic && std::getenv("NNPOPS_PAIRS_DIVIDE") && std::atoi(std::getenv("NNPOPS_PAIRS_DIVIDE"))) ? 2 : 0) |
                                   ((periodic && std::getenv("NNPOPS_PAIRS_TIE_TEST") && std::atoi(std::getenv("NNPOPS_PAIRS_TIE_TEST"))) ? 4 : 0);      // (4: keep the tie test, A/B)
        hipLaunchKernelGGL(pairs_cells_stage<T>, rgrid, rblock, 0, stream, N, pos, box, periodic_flags, cutoff2, w.grid, w.cell_start,
                           w.atom_cell, w.sorted_atom, w.sorted_pos, w.st_col, st_rec, w.row_count, ticket);
        hipLaunchKernelGGL(scan_rows, dim3(nscan), dim3(kScanBlock), 0, stream, N, w.row_count, w.row_offset, w.block_prefix, ticket,
                           num_pairs);
        hipLaunchKernelGGL(pairs_cells_emit<T>, rgrid, rblock, 0, stream, N, pos, box, periodic_flags, cutoff2, num_slots, w.grid,
                           w.cell_start, w.atom_cell, w.sorted_atom, w.sorted_pos, w.st_col, st_rec, w.row_count, w.row_offset,
                           w.block_prefix, neighbors, deltas, distances);
    } else {
        hipLaunchKernelGGL((pairs_allpairs<T, 0, false>), dim3(N), dim3(64), 0, stream, N, pos, box, periodic, cutoff2, num_slots,
                           w.row_count, w.row_offset, w.block_prefix, ticket, neighbors, deltas, distances, num_pairs);
        hipLaunchKernelGGL(scan_rows, dim3(nscan), dim3(kScanBlock), 0, stream, N, w.row_count, w.row_offset, w.block_prefix, ticket,
                           num_pairs);
        hipLaunchKernelGGL((pairs_allpairs<T, 1, false>), dim3(N), dim3(64), 0, stream, N, pos, box, periodic, cutoff2, num_slots,
                           w.row_count, w.row_offset, w.block_prefix, ticket, neighbors, deltas, distances, num_pairs);
    }
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

template <typename T>
int backward_impl(int N, long long num_slots, const int32_t* neighbors, const T* deltas, const T* distances,
                  const T* grad_deltas, const T* grad_distances, T* grad_positions, void* workspace, hipStream_t stream) {
    unsigned long long* scratch = (unsigned long long*)workspace;
    {
        const long long words = 2 * (2 + 7 * (long long)N);
        hipLaunchKernelGGL(zero_words, dim3((unsigned)std::min<long long>(div_up(words, 256), 4096)), dim3(256), 0, stream, words, (int*)scratch);
    }
    if (num_slots > 0) {
        const int nb_max = (int)std::min<long long>(div_up(num_slots, 256), 2048);
        hipLaunchKernelGGL(pairs_backward_max<T>, dim3(nb_max), dim3(256), 0, stream, num_slots, neighbors, deltas, distances, grad_deltas,
                           grad_distances, scratch);
        hipLaunchKernelGGL(pairs_backward_accumulate<T>, dim3(div_up(num_slots, 256)), dim3(256), 0, stream, num_slots, neighbors, deltas,
                           distances, grad_deltas, grad_distances, scratch, N);
    }
    hipLaunchKernelGGL(pairs_backward_finish<T>, dim3(div_up(3 * N, 256)), dim3(256), 0, stream, N, scratch, grad_positions);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

}  // namespace

extern "C" {

int64_t nnpops_neighbor_pairs_workspace_bytes(int num_atoms) {
    if (num_atoms < 0) return 0;
    return (int64_t)carve(nullptr, nullptr, num_atoms) + 256;
}

int nnpops_neighbor_pairs_forward(int dtype, int num_atoms, const void* positions, const void* box, double cutoff,
                                  int64_t max_num_pairs, int32_t* neighbors, void* deltas, void* distances,
                                  int32_t* num_pairs, void* workspace, void* stream) {
    NNPOPS_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (float32) or 1 (float64)");
    NNPOPS_REQUIRE(num_atoms > 0, "Expected the 1nd dimension size of \"positions\" to be more than 0");
    NNPOPS_REQUIRE(cutoff > 0, "Expected \"cutoff\" to be positive");
    NNPOPS_REQUIRE(max_num_pairs > 0 || max_num_pairs == -1, "Expected \"max_num_pairs\" to be positive or equal to -1");
    const long long slots = max_num_pairs == -1 ? (long long)num_atoms * (num_atoms - 1) / 2 : max_num_pairs;
    NNPOPS_REQUIRE(positions && num_pairs && workspace, "NULL device pointer");
    NNPOPS_REQUIRE(slots == 0 || (neighbors && deltas && distances), "NULL output pointer");
    NNPOPS_REQUIRE(max_num_pairs != -1 || num_atoms <= 65536,
                   "max_num_pairs == -1 needs one slot per pair; beyond 65536 atoms use a compacted list");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0)
        return forward_impl<float>(num_atoms, (const float*)positions, (const float*)box, cutoff, max_num_pairs, neighbors,
                                   (float*)deltas, (float*)distances, num_pairs, workspace, s);
    return forward_impl<double>(num_atoms, (const double*)positions, (const double*)box, cutoff, max_num_pairs, neighbors,
                                (double*)deltas, (double*)distances, num_pairs, workspace, s);
}

int64_t nnpops_neighbor_pairs_backward_workspace_bytes(int num_atoms) {
    return num_atoms < 0 ? 0 : (int64_t)sizeof(unsigned long long) * (2 + 7 * (int64_t)num_atoms);
}

int nnpops_neighbor_pairs_backward_ws(int dtype, int num_atoms, int64_t num_slots, const int32_t* neighbors, const void* deltas,
                                      const void* distances, const void* grad_deltas, const void* grad_distances,
                                      void* grad_positions, void* workspace, void* stream) {
    NNPOPS_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (float32) or 1 (float64)");
    NNPOPS_REQUIRE(num_atoms > 0 && num_slots >= 0, "bad sizes");
    NNPOPS_REQUIRE(grad_positions != nullptr && workspace != nullptr, "NULL device pointer");
    NNPOPS_REQUIRE(((uintptr_t)workspace & 7) == 0, "the workspace must be 8-byte aligned");
    NNPOPS_REQUIRE(num_slots == 0 || (neighbors && deltas && distances && grad_deltas && grad_distances), "NULL device pointer");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0)
        return backward_impl<float>(num_atoms, num_slots, neighbors, (const float*)deltas, (const float*)distances,
                                    (const float*)grad_deltas, (const float*)grad_distances, (float*)grad_positions, workspace, s);
    return backward_impl<double>(num_atoms, num_slots, neighbors, (const double*)deltas, (const double*)distances,
                                 (const double*)grad_deltas, (const double*)grad_distances, (double*)grad_positions, workspace, s);
}

int nnpops_neighbor_pairs_backward(int dtype, int num_atoms, int64_t num_slots, const int32_t* neighbors, const void* deltas,
                                   const void* distances, const void* grad_deltas, const void* grad_distances,
                                   void* grad_positions, void* stream) {
    // (the entry point of rounds 1-3, kept: takes its scratch from the stream-ordered allocator)
    NNPOPS_REQUIRE(num_atoms > 0, "bad sizes");
    void* ws = nullptr;
    NNPOPS_HIP_TRY(hipMallocAsync(&ws, (size_t)nnpops_neighbor_pairs_backward_workspace_bytes(num_atoms), (hipStream_t)stream));
    const int rc = nnpops_neighbor_pairs_backward_ws(dtype, num_atoms, num_slots, neighbors, deltas, distances, grad_deltas, grad_distances,
                                                     grad_positions, ws, stream);
    (void)hipFreeAsync(ws, (hipStream_t)stream);
    return rc;
}

}  // extern "C"
