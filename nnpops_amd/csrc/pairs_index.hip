// pairs_index.hip -- getNeighborPairs backward WITHOUT atomics: a transposed index of the pair list + an owner-computes gather.
//
// What is computed: reference src/pytorch/neighbors/getNeighborPairsCUDA.cu:80-101 -- for every pair k of the list
//     g = grad_deltas[k] + deltas[k] / distances[k] * grad_distances[k],   grad_positions[neighbors[0][k]] += g,   [neighbors[1][k]] -= g
// The reference scatters six floating-point atomicAdds per pair.  Rounds 4-5 replaced them by 64-bit fixed-point integer atomics
// (order independent, neighbor_pairs.hip: kept for lists of unknown origin); that pass is bound by its atomics, which this memory
// system executes behind the L2s at ~25 G/s: 0.44 ms per call at 100 000 atoms (2.95 M pairs), 3.4 x the forward op.
//
// A list the forward op emitted is GROUPED by neighbors[0] (rows ascending, neighbor_pairs.hip), so the first side of every pair
// is a segmented sum with one owner per atom.  For the second side the forward op builds, when the positions require a gradient,
// the TRANSPOSED index of the list -- the slots sorted by neighbors[1] (stable: ascending slot inside an atom's group) -- and hands
// it to autograd with the other saved tensors.  The backward is then
//     pairs_backward_terms    k -> G[k] = {g, 0}                 one streaming pass, 16-byte records
//     pairs_backward_gather   atom i (16 lanes): sum of G over its row segment (contiguous) minus the sum of G[order[p]] over its
//                             column segment (one 16-byte gather per pair), added up in float64 in a fixed order
// -- no atomics, no fixed point, no pass for the scale; bitwise reproducible; a NaN / infinite contribution reaches exactly the two
// atoms of its pair, as the reference's atomics do.
//
// The sort by column is written for this job (round 6; measured first: rocprim::radix_sort_pairs, 141 us for 3 M slots at 100 000
// atoms -- three look-back passes whose chain of 730 tiles is all latency at this size, plus seven fill launches -- and a plain LSD
// sort of 9-bit digits, 164 us: its 4-byte stores to 512 streams per tile reach the eight L2s as partial lines).  Two levels:
//   1. pairs_index_partition   the slots, as 8-byte records {column, slot}, into BUCKETS of 512 consecutive atoms (<= 512 buckets:
//                              262 144 atoms), stably: a histogram per tile of 4 096 slots (pairs_index_histogram), one scan launch
//                              (a workgroup per bucket over the tiles), and a scatter whose ranks come from wave-wide matches (nine
//                              ballots) ordered through a small LDS table.  Tiles are dealt to the XCDs in contiguous runs, so a
//                              bucket's region is written in eight contiguous pieces, each by one L2.
//   2. pairs_index_bucket      one workgroup per bucket sorts its records by atom -- counts per (wave, atom) in LDS, one scan, a stable
//                              scatter confined to the bucket's own 50 KB of the output -- and writes the atoms' column segments.
// Unused slots (-1) are dropped, not sorted: a list the forward op emitted keeps them behind the pairs, and a tile that starts
// with one is skipped.
#include "device_common.h"
#include "host_common.h"

using namespace nnpops;

namespace {

constexpr int kMaxBucketShift = 9, kMaxBucketAtoms = 1 << kMaxBucketShift;      // atoms per bucket (= bins of the second level): 2^shift, shift <= 9, picked per call
constexpr int kMaxBuckets = 512;                                       // bins of the first level
static_assert(kMaxBuckets * kMaxBucketAtoms == NNPOPS_PAIRS_INDEX_MAX_ATOMS, "include/nnpops_hip.h states the limit");
constexpr int kTile = 4096, kTileThreads = 256;      // slots per tile; a wave takes 1 024 consecutive ones in 16 rounds of 64
constexpr int kBucketWaves = 16;

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// workgroup b runs on XCD b % 8: give every XCD a contiguous eighth of the tiles (device_common.h: xcd_contiguous_wave_id)
__device__ __forceinline__ int xcd_contiguous_tile() {
    const int g = blockIdx.x, n = gridDim.x, xcd = g & 7;
    int start = 0;
    for (int c = 0; c < xcd; c++) start += (n - c + 7) >> 3;
    return start + (g >> 3);
}

__global__ __launch_bounds__(kTileThreads) void pairs_index_histogram(long long num_slots, int num_atoms, int shift, const int* __restrict__ cols, int ntiles,
                                                                     int* __restrict__ hist, int2* __restrict__ row_seg) {
    __shared__ int bins[kMaxBuckets];
    const int tid = threadIdx.x, tile = xcd_contiguous_tile();
    for (long long i = (long long)blockIdx.x * kTileThreads + tid; i < num_atoms; i += (long long)gridDim.x * kTileThreads) row_seg[i] = make_int2(0, 0);
    bins[tid] = 0; bins[tid + kTileThreads] = 0;
    __syncthreads();
    const long long base = (long long)tile * kTile;
    if (base < num_slots && cols[base] >= 0) {               // (an unused slot at the head of a tile: nothing but unused slots behind it)
#pragma unroll
        for (int r = 0; r < kTile / kTileThreads; r++) {
            const long long k = base + r * kTileThreads + tid;
            if (k < num_slots) {
                const int c = cols[k];
                if ((unsigned)c < (unsigned)num_atoms) atomicAdd(&bins[c >> shift], 1);      // (an id outside the system is treated as unused, never as a bin)      // (LDS, integer: the counts do not depend on the order)
            }
        }
    }
    __syncthreads();
    hist[(size_t)tid * ntiles + tile] = bins[tid];
    hist[(size_t)(tid + kTileThreads) * ntiles + tile] = bins[tid + kTileThreads];
}

// One workgroup per bucket: exclusive scan of its counts over the tiles (in place) and the bucket's total.
__global__ __launch_bounds__(256) void pairs_index_scan(int ntiles, int* __restrict__ hist, int* __restrict__ totals) {
    __shared__ int wave_tot[4];
    __shared__ int carry_s;
    int* row = hist + (size_t)blockIdx.x * ntiles;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += 256) {
        const int t = base + tid;
        const int v = t < ntiles ? row[t] : 0;
        const int incl = wave_prefix_sum(v);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; w++) before += wave_tot[w];
        if (t < ntiles) row[t] = before + incl - v;
        __syncthreads();
        if (tid == 255) carry_s = before + incl;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = carry_s;
}

// Exclusive scan of the 512 bucket totals by 256 threads (two per thread) into LDS; returns the grand total in every thread.
__device__ __forceinline__ int scan_bucket_totals(const int* __restrict__ totals, int* bucket_base, int* wave_tot) {
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    int incl = 0, a = 0, b = 0;
    if (threadIdx.x < 256) {
        a = totals[2 * tid]; b = totals[2 * tid + 1];
        incl = wave_prefix_sum(a + b);
        if (lane == 63) wave_tot[wave] = incl;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        int before = 0;
        for (int w = 0; w < wave; w++) before += wave_tot[w];
        bucket_base[2 * tid] = before + incl - a - b;
        bucket_base[2 * tid + 1] = before + incl - b;
    }
    __syncthreads();
    return wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

__global__ __launch_bounds__(kTileThreads) void pairs_index_partition(long long num_slots, int num_atoms, int shift, const int* __restrict__ rows,
                                                                     const int* __restrict__ cols, int ntiles, const int* __restrict__ hist,
                                                                     const int* __restrict__ totals, int2* __restrict__ records,
                                                                     int2* __restrict__ row_seg) {
    __shared__ int bucket_base[kMaxBuckets];
    __shared__ int wave_cnt[4][kMaxBuckets];
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = xcd_contiguous_tile();
#pragma unroll
    for (int w = 0; w < 4; w++) { wave_cnt[w][tid] = 0; wave_cnt[w][tid + kTileThreads] = 0; }
    scan_bucket_totals(totals, bucket_base, wave_tot);
    // where this tile's records of every bucket go: the bucket's start + what the tiles before this one put there
    bucket_base[tid] += hist[(size_t)tid * ntiles + tile];
    bucket_base[tid + kTileThreads] += hist[(size_t)(tid + kTileThreads) * ntiles + tile];
    const long long base = (long long)tile * kTile;
    if (base >= num_slots) return;                             // (uniform)
    // the row segments: equal rows are contiguous in a list the forward op emitted
    for (int r = 0; r < kTile / kTileThreads; r++) {
        const long long k = base + r * kTileThreads + tid;
        if (k >= num_slots) break;
        const int a = rows[k];
        if ((unsigned)a < (unsigned)num_atoms) {
            const int before = k > 0 ? rows[k - 1] : -2, after = k + 1 < num_slots ? rows[k + 1] : -2;
            if (a != before) row_seg[a].x = (int)k;
            if (a != after) row_seg[a].y = (int)(k + 1);
        }
    }
    if (cols[base] < 0) return;                                // (uniform: nothing but unused slots from here on)
    // wave w takes slots [base + 1024 w, base + 1024 (w + 1)) in sixteen rounds of 64: (wave, round, lane) is slot order
    constexpr int R = kTile / 4 / 64;
    int key[R], rank[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const long long k = base + wave * (kTile / 4) + r * 64 + lane;
        key[r] = k < num_slots ? cols[k] : -1;
        if ((unsigned)key[r] >= (unsigned)num_atoms) key[r] = -1;      // (as the histogram counted)
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const bool used = key[r] >= 0;
        const int d = used ? key[r] >> shift : 0;
        unsigned long long peers = __ballot(used);
#pragma unroll
        for (int b = 0; b < 9; b++) {
            const unsigned long long m = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? m : ~m;
        }
        const int old = used ? wave_cnt[wave][d] : 0;          // (LDS operations of a wave execute in order: everybody reads, then one lane writes)
        const int ahead = prefix_popc(peers);
        rank[r] = old + ahead;
        if (used && ahead == 0) wave_cnt[wave][d] = old + __popcll(peers);
        wave_fence();                                          // (the next round reads what this one wrote)
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (key[r] < 0) continue;
        const int d = key[r] >> shift;
        int at = bucket_base[d] + rank[r];
        for (int w = 0; w < wave; w++) at += wave_cnt[w][d];
        records[at] = make_int2(key[r], (int)(base + wave * (kTile / 4) + r * 64 + lane));
    }
}

// One workgroup of 16 waves per bucket: wave w takes the w-th sixteenth of the bucket's records (slot order), counts them per atom in
// its own LDS row, one scan puts the (atom, wave) runs in order, and a second sweep over the same records scatters the slots -- stable,
// every atom's slots ascending -- inside the bucket's own stretch of the output.
__global__ __launch_bounds__(64 * kBucketWaves) void pairs_index_bucket(int num_atoms, int shift, const int* __restrict__ totals, const int2* __restrict__ records,
                                                                       int* __restrict__ order, int2* __restrict__ col_seg) {
    __shared__ int cnt[kBucketWaves][kMaxBucketAtoms];          // counts, then cursors, of (wave, atom)
    const int kBucketAtoms = 1 << shift;
    __shared__ int bucket_base[kMaxBuckets];
    __shared__ int wave_tot[kBucketWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, bucket = blockIdx.x;
    for (int q = tid; q < kBucketWaves * kMaxBucketAtoms; q += 64 * kBucketWaves) (&cnt[0][0])[q] = 0;
    scan_bucket_totals(totals, bucket_base, wave_tot);           // (ends with a barrier)
    const int first = bucket_base[bucket], n = totals[bucket];
    const int per_wave = ((n + kBucketWaves - 1) / kBucketWaves + 63) & ~63;      // whole rounds of 64
    const int lo = min(n, wave * per_wave), hi = min(n, lo + per_wave);
    auto sweep = [&](auto second) {
        constexpr int GROUP = 4;                                 // rounds requested together: a wave's time here is its round trips to the L2
        for (int k0 = lo; k0 < hi; k0 += 64 * GROUP) {
            int2 rec[GROUP];
#pragma unroll
            for (int j = 0; j < GROUP; j++) {
                const int k = k0 + 64 * j + lane;
                rec[j] = k < hi ? records[first + k] : make_int2(0, 0);
            }
#pragma unroll
            for (int j = 0; j < GROUP; j++) {
                if (k0 + 64 * j >= hi) break;                    // (uniform)
                const bool used = k0 + 64 * j + lane < hi;
                const int d = rec[j].x & (kBucketAtoms - 1);
                unsigned long long peers = __ballot(used);
#pragma unroll
                for (int b = 0; b < kMaxBucketShift; b++) {
                    const unsigned long long m = __ballot((d >> b) & 1);
                    peers &= ((d >> b) & 1) ? m : ~m;
                }
                const int old = used ? cnt[wave][d] : 0;
                const int ahead = prefix_popc(peers);
                if (decltype(second)::value && used) order[first + old + ahead] = rec[j].y;
                if (used && ahead == 0) cnt[wave][d] = old + __popcll(peers);
                wave_fence();
            }
        }
    };
    sweep(std::false_type{});
    __syncthreads();
    // (atom, wave) runs in order: atom d starts at the sum of the atoms before it, wave w inside it behind the waves before it
    int tot = 0;
    if (tid < kBucketAtoms)
        for (int w = 0; w < kBucketWaves; w++) tot += cnt[w][tid];
    {
        const int incl = wave_prefix_sum(tid < kBucketAtoms ? tot : 0);
        __syncthreads();
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; w++) before += wave_tot[w];
        if (tid < kBucketAtoms) {
            int at = before + incl - tot;
            const int atom = bucket * kBucketAtoms + tid;
            if (atom < num_atoms) col_seg[atom] = make_int2(first + at, first + at + tot);
            for (int w = 0; w < kBucketWaves; w++) {
                const int c = cnt[w][tid];
                cnt[w][tid] = at;
                at += c;
            }
        }
    }
    __syncthreads();
    sweep(std::true_type{});
}

template <typename T> struct Term4 { T x, y, z, w; };

template <typename T>
__global__ __launch_bounds__(256) void pairs_backward_terms(long long num_slots, const int32_t* __restrict__ neighbors,
                                                            const T* __restrict__ deltas, const T* __restrict__ distances,
                                                            const T* __restrict__ grad_deltas, const T* __restrict__ grad_distances,
                                                            Term4<T>* __restrict__ terms) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= num_slots) return;
    Term4<T> g{T(0), T(0), T(0), T(0)};
    if (neighbors[k] >= 0) {                                   // CUDA.cu:93-94
        const T gd = grad_distances[k] / distances[k];
        g.x = grad_deltas[3 * k] + deltas[3 * k] * gd;         // CUDA.cu:96-99
        g.y = grad_deltas[3 * k + 1] + deltas[3 * k + 1] * gd;
        g.z = grad_deltas[3 * k + 2] + deltas[3 * k + 2] * gd;
    }
    terms[k] = g;
}

// 16 lanes per atom.  Lane l of the group takes entries l, l + 16, ... of the row segment, then of the column segment; the 16
// partial sums are added by a fixed xor tree: the order of the additions depends on the list only.
template <typename T>
__global__ __launch_bounds__(256) void pairs_backward_gather(int num_atoms, const int2* __restrict__ row_seg, const int2* __restrict__ col_seg,
                                                             const int* __restrict__ order, const Term4<T>* __restrict__ terms,
                                                             T* __restrict__ grad_positions) {
    const int i = (int)(((long long)blockIdx.x * 256 + threadIdx.x) >> 4), l = threadIdx.x & 15;
    const bool live = i < num_atoms;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    if (live) {
        const int2 rs = row_seg[i], cs = col_seg[i];
        for (int k = rs.x + l; k < rs.y; k += 16) {
            const Term4<T> g = terms[k];
            sx += (double)g.x; sy += (double)g.y; sz += (double)g.z;
        }
        for (int p = cs.x + l; p < cs.y; p += 16) {
            const Term4<T> g = terms[order[p]];
            sx -= (double)g.x; sy -= (double)g.y; sz -= (double)g.z;
        }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
        sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64); sz += __shfl_xor(sz, off, 64);
    }
    if (live && l == 0) {
        grad_positions[3 * (size_t)i] = (T)sx; grad_positions[3 * (size_t)i + 1] = (T)sy; grad_positions[3 * (size_t)i + 2] = (T)sz;
    }
}

}  // namespace

extern "C" {

int64_t nnpops_neighbor_pairs_index_ints(int num_atoms, int64_t num_slots) {
    if (num_atoms < 0 || num_slots < 0) return 0;
    return ((num_slots + 1) & ~1ll) + 4 * (int64_t)num_atoms;      // (the int2 segments sit 8-byte aligned behind the slots)
}

int64_t nnpops_neighbor_pairs_index_workspace_bytes(int num_atoms, int64_t num_slots) {
    if (num_atoms <= 0 || num_slots < 0) return 0;
    const size_t ntiles = (size_t)((num_slots + kTile - 1) / kTile) + 8;
    // the {column, slot} records by bucket | histogram [buckets][tiles] | bucket totals
    return (int64_t)(align256(sizeof(int2) * (size_t)num_slots) + align256(sizeof(int) * kMaxBuckets * ntiles) + align256(sizeof(int) * kMaxBuckets) + 512);
}

int nnpops_neighbor_pairs_build_index(int num_atoms, int64_t num_slots, const int32_t* neighbors, int32_t* index, void* workspace,
                                      void* stream) {
    NNPOPS_REQUIRE(num_atoms > 0 && num_slots >= 0, "bad sizes");
    NNPOPS_REQUIRE(num_slots < (1ll << 31) - kTile, "the transposed index holds 32-bit slots");
    if (num_atoms > kMaxBuckets * kMaxBucketAtoms)
        return fail(NNPOPS_ERR_UNSUPPORTED, "the transposed index is built for up to %d atoms (got %d): use nnpops_neighbor_pairs_backward_ws",
                    kMaxBuckets * kMaxBucketAtoms, num_atoms);
    int shift = 6;                                             // buckets of 64 ... 512 atoms: as many buckets (workgroups of the second level) as fit 512
    while (((num_atoms - 1) >> shift) + 1 > kMaxBuckets) shift++;
    NNPOPS_REQUIRE(index != nullptr && workspace != nullptr && (num_slots == 0 || neighbors != nullptr), "NULL device pointer");
    NNPOPS_REQUIRE(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)index & 7) == 0, "workspace must be 256-byte aligned, index 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int* order = index;
    int2* row_seg = (int2*)(index + ((num_slots + 1) & ~1ll));      // (int2: 8-byte aligned behind an odd number of slots)
    int2* col_seg = row_seg + num_atoms;
    const int ntiles = std::max(1, (int)((num_slots + kTile - 1) / kTile));
    char* ws = (char*)workspace;
    auto take = [&](size_t bytes) { char* p = ws; ws += align256(bytes); return p; };
    int2* records = (int2*)take(sizeof(int2) * (size_t)num_slots);
    int* hist = (int*)take(sizeof(int) * kMaxBuckets * ((size_t)ntiles + 8));
    int* totals = (int*)take(sizeof(int) * kMaxBuckets);
    const int* rows = neighbors;
    const int* cols = neighbors + num_slots;
    const int nbuckets = ((num_atoms - 1) >> shift) + 1;
    hipLaunchKernelGGL(pairs_index_histogram, dim3(ntiles), dim3(kTileThreads), 0, s, (long long)num_slots, num_atoms, shift, cols, ntiles, hist, row_seg);
    hipLaunchKernelGGL(pairs_index_scan, dim3(kMaxBuckets), dim3(256), 0, s, ntiles, hist, totals);
    hipLaunchKernelGGL(pairs_index_partition, dim3(ntiles), dim3(kTileThreads), 0, s, (long long)num_slots, num_atoms, shift, rows, cols, ntiles, (const int*)hist,
                       (const int*)totals, records, row_seg);
    hipLaunchKernelGGL(pairs_index_bucket, dim3(nbuckets), dim3(64 * kBucketWaves), 0, s, num_atoms, shift, (const int*)totals, (const int2*)records, order, col_seg);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

int64_t nnpops_neighbor_pairs_backward_indexed_workspace_bytes(int dtype, int64_t num_slots) {
    if (num_slots < 0) return 0;
    return (int64_t)align256((size_t)num_slots * (dtype == 1 ? 32 : 16)) + 256;
}

int nnpops_neighbor_pairs_backward_indexed(int dtype, int num_atoms, int64_t num_slots, const int32_t* neighbors, const void* deltas,
                                           const void* distances, const void* grad_deltas, const void* grad_distances,
                                           const int32_t* index, void* grad_positions, void* workspace, void* stream) {
    NNPOPS_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (float32) or 1 (float64)");
    NNPOPS_REQUIRE(num_atoms > 0 && num_slots >= 0 && num_slots < (1ll << 31), "bad sizes");
    NNPOPS_REQUIRE(grad_positions != nullptr && workspace != nullptr && index != nullptr, "NULL device pointer");
    NNPOPS_REQUIRE(((uintptr_t)workspace & 31) == 0, "the workspace must be 32-byte aligned");
    NNPOPS_REQUIRE(num_slots == 0 || (neighbors && deltas && distances && grad_deltas && grad_distances), "NULL device pointer");
    hipStream_t s = (hipStream_t)stream;
    const int* order = index;
    const int2* row_seg = (const int2*)(index + ((num_slots + 1) & ~1ll));
    const int2* col_seg = row_seg + num_atoms;
    const dim3 pgrid((unsigned)((num_slots + 255) / 256)), agrid((unsigned)(((long long)num_atoms * 16 + 255) / 256));
    if (dtype == 0) {
        Term4<float>* terms = (Term4<float>*)workspace;
        if (num_slots > 0)
            hipLaunchKernelGGL(pairs_backward_terms<float>, pgrid, dim3(256), 0, s, (long long)num_slots, neighbors, (const float*)deltas,
                               (const float*)distances, (const float*)grad_deltas, (const float*)grad_distances, terms);
        hipLaunchKernelGGL(pairs_backward_gather<float>, agrid, dim3(256), 0, s, num_atoms, row_seg, col_seg, order, terms, (float*)grad_positions);
    } else {
        Term4<double>* terms = (Term4<double>*)workspace;
        if (num_slots > 0)
            hipLaunchKernelGGL(pairs_backward_terms<double>, pgrid, dim3(256), 0, s, (long long)num_slots, neighbors, (const double*)deltas,
                               (const double*)distances, (const double*)grad_deltas, (const double*)grad_distances, terms);
        hipLaunchKernelGGL(pairs_backward_gather<double>, agrid, dim3(256), 0, s, num_atoms, row_seg, col_seg, order, terms, (double*)grad_positions);
    }
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

}  // extern "C"
