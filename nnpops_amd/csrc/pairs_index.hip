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
// The sort by column is a stable LSD radix sort written for this job (round 6: rocprim::radix_sort_pairs was measured first -- 141 us
// for 3 M slots at 100 000 atoms, three look-back passes of 30 us each whose chain of 730 tiles is all latency at this size, plus
// seven fill launches): 9-bit digits (two passes up to 262 144 atoms), per pass a histogram per tile of 2 048 slots in LDS, one
// scan launch (a workgroup per digit over the tiles), and a scatter whose ranks are STABLE -- a wave finds the lanes holding its
// digit with nine ballots, the four waves of a tile are ordered through a 4 x 512 table in LDS -- so the slots of one atom come out
// in ascending order whatever the hardware does.  Unused slots (-1) are dropped, not sorted: a list the forward op emitted keeps
// them behind the pairs, and a tile that starts with one is skipped.
#include "device_common.h"
#include "host_common.h"

using namespace nnpops;

namespace {

constexpr int kDigitBits = 9, kDigits = 1 << kDigitBits;      // 512 bins
constexpr int kTile = 2048, kTileThreads = 256, kRounds = kTile / kTileThreads;

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

int id_bits(int num_atoms) {
    int b = 1;
    while ((1ll << b) < (long long)num_atoms) b++;
    return b;
}

// Pass p sorts by digit p of the key.  FIRST: keys are neighbors[1] and the value of slot k is k itself; later passes read the
// (key, value) pairs the pass before left (n_valid of them: *count).
template <bool FIRST>
__global__ __launch_bounds__(kTileThreads) void pairs_sort_histogram(long long num_slots, int num_atoms, int shift, const int* __restrict__ keys,
                                                                    const int* __restrict__ count, int ntiles, int* __restrict__ hist,
                                                                    int2* __restrict__ row_seg, int2* __restrict__ col_seg) {
    __shared__ int bins[kDigits];
    const int tid = threadIdx.x, tile = blockIdx.x;
    if (FIRST) {                                               // (the segments of atoms without pairs: empty)
        for (long long i = (long long)tile * kTileThreads + tid; i < num_atoms; i += (long long)gridDim.x * kTileThreads) {
            row_seg[i] = make_int2(0, 0);
            col_seg[i] = make_int2(0, 0);
        }
    }
    bins[tid] = 0; bins[tid + kTileThreads] = 0;
    __syncthreads();
    const long long n = FIRST ? num_slots : (long long)min((long long)*count, num_slots);
    const long long base = (long long)tile * kTile;
    if (base < n && !(FIRST && keys[base] < 0)) {              // (an unused slot at the head of a tile: nothing but unused slots behind it)
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            const long long k = base + r * kTileThreads + tid;
            if (k < n) {
                const int key = keys[k];
                if (key >= 0) atomicAdd(&bins[(key >> shift) & (kDigits - 1)], 1);      // (LDS, integer: the counts do not depend on the order)
            }
        }
    }
    __syncthreads();
    hist[(size_t)tid * ntiles + tile] = bins[tid];
    hist[(size_t)(tid + kTileThreads) * ntiles + tile] = bins[tid + kTileThreads];
}

// One workgroup per digit: exclusive scan of its counts over the tiles (in place) and the digit's total.
__global__ __launch_bounds__(256) void pairs_sort_scan(int ntiles, int* __restrict__ hist, int* __restrict__ totals) {
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

template <bool FIRST>
__global__ __launch_bounds__(kTileThreads) void pairs_sort_scatter(long long num_slots, int shift, const int* __restrict__ keys,
                                                                  const int* __restrict__ vals, const int* __restrict__ count, int ntiles,
                                                                  const int* __restrict__ hist, const int* __restrict__ totals,
                                                                  int* __restrict__ keys_out, int* __restrict__ vals_out,
                                                                  int* __restrict__ count_out) {
    __shared__ int digit_base[kDigits];
    __shared__ int wave_cnt[4][kDigits];
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = blockIdx.x;
    {   // exclusive scan of the 512 digit totals (two per thread)
        const int a = totals[2 * tid], b = totals[2 * tid + 1];
        const int incl = wave_prefix_sum(a + b);
        if (lane == 63) wave_tot[wave] = incl;
#pragma unroll
        for (int w = 0; w < 4; w++) { wave_cnt[w][tid] = 0; wave_cnt[w][tid + kTileThreads] = 0; }
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; w++) before += wave_tot[w];
        digit_base[2 * tid] = before + incl - a - b;
        digit_base[2 * tid + 1] = before + incl - b;
        if (tile == 0 && tid == 255) *count_out = before + incl;      // number of used slots: what the next pass / the bounds read
    }
    __syncthreads();
    const long long n = FIRST ? num_slots : (long long)min((long long)*count, num_slots);
    const long long base = (long long)tile * kTile;
    if (base >= n || (FIRST && keys[base] < 0)) return;        // (uniform)
    // wave w takes slots [base + 512 w, base + 512 (w + 1)) in eight rounds of 64: (wave, round, lane) is slot order
    int key[kRounds], val[kRounds], rank[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const long long k = base + wave * (kTile / 4) + r * 64 + lane;
        key[r] = k < n ? keys[k] : -1;
        val[r] = FIRST ? (int)k : (k < n ? vals[k] : 0);
    }
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const bool used = key[r] >= 0;
        const int d = (key[r] >> shift) & (kDigits - 1);
        unsigned long long peers = __ballot(used);
#pragma unroll
        for (int b = 0; b < kDigitBits; b++) {
            const unsigned long long m = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? m : ~m;
        }
        const int old = used ? wave_cnt[wave][d] : 0;          // (LDS operations of a wave execute in order: everybody reads, then one lane writes)
        rank[r] = old + prefix_popc(peers);
        if (used && prefix_popc(peers) == 0) wave_cnt[wave][d] = old + __popcll(peers);
        wave_fence();                                          // (the next round reads what this one wrote)
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        if (key[r] < 0) continue;
        const int d = (key[r] >> shift) & (kDigits - 1);
        int at = digit_base[d] + hist[(size_t)d * ntiles + tile] + rank[r];
        for (int w = 0; w < wave; w++) at += wave_cnt[w][d];
        keys_out[at] = key[r];
        vals_out[at] = val[r];
    }
}

// rows: neighbors[0][0 .. num_slots); cols: the sorted keys [0 .. *count).  Equal ids are contiguous in both.
__global__ __launch_bounds__(256) void pairs_index_bounds(long long num_slots, int num_atoms, const int* __restrict__ rows,
                                                         const int* __restrict__ cols, const int* __restrict__ count,
                                                         int2* __restrict__ row_seg, int2* __restrict__ col_seg) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= num_slots) return;
    {
        const int a = rows[k];
        if ((unsigned)a < (unsigned)num_atoms) {
            const int before = k > 0 ? rows[k - 1] : -2, after = k + 1 < num_slots ? rows[k + 1] : -2;
            if (a != before) row_seg[a].x = (int)k;
            if (a != after) row_seg[a].y = (int)(k + 1);
        }
    }
    const long long n = min((long long)*count, num_slots);
    if (k < n) {
        const int a = cols[k];
        if ((unsigned)a < (unsigned)num_atoms) {
            const int before = k > 0 ? cols[k - 1] : -2, after = k + 1 < n ? cols[k + 1] : -2;
            if (a != before) col_seg[a].x = (int)k;
            if (a != after) col_seg[a].y = (int)(k + 1);
        }
    }
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
    const size_t ntiles = (size_t)((num_slots + kTile - 1) / kTile) + 1;
    // three (key, value) buffers at most in flight (ping, pong, sorted keys) | histogram [512][tiles] | digit totals | two counts
    return (int64_t)(4 * align256(sizeof(int) * (size_t)num_slots) + align256(sizeof(int) * kDigits * ntiles) + align256(sizeof(int) * kDigits) + 512);
}

int nnpops_neighbor_pairs_build_index(int num_atoms, int64_t num_slots, const int32_t* neighbors, int32_t* index, void* workspace,
                                      void* stream) {
    NNPOPS_REQUIRE(num_atoms > 0 && num_slots >= 0, "bad sizes");
    NNPOPS_REQUIRE(num_slots < (1ll << 31) - kTile, "the transposed index holds 32-bit slots");
    NNPOPS_REQUIRE(index != nullptr && workspace != nullptr && (num_slots == 0 || neighbors != nullptr), "NULL device pointer");
    NNPOPS_REQUIRE(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)index & 7) == 0, "workspace must be 256-byte aligned, index 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int* order = index;
    int2* row_seg = (int2*)(index + ((num_slots + 1) & ~1ll));      // (int2: 8-byte aligned behind an odd number of slots)
    int2* col_seg = row_seg + num_atoms;
    const int ntiles = (int)((num_slots + kTile - 1) / kTile) + (num_slots == 0 ? 1 : 0);
    char* ws = (char*)workspace;
    auto take = [&](size_t bytes) { char* p = ws; ws += align256(bytes); return p; };
    int* kbuf[2] = {(int*)take(sizeof(int) * (size_t)num_slots), (int*)take(sizeof(int) * (size_t)num_slots)};
    int* vbuf[2] = {(int*)take(sizeof(int) * (size_t)num_slots), (int*)take(sizeof(int) * (size_t)num_slots)};
    int* hist = (int*)take(sizeof(int) * kDigits * ((size_t)ntiles + 1));
    int* totals = (int*)take(sizeof(int) * kDigits);
    int* counts = (int*)take(256);                             // counts[p & 1]: used slots, as pass p found them
    const int bits = id_bits(num_atoms);
    const int passes = std::max(1, (bits + kDigitBits - 1) / kDigitBits);
    const int* cols = neighbors + num_slots;
    const int* sorted_keys = nullptr;
    for (int p = 0; p < passes; p++) {
        const bool first = p == 0, last = p == passes - 1;
        const int* kin = first ? cols : kbuf[(p - 1) & 1];
        const int* vin = first ? nullptr : (vbuf[(p - 1) & 1]);
        int* kout = kbuf[p & 1];
        int* vout = last ? order : vbuf[p & 1];
        const int* cin = first ? nullptr : counts + ((p - 1) & 1);
        if (first) {
            hipLaunchKernelGGL(pairs_sort_histogram<true>, dim3(ntiles), dim3(kTileThreads), 0, s, (long long)num_slots, num_atoms, 0, kin, cin, ntiles, hist,
                               row_seg, col_seg);
            hipLaunchKernelGGL(pairs_sort_scan, dim3(kDigits), dim3(256), 0, s, ntiles, hist, totals);
            hipLaunchKernelGGL(pairs_sort_scatter<true>, dim3(ntiles), dim3(kTileThreads), 0, s, (long long)num_slots, 0, kin, vin, cin, ntiles,
                               (const int*)hist, (const int*)totals, kout, vout, counts + (p & 1));
        } else {
            hipLaunchKernelGGL(pairs_sort_histogram<false>, dim3(ntiles), dim3(kTileThreads), 0, s, (long long)num_slots, num_atoms, p * kDigitBits, kin, cin,
                               ntiles, hist, row_seg, col_seg);
            hipLaunchKernelGGL(pairs_sort_scan, dim3(kDigits), dim3(256), 0, s, ntiles, hist, totals);
            hipLaunchKernelGGL(pairs_sort_scatter<false>, dim3(ntiles), dim3(kTileThreads), 0, s, (long long)num_slots, p * kDigitBits, kin, vin, cin,
                               ntiles, (const int*)hist, (const int*)totals, kout, vout, counts + (p & 1));
        }
        sorted_keys = kout;
    }
    if (num_slots > 0)
        hipLaunchKernelGGL(pairs_index_bounds, dim3((unsigned)((num_slots + 255) / 256)), dim3(256), 0, s, (long long)num_slots, num_atoms,
                           (const int*)neighbors, sorted_keys, (const int*)(counts + ((passes - 1) & 1)), row_seg, col_seg);
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
