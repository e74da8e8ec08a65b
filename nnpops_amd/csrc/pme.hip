// pme.hip -- direct-space part of Particle Mesh Ewald on a neighbour-pair list.
//
// Replaces the reference's computeDirect (reference src/pytorch/pme/pmeCUDA.cu:30-100, CPU form
// src/pytorch/pme/pmeCPU.cpp:75-163), the immediate consumer of getNeighborPairs' output
// (src/pytorch/pme/pme.py:163-165).  For every listed pair that is not excluded
//     E += k q1 q2 erfc(alpha r) / r
//     dE/dr / r = -k q1 q2 (erfc(alpha r) + 2 alpha r exp(-(alpha r)^2) / sqrt(pi)) / r^3      (force along delta)
// and for every excluded pair (taken WITHOUT periodic wrap, each once) the erf() part is subtracted again, because the
// reciprocal-space sum cannot leave it out.  Outputs: the energy and dE/dpositions, dE/dcharges (the autograd backward
// only scales them).  The reciprocal-space part (charge spreading + 3-D FFT) is outside this build's scope.
//
// Layout for MI355X: OWNER COMPUTES, no float atomics (the first version issued ~12 M device-scope float
// atomics at 100 000 atoms and was bound by them: 0.6 ms).  Two launches:
//   pme_direct_pairs   one lane per pair slot, streaming the four list arrays (28 B per slot, coalesced).  What a pair
//                      contributes to its SECOND atom is not added anywhere: the lane claims a slot in that atom's
//                      "incoming" row with ONE returning integer atomic and stores {fx, fy, fz, dE/dq} there (16 B).  The
//                      list getNeighborPairs produces is grouped by neighbors[0], so the 64 consecutive pairs of a wave
//                      touch two or three distinct first atoms: their contributions are added with a segmented scan over
//                      runs of consecutive equal atoms (runs found from neighbour comparisons only -- any pair order stays
//                      correct, just with shorter runs) and every run is ONE more incoming entry.
//   pme_direct_gather  16 lanes per atom: the atom adds up its incoming entries, adds the terms of its excluded pairs
//                      (each atom walks its own exclusion row: the table is symmetric, pme.py:66-73) and stores its
//                      derivatives once.  The entries arrive in an order that depends on the atomics, so the sum is made
//                      ORDER-INDEPENDENT: every entry is rounded to a multiple of 2^-40 of the atom's largest entry (far
//                      below fp32 resolution) and the multiples are added exactly in double precision.
// Energy: double per lane, per workgroup, partials summed in a fixed order.  Everything is bitwise reproducible -- the
// reference scatters eight float atomics per pair (pmeCUDA.cu:62-69) -- unless an atom receives more entries than its
// row holds (4 x the average pairs per atom + 32); such entries fall back to float atomics on a side array.
#include <cmath>

#include "device_common.h"
#include "host_common.h"

using namespace nnpops;

namespace {

constexpr int kPmeBlock = 256;
constexpr float kTwoOverSqrtPi = 1.12837916709551257390f;

// Inclusive segmented sum over runs of CONSECUTIVE lanes with the same key (lanes of a run end up with the sum from the
// run's first lane to themselves).  Runs are found from the neighbour comparison only, so any key order is handled.
__device__ __forceinline__ void segmented_scan4(int key, float& a, float& b, float& c, float& d) {
    const int lane = lane_id();
    const int prev = __shfl_up(key, 1, 64);
    const unsigned long long heads = __ballot(lane == 0 || prev != key);
    const int start = 63 - __builtin_clzll(heads & (~0ull >> (63 - lane)));      // first lane of my run
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float ua = __shfl_up(a, off, 64), ub = __shfl_up(b, off, 64), uc = __shfl_up(c, off, 64), ud = __shfl_up(d, off, 64);
        if (lane - off >= start) { a += ua; b += ub; c += uc; d += ud; }
    }
}

__global__ __launch_bounds__(kPmeBlock) void pme_direct_pairs(long long num_pairs, int max_excl, const int* __restrict__ nb0,
                                                             const int* __restrict__ nb1, const float* __restrict__ deltas,
                                                             const float* __restrict__ distances, const float* __restrict__ charge,
                                                             const int* __restrict__ excl, float alpha, float coulomb,
                                                             int* __restrict__ count, float4* __restrict__ incoming, int cap,
                                                             float* __restrict__ spill, double* __restrict__ partial) {
    __shared__ double red[kPmeBlock / 64];
    double energy = 0.0;
    const long long stride = (long long)gridDim.x * kPmeBlock;
    const long long first = (long long)blockIdx.x * kPmeBlock + threadIdx.x;
    auto deliver = [&](int atom, float x, float y, float z, float q) {
        const int slot = atomicAdd(&count[atom], 1);
        if (slot < cap) {
            incoming[(size_t)atom * cap + slot] = make_float4(x, y, z, q);
        } else {                                             // row full: side array, float atomics (not reproducible; never in practice)
            atomicAdd(&spill[4 * (size_t)atom], x); atomicAdd(&spill[4 * (size_t)atom + 1], y);
            atomicAdd(&spill[4 * (size_t)atom + 2], z); atomicAdd(&spill[4 * (size_t)atom + 3], q);
        }
    };
    // (whole waves iterate together: the segmented scan needs every lane of the wave in the loop)
    for (long long base = first - lane_id(); base < num_pairs; base += stride) {
        const long long i = base + lane_id();
        int atom1 = -1, atom2 = -1;
        bool include = false;
        float fx = 0.f, fy = 0.f, fz = 0.f, cd1 = 0.f, cd2 = 0.f;
        if (i < num_pairs) {
            atom1 = nb0[i];
            atom2 = nb1[i];
            include = atom1 > -1;
            // exclusion rows are sorted in descending order (pme.py:93): stop at the first entry below atom2  (ref :43-46)
            for (int j = 0; include && j < max_excl; j++) {
                const int e = excl[(long long)atom1 * max_excl + j];
                if (e < atom2) break;
                if (e == atom2) include = false;
            }
            if (include) {
                const float r = distances[i];
                const float inv_r = 1.0f / r, ar = alpha * r;
                const float ex = expf(-ar * ar), erfc_ar = erfcf(ar);
                const float pre = coulomb * inv_r;
                const float c1 = charge[atom1], c2 = charge[atom2];
                energy += (double)(pre * erfc_ar * c1 * c2);
                cd1 = pre * erfc_ar * c2;
                cd2 = pre * erfc_ar * c1;
                const float dedr = pre * c1 * c2 * (erfc_ar + ar * ex * kTwoOverSqrtPi) * inv_r * inv_r;
                fx = dedr * deltas[3 * i]; fy = dedr * deltas[3 * i + 1]; fz = dedr * deltas[3 * i + 2];
                deliver(atom2, fx, fy, fz, cd2);
            }
        }
        // first atom of the pair: one entry per run of equal atoms inside the wave
        const int key = include ? atom1 : -1 - lane_id();          // (excluded / empty slots never join a run)
        float sx = -fx, sy = -fy, sz = -fz, sc = cd1;
        segmented_scan4(key, sx, sy, sz, sc);
        const int next_key = __shfl_down(key, 1, 64);
        if (include && (lane_id() == 63 || next_key != key)) deliver(atom1, sx, sy, sz, sc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) energy += __shfl_xor(energy, off, 64);
    if (lane_id() == 0) red[threadIdx.x >> 6] = energy;
    __syncthreads();
    if (threadIdx.x == 0) {
        double e = 0.0;
        for (int w = 0; w < kPmeBlock / 64; w++) e += red[w];
        partial[blockIdx.x] = e;
    }
}

// Exact, order-independent sums: `scale` is a power of two, so v * scale, the rounding to an integer and the division back
// are exact or correctly rounded, and sums of integers below 2^53 are exact in double whatever their order.
__device__ __forceinline__ double quantise(float v, double scale) { return rint((double)v * scale); }

__device__ __forceinline__ double group16_sum(double v) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// 2^(40 - e) with 2^e > m >= 2^(e-1): every |entry| * scale < 2^40, so sums of thousands of entries stay exact in double
__device__ __forceinline__ double scale_for(float m) {
    int e = 0;
    (void)frexpf(m, &e);
    return m > 0.f ? ldexp(1.0, 40 - e) : 1.0;
}

// 16 lanes per atom: incoming entries + excluded pairs (each atom its own side; energy once, by the lower index) -> outputs
__global__ __launch_bounds__(kPmeBlock) void pme_direct_gather(int num_atoms, int max_excl, const float* __restrict__ pos,
                                                              const float* __restrict__ charge, const int* __restrict__ excl,
                                                              float alpha, float coulomb, const int* __restrict__ count,
                                                              const float4* __restrict__ incoming, int cap, const float* __restrict__ spill,
                                                              float* __restrict__ pos_deriv, float* __restrict__ charge_deriv,
                                                              double* __restrict__ partial) {
    __shared__ double red[kPmeBlock / 64];
    double energy = 0.0;
    const int sub = threadIdx.x & 15;
    const int groups = gridDim.x * (kPmeBlock / 16);
    // (whole waves iterate together: the group shuffles need every lane of the wave in the loop)
    for (int base = blockIdx.x * (kPmeBlock / 16); base < num_atoms; base += groups) {
        const int atom = base + (threadIdx.x >> 4);
        const bool live = atom < num_atoms;
        const int arrived = live ? count[atom] : 0;
        const int n = min(arrived, cap);
        const float4* row = incoming + (size_t)(live ? atom : 0) * cap;
        // pass 1: the largest magnitude per component fixes the quantum
        float mx = 0.f, my = 0.f, mz = 0.f, mq = 0.f;
        for (int k = sub; k < n; k += 16) {
            const float4 v = row[k];
            mx = fmaxf(mx, fabsf(v.x)); my = fmaxf(my, fabsf(v.y)); mz = fmaxf(mz, fabsf(v.z)); mq = fmaxf(mq, fabsf(v.w));
        }
        // excluded pairs of this atom: the erf() part reciprocal space cannot leave out, un-wrapped  (ref :71-99)
        float ex_x = 0.f, ex_y = 0.f, ex_z = 0.f, ex_q = 0.f;
        if (live) {
            const float px = pos[3 * atom], py = pos[3 * atom + 1], pz = pos[3 * atom + 2], c1 = charge[atom];
            for (int j = sub; j < max_excl; j += 16) {
                const int other = excl[(long long)atom * max_excl + j];
                if (other < 0 || other == atom) continue;
                const float dx = px - pos[3 * other], dy = py - pos[3 * other + 1], dz = pz - pos[3 * other + 2];
                const float r = sqrtf(dx * dx + dy * dy + dz * dz);
                const float inv_r = 1.0f / r, ar = alpha * r;
                const float e = expf(-ar * ar), erf_ar = erff(ar);
                const float pre = coulomb * inv_r;
                const float c2 = charge[other];
                if (other > atom) energy -= (double)(pre * erf_ar * c1 * c2);      // once per pair
                const float dedr = pre * c1 * c2 * (erf_ar - ar * e * kTwoOverSqrtPi) * inv_r * inv_r;
                ex_x += dedr * dx; ex_y += dedr * dy; ex_z += dedr * dz;
                ex_q -= pre * erf_ar * c2;
            }
        }
        mx = group16_max(fmaxf(mx, fabsf(ex_x))); my = group16_max(fmaxf(my, fabsf(ex_y)));
        mz = group16_max(fmaxf(mz, fabsf(ex_z))); mq = group16_max(fmaxf(mq, fabsf(ex_q)));
        const double kx = scale_for(mx), ky = scale_for(my), kz = scale_for(mz), kq = scale_for(mq);
        // pass 2: exact sum of the quantised entries
        double sx = quantise(ex_x, kx), sy = quantise(ex_y, ky), sz = quantise(ex_z, kz), sq = quantise(ex_q, kq);
        for (int k = sub; k < n; k += 16) {
            const float4 v = row[k];
            sx += quantise(v.x, kx); sy += quantise(v.y, ky); sz += quantise(v.z, kz); sq += quantise(v.w, kq);
        }
        sx = group16_sum(sx); sy = group16_sum(sy); sz = group16_sum(sz); sq = group16_sum(sq);
        if (live && sub == 0) {
            const float4 extra = arrived > cap ? *reinterpret_cast<const float4*>(spill + 4 * (size_t)atom) : make_float4(0.f, 0.f, 0.f, 0.f);
            pos_deriv[3 * atom] = (float)(sx / kx) + extra.x;
            pos_deriv[3 * atom + 1] = (float)(sy / ky) + extra.y;
            pos_deriv[3 * atom + 2] = (float)(sz / kz) + extra.z;
            charge_deriv[atom] = (float)(sq / kq) + extra.w;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) energy += __shfl_xor(energy, off, 64);
    if (lane_id() == 0) red[threadIdx.x >> 6] = energy;
    __syncthreads();
    if (threadIdx.x == 0) {
        double e = 0.0;
        for (int w = 0; w < kPmeBlock / 64; w++) e += red[w];
        partial[blockIdx.x] = e;
    }
}

// workgroup partials -> energy, always in the same order
__global__ __launch_bounds__(1024) void pme_sum_partials(const double* __restrict__ partial, int count, float* __restrict__ energy) {
    __shared__ double red[1024];
    double e = 0.0;
    for (int i = threadIdx.x; i < count; i += 1024) e += partial[i];      // (thousands of partials: one round trip per ten of them with 1 024 lanes)
    red[threadIdx.x] = e;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *energy = (float)red[0];
}

// ---- round 6: the same sums over the pair list's TRANSPOSED INDEX (pairs_index.hip) -- no atomics, no incoming rows -----------------
// For a list the forward op of getNeighborPairs emitted (grouped by neighbors[0]) and its index: pme_direct_terms is one streaming
// pass (per slot: the force on the second atom with dE/dq of the first, and the same force with dE/dq of the second, as two 16-byte records; the energy
// in double per workgroup), pme_direct_gather_indexed 16 lanes per atom -- its own pairs contiguous, the pairs it is second in through
// the index, its excluded pairs as before -- added up in double in a fixed order: bitwise reproducible without the quantisation the
// arrival order of pme_direct_pairs' entries makes necessary, and without its one returning atomic per pair (172 us at 100 000 atoms).
__global__ __launch_bounds__(kPmeBlock) void pme_direct_terms(long long num_pairs, int max_excl, const int* __restrict__ nb0,
                                                             const int* __restrict__ nb1, const float* __restrict__ deltas,
                                                             const float* __restrict__ distances, const float* __restrict__ charge,
                                                             const int* __restrict__ excl, float alpha, float coulomb,
                                                             float4* __restrict__ terms, float4* __restrict__ second, double* __restrict__ partial) {
    __shared__ double red[kPmeBlock / 64];
    double energy = 0.0;
    const long long stride = (long long)gridDim.x * kPmeBlock;
    for (long long i = (long long)blockIdx.x * kPmeBlock + threadIdx.x; i < num_pairs; i += stride) {
        const int atom1 = nb0[i], atom2 = nb1[i];
        bool include = atom1 > -1;
        for (int j = 0; include && j < max_excl; j++) {       // exclusion rows are sorted in descending order (pme.py:93; ref :43-46)
            const int e = excl[(long long)atom1 * max_excl + j];
            if (e < atom2) break;
            if (e == atom2) include = false;
        }
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        float cd2 = 0.f;
        if (include) {
            const float r = distances[i];
            const float inv_r = 1.0f / r, ar = alpha * r;
            const float ex = expf(-ar * ar), erfc_ar = erfcf(ar);
            const float pre = coulomb * inv_r;
            const float c1 = charge[atom1], c2 = charge[atom2];
            energy += (double)(pre * erfc_ar * c1 * c2);
            const float dedr = pre * c1 * c2 * (erfc_ar + ar * ex * kTwoOverSqrtPi) * inv_r * inv_r;
            t = make_float4(dedr * deltas[3 * i], dedr * deltas[3 * i + 1], dedr * deltas[3 * i + 2], pre * erfc_ar * c2);
            cd2 = pre * erfc_ar * c1;
        }
        terms[i] = t;                                          // what the FIRST atom of the pair reads (contiguously): force, its dE/dq
        second[i] = make_float4(t.x, t.y, t.z, cd2);           // what the SECOND atom gathers: ONE 16-byte record per pair
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) energy += __shfl_xor(energy, off, 64);
    if (lane_id() == 0) red[threadIdx.x >> 6] = energy;
    __syncthreads();
    if (threadIdx.x == 0) {
        double e = 0.0;
        for (int w = 0; w < kPmeBlock / 64; w++) e += red[w];
        partial[blockIdx.x] = e;
    }
}

__global__ __launch_bounds__(kPmeBlock) void pme_direct_gather_indexed(int num_atoms, int max_excl, const float* __restrict__ pos,
                                                                      const float* __restrict__ charge, const int* __restrict__ excl,
                                                                      float alpha, float coulomb, const int2* __restrict__ row_seg,
                                                                      const int2* __restrict__ col_seg, const int* __restrict__ order,
                                                                      const float4* __restrict__ terms, const float4* __restrict__ second,
                                                                      float* __restrict__ pos_deriv, float* __restrict__ charge_deriv,
                                                                      double* __restrict__ partial) {
    __shared__ double red[kPmeBlock / 64];
    double energy = 0.0;
    const int sub = threadIdx.x & 15;
    const int groups = gridDim.x * (kPmeBlock / 16);
    for (int base = blockIdx.x * (kPmeBlock / 16); base < num_atoms; base += groups) {       // (whole waves iterate together: group shuffles)
        const int atom = base + (threadIdx.x >> 4);
        const bool live = atom < num_atoms;
        double sx = 0.0, sy = 0.0, sz = 0.0, sq = 0.0;
        if (live) {
            const int2 rs = row_seg[atom], cs = col_seg[atom];
            for (int k = rs.x + sub; k < rs.y; k += 16) {     // pairs this atom is the FIRST of: minus the force, its own dE/dq
                const float4 t = terms[k];
                sx -= (double)t.x; sy -= (double)t.y; sz -= (double)t.z; sq += (double)t.w;
            }
            for (int p = cs.x + sub; p < cs.y; p += 16) {     // pairs it is the SECOND of
                const float4 t = second[order[p]];
                sx += (double)t.x; sy += (double)t.y; sz += (double)t.z; sq += (double)t.w;
            }
            // excluded pairs of this atom: the erf() part reciprocal space cannot leave out, un-wrapped  (ref :71-99)
            const float px = pos[3 * atom], py = pos[3 * atom + 1], pz = pos[3 * atom + 2], c1 = charge[atom];
            for (int j = sub; j < max_excl; j += 16) {
                const int other = excl[(long long)atom * max_excl + j];
                if (other < 0 || other == atom) continue;
                const float dx = px - pos[3 * other], dy = py - pos[3 * other + 1], dz = pz - pos[3 * other + 2];
                const float r = sqrtf(dx * dx + dy * dy + dz * dz);
                const float inv_r = 1.0f / r, ar = alpha * r;
                const float e = expf(-ar * ar), erf_ar = erff(ar);
                const float pre = coulomb * inv_r;
                const float c2 = charge[other];
                if (other > atom) energy -= (double)(pre * erf_ar * c1 * c2);      // once per pair
                const float dedr = pre * c1 * c2 * (erf_ar - ar * e * kTwoOverSqrtPi) * inv_r * inv_r;
                sx += (double)(dedr * dx); sy += (double)(dedr * dy); sz += (double)(dedr * dz);
                sq -= (double)(pre * erf_ar * c2);
            }
        }
        sx = group16_sum(sx); sy = group16_sum(sy); sz = group16_sum(sz); sq = group16_sum(sq);
        if (live && sub == 0) {
            pos_deriv[3 * atom] = (float)sx; pos_deriv[3 * atom + 1] = (float)sy; pos_deriv[3 * atom + 2] = (float)sz;
            charge_deriv[atom] = (float)sq;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) energy += __shfl_xor(energy, off, 64);
    if (lane_id() == 0) red[threadIdx.x >> 6] = energy;
    __syncthreads();
    if (threadIdx.x == 0) {
        double e = 0.0;
        for (int w = 0; w < kPmeBlock / 64; w++) e += red[w];
        partial[blockIdx.x] = e;
    }
}

int pair_blocks(long long num_pairs) { return (int)std::min<long long>(std::max<long long>(1, (num_pairs + kPmeBlock - 1) / kPmeBlock), 256 * 16); }
int gather_blocks_full(int num_atoms) { return (int)std::max<long long>(1, ((long long)num_atoms * 16 + kPmeBlock - 1) / kPmeBlock); }
int gather_blocks(int num_atoms) { return (int)std::min<long long>(std::max<long long>(1, ((long long)num_atoms * 16 + kPmeBlock - 1) / kPmeBlock), 256 * 16); }

// entries an atom's incoming row holds: a half list gives the atom with the lowest index ALL its neighbours (twice the
// average), dense spots more; plus one entry per run of the atom's own pairs
// (num_pairs is the CAPACITY of the caller's list, padding included: a generously padded list, or max_num_pairs = -1 with its
//  N (N - 1) / 2 slots, must not size the rows -- 2 048 entries are more than an atom has partners inside any cutoff PME is run with
//  (12 A at liquid density: 720), and an atom that does receive more falls back to the spill array: 32 KB per atom at most)
int incoming_capacity(long long num_pairs, int num_atoms) {
    const long long avg = (num_pairs + num_atoms - 1) / std::max(num_atoms, 1);
    return (int)std::min<long long>(4 * avg + 32, 2048);
}

struct PmeWorkspace {
    double* partial; int* count; float* spill; float4* incoming; int cap; size_t bytes;
};
PmeWorkspace carve(void* workspace, long long num_pairs, int num_atoms) {
    PmeWorkspace w{};
    uintptr_t p = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    auto take = [&](size_t bytes) { const uintptr_t at = p; p += (bytes + 255) & ~(size_t)255; return at; };
    w.cap = incoming_capacity(num_pairs, num_atoms);
    w.partial = (double*)take(sizeof(double) * (pair_blocks(num_pairs) + gather_blocks(num_atoms)));
    w.count = (int*)take(sizeof(int) * (size_t)num_atoms);               // count | spill are zeroed together
    w.spill = (float*)take(sizeof(float) * 4 * (size_t)num_atoms);
    w.incoming = (float4*)take(sizeof(float4) * (size_t)num_atoms * w.cap);
    w.bytes = (size_t)(p - (uintptr_t)workspace) + 256;
    return w;
}

}  // namespace

extern "C" {

int64_t nnpops_pme_direct_workspace_bytes(int64_t num_pairs, int num_atoms, int max_exclusions) {
    if (num_pairs < 0 || num_atoms < 0 || max_exclusions < 0) return 0;
    return (int64_t)carve(nullptr, num_pairs, num_atoms).bytes;
}

int nnpops_pme_direct(int num_atoms, int64_t num_pairs, int max_exclusions, const float* positions, const float* charges,
                      const int32_t* neighbors, const float* deltas, const float* distances, const int32_t* exclusions,
                      float alpha, float coulomb, float* energy, float* position_deriv, float* charge_deriv, void* workspace,
                      void* stream) {
    NNPOPS_REQUIRE(num_atoms > 0 && num_pairs >= 0 && max_exclusions >= 0, "bad sizes (atoms %d, pairs %lld, exclusions %d)", num_atoms,
                   (long long)num_pairs, max_exclusions);
    NNPOPS_REQUIRE(alpha > 0 && coulomb > 0, "alpha and coulomb must be positive");
    NNPOPS_REQUIRE(positions && charges && energy && position_deriv && charge_deriv && workspace, "NULL device pointer");
    NNPOPS_REQUIRE(num_pairs == 0 || (neighbors && deltas && distances), "NULL pair-list pointer");
    NNPOPS_REQUIRE(max_exclusions == 0 || exclusions, "NULL exclusions pointer");
    hipStream_t s = (hipStream_t)stream;
    const PmeWorkspace w = carve(workspace, num_pairs, num_atoms);
    const int pb = pair_blocks(num_pairs), gb = gather_blocks(num_atoms);
    NNPOPS_HIP_TRY(hipMemsetAsync(w.count, 0, (size_t)((char*)w.incoming - (char*)w.count), s));       // counters and the spill array
    hipLaunchKernelGGL(pme_direct_pairs, dim3(pb), dim3(kPmeBlock), 0, s, (long long)num_pairs, max_exclusions, neighbors,
                       neighbors + num_pairs, deltas, distances, charges, exclusions, alpha, coulomb, w.count, w.incoming, w.cap, w.spill,
                       w.partial);
    hipLaunchKernelGGL(pme_direct_gather, dim3(gb), dim3(kPmeBlock), 0, s, num_atoms, max_exclusions, positions, charges, exclusions, alpha,
                       coulomb, w.count, w.incoming, w.cap, w.spill, position_deriv, charge_deriv, w.partial + pb);
    hipLaunchKernelGGL(pme_sum_partials, dim3(1), dim3(1024), 0, s, w.partial, pb + gb, energy);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

int64_t nnpops_pme_direct_indexed_workspace_bytes(int64_t num_pairs, int num_atoms) {
    if (num_pairs < 0 || num_atoms < 0) return 0;
    const size_t a = (sizeof(double) * (size_t)(pair_blocks(num_pairs) + gather_blocks_full(num_atoms)) + 255) & ~(size_t)255;
    return (int64_t)(a + 2 * (((size_t)num_pairs * 16 + 255) & ~(size_t)255) + 512);
}

int nnpops_pme_direct_indexed(int num_atoms, int64_t num_pairs, int max_exclusions, const float* positions, const float* charges,
                              const int32_t* neighbors, const float* deltas, const float* distances, const int32_t* exclusions,
                              const int32_t* index, float alpha, float coulomb, float* energy, float* position_deriv,
                              float* charge_deriv, void* workspace, void* stream) {
    NNPOPS_REQUIRE(num_atoms > 0 && num_pairs >= 0 && max_exclusions >= 0, "bad sizes (atoms %d, pairs %lld, exclusions %d)", num_atoms,
                   (long long)num_pairs, max_exclusions);
    NNPOPS_REQUIRE(alpha > 0 && coulomb > 0, "alpha and coulomb must be positive");
    NNPOPS_REQUIRE(positions && charges && energy && position_deriv && charge_deriv && workspace && index, "NULL device pointer");
    NNPOPS_REQUIRE(num_pairs == 0 || (neighbors && deltas && distances), "NULL pair-list pointer");
    NNPOPS_REQUIRE(max_exclusions == 0 || exclusions, "NULL exclusions pointer");
    hipStream_t s = (hipStream_t)stream;
    const int pb = pair_blocks(num_pairs), gb = gather_blocks_full(num_atoms);
    uintptr_t p = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    auto take = [&](size_t bytes) { const uintptr_t at = p; p += (bytes + 255) & ~(size_t)255; return at; };
    double* partial = (double*)take(sizeof(double) * (size_t)(pb + gb));
    float4* terms = (float4*)take((size_t)num_pairs * 16);
    float4* second = (float4*)take((size_t)num_pairs * 16);
    const int* order = index;
    const int2* row_seg = (const int2*)(index + ((num_pairs + 1) & ~1ll));     // (the layout nnpops_neighbor_pairs_build_index writes)
    const int2* col_seg = row_seg + num_atoms;
    hipLaunchKernelGGL(pme_direct_terms, dim3(pb), dim3(kPmeBlock), 0, s, (long long)num_pairs, max_exclusions, neighbors, neighbors + num_pairs,
                       deltas, distances, charges, exclusions, alpha, coulomb, terms, second, partial);
    hipLaunchKernelGGL(pme_direct_gather_indexed, dim3(gb), dim3(kPmeBlock), 0, s, num_atoms, max_exclusions, positions, charges, exclusions,
                       alpha, coulomb, row_seg, col_seg, order, (const float4*)terms, (const float4*)second, position_deriv, charge_deriv,
                       partial + pb);
    hipLaunchKernelGGL(pme_sum_partials, dim3(1), dim3(1024), 0, s, partial, pb + gb, energy);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

}  // extern "C"
