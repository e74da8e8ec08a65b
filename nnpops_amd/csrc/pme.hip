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
// Layout for MI355X: one lane per pair slot, streaming the four list arrays (28 B per slot, coalesced).  The list that
// getNeighborPairs produces is grouped by neighbors[0], so the 64 consecutive pairs of a wave touch only two or three
// distinct first atoms: the wave adds their four contributions up with a segmented scan (shuffles) and issues ONE atomic
// per run, which halves the atomic traffic the reference pays (eight float atomics per pair); the second atom of a pair
// is scattered with float atomics.  Any other pair order is still correct, just with shorter runs.  The energy is
// accumulated in double per lane, reduced per workgroup, and the workgroup partials are summed in a fixed order by a
// second tiny kernel: the energy is bitwise reproducible, the derivatives are reproducible up to the order of the
// float atomics (as in the reference).
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
                                                             float* __restrict__ pos_deriv, float* __restrict__ charge_deriv,
                                                             double* __restrict__ partial) {
    __shared__ double red[kPmeBlock / 64];
    double energy = 0.0;
    const long long stride = (long long)gridDim.x * kPmeBlock;
    const long long first = (long long)blockIdx.x * kPmeBlock + threadIdx.x;
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
                atomicAdd(&pos_deriv[3 * atom2], fx);
                atomicAdd(&pos_deriv[3 * atom2 + 1], fy);
                atomicAdd(&pos_deriv[3 * atom2 + 2], fz);
                atomicAdd(&charge_deriv[atom2], cd2);
            }
        }
        // first atom of the pair: one atomic per run of equal atoms inside the wave
        const int key = include ? atom1 : -1 - lane_id();          // (excluded / empty slots never join a run)
        float sx = -fx, sy = -fy, sz = -fz, sc = cd1;
        segmented_scan4(key, sx, sy, sz, sc);
        const int next_key = __shfl_down(key, 1, 64);
        if (include && (lane_id() == 63 || next_key != key)) {
            atomicAdd(&pos_deriv[3 * atom1], sx);
            atomicAdd(&pos_deriv[3 * atom1 + 1], sy);
            atomicAdd(&pos_deriv[3 * atom1 + 2], sz);
            atomicAdd(&charge_deriv[atom1], sc);
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

// Excluded pairs, each once (atom2 > atom1), without periodic wrap: subtract what reciprocal space adds   (ref :71-99)
__global__ __launch_bounds__(kPmeBlock) void pme_direct_exclusions(int num_atoms, int max_excl, const float* __restrict__ pos,
                                                                  const float* __restrict__ charge, const int* __restrict__ excl,
                                                                  float alpha, float coulomb, float* __restrict__ pos_deriv,
                                                                  float* __restrict__ charge_deriv, double* __restrict__ partial) {
    __shared__ double red[kPmeBlock / 64];
    double energy = 0.0;
    const long long total = (long long)num_atoms * max_excl;
    for (long long idx = (long long)blockIdx.x * kPmeBlock + threadIdx.x; idx < total; idx += (long long)gridDim.x * kPmeBlock) {
        const int atom1 = (int)(idx / max_excl);
        const int atom2 = excl[idx];
        if (atom2 <= atom1) continue;
        const float dx = pos[3 * atom1] - pos[3 * atom2], dy = pos[3 * atom1 + 1] - pos[3 * atom2 + 1], dz = pos[3 * atom1 + 2] - pos[3 * atom2 + 2];
        const float r = sqrtf(dx * dx + dy * dy + dz * dz);
        const float inv_r = 1.0f / r, ar = alpha * r;
        const float ex = expf(-ar * ar), erf_ar = erff(ar);
        const float pre = coulomb * inv_r;
        const float c1 = charge[atom1], c2 = charge[atom2];
        energy -= (double)(pre * erf_ar * c1 * c2);
        atomicAdd(&charge_deriv[atom1], -pre * erf_ar * c2);
        atomicAdd(&charge_deriv[atom2], -pre * erf_ar * c1);
        const float dedr = pre * c1 * c2 * (erf_ar - ar * ex * kTwoOverSqrtPi) * inv_r * inv_r;
        atomicAdd(&pos_deriv[3 * atom1], dedr * dx); atomicAdd(&pos_deriv[3 * atom1 + 1], dedr * dy); atomicAdd(&pos_deriv[3 * atom1 + 2], dedr * dz);
        atomicAdd(&pos_deriv[3 * atom2], -dedr * dx); atomicAdd(&pos_deriv[3 * atom2 + 1], -dedr * dy); atomicAdd(&pos_deriv[3 * atom2 + 2], -dedr * dz);
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
__global__ __launch_bounds__(256) void pme_sum_partials(const double* __restrict__ partial, int count, float* __restrict__ energy) {
    __shared__ double red[256];
    double e = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) e += partial[i];
    red[threadIdx.x] = e;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *energy = (float)red[0];
}

int pair_blocks(long long num_pairs) { return (int)std::min<long long>(std::max<long long>(1, (num_pairs + kPmeBlock - 1) / kPmeBlock), 256 * 16); }
int excl_blocks(long long slots) { return (int)std::min<long long>(std::max<long long>(1, (slots + kPmeBlock - 1) / kPmeBlock), 256 * 4); }

}  // namespace

extern "C" {

int64_t nnpops_pme_direct_workspace_bytes(int64_t num_pairs, int num_atoms, int max_exclusions) {
    if (num_pairs < 0 || num_atoms < 0 || max_exclusions < 0) return 0;
    return (int64_t)sizeof(double) * (pair_blocks(num_pairs) + excl_blocks((long long)num_atoms * max_exclusions)) + 256;
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
    double* partial = (double*)(((uintptr_t)workspace + 7) & ~(uintptr_t)7);
    const int pb = pair_blocks(num_pairs), eb = excl_blocks((long long)num_atoms * max_exclusions);
    NNPOPS_HIP_TRY(hipMemsetAsync(position_deriv, 0, sizeof(float) * 3 * (size_t)num_atoms, s));
    NNPOPS_HIP_TRY(hipMemsetAsync(charge_deriv, 0, sizeof(float) * (size_t)num_atoms, s));
    hipLaunchKernelGGL(pme_direct_pairs, dim3(pb), dim3(kPmeBlock), 0, s, (long long)num_pairs, max_exclusions, neighbors,
                       neighbors + num_pairs, deltas, distances, charges, exclusions, alpha, coulomb, position_deriv, charge_deriv, partial);
    int count = pb;
    if (max_exclusions > 0) {
        hipLaunchKernelGGL(pme_direct_exclusions, dim3(eb), dim3(kPmeBlock), 0, s, num_atoms, max_exclusions, positions, charges,
                           exclusions, alpha, coulomb, position_deriv, charge_deriv, partial + pb);
        count += eb;
    }
    hipLaunchKernelGGL(pme_sum_partials, dim3(1), dim3(256), 0, s, partial, count, energy);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

}  // extern "C"
