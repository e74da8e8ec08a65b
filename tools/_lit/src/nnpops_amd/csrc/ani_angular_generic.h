// ani_angular_generic.h -- angular AEV forward for an ARBITRARY list of angular functions.
//
// The reference core accepts any std::vector<AngularFunction> {eta, rs, zeta, thetas} (reference
// src/ani/ANISymmetryFunctions.h:34-39, evaluated one by one in src/ani/CpuANISymmetryFunctions.cpp:153-194).  Every set
// a TorchANI model or the reference's torch binding can build is a full grid {(eta, rs)} x {(zeta, thetas)}, which is
// what the fast kernels (ani_angular_mfma.h, ani_angular_bwd.h) factor; a list that is not such a grid -- or whose grid
// has more than 16 x 8 factors -- takes the kernels in this file instead of NNPOPS_ERR_UNSUPPORTED.  Simple and
// deterministic rather than fast: per batch of 64 triples one pass with lane = triple computes the geometry, then
// lane = function walks the batch; triples are in bucket-major order, so a bucket is one contiguous run whose sum is
// STORED (no atomics).  The backward counterpart is the GENERIC instantiation of ani_angular_backward_pair.
#pragma once

#include "ani_kernels.h"

namespace nnpops {

__host__ __device__ inline size_t ang_fwd_generic_lds_bytes(int capA) {
    return (size_t)capA * 2 * sizeof(float4) + 64 * sizeof(float4) + 64 * sizeof(int);
}

template <bool TORCHANI>
__global__ __launch_bounds__(64 * kWavesPerGroup) void ani_angular_forward_generic(
    const AniParams* __restrict__ P, int cap, int capA, const float4* __restrict__ recA_g, const float4* __restrict__ recB_g,
    const int* __restrict__ tri_g, const int* __restrict__ cnt_a, const int* __restrict__ cnt_ro, float* __restrict__ angular,
    int ld_angular, int lds_per_wave) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int lane = lane_id();
    const int wig = __builtin_amdgcn_readfirstlane(wave_in_group());
    const int i = blockIdx.x * (blockDim.x >> 6) + wig;
    const int NB = P->NB, nA = P->nA;
    if (i >= P->N) return;
    char* cursor = lds_raw + (size_t)wig * lds_per_wave;
    float4* recA = (float4*)cursor;       cursor += (size_t)capA * sizeof(float4);
    float4* recB = (float4*)cursor;       cursor += (size_t)capA * sizeof(float4);
    float4* geo = (float4*)cursor;        cursor += 64 * sizeof(float4);       // {cos, sin, rbar, fc_ij fc_ik} of the batch's triples
    int* bkt = (int*)cursor;

    int n, nro;
    clamp_counts(cnt_a[i], cnt_ro[i], cap, capA, n, nro);
    const int T = (n * (n - 1)) / 2;
    const int* tri = tri_g + (size_t)i * triples_capacity(capA);
    float* out = angular + (size_t)i * ld_angular;
    for (int q = lane; q < NB * nA; q += 64) out[q] = 0.f;     // buckets without triples stay zero; the others are overwritten below
    load_angular_records(recA_g + (size_t)i * capA, recB_g + (size_t)i * capA, n, recA, recB);
    wave_fence();

    for (int mbase = 0; mbase < nA; mbase += 64) {
        const int m = min(mbase + lane, nA - 1);
        const bool live = mbase + lane < nA;
        const float cm = P->af_c[m], rs = P->af_rs[m], zeta = P->af_zeta[m], zc = P->af_cos[m], zs = P->af_sin[m];
        float acc = 0.f;
        int cur = -1;
        for (int base = 0; base < T; base += 64) {
            const int t = base + lane;
            if (t < T) {                                       // lane = triple: everything that does not depend on the function
                const int word = tri[t];
                const int p = word & 0xff, q = (word >> 8) & 0xff;
                const TripleGeom g = triple_geometry<TORCHANI>(recA[p], recB[p], recA[q], recB[q]);
                geo[lane] = make_float4(g.c, g.s, g.rbar, g.fcfc);
                bkt[lane] = word >> 16;
            }
            wave_fence();
            const int count = min(64, T - base);
            for (int u = 0; u < count; u++) {                  // lane = function
                const int b = __builtin_amdgcn_readfirstlane(bkt[u]);
                if (b != cur) {                                // a bucket is ONE contiguous run of the list: store its sum
                    if (cur >= 0 && live) out[cur * nA + mbase + lane] = acc;
                    acc = 0.f;
                    cur = b;
                }
                const float4 g = geo[u];
                const float x = fmaxf(1.0f + (g.x * zc + g.y * zs), 1e-30f);            // 1 + cos(theta - thetas_m)
                const float sh = g.z - rs;
                // (1+cos)^zeta * 2^(1-zeta) * exp(-eta (rbar - Rs)^2) as ONE exp2                         ref :176-190, :104-109
                acc += g.w * fast_exp2(fmaf(zeta, fast_log2(x), (1.0f - zeta) + cm * sh * sh));
            }
            wave_fence();
        }
        if (cur >= 0 && live) out[cur * nA + mbase + lane] = acc;
    }
}

// Forces of one triple for an arbitrary function list: the three chain-rule routes (ref :311-348) accumulated function by
// function.  Gb: the upstream gradients of the triple's bucket, Gb[m] for function m (global memory, caller's order).
template <bool TORCHANI>
__device__ __forceinline__ void triple_forces_generic(const AniParams* __restrict__ P, int nA, const float4& A, const float4& A2,
                                                      const float4& B, const float4& B2, const float* __restrict__ Gb,
                                                      float& alpha_p, float& alpha_q, float& beta) {
    const TripleGeom g = triple_geometry<TORCHANI>(A, A2, B, B2);
    float s0 = 0.f, sr = 0.f, sth = 0.f;
    for (int m = 0; m < nA; m++) {
        const float zeta = P->af_zeta[m], zc = P->af_cos[m], zs = P->af_sin[m];
        const float sh = g.rbar - P->af_rs[m];
        const float R = fast_exp2(P->af_c[m] * sh * sh);
        const float dR = -P->af_eta[m] * sh * R;               // rbar carries 1/2 (ref :306)
        const float cz = g.c * zc + g.s * zs, sz = g.s * zc - g.c * zs;
        const float x = fmaxf(1.0f + cz, 1e-30f);
        const float Zm1 = fast_exp2(fmaf(zeta - 1.0f, fast_log2(x), 1.0f - zeta));     // 2^(1-zeta) (1+cos)^(zeta-1)
        const float Z = Zm1 * x, dZ = -zeta * Zm1 * sz;
        const float G = Gb[m];
        s0 += G * R * Z;
        sr += G * dR * Z;
        sth += G * R * dZ;
    }
    const float t1 = A2.y * B2.x * s0 + g.fcfc * sr;
    const float t2 = A2.x * B2.y * s0 + g.fcfc * sr;
    const float t3 = g.fcfc * sth;
    const float dot = A.x * B.x + A.y * B.y + A.z * B.z;
    const float iprod = A2.z * B2.z;
    const float damp = TORCHANI ? 0.95f : 1.0f;
    const float dadd = -damp * fast_rcp(g.s) * iprod * t3;
    const float ka = dot * A2.z * A2.z, kb = dot * B2.z * B2.z;
    alpha_p = t1 * A2.z - dadd * ka;
    alpha_q = t2 * B2.z - dadd * kb;
    beta = dadd;
}

}  // namespace nnpops
