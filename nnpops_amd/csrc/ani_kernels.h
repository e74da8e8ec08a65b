// ani_kernels.h -- gfx950 kernels for the ANI atomic-environment-vector symmetry functions.
//
// What is computed (reference src/ani/CpuANISymmetryFunctions.cpp, maths in SURVEY.md App. A):
//   radial [i][s][k]  = scale_R * sum_{j: spec j = s, r_ij < Rcr} fc(r_ij;Rcr) exp(-eta_k (r_ij-Rs_k)^2)      (ref :112-151)
//   angular[i][b][m]  = 2^(1-zeta_m) * sum_{j<k nbrs of i, r < Rca, b = pair(spec j, spec k)}
//                         (1+cos(theta-ths_m))^zeta_m exp(-eta_m((r_ij+r_ik)/2-Rs_m)^2) fc(r_ij)fc(r_ik)       (ref :153-194)
//   and the analytic position gradients of both                                                              (ref :196-353)
//
// How it is laid out for CDNA4 (this is a new design, not the reference's CUDA launch shapes):
//   * one single-wave workgroup (64 lanes) per centre atom; everything an atom needs lives in that
//     wave's LDS slice, so the only HBM traffic is: positions/species gathers (L2 resident), the
//     neighbour row, and ONE coalesced write (forward) or read (backward) of the atom's AEV row.
//   * neighbour rows [N][cap]: angular neighbours (r < Rca) packed from the front, radial-only
//     neighbours (Rca <= r < Rcr) packed from the back; both in ascending atom order.
//   * the angular functions factor as R_a(rbar) x Z_z(theta) (8 x 4 for ANI-2x): a triple costs
//     nFR exp2 + nFZ (log2+exp2) instead of nA (powf+cosf+expf).
//   * forward, two phases per batch of 64 triples:
//       phase 1  lane = triple:           geometry, R_a, fc*fc*Z_z  -> LDS (no cross-lane traffic)
//       phase 2  lane = (stream, a):      acc[z] += R_a * Z_z over the stream's triples; triples are
//                                         sorted by species pair so a stream flushes its 4 partial sums
//                                         into the LDS output row only when the pair bucket changes.
//   * backward: lane = triple; the atom's upstream-gradient row sits in LDS in canonical
//     [bucket][a][z] order and is contracted with R, dR, Z, dZ in registers; forces on the two leg
//     atoms go to per-neighbour LDS accumulators, then one global atomic per neighbour component.
//   * radial backward is owner-computes (each atom walks its full row, reading both gradient
//     rows), so it needs no atomics and also initialises position_deriv.
#pragma once

#include "celllist.h"
#include "device_common.h"

namespace nnpops {

constexpr int kMaxRadialFns = 64;
constexpr int kMaxFactor = 16;       // max distinct (eta,rs) or (zeta,thetas) factors of the angular set
constexpr int kMaxAngularFns = 256;
constexpr int kMaxSpecies = 32;

struct AniParams {
    int N, S, nR, nA, NB, nFR, nFZ;
    int periodic, torchani;
    float rcr, rca, rcr2, rca2;
    float radial_scale;              // 0.25 (TorchANI) or 1          ref :99-103
    float angle_damp;                // 0.95 (TorchANI) or 1          ref :391-392
    float rad_c[kMaxRadialFns];      // -eta_k * log2(e)
    float rad_eta[kMaxRadialFns];
    float rad_rs[kMaxRadialFns];
    float fr_c[kMaxFactor];          // -eta_a * log2(e)
    float fr_eta[kMaxFactor];
    float fr_rs[kMaxFactor];
    float fz_zeta[kMaxFactor];
    float fz_cos[kMaxFactor];        // cos(thetas_z)
    float fz_sin[kMaxFactor];        // sin(thetas_z)
    float fz_scale[kMaxFactor];      // 2^(1-zeta_z)                  ref :104-109
    int m_of[kMaxAngularFns];        // canonical (a*nFZ+z) -> position m inside a species-pair block
};

// status words written by the neighbour builder
enum { kStatOverflow = 0, kStatMaxRow = 1, kStatMaxAngular = 2, kStatWords = 4 };

// Largest row / angular count of the last neighbour build, reduced from the per-atom counts only when
// the host asks (nnpops_ani_check).  Doing this with per-wave atomics inside the builders serialised
// 10k waves on one L2 word (measured ~240 us), so the builders publish nothing but their counts.
__global__ __launch_bounds__(256) void ani_row_stats(int N, const int* __restrict__ cnt_a, const int* __restrict__ cnt_ro,
                                                     int cap, int cap_angular, int* __restrict__ status) {
    __shared__ int red[2][256];
    int mrow = 0, mang = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        mrow = max(mrow, cnt_a[i] + cnt_ro[i]);
        mang = max(mang, cnt_a[i]);
    }
    red[0][threadIdx.x] = mrow;
    red[1][threadIdx.x] = mang;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] = max(red[0][threadIdx.x], red[0][threadIdx.x + off]);
            red[1][threadIdx.x] = max(red[1][threadIdx.x], red[1][threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMax(&status[kStatMaxRow], red[0][0]);
        atomicMax(&status[kStatMaxAngular], red[1][0]);
        if (red[0][0] > cap || red[1][0] > cap_angular) atomicOr(&status[kStatOverflow], 1);
    }
}

// =============================================================================================
// Neighbour rows, all-pairs scan (the reference's O(N^2) search, one wave per atom).
// =============================================================================================
template <bool PERIODIC>
__global__ __launch_bounds__(64) void ani_neighbors_allpairs(const AniParams* __restrict__ P,
                                                             const float* __restrict__ pos,
                                                             const float* __restrict__ box, int* __restrict__ nbr,
                                                             int cap, int cap_angular, int* __restrict__ cnt_a,
                                                             int* __restrict__ cnt_ro, int* __restrict__ status) {
    const int i = blockIdx.x;
    const int lane = lane_id();
    const int N = P->N;
    const float rcr2 = P->rcr2, rca2 = P->rca2;
    Box b{};
    if (PERIODIC) b = load_box(box);
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    int* row = nbr + (size_t)i * cap;
    int na = 0, nro = 0;
    for (int base = 0; base < N; base += 64) {
        const int j = base + lane;
        bool in_r = false, in_a = false;
        if (j < N && j != i) {
            float dx = pos[3 * j] - xi, dy = pos[3 * j + 1] - yi, dz = pos[3 * j + 2] - zi;
            min_image<PERIODIC>(dx, dy, dz, b);
            const float r2 = dx * dx + dy * dy + dz * dz;
            in_r = r2 < rcr2;
            in_a = in_r && (r2 < rca2);
        }
        const bool in_ro = in_r && !in_a;
        const unsigned long long ma = __ballot(in_a), mro = __ballot(in_ro);
        if (in_a) {
            const int slot = na + prefix_popc(ma);
            if (slot < cap) row[slot] = j;
        }
        if (in_ro) {
            const int slot = nro + prefix_popc(mro);
            if (slot < cap) row[cap - 1 - slot] = j;
        }
        na += __popcll(ma);
        nro += __popcll(mro);
    }
    if (lane == 0) {
        cnt_a[i] = na;
        cnt_ro[i] = nro;
    }
}

// =============================================================================================
// Neighbour rows from the cell grid (celllist.h): one wave per atom walks the 3x3x3 stencil of its
// cell; candidates are read as coalesced float4 {x,y,z,id} runs.  Same row format as above; the
// order inside a row is the (deterministic) stencil order.
// =============================================================================================
template <bool PERIODIC>
__global__ __launch_bounds__(64) void ani_neighbors_cells(const AniParams* __restrict__ P,
                                                          const float* __restrict__ box,
                                                          const CellGrid* __restrict__ grid,
                                                          const int* __restrict__ cell_start,
                                                          const int* __restrict__ atom_cell,
                                                          const float4* __restrict__ sorted_pos, int* __restrict__ nbr,
                                                          int cap, int cap_angular, int* __restrict__ cnt_a,
                                                          int* __restrict__ cnt_ro, int* __restrict__ status) {
    const int lane = lane_id();
    const CellGrid g = *grid;
    if (!g.ok) {                                           // box too small for the stencil: tell the host
        if (blockIdx.x == 0 && lane == 0) atomicOr(&status[kStatOverflow], 2);
        return;
    }
    const float rcr2 = P->rcr2, rca2 = P->rca2;
    Box b{};
    if (PERIODIC) b = load_box(box);
    const float4 me = sorted_pos[blockIdx.x];
    const int i = __float_as_int(me.w);
    const int c = atom_cell[i];
    const int cx = c % g.nx, cy = (c / g.nx) % g.ny, cz = c / (g.nx * g.ny);
    int* row = nbr + (size_t)i * cap;
    int na = 0, nro = 0;
    for_each_stencil_range(g, cell_start, cx, cy, cz, [&](int begin, int end) {
        for (int base = begin; base < end; base += 64) {
            const int k = base + lane;
            bool in_r = false, in_a = false;
            int j = -1;
            if (k < end) {
                const float4 pj = sorted_pos[k];
                j = __float_as_int(pj.w);
                if (j != i) {
                    float dx = pj.x - me.x, dy = pj.y - me.y, dz = pj.z - me.z;
                    min_image<PERIODIC>(dx, dy, dz, b);
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    in_r = r2 < rcr2;
                    in_a = in_r && (r2 < rca2);
                }
            }
            const bool in_ro = in_r && !in_a;
            const unsigned long long ma = __ballot(in_a), mro = __ballot(in_ro);
            if (in_a) {
                const int slot = na + prefix_popc(ma);
                if (slot < cap) row[slot] = j;
            }
            if (in_ro) {
                const int slot = nro + prefix_popc(mro);
                if (slot < cap) row[cap - 1 - slot] = j;
            }
            na += __popcll(ma);
            nro += __popcll(mro);
        }
    });
    if (lane == 0) {
        cnt_a[i] = na;
        cnt_ro[i] = nro;
    }
}

// Clamp the per-atom counts so that consumers never index outside a row even after an overflow
// (results of an overflowed compute are garbage and reported through nnpops_ani_check).
__device__ __forceinline__ void clamped_counts(const int* cnt_a, const int* cnt_ro, int i, int cap, int cap_angular,
                                               int& na, int& nro) {
    na = min(cnt_a[i], min(cap, cap_angular));
    nro = min(cnt_ro[i], cap - na);
}

// =============================================================================================
// Radial forward.  LDS: per-neighbour {r, fc, species} + the [S][nR] accumulator row.
// =============================================================================================
template <bool PERIODIC>
__global__ __launch_bounds__(64) void ani_radial_forward(const AniParams* __restrict__ P,
                                                         const float* __restrict__ pos,
                                                         const float* __restrict__ box,
                                                         const int* __restrict__ species,
                                                         const int* __restrict__ nbr, int cap, int cap_angular,
                                                         const int* __restrict__ cnt_a,
                                                         const int* __restrict__ cnt_ro, float* __restrict__ radial) {
    extern __shared__ float lds[];
    const int i = blockIdx.x, lane = lane_id();
    const int S = P->S, nR = P->nR, width = S * nR;
    float* acc = lds;                        // [S*nR]
    float* nb_r = acc + width;               // [cap]
    float* nb_fc = nb_r + cap;               // [cap]
    int* nb_sp = (int*)(nb_fc + cap);        // [cap]

    int na, nro;
    clamped_counts(cnt_a, cnt_ro, i, cap, cap_angular, na, nro);
    const int total = na + nro;
    const int* row = nbr + (size_t)i * cap;
    Box b{};
    if (PERIODIC) b = load_box(box);
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    const float rcr = P->rcr;

    for (int e = lane; e < total; e += 64) {
        const int j = e < na ? row[e] : row[cap - 1 - (e - na)];
        float dx = pos[3 * j] - xi, dy = pos[3 * j + 1] - yi, dz = pos[3 * j + 2] - zi;
        min_image<PERIODIC>(dx, dy, dz, b);
        const float r = sqrtf(dx * dx + dy * dy + dz * dz);
        nb_r[e] = r;
        nb_fc[e] = 0.5f * cosf(kPi * r / rcr) + 0.5f;
        nb_sp[e] = species[j];
    }
    __syncthreads();

    // lanes = (stream, k): KP = smallest power of two >= nR.  Each lane keeps one partial sum per
    // species in registers (select-accumulate; LDS float atomics cost ~3 cycles per lane on gfx950,
    // see tools/ubench/lds_atomic.hip), streams are folded with xor-shuffles at the end.
    int KP = 1;
    while (KP < nR) KP <<= 1;
    const int k = lane & (KP - 1), stream = lane / KP, nstreams = 64 / KP;
    const float ck = P->rad_c[min(k, nR - 1)], rs = P->rad_rs[min(k, nR - 1)];
    const float scale = P->radial_scale;
    float* out = radial + (size_t)i * width;
    constexpr int SCHUNK = 8;
    for (int s0 = 0; s0 < S; s0 += SCHUNK) {           // one pass per group of 8 species (one pass for ANI)
        float part[SCHUNK];
#pragma unroll
        for (int s = 0; s < SCHUNK; s++) part[s] = 0.f;
        for (int e = stream; e < total; e += nstreams) {
            const float sh = nb_r[e] - rs;
            const float v = nb_fc[e] * fast_exp2(ck * sh * sh);
            const int sp = nb_sp[e] - s0;
#pragma unroll
            for (int s = 0; s < SCHUNK; s++) part[s] += (sp == s) ? v : 0.f;
        }
#pragma unroll
        for (int s = 0; s < SCHUNK; s++) {
            float v = part[s];
            for (int off = KP; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            if (stream == 0 && k < nR && s0 + s < S) out[(s0 + s) * nR + k] = v * scale;
        }
    }
}

// =============================================================================================
// Radial backward (owner computes; writes position_deriv[i], no atomics).      ref :228-263
// =============================================================================================
template <bool PERIODIC>
__global__ __launch_bounds__(64) void ani_radial_backward(const AniParams* __restrict__ P,
                                                          const float* __restrict__ pos,
                                                          const float* __restrict__ box,
                                                          const int* __restrict__ species,
                                                          const int* __restrict__ nbr, int cap, int cap_angular,
                                                          const int* __restrict__ cnt_a,
                                                          const int* __restrict__ cnt_ro,
                                                          const float* __restrict__ radial_grad,
                                                          float* __restrict__ pos_grad) {
    extern __shared__ float lds[];
    const int i = blockIdx.x, lane = lane_id();
    const int S = P->S, nR = P->nR, width = S * nR;
    float* g_own = lds;                       // [S*nR] this atom's gradient row
    float* nb_r = g_own + width;              // [cap]
    float* nb_fc = nb_r + cap;
    float* nb_dfc = nb_fc + cap;
    float* nb_ux = nb_dfc + cap;              // unit vector i -> j
    float* nb_uy = nb_ux + cap;
    float* nb_uz = nb_uy + cap;
    int* nb_sp = (int*)(nb_uz + cap);
    int* nb_j = nb_sp + cap;

    int na, nro;
    clamped_counts(cnt_a, cnt_ro, i, cap, cap_angular, na, nro);
    const int total = na + nro;
    const int* row = nbr + (size_t)i * cap;
    Box b{};
    if (PERIODIC) b = load_box(box);
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    const float rcr = P->rcr;
    const int si = species[i];

    const float* gi = radial_grad + (size_t)i * width;
    for (int q = lane; q < width; q += 64) g_own[q] = gi[q];
    for (int e = lane; e < total; e += 64) {
        const int j = e < na ? row[e] : row[cap - 1 - (e - na)];
        float dx = pos[3 * j] - xi, dy = pos[3 * j + 1] - yi, dz = pos[3 * j + 2] - zi;
        min_image<PERIODIC>(dx, dy, dz, b);
        const float r = sqrtf(dx * dx + dy * dy + dz * dz);
        const float rinv = 1.0f / r;
        const float arg = kPi * r / rcr;
        nb_r[e] = r;
        nb_fc[e] = 0.5f * cosf(arg) + 0.5f;
        nb_dfc[e] = -(0.5f * kPi / rcr) * sinf(arg);
        nb_ux[e] = dx * rinv; nb_uy[e] = dy * rinv; nb_uz[e] = dz * rinv;
        nb_sp[e] = species[j];
        nb_j[e] = j;
    }
    __syncthreads();

    int KP = 1;
    while (KP < nR) KP <<= 1;
    const int k = lane & (KP - 1), stream = lane / KP, nstreams = 64 / KP;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (k < nR) {
        const float ck = P->rad_c[k], rs = P->rad_rs[k], eta = P->rad_eta[k];
        for (int e = stream; e < total; e += nstreams) {
            const float sh = nb_r[e] - rs;
            const float ex = fast_exp2(ck * sh * sh);
            const float dvdr = (nb_dfc[e] - nb_fc[e] * 2.f * eta * sh) * ex;
            const float dedv = g_own[nb_sp[e] * nR + k] + radial_grad[(size_t)nb_j[e] * width + si * nR + k];
            const float sc = dedv * dvdr;
            fx -= sc * nb_ux[e]; fy -= sc * nb_uy[e]; fz -= sc * nb_uz[e];
        }
    }
    fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
    if (lane == 0) {
        const float scale = P->radial_scale;
        pos_grad[3 * i] = fx * scale;
        pos_grad[3 * i + 1] = fy * scale;
        pos_grad[3 * i + 2] = fz * scale;
    }
}

// =============================================================================================
// Angular kernels.
// =============================================================================================
// Per-neighbour record kept in LDS, sorted by species so that triples come out grouped by bucket.
struct AngRec {
    float dx, dy, dz, r;       // displacement i -> j and its length
};
struct AngRec2 {
    float fc, dfc, rinv;
    int sp;
};

// (p, q) with p < q of the t-th pair in row-major order of the strict upper triangle of an n x n grid
__device__ __forceinline__ void decode_pair(int t, int n, int& p, int& q) {
    const float w = (float)(2 * n - 1);
    int pp = (int)((w - fast_sqrt(w * w - 8.0f * (float)t)) * 0.5f);
    pp = max(0, min(pp, n - 2));
    // offset(p) = p*(2n-p-1)/2 ; one fix-up step each way covers the rounding of the fast sqrt
    if (((pp + 1) * (2 * n - pp - 2)) / 2 <= t) pp++;
    if ((pp * (2 * n - pp - 1)) / 2 > t) pp--;
    p = pp;
    q = t - (pp * (2 * n - pp - 1)) / 2 + pp + 1;
}

// Load the angular neighbours of atom i into LDS, stably sorted by species.  Returns n.
// scratch: int[2*kMaxSpecies] in LDS.
template <bool PERIODIC>
__device__ __forceinline__ int load_sorted_angular_neighbors(const AniParams* __restrict__ P,
                                                             const float* __restrict__ pos, const Box& b,
                                                             const int* __restrict__ species,
                                                             const int* __restrict__ row, int n, int i,
                                                             AngRec* rec, AngRec2* rec2, int* rec_j, int* scratch) {
    const int lane = lane_id();
    const int S = P->S;
    int* tot = scratch;              // [S] species totals, then exclusive prefix
    int* run = scratch + kMaxSpecies;  // [S] running within-species offsets
    for (int s = lane; s < S; s += 64) { tot[s] = 0; run[s] = 0; }
    __syncthreads();
    for (int e = lane; e < n; e += 64) atomicAdd(&tot[species[row[e]]], 1);
    __syncthreads();
    if (lane == 0) {
        int accum = 0;
        for (int s = 0; s < S; s++) { const int c = tot[s]; tot[s] = accum; accum += c; }
    }
    __syncthreads();
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    const float rca = P->rca;
    for (int base = 0; base < n; base += 64) {
        const int e = base + lane;
        const bool valid = e < n;
        int j = 0, sp = -1;
        float dx = 0, dy = 0, dz = 0;
        if (valid) {
            j = row[e];
            sp = species[j];
            dx = pos[3 * j] - xi; dy = pos[3 * j + 1] - yi; dz = pos[3 * j + 2] - zi;
            min_image<PERIODIC>(dx, dy, dz, b);
        }
        int rank = 0;
        for (int s = 0; s < S; s++) {
            const unsigned long long m = __ballot(valid && sp == s);
            if (sp == s) rank = tot[s] + run[s] + prefix_popc(m);
            __syncthreads();
            if (lane == 0) run[s] += __popcll(m);
            __syncthreads();
        }
        if (valid) {
            const float r = sqrtf(dx * dx + dy * dy + dz * dz);
            const float arg = kPi * r / rca;
            rec[rank] = AngRec{dx, dy, dz, r};
            rec2[rank] = AngRec2{0.5f * cosf(arg) + 0.5f, -(0.5f * kPi / rca) * sinf(arg), 1.0f / r, sp};
            if (rec_j) rec_j[rank] = j;
        }
    }
    __syncthreads();
    return n;
}

// LDS carve-up shared by the two angular kernels (all offsets multiples of 16 bytes).
struct AngLds {
    AngRec* rec;      // [capA]
    AngRec2* rec2;    // [capA]
    int* rec_j;       // [capA]
    float* row;       // [NB][NFRP][NFZP] canonical accumulator (fwd) / scaled upstream gradient (bwd)
    float* facR;      // fwd: [64][NFRP]        bwd: per-neighbour force accumulators [capA][4]
    float* facZ;      // fwd: [64][NFZP]
    int* facB;        // fwd: [64]
    int* scratch;     // [2*kMaxSpecies]
};

template <int NFRP, int NFZP>
__host__ __device__ inline size_t ang_lds_bytes(int capA, int NB, bool forward) {
    size_t b = (size_t)capA * (sizeof(AngRec) + sizeof(AngRec2) + sizeof(int));
    b += (size_t)NB * NFRP * NFZP * sizeof(float);
    b += forward ? (size_t)64 * (NFRP + NFZP + 1) * sizeof(float) : (size_t)capA * 4 * sizeof(float);
    b += 2 * kMaxSpecies * sizeof(int);
    return b;
}

template <int NFRP, int NFZP>
__device__ __forceinline__ AngLds carve_lds(char* base, int capA, int NB, bool forward) {
    AngLds L;
    L.rec = (AngRec*)base;            base += (size_t)capA * sizeof(AngRec);
    L.rec2 = (AngRec2*)base;          base += (size_t)capA * sizeof(AngRec2);
    L.row = (float*)base;             base += (size_t)NB * NFRP * NFZP * sizeof(float);
    if (forward) {
        L.facR = (float*)base;        base += (size_t)64 * NFRP * sizeof(float);
        L.facZ = (float*)base;        base += (size_t)64 * NFZP * sizeof(float);
        L.facB = (int*)base;          base += (size_t)64 * sizeof(int);
    } else {
        L.facR = (float*)base;        base += (size_t)capA * 4 * sizeof(float);
        L.facZ = nullptr;
        L.facB = nullptr;
    }
    L.rec_j = (int*)base;             base += (size_t)capA * sizeof(int);
    L.scratch = (int*)base;
    return L;
}

// Geometry of one triple, shared by forward and backward.
struct TripleGeom {
    float c, s;        // cos / sin of the (damped) angle
    float rbar;        // (r_ij + r_ik)/2
    float fcfc;
    int bucket;
};

template <bool TORCHANI>
__device__ __forceinline__ TripleGeom triple_geometry(const AngRec& A, const AngRec2& A2, const AngRec& B,
                                                      const AngRec2& B2, int S) {
    TripleGeom g;
    const float dot = A.dx * B.dx + A.dy * B.dy + A.dz * B.dz;
    const float iprod = A2.rinv * B2.rinv;
    if (TORCHANI) {
        g.c = 0.95f * dot * iprod;                       // ref :391-393
        g.s = fast_sqrt(1.0f - g.c * g.c);               // |c| <= 0.95: no cancellation
    } else {
        g.c = fminf(fmaxf(dot * iprod, -1.0f), 1.0f);
        // sin from the cross product: accurate next to 0 and pi, which is what the reference's
        // asin branch (ref :396-404) is there for
        const float cx = A.dy * B.dz - A.dz * B.dy, cy = A.dz * B.dx - A.dx * B.dz, cz = A.dx * B.dy - A.dy * B.dx;
        g.s = fminf(fast_sqrt(cx * cx + cy * cy + cz * cz) * iprod, 1.0f);
    }
    g.rbar = 0.5f * (A.r + B.r);
    g.fcfc = A2.fc * B2.fc;
    const int sa = A2.sp, sb = B2.sp;                    // sorted: sa <= sb
    g.bucket = sa * S - (sa * (sa - 1)) / 2 + (sb - sa); // upper-triangular row-major, ref :39-43
    return g;
}

// ---------------------------------------------------------------------------------------------
// Angular forward.
// ---------------------------------------------------------------------------------------------
template <bool PERIODIC, bool TORCHANI, int NFRP, int NFZP>
__global__ __launch_bounds__(64) void ani_angular_forward(const AniParams* __restrict__ P,
                                                          const float* __restrict__ pos,
                                                          const float* __restrict__ box,
                                                          const int* __restrict__ species,
                                                          const int* __restrict__ nbr, int cap, int capA,
                                                          const int* __restrict__ cnt_a,
                                                          const int* __restrict__ cnt_ro,
                                                          float* __restrict__ angular) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int i = blockIdx.x, lane = lane_id();
    const int S = P->S, NB = P->NB, nA = P->nA, nFR = P->nFR, nFZ = P->nFZ;
    AngLds L = carve_lds<NFRP, NFZP>(lds_raw, capA, NB, true);
    constexpr int BLK = NFRP * NFZP;                      // padded canonical block
    const int rowlen = NB * BLK;

    int n, nro;
    clamped_counts(cnt_a, cnt_ro, i, cap, capA, n, nro);
    Box b{};
    if (PERIODIC) b = load_box(box);
    for (int q = lane; q < rowlen; q += 64) L.row[q] = 0.f;
    load_sorted_angular_neighbors<PERIODIC>(P, pos, b, species, nbr + (size_t)i * cap, n, i, L.rec, L.rec2, nullptr,
                                            L.scratch);

    // per-lane constants of the two factor families
    float frc[NFRP], frs[NFRP], zz[NFZP], zc[NFZP], zs[NFZP];
#pragma unroll
    for (int a = 0; a < NFRP; a++) { frc[a] = a < nFR ? P->fr_c[a] : 0.f; frs[a] = a < nFR ? P->fr_rs[a] : 0.f; }
#pragma unroll
    for (int z = 0; z < NFZP; z++) {
        zz[z] = z < nFZ ? P->fz_zeta[z] : 1.f;
        zc[z] = z < nFZ ? P->fz_cos[z] : 0.f;
        zs[z] = z < nFZ ? P->fz_sin[z] : 0.f;
    }

    constexpr int NSTREAM = 64 / NFRP;                    // phase-2 streams; each owns every NSTREAM-th triple
    const int a2 = lane & (NFRP - 1), stream = lane / NFRP;
    const int T = (n * (n - 1)) / 2;
    for (int base = 0; base < T; base += 64) {
        // ---------------- phase 1: lane = triple ----------------
        const int t = base + lane;
        if (t < T) {
            int p, q;
            decode_pair(t, n, p, q);
            const AngRec A = L.rec[p], B = L.rec[q];
            const AngRec2 A2 = L.rec2[p], B2 = L.rec2[q];
            const TripleGeom g = triple_geometry<TORCHANI>(A, A2, B, B2, S);
#pragma unroll
            for (int a = 0; a < NFRP; a++) {
                const float sh = g.rbar - frs[a];
                L.facR[lane * NFRP + a] = fast_exp2(frc[a] * sh * sh);
            }
#pragma unroll
            for (int z = 0; z < NFZP; z++) {
                const float x = fmaxf(1.0f + (g.c * zc[z] + g.s * zs[z]), 1e-30f);   // 1 + cos(theta - ths)
                L.facZ[lane * NFZP + z] = g.fcfc * fast_exp2(zz[z] * fast_log2(x));
            }
            L.facB[lane] = g.bucket;
        }
        __syncthreads();
        // ---------------- phase 2: lane = (stream, a) ----------------
        {
            const int count = min(64, T - base);
            float acc[NFZP];
#pragma unroll
            for (int z = 0; z < NFZP; z++) acc[z] = 0.f;
            int cur = -1;
            for (int u = stream; u < count; u += NSTREAM) {
                const int bkt = L.facB[u];
                if (bkt != cur) {
                    if (cur >= 0) {
#pragma unroll
                        for (int z = 0; z < NFZP; z++) atomicAdd(&L.row[cur * BLK + a2 * NFZP + z], acc[z]);
                    }
#pragma unroll
                    for (int z = 0; z < NFZP; z++) acc[z] = 0.f;
                    cur = bkt;
                }
                const float R = L.facR[u * NFRP + a2];
#pragma unroll
                for (int z = 0; z < NFZP; z++) acc[z] += R * L.facZ[u * NFZP + z];
            }
            if (cur >= 0) {
#pragma unroll
                for (int z = 0; z < NFZP; z++) atomicAdd(&L.row[cur * BLK + a2 * NFZP + z], acc[z]);
            }
        }
        __syncthreads();
    }

    // ---------------- epilogue: canonical LDS row -> reference column order, one coalesced write ----------------
    float* out = angular + (size_t)i * NB * nA;
    const int per_block = nFR * nFZ;                       // == nA
    for (int q = lane; q < NB * per_block; q += 64) {
        const int bkt = q / per_block, c = q - bkt * per_block;
        const int a = c / nFZ, z = c - a * nFZ;
        out[bkt * nA + P->m_of[c]] = L.row[bkt * BLK + a * NFZP + z] * P->fz_scale[z];
    }
}

// ---------------------------------------------------------------------------------------------
// Angular backward.                                                              ref :265-353
// ---------------------------------------------------------------------------------------------
template <bool PERIODIC, bool TORCHANI, int NFRP, int NFZP>
__global__ __launch_bounds__(64) void ani_angular_backward(const AniParams* __restrict__ P,
                                                           const float* __restrict__ pos,
                                                           const float* __restrict__ box,
                                                           const int* __restrict__ species,
                                                           const int* __restrict__ nbr, int cap, int capA,
                                                           const int* __restrict__ cnt_a,
                                                           const int* __restrict__ cnt_ro,
                                                           const float* __restrict__ angular_grad,
                                                           float* __restrict__ pos_grad) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int i = blockIdx.x, lane = lane_id();
    const int S = P->S, NB = P->NB, nA = P->nA, nFR = P->nFR, nFZ = P->nFZ;
    AngLds L = carve_lds<NFRP, NFZP>(lds_raw, capA, NB, false);
    constexpr int BLK = NFRP * NFZP;
    float* facc = L.facR;                                  // [capA][4] leg-atom force accumulators

    int n, nro;
    clamped_counts(cnt_a, cnt_ro, i, cap, capA, n, nro);
    if (n < 2) return;                                     // no triples: nothing to add (wave-uniform)
    Box b{};
    if (PERIODIC) b = load_box(box);

    // upstream gradient row -> canonical [bucket][a][z] order, pre-multiplied by 2^(1-zeta)
    for (int q = lane; q < NB * BLK; q += 64) L.row[q] = 0.f;
    __syncthreads();
    {
        const float* g = angular_grad + (size_t)i * NB * nA;
        const int per_block = nFR * nFZ;
        for (int q = lane; q < NB * per_block; q += 64) {
            const int bkt = q / per_block, c = q - bkt * per_block;
            const int a = c / nFZ, z = c - a * nFZ;
            L.row[bkt * BLK + a * NFZP + z] = g[bkt * nA + P->m_of[c]] * P->fz_scale[z];
        }
    }
    for (int q = lane; q < n * 4; q += 64) facc[q] = 0.f;
    load_sorted_angular_neighbors<PERIODIC>(P, pos, b, species, nbr + (size_t)i * cap, n, i, L.rec, L.rec2, L.rec_j,
                                            L.scratch);

    float frc[NFRP], frs[NFRP], fre[NFRP], zz[NFZP], zc[NFZP], zs[NFZP];
#pragma unroll
    for (int a = 0; a < NFRP; a++) {
        frc[a] = a < nFR ? P->fr_c[a] : 0.f;
        frs[a] = a < nFR ? P->fr_rs[a] : 0.f;
        fre[a] = a < nFR ? P->fr_eta[a] : 0.f;
    }
#pragma unroll
    for (int z = 0; z < NFZP; z++) {
        zz[z] = z < nFZ ? P->fz_zeta[z] : 1.f;
        zc[z] = z < nFZ ? P->fz_cos[z] : 0.f;
        zs[z] = z < nFZ ? P->fz_sin[z] : 0.f;
    }

    const int T = (n * (n - 1)) / 2;
    for (int t = lane; t < T; t += 64) {
        int p, q;
        decode_pair(t, n, p, q);
        const AngRec A = L.rec[p], B = L.rec[q];
        const AngRec2 A2 = L.rec2[p], B2 = L.rec2[q];
        const TripleGeom g = triple_geometry<TORCHANI>(A, A2, B, B2, S);

        float R[NFRP], dR[NFRP];
#pragma unroll
        for (int a = 0; a < NFRP; a++) {
            const float sh = g.rbar - frs[a];
            R[a] = fast_exp2(frc[a] * sh * sh);
            dR[a] = -fre[a] * sh * R[a];                   // d/dr_ij of exp(-eta (rbar-Rs)^2): rbar carries 1/2 (ref :306)
        }
        // contract the gradient block with R and dR:  U_z = sum_a G[a][z] R_a,  V_z = sum_a G[a][z] dR_a
        float U[NFZP], V[NFZP];
#pragma unroll
        for (int z = 0; z < NFZP; z++) { U[z] = 0.f; V[z] = 0.f; }
        const float* G = &L.row[g.bucket * BLK];
#pragma unroll
        for (int a = 0; a < NFRP; a++) {
#pragma unroll
            for (int z = 0; z < NFZP; z++) {
                const float gv = G[a * NFZP + z];
                U[z] += gv * R[a];
                V[z] += gv * dR[a];
            }
        }
        float S0 = 0.f, Sr = 0.f, Sth = 0.f;
#pragma unroll
        for (int z = 0; z < NFZP; z++) {
            const float cz = g.c * zc[z] + g.s * zs[z];    // cos(theta - ths)
            const float sz = g.s * zc[z] - g.c * zs[z];    // sin(theta - ths)
            const float x = fmaxf(1.0f + cz, 1e-30f);      // keeps 0 * -inf out of the zeta == 1 corner
            const float lg = fast_log2(x);
            const float Z = fast_exp2(zz[z] * lg);                         // (1+cos)^zeta
            const float dZ = -zz[z] * fast_exp2((zz[z] - 1.0f) * lg) * sz;  // d/dtheta            ref :337
            S0 += U[z] * Z;
            Sr += V[z] * Z;
            Sth += U[z] * dZ;
        }
        // three routes of the chain rule (ref :311-348), already summed over the functions m
        const float t1 = A2.dfc * B2.fc * S0 + g.fcfc * Sr;   // through r_ij
        const float t2 = A2.fc * B2.dfc * S0 + g.fcfc * Sr;   // through r_ik
        const float t3 = g.fcfc * Sth;                        // through theta
        // angle gradients (ref :410-433): dtheta/d(dot') = -damp / sin(theta)
        const float dot = A.dx * B.dx + A.dy * B.dy + A.dz * B.dz;
        const float iprod = A2.rinv * B2.rinv;
        const float damp = TORCHANI ? 0.95f : 1.0f;
        const float dadd = -damp * fast_rcp(g.s) * iprod * t3;
        const float ka = dot * A2.rinv * A2.rinv, kb = dot * B2.rinv * B2.rinv;
        const float s1 = t1 * A2.rinv, s2 = t2 * B2.rinv;
        const float fjx = s1 * A.dx + dadd * (B.dx - ka * A.dx);
        const float fjy = s1 * A.dy + dadd * (B.dy - ka * A.dy);
        const float fjz = s1 * A.dz + dadd * (B.dz - ka * A.dz);
        const float fkx = s2 * B.dx + dadd * (A.dx - kb * B.dx);
        const float fky = s2 * B.dy + dadd * (A.dy - kb * B.dy);
        const float fkz = s2 * B.dz + dadd * (A.dz - kb * B.dz);
        atomicAdd(&facc[p * 4 + 0], fjx); atomicAdd(&facc[p * 4 + 1], fjy); atomicAdd(&facc[p * 4 + 2], fjz);
        atomicAdd(&facc[q * 4 + 0], fkx); atomicAdd(&facc[q * 4 + 1], fky); atomicAdd(&facc[q * 4 + 2], fkz);
    }
    __syncthreads();
    // scatter: +F on each leg atom, -(sum) on the centre
    float cx = 0.f, cy = 0.f, cz = 0.f;
    for (int e = lane; e < n; e += 64) {
        const float fx = facc[e * 4], fy = facc[e * 4 + 1], fz = facc[e * 4 + 2];
        const int j = L.rec_j[e];
        atomicAdd(&pos_grad[3 * j], fx); atomicAdd(&pos_grad[3 * j + 1], fy); atomicAdd(&pos_grad[3 * j + 2], fz);
        cx -= fx; cy -= fy; cz -= fz;
    }
    cx = wave_sum(cx); cy = wave_sum(cy); cz = wave_sum(cz);
    if (lane == 0) {
        atomicAdd(&pos_grad[3 * i], cx); atomicAdd(&pos_grad[3 * i + 1], cy); atomicAdd(&pos_grad[3 * i + 2], cz);
    }
}

}  // namespace nnpops
