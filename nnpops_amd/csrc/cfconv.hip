// cfconv.hip -- SchNet continuous-filter convolution on gfx950 (C ABI: nnpops_cfconv_*).
//
// What is computed (reference src/schnet/CpuCFConv.cpp, maths in SURVEY.md App. A): for every pair
// (i, j) with r_ij < c
//     gamma_g = exp(-((r - mu_g)/sigma)^2 / 2),  mu_g = g*c/(G-1)                           ref :121-122, :151-154
//     y1 = act(W1 gamma + b1)      act = log((e^x + 1)/2)  or  tanh                          ref :158-166
//     y2 = fc(r) * (W2 y1 + b2)    fc = (cos(pi r/c) + 1)/2                                  ref :170-176
//     out[i] += y2 . x[j] ,  out[j] += y2 . x[i]           (elementwise over the W filters)  ref :180-183
// and the analytic derivatives with respect to x and to the positions                        ref :190-299
//
// How it is laid out for CDNA4 (new design; the reference CUDA code runs one warp per half pair and
// scatters 2W float atomics per pair):
//   * the neighbour build keeps a FULL list (rows of {dx, dy, dz, j} per atom, from the shared cell grid of
//     celllist.h or an all-pairs scan for small systems) and, behind it, a slot per PAIR that both ends know
//     (scan_half / half_slots).
//   * widths that are a multiple of 16 (16, 32, ... 128: every SchNet in use) evaluate the filter network ONCE
//     per pair on the MATRIX CORES -- 16 pair slots x W filters per tile, v_mfma_f32_16x16x4_f32 (exact fp32),
//     weights resident in LDS, 8 waves per CU (cfconv_filters_mfma; cfconv_filters_h2 runs the W x W layer as
//     split-fp16 products, three v_mfma_f32_16x16x32_f16 in place of eight fp32 ones) -- and spill the filter row; an
//     OWNER-COMPUTES gather (cfconv_gather: one wave per atom, lanes = channels) then accumulates out[i]
//     (backward: dE/dx[i], dE/dpos[i]) over the atom's full row.  No atomics, no scatter, deterministic.
//   * the same matrix-core code over the full rows -- every pair evaluated from both ends, nothing spilled --
//     is kept behind $NNPOPS_CFCONV_HALF=0 (cfconv_forward_mfma / cfconv_backward_mfma).
//   * other widths use the vector kernel: one wave per atom, lane = filter channel(s), pairs processed 8 at a
//     time so every weight read from LDS feeds 8 (x2 channels) FMAs; weights streamed through the caches
//     when they do not fit in LDS (W > 128).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "celllist.h"
#include "host_common.h"

using namespace nnpops;

namespace {

constexpr int kPairTile = 8;
constexpr int kMaxWavesPerBlock = 8;
constexpr int kMaxWidth = 512;      // <= 128: weights resident in LDS (matrix-core or vector kernels); above: streamed
constexpr int kMaxGauss = 256;
enum { kStOverflow = 0, kStMaxRow = 1, kStPairs = 2, kStUnmatched = 3, kStWordsN = 4 };

// ---------------------------------------------------------------------------------------------
// neighbour rows (full list): one wave per atom
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void append(float4* __restrict__ row, int* __restrict__ row_ids, int cap, bool keep, float dx, float dy,
                                       float dz, int j, int& n) {
    const unsigned long long m = __ballot(keep);
    if (keep) {
        const int slot = n + prefix_popc(m);
        if (slot < cap) {
            row[slot] = make_float4(dx, dy, dz, __int_as_float(j));
            row_ids[slot] = j;                              // the ids alone, 4 bytes apart: what half_slots searches
        }
    }
    n += __popcll(m);
}

// One lane per row: the row's length and how many of its pairs are with a higher index -- scan_half turns those into
// the row's range of pair slots (see "Half list" below).
__device__ __forceinline__ void publish_row(int i, int n, int n_lo, int* __restrict__ cnt, int* __restrict__ lo_cnt) {
    cnt[i] = n;
    lo_cnt[i] = n_lo;
}

template <bool PERIODIC>
__global__ __launch_bounds__(64) void rows_allpairs(int N, const float* __restrict__ pos, const float* __restrict__ box,
                                                    float cutoff2, float4* __restrict__ rows, int* __restrict__ ids, int cap,
                                                    int* __restrict__ cnt, int* __restrict__ lo_cnt) {
    const int i = blockIdx.x, lane = lane_id();
    Box b{};
    if (PERIODIC) b = load_box(box);
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    float4* row = rows + (size_t)i * cap;
    int n = 0, n_lo = 0;
    for (int base = 0; base < N; base += 64) {
        const int j = base + lane;
        bool keep = false;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (j < N && j != i) {
            dx = pos[3 * j] - xi; dy = pos[3 * j + 1] - yi; dz = pos[3 * j + 2] - zi;
            min_image<PERIODIC>(dx, dy, dz, b);
            keep = dx * dx + dy * dy + dz * dz < cutoff2;          // strict, on r^2 (ref :110)
        }
        append(row, ids + (size_t)i * cap, cap, keep, dx, dy, dz, j, n);
        n_lo += __popcll(__ballot(keep && j > i));
    }
    if (lane == 0) publish_row(i, n, n_lo, cnt, lo_cnt);
}

template <bool PERIODIC>
__global__ __launch_bounds__(64) void rows_cells(const float* __restrict__ box, float cutoff2,
                                                 const CellGrid* __restrict__ grid, const int* __restrict__ cell_start,
                                                 const int* __restrict__ sorted_cell, const float4* __restrict__ sorted_pos,
                                                 float4* __restrict__ rows, int* __restrict__ ids, int cap, int* __restrict__ cnt,
                                                 int* __restrict__ lo_cnt, int* __restrict__ status,
                                                 int* __restrict__ cell_hist) {
    const int lane = lane_id();
    clear_cell_histogram(cell_hist);
    const CellGrid g = *grid;
    if (!g.ok) {
        if (lane == 0) {
            if (blockIdx.x == 0) atomicOr(&status[kStOverflow], g.bin_overflow ? 6 : 2);   // 4: grow the cell bins
            publish_row((int)blockIdx.x, 0, 0, cnt, lo_cnt);
        }
        return;
    }
    Box b{};
    if (PERIODIC) b = load_box(box);
    const float4 me = sorted_pos[blockIdx.x];
    const int i = __float_as_int(me.w) & kIdMask;
    const int c = sorted_cell[blockIdx.x];                  // (in sorted order, next to the position: no load that waits for the id)
    const int cx = c % g.nx, cy = (c / g.nx) % g.ny, cz = c / (g.nx * g.ny);
    float4* row = rows + (size_t)i * cap;
    int n = 0, n_lo = 0;
    // the 27-cell stencil as one flat candidate space (celllist.h): full iterations, FOUR batches of candidates requested at a
    // time -- this wave's time is the sum of its dependent round trips to memory (~380 candidates: six batches, one after the
    // other with one load ahead, were five trips; now two)
    const Stencil st = gather_stencil(g, cell_start, cx, cy, cz);
    const int last = max(st.total - 1, 0);
    constexpr int GROUP = 4;
    for (int base = 0; base < st.total; base += 64 * GROUP) {
        float4 pj[GROUP];
#pragma unroll
        for (int q = 0; q < GROUP; q++) pj[q] = sorted_pos[stencil_slot(st, min(base + 64 * q + lane, last))];   // (every lane: ds_bpermute inside)
#pragma unroll
        for (int q = 0; q < GROUP; q++) {
            if (base + 64 * q >= st.total) break;               // wave-uniform
            const int k = base + 64 * q + lane;
            const float4 cur = pj[q];
            bool keep = false;
            int j = -1;
            float dx = 0.f, dy = 0.f, dz = 0.f;
            if (k < st.total) {
                j = __float_as_int(cur.w) & kIdMask;
                if (j != i) {
                    dx = cur.x - me.x; dy = cur.y - me.y; dz = cur.z - me.z;
                    min_image<PERIODIC>(dx, dy, dz, b);
                    keep = dx * dx + dy * dy + dz * dz < cutoff2;
                }
            }
            append(row, ids + (size_t)i * cap, cap, keep, dx, dy, dz, j, n);
            n_lo += __popcll(__ballot(keep && j > i));
        }
    }
    if (lane == 0) publish_row(i, n, n_lo, cnt, lo_cnt);
}

// ---------------------------------------------------------------------------------------------
// Half list behind the full rows.  The matrix-core kernels evaluate the filter network ONCE per pair {i, j}: every
// pair gets a slot (`pid`) -- the pairs a row holds with a higher index, in row order -- and every entry
// of the full rows learns the slot of its pair, so that the owner-computes gather (cfconv_gather) can fetch the
// filter row from either end.  half_off[i] = first slot of row i (scan_half); the entry (i -> j), j < i finds its slot by looking i up in row j.  An entry whose mirror image is missing (a pair within rounding of the cutoff, where the
// cosine cutoff makes its contribution vanish) points at the all-zero row `pair_cap`.
// ---------------------------------------------------------------------------------------------
// Exclusive scan of lo_cnt -> half_off, total in half_off[N].  One workgroup: every thread takes kScanPerThread
// consecutive rows (independent 16-byte loads), one block-wide scan of the thread totals per 16 K rows.
constexpr int kScanPerThread = 16;
__global__ __launch_bounds__(1024) void scan_half(int N, const int* __restrict__ lo_cnt, int* __restrict__ half_off) {
    __shared__ int wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int base = 0; base < N; base += 1024 * kScanPerThread) {
        const int k0 = base + threadIdx.x * kScanPerThread;
        int v[kScanPerThread], mine = 0;
        if (k0 + kScanPerThread <= N) {
#pragma unroll
            for (int q = 0; q < kScanPerThread; q += 4) {
                const int4 t = *reinterpret_cast<const int4*>(lo_cnt + k0 + q);
                v[q] = t.x; v[q + 1] = t.y; v[q + 2] = t.z; v[q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < kScanPerThread; q++) v[q] = k0 + q < N ? lo_cnt[k0 + q] : 0;
        }
#pragma unroll
        for (int q = 0; q < kScanPerThread; q++) mine += v[q];
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const int t = wsum[w];
            before += w < wave ? t : 0;
            total += t;
        }
        int run = carry + before + incl - mine;
#pragma unroll
        for (int q = 0; q < kScanPerThread; q++) {
            const int mine_q = run;
            run += v[q];
            v[q] = mine_q;
        }
        if (k0 + kScanPerThread <= N) {
#pragma unroll
            for (int q = 0; q < kScanPerThread; q += 4)
                *reinterpret_cast<int4*>(half_off + k0 + q) = make_int4(v[q], v[q + 1], v[q + 2], v[q + 3]);
        } else {
#pragma unroll
            for (int q = 0; q < kScanPerThread; q++)
                if (k0 + q < N) half_off[k0 + q] = v[q];
        }
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) half_off[N] = carry;
}

// (the reverse lookup of a pair is described where it happens, below)
__global__ __launch_bounds__(64) void half_slots(const float4* __restrict__ rows, const int* __restrict__ row_ids,
                                                 const int* __restrict__ cnt, const int* __restrict__ half_off, int cap,
                                                 int pair_cap, int* __restrict__ pid, float* __restrict__ half_r,
                                                 int2* __restrict__ half_ij, int* __restrict__ status, int N,
                                                 const float4* __restrict__ sorted_pos) {
    const int lane = lane_id();
    const int k0 = __builtin_amdgcn_readfirstlane(xcd_contiguous_wave_id());      // atoms in cell order when there is one: the rows looked up are L2-hot (wave-uniform: scalar loads)
    if (k0 >= N) return;
    const int i = sorted_pos ? __float_as_int(sorted_pos[k0].w) & kIdMask : k0;
    if (i >= N) return;
    const int n = min(cnt[i], cap);
    const int first = half_off[i];
    int lo_before = 0;
    for (int s0 = 0; s0 < n; s0 += 64) {
        const int s = s0 + lane;
        int j = i;
        bool lower = false;
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < n) {
            rec = rows[(size_t)i * cap + s];
            j = __float_as_int(rec.w) & kIdMask;
            lower = j > i;
        }
        const unsigned long long lm = __ballot(lower);
        int my_pid = pair_cap, unmatched = 0;
        if (lower) {
            int p = first + lo_before + prefix_popc(lm);
            if (p >= pair_cap) p = pair_cap;                 // (only after a row overflow: check() grows and rebuilds)
            my_pid = p;
            if (p < pair_cap) {
                half_r[p] = sqrtf(rec.x * rec.x + rec.y * rec.y + rec.z * rec.z);
                half_ij[p] = make_int2(i, j);
            }
        }
        lo_before += __popcll(lm);
        // entries towards a lower index j: slot = first slot of row j + rank of i among j's higher-index neighbours.  Every such lane
        // looks its own pair up: the whole id row of j (64 ids per pass) is requested at once, sixteen 16-byte loads per lane with
        // nothing between them, so all lookups of the atom cost ONE round trip to memory -- taking the rows eight at a time with the
        // wave scanning each one together (rounds 1-3) was four dependent round trips for the ~26 lower neighbours of an atom, and
        // this kernel is nothing but its chain of dependent first-touch loads (33 us for 10 000 atoms against 10 us without lookups).
        if (s < n && !lower) {
            const int nj = min(cnt[j], cap), fj = half_off[j];
            const int4* rid = reinterpret_cast<const int4*>(row_ids + (size_t)j * cap);      // (cap is a multiple of 4)
            const int pieces = cap >> 2;
            bool hit = false;
            int rank = 0;
            for (int t0 = 0; t0 < nj && !hit; t0 += 64) {
                int4 v[16];
#pragma unroll
                for (int q = 0; q < 16; q++) v[q] = rid[min((t0 >> 2) + q, pieces - 1)];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int id4[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const bool open = !hit && t0 + 4 * q + c < nj;          // (before the hit, inside the row)
                        hit = hit || (open && id4[c] == i);
                        rank += (open && id4[c] != i && id4[c] > j) ? 1 : 0;
                    }
                }
            }
            my_pid = hit ? min(fj + rank, pair_cap) : pair_cap;
            unmatched += hit ? 0 : 1;
        }
        if (s < n) pid[(size_t)i * cap + s] = my_pid;
        if (unmatched) atomicAdd(&status[kStUnmatched], 1);
    }
}

// max row length and number of half pairs (j > i) of the last build
__global__ __launch_bounds__(256) void row_stats(int N, const int* __restrict__ cnt, const float4* __restrict__ rows, int cap,
                                                 int* __restrict__ status) {
    __shared__ int red[2][256];
    int mrow = 0, half = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        const int n = cnt[i];
        mrow = max(mrow, n);
        const int m = min(n, cap);
        for (int e = 0; e < m; e++) half += (__float_as_int(rows[(size_t)i * cap + e].w) & kIdMask) > i;
    }
    red[0][threadIdx.x] = mrow;
    red[1][threadIdx.x] = half;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] = max(red[0][threadIdx.x], red[0][threadIdx.x + off]);
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMax(&status[kStMaxRow], red[0][0]);
        atomicAdd(&status[kStPairs], red[1][0]);
        if (red[0][0] > cap) atomicOr(&status[kStOverflow], 1);
    }
}

// ---------------------------------------------------------------------------------------------
// the convolution
// ---------------------------------------------------------------------------------------------
struct ConvParams {
    int N, W, G;
    float cutoff, sigma_inv;
    int activation;          // 0 shifted softplus, 1 tanh
    int skip_filter_store;   // backward, half-list path: the filter rows of this list are still in `filt` from the forward call
    // Split-fp16 kernels, backward: dY1 = dS1 act'(S1) is multiplied by this power of two before it is split into fp16 planes and dS2 by
    // its inverse (round 6).  Where an activation saturates (tanh) most entries of dY1 are orders of magnitude below the largest, and an
    // fp16 plane does not hold what lies below 2^-14 of ITS scale at full precision: unscaled, the forces under tanh sat 2e-5 ... 7e-5 of
    // the largest force from the oracle (the fp32 matrix kernel: 3e-6); scaled to the top of the fp16 range (the host's bound on
    // |dY1|, nnpops_cfconv_create) they sit at 3e-6.  1 where the split kernels are not in use.
    float dy_scale, dy_unscale;
};

template <int ACT>
__device__ __forceinline__ float activate(float s) {
    if (ACT == 0) return logf(0.5f * expf(s) + 0.5f);      // ref :163
    return tanhf(s);
}
// activation and its derivative in one go
template <int ACT>
__device__ __forceinline__ void activate_d(float s, float& y, float& dy) {
    if (ACT == 0) {
        const float e = expf(s);
        y = logf(0.5f * e + 0.5f);
        dy = e / (e + 1.0f);                               // ref :254-257
    } else {
        const float th = tanhf(s);
        y = th;
        dy = 1.0f - th * th;                               // ref :259-262
    }
}

// LDS: W2^T [W][W] | W1^T [G][W] | per wave: gam[G][8], y1[W][8], pair scalars [8][8] (+ dgam, dy1 backward)
__host__ __device__ inline size_t conv_weight_floats(int W, int G) {
    return ((size_t)W * W + (size_t)G * W + 3) & ~(size_t)3;          // keeps the per-wave slices 16-byte aligned
}
__host__ __device__ inline size_t conv_wave_floats(int W, int G, bool backward) {
    return (size_t)(backward ? 2 : 1) * ((size_t)G * kPairTile + (size_t)W * kPairTile) + 64;
}

// CPL = channels per lane (1: W <= 64, 2: W <= 128).  BACKWARD adds the d/dr path and the two gradients.
// WLDS = false: the weights do not fit in LDS next to one wave's tiles (W > 128): they are read through the
// caches instead.  Same arithmetic, a functional path for unusually wide layers, not a tuned one.
template <int ACT, int CPL, bool BACKWARD, bool WLDS = true>
__global__ __launch_bounds__(64 * kMaxWavesPerBlock) void cfconv_kernel(
    ConvParams P, const float* __restrict__ w1t, const float* __restrict__ b1, const float* __restrict__ w2t,
    const float* __restrict__ b2, const float4* __restrict__ rows, const int* __restrict__ cnt, int cap,
    const float* __restrict__ x, const float* __restrict__ gout,   // gout: upstream gradient (backward only)
    float* __restrict__ out,                                       // forward: output ; backward: input gradient
    float* __restrict__ pos_grad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int W = P.W, G = P.G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* s_w2t = WLDS ? lds : w2t;                // [W][W]   s_w2t[b*W + a] = w2[a][b]
    const float* s_w1t = WLDS ? lds + (size_t)W * W : w1t;   // [G][W]   s_w1t[g*W + a] = w1[a][g]
    const int waves_per_block = blockDim.x >> 6;
    float* wv = lds + (WLDS ? conv_weight_floats(W, G) : 0) + (size_t)wave * conv_wave_floats(W, G, BACKWARD);
    float* ps = wv;                                       // [8][8] per-pair scalars: r, fc, dfc, j, 1/r, dx, dy, dz
    float* gam = ps + 64;                                 // [G][8]
    float* y1b = gam + (size_t)G * kPairTile;             // [W][8]
    float* dgam = y1b + (size_t)W * kPairTile;            // [G][8]  (backward only)
    float* dy1b = dgam + (size_t)G * kPairTile;           // [W][8]  (backward only)

    if (WLDS) {
        for (int q = tid; q < W * W; q += blockDim.x) lds[q] = w2t[q];
        for (int q = tid; q < G * W; q += blockDim.x) lds[(size_t)W * W + q] = w1t[q];
        __syncthreads();
    }

    int ch[CPL];
    bool live[CPL];
    float bias1[CPL], bias2[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        ch[c] = lane + 64 * c;
        live[c] = ch[c] < W;
        if (!live[c]) ch[c] = 0;
        bias1[c] = b1[ch[c]];
        bias2[c] = b2[ch[c]];
    }
    const float mu_step = P.cutoff / (float)(G - 1);       // ref :121-122

    for (int i = blockIdx.x * waves_per_block + wave; i < P.N; i += gridDim.x * waves_per_block) {
        const int n = min(cnt[i], cap);
        const float4* row = rows + (size_t)i * cap;
        float acc[CPL];
        float xi[CPL], gi[CPL];
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            acc[c] = 0.f;
            xi[c] = BACKWARD ? x[(size_t)i * W + ch[c]] : 0.f;
            gi[c] = BACKWARD ? gout[(size_t)i * W + ch[c]] : 0.f;
        }
        float fx = 0.f, fy = 0.f, fz = 0.f;
        for (int t0 = 0; t0 < n; t0 += kPairTile) {
            const int np = min(kPairTile, n - t0);
            // ---- per-pair scalars (lanes 0..7) ----
            float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < np) rec = row[t0 + lane];
            if (lane < kPairTile) {
                const float r = lane < np ? sqrtf(rec.x * rec.x + rec.y * rec.y + rec.z * rec.z) : 1.0f;
                float sn, cs;
                sincospif(r / P.cutoff, &sn, &cs);
                ps[0 * 8 + lane] = r;
                ps[1 * 8 + lane] = lane < np ? 0.5f * cs + 0.5f : 0.f;                       // fc   (ref :301-303)
                ps[2 * 8 + lane] = lane < np ? -(0.5f * kPi / P.cutoff) * sn : 0.f;           // dfc  (ref :305-307)
                ps[3 * 8 + lane] = __int_as_float(lane < np ? (__float_as_int(rec.w) & kIdMask) : i);
                ps[4 * 8 + lane] = 1.0f / r;
                ps[5 * 8 + lane] = rec.x; ps[6 * 8 + lane] = rec.y; ps[7 * 8 + lane] = rec.z;
            }
            __builtin_amdgcn_wave_barrier();          // per-wave LDS slice: LDS ops of one wave execute in order
            // ---- Gaussians for the 8 pairs ----
            for (int q = lane; q < G * kPairTile; q += 64) {
                const int g = q >> 3, p = q & 7;
                const float xg = (ps[p] - (float)g * mu_step) * P.sigma_inv;
                const float gm = expf(-0.5f * xg * xg);                                       // ref :152-153
                gam[q] = gm;
                if (BACKWARD) dgam[q] = -xg * gm * P.sigma_inv;                                // ref :242
            }
            __builtin_amdgcn_wave_barrier();
            // ---- dense 1 + activation ----
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                float s[kPairTile], ds[kPairTile];
#pragma unroll
                for (int p = 0; p < kPairTile; p++) { s[p] = bias1[c]; ds[p] = 0.f; }
                for (int g = 0; g < G; g++) {
                    const float w = s_w1t[g * W + ch[c]];
                    const float4 ga = *reinterpret_cast<const float4*>(gam + g * 8), gb = *reinterpret_cast<const float4*>(gam + g * 8 + 4);
                    s[0] += ga.x * w; s[1] += ga.y * w; s[2] += ga.z * w; s[3] += ga.w * w;
                    s[4] += gb.x * w; s[5] += gb.y * w; s[6] += gb.z * w; s[7] += gb.w * w;
                    if (BACKWARD) {
                        const float4 da = *reinterpret_cast<const float4*>(dgam + g * 8), db = *reinterpret_cast<const float4*>(dgam + g * 8 + 4);
                        ds[0] += da.x * w; ds[1] += da.y * w; ds[2] += da.z * w; ds[3] += da.w * w;
                        ds[4] += db.x * w; ds[5] += db.y * w; ds[6] += db.z * w; ds[7] += db.w * w;
                    }
                }
                float yv[kPairTile], dyv[kPairTile];
#pragma unroll
                for (int p = 0; p < kPairTile; p++) {
                    if (BACKWARD) {
                        float dact;
                        activate_d<ACT>(s[p], yv[p], dact);
                        dyv[p] = ds[p] * dact;
                    } else {
                        yv[p] = activate<ACT>(s[p]);
                    }
                }
                if (live[c]) {
                    *reinterpret_cast<float4*>(y1b + ch[c] * 8) = make_float4(yv[0], yv[1], yv[2], yv[3]);
                    *reinterpret_cast<float4*>(y1b + ch[c] * 8 + 4) = make_float4(yv[4], yv[5], yv[6], yv[7]);
                    if (BACKWARD) {
                        *reinterpret_cast<float4*>(dy1b + ch[c] * 8) = make_float4(dyv[0], dyv[1], dyv[2], dyv[3]);
                        *reinterpret_cast<float4*>(dy1b + ch[c] * 8 + 4) = make_float4(dyv[4], dyv[5], dyv[6], dyv[7]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- dense 2, cutoff, accumulate ----
            float scale_p[kPairTile];
#pragma unroll
            for (int p = 0; p < kPairTile; p++) scale_p[p] = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                float s[kPairTile], ds[kPairTile];
#pragma unroll
                for (int p = 0; p < kPairTile; p++) { s[p] = bias2[c]; ds[p] = 0.f; }
                for (int b = 0; b < W; b++) {
                    const float w = s_w2t[b * W + ch[c]];
                    const float4 ya = *reinterpret_cast<const float4*>(y1b + b * 8), yb = *reinterpret_cast<const float4*>(y1b + b * 8 + 4);
                    s[0] += ya.x * w; s[1] += ya.y * w; s[2] += ya.z * w; s[3] += ya.w * w;
                    s[4] += yb.x * w; s[5] += yb.y * w; s[6] += yb.z * w; s[7] += yb.w * w;
                    if (BACKWARD) {
                        const float4 da = *reinterpret_cast<const float4*>(dy1b + b * 8), db = *reinterpret_cast<const float4*>(dy1b + b * 8 + 4);
                        ds[0] += da.x * w; ds[1] += da.y * w; ds[2] += da.z * w; ds[3] += da.w * w;
                        ds[4] += db.x * w; ds[5] += db.y * w; ds[6] += db.z * w; ds[7] += db.w * w;
                    }
                }
#pragma unroll
                for (int p = 0; p < kPairTile; p++) {
                    const float fc = ps[1 * 8 + p];
                    const int j = __float_as_int(ps[3 * 8 + p]);
                    const float y2 = fc * s[p];                                               // ref :175 / :275
                    if (!BACKWARD) {
                        const float xj = live[c] ? x[(size_t)j * W + ch[c]] : 0.f;
                        acc[c] += y2 * xj;                                                    // ref :181
                    } else {
                        const float gj = live[c] ? gout[(size_t)j * W + ch[c]] : 0.f;
                        const float xj = live[c] ? x[(size_t)j * W + ch[c]] : 0.f;
                        acc[c] += y2 * gj;                                                    // ref :284
                        const float dy2 = ps[2 * 8 + p] * s[p] + fc * ds[p];                   // ref :276
                        scale_p[p] += live[c] ? dy2 * (xj * gi[c] + xi[c] * gj) : 0.f;         // ref :286
                    }
                }
            }
            if (BACKWARD) {
#pragma unroll
                for (int p = 0; p < kPairTile; p++) {
                    const float sc = wave_sum(scale_p[p]) * ps[4 * 8 + p];                    // * 1/r
                    // position_deriv[i] -= sc * delta  (owner side of ref :287-291; delta = pos_j - pos_i)
                    fx -= sc * ps[5 * 8 + p]; fy -= sc * ps[6 * 8 + p]; fz -= sc * ps[7 * 8 + p];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int c = 0; c < CPL; c++)
            if (live[c]) out[(size_t)i * W + ch[c]] = acc[c];
        if (BACKWARD && lane == 0) {
            pos_grad[3 * i] = fx; pos_grad[3 * i + 1] = fy; pos_grad[3 * i + 2] = fz;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA forward: the two dense layers of the filter network as 16x16x4 fp32 matrix-core tiles.
//
// `v_mfma_f32_16x16x4_f32` is exact fp32 (an fmaf chain) at the fp32 matrix rate.  A tile is 16 pairs of
// ONE atom (owner computes, as above) x all W = 16*NCB filters; lane l owns pair/row `l & 15` as the A
// operand and column `l & 15` of each 16-wide column block as the B operand and result, k = 4*step + (l >> 4):
//   layer 1  A = Gaussians (computed in registers, straight in operand layout: no redundancy),
//            B = W1^T from LDS, C initialised with b1;  activation on the accumulators -> Y1 tile in LDS
//   layer 2  A = Y1 tile read back transposed (row stride W+1: conflict-free), B = W2^T from LDS, C = b2
//   output   acc[row][col] * fc[row] * x[j_row][col] summed over the tile's rows in registers.
// Result layout of the instruction: D[row = 4*(l >> 4) + reg][col = l & 15].
// ---------------------------------------------------------------------------------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;

// One dense layer of a 16-row tile on the matrix core: acc[cb] += A(16 x 4*ksteps) * B(4*ksteps x 16) for the NCB
// column blocks.  a_lane / b_lane point at this lane's operands of K step 0; a K step advances A by 4 floats and B
// by 4 rows of W floats.  The operands of step s + 1 are requested BEFORE the MFMAs of step s are issued (explicit
// double buffer): left to itself the compiler reuses one register pair for B and waits for LDS every two MFMAs.
template <int NCB, int W>
__device__ __forceinline__ void mfma_layer(const float* a_lane, const float* b_lane, int ksteps, f32x4 (&acc)[NCB]) {
    float a_cur = a_lane[0], b_cur[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) b_cur[cb] = b_lane[cb * 16];
#pragma unroll 2
    for (int s = 0; s < ksteps; s++) {
        const int nx = min(s + 1, ksteps - 1);                 // (the last step re-reads itself: no branch in the loop)
        const float a_nxt = a_lane[4 * nx];
        float b_nxt[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) b_nxt[cb] = b_lane[(size_t)4 * nx * W + cb * 16];
        __builtin_amdgcn_sched_barrier(0);                     // the requests above stay above the MFMAs below
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, b_cur[cb], acc[cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a_cur = a_nxt;
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) b_cur[cb] = b_nxt[cb];
    }
}
template <int ACT> __device__ __forceinline__ float activate_fast(float s);

__host__ __device__ inline size_t mfma_weight_floats(int W, int G) { return (size_t)W * W + (size_t)((G + 3) & ~3) * W; }
__host__ __device__ inline size_t mfma_wave_floats(int W) { return (size_t)16 * (W + 1) + 64; }

template <int ACT, int NCB>
__global__ __launch_bounds__(64 * kMaxWavesPerBlock) void cfconv_forward_mfma(
    ConvParams P, const float* __restrict__ w1t, const float* __restrict__ b1, const float* __restrict__ w2t,
    const float* __restrict__ b2, const float4* __restrict__ rows, const int* __restrict__ cnt, int cap,
    const float* __restrict__ x, float* __restrict__ out) {
    constexpr int W = NCB * 16, YS = W + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int G = P.G, Gp = (G + 3) & ~3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves_per_block = blockDim.x >> 6;
    float* s_w2t = lds;                                    // [W][W]
    float* s_w1t = s_w2t + (size_t)W * W;                  // [Gp][W], rows >= G are zero
    float* y1 = s_w1t + (size_t)Gp * W + (size_t)wave * mfma_wave_floats(W);   // [16][YS]
    float* ps = y1 + 16 * YS;                              // r[16] | fc[16] | j[16] (bit pattern)
    for (int q = tid; q < W * W; q += blockDim.x) s_w2t[q] = w2t[q];
    for (int q = tid; q < Gp * W; q += blockDim.x) s_w1t[q] = q < G * W ? w1t[q] : 0.f;
    __syncthreads();

    const int col = lane & 15, grp = lane >> 4;
    float b1v[NCB], b2v[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) { b1v[cb] = b1[cb * 16 + col]; b2v[cb] = b2[cb * 16 + col]; }
    const float mu_step = P.cutoff / (float)(G - 1);
    const float gscale = -0.5f * kLog2e * P.sigma_inv * P.sigma_inv;       // exp(-x^2/2) = exp2(gscale * (r - mu)^2)

    // Every wave owns a CONTIGUOUS run of atoms and walks the concatenation of their neighbour rows in tiles of
    // 16: a tile may straddle atoms (rows carry their owner), so only the last tile of a wave is ragged -- with one
    // tile sequence per atom, 52 +- 7 neighbours wasted 19 % of every matrix-core cycle on padding rows.
    const int total_waves = gridDim.x * waves_per_block;
    const int chunk = (P.N + total_waves - 1) / total_waves;
    const int a0 = min((blockIdx.x * waves_per_block + wave) * chunk, P.N), a1 = min(a0 + chunk, P.N);
    auto flush = [&](int owner, float (&sum)[NCB]) {        // fold the four row groups, write the owner's output row
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            float v = sum[cb];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (grp == 0) out[(size_t)owner * W + cb * 16 + col] = v;
        }
    };
    float oacc[NCB];                                        // partial output row of atom `carry`
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) oacc[cb] = 0.f;
    for (int a = a0; a < a1; a++)                           // atoms without neighbours never own a tile row
        if (min(cnt[a], cap) == 0) flush(a, oacc);
    int carry = -1;
    // Locating the 16 rows of a tile (a walk over the atoms' counts) and fetching their records is a chain of
    // dependent global loads: it is done for the NEXT tile while the matrix cores work on the current one.
    auto request = [&](int start_atom, int start_row, int& atom, int& e, float4& rec) {
        atom = start_atom;
        e = start_row + (lane & 15);                        // lanes 0..15 (the others mirror them)
        while (atom < a1) {
            const int n = min(cnt[atom], cap);
            if (e < n) break;
            e -= n;
            atom++;
        }
        rec = atom < a1 ? rows[(size_t)atom * cap + e] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    int my_atom, my_e;
    float4 rec;
    request(a0, 0, my_atom, my_e, rec);
    while (__shfl(my_atom, 0, 64) < a1) {                   // row 0 of the tile exists (wave-uniform)
        if (lane < 16) {
            float r = 1.0f, fc = 0.f;
            int j = a0, owner = -1;
            if (my_atom < a1) {
                r = sqrtf(rec.x * rec.x + rec.y * rec.y + rec.z * rec.z);
                fc = 0.5f * cospif(r / P.cutoff) + 0.5f;                            // ref :301-303
                j = __float_as_int(rec.w) & kIdMask;
                owner = my_atom;
            }
            ps[lane] = r; ps[16 + lane] = fc; ps[32 + lane] = __int_as_float(j); ps[48 + lane] = __int_as_float(owner);
        }
        // the next tile starts one past row 15 (lane 15 knows): request it now
        int next_atom, next_e;
        float4 next_rec;
        request(__shfl(my_atom, 15, 64), __shfl(my_e, 15, 64) + 1, next_atom, next_e, next_rec);
        wave_fence();
        // inputs of my four result rows, requested now so that the two GEMMs hide the latency
        float xv[NCB][4], fcq[4];
        int own[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int rr = grp * 4 + q;
            const int j = __float_as_int(ps[32 + rr]);
            fcq[q] = ps[16 + rr];
            own[q] = __float_as_int(ps[48 + rr]);
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) xv[cb][q] = x[(size_t)j * W + cb * 16 + col];
        }
        const int o_lo = __float_as_int(ps[48]);            // row 0 always exists
        int o_hi = o_lo;
#pragma unroll
        for (int r15 = 1; r15 < 16; r15++) o_hi = max(o_hi, __float_as_int(ps[48 + r15]));      // owners ascend; -1 = padding
        // ---- layer 1 ----
        f32x4 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[cb] = f32x4{b1v[cb], b1v[cb], b1v[cb], b1v[cb]};
        const float rp = ps[col];
        for (int s = 0; s < Gp / 4; s++) {
            const int g = 4 * s + grp;
            const float d = rp - (float)g * mu_step;
            const float a = g < G ? fast_exp2(gscale * d * d) : 0.f;                   // ref :151-154
            const float* wrow = s_w1t + g * W + col;
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wrow[cb * 16], acc[cb], 0, 0, 0);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; cb++)
#pragma unroll
            for (int q = 0; q < 4; q++) y1[(grp * 4 + q) * YS + cb * 16 + col] = activate_fast<ACT>(acc[cb][q]);
        wave_fence();
        // ---- layer 2 ----
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[cb] = f32x4{b2v[cb], b2v[cb], b2v[cb], b2v[cb]};
        mfma_layer<NCB, W>(y1 + col * YS + grp, s_w2t + grp * W + col, W / 4, acc);
        // ---- output: rows are summed into their owner (ref :175, :181); padding rows have fc = 0 ----
        if (carry >= 0 && carry != o_lo) {                  // the previous tile ended exactly on an atom boundary
            flush(carry, oacc);
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) oacc[cb] = 0.f;
        }
        if (o_lo == o_hi) {
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
#pragma unroll
                for (int q = 0; q < 4; q++) oacc[cb] += fcq[q] * acc[cb][q] * xv[cb][q];
        } else {
            for (int o = o_lo; o <= o_hi; o++) {            // wave-uniform; atoms in between without rows get zeros (again)
                float part[NCB];
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) {
                    part[cb] = o == o_lo ? oacc[cb] : 0.f;
#pragma unroll
                    for (int q = 0; q < 4; q++) part[cb] += own[q] == o ? fcq[q] * acc[cb][q] * xv[cb][q] : 0.f;
                }
                if (o < o_hi) {
                    flush(o, part);
                } else {
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++) oacc[cb] = part[cb];
                }
            }
        }
        carry = o_hi;
        my_atom = next_atom; my_e = next_e; rec = next_rec;
        wave_fence();
    }
    if (carry >= 0) flush(carry, oacc);
}

// Activations of the matrix-core kernels: single-instruction transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32,
// ~1 ulp) -- the 2 x 16 x W activations of a tile would otherwise cost as much issue time as its MFMAs.  For the
// shifted softplus the host has folded log2(e) into W1, b1 and ln(2) into W2 (nnpops_cfconv_create), so with
// s' = log2(e) s coming out of layer 1:   y' = log2((2^s' + 1) / 2) = y / ln 2,   dy/ds = 2^s' / (2^s' + 1),
// and layer 2 computes (ln2 W2) y' = W2 y; the d/dr path is scaled the same way (dS1' dy/ds = log2(e) dS1 dy/ds).
template <int ACT>
__device__ __forceinline__ float activate_fast(float s) {
    if (ACT == 0) return fast_log2(0.5f * fast_exp2(s) + 0.5f);                                   // ref :163
    return tanhf(s);
}
template <int ACT>
__device__ __forceinline__ void activate_d_fast(float s, float& y, float& dy) {
    if (ACT == 0) {
        const float e = fast_exp2(s);
        y = fast_log2(0.5f * e + 0.5f);
        dy = e * fast_rcp(e + 1.0f);                                                              // ref :254-257
    } else {
        const float th = tanhf(s);
        y = th;
        dy = 1.0f - th * th;                                                                      // ref :259-262
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA backward: same packed tiling as the forward kernel, with the d/dr path riding along.
//   layer 1   S1 = Gam W1^T + b1 and dS1 = dGam W1^T share every B operand (two MFMAs per LDS read)
//   Y1 = act(S1) goes to the LDS tile; dY1 = dS1 * act'(S1) waits in registers
//   layer 2   S2 = Y1 W2^T + b2, then the SAME LDS tile is refilled with dY1 for dS2 = dY1 W2^T
//             (one tile per wave instead of two keeps 7 waves per CU resident at W = 128)
//   epilogue  y2 = fc S2 -> input gradient ; dy2 = dfc S2 + fc dS2 -> force on the owner atom   ref :275-291
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline size_t mfma_wave_floats_bwd(int W) { return (size_t)16 * (W + 1) + 144; }

template <int ACT, int NCB>
__global__ __launch_bounds__(64 * kMaxWavesPerBlock) void cfconv_backward_mfma(
    ConvParams P, const float* __restrict__ w1t, const float* __restrict__ b1, const float* __restrict__ w2t,
    const float* __restrict__ b2, const float4* __restrict__ rows, const int* __restrict__ cnt, int cap,
    const float* __restrict__ x, const float* __restrict__ gout, float* __restrict__ xgrad, float* __restrict__ pos_grad) {
    constexpr int W = NCB * 16, YS = W + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int G = P.G, Gp = (G + 3) & ~3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves_per_block = blockDim.x >> 6;
    float* s_w2t = lds;
    float* s_w1t = s_w2t + (size_t)W * W;
    float* y1 = s_w1t + (size_t)Gp * W + (size_t)wave * mfma_wave_floats_bwd(W);   // [16][YS]
    float* ps = y1 + 16 * YS;                              // r | fc | dfc | j | 1/r | dx | dy | dz | owner, 16 each
    for (int q = tid; q < W * W; q += blockDim.x) s_w2t[q] = w2t[q];
    for (int q = tid; q < Gp * W; q += blockDim.x) s_w1t[q] = q < G * W ? w1t[q] : 0.f;
    __syncthreads();

    const int col = lane & 15, grp = lane >> 4;
    float b1v[NCB], b2v[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) { b1v[cb] = b1[cb * 16 + col]; b2v[cb] = b2[cb * 16 + col]; }
    const float mu_step = P.cutoff / (float)(G - 1);
    const float sig2 = P.sigma_inv * P.sigma_inv;
    const float gscale = -0.5f * kLog2e * sig2;

    // Tiles are packed across the atoms of a wave's contiguous run exactly as in the forward kernel (rows carry
    // their owner): per-atom tile sequences left 19 % of the rows of every matrix-core pass empty.
    const int total_waves = gridDim.x * waves_per_block;
    const int chunk = (P.N + total_waves - 1) / total_waves;
    const int a0 = min((blockIdx.x * waves_per_block + wave) * chunk, P.N), a1 = min(a0 + chunk, P.N);
    float xi[NCB], gi[NCB], gacc[NCB];                      // state of the atom whose rows are being consumed (`cur`)
    float fx = 0.f, fy = 0.f, fz = 0.f;
    int cur = -1;
    auto flush = [&]() {                                    // fold the four row groups, write cur's two results
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            float v = gacc[cb];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (grp == 0) xgrad[(size_t)cur * W + cb * 16 + col] = v;
        }
        fx += __shfl_xor(fx, 16, 64); fx += __shfl_xor(fx, 32, 64);
        fy += __shfl_xor(fy, 16, 64); fy += __shfl_xor(fy, 32, 64);
        fz += __shfl_xor(fz, 16, 64); fz += __shfl_xor(fz, 32, 64);
        if (lane == 0) { pos_grad[3 * cur] = fx; pos_grad[3 * cur + 1] = fy; pos_grad[3 * cur + 2] = fz; }
    };
    auto open = [&](int o) {
        cur = o;
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            xi[cb] = x[(size_t)o * W + cb * 16 + col];
            gi[cb] = gout[(size_t)o * W + cb * 16 + col];
            gacc[cb] = 0.f;
        }
        fx = fy = fz = 0.f;
    };
    for (int a = a0; a < a1; a++)                           // atoms without neighbours never own a tile row
        if (min(cnt[a], cap) == 0) {
            if (grp == 0)
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) xgrad[(size_t)a * W + cb * 16 + col] = 0.f;
            if (lane == 0) { pos_grad[3 * a] = 0.f; pos_grad[3 * a + 1] = 0.f; pos_grad[3 * a + 2] = 0.f; }
        }
    auto request = [&](int start_atom, int start_row, int& atom, int& e, float4& rec) {
        atom = start_atom;
        e = start_row + (lane & 15);                        // lanes 0..15 (the others mirror them)
        while (atom < a1) {
            const int n = min(cnt[atom], cap);
            if (e < n) break;
            e -= n;
            atom++;
        }
        rec = atom < a1 ? rows[(size_t)atom * cap + e] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    int my_atom, my_e;
    float4 rec;
    request(a0, 0, my_atom, my_e, rec);
    while (__shfl(my_atom, 0, 64) < a1) {                   // row 0 of the tile exists (wave-uniform)
        if (lane < 16) {
            float r = 1.0f, fc = 0.f, dfc = 0.f;
            int j = a0, owner = -1;
            if (my_atom < a1) {
                r = sqrtf(rec.x * rec.x + rec.y * rec.y + rec.z * rec.z);
                float sn, cs;
                sincospif(r / P.cutoff, &sn, &cs);
                fc = 0.5f * cs + 0.5f;                                              // ref :301-303
                dfc = -(0.5f * kPi / P.cutoff) * sn;                                // ref :305-307
                j = __float_as_int(rec.w) & kIdMask;
                owner = my_atom;
            }
            ps[lane] = r; ps[16 + lane] = fc; ps[32 + lane] = dfc;
            ps[48 + lane] = __int_as_float(j);
            ps[64 + lane] = 1.0f / r; ps[80 + lane] = rec.x; ps[96 + lane] = rec.y; ps[112 + lane] = rec.z;
            ps[128 + lane] = __int_as_float(owner);
        }
        // the next tile starts one past row 15 (lane 15 knows): its rows are requested now, used after the GEMMs
        int next_atom, next_e;
        float4 next_rec;
        request(__shfl(my_atom, 15, 64), __shfl(my_e, 15, 64) + 1, next_atom, next_e, next_rec);
        wave_fence();
        const int o_lo = __float_as_int(ps[128]);           // row 0 always exists
        int o_hi = o_lo;
#pragma unroll
        for (int r15 = 1; r15 < 16; r15++) o_hi = max(o_hi, __float_as_int(ps[128 + r15]));     // owners ascend; -1 = padding
        // ---- layer 1: value and d/dr together ----
        f32x4 acc[NCB], dacc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            acc[cb] = f32x4{b1v[cb], b1v[cb], b1v[cb], b1v[cb]};
            dacc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float rp = ps[col];
        for (int s = 0; s < Gp / 4; s++) {
            const int g = 4 * s + grp;
            const float d = rp - (float)g * mu_step;
            const float a = g < G ? fast_exp2(gscale * d * d) : 0.f;                   // ref :151-154
            const float da = -d * sig2 * a;                                            // ref :242
            const float* wrow = s_w1t + g * W + col;
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                const float b = wrow[cb * 16];
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[cb], 0, 0, 0);
                dacc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(da, b, dacc[cb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int cb = 0; cb < NCB; cb++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float yv, dact;
                activate_d_fast<ACT>(acc[cb][q], yv, dact);
                y1[(grp * 4 + q) * YS + cb * 16 + col] = yv;
                dacc[cb][q] *= dact;                                                   // dY1, kept in registers
            }
        wave_fence();
        // ---- layer 2 on Y1 ----
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[cb] = f32x4{b2v[cb], b2v[cb], b2v[cb], b2v[cb]};
        mfma_layer<NCB, W>(y1 + col * YS + grp, s_w2t + grp * W + col, W / 4, acc);
        wave_fence();
        // ---- refill the tile with dY1, layer 2 again ----
#pragma unroll
        for (int cb = 0; cb < NCB; cb++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                y1[(grp * 4 + q) * YS + cb * 16 + col] = dacc[cb][q];
                dacc[cb][q] = 0.f;
            }
        wave_fence();
        mfma_layer<NCB, W>(y1 + col * YS + grp, s_w2t + grp * W + col, W / 4, dacc);
        // ---- epilogue: my four rows of the tile, owner by owner (one owner for two tiles out of three) ----
        for (int o = o_lo; o <= o_hi; o++) {                // wave-uniform; atoms in between without rows get zeros (again)
            if (o != cur) {
                if (cur >= 0) flush();
                open(o);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int rr = grp * 4 + q;
                if (__float_as_int(ps[128 + rr]) == o) {    // uniform over the 16 lanes that share the row
                    const int j = __float_as_int(ps[48 + rr]);
                    const float fc = ps[16 + rr], dfc = ps[32 + rr];
                    float sc = 0.f;
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++) {
                        const float xj = x[(size_t)j * W + cb * 16 + col], gj = gout[(size_t)j * W + cb * 16 + col];
                        const float s2 = acc[cb][q];
                        gacc[cb] += fc * s2 * gj;                                              // ref :275, :284
                        const float dy2 = dfc * s2 + fc * dacc[cb][q];                         // ref :276
                        sc += dy2 * (xj * gi[cb] + xi[cb] * gj);                               // ref :286
                    }
                    sc += __shfl_xor(sc, 1, 64); sc += __shfl_xor(sc, 2, 64);
                    sc += __shfl_xor(sc, 4, 64); sc += __shfl_xor(sc, 8, 64);
                    sc *= ps[64 + rr];
                    // position_deriv[owner] -= sc * delta  (owner side of ref :287-291; delta = pos_j - pos_owner)
                    fx -= sc * ps[80 + rr]; fy -= sc * ps[96 + rr]; fz -= sc * ps[112 + rr];
                }
                __builtin_amdgcn_sched_barrier(0);          // one row's 2*NCB gathers in flight at a time: no spills at W = 128
            }
        }
        my_atom = next_atom; my_e = next_e; rec = next_rec;
        wave_fence();
    }
    if (cur >= 0) flush();
}

// ---------------------------------------------------------------------------------------------
// Half-list path (the default for the matrix-core widths): the filter network is evaluated once per PAIR.
//   cfconv_filters_mfma   tile = 16 consecutive pair slots (no owners, no raggedness); writes the filter row
//                         F[pid] = fc * (W2 y1 + b2) -- 512 B at W = 128 -- and, backward, the pair's radial
//                         force  s[pid] = sum_c (dfc S2 + fc dS2)_c (x_j g_i + x_i g_j)_c / r            ref :275-291
//   cfconv_gather         owner computes: out[i] = sum_e F[pid_e] * x[j_e]   (backward: the same sum over gout gives
//                         dE/dx[i], and dE/dpos[i] = -sum_e s[pid_e] delta_e) -- deterministic, no atomics.
// Compared with evaluating every pair from both ends this halves the matrix-core AND the activation work (which
// add up on a SIMD) for one round trip of F through HBM / the Infinity Cache.
// ---------------------------------------------------------------------------------------------
template <int ACT, int NCB, bool BWD>
__global__ __launch_bounds__(64 * kMaxWavesPerBlock) void cfconv_filters_mfma(
    ConvParams P, const float* __restrict__ w1t, const float* __restrict__ b1, const float* __restrict__ w2t,
    const float* __restrict__ b2, const int* __restrict__ half_off, const float* __restrict__ half_r,
    const int2* __restrict__ half_ij, int pair_cap, const float* __restrict__ x, const float* __restrict__ gout,
    float* __restrict__ filt, float* __restrict__ pair_s) {
    constexpr int W = NCB * 16, YS = W + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int G = P.G, Gp = (G + 3) & ~3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves_per_block = blockDim.x >> 6;
    float* s_w2t = lds;
    float* s_w1t = s_w2t + (size_t)W * W;
    float* y1 = s_w1t + (size_t)Gp * W + (size_t)wave * mfma_wave_floats_bwd(W);   // [16][YS]
    float* ps = y1 + 16 * YS;                              // r | fc | dfc | 1/r | i | j, 16 each
    for (int q = tid; q < W * W; q += blockDim.x) s_w2t[q] = w2t[q];
    for (int q = tid; q < Gp * W; q += blockDim.x) s_w1t[q] = q < G * W ? w1t[q] : 0.f;
    __syncthreads();
    if (blockIdx.x == 0) {                                  // the all-zero row behind the last slot (entries without a mirror image)
        for (int q = tid; q < W; q += blockDim.x) filt[(size_t)pair_cap * W + q] = 0.f;
        if (BWD && tid == 0) pair_s[pair_cap] = 0.f;
    }

    const int col = lane & 15, grp = lane >> 4;
    float b1v[NCB], b2v[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) { b1v[cb] = b1[cb * 16 + col]; b2v[cb] = b2[cb * 16 + col]; }
    const float mu_step = P.cutoff / (float)(G - 1);
    const float sig2 = P.sigma_inv * P.sigma_inv;
    const float gscale = -0.5f * kLog2e * sig2;

    const int pairs = min(half_off[P.N], pair_cap);
    const int tiles = (pairs + 15) >> 4;
    const int total_waves = gridDim.x * waves_per_block;
    int t = blockIdx.x * waves_per_block + wave;
    auto request = [&](int tile, float& r, int2& ij) {      // lanes 0..15 (the others mirror them)
        const int p = 16 * tile + (lane & 15);
        r = -1.f;
        ij = make_int2(0, 0);
        if (tile < tiles && p < pairs) {
            r = half_r[p];
            if constexpr (BWD) ij = half_ij[p];
        }
    };
    float my_r;
    int2 my_ij;
    request(t, my_r, my_ij);
    for (; t < tiles; t += total_waves) {
        if (lane < 16) {
            float r = 1.0f, fc = 0.f, dfc = 0.f;
            if (my_r >= 0.f) {
                r = my_r;
                if constexpr (BWD) {
                    float sn, cs;
                    sincospif(r / P.cutoff, &sn, &cs);
                    fc = 0.5f * cs + 0.5f;                                              // ref :301-303
                    dfc = -(0.5f * kPi / P.cutoff) * sn;                                // ref :305-307
                } else {
                    fc = 0.5f * cospif(r / P.cutoff) + 0.5f;
                }
            }
            ps[lane] = r; ps[16 + lane] = fc;
            if constexpr (BWD) {
                ps[32 + lane] = dfc; ps[48 + lane] = 1.0f / r;
                ps[64 + lane] = __int_as_float(my_ij.x); ps[80 + lane] = __int_as_float(my_ij.y);
            }
        }
        float next_r;
        int2 next_ij;
        request(t + total_waves, next_r, next_ij);          // used after the GEMMs
        wave_fence();
        // ---- layer 1 (backward: value and d/dr together) ----
        f32x4 acc[NCB], dacc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            acc[cb] = f32x4{b1v[cb], b1v[cb], b1v[cb], b1v[cb]};
            if constexpr (BWD) dacc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float rp = ps[col];
        for (int s = 0; s < Gp / 4; s++) {
            const int g = 4 * s + grp;
            const float d = rp - (float)g * mu_step;
            const float a = g < G ? fast_exp2(gscale * d * d) : 0.f;                   // ref :151-154
            const float da = -d * sig2 * a;                                            // ref :242
            const float* wrow = s_w1t + g * W + col;
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                const float b = wrow[cb * 16];
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[cb], 0, 0, 0);
                if constexpr (BWD) dacc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(da, b, dacc[cb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int cb = 0; cb < NCB; cb++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if constexpr (BWD) {
                    float yv, dact;
                    activate_d_fast<ACT>(acc[cb][q], yv, dact);
                    y1[(grp * 4 + q) * YS + cb * 16 + col] = yv;
                    dacc[cb][q] *= dact;                                               // dY1, kept in registers
                } else {
                    y1[(grp * 4 + q) * YS + cb * 16 + col] = activate_fast<ACT>(acc[cb][q]);
                }
            }
        wave_fence();
        // ---- layer 2 on Y1 ----
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[cb] = f32x4{b2v[cb], b2v[cb], b2v[cb], b2v[cb]};
        mfma_layer<NCB, W>(y1 + col * YS + grp, s_w2t + grp * W + col, W / 4, acc);
        if constexpr (BWD) {
            wave_fence();
            // ---- refill the tile with dY1, layer 2 again ----
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    y1[(grp * 4 + q) * YS + cb * 16 + col] = dacc[cb][q];
                    dacc[cb][q] = 0.f;
                }
            wave_fence();
            mfma_layer<NCB, W>(y1 + col * YS + grp, s_w2t + grp * W + col, W / 4, dacc);
        }
        // ---- my four pairs of the tile ----
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int rr = grp * 4 + q;
            const int p = 16 * t + rr;
            const float fc = ps[16 + rr];
            if (p < pairs) {                                // uniform over the 16 lanes of a row
                float* frow = filt + (size_t)p * W + col;
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) frow[cb * 16] = fc * acc[cb][q];      // ref :175
                if constexpr (BWD) {
                    const float dfc = ps[32 + rr];
                    const int i = __float_as_int(ps[64 + rr]), j = __float_as_int(ps[80 + rr]);
                    float sc = 0.f;
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++) {
                        const size_t c = (size_t)cb * 16 + col;
                        const float xi = x[(size_t)i * W + c], gi = gout[(size_t)i * W + c];
                        const float xj = x[(size_t)j * W + c], gj = gout[(size_t)j * W + c];
                        const float dy2 = dfc * acc[cb][q] + fc * dacc[cb][q];         // ref :276
                        sc += dy2 * (xj * gi + xi * gj);                               // ref :286
                    }
                    sc += __shfl_xor(sc, 1, 64); sc += __shfl_xor(sc, 2, 64);
                    sc += __shfl_xor(sc, 4, 64); sc += __shfl_xor(sc, 8, 64);
                    if (col == 0) pair_s[p] = sc * ps[48 + rr];
                }
            }
            if constexpr (BWD) __builtin_amdgcn_sched_barrier(0);     // one pair's 4*NCB gathers in flight at a time
        }
        my_r = next_r; my_ij = next_ij;
        wave_fence();
    }
}

// ---------------------------------------------------------------------------------------------
// cfconv_filters_h2: the filters kernel with its dense layers on the half-precision matrix instruction, every fp32
// operand split into two fp16 planes so that the result keeps fp32 accuracy:
//     x = hi + 2^-11 lo'   (hi = fp16(x), lo' = fp16((x - hi) 2^11): 22 significant bits, every plane in normal range)
//     A B = Ahi Bhi + 2^-11 (Ahi Blo' + Alo' Bhi) + O(2^-22)          -- three v_mfma_f32_16x16x32_f16 per 16x16x32
//     block (fp32 accumulation, two accumulators) instead of eight v_mfma_f32_16x16x4_f32: 48 instead of 256 issue cycles.
// Measured on a 16 x 128 x 128 tile (tools/ubench/split_f16_gemm.hip): 2.6x faster than the fp32 form including the
// split, and a SMALLER error against a double-precision product (1.6e-6 against 3.8e-6 at |y| ~ 9: exact fp16
// products summed in fp32 versus a chain of 128 rounded fp32 FMAs).  The host only takes this kernel when the weights
// bound every operand below the fp16 range (nnpops_cfconv_create); $NNPOPS_CFCONV_SPLIT=0 keeps the all-fp32 one.
//   layer 1   is computed TRANSPOSED (rows = filters, columns = pairs), so a lane ends up with four consecutive filters
//             of one pair -- after the activation exactly the 8-byte groups the A planes of layer 2 are written in.  b1
//             rides along as one more K index against a constant 1.  L1H (G + 1 <= 64): split products here too, B = the
//             pair's Gaussians computed and split in registers, A = the W1 planes [filter][k]; otherwise
//             v_mfma_f32_16x16x4_f32 with the operands swapped.
//   LDS       W2 planes [f2][k] and the per-wave A planes [pair][k] with the 16-byte slot index XORed by the row
//             (h2_slot: conflict-free ds_read_b128); W1 planes with rows an odd number of slots long (or W1^T fp32).
//             Same footprint as the fp32 kernel.
// ---------------------------------------------------------------------------------------------
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;

__host__ __device__ inline int h2_l1_rows(int G) { return ((G + 1) + 3) & ~3; }            // Gaussians + the bias row, padded to 4
__host__ __device__ inline size_t h2_weight_bytes(int W, int G) { return (size_t)2 * W * W * 2 + (size_t)h2_l1_rows(G) * W * 4; }
__host__ __device__ inline size_t h2_wave_bytes(int W) { return (size_t)2 * 16 * W * 2 + 96 * 4; }
// layer 1 as split products too (L1H): W1 planes [filter][k], k = Gaussians then the bias, rows an ODD number of 16-byte
// slots long (neighbouring rows then start 4 * odd banks apart: ds_read_b128 across rows is conflict-free without a swizzle)
__host__ __device__ inline int h2_l1_slots(int G) { return (G + 1 + 7) >> 3; }
__host__ __device__ inline int h2_l1_row_bytes(int G) { return (h2_l1_slots(G) | 1) * 16; }
__host__ __device__ inline size_t h2_weight_bytes_l1h(int W, int G) { return (size_t)2 * W * W * 2 + (size_t)2 * W * h2_l1_row_bytes(G); }

// Byte offset of the 16-byte slot `slot` of row `row` in a plane whose rows hold W halves.  ds_read_b128 serves a wave in
// four groups of 16 lanes that mix two K groups (lanes {0-3, 12-15} of one with {4-11} of the next): with the slot index
// XORed by the row, the two halves of such a group land in different quarters of the 256-byte bank row for any K step,
// so the 16 lanes hit 16 different slots (a rotation by the row, the first attempt, left every group 2-way conflicted:
// SQ_LDS_BANK_CONFLICT 47 % of the LDS cycles).  Row lengths that are not a power of two keep the rotation.
template <int W>
__device__ __forceinline__ int h2_slot(int row, int slot) {
    constexpr int kSlots = W / 8;
    if constexpr ((kSlots & (kSlots - 1)) == 0) return row * (2 * W) + ((slot ^ row) & (kSlots - 1)) * 16;
    return row * (2 * W) + ((slot + row) % kSlots) * 16;
}

__device__ __forceinline__ _Float16 split_lo(float v, _Float16 hi) { return (_Float16)((v - (float)hi) * kLoScale); }

// D = A B for one 16-row tile against all NCB column blocks: acc1 += Ahi Bhi, acc2 += Ahi Blo' + Alo' Bhi
// FRESH1 / FRESH2: the accumulator starts from zero -- passed as the (inline constant) C operand of its first MFMA instead
// of being cleared register by register beforehand.
template <int NCB, int W, bool TIGHT, bool FRESH1, bool FRESH2>   // TIGHT (backward): one K step's plane reads in flight, not two steps'
__device__ __forceinline__ void h2_layer(const char* a_h, const char* a_l, const char* b_h, const char* b_l, int row, int grp, int col,
                                         f32x4 (&acc1)[NCB], f32x4 (&acc2)[NCB]) {
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    auto step = [&](int s, auto first) {
        constexpr bool kFirst = decltype(first)::value;
        const int slot = 4 * s + grp;                       // this lane's 8 consecutive k of the step
        const f16x8 ah = *reinterpret_cast<const f16x8*>(a_h + h2_slot<W>(row, slot));
        const f16x8 al = *reinterpret_cast<const f16x8*>(a_l + h2_slot<W>(row, slot));
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            const int f2 = cb * 16 + col;
            const f16x8 bh = *reinterpret_cast<const f16x8*>(b_h + h2_slot<W>(f2, slot));
            const f16x8 bl = *reinterpret_cast<const f16x8*>(b_l + h2_slot<W>(f2, slot));
            acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, kFirst && FRESH1 ? zero : acc1[cb], 0, 0, 0);
            acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, kFirst && FRESH2 ? zero : acc2[cb], 0, 0, 0);
            acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc2[cb], 0, 0, 0);
        }
    };
    if constexpr (TIGHT) {                                  // (the caller has cleared / preset the accumulators)
        static_assert(!FRESH1 && !FRESH2, "the loop form does not peel its first step");
#pragma unroll 1
        for (int s = 0; s < W / 32; s++) step(s, std::false_type{});
    } else {
        step(0, std::true_type{});
#pragma unroll
        for (int s = 1; s < W / 32; s++) step(s, std::false_type{});
    }
}

template <int ACT, int NCB, bool BWD, bool L1H>
__global__ __launch_bounds__(64 * kMaxWavesPerBlock) void cfconv_filters_h2(
    ConvParams P, const float* __restrict__ w1b, const _Float16* __restrict__ w1h, const _Float16* __restrict__ w1l,
    const _Float16* __restrict__ w2h, const _Float16* __restrict__ w2l, const float* __restrict__ b2, const int* __restrict__ half_off, const float* __restrict__ half_r,
    const int2* __restrict__ half_ij, int pair_cap, const float* __restrict__ x, const float* __restrict__ gout,
    float* __restrict__ filt, float* __restrict__ pair_s) {
    constexpr int W = NCB * 16;
    static_assert(W % 32 == 0, "the K steps of layer 2 are 32 wide");
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const int G = P.G, Gq = h2_l1_rows(G);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves_per_block = blockDim.x >> 6;
    char* s_w2h = ldsb;                                      // [W][W] halves, slots rotated
    char* s_w2l = s_w2h + (size_t)W * W * 2;
    float* s_w1t = reinterpret_cast<float*>(s_w2l + (size_t)W * W * 2);     // [Gq][W]: rows < G = W1^T, row G = b1, rest 0
    const int row1 = h2_l1_row_bytes(G), slots1 = h2_l1_slots(G);           // L1H: W1 planes instead, [W][row1 bytes]
    char* s_w1h = reinterpret_cast<char*>(s_w1t);
    char* s_w1l = s_w1h + (size_t)W * row1;
    char* a_h = (L1H ? s_w1l + (size_t)W * row1 : reinterpret_cast<char*>(s_w1t + (size_t)Gq * W)) + (size_t)wave * h2_wave_bytes(W);
    char* a_l = a_h + 16 * W * 2;
    float* ps = reinterpret_cast<float*>(a_l + 16 * W * 2);  // r | fc | dfc | 1/r | i | j, 16 each
    for (int q = tid; q < W * (W / 8); q += blockDim.x) {    // 16-byte slots of the W2 planes
        const int f2 = q / (W / 8), slot = q % (W / 8);
        *reinterpret_cast<f16x8*>(s_w2h + h2_slot<W>(f2, slot)) = *reinterpret_cast<const f16x8*>(w2h + (size_t)f2 * W + slot * 8);
        *reinterpret_cast<f16x8*>(s_w2l + h2_slot<W>(f2, slot)) = *reinterpret_cast<const f16x8*>(w2l + (size_t)f2 * W + slot * 8);
    }
    if constexpr (L1H) {
        for (int q = tid; q < W * (row1 / 16); q += blockDim.x) {
            reinterpret_cast<f16x8*>(s_w1h)[q] = reinterpret_cast<const f16x8*>(w1h)[q];
            reinterpret_cast<f16x8*>(s_w1l)[q] = reinterpret_cast<const f16x8*>(w1l)[q];
        }
    } else {
        for (int q = tid; q < Gq * W; q += blockDim.x) s_w1t[q] = w1b[q];
    }
    __syncthreads();
    if (blockIdx.x == 0) {                                  // the all-zero row behind the last slot (entries without a mirror image)
        for (int q = tid; q < W; q += blockDim.x) filt[(size_t)pair_cap * W + q] = 0.f;
        if (BWD && tid == 0) pair_s[pair_cap] = 0.f;
    }

    const int col = lane & 15, grp = lane >> 4;
    float b2v[NCB];                                         // (backward: re-read per tile, the registers are needed)
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) b2v[cb] = b2[cb * 16 + col];
    const float mu_step = P.cutoff / (float)(G - 1);
    const float sig2 = P.sigma_inv * P.sigma_inv;
    const float gscale = -0.5f * kLog2e * sig2;

    const int pairs = min(half_off[P.N], pair_cap);
    const int tiles = (pairs + 15) >> 4;
    const int total_waves = gridDim.x * waves_per_block;
    int t = blockIdx.x * waves_per_block + wave;
    auto request = [&](int tile, float& r, int2& ij) {      // lanes 0..15 (the others mirror them)
        const int p = 16 * tile + (lane & 15);
        r = -1.f;
        ij = make_int2(0, 0);
        if (tile < tiles && p < pairs) {
            r = half_r[p];
            if constexpr (BWD) ij = half_ij[p];
        }
    };
    float my_r;
    int2 my_ij;
    request(t, my_r, my_ij);
    for (; t < tiles; t += total_waves) {
        if (lane < 16) {
            float r = 1.0f, fc = 0.f, dfc = 0.f;
            if (my_r >= 0.f) {
                r = my_r;
                if constexpr (BWD) {
                    float sn, cs;
                    sincospif(r / P.cutoff, &sn, &cs);
                    fc = 0.5f * cs + 0.5f;                                              // ref :301-303
                    dfc = -(0.5f * kPi / P.cutoff) * sn;                                // ref :305-307
                } else {
                    fc = 0.5f * cospif(r / P.cutoff) + 0.5f;
                }
            }
            ps[lane] = r; ps[16 + lane] = fc;
            if constexpr (BWD) {
                ps[32 + lane] = dfc; ps[48 + lane] = 1.0f / r;
                ps[64 + lane] = __int_as_float(my_ij.x); ps[80 + lane] = __int_as_float(my_ij.y);
            }
        }
        float next_r;
        int2 next_ij;
        request(t + total_waves, next_r, next_ij);          // used after the GEMMs
        wave_fence();
        // ---- layer 1, transposed: acc[cb][q] = S1 of filter 16 cb + 4 grp + q for the pair `col` ----
        const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 acc[NCB], dacc[NCB];
        if constexpr (!L1H) {
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (BWD) dacc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const float rp = ps[col];
        if constexpr (L1H) {
            // split products here too: B = the Gaussians of pair `col` (and their d/dr), eight consecutive g per lane and
            // step, split in registers; A = the W1 planes.  G + 1 <= 64: at most two K steps.
            // (backward: one pass for the values, one for d/dr -- four accumulator sets at once do not fit the registers;
            //  the second pass recomputes the Gaussians rather than keep their planes)
            const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
            auto l1_step = [&](auto deriv, auto first, int s, f32x4 (&hi)[NCB], f32x4 (&lo)[NCB]) {
                constexpr bool kFirst = decltype(first)::value;
                {
                    f16x8 gh, gl;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int g = 32 * s + 8 * grp + i;
                        const float d = rp - (float)g * mu_step;
                        float v = g < G ? fast_exp2(gscale * d * d) : 0.f;             // ref :151-154
                        if (decltype(deriv)::value) v *= -d * sig2;                    // ref :242
                        if (g == G) v = decltype(deriv)::value ? 0.f : 1.0f;           // the bias column of the planes
                        gh[i] = (_Float16)v;
                        gl[i] = split_lo(v, gh[i]);
                    }
                    const int slot = min(4 * s + grp, slots1 - 1);  // a slot past the row meets all-zero Gaussians
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++) {
                        const int off = (cb * 16 + col) * row1 + slot * 16;
                        const f16x8 wh = *reinterpret_cast<const f16x8*>(s_w1h + off);
                        const f16x8 wl = *reinterpret_cast<const f16x8*>(s_w1l + off);
                        hi[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, gh, kFirst ? zero : hi[cb], 0, 0, 0);
                        lo[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, gl, kFirst ? zero : lo[cb], 0, 0, 0);
                        lo[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, gh, lo[cb], 0, 0, 0);
                        if (BWD && (cb & 3) == 3) __builtin_amdgcn_sched_barrier(0);    // (four blocks' plane reads in flight, not eight)
                    }
                }
            };
            // (the accumulators start as the zero C operand of the first step; backward: one pass for the values, one for d/dr)
            auto l1_pass = [&](auto deriv, f32x4 (&hi)[NCB]) {
                f32x4 lo[NCB];
                if constexpr (BWD) {                        // (a loop, cleared accumulators: the registers do not allow both steps inline)
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++) { hi[cb] = zero; lo[cb] = zero; }
#pragma unroll 1
                    for (int s = 0; s < 2; s++) {
                        if (32 * s >= G + 1) break;         // (wave-uniform: fewer than 32 Gaussians)
                        l1_step(deriv, std::false_type{}, s, hi, lo);
                    }
                } else {
                    l1_step(deriv, std::true_type{}, 0, hi, lo);
                    if (32 < G + 1) l1_step(deriv, std::false_type{}, 1, hi, lo);      // (wave-uniform: more than 31 Gaussians)
                }
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) hi[cb] += kLoInv * lo[cb];
            };
            l1_pass(std::false_type{}, acc);
            if constexpr (BWD) {
                __builtin_amdgcn_sched_barrier(0);
                l1_pass(std::true_type{}, dacc);
            }
        } else {
            for (int s = 0; s < Gq / 4; s++) {
                const int g = 4 * s + grp;
                const float d = rp - (float)g * mu_step;
                float a = g < G ? fast_exp2(gscale * d * d) : 0.f;                     // ref :151-154
                float da = -d * sig2 * a;                                              // ref :242
                if (g == G) { a = 1.0f; da = 0.f; }                                    // the bias row
                const float* wrow = s_w1t + g * W + col;
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) {
                    const float w = wrow[cb * 16];
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, a, acc[cb], 0, 0, 0);
                    if constexpr (BWD) dacc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, da, dacc[cb], 0, 0, 0);
                }
            }
        }
        // ---- activation, split, A planes: pair `col`, filters 16 cb + 4 grp .. + 3 = half a 16-byte slot ----
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            f16x4 h, l;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float yv;
                if constexpr (BWD) {
                    float dact;
                    activate_d_fast<ACT>(acc[cb][q], yv, dact);
                    dacc[cb][q] *= dact;                                               // dY1, kept in registers
                } else {
                    yv = activate_fast<ACT>(acc[cb][q]);
                }
                h[q] = (_Float16)yv;
                l[q] = split_lo(yv, h[q]);
            }
            const int off = h2_slot<W>(col, 2 * cb + (grp >> 1)) + (grp & 1) * 8;
            *reinterpret_cast<f16x4*>(a_h + off) = h;
            *reinterpret_cast<f16x4*>(a_l + off) = l;
        }
        wave_fence();
        // ---- layer 2 on Y1: S2[pair 4 grp + q][filter 16 cb + col] ----
        f32x4 acc2[NCB];
        if constexpr (BWD) {
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                const float bias = b2[cb * 16 + col];
                acc[cb] = f32x4{bias, bias, bias, bias};
                acc2[cb] = zero4;
            }
            h2_layer<NCB, W, true, false, false>(a_h, a_l, s_w2h, s_w2l, col, grp, col, acc, acc2);
        } else {                                            // (zero C operands; the bias joins in the epilogue)
            h2_layer<NCB, W, false, true, true>(a_h, a_l, s_w2h, s_w2l, col, grp, col, acc, acc2);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[cb] += kLoInv * acc2[cb];
        if constexpr (BWD) {
            wave_fence();
            // ---- refill the planes with dY1, layer 2 again ----
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                f16x4 h, l;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float dv = dacc[cb][q] * P.dy_scale;      // (ConvParams::dy_scale)
                    h[q] = (_Float16)dv;
                    l[q] = split_lo(dv, h[q]);
                }
                const int off = h2_slot<W>(col, 2 * cb + (grp >> 1)) + (grp & 1) * 8;
                *reinterpret_cast<f16x4*>(a_h + off) = h;
                *reinterpret_cast<f16x4*>(a_l + off) = l;
                dacc[cb] = zero4;
                acc2[cb] = zero4;
            }
            wave_fence();
            h2_layer<NCB, W, true, false, false>(a_h, a_l, s_w2h, s_w2l, col, grp, col, dacc, acc2);
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) dacc[cb] = (dacc[cb] + kLoInv * acc2[cb]) * P.dy_unscale;
        }
        // ---- my four pairs of the tile: the filter rows first (backward: and dy2 = dfc S2 + fc dS2 in place of dS2, after
        //      which S2 is dead and its registers serve the gathers), then the pair forces ----
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int rr = grp * 4 + q;
            const int p = 16 * t + rr;
            const float fc = ps[16 + rr];
            if (p < pairs && !(BWD && P.skip_filter_store)) {      // uniform over the 16 lanes of a row
                float* frow = filt + (size_t)p * W + col;
#pragma unroll
                for (int cb = 0; cb < NCB; cb++)
                    frow[cb * 16] = BWD ? fc * acc[cb][q] : fc * (acc[cb][q] + b2v[cb]);      // ref :175
            }
            if constexpr (BWD) {
                const float dfc = ps[32 + rr];
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) dacc[cb][q] = dfc * acc[cb][q] + fc * dacc[cb][q];     // ref :276
            }
        }
        if constexpr (BWD) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int rr = grp * 4 + q;
                const int p = 16 * t + rr;
                // (straight-line loads: a padding row of the last tile carries i = j = 0 and is simply not stored;
                //  with a branch around them the compiler would drain the load queue where the paths meet)
                const int i = __float_as_int(ps[64 + rr]), j = __float_as_int(ps[80 + rr]);
                float sc = 0.f;
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) {
                    const size_t c = (size_t)cb * 16 + col;
                    const float xi = x[(size_t)i * W + c], gi = gout[(size_t)i * W + c];
                    const float xj = x[(size_t)j * W + c], gj = gout[(size_t)j * W + c];
                    sc += dacc[cb][q] * (xj * gi + xi * gj);                           // ref :286
                }
                sc += __shfl_xor(sc, 1, 64); sc += __shfl_xor(sc, 2, 64);
                sc += __shfl_xor(sc, 4, 64); sc += __shfl_xor(sc, 8, 64);
                if (col == 0 && p < pairs) pair_s[p] = sc * ps[48 + rr];
                if (q & 1) __builtin_amdgcn_sched_barrier(0);       // two pairs' gathers in flight at a time
            }
        }
        my_r = next_r; my_ij = next_ij;
        wave_fence();
    }
}

// ---------------------------------------------------------------------------------------------
// cfconv_filters_h2x2 (round 6): the FORWARD filters kernel with 32 pairs per wave and layer 2 fed from registers.
//
// What bound cfconv_filters_h2 forward (profiles/r05_cfconv_lds_mfma_counters_pmc.txt): not the matrix pipe (~30 % busy) but the LDS
// pipe (83 % of the busy cycles) -- every 16-pair tile streams BOTH W2 planes (64 KB) and the W1 planes through it, and writes and
// re-reads its own Y1 planes: 104 ds_read_b128 (1 KB each, 128 bytes per clock for the whole CU) for 144 matrix instructions.
// Two changes, both about operand traffic:
//   * a wave takes TWO 16-pair tiles per pass: every W1 / W2 fragment it reads from LDS feeds the matrix instructions of both
//     (weight reads per pair halve);
//   * layer 1 is computed transposed, so a lane ends with filters {16 cb + 4 grp + q} of ITS pair -- which IS a valid A fragment of
//     layer 2's 16 x 16 x 32 instruction if the K index of a step is read as {32 s + 4 grp + i, 32 s + 16 + 4 grp + i}: the order of
//     K inside a step is free as long as both operands agree, so the W2 planes are staged into LDS with that permutation of their
//     8-byte pieces and Y1 never goes through LDS at all (no plane writes, no plane reads, no fences, 8 KB of LDS per wave less).
// 96 ds_read_b128 per 32 pairs (48 per 16: less than half) for 288 matrix instructions.  Same arithmetic as cfconv_filters_h2
// (split-fp16 products, fp32 accumulation, bias behind the products); the rows agree with the 16-pair kernel's to the last bit or two
// (the matrix instruction adds the 32 products of a step in another order), tests/test_cfconv_gpu.py.
// ---------------------------------------------------------------------------------------------
template <int ACT, int NCB, int U, int WAVES>      // U: 16-pair tiles per pass of a wave (1 or 2); WAVES: waves per workgroup
__global__ __launch_bounds__(64 * WAVES) void cfconv_filters_h2x2(
    ConvParams P, const _Float16* __restrict__ w1h, const _Float16* __restrict__ w1l, const _Float16* __restrict__ w2h,
    const _Float16* __restrict__ w2l, const float* __restrict__ b2, const int* __restrict__ half_off, const float* __restrict__ half_r,
    int pair_cap, float* __restrict__ filt) {
    constexpr int W = NCB * 16;
    static_assert(W % 32 == 0, "the K steps of layer 2 are 32 wide");
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const int G = P.G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves_per_block = blockDim.x >> 6;
    char* s_w2h = ldsb;                                      // [W][W] halves, K pieces permuted inside every step, slots swizzled (h2_slot)
    char* s_w2l = s_w2h + (size_t)W * W * 2;
    const int row1 = h2_l1_row_bytes(G), slots1 = h2_l1_slots(G);
    char* s_w1h = s_w2l + (size_t)W * W * 2;                 // W1 planes [W][row1 bytes]
    char* s_w1l = s_w1h + (size_t)W * row1;
    float* ps = reinterpret_cast<float*>(s_w1l + (size_t)W * row1) + wave * 128;     // r | fc of the wave's 16 U pairs
    for (int q = tid; q < W * (W / 8); q += blockDim.x) {    // 16-byte slots of the W2 planes: slot 4 s + grp <- K pieces {32 s + 4 grp, 32 s + 16 + 4 grp}
        const int f2 = q / (W / 8), slot = q % (W / 8);
        const int k0 = 32 * (slot >> 2) + 4 * (slot & 3);
        const f16x4 h0 = *reinterpret_cast<const f16x4*>(w2h + (size_t)f2 * W + k0), h1 = *reinterpret_cast<const f16x4*>(w2h + (size_t)f2 * W + k0 + 16);
        const f16x4 l0 = *reinterpret_cast<const f16x4*>(w2l + (size_t)f2 * W + k0), l1 = *reinterpret_cast<const f16x4*>(w2l + (size_t)f2 * W + k0 + 16);
        *reinterpret_cast<f16x8*>(s_w2h + h2_slot<W>(f2, slot)) = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        *reinterpret_cast<f16x8*>(s_w2l + h2_slot<W>(f2, slot)) = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    }
    for (int q = tid; q < W * (row1 / 16); q += blockDim.x) {
        reinterpret_cast<f16x8*>(s_w1h)[q] = reinterpret_cast<const f16x8*>(w1h)[q];
        reinterpret_cast<f16x8*>(s_w1l)[q] = reinterpret_cast<const f16x8*>(w1l)[q];
    }
    __syncthreads();
    if (blockIdx.x == 0)                                    // the all-zero row behind the last slot (entries without a mirror image)
        for (int q = tid; q < W; q += blockDim.x) filt[(size_t)pair_cap * W + q] = 0.f;

    const int col = lane & 15, grp = lane >> 4;
    const float mu_step = P.cutoff / (float)(G - 1);
    const float gscale = -0.5f * kLog2e * P.sigma_inv * P.sigma_inv;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};

    const int pairs = min(half_off[P.N], pair_cap);
    constexpr int PP = 16 * U;                               // pairs per pass
    const int tiles = (pairs + PP - 1) / PP;
    const int total_waves = gridDim.x * waves_per_block;
    int t = blockIdx.x * waves_per_block + wave;
    auto request = [&](int tile) {                           // lanes 0..31 (the others mirror them)
        const int p = PP * tile + (lane & (PP - 1));
        return (tile < tiles && p < pairs) ? half_r[p] : -1.f;
    };
    float my_r = request(t);
    for (; t < tiles; t += total_waves) {
        if (lane < PP) {
            const float r = my_r >= 0.f ? my_r : 1.0f;
            ps[lane] = r;
            ps[64 + lane] = my_r >= 0.f ? 0.5f * cospif(r / P.cutoff) + 0.5f : 0.f;
        }
        const float next_r = request(t + total_waves);
        wave_fence();
        // ---- layer 1, transposed, both tiles: y[u][cb][q] = S1 of filter 16 cb + 4 grp + q for the pair `col` of tile u ----
        f32x4 y[U][NCB];
        {
            f32x4 lo[U][NCB];
            float rp[U];
#pragma unroll
            for (int u = 0; u < U; u++) rp[u] = ps[16 * u + col];
            auto l1_step = [&](auto first, int s) {
                constexpr bool kFirst = decltype(first)::value;
                f16x8 gh[U], gl[U];
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int g = 32 * s + 8 * grp + i;
                        const float d = rp[u] - (float)g * mu_step;
                        float v = g < G ? fast_exp2(gscale * d * d) : 0.f;             // ref :151-154
                        if (g == G) v = 1.0f;                                          // the bias column of the planes
                        gh[u][i] = (_Float16)v;
                        gl[u][i] = split_lo(v, gh[u][i]);
                    }
                const int slot = min(4 * s + grp, slots1 - 1);      // a slot past the row meets all-zero Gaussians
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) {
                    const int off = (cb * 16 + col) * row1 + slot * 16;
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(s_w1h + off);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(s_w1l + off);
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        y[u][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, gh[u], kFirst ? zero : y[u][cb], 0, 0, 0);
                        lo[u][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, gl[u], kFirst ? zero : lo[u][cb], 0, 0, 0);
                        lo[u][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, gh[u], lo[u][cb], 0, 0, 0);
                    }
                }
            };
            l1_step(std::true_type{}, 0);
            if (32 < G + 1) l1_step(std::false_type{}, 1);  // (wave-uniform: more than 31 Gaussians)
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) y[u][cb] += kLoInv * lo[u][cb];
        }
        // ---- activation and split in registers: the A fragments of layer 2, step s = {filters of cb 2 s, filters of cb 2 s + 1} ----
        f16x8 ah[U][NCB / 2], al[U][NCB / 2];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float yv = activate_fast<ACT>(y[u][cb][q]);
                    const _Float16 hq = (_Float16)yv;
                    ah[u][cb >> 1][(cb & 1) * 4 + q] = hq;
                    al[u][cb >> 1][(cb & 1) * 4 + q] = split_lo(yv, hq);
                }
        // ---- layer 2, transposed as well (the W2 fragment is the A operand, the Y1 fragment -- same lane layout -- the B operand):
        //      S2[u][filter 16 cb + 4 grp + q][pair col]: a lane ends with FOUR CONSECUTIVE filters of its pair, i.e. 16-byte pieces of
        //      the filter row (a store instruction of the wave covers 1 KB where the pair-major layout's 4-byte stores covered 256 bytes);
        //      every W2 fragment serves both tiles ----
        f32x4 acc2[U][NCB];
        auto l2_step = [&](auto first, int s) {
            constexpr bool kFirst = decltype(first)::value;
            const int slot = 4 * s + grp;
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                const int f2 = cb * 16 + col;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(s_w2h + h2_slot<W>(f2, slot));
                const f16x8 bl = *reinterpret_cast<const f16x8*>(s_w2l + h2_slot<W>(f2, slot));
#pragma unroll
                for (int u = 0; u < U; u++) {
                    y[u][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[u][s], kFirst ? zero : y[u][cb], 0, 0, 0);
                    acc2[u][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[u][s], kFirst ? zero : acc2[u][cb], 0, 0, 0);
                    acc2[u][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[u][s], acc2[u][cb], 0, 0, 0);
                }
            }
        };
        l2_step(std::true_type{}, 0);
#pragma unroll
        for (int s = 1; s < W / 32; s++) l2_step(std::false_type{}, s);
        // ---- the filter rows of my two pairs: 16 bytes per column block ----
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int p = PP * t + 16 * u + col;
            const float fc = ps[64 + 16 * u + col];
            if (p < pairs) {
                float* frow = filt + (size_t)p * W + 4 * grp;
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) {
                    const float4 bias = *reinterpret_cast<const float4*>(b2 + cb * 16 + 4 * grp);
                    const f32x4 v = y[u][cb] + kLoInv * acc2[u][cb];
                    *reinterpret_cast<float4*>(frow + cb * 16) = make_float4(fc * (v[0] + bias.x), fc * (v[1] + bias.y), fc * (v[2] + bias.z), fc * (v[3] + bias.w));      // ref :175
                    // (plain stores: `nt` / `sc1` rows were measured -- they leave the filters kernel at 54 - 66 us and cost the gather, which
                    //  then finds fewer of the rows in the caches, 12 us)
                }
            }
        }
        my_r = next_r;
        wave_fence();
    }
}

// ---------------------------------------------------------------------------------------------
// cfconv_filters_h2b (round 6): the BACKWARD filters kernel on the same footing -- no Y1 / dY1 planes in LDS, every weight
// fragment read once per tile.
//
// cfconv_filters_h2<.., BWD> runs four matrix passes per 16-pair tile (layer 1 values, layer 1 d/dr, layer 2 on Y1, layer 2 on dY1),
// each with its own walk over the weight planes in LDS (208 ds_read_b128 per tile), writes and re-reads two sets of A planes, and
// gathers x / gout with 128 four-byte loads per lane.  Here:
//   * layer 1: values and d/dr in ONE pass -- the derivative Gaussians are the value Gaussians times -(r - mu) / sigma^2, no second
//     set of exponentials -- every W1 fragment feeds six matrix instructions;
//   * activation, derivative and the two splits in registers; Y1 and dY1 ARE layer 2's operand fragments (K permuted inside a step,
//     cfconv_filters_h2x2), so nothing goes through LDS and the two layer-2 passes become one: every W2 fragment feeds six
//     matrix instructions -- 96 ds_read_b128 per tile in all;
//   * layer 2 transposed (the W2 fragment is the A operand): a lane ends with S2 and dS2 of FOUR CONSECUTIVE filters of its own pair,
//     so the filter row leaves as 16-byte pieces and x / gout of the pair's two atoms arrive as 16-byte loads (32 per lane instead
//     of 128), the contraction needs two cross-lane steps instead of sixteen.
// Same arithmetic as cfconv_filters_h2 (split-fp16 products, fp32 accumulation, second layer's bias in the accumulator).
// ---------------------------------------------------------------------------------------------
template <int ACT, int NCB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void cfconv_filters_h2b(
    ConvParams P, const _Float16* __restrict__ w1h, const _Float16* __restrict__ w1l, const _Float16* __restrict__ w2h,
    const _Float16* __restrict__ w2l, const float* __restrict__ b2, const int* __restrict__ half_off, const float* __restrict__ half_r,
    const int2* __restrict__ half_ij, int pair_cap, const float* __restrict__ x, const float* __restrict__ gout,
    float* __restrict__ filt, float* __restrict__ pair_s) {
    constexpr int W = NCB * 16;
    static_assert(W % 32 == 0, "the K steps of layer 2 are 32 wide");
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const int G = P.G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves_per_block = blockDim.x >> 6;
    char* s_w2h = ldsb;                                      // [W][W] halves, K pieces permuted inside every step, slots swizzled (h2_slot)
    char* s_w2l = s_w2h + (size_t)W * W * 2;
    const int row1 = h2_l1_row_bytes(G), slots1 = h2_l1_slots(G);
    char* s_w1h = s_w2l + (size_t)W * W * 2;                 // W1 planes [W][row1 bytes]
    char* s_w1l = s_w1h + (size_t)W * row1;
    float* s_b2 = reinterpret_cast<float*>(s_w1l + (size_t)W * row1);      // [W] (read per tile from here: as global loads the compiler hoists all of
                                                                           //      them out of the tile loop and spills them, 32 registers)
    float* ps = s_b2 + W + wave * 96;                        // r | fc | dfc | 1/r | i | j of the wave's 16 pairs
    for (int q = tid; q < W; q += blockDim.x) s_b2[q] = b2[q];
    for (int q = tid; q < W * (W / 8); q += blockDim.x) {    // slot 4 s + grp <- K pieces {32 s + 4 grp, 32 s + 16 + 4 grp}
        const int f2 = q / (W / 8), slot = q % (W / 8);
        const int k0 = 32 * (slot >> 2) + 4 * (slot & 3);
        const f16x4 h0 = *reinterpret_cast<const f16x4*>(w2h + (size_t)f2 * W + k0), h1 = *reinterpret_cast<const f16x4*>(w2h + (size_t)f2 * W + k0 + 16);
        const f16x4 l0 = *reinterpret_cast<const f16x4*>(w2l + (size_t)f2 * W + k0), l1 = *reinterpret_cast<const f16x4*>(w2l + (size_t)f2 * W + k0 + 16);
        *reinterpret_cast<f16x8*>(s_w2h + h2_slot<W>(f2, slot)) = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        *reinterpret_cast<f16x8*>(s_w2l + h2_slot<W>(f2, slot)) = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    }
    for (int q = tid; q < W * (row1 / 16); q += blockDim.x) {
        reinterpret_cast<f16x8*>(s_w1h)[q] = reinterpret_cast<const f16x8*>(w1h)[q];
        reinterpret_cast<f16x8*>(s_w1l)[q] = reinterpret_cast<const f16x8*>(w1l)[q];
    }
    __syncthreads();
    if (blockIdx.x == 0) {                                  // the all-zero row behind the last slot (entries without a mirror image)
        for (int q = tid; q < W; q += blockDim.x) filt[(size_t)pair_cap * W + q] = 0.f;
        if (tid == 0) pair_s[pair_cap] = 0.f;
    }

    const int col = lane & 15, grp = lane >> 4;
    const float mu_step = P.cutoff / (float)(G - 1);
    const float sig2 = P.sigma_inv * P.sigma_inv;
    const float gscale = -0.5f * kLog2e * sig2;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};

    const int pairs = min(half_off[P.N], pair_cap);
    const int tiles = (pairs + 15) >> 4;
    const int total_waves = gridDim.x * waves_per_block;
    int t = blockIdx.x * waves_per_block + wave;
    auto request = [&](int tile, float& r, int2& ij) {      // lanes 0..15 (the others mirror them)
        const int p = 16 * tile + (lane & 15);
        r = -1.f;
        ij = make_int2(0, 0);
        if (tile < tiles && p < pairs) { r = half_r[p]; ij = half_ij[p]; }
    };
    float my_r;
    int2 my_ij;
    request(t, my_r, my_ij);
    for (; t < tiles; t += total_waves) {
        if (lane < 16) {
            float r = 1.0f, fc = 0.f, dfc = 0.f;
            if (my_r >= 0.f) {
                r = my_r;
                float sn, cs;
                sincospif(r / P.cutoff, &sn, &cs);
                fc = 0.5f * cs + 0.5f;                                                  // ref :301-303
                dfc = -(0.5f * kPi / P.cutoff) * sn;                                    // ref :305-307
            }
            ps[lane] = r; ps[16 + lane] = fc; ps[32 + lane] = dfc; ps[48 + lane] = 1.0f / r;
            ps[64 + lane] = __int_as_float(my_ij.x); ps[80 + lane] = __int_as_float(my_ij.y);
        }
        float next_r;
        int2 next_ij;
        request(t + total_waves, next_r, next_ij);
        wave_fence();
        // ---- layer 1, transposed, values and d/dr together: S1 / dS1 of filter 16 cb + 4 grp + q for the pair `col`; then the activation,
        //      its derivative and the two splits: layer 2's operand fragments, step s = {filters of cb 2 s, of cb 2 s + 1}.  The column
        //      blocks in two halves (registers) ----
        constexpr int HB = NCB >= 4 ? NCB / 2 : NCB;         // column blocks per half
        f16x8 ah[NCB / 2], al[NCB / 2], dh[NCB / 2], dl[NCB / 2];
        const float rp = ps[col];
#pragma unroll
        for (int half = 0; half < NCB / HB; half++) {
            f32x4 y[HB], dy[HB], lo[HB], dlo[HB];
            auto l1_step = [&](auto first, int s) {
                constexpr bool kFirst = decltype(first)::value;
                f16x8 gh, gl, dgh, dgl;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int g = 32 * s + 8 * grp + i;
                    const float d = rp - (float)g * mu_step;
                    float v = g < G ? fast_exp2(gscale * d * d) : 0.f;                 // ref :151-154
                    float dv = -d * sig2 * v;                                          // ref :242
                    if (g == G) { v = 1.0f; dv = 0.f; }                                // the bias column of the planes
                    gh[i] = (_Float16)v;   gl[i] = split_lo(v, gh[i]);
                    dgh[i] = (_Float16)dv; dgl[i] = split_lo(dv, dgh[i]);
                }
                const int slot = min(4 * s + grp, slots1 - 1);      // a slot past the row meets all-zero Gaussians
#pragma unroll
                for (int c = 0; c < HB; c++) {
                    const int off = ((half * HB + c) * 16 + col) * row1 + slot * 16;
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(s_w1h + off);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(s_w1l + off);
                    y[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, gh, kFirst ? zero : y[c], 0, 0, 0);
                    dy[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, dgh, kFirst ? zero : dy[c], 0, 0, 0);
                    lo[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, gl, kFirst ? zero : lo[c], 0, 0, 0);
                    dlo[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, dgl, kFirst ? zero : dlo[c], 0, 0, 0);
                    lo[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, gh, lo[c], 0, 0, 0);
                    dlo[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, dgh, dlo[c], 0, 0, 0);
                }
            };
            l1_step(std::true_type{}, 0);
            if (32 < G + 1) l1_step(std::false_type{}, 1);  // (wave-uniform: more than 31 Gaussians)
#pragma unroll
            for (int c = 0; c < HB; c++) {
                const int cb = half * HB + c;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float yv, dact;
                    activate_d_fast<ACT>(y[c][q] + kLoInv * lo[c][q], yv, dact);
                    const float dv = (dy[c][q] + kLoInv * dlo[c][q]) * dact * P.dy_scale;      // dY1, at the top of the fp16 range (ConvParams::dy_scale; the scale LAST:
                                                                                            // as dS1 (dact scale) the same figures come out 10 x worse, tools/cfconv_split_error.py)
                    const _Float16 hq = (_Float16)yv, dq = (_Float16)dv;
                    ah[cb >> 1][(cb & 1) * 4 + q] = hq;
                    al[cb >> 1][(cb & 1) * 4 + q] = split_lo(yv, hq);
                    dh[cb >> 1][(cb & 1) * 4 + q] = dq;
                    dl[cb >> 1][(cb & 1) * 4 + q] = split_lo(dv, dq);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- layer 2 on Y1 and dY1 together, transposed: S2 / dS2 of filter 16 cb + 4 grp + q for the pair `col`; the column blocks in
        //      two halves (four accumulator sets of eight blocks do not fit the registers next to the 64 of the operand fragments), each
        //      followed by its part of the epilogue: the filter row (16-byte pieces), dy2 = dfc S2 + fc dS2, the contraction with
        //      x / gout of the pair's two atoms ----
        const int p = 16 * t + col;
        const float fc = ps[16 + col], dfc = ps[32 + col];
        const int ai = __float_as_int(ps[64 + col]), aj = __float_as_int(ps[80 + col]);      // (a padding row carries i = j = 0 and is not stored)
        const float* xi = x + (size_t)ai * W + 4 * grp;
        const float* xj = x + (size_t)aj * W + 4 * grp;
        const float* gi = gout + (size_t)ai * W + 4 * grp;
        const float* gj = gout + (size_t)aj * W + 4 * grp;
        const bool store = p < pairs && !P.skip_filter_store;
        float sc = 0.f;
#pragma unroll
        for (int half = 0; half < NCB / HB; half++) {
            f32x4 s2[HB], ds2[HB], lo2[HB], dlo2[HB];
            // (x / gout of the pair's two atoms for this half's filters: requested HERE, a matrix pass ahead of their use -- behind the
            //  pass their round trip to the L2 would be exposed once per half with two waves per SIMD to hide it)
            float4 vxi[HB], vgi[HB], vxj[HB], vgj[HB];
#pragma unroll
            for (int c = 0; c < HB; c++) {
                const int off = (half * HB + c) * 16;
                vxi[c] = *reinterpret_cast<const float4*>(xi + off); vgi[c] = *reinterpret_cast<const float4*>(gi + off);
                vxj[c] = *reinterpret_cast<const float4*>(xj + off); vgj[c] = *reinterpret_cast<const float4*>(gj + off);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < HB; c++) {
                const float4 bias = *reinterpret_cast<const float4*>(s_b2 + (half * HB + c) * 16 + 4 * grp);
                s2[c] = f32x4{bias.x, bias.y, bias.z, bias.w};
                ds2[c] = zero; lo2[c] = zero; dlo2[c] = zero;
            }
#pragma unroll
            for (int s = 0; s < W / 32; s++) {
                const int slot = 4 * s + grp;
#pragma unroll
                for (int c = 0; c < HB; c++) {
                    const int f2 = (half * HB + c) * 16 + col;
                    const f16x8 bh = *reinterpret_cast<const f16x8*>(s_w2h + h2_slot<W>(f2, slot));
                    const f16x8 bl = *reinterpret_cast<const f16x8*>(s_w2l + h2_slot<W>(f2, slot));
                    s2[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[s], s2[c], 0, 0, 0);
                    ds2[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, dh[s], ds2[c], 0, 0, 0);
                    lo2[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[s], lo2[c], 0, 0, 0);
                    dlo2[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, dh[s], dlo2[c], 0, 0, 0);
                    lo2[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[s], lo2[c], 0, 0, 0);
                    dlo2[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, dl[s], dlo2[c], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);      // (one step's weight fragments in flight: left alone the scheduler requests all four steps' at once)
            }
#pragma unroll
            for (int c = 0; c < HB; c++) {
                const int off = (half * HB + c) * 16;
                const f32x4 v = s2[c] + kLoInv * lo2[c], dv = (ds2[c] + kLoInv * dlo2[c]) * P.dy_unscale;
                if (store) *reinterpret_cast<float4*>(filt + (size_t)p * W + off + 4 * grp) = make_float4(fc * v[0], fc * v[1], fc * v[2], fc * v[3]);      // ref :175
                const f32x4 d2 = dfc * v + fc * dv;                                    // ref :276
                sc += d2[0] * (vxj[c].x * vgi[c].x + vxi[c].x * vgj[c].x) + d2[1] * (vxj[c].y * vgi[c].y + vxi[c].y * vgj[c].y) +
                      d2[2] * (vxj[c].z * vgi[c].z + vxi[c].z * vgj[c].z) + d2[3] * (vxj[c].w * vgi[c].w + vxi[c].w * vgj[c].w);      // ref :286
            }
            if (half + 1 < NCB / HB) __builtin_amdgcn_sched_barrier(0);      // (the second half's accumulators after the first half's epilogue)
        }
        sc += __shfl_xor(sc, 16, 64);
        sc += __shfl_xor(sc, 32, 64);
        if (grp == 0 && p < pairs) pair_s[p] = sc * ps[48 + col];
        my_r = next_r; my_ij = next_ij;
        wave_fence();
    }
}

// Owner-computes gather behind cfconv_filters_mfma: one wave per atom, lanes = filter channels.
//   forward   out[i]   = sum_e F[pid_e] * x[j_e]                                                     ref :180-183
//   backward  dE/dx[i] = sum_e F[pid_e] * gout[j_e] ,  dE/dpos[i] = -sum_e s[pid_e] * delta_e        ref :284-291
template <bool BWD, bool VEC2>
__global__ __launch_bounds__(256) void cfconv_gather(int N, int W, const float4* __restrict__ rows, const int* __restrict__ cnt,
                                                     int cap, const int* __restrict__ pid, int pair_cap,
                                                     const float* __restrict__ filt, const float* __restrict__ pair_s,
                                                     const float* __restrict__ v, float* __restrict__ out,
                                                     float* __restrict__ pos_grad, const float4* __restrict__ sorted_pos) {
    const int lane = lane_id();
    // atoms in cell order, an XCD taking a contiguous part of it: both ends of a pair then read its filter row through
    // the same L2, close in time
    const int k = __builtin_amdgcn_readfirstlane(xcd_contiguous_wave_id());      // (wave-uniform: the atom's id and count through the scalar cache)
    if (k >= N) return;
    const int i = sorted_pos ? __float_as_int(sorted_pos[k].w) & kIdMask : k;
    if (i >= N) return;                                     // (a grid that could not be built: check() reports it)
    const int n = min(cnt[i], cap);
    float acc0 = 0.f, acc1 = 0.f, fx = 0.f, fy = 0.f, fz = 0.f;
    const int c0 = VEC2 ? 2 * lane : lane, c1 = lane + 64;
    for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + lane;
        int my_p = pair_cap, my_j = i;                       // (the all-zero filter row)
        if (e < n) {
            const float4 rec = rows[(size_t)i * cap + e];
            my_j = __float_as_int(rec.w) & kIdMask;
            my_p = pid[(size_t)i * cap + e];
            if (BWD) {
                const float sc = pair_s[my_p];
                fx -= sc * rec.x; fy -= sc * rec.y; fz -= sc * rec.z;
            }
        }
        const int m = min(64, n - e0);
#pragma unroll 4
        for (int q = 0; q < m; q++) {
            const size_t fo = (size_t)__builtin_amdgcn_readlane(my_p, q) * W, vo = (size_t)__builtin_amdgcn_readlane(my_j, q) * W;
            if (VEC2) {
                const float2 f = *reinterpret_cast<const float2*>(filt + fo + c0);
                const float2 u = *reinterpret_cast<const float2*>(v + vo + c0);
                acc0 += f.x * u.x; acc1 += f.y * u.y;
            } else {
                if (c0 < W) acc0 += filt[fo + c0] * v[vo + c0];
                if (c1 < W) acc1 += filt[fo + c1] * v[vo + c1];
            }
        }
    }
    if (VEC2) {
        *reinterpret_cast<float2*>(out + (size_t)i * W + c0) = make_float2(acc0, acc1);
    } else {
        if (c0 < W) out[(size_t)i * W + c0] = acc0;
        if (c1 < W) out[(size_t)i * W + c1] = acc1;
    }
    if (BWD) {
        fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
        if (lane == 0) { pos_grad[3 * i] = fx; pos_grad[3 * i + 1] = fy; pos_grad[3 * i + 2] = fz; }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------------
struct nnpops_cfconv_neighbors {
    int N = 0;
    float cutoff = 0;
    bool periodic = false;
    int device = 0;
    hipStream_t stream = nullptr;
    int cap = 64;
    bool cells_disabled = false;
    bool built = false;
    float4* d_rows = nullptr;
    int* d_cnt = nullptr;
    int* d_status = nullptr;
    // cell grid
    CellGrid* d_grid = nullptr;
    int *d_cell_count = nullptr, *d_cell_start = nullptr, *d_atom_cell = nullptr, *d_atom_rank = nullptr;
    int* d_sorted_cell = nullptr;   // [N] cell of the atom in every sorted slot (rows_cells reads it with the slot's position)
    int *d_unsorted = nullptr, *d_sorted = nullptr;
    float4* d_sorted_pos = nullptr;
    int max_cells = 0;
    int* d_hist = nullptr;          // two-kernel cell build (celllist.h)
    int* d_bins = nullptr;
    int bin_cap = 64;
    // half list behind the rows (scan_half / half_slots): built with the rows once a matrix-core convolution has
    // asked for it, on demand before that
    int *d_lo_cnt = nullptr, *d_half_off = nullptr, *d_pid = nullptr, *d_ids = nullptr;
    float* d_half_r = nullptr;
    int2* d_half_ij = nullptr;
    bool want_half = false, half_built = false;
    unsigned long long epoch = 0;      // number of the last build (process-wide counter): what a convolution keeps per list (its filter rows) is valid for one build
    bool cell_ordered = false;      // the last build went through the cell grid: d_sorted_pos lists the atoms in cell order
    int pair_cap() const { return (int)std::min<size_t>((size_t)N * cap / 2, (size_t)INT32_MAX - 1); }
};

static int alloc_half(nnpops_cfconv_neighbors* h) {
    dev_free(h->d_pid); dev_free(h->d_half_r); dev_free(h->d_half_ij); dev_free(h->d_ids);
    h->d_ids = nullptr; h->d_pid = nullptr; h->d_half_r = nullptr; h->d_half_ij = nullptr;
    int rc;
    if ((rc = dev_alloc(&h->d_pid, (size_t)h->N * h->cap))) return rc;
    if ((rc = dev_alloc(&h->d_ids, (size_t)h->N * h->cap))) return rc;
    if ((rc = dev_alloc(&h->d_half_r, (size_t)h->pair_cap() + 1))) return rc;
    if ((rc = dev_alloc(&h->d_half_ij, (size_t)h->pair_cap() + 1))) return rc;
    if (hipMemset(h->d_half_r, 0, sizeof(float) * ((size_t)h->pair_cap() + 1)) != hipSuccess ||
        hipMemset(h->d_half_ij, 0, sizeof(int2) * ((size_t)h->pair_cap() + 1)) != hipSuccess ||
        hipMemset(h->d_pid, 0, sizeof(int) * (size_t)h->N * h->cap) != hipSuccess ||
        hipMemset(h->d_ids, 0, sizeof(int) * (size_t)h->N * h->cap) != hipSuccess)
        return fail(NNPOPS_ERR_HIP, "memset failed");
    return NNPOPS_OK;
}

// the two launches behind the rows that give every pair its slot
static int launch_half_build(nnpops_cfconv_neighbors* h, hipStream_t stream) {
    const float4* order = h->cell_ordered ? h->d_sorted_pos : nullptr;
    hipLaunchKernelGGL(scan_half, dim3(1), dim3(1024), 0, stream, h->N, h->d_lo_cnt, h->d_half_off);
    hipLaunchKernelGGL(half_slots, dim3(h->N), dim3(64), 0, stream, h->d_rows, h->d_ids, h->d_cnt, h->d_half_off, h->cap, h->pair_cap(),
                       h->d_pid, h->d_half_r, h->d_half_ij, h->d_status, h->N, order);
    NNPOPS_HIP_TRY(hipGetLastError());
    h->half_built = true;
    return NNPOPS_OK;
}

struct nnpops_cfconv {
    ConvParams p{};
    bool periodic = false;
    int device = 0;
    hipStream_t stream = nullptr;
    float *d_w1t = nullptr, *d_b1 = nullptr, *d_w2t = nullptr, *d_b2 = nullptr;
    // shifted softplus on the matrix-core path: log2(e) folded into layer 1 and ln(2) into W2, so that the
    // activation is exp2 / fma / log2 with no scaling multiplies (see activate_fast)
    float *d_w1t_s = nullptr, *d_b1_s = nullptr, *d_w2t_s = nullptr;
    int blocks = 256;
    bool force_valu = false;        // $NNPOPS_CFCONV_VALU=1: keep the matrix cores out (A/B timing, debugging)
    bool half_list = true;          // $NNPOPS_CFCONV_HALF=0: matrix-core kernels over the full rows (every pair from both ends)
    // split-fp16 second layer (cfconv_filters_h2): W1^T with the bias row, the two fp16 planes of W2 [out][in];
    // split_ok = width a multiple of 32 and every operand provably inside the fp16 range ($NNPOPS_CFCONV_SPLIT=0: never)
    float* d_w1b = nullptr;
    _Float16 *d_w2h = nullptr, *d_w2l = nullptr, *d_w1h = nullptr, *d_w1l = nullptr;
    bool split_ok = false, split_l1 = false;      // split_l1: layer 1 as split products too (G + 1 <= 64 and the planes fit in LDS)
    // filter rows F[pid][W] and pair forces s[pid] of the half-list path (+1: the all-zero row); sized on first use
    float *d_filt = nullptr, *d_pair_s = nullptr;
    size_t spill_rows = 0;
    // whose filter rows d_filt holds (the forward call writes them; a backward call on the same build of the same list reads them
    // back instead of storing them again: the same numbers to the last bit or two, 67 MB per call at config 3)
    const void* filt_list = nullptr;
    unsigned long long filt_epoch = 0;
    bool graph_seen = false;      // a filters launch of this convolution has been captured into a graph: replays write d_filt unseen
    bool reuse_filters = true;                     // $NNPOPS_CFCONV_REUSE_FILTERS=0: always store
};

extern "C" {

int nnpops_cfconv_neighbors_create(nnpops_cfconv_neighbors_t* out, int num_atoms, float cutoff, int periodic, int device) {
    NNPOPS_REQUIRE(out != nullptr, "out handle pointer is NULL");
    *out = nullptr;
    NNPOPS_REQUIRE(num_atoms > 0 && num_atoms <= kIdMask, "num_atoms must be in [1, %d]", kIdMask);
    NNPOPS_REQUIRE(cutoff > 0, "cutoff must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(NNPOPS_ERR_NO_DEVICE, "no HIP device available");
    NNPOPS_REQUIRE(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);
    auto* h = new nnpops_cfconv_neighbors();
    h->N = num_atoms; h->cutoff = cutoff; h->periodic = periodic != 0; h->device = device;
    h->max_cells = num_atoms + 4096;
    DeviceGuard guard(device);
    int rc;
    auto cleanup = [&](int code) { nnpops_cfconv_neighbors_destroy(h); return code; };
    if ((rc = dev_alloc(&h->d_rows, (size_t)num_atoms * h->cap))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_cnt, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_status, (size_t)kStWordsN))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_grid, 1))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_cell_count, (size_t)h->max_cells))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_cell_start, (size_t)h->max_cells + 1))) return cleanup(rc);
    if (const char* e = std::getenv("NNPOPS_CELL_BIN_CAP")) h->bin_cap = std::max(4, std::atoi(e) & ~3);   // tests: force growth
    if (periodic && num_atoms <= kBinnedAtoms) {
        if ((rc = dev_alloc(&h->d_hist, (size_t)kHistWords))) return cleanup(rc);
        if ((rc = dev_alloc(&h->d_bins, (size_t)kBinnedCells * h->bin_cap))) return cleanup(rc);
        if (hipMemset(h->d_hist, 0, sizeof(int) * kHistWords) != hipSuccess) return cleanup(fail(NNPOPS_ERR_HIP, "memset failed"));
    }
    if ((rc = dev_alloc(&h->d_lo_cnt, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_half_off, (size_t)num_atoms + 1))) return cleanup(rc);
    if (hipMemset(h->d_lo_cnt, 0, sizeof(int) * num_atoms) != hipSuccess || hipMemset(h->d_half_off, 0, sizeof(int) * ((size_t)num_atoms + 1)) != hipSuccess)
        return cleanup(fail(NNPOPS_ERR_HIP, "memset failed"));
    if ((rc = alloc_half(h))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_atom_cell, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_atom_rank, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_sorted_cell, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_unsorted, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_sorted, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_sorted_pos, (size_t)num_atoms))) return cleanup(rc);
    if (hipMemset(h->d_cnt, 0, sizeof(int) * num_atoms) != hipSuccess || hipMemset(h->d_status, 0, sizeof(int) * kStWordsN) != hipSuccess)
        return cleanup(fail(NNPOPS_ERR_HIP, "memset failed"));
    *out = h;
    return NNPOPS_OK;
}

int nnpops_cfconv_neighbors_destroy(nnpops_cfconv_neighbors_t h) {
    if (!h) return NNPOPS_OK;
    DeviceGuard guard(h->device);
    dev_free(h->d_rows); dev_free(h->d_cnt); dev_free(h->d_status);
    dev_free(h->d_hist); dev_free(h->d_bins);
    dev_free(h->d_lo_cnt); dev_free(h->d_half_off); dev_free(h->d_pid); dev_free(h->d_half_r); dev_free(h->d_half_ij); dev_free(h->d_ids);
    dev_free(h->d_grid); dev_free(h->d_cell_count); dev_free(h->d_cell_start); dev_free(h->d_atom_cell);
    dev_free(h->d_atom_rank); dev_free(h->d_sorted_cell); dev_free(h->d_unsorted); dev_free(h->d_sorted); dev_free(h->d_sorted_pos);
    delete h;
    return NNPOPS_OK;
}

int nnpops_cfconv_neighbors_set_stream(nnpops_cfconv_neighbors_t h, void* stream) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    h->stream = (hipStream_t)stream;
    return NNPOPS_OK;
}

int nnpops_cfconv_neighbors_build(nnpops_cfconv_neighbors_t h, const float* positions, const float* box) {
    NNPOPS_REQUIRE(h != nullptr && positions != nullptr, "NULL argument");
    NNPOPS_REQUIRE(!h->periodic || box, "periodic neighbour list needs box vectors");
    DeviceGuard guard(h->device);
    const int N = h->N;
    const bool per = h->periodic;
    const float c2 = h->cutoff * h->cutoff;
    // positions and box are read in place (the rows keep displacements, nothing refers back to them later); the
    // status words are cleared by create() and by check() after it has read them, not per build
    const bool use_cells = N >= 1024 && !h->cells_disabled;
    if (use_cells) {
        CellBuffers cb{h->d_grid, h->d_cell_count, h->d_cell_start, h->d_atom_cell, h->d_atom_rank, h->d_unsorted,
                       h->d_sorted, h->d_sorted_pos, h->max_cells, h->d_hist, h->d_bins, h->bin_cap};
        cb.sorted_cell = h->d_sorted_cell;
        launch_cell_build(h->stream, N, positions, box, per, h->cutoff, nullptr, cb);
        if (per)
            hipLaunchKernelGGL(rows_cells<true>, dim3(N), dim3(64), 0, h->stream, box, c2, h->d_grid, h->d_cell_start,
                               h->d_sorted_cell, h->d_sorted_pos, h->d_rows, h->d_ids, h->cap, h->d_cnt, h->d_lo_cnt, h->d_status, h->d_hist);
        else
            hipLaunchKernelGGL(rows_cells<false>, dim3(N), dim3(64), 0, h->stream, box, c2, h->d_grid, h->d_cell_start,
                               h->d_sorted_cell, h->d_sorted_pos, h->d_rows, h->d_ids, h->cap, h->d_cnt, h->d_lo_cnt, h->d_status, h->d_hist);
    } else if (per) {
        hipLaunchKernelGGL(rows_allpairs<true>, dim3(N), dim3(64), 0, h->stream, N, positions, box, c2, h->d_rows, h->d_ids, h->cap, h->d_cnt, h->d_lo_cnt);
    } else {
        hipLaunchKernelGGL(rows_allpairs<false>, dim3(N), dim3(64), 0, h->stream, N, positions, box, c2, h->d_rows, h->d_ids, h->cap, h->d_cnt, h->d_lo_cnt);
    }
    NNPOPS_HIP_TRY(hipGetLastError());
    h->built = true;
    h->half_built = false;
    static std::atomic<unsigned long long> build_counter{0};      // (unique across lists: a new list at a freed list's address is another build)
    h->epoch = ++build_counter;
    h->cell_ordered = use_cells;
    if (h->want_half) return launch_half_build(h, h->stream);
    return NNPOPS_OK;
}

int nnpops_cfconv_neighbors_check(nnpops_cfconv_neighbors_t h, int* num_pairs) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    DeviceGuard guard(h->device);
    int st[kStWordsN] = {0, 0, 0, 0};
    // the statistics are recomputed on every call; the overflow word keeps the builder's "grid unusable" bit
    NNPOPS_HIP_TRY(hipMemsetAsync(h->d_status + kStMaxRow, 0, 2 * sizeof(int), h->stream));
    hipLaunchKernelGGL(row_stats, dim3(std::min(64, div_up(h->N, 256))), dim3(256), 0, h->stream, h->N, h->d_cnt, h->d_rows,
                       h->cap, h->d_status);
    NNPOPS_HIP_TRY(hipGetLastError());
    NNPOPS_HIP_TRY(hipMemcpyAsync(st, h->d_status, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    NNPOPS_HIP_TRY(hipMemsetAsync(h->d_status, 0, sizeof(int) * kStWordsN, h->stream));   // consumed: builds do not clear it
    NNPOPS_HIP_TRY(hipStreamSynchronize(h->stream));
    if (num_pairs) *num_pairs = st[kStPairs];
    if (st[kStOverflow] & 4) {            // a cell holds more atoms than a bin of the two-kernel grid build: grow the bins
        const int old_bin = h->bin_cap;
        h->bin_cap *= 2;
        dev_free(h->d_bins);
        h->d_bins = nullptr;
        int rc = dev_alloc(&h->d_bins, (size_t)kBinnedCells * h->bin_cap);
        if (rc != NNPOPS_OK) return rc;
        h->built = false;
        return fail(NNPOPS_ERR_CAPACITY, "cell bins overflowed (%d ids per cell); grown to %d, call build() again", old_bin, h->bin_cap);
    }
    if (st[kStOverflow] & 2) {
        h->cells_disabled = true;
        h->built = false;
        return fail(NNPOPS_ERR_CAPACITY, "periodic box is fewer than 3 cells wide on some axis: switched to the all-pairs "
                                         "neighbour search, call build() again");
    }
    if (st[kStOverflow]) {
        const int old = h->cap;
        while (h->cap < st[kStMaxRow]) h->cap *= 2;
        dev_free(h->d_rows);
        h->d_rows = nullptr;
        int rc = dev_alloc(&h->d_rows, (size_t)h->N * h->cap);
        if (rc != NNPOPS_OK) return rc;
        if ((rc = alloc_half(h)) != NNPOPS_OK) return rc;
        h->built = false;
        return fail(NNPOPS_ERR_CAPACITY, "neighbour rows overflowed (max %d > %d); capacity grown to %d, call build() again",
                    st[kStMaxRow], old, h->cap);
    }
    return NNPOPS_OK;
}

int nnpops_cfconv_neighbors_export(nnpops_cfconv_neighbors_t h, int capacity, int32_t* pair_atoms, float* distances) {
    NNPOPS_REQUIRE(h != nullptr && pair_atoms && distances, "NULL argument");
    NNPOPS_REQUIRE(h->built, "export() must follow build()");
    DeviceGuard guard(h->device);
    std::vector<float4> rows((size_t)h->N * h->cap);
    std::vector<int> cnt(h->N);
    NNPOPS_HIP_TRY(hipStreamSynchronize(h->stream));
    NNPOPS_HIP_TRY(hipMemcpy(rows.data(), h->d_rows, rows.size() * sizeof(float4), hipMemcpyDeviceToHost));
    NNPOPS_HIP_TRY(hipMemcpy(cnt.data(), h->d_cnt, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
    int p = 0;
    for (int i = 0; i < h->N; i++) {
        std::vector<std::pair<int, float>> half;
        for (int e = 0; e < std::min(cnt[i], h->cap); e++) {
            const float4 r = rows[(size_t)i * h->cap + e];
            int j;
            std::memcpy(&j, &r.w, sizeof(int));
            j &= kIdMask;
            if (j > i) half.push_back({j, std::sqrt(r.x * r.x + r.y * r.y + r.z * r.z)});
        }
        std::sort(half.begin(), half.end());
        for (auto& e : half) {
            if (p >= capacity) return fail(NNPOPS_ERR_CAPACITY, "export capacity %d too small", capacity);
            pair_atoms[p] = i;
            pair_atoms[capacity + p] = e.first;
            distances[p] = e.second;
            p++;
        }
    }
    return NNPOPS_OK;
}

int nnpops_cfconv_create(nnpops_cfconv_t* out, int num_atoms, int width, int num_gaussians, float cutoff, int periodic,
                         float gaussian_width, int activation, const float* w1, const float* b1, const float* w2,
                         const float* b2, int device) {
    NNPOPS_REQUIRE(out != nullptr, "out handle pointer is NULL");
    *out = nullptr;
    NNPOPS_REQUIRE(num_atoms > 0, "num_atoms must be positive");
    NNPOPS_REQUIRE(w1 && b1 && w2 && b2, "NULL weight pointer");
    NNPOPS_REQUIRE(activation == 0 || activation == 1, "Invalid value of \"activation\"");
    NNPOPS_REQUIRE(cutoff > 0 && gaussian_width > 0, "cutoff and gaussian_width must be positive");
    NNPOPS_REQUIRE(num_gaussians >= 2, "num_gaussians must be at least 2 (centres are spaced cutoff/(G-1))");
    if (width < 1 || width > kMaxWidth || num_gaussians > kMaxGauss)
        return fail(NNPOPS_ERR_UNSUPPORTED, "this build supports width in [1, %d] and up to %d Gaussians (got %d, %d)",
                    kMaxWidth, kMaxGauss, width, num_gaussians);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(NNPOPS_ERR_NO_DEVICE, "no HIP device available");
    NNPOPS_REQUIRE(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);
    auto* h = new nnpops_cfconv();
    h->p.N = num_atoms; h->p.W = width; h->p.G = num_gaussians; h->p.cutoff = cutoff;
    h->p.sigma_inv = 1.0f / gaussian_width; h->p.activation = activation;
    h->p.skip_filter_store = 0; h->p.dy_scale = 1.0f; h->p.dy_unscale = 1.0f;
    h->periodic = periodic != 0; h->device = device;
    const int W = width, G = num_gaussians;
    std::vector<float> w1t((size_t)G * W), w2t((size_t)W * W);
    for (int a = 0; a < W; a++)
        for (int g = 0; g < G; g++) w1t[(size_t)g * W + a] = w1[(size_t)a * G + g];        // core layout [W][G], CpuCFConv.cpp:163
    for (int a = 0; a < W; a++)
        for (int b = 0; b < W; b++) w2t[(size_t)b * W + a] = w2[(size_t)a * W + b];        // [out][in], CpuCFConv.cpp:174
    DeviceGuard guard(device);
    int rc;
    auto cleanup = [&](int code) { nnpops_cfconv_destroy(h); return code; };
    if ((rc = dev_alloc(&h->d_w1t, w1t.size()))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_w2t, w2t.size()))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_b1, (size_t)W))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_b2, (size_t)W))) return cleanup(rc);
    if (hipMemcpy(h->d_w1t, w1t.data(), w1t.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->d_w2t, w2t.data(), w2t.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->d_b1, b1, (size_t)W * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->d_b2, b2, (size_t)W * 4, hipMemcpyHostToDevice) != hipSuccess)
        return cleanup(fail(NNPOPS_ERR_HIP, "weight upload failed"));
    if (activation == 0) {
        std::vector<float> w1s(w1t), w2s(w2t), b1s(b1, b1 + W);
        for (float& v : w1s) v *= kLog2e;
        for (float& v : b1s) v *= kLog2e;
        for (float& v : w2s) v *= 1.0f / kLog2e;
        if ((rc = dev_alloc(&h->d_w1t_s, w1s.size()))) return cleanup(rc);
        if ((rc = dev_alloc(&h->d_w2t_s, w2s.size()))) return cleanup(rc);
        if ((rc = dev_alloc(&h->d_b1_s, (size_t)W))) return cleanup(rc);
        if (hipMemcpy(h->d_w1t_s, w1s.data(), w1s.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_w2t_s, w2s.data(), w2s.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_b1_s, b1s.data(), (size_t)W * 4, hipMemcpyHostToDevice) != hipSuccess)
            return cleanup(fail(NNPOPS_ERR_HIP, "weight upload failed"));
    }
    if (W % 32 == 0 && W <= 128) {
        // operands of the split-fp16 second layer, in the scaled domain the kernels work in (see activate_fast)
        const float s1 = activation == 0 ? kLog2e : 1.0f, s2 = activation == 0 ? 1.0f / kLog2e : 1.0f;
        const int Gq = h2_l1_rows(G);
        std::vector<float> w1b((size_t)Gq * W, 0.f);
        std::vector<_Float16> w2h((size_t)W * W), w2l((size_t)W * W);
        double max_w2 = 0, max_y = 0, max_dy = 0;
        for (int f = 0; f < W; f++) {
            double row = 0;
            for (int g = 0; g < G; g++) {
                w1b[(size_t)g * W + f] = s1 * w1[(size_t)f * G + g];
                row += std::fabs((double)s1 * w1[(size_t)f * G + g]);
            }
            w1b[(size_t)G * W + f] = s1 * b1[f];
            max_y = std::max(max_y, row + std::fabs((double)s1 * b1[f]) + 1.0);     // |act(s)| <= |s| + 1 in either domain
            max_dy = std::max(max_dy, row * 0.61 / gaussian_width);                  // |d gamma / dr| <= 0.607 / sigma, act' <= 1
        }
        for (int f2 = 0; f2 < W; f2++)
            for (int k = 0; k < W; k++) {
                const float v = s2 * w2[(size_t)f2 * W + k];
                const _Float16 hi = (_Float16)v;
                w2h[(size_t)f2 * W + k] = hi;
                w2l[(size_t)f2 * W + k] = (_Float16)((v - (float)hi) * kLoScale);
                max_w2 = std::max(max_w2, (double)std::fabs(v));
            }
        const double limit = 3.0e4;                         // fp16 holds 65504; the low planes stay below 32 in any case
        h->split_ok = max_w2 < limit && max_y < limit && max_dy < limit;
        {   // dY1 goes into its fp16 planes scaled to the top of the range (ConvParams::dy_scale): the largest power of two, up to 2^12,
            // that keeps the bound on |dY1| below `limit`
            int k = 0;
            while (k < 12 && max_dy * std::ldexp(1.0, k + 1) < limit) k++;
            h->p.dy_scale = (float)std::ldexp(1.0, k);
            h->p.dy_unscale = (float)std::ldexp(1.0, -k);
        }
        int level = 2;                                      // 0: fp32 only, 1: second layer split, 2: both layers where possible
        if (const char* e = std::getenv("NNPOPS_CFCONV_SPLIT")) level = std::atoi(e);
        if (const char* e = std::getenv("NNPOPS_CFCONV_REUSE_FILTERS")) h->reuse_filters = std::atoi(e) != 0;
        h->split_ok = h->split_ok && level != 0;
        h->split_l1 = h->split_ok && level >= 2 && G + 1 <= 64 &&
                      h2_weight_bytes_l1h(W, G) + kMaxWavesPerBlock * h2_wave_bytes(W) <= (size_t)160 * 1024;
        if (h->split_l1) {
            const int rowh = h2_l1_row_bytes(G) / 2;        // halves per plane row
            std::vector<_Float16> w1h((size_t)W * rowh, (_Float16)0.f), w1l((size_t)W * rowh, (_Float16)0.f);
            for (int f = 0; f < W; f++)
                for (int k = 0; k <= G; k++) {
                    const float v = w1b[(size_t)k * W + f];  // (row G = the bias)
                    const _Float16 hi = (_Float16)v;
                    w1h[(size_t)f * rowh + k] = hi;
                    w1l[(size_t)f * rowh + k] = (_Float16)((v - (float)hi) * kLoScale);
                }
            if ((rc = dev_alloc(&h->d_w1h, w1h.size()))) return cleanup(rc);
            if ((rc = dev_alloc(&h->d_w1l, w1l.size()))) return cleanup(rc);
            if (hipMemcpy(h->d_w1h, w1h.data(), w1h.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(h->d_w1l, w1l.data(), w1l.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
                return cleanup(fail(NNPOPS_ERR_HIP, "weight upload failed"));
        }
        if (h->split_ok) {
            if ((rc = dev_alloc(&h->d_w1b, w1b.size()))) return cleanup(rc);
            if ((rc = dev_alloc(&h->d_w2h, w2h.size()))) return cleanup(rc);
            if ((rc = dev_alloc(&h->d_w2l, w2l.size()))) return cleanup(rc);
            if (hipMemcpy(h->d_w1b, w1b.data(), w1b.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(h->d_w2h, w2h.data(), w2h.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(h->d_w2l, w2l.data(), w2l.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
                return cleanup(fail(NNPOPS_ERR_HIP, "weight upload failed"));
        }
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->blocks = prop.multiProcessorCount;
    if (const char* e = std::getenv("NNPOPS_CFCONV_VALU")) h->force_valu = std::atoi(e) != 0;
    if (const char* e = std::getenv("NNPOPS_CFCONV_HALF")) h->half_list = std::atoi(e) != 0;
    *out = h;
    return NNPOPS_OK;
}

int nnpops_cfconv_destroy(nnpops_cfconv_t h) {
    if (!h) return NNPOPS_OK;
    DeviceGuard guard(h->device);
    dev_free(h->d_w1t); dev_free(h->d_w2t); dev_free(h->d_b1); dev_free(h->d_b2);
    dev_free(h->d_w1t_s); dev_free(h->d_w2t_s); dev_free(h->d_b1_s);
    dev_free(h->d_filt); dev_free(h->d_pair_s);
    dev_free(h->d_w1b); dev_free(h->d_w2h); dev_free(h->d_w2l); dev_free(h->d_w1h); dev_free(h->d_w1l);
    delete h;
    return NNPOPS_OK;
}

int nnpops_cfconv_set_stream(nnpops_cfconv_t h, void* stream) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    h->stream = (hipStream_t)stream;
    return NNPOPS_OK;
}

}  // extern "C"

namespace {

template <int ACT, int CPL, bool BWD, bool WLDS = true>
int launch_conv(nnpops_cfconv* h, nnpops_cfconv_neighbors* nb, const float* x, const float* gout, float* out, float* pos_grad) {
    // as many waves per workgroup as fit next to the shared weights in 160 KiB of LDS
    const size_t budget = 156 * 1024 / sizeof(float);
    const size_t wfl = WLDS ? conv_weight_floats(h->p.W, h->p.G) : 0, per_wave = conv_wave_floats(h->p.W, h->p.G, BWD);
    if (wfl + per_wave > budget)
        return fail(NNPOPS_ERR_UNSUPPORTED, "CFConv tiles (%zu floats) do not fit in LDS", wfl + per_wave);
    const int wpb = (int)std::min<size_t>(kMaxWavesPerBlock, (budget - wfl) / per_wave);
    const size_t lds = (wfl + (size_t)wpb * per_wave) * sizeof(float);
    auto k = cfconv_kernel<ACT, CPL, BWD, WLDS>;
    if (lds > 64 * 1024)
        NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int blocks = std::max(1, std::min(h->blocks, div_up(h->p.N, wpb)));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * wpb), lds, h->stream, h->p, h->d_w1t, h->d_b1, h->d_w2t, h->d_b2,
                       nb->d_rows, nb->d_cnt, nb->cap, x, gout, out, pos_grad);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

template <int ACT, int NCB>
int launch_forward_mfma(nnpops_cfconv* h, nnpops_cfconv_neighbors* nb, const float* x, float* out) {
    const size_t budget = 160 * 1024 / sizeof(float);   // all of a CU: 8 waves (2 per SIMD) at W = 128, G <= 52
    const size_t wfl = mfma_weight_floats(h->p.W, h->p.G), per_wave = mfma_wave_floats(h->p.W);
    const int wpb = (int)std::min<size_t>(kMaxWavesPerBlock, (budget - wfl) / per_wave);
    const size_t lds = (wfl + (size_t)wpb * per_wave) * sizeof(float);
    auto k = cfconv_forward_mfma<ACT, NCB>;
    if (lds > 64 * 1024)
        NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int blocks = std::max(1, std::min(h->blocks, div_up(h->p.N, wpb)));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * wpb), lds, h->stream, h->p, ACT == 0 ? h->d_w1t_s : h->d_w1t,
                       ACT == 0 ? h->d_b1_s : h->d_b1, ACT == 0 ? h->d_w2t_s : h->d_w2t, h->d_b2, nb->d_rows, nb->d_cnt, nb->cap, x, out);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

template <int ACT, int NCB>
int launch_backward_mfma(nnpops_cfconv* h, nnpops_cfconv_neighbors* nb, const float* x, const float* gout, float* xgrad, float* pos_grad) {
    const size_t budget = 160 * 1024 / sizeof(float);   // all of a CU: 8 waves (2 per SIMD) at W = 128, G <= 52
    const size_t wfl = mfma_weight_floats(h->p.W, h->p.G), per_wave = mfma_wave_floats_bwd(h->p.W);
    const int wpb = (int)std::min<size_t>(kMaxWavesPerBlock, (budget - wfl) / per_wave);
    const size_t lds = (wfl + (size_t)wpb * per_wave) * sizeof(float);
    auto k = cfconv_backward_mfma<ACT, NCB>;
    if (lds > 64 * 1024)
        NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int blocks = std::max(1, std::min(h->blocks, div_up(h->p.N, wpb)));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * wpb), lds, h->stream, h->p, ACT == 0 ? h->d_w1t_s : h->d_w1t,
                       ACT == 0 ? h->d_b1_s : h->d_b1, ACT == 0 ? h->d_w2t_s : h->d_w2t, h->d_b2, nb->d_rows, nb->d_cnt, nb->cap, x, gout,
                       xgrad, pos_grad);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

// Half-list path: filters once per pair, then the owner-computes gather.
int ensure_half_path(nnpops_cfconv* h, nnpops_cfconv_neighbors* nb) {
    nb->want_half = true;                                   // later builds carry the pair slots along
    if (!nb->half_built) {
        const int rc = launch_half_build(nb, h->stream);
        if (rc != NNPOPS_OK) return rc;
    }
    const size_t need = (size_t)nb->pair_cap() + 1;
    if (h->spill_rows < need) {                             // (first use, or the neighbour rows have grown: not capturable)
        dev_free(h->d_filt); dev_free(h->d_pair_s);
        h->d_filt = nullptr; h->d_pair_s = nullptr; h->spill_rows = 0; h->filt_list = nullptr;
        int rc;
        if ((rc = dev_alloc(&h->d_filt, need * h->p.W))) return rc;
        if ((rc = dev_alloc(&h->d_pair_s, need))) return rc;
        h->spill_rows = need;
    }

    return NNPOPS_OK;
}

template <int ACT, int NCB, bool BWD>
int launch_half_mfma(nnpops_cfconv* h, nnpops_cfconv_neighbors* nb, const float* x, const float* gout, float* out, float* pos_grad) {
    int rc = ensure_half_path(h, nb);
    if (rc != NNPOPS_OK) return rc;
    const int pair_cap = nb->pair_cap();
    bool launched = false;
    bool capturing = true;                                  // (unknown = treated as capturing)
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) capturing = false;
    }
    if constexpr (NCB % 2 == 0) {
        if (h->split_ok) {                                  // second layer as split-fp16 matrix products
            const size_t budget = 160 * 1024;
            const size_t wb = h->split_l1 ? h2_weight_bytes_l1h(h->p.W, h->p.G) : h2_weight_bytes(h->p.W, h->p.G);
            const size_t per_wave = h2_wave_bytes(h->p.W);
            if (wb + per_wave <= budget) {
                const int wpb = (int)std::min<size_t>(kMaxWavesPerBlock, (budget - wb) / per_wave);
                const size_t lds = wb + (size_t)wpb * per_wave;
                auto k = h->split_l1 ? cfconv_filters_h2<ACT, NCB, BWD, true> : cfconv_filters_h2<ACT, NCB, BWD, false>;
                if constexpr (!BWD) {
                    // forward, both layers split: 32 pairs per wave, layer 2 fed from registers (cfconv_filters_h2x2; $NNPOPS_CFCONV_FWD32=0: the 16-pair kernel)
                    const bool fwd32 = !(std::getenv("NNPOPS_CFCONV_FWD32") && std::atoi(std::getenv("NNPOPS_CFCONV_FWD32")) == 0);
                    if (h->split_l1 && fwd32) {
                        // (measured, 10 000 atoms: two tiles per pass on 8 waves 45.8 us; one tile per pass on 12 waves -- three per SIMD,
                        //  154 registers -- 50.5 us; on 16 waves, spilling, 68.6 us: the weight reads a second tile shares are worth more than a third wave)
                        // (... and on FOUR waves -- one per SIMD, 512 registers -- two tiles per pass 57.0 us, four tiles per pass 84.8 us)
                        auto k2 = cfconv_filters_h2x2<ACT, NCB, 2, kMaxWavesPerBlock>;
                        const int wv = kMaxWavesPerBlock;
                        const size_t lds2 = h2_weight_bytes_l1h(h->p.W, h->p.G) + (size_t)wv * 128 * sizeof(float);
                        if (lds2 > 64 * 1024)
                            NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
                        hipLaunchKernelGGL(k2, dim3(h->blocks), dim3(64 * wv), lds2, h->stream, h->p, h->d_w1h, h->d_w1l, h->d_w2h, h->d_w2l,
                                           h->d_b2, nb->d_half_off, nb->d_half_r, pair_cap, h->d_filt);
                        launched = true;
                    }
                }
                ConvParams cp = h->p;
                if (BWD && !capturing && !h->graph_seen && h->reuse_filters && h->filt_list == nb && h->filt_epoch == nb->epoch)
                    cp.skip_filter_store = 1;
                if constexpr (BWD) {
                    // backward, both layers split: one pass per layer, operands from registers (cfconv_filters_h2b; $NNPOPS_CFCONV_BWD1=0: the four-pass kernel)
                    const bool bwd1 = !(std::getenv("NNPOPS_CFCONV_BWD1") && std::atoi(std::getenv("NNPOPS_CFCONV_BWD1")) == 0);
                    if (h->split_l1 && bwd1) {
                        // FOUR waves per workgroup: one per SIMD, which gives the wave all 512 registers of its lanes.  The kernel holds four
                        // accumulator sets, four sets of operand fragments and the x / gout rows of its pair: ~410 registers.  With two waves per
                        // SIMD (256 each) it spills 444 bytes per lane and takes 206 us; with one, nothing is spilled: 99 us (the four-pass kernel,
                        // two waves per SIMD: 125.5 us).  Widths below 128 fit two waves per SIMD.
                        const int bw_env = std::getenv("NNPOPS_CFCONV_BWD_WAVES") ? std::atoi(std::getenv("NNPOPS_CFCONV_BWD_WAVES")) : 0;
                        const int wv = bw_env == 4 || bw_env == 8 ? bw_env : (NCB >= 6 ? 4 : 8);
                        auto k2 = wv == 4 ? cfconv_filters_h2b<ACT, NCB, 4> : cfconv_filters_h2b<ACT, NCB, 8>;
                        const size_t lds2 = h2_weight_bytes_l1h(h->p.W, h->p.G) + ((size_t)h->p.W + wv * 96) * sizeof(float);
                        if (lds2 > 64 * 1024)
                            NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
                        hipLaunchKernelGGL(k2, dim3(h->blocks), dim3(64 * wv), lds2, h->stream, cp, h->d_w1h, h->d_w1l, h->d_w2h, h->d_w2l,
                                           h->d_b2, nb->d_half_off, nb->d_half_r, nb->d_half_ij, pair_cap, x, gout, h->d_filt, h->d_pair_s);
                        launched = true;
                    }
                }
                if (!launched) {
                if (lds > 64 * 1024)
                    NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(k, dim3(h->blocks), dim3(64 * wpb), lds, h->stream, cp, h->d_w1b, h->d_w1h, h->d_w1l, h->d_w2h,
                                   h->d_w2l, h->d_b2, nb->d_half_off, nb->d_half_r, nb->d_half_ij, pair_cap, x, gout, h->d_filt,
                                   h->d_pair_s);
                launched = true;
                }
            }
        }
    }
    // Either kernel, either direction, leaves this list's rows in d_filt -- once it has RUN.  A launch that is only being captured has
    // written nothing yet, and a replay of the captured graph writes d_filt behind the host's back, for whatever build of whatever list
    // it was captured on: nothing is recorded while capturing, and a convolution that has ever been captured never
    // reads rows back again (graph_seen) -- the host cannot know what the last replay left there.
    // (Call-history dependence, documented: a backward call that reads the forward call's rows back differs from one that stores its
    //  own in the last bit or two -- the forward kernel adds the second layer's bias behind the products, the backward kernel carries
    //  it in the accumulator, docs/LAB_NOTEBOOK_r05.md #13; both are inside the parity bars, NNPOPS_CFCONV_REUSE_FILTERS=0 switches it off.)
    if (capturing) { h->graph_seen = true; h->filt_list = nullptr; }
    else { h->filt_list = nb; h->filt_epoch = nb->epoch; }
    if (!launched) {
        const size_t budget = 160 * 1024 / sizeof(float);
        const size_t wfl = mfma_weight_floats(h->p.W, h->p.G), per_wave = mfma_wave_floats_bwd(h->p.W);
        const int wpb = (int)std::min<size_t>(kMaxWavesPerBlock, (budget - wfl) / per_wave);
        const size_t lds = (wfl + (size_t)wpb * per_wave) * sizeof(float);
        auto k = cfconv_filters_mfma<ACT, NCB, BWD>;
        if (lds > 64 * 1024)
            NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3(h->blocks), dim3(64 * wpb), lds, h->stream, h->p, ACT == 0 ? h->d_w1t_s : h->d_w1t,
                           ACT == 0 ? h->d_b1_s : h->d_b1, ACT == 0 ? h->d_w2t_s : h->d_w2t, h->d_b2, nb->d_half_off, nb->d_half_r,
                           nb->d_half_ij, pair_cap, x, gout, h->d_filt, h->d_pair_s);
    }
    const int N = h->p.N;
    const float* v = BWD ? gout : x;
    const float4* order = nb->cell_ordered ? nb->d_sorted_pos : nullptr;
    if (h->p.W == 128)
        hipLaunchKernelGGL((cfconv_gather<BWD, true>), dim3(div_up(N, 4)), dim3(256), 0, h->stream, N, h->p.W, nb->d_rows, nb->d_cnt,
                           nb->cap, nb->d_pid, pair_cap, h->d_filt, h->d_pair_s, v, out, pos_grad, order);
    else
        hipLaunchKernelGGL((cfconv_gather<BWD, false>), dim3(div_up(N, 4)), dim3(256), 0, h->stream, N, h->p.W, nb->d_rows, nb->d_cnt,
                           nb->cap, nb->d_pid, pair_cap, h->d_filt, h->d_pair_s, v, out, pos_grad, order);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

template <int ACT, bool BWD>
int dispatch_half_mfma(nnpops_cfconv* h, nnpops_cfconv_neighbors* nb, const float* x, const float* gout, float* out,
                       float* pos_grad, bool& handled) {
    handled = true;
    switch (h->p.W) {
        case 16:  return launch_half_mfma<ACT, 1, BWD>(h, nb, x, gout, out, pos_grad);
        case 32:  return launch_half_mfma<ACT, 2, BWD>(h, nb, x, gout, out, pos_grad);
        case 48:  return launch_half_mfma<ACT, 3, BWD>(h, nb, x, gout, out, pos_grad);
        case 64:  return launch_half_mfma<ACT, 4, BWD>(h, nb, x, gout, out, pos_grad);
        case 80:  return launch_half_mfma<ACT, 5, BWD>(h, nb, x, gout, out, pos_grad);
        case 96:  return launch_half_mfma<ACT, 6, BWD>(h, nb, x, gout, out, pos_grad);
        case 112: return launch_half_mfma<ACT, 7, BWD>(h, nb, x, gout, out, pos_grad);
        case 128: return launch_half_mfma<ACT, 8, BWD>(h, nb, x, gout, out, pos_grad);
        default: handled = false; return NNPOPS_OK;
    }
}

template <int ACT>
int dispatch_backward_mfma(nnpops_cfconv* h, nnpops_cfconv_neighbors* nb, const float* x, const float* gout, float* xgrad,
                           float* pos_grad, bool& handled) {
    handled = true;
    switch (h->p.W) {
        case 16:  return launch_backward_mfma<ACT, 1>(h, nb, x, gout, xgrad, pos_grad);
        case 32:  return launch_backward_mfma<ACT, 2>(h, nb, x, gout, xgrad, pos_grad);
        case 48:  return launch_backward_mfma<ACT, 3>(h, nb, x, gout, xgrad, pos_grad);
        case 64:  return launch_backward_mfma<ACT, 4>(h, nb, x, gout, xgrad, pos_grad);
        case 80:  return launch_backward_mfma<ACT, 5>(h, nb, x, gout, xgrad, pos_grad);
        case 96:  return launch_backward_mfma<ACT, 6>(h, nb, x, gout, xgrad, pos_grad);
        case 112: return launch_backward_mfma<ACT, 7>(h, nb, x, gout, xgrad, pos_grad);
        case 128: return launch_backward_mfma<ACT, 8>(h, nb, x, gout, xgrad, pos_grad);
        default: handled = false; return NNPOPS_OK;
    }
}

// widths that are a multiple of 16 take the matrix-core forward path
template <int ACT>
int dispatch_forward_mfma(nnpops_cfconv* h, nnpops_cfconv_neighbors* nb, const float* x, float* out, bool& handled) {
    handled = true;
    switch (h->p.W) {
        case 16:  return launch_forward_mfma<ACT, 1>(h, nb, x, out);
        case 32:  return launch_forward_mfma<ACT, 2>(h, nb, x, out);
        case 48:  return launch_forward_mfma<ACT, 3>(h, nb, x, out);
        case 64:  return launch_forward_mfma<ACT, 4>(h, nb, x, out);
        case 80:  return launch_forward_mfma<ACT, 5>(h, nb, x, out);
        case 96:  return launch_forward_mfma<ACT, 6>(h, nb, x, out);
        case 112: return launch_forward_mfma<ACT, 7>(h, nb, x, out);
        case 128: return launch_forward_mfma<ACT, 8>(h, nb, x, out);
        default: handled = false; return NNPOPS_OK;
    }
}

template <bool BWD>
int dispatch_conv(nnpops_cfconv* h, nnpops_cfconv_neighbors* nb, const float* x, const float* gout, float* out, float* pos_grad) {
    // the matrix-core kernels keep both weight matrices in LDS beside at least one wave's tile; layers with very many
    // Gaussians (W = 128: G > ~180) do not fit and take the vector kernel, which streams its weights
    const bool mfma_fits = mfma_weight_floats(h->p.W, h->p.G) + mfma_wave_floats_bwd(h->p.W) <= (size_t)160 * 1024 / sizeof(float);
    const bool use_mfma = !h->force_valu && mfma_fits;
    if (use_mfma && h->half_list) {
        bool handled = false;
        const int rc = h->p.activation == 0 ? dispatch_half_mfma<0, BWD>(h, nb, x, gout, out, pos_grad, handled)
                                            : dispatch_half_mfma<1, BWD>(h, nb, x, gout, out, pos_grad, handled);
        if (handled) return rc;
    }
    if (!BWD && use_mfma) {
        bool handled = false;
        const int rc = h->p.activation == 0 ? dispatch_forward_mfma<0>(h, nb, x, out, handled)
                                            : dispatch_forward_mfma<1>(h, nb, x, out, handled);
        if (handled) return rc;
    }
    if (BWD && use_mfma) {
        bool handled = false;
        const int rc = h->p.activation == 0 ? dispatch_backward_mfma<0>(h, nb, x, gout, out, pos_grad, handled)
                                            : dispatch_backward_mfma<1>(h, nb, x, gout, out, pos_grad, handled);
        if (handled) return rc;
    }
    // vector kernels: weights in LDS when they fit beside one wave's tiles, else streamed through the caches
    const size_t budget = 156 * 1024 / sizeof(float);
    const bool fits = conv_weight_floats(h->p.W, h->p.G) + conv_wave_floats(h->p.W, h->p.G, BWD) <= budget && h->p.W <= 128;
    if (fits) {
        const bool two = h->p.W > 64;
        if (h->p.activation == 0)
            return two ? launch_conv<0, 2, BWD>(h, nb, x, gout, out, pos_grad) : launch_conv<0, 1, BWD>(h, nb, x, gout, out, pos_grad);
        return two ? launch_conv<1, 2, BWD>(h, nb, x, gout, out, pos_grad) : launch_conv<1, 1, BWD>(h, nb, x, gout, out, pos_grad);
    }
    const int cpl = div_up(h->p.W, 64);
    if (h->p.activation == 0) {
        if (cpl <= 2) return launch_conv<0, 2, BWD, false>(h, nb, x, gout, out, pos_grad);
        if (cpl <= 4) return launch_conv<0, 4, BWD, false>(h, nb, x, gout, out, pos_grad);
        return launch_conv<0, 8, BWD, false>(h, nb, x, gout, out, pos_grad);
    }
    if (cpl <= 2) return launch_conv<1, 2, BWD, false>(h, nb, x, gout, out, pos_grad);
    if (cpl <= 4) return launch_conv<1, 4, BWD, false>(h, nb, x, gout, out, pos_grad);
    return launch_conv<1, 8, BWD, false>(h, nb, x, gout, out, pos_grad);
}

int check_pair(nnpops_cfconv* h, nnpops_cfconv_neighbors* nb) {
    NNPOPS_REQUIRE(h != nullptr && nb != nullptr, "NULL handle");
    NNPOPS_REQUIRE(nb->built, "the neighbour list has not been built");
    NNPOPS_REQUIRE(nb->N == h->p.N, "neighbour list is for %d atoms, convolution for %d", nb->N, h->p.N);
    NNPOPS_REQUIRE(nb->cutoff == h->p.cutoff, "The cutoff of \"neighbors\" has changed");
    NNPOPS_REQUIRE(nb->device == h->device, "neighbour list and convolution live on different devices");
    return NNPOPS_OK;
}

}  // namespace

extern "C" {

int nnpops_cfconv_compute(nnpops_cfconv_t h, nnpops_cfconv_neighbors_t neighbors, const float* positions, const float* box,
                          const float* input, float* output) {
    (void)positions; (void)box;     // forward works from the distances stored by build() (CpuCFConv.cpp:146-147)
    int rc = check_pair(h, neighbors);
    if (rc != NNPOPS_OK) return rc;
    NNPOPS_REQUIRE(input && output, "NULL device pointer");
    DeviceGuard guard(h->device);
    return dispatch_conv<false>(h, neighbors, input, nullptr, output, nullptr);
}

int nnpops_cfconv_backprop(nnpops_cfconv_t h, nnpops_cfconv_neighbors_t neighbors, const float* positions, const float* box,
                           const float* input, const float* output_deriv, float* input_deriv, float* position_deriv) {
    (void)positions; (void)box;     // displacements were stored by build() from the same positions
    int rc = check_pair(h, neighbors);
    if (rc != NNPOPS_OK) return rc;
    NNPOPS_REQUIRE(input && output_deriv && input_deriv && position_deriv, "NULL device pointer");
    DeviceGuard guard(h->device);
    return dispatch_conv<true>(h, neighbors, input, output_deriv, input_deriv, position_deriv);
}

}  // extern "C"
