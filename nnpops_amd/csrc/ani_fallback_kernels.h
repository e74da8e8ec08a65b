// ani_fallback_kernels.h -- the round-1 kernels, kept as FALLBACKS and A/B arms.  None of them is on a default path.
//
//   ani_radial_backward          radial backward + force gather, lane = (neighbour stream, radial function).  Taken when the gradient
//                                rows cannot be read as aligned float4 (nR % 4 != 0, odd row stride, unaligned tensor) or the id rows
//                                are neither 32 nor 64 wide; default: ani_radial_backward_lanes (ani_radial_bwd.h).
//   ani_angular_forward          angular forward by run merging / by a chunked view of the triple list.  Taken when more than 32 species
//   ani_angular_forward_chunked  pairs can occur in the system (8+ species all present: the matrix-core kernel has 32 quads), or forced
//                                with $NNPOPS_ANI_FORWARD=0 / 1; default: ani_angular_forward_mfma (ani_angular_mfma.h).
//   ani_angular_backward         angular backward with a host-sized pair-matrix tile.  Taken when the full pair matrix of
//                                ani_angular_backward_pair (ani_angular_bwd.h) does not fit the LDS (more than 128 record slots), or
//                                forced with $NNPOPS_ANI_BACKWARD=0.
//
// They share the neighbour rows, records and triple list the builders of ani_kernels.h write, are held to the same parity tests
// (tests/test_ani_gpu.py forces each of them), and are 2-3 times slower than the kernels that replaced them.
#pragma once

#include "ani_kernels.h"

namespace nnpops {

__device__ __forceinline__ void decode_pair(int t, int n, int& p, int& q) { decode_pair_row_major(t, n, p, q); }

// =============================================================================================
// Radial backward + gather of the angular forces (owner computes; the only writer of
// position_deriv[i]; no atomics anywhere in the backward pass).                      ref :228-263, :310-344
// Runs AFTER ani_angular_backward.  The angular force on atom i is
//     centre_force[i]  +  sum over angular neighbours i' of  leg_force[i'][slot of i in the records of i']
// and the slot is found by scanning the compact id list of i' (<= capA ints, one or two cache lines).
// =============================================================================================
__global__ __launch_bounds__(64 * kWavesPerGroup) void ani_radial_backward(const AniParams* __restrict__ P,
                                                          const int* __restrict__ species,
                                                          const float4* __restrict__ nbr, int cap, int cap_angular,
                                                          const int* __restrict__ cnt_pos,   // rows and counts by position of the walk
                                                          const float* __restrict__ radial_grad, int ld_radial,
                                                          const int* __restrict__ ids,
                                                          const float4* __restrict__ leg_force,
                                                          const float4* __restrict__ centre_force,
                                                          const int* __restrict__ order,     // atoms in cell order, or NULL
                                                          float* __restrict__ pos_grad, int lds_per_wave, int w0, int nw) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    float* lds = (float*)(lds_raw + (size_t)wave_in_group() * lds_per_wave);
    const int lane = lane_id();
    // This kernel gathers rows of its atom's NEIGHBOURS (gradient rows, id rows, leg forces).  Atoms are walked in
    // cell order, an XCD-contiguous stretch per XCD, so that those rows are fetched into one L2 instead of eight.
    const int wl = order ? xcd_contiguous_wave_id() : wave_global_id();      // this launch covers positions [w0, w0 + nw)
    if (wl >= nw) return;
    const int w = w0 + wl;
    int i = order ? order[w] : w;
    if ((unsigned)i >= (unsigned)P->N) i = w;              // (a void grid build leaves no valid order: stay in bounds)
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
    const int counts = cnt_pos[w];
    int raw_a, raw_ro, packed_species;
    unpack_cnt_pos(counts, raw_a, raw_ro, packed_species);
    clamp_counts(raw_a, raw_ro, cap, cap_angular, na, nro);
    const int total = na + nro;
    const float4* row = nbr + (size_t)w * cap;
    const float inv_rcr = P->inv_rcr;
    const int si = species[i];

    const float* gi = radial_grad + (size_t)i * ld_radial;
    for (int q = lane; q < width; q += 64) g_own[q] = gi[q];
    for (int e = lane; e < total; e += 64) {
        const float4 rec = row[e];
        const int word = __float_as_int(rec.w);
        const float r = fast_sqrt(rec.x * rec.x + rec.y * rec.y + rec.z * rec.z);
        const float rinv = fast_rcp(r);
        float sn, cs;
        sincospi_unit(r * inv_rcr, sn, cs);
        nb_r[e] = r;
        nb_fc[e] = 0.5f * cs + 0.5f;
        nb_dfc[e] = -(0.5f * kPi * inv_rcr) * sn;
        nb_ux[e] = rec.x * rinv; nb_uy[e] = rec.y * rinv; nb_uz[e] = rec.z * rinv;
        nb_sp[e] = word >> kTagShift;
        nb_j[e] = word & kIdMask;
    }
    wave_fence();

    // Common shape (capA = 32, at most 64 angular neighbours): the id row of this lane's angular neighbour is
    // requested now and consumed after the radial loop, which hides the latency of the reverse lookup.
    const bool early = cap_angular == 32 && na <= 64;
    int4 idv[8];
    int ip_early = 0;
    if (early) {
        ip_early = lane < na ? nb_j[lane] : i;
        const int4* idrow = reinterpret_cast<const int4*>(ids + (size_t)ip_early * 32);
#pragma unroll
        for (int q = 0; q < 8; q++) idv[q] = idrow[q];
    }

    const int KP = 1 << P->kp_shift;
    const int k = lane & (KP - 1), stream = lane >> P->kp_shift, nstreams = 64 >> P->kp_shift;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (k < nR) {
        const float ck = P->rad_c[k], rs = P->rad_rs[k], eta = P->rad_eta[k];
#pragma unroll 4                  // four gathers of neighbour gradient rows in flight (8 was slower: registers)
        for (int e = stream; e < total; e += nstreams) {
            const float sh = nb_r[e] - rs;
            const float ex = fast_exp2(ck * sh * sh);
            const float dvdr = (nb_dfc[e] - nb_fc[e] * 2.f * eta * sh) * ex;
            const float dedv = g_own[nb_sp[e] * nR + k] + radial_grad[(size_t)nb_j[e] * ld_radial + si * nR + k];
            const float sc = dedv * dvdr;
            fx -= sc * nb_ux[e]; fy -= sc * nb_uy[e]; fz -= sc * nb_uz[e];
        }
    }
    const float scale = P->radial_scale;
    fx *= scale; fy *= scale; fz *= scale;
    // angular legs: lane e looks itself up in the records of angular neighbour e (rows are padded with -1)
    if (early) {
        int slot = -1;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            slot = idv[q].x == i ? 4 * q : slot;
            slot = idv[q].y == i ? 4 * q + 1 : slot;
            slot = idv[q].z == i ? 4 * q + 2 : slot;
            slot = idv[q].w == i ? 4 * q + 3 : slot;
        }
        if (lane < na && slot >= 0) {
            const float4 f = leg_force[(size_t)ip_early * 32 + slot];
            fx += f.x; fy += f.y; fz += f.z;
        }
    } else
    for (int e = lane; e < na; e += 64) {
        const int ip = nb_j[e];
        const int4* idrow = reinterpret_cast<const int4*>(ids + (size_t)ip * cap_angular);
        int k = -1;
        for (int q = 0; q < cap_angular; q += 4) {
            const int4 v = idrow[q >> 2];
            k = v.x == i ? q : k;
            k = v.y == i ? q + 1 : k;
            k = v.z == i ? q + 2 : k;
            k = v.w == i ? q + 3 : k;
        }
        if (k >= 0) {
            const float4 f = leg_force[(size_t)ip * cap_angular + k];
            fx += f.x; fy += f.y; fz += f.z;
        }
    }
    fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
    if (lane == 0) {
        if (na >= 2) {
            const float4 c = centre_force[i];
            fx += c.x; fy += c.y; fz += c.z;
        }
        pos_grad[3 * i] = fx;
        pos_grad[3 * i + 1] = fy;
        pos_grad[3 * i + 2] = fz;
    }
}

// ---------------------------------------------------------------------------------------------
// Angular forward.
//
// LDS per wave: sorted neighbour records, the atom's output row in canonical padded order
// [bucket][a][z], and the per-batch factor staging area.
//
// phase 2 ownership rules (triples of a batch are in bucket-major order, so every bucket is one
// contiguous run): stream s = lanes [s*NFRP, (s+1)*NFRP) owns the CH = NFRP consecutive triples
// [s*CH, (s+1)*CH) of the batch and lane (s, a) accumulates acc[z] += R_a * Z_z.  A run that starts
// and ends strictly inside a stream's chunk belongs to that stream alone and is added to the LDS
// row with a plain read-modify-write.  The first and the last run of each chunk may continue in a
// neighbouring stream; see "edge runs" below.  Batches whose 64 triples share one bucket (the common
// case for few-species systems) skip all of that: accumulate, fold the streams with xor-shuffles,
// one read-modify-write.
// ---------------------------------------------------------------------------------------------
template <int NFRP, int NFZP>
struct FwdLayout {
    static constexpr int CH = NFRP;                       // triples per stream per batch
    static constexpr int NSTREAM = 64 / NFRP;
    static constexpr int SR = NFRP * (CH + 1);            // stream stride of facR in floats (bank-conflict free)
    static constexpr int BLK = NFRP * NFZP;               // padded canonical block of one bucket
};

template <int NFRP, int NFZP>
__host__ __device__ inline size_t ang_fwd_lds_bytes(int capA, int NB) {
    using L = FwdLayout<NFRP, NFZP>;
    size_t b = (size_t)capA * 2 * sizeof(float4);
    b += (size_t)(NB + 1) * L::BLK * sizeof(float);       // + one dummy block that swallows masked-off stores
    b += (size_t)L::NSTREAM * L::SR * sizeof(float) + (size_t)64 * NFZP * sizeof(float) + 64 * sizeof(int);
    return b;
}

template <int NFZP>
__device__ __forceinline__ void row_add(float* dst, const float (&v)[NFZP]) {
#pragma unroll
    for (int z = 0; z < NFZP; z += 4) {
        float4 cur = *reinterpret_cast<float4*>(dst + z);
        cur.x += v[z]; cur.y += v[z + 1]; cur.z += v[z + 2]; cur.w += v[z + 3];
        *reinterpret_cast<float4*>(dst + z) = cur;
    }
}

template <bool TORCHANI, int NFRP, int NFZP>
__global__ __launch_bounds__(64 * kWavesPerGroup) void ani_angular_forward(const AniParams* __restrict__ P, int cap, int capA,
                                                          const float4* __restrict__ recA_g,
                                                          const float4* __restrict__ recB_g,
                                                          const int* __restrict__ tri_g,
                                                          const int* __restrict__ cnt_a,
                                                          const int* __restrict__ cnt_ro,
                                                          float* __restrict__ angular, int ld_angular, int lds_per_wave) {
    using L = FwdLayout<NFRP, NFZP>;
    constexpr int CH = L::CH, NSTREAM = L::NSTREAM, SR = L::SR, BLK = L::BLK;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int i = wave_global_id(), lane = lane_id();
    const int NB = P->NB, nA = P->nA, nFR = P->nFR, nFZ = P->nFZ;
    if (i >= P->N) return;

    char* cursor = lds_raw + (size_t)wave_in_group() * lds_per_wave;
    float4* recA = (float4*)cursor;       cursor += (size_t)capA * sizeof(float4);
    float4* recB = (float4*)cursor;       cursor += (size_t)capA * sizeof(float4);
    float* row = (float*)cursor;          cursor += (size_t)(NB + 1) * BLK * sizeof(float);
    float* facR = (float*)cursor;         cursor += (size_t)NSTREAM * SR * sizeof(float);
    float* facZ = (float*)cursor;         cursor += (size_t)64 * NFZP * sizeof(float);
    int* facB = (int*)cursor;

    int n, nro;
    clamp_counts(cnt_a[i], cnt_ro[i], cap, capA, n, nro);
    const int T = (n * (n - 1)) / 2;
    const int* tri = tri_g + (size_t)i * triples_capacity(capA);
    int word = lane < T ? tri[lane] : 0;                   // first batch of triple words, in flight early
    load_angular_records(recA_g + (size_t)i * capA, recB_g + (size_t)i * capA, n, recA, recB);
    const int rowlen = NB * BLK;
    for (int q = lane; q < rowlen; q += 64) row[q] = 0.f;

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
    wave_fence();

    const int a2 = lane & (NFRP - 1), stream = lane / NFRP;
    for (int base = 0; base < T; base += 64) {
        // ---------------- phase 1: lane = triple ----------------
        const int t = base + lane;
        const int next_word = (t + 64 < T) ? tri[t + 64] : 0;     // prefetch the next batch
        int bucket = -1;
        if (t < T) {
            const int p = word & 0xff, q = (word >> 8) & 0xff;
            bucket = word >> 16;
            const float4 A = recA[p], B = recA[q];
            const float4 A2 = recB[p], B2 = recB[q];
            const TripleGeom g = triple_geometry<TORCHANI>(A, A2, B, B2);
            float* dstR = facR + (lane / CH) * SR + (lane % CH) * NFRP;
#pragma unroll
            for (int a = 0; a < NFRP; a += 4) {
                float4 v;
                float sh;
                sh = g.rbar - frs[a];     v.x = fast_exp2(frc[a] * sh * sh);
                sh = g.rbar - frs[a + 1]; v.y = fast_exp2(frc[a + 1] * sh * sh);
                sh = g.rbar - frs[a + 2]; v.z = fast_exp2(frc[a + 2] * sh * sh);
                sh = g.rbar - frs[a + 3]; v.w = fast_exp2(frc[a + 3] * sh * sh);
                *reinterpret_cast<float4*>(dstR + a) = v;
            }
            float zv[NFZP];
#pragma unroll
            for (int z = 0; z < NFZP; z++) {
                const float x = fmaxf(1.0f + (g.c * zc[z] + g.s * zs[z]), 1e-30f);   // 1 + cos(theta - ths)
                zv[z] = g.fcfc * fast_exp2(zz[z] * fast_log2(x));
            }
#pragma unroll
            for (int z = 0; z < NFZP; z += 4)
                *reinterpret_cast<float4*>(facZ + lane * NFZP + z) = make_float4(zv[z], zv[z + 1], zv[z + 2], zv[z + 3]);
            facB[lane] = bucket;
        }
        word = next_word;
        const int b0 = __shfl(bucket, 0, 64);
        const bool uniform = __all(bucket == b0);          // implies all 64 lanes hold a triple
        wave_fence();
        // ---------------- phase 2: lane = (stream, a) ----------------
        const float* srcR = facR + stream * SR + a2;
        if (uniform) {
            float acc[NFZP];
#pragma unroll
            for (int z = 0; z < NFZP; z++) acc[z] = 0.f;
#pragma unroll
            for (int u = 0; u < CH; u++) {
                const float R = srcR[u * NFRP];
                const float* Z = facZ + (stream * CH + u) * NFZP;
#pragma unroll
                for (int z = 0; z < NFZP; z++) acc[z] += R * Z[z];
            }
#pragma unroll
            for (int off = NFRP; off < 64; off <<= 1) {
#pragma unroll
                for (int z = 0; z < NFZP; z++) acc[z] += __shfl_xor(acc[z], off, 64);
            }
            if (stream == 0) row_add<NFZP>(row + b0 * BLK + a2 * NFZP, acc);
        } else {
            // general path: pull the stream's whole chunk into registers first (one LDS wait), then find
            // the runs with register compares only.  An interior run (neither first nor last of the
            // chunk) is the ONLY contribution its bucket ever receives -- buckets are contiguous in the
            // atom's triple order -- so it is stored, not accumulated.
            const int count = min(64, T - base);
            int bk[CH];
            float Rv[CH];
            float Zv[CH][NFZP];
#pragma unroll
            for (int u = 0; u < CH; u += 4) {
                const int4 b4 = *reinterpret_cast<const int4*>(facB + stream * CH + u);
                bk[u] = b4.x; bk[u + 1] = b4.y; bk[u + 2] = b4.z; bk[u + 3] = b4.w;
            }
#pragma unroll
            for (int u = 0; u < CH; u++) {
                Rv[u] = srcR[u * NFRP];
#pragma unroll
                for (int z = 0; z < NFZP; z += 4) {
                    const float4 z4 = *reinterpret_cast<const float4*>(facZ + (stream * CH + u) * NFZP + z);
                    Zv[u][z] = z4.x; Zv[u][z + 1] = z4.y; Zv[u][z + 2] = z4.z; Zv[u][z + 3] = z4.w;
                }
            }
            // Everything below is straight-line, select-based code: the streams diverge at almost every
            // step for many-species systems, and exec-mask juggling was costing more than the arithmetic.
            // Stores that must not happen go to a dummy block behind the row (bucket index NB).
            float acc[NFZP], head[NFZP];
#pragma unroll
            for (int z = 0; z < NFZP; z++) { acc[z] = 0.f; head[z] = 0.f; }
            int cur = -1, hb = -1;
            bool first = true;
#pragma unroll
            for (int u = 0; u < CH; u++) {
                const bool valid = stream * CH + u < count;
                const int bkt = valid ? bk[u] : cur;
                const bool change = bkt != cur;
                const bool closes = change && cur >= 0;          // a run ends here
                const bool interior = closes && !first;
                const bool is_head = closes && first;
                float* dst = row + (interior ? cur : NB) * BLK + a2 * NFZP;
#pragma unroll
                for (int z = 0; z < NFZP; z += 4)
                    *reinterpret_cast<float4*>(dst + z) = make_float4(acc[z], acc[z + 1], acc[z + 2], acc[z + 3]);
                hb = is_head ? cur : hb;
                first = first && !closes;
                // slots past the end of the triple list hold stale LDS (possibly NaN bit patterns): select, don't scale
#pragma unroll
                for (int z = 0; z < NFZP; z++) {
                    head[z] = is_head ? acc[z] : head[z];
                    const float term = valid ? Rv[u] * Zv[u][z] : 0.f;
                    acc[z] = (change ? 0.f : acc[z]) + term;
                }
                cur = bkt;
            }
            // after the chunk: the open run is the head if it is the only one, else the tail
            const bool only = first;                               // chunk holds a single run (or nothing)
            hb = (only && cur >= 0) ? cur : hb;
            const int tb = (!only && cur >= 0) ? cur : -1;
#pragma unroll
            for (int z = 0; z < NFZP; z++) head[z] = only ? acc[z] : head[z];
            // edge runs.  The head run of a chunk may continue the previous stream's last run, and the last
            // run may continue into the next stream (runs are contiguous), possibly through several
            // single-run streams.  Each stream hands OUT_s forward: its tail if it has one, else its head plus
            // what it received -- a first-order recurrence OUT_s = a_s + c_s * OUT_{s-1} (c_s in {0,1}) that a
            // log-step scan over the streams resolves with shuffles.  The piece of a run that ends it adds
            // the run's total to the row; every run is therefore added exactly once, by one owner.
            const bool has_tail = tb >= 0;
            const int lastb = has_tail ? tb : hb;
            int prev_last = __shfl_up(lastb, NFRP, 64);
            int next_head = __shfl_down(hb, NFRP, 64);
            prev_last = stream == 0 ? -2 : prev_last;
            next_head = stream == NSTREAM - 1 ? -2 : next_head;
            const bool link = hb >= 0 && hb == prev_last;           // my head continues the previous stream
            float av[NFZP];
#pragma unroll
            for (int z = 0; z < NFZP; z++) av[z] = has_tail ? acc[z] : head[z];
            int cv = (!has_tail && link) ? 1 : 0;
#pragma unroll
            for (int off = NFRP; off < 64; off <<= 1) {
                const int cup = __shfl_up(cv, off, 64);
                const bool take = lane >= off && cv != 0;
#pragma unroll
                for (int z = 0; z < NFZP; z++) {
                    const float up = __shfl_up(av[z], off, 64);
                    av[z] += take ? up : 0.f;
                }
                cv = (lane >= off) ? (cv & cup) : cv;
            }
            // av is now OUT_s; what I receive is OUT_{s-1} when linked
            float total[NFZP];
#pragma unroll
            for (int z = 0; z < NFZP; z++) {
                const float in = __shfl_up(av[z], NFRP, 64);
                total[z] = head[z] + (link ? in : 0.f);
            }
            const bool head_ends_here = hb >= 0 && (has_tail || next_head != hb);
            const bool tail_ends_here = has_tail && next_head != tb;
            row_add<NFZP>(row + (head_ends_here ? hb : NB) * BLK + a2 * NFZP, total);
            row_add<NFZP>(row + (tail_ends_here ? tb : NB) * BLK + a2 * NFZP, acc);
        }
        wave_fence();
    }

    // ---------------- epilogue: canonical LDS row -> reference column order, coalesced rows ----------------
    float* out = angular + (size_t)i * ld_angular;
    if (nA <= 32) {                                        // two buckets per pass
        const int m = lane & 31, half = lane >> 5;
        const bool live = m < nA;
        const int c = live ? P->c_of_m[m] : 0;             // canonical slot a*NFZP+z of function m
        const float sc = live ? P->scale_m[m] : 0.f;
        for (int bk0 = half; bk0 < NB; bk0 += 16) {        // 8 LDS reads in flight, then 8 coalesced stores
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int bk = bk0 + 2 * k;
                v[k] = (live && bk < NB) ? row[bk * BLK + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int bk = bk0 + 2 * k;
                if (live && bk < NB) out[bk * nA + m] = v[k] * sc;
            }
        }
    } else {
        for (int bk = 0; bk < NB; bk++)
            for (int m = lane; m < nA; m += 64) out[bk * nA + m] = row[bk * BLK + P->c_of_m[m]] * P->scale_m[m];
    }
}

// ---------------------------------------------------------------------------------------------
// Angular forward, chunked view (the default when NB <= 64).
//
// Same phase 1 as above, but the triples are visited through a PADDED view of the builder's list: every bucket is
// rounded up to whole chunks of CH = NFRP triples (the builder publishes the bucket offsets; padding slots produce
// zero factors).  A stream's chunk then belongs to ONE bucket, and phase 2 collapses to: accumulate the chunk
// (32 FMAs per lane), add up adjacent streams that hold the same bucket (a 3-step segmented scan over the streams,
// shuffles only), one read-modify-write of the LDS row by the last stream of each bucket.  No run detection, no
// head/tail bookkeeping, no dummy stores.  Costs ~40 % more phase-1 slots for a many-species system (5.5 triples
// per bucket padded to 8) and nothing for water; phase 2 drops from ~270 to ~75 vector instructions per batch.
// ---------------------------------------------------------------------------------------------
template <int NFRP, int NFZP>
__host__ __device__ inline size_t ang_fwd_chunked_lds_bytes(int capA, int NB) {
    using L = FwdLayout<NFRP, NFZP>;
    const int max_chunks = (capA * (capA - 1) / 2) / L::CH + NB + 1;
    size_t b = (size_t)capA * 2 * sizeof(float4);
    b += (size_t)NB * L::BLK * sizeof(float);
    b += (size_t)L::NSTREAM * L::SR * sizeof(float) + (size_t)64 * NFZP * sizeof(float);
    b += (size_t)(2 * (NB + 1) + max_chunks + 3) / 4 * 4 * sizeof(int);        // bucket offsets, chunk starts, chunk -> bucket
    return b;
}

template <bool TORCHANI, int NFRP, int NFZP>
__global__ __launch_bounds__(64 * kWavesPerGroup) void ani_angular_forward_chunked(
    const AniParams* __restrict__ P, int cap, int capA, const float4* __restrict__ recA_g, const float4* __restrict__ recB_g,
    const int* __restrict__ tri_g, const int* __restrict__ cnt_a, const int* __restrict__ cnt_ro, float* __restrict__ angular,
    int ld_angular, int lds_per_wave) {
    using L = FwdLayout<NFRP, NFZP>;
    constexpr int CH = L::CH, NSTREAM = L::NSTREAM, SR = L::SR, BLK = L::BLK;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int i = wave_global_id(), lane = lane_id();
    const int NB = P->NB, nA = P->nA, nFR = P->nFR, nFZ = P->nFZ;
    if (i >= P->N) return;

    char* cursor = lds_raw + (size_t)wave_in_group() * lds_per_wave;
    float4* recA = (float4*)cursor;       cursor += (size_t)capA * sizeof(float4);
    float4* recB = (float4*)cursor;       cursor += (size_t)capA * sizeof(float4);
    float* row = (float*)cursor;          cursor += (size_t)NB * BLK * sizeof(float);
    float* facR = (float*)cursor;         cursor += (size_t)NSTREAM * SR * sizeof(float);
    float* facZ = (float*)cursor;         cursor += (size_t)64 * NFZP * sizeof(float);
    int* boff = (int*)cursor;             // [NB + 1] first triple of bucket b in the builder's list
    int* cstart = boff + NB + 1;          // [NB + 1] first chunk of bucket b in the padded view
    int* cbkt = cstart + NB + 1;          // [chunks] bucket of chunk c

    int n, nro;
    clamp_counts(cnt_a[i], cnt_ro[i], cap, capA, n, nro);
    const int* tri = tri_g + (size_t)i * triples_capacity(capA);
    // bucket offsets -> chunks per bucket -> chunk starts (wave scan; NB <= 64)
    const int* boff_g = P->bucket_offsets + (size_t)i * (NB + 1);
    const int my_lo = lane <= NB ? boff_g[lane] : 0;
    const int my_hi = lane < NB ? boff_g[lane + 1] : my_lo;
    load_angular_records(recA_g + (size_t)i * capA, recB_g + (size_t)i * capA, n, recA, recB);
    const int rowlen = NB * BLK;
    for (int q = lane; q < rowlen; q += 64) row[q] = 0.f;
    const int my_chunks = n >= 2 ? (my_hi - my_lo + CH - 1) / CH : 0;
    int incl = my_chunks;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    const int my_first = incl - my_chunks;
    if (lane <= NB) { boff[lane] = my_lo; cstart[lane] = my_first; }
    for (int c = 0; c < my_chunks; c++) cbkt[my_first + c] = lane;        // (lanes >= NB have no chunks)
    const int chunks = __shfl(incl, 63, 64);

    float frc[NFRP], frs[NFRP], zz[NFZP], zc[NFZP], zs[NFZP];
#pragma unroll
    for (int a = 0; a < NFRP; a++) { frc[a] = a < nFR ? P->fr_c[a] : 0.f; frs[a] = a < nFR ? P->fr_rs[a] : 0.f; }
#pragma unroll
    for (int z = 0; z < NFZP; z++) {
        zz[z] = z < nFZ ? P->fz_zeta[z] : 1.f;
        zc[z] = z < nFZ ? P->fz_cos[z] : 0.f;
        zs[z] = z < nFZ ? P->fz_sin[z] : 0.f;
    }
    wave_fence();

    const int a2 = lane & (NFRP - 1), stream = lane / NFRP;
    // triple of (chunk, slot) in the builder's list, or -1 for padding
    auto triple_of = [&](int chunk, int u) {
        if (chunk >= chunks) return -1;
        const int b = cbkt[chunk];
        const int t = boff[b] + (chunk - cstart[b]) * CH + u;
        return t < boff[b + 1] ? t : -1;
    };
    int t_mine = triple_of(lane / CH, lane % CH);
    int word = t_mine >= 0 ? tri[t_mine] : 0;
    for (int cb = 0; cb < chunks; cb += NSTREAM) {
        // ---------------- phase 1: lane = (chunk, slot) ----------------
        const int t_next = triple_of(cb + NSTREAM + lane / CH, lane % CH);
        const int next_word = t_next >= 0 ? tri[t_next] : 0;              // next batch in flight
        float* dstR = facR + (lane / CH) * SR + (lane % CH) * NFRP;
        if (t_mine >= 0) {
            const int p = word & 0xff, q = (word >> 8) & 0xff;
            const float4 A = recA[p], B = recA[q];
            const float4 A2 = recB[p], B2 = recB[q];
            const TripleGeom g = triple_geometry<TORCHANI>(A, A2, B, B2);
#pragma unroll
            for (int a = 0; a < NFRP; a += 4) {
                float4 v;
                float sh;
                sh = g.rbar - frs[a];     v.x = fast_exp2(frc[a] * sh * sh);
                sh = g.rbar - frs[a + 1]; v.y = fast_exp2(frc[a + 1] * sh * sh);
                sh = g.rbar - frs[a + 2]; v.z = fast_exp2(frc[a + 2] * sh * sh);
                sh = g.rbar - frs[a + 3]; v.w = fast_exp2(frc[a + 3] * sh * sh);
                *reinterpret_cast<float4*>(dstR + a) = v;
            }
            float zv[NFZP];
#pragma unroll
            for (int z = 0; z < NFZP; z++) {
                const float x = fmaxf(1.0f + (g.c * zc[z] + g.s * zs[z]), 1e-30f);   // 1 + cos(theta - ths)
                zv[z] = g.fcfc * fast_exp2(zz[z] * fast_log2(x));
            }
#pragma unroll
            for (int z = 0; z < NFZP; z += 4)
                *reinterpret_cast<float4*>(facZ + lane * NFZP + z) = make_float4(zv[z], zv[z + 1], zv[z + 2], zv[z + 3]);
        } else {                                                           // padding: contributes nothing
#pragma unroll
            for (int z = 0; z < NFZP; z += 4)
                *reinterpret_cast<float4*>(facZ + lane * NFZP + z) = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int a = 0; a < NFRP; a += 4) *reinterpret_cast<float4*>(dstR + a) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        word = next_word;
        t_mine = t_next;
        wave_fence();
        // ---------------- phase 2: lane = (stream, a), one bucket per stream ----------------
        {
            const int chunk = cb + stream;
            const int bs = chunk < chunks ? cbkt[chunk] : -1;
            const float* srcR = facR + stream * SR + a2;
            float acc[NFZP];
#pragma unroll
            for (int z = 0; z < NFZP; z++) acc[z] = 0.f;
#pragma unroll
            for (int u = 0; u < CH; u++) {
                const float R = srcR[u * NFRP];
                const float* Z = facZ + (stream * CH + u) * NFZP;
#pragma unroll
                for (int z = 0; z < NFZP; z += 4) {
                    const float4 z4 = *reinterpret_cast<const float4*>(Z + z);
                    acc[z] += R * z4.x; acc[z + 1] += R * z4.y; acc[z + 2] += R * z4.z; acc[z + 3] += R * z4.w;
                }
            }
            // streams holding the same bucket are adjacent: segmented inclusive scan over the streams
#pragma unroll
            for (int off = NFRP; off < 64; off <<= 1) {
                const int ub = __shfl_up(bs, off, 64);
                const bool take = lane >= off && ub == bs;
#pragma unroll
                for (int z = 0; z < NFZP; z++) {
                    const float up = __shfl_up(acc[z], off, 64);
                    acc[z] += take ? up : 0.f;
                }
            }
            const int nb_next = __shfl_down(bs, NFRP, 64);
            const bool closes = bs >= 0 && (stream == NSTREAM - 1 || nb_next != bs);   // last stream of its bucket in this batch
            if (closes) row_add<NFZP>(row + bs * BLK + a2 * NFZP, acc);
        }
        wave_fence();
    }

    // ---------------- epilogue: canonical LDS row -> reference column order, coalesced rows ----------------
    float* out = angular + (size_t)i * ld_angular;
    if (nA <= 32) {                                        // two buckets per pass
        const int m = lane & 31, half = lane >> 5;
        const bool live = m < nA;
        const int c = live ? P->c_of_m[m] : 0;             // canonical slot a*NFZP+z of function m
        const float sc = live ? P->scale_m[m] : 0.f;
        for (int bk0 = half; bk0 < NB; bk0 += 16) {        // 8 LDS reads in flight, then 8 coalesced stores
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int bk = bk0 + 2 * k;
                v[k] = (live && bk < NB) ? row[bk * BLK + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int bk = bk0 + 2 * k;
                if (live && bk < NB) out[bk * nA + m] = v[k] * sc;
            }
        }
    } else {
        for (int bk = 0; bk < NB; bk++)
            for (int m = lane; m < nA; m += 64) out[bk * nA + m] = row[bk * BLK + P->c_of_m[m]] * P->scale_m[m];
    }
}

// ---------------------------------------------------------------------------------------------
// Angular backward.                                                              ref :265-353
//
// lane = triple (p < q sorted neighbour slots).  The force a triple puts on its two leg atoms is
// written to an LDS "pair matrix" M[p][q] (force on p) / M[q][p] (force on q): every entry has
// exactly one writer, so there are no atomics and no reductions across lanes; afterwards lane e sums
// row e.  The centre atom receives minus the total.  The host sizes `tile` (<= 32, the matrix edge)
// from the largest angular neighbour count it has seen, so an atom is normally one tile and takes
// its triples straight from the builder's word list; an atom that outgrows the tile is processed
// tile pair by tile pair (enumerating pairs itself, off-diagonal tiles in two passes).
// ---------------------------------------------------------------------------------------------
// `compact`: no atom of the system has more than `tile` angular neighbours (the host knows from check()), so the
// tile-pair fallback cannot run: beta is stored once per unordered pair and the per-slot force accumulators
// reuse the gradient block's space.
template <int NFRP, int NFZP>
__host__ __device__ inline size_t ang_bwd_lds_bytes(int capA, int NB, int tile, bool compact) {
    size_t b = (size_t)capA * 2 * sizeof(float4);
    if (!compact) b += (size_t)capA * 4 * sizeof(float);
    b += (size_t)NB * NFRP * NFZP * sizeof(float);
    if (compact) b += ((size_t)tile * (tile + 1) + (size_t)tile * (tile - 1) / 2) * sizeof(float);   // alpha square + beta triangle
    else b += (size_t)2 * tile * (tile + 1) * sizeof(float);                                        // {alpha, beta} square
    b = std::max(b, (size_t)capA * (2 * sizeof(float4) + 4 * sizeof(float)) + (size_t)NB * NFRP * NFZP * sizeof(float));
    return b;
}

// Forces of one triple on its two leg atoms, given the scaled upstream-gradient block, as three scalars:
//     F_p = alpha_p * A + beta * B,      F_q = alpha_q * B + beta * A        (A, B = displacements of the legs)
// -- the pair matrix then holds two floats per entry instead of a vector, which is what lets 15 instead of 11
// waves share a CU's LDS (the kernel is latency bound at that occupancy: time x waves is constant).
template <bool TORCHANI, int NFRP, int NFZP>
__device__ __forceinline__ void triple_forces(const float4& A, const float4& A2, const float4& B, const float4& B2,
                                              const float* Gb, const float (&frc)[NFRP], const float (&frs)[NFRP],
                                              const float (&fre)[NFRP], const float (&zz)[NFZP], const float (&zc)[NFZP],
                                              const float (&zs)[NFZP], float& alpha_p, float& alpha_q, float& beta) {
    const TripleGeom g = triple_geometry<TORCHANI>(A, A2, B, B2);
    float R[NFRP], dR[NFRP];
#pragma unroll
    for (int a = 0; a < NFRP; a++) {
        const float sh = g.rbar - frs[a];
        R[a] = fast_exp2(frc[a] * sh * sh);
        dR[a] = -fre[a] * sh * R[a];       // d/dr_ij of exp(-eta (rbar-Rs)^2): rbar carries 1/2 (ref :306)
    }
    // contract the gradient block with R and dR:  U_z = sum_a G[a][z] R_a,  V_z = sum_a G[a][z] dR_a
    float U[NFZP], V[NFZP];
#pragma unroll
    for (int z = 0; z < NFZP; z++) { U[z] = 0.f; V[z] = 0.f; }
#pragma unroll
    for (int a = 0; a < NFRP; a++) {
#pragma unroll
        for (int z = 0; z < NFZP; z += 4) {
            const float4 gv = *reinterpret_cast<const float4*>(Gb + a * NFZP + z);
            U[z] += gv.x * R[a];     V[z] += gv.x * dR[a];
            U[z + 1] += gv.y * R[a]; V[z + 1] += gv.y * dR[a];
            U[z + 2] += gv.z * R[a]; V[z + 2] += gv.z * dR[a];
            U[z + 3] += gv.w * R[a]; V[z + 3] += gv.w * dR[a];
        }
    }
    float S0 = 0.f, Sr = 0.f, Sth = 0.f;
#pragma unroll
    for (int z = 0; z < NFZP; z++) {
        const float cz = g.c * zc[z] + g.s * zs[z];    // cos(theta - ths)
        const float sz = g.s * zc[z] - g.c * zs[z];    // sin(theta - ths)
        const float x = fmaxf(1.0f + cz, 1e-30f);      // keeps 0 * -inf out of the zeta == 1 corner
        const float lg = fast_log2(x);
        const float Z = fast_exp2(zz[z] * lg);                          // (1+cos)^zeta
        const float dZ = -zz[z] * fast_exp2((zz[z] - 1.0f) * lg) * sz;   // d/dtheta           ref :337
        S0 += U[z] * Z;
        Sr += V[z] * Z;
        Sth += U[z] * dZ;
    }
    // three routes of the chain rule (ref :311-348), already summed over the functions m
    const float t1 = A2.y * B2.x * S0 + g.fcfc * Sr;   // through r_ij   (A2.y = dfc_ij, B2.x = fc_ik)
    const float t2 = A2.x * B2.y * S0 + g.fcfc * Sr;   // through r_ik
    const float t3 = g.fcfc * Sth;                     // through theta
    // angle gradients (ref :410-433): dtheta/d(dot') = -damp / sin(theta)
    const float dot = A.x * B.x + A.y * B.y + A.z * B.z;
    const float iprod = A2.z * B2.z;
    const float damp = TORCHANI ? 0.95f : 1.0f;
    const float dadd = -damp * fast_rcp(g.s) * iprod * t3;
    const float ka = dot * A2.z * A2.z, kb = dot * B2.z * B2.z;
    const float s1 = t1 * A2.z, s2 = t2 * B2.z;
    // F_p = s1 A + dadd (B - ka A),  F_q = s2 B + dadd (A - kb B)
    alpha_p = s1 - dadd * ka;
    alpha_q = s2 - dadd * kb;
    beta = dadd;
}

template <bool TORCHANI, int NFRP, int NFZP>
__global__ __launch_bounds__(64 * kWavesPerGroup) void ani_angular_backward(const AniParams* __restrict__ P, int cap, int capA, int tile,
                                                           const float4* __restrict__ recA_g,
                                                           const float4* __restrict__ recB_g,
                                                           const int* __restrict__ tri_g,
                                                           const int* __restrict__ cnt_a,
                                                           const int* __restrict__ cnt_ro,
                                                           const float* __restrict__ angular_grad, int ld_angular,
                                                           float4* __restrict__ leg_force,      // [N][capA]
                                                           float4* __restrict__ centre_force,   // [N]
                                                           int lds_per_wave, int compact) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int i = wave_global_id(), lane = lane_id();
    const int S = P->S, NB = P->NB, nA = P->nA, nFR = P->nFR, nFZ = P->nFZ;
    constexpr int BLK = NFRP * NFZP;
    const int tstride = tile + 1;
    if (i >= P->N) return;

    char* cursor = lds_raw + (size_t)wave_in_group() * lds_per_wave;
    float4* recA = (float4*)cursor;       cursor += (size_t)capA * sizeof(float4);
    float4* recB = (float4*)cursor;       cursor += (size_t)capA * sizeof(float4);
    float* facc = (float*)cursor;         // per-slot forces; compact layout: the space of `grow`, dead by then
    if (!compact) cursor += (size_t)capA * 4 * sizeof(float);
    float* grow = (float*)cursor;         cursor += (size_t)NB * BLK * sizeof(float);   // scaled upstream gradient row
    float2* M = (float2*)cursor;          // tile-pair fallback: [tile][tile + 1] {alpha of the row's slot, beta} per ordered pair
    float* Ma = (float*)cursor;           // common path: alpha[tile][tile + 1] ...
    float* Mb = Ma + tile * tstride;      // ... and beta, once per unordered pair (p < q), triangular

    int n, nro;
    clamp_counts(cnt_a[i], cnt_ro[i], cap, capA, n, nro);
    if (n < 2) {                             // no triples (wave-uniform): a lone leg carries no force
        if (n == 1 && lane == 0) leg_force[(size_t)i * capA] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int T = (n * (n - 1)) / 2;
    const int* tri = tri_g + (size_t)i * triples_capacity(capA);
    int word = lane < T ? tri[lane] : 0;

    // upstream gradient row -> canonical [bucket][a][z] order, pre-multiplied by 2^(1-zeta).
    // All global loads of a group are issued before LDS is touched (a load-store-load loop waits every trip).
    {
        const float* g = angular_grad + (size_t)i * ld_angular;
        if (nA <= 32) {
            const int m = lane & 31, half = lane >> 5;
            const bool live = m < nA;
            const int c = live ? P->c_of_m[m] : 0;
            const float sc = live ? P->scale_m[m] : 0.f;
            if (BLK != nA)
                for (int q = lane; q < NB * BLK; q += 64) grow[q] = 0.f;      // padded slots must read as zero
            for (int bk0 = half; bk0 < NB; bk0 += 16) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int bk = bk0 + 2 * k;
                    v[k] = (live && bk < NB) ? g[bk * nA + m] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int bk = bk0 + 2 * k;
                    if (live && bk < NB) grow[bk * BLK + c] = v[k] * sc;
                }
            }
        } else {
            for (int q = lane; q < NB * BLK; q += 64) grow[q] = 0.f;
            wave_fence();
            for (int bk = 0; bk < NB; bk++)
                for (int m = lane; m < nA; m += 64) grow[bk * BLK + P->c_of_m[m]] = g[bk * nA + m] * P->scale_m[m];
        }
    }
    if (!compact)                                          // (only the tile-pair fallback accumulates into facc)
        for (int q = lane; q < n * 4; q += 64) facc[q] = 0.f;
    load_angular_records(recA_g + (size_t)i * capA, recB_g + (size_t)i * capA, n, recA, recB);

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
    wave_fence();

    float* fsum = facc;                                    // where the per-slot forces end up
    if (n <= tile) {
        // ---------------- common case: one tile, triples from the builder's list ----------------
        for (int base = 0; base < T; base += 64) {
            const int t = base + lane;
            const int next_word = (t + 64 < T) ? tri[t + 64] : 0;
            if (t < T) {
                const int p = word & 0xff, q = (word >> 8) & 0xff, bucket = word >> 16;
                float ap, aq, bt;
                triple_forces<TORCHANI, NFRP, NFZP>(recA[p], recB[p], recA[q], recB[q], grow + bucket * BLK, frc, frs, fre,
                                                    zz, zc, zs, ap, aq, bt);
                Ma[p * tstride + q] = ap;
                Ma[q * tstride + p] = aq;
                Mb[p * (2 * tile - p - 1) / 2 + (q - p - 1)] = bt;            // p < q: once per unordered pair
            }
            word = next_word;
        }
        wave_fence();
        // row sums: lane (e, half) walks up to 16 columns of row e; halves folded by one shuffle
        //   F_e = (sum_x alpha[e][x]) * A_e + sum_x beta{e,x} * A_x
        const int e = lane & 31, half = lane >> 5;
        float fx = 0.f, fy = 0.f, fz = 0.f, as = 0.f;
        if (e < n) {
            const int x0 = half * 16, x1 = min(n, x0 + 16);
            const float* ma = Ma + e * tstride;
            // index of beta{e,x} in the triangle: x < e: x(2T-x-1)/2 + e-x-1 (grows by T-x-2 per step), x > e: base_e + x-e-1
            int below = x0 * (2 * tile - x0 - 1) / 2 + (e - x0 - 1);
            const int above0 = e * (2 * tile - e - 1) / 2 - e - 1;
#pragma unroll 4
            for (int x = x0; x < x1; x++) {
                const bool use = x != e;
                const int bi = x < e ? below : above0 + x;
                const float al = ma[x];
                const float b = use ? Mb[use ? bi : 0] : 0.f;
                const float4 Ax = recA[x];
                as += use ? al : 0.f;
                fx += b * Ax.x; fy += b * Ax.y; fz += b * Ax.z;
                below += tile - x - 2;
            }
        }
        as += __shfl_xor(as, 32, 64);
        if (e < n) {
            const float4 Ae = recA[e];
            const float own = half == 0 ? as : 0.f;          // counted once
            fx += own * Ae.x; fy += own * Ae.y; fz += own * Ae.z;
        }
        fx += __shfl_xor(fx, 32, 64); fy += __shfl_xor(fy, 32, 64); fz += __shfl_xor(fz, 32, 64);
        if (half == 0 && e < n) { facc[e * 4] = fx; facc[e * 4 + 1] = fy; facc[e * 4 + 2] = fz; }
        wave_fence();
    } else {
        // ---------------- an atom larger than the pair matrix: tile pairs ----------------
        // (compact layout: the host did not expect this atom -- it appeared after the last check().  Still exact:
        // the accumulators move to the end of the matrix region and the tile shrinks to what is left.)
        int ft = tile;
        if (compact) {
            const int region = tile * tstride + tile * (tile - 1) / 2;
            fsum = Ma + region - capA * 4;
            while (2 * ft * (ft + 1) > region - capA * 4) ft--;
            for (int q = lane; q < n * 4; q += 64) fsum[q] = 0.f;
            wave_fence();
        }
        const int fs = ft + 1;
        const int nblk = (n + ft - 1) / ft;
        for (int PB = 0; PB < nblk; PB++) {
            for (int QB = PB; QB < nblk; QB++) {
                const int p0 = PB * ft, q0 = QB * ft;
                const int np = min(ft, n - p0), nq = min(ft, n - q0);
                const bool diag = PB == QB;
                const int Tt = diag ? (np * (np - 1)) / 2 : np * nq;
                const int npass = diag ? 1 : 2;           // off-diagonal: forces on the p block, then on the q block
                for (int pass = 0; pass < npass; pass++) {
                    for (int t = lane; t < Tt; t += 64) {
                        int pl, ql;
                        if (diag) decode_pair(t, np, pl, ql);
                        else { pl = t / nq; ql = t - pl * nq; }
                        const int p = p0 + pl, q = q0 + ql;
                        const float4 A2 = recB[p], B2 = recB[q];
                        const int sa = __float_as_int(A2.w) >> kTagShift, sb = __float_as_int(B2.w) >> kTagShift;
                        const int bucket = sa * S - (sa * (sa - 1)) / 2 + (sb - sa);      // sorted: sa <= sb, ref :39-43
                        float ap, aq, bt;
                        triple_forces<TORCHANI, NFRP, NFZP>(recA[p], A2, recA[q], B2, grow + bucket * BLK, frc, frs, fre, zz,
                                                            zc, zs, ap, aq, bt);
                        if (pass == 0) M[pl * fs + ql] = make_float2(ap, bt);
                        if (diag || pass == 1) M[ql * fs + pl] = make_float2(aq, bt);
                    }
                    wave_fence();
                    const int e = lane & 31, half = lane >> 5;
                    const bool p_rows = diag || pass == 0;                 // rows are slots of the p block
                    const int rows = p_rows ? np : nq;
                    const int cols = diag ? np : (pass == 0 ? nq : np);
                    const int row0 = p_rows ? p0 : q0, col0 = diag ? p0 : (pass == 0 ? q0 : p0);
                    float fx = 0.f, fy = 0.f, fz = 0.f, as = 0.f;
                    if (e < rows) {
                        const int x0 = half * 16, x1 = min(cols, x0 + 16);
                        for (int x = x0; x < x1; x++) {
                            if (diag && x == e) continue;
                            const float2 ab = M[e * fs + x];
                            const float4 Ax = recA[col0 + x];
                            as += ab.x;
                            fx += ab.y * Ax.x; fy += ab.y * Ax.y; fz += ab.y * Ax.z;
                        }
                    }
                    as += __shfl_xor(as, 32, 64);
                    fx += __shfl_xor(fx, 32, 64); fy += __shfl_xor(fy, 32, 64); fz += __shfl_xor(fz, 32, 64);
                    if (half == 0 && e < rows) {
                        const int slot = row0 + e;
                        const float4 Ae = recA[slot];
                        fsum[slot * 4] += fx + as * Ae.x; fsum[slot * 4 + 1] += fy + as * Ae.y; fsum[slot * 4 + 2] += fz + as * Ae.z;
                    }
                    wave_fence();
                }
            }
        }
    }
    // No scatter: the force on leg e of this atom is parked in leg_force[i][e] (record order) and the reaction
    // on the centre in centre_force[i]; ani_radial_backward_gather, which owns position_deriv[j], picks the
    // legs up from the other side.  No atomics, bitwise reproducible forces.
    float cx = 0.f, cy = 0.f, cz = 0.f;
    float4* out = leg_force + (size_t)i * capA;
    for (int e = lane; e < n; e += 64) {
        const float fx = fsum[e * 4], fy = fsum[e * 4 + 1], fz = fsum[e * 4 + 2];
        out[e] = make_float4(fx, fy, fz, 0.f);
        cx -= fx; cy -= fy; cz -= fz;
    }
    cx = wave_sum(cx); cy = wave_sum(cy); cz = wave_sum(cz);
    if (lane == 0) centre_force[i] = make_float4(cx, cy, cz, 0.f);
}

}  // namespace nnpops
