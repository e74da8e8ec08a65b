// ani_angular_mfma.h -- angular AEV forward with the species-pair scatter done by the matrix cores.
//
// What is computed: reference src/ani/CpuANISymmetryFunctions.cpp:153-194 (angular[i][bucket][m] summed over the
// neighbour pairs of atom i), with the function set factored as R_a(rbar) x Z_z(theta) (ani_kernels.h).
//
// Why a third forward kernel.  The arithmetic of a triple is 12 transcendentals and a 8 x 4 outer product; what made
// the first two kernels slow was not that but the bookkeeping of adding the outer products of ~150 triples into 28
// species-pair blocks of the output row with 64 lanes (run detection, segmented scans, LDS read-modify-writes:
// ~270 vector instructions per batch of 64 triples against 32 FMAs of real work).  Here that scatter is the operand
// shape of one instruction:
//
//     v_mfma_f32_4x4x1_16b_f32     16 INDEPENDENT 4x4 outer products per issue, D_b[i][j] += A_b[i] * B_b[j]
//                                  (A: lane 4b+i, B: lane 4b+j, D: register i of lane 4b+j; exact fp32, vector rate)
//
// A quad of lanes (one of the 16 blocks) owns ONE species-pair bucket for the whole atom and walks that bucket's
// triples -- contiguous in the builder's bucket-major list -- one per step: A = Z_z(t), B = R_a(t) (two issues for
// a = 0..3 and 4..7).  The accumulators ARE the output block: no LDS row, no zero fill, no run logic, no shuffles,
// and the row leaves the registers as 16-byte stores.  Two sets of 16 quads cover up to 32 buckets; when fewer
// species pairs can occur in the system (water: 3) every bucket is split over K = 2, 4 or 8 quads (triples dealt
// round-robin, partial blocks added with K-1 xor-shuffles at the end), so the number of steps is
// max_b ceil(n_b / K) whatever the composition.
//
// Per atom: phase 1 (lane = triple, 64 per batch, CH triples staged per chunk) writes the 12 factors of every
// triple to LDS as one 48-byte record; phase 2 (lane = (quad, column)) reads one 8-byte and one 4-byte piece per
// step and issues two MFMAs.  Quads whose bucket is exhausted read an all-zero record.
#pragma once

#include "ani_kernels.h"

namespace nnpops {

typedef float mfma_f4 __attribute__((ext_vector_type(4)));

constexpr int kFwdSlots = 32;          // 2 sets x 16 quads

template <int NFRP, int NFZP>
__host__ __device__ inline size_t ang_fwd_mfma_lds_bytes(int capA, int CH) {
    // records | staged factors (+ the zero record) | 32 x 2 ints: the per-atom quad table of the balanced phase 2 (DYN)
    return (size_t)capA * 2 * sizeof(float4) + (size_t)(CH + 1) * (NFRP + NFZP) * sizeof(float) + 64 * sizeof(int);
}

template <int W>
__device__ __forceinline__ void lds_read_vec(const float* p, float (&v)[W]) {
    if constexpr (W == 1) {
        v[0] = p[0];
    } else if constexpr (W == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x; v[1] = t.y;
    } else {
        static_assert(W == 4, "operand width");
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
}

// 16-byte store of an output row piece.  mode 0: plain; 1: sc1 (write-through at agent scope); 2: sc0 sc1; 3: nt.  The 36 MB
// of rows a 10 000-atom launch writes otherwise sit dirty in the XCDs' L2s until the write-back at the end of the kernel:
// any of the three streaming flavours lets them drain while the kernel still runs (measured: -1 us on this kernel, -0.9 us
// on the kernel that follows; the same treatment of the neighbour build's arrays LOSES, because the next kernel reads them
// through the L2).
__device__ __forceinline__ void store_row16(float* p, const mfma_f4& v, int mode) {
    if (mode == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (mode == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (mode == 3) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ const float* lds_ptr(int byte_address) {
    return (const float*)(__attribute__((address_space(3))) const float*)(uintptr_t)(unsigned)byte_address;
}

// WPA = waves per atom.  1: a wave owns an atom (both quad sets).  2: a 128-lane workgroup owns an atom -- the two
// waves stage alternate batches of phase 1 into ONE shared staging area, meet at a barrier, and each runs the step loop of
// one quad set: same instructions in total, half the LDS per wave (twice the waves per CU) and half the latency per atom.
//
// The per-atom work is a struct so that two kernels can run it: ani_angular_forward_mfma below (records and triple
// list read back from global memory) and ani_build_forward (ani_build_forward.h: the neighbour build of the same atom
// runs first in the same workgroup and leaves them in LDS).
// UNI: every radial factor has the same eta and every angular factor the same zeta (ANI-1x / 1ccx / 2x: one EtaA, one Zeta)
// (1; 2: ... and the constants are those of the published ANI-2x set, compiled in as literals: Ani2xAngular, ani_kernels.h)
// and no factor slot is padding: 13 fewer wave-uniform constants to hold in scalar registers through phase 1 -- the kernel
// parks scalars in vector lanes when they run out (one vector instruction to park, one to fetch back).
// DYN (two waves per atom, row assembled in LDS, at most 63 buckets): the quads are dealt out PER ATOM.  With a fixed quad per species
// pair the step loop runs max_b n_b times -- 14 where the mean is 5.4 for seven equally likely species -- and the other quads
// multiply zeros.  Here the first wave finds, from the atom's bucket sizes, the smallest piece length L for which
// sum_b ceil(n_b / L) <= 32 quads, and writes a table {bucket, first triple, count, part, parts} per quad; a bucket larger than L is
// shared by consecutive quads of ONE wave (never across the two), whose partial blocks are added with a segmented shuffle before
// the row is assembled; species pairs without triples get no quad at all (the row is zero-filled first).  Steps: 14 -> ~9 for the
// 7-species liquid, 9 -> 5 for water.
template <bool TORCHANI, int NFRP, int NFZP, int WPA, int UNI = 0, bool DYN = false>
struct MfmaForward {
    static_assert(!DYN || WPA == 2, "the per-atom quad table is built by the first of two waves");
    static constexpr int NR4 = NFRP / 4, NZ4 = NFZP / 4, REC = NFRP + NFZP;
    static constexpr int NS = 2 / WPA;                         // quad sets run by this wave

    const AniParams* P;
    int capA, CH, vec_ok, ld_angular;
    float* angular;
    float4 *recA, *recB;                  // [capA] each, LDS
    float* fac;                           // [CH + 1][REC], LDS; the last record stays zero
    int* qtab;                            // [32][2], LDS (DYN): quad slot -> {first triple of the bucket | its size << 16, bucket | part << 8 | parts << 16}
    int dpart, dparts;                    // DYN: this quad's part of its bucket, and how many quads share the bucket
    float dinv;                           //      1 / dparts (quotients by it: exact for the sizes that occur, see phase 2)
    int lane, role, quad, nn;
    int NB, nA, K, logK;
    int fac_addr, zdelta, zero_addr;      // (LDS addresses of phase 2 are kept as 32-bit byte offsets: one register per quad stream)
    int sbk[NS], spart[NS];
    float frc[NFRP], frs[NFRP], zz[NFZP], zc[NFZP], zs[NFZP], zb[NFZP];   // constants of the two factor families (wave-uniform)
    float frc0, zz0, zb0;                                                 // UNI: the shared eta / zeta / bias
    GeoRadial geo;                                                        // UNI with eight radial factors: the recurrence's constants

    __device__ __forceinline__ void sync() const {
        if constexpr (WPA == 2) __syncthreads();
        else wave_fence();
    }

    // lds: this atom's area, recA | recB | fac.  role: which half of the work of the atom (0 when WPA == 1).
    __device__ __forceinline__ void init(const AniParams* P_, int capA_, int CH_, int vec_ok_, float* angular_, int ld_angular_,
                                         char* lds, int role_) {
        P = P_; capA = capA_; CH = CH_; vec_ok = vec_ok_; angular = angular_; ld_angular = ld_angular_; role = role_;
        lane = lane_id();
        NB = P->NB; nA = P->nA;
        const int nFR = P->nFR, nFZ = P->nFZ;
        K = P->fwd_split; logK = 31 - __builtin_clz(K);
        recA = (float4*)lds;
        recB = recA + capA;
        fac = (float*)(recB + capA);
        qtab = (int*)(fac + (size_t)(CH + 1) * REC);
        fac_addr = (int)(uintptr_t)fac + (lane & 3) * (NR4 * 4);                   // this lane's R pieces in record 0
        zdelta = NFRP * 4 + (lane & 3) * (NZ4 * 4) - (lane & 3) * (NR4 * 4);       // from the R pieces to the Z pieces
        zero_addr = fac_addr + CH * (REC * 4);
        quad = lane >> 2; nn = lane & 3;                       // per-lane view of phase 2: quad = block of the MFMA, nn = column inside the block
#pragma unroll
        for (int s = 0; s < NS; s++) {
            sbk[s] = P->fwd_slot_bucket[(role * NS + s) * 16 + quad];          // -1: unused slot
            spart[s] = ((role * NS + s) * 16 + quad) & (K - 1);
        }
        if constexpr (UNI == 2) {
            static_assert(UNI != 2 || (NFRP == 8 && NFZP == 4 && NR4 == 2), "the literal set is ANI-2x's 8 x 4");
            zz0 = Ani2xAngular::zeta; zb0 = Ani2xAngular::zbias;
            geo.rs1 = Ani2xAngular::rs1; geo.c = Ani2xAngular::c; geo.k1 = Ani2xAngular::k1; geo.k0 = Ani2xAngular::k0;
            geo.q = Ani2xAngular::q; geo.q4 = Ani2xAngular::q4; geo.qi4 = Ani2xAngular::qi4; geo.d4 = Ani2xAngular::d4;
#pragma unroll
            for (int z = 0; z < NFZP; z++) {
                constexpr float c4[4] = {Ani2xAngular::zc0, Ani2xAngular::zc1, Ani2xAngular::zc2, Ani2xAngular::zc3};
                constexpr float s4[4] = {Ani2xAngular::zs0, Ani2xAngular::zs1, Ani2xAngular::zs2, Ani2xAngular::zs3};
                zc[z] = c4[z & 3]; zs[z] = s4[z & 3];
            }
        } else {
#pragma unroll
            for (int a = 0; a < NFRP; a++) { frc[a] = a < nFR ? P->fr_c[a] : 0.f; frs[a] = a < nFR ? P->fr_rs[a] : 0.f; }
#pragma unroll
            for (int z = 0; z < NFZP; z++) {
                zz[z] = z < nFZ ? P->fz_zeta[z] : 1.f;
                zc[z] = z < nFZ ? P->fz_cos[z] : 0.f;
                zs[z] = z < nFZ ? P->fz_sin[z] : 0.f;
                zb[z] = z < nFZ ? P->fz_bias[z] : 0.f;             // 1 - zeta: the 2^(1-zeta) of ref :104-109 folded into the exponent
            }
            frc0 = P->fr_c[0]; zz0 = P->fz_zeta[0]; zb0 = P->fz_bias[0];
            if constexpr (UNI && NFRP == 8) {
                static_assert(NFRP != 8 || NR4 == 2, "factor order of the 16-byte records");
                geo.rs1 = P->geo.rs1; geo.c = P->geo.c; geo.k1 = P->geo.k1; geo.k0 = P->geo.k0;
                geo.q = P->geo.q; geo.q4 = P->geo.q4; geo.qi4 = P->geo.qi4; geo.d4 = P->geo.d4;
            }
        }
    }
    __device__ __forceinline__ void write_zero_record() const {
        if (role == 0 && lane < REC) fac[CH * REC + lane] = 0.f;
    }

    // One atom with n angular neighbours.  tri_at(t): word of triple t; boff_at(b): first triple of bucket b;
    // stage_records(): called once, before the first barrier -- the records of the atom must be in recA / recB after it.
    template <class TriAt, class BoffAt, class StageRecords>
    __device__ __forceinline__ void atom(int i, int n, TriAt&& tri_at, BoffAt&& boff_at, StageRecords&& stage_records) {
        const int T = (n * (n - 1)) / 2;
        int word = role * 64 + lane < T ? tri_at(role * 64 + lane) : 0;        // my first batch of triple words, in flight early
        int sstart[NS], send[NS];
        if constexpr (DYN) {
            if (role == 0) {
                // lane b = bucket b: its size from two neighbouring offsets (one coalesced load of the atom's NB + 1 offsets)
                const int off = lane <= NB ? boff_at(lane) : 0;
                const int off1 = __shfl_down(off, 1, 64);
                const int nb = (lane < NB && T > 0) ? off1 - off : 0;
                if (lane < 32) { qtab[2 * lane] = 0; qtab[2 * lane + 1] = 0; }
                // piece length: the estimate T / (32 - nonempty / 2) (half a piece is lost per bucket to rounding), then up until it fits
                const int nz = __popcll(__ballot(nb > 0));
                int L = max(1, (2 * T + (63 - nz)) / max(1, 64 - nz));
                int k, q0, shift;
                for (;;) {
                    k = (int)(((float)(nb + L - 1) + 0.5f) * __builtin_amdgcn_rcpf((float)L));     // ceil(nb / L), exact for these sizes
                    const int incl = wave_prefix_sum(k);
                    q0 = incl - k;
                    // a bucket's quads must sit in ONE wave (slots 0-15 / 16-31): the one that would straddle starts at 16
                    shift = wave_max_nonneg((q0 < 16 && q0 + k > 16) ? 16 - q0 : 0);
                    if (__builtin_amdgcn_readlane(incl, 63) + shift <= 32) break;
                    L++;
                }
                if (q0 + k > 16) q0 += shift;
                const int kmax = wave_max_nonneg(k);
                wave_fence();                                  // (the zeros above)
                // (round 5) the k quads of a bucket take its triples ROUND-ROBIN (triple j goes to quad j mod k), not as k consecutive runs
                // of L: the staging area holds a window of the bucket-major list, and a window of CH triples meets only ~CH / L of
                // the runs -- those quads walked L steps per chunk while the others multiplied zeros, so an atom of c chunks ran c
                // times the steps it needs (dense molecules: 800 triples, 4 chunks, 100 steps instead of 25).  Dealt round-robin every
                // quad of a bucket has its share of every window: ~CH / 32 steps per chunk whatever the atom.
                for (int j = 0; j < kmax; j++)
                    if (j < k) {
                        qtab[2 * (q0 + j)] = off | (nb << 16);
                        qtab[2 * (q0 + j) + 1] = lane | (j << 8) | (k << 16);
                    }
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int b = max(sbk[s], 0);
                const int lo = boff_at(b), hi = boff_at(b + 1);
                sstart[s] = (sbk[s] >= 0 && T > 0) ? lo : 0;
                send[s] = (sbk[s] >= 0 && T > 0) ? hi : 0;
            }
        }
        stage_records();

        mfma_f4 acc[NS][NZ4][NR4];
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int zh = 0; zh < NZ4; zh++)
#pragma unroll
                for (int rh = 0; rh < NR4; rh++) acc[s][zh][rh] = mfma_f4{0.f, 0.f, 0.f, 0.f};
        sync();
        if constexpr (DYN) {                                   // my quad's entry of the table
            const int2 e = *reinterpret_cast<const int2*>(qtab + 2 * (role * 16 + quad));
            dparts = e.y >> 16; dpart = (e.y >> 8) & 0xff;
            sbk[0] = dparts > 0 ? (e.y & 0xff) : -1;
            spart[0] = dpart;
            sstart[0] = e.x & 0xffff;                          // the whole bucket: this quad's triples are first + part + j * parts
            send[0] = sstart[0] + (int)((unsigned)e.x >> 16);
            dinv = __builtin_amdgcn_rcpf((float)max(dparts, 1));
        }

        for (int c0 = 0; c0 < T; c0 += CH) {
            const int c1 = min(c0 + CH, T);
            // ---------------- phase 1: lane = triple, records of the chunk to LDS ----------------
            const int first_sub = c0 + role * 64;
            if (c0 > 0 && (CH & (64 * WPA - 1)) != 0) word = first_sub + lane < T ? tri_at(first_sub + lane) : 0;   // (ragged chunks)
            for (int sub = first_sub; sub < c1; sub += 64 * WPA) {
                const int t = sub + lane;
                const int next_word = (t + 64 * WPA < T) ? tri_at(t + 64 * WPA) : 0;
                if (t < c1) {
                    const int p = word & 0xff, q = (word >> 8) & 0xff;
                    const float4 A = recA[p], B = recA[q];
                    const float4 A2 = recB[p], B2 = recB[q];
                    const TripleGeom g = triple_geometry<TORCHANI>(A, A2, B, B2);
                    float vr[NFRP], vz[NFZP];
                    if constexpr (UNI && NFRP == 8) {          // one eta, equally spaced shifts: four transcendentals for the eight factors
                        v2f R04, R15, R26, R37, Y;             // the record holds {R0, R4, R1, R5 | R2, R6, R3, R7}: the pairs as they come
                        radial_factors_geo8(g.rbar, geo, R04, R15, R26, R37, Y);
                        vr[0] = R04.x; vr[1] = R04.y; vr[2] = R15.x; vr[3] = R15.y;
                        vr[4] = R26.x; vr[5] = R26.y; vr[6] = R37.x; vr[7] = R37.y;
                    } else {
#pragma unroll
                        for (int a = 0; a < NFRP; a++) {       // stored so that column nn finds its NR4 values together
                            const float sh = g.rbar - frs[a];
                            vr[(a & 3) * NR4 + (a >> 2)] = fast_exp2((UNI ? frc0 : frc[a]) * sh * sh);
                        }
                    }
#pragma unroll
                    for (int z = 0; z < NFZP; z++) {
                        const float x = fmaxf(1.0f + (g.c * zc[z] + g.s * zs[z]), 1e-30f);   // 1 + cos(theta - ths)
                        vz[(z & 3) * NZ4 + (z >> 2)] = g.fcfc * fast_exp2(fmaf(UNI ? zz0 : zz[z], fast_log2(x), UNI ? zb0 : zb[z]));
                    }
                    float* dst = fac + (t - c0) * REC;
#pragma unroll
                    for (int a = 0; a < NFRP; a += 4)
                        *reinterpret_cast<float4*>(dst + a) = make_float4(vr[a], vr[a + 1], vr[a + 2], vr[a + 3]);
#pragma unroll
                    for (int z = 0; z < NFZP; z += 4)
                        *reinterpret_cast<float4*>(dst + NFRP + z) = make_float4(vz[z], vz[z + 1], vz[z + 2], vz[z + 3]);
                }
                word = next_word;
            }
            sync();
            // ---------------- phase 2: lane = (quad, column); one triple per quad per step ----------------
            // LDS byte addresses: `ra` of this lane's R pieces in the record of the quad's next triple, Z pieces at ra + zdelta
            int cnt[NS], ra[NS];
            int cmax = 0;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int lo = max(sstart[s], c0), hi = min(send[s], c1);
                if constexpr (DYN) {                           // every dparts-th triple of the bucket, from dpart on
                    // (floor(x / dparts) as (int)((x + 0.5) / dparts) in float: x < 2^15, dparts <= 32 -- the error of the product is
                    //  below 1e-3 where the nearest integer boundary is 0.5 / 32 away)
                    const int a = max(lo - sstart[s], 0), pp = max(dparts, 1);
                    const int r = a - pp * (int)(((float)a + 0.5f) * dinv);
                    int d = dpart - r;
                    d += d < 0 ? pp : 0;
                    const int first = lo + d;
                    cnt[s] = hi > first ? (int)(((float)(hi - first + pp - 1) + 0.5f) * dinv) : 0;
                    ra[s] = fac_addr + (first - c0) * (REC * 4);
                } else {
                    const int first = lo + ((spart[s] - (lo - sstart[s])) & (K - 1));     // triples of a bucket are dealt round-robin
                    cnt[s] = max(0, (hi - first + K - 1) >> logK);
                    ra[s] = fac_addr + (first - c0) * (REC * 4);
                }
                cmax = max(cmax, cnt[s]);
            }
            const int steps = wave_max_nonneg(cmax);             // wave-uniform trip count
            const int stride = (DYN ? max(dparts, 1) : K) * REC * 4;
            float ar[NS][NR4], az[NS][NZ4], br[NS][NR4], bz[NS][NZ4];      // operand registers, ping-pong
            auto fetch = [&](int k, float (&r)[NS][NR4], float (&z)[NS][NZ4]) {
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const int at = k < cnt[s] ? ra[s] : zero_addr;     // an exhausted quad multiplies zeros
                    ra[s] += stride;
                    lds_read_vec<NR4>(lds_ptr(at), r[s]);
                    lds_read_vec<NZ4>(lds_ptr(at + zdelta), z[s]);
                }
            };
            auto issue = [&](const float (&r)[NS][NR4], const float (&z)[NS][NZ4]) {
#pragma unroll
                for (int s = 0; s < NS; s++)
#pragma unroll
                    for (int zh = 0; zh < NZ4; zh++)
#pragma unroll
                        for (int rh = 0; rh < NR4; rh++)
                            acc[s][zh][rh] = __builtin_amdgcn_mfma_f32_4x4x1f32(z[s][zh], r[s][rh], acc[s][zh][rh], 0, 0, 0);
            };
            fetch(0, ar, az);
            int k = 0;
            for (; k + 1 < steps; k += 2) {                    // operands of the next step in flight during the MFMAs
                fetch(k + 1, br, bz);
                issue(ar, az);
                fetch(k + 2, ar, az);
                issue(br, bz);
            }
            if (k < steps) issue(ar, az);
            if (c1 < T) sync();                                // (the staging area is about to be overwritten)
        }

        if constexpr (DYN) {
            // the quads of a bucket are consecutive and inside one wave: segmented suffix sum, part 0 ends with the total
            const int kw = wave_max_nonneg(dparts);
            for (int d = 1; d < kw; d <<= 1) {
                const bool take = dpart + d < dparts;
#pragma unroll
                for (int zh = 0; zh < NZ4; zh++)
#pragma unroll
                    for (int rh = 0; rh < NR4; rh++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float o = __shfl_down(acc[0][zh][rh][r], 4 * d, 64);
                            acc[0][zh][rh][r] += take ? o : 0.f;
                        }
            }
        } else {
        // the K quads of a bucket hold partial blocks: add them up (all end with the total)
        for (int off = 4; off < 4 * K; off <<= 1) {
#pragma unroll
            for (int s = 0; s < NS; s++)
#pragma unroll
                for (int zh = 0; zh < NZ4; zh++)
#pragma unroll
                    for (int rh = 0; rh < NR4; rh++)
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[s][zh][rh][r] += __shfl_xor(acc[s][zh][rh][r], off, 64);
        }
        }

        // ---------------- epilogue: registers -> the atom's output row ----------------
        // register r of lane (quad, nn) of block (zh, rh) is canonical slot (a = nn + 4 rh, z = r + 4 zh) of the quad's bucket
        float* out = angular + (size_t)i * ld_angular;
        const bool via_lds = vec_ok && (vec_ok & 8) && NB * nA <= CH * REC;
        if (via_lds) {
            // The row is assembled in LDS (the staging area is free now) and leaves as whole 1 KB wave stores: a quad holds
            // 64-byte halves of 128-byte blocks, and stores of that shape reached memory as partial lines (47.9 MB of write
            // traffic for 35.8 MB of rows, profiles/r02b_hbm_traffic_pmc.txt).
            float* rowbuf = fac;
            constexpr int NT = 64 * WPA;
            const int tid = role * 64 + lane, pieces = (NB * nA) >> 2;
            const int nabs = P->fwd_nabsent;
            sync();                                            // every wave is done with the staged factors
            if (DYN || nabs > 0) {                             // blocks nobody owns are zero (DYN: species pairs without triples have no quad)
                for (int q = tid; q < pieces; q += NT) reinterpret_cast<float4*>(rowbuf)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                sync();
            }
#pragma unroll
            for (int s = 0; s < NS; s++) {
                if (sbk[s] < 0 || spart[s] != 0) continue;
#pragma unroll
                for (int zh = 0; zh < NZ4; zh++)
#pragma unroll
                    for (int rh = 0; rh < NR4; rh++) {
                        const mfma_f4 v = acc[s][zh][rh];
                        *reinterpret_cast<float4*>(rowbuf + sbk[s] * nA + (nn + 4 * rh) * NFZP + 4 * zh) = make_float4(v[0], v[1], v[2], v[3]);
                    }
            }
            sync();
            for (int q = tid; q < pieces; q += NT) {
                const float4 v = reinterpret_cast<const float4*>(rowbuf)[q];
                store_row16(out + 4 * q, mfma_f4{v.x, v.y, v.z, v.w}, (vec_ok >> 1) & 3);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; s++) {
                if (sbk[s] < 0 || spart[s] != 0) continue;
                float* ob = out + sbk[s] * nA;
#pragma unroll
                for (int zh = 0; zh < NZ4; zh++)
#pragma unroll
                    for (int rh = 0; rh < NR4; rh++) {
                        const int c = (nn + 4 * rh) * NFZP + 4 * zh;
                        const mfma_f4 v = acc[s][zh][rh];
                        if (vec_ok) {                              // function m sits at canonical slot m: one 16-byte store
                            store_row16(ob + c, v, (vec_ok >> 1) & 3);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int m = P->m_of_c[c + r];
                                if (m >= 0) ob[m] = v[r];
                            }
                        }
                    }
            }
            // species pairs that cannot occur in this system have no quad: their blocks are zero
            const int nabs = P->fwd_nabsent;
            if (nabs > 0 && role == 0) {
                if (vec_ok) {                                      // 2^fwd_zero_shift lanes per block, one 16-byte piece each
                    const int sh = P->fwd_zero_shift, piece = lane & ((1 << sh) - 1), per_pass = 64 >> sh;
                    for (int a0 = 0; a0 < nabs; a0 += per_pass) {
                        const int a = a0 + (lane >> sh);
                        if (a < nabs && piece * 4 < nA)
                            *reinterpret_cast<float4*>(out + P->fwd_absent[a] * nA + piece * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                } else {
                    for (int a = 0; a < nabs; a++) {
                        float* ob = out + P->fwd_absent[a] * nA;
                        for (int m = lane; m < nA; m += 64) ob[m] = 0.f;
                    }
                }
            }
        }
    }
};

template <bool TORCHANI, int NFRP, int NFZP, int WPA, int OCC, int UNI = 0, bool DYN = false>
__global__ __launch_bounds__(WPA == 2 ? 128 : 64 * kWavesPerGroup, OCC) void ani_angular_forward_mfma(
    const AniParams* __restrict__ P, int cap, int capA, int CH, const float4* __restrict__ recA_g,
    const float4* __restrict__ recB_g, const int* __restrict__ tri_g, const int* __restrict__ cnt_a,
    const int* __restrict__ cnt_ro, float* __restrict__ angular, int ld_angular, int vec_ok, int lds_per_atom,
    const int* __restrict__ order, int w0, int nw) {      // this launch covers positions [w0, w0 + nw) of `order` (NULL: atom = position)
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int wig = __builtin_amdgcn_readfirstlane(wave_in_group());        // wave-uniform: keeps per-atom addressing scalar
    const int slot_in_group = WPA == 2 ? 0 : wig;               // which atom of the workgroup
    MfmaForward<TORCHANI, NFRP, NFZP, WPA, UNI, DYN> F;
    F.init(P, capA, CH, vec_ok, angular, ld_angular, lds_raw + (size_t)slot_in_group * lds_per_atom, WPA == 2 ? wig : 0);
    F.write_zero_record();
    const int lane = F.lane, NB = F.NB;

    const int natoms_group = WPA == 2 ? 1 : (blockDim.x >> 6);
    const int stride_atoms = gridDim.x * natoms_group;
    for (int w = blockIdx.x * natoms_group + slot_in_group; w < nw; w += stride_atoms) {
        int i = order ? order[w0 + w] : w0 + w;
        if ((unsigned)i >= (unsigned)P->N) i = w0 + w;         // (a void grid build leaves no valid order: stay in bounds)
        int n, nro;
        clamp_counts(cnt_a[i], cnt_ro[i], cap, capA, n, nro);
        const int* tri = tri_g + (size_t)i * triples_capacity(capA);
        const int* boff_g = P->bucket_offsets + (size_t)i * (NB + 1);
        // (Requesting the first triple words and records BEFORE the counts are known -- one dependent round trip less per atom --
        //  was built and measured in round 4: no gain, and the values it keeps alive cost this kernel, which sits exactly at the
        //  72 registers of seven waves per SIMD, 20 bytes of scratch: 17.5 -> 18.7 us.  The backward kernel keeps that form.)
        F.atom(i, n, [&](int t) { return tri[t]; }, [&](int b) { return boff_g[b]; },
               [&]() {
                   if constexpr (WPA == 2) {                   // one array each
                       const float4* src = (F.role == 0 ? recA_g : recB_g) + (size_t)i * capA;
                       float4* dst = F.role == 0 ? F.recA : F.recB;
                       for (int e = lane; e < n; e += 64) dst[e] = src[e];
                   } else {
                       load_angular_records(recA_g + (size_t)i * capA, recB_g + (size_t)i * capA, n, F.recA, F.recB);
                   }
               });
        if (w + stride_atoms < nw) F.sync();                   // (another atom follows: records and staging area must be free)
    }
}

}  // namespace nnpops
