// ani_angular_bwd.h -- angular AEV backward, a 128-lane workgroup (two waves) per atom.
//
// What is computed: reference src/ani/CpuANISymmetryFunctions.cpp:265-353 -- for every neighbour pair (j, k) of
// atom i the three chain-rule routes (through r_ij, r_ik and theta) of all angular functions, contracted with the
// upstream gradient block of the pair's species bucket, as forces on the two legs and on the centre.
//
// Same mathematics and the same conflict-free accumulation as ani_angular_backward (ani_kernels.h): lane = triple,
// the force of a triple on its legs is  F_p = alpha_p A + beta B,  F_q = alpha_q B + beta A  (A, B = leg
// displacements), the scalars go to an LDS pair matrix with exactly one writer per entry, row sums give the per-slot
// forces, which are parked in leg_force / centre_force for ani_radial_backward to gather.  What changed, all of it
// because these kernels are bound by vector-instruction issue (one VALU instruction per ~4 cycles per SIMD; the
// measured time of every per-atom kernel is its instruction count times that):
//   * two waves per atom share ONE set of LDS arrays and take alternate batches of 64 triples: half the LDS per wave
//     (28 instead of 15-19 waves per CU), half the latency per atom;
//   * the per-triple arithmetic is written two factors at a time (v_pk_fma/mul/add_f32 run at twice the scalar rate):
//     the 8 x 4 x {R, dR} contraction with the gradient block is 32 packed FMAs instead of 64;
//   * 2^(1-zeta) is folded into the angular factor instead of into a scaled copy of the gradient row, so the row goes
//     global -> LDS as 16-byte pieces without arithmetic; (1+cos)^zeta = (1+cos)^(zeta-1) (1+cos): one exp2 for the
//     value and the derivative;
//   * the pair matrix always covers cap_angular slots (no host-sized tile, no tile-pair fallback: an atom that grew
//     since the last check() is handled like any other), and the row sums split the columns in two equal halves.
#pragma once

#include "ani_angular_generic.h"
#include "ani_angular_mfma.h"

namespace nnpops {


template <bool TORCHANI, int NFRP, int NFZP>
__device__ __forceinline__ void triple_forces_pk(const float4& A, const float4& A2, const float4& B, const float4& B2,
                                                 const float* Gb, const float (&frc)[NFRP], const float (&frs)[NFRP],
                                                 const float (&fren)[NFRP], const float (&zz)[NFZP], const float (&zc)[NFZP],
                                                 const float (&zs)[NFZP], const float (&zb)[NFZP], float& alpha_p,
                                                 float& alpha_q, float& beta) {
    const TripleGeom g = triple_geometry<TORCHANI>(A, A2, B, B2);
    const v2f rb2 = {g.rbar, g.rbar}, c2 = {g.c, g.c}, s2 = {g.s, g.s}, one = {1.0f, 1.0f};
    float R[NFRP], dR[NFRP];
    // (The forward kernel's recurrence for eight equally spaced shifts -- radial_factors_geo8, four transcendentals instead of
    //  eight -- was built here too and measured 15.5 -> 15.7 us: the derivative needs every (rbar - Rs_a) anyway, and the packed
    //  sub / mul / mul below already handles two factors per instruction.)
#pragma unroll
    for (int a = 0; a < NFRP; a += 2) {
        const v2f sh = rb2 - v2f{frs[a], frs[a + 1]};
        const v2f arg = v2f{frc[a], frc[a + 1]} * sh * sh;
        const v2f r = {fast_exp2(arg.x), fast_exp2(arg.y)};
        const v2f d = (v2f{fren[a], fren[a + 1]} * sh) * r;      // d/dr_ij of exp(-eta (rbar-Rs)^2): rbar carries 1/2 (ref :306)
        R[a] = r.x; R[a + 1] = r.y; dR[a] = d.x; dR[a + 1] = d.y;
    }
    // contract the gradient block with R and dR:  U_z = sum_a G[a][z] R_a,  V_z = sum_a G[a][z] dR_a   (pairs of z)
    v2f U[NFZP / 2], V[NFZP / 2];
#pragma unroll
    for (int z = 0; z < NFZP / 2; z++) { U[z] = v2f{0.f, 0.f}; V[z] = v2f{0.f, 0.f}; }
#pragma unroll
    for (int a = 0; a < NFRP; a++) {
        const v2f ra = {R[a], R[a]}, da = {dR[a], dR[a]};
#pragma unroll
        for (int z = 0; z < NFZP; z += 4) {
            const float4 gv = *reinterpret_cast<const float4*>(Gb + a * NFZP + z);
            const v2f g01 = {gv.x, gv.y}, g23 = {gv.z, gv.w};
            U[z / 2] += g01 * ra;     V[z / 2] += g01 * da;
            U[z / 2 + 1] += g23 * ra; V[z / 2 + 1] += g23 * da;
        }
    }
    v2f S0 = {0.f, 0.f}, Sr = {0.f, 0.f}, Sth = {0.f, 0.f};
#pragma unroll
    for (int z = 0; z < NFZP; z += 2) {
        const v2f zc2 = {zc[z], zc[z + 1]}, zs2 = {zs[z], zs[z + 1]};
        const v2f cz = c2 * zc2 + s2 * zs2;               // cos(theta - ths)
        const v2f sz = s2 * zc2 - c2 * zs2;               // sin(theta - ths)
        v2f x = one + cz;
        x.x = fmaxf(x.x, 1e-30f); x.y = fmaxf(x.y, 1e-30f);      // keeps 0 * -inf out of the zeta == 1 corner
        const v2f lg = {fast_log2(x.x), fast_log2(x.y)};
        const v2f e = v2f{zz[z] - 1.0f, zz[z + 1] - 1.0f} * lg + v2f{zb[z], zb[z + 1]};
        const v2f Zm1 = {fast_exp2(e.x), fast_exp2(e.y)};       // 2^(1-zeta) (1+cos)^(zeta-1)       (scale: ref :104-109)
        const v2f Z = Zm1 * x;
        const v2f dZ = (v2f{-zz[z], -zz[z + 1]} * Zm1) * sz;    // d/dtheta                          ref :337
        S0 += U[z / 2] * Z;
        Sr += V[z / 2] * Z;
        Sth += U[z / 2] * dZ;
    }
    const float s0 = S0.x + S0.y, sr = Sr.x + Sr.y, sth = Sth.x + Sth.y;
    // three routes of the chain rule (ref :311-348), already summed over the functions m
    const float t1 = A2.y * B2.x * s0 + g.fcfc * sr;   // through r_ij   (A2.y = dfc_ij, B2.x = fc_ik)
    const float t2 = A2.x * B2.y * s0 + g.fcfc * sr;   // through r_ik
    const float t3 = g.fcfc * sth;                     // through theta
    // angle gradients (ref :410-433): dtheta/d(dot') = -damp / sin(theta)
    const float dot = A.x * B.x + A.y * B.y + A.z * B.z;
    const float iprod = A2.z * B2.z;
    const float damp = TORCHANI ? 0.95f : 1.0f;
    const float dadd = -damp * fast_rcp(g.s) * iprod * t3;
    const float ka = dot * A2.z * A2.z, kb = dot * B2.z * B2.z;
    const float s1 = t1 * A2.z, s2f = t2 * B2.z;
    // F_p = s1 A + dadd (B - ka A),  F_q = s2 B + dadd (A - kb B)
    alpha_p = s1 - dadd * ka;
    alpha_q = s2f - dadd * kb;
    beta = dadd;
}

// WPA: waves per atom (1: a 64-lane workgroup owns an atom; 2: a 128-lane workgroup, the waves take alternate batches).
// GLDS: the atom's upstream gradient row is staged in LDS (true), or every triple reads its 128-byte block straight from
// global memory through the vector L1 (false; needs the 16-byte layout `vec_ok`): 3.5 KB less LDS per atom, i.e. more
// atoms in flight per CU -- these kernels are bound by latency x occupancy, not by issue slots or bytes.
template <int NFRP, int NFZP>
__host__ __device__ inline size_t ang_bwd_pair_lds_bytes(int capA, int NB, bool glds) {
    return (size_t)capA * 2 * sizeof(float4) + (glds ? (size_t)NB * NFRP * NFZP * sizeof(float) : 0) +
           ((size_t)capA * (capA + 1) + (size_t)capA * (capA - 1) / 2) * sizeof(float);
    // (LDS is handed out in 128 pieces of 1 280 bytes per CU: at 64 record slots this is 26 752 bytes = 21 pieces, six workgroups per
    //  CU; 256 bytes more -- a table of the receivers' slots, round 5 -- were 22 pieces, five workgroups, and 12 % of the kernel's speed:
    //  the table lives in a dead field of the records instead)
}

// GENERIC: the function list does not factor (ani_angular_generic.h): functions evaluated one by one, gradients read from
// global memory in the caller's order (GLDS must be false, NFRP / NFZP are not used).
// UNI: one eta for every radial factor, one zeta for every angular factor, no padded factor slots (ani_angular_mfma.h):
// 20 fewer wave-uniform constants in scalar registers.
// CLASSES: 1 -- the instantiation the launches by class of atoms use, 2 -- the clean-up launch behind them (class_word below); 0 -- the
// class logic is not compiled in.  This kernel spills scalar registers as it is, and every value kept alive across the atom costs
// time (measured on the 1 024-conformer batch: the three class launches 233 -> 243 us with the flag's address, the stamp and the
// per-atom limits all alive in one instantiation): a class launch keeps ONE extra scalar (the stamp) and finds the flag behind the
// last centre force, the clean-up launch -- whose speed does not matter -- carries the rest.
// SCAT: the two-wave instantiation that stores the leg forces in the receiving atoms' rows (see "where the leg forces go" below);
// everybody else is compiled without that code -- written into one kernel behind a run-time pointer test it cost the two-wave
// kernel a third of its speed (18-22 spilled scalar registers: 32.6 -> 44 us on a 7 600-atom block of conformers).
template <bool TORCHANI, int NFRP, int NFZP, int OCC, int WPA, bool GLDS, bool GENERIC = false, int UNI = 0, int CLASSES = 0, bool SCAT = false>
__global__ __launch_bounds__(WPA == 2 ? 128 : 64 * kWavesPerGroup, OCC) void ani_angular_backward_pair(
    const AniParams* __restrict__ P, const AngularConsts C, int cap, int capA, int tile, const float4* __restrict__ recA_g,
    const float4* __restrict__ recB_g, const int* __restrict__ tri_g, const int* __restrict__ cnt_a, const int* __restrict__ cnt_ro,
    const float* __restrict__ angular_grad, int ld_angular, float4* __restrict__ leg_force, float4* __restrict__ centre_force,
    const int* __restrict__ ids_g, float4* __restrict__ recv,      // recv != NULL (two waves per atom only): see "where the leg forces go"
    int vec_ok, int NB, int lds_per_atom, const int* __restrict__ order, int w0, int nw,     // positions [w0, w0 + nw) of `order`
    int class_word) {      // this backprop()'s stamp (CLASSES != 0)
    // CLASSES (launches by class of atoms, nnpops_ani_check): 0 -- one launch for everybody, `tile` covers every record slot;
    // 1 -- the launch of one class: an atom that has outgrown the class since check() (more angular neighbours than `tile`) is LEFT
    // OUT here -- the launch says so by writing this backprop()'s stamp to class_flag -- and evaluated by the launch of mode 2, which
    // runs behind the classes with the full-size pair matrix, returns at once unless the flag carries the stamp, and takes exactly
    // the atoms the classes left out.  Forces are right whether or not anybody calls check() between the frames (graph replays,
    // check intervals > 1: a replayed stamp keeps the clean-up launch on from the first outgrown atom on, which is only slower).
    // (the flag is the word behind the last centre force, centre_force[N].x; the per-atom limits of the clean-up launch come through the
    //  parameter block, P->class_tile)
    constexpr int BLK = NFRP * NFZP;
    const int stamp = class_word;
    if constexpr (CLASSES == 2) {
        if (__float_as_int(centre_force[C.N].x) != stamp) return;
    }
    constexpr int NT = 64 * WPA;                               // lanes of the workgroup
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int lane = lane_id();
    const int wig = __builtin_amdgcn_readfirstlane(wave_in_group());
    const int role = WPA == 2 ? wig : 0;
    // one wave per atom: a workgroup holds blockDim / 64 independent atoms (fewer, larger dispatches), each with its own LDS slice
    const int atoms_per_group = WPA == 2 ? 1 : (int)(blockDim.x >> 6);
    const int slot_in_group = WPA == 2 ? 0 : wig;
    const int nA = C.nA;
    // tile: edge of the LDS pair matrix and number of record slots of THIS launch (<= capA, the stride of the global arrays): check()
    // groups the atoms by their number of angular neighbours and the groups are launched one after the other, each with the
    // LDS its atoms need -- 7 KB for up to 32 neighbours, 15 KB for 48, 27 KB for 64: in a batch of compact molecules (BASELINE
    // config 4) two thirds of the atoms have at most 32 and ran six to a CU because 1 % have more than 48.
    const int tstride = tile + 1;
    auto sync = [&]() {
        if constexpr (WPA == 2) __syncthreads();
        else wave_fence();
    };

    char* cursor = lds_raw + (size_t)slot_in_group * lds_per_atom;
    float4* recA = (float4*)cursor;       cursor += (size_t)tile * sizeof(float4);
    float4* recB = (float4*)cursor;       cursor += (size_t)tile * sizeof(float4);
    float* grow = (float*)cursor;         if (GLDS) cursor += (size_t)NB * BLK * sizeof(float);   // upstream gradient row, canonical [bucket][a][z]
    float* Ma = (float*)cursor;           // alpha[tile][tile + 1]: Ma[e][x] = coefficient of A_e in the force of triple {e, x} on e
    float* Mb = Ma + tile * tstride;      // beta, once per unordered pair (p < q), triangular
    // (SCAT: leg e -> slot of THIS atom in the records of the atom on that leg is kept in recB[e].x -- fc of the leg, dead once the
    //  triples are done; .w, the leg's atom id, is still needed)
    static_assert(!SCAT || WPA == 2, "the look-up of the receivers' slots is the second wave's work");

    // Where the leg forces go.  One wave per atom (liquids): parked in leg_force[i][e]; every atom then finds itself in the id rows of
    // its angular neighbours inside the radial backward kernel.  Two waves per atom (dense systems: molecules, where a 64-slot id row
    // is 256 bytes and that search is most of the radial backward's 7.7 KB per atom) and recv != NULL (round 5): the SECOND wave, idle
    // while the first adds up the rows of the pair matrix, looks the slot of this atom up in the id row of the atom on every leg, and
    // the force is stored straight into the RECEIVING atom's row, recv[j][slot of i in the records of j] -- one writer per entry (the
    // pair relation is symmetric: the same r^2 from both ends), the receiver reads its own row: contiguous, no search, no atomics,
    // bitwise reproducible.  (With ONE wave per atom the same look-up was measured at +7 ... +16 us on a 15 us kernel at its
    // register limit, docs/LAB_NOTEBOOK_r05.md: it stays in the radial backward there.)
    // An id row is capA ints = capA / 4 16-byte pieces; QL lanes scan one row, RPP rows per pass of the wave, 8 passes in flight.
    auto lookup_receiver_slots = [&](int i, int n) {
        constexpr int NP = 8;
        //  a leg whose atom does not list THIS atom -- its row was cut short by an overflow of cap_angular, so the
        // pair relation is not symmetric in this frame -- must not leave fc's bits behind as a "slot": -1 = no receiver, the store is
        // skipped (the frame is reported through the overflow word and evaluated again; until then nothing is written out of bounds).
        for (int e = lane; e < n; e += 64) recB[e].x = __int_as_float(-1);      // (LDS operations of a wave execute in order)
        const int ql_shift = 29 - __builtin_clz((unsigned)capA);      // log2(capA / 4); capA is a power of two >= 32
        const int QL = 1 << ql_shift, rpp_shift = 6 - ql_shift;
        for (int t0 = 0; (t0 << rpp_shift) < n; t0 += NP) {
            int4 idv[NP];
#pragma unroll
            for (int t = 0; t < NP; t++) {
                const int et = (lane >> ql_shift) + ((t0 + t) << rpp_shift);
                idv[t] = make_int4(-1, -1, -1, -1);
                if (et < n) {
                    const int jt = __float_as_int(recB[et].w) & kIdMask;
                    idv[t] = reinterpret_cast<const int4*>(ids_g + (size_t)jt * capA)[lane & (QL - 1)];
                }
            }
#pragma unroll
            for (int t = 0; t < NP; t++) {
                const int et = (lane >> ql_shift) + ((t0 + t) << rpp_shift);
                int slot = -1;
                slot = idv[t].x == i ? 0 : slot;
                slot = idv[t].y == i ? 1 : slot;
                slot = idv[t].z == i ? 2 : slot;
                slot = idv[t].w == i ? 3 : slot;
                if (slot >= 0) recB[et].x = __int_as_float(4 * (lane & (QL - 1)) + slot);
            }
        }
    };

    static_assert(!(GENERIC && GLDS), "generic function lists read their gradients from global memory");
    // (constants from the by-value block, padded on the host: no dependent scalar loads in the workgroup's prologue, ani_kernels.h)
    float frc[NFRP], frs[NFRP], fren[NFRP], zz[NFZP], zc[NFZP], zs[NFZP], zb[NFZP];
    if constexpr (UNI == 2) {                                  // the published ANI-2x set, compiled in (ani_kernels.h: Ani2xAngular)
        static_assert(UNI != 2 || (NFRP == 8 && NFZP == 4), "the literal set is ANI-2x's 8 x 4");
        using L = Ani2xAngular;
        constexpr float rs8[8] = {L::rs_0, L::rs_1, L::rs_2, L::rs_3, L::rs_4, L::rs_5, L::rs_6, L::rs_7};
        constexpr float c4[4] = {L::zc0, L::zc1, L::zc2, L::zc3}, s4[4] = {L::zs0, L::zs1, L::zs2, L::zs3};
#pragma unroll
        for (int a = 0; a < NFRP; a++) { frc[a] = L::c; frs[a] = rs8[a & 7]; fren[a] = L::negeta; }
#pragma unroll
        for (int z = 0; z < NFZP; z++) { zz[z] = L::zeta; zc[z] = c4[z & 3]; zs[z] = s4[z & 3]; zb[z] = L::zbias; }
    } else {
#pragma unroll
        for (int a = 0; a < NFRP; a++) {
            frc[a] = C.fr_c[UNI ? 0 : a];
            frs[a] = C.fr_rs[a];
            fren[a] = C.fr_negeta[UNI ? 0 : a];
        }
#pragma unroll
        for (int z = 0; z < NFZP; z++) {
            zz[z] = C.fz_zeta[UNI ? 0 : z];
            zc[z] = C.fz_cos[z];
            zs[z] = C.fz_sin[z];
            zb[z] = C.fz_bias[UNI ? 0 : z];
        }
    }

    const int stride_atoms = gridDim.x * atoms_per_group;
    for (int w = blockIdx.x * atoms_per_group + slot_in_group; w < nw; w += stride_atoms) {
        int i = order ? order[w0 + w] : w0 + w;
        if ((unsigned)i >= (unsigned)C.N) i = w0 + w;          // (a void grid build leaves no valid order: stay in bounds)
        // (the first batch of triple words and the first 64 records are requested before the counts are known: what lies behind
        //  the atom's last triple / record is allocated and ignored -- one dependent round trip less per atom, ani_angular_mfma.h)
        const int tid = role * 64 + lane;
        const int capT = triples_capacity(capA);
        const int* tri = tri_g + (size_t)i * capT;
        int word = tid < capT ? tri[tid] : 0;
        float4 recA_first = make_float4(0.f, 0.f, 0.f, 0.f), recB_first = recA_first;
        if constexpr (WPA == 2) {
            recA_first = (role == 0 ? recA_g : recB_g)[(size_t)i * capA + min(lane, capA - 1)];
        } else {
            recA_first = recA_g[(size_t)i * capA + min(lane, capA - 1)];
            recB_first = recB_g[(size_t)i * capA + min(lane, capA - 1)];
        }
        int n, nro;
        clamp_counts(cnt_a[i], cnt_ro[i], cap, capA, n, nro);
        if constexpr (CLASSES == 1) {
            if (n > tile) {                                    // outgrew its class: the clean-up launch evaluates it (uniform for the workgroup)
                if (role == 0 && lane == 0) centre_force[C.N].x = __int_as_float(stamp);
                continue;
            }
        }
        if constexpr (CLASSES == 2) {
            if (n <= (int)P->class_tile[i]) continue;          // its class evaluated it
        }
        n = min(n, tile);                                      // (tile >= capA outside class launches: a no-op)
        if (n < 2) {                                           // no triples (uniform for the workgroup): a lone leg carries no force
            if (n == 1 && role == 0) {
                if constexpr (SCAT) {                          // ... and says so in the row of the atom on it
                    const int j = ids_g[(size_t)i * capA];
                    for (int e = lane; e < capA; e += 64)
                        if (ids_g[(size_t)j * capA + e] == i) recv[(size_t)j * capA + e] = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    if (lane == 0) leg_force[(size_t)i * capA] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            continue;
        }
        const int T = (n * (n - 1)) / 2;
        if (tid >= T) word = 0;
        const float* g = angular_grad + (size_t)i * ld_angular;

        // upstream gradient row -> LDS in canonical [bucket][a][z] order (a copy: no scaling, see triple_forces_pk)
        if constexpr (GLDS) {
            if (vec_ok) {                                      // function m sits at canonical slot m: 16-byte pieces
                const int pieces = (NB * BLK) >> 2;
                for (int q = tid; q < pieces; q += NT) reinterpret_cast<float4*>(grow)[q] = reinterpret_cast<const float4*>(g)[q];
            } else {
                for (int c = tid; c < NB * BLK; c += NT) {
                    const int m = P->m_of_c[c % BLK];
                    grow[c] = m >= 0 ? g[(c / BLK) * nA + m] : 0.f;      // padded slots must read as zero
                }
            }
        }
        if constexpr (WPA == 2) {
            const float4* src = (role == 0 ? recA_g : recB_g) + (size_t)i * capA;
            float4* dst = role == 0 ? recA : recB;
            if (lane < tile) dst[lane] = recA_first;           // (not "< n": the compiler would sink the load behind the counts)
            for (int e = lane + 64; e < n; e += 64) dst[e] = src[e];
        } else {
            if (lane < tile) { recA[lane] = recA_first; recB[lane] = recB_first; }
            for (int e = lane + 64; e < n; e += 64) { recA[e] = recA_g[(size_t)i * capA + e]; recB[e] = recB_g[(size_t)i * capA + e]; }
        }
        sync();

        // ---------------- lane = triple; with two waves, alternate batches ----------------
        for (int base = role * 64; base < T; base += NT) {
            const int t = base + lane;
            const int next_word = (t + NT < T) ? tri[t + NT] : 0;
            if (t < T) {
                const int p = word & 0xff, q = (word >> 8) & 0xff, bucket = word >> 16;
                float ap, aq, bt;
                if constexpr (GENERIC) {
                    triple_forces_generic<TORCHANI>(P, nA, recA[p], recB[p], recA[q], recB[q], g + bucket * nA, ap, aq, bt);
                } else {
                    const float* Gb = GLDS ? grow + bucket * BLK : g + bucket * BLK;
                    triple_forces_pk<TORCHANI, NFRP, NFZP>(recA[p], recB[p], recA[q], recB[q], Gb, frc, frs, fren,
                                                           zz, zc, zs, zb, ap, aq, bt);
                }
                Ma[__mul24(p, tstride) + q] = ap;                          // (24-bit multiplies: full rate)
                Ma[__mul24(q, tstride) + p] = aq;
                Mb[__mul24(p, 2 * tile - p - 1) / 2 + (q - p - 1)] = bt;     // p < q: once per unordered pair
            }
            word = next_word;
        }
        sync();

        // ---------------- row sums (first wave):  F_e = (sum_x alpha[e][x]) A_e + sum_x beta{e,x} A_x ----------------
        if constexpr (SCAT) {
            if (role == 1) {                                   // (the second wave meanwhile: the receivers' slots)
                lookup_receiver_slots(i, n);
                sync();                                        // pairs with the one in front of the first wave's first store
            }
        }
        if (role == 0) {
            float cx = 0.f, cy = 0.f, cz = 0.f;
            const int el = lane & 31, half = lane >> 5;
            const int hc = (n + 1) >> 1;                       // columns per half
            for (int eb = 0; eb < n; eb += 32) {
                const int e = eb + el;
                float fx = 0.f, fy = 0.f, fz = 0.f, as = 0.f;
                if (e < n) {
                    const int x0 = half * hc, x1 = min(n, x0 + hc);
                    const float* ma = Ma + __mul24(e, tstride);
                    // index of beta{e,x} in the triangle: x < e: x(2T-x-1)/2 + e-x-1 (grows by T-x-2 per step), x > e: base_e + x-e-1
                    int below = __mul24(x0, 2 * tile - x0 - 1) / 2 + (e - x0 - 1);
                    const int above0 = __mul24(e, 2 * tile - e - 1) / 2 - e - 1;
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
                    const float own = half == 0 ? as : 0.f;      // counted once
                    fx += own * Ae.x; fy += own * Ae.y; fz += own * Ae.z;
                }
                fx += __shfl_xor(fx, 32, 64); fy += __shfl_xor(fy, 32, 64); fz += __shfl_xor(fz, 32, 64);
                if constexpr (SCAT) {
                    if (eb == 0) sync();                       // the receivers' slots are there
                }
                // The force on leg e goes to the atom on that leg (recv, see above) or is parked in leg_force[i][e] (record order); the
                // reaction on the centre in centre_force[i]; the radial backward, which owns position_deriv[j], picks them up.
                if (half == 0 && e < n) {
                    if constexpr (SCAT) {
                        const float4 leg = recB[e];
                        const int rslot = __float_as_int(leg.x);
                        if ((unsigned)rslot < (unsigned)capA)      // (-1: the receiver's row does not hold this atom -- overflowed frame)
                            recv[(size_t)(__float_as_int(leg.w) & kIdMask) * capA + rslot] = make_float4(fx, fy, fz, 0.f);
                    } else leg_force[(size_t)i * capA + e] = make_float4(fx, fy, fz, 0.f);
                    cx -= fx; cy -= fy; cz -= fz;
                }
            }
            cx = wave_sum_lane63(cx); cy = wave_sum_lane63(cy); cz = wave_sum_lane63(cz);
            if (lane == 63) centre_force[i] = make_float4(cx, cy, cz, 0.f);
        }
        if (w + stride_atoms < nw) sync();                    // (another atom follows: the LDS arrays must be free)
    }
}

}  // namespace nnpops
