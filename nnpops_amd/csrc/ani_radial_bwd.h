// ani_radial_bwd.h -- radial backward + gather of the angular leg forces, lane = neighbour.
//
// ani_radial_backward (ani_kernels.h) gives a lane one (neighbour stream, radial function k): 17 dependent steps per
// atom, each with its own gather of a neighbour's gradient row -- five round trips to memory in a row even when
// unrolled by four, and nine LDS reads per step.  The kernel is bound by that latency times the number of occupancy
// rounds, not by arithmetic.  Here a lane owns one NEIGHBOUR: its 16 gradient values arrive as four 16-byte loads
// issued at once (one round trip), the sum over k runs in registers with the function parameters as scalar operands,
// and the reverse lookup of the angular leg forces is done by all 64 lanes together:
//     row of the atom (requested before its counts are known) -> gradient rows of its neighbours -> their id rows
//     -> leg forces.
// ~400 vector instructions per atom instead of 680, 64 registers (8 waves per SIMD).  Used when rows can be read as
// float4 (nR % 4 == 0, ld % 4 == 0, 16-byte aligned gradient tensor) and id rows are 32 or 64 wide; everything else takes
// ani_radial_backward.
//
// Semantics: reference src/ani/CpuANISymmetryFunctions.cpp:228-263 (radial part), :310-344 (angular legs, here
// gathered by the owner from leg_force / centre_force written by the angular backward kernel).
#pragma once

#include "ani_kernels.h"

namespace nnpops {

// LAT: systems that fit the chip in one round of waves (a molecule, a small box): registers are free, so every piece of the id
// rows is requested at once and early -- the wave's time is its chain of dependent round trips, nothing else.
// RECV (round 5): the angular backward has stored every leg force in the RECEIVING atom's row (two-wave kernels, ani_angular_bwd.h):
// `leg_force` is then recv[i][e], the force on i from the triples centred on its e-th angular neighbour -- one contiguous load per
// lane, requested with the atom's gradient row; no id rows, no search.
template <int NR4, int CAPA, bool LAT = false, bool RECV = false>
__global__ __launch_bounds__(64 * kWavesPerGroup, LAT ? 4 : 8) void ani_radial_backward_lanes(
    const AniParams* __restrict__ P, const int* __restrict__ species, const float4* __restrict__ nbr, int cap,
    const int* __restrict__ cnt_pos, const float* __restrict__ radial_grad, int ld_radial,
    const int* __restrict__ ids, const float4* __restrict__ leg_force, const float4* __restrict__ centre_force,
    const int* __restrict__ order,     // atoms in cell order, or NULL
    float* __restrict__ pos_grad, int lds_per_wave, int w0, int nw) {
    constexpr int NR = 4 * NR4;
    constexpr int QL = CAPA / 4, RPP = 64 / QL, NT = CAPA / RPP;      // id row = QL 16-byte pieces; RPP rows per pass of the wave; NT passes cover a row's worth of neighbours
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    float* g_own = (float*)(lds_raw + (size_t)wave_in_group() * lds_per_wave);      // [S * NR] this atom's gradient row
    const int lane = lane_id();
    const int wl = order ? xcd_contiguous_wave_id() : wave_global_id();      // this launch covers positions [w0, w0 + nw)
    if (wl >= nw) return;
    const int w = __builtin_amdgcn_readfirstlane(w0 + wl);         // (wave-uniform: counts and atom id through the scalar cache)
    // Rows and counts are stored by POSITION in this walk (the builders' slot in cell order, or the atom index): the row, its counts
    // and the atom's id come back in ONE round trip -- the kernel is a chain of four dependent ones otherwise.
    const float4* row = nbr + (size_t)w * cap;
    const float4 first = row[min(lane, cap - 1)];          // rows are contiguous (flush_row): requested before the counts are known
    const int counts = cnt_pos[w];
    int i = order ? order[w] : w;
    if ((unsigned)i >= (unsigned)P->N) i = w;              // (a void grid build leaves no valid order: stay in bounds)
    const int width = P->S * NR;
    int na, nro;
    int raw_a, raw_ro, my_species;                         // (the species rides in the same word: the neighbours' gradient rows are
    unpack_cnt_pos(counts, raw_a, raw_ro, my_species);     //  addressed with it, and species[i] would be one more dependent round trip)
    clamp_counts(raw_a, raw_ro, cap, CAPA, na, nro);
    const int total = na + nro;
    const float inv_rcr = P->inv_rcr;
    const int col = my_species * NR;                       // where this atom's species sits in a neighbour's row

    const float* gi = radial_grad + (size_t)i * ld_radial;
    // (the centre force through the SCALAR cache, requested here: no vector register -- the 65th would cost a wave per SIMD -- and
    //  no round trip behind the wave sum at the end)
    const float4 centre = centre_force[__builtin_amdgcn_readfirstlane(i)];
    float4 own_leg = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (RECV) own_leg = leg_force[(size_t)i * CAPA + min(lane, CAPA - 1)];      // (slots behind na: stale, masked below)
    for (int q = lane; q < width; q += 64) g_own[q] = gi[q];
    wave_fence();

    float fx = 0.f, fy = 0.f, fz = 0.f;
    for (int base = 0; base < total; base += 64) {         // one pass for up to 64 neighbours
        const int e = base + lane;
        const bool live = e < total;
        float4 rec = base == 0 ? first : row[min(e, cap - 1)];
        if (!live) rec = make_float4(1.f, 0.f, 0.f, __int_as_float(i));
        const int word = __float_as_int(rec.w), j = word & kIdMask;
        // everything that depends on the neighbour's id, in flight together: its gradient row, and (first pass: the angular
        // neighbours are the first na <= CAPA <= 64 of the row) the pieces of the id rows for the reverse lookup of the angular
        // legs -- lane l scans piece (l % QL) of the id rows of angular neighbours l / QL + RPP t, one 16-byte load per lane and t,
        // four t in flight; whoever finds this atom in a neighbour's id row (rows are padded with -1) requests that leg, and the
        // radial arithmetic below runs while the legs are on their way (the partial forces meet in the wave sum): the kernel is a
        // chain of dependent round trips, and this order makes it row -> {gradient rows, id rows} -> legs with the arithmetic inside
        // the last one, where it was row -> gradient rows -> arithmetic -> id rows -> legs.
        const float4* grow = reinterpret_cast<const float4*>(radial_grad + (size_t)j * ld_radial + col);
        float4 gj[NR4];
#pragma unroll
        for (int c = 0; c < NR4; c++) gj[c] = grow[c];
        // (id rows of 64 slots keep the old order -- lookups behind the arithmetic: early they cost 34 spilled registers -- unless
        //  registers are free, LAT)
        constexpr bool EARLY = CAPA == 32 || LAT;
        constexpr int IF = LAT ? (NT < 16 ? NT : 16) : 4;     // pieces in flight per lane in the early round
        float4 leg[IF];
#pragma unroll
        for (int t = 0; t < IF; t++) leg[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool look = !RECV && base == 0 && na > 0;
        if (EARLY && look) {
            int4 idv[IF];
            int jt[IF];
#pragma unroll
            for (int t = 0; t < IF; t++) {
                const int et = lane / QL + RPP * t;
                jt[t] = __shfl(j, et & 63, 64);
                idv[t] = make_int4(-1, -1, -1, -1);
                if (t < NT && et < na) idv[t] = reinterpret_cast<const int4*>(ids + (size_t)jt[t] * CAPA)[lane % QL];
            }
#pragma unroll
            for (int t = 0; t < IF; t++) {
                int slot = -1;
                slot = idv[t].x == i ? 0 : slot;
                slot = idv[t].y == i ? 1 : slot;
                slot = idv[t].z == i ? 2 : slot;
                slot = idv[t].w == i ? 3 : slot;
                if (slot >= 0) leg[t] = leg_force[(size_t)jt[t] * CAPA + 4 * (lane % QL) + slot];
            }
        }
        const float r = fast_sqrt(rec.x * rec.x + rec.y * rec.y + rec.z * rec.z);
        const float rinv = fast_rcp(r);
        float sn, cs;
        sincospi_unit(r * inv_rcr, sn, cs);
        const float fc2 = -cs - 1.0f, dfc = -(0.5f * kPi * inv_rcr) * sn;       // fc2 = -2 fc
        const float4* own = reinterpret_cast<const float4*>(g_own + (word >> kTagShift) * NR);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NR4; c++) {
            const float4 o = own[c];
            const float d[4] = {o.x + gj[c].x, o.y + gj[c].y, o.z + gj[c].z, o.w + gj[c].w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int k = 4 * c + t;                   // compile-time: the parameters are scalar operands
                const float sh = r - P->rad_rs[k];
                const float ex = fast_exp2(P->rad_c[k] * sh * sh);
                s = fmaf(d[t], fmaf(fc2 * sh, P->rad_eta[k], dfc) * ex, s);
            }
        }
        s = live ? s * P->radial_scale * rinv : 0.f;
        fx -= s * rec.x; fy -= s * rec.y; fz -= s * rec.z;
#pragma unroll
        for (int t = 0; t < IF; t++) { fx += leg[t].x; fy += leg[t].y; fz += leg[t].z; }
        if (look) {                                            // id rows of more than four passes (CAPA = 64 with many angular neighbours)
            for (int t0 = EARLY ? IF : 0; t0 < NT && t0 * RPP < na; t0 += 4) {
                int4 idv[4];
                int jt[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int et = lane / QL + RPP * (t0 + t);
                    jt[t] = __shfl(j, et & 63, 64);
                    idv[t] = make_int4(-1, -1, -1, -1);
                    if (et < na) idv[t] = reinterpret_cast<const int4*>(ids + (size_t)jt[t] * CAPA)[lane % QL];
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    int slot = -1;
                    slot = idv[t].x == i ? 0 : slot;
                    slot = idv[t].y == i ? 1 : slot;
                    slot = idv[t].z == i ? 2 : slot;
                    slot = idv[t].w == i ? 3 : slot;
                    if (slot >= 0) {
                        const float4 f = leg_force[(size_t)jt[t] * CAPA + 4 * (lane % QL) + slot];
                        fx += f.x; fy += f.y; fz += f.z;
                    }
                }
            }
        }
    }
    if (RECV && lane < na) { fx += own_leg.x; fy += own_leg.y; fz += own_leg.z; }
    fx = wave_sum_lane63(fx); fy = wave_sum_lane63(fy); fz = wave_sum_lane63(fz);
    if (lane == 63) {
        if (na >= 2) { fx += centre.x; fy += centre.y; fz += centre.z; }
        pos_grad[3 * i] = fx;
        pos_grad[3 * i + 1] = fy;
        pos_grad[3 * i + 2] = fz;
    }
}

}  // namespace nnpops
