// ani_build_forward.h -- neighbour build + radial AEV + angular AEV of an atom in ONE workgroup.
//
// The angular forward of atom i needs nothing but what the neighbour build of atom i has just produced (sorted
// records, bucket offsets, triple list): the dependency between the two kernels is per atom, yet as two launches it
// costs a device-wide boundary (~2.3 us of dependent-launch latency), a second ramp-up and a second tail (10 000
// per-atom workgroups are ~1.4 occupancy rounds: each kernel ends with the chip a third full), and a round trip of the
// records and the triple list through L2.  Here a 128-lane workgroup owns the atom from the stencil walk to the
// stores of its AEV row:
//     wave 0: candidate scan                     | wave 1: waits
//     wave 0: species sort, records, triple list | wave 1: neighbour row to memory + radial AEV        (concurrently)
//     both:   angular forward (MfmaForward, WPA = 2) on the records / triple list left in LDS
// What backward needs (rows, records, ids, triple list, bucket offsets, counts) is still written to memory, once.
//
// Semantics: reference src/ani/CpuANISymmetryFunctions.cpp:61-135 (neighbours), :137-151 (radial), :153-194 (angular).
#pragma once

#include "ani_angular_mfma.h"

namespace nnpops {

// LDS of one workgroup: recA | recB | max(forward staging, builder scratch + 4 shared ints) | triple list
template <int NFRP, int NFZP>
__host__ __device__ inline size_t build_forward_lds_bytes(int cap, int capA, int S, int NB, int CH, int* tri_offset = nullptr) {
    const size_t recs = (size_t)capA * 2 * sizeof(float4);
    const size_t staging = (size_t)(CH + 1) * (NFRP + NFZP) * sizeof(float) + 64 * sizeof(int);      // (+ the quad table of the balanced phase 2)
    const size_t builder = ((builder_lds_bytes(cap, S, NB) + 15) & ~(size_t)15) + 32;
    const size_t mid = ((staging > builder ? staging : builder) + 15) & ~(size_t)15;
    if (tri_offset) *tri_offset = (int)(recs + mid);
    return recs + mid + (size_t)triples_capacity(capA) * sizeof(int);
}

struct BuildInputs {
    // cell-grid search
    const float* box;
    const CellGrid* grid;
    const int* cell_start;
    const int* sorted_cell;
    const float4* sorted_pos;
    int* cell_hist;
    // all-pairs search
    const float* pos;
    const int* species;
    const int2* segment;
    int use_cells, periodic;
};

struct BuildOutputs {
    float4* nbr;
    float4 *recA, *recB;
    int *ids, *tri, *cnt_a, *cnt_ro, *cnt_pos, *status;      // (nbr rows and cnt_pos go by position: slot in cell order, or the atom index)
    float* radial;
    int ld_radial;
};

template <bool TORCHANI, int NFRP, int NFZP, int OCC, int UNI = 0, bool DYN = false>
__global__ __launch_bounds__(128, OCC) void ani_build_forward(const AniParams* __restrict__ P, BuildInputs in, BuildOutputs out, int cap,
                                                              int capA, int CH, float* __restrict__ angular, int ld_angular,
                                                              int vec_ok, int tri_offset, int w0, int nw) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int role = __builtin_amdgcn_readfirstlane(wave_in_group());
    const int lane = lane_id();
    // recA | recB | forward staging; the builder's scratch shares the staging area (dead before phase 1 of the forward writes there)
    float4* recA_l = (float4*)lds_raw;
    float4* recB_l = recA_l + capA;
    float4* stage = recB_l + capA;
    float* rscratch = (float*)(stage + cap);
    int* groups = (int*)(rscratch + 3 * cap + 64 * P->S);
    const AtomGroups G = carve_groups(groups, P->S, P->NB);
    int* shared = groups + ((group_ints(P->S, P->NB, cap) + 3) & ~(size_t)3);       // [0] angular count, [1] radial-only count
    int* tri_l = (int*)(lds_raw + tri_offset);
    clear_cell_histogram(in.use_cells ? in.cell_hist : nullptr);

    const float rcr2 = P->rcr2, rca2 = P->rca2;
    Box b{};
    if (in.periodic) b = load_box(in.box);
    CellGrid g{};
    if (in.use_cells) g = *in.grid;

    {   // one atom per workgroup (no grid-stride loop: every kernel argument would stay live across the iterations)
        const int w = blockIdx.x;
        const int slot_id = w0 + w;                            // cell order (grid) or atom index (all pairs)
        int i = slot_id;
        float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool void_grid = in.use_cells && !g.ok;
        if (in.use_cells && !void_grid) {
            me = in.sorted_pos[slot_id];
            i = __float_as_int(me.w) & kIdMask;
        } else if (!in.use_cells) {
            me = make_float4(in.pos[3 * i], in.pos[3 * i + 1], in.pos[3 * i + 2], 0.f);
        }
        if (role == 0) {
            int na = 0, nro = 0;
            if (void_grid) {                                   // box too small for the stencil: tell the host
                if (lane == 0 && slot_id == 0) atomicOr(&out.status[kStatOverflow], g.bin_overflow ? 6 : 2);   // 4: grow the cell bins
            } else if (in.use_cells) {
                const int c = in.sorted_cell[slot_id];
                int cx, cy, cz;
                split_cell(g, c, cx, cy, cz);
                const WideStencil st = gather_wide_stencil(g, in.cell_start, cx, cy, cz);
                int* strip = (int*)rscratch;                   // the radial scratch is idle during the scan
                int carry = 0;
                constexpr int GROUP = 4;
                for (int base = 0; base < st.total; base += 64 * GROUP) {
                    float4 pj[GROUP];
#pragma unroll
                    for (int b4 = 0; b4 < GROUP; b4++)
                        if (base + 64 * b4 < st.total) pj[b4] = in.sorted_pos[wide_stencil_slot(st, base + 64 * b4, strip, carry)];
#pragma unroll
                    for (int b4 = 0; b4 < GROUP; b4++) {
                        if (base + 64 * b4 >= st.total) break;         // wave-uniform
                        const float4 cur = pj[b4];
                        const int word = __float_as_int(cur.w);
                        float dx = cur.x - me.x, dy = cur.y - me.y, dz = cur.z - me.z;
                        if (in.periodic) min_image<true>(dx, dy, dz, b);
                        const float r2 = dx * dx + dy * dy + dz * dz;
                        const bool in_r = (base + 64 * b4 + lane < st.total) & ((word & kIdMask) != i) & (r2 < rcr2);
                        const bool in_a = in_r & (r2 < rca2);
                        append_to_row(cap, stage, in_a, in_r & !in_a, dx, dy, dz, word, na, nro);
                    }
                }
            } else {
                // the reference's O(N^2) search; batched molecules: an atom only sees the atoms of its own molecule
                const int lo = in.segment ? in.segment[i].x : 0, hi = in.segment ? in.segment[i].y : P->N;
                for (int base = lo; base < hi; base += 64) {
                    const int j = base + lane;
                    bool in_r = false, in_a = false;
                    int word = 0;
                    float dx = 0.f, dy = 0.f, dz = 0.f;
                    if (j < hi && j != i) {
                        word = j | (in.species[j] << kTagShift);
                        dx = in.pos[3 * j] - me.x; dy = in.pos[3 * j + 1] - me.y; dz = in.pos[3 * j + 2] - me.z;
                        if (in.periodic) min_image<true>(dx, dy, dz, b);
                        const float r2 = dx * dx + dy * dy + dz * dz;
                        in_r = r2 < rcr2;
                        in_a = in_r && (r2 < rca2);
                    }
                    append_to_row(cap, stage, in_a, in_r && !in_a, dx, dy, dz, word, na, nro);
                }
            }
            if (lane == 0) {
                out.cnt_a[i] = na; out.cnt_ro[i] = nro;
                out.cnt_pos[slot_id] = pack_cnt_pos(na, nro, in.use_cells ? (__float_as_int(me.w) >> kTagShift) : in.species[i]);
                shared[0] = na; shared[1] = nro;
                if (na > capA || na + nro > cap) atomicOr(&out.status[kStatOverflow], 1);      // (ani_kernels.h: builders flag their own overflow)
                else if (P->class_tile[i] != 255 && na > (int)P->class_tile[i]) atomicOr(&out.status[kStatOverflow], 8);
            }
        }
        __syncthreads();
        int n, nro_c;
        const int na_raw = shared[0], nro_raw = shared[1];
        clamp_counts(na_raw, nro_raw, cap, capA, n, nro_c);
        if (!void_grid) {
            if (role == 1) {
                flush_row(out.nbr + (size_t)slot_id * cap, stage, cap, na_raw, nro_raw);
                radial_forward_from_lds(P, stage, cap, n, nro_c, rscratch, out.radial + (size_t)i * out.ld_radial);
            } else {
                finalize_angular(P, stage, n, out.recA + (size_t)i * capA, out.recB + (size_t)i * capA, out.ids + (size_t)i * capA, capA,
                                 out.tri + (size_t)i * triples_capacity(capA), P->bucket_offsets + (size_t)i * (P->NB + 1), G,
                                 recA_l, recB_l, tri_l);
            }
        }
        __syncthreads();
        // (the forward's constants are set up here, not before the build: they would be live across it -- 32 scalars the
        //  build has no room for)
        MfmaForward<TORCHANI, NFRP, NFZP, 2, UNI, DYN> F;
        F.init(P, capA, CH, vec_ok, angular, ld_angular, lds_raw, role);
        F.atom(i, n, [&](int t) { return tri_l[t]; }, [&](int bk) { return G.boff[bk]; }, [&]() { F.write_zero_record(); });
    }
}

}  // namespace nnpops
