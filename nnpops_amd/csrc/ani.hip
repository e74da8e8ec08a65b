// ani.hip -- C-ABI entry points for the ANI symmetry functions (see include/nnpops_hip.h).
//
// Host-side counterpart of the reference's ANISymmetryFunctions object
// (src/ani/ANISymmetryFunctions.h:41-154): construction parameters are frozen in the handle,
// compute() leaves positions / box / neighbour rows behind for backprop().
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ani_kernels.h"
#include "ani_fallback_kernels.h"
#include "ani_angular_mfma.h"
#include "ani_angular_bwd.h"
#include "ani_angular_generic.h"
#include "ani_radial_bwd.h"
#include "ani_build_forward.h"
#include "host_common.h"

using namespace nnpops;

constexpr int kRbwdLatencyAtoms = 8192;  // up to this many atoms the radial backward runs its latency variant (ani_radial_bwd.h: LAT)
constexpr int kFuseAtoms = 4096;       // systems of up to this many atoms build and run the angular forward in one workgroup (ani_build_forward.h)
struct nnpops_ani {
    AniParams hp{};                 // host copy of the parameter block
    AngularConsts ac{};             // what the angular backward kernel needs of it, passed by value (ani_kernels.h)
    AniParams* d_params = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    int nfrp = 0, nfzp = 0;         // padded factor counts selecting the kernel instantiation
    int algorithm = 0;              // 0 auto, 1 all-pairs, 2 cell list
    // device state
    int32_t* d_species = nullptr;
    int2* d_segment = nullptr;      // [N] per-atom [lo, hi) of its molecule (batched handles only)
    float4* d_nbr = nullptr;        // [N][cap] records {dx, dy, dz, (species<<24)|atom}, rows by POSITION of the radial backward's walk (slot in cell order / atom index)
    int* d_cnt_pos = nullptr;       // [N] na | nro << 16, by the same position
    float4* d_recA = nullptr;       // [N][cap_angular] sorted angular records {dx,dy,dz,r}
    float4* d_recB = nullptr;       // [N][cap_angular]                        {fc,dfc,1/r,word}
    int* d_ids = nullptr;           // [N][cap_angular] atom ids in record order (reverse lookup of the backward gather)
    float4* d_leg_force = nullptr;  // [N][cap_angular] angular backward: force on each leg, record order
    float4* d_centre_force = nullptr; // [N]            angular backward: reaction on the centre atom
    int* d_tri = nullptr;           // [N][cap_angular*(cap_angular-1)/2] bucket-major triple words
    int* d_bucket_offsets = nullptr; // [N][NB + 1] first triple of every bucket (chunked forward view)
    bool chunked_forward = true;
    int forward_kernel = 2;         // 2: matrix-core scatter (ani_angular_mfma.h), 1: chunked view, 0: run merging
    bool generic = false;           // the angular functions do not factor (or have too many factors): generic kernels
    bool mfma_ok = false;           // at most 32 species pairs can occur in this system
    bool fwd_identity = false;      // angular function m sits at canonical slot m: 16-byte stores of the row
    int fwd_chunk = 192;            // triples staged in LDS per chunk of the matrix-core forward kernel (large systems; forward_chunk())
    bool fwd_chunk_forced = false;  // $NNPOPS_ANI_FWD_CHUNK given
    int fwd_waves_per_atom = 2;     // 2: a 128-lane workgroup per atom (half the LDS per wave), 1: a wave per atom
    // Atoms are evaluated in `nstreams` spans on as many HIP streams (fork after the cell grid, join before the caller's
    // stream continues): the per-atom kernels of a span only depend on the same span of the kernel before, so the ramp and
    // the tail of every launch overlap with the steady state of the other spans' launches (two half-size evaluations on two
    // streams finish in 0.83x the time of one full-size evaluation on one stream, tools/two_streams.py).
    bool backward_forced = false;   // $NNPOPS_ANI_BACKWARD given: no automatic choice of the two-wave kernel for dense systems
    bool fwd_uniform = false;       // every radial factor shares its eta, every angular factor its zeta (set at create; $NNPOPS_ANI_FWD_UNI=0)
    bool fwd_grid = true;           // ... and eight radial factors sit on equally spaced shifts (what the UNI forward kernel assumes of eight)
    bool bwd_literal = true;        // ($NNPOPS_ANI_BWD_LITERAL=0: the backward kernel keeps its constants in registers)
    bool fwd_literal = false;       // ... and every derived constant equals the compiled-in ANI-2x set bit for bit (Ani2xAngular): literal kernels
    int fuse_forward = -1;          // neighbour build and angular forward of an atom in one workgroup (ani_build_forward.h).  -1: for
                                    // systems of up to kFuseAtoms atoms, where a launch less is worth 6-13 % of a step (600 atoms:
                                    // 33.7 -> 29.6 us, 3 000: 49.5 -> 46.3) -- at 5 000 it breaks even and at 10 000 its lower
                                    // occupancy makes it equal to the two launches; $NNPOPS_ANI_FUSE=0 / 1 forces
    bool rbwd_lanes = true;         // radial backward with a lane per neighbour (ani_radial_bwd.h) where rows read as float4
    int scatter_mode = -1;          // leg forces stored in the receiving atom's row by the two-wave angular backward (ani_angular_bwd.h):
                                    // -1 where every backward launch runs two waves per atom (dense systems), 0 / 1 forced ($NNPOPS_ANI_SCATTER)
    bool scatter_now = false;       // ... decided by backprop() for the call in progress
    bool last_fused_build = false;  // the last compute() ran the one-launch build + forward (what describe() reports)
    bool fine_grid = true;          // cell grid of half-cutoff cells where it fits (celllist.h: decide_grid)
    bool fwd_row_via_lds = true;    // the angular row leaves as whole-wave stores from an LDS copy
    int fwd_occ = 7;                // A/B: register budget of the forward kernel (waves per SIMD)
    bool fwd_dynamic = false;       // quads dealt out per atom from its bucket sizes (ani_angular_mfma.h: DYN): set at create from the
                                    // composition (forward_dynamic_pays), $NNPOPS_ANI_FWD_DYN=0 / 1 forces
    int nstreams = 1;
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    bool can_split = false;         // the selected kernels take atom ranges
    int bwd_atoms_per_group = 1;    // one-wave backward kernel: atoms (waves) per workgroup
    int store_mode = 3;             // A/B: 0 plain, 1 sc1, 2 sc0 sc1, 3 nt stores of the angular rows
    bool occ6 = false;              // register budget of the two-wave kernels: 6 (80 VGPRs) or 5 (96) waves per SIMD
    int backward_kernel = 1;        // 1: two waves per atom, packed arithmetic (ani_angular_bwd.h), 0: the one-wave kernel
    int fwd_atoms_per_group = 1;    // > 1: every wave / workgroup walks that many atoms (amortises its prologue)
    int* d_cnt_a = nullptr;         // [N]
    int* d_cnt_ro = nullptr;        // [N]
    int* d_status = nullptr;        // [kStatAlloc]: check()'s kStatWords words, then the class launches' flag (ani_kernels.h)
    int* h_status = nullptr;        // pinned, device-visible host words {stamp, overflow} (nnpops_ani_check_begin / _end)
    int* h_status_dev = nullptr;    // the device's address of the same words
    int check_stamp = 0;
    bool check_pending = false;
    // cell grid (celllist.h)
    CellGrid* d_grid = nullptr;
    int* d_cell_count = nullptr;    // [max_cells]
    int* d_cell_start = nullptr;    // [max_cells+1]
    int* d_atom_cell = nullptr;     // [N]
    int* d_atom_rank = nullptr;     // [N]
    int* d_sorted_cell = nullptr;   // [N] cell of the atom in every sorted slot
    int* d_tile_total = nullptr;    // [max_cells / 8192 + 1] partial sums of the parallel cell scan
    int* d_unsorted_atom = nullptr; // [N] cell segments in arrival order
    int* d_sorted_atom = nullptr;   // [N]
    float4* d_sorted_pos = nullptr; // [N]
    int max_cells = 0;
    int* d_hist = nullptr;          // two-kernel cell build (celllist.h): [kBinnedCells + 1]
    int* d_bins = nullptr;          // [kBinnedCells][bin_cap]
    int bin_cap = 64;
    bool cells_disabled = false;    // set when a box turned out too small for the 27-cell stencil
    int cap = 0;                    // row capacity (angular + radial-only neighbours)
    bool cap_fitted = false;        // the first clean check() shrinks `cap` to the system
    int cap_angular = 0;            // LDS capacity of the angular kernels
    int tile = 32;                  // pair-matrix edge of the angular backward kernel (<= 32, sized in check())
    bool compact_bwd = false;       // check() saw no atom with more than `tile` angular neighbours: compact LDS layout
    bool computed = false;
    // optional per-kernel HIP-event timing (nnpops_ani_enable_timing)
    int ld_radial = 0, ld_angular = 0;   // row strides (floats) of the AEV / gradient arrays of the call in progress
    bool last_used_cells = false;   // the last compute() built a cell grid (d_sorted_atom is a permutation in cell order)
    int* d_work_order = nullptr;    // [N] atoms by DECREASING number of angular neighbours (check() builds it): the schedule of the
    bool work_order_valid = false;  //     angular kernels -- heaviest atoms first, the light ones fill the tail
    unsigned char* d_class_tile = nullptr;   // [N] pair-matrix edge of the backward launch every atom belongs to (255: no limit)
    struct BwdClass { int tile, w0, nw; bool two_waves; };
    std::vector<BwdClass> bwd_classes;       // stretches of d_work_order, by decreasing tile (check() builds them with the order)
    bool bwd_by_class = true;                // $NNPOPS_ANI_BWD_CLASSES=0: one launch with the full-size pair matrix
    bool bwd_two_waves = false;              // the atoms average 200 triples or more (check()): two waves per atom in the backward kernel
    double mean_triples = 0;                 // triples per atom of the frame check() last looked at
    int bwd_class_min = 512;                 // smallest class launched on its own ($NNPOPS_ANI_BWD_CLASS_MIN)
    int bwd_class_atoms = 16384;             // systems of fewer atoms take one backward launch ($NNPOPS_ANI_BWD_CLASS_ATOMS)
    int cell_atoms = 1800;          // systems of at least this many atoms search their neighbours through the cell grid ($NNPOPS_ANI_CELL_ATOMS)
    int lpt = 2;                    // $NNPOPS_ANI_LPT: 0 off, 1 only where there is no cell order, 2 (default) also instead of the cell order
    int backprop_stamp = 0;         // counts backprop() calls (never 0): what the class launches write to the class flag (ani_angular_bwd.h)
    unsigned timing_mask = 0;       // bit k: kernel id k is bracketed by events
    int timing_every = 1;           // ... on every timing_every-th launch
    bool timing_merge = false;      // nnpops_ani_set_timing_merge: ONE bracket around neighbour build + angular forward (reported as the
                                    // build) and ONE around angular + radial backward (reported as the angular backward); the sum of
                                    // the two single brackets minus the merged one is what a bracket costs, measured in place
    unsigned timing_seen[NNPOPS_ANI_NUM_KERNELS] = {};
    std::vector<hipEvent_t> ev_start[NNPOPS_ANI_NUM_KERNELS], ev_stop[NNPOPS_ANI_NUM_KERNELS];
    size_t ev_used[NNPOPS_ANI_NUM_KERNELS] = {};
};

namespace {

// Brackets one kernel launch with a pair of events on the handle's stream when timing is on.
struct KernelTimer {
    nnpops_ani* h;
    int id;
    hipStream_t stream;
    KernelTimer(nnpops_ani* h_, int id_, hipStream_t s_ = nullptr, bool merged = false) : h(h_), id(id_), stream(s_ ? s_ : h_->stream) {
        if (!(h->timing_mask >> id & 1)) return;
        if (h->timing_merge != merged && id != NNPOPS_ANI_K_CELL_GRID) return;      // (merge mode: only the two merged brackets and the grid's)
        if (h->timing_seen[id]++ % (unsigned)h->timing_every != 0) return;
        active = true;
        if (h->ev_used[id] == h->ev_start[id].size()) {
            hipEvent_t a, b;
            // timing-only events: no system-scope fence (and L2 write-back) when they complete
            (void)hipEventCreateWithFlags(&a, hipEventDisableSystemFence);
            (void)hipEventCreateWithFlags(&b, hipEventDisableSystemFence);
            h->ev_start[id].push_back(a);
            h->ev_stop[id].push_back(b);
        }
        slot = h->ev_used[id]++;
        (void)hipEventRecord(h->ev_start[id][slot], stream);
    }
    size_t slot = 0;
    bool active = false;
    ~KernelTimer() {
        if (!active) return;
        (void)hipEventRecord(h->ev_stop[id][slot], stream);
    }
};

// Waves per workgroup for a per-atom kernel that needs `lds_wave` bytes of LDS per wave: the largest of
// {4, 2, 1} that does not lower the number of waves a CU can hold (32 wave slots; 160 KiB of LDS handed out in 128 pieces of
// 1 280 bytes -- a workgroup occupies whole pieces, round 5).
int waves_per_group(size_t lds_wave) {
    auto resident = [&](int wpg) { return (int)std::min<size_t>(32, (128 / std::max<size_t>(1, (lds_wave * wpg + 1279) / 1280)) * wpg); };
    int best = 1;
    for (int wpg : {2, 4})
        if (resident(wpg) >= resident(best)) best = wpg;
    return best;
}

// Triples staged per chunk of the matrix-core forward.  192 is the optimum of the 10 000-atom liquid, where LDS per atom is
// occupancy (128 -> 20.9 us, 192 -> 19.0, 256 -> 20.2).  A system small enough for ALL its atoms to be resident at once has no
// occupancy to lose: there the chunk is as large as the whole triple list of the busiest atom the records allow (one phase 1, one
// phase 2, two barriers instead of six for a 300-triple atom): the 50-atom molecule of BASELINE config 1 36.3 -> 31.1 us per
// forward+backward evaluation at 512 (round 4).  fixed_lds_bytes: what the workgroup needs besides the staged factors.
int forward_chunk(const nnpops_ani* h, size_t fixed_lds_bytes, size_t bytes_per_triple) {
    if (h->fwd_chunk_forced) return h->fwd_chunk;
    const int want = std::min(512, (triples_capacity(h->cap_angular) + 15) & ~15);
    for (int ch : {want, 384, 256}) {
        if (ch > want || ch <= h->fwd_chunk) continue;
        const size_t lds = fixed_lds_bytes + (size_t)(ch + 1) * bytes_per_triple;
        if (lds <= 160 * 1024 && h->hp.N <= 256L * (long)(160 * 1024 / lds)) return ch;
    }
    // dense systems (compact molecules: 400+ triples per atom): 256 triples per chunk are two passes where 192 are three -- a
    // 7 600-atom block of the conformer batch 40.1 -> 36.5 us, the whole batch equal
    // (round 5: 240, not 256 -- with 64-slot records 241 staged triples are 11 of the CU's 1 280-byte LDS pieces, eleven workgroups per
    //  CU, 257 are 12 pieces and ten: forward of a 7 600-atom block 31.1 -> 30.6 us, of the whole batch 254.5 -> 250.2)
    if (h->mean_triples >= 256.0 && h->fwd_chunk < 240 && fixed_lds_bytes + 241 * bytes_per_triple <= 160 * 1024) return 240;
    return h->fwd_chunk;
}

int pad_pow2(int n, int lo) {
    int p = lo;
    while (p < n) p <<= 1;
    return p;
}

// Split the angular set into {(eta,rs)} x {(zeta,thetas)}; fills hp.fr_*, hp.fz_*, hp.m_of.
int factor_angular(AniParams& hp, const float* af, int nA) {
    std::vector<std::pair<float, float>> fr, fz;
    std::vector<int> ia(nA), iz(nA);
    for (int m = 0; m < nA; m++) {
        const std::pair<float, float> r{af[4 * m], af[4 * m + 1]}, z{af[4 * m + 2], af[4 * m + 3]};
        size_t a = 0, q = 0;
        while (a < fr.size() && fr[a] != r) a++;
        if (a == fr.size()) fr.push_back(r);
        while (q < fz.size() && fz[q] != z) q++;
        if (q == fz.size()) fz.push_back(z);
        ia[m] = (int)a;
        iz[m] = (int)q;
    }
    const int nFR = (int)fr.size(), nFZ = (int)fz.size();
    if (nFR * nFZ != nA || nFR > kMaxFactor || nFZ > 8) return 1;      // not a (small enough) full grid: generic kernels
    std::vector<int> seen(nA, -1);
    const int nfzp = pad_pow2(nFZ, 4);
    for (int m = 0; m < nA; m++) {
        const int c = ia[m] * nFZ + iz[m];
        if (seen[c] >= 0) return 1;                            // a duplicated function: not a grid either
        seen[c] = m;
        hp.c_of_m[m] = ia[m] * nfzp + iz[m];
        hp.scale_m[m] = powf(2.0f, 1.0f - af[4 * m + 2]);
    }
    hp.nFR = nFR;
    hp.nFZ = nFZ;
    for (int a = 0; a < nFR; a++) {
        hp.fr_eta[a] = fr[a].first;
        hp.fr_rs[a] = fr[a].second;
        hp.fr_c[a] = -fr[a].first * kLog2e;
    }
    for (int z = 0; z < nFZ; z++) {
        hp.fz_zeta[z] = fz[z].first;
        hp.fz_cos[z] = (float)std::cos((double)fz[z].second);
        hp.fz_sin[z] = (float)std::sin((double)fz[z].second);
        hp.fz_bias[z] = 1.0f - fz[z].first;
    }
    return NNPOPS_OK;
}

int alloc_rows(nnpops_ani* h) {
    dev_free(h->d_nbr); dev_free(h->d_recA); dev_free(h->d_recB); dev_free(h->d_tri); dev_free(h->d_ids); dev_free(h->d_leg_force);
    int rc;
    if ((rc = dev_alloc(&h->d_nbr, (size_t)h->hp.N * h->cap))) return rc;
    if ((rc = dev_alloc(&h->d_recA, (size_t)h->hp.N * h->cap_angular))) return rc;
    if ((rc = dev_alloc(&h->d_recB, (size_t)h->hp.N * h->cap_angular))) return rc;
    if ((rc = dev_alloc(&h->d_ids, (size_t)h->hp.N * h->cap_angular))) return rc;
    if ((rc = dev_alloc(&h->d_leg_force, (size_t)h->hp.N * h->cap_angular))) return rc;
    if ((rc = dev_alloc(&h->d_tri, (size_t)h->hp.N * triples_capacity(h->cap_angular)))) return rc;
    {   // the walk of the builders' triple loop follows the footprint of the lists (ani_kernels.h: decode_pair_folded)
        int row_major = (size_t)h->hp.N * triples_capacity(h->cap_angular) * sizeof(int) > ((size_t)256 << 20) ? 1 : 0;
        if (const char* e = std::getenv("NNPOPS_ANI_TRI_ROW_MAJOR")) row_major = std::atoi(e) != 0;
        if (row_major != h->hp.tri_row_major && h->d_params) {
            h->hp.tri_row_major = row_major;
            NNPOPS_HIP_TRY(hipMemcpy(h->d_params, &h->hp, sizeof(AniParams), hipMemcpyHostToDevice));
        }
        h->hp.tri_row_major = row_major;
    }
    return NNPOPS_OK;
}

// A contiguous stretch of the atoms, in cell order when the build used the cell grid (order = sorted atom ids) or in index
// order (order = NULL), and the stream its kernels are launched on.
struct Span {
    hipStream_t stream;
    const int* order;
    int w0, nw;
    const int* ang_order;      // what the two angular kernels walk: the work-sorted schedule (check() builds it), else `order`
};

// The radial backward with a lane per neighbour (ani_radial_bwd.h) takes this call's gradient array?
bool radial_lanes(const nnpops_ani* h, const float* radial_deriv) {
    const int nr4 = h->hp.nR / 4;
    return h->rbwd_lanes && h->hp.nR % 4 == 0 && nr4 >= 1 && nr4 <= 8 && h->ld_radial % 4 == 0 &&
           (h->cap_angular == 32 || h->cap_angular == 64) && (reinterpret_cast<uintptr_t>(radial_deriv) & 15) == 0;
}

// ---- kernel dispatch over (TORCHANI, NFRP, NFZP) ----
template <bool TA, int NFRP, int NFZP>
int launch_angular(nnpops_ani* h, bool forward, const float* grad_or_null, float* out, const Span& sp) {
    const int N = sp.nw;                                       // atoms of this launch
    const size_t lds = forward ? (h->chunked_forward ? ang_fwd_chunked_lds_bytes<NFRP, NFZP>(h->cap_angular, h->hp.NB)
                                                    : ang_fwd_lds_bytes<NFRP, NFZP>(h->cap_angular, h->hp.NB))
                               : ang_bwd_lds_bytes<NFRP, NFZP>(h->cap_angular, h->hp.NB, h->tile, h->compact_bwd);
    if (lds > 160 * 1024)
        return fail(NNPOPS_ERR_UNSUPPORTED, "angular kernel needs %zu bytes of LDS per wave (> 160 KiB)", lds);
    const int lds_wave = (int)((lds + 15) & ~(size_t)15);
    const int wpg = waves_per_group(lds_wave);
    const size_t lds_group = (size_t)lds_wave * wpg;
    const dim3 grid(div_up(N, wpg)), block(64 * wpg);
    if (forward && h->forward_kernel == 2) {
        const int CH = forward_chunk(h, (size_t)h->cap_angular * 2 * sizeof(float4), (size_t)(NFRP + NFZP) * sizeof(float));
        const size_t lds2 = ang_fwd_mfma_lds_bytes<NFRP, NFZP>(h->cap_angular, CH);
        const int lw = (int)((lds2 + 15) & ~(size_t)15);
        int vec_ok = h->fwd_identity && h->ld_angular % 4 == 0 && ((uintptr_t)out & 15) == 0;
        if (vec_ok) vec_ok |= (h->store_mode << 1) | (h->fwd_row_via_lds ? 8 : 0);     // (bits 1-2: flavour of the row stores, bit 3: row assembled in LDS)
        if (h->fwd_waves_per_atom == 2) {                      // a 128-lane workgroup per atom
            int groups = N;
            if (h->fwd_atoms_per_group > 1) groups = div_up(N, h->fwd_atoms_per_group);
            const bool uni = h->fwd_uniform && h->fwd_grid && h->hp.nFR == NFRP && h->hp.nFZ == NFZP;      // one eta, one zeta, no padded factor slots
            // balanced phase 2 (per-atom quad table, ani_angular_mfma.h: DYN): needs the row assembled in LDS and a lane per bucket
            const bool dyn = h->fwd_dynamic && (vec_ok & 8) && h->hp.NB <= 63 && h->hp.NB * h->hp.nA <= CH * (NFRP + NFZP);
            auto k = h->fwd_occ == 6 ? ani_angular_forward_mfma<TA, NFRP, NFZP, 2, 6> : h->fwd_occ == 8 ? ani_angular_forward_mfma<TA, NFRP, NFZP, 2, 8>
                   : dyn ? (uni ? ani_angular_forward_mfma<TA, NFRP, NFZP, 2, 7, 1, true> : ani_angular_forward_mfma<TA, NFRP, NFZP, 2, 7, 0, true>)
                   : uni ? ani_angular_forward_mfma<TA, NFRP, NFZP, 2, 7, 1> : ani_angular_forward_mfma<TA, NFRP, NFZP, 2, 7>;
            if constexpr (NFRP == 8 && NFZP == 4) {            // the published ANI-2x constants: kernels that carry them as literals
                if (uni && h->fwd_literal && h->fwd_occ != 6 && h->fwd_occ != 8)
                    k = dyn ? ani_angular_forward_mfma<TA, 8, 4, 2, 7, 2, true> : ani_angular_forward_mfma<TA, 8, 4, 2, 7, 2>;
            }
            if (lw > 64 * 1024) NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lw));
            hipLaunchKernelGGL(k, dim3(groups), dim3(128), (size_t)lw, sp.stream, h->d_params, h->cap, h->cap_angular, CH, h->d_recA,
                               h->d_recB, h->d_tri, h->d_cnt_a, h->d_cnt_ro, out, h->ld_angular, vec_ok, lw, sp.ang_order, sp.w0, sp.nw);
        } else {
            const int wpg2 = waves_per_group(lw);
            const size_t lg = (size_t)lw * wpg2;
            int groups = div_up(N, wpg2);
            if (h->fwd_atoms_per_group > 1) groups = div_up(groups, h->fwd_atoms_per_group);
            auto k = ani_angular_forward_mfma<TA, NFRP, NFZP, 1, 5>;
            if (lg > 64 * 1024) NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lg));
            hipLaunchKernelGGL(k, dim3(groups), dim3(64 * wpg2), lg, sp.stream, h->d_params, h->cap, h->cap_angular, CH, h->d_recA,
                               h->d_recB, h->d_tri, h->d_cnt_a, h->d_cnt_ro, out, h->ld_angular, vec_ok, lw, sp.ang_order, sp.w0, sp.nw);
        }
    } else if (forward) {
        auto k = h->chunked_forward ? ani_angular_forward_chunked<TA, NFRP, NFZP> : ani_angular_forward<TA, NFRP, NFZP>;
        if (lds_group > 64 * 1024) NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_group));
        hipLaunchKernelGGL(k, grid, block, lds_group, h->stream, h->d_params, h->cap, h->cap_angular, h->d_recA, h->d_recB,
                           h->d_tri, h->d_cnt_a, h->d_cnt_ro, out, h->ld_angular, lds_wave);
    } else if (h->backward_kernel >= 1 && ang_bwd_pair_lds_bytes<NFRP, NFZP>(h->cap_angular, h->hp.NB, true) <= 160 * 1024) {
        // backward_kernel: 1 = one wave per atom, every triple reads its gradient block through the L1 (needs the 16-byte
        // layout); 2 = one wave, gradient row staged in LDS; 3 / 4 = the same two with two waves per atom (A/B only)
        const int vec_ok = h->fwd_identity && h->ld_angular % 4 == 0 && ((uintptr_t)grad_or_null & 15) == 0;
        // One launch per class of atoms (by their number of angular neighbours, nnpops_ani_check), each with the pair matrix its
        // atoms need; without classes -- no work order yet, several spans, $NNPOPS_ANI_BWD_CLASSES=0 -- one launch at full size.
        nnpops_ani::BwdClass whole{h->cap_angular, sp.w0, sp.nw, h->bwd_two_waves};
        // (class launches take the kernels that read the gradient blocks through the L1 -- the ones with a CLASSES instantiation)
        const bool by_class = h->bwd_by_class && !h->bwd_classes.empty() && sp.ang_order == h->d_work_order && sp.w0 == 0 && sp.nw == h->hp.N &&
                              vec_ok && h->backward_kernel == 1 && !h->occ6;
        const nnpops_ani::BwdClass* classes = by_class ? h->bwd_classes.data() : &whole;
        const int nclasses = by_class ? (int)h->bwd_classes.size() : 1;
        static const bool debug_no_cleanup = std::getenv("NNPOPS_ANI_DEBUG_NO_CLEANUP") != nullptr;      // (timing experiments only: forces are wrong for outgrown atoms)
        for (int c = 0; c < nclasses + (by_class && !debug_no_cleanup ? 1 : 0); c++) {
        const nnpops_ani::BwdClass& cl = c < nclasses ? classes[c] : whole;
        const int tile = std::min(cl.tile, h->cap_angular), cw0 = cl.w0, cnw = cl.nw;
        if (cnw <= 0) continue;
        int mode = h->backward_kernel;
        // Dense systems (64 or more record slots): the pair matrix is 27 KB per atom and only 6 atoms fit a CU -- six waves
        // where twenty could run.  Two waves per atom double the waves on the same LDS (1 024 conformers: 1.03 -> 0.97 ms per
        // batch); with the usual 32 slots one wave per atom wins (section 3.5 of DESIGN.md).
        // (round 4: what decides is the work per atom, not the LDS -- with the classes above every class of the conformer batch, the
        //  32-slot one included, is faster with two waves per atom: 304 us in one launch, 275 by class with this rule on LDS, 246
        //  with two waves everywhere; the 153-triple atoms of a liquid stay with one wave, 15.8 against 25 us)
        if (mode == 1 && !h->backward_forced && (cl.two_waves || h->scatter_now || ang_bwd_pair_lds_bytes<NFRP, NFZP>(tile, h->hp.NB, false) > 16 * 1024)) mode = 3;
        if (!vec_ok && (mode == 1 || mode == 3)) mode++;
        const bool glds = mode == 2 || mode == 4;
        const size_t lb = (ang_bwd_pair_lds_bytes<NFRP, NFZP>(tile, h->hp.NB, glds) + 15) & ~(size_t)15;
        void (*k)(const AniParams*, const AngularConsts, int, int, int, const float4*, const float4*, const int*, const int*, const int*, const float*, int,
                  float4*, float4*, const int*, float4*, int, int, int, const int*, int, int, int) =
            mode == 1 ? (h->occ6 ? ani_angular_backward_pair<TA, NFRP, NFZP, 6, 1, false>
                         : (h->fwd_uniform && h->hp.nFR == NFRP && h->hp.nFZ == NFZP) ? ani_angular_backward_pair<TA, NFRP, NFZP, 5, 1, false, false, 1>
                                                                                       : ani_angular_backward_pair<TA, NFRP, NFZP, 5, 1, false>)
          : mode == 2 ? ani_angular_backward_pair<TA, NFRP, NFZP, 5, 1, true>
          : mode == 3 ? ((h->fwd_uniform && h->hp.nFR == NFRP && h->hp.nFZ == NFZP) ? ani_angular_backward_pair<TA, NFRP, NFZP, 5, 2, false, false, 1>
                                                                                       : ani_angular_backward_pair<TA, NFRP, NFZP, 5, 2, false>)
                      : ani_angular_backward_pair<TA, NFRP, NFZP, 5, 2, true>;
        if constexpr (NFRP == 8 && NFZP == 4) {                // the published ANI-2x constants as literals (Ani2xAngular)
            if (h->fwd_literal && h->bwd_literal && h->fwd_uniform && h->hp.nFR == 8 && h->hp.nFZ == 4 && !h->occ6 && (mode == 1 || mode == 3))
                k = mode == 1 ? ani_angular_backward_pair<TA, 8, 4, 5, 1, false, false, 2> : ani_angular_backward_pair<TA, 8, 4, 5, 2, false, false, 2>;
        }
        if (by_class && c < nclasses) {                        // (mode is 1 or 3 here) the launch of a class
            const bool uni = h->fwd_uniform && h->hp.nFR == NFRP && h->hp.nFZ == NFZP;
            k = mode == 1 ? (uni ? ani_angular_backward_pair<TA, NFRP, NFZP, 5, 1, false, false, 1, 1> : ani_angular_backward_pair<TA, NFRP, NFZP, 5, 1, false, false, 0, 1>)
                          : (uni ? ani_angular_backward_pair<TA, NFRP, NFZP, 5, 2, false, false, 1, 1> : ani_angular_backward_pair<TA, NFRP, NFZP, 5, 2, false, false, 0, 1>);
            if constexpr (NFRP == 8 && NFZP == 4) {
                if (h->fwd_literal && h->bwd_literal && uni && h->hp.nFR == 8 && h->hp.nFZ == 4)
                    k = mode == 1 ? ani_angular_backward_pair<TA, 8, 4, 5, 1, false, false, 2, 1> : ani_angular_backward_pair<TA, 8, 4, 5, 2, false, false, 2, 1>;
            }
        } else if (by_class) {                                 // the clean-up launch: one instantiation per wave count will do (its speed does not matter)
            k = mode == 1 ? ani_angular_backward_pair<TA, NFRP, NFZP, 5, 1, false, false, 0, 2> : ani_angular_backward_pair<TA, NFRP, NFZP, 5, 2, false, false, 0, 2>;
        }
        bool scat = false;
        if constexpr (NFRP == 8 && NFZP == 4) {                // leg forces into the receivers' rows: backprop() decided (scatter_now) -- two waves, literal constants
            if (h->scatter_now && mode == 3) {
                scat = true;
                k = !by_class ? ani_angular_backward_pair<TA, 8, 4, 5, 2, false, false, 2, 0, true>
                  : c < nclasses ? ani_angular_backward_pair<TA, 8, 4, 5, 2, false, false, 2, 1, true>
                                 : ani_angular_backward_pair<TA, 8, 4, 5, 2, false, false, 2, 2, true>;
            }
        }
        const int apg = mode >= 3 ? 1 : std::max(1, std::min(kWavesPerGroup, h->bwd_atoms_per_group));
        const int threads = mode >= 3 ? 128 : 64 * apg;
        if (lb * apg > 64 * 1024) NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lb * apg)));
        // (c == nclasses: the clean-up launch behind the classes -- every atom, full-size pair matrix, one workgroup per CU: 4.5 us when it has
        //  nothing to do;
        //  it returns at once unless a class launch left an atom out: ani_angular_bwd.h)
        const int class_mode = !by_class ? 0 : c < nclasses ? 1 : 2;
        const int groups = class_mode == 2 ? std::min(div_up(cnw, apg), 256) : div_up(cnw, apg);
        hipLaunchKernelGGL(k, dim3(groups), dim3(threads), lb * apg, sp.stream, h->d_params, h->ac, h->cap, h->cap_angular, tile, h->d_recA, h->d_recB, h->d_tri,
                           h->d_cnt_a, h->d_cnt_ro, grad_or_null, h->ld_angular, h->d_leg_force, h->d_centre_force, h->d_ids,
                           scat ? h->d_leg_force : nullptr, vec_ok, h->hp.NB, (int)lb,
                           sp.ang_order, cw0, cnw, h->backprop_stamp);
        }
    } else {
        auto k = ani_angular_backward<TA, NFRP, NFZP>;
        if (lds_group > 64 * 1024) NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_group));
        hipLaunchKernelGGL(k, grid, block, lds_group, h->stream, h->d_params, h->cap, h->cap_angular, h->tile, h->d_recA,
                           h->d_recB, h->d_tri, h->d_cnt_a, h->d_cnt_ro, grad_or_null, h->ld_angular, h->d_leg_force, h->d_centre_force,
                           lds_wave, (int)h->compact_bwd);
    }
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

template <bool TA>
int dispatch_factors(nnpops_ani* h, bool forward, const float* g, float* out, const Span& sp) {
    const int key = h->nfrp * 100 + h->nfzp;
    switch (key) {
        case 804:  return launch_angular<TA, 8, 4>(h, forward, g, out, sp);
#ifndef NNPOPS_ONLY_ANI2X_SHAPE      // (a development build instantiates the ANI-1x / 2x factor shape only: a quarter of the compile time)
        case 404:  return launch_angular<TA, 4, 4>(h, forward, g, out, sp);
        case 408:  return launch_angular<TA, 4, 8>(h, forward, g, out, sp);
        case 808:  return launch_angular<TA, 8, 8>(h, forward, g, out, sp);
        case 1604: return launch_angular<TA, 16, 4>(h, forward, g, out, sp);
        case 1608: return launch_angular<TA, 16, 8>(h, forward, g, out, sp);
#endif
        default:
            return fail(NNPOPS_ERR_UNSUPPORTED, "no angular kernel for %d x %d factors", h->hp.nFR, h->hp.nFZ);
    }
}

template <bool TA>
int launch_generic(nnpops_ani* h, bool forward, const float* g, float* out, const Span& sp) {
    const int N = h->hp.N;                                     // (the generic forward kernel has no ranges: spans are never split)
    if (forward) {
        const int lw = (int)((ang_fwd_generic_lds_bytes(h->cap_angular) + 15) & ~(size_t)15);
        const int wpg = waves_per_group(lw);
        hipLaunchKernelGGL(ani_angular_forward_generic<TA>, dim3(div_up(N, wpg)), dim3(64 * wpg), (size_t)lw * wpg, h->stream, h->d_params,
                           h->cap, h->cap_angular, h->d_recA, h->d_recB, h->d_tri, h->d_cnt_a, h->d_cnt_ro, out, h->ld_angular, lw);
    } else {
        const size_t lb = (ang_bwd_pair_lds_bytes<4, 4>(h->cap_angular, h->hp.NB, false) + 15) & ~(size_t)15;
        if (lb > 160 * 1024) return fail(NNPOPS_ERR_UNSUPPORTED, "generic angular backward needs %zu bytes of LDS (cap_angular %d)", lb, h->cap_angular);
        auto k = ani_angular_backward_pair<TA, 4, 4, 4, 1, false, true>;
        if (lb > 64 * 1024) NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
        hipLaunchKernelGGL(k, dim3(N), dim3(64), lb, sp.stream, h->d_params, h->ac, h->cap, h->cap_angular, h->cap_angular, h->d_recA, h->d_recB, h->d_tri,
                           h->d_cnt_a, h->d_cnt_ro, g, h->ld_angular, h->d_leg_force, h->d_centre_force, h->d_ids, nullptr, 0, h->hp.NB, (int)lb, nullptr, 0, N, 0);
    }
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

int dispatch_angular(nnpops_ani* h, bool forward, const float* g, float* out, const Span& sp) {
    KernelTimer timer(h, forward ? NNPOPS_ANI_K_ANGULAR_FWD : NNPOPS_ANI_K_ANGULAR_BWD, sp.stream);
    if (h->generic) return h->hp.torchani ? launch_generic<true>(h, forward, g, out, sp) : launch_generic<false>(h, forward, g, out, sp);
    return h->hp.torchani ? dispatch_factors<true>(h, forward, g, out, sp) : dispatch_factors<false>(h, forward, g, out, sp);
}

// The fused neighbour build + angular forward (ani_build_forward.h): the ANI-1x / ANI-2x factor shape, two waves per atom.
// (fused kernel: records + the triple list in LDS besides the staged factors; the builder's scratch shares the staging area)
int fused_chunk(const nnpops_ani* h) {
    return forward_chunk(h, (size_t)h->cap_angular * 2 * sizeof(float4) + (size_t)triples_capacity(h->cap_angular) * sizeof(int) + 64, 12 * sizeof(float));
}

bool build_forward_fused(const nnpops_ani* h, const float* angular) {
    const bool want = h->fuse_forward < 0 ? h->hp.N <= kFuseAtoms : h->fuse_forward != 0;
    return want && !h->generic && h->forward_kernel == 2 && h->fwd_waves_per_atom == 2 && h->nfrp == 8 && h->nfzp == 4 &&
           h->nstreams == 1 && h->fwd_identity && h->ld_angular % 4 == 0 && ((uintptr_t)angular & 15) == 0 &&
           build_forward_lds_bytes<8, 4>(h->cap, h->cap_angular, h->hp.S, h->hp.NB, fused_chunk(h)) <= 160 * 1024;
}

template <bool TA>
int launch_build_forward(nnpops_ani* h, const BuildInputs& in, const BuildOutputs& out, float* angular, const Span& sp) {
    int tri_offset = 0;
    const int CH = fused_chunk(h);
    const size_t lds = (build_forward_lds_bytes<8, 4>(h->cap, h->cap_angular, h->hp.S, h->hp.NB, CH, &tri_offset) + 15) & ~(size_t)15;
    const int vec_ok = 1 | (h->store_mode << 1) | (h->fwd_row_via_lds ? 8 : 0);
    const bool uni = h->fwd_uniform && h->fwd_grid && h->hp.nFR == 8 && h->hp.nFZ == 4;
    const bool dyn = h->fwd_dynamic && h->fwd_row_via_lds && h->hp.NB <= 63 && h->hp.NB * h->hp.nA <= CH * 12;
    auto k = h->fwd_occ == 6 ? ani_build_forward<TA, 8, 4, 6>
           : dyn ? (uni ? ani_build_forward<TA, 8, 4, 7, 1, true> : ani_build_forward<TA, 8, 4, 7, 0, true>)
           : uni ? ani_build_forward<TA, 8, 4, 7, 1> : ani_build_forward<TA, 8, 4, 7>;
    if (uni && h->fwd_literal && h->fwd_occ != 6) k = dyn ? ani_build_forward<TA, 8, 4, 7, 2, true> : ani_build_forward<TA, 8, 4, 7, 2>;
    if (lds > 64 * 1024) NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(sp.nw), dim3(128), lds, sp.stream, h->d_params, in, out, h->cap, h->cap_angular, CH, angular,
                       h->ld_angular, vec_ok, tri_offset, sp.w0, sp.nw);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

bool pair_backward_fits(const nnpops_ani* h) {
    const size_t m = (size_t)h->cap_angular * (h->cap_angular + 1) + (size_t)h->cap_angular * (h->cap_angular - 1) / 2;
    return (size_t)h->cap_angular * 32 + (size_t)h->hp.NB * h->nfrp * h->nfzp * 4 + m * 4 <= 160 * 1024;
}

// The spans of one evaluation: nstreams stretches of the (cell-ordered) atoms when the kernels in use take ranges, else one.
int make_spans(nnpops_ani* h, Span (&spans)[4]) {
    const int N = h->hp.N;
    const int* order = h->last_used_cells ? h->d_sorted_atom : nullptr;
    int k = h->can_split && N >= 2048 ? std::max(1, std::min(4, h->nstreams)) : 1;
    if (k > 1 && (!h->ev_fork || !h->side[k - 2])) {           // first use: streams cannot be created while the caller captures a graph
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) k = 1;
    }
    for (int q = 1; q < k; q++) {                              // side streams and fork / join events, created on first use
        if (!h->side[q - 1] && hipStreamCreateWithFlags(&h->side[q - 1], hipStreamNonBlocking) != hipSuccess) k = 1;
        if (k > 1 && !h->ev_join[q - 1] && hipEventCreateWithFlags(&h->ev_join[q - 1], hipEventDisableTiming) != hipSuccess) k = 1;
    }
    if (k > 1 && !h->ev_fork && hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) k = 1;
    const int per = ((N + k - 1) / k + 3) & ~3;                // (whole workgroups of up to four atoms)
    for (int q = 0; q < k; q++) {
        const int w0 = std::min(N, q * per), w1 = q == k - 1 ? N : std::min(N, (q + 1) * per);
        spans[q] = Span{q == 0 ? h->stream : h->side[q - 1], order, w0, w1 - w0,
                        h->work_order_valid && k == 1 && (!order || h->lpt == 2) ? h->d_work_order : order};
    }
    return k;
}

int fork_streams(nnpops_ani* h, const Span* spans, int k) {
    if (k <= 1) return NNPOPS_OK;
    NNPOPS_HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
    for (int q = 1; q < k; q++) NNPOPS_HIP_TRY(hipStreamWaitEvent(spans[q].stream, h->ev_fork, 0));
    return NNPOPS_OK;
}

int join_streams(nnpops_ani* h, const Span* spans, int k) {
    for (int q = 1; q < k; q++) {
        NNPOPS_HIP_TRY(hipEventRecord(h->ev_join[q - 1], spans[q].stream));
        NNPOPS_HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join[q - 1], 0));
    }
    return NNPOPS_OK;
}

}  // namespace

extern "C" {

int nnpops_ani_create(nnpops_ani_t* out, int num_atoms, int num_species, float radial_cutoff, float angular_cutoff,
                      int periodic, const int32_t* atom_species, int num_radial, const float* radial_eta_rs,
                      int num_angular, const float* angular_eta_rs_zeta_ths, int torchani, int device) {
    NNPOPS_REQUIRE(out != nullptr, "out handle pointer is NULL");
    *out = nullptr;
    NNPOPS_REQUIRE(num_atoms > 0 && num_atoms <= kIdMask, "num_atoms must be in [1, %d] (got %d)", kIdMask, num_atoms);
    NNPOPS_REQUIRE(num_species > 0 && num_species <= kMaxSpecies, "num_species must be in [1, %d] (got %d)", kMaxSpecies, num_species);
    NNPOPS_REQUIRE(radial_cutoff > 0 && angular_cutoff > 0, "cutoffs must be positive");
    NNPOPS_REQUIRE(angular_cutoff <= radial_cutoff,
                   "angular cutoff (%g) must not exceed the radial cutoff (%g): angular neighbours are drawn from the radial scan "
                   "(reference CpuANISymmetryFunctions.cpp:129-135)", angular_cutoff, radial_cutoff);
    NNPOPS_REQUIRE(num_radial > 0 && num_radial <= kMaxRadialFns, "num_radial must be in [1, %d]", kMaxRadialFns);
    NNPOPS_REQUIRE(num_angular > 0 && num_angular <= kMaxAngularFns, "num_angular must be in [1, %d]", kMaxAngularFns);
    NNPOPS_REQUIRE(atom_species && radial_eta_rs && angular_eta_rs_zeta_ths, "NULL parameter array");
    for (int i = 0; i < num_atoms; i++)
        NNPOPS_REQUIRE(atom_species[i] >= 0 && atom_species[i] < num_species, "atom_species[%d] = %d is outside [0, %d)", i,
                       atom_species[i], num_species);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(NNPOPS_ERR_NO_DEVICE, "no HIP device available");
    NNPOPS_REQUIRE(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);

    nnpops_ani* h = new nnpops_ani();
    AniParams& hp = h->hp;
    hp.N = num_atoms; hp.S = num_species; hp.nR = num_radial; hp.nA = num_angular;
    hp.NB = num_species * (num_species + 1) / 2;
    hp.periodic = periodic != 0; hp.torchani = torchani != 0;
    hp.rcr = radial_cutoff; hp.rca = angular_cutoff;
    hp.rcr2 = radial_cutoff * radial_cutoff; hp.rca2 = angular_cutoff * angular_cutoff;
    hp.radial_scale = torchani ? 0.25f : 1.0f;
    hp.inv_rcr = 1.0f / radial_cutoff; hp.inv_rca = 1.0f / angular_cutoff;
    hp.kp_shift = 0;
    while ((1 << hp.kp_shift) < num_radial) hp.kp_shift++;
    hp.angle_damp = torchani ? 0.95f : 1.0f;
    for (int k = 0; k < num_radial; k++) {
        hp.rad_eta[k] = radial_eta_rs[2 * k];
        hp.rad_rs[k] = radial_eta_rs[2 * k + 1];
        hp.rad_c[k] = -radial_eta_rs[2 * k] * kLog2e;
    }
    for (int a = 0, bk = 0; a < num_species; a++)
        for (int b = a; b < num_species; b++, bk++) { hp.bkt_a[bk] = a; hp.bkt_b[bk] = b; }
    for (int m = 0; m < num_angular; m++) {                    // the list as given, for the generic kernels
        const float* f = angular_eta_rs_zeta_ths + 4 * m;
        hp.af_eta[m] = f[0]; hp.af_rs[m] = f[1]; hp.af_zeta[m] = f[2];
        hp.af_c[m] = -f[0] * kLog2e;
        hp.af_cos[m] = (float)std::cos((double)f[3]);
        hp.af_sin[m] = (float)std::sin((double)f[3]);
    }
    int rc = factor_angular(hp, angular_eta_rs_zeta_ths, num_angular);
    h->generic = rc != NNPOPS_OK;
    if (h->generic) {                                          // (factor fields are not used; keep them harmless)
        hp.nFR = hp.nFZ = 1;
        for (int m = 0; m < num_angular; m++) { hp.c_of_m[m] = 0; hp.scale_m[m] = 1.f; }
    }
    if (const char* e = std::getenv("NNPOPS_ANI_GENERIC")) h->generic = h->generic || std::atoi(e) != 0;      // tests: force
    h->nfrp = pad_pow2(hp.nFR, 4);
    h->nfzp = pad_pow2(hp.nFZ, 4);
    h->fwd_uniform = !h->generic;
    for (int a = 1; a < hp.nFR; a++) h->fwd_uniform = h->fwd_uniform && hp.fr_c[a] == hp.fr_c[0];
    for (int z = 1; z < hp.nFZ; z++) h->fwd_uniform = h->fwd_uniform && hp.fz_zeta[z] == hp.fz_zeta[0];
    h->fwd_grid = true;
    if (h->fwd_uniform && hp.nFR == 8) {
        // Eight radial factors of one eta on equally spaced shifts (every ANI-2x-shaped model): the UNI forward kernel gets them
        // by recurrence (GeoRadial, ani_kernels.h).  A list that is not such a grid takes the kernel that evaluates factor by factor.
        const double rs0 = hp.fr_rs[0], d = ((double)hp.fr_rs[7] - rs0) / 7.0, c = hp.fr_c[0];
        for (int a = 1; a < 8; a++) h->fwd_grid = h->fwd_grid && std::fabs((double)hp.fr_rs[a] - (rs0 + a * d)) <= 1e-6 * std::fabs(d);
        // (the ratios must stay inside the fp32 range over 0 <= x <= Rca: exponents k0 +- k1 (x - Rs_1), then +- 8 c d^2)
        const double rs1 = rs0 + d;
        const double reach = 2.0 * std::fabs(c * d) * std::max(std::fabs(rs1), std::fabs((double)hp.rca - rs1)) + 9.0 * std::fabs(c) * d * d;
        h->fwd_grid = h->fwd_grid && d != 0.0 && reach < 120.0;
        GeoRadial& g = hp.geo;
        g.rs1 = (float)(rs0 + d); g.c = (float)c; g.k1 = (float)(-2.0 * c * d); g.k0 = (float)(c * d * d);
        g.q = (float)std::exp2(2.0 * c * d * d); g.q4 = (float)std::exp2(8.0 * c * d * d); g.qi4 = (float)std::exp2(-8.0 * c * d * d);
        g.d4 = (float)(4.0 * d);
    }
    if (const char* e = std::getenv("NNPOPS_ANI_FWD_UNI")) h->fwd_uniform = h->fwd_uniform && std::atoi(e) != 0;
    {   // the published ANI-2x constants? (bit-for-bit: a model that differs in the last place keeps its constants in registers)
        using L = Ani2xAngular;
        const GeoRadial& g = hp.geo;
        const float lz[4] = {L::zc0, L::zc1, L::zc2, L::zc3}, ls[4] = {L::zs0, L::zs1, L::zs2, L::zs3};
        bool same = h->fwd_uniform && h->fwd_grid && hp.nFR == 8 && hp.nFZ == 4 && hp.fz_zeta[0] == L::zeta && hp.fz_bias[0] == L::zbias &&
                    g.rs1 == L::rs1 && g.c == L::c && g.k1 == L::k1 && g.k0 == L::k0 && g.q == L::q && g.q4 == L::q4 && g.qi4 == L::qi4 &&
                    g.d4 == L::d4;
        for (int z = 0; z < 4 && same; z++) same = hp.fz_cos[z] == lz[z] && hp.fz_sin[z] == ls[z];
        h->fwd_literal = same;
        if (const char* e = std::getenv("NNPOPS_ANI_FWD_LITERAL")) h->fwd_literal = h->fwd_literal && std::atoi(e) != 0;
        if (const char* e = std::getenv("NNPOPS_ANI_BWD_LITERAL")) h->bwd_literal = std::atoi(e) != 0;
    }
    // Matrix-core forward kernel: quads are handed the species pairs that can occur among this system's atoms.
    {
        std::vector<char> present(num_species, 0);
        for (int i = 0; i < num_atoms; i++) present[atom_species[i]] = 1;
        std::vector<int> live;
        hp.fwd_nabsent = 0;
        for (int bk = 0; bk < hp.NB; bk++) {
            if (present[hp.bkt_a[bk]] && present[hp.bkt_b[bk]]) live.push_back(bk);
            else hp.fwd_absent[hp.fwd_nabsent++] = bk;
        }
        for (int q = 0; q < kFwdSlots; q++) hp.fwd_slot_bucket[q] = -1;
        hp.fwd_split = 1;
        h->mfma_ok = !live.empty() && (int)live.size() <= kFwdSlots;
        if (h->mfma_ok) {
            while (hp.fwd_split < 8 && (int)live.size() * hp.fwd_split * 2 <= kFwdSlots) hp.fwd_split *= 2;
            for (size_t b = 0; b < live.size(); b++)
                for (int p = 0; p < hp.fwd_split; p++) hp.fwd_slot_bucket[b * hp.fwd_split + p] = live[b];
        }
        for (int c = 0; c < kMaxAngularFns; c++) hp.m_of_c[c] = -1;
        h->fwd_identity = num_angular == h->nfrp * h->nfzp;
        for (int m = 0; m < num_angular; m++) {
            hp.m_of_c[hp.c_of_m[m]] = m;
            if (hp.c_of_m[m] != m) h->fwd_identity = false;
        }
        hp.fwd_zero_shift = 0;
        while ((4 << hp.fwd_zero_shift) < num_angular && hp.fwd_zero_shift < 6) hp.fwd_zero_shift++;
        h->fwd_identity = h->fwd_identity && num_angular <= 256;
        h->forward_kernel = h->mfma_ok ? 2 : -1;
        if (const char* e = std::getenv("NNPOPS_ANI_FWD_CHUNK")) { h->fwd_chunk = std::min(512, std::max(64, (std::atoi(e) + 15) / 16 * 16)); h->fwd_chunk_forced = true; }
        if (const char* e = std::getenv("NNPOPS_ANI_FUSE")) h->fuse_forward = std::atoi(e) != 0 ? 1 : 0;
        if (const char* e = std::getenv("NNPOPS_ANI_LPT")) h->lpt = std::atoi(e);
        if (const char* e = std::getenv("NNPOPS_ANI_CELL_ATOMS")) h->cell_atoms = std::max(1, std::atoi(e));
        if (const char* e = std::getenv("NNPOPS_ANI_RBWD")) h->rbwd_lanes = std::atoi(e) != 0;
        if (const char* e = std::getenv("NNPOPS_ANI_SCATTER")) h->scatter_mode = std::atoi(e) != 0 ? 1 : 0;
        if (const char* e = std::getenv("NNPOPS_ANI_FINE_GRID")) h->fine_grid = std::atoi(e) != 0;
        if (const char* e = std::getenv("NNPOPS_ANI_FWD_ROWLDS")) h->fwd_row_via_lds = std::atoi(e) != 0;
        if (const char* e = std::getenv("NNPOPS_ANI_FWD_OCC")) h->fwd_occ = std::atoi(e);
        {   // Does the per-atom quad table pay?  With one quad set per species pair the step loop of phase 2 runs max_b n_b / K times,
            // balanced it runs ~T / 32 times; the table costs ~150-250 instructions per atom (two waves).  From the composition:
            // the largest bucket's expected share of the triples, f = p_a p_b (x 2 for a != b), against 1 / 32 per quad and the K
            // quads it already has.  Seven equally likely species 1.3, water 1.8 (measured there: 17.6 -> 21.1 us and 16.1 -> 22.9 us
            // with the table: it loses); H/C/N/O molecules 4.8 (64 steps against 14 for a 400-triple atom: it wins).
            std::vector<double> frac(num_species, 0.0);
            for (int i = 0; i < num_atoms; i++) frac[atom_species[i]] += 1.0 / num_atoms;
            double fmax = 0;
            for (int a = 0; a < num_species; a++)
                for (int b = a; b < num_species; b++) fmax = std::max(fmax, frac[a] * frac[b] * (a == b ? 1.0 : 2.0));
            h->fwd_dynamic = h->mfma_ok && 32.0 * fmax / hp.fwd_split > 3.0;
        }
        if (const char* e = std::getenv("NNPOPS_ANI_FWD_DYN")) h->fwd_dynamic = std::atoi(e) != 0;
        if (const char* e = std::getenv("NNPOPS_ANI_STREAMS")) h->nstreams = std::max(1, std::min(4, std::atoi(e)));
        if (const char* e = std::getenv("NNPOPS_ANI_BWD_APG")) h->bwd_atoms_per_group = std::atoi(e);
        if (const char* e = std::getenv("NNPOPS_ANI_STORE")) h->store_mode = std::atoi(e) & 3;
        if (const char* e = std::getenv("NNPOPS_ANI_OCC")) h->occ6 = std::atoi(e) >= 6;
        if (const char* e = std::getenv("NNPOPS_ANI_BACKWARD")) { h->backward_kernel = std::min(4, std::max(0, std::atoi(e))); h->backward_forced = true; }
        if (const char* e = std::getenv("NNPOPS_ANI_FWD_WPA")) h->fwd_waves_per_atom = std::atoi(e) == 1 ? 1 : 2;
        if (const char* e = std::getenv("NNPOPS_ANI_FWD_APG")) h->fwd_atoms_per_group = std::max(1, std::atoi(e));
    }
    h->device = device;
    h->cap = 128;
    h->cap_angular = 32;
    if (const char* e = std::getenv("NNPOPS_ANI_CAP")) h->cap = std::max(16, std::atoi(e) & ~15);      // (initial row capacity; grows on demand)

    DeviceGuard guard(device);
    if (!guard.ok) { delete h; return fail(NNPOPS_ERR_HIP, "cannot select device %d", device); }
    auto cleanup = [&](int code) { nnpops_ani_destroy(h); return code; };
    if ((rc = dev_alloc(&h->d_params, 1))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_species, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_cnt_a, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_cnt_ro, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_cnt_pos, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_centre_force, (size_t)num_atoms + 1))) return cleanup(rc);      // (+ the class launches' flag word, ani_angular_bwd.h)
    if (hipMemset(h->d_centre_force + num_atoms, 0, sizeof(float4)) != hipSuccess) return cleanup(fail(NNPOPS_ERR_HIP, "memset failed"));
    if ((rc = dev_alloc(&h->d_status, (size_t)kStatAlloc))) return cleanup(rc);
    if ((rc = alloc_rows(h))) return cleanup(rc);
    h->max_cells = num_atoms + 4096;
    if ((rc = dev_alloc(&h->d_grid, 1))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_cell_count, (size_t)h->max_cells))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_cell_start, (size_t)h->max_cells + 1))) return cleanup(rc);
    if (const char* e = std::getenv("NNPOPS_CELL_BIN_CAP")) h->bin_cap = std::max(4, std::atoi(e) & ~3);   // tests: force growth
    if (periodic && num_atoms <= kBinnedAtoms) {
        if ((rc = dev_alloc(&h->d_hist, (size_t)kHistWords))) return cleanup(rc);
        if ((rc = dev_alloc(&h->d_bins, (size_t)kBinnedCells * h->bin_cap))) return cleanup(rc);
        if (hipMemset(h->d_hist, 0, sizeof(int) * kHistWords) != hipSuccess) return cleanup(fail(NNPOPS_ERR_HIP, "memset failed"));
    }
    if ((rc = dev_alloc(&h->d_atom_cell, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_atom_rank, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_sorted_cell, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_tile_total, (size_t)h->max_cells / kScanTile + 2))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_sorted_atom, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_work_order, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_class_tile, (size_t)num_atoms))) return cleanup(rc);
    if (hipMemset(h->d_class_tile, 255, (size_t)num_atoms) != hipSuccess) return cleanup(fail(NNPOPS_ERR_HIP, "memset failed"));
    hp.class_tile = h->d_class_tile;
    if (const char* e = std::getenv("NNPOPS_ANI_BWD_CLASSES")) h->bwd_by_class = std::atoi(e) != 0;
    if (const char* e = std::getenv("NNPOPS_ANI_BWD_CLASS_MIN")) h->bwd_class_min = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("NNPOPS_ANI_BWD_CLASS_ATOMS")) h->bwd_class_atoms = std::max(0, std::atoi(e));
    if ((rc = dev_alloc(&h->d_unsorted_atom, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_sorted_pos, (size_t)num_atoms))) return cleanup(rc);
    if ((rc = dev_alloc(&h->d_bucket_offsets, (size_t)num_atoms * (hp.NB + 1)))) return cleanup(rc);
    hp.bucket_offsets = h->d_bucket_offsets;
    if (hipMemset(h->d_bucket_offsets, 0, sizeof(int) * (size_t)num_atoms * (hp.NB + 1)) != hipSuccess)
        return cleanup(fail(NNPOPS_ERR_HIP, "memset failed"));
    // Two angular forward kernels, chosen per system from its composition.  The chunked view pads every species-pair
    // bucket to whole chunks of 8 triples: ideal for water or organic molecules (a few well-filled buckets; measured
    // 8-10 % faster than the run-merging kernel), wasteful when many species are equally likely (7 uniform species:
    // 6 triples per bucket, 8 % slower).  Estimate the useful fraction of the padded view for ~18 angular neighbours.
    {
        std::vector<double> frac(num_species, 0.0);
        for (int i = 0; i < num_atoms; i++) frac[atom_species[i]] += 1.0 / num_atoms;
        const double triples = 18.0 * 17.0 / 2.0;
        double useful = 0, padded = 0;
        for (int a = 0; a < num_species; a++)
            for (int b = a; b < num_species; b++) {
                const double t = triples * frac[a] * frac[b] * (a == b ? 1.0 : 2.0);     // expected triples of the bucket
                useful += t;
                padded += 8.0 * std::max(t / 8.0 + 0.5, 1.0) * (1.0 - std::exp(-t));       // ~E[8 * ceil(X / 8)], X ~ Poisson(t)
            }
        // measured: water 0.93 and H/C/N/O 0.79 favour the chunked view, seven equally likely species 0.57 do not
        h->chunked_forward = hp.NB < 64 && useful >= 0.70 * padded;     // (the chunked view scans the buckets with one wave)
    }
    if (h->forward_kernel < 0) h->forward_kernel = h->chunked_forward ? 1 : 0;
    if (const char* e = std::getenv("NNPOPS_ANI_FORWARD")) {          // tests / A-B: 0 run merging, 1 chunked view, 2 matrix cores
        const int want = std::atoi(e);
        if (want == 2 && h->mfma_ok) h->forward_kernel = 2;
        else if (want != 2) { h->chunked_forward = hp.NB < 64 && want != 0; h->forward_kernel = h->chunked_forward ? 1 : 0; }
    }
    {   // the angular backward kernel's constants, by value: every factor slot behind the real ones holds its neutral value
        AngularConsts& c = h->ac;
        c.N = hp.N; c.nA = hp.nA;
        for (int a = 0; a < kMaxFactor; a++) {
            const bool live = !h->generic && a < hp.nFR;
            c.fr_c[a] = live ? hp.fr_c[a] : 0.f; c.fr_rs[a] = live ? hp.fr_rs[a] : 0.f; c.fr_negeta[a] = live ? -hp.fr_eta[a] : 0.f;
        }
        for (int z = 0; z < 8; z++) {
            const bool live = !h->generic && z < hp.nFZ;
            c.fz_zeta[z] = live ? hp.fz_zeta[z] : 1.f; c.fz_cos[z] = live ? hp.fz_cos[z] : 0.f;
            c.fz_sin[z] = live ? hp.fz_sin[z] : 0.f; c.fz_bias[z] = live ? hp.fz_bias[z] : 0.f;
        }
    }
    if (hipMemcpy(h->d_params, &hp, sizeof(AniParams), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->d_species, atom_species, sizeof(int32_t) * num_atoms, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(h->d_status, 0, sizeof(int) * kStatAlloc) != hipSuccess ||
        hipMemset(h->d_cnt_a, 0, sizeof(int) * num_atoms) != hipSuccess ||
        hipMemset(h->d_cnt_ro, 0, sizeof(int) * num_atoms) != hipSuccess || hipMemset(h->d_cnt_pos, 0, sizeof(int) * num_atoms) != hipSuccess)
        return cleanup(fail(NNPOPS_ERR_HIP, "parameter upload failed: %s", hipGetErrorString(hipGetLastError())));
    *out = h;
    return NNPOPS_OK;
}

int nnpops_ani_destroy(nnpops_ani_t h) {
    if (!h) return NNPOPS_OK;
    DeviceGuard guard(h->device);
    dev_free(h->d_params); dev_free(h->d_species); dev_free(h->d_segment);
    dev_free(h->d_nbr); dev_free(h->d_recA); dev_free(h->d_recB); dev_free(h->d_tri); dev_free(h->d_cnt_a); dev_free(h->d_cnt_ro); dev_free(h->d_cnt_pos); dev_free(h->d_status);
    dev_free(h->d_ids); dev_free(h->d_leg_force); dev_free(h->d_centre_force); dev_free(h->d_bucket_offsets);
    dev_free(h->d_hist); dev_free(h->d_bins);
    dev_free(h->d_grid); dev_free(h->d_cell_count); dev_free(h->d_cell_start); dev_free(h->d_atom_cell);
    dev_free(h->d_atom_rank); dev_free(h->d_sorted_cell); dev_free(h->d_tile_total); dev_free(h->d_sorted_atom); dev_free(h->d_work_order); dev_free(h->d_class_tile); dev_free(h->d_unsorted_atom); dev_free(h->d_sorted_pos);
    for (int q = 0; q < 3; q++) {
        if (h->side[q]) (void)hipStreamDestroy(h->side[q]);
        if (h->ev_join[q]) (void)hipEventDestroy(h->ev_join[q]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->h_status) (void)hipHostFree(h->h_status);
    for (int k = 0; k < NNPOPS_ANI_NUM_KERNELS; k++) {
        for (hipEvent_t e : h->ev_start[k]) (void)hipEventDestroy(e);
        for (hipEvent_t e : h->ev_stop[k]) (void)hipEventDestroy(e);
    }
    delete h;
    return NNPOPS_OK;
}

int nnpops_ani_set_stream(nnpops_ani_t h, void* stream) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    h->stream = (hipStream_t)stream;
    return NNPOPS_OK;
}

int nnpops_ani_set_molecules(nnpops_ani_t h, int num_molecules, const int32_t* molecule_offsets) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    DeviceGuard guard(h->device);
    if (num_molecules <= 0 || molecule_offsets == nullptr) {     // back to one system
        dev_free(h->d_segment);
        h->computed = false;
        return NNPOPS_OK;
    }
    NNPOPS_REQUIRE(!h->hp.periodic, "batched molecules are non-periodic systems");
    NNPOPS_REQUIRE(molecule_offsets[0] == 0 && molecule_offsets[num_molecules] == h->hp.N,
                   "molecule_offsets must start at 0 and end at num_atoms (%d)", h->hp.N);
    std::vector<int2> seg(h->hp.N);
    for (int m = 0; m < num_molecules; m++) {
        NNPOPS_REQUIRE(molecule_offsets[m] < molecule_offsets[m + 1], "molecule %d is empty or offsets are not increasing", m);
        for (int i = molecule_offsets[m]; i < molecule_offsets[m + 1]; i++) seg[i] = int2{molecule_offsets[m], molecule_offsets[m + 1]};
    }
    int rc;
    if (!h->d_segment && (rc = dev_alloc(&h->d_segment, (size_t)h->hp.N))) return rc;
    NNPOPS_HIP_TRY(hipMemcpy(h->d_segment, seg.data(), sizeof(int2) * seg.size(), hipMemcpyHostToDevice));
    h->computed = false;
    return NNPOPS_OK;
}

int nnpops_ani_set_neighbor_algorithm(nnpops_ani_t h, int algorithm) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    NNPOPS_REQUIRE(algorithm >= 0 && algorithm <= 2, "algorithm must be 0 (auto), 1 (all pairs) or 2 (cell list)");
    h->algorithm = algorithm;
    return NNPOPS_OK;
}

int nnpops_ani_compute(nnpops_ani_t h, const float* positions, const float* box, float* radial, float* angular) {
    return nnpops_ani_compute_strided(h, positions, box, radial, 0, angular, 0);
}

int nnpops_ani_compute_strided(nnpops_ani_t h, const float* positions, const float* box, float* radial, int radial_ld,
                               float* angular, int angular_ld) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    {
        const int wr = h->hp.S * h->hp.nR, wa = h->hp.NB * h->hp.nA;
        NNPOPS_REQUIRE((radial_ld == 0 || radial_ld >= wr) && (angular_ld == 0 || angular_ld >= wa),
                       "row strides must be 0 (dense) or at least the row widths (%d, %d)", wr, wa);
        h->ld_radial = radial_ld ? radial_ld : wr;
        h->ld_angular = angular_ld ? angular_ld : wa;
    }
    NNPOPS_REQUIRE(positions && radial && angular, "NULL device pointer");
    NNPOPS_REQUIRE(!h->hp.periodic || box, "periodic handle needs box vectors");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(NNPOPS_ERR_HIP, "cannot select device %d", h->device);
    const int N = h->hp.N;
    const bool per = h->hp.periodic;
    // What backprop needs from this call (ANISymmetryFunctions.h:83-84) is kept in the neighbour rows and
    // records the builder writes (displacements, not positions), so positions and box are read in place:
    // no copies, no memsets on the hot path.

    // neighbour search + radial AEV: cell grid for large systems, the reference's all-pairs scan for small ones
    // (or when a previous compute found the box too small for the 27-cell stencil)
    const int lds_bw = (int)((builder_lds_bytes(h->cap, h->hp.S, h->hp.NB) + 15) & ~(size_t)15);
    const int wpg_b = waves_per_group(lds_bw);
    const size_t lds_b = (size_t)lds_bw * wpg_b;
    const dim3 ablock(64 * wpg_b);
    // (below ~1 800 atoms scanning every atom -- 28 batches of 64 -- costs less than the two launches of the grid build:
    //  1 500 atoms 38.8 -> 34.4 us per fwd+bwd step, 2 001 atoms equal, 3 000 atoms 45 vs 52 us)
    const bool use_cells = !h->d_segment && (h->algorithm == 2 || (h->algorithm == 0 && N >= h->cell_atoms && !h->cells_disabled));
    h->last_used_cells = use_cells;
    if (use_cells) {
        KernelTimer timer(h, NNPOPS_ANI_K_CELL_GRID);
        const CellBuffers cb{h->d_grid, h->d_cell_count, h->d_cell_start, h->d_atom_cell, h->d_atom_rank,
                             h->d_unsorted_atom, h->d_sorted_atom, h->d_sorted_pos, h->max_cells,
                             h->d_hist, h->d_bins, h->bin_cap, h->fine_grid ? 1 : 0, h->d_sorted_cell, h->d_tile_total};
        launch_cell_build(h->stream, N, positions, box, per, h->hp.rcr, h->d_species, cb);
    }
    // The per-atom kernels, span by span: every span's chain (neighbour build -> angular forward) runs on its own stream.
    h->can_split = !h->generic && h->forward_kernel == 2 && h->backward_kernel >= 1 &&
                   pair_backward_fits(h);
    Span spans[4];
    const int nspans = make_spans(h, spans);
    int rc = fork_streams(h, spans, nspans);
    if (rc != NNPOPS_OK) return rc;
    for (int q = 0; q < nspans; q++) {
        const Span& sp = spans[q];
        const dim3 sgrid(div_up(sp.nw, wpg_b));
        h->last_fused_build = build_forward_fused(h, angular);
        if (h->last_fused_build) {                             // one launch: build + radial + angular forward (timed as the build)
            KernelTimer timer(h, NNPOPS_ANI_K_NEIGHBORS, sp.stream);
            KernelTimer merged_timer(h, NNPOPS_ANI_K_NEIGHBORS, sp.stream, /*merged=*/true);      // (merge mode brackets the same launch: ADVICE r04)
            const BuildInputs in{box, h->d_grid, h->d_cell_start, h->d_sorted_cell, h->d_sorted_pos, h->d_hist, positions, h->d_species,
                                 h->d_segment, use_cells ? 1 : 0, per ? 1 : 0};
            const BuildOutputs out{h->d_nbr, h->d_recA, h->d_recB, h->d_ids, h->d_tri, h->d_cnt_a, h->d_cnt_ro, h->d_cnt_pos, h->d_status, radial, h->ld_radial};
            rc = h->hp.torchani ? launch_build_forward<true>(h, in, out, angular, sp) : launch_build_forward<false>(h, in, out, angular, sp);
            if (rc != NNPOPS_OK) return rc;
            continue;
        }
        KernelTimer merged_timer(h, NNPOPS_ANI_K_NEIGHBORS, sp.stream, /*merged=*/true);      // (spans the angular forward below)
        {
        KernelTimer timer(h, NNPOPS_ANI_K_NEIGHBORS, sp.stream);
        if (use_cells) {
            if (per)
                hipLaunchKernelGGL(ani_neighbors_cells<true>, sgrid, ablock, lds_b, sp.stream, h->d_params, box, h->d_grid,
                                   h->d_cell_start, h->d_sorted_cell, h->d_sorted_pos, h->d_nbr, h->cap, h->cap_angular, h->d_recA,
                                   h->d_recB, h->d_ids, h->d_tri, h->d_cnt_a, h->d_cnt_ro, h->d_cnt_pos, h->d_status, radial, h->ld_radial, lds_bw, h->d_hist,
                                   sp.w0, sp.nw);
            else
                hipLaunchKernelGGL(ani_neighbors_cells<false>, sgrid, ablock, lds_b, sp.stream, h->d_params, box, h->d_grid,
                                   h->d_cell_start, h->d_sorted_cell, h->d_sorted_pos, h->d_nbr, h->cap, h->cap_angular, h->d_recA,
                                   h->d_recB, h->d_ids, h->d_tri, h->d_cnt_a, h->d_cnt_ro, h->d_cnt_pos, h->d_status, radial, h->ld_radial, lds_bw, h->d_hist,
                                   sp.w0, sp.nw);
        } else if (per)
            hipLaunchKernelGGL(ani_neighbors_allpairs<true>, sgrid, ablock, lds_b, sp.stream, h->d_params, positions, box,
                               h->d_species, h->d_segment, h->d_nbr, h->cap, h->cap_angular, h->d_recA, h->d_recB, h->d_ids, h->d_tri,
                               h->d_cnt_a, h->d_cnt_ro, h->d_cnt_pos, h->d_status, radial, h->ld_radial, lds_bw, sp.w0, sp.nw);
        else
            hipLaunchKernelGGL(ani_neighbors_allpairs<false>, sgrid, ablock, lds_b, sp.stream, h->d_params, positions, box,
                               h->d_species, h->d_segment, h->d_nbr, h->cap, h->cap_angular, h->d_recA, h->d_recB, h->d_ids, h->d_tri,
                               h->d_cnt_a, h->d_cnt_ro, h->d_cnt_pos, h->d_status, radial, h->ld_radial, lds_bw, sp.w0, sp.nw);
        }
        NNPOPS_HIP_TRY(hipGetLastError());
        // (the radial AEV is written by the builder wave itself: radial_forward_from_lds)
        rc = dispatch_angular(h, true, nullptr, angular, sp);
        if (rc != NNPOPS_OK) return rc;
    }
    rc = join_streams(h, spans, nspans);
    if (rc != NNPOPS_OK) return rc;
    h->computed = true;
    return NNPOPS_OK;
}

int nnpops_ani_backprop(nnpops_ani_t h, const float* radial_deriv, const float* angular_deriv, float* position_deriv) {
    return nnpops_ani_backprop_strided(h, radial_deriv, 0, angular_deriv, 0, position_deriv);
}

int nnpops_ani_backprop_strided(nnpops_ani_t h, const float* radial_deriv, int radial_ld, const float* angular_deriv,
                                int angular_ld, float* position_deriv) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    {
        const int wr = h->hp.S * h->hp.nR, wa = h->hp.NB * h->hp.nA;
        NNPOPS_REQUIRE((radial_ld == 0 || radial_ld >= wr) && (angular_ld == 0 || angular_ld >= wa),
                       "row strides must be 0 (dense) or at least the row widths (%d, %d)", wr, wa);
        h->ld_radial = radial_ld ? radial_ld : wr;
        h->ld_angular = angular_ld ? angular_ld : wa;
    }
    NNPOPS_REQUIRE(radial_deriv && angular_deriv && position_deriv, "NULL device pointer");
    NNPOPS_REQUIRE(h->computed, "backprop() must follow compute() (ANISymmetryFunctions.h:83-84)");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(NNPOPS_ERR_HIP, "cannot select device %d", h->device);
    const int N = h->hp.N;
    const int lds_rw = (int)((((size_t)h->hp.S * h->hp.nR + 8 * (size_t)h->cap) * sizeof(float) + 15) & ~(size_t)15);
    const int wpg_r = waves_per_group(lds_rw);
    const size_t lds_r = (size_t)lds_rw * wpg_r;
    if (lds_r > 64 * 1024) return fail(NNPOPS_ERR_UNSUPPORTED, "radial backward needs %zu bytes of LDS", lds_r);
    const dim3 ablock(64 * wpg_r);
    // 1. angular backward parks the per-leg forces in leg_force / centre_force (no scatter), span by span on the streams ...
    Span spans[4];
    const int nspans = make_spans(h, spans);
    int rc = fork_streams(h, spans, nspans);
    if (rc != NNPOPS_OK) return rc;
    h->backprop_stamp = h->backprop_stamp >= 0x3fffffff ? 1 : h->backprop_stamp + 1;      // (as a float's bits: never a NaN pattern, never 0)
    {   // Leg forces straight into the receiving atoms' rows?  Only where EVERY launch of the angular backward runs two waves per atom
        // (the second wave does the look-up under the first one's row sums) and the radial backward is the kernel that reads such rows:
        // dense systems (the atoms average 200 triples or more, check()), the pair-matrix kernels with 16-byte gradient loads.
        const bool vec_ok = h->fwd_identity && h->ld_angular % 4 == 0 && (reinterpret_cast<uintptr_t>(angular_deriv) & 15) == 0;
        const bool can = !h->generic && h->backward_kernel == 1 && !h->backward_forced && vec_ok && pair_backward_fits(h) &&
                         radial_lanes(h, radial_deriv) && h->hp.nR == 16 && h->nstreams == 1 &&
                         h->nfrp == 8 && h->nfzp == 4 && h->fwd_literal && h->bwd_literal && h->fwd_uniform && h->hp.nFR == 8 && h->hp.nFZ == 4 && !h->occ6;
        h->scatter_now = can && (h->scatter_mode < 0 ? h->bwd_two_waves : h->scatter_mode != 0);
    }
    KernelTimer merged_timer(h, NNPOPS_ANI_K_ANGULAR_BWD, spans[0].stream, /*merged=*/true);      // (spans the radial backward below)
    for (int q = 0; q < nspans; q++) {
        rc = dispatch_angular(h, false, angular_deriv, nullptr, spans[q]);
        if (rc != NNPOPS_OK) return rc;
    }
    rc = join_streams(h, spans, nspans);                       // (every leg force must be there before anybody gathers)
    if (rc != NNPOPS_OK) return rc;
    // 2. ... and the radial backward wave of every atom, the only writer of position_deriv[i], gathers them
    rc = fork_streams(h, spans, nspans);
    if (rc != NNPOPS_OK) return rc;
    for (int q = 0; q < nspans; q++) {
        const Span& sp = spans[q];
        KernelTimer timer(h, NNPOPS_ANI_K_RADIAL_BWD, sp.stream);
        const int nr4 = h->hp.nR / 4;
        const bool lanes = radial_lanes(h, radial_deriv);
        if (lanes) {
            // a lane per neighbour: only this atom's own gradient row is staged in LDS
            const int lw = (int)(((size_t)h->hp.S * h->hp.nR * sizeof(float) + 15) & ~(size_t)15);
            const int wpg = kWavesPerGroup;
            const bool wide = h->cap_angular == 64;
            auto k = ani_radial_backward_lanes<4, 32>;
            if (h->scatter_now) {                              // (reads the legs from the atom's own row; sixteen radial functions: backprop() asks for nothing else)
                k = sp.nw <= kRbwdLatencyAtoms ? (wide ? ani_radial_backward_lanes<4, 64, true, true> : ani_radial_backward_lanes<4, 32, true, true>)
                                                              : (wide ? ani_radial_backward_lanes<4, 64, false, true> : ani_radial_backward_lanes<4, 32, false, true>);
            } else if (nr4 == 4 && sp.nw <= kRbwdLatencyAtoms) k = wide ? ani_radial_backward_lanes<4, 64, true> : ani_radial_backward_lanes<4, 32, true>;
            else
            switch (nr4) {
                case 1: k = wide ? ani_radial_backward_lanes<1, 64> : ani_radial_backward_lanes<1, 32>; break;
                case 2: k = wide ? ani_radial_backward_lanes<2, 64> : ani_radial_backward_lanes<2, 32>; break;
                case 3: k = wide ? ani_radial_backward_lanes<3, 64> : ani_radial_backward_lanes<3, 32>; break;
                case 4: k = wide ? ani_radial_backward_lanes<4, 64> : ani_radial_backward_lanes<4, 32>; break;
                case 5: k = wide ? ani_radial_backward_lanes<5, 64> : ani_radial_backward_lanes<5, 32>; break;
                case 6: k = wide ? ani_radial_backward_lanes<6, 64> : ani_radial_backward_lanes<6, 32>; break;
                case 7: k = wide ? ani_radial_backward_lanes<7, 64> : ani_radial_backward_lanes<7, 32>; break;
                default: k = wide ? ani_radial_backward_lanes<8, 64> : ani_radial_backward_lanes<8, 32>; break;
            }
            hipLaunchKernelGGL(k, dim3(div_up(sp.nw, wpg)), dim3(64 * wpg), (size_t)lw * wpg, sp.stream, h->d_params, h->d_species, h->d_nbr,
                               h->cap, h->d_cnt_pos, radial_deriv, h->ld_radial, h->d_ids, h->d_leg_force, h->d_centre_force,
                               sp.order, position_deriv, lw, sp.w0, sp.nw);
        } else
        hipLaunchKernelGGL(ani_radial_backward, dim3(div_up(sp.nw, wpg_r)), ablock, lds_r, sp.stream, h->d_params, h->d_species, h->d_nbr, h->cap,
                           h->cap_angular, h->d_cnt_pos, radial_deriv, h->ld_radial, h->d_ids, h->d_leg_force, h->d_centre_force,
                           sp.order, position_deriv, lds_rw, sp.w0, sp.nw);
    }
    rc = join_streams(h, spans, nspans);
    if (rc != NNPOPS_OK) return rc;
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

// The capacity check in two halves, so that its host round trip does not stop the device: _begin queues ONE tiny launch
// behind the neighbour build that publishes the builders' overflow word, and a stamp after it, into pinned host memory the
// device can write (system-scope stores: no copy command, no event -- those cost the host ~35 us per step, more than the
// launch of every other kernel of the step together); the caller goes on launching the work that consumes the rows
// (consumers clamp their counts, an overflowed row is incomplete but harmless); _end polls the stamp -- long there by then
// -- and, when the word is clean, that was all.  Otherwise it is nnpops_ani_check().
}  // extern "C"
namespace {
__global__ void ani_publish_status(const int* __restrict__ status, int* __restrict__ host_words, int stamp) {
    const int overflow = status[kStatOverflow];
    __hip_atomic_store(&host_words[1], overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host_words[0], stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
extern "C" {

// -> 1: the handle can defer its check and has a fresh stamp (host words allocated); 0: not deferrable
static int check_begin_prepare(nnpops_ani_t h) {
    h->check_pending = false;
    if (!h->cap_fitted || h->backward_kernel == 0) return 0;          // the full check has decisions to make: not deferrable
    DeviceGuard guard(h->device);
    if (!guard.ok) return 0;
    if (!h->h_status) {
        if (hipHostMalloc((void**)&h->h_status, 2 * sizeof(int), hipHostMallocMapped) != hipSuccess) { h->h_status = nullptr; return 0; }
        h->h_status[0] = 0; h->h_status[1] = 0;
        if (hipHostGetDevicePointer((void**)&h->h_status_dev, h->h_status, 0) != hipSuccess) {
            (void)hipHostFree(h->h_status);
            h->h_status = nullptr;
            return 0;
        }
    }
    h->check_stamp = h->check_stamp == 0x7fffffff ? 1 : h->check_stamp + 1;
    return 1;
}

int nnpops_ani_check_begin(nnpops_ani_t h) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    if (!check_begin_prepare(h)) return 0;
    DeviceGuard guard(h->device);
    hipLaunchKernelGGL(ani_publish_status, dim3(1), dim3(1), 0, h->stream, h->d_status, h->h_status_dev, h->check_stamp);
    if (hipGetLastError() != hipSuccess) return 0;
    h->check_pending = true;
    return 1;
}

int nnpops_ani_check_begin_with(nnpops_ani_t h, const int32_t** word, int32_t** publish_to, int32_t* stamp) {
    NNPOPS_REQUIRE(h != nullptr && word != nullptr && publish_to != nullptr && stamp != nullptr, "NULL argument");
    if (!check_begin_prepare(h)) return 0;
    *word = reinterpret_cast<const int32_t*>(h->d_status + kStatOverflow);
    *publish_to = reinterpret_cast<int32_t*>(h->h_status_dev);
    *stamp = h->check_stamp;
    h->check_pending = true;
    return 1;
}

int nnpops_ani_check_end(nnpops_ani_t h) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    if (!h->check_pending) return nnpops_ani_check(h, nullptr, nullptr);
    h->check_pending = false;
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(NNPOPS_ERR_HIP, "cannot select device %d", h->device);
    volatile int* words = h->h_status;
    bool seen = false;
    for (long spin = 0; spin < 2000000 && !(seen = __atomic_load_n(&h->h_status[0], __ATOMIC_ACQUIRE) == h->check_stamp); spin++)
        __builtin_ia32_pause();
    if (!seen) {                                                       // (a very slow device, or the store got lost: ask the stream)
        NNPOPS_HIP_TRY(hipStreamSynchronize(h->stream));
        if (__atomic_load_n(&h->h_status[0], __ATOMIC_ACQUIRE) != h->check_stamp) return nnpops_ani_check(h, nullptr, nullptr);
    }
    if (words[1] == 0) return NNPOPS_OK;
    return nnpops_ani_check(h, nullptr, nullptr);                      // (statistics, growth, NNPOPS_ERR_CAPACITY)
}

int nnpops_ani_check(nnpops_ani_t h, int* max_radial_neighbors, int* max_angular_neighbors) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(NNPOPS_ERR_HIP, "cannot select device %d", h->device);
    int st[kStatWords] = {0, 0, 0, 0};
    // The builders flag their own overflow (and a void grid): when nobody asks for the statistics and nothing was flagged,
    // this is one 4-byte copy and a synchronisation -- no pass over the counts, no memset.  (The torch op checks after
    // every forward.)
    const bool want_stats = max_radial_neighbors || max_angular_neighbors || !h->cap_fitted || h->backward_kernel == 0;
    if (!want_stats) {
        NNPOPS_HIP_TRY(hipMemcpyAsync(st, h->d_status, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        NNPOPS_HIP_TRY(hipStreamSynchronize(h->stream));
        if (st[kStatOverflow] == 0) return NNPOPS_OK;
    }
    hipLaunchKernelGGL(ani_row_stats, dim3(std::min(64, div_up(h->hp.N, 256))), dim3(256), 0, h->stream, h->hp.N, h->d_cnt_a,
                       h->d_cnt_ro, h->cap, h->cap_angular, h->d_status);
    NNPOPS_HIP_TRY(hipGetLastError());
    NNPOPS_HIP_TRY(hipMemcpyAsync(st, h->d_status, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    NNPOPS_HIP_TRY(hipMemsetAsync(h->d_status, 0, sizeof(int) * kStatWords, h->stream));      // clean slate for the next build
    // The two angular kernels touch nothing but their own atom's records, so the order in which they walk the atoms is free.
    // In index or cell order the few atoms with the most triples -- work grows with the SQUARE of the angular neighbours: 4x
    // between the core and the surface of a 60-atom molecule, 45..300 triples in a liquid -- decide when the last occupancy
    // round ends.  Longest first: the schedule is a permutation by decreasing neighbour count, built here from the counts of
    // this frame and kept (a stale one is only a schedule; rebuilt whenever the statistics are).  128 conformers (7 589
    // atoms): angular forward 70 -> 46 us, backward 48 -> 38; 10 000-atom liquid: 19.2 -> 17.6 and 17.6 -> 15.7 us.
    std::vector<int> counts;
    const bool want_order = (!h->last_used_cells || h->lpt == 2) && h->hp.N >= 256 && h->lpt;
    if (want_order) {
        counts.resize((size_t)h->hp.N);
        NNPOPS_HIP_TRY(hipMemcpyAsync(counts.data(), h->d_cnt_a, sizeof(int) * counts.size(), hipMemcpyDeviceToHost, h->stream));
    }
    NNPOPS_HIP_TRY(hipStreamSynchronize(h->stream));
    if (want_order) {
        std::vector<int> perm(counts.size());
        for (size_t a = 0; a < perm.size(); a++) perm[a] = (int)a;
        std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return counts[a] > counts[b]; });
        NNPOPS_HIP_TRY(hipMemcpyAsync(h->d_work_order, perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice, h->stream));
        // The angular backward is launched class by class along this order (launch_angular): atoms with more than 44 angular
        // neighbours with the full pair matrix, up to 44 with a 48-slot one, up to 28 with a 32-slot one -- four slots of room for
        // the frames until the next check(); an atom that outgrows its class is flagged by the builder (kStatOverflow bit 3) and
        // the classes are rebuilt here.  Records of 32 slots: one class.
        std::vector<unsigned char> tile_of(perm.size(), 255);
        h->bwd_classes.clear();
        {
            double triples = 0;
            for (int cnt : counts) triples += 0.5 * cnt * (cnt - 1);
            h->mean_triples = triples / std::max<size_t>(counts.size(), 1);
            h->bwd_two_waves = h->mean_triples >= 200.0;
        }
        // (classes pay when every launch can fill the chip: the 61 199-atom conformer batch 304 -> 246 us; on a 7 600-atom block of
        //  it three launches are 3 us slower than one)
        if (h->cap_angular > 48 && h->bwd_by_class && h->hp.N >= h->bwd_class_atoms) {          // (record capacities are 32, 64, 128, ...)
            // The pair matrix of a class is as large as the 1 280-byte pieces it occupies anyway allow (a CU hands its LDS out in 128
            // such pieces: ani_angular_bwd.h): for "up to 44 neighbours" 47 slots are 12 pieces like 46, ten workgroups per CU where 48
            // slots are 13 pieces and nine; the top class is sized by the busiest atom of the frame (+ 2), not by the record capacity --
            // 59 slots are 18 pieces, seven workgroups, 64 slots 21 pieces, six.  An atom that outgrows its class is evaluated by the
            // clean-up launch (full capacity).
            auto pieces = [](int t) { return (32 * t + 4 * (t * (t + 1) + t * (t - 1) / 2) + 1279) / 1280; };
            auto snap = [&](int need) {
                int t = std::min(need, h->cap_angular);
                while (t < h->cap_angular && pieces(t + 1) == pieces(t)) t++;
                return t;
            };
            int busiest = 0;
            for (int cnt : counts) busiest = std::max(busiest, cnt);
            const int tiles[3] = {snap(busiest + 2), snap(46), snap(30)}, above[3] = {44, 28, -1};      // class c: atoms with more than above[c] neighbours
            // (a class of a few hundred atoms is a launch that cannot fill the chip: it takes the next class with it -- at the
            //  larger pair matrix -- until it has bwd_class_min atoms)
            int start = 0, pending = 0;
            for (int c = 0; c < 3; c++) {
                int end = start;
                while (end < (int)perm.size() && counts[perm[end]] > above[c]) end++;
                if (c < 2 && end - start < h->bwd_class_min) continue;     // (start stays: the next class begins where this one would have)
                double triples = 0;
                for (int q = start; q < end; q++) triples += 0.5 * counts[perm[q]] * (counts[perm[q]] - 1);
                const int tile = tiles[pending];               // (the largest matrix of the classes merged into this launch)
                h->bwd_classes.push_back({tile, start, end - start, end > start && triples / (end - start) >= 200.0});
                for (int q = start; q < end; q++) tile_of[perm[q]] = (unsigned char)std::min(255, tile);
                start = end;
                pending = c + 1;
            }
        }
        NNPOPS_HIP_TRY(hipMemcpyAsync(h->d_class_tile, tile_of.data(), tile_of.size(), hipMemcpyHostToDevice, h->stream));
        NNPOPS_HIP_TRY(hipStreamSynchronize(h->stream));      // (perm and tile_of are locals)
        h->work_order_valid = true;
    }
    // the backward pair matrix only needs to cover the busiest atom (larger atoms still work, tile by tile)
    h->tile = std::min(32, std::max(8, st[kStatMaxAngular]));      // exact: every row of LDS saved is occupancy
    // (tiny tiles: nothing to gain; and the fallback of an atom that outgrows the tile needs the per-slot accumulators plus a
    //  2 x 2 pair block inside the matrix region)
    h->compact_bwd = st[kStatMaxAngular] <= h->tile && h->tile >= 16 &&
                     h->tile * (h->tile + 1) + h->tile * (h->tile - 1) / 2 >= h->cap_angular * 4 + 12;
    if (max_radial_neighbors) *max_radial_neighbors = st[kStatMaxRow];
    if (max_angular_neighbors) *max_angular_neighbors = st[kStatMaxAngular];
    if (st[kStatOverflow] == 8) {         // nothing overflowed, but an atom outgrew its backward class: regrouped above.  The evaluation
                                          // itself stands (the clean-up launch of the backward takes such atoms, ani_angular_bwd.h): no
                                          // recompute, the new classes only make the next backward launches cheaper again.
        if (!want_order) {                 // (no order was built: lift every limit)
            NNPOPS_HIP_TRY(hipMemset(h->d_class_tile, 255, (size_t)h->hp.N));
            h->bwd_classes.clear();
        }
        return NNPOPS_OK;
    }
    if (st[kStatOverflow] & 4) {          // a cell holds more atoms than a bin of the two-kernel grid build: grow the bins
        const int old_bin = h->bin_cap;
        h->bin_cap *= 2;
        dev_free(h->d_bins);
        h->d_bins = nullptr;
        int rc = dev_alloc(&h->d_bins, (size_t)kBinnedCells * h->bin_cap);
        if (rc != NNPOPS_OK) return rc;
        h->computed = false;
        return fail(NNPOPS_ERR_CAPACITY, "cell bins overflowed (%d ids per cell); grown to %d, call compute() again", old_bin, h->bin_cap);
    }
    if (st[kStatOverflow] & 2) {
        if (h->algorithm == 2)
            return fail(NNPOPS_ERR_UNSUPPORTED, "cell list forced but the periodic box is fewer than 3 cells wide on some axis");
        h->cells_disabled = true;
        h->computed = false;
        return fail(NNPOPS_ERR_CAPACITY, "periodic box is fewer than 3 cells wide on some axis: switched to the all-pairs "
                                         "neighbour search, call compute() again");
    }
    if (st[kStatOverflow]) {
        const int old_cap = h->cap, old_ca = h->cap_angular;
        while (h->cap < st[kStatMaxRow]) h->cap *= 2;
        while (h->cap_angular < st[kStatMaxAngular]) h->cap_angular *= 2;
        if (h->cap_angular != old_ca) {    // (the classes above were cut for the old capacity)
            h->bwd_classes.clear();
            NNPOPS_HIP_TRY(hipMemset(h->d_class_tile, 255, (size_t)h->hp.N));
        }
        if (h->cap_angular > kMaxAngularCap)
            return fail(NNPOPS_ERR_UNSUPPORTED, "an atom has %d neighbours inside the angular cutoff (limit %d)",
                        st[kStatMaxAngular], kMaxAngularCap);
        int rc = alloc_rows(h);
        if (rc != NNPOPS_OK) return rc;
        h->computed = false;
        return fail(NNPOPS_ERR_CAPACITY,
                    "neighbour rows overflowed (max row %d > %d or max angular %d > %d); capacities grown to %d / %d, "
                    "call compute() again", st[kStatMaxRow], old_cap, st[kStatMaxAngular], old_ca, h->cap, h->cap_angular);
    }
    if (!h->cap_fitted) {
        // First clean check: fit the row capacity to the system (25 % + 8 entries of slack, multiple of 16: callers that
        // stop checking afterwards -- graph replays, a check interval of 0 -- keep that much room for density fluctuations;
        // include/nnpops_hip.h tells them to call check() from time to time).  The builder's
        // LDS per wave is proportional to it -- at the initial 128 a CU holds 27 builder waves, at 96 all 32 (-1 us per
        // launch at 10 000 atoms) -- and every row-indexed array shrinks with it.  Growth stays on demand, as before.
        h->cap_fitted = true;
        const int fit = std::max(32, (st[kStatMaxRow] + st[kStatMaxRow] / 4 + 8 + 15) & ~15);
        if (fit < h->cap && !std::getenv("NNPOPS_ANI_CAP")) {
            const int old_cap = h->cap;
            h->cap = fit;
            int rc = alloc_rows(h);
            if (rc != NNPOPS_OK) return rc;
            h->computed = false;
            return fail(NNPOPS_ERR_CAPACITY, "row capacity fitted to the system (longest row %d: %d -> %d), call compute() again",
                        st[kStatMaxRow], old_cap, h->cap);
        }
    }
    return NNPOPS_OK;
}

int nnpops_ani_overflow_word(nnpops_ani_t h, const int32_t** word) {
    NNPOPS_REQUIRE(h != nullptr && word != nullptr, "NULL argument");
    *word = reinterpret_cast<const int32_t*>(h->d_status + kStatOverflow);
    return NNPOPS_OK;
}

int nnpops_ani_read_overflow(nnpops_ani_t h, int32_t* value) {
    NNPOPS_REQUIRE(h != nullptr && value != nullptr, "NULL argument");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(NNPOPS_ERR_HIP, "cannot select device %d", h->device);
    NNPOPS_HIP_TRY(hipMemcpyAsync(value, h->d_status + kStatOverflow, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    NNPOPS_HIP_TRY(hipStreamSynchronize(h->stream));
    return NNPOPS_OK;
}

int nnpops_ani_describe(nnpops_ani_t h, char* text, int capacity) {
    NNPOPS_REQUIRE(h != nullptr && text != nullptr && capacity > 0, "NULL argument");
    const bool uni = h->fwd_uniform && h->hp.nFR == h->nfrp && h->hp.nFZ == h->nfzp;
    const bool shape2x = h->nfrp == 8 && h->nfzp == 4;
    std::snprintf(text, (size_t)capacity,
                  "forward=%s backward=%d generic=%d uniform=%d grid=%d literal=%d dynamic_quads=%d fused_build=%d cap=%d cap_angular=%d "
                  "chunk=%d classes=%d cells=%d scatter=%d row_major_walk=%d",
                  h->forward_kernel == 2 ? "mfma" : h->forward_kernel == 1 ? "chunked" : "merge", h->backward_kernel, (int)h->generic,
                  (int)uni, (int)(uni && h->fwd_grid), (int)(uni && h->fwd_grid && shape2x && h->fwd_literal), (int)h->fwd_dynamic,
                  (int)(h->computed ? h->last_fused_build : (h->fuse_forward < 0 ? h->hp.N <= kFuseAtoms : h->fuse_forward != 0)), h->cap, h->cap_angular,
                  // (the chunk the forward kernel is launched with, not the configured floor: ADVICE r04)
                  h->generic ? h->fwd_chunk : forward_chunk(h, (size_t)h->cap_angular * 2 * sizeof(float4), (size_t)(h->nfrp + h->nfzp) * sizeof(float)),
                  (int)h->bwd_classes.size(), (int)h->last_used_cells,
                  (int)h->scatter_now, h->hp.tri_row_major);      // (scatter: the last backprop() stored the leg forces in the receivers' rows)
    return NNPOPS_OK;
}

int nnpops_ani_enable_timing(nnpops_ani_t h, int enable) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    h->timing_mask = enable == 1 ? ~0u : (unsigned)enable >> 1;      // 1: all kernels; else bit (id + 1) selects kernel id
    for (int k = 0; k < NNPOPS_ANI_NUM_KERNELS; k++) { h->ev_used[k] = 0; h->timing_seen[k] = 0; }
    return NNPOPS_OK;
}

int nnpops_ani_set_timing_stride(nnpops_ani_t h, int every) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    NNPOPS_REQUIRE(every >= 1, "timing stride must be at least 1 (got %d)", every);
    h->timing_every = every;
    return NNPOPS_OK;
}

int nnpops_ani_set_timing_merge(nnpops_ani_t h, int merge) {
    NNPOPS_REQUIRE(h != nullptr, "NULL handle");
    h->timing_merge = merge != 0;
    return NNPOPS_OK;
}

int nnpops_ani_timing_overhead(nnpops_ani_t h, double* ms) {
    NNPOPS_REQUIRE(h != nullptr && ms, "NULL argument");
    DeviceGuard guard(h->device);
    hipEvent_t a, b;
    NNPOPS_HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableSystemFence));
    NNPOPS_HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableSystemFence));
    std::vector<float> samples;
    for (int k = 0; k < 21; k++) {                          // an EMPTY bracket, same events, same stream
        NNPOPS_HIP_TRY(hipEventRecord(a, h->stream));
        NNPOPS_HIP_TRY(hipEventRecord(b, h->stream));
        NNPOPS_HIP_TRY(hipStreamSynchronize(h->stream));
        float t = 0;
        NNPOPS_HIP_TRY(hipEventElapsedTime(&t, a, b));
        samples.push_back(t);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    std::sort(samples.begin(), samples.end());
    *ms = samples[samples.size() / 2];
    return NNPOPS_OK;
}

int nnpops_ani_get_timing(nnpops_ani_t h, double* total_ms, int* launches) {
    NNPOPS_REQUIRE(h != nullptr && total_ms && launches, "NULL argument");
    DeviceGuard guard(h->device);
    NNPOPS_HIP_TRY(hipStreamSynchronize(h->stream));
    for (int k = 0; k < NNPOPS_ANI_NUM_KERNELS; k++) {
        double sum = 0;
        for (size_t q = 0; q < h->ev_used[k]; q++) {
            float ms = 0;
            NNPOPS_HIP_TRY(hipEventElapsedTime(&ms, h->ev_start[k][q], h->ev_stop[k][q]));
            sum += ms;
        }
        total_ms[k] = sum;
        launches[k] = (int)h->ev_used[k];
        h->ev_used[k] = 0;
    }
    return NNPOPS_OK;
}

}  // extern "C"
