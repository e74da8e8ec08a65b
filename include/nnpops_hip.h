/*
 * nnpops_hip.h -- C ABI of libnnpops_hip.so, the MI355X (gfx950) implementation of the NNPOps
 * per-atom hot path: ANI symmetry functions, SchNet CFConv (+ neighbour list) and
 * getNeighborPairs.
 *
 * This is the drop-in boundary.  Every entry point replaces one method of the reference's
 * device-agnostic C++ core (the layer its torch binding calls into) or one torch dispatcher
 * kernel, and keeps that method's argument meaning:
 *
 *   nnpops_ani_create / _destroy     ANISymmetryFunctions ctor/dtor        src/ani/ANISymmetryFunctions.h:60-66
 *   nnpops_ani_set_stream            CudaANISymmetryFunctions::setStream   src/ani/CudaANISymmetryFunctions.h (setStream)
 *   nnpops_ani_compute               computeSymmetryFunctions              src/ani/ANISymmetryFunctions.h:78
 *   nnpops_ani_backprop              backprop                              src/ani/ANISymmetryFunctions.h:92
 *   nnpops_cfconv_neighbors_*        CFConvNeighbors ctor / build          src/schnet/CFConv.h:37-85
 *   nnpops_cfconv_create / compute / backprop   CFConv ctor / compute / backprop   src/schnet/CFConv.h:109-217
 *   nnpops_neighbor_pairs_forward / _backward   neighbors::getNeighborPairs forward/backward kernels
 *                                               src/pytorch/neighbors/getNeighborPairsCUDA.cu:31-101
 *   nnpops_pme_direct                           pme::pme_direct (computeDirect)   src/pytorch/pme/pmeCUDA.cu:30-100
 *
 * Conventions
 *   - plain C: opaque handles, raw pointers, sizes.  No torch / C++ types cross this boundary.
 *   - every pointer documented "device" is a HIP device pointer valid on the handle's device;
 *     "host" pointers are read during the call only.  float = IEEE fp32, indices = int32.
 *   - all work is enqueued on the handle's stream (default: the null stream); nothing
 *     synchronises the host unless the function says so.  Calls are graph-capturable unless noted.
 *   - every function returns 0 on success or a negative nnpops_status; nnpops_last_error()
 *     returns a thread-local message for the last failure.
 *   - a handle is not thread-safe and not re-entrant, like the reference's objects
 *     (backprop consumes state left by the last compute -- ANISymmetryFunctions.h:83-84).
 */
#ifndef NNPOPS_HIP_H
#define NNPOPS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    NNPOPS_OK = 0,
    NNPOPS_ERR_INVALID_ARGUMENT = -1,
    NNPOPS_ERR_HIP = -2,            /* a HIP runtime call failed; message carries hipGetErrorString */
    NNPOPS_ERR_UNSUPPORTED = -3,    /* configuration outside what the kernels were built for */
    NNPOPS_ERR_CAPACITY = -4,       /* a neighbour list outgrew its buffers (see *_check) */
    NNPOPS_ERR_NO_DEVICE = -5
} nnpops_status;

const char* nnpops_last_error(void);
/* Library / build identification: "nnpops_hip <version> gfx950". */
const char* nnpops_version(void);
/* Number of HIP devices visible; negative nnpops_status when the runtime is unusable. */
int nnpops_device_count(void);

/* ------------------------------------------------------------------------------------------
 * ANI symmetry functions (replaces ANISymmetryFunctions / CudaANISymmetryFunctions)
 * ------------------------------------------------------------------------------------------ */
typedef struct nnpops_ani* nnpops_ani_t;

/* radial_eta_rs:            host, [num_radial][2]   = {eta, rs}                 (RadialFunction,  ANISymmetryFunctions.h:29-32)
 * angular_eta_rs_zeta_ths:  host, [num_angular][4]  = {eta, rs, zeta, thetas}   (AngularFunction, ANISymmetryFunctions.h:34-39)
 * atom_species:             host, [num_atoms], values in [0, num_species)
 * periodic / torchani:      as the reference constructor flags
 * device:                   HIP device ordinal the handle lives on
 * Any list of angular functions is accepted, like the reference core (CpuANISymmetryFunctions.cpp:153-194).  A list that
 * is a full grid {(eta,rs)} x {(zeta,thetas)} of at most 16 x 8 distinct factors -- every set the reference's torch binding
 * can build (SymmetryFunctions.cpp:115-120), ANI-1x/1ccx/2x included -- runs on the factored matrix-core kernels; any other
 * list runs on generic kernels (same results, several times slower). */
int nnpops_ani_create(nnpops_ani_t* out, int num_atoms, int num_species, float radial_cutoff, float angular_cutoff,
                      int periodic, const int32_t* atom_species, int num_radial, const float* radial_eta_rs,
                      int num_angular, const float* angular_eta_rs_zeta_ths, int torchani, int device);
int nnpops_ani_destroy(nnpops_ani_t h);
/* stream: a hipStream_t passed as void* (NULL = null stream). */
int nnpops_ani_set_stream(nnpops_ani_t h, void* stream);

/* positions: device [num_atoms][3]; box: device [3][3] rows = box vectors (ignored, may be NULL,
 * when the handle is not periodic); radial: device [num_atoms][num_species][num_radial];
 * angular: device [num_atoms][num_species*(num_species+1)/2][num_angular].  Outputs are fully
 * overwritten.  Positions / box / neighbour lists are retained for backprop. */
int nnpops_ani_compute(nnpops_ani_t h, const float* positions, const float* box, float* radial, float* angular);
/* radial_deriv / angular_deriv: device, shapes as the outputs above; position_deriv: device
 * [num_atoms][3], fully overwritten.  Must follow a compute() on the same handle. */
int nnpops_ani_backprop(nnpops_ani_t h, const float* radial_deriv, const float* angular_deriv, float* position_deriv);
/* The same two calls for AEV / gradient arrays whose rows are embedded in wider rows: row i of the radial part
 * starts at radial + i * radial_ld (floats), likewise angular; 0 = dense.  Lets a caller keep ONE [num_atoms][W_r + W_a]
 * array (radial = aev, angular = aev + W_r, both strides W_r + W_a) -- the layout TorchANI's AEVComputer returns --
 * without a concatenation copy forward and a split copy backward. */
int nnpops_ani_compute_strided(nnpops_ani_t h, const float* positions, const float* box, float* radial, int radial_ld,
                               float* angular, int angular_ld);
int nnpops_ani_backprop_strided(nnpops_ani_t h, const float* radial_deriv, int radial_ld, const float* angular_deriv,
                                int angular_ld, float* position_deriv);
/* Blocks on the handle's stream and reports whether the last compute() overflowed a neighbour
 * buffer (NNPOPS_ERR_CAPACITY; the handle has then already grown its buffers, so simply call
 * compute() again).  max_radial_neighbors / max_angular_neighbors (host, may be NULL) receive the
 * largest per-atom counts seen.  Not graph-capturable.
 * The first clean check() fits the row capacity to the system (longest row + 25 % + 8, NNPOPS_ERR_CAPACITY once: call
 * compute() again).  A caller that then stops checking -- a captured graph replayed for many steps -- relies on that
 * slack: a row that outgrows it is clamped (the builders raise a device-side flag that only check() reads), so call
 * check() every few hundred replays, or after anything that can change the density. */
int nnpops_ani_check(nnpops_ani_t h, int* max_radial_neighbors, int* max_angular_neighbors);
/* The device-side flag itself: *word receives the DEVICE address of one int32 that the neighbour builders OR into whenever a
 * compute() overflows a capacity (bit 0 rows / records, bit 1 box too small for the cell stencil, bit 2 cell bins, bit 3 an
 * atom outgrew the class of its backward launch) and that only nnpops_ani_check() clears.  It is STICKY: a captured graph
 * that is replayed without any check leaves its overflows there, so a caller can (a) put its own 4-byte asynchronous copy of
 * the word into the graph, or read it every N replays, without calling into this library, and (b) rely on the next eager
 * check() -- any later compute() followed by check() -- to report NNPOPS_ERR_CAPACITY for what happened during the replays:
 * a non-zero word found there means that every evaluation since the previous check may have been incomplete.  The address is
 * valid for the life of the handle. */
int nnpops_ani_overflow_word(nnpops_ani_t h, const int32_t** word);
/* The same word read for the caller: waits for the handle's stream, copies the four bytes with THIS library's HIP runtime (a host
 * layer that opens a HIP runtime of its own to copy from the address above may get a second runtime that does not know the
 * pointer) and leaves the word as it is -- only nnpops_ani_check() clears it.  Additive. */
int nnpops_ani_read_overflow(nnpops_ani_t h, int32_t* value);
/* Which kernels this handle runs, as one line of `key=value` words (for logs and tests; the words may grow): forward= merge |
 * chunked | mfma; backward= kernel number; generic= the function list does not factor; uniform= one eta and one zeta; grid= eight
 * radial factors on equally spaced shifts (taken by recurrence); literal= the constants are the published ANI-2x set and the
 * forward kernels carry them as literals; dynamic_quads, fused_build, cap, cap_angular, chunk, classes, cells: state after the last
 * compute() / check().  Writes at most `capacity` bytes including the terminator.  Additive. */
int nnpops_ani_describe(nnpops_ani_t h, char* text, int capacity);
/* The same check in two halves for callers that have more work to queue behind compute(): _begin (right after compute) queues
 * one single-thread launch that publishes the overflow word and a stamp into pinned host memory and returns 1 -- or 0 when the check cannot be deferred (first calls, while
 * capacities are still being fitted): call nnpops_ani_check() then.  _end (after the consumers of this build have been launched;
 * they clamp their counts, so an overflowed build is incomplete but harmless to run) polls that stamp (no API call) and returns what
 * nnpops_ani_check() would: NNPOPS_OK, or NNPOPS_ERR_CAPACITY after growing the buffers -- compute() and everything queued
 * behind it must then be issued again.  Additive (the reference has no capacity check: its neighbour list is the N x N matrix). */
int nnpops_ani_check_begin(nnpops_ani_t h);
/* _begin without its launch, for a caller whose NEXT kernel on the stream can do the publishing itself (nnpops_mlp_forward with
 * the frame's publish_* fields): returns 1 and the three values to hand to that kernel, or 0 when the check cannot be deferred.
 * The caller must launch that kernel before nnpops_ani_check_end(), which otherwise waits for the stream and runs the full check. */
int nnpops_ani_check_begin_with(nnpops_ani_t h, const int32_t** word, int32_t** publish_to, int32_t* stamp);
int nnpops_ani_check_end(nnpops_ani_t h);
/* Neighbour search used by compute(): 0 = automatic, 1 = all-pairs scan (the reference's
 * algorithm, O(N^2)), 2 = cell list (O(N); periodic boxes must be at least 3 cells wide per axis). */
int nnpops_ani_set_neighbor_algorithm(nnpops_ani_t h, int algorithm);

/* Batched evaluation of independent NON-PERIODIC molecules in one handle (the reference has no batch
 * dimension: src/pytorch/SymmetryFunctions.py:110; this is the additive API SURVEY.md s8f ranks second).
 * Atoms [molecule_offsets[m], molecule_offsets[m+1]) form molecule m (host array, num_molecules+1 entries,
 * first 0, last num_atoms); atoms of different molecules never see each other.  compute()/backprop() are
 * unchanged -- every kernel is per atom -- so one launch sequence evaluates the whole batch.
 * num_molecules <= 0 restores the single-system behaviour. */
int nnpops_ani_set_molecules(nnpops_ani_t h, int num_molecules, const int32_t* molecule_offsets);

/* Per-kernel timing with HIP events recorded on the handle's stream around kernel launches
 * (off by default; not for use during graph capture).  enable: 0 = off, 1 = every kernel, any other
 * value = a mask with bit (id + 1) set for each kernel id to time (an event pair costs ~3 us of stream
 * time on MI355X, so a benchmark times only the kernel it reports on).  get_timing blocks on the
 * stream, returns for each kernel id the summed duration in milliseconds and the number of launches
 * since the last call / enable, and resets the counters.  Arrays have NNPOPS_ANI_NUM_KERNELS entries. */
enum {
    NNPOPS_ANI_K_NEIGHBORS = 0,       /* neighbour rows + records + triple lists + radial AEV (one launch) */
    NNPOPS_ANI_K_RADIAL_FWD = 1,      /* always 0 launches: the radial AEV is written by the neighbour kernel */
    NNPOPS_ANI_K_ANGULAR_FWD = 2,     /* 0 launches for small systems: build + radial + angular forward are one kernel there, timed as NEIGHBORS */
    NNPOPS_ANI_K_RADIAL_BWD = 3,
    NNPOPS_ANI_K_ANGULAR_BWD = 4,
    NNPOPS_ANI_K_CELL_GRID = 5,       /* the grid build in front of the neighbour kernel (2 or 5 launches; 0 for all-pairs) */
    NNPOPS_ANI_NUM_KERNELS = 6
};
int nnpops_ani_enable_timing(nnpops_ani_t h, int enable);
/* Bracket only every `every`-th launch of each selected kernel (default 1 = every launch): a benchmark that must
 * measure its kernel inside the timed region pays the ~3 us per event that way on a sample of the steps only. */
int nnpops_ani_set_timing_stride(nnpops_ani_t h, int every);
int nnpops_ani_get_timing(nnpops_ani_t h, double* total_ms, int* launches);
/* What an event pair reports for an EMPTY bracket on the handle's stream (median of 21, milliseconds; blocks):
 * subtract it from a per-launch average to compare with a profiler's kernel durations. */
int nnpops_ani_timing_overhead(nnpops_ani_t h, double* ms);
/* merge != 0: while timing is enabled there is ONE bracket around neighbour build + angular forward (its time is reported under the
 * neighbour build) and ONE around angular backward + radial backward (reported under the angular backward); the four single
 * brackets are off, the cell grid's stays.  (single bracket A) + (single bracket B) - (merged bracket A+B) is what ONE bracket adds
 * to the stream -- measured in place, same launches, same cache state -- which is how bench.py makes its event figures comparable
 * with rocprofv3's kernel durations.  merge = 0 restores the per-kernel brackets.  Nothing changes when timing is off. */
int nnpops_ani_set_timing_merge(nnpops_ani_t h, int merge);

/* ------------------------------------------------------------------------------------------
 * SchNet continuous-filter convolution (replaces CFConvNeighbors / CFConv and their Cuda* subclasses)
 * ------------------------------------------------------------------------------------------ */
typedef struct nnpops_cfconv_neighbors* nnpops_cfconv_neighbors_t;
typedef struct nnpops_cfconv* nnpops_cfconv_t;

int nnpops_cfconv_neighbors_create(nnpops_cfconv_neighbors_t* out, int num_atoms, float cutoff, int periodic, int device);
int nnpops_cfconv_neighbors_destroy(nnpops_cfconv_neighbors_t h);
int nnpops_cfconv_neighbors_set_stream(nnpops_cfconv_neighbors_t h, void* stream);
/* positions: device [num_atoms][3]; box: device [3][3] or NULL.  Builds the half list
 * {(i,j): j>i, r_ij^2 < cutoff^2} with stored distances (CFConv.h:57, CpuCFConv.cpp:104-115). */
int nnpops_cfconv_neighbors_build(nnpops_cfconv_neighbors_t h, const float* positions, const float* box);
/* Blocks; num_pairs (host) receives the number of half pairs of the last build;
 * NNPOPS_ERR_CAPACITY if that build overflowed (buffers already grown: build again). */
int nnpops_cfconv_neighbors_check(nnpops_cfconv_neighbors_t h, int* num_pairs);
/* Blocks; copies the half list of the last build to host arrays (test/diagnostic use):
 * pair_atoms [2][capacity] (row 0 = i, row 1 = j), distances [capacity]. */
int nnpops_cfconv_neighbors_export(nnpops_cfconv_neighbors_t h, int capacity, int32_t* pair_atoms, float* distances);

/* activation: 0 = shifted softplus, 1 = tanh (CFConv.h:114-117).
 * w1: host [width][num_gaussians] (the layout the reference core indexes, CpuCFConv.cpp:163);
 * b1: host [width]; w2: host [width][width] ([out][in]); b2: host [width]. */
int nnpops_cfconv_create(nnpops_cfconv_t* out, int num_atoms, int width, int num_gaussians, float cutoff, int periodic,
                         float gaussian_width, int activation, const float* w1, const float* b1, const float* w2,
                         const float* b2, int device);
int nnpops_cfconv_destroy(nnpops_cfconv_t h);
int nnpops_cfconv_set_stream(nnpops_cfconv_t h, void* stream);
/* input / output: device [num_atoms][width]; output fully overwritten (CFConv.h:169-171).
 * Widths 16, 32, ... 128 evaluate the filter network once per pair and keep one filter row per pair in a
 * scratch buffer owned by the convolution (num_atoms * row capacity / 2 rows of `width` floats; 164 MB for 10 000
 * atoms at width 128): it is allocated by the first compute()/backprop() with a given neighbour list, so run one
 * step before capturing a HIP graph.  For widths 32, 64, 96, 128 the dense layers are evaluated as split-fp16
 * matrix products with fp32 accumulation (every fp32 operand = two fp16 planes, 22 significant bits; at least as
 * accurate as a chain of fp32 FMAs) whenever the weights keep all operands inside the fp16 range; otherwise, or with
 * NNPOPS_CFCONV_SPLIT=0 in the environment at creation, on the fp32 matrix instruction. */
int nnpops_cfconv_compute(nnpops_cfconv_t h, nnpops_cfconv_neighbors_t neighbors, const float* positions,
                          const float* box, const float* input, float* output);
/* output_deriv: device [num_atoms][width]; input_deriv: device [num_atoms][width];
 * position_deriv: device [num_atoms][3]; both fully overwritten (CFConv.h:186-189). */
int nnpops_cfconv_backprop(nnpops_cfconv_t h, nnpops_cfconv_neighbors_t neighbors, const float* positions,
                           const float* box, const float* input, const float* output_deriv, float* input_deriv,
                           float* position_deriv);

/* ------------------------------------------------------------------------------------------
 * getNeighborPairs (replaces the neighbors::getNeighborPairs CUDA kernels)
 * ------------------------------------------------------------------------------------------ */
/* dtype: 0 = float32, 1 = float64 (positions, box, deltas, distances share it).
 * positions: device [num_atoms][3]; box: device [3][3] or NULL (no periodic wrap);
 * max_num_pairs: -1 = one slot per lower-triangle pair, k -> (row, col<row) as tril_indices;
 *                >0 = compacted list of that many slots.
 * neighbors: device int32 [2][P]; deltas: device [P][3]; distances: device [P]; num_pairs: device int32[1]
 * with P = num_atoms*(num_atoms-1)/2 or max_num_pairs.  All four are fully written:
 * unused slots hold -1 / NaN / NaN (getNeighborPairsCUDA.cu:137-139); num_pairs receives the number
 * of pairs within the cutoff, which may exceed P in compacted mode (surplus pairs are dropped).
 * In compacted mode the list is emitted in ascending pair order (deterministic, unlike the reference).
 * workspace: device scratch of at least nnpops_neighbor_pairs_workspace_bytes(num_atoms) bytes. */
int64_t nnpops_neighbor_pairs_workspace_bytes(int num_atoms);
int nnpops_neighbor_pairs_forward(int dtype, int num_atoms, const void* positions, const void* box, double cutoff,
                                  int64_t max_num_pairs, int32_t* neighbors, void* deltas, void* distances,
                                  int32_t* num_pairs, void* workspace, void* stream);
/* grad_positions: device [num_atoms][3], fully overwritten
 * (getNeighborPairsCUDA.cu:80-101: +g on neighbors[0], -g on neighbors[1], g = grad_deltas + deltas/distance*grad_distances). */
int nnpops_neighbor_pairs_backward(int dtype, int num_atoms, int64_t num_slots, const int32_t* neighbors,
                                   const void* deltas, const void* distances, const void* grad_deltas,
                                   const void* grad_distances, void* grad_positions, void* stream);
/* The same with caller-provided scratch (device, 8-byte aligned, nnpops_neighbor_pairs_backward_workspace_bytes(num_atoms) bytes;
 * the entry point above takes it from the stream-ordered allocator).  Where the reference adds six floating-point atomics per pair
 * (getNeighborPairsCUDA.cu:96-100), the contributions are added as fixed-point numbers (two 64-bit words per component) on one
 * scale for the call: the result does not depend on the order of the additions -- bitwise reproducible -- and is within 2^-80 of
 * the largest contribution per term of the exact sum.  A NaN / infinite contribution makes the gradients of ITS two atoms NaN
 * (the atoms the reference's atomicAdds poison) and leaves the others what they are without it. */
int64_t nnpops_neighbor_pairs_backward_workspace_bytes(int num_atoms);
int nnpops_neighbor_pairs_backward_ws(int dtype, int num_atoms, int64_t num_slots, const int32_t* neighbors,
                                      const void* deltas, const void* distances, const void* grad_deltas,
                                      const void* grad_distances, void* grad_positions, void* workspace, void* stream);

/* Backward WITHOUT atomics for a list the forward op emitted (round 6; replaces the atomicAdd scatter of getNeighborPairsCUDA.cu:80-101
 * by an owner-computes gather).  Such a list is grouped by neighbors[0] (rows ascending, see above); nnpops_neighbor_pairs_build_index
 * adds the TRANSPOSED view -- the slots sorted by neighbors[1], ascending slot inside an atom's group -- plus the first / last slot of
 * every atom's group on either side:
 *   index: device int32 [nnpops_neighbor_pairs_index_ints(num_atoms, num_slots)], 8-byte aligned; a function of `neighbors` alone, so
 *          the torch op builds it once in forward() when the positions require a gradient and saves it with the list;
 *   workspace: device scratch, 256-byte aligned, nnpops_neighbor_pairs_index_workspace_bytes(...) bytes (free after the call).
 * nnpops_neighbor_pairs_backward_indexed then computes the same grad_positions as nnpops_neighbor_pairs_backward_ws: one pass forms
 * g for every slot, one pass adds them up per atom (float64 accumulation, fixed order: bitwise reproducible; a NaN / infinite g
 * reaches exactly the two atoms of its pair, as the reference's atomics do); workspace: 32-byte aligned,
 * nnpops_neighbor_pairs_backward_indexed_workspace_bytes(dtype, num_slots) bytes.  A list that is NOT grouped by neighbors[0] (edited,
 * shuffled, of unknown origin) must go through nnpops_neighbor_pairs_backward_ws, which assumes nothing.  Additive. */
#define NNPOPS_PAIRS_INDEX_MAX_ATOMS 262144      /* nnpops_neighbor_pairs_build_index: 512 buckets of up to 512 atoms; beyond, NNPOPS_ERR_UNSUPPORTED */
int64_t nnpops_neighbor_pairs_index_ints(int num_atoms, int64_t num_slots);
int64_t nnpops_neighbor_pairs_index_workspace_bytes(int num_atoms, int64_t num_slots);
int nnpops_neighbor_pairs_build_index(int num_atoms, int64_t num_slots, const int32_t* neighbors, int32_t* index, void* workspace,
                                      void* stream);
int64_t nnpops_neighbor_pairs_backward_indexed_workspace_bytes(int dtype, int64_t num_slots);
int nnpops_neighbor_pairs_backward_indexed(int dtype, int num_atoms, int64_t num_slots, const int32_t* neighbors, const void* deltas,
                                           const void* distances, const void* grad_deltas, const void* grad_distances,
                                           const int32_t* index, void* grad_positions, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * PME, direct-space part (replaces computeDirect: src/pytorch/pme/pmeCUDA.cu:30-100, pmeCPU.cpp:75-163) -- the immediate
 * consumer of the pair list above (src/pytorch/pme/pme.py:163-165).  The reciprocal-space part is not built.
 * ------------------------------------------------------------------------------------------ */
/* positions: device [num_atoms][3]; charges: device [num_atoms]; neighbors int32 [2][num_pairs], deltas [num_pairs][3],
 * distances [num_pairs]: the outputs of nnpops_neighbor_pairs_forward (float32), slots holding -1 are skipped;
 * exclusions: device int32 [num_atoms][max_exclusions], every row sorted in DESCENDING order and padded with -1, symmetric
 * (pme.py:66-73,93); may be NULL when max_exclusions == 0.  alpha: Ewald splitting parameter; coulomb: 1/(4 pi eps0) in the
 * caller's units.  energy: device float[1]; position_deriv: device [num_atoms][3] = dE/dpositions; charge_deriv: device
 * [num_atoms] = dE/dcharges; all three fully overwritten.  workspace: device scratch of at least
 * nnpops_pme_direct_workspace_bytes(...) bytes -- about num_atoms x min(4 x num_pairs / num_atoms + 32, 2048) x 16 bytes (a row of
 * incoming contributions per atom; num_pairs is the capacity of the list, so a heavily padded list costs workspace up to that
 * clamp: 32 KiB per atom).  The table of exclusions must be SYMMETRIC: every atom takes the terms of its excluded pairs from its own
 * row (the Python wrapper checks it once).  Graph-capturable. */
int64_t nnpops_pme_direct_workspace_bytes(int64_t num_pairs, int num_atoms, int max_exclusions);
int nnpops_pme_direct(int num_atoms, int64_t num_pairs, int max_exclusions, const float* positions, const float* charges,
                      const int32_t* neighbors, const float* deltas, const float* distances, const int32_t* exclusions,
                      float alpha, float coulomb, float* energy, float* position_deriv, float* charge_deriv, void* workspace,
                      void* stream);
/* The same sums over the pair list's transposed index (round 6: nnpops_neighbor_pairs_build_index above) for a list the forward op of
 * getNeighborPairs emitted: one streaming pass over the slots + an owner-computes gather, no atomics, no rows of incoming entries,
 * float64 accumulation in a fixed order (bitwise reproducible).  `index` must have been built from exactly this `neighbors`; workspace:
 * nnpops_pme_direct_indexed_workspace_bytes(num_pairs, num_atoms) bytes (32 bytes per slot).  Everything else as nnpops_pme_direct,
 * which remains the entry point for lists of unknown origin.  Additive. */
int64_t nnpops_pme_direct_indexed_workspace_bytes(int64_t num_pairs, int num_atoms);
int nnpops_pme_direct_indexed(int num_atoms, int64_t num_pairs, int max_exclusions, const float* positions, const float* charges,
                              const int32_t* neighbors, const float* deltas, const float* distances, const int32_t* exclusions,
                              const int32_t* index, float alpha, float coulomb, float* energy, float* position_deriv,
                              float* charge_deriv, void* workspace, void* stream);

/* ---- dense layers of the ANI atomic networks (reference src/pytorch/BatchedNN.cpp:30-50, BatchedNN.py:37-122) ----
 * C[M x N] = A[M x K] B with fp32 in and out; the products run on the half-precision matrix instruction with every
 * operand carried as two fp16 planes (22 significant bits) and fp32 accumulation.  B arrives pre-split:
 * nnpops_split_planes writes the planes [rows][ldp] (ldp a multiple of 32, zero padded) of W [rows][cols] -- or of its
 * transpose: rows/cols then describe the OUTPUT, W is [cols][rows].  For y = x W^T with a torch Linear weight
 * W [out][in], B = planes(W) with rows = N = out, cols = K = in.  All pointers are device pointers. */
int nnpops_split_planes(void* stream, int rows, int cols, const float* w, long ldw, int transpose, void* hi, void* lo, long ldp);
/* batch: independent problems `stride*` elements apart (0 = shared operand).
 * epilogue 0: none; 1: C = CELU(C + bias[N], alpha); 2: C *= CELU'(Y) with Y [M][ldy] a saved CELU OUTPUT.
 * prologue 0: A as given; 1: A[m][k] = pv[k] * CELU'(PY[m][k]) (A itself is not read).
 * a_scale: A is multiplied by it before the split (and C divided by it): keep |A| * a_scale below 6e4.
 * a_rows / c_rows (optional, batch == 1): row m of A is read from row a_rows[m], row m of C is written to row c_rows[m]
 * -- the atoms of a species need not be gathered into a contiguous block first, nor their gradients scattered back. */
/* out[m] = A[m][0..K) . w + bias (the networks' last layer: one output per member, summed over the members); with
 * out_rows the result of row m goes to out[out_rows[m]]. */
int nnpops_rows_dot(void* stream, int M, int K, const float* A, long lda, const float* w, float bias, float* out, const int* out_rows);
int nnpops_gemm_split(void* stream, int M, int N, int K, int batch, const float* A, long lda, long strideA, const void* Bh,
                      const void* Bl, long ldb, long strideB, float* C, long ldc, long strideC, int epilogue, const float* bias,
                      long strideBias, const float* Y, long ldy, long strideY, int prologue, const float* PY, long ldpy,
                      long stridePY, const float* pv, long stridePv, float alpha, float a_scale, const int* a_rows, const int* c_rows);

/* ---- the atomic networks of a whole frame in two launches (replaces the four BatchedLinear + CELU calls and the
 * sum of BatchedNN.py:100-111 inside OptimizedTorchANI.py:49-52, and their autograd backward to the AEV) ----
 * Per atom and ensemble member: Linear(F,H1) CELU Linear(H1,H2) CELU Linear(H2,H3) CELU Linear(H3,1).  The atoms are
 * grouped by species ("kind"): rows[] lists, kind after kind, the row of x (and of dx) that holds each atom's AEV.
 * A workgroup carries 64 atoms of one kind and one member through all four layers with the activations in LDS /
 * registers; with_gradient the same launch runs the backward pass of layers 6, 4 and 2 and leaves dE/dy1 (split fp16
 * planes, workspace d1); nnpops_mlp_input_grad then forms dE/dx = W0^T dE/dy1 and writes the rows of dx.
 * Arithmetic: fp32 in and out; every operand of a product is carried as two fp16 planes (22 significant bits after a
 * scale of 2^-act_scale_log2 -- 1/16 by default --, exact products, fp32 accumulation) -- keep |activation| below 65 504 / scale.
 * Weights arrive packed (nnpops_mlp_pack): W [rows][cols] fp32 -> fragment planes of nnpops_mlp_packed_halves(rows, cols)
 * fp16 values; rows = outputs, cols = inputs of the product the planes are the left operand of.  For a torch Linear
 * weight W_l [out][in] of member m, with widths padded to multiples of 32 (zero rows / columns):
 *   w0  = pack(h1, F,  W_0, permute 0)                     w2 = pack(h2, h1, W_2, permute 1)    w4 = pack(h3, h2, W_4, 1)
 *   w4t = pack(h2, h3, W_4, transpose 1, permute 1)        w2t = pack(h1, h2, W_2, transpose 1, permute 1)
 * members one after the other in each buffer; w0t = pack(F, M*h1, [W_0 of all members stacked: M*h1 x F], transpose 1,
 * permute 1), one buffer for all members.  permute 1 selects the K order in which a matrix-core accumulator hands its
 * rows to the next product (mlp_fused.hip).  Biases b0 [M][h1], b2 [M][h2], b4 [M][h3], last layer w6 [M][h3], b6 [M]. */
#define NNPOPS_MLP_MAX_KINDS 8
typedef struct {
    int num_atoms;                       /* atoms of this kind: the next num_atoms entries of rows[] */
    int h1, h2, h3;                      /* packed layer widths: multiples of 32 in 32..256 */
    const void *w0, *w2, *w4;            /* forward planes (device) */
    const void *w4t, *w2t, *w0t;         /* gradient planes (device; may be NULL when with_gradient == 0) */
    const float *b0, *b2, *b4, *w6, *b6; /* device */
    void* d1;                            /* device workspace, nnpops_mlp_d1_halves(num_atoms, M, h1) fp16 values (gradient only) */
    const void* w0tm;                    /* optional: W_0^T member by member, pack(F, h1, W_0 of member m, transpose 1, permute 1), members one
                                          * after the other -- what the forward launch multiplies dE/dy1 with when the frame has dx_partial */
} nnpops_mlp_kind;
typedef struct {
    int num_kinds, num_features, num_members;
    const float* x; int ldx;             /* device [atoms][ldx], ldx >= num_features, rows 16-byte aligned; num_features % 8 == 0 */
    const int32_t* rows;                 /* device [sum of num_atoms] */
    float* energies;                     /* device [sum of num_atoms][num_members]: output of every (grouped atom, member) network */
    float alpha;                         /* CELU alpha (0.1 in TorchANI) */
    float* dx; int lddx;                 /* nnpops_mlp_input_grad: device [atoms][lddx]; rows listed in rows[] are overwritten */
    const float* upstream;               /* optional device scalar: dx is multiplied by it (dE_total/dE of this sum); NULL = 1 */
    float dx_scale;                      /* host scalar, also multiplied into dx (e.g. 1 / num_members for an ensemble mean); 0 is read as 1 */
    nnpops_mlp_kind kinds[NNPOPS_MLP_MAX_KINDS];
    /* Networks over a SUBSET of the columns of x.  The AEV blocks of species a molecule does not contain are structurally zero
     * (no neighbour of that species, no pair with it): their products need not be formed and their weights not be read.  With
     * x_groups (device, num_features / 16 entries; num_features % 16 == 0) the planes w0 / w0t / w0tm are packed over
     * num_features = 16 * (number of live blocks) columns and feature block f reads columns 16 * x_groups[f] .. + 15 of x (and
     * writes those of dx).  dead_groups lists the other 16-column blocks of dx: nnpops_mlp_input_grad sets them to zero.  The
     * result of the full product whenever the skipped columns of x are zero (up to the order of the fp32 additions: the live
     * columns share K steps differently).  NULL / 0: all columns, as before. */
    const int32_t* x_groups;
    const int32_t* dead_groups; int num_dead_groups;
    /* Optional device workspace [num_members][sum of num_atoms][num_features] floats, for num_features <= 256 and kinds with w0tm:
     * nnpops_mlp_forward(with_gradient) then also forms every member's W_0^T dE/dy1 (d1 never leaves the CU) and
     * nnpops_mlp_input_grad only adds the members up, in a fixed order. */
    float* dx_partial;
    /* Optional, honoured by nnpops_mlp_input_grad when dx_partial is set: the launch that adds the members up also takes the
     * energy mean of nnpops_mlp_energy_mean (mean_out, a device float) or nnpops_mlp_energy_mean_shifted (mean_shift and
     * mean_out_shifted, device doubles) with scale mean_scale -- same numbers, one launch fewer. */
    float mean_scale; float* mean_out; const double* mean_shift; double* mean_out_shifted;
    /* Optional, honoured by nnpops_mlp_forward: its first thread copies *publish_word to publish_to[1] and then stores publish_stamp
     * to publish_to[0] (system scope, release) -- the deferred capacity check of an ANI handle (nnpops_ani_check_begin_with) riding
     * along instead of taking a launch of its own.  All three come from that call; NULL / 0: nothing is published. */
    const int32_t* publish_word; int32_t* publish_to; int32_t publish_stamp;
    /* Operand scale of the fp16 planes: activations (and back-propagated gradients) are split after a scale of 2^-act_scale_log2,
     * so |activation| must stay below 65 504 * 2^act_scale_log2.  0 is read as 4 -- the fixed 1/16 (|activation| < 1e6) of rounds
     * 1-4; 4 .. 12 accepted.  A larger exponent buys range for networks whose weights admit large activations at the price of
     * the smallest activations' last bits (below 6e-5 * 2^act_scale_log2 the high plane is a denormal; the low plane still carries
     * the remainder): callers derive it from a bound on the activations (nnpops_amd/BatchedNN.py: _refresh_planes). */
    int act_scale_log2;
} nnpops_mlp_frame;
int64_t nnpops_mlp_packed_halves(int rows, int cols);
int64_t nnpops_mlp_d1_halves(int num_atoms, int num_members, int h1);
int nnpops_mlp_pack(void* stream, int rows, int cols, const float* w, long ldw, int transpose, int permute, void* out);
int nnpops_mlp_forward(void* stream, const nnpops_mlp_frame* frame, int with_gradient);
int nnpops_mlp_input_grad(void* stream, const nnpops_mlp_frame* frame);
/* out[0] = scale * sum(energies[0 .. count)) in double precision and a fixed order: the sum over atoms and the mean over
 * members of BatchedNN.py:109 (scale = 1 / num_members) in one small launch.  Device pointers. */
int nnpops_mlp_energy_mean(void* stream, const float* energies, int64_t count, float scale, float* out);
/* The same sum, promoted to double and shifted by the molecule's self energy: out[0] = (double)(float)(scale * sum) + shift[0] --
 * the `energies + self_energies` of the reference's EnergyShifter (pytorch/EnergyShifter.py:52) without a launch of its own.
 * `shift` and `out` are device doubles. */
int nnpops_mlp_energy_mean_shifted(void* stream, const float* energies, int64_t count, float scale, const double* shift, double* out);
/* out[i] = in[i] * (float)factor[0] for i < count; `factor` is a DEVICE scalar, a double when factor_is_double != 0 and a float
 * otherwise.  The backward of the one-node OptimizedTorchANI step (pytorch/OptimizedTorchANI.py:49-52): forces kept by the forward
 * pass times the gradient autograd hands in, in one launch. */
int nnpops_scale_by_scalar(void* stream, const float* in, int64_t count, const void* factor, int factor_is_double, float* out);

#ifdef __cplusplus
}
#endif
#endif /* NNPOPS_HIP_H */
