// ani_kernels.h -- gfx950 kernels for the ANI atomic-environment-vector symmetry functions.
//
// What is computed (reference src/ani/CpuANISymmetryFunctions.cpp, maths in SURVEY.md App. A):
//   radial [i][s][k]  = scale_R * sum_{j: spec j = s, r_ij < Rcr} fc(r_ij;Rcr) exp(-eta_k (r_ij-Rs_k)^2)      (ref :112-151)
//   angular[i][b][m]  = 2^(1-zeta_m) * sum_{j<k nbrs of i, r < Rca, b = pair(spec j, spec k)}
//                         (1+cos(theta-ths_m))^zeta_m exp(-eta_m((r_ij+r_ik)/2-Rs_m)^2) fc(r_ij)fc(r_ik)       (ref :153-194)
//   and the analytic position gradients of both                                                              (ref :196-353)
//
// How it is laid out for CDNA4 (a new design, not the reference's CUDA launch shapes):
//   * one WAVE (64 lanes) per centre atom, 1-4 waves per workgroup; everything an atom needs lives in that
//     wave's LDS slice, waves never talk to each other (wave_fence() only).
//   * the NEIGHBOUR BUILD does all the geometry and all the integer bookkeeping, once per evaluation -- and the
//     radial AEV while the row is in LDS -- and leaves behind streaming-friendly arrays; the kernels that follow
//     are pure load + arithmetic:
//       rows  [N][cap]     16-byte records {dx, dy, dz, (species<<24)|atom} of every neighbour within
//                          Rcr: angular ones (r < Rca) packed from the front, radial-only from the back;
//       recA  [N][capA]    {dx, dy, dz, r}                    angular neighbours SORTED BY SPECIES
//       recB  [N][capA]    {fc, dfc/dr, 1/r, (species<<24)|atom}
//       ids   [N][capA]    atom ids in record order, -1 padded (reverse lookup of the backward gather)
//       tri   [N][capT]    the atom's n(n-1)/2 neighbour pairs in bucket-major order, one word
//                          p | q<<8 | bucket<<16 (bucket = species-pair block of the output row)
//       bucket_offsets [N][NB+1]   first triple of every bucket (chunked forward kernel)
//   * the angular functions factor as R_a(rbar) x Z_z(theta) (8 x 4 for ANI-2x): a triple costs
//     nFR exp2 + nFZ (log2+exp2) instead of nA (powf+cosf+expf).
//   * (the round-1 forward / backward kernels this header used to hold are in ani_fallback_kernels.h: fallbacks only)
//   * no float atomics anywhere, LDS or global: ds_add_f32 costs ~3 cycles per active lane on gfx950
//     (tools/ubench/lds_atomic.hip); every accumulation below has exactly one owner, results are bitwise
//     reproducible.
//   * forward, per batch of 64 triples: phase 1 lane = triple computes the 12 factors into LDS;
//     phase 2 lane = (stream, a) contracts them into the atom's LDS output row.  Two kernels: run merging
//     (many equally likely species) and a chunked view of the triple list (few well-filled species pairs).
//   * backward: lane = triple; the force of a triple on its legs is alpha_p A + beta B / alpha_q B + beta A, the
//     scalars go to an LDS pair matrix (one writer per entry), row sums give per-slot forces, which are PARKED
//     in leg_force / centre_force ...
//   * ... and gathered by the radial backward kernel, which is owner-computes (each atom walks its full row,
//     reading both gradient rows, then looks itself up in its angular neighbours' id rows) and is the only
//     writer of position_deriv.
#pragma once

#include "celllist.h"
#include "device_common.h"

namespace nnpops {

constexpr int kMaxRadialFns = 64;
constexpr int kMaxFactor = 16;       // max distinct (eta,rs) or (zeta,thetas) factors of the angular set
constexpr int kMaxAngularFns = 256;
constexpr int kMaxSpecies = 16;
constexpr int kMaxBuckets = kMaxSpecies * (kMaxSpecies + 1) / 2;
constexpr int kMaxAngularCap = 256;  // p and q of a triple word are 8 bits each

// Eight radial factors of one eta on equally spaced shifts (ANI-2x: ShfA = 0.8 + 0.3375 a): g_a = 2^(c (x - Rs_a)^2) obeys
//     g_{a+1} / g_a = u_a = 2^(c d^2 - 2 c d y_a),   g_{a-1} / g_a = 2^(c d^2 + 2 c d y_a),   u_{a+1} = u_a 2^(2 c d^2)        (y_a = x - Rs_a)
// so the eight of them cost FOUR transcendentals (g_1, g_5 and the two ratios at a = 1) and twelve multiplies instead of eight
// (v_exp_f32 runs at a quarter of the rate of a multiply).  Two starting points keep every chain at two steps: the rounding of the
// ratios' exponents (|e| < 32: 1e-6 absolute) enters a factor at most three times.  Constants from the host, in double.
struct GeoRadial {
    float rs1, c, k1, k0;            // Rs_1;  -eta log2(e);  -2 c d;  c d^2
    float q, q4, qi4, d4;            // 2^(2 c d^2), its 4th power and the inverse of that; 4 d
};
// The angular factor constants of the published ANI-2x set (EtaA 12.5, Zeta 14.1, ShfA 0.8 + 0.3375 a, ShfZ (2 z + 1) pi / 8;
// reference src/ani/BenchmarkCudaANISymmetryFunctions.cu:101-153), exactly as nnpops_ani_create derives them from those parameters.
// A handle whose derived constants equal these bit for bit (checked there; anything else keeps the constants in registers) runs
// forward kernels that carry them as LITERALS: the matrix-core forward kernel keeps ~110 wave-uniform values alive against 94 scalar
// registers at seven waves per SIMD, the overflow is parked in vector lanes (v_writelane / v_readlane: vector-issue slots, the
// resource these kernels are bound by), and these 18 constants are most of it -- 19 parked values become 6, the forward kernel
// 17.0 -> 15.8 us at 10 000 atoms (round 4, interleaved A/B).
struct Ani2xAngular {
    static constexpr float zeta = 0x1.c333340000000p+3f, zbias = -0x1.a333340000000p+3f;
    static constexpr float zc0 = 0x1.d906bc0000000p-1f, zc1 = 0x1.87de2a0000000p-2f, zc2 = -0x1.87de280000000p-2f, zc3 = -0x1.d906bc0000000p-1f;      // cos(ShfZ)
    static constexpr float zs0 = 0x1.87de2c0000000p-2f, zs1 = 0x1.d906bc0000000p-1f, zs2 = 0x1.d906be0000000p-1f, zs3 = 0x1.87de2e0000000p-2f;      // sin(ShfZ)
    static constexpr float rs_0 = 0x1.99999a0000000p-1f, rs_1 = 0x1.2333340000000p+0f, rs_2 = 0x1.79999a0000000p+0f, rs_3 = 0x1.d000000000000p+0f, rs_4 = 0x1.1333340000000p+1f, rs_5 = 0x1.3e66660000000p+1f, rs_6 = 0x1.69999a0000000p+1f, rs_7 = 0x1.94cccc0000000p+1f;      // ShfA
    static constexpr float negeta = -0x1.9000000000000p+3f;
    static constexpr float rs1 = 0x1.2333340000000p+0f, c = -0x1.2089fc0000000p+4f, k1 = 0x1.8587140000000p+3f, k0 = -0x1.06ee600000000p+1f, q = 0x1.daf9060000000p-5f, q4 = 0x1.7b326e0000000p-17f, qi4 = 0x1.59a8220000000p+16f, d4 = 0x1.5999980000000p+0f;      // GeoRadial
};
typedef float v2f __attribute__((ext_vector_type(2)));
// The two chains (from g_1 and from g_5) side by side in the halves of packed registers: R04 = {g_0, g_4}, R15 = {g_1, g_5},
// R26, R37; Y = {x - Rs_1, x - Rs_5}.
__device__ __forceinline__ void radial_factors_geo8(float x, const GeoRadial& G, v2f& R04, v2f& R15, v2f& R26, v2f& R37, v2f& Y) {
    const float y1 = x - G.rs1;
    Y = v2f{y1, y1 - G.d4};
    const v2f arg = (v2f{G.c, G.c} * Y) * Y;
    R15 = v2f{fast_exp2(arg.x), fast_exp2(arg.y)};
    const float e1 = G.k1 * y1;
    const float u1 = fast_exp2(G.k0 + e1), d1 = fast_exp2(G.k0 - e1);
    const v2f U = v2f{u1, u1} * v2f{1.0f, G.q4}, D = v2f{d1, d1} * v2f{1.0f, G.qi4};
    R04 = R15 * D;
    R26 = R15 * U;
    R37 = R26 * (U * v2f{G.q, G.q});
}

struct AniParams {
    int N, S, nR, nA, NB, nFR, nFZ;
    int periodic, torchani;
    float rcr, rca, rcr2, rca2;
    float inv_rcr, inv_rca;          // 1 / cutoff, rounded once on the host
    int kp_shift;                    // log2 of KP, the smallest power of two >= nR (lane = (stream, k) layouts)
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
    float fz_bias[kMaxFactor];       // 1 - zeta_z
    float scale_m[kMaxAngularFns];   // 2^(1-zeta_m) of angular function m                       ref :104-109
    int c_of_m[kMaxAngularFns];      // function m -> slot a*NFZP+z inside a padded canonical bucket block
    int bkt_a[kMaxBuckets];          // bucket b -> its species pair (A <= B), upper-triangular row-major
    int bkt_b[kMaxBuckets];          //                                                          ref :39-43
    // the angular functions as given (generic kernels for sets that do not factor; function m of the caller's list)
    float af_c[kMaxAngularFns];      // -eta_m * log2(e)
    float af_eta[kMaxAngularFns];
    float af_rs[kMaxAngularFns];
    float af_zeta[kMaxAngularFns];
    float af_cos[kMaxAngularFns];    // cos(thetas_m)
    float af_sin[kMaxAngularFns];    // sin(thetas_m)
    int* bucket_offsets;             // [N][NB + 1] device array the builders fill: offsets of the buckets in an atom's triple list
    const unsigned char* class_tile; // [N] pair-matrix edge of the backward launch this atom was put in by check() (255: no limit);
                                     //     a builder that finds more angular neighbours than that flags kStatOverflow bit 3
    int tri_row_major;               // builders walk the pairs of an atom row-major (1) or as a folded rectangle (0): decode_pair_folded
    // matrix-core forward kernel (ani_angular_mfma.h)
    int m_of_c[kMaxAngularFns];      // canonical slot a*NFZP+z -> function m, -1 for padding slots
    int fwd_split;                   // K: every species pair that can occur is shared by K quads (1, 2, 4 or 8)
    int fwd_slot_bucket[32];         // quad slot (set * 16 + quad) -> bucket, -1 unused; slots of a bucket are consecutive
    int fwd_nabsent;                 // species pairs that cannot occur in this system (their output blocks are zero)
    int fwd_absent[kMaxBuckets];
    int fwd_zero_shift;              // log2 of the lanes that zero one absent block (16 bytes each), <= 6
    GeoRadial geo;                   // UNI forward kernel with eight radial factors (valid when the host set nnpops_ani::fwd_grid)
};

// What the angular BACKWARD kernel needs of the parameter block, passed BY VALUE in the kernel arguments.  Read through the AniParams
// pointer, the factor constants cost every workgroup a chain of dependent scalar loads before its first useful instruction (kernel
// argument -> pointer -> counts -> the conditionally loaded constants).  The arrays are padded on the host with the neutral value of
// every factor slot, so the loads are unconditional and leave with the kernel arguments' own request: 15.9 -> 15.4 us at 10 000 atoms
// (round 4, interleaved A/B against the round-3 library).  The FORWARD kernel keeps the pointer: with these 60 values as kernel
// arguments it parks even more scalars in vector lanes and measured 17.7 -> 18.5 us.
struct AngularConsts {
    int N, nA;
    float fr_c[kMaxFactor], fr_rs[kMaxFactor], fr_negeta[kMaxFactor];       // -eta log2(e), Rs, -eta; 0 behind nFR
    float fz_zeta[8], fz_cos[8], fz_sin[8], fz_bias[8];                     // zeta (1 behind nFZ), cos / sin(thetas), 1 - zeta
};

// status words reported by nnpops_ani_check
enum { kStatOverflow = 0, kStatMaxRow = 1, kStatMaxAngular = 2, kStatWords = 4, kStatAlloc = 8 };

__host__ __device__ inline int triples_capacity(int capA) { return capA * (capA - 1) / 2; }

// Largest row / angular count of the last neighbour build, reduced from the per-atom counts only when
// the host asks (nnpops_ani_check).  Doing this with per-wave atomics inside the builders serialised
// 10k waves on one L2 word (measured ~240 us), so the builders publish nothing but their counts.
static __global__ __launch_bounds__(256) void ani_row_stats(int N, const int* __restrict__ cnt_a, const int* __restrict__ cnt_ro,
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

// Clamp the per-atom counts so that nobody indexes outside a row even after an overflow (results of
// an overflowed compute are garbage and reported through nnpops_ani_check).  Builders and consumers
// use the same clamp.
// cnt_pos word of an atom (by position of the radial backward's walk): angular count (9 bits), species (7), radial-only count (16)
__device__ __forceinline__ int pack_cnt_pos(int na, int nro, int species) { return min(na, 0x1ff) | ((species & 0x7f) << 9) | (min(nro, 0xffff) << 16); }
__device__ __forceinline__ void unpack_cnt_pos(int word, int& na, int& nro, int& species) {
    na = word & 0x1ff; species = (word >> 9) & 0x7f; nro = (int)((unsigned)word >> 16);
}

__device__ __forceinline__ void clamp_counts(int raw_a, int raw_ro, int cap, int cap_angular, int& na, int& nro) {
    na = max(0, min(raw_a, min(cap, cap_angular)));
    nro = max(0, min(raw_ro, cap - na));
}

// =============================================================================================
// Triple enumeration helpers (used by the builders; the consumers just read the word lists).
// =============================================================================================
// The pairs p < q of n sorted slots as a rectangle without a square root (round 5): row p of the strict upper triangle holds
// n - 1 - p pairs, so rows R and n - 1 - R together hold n - 1 -- fold the triangle into ceil(n / 2) rows of n - 1 columns:
//     t -> (R, C) = divmod(t, n - 1);   C < n - 1 - R:  pair (R, R + 1 + C);   else:  pair (n - 1 - R, C + 1)
// Every pair appears exactly once; for odd n the middle row pairs with itself and its second half is void (valid = false).  The
// quotient comes from one multiplication by 1 / (n - 1): exact for every t < 2^15, n <= 256 (the error of the product is below
// 2e-5 where the nearest quotient boundary is 0.5 / 255 away).  ~9 vector instructions where the row-major decode with its square
// root, two correction steps and two 24-bit products took ~25; the ORDER in which a wave meets the pairs is irrelevant: every pair
// computes its own place in the bucket-major list.
// Which walk a handle takes is decided by where the lists go: while everything a build writes stays inside the 256 MiB Infinity
// Cache the builders are bound by instruction issue and the cheaper decode wins (10 000-atom liquid 19.9 -> 19.1 us, a 7 600-atom
// block of conformers 18.3 -> 16.4 us); the 1 024-conformer batch writes 490 MB of list capacity, is bound by its stores, and loses
// 4 % with the folded walk (two rows of different species per wave: the 4-byte stores of one instruction touch more lines) --
// AniParams::tri_row_major, set by the host from N x capacity (ani.hip: alloc_rows).
__device__ __forceinline__ int folded_pair_count(int n) { return __mul24((n + 1) >> 1, n - 1); }
// (p, q) with p < q of the t-th pair in row-major order of the strict upper triangle of an n x n grid
__device__ __forceinline__ void decode_pair_row_major(int t, int n, int& p, int& q) {
    const float w = (float)(2 * n - 1);
    int pp = (int)((w - fast_sqrt(fmaxf(w * w - 8.0f * (float)t, 0.f))) * 0.5f);
    pp = max(0, min(pp, n - 2));
    // offset(p) = p*(2n-p-1)/2 ; one fix-up step each way covers the rounding of the fast sqrt
    if (__mul24(pp + 1, 2 * n - pp - 2) / 2 <= t) pp++;
    if (__mul24(pp, 2 * n - pp - 1) / 2 > t) pp--;
    p = pp;
    q = t - __mul24(pp, 2 * n - pp - 1) / 2 + pp + 1;
}
__device__ __forceinline__ bool decode_pair_folded(int t, int n, float inv_nm1, int& p, int& q) {
    const int R = (int)(((float)t + 0.5f) * inv_nm1);
    const int C = t - __mul24(R, n - 1);
    const int mirror = n - 1 - R;
    const bool first = C < mirror;
    p = first ? R : mirror;
    q = C + 1 + (first ? R : 0);
    return first || mirror != R;
}

// Per-atom species bookkeeping in LDS (ints), sized by the actual species / bucket counts.
struct AtomGroups {
    int* gs;      // [S]  first sorted slot of species s
    int* gn;      // [S]  number of neighbours of species s
    int* run;     // [S]  scratch for the stable sort
    int* boff;    // [NB + 1] exclusive offsets of the species-pair buckets in bucket-major triple order
    int* ba;      // [NB] first species of bucket b   (LDS copy of AniParams::bkt_a)
    int* bb;      // [NB] second species of bucket b
    unsigned char* ssp;   // [cap] species of the neighbour in every sorted slot
};
__host__ __device__ inline size_t group_ints(int S, int NB, int cap) { return (size_t)3 * S + 3 * NB + 1 + (cap + 3) / 4; }

__device__ __forceinline__ AtomGroups carve_groups(int* base, int S, int NB) {
    AtomGroups G;
    G.gs = base; G.gn = G.gs + S; G.run = G.gn + S; G.boff = G.run + S;
    G.ba = G.boff + NB + 1; G.bb = G.ba + NB;
    G.ssp = (unsigned char*)(G.bb + NB);
    return G;
}

// Bucket-major triple order: bucket b = (A <= B) holds gn[A]*gn[B] pairs (A < B, row-major over
// (ia, ib)) or gn[A](gn[A]-1)/2 pairs (A == B, strict upper triangle).  Returns the triple count.
__device__ __forceinline__ int build_bucket_offsets(int NB, const AtomGroups& G) {
    const int lane = lane_id();
    int carry = 0;
    for (int base = 0; base < NB; base += 64) {
        const int bk = base + lane;
        int c = 0;
        if (bk < NB) {
            const int A = G.ba[bk], B = G.bb[bk];
            const int ga = G.gn[A], gb = G.gn[B];
            c = (A == B) ? __mul24(ga, ga - 1) / 2 : __mul24(ga, gb);          // (24-bit multiplies run at full rate, v_mul_lo_u32 at a quarter)
        }
        const int incl = wave_prefix_sum(c);
        if (bk < NB) G.boff[bk] = carry + incl - c;
        carry += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) G.boff[NB] = carry;
    wave_fence();
    return carry;
}

// =============================================================================================
// Neighbour build.
// =============================================================================================
// Appends the lanes flagged in_a / in_ro to the two ends of the row being assembled in LDS (`stage`, same layout
// as the global row; ballot compaction keeps scan order).  The global row is written once, coalesced, after
// the scan (flush_row): the radial sums and the species sort that follow work from the LDS copy.
__device__ __forceinline__ void append_to_row(int cap, float4* stage, bool in_a, bool in_ro, float dx, float dy, float dz,
                                              int word, int& na, int& nro) {
    const unsigned long long ma = __ballot(in_a), mro = __ballot(in_ro);
    const float4 rec = make_float4(dx, dy, dz, __int_as_float(word));
    if (in_a) {
        const int slot = na + prefix_popc(ma);
        if (slot < cap) stage[slot] = rec;
    }
    if (in_ro) {
        const int slot = nro + prefix_popc(mro);
        if (slot < cap) stage[cap - 1 - slot] = rec;
    }
    na += __popcll(ma);
    nro += __popcll(mro);
}

__device__ __forceinline__ void flush_row(float4* __restrict__ row, const float4* stage, int cap, int na, int nro) {
    // the global row is CONTIGUOUS: angular neighbours first, the radial-only ones behind them (the stage keeps them
    // at its two ends because the counts are only known after the scan); a consumer can issue row[lane] before it
    // has the counts
    const int front = min(na, cap), back = min(nro, cap - front);
    for (int e = lane_id(); e < front + back; e += 64) store_wt(row + e, e < front ? stage[e] : stage[cap - 1 - (e - front)]);
}

// After the scan: sort the staged angular neighbours by species (stable), evaluate everything that
// depends on one neighbour only (r, cutoff function and derivative, 1/r), write the sorted records
// and the bucket-major triple list.  Runs once per atom per evaluation; forward and backward reuse it.
__device__ __forceinline__ void finalize_angular(const AniParams* __restrict__ P, const float4* stage, int n,
                                                 float4* __restrict__ recA, float4* __restrict__ recB,
                                                 int* __restrict__ ids, int capA, int* __restrict__ tri,
                                                 int* __restrict__ boff_out, const AtomGroups& G,
                                                 float4* recA_l = nullptr, float4* recB_l = nullptr, int* tri_l = nullptr) {
    // (recA_l / recB_l / tri_l: LDS copies for a forward pass that follows in the same workgroup, ani_build_forward.h)
    const int lane = lane_id();
    const int S = P->S, NB = P->NB;
    const float inv_rca = P->inv_rca;
    for (int bk = lane; bk < NB; bk += 64) { G.ba[bk] = P->bkt_a[bk]; G.bb[bk] = P->bkt_b[bk]; }
    auto emit = [&](const float4& r4, int rank) {
        const float r = fast_sqrt(r4.x * r4.x + r4.y * r4.y + r4.z * r4.z);       // ~1 ulp, like everything downstream
        float sn, cs;
        sincospi_unit(r * inv_rca, sn, cs);                // fc = (cos(pi r/Rc)+1)/2, ref :381-387
        const float4 a = make_float4(r4.x, r4.y, r4.z, r);
        const float4 b2 = make_float4(0.5f * cs + 0.5f, -(0.5f * kPi * inv_rca) * sn, fast_rcp(r), r4.w);
        G.ssp[rank] = (unsigned char)(__float_as_int(r4.w) >> kTagShift);
        store_wt(recA + rank, a);
        store_wt(recB + rank, b2);
        if (recA_l) { recA_l[rank] = a; recB_l[rank] = b2; }
        store_wt(ids + rank, __float_as_int(r4.w) & kIdMask);        // compact copy for the backward gather's reverse lookup
    };
    if (n <= 64) {
        // one neighbour per lane: the stable species sort is S ballots, no LDS traffic, no fences
        const bool valid = lane < n;
        const float4 r4 = valid ? stage[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        const int sp = valid ? (__float_as_int(r4.w) >> kTagShift) : -1;
        int rank = 0, accum = 0, my_gs = 0, my_gn = 0;
        for (int s = 0; s < S; s++) {
            const unsigned long long m = __ballot(sp == s);
            const int c = __popcll(m);
            if (sp == s) rank = accum + prefix_popc(m);
            if (lane == s) { my_gs = accum; my_gn = c; }
            accum += c;
        }
        if (lane < S) { G.gs[lane] = my_gs; G.gn[lane] = my_gn; }     // S <= kMaxSpecies <= 64
        if (valid) emit(r4, rank);
        wave_fence();
    } else {
        for (int s = lane; s < S; s += 64) { G.gn[s] = 0; G.run[s] = 0; }
        wave_fence();
        for (int e = lane; e < n; e += 64) atomicAdd(&G.gn[__float_as_int(stage[e].w) >> kTagShift], 1);   // int LDS atomics
        wave_fence();
        if (lane == 0) {
            int accum = 0;
            for (int s = 0; s < S; s++) { G.gs[s] = accum; accum += G.gn[s]; }
        }
        wave_fence();
        for (int base = 0; base < n; base += 64) {
            const int e = base + lane;
            const bool valid = e < n;
            const float4 r4 = valid ? stage[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            const int sp = valid ? (__float_as_int(r4.w) >> kTagShift) : -1;
            int rank = 0;
            for (int s = 0; s < S; s++) {
                const unsigned long long m = __ballot(valid && sp == s);
                if (m == 0) continue;                         // wave-uniform
                if (sp == s) rank = G.gs[s] + G.run[s] + prefix_popc(m);
                wave_fence();
                if (lane == 0) G.run[s] += __popcll(m);
                wave_fence();
            }
            if (valid) emit(r4, rank);
        }
    }
    for (int e = n + lane; e < capA; e += 64) store_wt(ids + e, -1);    // the gather scans whole rows: no stale ids behind the list
    const int T = build_bucket_offsets(NB, G);
    for (int bk = lane; bk <= NB; bk += 64) store_wt(boff_out + bk, G.boff[bk]);       // for the forward kernel's chunked view
    // The triple list, bucket-major (bucket b = (A <= B) holds its pairs in row-major (ia, ib) order), written by lanes that
    // walk the PAIRS (p < q) of the sorted slots in row-major order and compute where each one goes: the bucket is the
    // species pair of the two slots and the place inside it a multiply-add, where finding the pair of a given place costs a
    // binary search over the bucket offsets and a division or a square root per lane (~100 vector instructions per batch
    // of 64 against ~45).  Same list, scattered 4-byte stores inside the atom's own few cache lines.
    wave_fence();                                              // (G.ssp, G.boff)
    // (P->tri_row_major: the handle asks for the row-major walk -- see AniParams)
    const bool row_major = P->tri_row_major != 0;
    const int E = T <= 0 ? 0 : row_major ? T : folded_pair_count(n);      // (n <= 1: no pairs, and no 1 / (n - 1))
    const float inv_nm1 = __builtin_amdgcn_rcpf((float)max(n - 1, 1));
    for (int t = lane; t < E; t += 64) {
        int p, q;
        if (row_major) decode_pair_row_major(t, n, p, q);
        else if (!decode_pair_folded(t, n, inv_nm1, p, q)) continue;
        const int A = G.ssp[p], B = G.ssp[q];                  // A <= B: the slots are sorted by species
        const int bucket = __mul24(A, S) - __mul24(A, A - 1) / 2 + (B - A);        // upper-triangular row-major, as AniParams::bkt_a / bkt_b
        const int ia = p - G.gs[A], ib = q - G.gs[B], gb = G.gn[B];
        const int local = A == B ? __mul24(ia, 2 * gb - ia - 1) / 2 + (ib - ia - 1) : __mul24(ia, gb) + ib;
        const int at = G.boff[bucket] + local;
        const int word = p | (q << 8) | (bucket << 16);
        store_wt(tri + at, word);
        if (tri_l) tri_l[at] = word;
    }
    // (Round 4: assembling the list in LDS -- the row mirror is dead by then -- and writing it out as ~10 wave-wide 16-byte stores
    //  instead of 153 scattered 4-byte ones was built and measured: 19.6 -> 19.8 us, interleaved.  The 3 us this list costs the
    //  builder (profiles/r04a_probe_10k.json) are its decode / place arithmetic and its six dependent LDS lookups per pair, not
    //  its stores.)
}

// LDS of one builder wave: the row mirror [cap] float4 | radial scratch r, fc, species [3][cap] | radial bins
// [64 * S] | species groups
__host__ __device__ inline size_t builder_lds_bytes(int cap, int S, int NB) {
    return (size_t)cap * (sizeof(float4) + 3 * sizeof(float)) + (size_t)64 * S * sizeof(float) + group_ints(S, NB, cap) * sizeof(int);
}

// =============================================================================================
// Radial forward, run by the builder wave on the row it has just assembled (LDS mirror): the radial
// AEV needs nothing else, and a separate kernel would cost a launch boundary plus a re-read of the row.
// =============================================================================================
__device__ __forceinline__ void radial_forward_from_lds(const AniParams* __restrict__ P, const float4* stage, int cap,
                                                        int na, int nro, float* scratch, float* __restrict__ out) {
    const int lane = lane_id();
    const int S = P->S, nR = P->nR;
    float* nb_r = scratch;                   // [cap]
    float* nb_fc = nb_r + cap;               // [cap]
    int* nb_bin = (int*)(nb_fc + cap);       // [cap]  species * KP: where this neighbour's bins start inside a stream
    const int total = na + nro;
    const float inv_rcr = P->inv_rcr;
    const int kshift = P->kp_shift, KP = 1 << kshift;
    for (int e = lane; e < total; e += 64) {
        const float4 rec = e < na ? stage[e] : stage[cap - 1 - (e - na)];
        const float r = fast_sqrt(rec.x * rec.x + rec.y * rec.y + rec.z * rec.z);
        nb_r[e] = r;
        float sn_unused, cs;
        sincospi_unit(r * inv_rcr, sn_unused, cs);
        nb_fc[e] = 0.5f * cs + 0.5f;
        nb_bin[e] = (__float_as_int(rec.w) >> kTagShift) << kshift;
    }

    // lanes = (stream, k): KP = smallest power of two >= nR.  Every (stream, species, k) has a private LDS bin, so
    // the scatter by species is one plain read-modify-write per neighbour (LDS operations of a wave execute in
    // order: back-to-back hits on one bin are safe) instead of a compare/select per species in registers.  The next
    // neighbour is read before the bin of this one (LDS float atomics would need no read at all, but they run at a
    // fraction of the rate: 65 us instead of 26 for the kernel).
    // (Round 4: the scatter as a matrix product on the matrix core -- v_mfma_f32_16x16x4_f32, A = one-hot species of four
    //  neighbours, B = their terms, out[species][k] in four accumulator registers, no bins -- was built, passes the parity
    //  tests and measured SLOWER: builder 19.6 -> 21.7 us at 10 000 atoms, interleaved.  14 issues of 32 cycles per atom are
    //  1.8 us of matrix pipe for the whole frame, which the bins' LDS round trips, hidden behind the other waves, do not cost.)
    const int k = lane & (KP - 1), stream = lane >> kshift, nstreams = 64 >> kshift;
    const float ck = P->rad_c[min(k, nR - 1)], rs = P->rad_rs[min(k, nR - 1)];
    float* bins = (float*)(nb_bin + cap);              // [nstreams][S][KP] = 64 * S floats
    for (int s = 0; s < S; s++) bins[s * 64 + lane] = 0.f;
    wave_fence();
    float* mine = bins + stream * S * KP + k;
    int e = stream;
    float r = 0.f, fc = 0.f;
    int bin = 0;
    if (e < total) { r = nb_r[e]; fc = nb_fc[e]; bin = nb_bin[e]; }
    while (e < total) {
        const float sh = r - rs, w = fc;
        float* dst = mine + bin;
        e += nstreams;
        if (e < total) { r = nb_r[e]; fc = nb_fc[e]; bin = nb_bin[e]; }
        *dst += w * fast_exp2(ck * sh * sh);
    }
    wave_fence();
    const float scale = P->radial_scale;
    for (int q = lane; q < S * KP; q += 64) {              // q = species * KP + k: the bins of stream 0
        const int sp = q >> kshift, kk = q & (KP - 1);
        float v = 0.f;
        for (int st = 0; st < nstreams; st++) v += bins[st * S * KP + q];
        if (kk < nR) store_wt(out + sp * nR + kk, v * scale);
    }
}

// All-pairs scan (the reference's O(N^2) search, one wave per atom; used for small systems and for
// boxes too small for the cell stencil).  Row order = ascending atom id.
template <bool PERIODIC>
__global__ __launch_bounds__(64 * kWavesPerGroup) void ani_neighbors_allpairs(const AniParams* __restrict__ P,
                                                             const float* __restrict__ pos,
                                                             const float* __restrict__ box,
                                                             const int* __restrict__ species,
                                                             const int2* __restrict__ segment,   // per-atom [lo, hi) or NULL
                                                             float4* __restrict__ nbr,
                                                             int cap, int capA, float4* __restrict__ recA,
                                                             float4* __restrict__ recB, int* __restrict__ ids,
                                                             int* __restrict__ tri,
                                                             int* __restrict__ cnt_a, int* __restrict__ cnt_ro, int* __restrict__ cnt_pos, int* __restrict__ status,
                                                             float* __restrict__ radial, int ld_radial, int lds_per_wave, int w0, int nw) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    float4* stage = (float4*)(lds_raw + (size_t)wave_in_group() * lds_per_wave);
    float* rscratch = (float*)(stage + cap);
    const AtomGroups G = carve_groups((int*)(rscratch + 3 * cap + 64 * P->S), P->S, P->NB);
    if (wave_global_id() >= nw) return;
    const int i = w0 + wave_global_id();                   // this launch covers atoms [w0, w0 + nw)
    const int lane = lane_id();
    const int N = P->N;
    const float rcr2 = P->rcr2, rca2 = P->rca2;
    Box b{};
    if (PERIODIC) b = load_box(box);
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    float4* row = nbr + (size_t)i * cap;                  // (rows and cnt_pos go by POSITION in the walk of the radial backward: here the atom index)
    // batched molecules: an atom only sees the atoms of its own molecule (independent systems in one handle)
    const int lo = segment ? segment[i].x : 0, hi = segment ? segment[i].y : N;
    int na = 0, nro = 0;
    for (int base = lo; base < hi; base += 64) {
        const int j = base + lane;
        bool in_r = false, in_a = false;
        int word = 0;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (j < hi && j != i) {
            word = j | (species[j] << kTagShift);
            dx = pos[3 * j] - xi; dy = pos[3 * j + 1] - yi; dz = pos[3 * j + 2] - zi;
            min_image<PERIODIC>(dx, dy, dz, b);
            const float r2 = dx * dx + dy * dy + dz * dz;
            in_r = r2 < rcr2;
            in_a = in_r && (r2 < rca2);
        }
        append_to_row(cap, stage, in_a, in_r && !in_a, dx, dy, dz, word, na, nro);
    }
    if (lane == 0) {
        cnt_a[i] = na; cnt_ro[i] = nro;
        cnt_pos[i] = pack_cnt_pos(na, nro, species[i]);
        // (an atom that outgrew its row or its records says so itself: check() then needs no pass over the counts to
        //  know that nothing overflowed; an atomic only in that rare case)
        if (na > capA || na + nro > cap) atomicOr(&status[kStatOverflow], 1);
        else if (P->class_tile[i] != 255 && na > (int)P->class_tile[i]) atomicOr(&status[kStatOverflow], 8);     // outgrew its backward class: check() regroups
    }
    int n, nro_c;
    clamp_counts(na, nro, cap, capA, n, nro_c);
    wave_fence();
    flush_row(row, stage, cap, na, nro);
    radial_forward_from_lds(P, stage, cap, n, nro_c, rscratch, radial + (size_t)i * ld_radial);
    finalize_angular(P, stage, n, recA + (size_t)i * capA, recB + (size_t)i * capA, ids + (size_t)i * capA, capA,
                     tri + (size_t)i * triples_capacity(capA), P->bucket_offsets + (size_t)i * (P->NB + 1), G);
}

// Cell-grid search (celllist.h): one wave per atom walks the 3x3x3 stencil of its cell; candidates
// are read as coalesced float4 {x,y,z,(species<<24)|id} runs.  Row order = stencil order.
template <bool PERIODIC>
__global__ __launch_bounds__(64 * kWavesPerGroup) void ani_neighbors_cells(const AniParams* __restrict__ P,
                                                          const float* __restrict__ box,
                                                          const CellGrid* __restrict__ grid,
                                                          const int* __restrict__ cell_start,
                                                          const int* __restrict__ sorted_cell,
                                                          const float4* __restrict__ sorted_pos, float4* __restrict__ nbr,
                                                          int cap, int capA, float4* __restrict__ recA,
                                                          float4* __restrict__ recB, int* __restrict__ ids,
                                                          int* __restrict__ tri,
                                                          int* __restrict__ cnt_a, int* __restrict__ cnt_ro, int* __restrict__ cnt_pos,
                                                          int* __restrict__ status, float* __restrict__ radial,
                                                          int ld_radial, int lds_per_wave, int* __restrict__ cell_hist, int slot0, int nslots) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    float4* stage = (float4*)(lds_raw + (size_t)wave_in_group() * lds_per_wave);
    float* rscratch = (float*)(stage + cap);
    const AtomGroups G = carve_groups((int*)(rscratch + 3 * cap + 64 * P->S), P->S, P->NB);
    const int lane = lane_id();
    const int slot_id = slot0 + wave_global_id();          // position in cell order; this launch covers [slot0, slot0 + nslots)
    clear_cell_histogram(cell_hist);
    if (wave_global_id() >= nslots) return;
    const CellGrid g = *grid;
    if (!g.ok) {                                           // box too small for the stencil: tell the host
        if (lane == 0) {
            if (slot_id == 0) atomicOr(&status[kStatOverflow], g.bin_overflow ? 6 : 2);   // 4: grow the cell bins
            cnt_a[slot_id] = 0;                            // keep the consumers of this (void) build harmless
            cnt_ro[slot_id] = 0;
            cnt_pos[slot_id] = 0;
        }
        return;
    }
    const float rcr2 = P->rcr2, rca2 = P->rca2;
    Box b{};
    if (PERIODIC) b = load_box(box);
    const float4 me = sorted_pos[slot_id];
    const int c = sorted_cell[slot_id];                    // (written in sorted order by the grid build: no load that waits for the id)
    const int i = __float_as_int(me.w) & kIdMask;
    int cx, cy, cz;
    split_cell(g, c, cx, cy, cz);                          // (no integer division; exact: celllist.h)
    float4* row = nbr + (size_t)slot_id * cap;            // (by position in cell order: the radial backward walks the same order and needs no atom id to find it)
    int na = 0, nro = 0;
    const WideStencil st = gather_wide_stencil(g, cell_start, cx, cy, cz);
    int* strip = (int*)rscratch;                           // the radial scratch is idle during the scan
    int carry = 0;
    // This wave's time is the sum of its dependent round trips to memory (the kernel runs ~1.5 occupancy rounds of it),
    // so the candidates are requested four batches at a time -- the whole stencil of a half-cutoff grid in one trip.
    constexpr int GROUP = 4;
    for (int base = 0; base < st.total; base += 64 * GROUP) {
        float4 pj[GROUP];
#pragma unroll
        for (int b4 = 0; b4 < GROUP; b4++)
            if (base + 64 * b4 < st.total) pj[b4] = sorted_pos[wide_stencil_slot(st, base + 64 * b4, strip, carry)];
#pragma unroll
        for (int b4 = 0; b4 < GROUP; b4++) {
            if (base + 64 * b4 >= st.total) break;         // wave-uniform
            const float4 cur = pj[b4];
            const int word = __float_as_int(cur.w);
            float dx = cur.x - me.x, dy = cur.y - me.y, dz = cur.z - me.z;
            min_image<PERIODIC>(dx, dy, dz, b);
            const float r2 = dx * dx + dy * dy + dz * dz;
            const bool in_r = (base + 64 * b4 + lane < st.total) & ((word & kIdMask) != i) & (r2 < rcr2);
            const bool in_a = in_r & (r2 < rca2);
            append_to_row(cap, stage, in_a, in_r & !in_a, dx, dy, dz, word, na, nro);
        }
    }
    if (lane == 0) {
        cnt_a[i] = na; cnt_ro[i] = nro;
        cnt_pos[slot_id] = pack_cnt_pos(na, nro, __float_as_int(me.w) >> kTagShift);
        // (an atom that outgrew its row or its records says so itself: check() then needs no pass over the counts to
        //  know that nothing overflowed; an atomic only in that rare case)
        if (na > capA || na + nro > cap) atomicOr(&status[kStatOverflow], 1);
        else if (P->class_tile[i] != 255 && na > (int)P->class_tile[i]) atomicOr(&status[kStatOverflow], 8);     // outgrew its backward class: check() regroups
    }
    int n, nro_c;
    clamp_counts(na, nro, cap, capA, n, nro_c);
    wave_fence();
    flush_row(row, stage, cap, na, nro);
    radial_forward_from_lds(P, stage, cap, n, nro_c, rscratch, radial + (size_t)i * ld_radial);
    finalize_angular(P, stage, n, recA + (size_t)i * capA, recB + (size_t)i * capA, ids + (size_t)i * capA, capA,
                     tri + (size_t)i * triples_capacity(capA), P->bucket_offsets + (size_t)i * (P->NB + 1), G);
}

// =============================================================================================
// Angular kernels.
// =============================================================================================
// Geometry of one triple, shared by forward and backward.  A = {dx,dy,dz,r}, A2 = {fc,dfc,1/r,word}.
struct TripleGeom {
    float c, s;        // cos / sin of the (damped) angle
    float rbar;        // (r_ij + r_ik)/2
    float fcfc;
};

template <bool TORCHANI>
__device__ __forceinline__ TripleGeom triple_geometry(const float4& A, const float4& A2, const float4& B, const float4& B2) {
    TripleGeom g;
    const float dot = A.x * B.x + A.y * B.y + A.z * B.z;
    const float iprod = A2.z * B2.z;
    if (TORCHANI) {
        g.c = 0.95f * dot * iprod;                       // ref :391-393
        g.s = fast_sqrt(1.0f - g.c * g.c);               // |c| <= 0.95: no cancellation
    } else {
        g.c = fminf(fmaxf(dot * iprod, -1.0f), 1.0f);
        // sin from the cross product: accurate next to 0 and pi, which is what the reference's
        // asin branch (ref :396-404) is there for
        const float cx = A.y * B.z - A.z * B.y, cy = A.z * B.x - A.x * B.z, cz = A.x * B.y - A.y * B.x;
        g.s = fminf(fast_sqrt(cx * cx + cy * cy + cz * cz) * iprod, 1.0f);
    }
    g.rbar = 0.5f * (A.w + B.w);
    g.fcfc = A2.x * B2.x;
    return g;
}

// Copy the atom's sorted angular records into LDS (one coalesced access per array).
__device__ __forceinline__ void load_angular_records(const float4* __restrict__ recA_g, const float4* __restrict__ recB_g,
                                                     int n, float4* recA, float4* recB) {
    for (int e = lane_id(); e < n; e += 64) {
        const float4 a = recA_g[e], b = recB_g[e];
        recA[e] = a;
        recB[e] = b;
    }
}

}  // namespace nnpops
