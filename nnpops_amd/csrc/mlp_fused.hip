// mlp_fused.hip -- the ANI atomic networks of a frame as ONE forward launch and ONE input-gradient launch
// (C ABI: nnpops_mlp_pack, nnpops_mlp_forward, nnpops_mlp_input_grad).
//
// What it serves: TorchANIBatchedNN inside OptimizedTorchANI (reference src/pytorch/BatchedNN.py:37-122,
// OptimizedTorchANI.py:49-52): per atom and ensemble member  Linear(F,H1) CELU Linear(H1,H2) CELU Linear(H2,H3) CELU
// Linear(H3,1), summed over atoms, averaged over members -- and its gradient with respect to the AEV.
//
// Shape of the work: with the atoms grouped by species, a workgroup owns a TILE of
// 64 atoms of one species and ONE ensemble member and carries the tile through all four layers -- and straight back through
// the two small layers of the gradient -- without its activations ever leaving the CU:
//   * everything is computed TRANSPOSED, Y^T[features x atoms] = W[features x K] . X^T[K x atoms]: the weights are the A
//     operand of v_mfma_f32_16x16x32_f16 (rows = output features), the activations the B operand (columns = atoms).  A wave
//     (eight per workgroup, two per SIMD) owns an eighth of the output features (row blocks w, w + 8) for all 64 atoms, so its
//     weight fragments are its own:
//     they go from L2 straight into registers in the instruction's operand layout (the packed planes are stored fragment
//     by fragment, 1 KiB per wave load) and never touch LDS;
//   * the activations of a layer are the B operand of the next one: they cross the waves through LDS in B-FRAGMENT layout
//     ([K step][column block][plane][lane] x 16 B -- linear in the lane, so reads and writes are free of bank conflicts).
//     A lane's four accumulator rows of an even/odd row block are exactly the low/high half of a 16-byte fragment slot once
//     the K order inside a step is permuted to (rows 4g..4g+3 of the even block, then of the odd block); the weight planes
//     of every layer fed from an accumulator are packed with that permutation, so the hand-over is one 8-byte LDS store
//     per block and no shuffle;
//   * fp32 in, fp32 out; every operand is carried as two fp16 planes (x/16 = hi + 2^-11 lo', products exact, fp32
//     accumulation, three matrix instructions per block): the arithmetic of batched_nn.hip / cfconv.hip, measured there
//     to be at least as close to a double-precision product as a chain of fp32 FMAs;
//   * CELU'(.) of the three hidden layers stays in registers (160 per lane): the backward pass of layers 6, 4 and 2 runs in
//     the same launch and the kernel's only large output is dE/dy1 [atoms x members*H1], already split into B-fragment
//     planes for the second kernel;
//   * nnpops_mlp_input_grad is the one big product left, dE/dAEV^T[F x atoms] = W0^T[F x members*H1] . dE/dy1^T, tiled
//     64 atoms x 128 AEV columns per workgroup, and writes the gradient rows where the AEV kernels read them.
// All species of a frame run in ONE launch (a table of per-species descriptors travels in the kernel arguments), and
// workgroup b works for member b mod M: with 8 members every XCD's L2 holds the weights of exactly one member.
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "host_common.h"

using namespace nnpops;

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

constexpr int kWaves = 8;                   // waves per workgroup: two per SIMD, so that one computes while the other waits for its weights
constexpr int kThreads = 64 * kWaves;
constexpr int kRB = 2;                      // row blocks (16 output features each) per wave: layer widths up to 16 * kRB * kWaves = 256
constexpr int kCB = 4;                      // column blocks (16 atoms each) per workgroup: tiles of 64 atoms
constexpr int kTile = 16 * kCB;
constexpr int kFrag = 512;                  // halves per fragment plane: 64 lanes x 8
constexpr float kDefaultScaleLog2 = 4;     // operands are split after a scale of 2^-k (MlpArgs::scale): k = 4, |activation| up to 1e6, unless the frame says otherwise
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;
constexpr int kActBytes = 16 * kCB * 2 * 1024 / 2;        // one activation region: up to 8 K steps (256 features): 64 KiB
constexpr int kStageBytes = kCB * 2 * 1024;               // one K step of B fragments for the whole tile: 8 KiB

struct KindDesc {                           // one species: its atoms (a contiguous run of `rows`) and its networks
    int n, first;                           // atoms of the kind, position of the first one in `rows`
    int h1, h2, h3;                         // layer widths rounded up to 32 (zero padded)
    int tiles, block0;                      // 64-atom tiles, first workgroup of the kind
    const _Float16 *w0, *w2, *w4, *w4t, *w2t, *w0t;     // packed fragment planes, members one after the other
    const _Float16* w0tm;                   // W0^T member by member ([M][F/16 row blocks][h1/32 steps]); only with dx_partial
    const float *b0, *b2, *b4, *w6, *b6;    // [M][h1] [M][h2] [M][h3] [M][h3] [M]
    _Float16* d1;                           // [tiles][M][h1/32][kCB][2][kFrag]: dE/dy1 in B-fragment planes
};

struct MlpArgs {
    int num_kinds, F, M, ldx;
    const float* x;                         // [atoms][ldx]
    const int* rows;                        // atoms grouped by kind: row of x / dx of the r-th grouped atom
    float* energies;                        // [grouped atoms][M]
    float alpha;
    float* dx; int lddx;                    // input_grad only
    const float* upstream;                  // optional device scalar multiplying dx
    float dx_scale;                         // host scalar multiplying dx
    const int* x_groups;                    // optional: feature block f lives in columns 16 x_groups[f] .. of x / dx (NULL: f)
    const int* dead_groups; int num_dead;   // 16-column blocks of dx the gradient pass sets to zero
    float* dx_partial; int n_grouped;       // optional [M][n_grouped][F]: every member's W0^T dE/dy1, formed by the forward launch
    float mean_scale; float* mean_out; const double* mean_shift; double* mean_out_shifted;   // optional: the energy mean rides along (mlp_sum_members)
    const int* publish_word; int* publish_to; int publish_stamp;                             // optional: an ANI handle's deferred capacity check rides along (mlp_forward)
    float scale, unscale;                   // 2^-act_scale_log2 and its inverse: the operand scale of the fp16 planes (nnpops_hip.h; 1/16 by default)
    KindDesc kinds[NNPOPS_MLP_MAX_KINDS];
};

// v (already scaled into range) -> hi + 2^-11 lo'.  The high plane is rounded toward zero two values at a time
// (v_cvt_pkrtz_f16_f32: any fp16 near v will do, the low plane carries the exact remainder), the low plane to nearest.
__device__ __forceinline__ void split4(const f32x4& v, f16x4& h, f16x4& l) {
    typedef __fp16 pk2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        const pk2 hh = __builtin_amdgcn_cvt_pkrtz(v[i], v[i + 1]);
        h[i] = (_Float16)hh[0]; h[i + 1] = (_Float16)hh[1];
        l[i] = (_Float16)((v[i] - (float)hh[0]) * kLoScale);
        l[i + 1] = (_Float16)((v[i + 1] - (float)hh[1]) * kLoScale);
    }
}

struct AFrag { f16x8 h[kRB], l[kRB]; };

// fragments of K step s of this wave's row blocks (w, w + kWaves, ...); blocks past the layer are read from the last valid one
// (straight-line loads: a branch around a load makes the compiler drain the whole load queue where the paths meet)
__device__ __forceinline__ void load_a(AFrag& a, const _Float16* __restrict__ wp, int steps, int nb, int s, int w, int lane) {
#pragma unroll
    for (int j = 0; j < kRB; j++) {
        const int rb = min(w + kWaves * j, nb - 1);
        const _Float16* p = wp + ((size_t)(rb * steps + s) * 2) * kFrag + lane * 8;
        a.h[j] = *reinterpret_cast<const f16x8*>(p);
        a.l[j] = *reinterpret_cast<const f16x8*>(p + kFrag);
    }
}

__device__ __forceinline__ void mma_step(const AFrag& a, const char* __restrict__ bstage, int nmine, int lane, f32x4 (&acc1)[kRB][kCB],
                                         f32x4 (&acc2)[kRB][kCB]) {
    f16x8 bh[kCB], bl[kCB];
#pragma unroll
    for (int cb = 0; cb < kCB; cb++) {
        bh[cb] = *reinterpret_cast<const f16x8*>(bstage + (cb * 2) * 1024 + lane * 16);
        bl[cb] = *reinterpret_cast<const f16x8*>(bstage + (cb * 2 + 1) * 1024 + lane * 16);
    }
#pragma unroll
    for (int j = 0; j < kRB; j++) {
        if (j < nmine) {
#pragma unroll
            for (int cb = 0; cb < kCB; cb++) {
                acc1[j][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h[j], bh[cb], acc1[j][cb], 0, 0, 0);
                acc2[j][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h[j], bl[cb], acc2[j][cb], 0, 0, 0);
                acc2[j][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l[j], bh[cb], acc2[j][cb], 0, 0, 0);
            }
        }
    }
}

__device__ __forceinline__ void zero_acc(f32x4 (&acc1)[kRB][kCB], f32x4 (&acc2)[kRB][kCB]) {
#pragma unroll
    for (int j = 0; j < kRB; j++)
#pragma unroll
        for (int cb = 0; cb < kCB; cb++) { acc1[j][cb] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[j][cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
}

// A layer whose B operand is already resident in LDS (an activation region in fragment layout): no barrier inside.
// The weights are requested three K steps ahead (four register sets of 16: an L2 round trip is longer than a step).
__device__ __forceinline__ void layer_resident(const _Float16* __restrict__ wp, int steps, int nb, int nmine, int w, int lane,
                                               const char* __restrict__ act, f32x4 (&acc1)[kRB][kCB], f32x4 (&acc2)[kRB][kCB]) {
    zero_acc(acc1, acc2);
    AFrag a0, a1, a2, a3;
    load_a(a0, wp, steps, nb, 0, w, lane);
    load_a(a1, wp, steps, nb, min(1, steps - 1), w, lane);
    load_a(a2, wp, steps, nb, min(2, steps - 1), w, lane);
    auto step = [&](int s, const AFrag& cur, AFrag& refill) {
        load_a(refill, wp, steps, nb, min(s + 3, steps - 1), w, lane);
        mma_step(cur, act + (size_t)s * kStageBytes, nmine, lane, acc1, acc2);
    };
    for (int s = 0; s < steps; s += 4) {
        step(s, a0, a3);
        if (s + 1 < steps) step(s + 1, a1, a0);
        if (s + 2 < steps) step(s + 2, a2, a1);
        if (s + 3 < steps) step(s + 3, a3, a2);
    }
}

// accumulator of row block rb (rows 4g + q of lane (g = lane >> 4, atom = lane & 15)) -> fp32 value, then `f` decides
template <typename F>
__device__ __forceinline__ void for_blocks(int nmine, int w, F&& f) {
#pragma unroll
    for (int j = 0; j < kRB; j++)
        if (j < nmine) f(j, w + kWaves * j);
}

// hand a block of fp32 values (D layout) to the next layer: 8-byte half slot of its B fragment, both planes
__device__ __forceinline__ void put_fragment(char* __restrict__ act, int rb, int cb, int lane, const f32x4& v) {
    f16x4 h, l;
    split4(v, h, l);
    char* p = act + (size_t)(rb >> 1) * kStageBytes + (cb * 2) * 1024 + lane * 16 + 8 * (rb & 1);
    *reinterpret_cast<f16x4*>(p) = h;
    *reinterpret_cast<f16x4*>(p + 1024) = l;
}

__device__ __forceinline__ int blocks_of_wave(int nb, int w) { return nb > w ? (nb - w + kWaves - 1) / kWaves : 0; }

__device__ __forceinline__ const KindDesc& kind_of_block(const MlpArgs& g, int b) {
    int k = 0;
#pragma unroll
    for (int i = 1; i < NNPOPS_MLP_MAX_KINDS; i++)
        if (i < g.num_kinds && b >= g.kinds[i].block0) k = i;
    return g.kinds[k];
}

// =============================================================================================
// forward through all four layers (+ backward through layers 6, 4, 2 when GRAD)
// =============================================================================================
template <bool GRAD>
__global__ __launch_bounds__(kThreads, 2) void mlp_forward(const MlpArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* xstage = lds;                                     // 2 x 8 KiB: the AEV columns of a K step, split, fragment layout
    char* actA = lds + 2 * kStageBytes;                     // 64 KiB
    char* actB = actA + kActBytes;                          // 64 KiB
    int* xgroup = reinterpret_cast<int*>(actB + kActBytes); // 64 ints: column block of x behind every 16-feature block
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const KindDesc& kd = kind_of_block(g, blockIdx.x);
    const int local = blockIdx.x - kd.block0;
    const int m = local % g.M, tile = local / g.M;
    if (tid < 64) xgroup[tid] = 16 * tid < g.F ? (g.x_groups ? g.x_groups[tid] : tid) : 0;
    if (g.publish_to && blockIdx.x == 0 && tid == 64) {     // (nnpops_hip.h: publish_*; what ani_publish_status would need a launch for)
        const int word = __hip_atomic_load(g.publish_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&g.publish_to[1], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&g.publish_to[0], g.publish_stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    const int r0 = tile * kTile;                            // first atom of the tile inside the kind
    const int kgD = lane >> 4, a16 = lane & 15;
    const float inv_alpha = 1.0f / g.alpha;

    const int nb1 = kd.h1 >> 4, nb2 = kd.h2 >> 4, nb3 = kd.h3 >> 4;
    const int s0 = (g.F + 31) >> 5, s1 = kd.h1 >> 5, s2 = kd.h2 >> 5, s3 = kd.h3 >> 5;
    const int n1 = blocks_of_wave(nb1, w), n2 = blocks_of_wave(nb2, w), n3 = blocks_of_wave(nb3, w);

    f32x4 acc1[kRB][kCB], acc2[kRB][kCB];
    f32x4 c1[kRB][kCB], c2[kRB][kCB];                      // CELU' of hidden layers 1 and 2 (GRAD; layer 3's is used on the spot)

    // ---------------- layer 0: the AEV rows of the tile stream through LDS, 32 columns per step ----------------
    {
        const _Float16* wp = kd.w0 + (size_t)m * nb1 * s0 * 2 * kFrag;
        // staging role: thread -> (atom a = tid / 8, piece = tid % 8): four consecutive columns, one 8-byte half of a fragment slot
        const int sa = tid >> 3, piece = tid & 7;
        const int srow = g.rows[kd.first + min(r0 + sa, kd.n - 1)];
        const float* xsrc = g.x + (size_t)srow * g.ldx + (piece & 3) * 4;
        char* xdst = xstage + ((sa >> 4) * 2) * 1024 + ((sa & 15) + 16 * (piece >> 1)) * 16 + 8 * (piece & 1);
        // (the loaded columns are NOT touched before they are staged one step later: an instruction that consumes them right
        //  after the load makes the compiler wait for EVERY outstanding load there -- s_waitcnt vmcnt(0) after each barrier,
        //  which drained the weight fragments requested three steps ahead with it)
        float4 xraw;
        float xscale;
        auto fetch_x = [&](int s) {
            const int k = 32 * s + piece * 4;
            const bool in = k < g.F;                         // (F is a multiple of 8: all four or none)
            xraw = *reinterpret_cast<const float4*>(xsrc + 16 * xgroup[in ? 2 * s + (piece >> 2) : 0]);
            xscale = in ? g.scale : 0.0f;
        };
        auto stage_x = [&](int stage) {
            const f32x4 xv = {xraw.x * xscale, xraw.y * xscale, xraw.z * xscale, xraw.w * xscale};
            f16x4 h, l;
            split4(xv, h, l);
            *reinterpret_cast<f16x4*>(xdst + stage * kStageBytes) = h;
            *reinterpret_cast<f16x4*>(xdst + stage * kStageBytes + 1024) = l;
        };
        zero_acc(acc1, acc2);
        AFrag a0, a1, a2, a3;                                // weights three steps ahead, the AEV columns two (one in LDS, one in registers)
        fetch_x(0);
        load_a(a0, wp, s0, nb1, 0, w, lane);
        load_a(a1, wp, s0, nb1, min(1, s0 - 1), w, lane);
        load_a(a2, wp, s0, nb1, min(2, s0 - 1), w, lane);
        stage_x(0);
        fetch_x(min(1, s0 - 1));
        __syncthreads();
        auto step = [&](int s, const AFrag& cur, AFrag& refill) {
            load_a(refill, wp, s0, nb1, min(s + 3, s0 - 1), w, lane);
            mma_step(cur, xstage + (s & 1) * kStageBytes, n1, lane, acc1, acc2);
            stage_x((s + 1) & 1);                            // step s + 1 (fetched one iteration ago)
            fetch_x(min(s + 2, s0 - 1));
            __syncthreads();
        };
        for (int s = 0; s < s0; s += 4) {
            step(s, a0, a3);
            if (s + 1 < s0) step(s + 1, a1, a0);
            if (s + 2 < s0) step(s + 2, a2, a1);
            if (s + 3 < s0) step(s + 3, a3, a2);
        }
    }
    // bias + CELU; y1 -> region A.  Everything is kept divided by 16 (the operand scale): vs = v / 16 comes straight out of the
    // accumulators (acc1 + acc2 / 2048 + b / 16), CELU(v) / 16 = vs > 0 ? vs : (alpha / 16) (exp(16 vs / alpha) - 1).
    const float kScale = g.scale, kUnscale = g.unscale;     // (run-time powers of two since round 5; the names of the constants they replace)
    const float exp_scale = kUnscale * inv_alpha * 1.44269504089f, alpha_s = g.alpha * kScale;
    auto activate = [&](const float* __restrict__ bias, int nmine, f32x4 (&c)[kRB][kCB], char* act) {
        for_blocks(nmine, w, [&](int j, int rb) {
            const float4 b = *reinterpret_cast<const float4*>(bias + rb * 16 + kgD * 4);
            const float bq[4] = {b.x * kScale, b.y * kScale, b.z * kScale, b.w * kScale};
#pragma unroll
            for (int cb = 0; cb < kCB; cb++) {
                f32x4 y;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float vs = fmaf(kLoInv, acc2[j][cb][q], acc1[j][cb][q]) + bq[q];
                    const float e = __builtin_amdgcn_exp2f(fminf(vs, 0.f) * exp_scale);       // CELU'(v) for v <= 0
                    y[q] = vs > 0.f ? vs : fmaf(alpha_s, e, -alpha_s);                        // CELU / 16 (BatchedNN.py:103-109)
                    if (GRAD) c[j][cb][q] = vs > 0.f ? 1.0f : e;
                }
                put_fragment(act, rb, cb, lane, y);
            }
        });
    };
    activate(kd.b0 + (size_t)m * kd.h1, n1, c1, actA);
    __syncthreads();
    // ---------------- layer 2 ----------------
    layer_resident(kd.w2 + (size_t)m * nb2 * s1 * 2 * kFrag, s1, nb2, n2, w, lane, actA, acc1, acc2);
    activate(kd.b2 + (size_t)m * kd.h2, n2, c2, actB);
    __syncthreads();
    // ---------------- layer 4 ----------------
    layer_resident(kd.w4 + (size_t)m * nb3 * s2 * 2 * kFrag, s2, nb3, n3, w, lane, actB, acc1, acc2);
    // bias + CELU, then layer 6 on the spot: e[atom] = sum_f w6[f] y3[f][atom] + b6
    float part[kCB] = {0.f, 0.f, 0.f, 0.f};
    {
        const float* bias = kd.b4 + (size_t)m * kd.h3;
        const float* w6 = kd.w6 + (size_t)m * kd.h3;
        for_blocks(n3, w, [&](int j, int rb) {
            const float4 b = *reinterpret_cast<const float4*>(bias + rb * 16 + kgD * 4);
            const float4 wl = *reinterpret_cast<const float4*>(w6 + rb * 16 + kgD * 4);
            const float bq[4] = {b.x * kScale, b.y * kScale, b.z * kScale, b.w * kScale};
            const float wq[4] = {wl.x * kUnscale, wl.y * kUnscale, wl.z * kUnscale, wl.w * kUnscale};      // (times the scaled y3)
            const float ws[4] = {wl.x * kScale, wl.y * kScale, wl.z * kScale, wl.w * kScale};              // d3 / 16
#pragma unroll
            for (int cb = 0; cb < kCB; cb++) {
                f32x4 d;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float vs = fmaf(kLoInv, acc2[j][cb][q], acc1[j][cb][q]) + bq[q];
                    const float e = __builtin_amdgcn_exp2f(fminf(vs, 0.f) * exp_scale);
                    const float ys = vs > 0.f ? vs : fmaf(alpha_s, e, -alpha_s);
                    part[cb] = fmaf(wq[q], ys, part[cb]);
                    d[q] = vs > 0.f ? ws[q] : ws[q] * e;                    // dE/d(pre-activation 3) / 16 = w6 CELU' / 16
                }
                if (GRAD) put_fragment(actA, rb, cb, lane, d);              // (region A: layer 2 is done with it)
            }
        });
    }
    // energies: sum over the K groups of a wave (lanes 16 apart), then over the waves
    float* red = reinterpret_cast<float*>(xstage);          // kWaves x 64 atoms (the x stages are idle now)
#pragma unroll
    for (int cb = 0; cb < kCB; cb++) {
        float p = part[cb];
        p += __shfl_xor(p, 16, 64);
        p += __shfl_xor(p, 32, 64);
        if (kgD == 0) red[w * kTile + cb * 16 + a16] = p;
    }
    __syncthreads();
    if (tid < kTile && r0 + tid < kd.n)
    {
        float e = kd.b6[m];
#pragma unroll
        for (int v = 0; v < kWaves; v++) e += red[v * kTile + tid];
        g.energies[(size_t)(kd.first + r0 + tid) * g.M + m] = e;
    }
    if (!GRAD) return;
    // ---------------- backward: d2 = (W4^T d3) * CELU'(2), d1 = (W2^T d2) * CELU'(1) ----------------
    layer_resident(kd.w4t + (size_t)m * nb2 * s3 * 2 * kFrag, s3, nb2, n2, w, lane, actA, acc1, acc2);
    for_blocks(n2, w, [&](int j, int rb) {
#pragma unroll
        for (int cb = 0; cb < kCB; cb++) {
            f32x4 d;
#pragma unroll
            for (int q = 0; q < 4; q++) d[q] = fmaf(kLoInv, acc2[j][cb][q], acc1[j][cb][q]) * c2[j][cb][q];      // d2 / 16
            put_fragment(actB, rb, cb, lane, d);
        }
    });
    __syncthreads();
    layer_resident(kd.w2t + (size_t)m * nb1 * s2 * 2 * kFrag, s2, nb1, n1, w, lane, actB, acc1, acc2);
    if (g.dx_partial) {
        // Few input columns (the AEV blocks of the species the molecule has: x_groups): this member's share of dE/dx,
        // W0_m^T d1, is one more small product on the spot -- d1 goes to region A like d2 went to region B -- instead of a
        // round trip of d1 through memory and a launch that streams W0^T of every member past every tile.
        for_blocks(n1, w, [&](int j, int rb) {
#pragma unroll
            for (int cb = 0; cb < kCB; cb++) {
                f32x4 d;
#pragma unroll
                for (int q = 0; q < 4; q++) d[q] = fmaf(kLoInv, acc2[j][cb][q], acc1[j][cb][q]) * c1[j][cb][q];      // d1 / 16
                put_fragment(actA, rb, cb, lane, d);
            }
        });
        __syncthreads();
        const int nbf = (g.F + 15) >> 4, nf = blocks_of_wave(nbf, w);
        layer_resident(kd.w0tm + (size_t)m * nbf * s1 * 2 * kFrag, s1, nbf, nf, w, lane, actA, acc1, acc2);
        float* pdst = g.dx_partial + ((size_t)m * g.n_grouped + kd.first + r0) * g.F;
        for_blocks(nf, w, [&](int j, int rb) {
            const int col = rb * 16 + kgD * 4;
            if (col >= g.F) return;
#pragma unroll
            for (int cb = 0; cb < kCB; cb++) {
                const int r = cb * 16 + a16;
                if (r0 + r >= kd.n) continue;
                float4 v;
                v.x = fmaf(kLoInv, acc2[j][cb][0], acc1[j][cb][0]); v.y = fmaf(kLoInv, acc2[j][cb][1], acc1[j][cb][1]);
                v.z = fmaf(kLoInv, acc2[j][cb][2], acc1[j][cb][2]); v.w = fmaf(kLoInv, acc2[j][cb][3], acc1[j][cb][3]);
                *reinterpret_cast<float4*>(pdst + (size_t)r * g.F + col) = v;
            }
        });
        return;
    }
    _Float16* d1 = kd.d1 + ((size_t)(tile * g.M + m) * s1) * (kCB * 2 * kFrag);
    for_blocks(n1, w, [&](int j, int rb) {
#pragma unroll
        for (int cb = 0; cb < kCB; cb++) {
            f32x4 d;
#pragma unroll
            for (int q = 0; q < 4; q++) d[q] = fmaf(kLoInv, acc2[j][cb][q], acc1[j][cb][q]) * c1[j][cb][q];      // d1 / 16
            f16x4 h, l;
            split4(d, h, l);
            _Float16* p = d1 + ((size_t)(rb >> 1) * kCB + cb) * 2 * kFrag + lane * 8 + 4 * (rb & 1);
            *reinterpret_cast<f16x4*>(p) = h;
            *reinterpret_cast<f16x4*>(p + kFrag) = l;
        }
    });
}

// =============================================================================================
// dE/dAEV^T [F x atoms] = W0^T [F x M*H1] . dE/dy1^T: workgroup = (tile of 64 atoms, 8 row blocks = 128 AEV columns)
// =============================================================================================
constexpr int kGradBlocks = kWaves;         // row blocks of F per workgroup: one per wave

__global__ __launch_bounds__(kThreads, 2) void mlp_input_grad(const MlpArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];     // 2 stages x 8 KiB of B fragments
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const KindDesc& kd = kind_of_block(g, blockIdx.x);
    const int local = blockIdx.x - kd.block0;
    const int nbf = (g.F + 15) >> 4, chunks = (nbf + kGradBlocks - 1) / kGradBlocks;
    const int chunk = local % chunks, tile = local / chunks;
    const int r0 = tile * kTile;
    const int s1 = kd.h1 >> 5, steps = g.M * s1;
    const int rb = chunk * kGradBlocks + w;                 // this wave's row block of F
    const bool mine = rb < nbf;
    const _Float16* bsrc = kd.d1 + (size_t)tile * steps * (kCB * 2 * kFrag) + tid * 8;     // 512 threads x 16 B = 8 KiB per step
    const _Float16* wp = kd.w0t + ((size_t)min(rb, nbf - 1) * steps * 2) * kFrag + lane * 8;

    f32x4 acc1[kCB], acc2[kCB];
#pragma unroll
    for (int cb = 0; cb < kCB; cb++) { acc1[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    struct A1 { f16x8 h, l; };
    auto load_a1 = [&](A1& a, int s) {
        const _Float16* p = wp + (size_t)s * 2 * kFrag;
        a.h = *reinterpret_cast<const f16x8*>(p);
        a.l = *reinterpret_cast<const f16x8*>(p + kFrag);
    };
    f16x8 bv;
    auto fetch_b = [&](int s) { bv = *reinterpret_cast<const f16x8*>(bsrc + (size_t)s * (kCB * 2 * kFrag)); };
    auto stage_b = [&](int stage) { *reinterpret_cast<f16x8*>(lds + stage * kStageBytes + tid * 16) = bv; };
    auto mma = [&](const A1& a, const char* bstage) {
        f16x8 bh[kCB], bl[kCB];
#pragma unroll
        for (int cb = 0; cb < kCB; cb++) {
            bh[cb] = *reinterpret_cast<const f16x8*>(bstage + (cb * 2) * 1024 + lane * 16);
            bl[cb] = *reinterpret_cast<const f16x8*>(bstage + (cb * 2 + 1) * 1024 + lane * 16);
        }
        if (mine) {
#pragma unroll
            for (int cb = 0; cb < kCB; cb++) {
                acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, bh[cb], acc1[cb], 0, 0, 0);
                acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, bl[cb], acc2[cb], 0, 0, 0);
                acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l, bh[cb], acc2[cb], 0, 0, 0);
            }
        }
    };
    // weights four steps ahead (four register sets, 8 registers each), B fragments two steps ahead through two LDS stages
    A1 a0, a1, a2, a3;
    fetch_b(0);
    load_a1(a0, 0); load_a1(a1, min(1, steps - 1)); load_a1(a2, min(2, steps - 1));
    stage_b(0);
    fetch_b(min(1, steps - 1));
    __syncthreads();
    auto step = [&](int s, const A1& cur, A1& refill) {
        load_a1(refill, min(s + 3, steps - 1));
        mma(cur, lds + (s & 1) * kStageBytes);
        stage_b((s + 1) & 1);
        fetch_b(min(s + 2, steps - 1));
        __syncthreads();
    };
    for (int s = 0; s < steps; s += 4) {
        step(s, a0, a3);
        if (s + 1 < steps) step(s + 1, a1, a0);
        if (s + 2 < steps) step(s + 2, a2, a1);
        if (s + 3 < steps) step(s + 3, a3, a2);
    }
    const float up = (g.upstream ? *g.upstream : 1.0f) * g.dx_scale;
    const int kgD = lane >> 4, a16 = lane & 15;
    if (chunk == 0 && g.num_dead > 0) {                      // column blocks no network reads: their gradient is zero
        const int per_row = g.num_dead * 4;
        for (int q = tid; q < kTile * per_row; q += kThreads) {
            const int r = r0 + q / per_row, e = q % per_row;
            if (r < kd.n)
                *reinterpret_cast<float4*>(g.dx + (size_t)g.rows[kd.first + r] * g.lddx + 16 * g.dead_groups[e >> 2] + 4 * (e & 3)) =
                    make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const int col = 16 * (g.x_groups ? g.x_groups[min(rb, nbf - 1)] : rb) + kgD * 4;
    if (mine && rb * 16 + kgD * 4 < g.F) {                   // (F is a multiple of 4)
#pragma unroll
        for (int cb = 0; cb < kCB; cb++) {
            const int r = r0 + cb * 16 + a16;
            if (r >= kd.n) continue;
            float4 v;
            v.x = (acc1[cb][0] + kLoInv * acc2[cb][0]) * g.unscale * up;
            v.y = (acc1[cb][1] + kLoInv * acc2[cb][1]) * g.unscale * up;
            v.z = (acc1[cb][2] + kLoInv * acc2[cb][2]) * g.unscale * up;
            v.w = (acc1[cb][3] + kLoInv * acc2[cb][3]) * g.unscale * up;
            *reinterpret_cast<float4*>(g.dx + (size_t)g.rows[kd.first + r] * g.lddx + col) = v;
        }
    }
}

// (value per thread: a sum over the elements i = tid, tid + 1024, ... whatever the width of the workgroup that calls it -- a
//  narrower one walks several of those strides -- so the result does not depend on which kernel took the mean)
template <int THREADS>
__device__ __forceinline__ void energy_mean_block(const float* __restrict__ v, long n, float scale, float* __restrict__ out,
                                                  const double* __restrict__ shift, double* __restrict__ out_shifted, double* red) {
    static_assert(1024 % THREADS == 0 && THREADS % 64 == 0, "");
    constexpr int SUBS = 1024 / THREADS, AHEAD = 8;
    double acc[SUBS];
#pragma unroll
    for (int q = 0; q < SUBS; q++) acc[q] = 0.0;
    for (long base = 0; base < n; base += 1024L * AHEAD) {             // (loads of a round issued together, added in index order)
        float t[SUBS][AHEAD];
#pragma unroll
        for (int q = 0; q < SUBS; q++)
#pragma unroll
            for (int k = 0; k < AHEAD; k++) {
                const long i = base + 1024L * k + q * THREADS + threadIdx.x;
                const float x = v[min(i, n - 1)];                       // (a load behind a branch would be waited for on the spot)
                t[q][k] = i < n ? x : 0.0f;
            }
#pragma unroll
        for (int q = 0; q < SUBS; q++)
#pragma unroll
            for (int k = 0; k < AHEAD; k++) acc[q] += (double)t[q][k];
    }
#pragma unroll
    for (int q = 0; q < SUBS; q++) {                                   // the 1024 strided partial sums, THREADS at a time
        double a = acc[q];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
        const int sub = q * THREADS + threadIdx.x;
        if ((sub & 63) == 0) red[sub >> 6] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double e = 0.0;
        for (int w = 0; w < 1024 / 64; w++) e += red[w];
        const float mean = (float)(e * (double)scale);
        if (shift) out_shifted[0] = (double)mean + shift[0];
        else out[0] = mean;
    }
}

// dx = scale * sum over the members of dx_partial (what mlp_forward left when the frame carries dx_partial), written to the
// columns the feature blocks live in; dead column blocks are set to zero.  One thread per (grouped atom, four columns).
__global__ __launch_bounds__(256) void mlp_sum_members(const MlpArgs g) {
    __shared__ double red[1024 / 64];
    if (blockIdx.x == 0 && (g.mean_out || g.mean_out_shifted))                  // (the energy mean rides along: one launch fewer)
        energy_mean_block<256>(g.energies, (long)g.n_grouped * g.M, g.mean_scale, g.mean_out, g.mean_shift, g.mean_out_shifted, red);
    const int quads = g.F >> 2, per_row = quads + g.num_dead * 4;
    const long total = (long)g.n_grouped * per_row;
    const float up = (g.upstream ? *g.upstream : 1.0f) * g.dx_scale * g.unscale;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int r = (int)(t / per_row), e = (int)(t % per_row);
        float* row = g.dx + (size_t)g.rows[r] * g.lddx;
        if (e < quads) {
            const int c = 4 * e;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int m = 0; m < g.M; m++) {                  // fixed order: bitwise reproducible
                const float4 v = *reinterpret_cast<const float4*>(g.dx_partial + ((size_t)m * g.n_grouped + r) * g.F + c);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            const int grp = g.x_groups ? g.x_groups[c >> 4] : (c >> 4);
            *reinterpret_cast<float4*>(row + 16 * grp + (c & 15)) = make_float4(acc.x * up, acc.y * up, acc.z * up, acc.w * up);
        } else {
            const int d = e - quads;
            *reinterpret_cast<float4*>(row + 16 * g.dead_groups[d >> 2] + 4 * (d & 3)) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// =============================================================================================
// packing: W [rows][cols] fp32 (or its transpose) -> A-fragment planes [row block][K step][plane][lane][8 halves]
// =============================================================================================
// element (rb, s, plane, lane = r16 + 16 kg, i) = split(W[16 rb + r16][k]),  k = 32 s + 8 kg + i in natural K order, or
// with the accumulator permutation 32 s + (i < 4 ? 4 kg + i : 16 + 4 kg + i - 4).  No scale: only activations are scaled.
__global__ __launch_bounds__(256) void mlp_pack(int rows, int cols, const float* __restrict__ W, long ldw, int transpose, int permute,
                                                _Float16* __restrict__ out) {
    const int nb = (rows + 15) >> 4, steps = (cols + 31) >> 5;
    const long total = (long)nb * steps * kFrag;             // one thread per (rb, s, lane, i): writes both planes
    for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int i = (int)(t & 7), lane = (int)((t >> 3) & 63);
        const long f = t >> 9;
        const int s = (int)(f % steps), rb = (int)(f / steps);
        const int r16 = lane & 15, kg = lane >> 4;
        const int row = rb * 16 + r16;
        const int k = 32 * s + (permute ? (i < 4 ? 4 * kg + i : 16 + 4 * kg + i - 4) : 8 * kg + i);
        float v = 0.f;
        if (row < rows && k < cols) v = transpose ? W[(size_t)k * ldw + row] : W[(size_t)row * ldw + k];
        const _Float16 h = (_Float16)v;
        _Float16* p = out + ((size_t)f * 2) * kFrag + lane * 8 + i;
        p[0] = h;
        p[kFrag] = (_Float16)((v - (float)h) * kLoScale);
    }
}

// out[0] = scale * sum of v[0 .. n): one workgroup, double accumulation, fixed order (bitwise reproducible)
// (shift != NULL: out is a double and gets (double)(float)mean + shift[0] -- the float energy promoted and shifted exactly as
//  `energies + self_energies` does it in the reference's EnergyShifter.py:52)
__global__ __launch_bounds__(1024) void mlp_energy_mean(const float* __restrict__ v, long n, float scale, float* __restrict__ out,
                                                        const double* __restrict__ shift, double* __restrict__ out_shifted) {
    __shared__ double red[1024 / 64];
    energy_mean_block<1024>(v, n, scale, out, shift, out_shifted, red);
}

// out[i] = in[i] * (float)factor[0], the factor a device scalar of either precision (the chain-rule factor autograd hands to the
// backward of the energy node: a double once the energy has been shifted)
template <typename F>
__global__ __launch_bounds__(256) void scale_by_scalar(const float* __restrict__ in, long n, const F* __restrict__ factor, float* __restrict__ out) {
    const float f = (float)factor[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = in[i] * f;
}

int check_and_fill(const nnpops_mlp_frame* fr, MlpArgs& g, bool grad, int blocks_per_tile_grad, int* total_blocks) {
    NNPOPS_REQUIRE(fr != nullptr, "NULL frame descriptor");
    NNPOPS_REQUIRE(fr->num_kinds >= 1 && fr->num_kinds <= NNPOPS_MLP_MAX_KINDS, "1..%d kinds per launch (got %d)", NNPOPS_MLP_MAX_KINDS, fr->num_kinds);
    NNPOPS_REQUIRE(fr->num_features > 0 && fr->num_features % 8 == 0, "the input width must be a positive multiple of 8 (got %d)", fr->num_features);
    // (mlp_forward keeps the column block behind every 16-feature block of x in a 64-entry LDS table: 64 * 16 features)
    NNPOPS_REQUIRE(fr->num_features <= 1024, "the fused networks take at most 1024 input features (got %d): use the per-layer GEMM layout",
                   fr->num_features);
    NNPOPS_REQUIRE(fr->num_members >= 1, "at least one ensemble member");
    NNPOPS_REQUIRE(fr->x && fr->rows && fr->energies, "NULL device pointer");
    NNPOPS_REQUIRE(fr->ldx >= fr->num_features && fr->ldx % 4 == 0 && ((uintptr_t)fr->x % 16) == 0, "x rows must be 16-byte aligned (ldx %d)", fr->ldx);
    NNPOPS_REQUIRE(fr->alpha > 0, "alpha must be positive");
    g.num_kinds = fr->num_kinds; g.F = fr->num_features; g.M = fr->num_members; g.ldx = fr->ldx;
    g.x = fr->x; g.rows = fr->rows; g.energies = fr->energies; g.alpha = fr->alpha;
    g.dx = fr->dx; g.lddx = fr->lddx; g.upstream = fr->upstream; g.dx_scale = fr->dx_scale == 0.f ? 1.0f : fr->dx_scale;
    g.x_groups = fr->x_groups; g.dead_groups = fr->dead_groups; g.num_dead = fr->dead_groups ? fr->num_dead_groups : 0;
    g.dx_partial = grad ? fr->dx_partial : nullptr;
    g.mean_scale = fr->mean_scale; g.mean_out = fr->mean_out; g.mean_shift = fr->mean_shift; g.mean_out_shifted = fr->mean_out_shifted;
    g.publish_word = fr->publish_word; g.publish_to = fr->publish_word ? fr->publish_to : nullptr; g.publish_stamp = fr->publish_stamp;
    {
        const int k = fr->act_scale_log2 == 0 ? (int)kDefaultScaleLog2 : fr->act_scale_log2;
        NNPOPS_REQUIRE(k >= 4 && k <= 12, "act_scale_log2 must be 0 (= 4) or in 4..12 (got %d)", fr->act_scale_log2);
        g.scale = 1.0f / (float)(1 << k); g.unscale = (float)(1 << k);
    }
    NNPOPS_REQUIRE(!fr->publish_word || fr->publish_to, "publish_word without publish_to");
    NNPOPS_REQUIRE(!(fr->mean_out && fr->mean_out_shifted) && (!fr->mean_out_shifted || fr->mean_shift), "one energy mean: float, or shifted double with its shift");
    NNPOPS_REQUIRE(!fr->x_groups || fr->num_features % 16 == 0, "x_groups maps blocks of 16 features: the feature count must be a multiple of 16 (got %d)",
                   fr->num_features);
    NNPOPS_REQUIRE(fr->num_dead_groups >= 0 && (fr->num_dead_groups == 0 || fr->dead_groups), "dead_groups is NULL");
    NNPOPS_REQUIRE(!g.dx_partial || (fr->num_features <= 256 && fr->num_features % 16 == 0),
                   "dx_partial (the input gradient formed by the forward launch) takes at most 256 features in blocks of 16 (got %d)", fr->num_features);
    int blocks = 0, first = 0;
    for (int k = 0; k < fr->num_kinds; k++) {
        const nnpops_mlp_kind& s = fr->kinds[k];
        KindDesc& d = g.kinds[k];
        NNPOPS_REQUIRE(s.num_atoms >= 0, "negative atom count");
        for (int h : {s.h1, s.h2, s.h3})
            NNPOPS_REQUIRE(h >= 32 && h <= 256 && h % 32 == 0, "packed layer widths must be multiples of 32 in 32..256 (got %d)", h);
        NNPOPS_REQUIRE(s.w0 && s.w2 && s.w4 && s.b0 && s.b2 && s.b4 && s.w6 && s.b6, "NULL weight pointer (kind %d)", k);
        NNPOPS_REQUIRE(!grad || (s.w4t && s.w2t && (fr->dx_partial || (s.w0t && s.d1))), "gradient pass needs the transposed planes and the d1 workspace (kind %d)", k);
        d.n = s.num_atoms; d.first = first; d.h1 = s.h1; d.h2 = s.h2; d.h3 = s.h3;
        d.tiles = (s.num_atoms + kTile - 1) / kTile;
        d.block0 = blocks;
        d.w0 = (const _Float16*)s.w0; d.w2 = (const _Float16*)s.w2; d.w4 = (const _Float16*)s.w4;
        d.w4t = (const _Float16*)s.w4t; d.w2t = (const _Float16*)s.w2t; d.w0t = (const _Float16*)s.w0t;
        d.b0 = s.b0; d.b2 = s.b2; d.b4 = s.b4; d.w6 = s.w6; d.b6 = s.b6; d.d1 = (_Float16*)s.d1;
        d.w0tm = (const _Float16*)s.w0tm;
        NNPOPS_REQUIRE(!g.dx_partial || s.w0tm, "dx_partial needs the member-by-member planes w0tm (kind %d)", k);
        blocks += d.tiles * blocks_per_tile_grad;
        first += s.num_atoms;
    }
    for (int k = fr->num_kinds; k < NNPOPS_MLP_MAX_KINDS; k++) g.kinds[k] = g.kinds[0];
    g.n_grouped = first;
    *total_blocks = blocks;
    return NNPOPS_OK;
}

}  // namespace

extern "C" {

int64_t nnpops_mlp_packed_halves(int rows, int cols) {
    return (int64_t)((rows + 15) / 16) * ((cols + 31) / 32) * 2 * kFrag;
}

int64_t nnpops_mlp_d1_halves(int num_atoms, int num_members, int h1) {
    return (int64_t)((num_atoms + kTile - 1) / kTile) * num_members * (h1 / 32) * kCB * 2 * kFrag;
}

int nnpops_mlp_pack(void* stream, int rows, int cols, const float* w, long ldw, int transpose, int permute, void* out) {
    NNPOPS_REQUIRE(w && out, "NULL device pointer");
    NNPOPS_REQUIRE(rows > 0 && cols > 0 && ldw >= (transpose ? rows : cols), "bad matrix shape");
    const long total = (long)((rows + 15) / 16) * ((cols + 31) / 32) * kFrag;
    const int blocks = (int)std::min<long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(mlp_pack, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rows, cols, w, ldw, transpose, permute, (_Float16*)out);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

int nnpops_mlp_forward(void* stream, const nnpops_mlp_frame* frame, int with_gradient) {
    MlpArgs g{};
    int blocks = 0;
    int rc = check_and_fill(frame, g, with_gradient != 0, frame ? frame->num_members : 1, &blocks);
    if (rc != NNPOPS_OK) return rc;
    if (blocks == 0) return NNPOPS_OK;
    const size_t lds = 2 * kStageBytes + 2 * kActBytes + 256;    // 144 KiB (+ the column-block table): above the default limit of dynamic LDS, raised per device
    {   // (once per device and kernel: the attribute is sticky, and the call is not free on the launch path)
        static bool raised[2][64] = {};
        int dev = 0;
        NNPOPS_HIP_TRY(hipGetDevice(&dev));
        bool& done = raised[with_gradient ? 1 : 0][dev & 63];
        if (!done || dev >= 64) {
            if (with_gradient) NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)mlp_forward<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            else NNPOPS_HIP_TRY(hipFuncSetAttribute((const void*)mlp_forward<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            done = true;
        }
    }
    if (with_gradient) hipLaunchKernelGGL(mlp_forward<true>, dim3(blocks), dim3(kThreads), lds, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(mlp_forward<false>, dim3(blocks), dim3(kThreads), lds, (hipStream_t)stream, g);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

int nnpops_mlp_energy_mean(void* stream, const float* energies, int64_t count, float scale, float* out) {
    NNPOPS_REQUIRE(energies && out && count > 0, "NULL device pointer or empty sum");
    hipLaunchKernelGGL(mlp_energy_mean, dim3(1), dim3(1024), 0, (hipStream_t)stream, energies, (long)count, scale, out,
                       (const double*)nullptr, (double*)nullptr);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

int nnpops_mlp_energy_mean_shifted(void* stream, const float* energies, int64_t count, float scale, const double* shift, double* out) {
    NNPOPS_REQUIRE(energies && out && shift && count > 0, "NULL device pointer or empty sum");
    hipLaunchKernelGGL(mlp_energy_mean, dim3(1), dim3(1024), 0, (hipStream_t)stream, energies, (long)count, scale, (float*)nullptr, shift,
                       out);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

int nnpops_scale_by_scalar(void* stream, const float* in, int64_t count, const void* factor, int factor_is_double, float* out) {
    NNPOPS_REQUIRE(in && out && factor && count >= 0, "NULL device pointer");
    if (count == 0) return NNPOPS_OK;
    const int blocks = (int)std::min<int64_t>((count + 255) / 256, 2048);
    if (factor_is_double)
        hipLaunchKernelGGL(scale_by_scalar<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, (long)count, (const double*)factor, out);
    else
        hipLaunchKernelGGL(scale_by_scalar<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, (long)count, (const float*)factor, out);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

int nnpops_mlp_input_grad(void* stream, const nnpops_mlp_frame* frame) {
    MlpArgs g{};
    int blocks = 0;
    NNPOPS_REQUIRE(frame && frame->dx && frame->lddx >= frame->num_features && frame->lddx % 4 == 0 && ((uintptr_t)frame->dx % 16) == 0,
                   "dx rows must be 16-byte aligned");
    const int chunks = ((frame->num_features + 15) / 16 + kGradBlocks - 1) / kGradBlocks;
    int rc = check_and_fill(frame, g, true, chunks, &blocks);
    if (rc != NNPOPS_OK) return rc;
    if (blocks == 0) return NNPOPS_OK;
    if (g.dx_partial) {                                      // the forward launch has formed every member's share: add them up
        const long total = (long)g.n_grouped * ((g.F >> 2) + g.num_dead * 4);
        hipLaunchKernelGGL(mlp_sum_members, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, g);
        NNPOPS_HIP_TRY(hipGetLastError());
        return NNPOPS_OK;
    }
    hipLaunchKernelGGL(mlp_input_grad, dim3(blocks), dim3(kThreads), 2 * kStageBytes, (hipStream_t)stream, g);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

}  // extern "C"
