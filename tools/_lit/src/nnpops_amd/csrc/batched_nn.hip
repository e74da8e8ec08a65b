// batched_nn.hip -- dense layers of the ANI atomic networks on the matrix cores (C ABI: nnpops_gemm_split,
// nnpops_split_planes).
//
// What it serves: TorchANIBatchedNN (reference src/pytorch/BatchedNN.py:37-122, BatchedNN.cpp:30-50) evaluates, per atom
// and per ensemble member, Linear -> CELU(0.1) -> Linear -> CELU -> Linear -> CELU -> Linear.  With the atoms grouped by
// species every layer is a plain GEMM  C[M x N] = A[M x K] B  (M = atoms of the species, K / N = layer widths, batched
// over the ensemble members), and so is every step of the input-gradient pass.
//
// How: fp32 in, fp32 out, the products on v_mfma_f32_16x16x32_f16 with every operand carried as two fp16 planes
// (x = hi + 2^-11 lo', 22 significant bits; A B = Ahi Bhi + 2^-11 (Ahi Blo' + Alo' Bhi), fp32 accumulation) -- the same
// arithmetic as cfconv_filters_h2 (cfconv.hip; accuracy and rate measured in tools/ubench/split_f16_gemm.hip): the
// fp32-input MFMA runs at the vector rate on this chip, the half-precision one 16x faster.  B (the weights) is split
// once (nnpops_split_planes), A is split while it is staged into LDS.  The epilogue adds the bias and applies CELU, or
// multiplies by CELU' of a saved activation (backward); an optional prologue forms A = v[k] * CELU'(Y[m][k]) on the
// fly (the first step of the backward pass), so no elementwise kernel runs between the GEMMs.
//
// Tiling: workgroup = 4 waves = 64 x 128 of C, K in steps of 32; a wave owns 32 x 64 (2 x 4 MFMA blocks, two
// accumulator sets).  LDS: two stages of {A planes 64 x 32 halves, B planes 128 x 32 halves} = 48 KB; rows are 64 B =
// four 16-byte slots, the slot index XORed with a function of the row so that the lane groups of ds_read_b128 (which
// mix two K groups) find 16 different bank quarters.  Global loads of step s + 1 are in flight during the MFMAs of s.
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "host_common.h"

using namespace nnpops;

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;
constexpr int BM = 64, BK = 32;                           // BN = 128 or 64 (template): small problems need the workgroups
constexpr int stage_bytes(int BN) { return 2 * BM * 64 + 2 * BN * 64; }     // A planes + B planes, 64-byte rows

// byte offset of 16-byte slot `slot` (0..3) of row `row` inside a plane of 64-byte rows
__device__ __forceinline__ int sw(int row, int slot) {
    const int f = (0x78 >> (2 * ((row >> 2) & 3))) & 3;     // 0, 2, 3, 1 for rows 0-3, 4-7, 8-11, 12-15 (mod 16)
    return row * 64 + ((slot ^ f) << 4);
}

__device__ __forceinline__ float celu_grad_from_output(float y, float inv_alpha) { return y > 0.f ? 1.0f : y * inv_alpha + 1.0f; }

struct GemmArgs {
    int M, N, K;
    const float* A; long lda, strideA;
    const _Float16 *Bh, *Bl; long ldb, strideB;             // planes [N][ldb], ldb >= K rounded up to 32, zero padded
    float* C; long ldc, strideC;
    const float* bias; long strideBias;                      // EPI 1: [N]
    const float* Y; long ldy, strideY;                       // EPI 2: saved activation, same shape as C
    const float* PY; long ldpy, stridePY;                    // PRO 1: activation whose CELU' scales A ...
    const float* pv; long stridePv;                          // ... and the vector [K] that multiplies it (A itself is unused)
    float alpha, a_scale;
    const int *a_rows, *c_rows;                             // optional row maps: A row m is a_rows[m], C row m goes to c_rows[m]
};

// EPI: 0 plain, 1 bias + CELU, 2 times CELU'(Y).   PRO: 0 A as given, 1 A[m][k] = pv[k] * CELU'(PY[m][k])
// KS = 2: eight waves, K in steps of 64 -- waves 0-3 take the first 32 of a step, waves 4-7 the second, and the two halves
// are added through LDS at the end.  For problems too small to give every CU several workgroups (the layers of one
// species of a 2 000-atom frame) this doubles the waves that hide each other's latency and halves the number of steps.
// FAST (K a multiple of 8, rows of A 16-byte aligned: every layer of the networks): the staging loads are straight-line
// 16-byte loads from clamped addresses, masked afterwards -- a branch around a load makes the compiler drain the whole
// load queue (s_waitcnt vmcnt(0)) where the paths meet, inside the very function that is meant to run ahead.
template <int EPI, int PRO, int BN, int KS, bool FAST>
__global__ __launch_bounds__(256 * KS) void gemm_h2(GemmArgs g) {
    constexpr int NB = BN / 32;                             // 16-column blocks per wave (2 x NB MFMA blocks)
    constexpr int kSubBytes = stage_bytes(BN);              // one 32-wide K sub-step: A planes + B planes
    constexpr int kStageBytes = KS * kSubBytes;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave8 = threadIdx.x >> 6;
    const int sub = KS == 2 ? wave8 >> 2 : 0;               // which half of a K step this wave (and its loads) serves
    const int tid = threadIdx.x & 255, wave = wave8 & 3;
    const int wm = wave & 1, wn = wave >> 1;
    const int kpad = (g.K + 31) & ~31;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, b = blockIdx.z;
    const float* A = PRO == 0 ? g.A + (size_t)b * g.strideA : g.PY + (size_t)b * g.stridePY;
    const long lda = PRO == 0 ? g.lda : g.ldpy;
    const float* pv = PRO == 1 ? g.pv + (size_t)b * g.stridePv : nullptr;
    const _Float16* Bh = g.Bh + (size_t)b * g.strideB;
    const _Float16* Bl = g.Bl + (size_t)b * g.strideB;
    const float inv_alpha = 1.0f / g.alpha;

    // staging roles: A: thread -> (row = tid / 4, slot = tid % 4): 8 consecutive k;  B: rows tid / 4 and tid / 4 + 64
    const int srow = tid >> 2, sslot = tid & 3;
    const bool a_row_ok = m0 + srow < g.M;
    const int a_row = m0 + (a_row_ok ? srow : 0);
    const float* a_src = A + (size_t)(g.a_rows ? g.a_rows[a_row] : a_row) * lda + sslot * 8;
    const bool b_ok0 = n0 + srow < g.N, b_ok1 = BN == 128 && n0 + srow + 64 < g.N;
    const size_t b_off0 = (size_t)(n0 + (b_ok0 ? srow : 0)) * g.ldb + sslot * 8;
    const size_t b_off1 = (size_t)(n0 + (b_ok1 ? srow + 64 : 0)) * g.ldb + sslot * 8;

    struct Regs { float av[8]; float pw[PRO == 1 ? 8 : 1]; f16x8 bh0, bl0, bh1, bl1; bool a_ok, b_ok; };
    auto fetch = [&](int kstep, Regs& R) {
        const int k0 = kstep + 32 * sub;
        float (&av)[8] = R.av;
        f16x8 &bh0 = R.bh0, &bl0 = R.bl0, &bh1 = R.bh1, &bl1 = R.bl1;
        const int k = k0 + sslot * 8;
        const bool k_ok = k0 < kpad;                         // (KS = 2: the second half of the last step may lie past the planes)
        if constexpr (FAST) {
            const bool k_in = k < g.K;                      // (all eight or none)
            const float* src = a_src + (k_in ? k0 : -sslot * 8);
            const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
            av[0] = lo.x; av[1] = lo.y; av[2] = lo.z; av[3] = lo.w; av[4] = hi.x; av[5] = hi.y; av[6] = hi.z; av[7] = hi.w;
            if constexpr (PRO == 1) {
                const float* ps = pv + (k_in ? k : 0);
                const float4 plo = *reinterpret_cast<const float4*>(ps), phi = *reinterpret_cast<const float4*>(ps + 4);
                R.pw[0] = plo.x; R.pw[1] = plo.y; R.pw[2] = plo.z; R.pw[3] = plo.w; R.pw[4] = phi.x; R.pw[5] = phi.y; R.pw[6] = phi.z; R.pw[7] = phi.w;
            }
            const int kb = k_ok ? k0 : 0;
            bh0 = *reinterpret_cast<const f16x8*>(Bh + b_off0 + kb);
            bl0 = *reinterpret_cast<const f16x8*>(Bl + b_off0 + kb);
            if (BN == 128) {
                bh1 = *reinterpret_cast<const f16x8*>(Bh + b_off1 + kb);
                bl1 = *reinterpret_cast<const f16x8*>(Bl + b_off1 + kb);
            }
            R.a_ok = a_row_ok && k_in;
            R.b_ok = k_ok;
            return;
        }
        R.a_ok = true; R.b_ok = true;
#pragma unroll
        for (int i = 0; i < 8; i++) av[i] = 0.f;
        if (a_row_ok) {
            if (k + 8 <= g.K && (lda & 3) == 0) {
                const float4 lo = *reinterpret_cast<const float4*>(a_src + k0), hi = *reinterpret_cast<const float4*>(a_src + k0 + 4);
                av[0] = lo.x; av[1] = lo.y; av[2] = lo.z; av[3] = lo.w; av[4] = hi.x; av[5] = hi.y; av[6] = hi.z; av[7] = hi.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) if (k + i < g.K) av[i] = a_src[k0 + i];
            }
            if (PRO == 1) {
#pragma unroll
                for (int i = 0; i < 8; i++) av[i] = k + i < g.K ? pv[k + i] * celu_grad_from_output(av[i], inv_alpha) : 0.f;
            }
        }
        const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        bh0 = b_ok0 && k_ok ? *reinterpret_cast<const f16x8*>(Bh + b_off0 + k0) : zero;
        bl0 = b_ok0 && k_ok ? *reinterpret_cast<const f16x8*>(Bl + b_off0 + k0) : zero;
        if (BN == 128) {
            bh1 = b_ok1 && k_ok ? *reinterpret_cast<const f16x8*>(Bh + b_off1 + k0) : zero;
            bl1 = b_ok1 && k_ok ? *reinterpret_cast<const f16x8*>(Bl + b_off1 + k0) : zero;
        }
    };
    auto stage = [&](char* stage_base, const Regs& R) {
        char* base = stage_base + sub * kSubBytes;
        const float (&av)[8] = R.av;
        const f16x8 &bh0 = R.bh0, &bl0 = R.bl0, &bh1 = R.bh1, &bl1 = R.bl1;
        char* a_h = base; char* a_l = base + BM * 64;
        char* s_bh = base + 2 * BM * 64; char* s_bl = s_bh + BN * 64;
        const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        const float scale = R.a_ok ? g.a_scale : 0.f;       // (FAST: out-of-range rows / k were read from a valid address)
        f16x8 h, l;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float v = av[i];
            if constexpr (FAST && PRO == 1) v = R.pw[i] * celu_grad_from_output(v, inv_alpha);
            v *= scale;
            h[i] = (_Float16)v;
            l[i] = (_Float16)((v - (float)h[i]) * kLoScale);
        }
        const bool k0_ok = !FAST || (b_ok0 && R.b_ok), k1_ok = !FAST || (b_ok1 && R.b_ok);
        *reinterpret_cast<f16x8*>(a_h + sw(srow, sslot)) = h;
        *reinterpret_cast<f16x8*>(a_l + sw(srow, sslot)) = l;
        *reinterpret_cast<f16x8*>(s_bh + sw(srow, sslot)) = k0_ok ? bh0 : zero;
        *reinterpret_cast<f16x8*>(s_bl + sw(srow, sslot)) = k0_ok ? bl0 : zero;
        if (BN == 128) {
            *reinterpret_cast<f16x8*>(s_bh + sw(srow + 64, sslot)) = k1_ok ? bh1 : zero;
            *reinterpret_cast<f16x8*>(s_bl + sw(srow + 64, sslot)) = k1_ok ? bl1 : zero;
        }
    };

    f32x4 acc1[2][NB], acc2[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; mb++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++) { acc1[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int ksteps = (g.K + BK * KS - 1) / (BK * KS);
    const int r16 = lane & 15, kg = lane >> 4;
    auto compute = [&](const char* stage_base) {
        const char* cur = stage_base + sub * kSubBytes;
        const char* a_h = cur; const char* a_l = cur + BM * 64;
        const char* s_bh = cur + 2 * BM * 64; const char* s_bl = s_bh + BN * 64;
        f16x8 ah[2], al[2];
#pragma unroll
        for (int mb = 0; mb < 2; mb++) {
            const int row = wm * 32 + mb * 16 + r16;
            ah[mb] = *reinterpret_cast<const f16x8*>(a_h + sw(row, kg));
            al[mb] = *reinterpret_cast<const f16x8*>(a_l + sw(row, kg));
        }
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            const int row = wn * (BN / 2) + nb * 16 + r16;
            const f16x8 bh = *reinterpret_cast<const f16x8*>(s_bh + sw(row, kg));
            const f16x8 bl = *reinterpret_cast<const f16x8*>(s_bl + sw(row, kg));
#pragma unroll
            for (int mb = 0; mb < 2; mb++) {
                acc1[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], bh, acc1[mb][nb], 0, 0, 0);
                acc2[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], bl, acc2[mb][nb], 0, 0, 0);
                acc2[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mb], bh, acc2[mb][nb], 0, 0, 0);
            }
        }
    };
    // The loads of K step s + 1 are issued before the MFMAs of step s and written to LDS after them.  (A second register
    // set, two steps ahead, costs the third workgroup per CU -- 176 registers -- and loses more than it hides.)
    Regs R;
    fetch(0, R);
    stage(lds, R);
    __syncthreads();
    for (int s = 0; s < ksteps; s++) {
        if (s + 1 < ksteps) fetch((s + 1) * BK * KS, R);
        compute(lds + (s & 1) * kStageBytes);
        if (s + 1 < ksteps) stage(lds + ((s + 1) & 1) * kStageBytes, R);
        __syncthreads();
    }

    if constexpr (KS == 2) {                                // the second half hands its sums over (16 * NB floats per lane)
        float* red = reinterpret_cast<float*>(lds);
        if (sub == 1) {
#pragma unroll
            for (int mb = 0; mb < 2; mb++)
#pragma unroll
                for (int nb = 0; nb < NB; nb++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        red[((mb * NB + nb) * 4 + q) * 256 + tid] = acc1[mb][nb][q] + kLoInv * acc2[mb][nb][q];
        }
        __syncthreads();
        if (sub == 1) return;
#pragma unroll
        for (int mb = 0; mb < 2; mb++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
#pragma unroll
                for (int q = 0; q < 4; q++) acc1[mb][nb][q] += red[((mb * NB + nb) * 4 + q) * 256 + tid];
    }
    // ---- epilogue: D[row = 4 * (lane >> 4) + q][col = lane & 15] of every block ----
    const float out_scale = 1.0f / g.a_scale;
    float* C = g.C + (size_t)b * g.strideC;
    const float* bias = EPI == 1 ? g.bias + (size_t)b * g.strideBias : nullptr;
    const float* Y = EPI == 2 ? g.Y + (size_t)b * g.strideY : nullptr;
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        const int col = n0 + wn * (BN / 2) + nb * 16 + r16;
        if (col >= g.N) continue;
        const float bv = EPI == 1 ? bias[col] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; mb++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int row = m0 + wm * 32 + mb * 16 + 4 * kg + q;
                if (row >= g.M) continue;
                float v = (acc1[mb][nb][q] + kLoInv * acc2[mb][nb][q]) * out_scale;
                if (EPI == 1) {
                    v += bv;
                    v = v > 0.f ? v : g.alpha * (__expf(v * inv_alpha) - 1.0f);           // CELU (BatchedNN.py:103-109)
                } else if (EPI == 2) {
                    v *= celu_grad_from_output(Y[(size_t)row * g.ldy + col], inv_alpha);
                }
                C[(size_t)(g.c_rows ? g.c_rows[row] : row) * g.ldc + col] = v;
            }
    }
}

// W [rows][cols] fp32 -> planes [rows][ldp] (ldp >= cols, zero padded), optionally of the transpose
__global__ __launch_bounds__(256) void split_planes(int rows, int cols, const float* __restrict__ W, long ldw, int transpose,
                                                    _Float16* __restrict__ hi, _Float16* __restrict__ lo, long ldp) {
    // output plane element (r, c), r < rows, c < ldp;  source = W[r][c] or, transposed, W[c][r]
    const long total = (long)rows * ldp;
    for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const int r = (int)(q / ldp), c = (int)(q % ldp);
        float v = 0.f;
        if (c < cols) v = transpose ? W[(size_t)c * ldw + r] : W[(size_t)r * ldw + c];
        const _Float16 h = (_Float16)v;
        hi[q] = h;
        lo[q] = (_Float16)((v - (float)h) * kLoScale);
    }
}

// out[out_rows ? out_rows[m] : m] = A[m][0..K) . w + bias : the last layer of the networks (one output per member, summed over the members).
// One wave per row, float4 loads.
__global__ __launch_bounds__(256) void rows_dot(int M, int K, const float* __restrict__ A, long lda, const float* __restrict__ w,
                                                float bias, float* __restrict__ out, const int* __restrict__ out_rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* a = A + (size_t)row * lda;
    float acc = 0.f;
    if ((K & 3) == 0 && (lda & 3) == 0) {
        for (int k = 4 * lane; k < K; k += 256) {
            const float4 x = *reinterpret_cast<const float4*>(a + k), y = *reinterpret_cast<const float4*>(w + k);
            acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
    } else {
        for (int k = lane; k < K; k += 64) acc += a[k] * w[k];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[out_rows ? out_rows[row] : row] = acc + bias;
}

}  // namespace

extern "C" {

int nnpops_rows_dot(void* stream, int M, int K, const float* A, long lda, const float* w, float bias, float* out, const int* out_rows) {
    NNPOPS_REQUIRE(M > 0 && K > 0 && A && w && out, "empty problem or NULL device pointer");
    hipLaunchKernelGGL(rows_dot, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, M, K, A, lda, w, bias, out, out_rows);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}


int nnpops_split_planes(void* stream, int rows, int cols, const float* w, long ldw, int transpose, void* hi, void* lo, long ldp) {
    NNPOPS_REQUIRE(w && hi && lo, "NULL device pointer");
    NNPOPS_REQUIRE(rows > 0 && cols > 0 && ldp >= cols && ldp % 32 == 0, "planes need a row length that is a multiple of 32 and >= cols");
    const long total = (long)rows * ldp;
    const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(split_planes, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rows, cols, w, ldw, transpose,
                       (_Float16*)hi, (_Float16*)lo, ldp);
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

int nnpops_gemm_split(void* stream, int M, int N, int K, int batch, const float* A, long lda, long strideA, const void* Bh,
                      const void* Bl, long ldb, long strideB, float* C, long ldc, long strideC, int epilogue, const float* bias,
                      long strideBias, const float* Y, long ldy, long strideY, int prologue, const float* PY, long ldpy,
                      long stridePY, const float* pv, long stridePv, float alpha, float a_scale, const int* a_rows, const int* c_rows) {
    NNPOPS_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "empty GEMM");
    NNPOPS_REQUIRE(Bh && Bl && C, "NULL device pointer");
    NNPOPS_REQUIRE(ldb % 8 == 0 && ldb >= ((K + 31) & ~31), "planes need ldb >= K rounded up to 32 (zero padded)");
    NNPOPS_REQUIRE(epilogue >= 0 && epilogue <= 2 && prologue >= 0 && prologue <= 1, "unknown epilogue / prologue");
    NNPOPS_REQUIRE(prologue == 1 ? (PY && pv) : (A != nullptr), "NULL A operand");
    NNPOPS_REQUIRE(epilogue != 1 || bias, "bias + CELU epilogue needs a bias");
    NNPOPS_REQUIRE(epilogue != 2 || Y, "CELU' epilogue needs the saved activation");
    NNPOPS_REQUIRE(alpha > 0 && a_scale > 0, "alpha and a_scale must be positive");
    NNPOPS_REQUIRE((!a_rows && !c_rows) || batch == 1, "row maps are for single problems");
    GemmArgs g{M, N, K, A, lda, strideA, (const _Float16*)Bh, (const _Float16*)Bl, ldb, strideB, C, ldc, strideC, bias, strideBias,
               Y, ldy, strideY, PY, ldpy, stridePY, pv, stridePv, alpha, a_scale, a_rows, c_rows};
    // 64 x 128 tiles, four waves, when they still give every CU a few workgroups; otherwise 64 x 64 tiles with eight waves
    // splitting every K step (the layers of one species of a 2 000-atom frame are 10-300 tiles: one wave per SIMD hides
    // nothing)
    const long tiles128 = (long)((N + 127) / 128) * ((M + BM - 1) / BM) * batch;
    const long tiles64 = (long)((N + 63) / 64) * ((M + BM - 1) / BM) * batch;
    const int shape = tiles128 >= 512 ? 0 : tiles64 >= 600 ? 1 : 2;       // 0: 64x128 | 1: 64x64 | 2: 64x64, eight waves splitting K
    const int bn = shape == 0 ? 128 : 64;
    const dim3 grid((N + bn - 1) / bn, (M + BM - 1) / BM, batch);
    const size_t lds = (shape == 2 ? 2 : 1) * 2 * stage_bytes(bn);
    hipStream_t st = (hipStream_t)stream;
    const long lda_eff = prologue == 1 ? ldpy : lda;
    const float* a_eff = prologue == 1 ? PY : A;
    const bool fast = (K % 8) == 0 && (lda_eff % 4) == 0 && ((uintptr_t)a_eff % 16) == 0 &&
                      (batch == 1 || ((prologue == 1 ? stridePY : strideA) % 4) == 0) &&
                      (prologue == 0 || (((uintptr_t)pv % 16) == 0 && (stridePv % 4) == 0)) &&
                      !(std::getenv("NNPOPS_GEMM_FAST") && std::atoi(std::getenv("NNPOPS_GEMM_FAST")) == 0);
#define NNPOPS_LAUNCH_GEMM_F(E, P, F) \
    do { if (shape == 0) hipLaunchKernelGGL((gemm_h2<E, P, 128, 1, F>), grid, dim3(256), lds, st, g); \
         else if (shape == 1) hipLaunchKernelGGL((gemm_h2<E, P, 64, 1, F>), grid, dim3(256), lds, st, g); \
         else hipLaunchKernelGGL((gemm_h2<E, P, 64, 2, F>), grid, dim3(512), lds, st, g); } while (0)
#define NNPOPS_LAUNCH_GEMM(E, P) do { if (fast) NNPOPS_LAUNCH_GEMM_F(E, P, true); else NNPOPS_LAUNCH_GEMM_F(E, P, false); } while (0)
    if (prologue == 0) {
        if (epilogue == 0) NNPOPS_LAUNCH_GEMM(0, 0); else if (epilogue == 1) NNPOPS_LAUNCH_GEMM(1, 0); else NNPOPS_LAUNCH_GEMM(2, 0);
    } else {
        if (epilogue == 0) NNPOPS_LAUNCH_GEMM(0, 1); else if (epilogue == 1) NNPOPS_LAUNCH_GEMM(1, 1); else NNPOPS_LAUNCH_GEMM(2, 1);
    }
#undef NNPOPS_LAUNCH_GEMM
#undef NNPOPS_LAUNCH_GEMM_F
    NNPOPS_HIP_TRY(hipGetLastError());
    return NNPOPS_OK;
}

}  // extern "C"
