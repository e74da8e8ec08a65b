// Micro-benchmark: do the matrix cores and the vector ALUs of one SIMD overlap ACROSS waves?
// A workgroup is 8 waves = 2 per SIMD.  Modes per wave (chosen by wave parity so that each SIMD gets one of each):
//   M = only v_mfma_f32_16x16x4_f32 (8 independent accumulators, operands in registers, no LDS)
//   V = only v_fma_f32 (8 independent chains)
//   T = only v_exp_f32 (quarter rate)
//   H = only v_mfma_f32_16x16x32_f16 (the real matrix core; 16 per iteration)
// Times: M alone, V alone, M next to V, M next to T, and M+V interleaved inside ONE wave.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ void mfma_block(f32x4 (&acc)[8], float a, float b, int iters) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < 8; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
}
__device__ __forceinline__ void hmfma_block(f32x4 (&acc)[8], f16x8 a, f16x8 b, int iters) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 8; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
    }
}
__device__ __forceinline__ void fma_block(float (&v)[8], float a, float b, int iters) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = __builtin_fmaf(v[c], a, b);      // 64 v_fma per iteration = 256 issue cycles
    }
}
__device__ __forceinline__ void exp_block(float (&v)[8], int iters) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = __builtin_amdgcn_exp2f(v[c]);    // 16 v_exp per iteration = 256 issue cycles
    }
}

// mode: bit0 = even waves run M, bit1 = odd waves run V, bit2 = odd waves run T, bit3 = every wave interleaves M and V
__global__ __launch_bounds__(512) void k(float* out, int mode, int iters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // waves go to SIMDs 0,2,1,3,0,2,1,3: waves w and w + 4 share a SIMD
    const bool first = wave < 4;
    f32x4 acc[8];
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; c++) { acc[c] = f32x4{0.f, 0.f, 0.f, 0.f}; v[c] = 0.001f * (lane + c); }
    const float a = 1.0f + 1e-7f * lane, b = 1e-9f;
    f16x8 ha, hb;
#pragma unroll
    for (int c = 0; c < 8; c++) { ha[c] = (_Float16)(0.01f * lane); hb[c] = (_Float16)(0.001f * c); }
    if (mode & 32) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[c], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; q++) v[(4 * c + q) & 7] = __builtin_fmaf(v[(4 * c + q) & 7], a, b);   // 4 v_fma per MFMA
                }
        }
    } else if (mode & 8) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 8; r++) v[r] = __builtin_fmaf(v[r], a, b);   // 8 v_fma (32 cycles) in each MFMA's shadow
            }
        }
    } else if (mode & 64) {
        fma_block(v, a, b, iters);                                  // every wave: two V waves per SIMD
    } else if (mode & 128) {
        if (first) exp_block(v, iters); else fma_block(v, a, b, iters);   // T on one wave, V on the other wave of the SIMD
    } else if (first) {
        if (mode & 1) mfma_block(acc, a, b, iters);
        if (mode & 16) hmfma_block(acc, ha, hb, iters);
    } else {
        if (mode & 2) fma_block(v, a, b, iters);
        if (mode & 4) exp_block(v, iters);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3] + v[c];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float* out;
    const int blocks = 256, iters = 2000;
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    auto run = [&](int mode, const char* name) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, mode, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, mode, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %.3f ms  (%.0f cycles per iteration at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
    };
    run(1, "M alone: 8 MFMA / iteration (ideal 256 cycles)");
    run(2, "V alone: 64 v_fma / iteration (ideal 256 cycles)");
    run(4, "T alone: 16 v_exp / iteration (ideal 256 cycles)");
    run(3, "M on one wave, V on the other wave of the SIMD");
    run(5, "M on one wave, T on the other wave of the SIMD");
    run(8, "one wave: 8 x (MFMA + 8 v_fma)  [2 such waves per SIMD]");
    run(64, "V on both waves of the SIMD (2 x 64 v_fma / iteration)");
    run(128, "T on one wave, V on the other wave of the SIMD");
    run(16, "H alone: 16 f16 MFMA 16x16x32 / iteration");
    run(18, "H on one wave, V on the other wave of the SIMD");
    run(20, "H on one wave, T on the other wave of the SIMD");
    run(32, "one wave: 16 x (f16 MFMA + 4 v_fma)  [2 such waves per SIMD]");
    return 0;
}
