// Micro-benchmark: marginal cost of VALU "filler" instructions between back-to-back f16 MFMAs, for the two
// shapes (16x16x32: 16 K FLOP, 32x32x16: 32 K FLOP), one or two waves per SIMD.  Prints cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int SHAPE, int FILL, int KIND>   // KIND 0 = v_fma_f32, 1 = v_exp_f32, 2 = v_cvt_pkrtz
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 ha, hb;
#pragma unroll
    for (int c = 0; c < 8; c++) { ha[c] = (_Float16)(0.01f * lane); hb[c] = (_Float16)(0.001f * c); }
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; c++) v[c] = 0.001f * (lane + c);
    const float a = 1.0f + 1e-7f * lane, b = 1e-9f;
    f32x4 acc4[8];
    f32x16 acc16[4];
#pragma unroll
    for (int c = 0; c < 8; c++) acc4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc16[c][q] = 0.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            if (SHAPE == 16) acc4[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4[c], 0, 0, 0);
            else acc16[c & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc16[c & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < FILL; q++) {
                float& t = v[(c * FILL + q) & 7];
                if (KIND == 0) t = __builtin_fmaf(t, a, b);
                else if (KIND == 1) t = __builtin_amdgcn_exp2f(t);
                else { auto h = __builtin_amdgcn_cvt_pkrtz(t, a); t = (float)h[0] + b; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; c++) s += acc4[c][0] + acc4[c][3] + v[c];
#pragma unroll
    for (int c = 0; c < 4; c++) s += acc16[c][0] + acc16[c][15];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int FILL, int KIND>
void run(float* out, int waves_per_simd) {
    const int blocks = 256, iters = 2000, threads = 256 * waves_per_simd;
    hipLaunchKernelGGL((k<SHAPE, FILL, KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, FILL, KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const char* kind = KIND == 0 ? "v_fma" : KIND == 1 ? "v_exp" : "cvt_pk+cvt+add";
    printf("%dx%d f16, %d wave/SIMD, %d %s per MFMA: %.1f cycles per MFMA per SIMD\n", SHAPE, SHAPE, waves_per_simd, FILL, kind,
           ms * 1e-3 * 2.4e9 / (iters * 8.0 * waves_per_simd));
}

int main() {
    float* out;
    (void)hipMalloc(&out, (size_t)256 * 512 * 4);
    for (int w = 1; w <= 2; w++) {
        run<16, 0, 0>(out, w); run<16, 1, 0>(out, w); run<16, 2, 0>(out, w); run<16, 4, 0>(out, w); run<16, 8, 0>(out, w);
        run<16, 1, 1>(out, w); run<16, 2, 1>(out, w);
        run<32, 0, 0>(out, w); run<32, 2, 0>(out, w); run<32, 4, 0>(out, w); run<32, 8, 0>(out, w); run<32, 16, 0>(out, w);
        run<32, 2, 1>(out, w);
    }
    return 0;
}
