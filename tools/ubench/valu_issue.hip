// valu_issue.hip -- what one vector instruction costs a SIMD of gfx950, by instruction type and by the number of
// waves resident on the SIMD.  Every wave runs LOOPS iterations of 32 instructions of one type on 8 independent
// register chains; a workgroup is 256 lanes (one wave per SIMD), W workgroups per CU.  Prints shader-clock cycles per
// instruction per SIMD (s_memtime around the loop of one wave, times 1/W would be the per-wave view).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue.hip -o tools/ubench/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define LOOPS 2048

template <int KIND>
__global__ __launch_bounds__(256) void spin(float* out, long long* cyc, float a, float b) {
    float x[8];
    f2 y[8];
    f4 m[8];
    int q[8];
    for (int k = 0; k < 8; k++) { x[k] = a + k; y[k] = f2{a + k, b + k}; m[k] = f4{a, b, a, b}; q[k] = (int)a + k; }
    const f2 a2 = {a, a}, b2 = {b, b};
    int sacc = (int)a;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < LOOPS; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y[k]) : "v"(a2), "v"(b2));
                if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k]));
                if (KIND == 3) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(m[k]) : "v"(a), "v"(b));
                if (KIND == 4) asm volatile("v_add_u32 %0, %1, %0" : "+v"(q[k]) : "v"(q[(k + 1) & 7]));
                if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a));
                if (KIND == 6) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc)); }
                if (KIND == 7) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[k]) : "v"(a));
                if (KIND == 8) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[k]) : "v"(a), "v"(b));
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int k = 0; k < 8; k++) s += x[k] + y[k].x + y[k].y + m[k][0] + m[k][3] + q[k];
    out[blockIdx.x * 256 + threadIdx.x] = s + sacc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, long long* cyc) {
    for (int W : {1, 2, 4, 8}) {
        const int blocks = 256 * W;
        hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f, 0.5f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
        double avg = 0; for (long long v : h) avg += (double)v; avg /= blocks;
        const double instr = (double)LOOPS * 32 * (KIND == 6 ? 2 : 1);
        // s_memtime ticks at a constant 100 MHz on this family; report both the wall-time and the tick view
        printf("%-22s W=%d  %8.1f us   ns per instr per SIMD (W waves interleaved): %.3f   ticks/wave %.0f\n", name, W, ms * 1e3,
               ms * 1e6 / (instr * W), avg);
    }
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 8); hipMalloc(&cyc, sizeof(long long) * 256 * 8);
    run<0>("v_fma_f32", out, cyc);
    run<7>("v_mul_f32", out, cyc);
    run<1>("v_pk_fma_f32", out, cyc);
    run<2>("v_exp_f32", out, cyc);
    run<3>("v_mfma_4x4x1_16b", out, cyc);
    run<8>("v_mfma_16x16x4_f32", out, cyc);
    run<4>("v_add_u32", out, cyc);
    run<5>("v_cndmask_b32", out, cyc);
    run<6>("v_fma+s_add pairs", out, cyc);
    return 0;
}
