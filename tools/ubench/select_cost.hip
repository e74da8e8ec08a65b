// select_cost.hip -- what the instructions the per-atom ANI kernels are made of cost a SIMD of gfx950: selects, compares, lane
// traffic, integer multiplies, LDS reads -- same method as valu_issue.hip (8 independent chains, LOOPS x 32 instructions per wave,
// W waves per SIMD), one line per instruction.  Round 5: valu_issue.hip priced v_cndmask_b32 at 8.6 simple instructions; this
// file is there to find out which forms of a select are cheap.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/select_cost.hip -o tools/ubench/select_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
static const char* g_filter = nullptr;      // argv[1]: only the lines whose name contains it
typedef float f2 __attribute__((ext_vector_type(2)));
#define LOOPS 1024

template <int KIND>
__global__ __launch_bounds__(256) void spin(float* out, float a, float b, int n) {
    __shared__ float lds[1024];
    float x[8];
    f2 y[8];
    int q[8];
    for (int k = 0; k < 8; k++) { x[k] = a + k + threadIdx.x; y[k] = f2{a + k, b + k}; q[k] = (int)a + k + threadIdx.x; }
    lds[threadIdx.x] = a; lds[threadIdx.x + 256] = b; lds[threadIdx.x + 512] = a; lds[threadIdx.x + 768] = b;
    __syncthreads();
    unsigned long long sm = n > 5 ? 0x5555555555555555ull : 0x3333333333333333ull;
    int sacc = n;
    const f2 a2 = {a, a};
    for (int it = 0; it < LOOPS; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[k]) : "v"(a));
                if (KIND == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a));
                if (KIND == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "s"(sm));
                if (KIND == 3) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[k]), "v"(a) : "vcc");
                if (KIND == 4) { asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : "vcc"); }
                if (KIND == 5) asm volatile("v_max_f32 %0, %1, %0" : "+v"(x[k]) : "v"(a));
                if (KIND == 6) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(q[k]) : "v"(q[(k + 1) & 7]), "v"(q[(k + 2) & 7]));
                if (KIND == 7) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sacc) : "v"(x[k]));
                if (KIND == 9) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[k]));
                if (KIND == 10) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(q[k]) : "v"(q[(k + 1) & 7]));
                if (KIND == 11) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(q[k]) : "v"(q[(k + 1) & 7]));
                if (KIND == 12) asm volatile("v_mad_u32_u24 %0, %1, %0, %0" : "+v"(q[k]) : "v"(q[(k + 1) & 7]));
                if (KIND == 13) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(y[k]) : "v"(a2));
                if (KIND == 14) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(y[k]) : "v"(a2));
                if (KIND == 15) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[k]));
                if (KIND == 16) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[k]));
                if (KIND == 17) asm volatile("v_log_f32 %0, %0" : "+v"(x[k]));
                if (KIND == 18) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(q[k]) : "s"((unsigned)sm));
                if (KIND == 19) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x[k]));
                if (KIND == 20) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(q[k]) : "v"(q[(k + 1) & 7]));
                if (KIND == 21) asm volatile("v_and_b32 %0, %1, %0" : "+v"(q[k]) : "v"(q[(k + 1) & 7]));
                if (KIND == 22) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(q[k]), "v"(q[(k + 1) & 7]) : "vcc");
                if (KIND == 23) asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(sm) : "v"(q[k]), "v"(q[(k + 1) & 7]));
                if (KIND == 24) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x[k]) : "v"((q[k] & 255) * 4));
                if (KIND == 25) asm volatile("ds_read_b128 %0, %1" : "=v"(*(float4*)&x[(k & 1) * 4]) : "v"((q[k] & 63) * 16));
                if (KIND == 26) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b));
                if (KIND == 27) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
                if (KIND == 28) asm volatile("v_min_i32 %0, %1, %0" : "+v"(q[k]) : "v"(q[(k + 1) & 7]));
                if (KIND == 29) asm volatile("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(q[k]));
                if (KIND == 30) asm volatile("s_and_saveexec_b64 %0, vcc\n\ts_or_b64 exec, exec, %0" : "=s"(sm) : : "vcc");
                if (KIND == 31) asm volatile("v_add_f32 %0, %1, %0\n\ts_nop 0" : "+v"(x[k]) : "v"(a));
                if (KIND == 32) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y[k]) : "v"(a2), "v"(a2));
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    float s = 0;
    for (int k = 0; k < 8; k++) s += x[k] + y[k].x + y[k].y + q[k];
    out[blockIdx.x * 256 + threadIdx.x] = s + sacc + (float)(sm & 1);
}

template <int KIND>
void run(const char* name, float* out, int per = 1) {
    if (g_filter && !strstr(name, g_filter)) return;
    for (int W : {1, 4, 8}) {
        const int blocks = 256 * W;
        hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.5f, 3);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.5f, 3);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)LOOPS * 32 * per;
        printf("%-34s W=%d  %8.1f us   ns per instr per SIMD: %.3f\n", name, W, ms * 1e3, ms * 1e6 / (instr * W));
    }
}

int main(int argc, char** argv) {
    if (argc > 1) g_filter = argv[1];
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
    run<0>("v_add_f32", out);
    run<26>("v_fma_f32", out);
    run<27>("v_fmac_f32", out);
    run<31>("v_add_f32 + s_nop", out);
    run<1>("v_cndmask_b32 (vcc)", out);
    run<2>("v_cndmask_b32_e64 (sgpr mask)", out);
    run<3>("v_cmp_lt_f32 -> vcc", out);
    run<22>("v_cmp_lt_i32 -> vcc", out);
    run<23>("v_cmp_lt_i32_e64 -> sgpr", out);
    run<4>("v_cmp + v_cndmask pair", out, 2);
    run<5>("v_max_f32", out);
    run<28>("v_min_i32", out);
    run<6>("v_bfi_b32", out);
    run<21>("v_and_b32", out);
    run<20>("v_lshl_add_u32", out);
    run<7>("v_readlane_b32", out);
    run<9>("v_mov_b32_dpp row_shr", out);
    run<29>("v_max_i32_dpp row_shr", out);
    run<18>("v_mbcnt_lo", out);
    run<10>("v_mul_lo_u32", out);
    run<11>("v_mul_u32_u24", out);
    run<12>("v_mad_u32_u24", out);
    run<19>("v_cvt_f32_i32", out);
    run<13>("v_pk_mul_f32", out);
    run<14>("v_pk_add_f32", out);
    run<32>("v_pk_fma_f32", out);
    run<15>("v_sqrt_f32", out);
    run<16>("v_rcp_f32", out);
    run<17>("v_log_f32", out);
    run<24>("ds_read_b32 + wait", out);
    run<25>("ds_read_b128 (no wait)", out);
    run<30>("s_and_saveexec + s_or exec", out, 2);
    return 0;
}
