// Micro-benchmark: cost of ds_add_f32 under different conflict patterns (cycles per wave-instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, int iters) {
    __shared__ float buf[4096];
    const int lane = threadIdx.x;
    for (int q = lane; q < 4096; q += 64) buf[q] = 0.f;
    __syncthreads();
    int idx;
    if (MODE == 0) idx = lane;                 // conflict-free, distinct banks (2 lanes / bank over 64 lanes)
    else if (MODE == 1) idx = lane & 7;        // 8 lanes share each address
    else if (MODE == 2) idx = 0;               // all 64 lanes same address
    else if (MODE == 3) idx = (lane & 7) * 32; // 8 addresses all in the same bank
    else if (MODE == 4) idx = lane;            // plain read-modify-write (no atomic), distinct
    else idx = lane * 33 & 4095;
    long long t0 = clock64();
    float v = 1.0f + lane;
    for (int i = 0; i < iters; i++) {
        if (MODE == 4) { buf[idx] += v; }
        else if (MODE == 6) { if (lane < 8) atomicAdd(&buf[idx], v); }
        else atomicAdd(&buf[idx], v);
        v += 1.0f;
    }
    __syncthreads();
    long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + lane] = buf[lane];
}
int main() {
    float* out; long long* cyc;
    const int blocks = 256 * 8, iters = 1000;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    std::vector<long long> h(blocks);
    auto run = [&](auto kern, const char* name) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto c : h) avg += c; avg /= blocks;
        // chip-level: blocks*iters instrs over 256 CUs
        double cyc_per_instr_per_cu = ms * 1e-3 * 2.4e9 / (double(blocks) * iters / 256.0);
        printf("%-44s wave-latency %.1f clk/instr ; CU-throughput %.1f clk/instr (kernel %.3f ms)\n", name, avg / iters, cyc_per_instr_per_cu, ms);
    };
    run(k<0>, "ds_add_f32 distinct addresses");
    run(k<1>, "ds_add_f32 8 lanes per address");
    run(k<2>, "ds_add_f32 64 lanes same address");
    run(k<3>, "ds_add_f32 8 addresses, one bank");
    run(k<4>, "plain RMW distinct addresses");
    run(k<6>, "ds_add_f32 only 8 lanes active, distinct");
    return 0;
}
