// launch_floor.hip -- what does a launch of N short-lived workgroups cost before any of them does anything?
// (round 4: the angular forward kernel with no atom work at all took 6.4 of its 20.5 us, profiles/r04b_probe.json)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_floor.hip -o tools/ubench/launch_floor && tools/ubench/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int VG>
__global__ void empty_kernel(int* out, int n) {
    extern __shared__ char lds[];
    if (n < 0) {                                   // never true: keeps the LDS allocation and the registers alive
        float v[VG];
        for (int k = 0; k < VG; k++) v[k] = lds[k] * (float)threadIdx.x;
        float s = 0;
        for (int k = 0; k < VG; k++) s += v[k] * v[(k * 7) % VG];
        out[threadIdx.x] = (int)s;
    }
}

__global__ void one_load_kernel(const int* __restrict__ src, int* out, int n) {
    extern __shared__ char lds[];
    const int v = src[blockIdx.x];                 // one dependent round trip per workgroup
    if (v == 123456789) out[0] = v;
}

__global__ void chain_kernel(const int* __restrict__ src, int* out, int n, int hops) {
    extern __shared__ char lds[];
    int v = blockIdx.x;
    for (int h = 0; h < hops; h++) v = src[v];     // `hops` dependent round trips (src[i] = i)
    if (v == 123456789) out[0] = v;
}

template <class F>
float time_us(F&& launch, int reps = 200) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 20; r++) launch();
    std::vector<float> t;
    for (int round = 0; round < 9; round++) {
        hipEventRecord(a);
        for (int r = 0; r < reps; r++) launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        t.push_back(1e3f * ms / reps);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    int *src, *out;
    const int N = 1 << 16;
    hipMalloc(&src, N * sizeof(int)); hipMalloc(&out, 1024 * sizeof(int));
    std::vector<int> h(N);
    for (int i = 0; i < N; i++) h[i] = i;
    hipMemcpy(src, h.data(), N * sizeof(int), hipMemcpyHostToDevice);
    printf("back-to-back launches of one kernel, us per launch (median of 9 x 200)\n");
    for (int groups : {256, 2500, 5000, 10000, 20000})
        for (int threads : {64, 128, 256})
            for (int ldsb : {0, 10560, 32768}) {
                const float t0 = time_us([&] { hipLaunchKernelGGL(empty_kernel<8>, dim3(groups), dim3(threads), ldsb, 0, out, 1); });
                const float t1 = time_us([&] { hipLaunchKernelGGL(one_load_kernel, dim3(groups), dim3(threads), ldsb, 0, src, out, 1); });
                const float t3 = time_us([&] { hipLaunchKernelGGL(chain_kernel, dim3(groups), dim3(threads), ldsb, 0, src, out, 1, 3); });
                printf("groups %6d x %3d threads, %5d B LDS: empty %6.2f   one load %6.2f   3 dependent loads %6.2f\n", groups, threads, ldsb, t0, t1, t3);
            }
    return 0;
}
