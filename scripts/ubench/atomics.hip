// Micro-benchmark: throughput of global integer atomics onto a small table of hot counters
// (the tile-count pattern of gs_bin.hip) under different layouts / scopes.
//   hipcc --offload-arch=gfx950 -O3 atomics.hip -o atomics && ./atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}

template <int SCOPE, bool RET>
__global__ void k_atomics(int n, int per, int tiles, int stride, int copies, int *counters, int *sink) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int copy = 0;
    if (copies > 1) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        copy = (xcc & 0xF) % copies;
    }
    int *base = counters + (size_t)copy * tiles * stride;
    int acc = 0;
    for (int j = 0; j < per; j++) {
        int t = hash(i * 7919u + j) % tiles;
        if (RET) acc += __hip_atomic_fetch_add(base + (size_t)t * stride, 1, __ATOMIC_RELAXED, SCOPE);
        else (void)__hip_atomic_fetch_add(base + (size_t)t * stride, 1, __ATOMIC_RELAXED, SCOPE);
    }
    if (RET && acc == 0x7fffffff) sink[0] = acc;
}

// LDS-privatised histogram per workgroup, flushed with global atomics for non-zero bins
__global__ void k_lds(int n, int per, int tiles, int *counters) {
    extern __shared__ int h[];
    for (int t = threadIdx.x; t < tiles; t += blockDim.x) h[t] = 0;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int j = 0; j < per; j++) atomicAdd(&h[hash(i * 7919u + j) % tiles], 1);
    __syncthreads();
    for (int t = threadIdx.x; t < tiles; t += blockDim.x)
        if (h[t]) atomicAdd(&counters[t], h[t]);
}

// Gradient-scatter pattern of the backward compositing kernel: one wave-instruction with nine active
// lanes per (tile, Gaussian) entry.  MODE 0: four separate arrays (v_xy[N,2] v_conic[N,3]
// v_colors[N,3] v_opacity[N]); MODE 1: one record of `rec` floats per Gaussian, 9 used.
template <int MODE>
__global__ void k_scatter9(int entries_per_wave, int N, int rec, float *a0, float *a1, float *a2, float *a3) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float *base; int stride;
    if (MODE == 0) {
        base = lane < 2 ? a0 + lane : lane < 5 ? a1 + (lane - 2) : lane < 8 ? a2 + (lane - 5) : a3;
        stride = lane < 2 ? 2 : lane < 8 ? 3 : 1;
    } else {
        base = a0 + lane; stride = rec;
    }
    for (int e = 0; e < entries_per_wave; e++) {
        const int g = hash(wave * 1315423911u + e) % N;
        if (lane < 9) atomicAdd(base + (size_t)g * stride, 1.0f);
    }
}

template <typename F>
float timeit(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; r++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1000.f;
}

int main() {
    const int n = 1000000, per = 2, tiles = 8160;
    int *c, *sink;
    CK(hipMalloc(&c, (size_t)tiles * 64 * 8 * sizeof(int) + 4096));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(c, 0, (size_t)tiles * 64 * 8 * sizeof(int)));
    dim3 g((n + 255) / 256), b(256);
    for (int stride : {1, 16, 64}) {
        float us = timeit([&] { hipLaunchKernelGGL((k_atomics<__HIP_MEMORY_SCOPE_AGENT, false>), g, b, 0, 0, n, per, tiles, stride, 1, c, sink); });
        printf("agent scope, no return, stride %2d ints           : %7.1f us\n", stride, us);
        us = timeit([&] { hipLaunchKernelGGL((k_atomics<__HIP_MEMORY_SCOPE_AGENT, true>), g, b, 0, 0, n, per, tiles, stride, 1, c, sink); });
        printf("agent scope, returning, stride %2d ints           : %7.1f us\n", stride, us);
    }
    for (int stride : {1, 16}) {
        float us = timeit([&] { hipLaunchKernelGGL((k_atomics<__HIP_MEMORY_SCOPE_WORKGROUP, false>), g, b, 0, 0, n, per, tiles, stride, 8, c, sink); });
        printf("workgroup scope, 8 XCC-private copies, stride %2d : %7.1f us\n", stride, us);
        us = timeit([&] { hipLaunchKernelGGL((k_atomics<__HIP_MEMORY_SCOPE_WORKGROUP, true>), g, b, 0, 0, n, per, tiles, stride, 8, c, sink); });
        printf("  .. returning                                    : %7.1f us\n", us);
        us = timeit([&] { hipLaunchKernelGGL((k_atomics<__HIP_MEMORY_SCOPE_AGENT, false>), g, b, 0, 0, n, per, tiles, stride, 8, c, sink); });
        printf("agent scope, 8 XCC-private copies, stride %2d     : %7.1f us\n", stride, us);
    }
    for (int blocks : {256, 1024, 2048}) {
        float us = timeit([&] { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), tiles * sizeof(int), 0, n, per, tiles, c); });
        printf("LDS-privatised, %4d workgroups                   : %7.1f us\n", blocks, us);
    }
    // larger table (gradient-scatter-like): 36 MB of floats, random addresses
    {
        const int big = 9000000;
        int *cb; CK(hipMalloc(&cb, (size_t)big * 4));
        float us = timeit([&] { hipLaunchKernelGGL((k_atomics<__HIP_MEMORY_SCOPE_AGENT, false>), dim3((2000000 + 255) / 256), b, 0, 0, 2000000, 9, big, 1, 1, cb, sink); });
        printf("agent scope, 18 M atomics over a 36 MB table      : %7.1f us\n", us);
    }
    {
        const int N = 1000000, waves = 8160, epw = 254;  // 2.07 M entries
        float *buf; CK(hipMalloc(&buf, (size_t)N * 32 * 4 + 4096));
        CK(hipMemset(buf, 0, (size_t)N * 32 * 4));
        float *a0 = buf, *a1 = a0 + 2 * N, *a2 = a1 + 3 * N, *a3 = a2 + 3 * N;
        float us = timeit([&] { hipLaunchKernelGGL(k_scatter9<0>, dim3(waves / 4), dim3(256), 0, 0, epw, N, 0, a0, a1, a2, a3); });
        printf("9-lane scatter, 4 separate arrays (2.07 M entries) : %7.1f us\n", us);
        for (int rec : {9, 12, 16, 32}) {
            us = timeit([&] { hipLaunchKernelGGL(k_scatter9<1>, dim3(waves / 4), dim3(256), 0, 0, epw, N, rec, a0, a1, a2, a3); });
            printf("9-lane scatter, one %2d-float record per Gaussian   : %7.1f us\n", rec, us);
        }
    }
    return 0;
}
