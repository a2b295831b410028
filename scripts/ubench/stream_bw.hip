// Achievable HBM streaming bandwidth on MI355X at the footprints of the per-Gaussian kernels
// (what k_sh_forward_fused16_quad / k_project_pack / k_gaussian_backward can be held
// against).  Read-only, write-only and copy, 16 B per lane, one pass over the buffer per launch;
// buffers of 216 MB (the SH coefficients of 1 M Gaussians) and 1.5 GB (beyond the 256 MiB Infinity
// Cache), each timed after a pass over a DIFFERENT 1.5 GB buffer so that nothing is resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int UNROLL>
__global__ void __launch_bounds__(256) k_read(const float4 *__restrict__ p, size_t n, float *out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) v[k] = p[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    for (; i < n; i += stride) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(float4 *__restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void __launch_bounds__(256) k_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b[i] = a[i];
}

int main() {
    const size_t big = (size_t)1536 << 20;
    float4 *A, *B, *F; float *out;
    CK(hipMalloc(&A, big)); CK(hipMalloc(&B, big)); CK(hipMalloc(&F, big)); CK(hipMalloc(&out, 4));
    CK(hipMemset(A, 0, big)); CK(hipMemset(B, 0, big)); CK(hipMemset(F, 0, big));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sizes[2] = {(size_t)216000000, big};
    printf("{");
    bool first = true;
    for (int si = 0; si < 2; si++) {
        const size_t n = sizes[si] / 16;
        for (int grid : {1024, 4096, 16384, 65536}) {
            for (int mode = 0; mode < 5; mode++) {
                float best = 1e30f;
                for (int rep = 0; rep < 5; rep++) {
                    hipLaunchKernelGGL(k_read<1>, dim3(8192), dim3(256), 0, 0, F, big / 16, out);  // flush
                    CK(hipEventRecord(e0));
                    switch (mode) {
                    case 0: hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(256), 0, 0, A, n, out); break;
                    case 1: hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, A, n, out); break;
                    case 2: hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, B, n); break;
                    case 3: hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, A, B, n); break;
                    case 4: hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, 0, A, n, out); break;
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                const char *names[5] = {"read", "read_x4", "write", "copy", "read_x8"};
                const double bytes = (double)n * 16 * (mode == 3 ? 2 : 1);
                printf("%s\n \"%s_%zuMB_grid%d\": {\"us\": %.1f, \"TBps\": %.2f}", first ? "" : ",", names[mode],
                       sizes[si] / 1000000, grid, best * 1e3, bytes / (best * 1e-3) / 1e12);
                first = false;
            }
        }
    }
    printf("\n}\n");
    return 0;
}
