// Shader clock under load: clock64() (s_memtime, shader cycles) against wall_clock64() (constant
// 100 MHz) while every SIMD runs 8 waves of fp64 FMA / fp32 multiply chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int KIND>
__global__ void __launch_bounds__(64) k(int iters, unsigned long long *out) {
    double d[8]; float a[8];
    for (int i = 0; i < 8; i++) { d[i] = 1.0 + i + threadIdx.x; a[i] = (float)d[i]; }
    const double dm = 1.0000000001; const float m = 1.0000001f;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dm));
                else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    double s = 0; for (int i = 0; i < 8; i++) s += d[i] + a[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (unsigned long long)s; }
}
int main() {
    unsigned long long *out, h[3];
    CK(hipMalloc(&out, 64));
    for (int kind = 0; kind < 2; kind++)
        for (int rep = 0; rep < 3; rep++) {
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(8192), dim3(64), 0, 0, 2000, out);
            else hipLaunchKernelGGL(k<1>, dim3(8192), dim3(64), 0, 0, 2000, out);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
            printf("%s: %llu shader cycles in %llu x 10 ns -> %.0f MHz; %.2f cycles per instruction of this wave (8 waves/SIMD)\n",
                   kind ? "v_mul_f32" : "v_fma_f64", h[0], h[1], (double)h[0] / (h[1] * 0.01), (double)h[0] / (2000.0 * 64));
        }
    return 0;
}
