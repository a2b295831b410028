// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the access patterns of the compositing
// kernels: the guide's "x2" rule is calibrated for wide coalesced streaming
// reads only.  Every kernel below moves a KNOWN number of bytes over a working set far larger than
// the 256 MiB Infinity Cache; run under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
// and compare the counters with the "requested" / "line" byte counts this program prints.
//   k_stream_read16     coalesced 16 B / lane                       (the guide's calibrated case)
//   k_gather48_random   one random 48-B record per lane, 3 x dwordx4 (uniform over 1.5 GB)
//   k_gather48_tile     records gathered like a tile list: 64 consecutive list entries point into a
//                       window of ~2000 records (what neighbouring tiles share), windows revisited
//   k_atomic_records    nine-lane float atomics into random 64-B records (the backward's scatter)
//   k_stream_write16    coalesced 16-B stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void __launch_bounds__(256) k_stream_read16(const float4 *__restrict__ p, size_t n, float *out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_stream_write16(float4 *__restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__device__ __forceinline__ float gather48(const float4 *__restrict__ rec, int g) {
    const float4 a = rec[3 * (size_t)g], b = rec[3 * (size_t)g + 1], c = rec[3 * (size_t)g + 2];
    return a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w;
}
__global__ void __launch_bounds__(256) k_gather48_random(const float4 *__restrict__ rec, const int *__restrict__ idx, size_t n, float *out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += gather48(rec, idx[i]);
    if (acc == 123.456f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_gather48_tile(const float4 *__restrict__ rec, const int *__restrict__ idx, size_t n, float *out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += gather48(rec, idx[i]);
    if (acc == 123.456f) out[0] = acc;
}
__global__ void __launch_bounds__(64) k_atomic_records(float *__restrict__ rec, const int *__restrict__ idx, size_t n) {
    // one wave = 7 records x 9 lanes per instruction (63 lanes), like the flush of gradient partials
    const int lane = threadIdx.x, r = lane / 9, c = lane % 9;
    for (size_t i = (size_t)blockIdx.x; i * 7 + r < n; i += gridDim.x)
        if (lane < 63) atomicAdd(rec + 16 * (size_t)(idx[i * 7 + r] & ((8 << 20) - 1)) + c, 1.0f);
}

int main(int argc, char **argv) {
    const size_t nrec = 32u << 20;            // 32 Mi records x 48 B = 1.5 GiB
    const size_t ngather = 16u << 20;         // 16 Mi gathers = 768 MiB requested
    float4 *rec; int *idx_r, *idx_t; float *out; float *arec;
    CK(hipMalloc(&rec, nrec * 48)); CK(hipMalloc(&idx_r, ngather * 4)); CK(hipMalloc(&idx_t, ngather * 4));
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&arec, (size_t)(8u << 20) * 64));  // 8 Mi x 64-B records = 512 MiB
    CK(hipMemset(rec, 0, nrec * 48)); CK(hipMemset(arec, 0, (size_t)(8u << 20) * 64));
    std::vector<int> hr(ngather), ht(ngather);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t i = 0; i < ngather; i++) hr[i] = (int)(rnd() % nrec);
    // tile-like: consecutive groups of 256 gathers draw from a window of 2048 records; the window
    // advances by 512 records per group (neighbouring tiles overlap 75 %), wrapping over the table
    for (size_t i = 0; i < ngather; i++) {
        const size_t grp = i / 256;
        ht[i] = (int)((grp * 512 + rnd() % 2048) % nrec);
    }
    CK(hipMemcpy(idx_r, hr.data(), ngather * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(idx_t, ht.data(), ngather * 4, hipMemcpyHostToDevice));
    // distinct 64-B lines / 128-B lines touched by the random gather (expected values of what must move)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch, double req_bytes, const char *note) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"kernel\": \"%s\", \"requested_bytes\": %.0f, \"ms\": %.4f, \"GBs\": %.1f, \"note\": \"%s\"}\n",
               name, req_bytes, ms, req_bytes / ms / 1e6, note);
    };
    const size_t nstream = nrec * 3;   // float4 elements in the table
    timeit("k_stream_read16", [&] { hipLaunchKernelGGL(k_stream_read16, dim3(4096), dim3(256), 0, 0, rec, nstream, out); },
           (double)nstream * 16, "coalesced 16 B per lane over 1.5 GiB");
    timeit("k_stream_write16", [&] { hipLaunchKernelGGL(k_stream_write16, dim3(4096), dim3(256), 0, 0, rec, nstream); },
           (double)nstream * 16, "coalesced 16-B stores over 1.5 GiB");
    CK(hipMemset(rec, 0, nrec * 48));
    timeit("k_gather48_random", [&] { hipLaunchKernelGGL(k_gather48_random, dim3(4096), dim3(256), 0, 0, rec, idx_r, ngather, out); },
           (double)ngather * 52, "48-B record + 4-B index per gather; a 16-B-aligned 48-B record spans 1.5 64-B lines / 1.25 128-B lines on average");
    timeit("k_gather48_tile", [&] { hipLaunchKernelGGL(k_gather48_tile, dim3(4096), dim3(256), 0, 0, rec, idx_t, ngather, out); },
           (double)ngather * 52, "windowed gather (2048-record windows advancing by 512 per 256 gathers): compulsory traffic = table touched once = 16Mi/256*512*48 B + indices");
    const size_t natom = 14u << 20;
    timeit("k_atomic_records", [&] { hipLaunchKernelGGL(k_atomic_records, dim3(16384), dim3(64), 0, 0, arec, idx_r /* < 32 Mi: fold */, natom); },
           (double)natom * 36, "9 float atomics (36 B) into a random 64-B record; indices folded below");
    return 0;
}
