// Do 16-byte global loads that are only 4-byte aligned run at full rate on gfx950?  (The SH rows of
// OpenSplat's features_rest are 180 bytes: three rows out of four start off a 16-byte boundary.)
//   mode 0: lane i reads 16 B at 16 i            (aligned, contiguous)
//   mode 1: lane i reads 16 B at 16 i + 4        (same stream, every load misaligned by 4 bytes)
//   mode 2: lane i reads 3 x 16 B at 48 i        (aligned start, lane stride 48 B: the quad kernel's shape)
//   mode 3: lane i reads 3 x 16 B at 48 i + 4    (the same, misaligned)
//   mode 4: 180-B rows, four lanes per row, 3 x 16 B each at row + 48 q (- overlap): the real pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(const float *__restrict__ p, size_t nfloats, float *out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    if (MODE <= 1) {
        const size_t o = 4 * t + (MODE == 1 ? 1 : 0);
        if (o + 4 <= nfloats) { f4u v = *reinterpret_cast<const f4u *>(p + o); acc = v.x + v.y + v.z + v.w; }
    } else if (MODE <= 3) {
        const size_t o = 12 * t + (MODE == 3 ? 1 : 0);
        if (o + 12 <= nfloats) {
            f4u a = *reinterpret_cast<const f4u *>(p + o), b = *reinterpret_cast<const f4u *>(p + o + 4),
                c = *reinterpret_cast<const f4u *>(p + o + 8);
            acc = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w;
        }
    } else {
        const size_t g = t >> 2; const int q = (int)(t & 3);
        const size_t o = 45 * g + 12 * q;
        if (45 * (g + 1) <= nfloats) {
            f4u a = *reinterpret_cast<const f4u *>(p + o), b = *reinterpret_cast<const f4u *>(p + o + 4),
                c = *reinterpret_cast<const f4u *>(p + o + (q == 3 ? 5 : 8));
            acc = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE> static void run(const float *A, const float *F, size_t nfloats, size_t big_floats, float *out, const char *name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    size_t threads = MODE <= 1 ? nfloats / 4 : MODE <= 3 ? nfloats / 12 : nfloats / 45 * 4;
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL(k<0>, dim3((unsigned)((big_floats / 4 + 255) / 256)), dim3(256), 0, 0, F, big_floats, out);  // flush caches
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, A, nfloats, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf(" \"%s\": {\"us\": %.1f, \"TBps\": %.2f},\n", name, best * 1e3, (double)nfloats * 4 / (best * 1e-3) / 1e12);
}

int main() {
    const size_t nfloats = 45000000;            // 180 MB: features_rest of 1 M Gaussians
    const size_t big = (size_t)1536 << 20;
    float *A, *F, *out;
    CK(hipMalloc(&A, nfloats * 4 + 64)); CK(hipMalloc(&F, big)); CK(hipMalloc(&out, 4));
    CK(hipMemset(A, 0, nfloats * 4 + 64)); CK(hipMemset(F, 0, big));
    printf("{\n");
    run<0>(A, F, nfloats, big / 4, out, "aligned_contiguous");
    run<1>(A, F, nfloats, big / 4, out, "misaligned_contiguous");
    run<2>(A, F, nfloats, big / 4, out, "aligned_stride48");
    run<3>(A, F, nfloats, big / 4, out, "misaligned_stride48");
    run<4>(A, F, nfloats, big / 4, out, "rows180_quad");
    printf(" \"bytes\": %zu\n}\n", nfloats * 4);
    return 0;
}
