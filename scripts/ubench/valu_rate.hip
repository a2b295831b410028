// Micro-benchmark: issue rate of the VALU instruction classes the compositing kernels use.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
// Each wave runs ITER x 64 independent instructions of one kind (8 accumulator chains, so latency
// is hidden inside one wave); 1024 SIMDs x `waves` waves.  Reported: cycles per wave-instruction
// per SIMD at the measured clock estimate (2.4 GHz nominal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(64) k(int iters, float *out, float seed) {
    float a[8]; f2 p[8]; double d[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i] = (f2)(a[i]); d[i] = a[i]; }
    unsigned long long msk = 0x5555aaaa5555aaaaull ^ (unsigned long long)iters;
    const float m = 1.0000001f; const f2 pm = (f2)(m); const double dm = 1.0000000001;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
                if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dm));
                if (KIND == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
                if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
                if (KIND == 7) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
                if (KIND == 8) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
                if (KIND == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(pm));
                if (KIND == 10) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
                if (KIND == 11) asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(a[i]), "v"(m) : "vcc");
                if (KIND == 12) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 13) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
                if (KIND == 15) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(msk));
                if (KIND == 16) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(m), "v"(seed));
                if (KIND == 17) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(m));
                if (KIND == 18) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 19) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 20) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 21) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (KIND == 22) asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(msk) : "v"(a[i]), "v"(m));
                if (KIND == 23) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 24) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "s"(seed));
                if (KIND == 25) asm volatile("v_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
                if (KIND == 26) asm volatile("v_cmp_le_f32_e64 %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(msk));
                if (KIND == 27) asm volatile("v_cmp_le_f32 vcc, %0, %1\n\ts_nop 3\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
                if (KIND == 14) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)d[i];
    if (s == 12345.678f) out[0] = s;
}


// MIXED streams in the class proportions of the two compositing kernels' loop bodies (scripts/issue_roofline.py):
// do the per-class prices add up when the classes are interleaved?  MIX 0 (forward): per 16 instructions 7 plain,
// 3 compares, 3 with an SGPR source, 2 fp64, 1 v_cndmask (SGPR mask); MIX 1 (backward): per 32: 19 plain, 4 compares,
// 3 SGPR source, 3 fp64, 1 v_cndmask, 1 DPP, 1 transcendental.
template <int MIX>
__global__ void __launch_bounds__(64) kmix(int iters, float *out, float seed) {
    float a[8]; double d[4];
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x;
    for (int i = 0; i < 4; i++) d[i] = a[i];
    unsigned long long msk = 0x5555aaaa5555aaaaull ^ (unsigned long long)iters;
    const float m = 1.0000001f; const double dm = 1.0000000001;
#define PL(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m))
#define CM(i) asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(msk) : "v"(a[i]), "v"(m))
#define SG(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "s"(seed))
#define FD(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dm))
#define CN(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(msk))
#define DP(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]))
#define TR(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]))
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < (MIX == 0 ? 4 : 2); r++) {
            if (MIX == 0) {
                PL(0); CM(1); SG(2); PL(3); FD(0); PL(4); CM(5); SG(6); PL(7); FD(1); PL(1); CM(2); SG(3); PL(5); CN(6); PL(0);
            } else {
                PL(0); CM(1); PL(2); SG(3); PL(4); FD(0); PL(5); PL(6); CM(7); PL(1); SG(2); PL(3); FD(1); PL(4); PL(5); CN(6);
                PL(7); CM(0); PL(1); DP(2); PL(3); FD(2); PL(4); PL(5); CM(6); PL(7); SG(0); PL(1); TR(2); PL(3); PL(4); PL(6);
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i]; for (int i = 0; i < 4; i++) s += (float)d[i];
    if (s == 12345.678f) out[0] = s + (float)msk;
}

template <int MIX> void runmix(const char *name, int waves_per_simd) {
    const int iters = 200;
    float *out; CK(hipMalloc(&out, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    dim3 g(1024 * waves_per_simd), blk(64);
    hipLaunchKernelGGL(kmix<MIX>, g, blk, 0, 0, iters, out, 1.0f); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kmix<MIX>, g, blk, 0, 0, iters, out, 1.0f);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double insts_per_simd = (double)iters * 64 * waves_per_simd;
    printf("%-18s %d waves/SIMD: %8.1f us  -> %5.2f cycles / wave-instruction (at 2.4 GHz)\n", name,
           waves_per_simd, ms * 1e3, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

template <int KIND> void run(const char *name, int waves_per_simd) {
    const int iters = 200;
    float *out; CK(hipMalloc(&out, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    dim3 g(1024 * waves_per_simd), blk(64);
    hipLaunchKernelGGL(k<KIND>, g, blk, 0, 0, iters, out, 1.0f); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<KIND>, g, blk, 0, 0, iters, out, 1.0f);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double insts_per_simd = (double)iters * 64 * waves_per_simd;
    printf("%-18s %d waves/SIMD: %8.1f us  -> %5.2f cycles / wave-instruction (at 2.4 GHz)\n", name,
           waves_per_simd, ms * 1e3, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

// usage: valu_rate [waves per SIMD ...]   (default 8; round 6: 4 and 5 — what the compositing kernels run at — under
// rocprofv3 --pmc, scripts/gpu_valu_calib.sh, to read what SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES shows for a
// SATURATED loop of each class)
int main(int argc, char **argv) {
    int ws[8], nw = 0;
    for (int i = 1; i < argc && nw < 8; i++) ws[nw++] = atoi(argv[i]);
    if (nw == 0) ws[nw++] = 8;
    for (int wi = 0; wi < nw; wi++) {
        const int w = ws[wi];
        run<0>("v_mul_f32", w); run<1>("v_pk_mul_f32", w); run<9>("v_pk_fma_f32", w);
        run<2>("v_fma_f64", w); run<3>("v_mul_f64", w); run<10>("v_add_f64", w);
        run<7>("v_cvt_f64_f32", w); run<8>("v_cvt_f32_f64", w);
        run<4>("v_exp_f32", w); run<5>("v_rcp_f32", w); run<6>("v_cndmask_b32", w);
        run<11>("v_cmp_le_f32", w); run<12>("v_med3_f32", w); run<13>("v_permlane32_swap", w);
        run<14>("v_add_f32_dpp", w);
        run<15>("v_cndmask_e64 sgpr", w); run<16>("v_cndmask indep", w); run<17>("v_mov_b32", w);
        run<18>("v_add_u32", w); run<19>("v_and_b32", w); run<20>("v_add_f32", w); run<21>("v_fma_f32", w);
        run<25>("cmp+cndmask vcc (2)", w); run<26>("cmp+cndmask sgpr(2)", w); run<27>("cmp,nop3,cndmask vcc", w);
        run<22>("v_cmp_e64 ->sgpr", w); run<23>("v_max_f32", w); run<24>("v_mul_f32 sgpr src", w);
        runmix<0>("mix forward", w); runmix<1>("mix backward", w);
    }
    return 0;
}
