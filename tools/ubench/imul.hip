// Micro-benchmark: issue rate of the integer multiplies behind Philox4x32 (v_mad_u64_u32, v_mul_hi_u32, v_mul_lo_u32)
// against v_fma_f64 and v_xor_b32, one to eight waves per SIMD.  Diagnostic only: it explains what mc_sample_kernel
// waits on (SQ_WAIT_INST_ANY: the SIMD is busy with a multi-pass instruction).
//   hipcc -O3 --offload-arch=gfx950 imul.hip -o imul && ./imul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void rate(unsigned long long* o, long long* cyc, int iters, unsigned seed) {
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
    unsigned long long p0 = 1, p1 = 2, p2 = 3, p3 = 4, p4 = 5, p5 = 6, p6 = 7, p7 = 8;
    double d0 = 1, d1 = 2, d2 = 3, d3 = 4, d4 = 5, d5 = 6, d6 = 7, d7 = 8, c = 1.0 + 1e-9 * threadIdx.x;
    const unsigned M = 0xD2511F53u;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // v_mad_u64_u32: 8 independent chains
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %10, %1\n v_mad_u64_u32 %2, vcc, %8, %11, %2\n v_mad_u64_u32 %3, vcc, %8, %12, %3\n"
                         "v_mad_u64_u32 %4, vcc, %8, %13, %4\n v_mad_u64_u32 %5, vcc, %8, %14, %5\n v_mad_u64_u32 %6, vcc, %8, %15, %6\n v_mad_u64_u32 %7, vcc, %8, %16, %7\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                         : "v"(M), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
        } else if (MODE == 1) {   // v_mul_hi_u32
            asm volatile("v_mul_hi_u32 %0, %8, %0\n v_mul_hi_u32 %1, %8, %1\n v_mul_hi_u32 %2, %8, %2\n v_mul_hi_u32 %3, %8, %3\n"
                         "v_mul_hi_u32 %4, %8, %4\n v_mul_hi_u32 %5, %8, %5\n v_mul_hi_u32 %6, %8, %6\n v_mul_hi_u32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(M));
        } else if (MODE == 2) {   // v_mul_lo_u32
            asm volatile("v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %1, %8, %1\n v_mul_lo_u32 %2, %8, %2\n v_mul_lo_u32 %3, %8, %3\n"
                         "v_mul_lo_u32 %4, %8, %4\n v_mul_lo_u32 %5, %8, %5\n v_mul_lo_u32 %6, %8, %6\n v_mul_lo_u32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(M));
        } else if (MODE == 3) {   // v_fma_f64
            asm volatile("v_fma_f64 %0, %8, %8, %0\n v_fma_f64 %1, %8, %8, %1\n v_fma_f64 %2, %8, %8, %2\n v_fma_f64 %3, %8, %8, %3\n"
                         "v_fma_f64 %4, %8, %8, %4\n v_fma_f64 %5, %8, %8, %5\n v_fma_f64 %6, %8, %8, %6\n v_fma_f64 %7, %8, %8, %7\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(c));
        } else if (MODE == 4) {   // v_xor_b32 (full-rate reference)
            asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n"
                         "v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(M));
        } else if (MODE == 6) {   // v_pk_fma_f32 (two fp32 fma per lane)
            asm volatile("v_pk_fma_f32 %0, %8, %8, %0\n v_pk_fma_f32 %1, %8, %8, %1\n v_pk_fma_f32 %2, %8, %8, %2\n v_pk_fma_f32 %3, %8, %8, %3\n"
                         "v_pk_fma_f32 %4, %8, %8, %4\n v_pk_fma_f32 %5, %8, %8, %5\n v_pk_fma_f32 %6, %8, %8, %6\n v_pk_fma_f32 %7, %8, %8, %7\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(c));
        } else if (MODE == 7) {   // v_fma_f32
            asm volatile("v_fma_f32 %0, %8, %8, %0\n v_fma_f32 %1, %8, %8, %1\n v_fma_f32 %2, %8, %8, %2\n v_fma_f32 %3, %8, %8, %3\n"
                         "v_fma_f32 %4, %8, %8, %4\n v_fma_f32 %5, %8, %8, %5\n v_fma_f32 %6, %8, %8, %6\n v_fma_f32 %7, %8, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(M));
        } else if (MODE == 8) {   // v_min_f64
            asm volatile("v_min_f64 %0, %8, %0\n v_min_f64 %1, %8, %1\n v_min_f64 %2, %8, %2\n v_min_f64 %3, %8, %3\n"
                         "v_min_f64 %4, %8, %4\n v_min_f64 %5, %8, %5\n v_min_f64 %6, %8, %6\n v_min_f64 %7, %8, %7\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(c));
        } else if (MODE == 5) {   // v_mul_u32_u24
            asm volatile("v_mul_u32_u24 %0, %8, %0\n v_mul_u32_u24 %1, %8, %1\n v_mul_u32_u24 %2, %8, %2\n v_mul_u32_u24 %3, %8, %3\n"
                         "v_mul_u32_u24 %4, %8, %4\n v_mul_u32_u24 %5, %8, %5\n v_mul_u32_u24 %6, %8, %6\n v_mul_u32_u24 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(M));
        }
    }
    long long t1 = clock64();
    o[blockIdx.x * blockDim.x + threadIdx.x] = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7 + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 +
                                               (unsigned long long)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

int main() {
    const int iters = 4000;
    const char* names[] = {"v_mad_u64_u32", "v_mul_hi_u32", "v_mul_lo_u32", "v_fma_f64", "v_xor_b32", "v_mul_u32_u24", "v_pk_fma_f32", "v_fma_f32", "v_min_f64"};
    for (int wps : {1, 2, 4, 8}) {   // waves per SIMD: blocks of 256 threads (one wave per SIMD each), wps blocks per CU
        const int blocks = 256 * wps, threads = 256, nw = blocks * threads / 64;
        unsigned long long* o; long long* cyc;
        CK(hipMalloc(&o, (size_t)blocks * threads * 8)); CK(hipMalloc(&cyc, (size_t)nw * 8));
        for (int mode = 0; mode < 9; ++mode) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                switch (mode) {
                    case 0: hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(threads), 0, 0, o, cyc, iters, 1u); break;
                    case 1: hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(threads), 0, 0, o, cyc, iters, 1u); break;
                    case 2: hipLaunchKernelGGL(rate<2>, dim3(blocks), dim3(threads), 0, 0, o, cyc, iters, 1u); break;
                    case 3: hipLaunchKernelGGL(rate<3>, dim3(blocks), dim3(threads), 0, 0, o, cyc, iters, 1u); break;
                    case 4: hipLaunchKernelGGL(rate<4>, dim3(blocks), dim3(threads), 0, 0, o, cyc, iters, 1u); break;
                    case 5: hipLaunchKernelGGL(rate<5>, dim3(blocks), dim3(threads), 0, 0, o, cyc, iters, 1u); break;
                    case 6: hipLaunchKernelGGL(rate<6>, dim3(blocks), dim3(threads), 0, 0, o, cyc, iters, 1u); break;
                    case 7: hipLaunchKernelGGL(rate<7>, dim3(blocks), dim3(threads), 0, 0, o, cyc, iters, 1u); break;
                    default: hipLaunchKernelGGL(rate<8>, dim3(blocks), dim3(threads), 0, 0, o, cyc, iters, 1u); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            // SIMD-cycles per wave-instruction at 2.4 GHz: kernel time x clock / (instructions issued per SIMD)
            const double per_simd = (double)iters * 8 * wps;
            printf("waves/SIMD %d  %-14s %8.3f ms  %6.2f SIMD-cycles per wave-instruction (at 2.4 GHz)\n", wps, names[mode], ms, ms * 1e-3 * 2.4e9 / per_simd);
        }
        CK(hipFree(o)); CK(hipFree(cyc));
    }
    return 0;
}
