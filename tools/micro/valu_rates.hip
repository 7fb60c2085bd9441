// Issue cost of the instruction kinds the register engine is made of, measured with
// every SIMD full (8 waves) and with one wave per SIMD: cycles per wave-instruction.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ void k(double* out, int iters, long long* cyc) {
    double a[8];
    int b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3 + i; b[i] = threadIdx.x + i; }
    const double c = 1.0000001, d = 1e-9;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {  // v_fma_f64, 8 independent chains
            REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                               "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c), "v"(d));)
        } else if (KIND == 1) {  // v_mov_b32_dpp quad_perm
            REP16(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                               : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));)
        } else if (KIND == 2) {  // v_mov_b32_dpp row_ror:8
            REP16(asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %2, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %4, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %6, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %0 row_ror:8 row_mask:0xf bank_mask:0xf"
                               : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));)
        } else if (KIND == 3) {  // plain v_xor_b32
            REP16(asm volatile("v_xor_b32 %0, %1, %0\n\tv_xor_b32 %1, %2, %1\n\tv_xor_b32 %2, %3, %2\n\tv_xor_b32 %3, %4, %3\n\t"
                               "v_xor_b32 %4, %5, %4\n\tv_xor_b32 %5, %6, %5\n\tv_xor_b32 %6, %7, %6\n\tv_xor_b32 %7, %0, %7"
                               : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));)
        } else if (KIND == 4) {  // the lane-rotation mix: 4 dpp + 2 fma
            REP16(asm volatile("v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %2, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_fma_f64 %8, %10, %8, %10\n\tv_fma_f64 %9, %10, %9, %10\n\ts_nop 0\n\ts_nop 0"
                               : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(a[0]), "+v"(a[1]) : "v"(c));)
        } else if (KIND == 5) {  // v_mov_b64
            REP16(asm volatile("v_mov_b64 %0, %1\n\tv_mov_b64 %1, %2\n\tv_mov_b64 %2, %3\n\tv_mov_b64 %3, %4\n\tv_mov_b64 %4, %5\n\tv_mov_b64 %5, %6\n\tv_mov_b64 %6, %7\n\tv_mov_b64 %7, %0"
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));)
        } else if (KIND == 6) {  // v_mul_f64
            REP16(asm volatile("v_mul_f64 %0, %0, %8\n\tv_mul_f64 %1, %1, %8\n\tv_mul_f64 %2, %2, %8\n\tv_mul_f64 %3, %3, %8\n\t"
                               "v_mul_f64 %4, %4, %8\n\tv_mul_f64 %5, %5, %8\n\tv_mul_f64 %6, %6, %8\n\tv_mul_f64 %7, %7, %8"
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));)
        } else if (KIND == 7) {  // v_permlane32_swap
            REP16(asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                               "v_permlane32_swap_b32 %1, %2\n\tv_permlane32_swap_b32 %3, %4\n\tv_permlane32_swap_b32 %5, %6\n\tv_permlane32_swap_b32 %7, %0"
                               : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));)
        }
    }
    long long t1 = clock64();
    double s = 0; int q = 0;
    for (int i = 0; i < 8; ++i) { s += a[i]; q ^= b[i]; }
    if (s == 12345.678 && q == 77) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_iter) {
    double* out; long long* cyc;
    hipMalloc(&out, 64); hipMalloc(&cyc, 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int iters = 2000;
    for (int waves = 1; waves <= 8; waves *= 2) {
        // `waves` waves per SIMD: blocks of 256 threads (one wave per SIMD each), `waves` blocks per CU
        hipLaunchKernelGGL(k<KIND>, dim3(p.multiProcessorCount * waves), dim3(256), 0, 0, out, iters, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(p.multiProcessorCount * waves), dim3(256), 0, 0, out, iters, cyc);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double winst = (double)iters * 16 * per_iter * 4.0 * p.multiProcessorCount * waves;   // wave-instructions
        printf("   wall %.3f ms -> %.2f core cycles (2.4 GHz) per wave-instruction per SIMD; ", ms,
               ms * 1e-3 * 2.4e9 / (winst / (4.0 * p.multiProcessorCount)));
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-28s waves/SIMD %d: %.2f cycles per instruction per wave, %.2f per SIMD\n", name, waves,
               (double)c / ((double)iters * 16 * per_iter), (double)c / ((double)iters * 16 * per_iter * waves));
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_fma_f64", 8);
    run<6>("v_mul_f64", 8);
    run<1>("v_mov_b32_dpp quad_perm", 8);
    run<2>("v_mov_b32_dpp row_ror:8", 8);
    run<3>("v_xor_b32", 8);
    run<5>("v_mov_b64", 8);
    run<7>("v_permlane32_swap_b32", 8);
    run<4>("4 dpp + 2 fma_f64 (+2 nop)", 6);
    return 0;
}
