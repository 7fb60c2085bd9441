// Dependent-issue latency of v_fma_f64 (and v_mul_f64 / v_mfma_f64_4x4x4) on one wave per SIMD:
// K independent chains interleaved, N links each; prints cycles per instruction for K = 1, 2, 4, 8, 16.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/fma_latency.hip -o tools/micro/fma_latency
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int K>
__global__ void k_chain(double* out, double m, double c, int n, long long* cyc) {
    double v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = threadIdx.x * 1e-3 + k;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[k]) : "v"(m), "v"(c));
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int K>
__global__ void k_mfma(double* out, double m, int n, long long* cyc) {
    double v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = threadIdx.x * 1e-3 + k;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(m, v[k], 0.0, 0, 0, 0);
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int K>
void run(double* d_out, long long* d_cyc, int waves) {
    const int n = 20000;
    hipLaunchKernelGGL(k_chain<K>, dim3(1), dim3(64 * waves), 0, 0, d_out, 0.999999, 1e-9, n, d_cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_chain<K>, dim3(1), dim3(64 * waves), 0, 0, d_out, 0.999999, 1e-9, n, d_cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    printf("fma  chains %2d, %d waves in the workgroup: %.2f ns per instruction per wave (%.1f cycles at 2.4 GHz), clock64 units per instruction %.2f\n",
           K, waves, ms * 1e6 / ((double)n * K), ms * 1e6 / ((double)n * K) * 2.4, (double)cyc / ((double)n * K));
}
template <int K>
void run_mfma(double* d_out, long long* d_cyc) {
    const int n = 20000;
    hipLaunchKernelGGL(k_mfma<K>, dim3(1), dim3(64), 0, 0, d_out, 0.999999, n, d_cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<K>, dim3(1), dim3(64), 0, 0, d_out, 0.999999, n, d_cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mfma chains %2d: %.2f ns per instruction (%.1f cycles at 2.4 GHz)\n", K, ms * 1e6 / ((double)n * K), ms * 1e6 / ((double)n * K) * 2.4);
}

int main() {
    double* d_out; long long* d_cyc;
    hipMalloc(&d_out, 1 << 20); hipMalloc(&d_cyc, 8);
    run<1>(d_out, d_cyc, 1); run<2>(d_out, d_cyc, 1); run<4>(d_out, d_cyc, 1); run<8>(d_out, d_cyc, 1); run<16>(d_out, d_cyc, 1);
    run<1>(d_out, d_cyc, 4); run<4>(d_out, d_cyc, 4); run<8>(d_out, d_cyc, 8); run<16>(d_out, d_cyc, 12);
    run_mfma<1>(d_out, d_cyc); run_mfma<2>(d_out, d_cyc); run_mfma<4>(d_out, d_cyc); run_mfma<8>(d_out, d_cyc);
    return 0;
}
