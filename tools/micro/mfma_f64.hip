// FP64 MFMA as a cross-lane butterfly: operand layout of v_mfma_f64_4x4x4_4b_f64 decoded by
// experiment, its issue rate alone, and beside v_fma_f64 streams of other waves on the same SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_f64 mfma_f64.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// D[l] for A = one-hot at lane la (value 1), B[l] = l + 1: which lane's B value lands where
__global__ void k_layout(double* out) {
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la) {
        const double a = l == la ? 1.0 : 0.0, b = l + 1.0;
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[la * 64 + l] = d;
    }
}

#define REP8(x) x x x x x x x x
// KIND 0: 32 independent MFMAs per iteration; 1: 48 v_fma_f64 (6 x 8 chains); 2: both interleaved in one wave
template <int KIND>
__global__ void k_rate(double* out, int iters, long long* cyc) {
    double v[32];
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3 + i;
    double f[8];
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 1e-4 + i;
    const double a = 0.5, c = 1.0000001, d = 1e-9;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0 || KIND == 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, v[i], 0.0, 0, 0, 0);
        }
        if (KIND == 1 || KIND == 2) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
                asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                             "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                             : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c), "v"(d));
        }
    }
    const long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 32; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 12345.678) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// waves of a workgroup alternate roles: even waves MFMA streams, odd waves FMA streams
__global__ void k_mixed_roles(double* out, int iters, long long* cyc) {
    double v[32];
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3 + i;
    double f[8];
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 1e-4 + i;
    const double a = 0.5, c = 1.0000001, d = 1e-9;
    const int role = (threadIdx.x >> 6) >> 2 & 1;   // waves 0-3 -> SIMD 0-3 first round, waves 4-7 second round
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (role == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, v[i], 0.0, 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 6; ++r)
                asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                             "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                             : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c), "v"(d));
        }
    }
    const long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 32; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 12345.678) out[0] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

int main() {
    double* d; long long* c;
    hipMalloc(&d, 64 * 64 * 8); hipMalloc(&c, 64 * 8);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d);
    std::vector<double> h(64 * 64);
    hipMemcpy(h.data(), d, 64 * 64 * 8, hipMemcpyDeviceToHost);
    // for each A lane la: the (output lane, source B lane) pairs it feeds
    printf("v_mfma_f64_4x4x4_4b_f64: A one-hot at lane la -> D[l] = B[lb]\n");
    for (int la = 0; la < 64; ++la) {
        printf("la=%2d:", la);
        for (int l = 0; l < 64; ++l)
            if (h[la * 64 + l] != 0.0) printf(" D[%d]<-B[%d]", l, (int)h[la * 64 + l] - 1);
        printf("\n");
    }
    const int iters = 2000;
    for (int waves = 1; waves <= 3; ++waves) {
        long long hc[3];
        hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(256 * waves), 0, 0, d, iters, c); hipMemcpy(&hc[0], c, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(256 * waves), 0, 0, d, iters, c); hipMemcpy(&hc[1], c, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_rate<2>, dim3(256), dim3(256 * waves), 0, 0, d, iters, c); hipMemcpy(&hc[2], c, 8, hipMemcpyDeviceToHost);
        printf("%d waves/SIMD: 32 mfma %.1f cycles/iter/wave (%.2f per mfma per SIMD) | 48 fma %.1f (%.2f per fma per SIMD) | both in one wave %.1f\n", waves,
               (double)hc[0] / iters, (double)hc[0] / iters / 32 / waves, (double)hc[1] / iters, (double)hc[1] / iters / 48 / waves, (double)hc[2] / iters);
    }
    long long hc[8];
    hipLaunchKernelGGL(k_mixed_roles, dim3(256), dim3(512), 0, 0, d, iters, c);
    hipMemcpy(hc, c, 64, hipMemcpyDeviceToHost);
    printf("2 waves/SIMD, one streams 32 mfma, the other 48 fma per iteration: mfma wave %.1f cycles/iter, fma wave %.1f cycles/iter\n",
           (double)hc[0] / iters, (double)hc[4] / iters);
    // wall-clock rates (the in-kernel counter does not show contention reliably)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto wall = [&](auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return (double)ms;
    };
    const int it2 = 20000;
    for (int waves = 1; waves <= 4; ++waves) {
        const double ta = wall([&] { hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(256 * waves), 0, 0, d, it2, c); });
        const double tb = wall([&] { hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(256 * waves), 0, 0, d, it2, c); });
        printf("wall, %d waves/SIMD: 32 mfma/iter %.3f ms (%.2f cycles per mfma per SIMD at 2.4 GHz) | 48 fma/iter %.3f ms (%.2f per fma per SIMD)\n", waves,
               ta, ta * 1e-3 * 2.4e9 / it2 / 32 / waves, tb, tb * 1e-3 * 2.4e9 / it2 / 48 / waves);
    }
    const double tc = wall([&] { hipLaunchKernelGGL(k_mixed_roles, dim3(256), dim3(512), 0, 0, d, it2, c); });
    printf("wall, 2 waves/SIMD with roles (one 32 mfma/iter, one 48 fma/iter): %.3f ms\n", tc);
    return 0;
}
