// Micro-benchmarks that measure this GPU's own ceilings for the three
// resources the engines can be bound by: FP64 vector FMA issue, LDS
// read+write streaming, and HBM copy.  Built as libvqapeaks.so; bench.py
// prints the measured figures beside the vendor ones.  Measurement tooling,
// not part of the evaluation path.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

// 16 independent FMA chains per lane; 2 flops per FMA.
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a, double b) {
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678) out[0] = s;  // keeps the chains alive, never true in practice
}

// every lane streams 16-byte reads and writes over a 64 KB LDS buffer
__global__ __launch_bounds__(256) void k_lds(double* out, int iters) {
    extern __shared__ double2 buf[];
    const int n = 4096;  // 64 KB of double2
    for (int i = threadIdx.x; i < n; i += 256) buf[i] = make_double2(i, -i);
    __syncthreads();
    double2 acc = make_double2(0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int i = (threadIdx.x + 256 * k) & (n - 1);
            double2 v = buf[i];
            acc.x += v.x;
            acc.y += v.y;
            buf[i] = make_double2(v.y, v.x);
        }
        __syncthreads();
    }
    if (acc.x == 12345.678) out[0] = acc.y;
}

// Copy of n 16-byte words: a workgroup takes a contiguous tile of 256 * U words and every lane
// issues its U loads before its first store (U requests in flight per lane, each wave-instruction a
// contiguous 1 KiB).  NT: non-temporal loads and stores (a stream that is read and written once).
typedef float float4v __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const float4v* __restrict__ src, float4v* __restrict__ dst, size_t n) {
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    float4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i);
            else dst[i] = v[u];
        }
    }
}

template <class F>
int timed(F launch, int reps, double* ms) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
    launch();
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return -1;
    float t = 0;
    hipEventElapsedTime(&t, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *ms = t / reps;
    return 0;
}

}  // namespace

extern "C" {

// FP64 vector FMA throughput in TFLOP/s (2 flops per FMA).
int vqa_peak_fp64_fma(int device, double* tflops) {
    if (hipSetDevice(device) != hipSuccess) return -1;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return -1;
    double* d = nullptr;
    if (hipMalloc(&d, 64) != hipSuccess) return -1;
    const int iters = 4096, blocks = p.multiProcessorCount * 16;
    double ms = 0;
    int rc = timed([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9); }, 3, &ms);
    hipFree(d);
    if (rc) return rc;
    *tflops = 2.0 * 16.0 * iters * 256.0 * blocks / (ms * 1e-3) / 1e12;
    return 0;
}

// LDS streaming bandwidth (bytes read + bytes written) in GB/s, all CUs busy.
int vqa_peak_lds(int device, double* gbs) {
    if (hipSetDevice(device) != hipSuccess) return -1;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return -1;
    double* d = nullptr;
    if (hipMalloc(&d, 64) != hipSuccess) return -1;
    const int iters = 512, blocks = p.multiProcessorCount * 2;
    double ms = 0;
    int rc = timed([&] { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 65536, 0, d, iters); }, 3, &ms);
    hipFree(d);
    if (rc) return rc;
    *gbs = 2.0 * 16.0 * 16.0 * iters * 256.0 * blocks / (ms * 1e-3) / 1e9;
    return 0;
}

// HBM copy bandwidth (bytes read + bytes written) in GB/s on a 2 x 2 GiB pair (far beyond the
// 256 MiB Infinity Cache): the best of a few shapes of the same copy; variant[] (may be null)
// receives the rate of each: {U=4, U=8, U=4 non-temporal, U=8 non-temporal}.
int vqa_peak_hbm_copy_variants(int device, double* gbs, double* variant) {
    if (hipSetDevice(device) != hipSuccess) return -1;
    const size_t bytes = (size_t)2 << 30, n = bytes / sizeof(float4v);
    float4v *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess) return -1;
    if (hipMalloc(&b, bytes) != hipSuccess) { hipFree(a); return -1; }
    hipMemset(a, 1, bytes);
    double best = 0.0;
    int rc = 0;
    for (int k = 0; k < 4 && !rc; ++k) {
        const int U = (k & 1) ? 8 : 4;
        const unsigned blocks = (unsigned)((n + 256 * (size_t)U - 1) / (256 * (size_t)U));
        double ms = 0;
        rc = timed([&] {
            if (k == 0) hipLaunchKernelGGL((k_copy<4, false>), dim3(blocks), dim3(256), 0, 0, a, b, n);
            else if (k == 1) hipLaunchKernelGGL((k_copy<8, false>), dim3(blocks), dim3(256), 0, 0, a, b, n);
            else if (k == 2) hipLaunchKernelGGL((k_copy<4, true>), dim3(blocks), dim3(256), 0, 0, a, b, n);
            else hipLaunchKernelGGL((k_copy<8, true>), dim3(blocks), dim3(256), 0, 0, a, b, n);
        }, 5, &ms);
        const double r = 2.0 * bytes / (ms * 1e-3) / 1e9;
        if (variant) variant[k] = r;
        if (r > best) best = r;
    }
    hipFree(a);
    hipFree(b);
    if (rc) return rc;
    *gbs = best;
    return 0;
}
int vqa_peak_hbm_copy(int device, double* gbs) { return vqa_peak_hbm_copy_variants(device, gbs, nullptr); }

}  // extern "C"
