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

__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
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

// HBM copy bandwidth (bytes read + bytes written) in GB/s on a 2 x 1 GiB pair.
int vqa_peak_hbm_copy(int device, double* gbs) {
    if (hipSetDevice(device) != hipSuccess) return -1;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return -1;
    const size_t bytes = (size_t)1 << 30, n = bytes / sizeof(double2);
    double2 *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess) return -1;
    if (hipMalloc(&b, bytes) != hipSuccess) { hipFree(a); return -1; }
    hipMemset(a, 1, bytes);
    double ms = 0;
    int rc = timed([&] { hipLaunchKernelGGL(k_copy, dim3(p.multiProcessorCount * 8), dim3(256), 0, 0, a, b, n); }, 5, &ms);
    hipFree(a);
    hipFree(b);
    if (rc) return rc;
    *gbs = 2.0 * bytes / (ms * 1e-3) / 1e9;
    return 0;
}

}  // extern "C"
