// streambench.hip -- HBM calibration microbenchmarks for the L-BFGS kernels (dev tool).
//   hipcc -O3 --offload-arch=gfx950 -o streambench tools/streambench.hip && ./streambench
// Measures (fp64, n = 10^7 per vector unless noted):
//   copy / triad            the classic achievable-peak probes
//   multi-stream read-sum   K separate slabs (the ring layout) vs one blocked slab
//                           [chunk][slot][CH] (all K slots of an index range contiguous)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy_k(const double2 *__restrict__ a, double2 *__restrict__ b, long nv) {
    for (long i = (long)blockIdx.x * 256 * 4 + threadIdx.x; i < nv; i += (long)gridDim.x * 256 * 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { long j = i + u * 256; if (j < nv) b[j] = a[j]; }
    }
}
__global__ __launch_bounds__(256) void triad_k(const double2 *__restrict__ a, const double2 *__restrict__ b, double2 *__restrict__ c, double s, long nv) {
    for (long i = (long)blockIdx.x * 256 * 4 + threadIdx.x; i < nv; i += (long)gridDim.x * 256 * 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { long j = i + u * 256; if (j < nv) { double2 x = a[j], y = b[j]; c[j] = make_double2(x.x + s * y.x, x.y + s * y.y); } }
    }
}
// K streams, separate slabs: element e of stream s at base + s*stride + e
template <int U, bool NT>
__global__ __launch_bounds__(256) void multi_read_k(const double2 *__restrict__ base, long stride_v, int K, long nv, double *out) {
    double acc = 0;
    for (long i = (long)blockIdx.x * 256 * U + threadIdx.x; i < nv; i += (long)gridDim.x * 256 * U) {
        for (int s = 0; s < K; ++s) {
            const double2 *p = base + (long)s * stride_v;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                long j = i + u * 256;
                if (j < nv) {
                    double2 v;
                    if (NT) { v.x = __builtin_nontemporal_load(&p[j].x); v.y = __builtin_nontemporal_load(&p[j].y); }
                    else v = p[j];
                    acc += v.x + v.y;
                }
            }
        }
    }
    if (acc == 1.2345e300) out[0] = acc;
}
// K nt-read streams + one written stream (the combine kernel's traffic shape)
template <int U, bool NTSTORE>
__global__ __launch_bounds__(256) void multi_read_write_k(const double2 *__restrict__ base, long stride_v, int K, long nv, double2 *__restrict__ out) {
    typedef double v2 __attribute__((ext_vector_type(2)));
    for (long i = (long)blockIdx.x * 256 * U + threadIdx.x; i < nv; i += (long)gridDim.x * 256 * U) {
        double2 acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = make_double2(0, 0);
#pragma unroll 4
        for (int s = 0; s < K; ++s) {
            const double2 *p = base + (long)s * stride_v;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                long j = i + u * 256;
                if (j < nv) {
                    v2 v = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(&p[j]));
                    acc[u].x = __builtin_fma(1.0001, v.x, acc[u].x);
                    acc[u].y = __builtin_fma(1.0001, v.y, acc[u].y);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long j = i + u * 256;
            if (j < nv) {
                if (NTSTORE) { v2 w; w.x = acc[u].x; w.y = acc[u].y; __builtin_nontemporal_store(w, reinterpret_cast<v2 *>(&out[j])); }
                else out[j] = acc[u];
            }
        }
    }
}

// K nt-read streams + W nt-written streams (the single-pass step's traffic shape: 42 + 7)
template <int U>
__global__ __launch_bounds__(256) void multi_read_multi_write_k(const double2 *__restrict__ base, long stride_v, int K, int W, long nv, double2 *__restrict__ out, long ostride_v) {
    typedef double v2 __attribute__((ext_vector_type(2)));
    for (long i = (long)blockIdx.x * 256 * U + threadIdx.x; i < nv; i += (long)gridDim.x * 256 * U) {
        double2 acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = make_double2(0, 0);
#pragma unroll 4
        for (int s = 0; s < K; ++s) {
            const double2 *p = base + (long)s * stride_v;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                long j = i + u * 256;
                if (j < nv) {
                    v2 v = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(&p[j]));
                    acc[u].x = __builtin_fma(1.0001, v.x, acc[u].x);
                    acc[u].y = __builtin_fma(1.0001, v.y, acc[u].y);
                }
            }
        }
        for (int w = 0; w < W; ++w) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                long j = i + u * 256;
                if (j < nv) { v2 o; o.x = acc[u].x + w; o.y = acc[u].y; __builtin_nontemporal_store(o, reinterpret_cast<v2 *>(&out[(long)w * ostride_v + j])); }
            }
        }
    }
}

// single-pass shape on a BLOCKED history mirror [row][slot][64 vectors]: per wave-row 40 slots read
// (one contiguous 40 KiB), 2 slots written in place, plus 2 separate read streams (x, g) and
// 5 + 2 separate written streams (d, x, g, backups + the canonical copies of the new pair)
__global__ __launch_bounds__(256) void blocked_rw_k(const double2 *__restrict__ hist, int slots, long rows,
                                                    const double2 *__restrict__ xs, long xstride, double2 *__restrict__ out, long ostride, int W) {
    typedef double v2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
        const double2 *h = hist + row * (long)slots * 64 + lane;
        double2 acc = make_double2(0, 0);
#pragma unroll 8
        for (int s = 0; s < 40; ++s) {
            v2 v = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(&h[(long)s * 64]));
            acc.x = __builtin_fma(1.0001, v.x, acc.x); acc.y = __builtin_fma(1.0001, v.y, acc.y);
        }
        for (int s = 0; s < 2; ++s) { double2 v = xs[(long)s * xstride + row * 64 + lane]; acc.x += v.x; acc.y += v.y; }
        v2 o; o.x = acc.x; o.y = acc.y;
        double2 *hw = const_cast<double2 *>(hist) + row * (long)slots * 64 + lane;
        __builtin_nontemporal_store(o, reinterpret_cast<v2 *>(&hw[40L * 64]));
        __builtin_nontemporal_store(o, reinterpret_cast<v2 *>(&hw[41L * 64]));
        for (int w = 0; w < W; ++w) __builtin_nontemporal_store(o, reinterpret_cast<v2 *>(&out[(long)w * ostride + row * 64 + lane]));
    }
}

// blocked layout: [chunk][slot][CHV vectors]; block-iteration handles one chunk of CHV vectors
template <int CHV>
__global__ __launch_bounds__(256) void blocked_read_k(const double2 *__restrict__ base, int K, int slots, long nchunks, double *out) {
    double acc = 0;
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const double2 *p = base + c * (long)slots * CHV;
        for (int s = 0; s < K; ++s) {
#pragma unroll
            for (int u = 0; u < CHV / 256; ++u) { double2 v = p[(long)s * CHV + u * 256 + threadIdx.x]; acc += v.x + v.y; }
        }
    }
    if (acc == 1.2345e300) out[0] = acc;
}

template <typename F> double time_ms(F f, int reps = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char **argv) {
    const long n = 10000000, nv = n / 2;
    const int KMAX = 41;
    double2 *buf, *out2; double *out;
    const long stride_v = nv;   // exactly as the ring: n rounded to 64 elements
    CK(hipMalloc(&buf, sizeof(double2) * stride_v * (KMAX + 1)));
    CK(hipMalloc(&out2, sizeof(double2) * nv));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, sizeof(double2) * stride_v * (KMAX + 1)));
    CK(hipDeviceSynchronize());
    int cus = 256;
    for (int g : {8}) {
        int grid = cus * g;
        double ms = time_ms([&] { hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, buf, out2, nv); });
        printf("copy   grid=%5d  %.1f us  %.0f GB/s (r+w)\n", grid, ms * 1e3, 2.0 * n * 8 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(triad_k, dim3(grid), dim3(256), 0, 0, buf, buf + stride_v, out2, 1.5, nv); });
        printf("triad  grid=%5d  %.1f us  %.0f GB/s (2r+w)\n", grid, ms * 1e3, 3.0 * n * 8 / ms / 1e6);
    }
    for (int K : {41}) {
        for (int g : {4, 8, 16}) {
            int grid = cus * g;
            double ms = time_ms([&] { hipLaunchKernelGGL((multi_read_k<2, false>), dim3(grid), dim3(256), 0, 0, buf, stride_v, K, nv, out); }, 10);
            double ms4 = time_ms([&] { hipLaunchKernelGGL((multi_read_k<4, false>), dim3(grid), dim3(256), 0, 0, buf, stride_v, K, nv, out); }, 10);
            double msn = time_ms([&] { hipLaunchKernelGGL((multi_read_k<2, true>), dim3(grid), dim3(256), 0, 0, buf, stride_v, K, nv, out); }, 10);
            printf("read K=%2d grid=%5d  U2 %.0f GB/s  U4 %.0f GB/s  U2-nt %.0f GB/s\n", K, grid, (double)K * n * 8 / ms / 1e6, (double)K * n * 8 / ms4 / 1e6, (double)K * n * 8 / msn / 1e6);
        }
    }
    for (int g : {4, 8, 16}) {
        int grid = cus * g;
        double m2 = time_ms([&] { hipLaunchKernelGGL((multi_read_write_k<2, false>), dim3(grid), dim3(256), 0, 0, buf, stride_v, 41, nv, out2); }, 10);
        double m4 = time_ms([&] { hipLaunchKernelGGL((multi_read_write_k<4, false>), dim3(grid), dim3(256), 0, 0, buf, stride_v, 41, nv, out2); }, 10);
        double m4n = time_ms([&] { hipLaunchKernelGGL((multi_read_write_k<4, true>), dim3(grid), dim3(256), 0, 0, buf, stride_v, 41, nv, out2); }, 10);
        printf("read41+write1 grid=%5d  U2 %.1f us (%.0f GB/s)  U4 %.1f us (%.0f GB/s)  U4-ntstore %.1f us (%.0f GB/s)\n", grid,
               m2 * 1e3, 42.0 * n * 8 / m2 / 1e6, m4 * 1e3, 42.0 * n * 8 / m4 / 1e6, m4n * 1e3, 42.0 * n * 8 / m4n / 1e6);
    }
    {   // 42 reads + 7 writes (single-pass step)
        double2 *wout; const long ostride = nv + 64;
        CK(hipMalloc(&wout, sizeof(double2) * ostride * 7));
        for (int g : {1, 2, 4, 8}) {
            int grid = cus * g;
            double m1 = time_ms([&] { hipLaunchKernelGGL((multi_read_multi_write_k<1>), dim3(grid), dim3(256), 0, 0, buf, stride_v, 41, 7, nv, wout, ostride); }, 10);
            double m2 = time_ms([&] { hipLaunchKernelGGL((multi_read_multi_write_k<2>), dim3(grid), dim3(256), 0, 0, buf, stride_v, 41, 7, nv, wout, ostride); }, 10);
            double m4 = time_ms([&] { hipLaunchKernelGGL((multi_read_multi_write_k<4>), dim3(grid), dim3(256), 0, 0, buf, stride_v, 41, 7, nv, wout, ostride); }, 10);
            printf("read41+write7 grid=%5d  U1 %.1f us (%.0f GB/s)  U2 %.1f us (%.0f GB/s)  U4 %.1f us (%.0f GB/s)\n", grid,
                   m1 * 1e3, 48.0 * n * 8 / m1 / 1e6, m2 * 1e3, 48.0 * n * 8 / m2 / 1e6, m4 * 1e3, 48.0 * n * 8 / m4 / 1e6);
        }
        CK(hipFree(wout));
    }
    {   // single-pass shape on a blocked mirror
        double2 *wout; const long ostride = nv + 64;
        CK(hipMalloc(&wout, sizeof(double2) * ostride * 7));
        const long rows = nv / 64;
        for (int g : {1, 2, 4}) {
            int grid = cus * g;
            double m5 = time_ms([&] { hipLaunchKernelGGL(blocked_rw_k, dim3(grid), dim3(256), 0, 0, buf, 42, rows, buf, stride_v, wout, ostride, 5); }, 10);
            double m7 = time_ms([&] { hipLaunchKernelGGL(blocked_rw_k, dim3(grid), dim3(256), 0, 0, buf, 42, rows, buf, stride_v, wout, ostride, 7); }, 10);
            printf("blocked mirror 40r(contig)+2r + 2w(in place)+5w grid=%5d  %.1f us (%.0f GB/s);  +7w: %.1f us (%.0f GB/s)\n", grid,
                   m5 * 1e3, 49.0 * n * 8 / m5 / 1e6, m7 * 1e3, 51.0 * n * 8 / m7 / 1e6);
        }
        CK(hipFree(wout));
    }
    {   // blocked layout with 41 of 42 slots
        const int slots = 42;
        for (int g : {4, 8, 16}) {
            int grid = cus * g;
            {
                constexpr int CHV = 256;   // 4 KiB per slot per chunk
                long nchunks = nv / CHV;
                double ms = time_ms([&] { hipLaunchKernelGGL((blocked_read_k<CHV>), dim3(grid), dim3(256), 0, 0, buf, 41, slots, nchunks, out); }, 10);
                printf("blocked CH=4KiB  K=41 grid=%5d  %.0f GB/s\n", grid, 41.0 * nchunks * CHV * 16 / ms / 1e6);
            }
            {
                constexpr int CHV = 1024;  // 16 KiB per slot per chunk
                long nchunks = nv / CHV;
                double ms = time_ms([&] { hipLaunchKernelGGL((blocked_read_k<CHV>), dim3(grid), dim3(256), 0, 0, buf, 41, slots, nchunks, out); }, 10);
                printf("blocked CH=16KiB K=41 grid=%5d  %.0f GB/s\n", grid, 41.0 * nchunks * CHV * 16 / ms / 1e6);
            }
        }
    }
    return 0;
}
