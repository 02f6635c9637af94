// pointbench21.hip -- the MEMORY SHAPE of the round-3 point pass without its arithmetic (dev tool; the round-2 shape,
// 42 read tiles + 2 written, is tools/pointbench.hip).
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/pointbench21 tools/pointbench21.hip && tools/bin/pointbench21
// STREAM-major ring: tile (stream q, row r) at q * stream_bytes + r * 1 KiB.  Per wave-row the R = 21 tiles of streams
// 0..20 are loaded (non-temporal, 16 B per lane) and "computed" (summed, plus SPIN dependent fma pairs per loaded vector
// to stand in for the real work); W = 0 / 1 tiles of stream 21 are stored in bursts of BATCH rows.  Two forms:
//   SETS = 2: two register sets, one wave per SIMD  (grid = CUs blocks; the round-2 schedule)
//   SETS = 1: ONE register set refilled tile by tile from inside the compute loop, two waves per SIMD (grid = 2 CUs
//             blocks) -- the schedule of lbfgs_point_pass_kernel<double, 20, false, 1>
// Reports the time of one sweep over rows = n / (2 * 62) and the bandwidth over (R + W) tiles per row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double v2 __attribute__((ext_vector_type(2)));
constexpr int R = 21;

template <int SETS, int W, bool NTS, int SPIN, int BATCH>
__global__ __launch_bounds__(256, SETS == 1 ? 2 : 1) void sweep21(char *ring, long rows, long stream_bytes, double *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long stride = (long)gridDim.x * 4;
    v2 A[R], B[SETS == 2 ? R : 1];
    double acc = 0;
    auto clampr = [&](long r) { return r < rows ? r : rows - 1; };
    auto base = [&](long row) { return ring + (unsigned long)__builtin_amdgcn_readfirstlane((int)row) * 1024 + lane * 16; };
    auto issue = [&](long row, v2 *r) {
        const char *rb = base(row);
#pragma unroll
        for (int t = 0; t < R; ++t) r[t] = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(rb + (unsigned long)t * stream_bytes));
    };
    const long first = (long)blockIdx.x * 4 + wave;
    auto finish_row = [&](long row, v2 s) {
        const long it = (row - first) / stride;
        if (W && it % BATCH == BATCH - 1) {
            for (int b = 0; b < BATCH; ++b) {
                const long r2 = row - (long)b * stride;
                char *wb = ring + (unsigned long)R * stream_bytes + (unsigned long)__builtin_amdgcn_readfirstlane((int)r2) * 1024 + lane * 16;
                if (lane >= 1 && lane < 63) { if (NTS) __builtin_nontemporal_store(s, reinterpret_cast<v2 *>(wb)); else *reinterpret_cast<v2 *>(wb) = s; }
            }
        }
        acc += s.x + s.y;
    };
    auto eat = [&](v2 v, v2 &s) {
#pragma unroll
        for (int k = 0; k < SPIN; ++k) { s.x = __builtin_fma(v.x, 1.0000001, s.x); s.y = __builtin_fma(v.y, 0.9999999, s.y); }
        if (SPIN == 0) { s.x += v.x; s.y += v.y; }
    };
    long row = first;
    issue(clampr(row), A);
    if (SETS == 2) {
        auto compute = [&](long rw, v2 *r) {
            v2 s = {0, 0};
#pragma unroll
            for (int t = 0; t < R; ++t) eat(r[t], s);
            finish_row(rw, s);
        };
        while (row < rows) {
            issue(clampr(row + stride), B);
            compute(row, A);
            row += stride;
            if (row >= rows) break;
            issue(clampr(row + stride), A);
            compute(row, B);
            row += stride;
        }
    } else {
        while (row < rows) {
            const char *nb = base(clampr(row + stride));
            v2 s = {0, 0};
#pragma unroll
            for (int t = 0; t < R; ++t) {
                eat(A[t], s);
                A[t] = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(nb + (unsigned long)t * stream_bytes));   // refill behind its last reader
            }
            finish_row(row, s);
            row += stride;
        }
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

template <typename F> static double time_us(F f, int reps = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return 1e3 * ms / reps;
}

int main() {
    const long n = 10000000, nvec = n / 2, rows = (nvec + 61) / 62;
    const long stream_bytes = ((rows * 1024 + 1023) / 1024 | 1) * 1024;      // an odd number of KiB
    char *ring; double *sink;
    CK(hipMalloc(&ring, (size_t)stream_bytes * (R + 1))); CK(hipMalloc(&sink, 64));
    CK(hipMemset(ring, 0, (size_t)stream_bytes * (R + 1))); CK(hipDeviceSynchronize());
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double rd = (double)rows * R * 1024, wr1 = (double)rows * 992;
    printf("n = %ld fp64, rows = %ld, %d CUs; algorithmic bytes of the real pass (k+2) n T = %.3f GB\n", n, rows, cus, 22.0 * n * 8 / 1e9);
#define RUN(SETS, W, NTS, SPIN, BATCH) { double us = time_us([&] { hipLaunchKernelGGL((sweep21<SETS, W, NTS, SPIN, BATCH>), dim3(cus * (SETS == 1 ? 2 : 1)), dim3(256), 0, 0, ring, rows, stream_bytes, sink); }); \
        printf("%d set(s), %d wave(s)/SIMD: 21 tile reads + %d tile write (%s, bursts of %2d rows), %3d fma per vector: %7.1f us  %6.0f GB/s moved, %6.0f GB/s by (k+2) n T\n", \
               SETS, SETS == 1 ? 2 : 1, W, NTS ? "nt   " : "plain", BATCH, 2 * SPIN, us, (rd + W * wr1) / us / 1e3, 22.0 * n * 8 / us / 1e3); }
    for (int rep = 0; rep < 2; ++rep) {
        RUN(1, 0, true, 0, 16) RUN(1, 1, true, 0, 16) RUN(1, 1, false, 0, 16) RUN(1, 1, true, 0, 1)
        RUN(1, 1, true, 8, 16) RUN(1, 1, true, 24, 16) RUN(1, 1, true, 48, 16) RUN(1, 1, true, 96, 16)
        RUN(2, 0, true, 0, 16) RUN(2, 1, true, 0, 16) RUN(2, 1, true, 24, 16) RUN(2, 1, true, 48, 16) RUN(2, 1, true, 96, 16)
    }
    return 0;
}
