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

template <int SETS, int W, bool NTS, int SPIN, int BATCH, int CH = 1, bool BSYNC = false>
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
    // CH > 1: a wave takes CH CONSECUTIVE rows at a time (rows (c * waves + first) * CH + j), so that a burst of BATCH = CH
    // rows is one contiguous piece of the written stream (4 CH KiB per block)
    auto row_of = [&](long it) { return CH == 1 ? first + it * stride : ((it / CH) * stride + first) * CH + it % CH; };
    auto finish_row = [&](long it, v2 s) {
        if (W && it % BATCH == BATCH - 1) {
            if (BSYNC) __syncthreads();
            for (int b = 0; b < BATCH; ++b) {
                const long r2 = clampr(row_of(it - b));
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
    const long its = CH == 1 ? (rows - first + stride - 1) / stride : ((rows + CH - 1) / CH - first + stride - 1) / stride * CH;   // (CH > 1: whole chunks, clamped rows)
    long it = 0;
    issue(clampr(row_of(0)), A);
    if (SETS == 2) {
        auto compute = [&](long i, v2 *r) {
            v2 s = {0, 0};
#pragma unroll
            for (int t = 0; t < R; ++t) eat(r[t], s);
            finish_row(i, s);
        };
        while (it < its) {
            issue(clampr(row_of(it + 1)), B);
            compute(it, A);
            ++it;
            if (it >= its) break;
            issue(clampr(row_of(it + 1)), A);
            compute(it, B);
            ++it;
        }
    } else {
        while (it < its) {
            const char *nb = base(clampr(row_of(it + 1)));
            v2 s = {0, 0};
#pragma unroll
            for (int t = 0; t < R; ++t) {
                eat(A[t], s);
                A[t] = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(nb + (unsigned long)t * stream_bytes));   // refill behind its last reader
            }
            finish_row(it, s);
            ++it;
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
#define RUNX(SETS, W, NTS, SPIN, BATCH, CH, BSYNC) { double us = time_us([&] { hipLaunchKernelGGL((sweep21<SETS, W, NTS, SPIN, BATCH, CH, BSYNC>), dim3(cus * (SETS == 1 ? 2 : 1)), dim3(256), 0, 0, ring, rows, stream_bytes, sink); }); \
        printf("%d set(s): 21 reads + %d write (%s), bursts of %2d rows, %2d CONSECUTIVE rows per wave%s, %3d fma per vector: %7.1f us\n", \
               SETS, W, NTS ? "nt   " : "plain", BATCH, CH, BSYNC ? ", block-synchronised bursts" : "", 2 * SPIN, us); }
    for (int rep = 0; rep < 2; ++rep) {
        RUNX(1, 1, true, 0, 16, 16, false) RUNX(1, 1, true, 0, 16, 16, true) RUNX(1, 1, true, 0, 8, 8, false) RUNX(1, 1, true, 0, 4, 4, false)
        RUNX(1, 0, true, 0, 16, 16, false) RUNX(1, 1, true, 48, 16, 16, false) RUNX(1, 1, false, 0, 16, 16, false)
        // (bursts longer than the ~39 rows a wave has at n = 1e7 would simply never be written: BATCH <= 16 only)
        RUN(1, 0, true, 0, 16) RUN(1, 1, true, 0, 16) RUN(1, 1, false, 0, 16) RUN(1, 1, true, 0, 1)
        RUN(1, 1, true, 8, 16) RUN(1, 1, true, 24, 16) RUN(1, 1, true, 48, 16) RUN(1, 1, true, 96, 16)
        RUN(2, 0, true, 0, 16) RUN(2, 1, true, 0, 16) RUN(2, 1, true, 24, 16) RUN(2, 1, true, 48, 16) RUN(2, 1, true, 96, 16)
    }
    return 0;
}
