// pointbench.hip -- what the MEMORY SHAPE of the point pass costs without its arithmetic (dev tool).
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/pointbench tools/pointbench.hip && tools/bin/pointbench
// Ring [row][slot 0..21][x tile | g tile][1 KiB]; one wave per SIMD; per wave-row all 42 tiles of slots 0..20 are
// loaded (non-temporal, 16 B per lane) into one of two register sets while the other set is "computed" (summed,
// plus SPIN dependent fma per loaded vector to stand in for the real work), then W of the two tiles of slot 21
// are stored (plain or non-temporal).  Reports the time of one sweep over rows = n / (2 * 62) and the bandwidth.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double v2 __attribute__((ext_vector_type(2)));
constexpr int kTiles = 42, kRowBytes = 44 * 1024;

template <int W, bool NTS, int SPIN, int BATCH = 1, int HALO = 0>
__global__ __launch_bounds__(256, 1) void sweep(char *ring, long rows, double *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long stride = (long)gridDim.x * 4;
    v2 A[kTiles], B[kTiles];
    double acc = 0;
    auto issue = [&](long row, v2 (&r)[kTiles]) {
        const char *rb = ring + (unsigned long)__builtin_amdgcn_readfirstlane((int)row) * kRowBytes + lane * 16;
#pragma unroll
        for (int t = 0; t < kTiles; ++t) r[t] = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(rb + t * 1024));
    };
    auto compute = [&](long row, v2 (&r)[kTiles]) {
        v2 s = {0, 0};
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            v2 v = r[t];
#pragma unroll
            for (int k = 0; k < SPIN; ++k) { s.x = __builtin_fma(v.x, 1.0000001, s.x); s.y = __builtin_fma(v.y, 0.9999999, s.y); }
            if (SPIN == 0) { s.x += v.x; s.y += v.y; }
        }
        if (BATCH == 1) {
            char *wb = ring + (unsigned long)__builtin_amdgcn_readfirstlane((int)row) * kRowBytes + 42 * 1024 + lane * 16;
            if (lane >= 1 && lane < 63) {
                if (W >= 1) { if (NTS) __builtin_nontemporal_store(s, reinterpret_cast<v2 *>(wb)); else *reinterpret_cast<v2 *>(wb) = s; }
                if (W >= 2) { if (NTS) __builtin_nontemporal_store(s, reinterpret_cast<v2 *>(wb + 1024)); else *reinterpret_cast<v2 *>(wb + 1024) = s; }
            }
        } else {
            // the outputs of BATCH consecutive rows of this wave written together (as if staged in LDS)
            const long it = (row - ((long)blockIdx.x * 4 + wave)) / stride;
            if (it % BATCH == BATCH - 1) {
                for (int b = 0; b < BATCH; ++b) {
                    const long r = row - (long)b * stride;
                    char *wb = ring + (unsigned long)__builtin_amdgcn_readfirstlane((int)r) * kRowBytes + 42 * 1024 + lane * 16;
                    if (lane >= 1 && lane < 63) {
                        if (W >= 1) { if (NTS) __builtin_nontemporal_store(s, reinterpret_cast<v2 *>(wb)); else *reinterpret_cast<v2 *>(wb) = s; }
                        if (W >= 2) { if (NTS) __builtin_nontemporal_store(s, reinterpret_cast<v2 *>(wb + 1024)); else *reinterpret_cast<v2 *>(wb + 1024) = s; }
                    }
                    if (HALO) {                                     // the halo copies of the real pass: 16-B stores into the neighbouring rows' tiles
                        char *t0 = ring + (unsigned long)__builtin_amdgcn_readfirstlane((int)r) * kRowBytes + 42 * 1024;
                        if (lane == 1 && r > 0) { *reinterpret_cast<v2 *>(t0 - kRowBytes + 63 * 16) = s; if (HALO > 1) *reinterpret_cast<v2 *>(t0 - kRowBytes + 1024 + 63 * 16) = s; }
                        if (lane == 62 && r + 1 < rows) { *reinterpret_cast<v2 *>(t0 + kRowBytes) = s; if (HALO > 1) *reinterpret_cast<v2 *>(t0 + kRowBytes + 1024) = s; }
                    }
                }
            }
        }
        acc += s.x + s.y;
    };
    long row = (long)blockIdx.x * 4 + wave;
    auto clampr = [&](long r) { return r < rows ? r : rows - 1; };
    issue(clampr(row), A);
    while (row < rows) {
        issue(clampr(row + stride), B);
        compute(row, A);
        row += stride;
        if (row >= rows) break;
        issue(clampr(row + stride), A);
        compute(row, B);
        row += stride;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}


// the same sweep over a STREAM-major ring: tile (stream q, row r) at q * stream_bytes + r * 1024 -- every stream is
// contiguous, so what a block's four waves write (rows 4b .. 4b+3) is one 4-KiB piece per written stream
template <int W, bool NTS, int SPIN, int BATCH, bool CHUNK = false>
__global__ __launch_bounds__(256, 1) void sweep_sm(char *ring, long rows, long stream_bytes, double *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nwaves = (long)gridDim.x * 4;
    const long per = (rows + nwaves - 1) / nwaves;                   // CHUNK: a wave takes `per` CONSECUTIVE rows
    const long stride = CHUNK ? 1 : nwaves;
    v2 A[kTiles], B[kTiles];
    double acc = 0;
    auto issue = [&](long row, v2 (&r)[kTiles]) {
        const char *rb = ring + (unsigned long)__builtin_amdgcn_readfirstlane((int)row) * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < kTiles; ++t) r[t] = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(rb + (unsigned long)t * stream_bytes));
    };
    v2 held[BATCH];
    auto compute = [&](long row, v2 (&r)[kTiles]) {
        v2 s = {0, 0};
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            v2 v = r[t];
#pragma unroll
            for (int k = 0; k < SPIN; ++k) { s.x = __builtin_fma(v.x, 1.0000001, s.x); s.y = __builtin_fma(v.y, 0.9999999, s.y); }
            if (SPIN == 0) { s.x += v.x; s.y += v.y; }
        }
        const long first = CHUNK ? ((long)blockIdx.x * 4 + wave) * per : ((long)blockIdx.x * 4 + wave);
        const long it = (row - first) / stride;
        if (it % BATCH == BATCH - 1) {
            for (int b = 0; b < BATCH; ++b) {
                const long r2 = row - (long)b * stride;
                char *wb = ring + 42 * stream_bytes + (unsigned long)__builtin_amdgcn_readfirstlane((int)r2) * 1024 + lane * 16;
                if (lane >= 1 && lane < 63) {
                    if (W >= 1) { if (NTS) __builtin_nontemporal_store(s, reinterpret_cast<v2 *>(wb)); else *reinterpret_cast<v2 *>(wb) = s; }
                    if (W >= 2) { if (NTS) __builtin_nontemporal_store(s, reinterpret_cast<v2 *>(wb + stream_bytes)); else *reinterpret_cast<v2 *>(wb + stream_bytes) = s; }
                }
            }
        }
        acc += s.x + s.y;
    };
    long row = CHUNK ? ((long)blockIdx.x * 4 + wave) * per : (long)blockIdx.x * 4 + wave;
    const long end = CHUNK ? (row + per < rows ? row + per : rows) : rows;
    auto clampr = [&](long r) { return r < rows ? r : rows - 1; };
    issue(clampr(row), A);
    while (row < end) {
        issue(clampr(row + stride), B);
        compute(row, A);
        row += stride;
        if (row >= end) break;
        issue(clampr(row + stride), A);
        compute(row, B);
        row += stride;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

template <typename F> static double time_us(F f, int reps = 8) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return 1e3 * ms / reps;
}

int main() {
    const long n = 10000000, nvec = n / 2, rows = (nvec + 61) / 62;
    char *ring; double *sink;
    CK(hipMalloc(&ring, (size_t)rows * kRowBytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(ring, 0, (size_t)rows * kRowBytes)); CK(hipDeviceSynchronize());
    const double rd = (double)rows * kTiles * 1024, wr1 = (double)rows * 992;
#define RUN(W, NTS, SPIN) { double us = time_us([&] { hipLaunchKernelGGL((sweep<W, NTS, SPIN>), dim3(256), dim3(256), 0, 0, ring, rows, sink); }); \
        printf("42 tile reads + %d tile writes (%s), %2d fma per vector: %7.1f us  %6.0f GB/s\n", W, NTS ? "nt   " : "plain", 2 * SPIN, us, (rd + W * wr1) / us / 1e3); }
#define RUNB(W, NTS, SPIN, BATCH) { double us = time_us([&] { hipLaunchKernelGGL((sweep<W, NTS, SPIN, BATCH>), dim3(256), dim3(256), 0, 0, ring, rows, sink); }); \
        printf("42 tile reads + %d tile writes (%s) in batches of %2d rows, %2d fma per vector: %7.1f us  %6.0f GB/s\n", W, NTS ? "nt   " : "plain", BATCH, 2 * SPIN, us, (rd + W * wr1) / us / 1e3); }
    {
        const long stream_bytes = ((rows * 1024 + 1023) / 1024 | 1) * 1024;      // an odd number of KiB
        char *ring2; CK(hipMalloc(&ring2, (size_t)stream_bytes * 44)); CK(hipMemset(ring2, 0, (size_t)stream_bytes * 44)); CK(hipDeviceSynchronize());
#define RUNS(W, NTS, SPIN, BATCH) { double us = time_us([&] { hipLaunchKernelGGL((sweep_sm<W, NTS, SPIN, BATCH>), dim3(256), dim3(256), 0, 0, ring2, rows, stream_bytes, sink); }); \
        printf("STREAM-major: 42 tile reads + %d tile writes (%s) in batches of %2d rows: %7.1f us  %6.0f GB/s\n", W, NTS ? "nt   " : "plain", BATCH, us, (rd + W * wr1) / us / 1e3); }
#define RUNC(W, NTS, SPIN, BATCH) { double us = time_us([&] { hipLaunchKernelGGL((sweep_sm<W, NTS, SPIN, BATCH, true>), dim3(256), dim3(256), 0, 0, ring2, rows, stream_bytes, sink); }); \
        printf("STREAM-major, consecutive rows per wave: 42 tile reads + %d tile writes (%s) in batches of %2d rows: %7.1f us  %6.0f GB/s\n", W, NTS ? "nt   " : "plain", BATCH, us, (rd + W * wr1) / us / 1e3); }
        RUNC(0, true, 8, 1) RUNC(2, true, 8, 1) RUNC(2, true, 8, 16) RUNC(2, false, 8, 16)
        RUNS(0, true, 8, 1) RUNS(2, true, 8, 1) RUNS(2, false, 8, 1) RUNS(2, false, 8, 16) RUNS(2, true, 8, 16) RUNS(1, false, 8, 1)
        CK(hipFree(ring2));
    }
#define RUNH(W, NTS, SPIN, BATCH, HALO) { double us = time_us([&] { hipLaunchKernelGGL((sweep<W, NTS, SPIN, BATCH, HALO>), dim3(256), dim3(256), 0, 0, ring, rows, sink); }); \
        printf("42 tile reads + %d tile writes (%s) in batches of %2d rows + halo copies of %d streams: %7.1f us\n", W, NTS ? "nt   " : "plain", BATCH, HALO, us); }
    RUNH(2, false, 8, 16, 0) RUNH(2, false, 8, 16, 1) RUNH(2, false, 8, 16, 2) RUNH(2, false, 8, 16, 0) RUNH(2, false, 8, 16, 2)
    RUNB(2, false, 8, 4) RUNB(2, false, 8, 8) RUNB(2, false, 8, 16) RUNB(2, true, 8, 8) RUNB(2, true, 8, 16)
    RUN(0, true, 0) RUN(1, true, 0) RUN(2, true, 0) RUN(1, false, 0) RUN(2, false, 0)
    RUN(0, true, 8) RUN(2, true, 8) RUN(2, false, 8)
    RUN(0, true, 24) RUN(2, true, 24) RUN(2, false, 24)
    RUN(0, true, 48) RUN(2, false, 48)
    return 0;
}
