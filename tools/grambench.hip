// grambench.hip -- feature-ablation microbenchmark for the L-BFGS Gram pass (dev tool).
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -o grambench tools/grambench.hip
// Re-creates the Gram pass's traffic (g + pivot pair + k pairs of 16-B-per-lane streams, fp64)
// with switchable parts, to find out which part keeps it below the plain 41-stream read rate of
// tools/streambench.hip.  Results are garbage numerically when a part is switched off; only
// the time matters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef double v2 __attribute__((ext_vector_type(2)));
constexpr int kBlock = 256;
constexpr int kMaxK = 64;

struct Params {
    long nvec;
    int k;
    const v2 *g, *sp, *yp;
    const v2 *s[kMaxK];
    const v2 *y[kMaxK];
    double *partials;
};

__device__ __forceinline__ v2 ldnt(const v2 *p) { return __builtin_nontemporal_load(p); }

__device__ __forceinline__ double readlane_f64(double v, int l) {
    union { double d; int i[2]; } a, b;
    a.d = v;
    b.i[0] = __builtin_amdgcn_readlane(a.i[0], l);
    b.i[1] = __builtin_amdgcn_readlane(a.i[1], l);
    return b.d;
}

__device__ __forceinline__ void wave_sum5(const double (&t)[5], int lane, double (&tot)[5]) {
    const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0;
    const double r0 = __shfl_xor(b5 ? t[0] : t[3], 32, 64);
    const double r1 = __shfl_xor(b5 ? t[1] : t[4], 32, 64);
    const double r2 = __shfl_xor(b5 ? t[2] : 0.0, 32, 64);
    const double a0 = (b5 ? t[3] : t[0]) + r0;
    const double a1 = (b5 ? t[4] : t[1]) + r1;
    const double a2 = t[2] + r2;
    const double u0 = b5 ? (b4 ? a0 : a1) : (b4 ? a0 : a2);
    const double u1 = (!b5 && b4) ? a1 : 0.0;
    const double v0 = __shfl_xor(u0, 16, 64);
    const double v1 = __shfl_xor(u1, 16, 64);
    const double c0 = (b5 ? (b4 ? a1 : a0) : (b4 ? a2 : a0)) + v0;
    const double c1 = a1 + v1;
    const bool g00 = !b5 && !b4;
    const double x = __shfl_xor(g00 ? (b3 ? c0 : c1) : c0, 8, 64);
    double e = (g00 ? (b3 ? c1 : c0) : c0) + x;
    e += __shfl_xor(e, 4, 64);
    e += __shfl_xor(e, 2, 64);
    e += __shfl_xor(e, 1, 64);
    tot[0] = readlane_f64(e, 0);
    tot[1] = readlane_f64(e, 8);
    tot[2] = readlane_f64(e, 16);
    tot[3] = readlane_f64(e, 32);
    tot[4] = readlane_f64(e, 48);
}

// U   : 16-B vectors per lane per stream step
// DB  : two register sets (prefetch pair i+1 while pair i is consumed)
// NF  : 5 = the five dot products, 1 = one fma per element, 0 = plain add
// BF  : 1 = transposed butterfly + lane-distributed accumulate, 0 = lane-local accumulate
// WT  : 1 = wave-granular tiles (each wave walks its own tiles), 0 = block tiles
template <int U, bool DB, int NF, bool BF, bool WT>
__global__ __launch_bounds__(kBlock) void gram_like(Params p) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double acc[5] = {0, 0, 0, 0, 0};
    const int k = p.k;
    const long tile_v = WT ? 64L * U : (long)kBlock * U;
    const long stride_u = WT ? 64 : kBlock;
    const long tid = WT ? lane : threadIdx.x;
    const long first = WT ? (long)blockIdx.x * 4 + wave : blockIdx.x;
    const long step = WT ? (long)gridDim.x * 4 : gridDim.x;
    const long full_tiles = p.nvec / tile_v;
    for (long tile = first; tile < full_tiles; tile += step) {
        const long base = tile * tile_v + tid;
        v2 gv[U], spv[U], ypv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            gv[u] = ldnt(p.g + base + u * stride_u);
            if (NF == 5) { spv[u] = ldnt(p.sp + base + u * stride_u); ypv[u] = ldnt(p.yp + base + u * stride_u); }
        }
        v2 sA[U], yA[U], sB[U], yB[U];
        auto fetch = [&](int i, v2 (&sv)[U], v2 (&yv)[U]) {
            const v2 *si = p.s[i], *yi = p.y[i];
#pragma unroll
            for (int u = 0; u < U; ++u) { sv[u] = ldnt(si + base + u * stride_u); yv[u] = ldnt(yi + base + u * stride_u); }
        };
        auto consume = [&](int i, const v2 (&sv)[U], const v2 (&yv)[U]) {
            double t[5] = {0, 0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const double sx = sv[u][j], yx = yv[u][j];
                    if (NF == 5) {
                        t[0] = __builtin_fma(sx, gv[u][j], t[0]);
                        t[1] = __builtin_fma(yx, gv[u][j], t[1]);
                        t[2] = __builtin_fma(yx, ypv[u][j], t[2]);
                        t[3] = __builtin_fma(yx, spv[u][j], t[3]);
                        t[4] = __builtin_fma(sx, ypv[u][j], t[4]);
                    } else if (NF == 1) {
                        t[0] = __builtin_fma(sx, gv[u][j], t[0]);
                        t[1] = __builtin_fma(yx, gv[u][j], t[1]);
                    } else {
                        t[0] += sx; t[1] += yx;
                    }
                }
            }
            if (BF) {
                double tot[5];
                wave_sum5(t, lane, tot);
                if (lane == i) {
#pragma unroll
                    for (int c = 0; c < 5; ++c) acc[c] += tot[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 5; ++c) acc[c] += t[c];
            }
        };
        if (DB) {
            fetch(0, sA, yA);
            for (int i = 0; i < k; i += 2) {
                if (i + 1 < k) fetch(i + 1, sB, yB);
                consume(i, sA, yA);
                if (i + 2 < k) fetch(i + 2, sA, yA);
                if (i + 1 < k) consume(i + 1, sB, yB);
            }
        } else {
            for (int i = 0; i < k; ++i) { fetch(i, sA, yA); consume(i, sA, yA); }
        }
    }
    __shared__ double wacc[4][kMaxK][5];
    if (lane < k) {
#pragma unroll
        for (int c = 0; c < 5; ++c) wacc[wave][lane][c] = acc[c];
    }
    __syncthreads();
    if (wave == 0 && lane < k) {
#pragma unroll
        for (int c = 0; c < 5; ++c)
            p.partials[(long)(lane * 5 + c) * gridDim.x + blockIdx.x] = (wacc[0][lane][c] + wacc[1][lane][c]) + (wacc[2][lane][c] + wacc[3][lane][c]);
    }
}

template <typename F> double time_us(F f, int reps = 12) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return t[t.size() / 2] * 1e3;
}

int main(int argc, char **argv) {
    const long nmax = 11000000;
    const int k = 20;
    const long slot = ((nmax * 8 + 1023) / 1024 | 1) * 1024 / 16;   // odd-KiB stride, in v2 units
    v2 *buf; double *partials;
    CK(hipMalloc(&buf, sizeof(v2) * slot * (2 * (k + 1) + 1)));
    CK(hipMalloc(&partials, sizeof(double) * 5 * kMaxK * 8192));
    if (argc > 1 && atoi(argv[1]) == 0) {
        CK(hipMemset(buf, 0, sizeof(v2) * slot * (2 * (k + 1) + 1)));
        printf("data: zeros\n");
    } else {
        // random mantissas: DRAM / datapath toggling like real history vectors
        const size_t cnt = (size_t)slot * (2 * (k + 1) + 1) * 2;
        std::vector<double> h(1 << 22);
        unsigned long long st = 88172645463325252ULL;
        for (auto &x : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x = (double)(st >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
        for (size_t off = 0; off < cnt; off += h.size())
            CK(hipMemcpy((double *)buf + off, h.data(), sizeof(double) * std::min(h.size(), cnt - off), hipMemcpyHostToDevice));
        printf("data: random\n");
    }
    CK(hipDeviceSynchronize());
    Params p;
    p.k = k; p.partials = partials;
    p.g = buf + slot * 2 * (k + 1);
    for (int i = 0; i < k; ++i) { p.s[i] = buf + slot * (2 * i); p.y[i] = buf + slot * (2 * i + 1); }
    p.sp = p.s[0]; p.yp = p.y[0];
    const double bytes_per_n = (2.0 * k + 1) * 8;

#define RUN(NAME, U, DB, NF, BF, WT, GRID, N)                                                         \
    do {                                                                                              \
        p.nvec = (N) / 2;                                                                             \
        double us = time_us([&] { hipLaunchKernelGGL((gram_like<U, DB, NF, BF, WT>), dim3(GRID), dim3(kBlock), 0, 0, p); }); \
        printf("%-34s U=%d grid=%5d n=%9ld  %7.1f us  %6.0f GB/s\n", NAME, U, GRID, (long)(N), us, bytes_per_n * (N) / us / 1e3); \
        fflush(stdout);                                                                               \
    } while (0)

    const long n = 10000000;
    for (int grid : {512, 768, 1024, 2048}) {
        RUN("full (shipped shape)", 4, true, 5, true, false, grid, n);
        RUN("full", 2, true, 5, true, false, grid, n);
        RUN("no butterfly", 4, true, 5, false, false, grid, n);
        RUN("no butterfly", 2, true, 5, false, false, grid, n);
        RUN("no butterfly, 2 fma", 4, true, 1, false, false, grid, n);
        RUN("no butterfly, 2 fma", 2, true, 1, false, false, grid, n);
        RUN("no butterfly, add", 2, true, 0, false, false, grid, n);
        RUN("no butterfly, add, single buffer", 2, false, 0, false, false, grid, n);
        RUN("no butterfly, add, single buffer", 4, false, 0, false, false, grid, n);
        RUN("full, single buffer", 4, false, 5, true, false, grid, n);
        RUN("full, wave tiles", 4, true, 5, true, true, grid, n);
        RUN("full, wave tiles", 2, true, 5, true, true, grid, n);
        RUN("full, wave tiles", 8, true, 5, true, true, grid, n);
    }
    // sawtooth in n (tail / quantisation effect), shipped shape
    for (long nn = 9000000; nn <= 11000000; nn += 250000) RUN("full (n sweep)", 4, true, 5, true, false, 768, nn);
    return 0;
}
