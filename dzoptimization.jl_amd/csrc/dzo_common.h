// dzo_common.h -- shared device helpers and host-side plumbing for libdzo_hip.so (gfx950).
//
// Design rules:
//   * every hot path here is HBM-bound BLAS-1 style work: 16-byte-per-lane coalesced loads,
//     wave64 shuffle reductions, LDS only for the 4-wave block combine;
//   * reductions are deterministic: fixed grid, per-block partials in HBM, a fixed-order
//     second stage.  No floating-point atomics anywhere;
//   * accumulation is always in fp64, also for fp32 vectors (VALU is idle on these kernels);
//   * elementwise arithmetic uses explicit fma so results are bit-identical to the oracle.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dzo.h"

namespace dzo {

constexpr int kBlock = 256;          // 4 waves of 64
constexpr int kWaves = kBlock / 64;
constexpr int kMaxPartialBlocks = 2048;  // upper bound on the grid of any reducing kernel
constexpr int kMaxHistory = 64;      // recurrence kernel runs one lane per pair

// ------------------------------------------------------------------------------ errors
void set_error(const char *fmt, ...);
int32_t hip_fail(hipError_t e, const char *what, const char *file, int line);

#define DZO_HIP(call)                                                        \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) return ::dzo::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

#define DZO_TRY(call)                        \
    do {                                     \
        int32_t rc__ = (call);               \
        if (rc__ != DZO_OK) return rc__;     \
    } while (0)

#define DZO_REQUIRE(cond, code, ...)         \
    do {                                     \
        if (!(cond)) {                       \
            ::dzo::set_error(__VA_ARGS__);   \
            return (code);                   \
        }                                    \
    } while (0)

// ------------------------------------------------------------------------------ context
struct Context {
    bool ready = false;
    int device = -1;
    int cus = 256;
    int64_t hbm_bytes = 0;
    char name[128] = {0};
    hipStream_t stream = nullptr;       // default stream for handle-less primitives
    double *scratch = nullptr;          // device: partial sums for handle-less reductions
    double *host_scalar = nullptr;      // pinned host: results of blocking reductions
};
// One context per HIP device; ctx() is the context of the device the CALLING THREAD last selected
// with dzo_init (or entered through a DeviceScope), the first initialised device otherwise.
Context &ctx();
int32_t require_init();
// Spin until the pinned host word `*word` holds `ticket` (a kernel on `s` writes it after its results, behind a
// system-scope fence).  Cheaper than hipStreamSynchronize / an event: no barrier packet on the stream, no wake-up
// through the runtime's signal wait.  The stream is queried every 64 Ki spins so that a failed launch cannot hang
// the caller.  DZO_TUNE_POLL=0: hipStreamSynchronize instead.
int32_t wait_ticket(hipStream_t s, const double *word, double ticket);
// ... and then until the `count` doubles at `data` carry the publishing kernel's SEAL: the xor of their bit patterns and
// of the ticket's, which the kernel stores in `*seal` next to them.  Data and ticket travel to host memory as separate
// writes; a system-scope fence between them orders them on the GPU, and still the host has been seen to read the new
// ticket next to the previous publish's data when the two lie in different cache lines (1 to 5 times in 10 000 waits,
// MI355X over PCIe).  The seal does not depend on any ordering.  (Results that share the ticket's 64-byte line --
// core_wait_decision -- have never been seen torn and are not sealed.)
int32_t wait_sealed(hipStream_t s, const double *data, int count, const double *seal, double ticket);
int64_t unsealed_first_reads();                  // how often wait_sealed had to look twice (process-wide, diagnostic)
constexpr int kMaxDevices = 32;

// Enter `device` for the lifetime of the scope (HIP's current device of this thread and the
// library context), then return to where the thread was.  Entry points of handles that record
// their device (the batched optimizer, the communicator) use it so that one host thread can drive
// shards on several GPUs.
struct DeviceScope {
    int prev_hip = -1, prev_ctx = -1;
    bool active = false;
    explicit DeviceScope(int device);
    ~DeviceScope();
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// streaming-kernel grid: enough blocks to fill 256 CUs x 8, grid-stride the rest
inline int stream_grid(int64_t n, int elems_per_thread) {
    int64_t per_block = (int64_t)kBlock * elems_per_thread;
    int64_t blocks = (n + per_block - 1) / per_block;
    int64_t cap = (int64_t)ctx().cus * 8;
    if (cap > kMaxPartialBlocks) cap = kMaxPartialBlocks;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// Optimizers may keep the current point / gradient in an internal twin of the array they alias (see
// dzo_lbfgs.hip); every entry point through which the HOST is about to look at device memory
// (dzo_synchronize, dzo_memcpy_*) first writes them back.  Cheap when nothing is pending.
int32_t settle_all_optimizers();
// registry behind it: a handle registers itself (with the function that settles it and removes it again)
// while its live copy sits in a twin
void unsettled_add(void *handle, int32_t (*settle)(void *handle));
void unsettled_remove(void *handle);
// destroy: remove the handle and wait until no OTHER thread is inside a settle call on it (call without the
// handle's own mutex held, after the handle has been settled by the caller)
void unsettled_retire(void *handle);

// The reference constructors' backend asserts (src/DZOptimization.jl:363-364, 376-378, 410, 420; AdGD :216-226):
// every array a constructor is given must live on the device of the calling thread's library context, where the
// constructor allocates (`similar`) everything else.  DZO_ERR_ASSERT with the reference's wording for a host
// pointer, memory HIP does not know, or another GPU's memory.  Null pointers are skipped.
int32_t require_same_backend(const char *where, const char *cite, const void *a, const char *a_name, const void *b, const char *b_name);
// device that owns a device pointer; -1 for host memory / unknown pointers
int32_t pointer_device(const void *p);

// ------------------------------------------------------------------------------ profiling
// HIP-event pairs recorded on the launching stream around each kernel (bench roofline leg).
struct ProfileEntry {
    std::string name;
    int64_t launches = 0;
    double total_ms = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};
bool profiling_on();
void profile_begin(const char *name, hipStream_t s, hipEvent_t *stop_out);
void profile_end(hipEvent_t stop, hipStream_t s);

struct ScopedKernelTimer {
    hipEvent_t stop = nullptr;
    hipStream_t s;
    ScopedKernelTimer(const char *name, hipStream_t stream) : s(stream) {
        if (profiling_on()) profile_begin(name, s, &stop);
    }
    ~ScopedKernelTimer() {
        if (stop) profile_end(stop, s);
    }
};
#define DZO_TIMED(name, stream) ::dzo::ScopedKernelTimer timer__(name, stream)

// ------------------------------------------------------------------------------ device side
__device__ __forceinline__ unsigned long long seal_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ void store_seal(double *slot, unsigned long long seal) { *reinterpret_cast<unsigned long long *>(slot) = seal; }

template <typename T> struct Vec16;   // 16-byte vector of T
template <> struct Vec16<double> { using type = double2; static constexpr int N = 2; };
template <> struct Vec16<float>  { using type = float4;  static constexpr int N = 4; };

template <typename T> __device__ __forceinline__ T dfma(T a, T b, T c);
template <> __device__ __forceinline__ double dfma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float dfma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T> __device__ __forceinline__ void load16(const T *p, T (&v)[Vec16<T>::N]) {
    using V = typename Vec16<T>::type;
    V t = *reinterpret_cast<const V *>(p);
    if constexpr (Vec16<T>::N == 2) { v[0] = t.x; v[1] = t.y; }
    else { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
}
template <typename T> __device__ __forceinline__ void store16(T *p, const T (&v)[Vec16<T>::N]) {
    using V = typename Vec16<T>::type;
    V t;
    if constexpr (Vec16<T>::N == 2) { t.x = v[0]; t.y = v[1]; }
    else { t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; }
    *reinterpret_cast<V *>(p) = t;
}

// Non-temporal 16-byte load: streamed-once operands (the (s, y) history) should not displace
// reusable lines; measured +10 % on a 41-stream fp64 read (tools/streambench.hip).
template <typename T> __device__ __forceinline__ void load16_nt(const T *p, T (&v)[Vec16<T>::N]) {
    if constexpr (Vec16<T>::N == 2) {
        typedef double v2f64 __attribute__((ext_vector_type(2)));
        const v2f64 t = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(p));
        v[0] = t.x; v[1] = t.y;
    } else {
        typedef float v4f32 __attribute__((ext_vector_type(4)));
        const v4f32 t = __builtin_nontemporal_load(reinterpret_cast<const v4f32 *>(p));
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
}

template <typename T> __device__ __forceinline__ void store16_nt(T *p, const T (&v)[Vec16<T>::N]) {
    if constexpr (Vec16<T>::N == 2) {
        typedef double v2f64 __attribute__((ext_vector_type(2)));
        v2f64 t; t.x = v[0]; t.y = v[1];
        __builtin_nontemporal_store(t, reinterpret_cast<v2f64 *>(p));
    } else {
        typedef float v4f32 __attribute__((ext_vector_type(4)));
        v4f32 t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
        __builtin_nontemporal_store(t, reinterpret_cast<v4f32 *>(p));
    }
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

// Wave-wide sums of FIVE values with 9 cross-lane exchanges instead of 5 x 6: a transposed
// ("reduce-scatter") butterfly -- at the first three steps a lane keeps only part of the values
// and ships the rest to its partner, so that afterwards every lane owns ONE value, which three
// plain pairwise sums finish.  Totals come back through v_readlane (scalar).
// No LDS anywhere: the exchanges are DPP moves (quad_perm for lane^1 and lane^2, row_ror:8 for
// lane^8, two bank-masked row shifts for lane^4) and the gfx950 v_permlane16/32_swap for the
// cross-row sums.  __shfl_xor compiles to ds_bpermute: six dependent LDS round trips per
// butterfly, which a kernel running one wave per SIMD (the single-pass step) cannot hide.
// Fixed association order -> deterministic.
template <int CTRL, int BANK> __device__ __forceinline__ double dpp_f64(double old, double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, CTRL, 0xF, BANK, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, CTRL, 0xF, BANK, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_xor1(double v) { return dpp_f64<0xB1, 0xF>(v, v); }    // quad_perm:[1,0,3,2]
__device__ __forceinline__ double lane_xor2(double v) { return dpp_f64<0x4E, 0xF>(v, v); }    // quad_perm:[2,3,0,1]
__device__ __forceinline__ double lane_xor8(double v) { return dpp_f64<0x128, 0xF>(v, v); }   // row_ror:8
__device__ __forceinline__ double lane_xor4(double v) {
    const double up = dpp_f64<0x104, 0x5>(v, v);        // row_shl:4 into banks 0, 2 (lanes with bit 2 clear)
    return dpp_f64<0x114, 0xA>(up, v);                  // row_shr:4 into banks 1, 3
}
// v[l] + v[l ^ 16] (resp. ^ 32) in every lane: the swap leaves {own, own} in one register and
// {partner, partner} in the other
__device__ __forceinline__ double sum_xor16(double v) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const u2 rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const u2 rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)rh.x, (int)rl.x) + __hiloint2double((int)rh.y, (int)rl.y);
}
__device__ __forceinline__ double sum_xor32(double v) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const u2 rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const u2 rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)rh.x, (int)rl.x) + __hiloint2double((int)rh.y, (int)rl.y);
}

// wave64 sums in the association order of the classic shuffle trees (offsets 32, 16, 8, 4, 2, 1), without LDS:
// every step adds the SAME two operands the __shfl_down / __shfl_xor form adds (a + b == b + a bit for bit), so the
// results are those of the shuffle forms -- which compile to ds_bpermute, six dependent LDS round trips per sum.
// wave_sum: result valid in lane 0.  wave_sum_all: the same value in every lane.  All 64 lanes must be active.
__device__ __forceinline__ double wave_sum_all(double v) {
    v = sum_xor32(v);
    v = sum_xor16(v);
    v += lane_xor8(v);
    v += lane_xor4(v);
    v += lane_xor2(v);
    v += lane_xor1(v);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_all(v); }

// Neighbour lanes without LDS: the value of lane - 1 (lane 0 keeps its own) / lane + 1 (lane 63 keeps its own) -- what
// __shfl_up(v, 1) / __shfl_down(v, 1) return, but as one DPP move per dword (wave_shr:1 / wave_shl:1, whole-wave shifts
// of the gfx9 family) instead of ds_bpermute's LDS round trip.  The stencil kernels run one wave per SIMD and cannot hide
// that latency (three exchanges per wave-row in the point pass).
template <typename T> __device__ __forceinline__ T lane_prev(T v);
template <typename T> __device__ __forceinline__ T lane_next(T v);
template <> __device__ __forceinline__ double lane_prev<double>(double v) { return dpp_f64<0x138, 0xF>(v, v); }   // wave_shr:1
template <> __device__ __forceinline__ double lane_next<double>(double v) { return dpp_f64<0x130, 0xF>(v, v); }   // wave_shl:1
template <> __device__ __forceinline__ float lane_prev<float>(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
template <> __device__ __forceinline__ float lane_next<float>(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xF, 0xF, false));
}

// the same where the lane without a neighbour may receive anything: it gets 0 (bound_ctrl), and the shift needs no copy
// of the source to serve as the DPP's `old` operand (4 moves per fp64 value pair in the point pass's stencils)
template <typename T> __device__ __forceinline__ T lane_prev0(T v);
template <typename T> __device__ __forceinline__ T lane_next0(T v);
template <> __device__ __forceinline__ double lane_prev0<double>(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <> __device__ __forceinline__ double lane_next0<double>(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <> __device__ __forceinline__ float lane_prev0<float>(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, true));
}
template <> __device__ __forceinline__ float lane_next0<float>(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true));
}

// the same with a value for the lane that has no neighbour (lane 0 / lane 63 keep `edge`: the DPP's `old` operand).
// (A v_writelane after the plain shift measured 84 instructions MORE per wave-row of the point pass at K = 20.)
template <typename T> __device__ __forceinline__ T lane_prev_or(T v, T edge);
template <typename T> __device__ __forceinline__ T lane_next_or(T v, T edge);
template <> __device__ __forceinline__ double lane_prev_or<double>(double v, double edge) { return dpp_f64<0x138, 0xF>(edge, v); }
template <> __device__ __forceinline__ double lane_next_or<double>(double v, double edge) { return dpp_f64<0x130, 0xF>(edge, v); }
template <> __device__ __forceinline__ float lane_prev_or<float>(float v, float edge) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
template <> __device__ __forceinline__ float lane_next_or<float>(float v, float edge) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xF, 0xF, false));
}

// wave64 sum in every lane without touching LDS (DPP + permlane swaps; fixed order)
__device__ __forceinline__ double wave_sum_all_dpp(double v) {
    v += lane_xor1(v);
    v += lane_xor2(v);
    v += lane_xor4(v);
    v += lane_xor8(v);
    v = sum_xor16(v);
    return sum_xor32(v);
}

// ---------------------------------------------------------------------------------------------------------------
// Streaming transposed reduction: wave-wide sums of MANY values (the 5 (K + 1) dot-product partials of a wave-row of
// the point pass) at ~5 instructions per value instead of a 9-exchange butterfly + 10 v_readlane + 5 masked adds
// per five of them.
//
// combine<L>(a, b) takes two per-lane values and returns ONE register in which the lanes with bit (5 - L) clear hold
// a[l] + a[l ^ bit] and the lanes with that bit set hold b[l] + b[l ^ bit]: one level of a reduction tree for two
// values at the price of one.  Levels 0 and 1 are the gfx950 v_permlane32/16_swap (which exchange half-waves / odd
// and even rows BETWEEN two registers: no select needed), levels 2 and 3 bank-masked DPP row shifts by 8 and 4 (the
// `old` operand supplies the lanes that keep their own value: no select either), levels 4 and 5 quad_perm exchanges
// behind a select.  Pushing values through a binary counter of pending subtrees (push<v>) sends 64 values through
// 32 + 16 + 8 + 4 + 2 + 1 combines; afterwards lane l holds the complete wave-wide sum of value bitrev6(l).
// No LDS, fixed association order -> deterministic.
template <int L> __device__ __forceinline__ double tree_combine(double a, double b, int lane) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    if constexpr (L == 0 || L == 1) {
        const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
        const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
        u2 rl, rh;
        if constexpr (L == 0) { rl = __builtin_amdgcn_permlane32_swap(alo, blo, false, false); rh = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false); }
        else { rl = __builtin_amdgcn_permlane16_swap(alo, blo, false, false); rh = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false); }
        // .x = [a.low half | b.low half], .y = [a.high half | b.high half]  (halves of the wave / of each row pair)
        return __hiloint2double((int)rh.x, (int)rl.x) + __hiloint2double((int)rh.y, (int)rl.y);
    } else if constexpr (L == 2) {
        const double r1 = dpp_f64<0x108, 0x3>(b, a);     // row_shl:8 into banks 0, 1: a[l + 8]; the other lanes keep b[l]
        const double r2 = dpp_f64<0x118, 0xC>(a, b);     // row_shr:8 into banks 2, 3: b[l - 8]; the other lanes keep a[l]
        return r1 + r2;
    } else if constexpr (L == 3) {
        const double r1 = dpp_f64<0x104, 0x5>(b, a);     // row_shl:4 into banks 0, 2
        const double r2 = dpp_f64<0x114, 0xA>(a, b);     // row_shr:4 into banks 1, 3
        return r1 + r2;
    } else if constexpr (L == 4) {
        const bool hi = (lane & 2) != 0;
        return (hi ? b : a) + lane_xor2(hi ? a : b);
    } else {
        const bool hi = (lane & 1) != 0;
        return (hi ? b : a) + lane_xor1(hi ? a : b);
    }
}

template <int NVALUES> struct TreeSum {
    static constexpr int kGroups = (NVALUES + 63) / 64;
    double lv[6];
    double acc[kGroups];           // per lane: the running total (over the wave-rows) of value 64 g + bitrev6(lane)
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int g = 0; g < kGroups; ++g) acc[g] = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) lv[b] = 0;
    }
    template <int B> __device__ __forceinline__ double level(double a, double b, int lane) { return tree_combine<B>(a, b, lane); }
    // value number V of the current wave-row (call with V = 0, 1, ..., NVALUES - 1 in this order, then finish_row)
    template <int V> __device__ __forceinline__ void push(double x, int lane) {
        constexpr int w = V & 63;
        if constexpr ((w & 1) == 0) { lv[0] = x; return; }
        x = tree_combine<0>(lv[0], x, lane);
        if constexpr ((w & 2) == 0) { lv[1] = x; return; }
        x = tree_combine<1>(lv[1], x, lane);
        if constexpr ((w & 4) == 0) { lv[2] = x; return; }
        x = tree_combine<2>(lv[2], x, lane);
        if constexpr ((w & 8) == 0) { lv[3] = x; return; }
        x = tree_combine<3>(lv[3], x, lane);
        if constexpr ((w & 16) == 0) { lv[4] = x; return; }
        x = tree_combine<4>(lv[4], x, lane);
        if constexpr ((w & 32) == 0) { lv[5] = x; return; }
        x = tree_combine<5>(lv[5], x, lane);
        acc[V >> 6] += x;
    }
    // the last group of a row is incomplete when NVALUES is not a multiple of 64: its pending subtrees go up the
    // remaining levels against zeros (a value keeps the lane its number says)
    __device__ __forceinline__ void finish_row(int lane) {
        constexpr int rem = NVALUES & 63;
        if constexpr (rem != 0) {
            double x = 0;
            bool have = false;
#define DZO_TS_LEVEL(B)                                                                        \
            if constexpr ((rem >> B) & 1) { x = tree_combine<B>(lv[B], have ? x : 0.0, lane); have = true; } \
            else if (have) { x = tree_combine<B>(x, 0.0, lane); }
            DZO_TS_LEVEL(0) DZO_TS_LEVEL(1) DZO_TS_LEVEL(2) DZO_TS_LEVEL(3) DZO_TS_LEVEL(4) DZO_TS_LEVEL(5)
#undef DZO_TS_LEVEL
            acc[kGroups - 1] += x;
        }
    }
    // the value whose total this lane holds in acc[g]
    static __device__ __forceinline__ int value_of(int g, int lane) { return 64 * g + (int)(__builtin_bitreverse32((unsigned)lane) >> 26); }
};

// Wave-wide sums of EIGHT values with 10 cross-lane exchanges instead of 8 x 6 (transposed butterfly,
// like wave_sum5 in dzo_lbfgs.hip): three halving steps leave ONE value per lane, three plain steps
// finish it.  Afterwards every lane holds the total of value  4*(lane & 1) + 2*((lane >> 1) & 1) +
// ((lane >> 2) & 1).  Fixed association order -> deterministic.
__device__ __forceinline__ double wave_sum8(const double (&t)[8], int lane) {
    const bool A = (lane & 1) != 0, B = (lane & 2) != 0, C = (lane & 4) != 0;
    double a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (A ? t[i + 4] : t[i]) + lane_xor1(A ? t[i] : t[i + 4]);
    double b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = (B ? a[i + 2] : a[i]) + lane_xor2(B ? a[i] : a[i + 2]);
    double e = (C ? b[1] : b[0]) + lane_xor4(C ? b[0] : b[1]);
    e += lane_xor8(e);
    e = sum_xor16(e);
    return sum_xor32(e);
}
__device__ __forceinline__ int wave_sum8_owner(int lane) { return 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1); }

// Block sum for kBlock threads; `lds` holds kWaves doubles.  Result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double *lds) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) r += lds[w];
    }
    __syncthreads();
    return r;
}

// R block sums behind ONE pair of barriers (block_sum called R times pays 2 R); `lds` holds R * kWaves doubles.  Every
// sum is bit-identical to block_sum of the same value.  Results valid in thread 0.
template <int R> __device__ __forceinline__ void block_sum_multi(const double (&v)[R], double *lds, double (&out)[R]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double w = wave_sum(v[r]);
        if (lane == 0) lds[r * kWaves + wave] = w;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double t = 0;
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) t += lds[r * kWaves + w];
        }
        out[r] = t;
    }
    __syncthreads();
}

// Every thread of every block obtains the same sum of `count` per-block partials, read in
// a fixed order (the second stage of the two-stage reductions).  `lds` holds kWaves doubles.
__device__ __forceinline__ double reduce_partials_all(const double *__restrict__ partials,
                                                      int count, double *lds) {
    double v = 0;
    for (int i = threadIdx.x; i < count; i += kBlock) v += partials[i];
    v = wave_sum_all(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) r += lds[w];
    __syncthreads();
    return r;
}

// Raise a boolean flag in global memory when any thread of the block has `p` set.  ONE plain
// store per block: every writer stores the same value, so no atomic is needed -- thousands of
// atomics on one word serialise at ~12 ns each (MI355X_MICROARCH.md, row "fanin") and were
// measured to double the time of a 4n streaming kernel.
__device__ __forceinline__ void block_raise_flag(bool p, int32_t *flag, int *lds_flag) {
    if (threadIdx.x == 0) *lds_flag = 0;
    __syncthreads();
    if (__any(p) && (threadIdx.x & 63) == 0) *lds_flag = 1;
    __syncthreads();
    if (threadIdx.x == 0 && *lds_flag) *flag = 1;
}

// Base.isequal for floats: bitwise equal, except that every NaN equals every NaN.
__device__ __forceinline__ bool is_equal(double a, double b) {
    return (__double_as_longlong(a) == __double_as_longlong(b)) || (a != a && b != b);
}
__device__ __forceinline__ bool is_equal(float a, float b) {
    return (__float_as_int(a) == __float_as_int(b)) || (a != a && b != b);
}

// host-side dtype dispatch
#define DZO_DISPATCH(dtype, ...)                                   \
    do {                                                           \
        if ((dtype) == DZO_F64) { using T = double; __VA_ARGS__; } \
        else if ((dtype) == DZO_F32) { using T = float; __VA_ARGS__; } \
        else { ::dzo::set_error("bad dtype %d", (int)(dtype)); return DZO_ERR_INVALID; } \
    } while (0)

inline size_t dtype_size(int32_t dtype) { return dtype == DZO_F64 ? 8 : 4; }

// ------------------------------------------------------------------------------ shared launchers
// (defined in dzo_vec.hip; used by the optimizers)
template <typename T> void launch_axpy(hipStream_t s, int64_t n, T a, const T *x, T *y);
template <typename T> void launch_axpy_oop(hipStream_t s, int64_t n, T *dst, T a, const T *x, const T *y);
template <typename T> void launch_axpby(hipStream_t s, int64_t n, T a, const T *x, T b, T *y);
template <typename T> void launch_scal(hipStream_t s, int64_t n, T a, T *x);
template <typename T> void launch_scal_oop(hipStream_t s, int64_t n, T *dst, T a, const T *x);
template <typename T> void launch_fill(hipStream_t s, int64_t n, T a, T *x);
// two-stage dot: writes the final sum to result_dev[0]; partials_dev needs kMaxPartialBlocks doubles
template <typename T> void launch_dot(hipStream_t s, int64_t n, const T *x, const T *y,
                                      double *partials_dev, double *result_dev);
template <typename T> void launch_isequal(hipStream_t s, int64_t n, const T *a, const T *b, int32_t *differs_dev);

// blocking scalar helpers on a stream: run dot and copy the result to the host
int32_t dot_blocking(hipStream_t s, int64_t n, int32_t dtype, const void *x, const void *y,
                     double *partials_dev, double *host_pinned, double *out);

}  // namespace dzo
