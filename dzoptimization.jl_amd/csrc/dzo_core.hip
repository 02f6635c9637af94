// dzo_core.hip -- library context, error reporting, device memory and kernel timing.
#include <cstdarg>
#include <map>

#include "dzo_common.h"

#include <atomic>
#include <condition_variable>
#include <thread>

namespace dzo {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int32_t hip_fail(hipError_t e, const char *what, const char *file, int line) {
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    (void)hipGetLastError();    // (reported once: the runtime keeps the last error until somebody asks, and a later, unrelated hipGetLastError() check would report it again)
    return DZO_ERR_HIP;
}

static Context g_ctx[kMaxDevices];
static int g_default_device = -1;              // first device initialised in this process
static thread_local int tl_device = -1;        // device selected by this thread (dzo_init / DeviceScope)

Context &ctx() {
    const int d = tl_device >= 0 ? tl_device : g_default_device;
    return g_ctx[d >= 0 ? d : 0];
}

DeviceScope::DeviceScope(int device) {
    if (device < 0 || device >= kMaxDevices) return;
    prev_ctx = tl_device;
    if (hipGetDevice(&prev_hip) != hipSuccess) prev_hip = -1;
    if (prev_hip != device) (void)hipSetDevice(device);
    tl_device = device;
    active = true;
}

DeviceScope::~DeviceScope() {
    if (!active) return;
    tl_device = prev_ctx;
    int cur = -1;
    if (prev_hip >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev_hip) (void)hipSetDevice(prev_hip);
}

int32_t wait_ticket(hipStream_t s, const double *word, double ticket) {
    static const bool poll = getenv("DZO_TUNE_POLL") ? atoi(getenv("DZO_TUNE_POLL")) != 0 : true;
    if (!poll) { DZO_HIP(hipStreamSynchronize(s)); return DZO_OK; }
    volatile const double *h = word;
    for (uint64_t spins = 1;; ++spins) {
        if (*h == ticket) break;
        if ((spins & 0xFFFF) == 0) {
            hipError_t e = hipStreamQuery(s);
            if (e == hipSuccess) {                                   // everything ran: the ticket must be there
                if (*h == ticket) break;
                set_error("a kernel finished without publishing its results to the host");
                return DZO_ERR_HIP;
            }
            if (e != hipErrorNotReady) { DZO_HIP(e); }
        }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return DZO_OK;
}

static std::atomic<int64_t> g_unsealed_reads{0};
int64_t unsealed_first_reads() { return g_unsealed_reads.load(); }

int32_t wait_sealed(hipStream_t s, const double *data, int count, const double *seal, double ticket) {
    volatile const double *d = data;
    volatile const unsigned long long *sl = reinterpret_cast<volatile const unsigned long long *>(seal);
    unsigned long long tb;
    memcpy(&tb, &ticket, sizeof(tb));
    for (int64_t spins = 0;; ++spins) {
        unsigned long long x = tb;
        for (int i = 0; i < count; ++i) { const double v = d[i]; unsigned long long b; memcpy(&b, &v, sizeof(b)); x ^= b; }
        if (x == *sl) break;
        if (spins == 0) g_unsealed_reads += 1;
        if (spins == 1000000) DZO_HIP(hipStreamSynchronize(s));      // (everything the stream wrote is visible after this)
        if (spins >= 1001000) { set_error("results published to the host never matched their seal"); return DZO_ERR_HIP; }
        __builtin_ia32_pause();
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return DZO_OK;
}

int32_t require_init() {
    if (!ctx().ready) {
        set_error("dzo_init() has not been called (or no HIP device is available)");
        return DZO_ERR_STATE;
    }
    return DZO_OK;
}

// ---------------------------------------------------------------------------- twin-buffer registry
// g_unsettled: the handles whose live point / gradient sit in internal buffers.  g_settling: the handles some
// thread is settling right now OUTSIDE the registry lock (the settle function takes the handle's own mutex and
// waits on its stream).  A destroy on another thread -- Julia finalizers run on any thread -- must not free a
// handle that is in g_settling: unsettled_retire() removes the entry and waits until no other thread is inside
// a settle call on it.
static std::mutex g_unsettled_mu;
static std::condition_variable g_unsettled_cv;
static std::vector<std::pair<void *, int32_t (*)(void *)>> g_unsettled;
static std::vector<std::pair<void *, std::thread::id>> g_settling;

void unsettled_add(void *handle, int32_t (*settle)(void *)) {
    std::lock_guard<std::mutex> lk(g_unsettled_mu);
    for (auto &e : g_unsettled) if (e.first == handle) return;
    g_unsettled.emplace_back(handle, settle);
}

static void unsettled_erase_locked(void *handle) {
    for (size_t i = 0; i < g_unsettled.size(); ++i)
        if (g_unsettled[i].first == handle) { g_unsettled.erase(g_unsettled.begin() + (long)i); return; }
}

void unsettled_remove(void *handle) {
    std::lock_guard<std::mutex> lk(g_unsettled_mu);
    unsettled_erase_locked(handle);
}

void unsettled_retire(void *handle) {
    std::unique_lock<std::mutex> lk(g_unsettled_mu);
    unsettled_erase_locked(handle);
    const std::thread::id me = std::this_thread::get_id();
    g_unsettled_cv.wait(lk, [&] {
        for (auto &b : g_settling) if (b.first == handle && b.second != me) return false;
        return true;
    });
}

int32_t settle_all_optimizers() {
    const std::thread::id me = std::this_thread::get_id();
    for (int guard = 0; guard < 1 << 20; ++guard) {
        std::pair<void *, int32_t (*)(void *)> e;
        {
            std::unique_lock<std::mutex> lk(g_unsettled_mu);
            if (g_unsettled.empty()) return DZO_OK;
            e = g_unsettled.back();
            g_settling.emplace_back(e.first, me);        // the handle stays alive until this entry is gone
        }
        const int32_t rc = e.second(e.first);            // settles, waits for the copies, removes itself
        {
            std::lock_guard<std::mutex> lk(g_unsettled_mu);
            for (size_t i = 0; i < g_settling.size(); ++i)
                if (g_settling[i].first == e.first && g_settling[i].second == me) { g_settling.erase(g_settling.begin() + (long)i); break; }
            if (rc != DZO_OK) unsettled_erase_locked(e.first);
        }
        g_unsettled_cv.notify_all();
        if (rc != DZO_OK) return rc;
    }
    return DZO_OK;
}

// ---------------------------------------------------------------------------- backend assert (a8)
int32_t pointer_device(const void *p) {
    hipPointerAttribute_t a;
    if (p && hipPointerGetAttributes(&a, p) == hipSuccess && (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged)) return a.device;
    (void)hipGetLastError();
    return -1;
}

int32_t require_same_backend(const char *where, const char *cite, const void *a, const char *a_name, const void *b, const char *b_name) {
    const int want = ctx().device;
    const void *ptrs[2] = {a, b};
    const char *names[2] = {a_name, b_name};
    for (int i = 0; i < 2; ++i) {
        if (!ptrs[i]) continue;
        const int dev = pointer_device(ptrs[i]);
        if (dev < 0) {
            set_error("%s: @assert backend == get_backend(%s) (%s): %s is not device memory (a host pointer, or memory HIP does not know)",
                      where, names[i], cite, names[i]);
            return DZO_ERR_ASSERT;
        }
        if (dev != want) {
            set_error("%s: @assert backend == get_backend(%s) (%s): %s lives on device %d, the calling thread's library context is device %d",
                      where, names[i], cite, names[i], dev, want);
            return DZO_ERR_ASSERT;
        }
    }
    return DZO_OK;
}

// ---------------------------------------------------------------------------- profiling
// HIP-event pairs around kernel launches, recorded on the launching stream.  Events come from
// a pool (creating two events per launch cost ~7 % of a 1.3 ms step); level 1 times only the
// kernels whose name starts with "lbfgs_single_pass", "lbfgs_gram_pass", "lbfgs_combine", "lbfgs_chain", "bfgs_update",
// "bfgs_symv", "bfgs_scalars" or "bfgs_batch" (the roofline kernels: one bracket per step of the headline workload -- a
// bracket costs ~8 us of stream time, and with the 5-us reduce and 10-us finish kernels bracketed too the events
// took 5.5 % of a 0.41-ms step), level 2 times every kernel.
static int g_profile = 0;
static std::mutex g_profile_mu;
static std::vector<ProfileEntry> g_entries;
static std::map<std::string, int> g_index;
static std::vector<hipEvent_t> g_pool;
static thread_local int g_current = -1;
static thread_local hipEvent_t g_current_start = nullptr;

bool profiling_on() { return g_profile != 0; }

static bool is_roofline_kernel(const char *name) {
    static const char *keys[] = {"lbfgs_single_pass", "lbfgs_gram_pass", "lbfgs_combine", "lbfgs_chain", "bfgs_update", "bfgs_symv", "bfgs_scalars", "bfgs_batch"};
    for (const char *k : keys)
        if (strncmp(name, k, strlen(k)) == 0) return true;
    return false;
}

static hipEvent_t pool_get() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void profile_begin(const char *name, hipStream_t s, hipEvent_t *stop_out) {
    if (g_profile == 1 && !is_roofline_kernel(name)) return;
    std::lock_guard<std::mutex> lk(g_profile_mu);
    hipEvent_t a = pool_get(), b = pool_get();
    if (!a || !b) return;
    auto it = g_index.find(name);
    if (it == g_index.end()) {
        g_index[name] = (int)g_entries.size();
        g_entries.emplace_back();
        g_entries.back().name = name;
        g_current = (int)g_entries.size() - 1;
    } else {
        g_current = it->second;
    }
    g_current_start = a;
    (void)hipEventRecord(a, s);
    *stop_out = b;
}

void profile_end(hipEvent_t stop, hipStream_t s) {
    (void)hipEventRecord(stop, s);
    std::lock_guard<std::mutex> lk(g_profile_mu);
    if (g_current >= 0) g_entries[g_current].pending.emplace_back(g_current_start, stop);
}

static void profile_drain() {
    std::lock_guard<std::mutex> lk(g_profile_mu);
    for (auto &e : g_entries) {
        for (auto &p : e.pending) {
            (void)hipEventSynchronize(p.second);
            float ms = 0;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
                e.total_ms += ms;
                e.launches += 1;
            }
            g_pool.push_back(p.first);
            g_pool.push_back(p.second);
        }
        e.pending.clear();
    }
}

}  // namespace dzo

using namespace dzo;

extern "C" {

int32_t dzo_version(void) { return DZO_VERSION; }

const char *dzo_last_error(void) { return g_error; }

int32_t dzo_init(int32_t device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (hipGetDeviceCount: %s); this library has no CPU path",
                  hipGetErrorString(e));
        return DZO_ERR_HIP;
    }
    DZO_REQUIRE(device >= 0 && device < count && device < kMaxDevices, DZO_ERR_INVALID, "device %d out of range [0,%d)",
                device, count < kMaxDevices ? count : kMaxDevices);
    DZO_HIP(hipSetDevice(device));
    tl_device = device;                        // this thread now works on `device`
    static std::mutex init_mu;
    std::lock_guard<std::mutex> lk(init_mu);
    if (g_default_device < 0) g_default_device = device;
    Context &c = g_ctx[device];
    if (c.ready) return DZO_OK;
    hipDeviceProp_t prop;
    DZO_HIP(hipGetDeviceProperties(&prop, device));
    c.device = device;
    c.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c.hbm_bytes = (int64_t)prop.totalGlobalMem;
    snprintf(c.name, sizeof(c.name), "%s (%s)", prop.name, prop.gcnArchName);
    DZO_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    DZO_HIP(hipMalloc((void **)&c.scratch, sizeof(double) * (kMaxPartialBlocks + 8)));
    DZO_HIP(hipHostMalloc((void **)&c.host_scalar, sizeof(double) * 8, hipHostMallocDefault));
    c.ready = true;
    return DZO_OK;
}

int32_t dzo_shutdown(void) {
    for (int d = 0; d < kMaxDevices; ++d) {
        Context &c = g_ctx[d];
        if (!c.ready) continue;
        DeviceScope scope(d);
        (void)hipStreamSynchronize(c.stream);
        (void)hipFree(c.scratch);
        (void)hipHostFree(c.host_scalar);
        (void)hipStreamDestroy(c.stream);
        c = Context();
    }
    g_default_device = -1;
    tl_device = -1;
    return DZO_OK;
}

int32_t dzo_device_info(char *name, int32_t name_len, int32_t *compute_units, int64_t *hbm_bytes) {
    DZO_TRY(require_init());
    if (name && name_len > 0) {
        strncpy(name, ctx().name, (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (compute_units) *compute_units = ctx().cus;
    if (hbm_bytes) *hbm_bytes = ctx().hbm_bytes;
    return DZO_OK;
}

// the number of HIP devices this process sees (needs no dzo_init and creates no context: a host that drives one shard
// per GPU sizes its loop with it)
int32_t dzo_device_count(int32_t *count) {
    DZO_REQUIRE(count, DZO_ERR_INVALID, "null argument");
    int c = 0;
    DZO_HIP(hipGetDeviceCount(&c));
    *count = c;
    return DZO_OK;
}

int32_t dzo_synchronize(void) {
    DZO_TRY(require_init());
    DZO_TRY(settle_all_optimizers());
    DZO_HIP(hipDeviceSynchronize());
    return DZO_OK;
}

int32_t dzo_profile_enable(int32_t on) {
    g_profile = on < 0 ? 0 : (on > 2 ? 2 : on);
    return DZO_OK;
}

int32_t dzo_unsealed_first_reads(int64_t *count) {
    DZO_REQUIRE(count, DZO_ERR_INVALID, "null argument");
    *count = unsealed_first_reads();
    return DZO_OK;
}

int32_t dzo_profile_reset(void) {
    profile_drain();
    std::lock_guard<std::mutex> lk(g_profile_mu);
    for (auto &e : g_entries) {
        e.launches = 0;
        e.total_ms = 0;
    }
    return DZO_OK;
}

int32_t dzo_profile_count(int32_t *count) {
    profile_drain();
    std::lock_guard<std::mutex> lk(g_profile_mu);
    *count = (int32_t)g_entries.size();
    return DZO_OK;
}

int32_t dzo_profile_get(int32_t i, char *name, int32_t name_len, int64_t *launches, double *total_ms) {
    profile_drain();
    std::lock_guard<std::mutex> lk(g_profile_mu);
    DZO_REQUIRE(i >= 0 && i < (int32_t)g_entries.size(), DZO_ERR_INVALID, "profile index %d", i);
    if (name && name_len > 0) {
        strncpy(name, g_entries[i].name.c_str(), (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (launches) *launches = g_entries[i].launches;
    if (total_ms) *total_ms = g_entries[i].total_ms;
    return DZO_OK;
}

// device that owns a device pointer (-1: unknown / host memory -> stay on the current device)
static int device_of(const void *p) { return pointer_device(p); }

int32_t dzo_malloc(void **ptr_dev, int64_t bytes) {
    DZO_TRY(require_init());
    DZO_REQUIRE(ptr_dev && bytes >= 0, DZO_ERR_INVALID, "dzo_malloc: bad arguments");
    hipError_t e = hipMalloc(ptr_dev, (size_t)(bytes > 0 ? bytes : 16));
    if (e == hipErrorOutOfMemory) {
        set_error("hipMalloc(%lld bytes): out of device memory", (long long)bytes);
        (void)hipGetLastError(); return DZO_ERR_NOMEM;
    }
    DZO_HIP(e);
    return DZO_OK;
}

int32_t dzo_free(void *ptr_dev) {
    DZO_TRY(require_init());
    DeviceScope scope(device_of(ptr_dev));
    if (ptr_dev) DZO_HIP(hipFree(ptr_dev));
    return DZO_OK;
}

int32_t dzo_memcpy_h2d(void *dst_dev, const void *src_host, int64_t bytes) {
    DZO_TRY(require_init());
    DZO_TRY(settle_all_optimizers());          // (a write into an aliased array must land in the live copy)
    DeviceScope scope(device_of(dst_dev));
    DZO_HIP(hipDeviceSynchronize());
    if (bytes > 0) DZO_HIP(hipMemcpy(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice));
    DZO_HIP(hipDeviceSynchronize());
    return DZO_OK;
}

int32_t dzo_memcpy_d2h(void *dst_host, const void *src_dev, int64_t bytes) {
    DZO_TRY(require_init());
    DZO_TRY(settle_all_optimizers());
    DeviceScope scope(device_of(src_dev));
    // every library stream is non-blocking w.r.t. the null stream: drain the device first
    DZO_HIP(hipDeviceSynchronize());
    if (bytes > 0) DZO_HIP(hipMemcpy(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost));
    return DZO_OK;
}

int32_t dzo_memcpy_d2d(void *dst_dev, const void *src_dev, int64_t bytes) {
    DZO_TRY(require_init());
    DZO_TRY(settle_all_optimizers());
    DeviceScope scope(device_of(dst_dev));
    DZO_HIP(hipDeviceSynchronize());
    if (bytes > 0) DZO_HIP(hipMemcpy(dst_dev, src_dev, (size_t)bytes, hipMemcpyDeviceToDevice));
    DZO_HIP(hipDeviceSynchronize());
    return DZO_OK;
}

}  // extern "C"
