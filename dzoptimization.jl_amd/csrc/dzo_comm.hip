// dzo_comm.hip -- the ONE collective of the design: the global convergence flag of the batched /
// sharded mode, all-reduced (MIN over one int32 per rank) over RCCL / xGMI.
//
// "run multiple optimizers in parallel" (README.md:12): optimizer instances share nothing, so the
// data path has no collective at all; a host that shards instances over the GPUs
// of a node only needs to know when EVERY shard has terminated.  Two ways to build the communicator:
//   dzo_comm_init_all   one process drives several devices (a Julia host with one task per GPU):
//                       ncclCommInitAll over the listed devices;
//   dzo_comm_init_rank  one process per GPU (torchrun / MPI style): rank 0 calls
//                       dzo_comm_unique_id, the 128 bytes travel to the other ranks by whatever
//                       means the launcher has, every rank calls dzo_comm_init_rank.
// RCCL is loaded at first use with dlopen (librccl.so.1), so the library has no link-time
// dependency on it and shares the copy a host process may already have loaded (PyTorch bundles one).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <vector>

#include "dzo_common.h"

struct dzo_bfgs_batch_s;
namespace dzo {
// defined in dzo_batch.hip
int32_t batch_count_active_enqueue(dzo_bfgs_batch_s *b);
int32_t batch_count_active_finish(dzo_bfgs_batch_s *b, int64_t *active);
int batch_device(const dzo_bfgs_batch_s *b);
}  // namespace dzo

namespace dzo {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;                    // dlerror() of the failed dlopen / the missing symbol, captured when it happened
};

static Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            r.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            const char *e = dlerror();                      // (read once: the call clears it)
            r.why += std::string(r.why.empty() ? "" : "; ") + (e ? e : "dlopen failed");
        }
        if (!r.handle) return;
#define SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name))
        SYM(GetUniqueId, "ncclGetUniqueId");
        SYM(CommInitRank, "ncclCommInitRank");
        SYM(CommInitAll, "ncclCommInitAll");
        SYM(CommDestroy, "ncclCommDestroy");
        SYM(AllReduce, "ncclAllReduce");
        SYM(GroupStart, "ncclGroupStart");
        SYM(GroupEnd, "ncclGroupEnd");
        SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.AllReduce && r.GroupStart && r.GroupEnd &&
               r.GetErrorString;
        if (!r.ok) r.why = "librccl was loaded but lacks one of the nccl* entry points this library binds";
    });
    return r;
}

static int32_t require_rccl() {
    if (!rccl().ok) {
        set_error("RCCL is not available: %s", rccl().why.c_str());
        return DZO_ERR_UNSUPPORTED;
    }
    return DZO_OK;
}

#define DZO_NCCL(call)                                                                              \
    do {                                                                                            \
        ncclResult_t r__ = (call);                                                                  \
        if (r__ != ncclSuccess) {                                                                   \
            ::dzo::set_error("RCCL error %d (%s) in %s at %s:%d", (int)r__, rccl().GetErrorString(r__), #call, __FILE__, __LINE__); \
            return DZO_ERR_HIP;                                                                     \
        }                                                                                           \
    } while (0)

}  // namespace dzo

struct dzo_comm_s {
    int nranks = 0;                     // ranks of the communicator (GPUs taking part)
    int first_rank = 0;                 // rank of this process's first device
    std::vector<int> devices;           // local devices, one rank each
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<int32_t *> dev_flags;   // per local rank: [0] = send, [1] = recv
    int32_t *host = nullptr;            // pinned: [nlocal] send, [nlocal] recv
    int64_t collectives = 0;
};

using namespace dzo;

static int32_t comm_alloc_local(dzo_comm_s *c) {
    const size_t nl = c->devices.size();
    c->streams.assign(nl, nullptr);
    c->dev_flags.assign(nl, nullptr);
    DZO_HIP(hipHostMalloc((void **)&c->host, sizeof(int32_t) * 2 * nl, hipHostMallocDefault));
    for (size_t i = 0; i < nl; ++i) {
        DeviceScope scope(c->devices[i]);
        DZO_HIP(hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking));
        DZO_HIP(hipMalloc((void **)&c->dev_flags[i], sizeof(int32_t) * 2));
    }
    return DZO_OK;
}

extern "C" {

int32_t dzo_comm_unique_id(void *id128) {
    DZO_TRY(require_init());
    DZO_TRY(require_rccl());
    DZO_REQUIRE(id128, DZO_ERR_INVALID, "null id buffer");
    static_assert(sizeof(ncclUniqueId) == DZO_COMM_UNIQUE_ID_BYTES, "RCCL unique id size changed");
    ncclUniqueId id;
    DZO_NCCL(rccl().GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return DZO_OK;
}

int32_t dzo_comm_destroy(dzo_comm_t c) {
    if (!c) return DZO_OK;
    for (size_t i = 0; i < c->devices.size(); ++i) {
        DeviceScope scope(c->devices[i]);
        if (i < c->streams.size() && c->streams[i]) (void)hipStreamSynchronize(c->streams[i]);
        if (i < c->comms.size() && c->comms[i] && rccl().ok) (void)rccl().CommDestroy(c->comms[i]);
        if (i < c->dev_flags.size() && c->dev_flags[i]) (void)hipFree(c->dev_flags[i]);
        if (i < c->streams.size() && c->streams[i]) (void)hipStreamDestroy(c->streams[i]);
    }
    if (c->host) (void)hipHostFree(c->host);
    delete c;
    return DZO_OK;
}

int32_t dzo_comm_init_rank(const void *id128, int32_t nranks, int32_t rank, dzo_comm_t *out) {
    DZO_TRY(require_init());
    DZO_TRY(require_rccl());
    DZO_REQUIRE(id128 && out, DZO_ERR_INVALID, "null argument");
    DZO_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, DZO_ERR_INVALID, "rank %d out of range [0,%d)", rank, nranks);
    dzo_comm_s *c = new dzo_comm_s();
    c->nranks = nranks; c->first_rank = rank;
    c->devices.push_back(ctx().device);                   // the device this process selected with dzo_init
    c->comms.assign(1, nullptr);
    int32_t rc = comm_alloc_local(c);
    if (rc != DZO_OK) { dzo_comm_destroy(c); return rc; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    {
        DeviceScope scope(c->devices[0]);
        ncclResult_t r = rccl().CommInitRank(&c->comms[0], nranks, id, rank);
        if (r != ncclSuccess) {
            set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, rccl().GetErrorString(r));
            dzo_comm_destroy(c);
            return DZO_ERR_HIP;
        }
    }
    *out = c;
    return DZO_OK;
}

int32_t dzo_comm_init_all(const int32_t *devices, int32_t ndev, dzo_comm_t *out) {
    DZO_TRY(require_rccl());
    DZO_REQUIRE(devices && out && ndev >= 1, DZO_ERR_INVALID, "bad argument");
    int count = 0;
    DZO_HIP(hipGetDeviceCount(&count));
    for (int i = 0; i < ndev; ++i) {
        DZO_REQUIRE(devices[i] >= 0 && devices[i] < count, DZO_ERR_INVALID, "device %d out of range [0,%d)", devices[i], count);
        for (int j = 0; j < i; ++j) DZO_REQUIRE(devices[i] != devices[j], DZO_ERR_INVALID, "device %d listed twice", devices[i]);
    }
    const int prev = ctx().ready ? ctx().device : -1;
    for (int i = 0; i < ndev; ++i) DZO_TRY(dzo_init(devices[i]));   // a library context (stream, scratch) per device
    if (prev >= 0) DZO_TRY(dzo_init(prev)); else DZO_TRY(dzo_init(devices[0]));
    dzo_comm_s *c = new dzo_comm_s();
    c->nranks = ndev; c->first_rank = 0;
    c->devices.assign(devices, devices + ndev);
    c->comms.assign((size_t)ndev, nullptr);
    int32_t rc = comm_alloc_local(c);
    if (rc != DZO_OK) { dzo_comm_destroy(c); return rc; }
    std::vector<int> devs(devices, devices + ndev);
    ncclResult_t r = rccl().CommInitAll(c->comms.data(), ndev, devs.data());
    if (r != ncclSuccess) {
        set_error("ncclCommInitAll over %d devices failed: %s", ndev, rccl().GetErrorString(r));
        dzo_comm_destroy(c);
        return DZO_ERR_HIP;
    }
    *out = c;
    return DZO_OK;
}

int32_t dzo_comm_info(dzo_comm_t c, int32_t *nranks, int32_t *nlocal, int32_t *first_rank, int64_t *collectives) {
    DZO_REQUIRE(c, DZO_ERR_INVALID, "null communicator");
    if (nranks) *nranks = c->nranks;
    if (nlocal) *nlocal = (int32_t)c->devices.size();
    if (first_rank) *first_rank = c->first_rank;
    if (collectives) *collectives = c->collectives;
    return DZO_OK;
}

// MIN over all ranks of one int32 per rank.  local_flags has one entry per LOCAL rank (one for a
// process-per-GPU communicator).  Blocking: returns when *global_flag is known.
int32_t dzo_flag_allreduce_min_n(dzo_comm_t c, const int32_t *local_flags, int32_t nflags, int32_t *global_flag) {
    DZO_REQUIRE(c && local_flags && global_flag, DZO_ERR_INVALID, "null argument");
    DZO_REQUIRE((size_t)nflags == c->devices.size(), DZO_ERR_INVALID, "one flag per local rank of the communicator (%d given, %zu local ranks)",
                nflags, c->devices.size());
    return dzo_flag_allreduce_min(c, local_flags, global_flag);
}

int32_t dzo_flag_allreduce_min(dzo_comm_t c, const int32_t *local_flags, int32_t *global_flag) {
    DZO_REQUIRE(c && local_flags && global_flag, DZO_ERR_INVALID, "null argument");
    const size_t nl = c->devices.size();
    for (size_t i = 0; i < nl; ++i) c->host[i] = local_flags[i];
    for (size_t i = 0; i < nl; ++i) {
        DeviceScope scope(c->devices[i]);
        DZO_HIP(hipMemcpyAsync(c->dev_flags[i], c->host + i, sizeof(int32_t), hipMemcpyHostToDevice, c->streams[i]));
    }
    DZO_NCCL(rccl().GroupStart());
    for (size_t i = 0; i < nl; ++i) {
        DeviceScope scope(c->devices[i]);
        ncclResult_t r = rccl().AllReduce(c->dev_flags[i], c->dev_flags[i] + 1, 1, ncclInt32, ncclMin, c->comms[i], c->streams[i]);
        if (r != ncclSuccess) {
            (void)rccl().GroupEnd();
            set_error("ncclAllReduce failed on local rank %zu: %s", i, rccl().GetErrorString(r));
            return DZO_ERR_HIP;
        }
    }
    DZO_NCCL(rccl().GroupEnd());
    for (size_t i = 0; i < nl; ++i) {
        DeviceScope scope(c->devices[i]);
        DZO_HIP(hipMemcpyAsync(c->host + nl + i, c->dev_flags[i] + 1, sizeof(int32_t), hipMemcpyDeviceToHost, c->streams[i]));
    }
    for (size_t i = 0; i < nl; ++i) {
        DeviceScope scope(c->devices[i]);
        DZO_HIP(hipStreamSynchronize(c->streams[i]));
    }
    *global_flag = c->host[nl];                           // every rank holds the same value
    c->collectives += 1;
    return DZO_OK;
}

// all_done over every instance of every shard: the local shards count their live instances (the
// count kernels of all local devices run concurrently), then ONE 4-byte all-reduce.  comm may be NULL
// (single shard, no collective).
int32_t dzo_bfgs_batch_all_done(dzo_comm_t comm, const dzo_bfgs_batch_t *batches, int32_t nbatches, int32_t *all_done) {
    DZO_REQUIRE(batches && all_done && nbatches >= 1, DZO_ERR_INVALID, "bad argument");
    DZO_REQUIRE(!comm || (size_t)nbatches == comm->devices.size(), DZO_ERR_INVALID,
                "one batch per local rank of the communicator (%d given, %zu local ranks)", nbatches, comm ? comm->devices.size() : (size_t)0);
    for (int i = 0; i < nbatches; ++i) {
        DZO_REQUIRE(batches[i], DZO_ERR_INVALID, "null batch %d", i);
        DZO_REQUIRE(!comm || batch_device(batches[i]) == comm->devices[(size_t)i], DZO_ERR_INVALID,
                    "batch %d lives on device %d, local rank %d of the communicator on device %d", i, batch_device(batches[i]), i,
                    comm ? comm->devices[(size_t)i] : -1);
        DZO_TRY(batch_count_active_enqueue(batches[i]));
    }
    std::vector<int32_t> flags((size_t)nbatches);
    int32_t local_min = 1;
    for (int i = 0; i < nbatches; ++i) {
        int64_t active = 0;
        DZO_TRY(batch_count_active_finish(batches[i], &active));
        flags[(size_t)i] = active == 0 ? 1 : 0;
        if (!flags[(size_t)i]) local_min = 0;
    }
    if (!comm || comm->nranks == 1) { *all_done = local_min; return DZO_OK; }
    return dzo_flag_allreduce_min(comm, flags.data(), all_done);
}

}  // extern "C"
