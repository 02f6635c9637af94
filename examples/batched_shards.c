/* batched_shards.c -- "run multiple optimizers in parallel" (README.md:12) from ONE host process over every GPU it
 * sees, through the plain C ABI of include/dzo.h (single process, one stream per device, the
 * communicator from ncclCommInitAll; no Python, no PyTorch, no launcher):
 *
 *   - B independent dense-BFGS optimizers per GPU (legacy/DZOptimization.jl:733-994 each), chained Rosenbrock,
 *     instance i of the job starts from its own point -- block partition of the instances over the devices;
 *   - no data-path collective: the only exchange is the global convergence flag, one all-reduce(MIN) of an int32
 *     over RCCL / xGMI every POLL steps (dzo_bfgs_batch_all_done);
 *   - the loop of the reference, `while !all(opt.has_converged[] for opt in opts); step!.(opts); end`.
 *
 *   gcc -O2 -Iinclude examples/batched_shards.c -Ldzoptimization.jl_amd -ldzo_hip \
 *       -Wl,-rpath,$PWD/dzoptimization.jl_amd -o batched_shards && ./batched_shards [instances_per_gpu [n]]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "dzo.h"

#define MAX_DEV 16
#define POLL 10

#define CHECK(call)                                                                  \
    do {                                                                             \
        int32_t rc_ = (call);                                                        \
        if (rc_ != DZO_OK) {                                                         \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, dzo_last_error());   \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char **argv) {
    const int64_t B = argc > 1 ? atoll(argv[1]) : 64;          /* instances per GPU */
    const int64_t n = argc > 2 ? atoll(argv[2]) : 16;          /* even, 2..1024 */
    int32_t ndev = 0;
    CHECK(dzo_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no HIP device\n"); return 1; }
    if (ndev > MAX_DEV) ndev = MAX_DEV;

    int32_t devices[MAX_DEV];
    dzo_bfgs_batch_t shard[MAX_DEV];
    void *x0_dev[MAX_DEV];
    double *x0 = (double *)malloc((size_t)(B * n) * sizeof(double));
    if (!x0) return 1;
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    for (int d = 0; d < ndev; ++d) {
        devices[d] = d;
        CHECK(dzo_init(d));                                    /* this thread now allocates on device d */
        for (int64_t i = 0; i < B * n; ++i) {                  /* instance (d, b): its own start in [0, 1)^n */
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            x0[i] = (double)(lcg >> 11) * (1.0 / 9007199254740992.0);
        }
        CHECK(dzo_malloc(&x0_dev[d], B * n * (int64_t)sizeof(double)));
        CHECK(dzo_memcpy_h2d(x0_dev[d], x0, B * n * (int64_t)sizeof(double)));
        CHECK(dzo_bfgs_batch_create_on(d, DZO_PROBLEM_ROSENBROCK_CHAIN, B, n, DZO_F64, x0_dev[d], 1.0, &shard[d]));
    }
    dzo_comm_t comm = NULL;
    CHECK(dzo_comm_init_all(devices, ndev, &comm));            /* one process, ndev local ranks */

    int32_t all_done = 0;
    int64_t rounds = 0;
    const double t0 = now_s();
    while (!all_done && rounds < 2000) {
        for (int d = 0; d < ndev; ++d) CHECK(dzo_bfgs_batch_step(shard[d], POLL, NULL));   /* enqueue only: the shards run side by side */
        CHECK(dzo_bfgs_batch_all_done(comm, shard, ndev, &all_done));                      /* local counts + one 4-byte all-reduce */
        ++rounds;
    }
    const double el = now_s() - t0;

    /* every instance must have reached the minimiser (1, ..., 1) */
    double worst = 0;
    int64_t steps = 0;
    int64_t *iters = (int64_t *)malloc((size_t)B * sizeof(int64_t));
    if (!iters) return 1;
    for (int d = 0; d < ndev; ++d) {
        void *xp = NULL, *ip = NULL;
        CHECK(dzo_bfgs_batch_get_ptr(shard[d], 0, &xp));
        CHECK(dzo_memcpy_d2h(x0, xp, B * n * (int64_t)sizeof(double)));
        for (int64_t i = 0; i < B * n; ++i) worst = fmax(worst, fabs(x0[i] - 1.0));
        CHECK(dzo_bfgs_batch_get_ptr(shard[d], 5, &ip));
        CHECK(dzo_memcpy_d2h(iters, ip, B * (int64_t)sizeof(int64_t)));
        for (int64_t b = 0; b < B; ++b) steps += iters[b];
    }
    int32_t nranks = 0, nlocal = 0;
    int64_t collectives = 0;
    CHECK(dzo_comm_info(comm, &nranks, &nlocal, NULL, &collectives));
    printf("%d GPU(s) x %lld instances, n = %lld: all_done = %d after %lld rounds of %d steps, %lld instance-steps in %.3f s "
           "(%.0f /s), %lld all-reduces over %d ranks (%d local), max |x - 1| = %.3e\n",
           ndev, (long long)B, (long long)n, all_done, (long long)rounds, POLL, (long long)steps, el, (double)steps / el,
           (long long)collectives, nranks, nlocal, worst);
    for (int d = 0; d < ndev; ++d) {
        CHECK(dzo_bfgs_batch_destroy(shard[d]));
        CHECK(dzo_init(d));
        CHECK(dzo_free(x0_dev[d]));
    }
    CHECK(dzo_comm_destroy(comm));
    free(iters);
    free(x0);
    if (!all_done || worst > 1e-6) { printf("FAILED\n"); return 1; }
    printf("OK\n");
    return 0;
}
