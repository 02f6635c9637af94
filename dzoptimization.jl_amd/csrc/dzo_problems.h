// dzo_problems.h -- internal interface of the built-in objectives (see dzo_problems.hip).
#pragma once
#include "dzo_common.h"

struct dzo_problem_s {
    int32_t kind = 0;
    int64_t n = 0;
    int32_t dtype = DZO_F64;
    const void *A = nullptr;   // QUADRATIC: n x n column-major symmetric, device
    const void *c = nullptr;   // LSE: centre, device
    double lambda = 0;
    // decorators (legacy/DZOptimization.jl:219-296)
    double l2 = 0;                         // L2 wrappers' lambda; 0 = off
    bool bg_on = false; double bg_lo = 0, bg_hi = 0;       // UniformBoxGradientWrapper
    bool cons_on = false; double cons_lo = 0, cons_hi = 0; // UniformBoxConstraint as constraint_function!
    // Workspace.  A problem handle that an optimizer was created from is never used directly by that
    // optimizer: each optimizer works on its own VIEW (same definition, private scratch / result /
    // host scalars; `parent` points at the user's handle), so several optimizers built on one
    // dzo_problem_t can run on their own streams / host threads without sharing partial sums.
    dzo_problem_s *parent = nullptr;
    double *scratch = nullptr; // device partials
    int64_t scratch_doubles = 0;   // usable partials in scratch
    double *result = nullptr;  // device: [f, ...] inside scratch
    double *host = nullptr;    // pinned host scalars
    double *tri_part = nullptr; // QUADRATIC: row / column parts of the lower-triangle evaluations (quadratic_tri6_kernel)
};

namespace dzo {
// per-optimizer view of a user's problem handle (see dzo_problem_s::parent); sync re-reads the
// definition and the decorators from the parent (they may be changed between steps)
int32_t problem_view_create(dzo_problem_s *parent, dzo_problem_s **out);
void problem_view_sync(dzo_problem_s *view);
void problem_view_destroy(dzo_problem_s *view);
// Enqueue f(x) on `s`; result_dev[0] receives the value (fp64, rounded to T by the caller).
int32_t problem_eval_async(dzo_problem_s *p, hipStream_t s, const void *x, double *result_dev);
// Objective partials only (no final sum); false if this objective needs its own finish kernel.
bool problem_eval_partials_async(dzo_problem_s *p, hipStream_t s, const void *x, const double **partials,
                                 int64_t *count, double *scale);
// x = fma(t, d, x_old) + changed flag + objective partials of the new point in ONE pass; false when
// this objective / these operands have no fused kernel (the caller then runs trial + objective).
bool problem_trial_eval_async(dzo_problem_s *p, hipStream_t s, void *x, void *backup, const void *d, double t,
                              bool first, int32_t *changed, const double **partials, int64_t *count, double *scale);
// f(x + ts*dir) without a trial-point launch (dense quadratic): point to point_out; value to
// result_dev[0] and, when `flags` (3 zeroed int32 on the device) is given, {point != x, dir != 0,
// point != ref} to the int32 view of result_dev[4..5] (result_dev may be pinned host memory); the
// device flags are re-armed.  false when there is no such kernel.
bool problem_phi_async(dzo_problem_s *p, hipStream_t s, const void *x, const void *dir, double ts, void *point_out,
                       int32_t *flags, double *result_dev, const void *ref = nullptr);
// up to three step sizes along one direction, evaluated in the same pass over A
struct PhiDirHost {
    const void *dir = nullptr;
    double ts[3] = {0, 0, 0};
    void *point_out[3] = {nullptr, nullptr, nullptr};
    const void *ref[3] = {nullptr, nullptr, nullptr};
    int ref_req[3] = {-1, -1, -1};
    void *grad_out[3] = {nullptr, nullptr, nullptr};
    bool active[3] = {false, false, false};
};
// the requests of one launch as the DEVICE posts them (device-driven line searches of the dense BFGS step): step sizes
// (signed, already rounded to the dtype), which of the six run, and what each one's :150 test compares with
struct PhiReqDev {
    double ts[2][3];
    int32_t active[2][3];
    int32_t ref_req[2][3];      // the request of the same direction in this launch to compare with, or -1
    int32_t use_ref[2][3];      // [side][0]: 0 = no stored reference point, 1 + b = the direction's trial-point buffer b (point_out[b]) as it stands
};
// dreq: take step sizes / activity / references from there (the buffers still from req; req's ref pointers are not used) and
// leave the finish to the caller
bool problem_phi6_async(dzo_problem_s *p, hipStream_t s, const void *x, const PhiDirHost req[2], int32_t *flags, double *result_dev, double ticket,
                        const PhiReqDev *dreq = nullptr);   // result_dev[20] <- ticket, last
// Enqueue g = grad f(x) on `s`.
int32_t problem_grad_async(dzo_problem_s *p, hipStream_t s, void *g, const void *x);
// Fused accept + gradient + delta_gradient + rho partials (chained Rosenbrock, aligned operands).
bool problem_has_fused_post(const dzo_problem_s *p, const void *x, const void *dx, const void *g, const void *dg);
int32_t problem_fused_post_async(dzo_problem_s *p, hipStream_t s, const void *x, void *dx, void *g, void *dg,
                                 double *partials, int *grid_out, const int32_t *gate = nullptr, const void *xold_src = nullptr,
                                 const void *gold_src = nullptr);
// x[i] = clamp(x[i], lo, hi) on stream s (UniformBoxConstraint, legacy :264-272)
int32_t box_clamp_async(hipStream_t s, int64_t n, int32_t dtype, void *x, double lo, double hi);
}  // namespace dzo
