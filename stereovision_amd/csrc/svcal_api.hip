// Host side of the calibration reprojection path: the C ABI of include/svcal.h.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "svcal.h"
#include "svcal_device.hpp"
#include "svicp.h"

using namespace svcal;

struct svcal_ctx {
    int device = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;
    std::string err;
    int V = 0, C = 0, k = 5, stereo = 0;
    bool have_obs = false, have_params = false, evaluated = false;
    double *world = nullptr, *left = nullptr, *right = nullptr, *params = nullptr;
    double *gram_view = nullptr, *gram_sum = nullptr, *residual = nullptr;
    size_t cap_params = 0;
    // the damped solve: per-view solutions and contributions, shared step, whole step, flags
    double *sv_W = nullptr, *sv_contrib = nullptr, *sv_dc = nullptr, *sv_step = nullptr, *sv_pred = nullptr;
    double *sv_csum = nullptr;       // the views' summed contribution to the shared system
    double *h_step = nullptr;        // pinned: the step and its tail as they come off the device
    double *h_params = nullptr;      // pinned: the parameter vector on its way to the device
    double *h_small = nullptr;       // pinned: a few doubles on their way back (|e|^2)
    int *sv_flags = nullptr;
    int sv_views = 0;
    bool solve_begun = false;        // svcal_solve_begin ran for the blocks now on the device
    double begun_lam = 0.0;          // the damping svcal_solve_begin was given
    bool imported = false;           // gram_sum holds sums over ALL ranks' views (svcal_shard_import)
    int freeze_intrinsics = 0;       // 1: update_intrinsics = false of doStereoCalibrate
};

namespace {
thread_local std::string g_cal_create_error;

// released on every path out of the scope that made them
struct ScopedEvent {
    hipEvent_t ev = nullptr;
    hipError_t create() { return hipEventCreate(&ev); }
    ~ScopedEvent() { if (ev) (void)hipEventDestroy(ev); }
    ScopedEvent() = default;
    ScopedEvent(const ScopedEvent &) = delete;
    ScopedEvent &operator=(const ScopedEvent &) = delete;
};
struct ScopedDevice {
    void *p = nullptr;
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    ~ScopedDevice() { if (p) (void)hipFree(p); }
    ScopedDevice() = default;
    ScopedDevice(const ScopedDevice &) = delete;
    ScopedDevice &operator=(const ScopedDevice &) = delete;
};

int cfail(svcal_ctx *c, int code, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    if (c) c->err = buf; else g_cal_create_error = buf;
    return code;
}
#define CHK(call)                                                            \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return cfail(ctx, SVICP_ERR_HIP, #call, e_);   \
    } while (0)

void free_obs(svcal_ctx *ctx)
{
    void *bufs[] = { ctx->world, ctx->left, ctx->right, ctx->gram_view, ctx->gram_sum, ctx->residual };
    for (void *b : bufs) if (b) (void)hipFree(b);
    ctx->world = ctx->left = ctx->right = ctx->gram_view = ctx->gram_sum = ctx->residual = nullptr;
    void *sv[] = { ctx->sv_W, ctx->sv_contrib, ctx->sv_dc, ctx->sv_step, ctx->sv_pred, ctx->sv_flags,
                   ctx->sv_csum };
    for (void *b : sv) if (b) (void)hipFree(b);
    ctx->sv_W = ctx->sv_contrib = ctx->sv_dc = ctx->sv_step = ctx->sv_pred = ctx->sv_csum = nullptr;
    ctx->sv_flags = nullptr;
    ctx->sv_views = 0;
    ctx->solve_begun = false;
    if (ctx->h_step) (void)hipHostFree(ctx->h_step);
    ctx->h_step = nullptr;
}

int param_count(const svcal_ctx *ctx, int k)
{
    const int nI = 4 + k;
    return ctx->stereo ? 2 * nI + 6 * ctx->V + 6 : nI + 6 * ctx->V;
}
}  // namespace

extern "C" {

const char *svcal_last_error(const svcal_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_cal_create_error.c_str();
}

int svcal_create(int device_id, svcal_ctx **out)
{
    svcal_ctx *ctx = nullptr;
    if (!out) return cfail(nullptr, SVICP_ERR_ARG, "svcal_create: out is NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return cfail(nullptr, SVICP_ERR_NO_DEVICE, "svcal_create: no HIP device is visible");
    if (device_id < 0 || device_id >= count)
        return cfail(nullptr, SVICP_ERR_ARG, "svcal_create: device_id out of range");
    hipDeviceProp_t prop;
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&prop, device_id) != hipSuccess)
        return cfail(nullptr, SVICP_ERR_HIP, "svcal_create: cannot query the device");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return cfail(nullptr, SVICP_ERR_NO_DEVICE, "svcal_create: the library holds gfx950 code only");
    ctx = new svcal_ctx();
    ctx->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return cfail(nullptr, SVICP_ERR_HIP, "hipStreamCreate", e); }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return SVICP_OK;
}

int svcal_destroy(svcal_ctx *ctx)
{
    if (!ctx) return SVICP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_obs(ctx);
    if (ctx->params) (void)hipFree(ctx->params);
    if (ctx->h_params) (void)hipHostFree(ctx->h_params);
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return SVICP_OK;
}

int svcal_set_observations(svcal_ctx *ctx, const double *world_xy, int corners,
                           const double *left_xy, const double *right_xy, int views)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (!world_xy || !left_xy || corners < 1 || views < 1)
        return cfail(ctx, SVICP_ERR_ARG, "svcal_set_observations: bad argument");
    CHK(hipSetDevice(ctx->device));
    CHK(hipStreamSynchronize(ctx->stream));
    free_obs(ctx);
    ctx->have_obs = ctx->have_params = ctx->evaluated = false;
    ctx->V = views; ctx->C = corners; ctx->stereo = right_xy ? 1 : 0;
    const size_t nobs = sizeof(double) * 2 * (size_t)views * corners;
    CHK(hipMalloc((void **)&ctx->world, sizeof(double) * 2 * corners));
    CHK(hipMalloc((void **)&ctx->left, nobs));
    CHK(hipMalloc((void **)&ctx->gram_view, sizeof(double) * 2 * kGram * (size_t)views));
    CHK(hipMalloc((void **)&ctx->gram_sum, sizeof(double) * 2 * kGram));
    CHK(hipMalloc((void **)&ctx->residual, nobs * (ctx->stereo ? 2 : 1)));
    CHK(hipMemcpyAsync(ctx->world, world_xy, sizeof(double) * 2 * corners, hipMemcpyHostToDevice, ctx->stream));
    CHK(hipMemcpyAsync(ctx->left, left_xy, nobs, hipMemcpyHostToDevice, ctx->stream));
    if (right_xy) {
        CHK(hipMalloc((void **)&ctx->right, nobs));
        CHK(hipMemcpyAsync(ctx->right, right_xy, nobs, hipMemcpyHostToDevice, ctx->stream));
    }
    CHK(hipMemsetAsync(ctx->gram_view, 0, sizeof(double) * 2 * kGram * (size_t)views, ctx->stream));
    CHK(hipMemsetAsync(ctx->gram_sum, 0, sizeof(double) * 2 * kGram, ctx->stream));
    CHK(hipStreamSynchronize(ctx->stream));
    ctx->have_obs = true;
    return SVICP_OK;
}

int svcal_set_params(svcal_ctx *ctx, const double *params, int k_size)
{
    if (!ctx || !params) return cfail(ctx, SVICP_ERR_ARG, "svcal_set_params: NULL argument");
    if (!ctx->have_obs) return cfail(ctx, SVICP_ERR_ARG, "svcal_set_params: set the observations first");
    if (k_size != 2 && k_size != 5)
        return cfail(ctx, SVICP_ERR_ARG, "svcal_set_params: k_size must be 2 or 5");
    if (!ctx->stereo && k_size != 5)
        return cfail(ctx, SVICP_ERR_ARG, "svcal_set_params: the monocular path has five distortion terms");
    CHK(hipSetDevice(ctx->device));
    const size_t bytes = sizeof(double) * (size_t)param_count(ctx, k_size);
    if (bytes > ctx->cap_params) {
        CHK(hipStreamSynchronize(ctx->stream));
        if (ctx->params) CHK(hipFree(ctx->params));
        if (ctx->h_params) CHK(hipHostFree(ctx->h_params));
        ctx->params = ctx->h_params = nullptr; ctx->cap_params = 0;
        CHK(hipMalloc((void **)&ctx->params, bytes));
        CHK(hipHostMalloc((void **)&ctx->h_params, bytes, 0));
        ctx->cap_params = bytes;
    }
    ctx->k = k_size;
    // through the context's own pinned buffer: the caller may reuse its vector at once, and
    // the copy to the device need not be waited for (a copy from pageable memory would be)
    CHK(hipStreamSynchronize(ctx->stream));   // the last upload has left the pinned buffer
    std::memcpy(ctx->h_params, params, bytes);
    CHK(hipMemcpyAsync(ctx->params, ctx->h_params, bytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->have_params = true;
    return SVICP_OK;
}

int svcal_eval(svcal_ctx *ctx)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (!ctx->have_params) return cfail(ctx, SVICP_ERR_ARG, "svcal_eval: set the parameters first");
    CHK(hipSetDevice(ctx->device));
    if (ctx->k == 5)
        hipLaunchKernelGGL(cal_eval_kernel<5>, dim3(ctx->V), dim3(kThreads), 0, ctx->stream,
                           ctx->params, ctx->V, ctx->C, ctx->stereo, ctx->world, ctx->left,
                           ctx->right, ctx->gram_view, ctx->residual);
    else
        hipLaunchKernelGGL(cal_eval_kernel<2>, dim3(ctx->V), dim3(kThreads), 0, ctx->stream,
                           ctx->params, ctx->V, ctx->C, ctx->stereo, ctx->world, ctx->left,
                           ctx->right, ctx->gram_view, ctx->residual);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cfail(ctx, SVICP_ERR_HIP, "cal_eval_kernel", e);
    hipLaunchKernelGGL(cal_reduce_kernel, dim3((ctx->stereo ? 2 : 1) * kGram), dim3(256), 0,
                       ctx->stream, ctx->gram_view, ctx->V, ctx->gram_sum);
    e = hipGetLastError();
    if (e != hipSuccess) return cfail(ctx, SVICP_ERR_HIP, "cal_reduce_kernel", e);
    ctx->evaluated = true;
    ctx->solve_begun = false;
    ctx->imported = false;
    return SVICP_OK;
}

int svcal_fetch(svcal_ctx *ctx, svcal_blocks *out)
{
    if (!ctx || !out) return cfail(ctx, SVICP_ERR_ARG, "svcal_fetch: NULL argument");
    if (!ctx->evaluated) return cfail(ctx, SVICP_ERR_ARG, "svcal_fetch: nothing evaluated yet");
    CHK(hipSetDevice(ctx->device));
    const int V = ctx->V, nI = 4 + ctx->k;
    std::vector<double> gv(2 * kGram * (size_t)V), gs(2 * kGram);
    CHK(hipMemcpyAsync(gv.data(), ctx->gram_view, sizeof(double) * gv.size(), hipMemcpyDeviceToHost, ctx->stream));
    CHK(hipMemcpyAsync(gs.data(), ctx->gram_sum, sizeof(double) * gs.size(), hipMemcpyDeviceToHost, ctx->stream));
    if (out->residual)
        CHK(hipMemcpyAsync(out->residual, ctx->residual,
                           sizeof(double) * 2 * (size_t)V * ctx->C * (ctx->stereo ? 2 : 1),
                           hipMemcpyDeviceToHost, ctx->stream));
    CHK(hipStreamSynchronize(ctx->stream));
    // entry (a, b), a <= b, of a 16 x 16 upper triangle stored row by row
    auto at = [](int a, int b) { if (a > b) std::swap(a, b); return a * 16 - (a * (a - 1)) / 2 + (b - a); };
    const double *GL = gs.data(), *GR = gs.data() + kGram;
    for (int a = 0; a < nI; ++a) {
        for (int b = 0; b < nI; ++b) {
            if (out->LL) out->LL[a * nI + b] = GL[at(a, b)];
            if (out->RR && ctx->stereo) out->RR[a * nI + b] = GR[at(a, b)];
        }
        if (out->gL) out->gL[a] = GL[at(a, 15)];
        if (ctx->stereo) {
            if (out->gR) out->gR[a] = GR[at(a, 15)];
            if (out->RS) for (int q = 0; q < 6; ++q) out->RS[a * 6 + q] = GR[at(a, nI + q)];
        }
    }
    if (ctx->stereo)
        for (int q = 0; q < 6; ++q) {
            if (out->gS) out->gS[q] = GR[at(nI + q, 15)];
            if (out->SS) for (int p = 0; p < 6; ++p) out->SS[q * 6 + p] = GR[at(nI + q, nI + p)];
        }
    for (int v = 0; v < V; ++v) {
        const double *G = gv.data() + (size_t)v * 2 * kGram;
        for (int q = 0; q < 6; ++q) {
            if (out->gE) out->gE[(size_t)v * 6 + q] = G[at(nI + q, 15)];
            if (out->EE) for (int p = 0; p < 6; ++p) out->EE[((size_t)v * 6 + q) * 6 + p] = G[at(nI + q, nI + p)];
            if (out->LE) for (int a = 0; a < nI; ++a) out->LE[((size_t)v * nI + a) * 6 + q] = G[at(a, nI + q)];
        }
    }
    out->sq_residual = GL[at(15, 15)] + (ctx->stereo ? GR[at(15, 15)] : 0.0);
    return SVICP_OK;
}

int svcal_sq_residual(svcal_ctx *ctx, double *out)
{
    if (!ctx || !out) return cfail(ctx, SVICP_ERR_ARG, "svcal_sq_residual: NULL argument");
    if (!ctx->evaluated) return cfail(ctx, SVICP_ERR_ARG, "svcal_sq_residual: nothing evaluated yet");
    CHK(hipSetDevice(ctx->device));
    if (!ctx->h_small) CHK(hipHostMalloc((void **)&ctx->h_small, sizeof(double) * 4, 0));
    double *e2 = ctx->h_small;                           // entry (15, 15) of each summed triangle
    e2[0] = e2[1] = 0.0;
    CHK(hipMemcpyAsync(&e2[0], ctx->gram_sum + kGram - 1, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->stereo)
        CHK(hipMemcpyAsync(&e2[1], ctx->gram_sum + 2 * kGram - 1, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CHK(hipStreamSynchronize(ctx->stream));
    *out = e2[0] + e2[1];
    return SVICP_OK;
}

namespace {
int ensure_solve_buffers(svcal_ctx *ctx)
{
    const int V = ctx->V;
    if (ctx->sv_views == V) return SVICP_OK;
    void *sv[] = { ctx->sv_W, ctx->sv_contrib, ctx->sv_dc, ctx->sv_step, ctx->sv_pred, ctx->sv_flags,
                   ctx->sv_csum };
    for (void *b : sv) if (b) CHK(hipFree(b));
    ctx->sv_W = ctx->sv_contrib = ctx->sv_dc = ctx->sv_step = ctx->sv_pred = ctx->sv_csum = nullptr;
    ctx->sv_flags = nullptr;
    ctx->sv_views = 0;
    // sized for the longest parameter vector (five distortion terms, stereo)
    CHK(hipMalloc((void **)&ctx->sv_W, sizeof(double) * (size_t)V * 6 * 10));
    CHK(hipMalloc((void **)&ctx->sv_contrib, sizeof(double) * (size_t)V * 9 * 10));
    CHK(hipMalloc((void **)&ctx->sv_dc, sizeof(double) * (kMaxShared + 1)));
    CHK(hipMalloc((void **)&ctx->sv_step, sizeof(double) * (2 * 9 + 6 * (size_t)V + 6 + 3)));
    if (ctx->h_step) CHK(hipHostFree(ctx->h_step));
    ctx->h_step = nullptr;
    CHK(hipHostMalloc((void **)&ctx->h_step, sizeof(double) * (2 * 9 + 6 * (size_t)V + 6 + 3), 0));
    CHK(hipMalloc((void **)&ctx->sv_pred, sizeof(double) * (size_t)V));
    CHK(hipMalloc((void **)&ctx->sv_flags, sizeof(int)));
    CHK(hipMalloc((void **)&ctx->sv_csum, sizeof(double) * kSolveSums));
    ctx->sv_views = V;
    return SVICP_OK;
}

// the views' blocks eliminated; their summed contribution to the shared system in sv_csum
int solve_first_half(svcal_ctx *ctx, double lam, const char *who)
{
    if (!ctx->evaluated) return cfail(ctx, SVICP_ERR_ARG, "svcal_solve: nothing evaluated yet");
    if (!(lam >= 0.0) || !std::isfinite(lam))
        return cfail(ctx, SVICP_ERR_ARG, "svcal_solve: the damping must be finite and >= 0");
    CHK(hipSetDevice(ctx->device));
    int rc = ensure_solve_buffers(ctx);
    if (rc) return rc;
    const int V = ctx->V, nI = 4 + ctx->k;
    CHK(hipMemsetAsync(ctx->sv_flags, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(cal_view_eliminate_kernel, dim3((V * (nI + 1) + 63) / 64), dim3(64), 0, ctx->stream,
                       ctx->gram_view, V, nI, lam, ctx->sv_W, ctx->sv_contrib, ctx->sv_flags);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cfail(ctx, SVICP_ERR_HIP, who, e);
    hipLaunchKernelGGL(cal_contrib_sum_kernel, dim3(kSolveSums), dim3(256), 0, ctx->stream, ctx->sv_contrib,
                       V, nI, ctx->sv_csum);
    e = hipGetLastError();
    if (e != hipSuccess) return cfail(ctx, SVICP_ERR_HIP, who, e);
    return SVICP_OK;
}

// the shared system solved from sv_csum, the views' steps from it; predicted[0]: the shared
// parameters' part of step^T (lam step - G), predicted[1]: this context's views' part
int solve_second_half(svcal_ctx *ctx, double lam, double *step_out, double predicted[2], const char *who)
{
    const int V = ctx->V, nI = 4 + ctx->k, P = param_count(ctx, ctx->k);
    const int blocks = (V + 63) / 64;
    hipLaunchKernelGGL(cal_shared_solve_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->gram_sum,
                       ctx->sv_csum, nI, ctx->stereo, ctx->freeze_intrinsics, lam, ctx->sv_dc, ctx->sv_flags);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cfail(ctx, SVICP_ERR_HIP, who, e);
    hipLaunchKernelGGL(cal_view_step_kernel, dim3(blocks), dim3(64), 0, ctx->stream, ctx->gram_view,
                       ctx->sv_W, ctx->sv_dc, V, nI, ctx->stereo, lam, ctx->sv_step, ctx->sv_pred);
    e = hipGetLastError();
    if (e != hipSuccess) return cfail(ctx, SVICP_ERR_HIP, who, e);
    hipLaunchKernelGGL(cal_step_tail_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->sv_pred, V, ctx->sv_dc,
                       ctx->sv_flags, ctx->sv_step + P);
    e = hipGetLastError();
    if (e != hipSuccess) return cfail(ctx, SVICP_ERR_HIP, who, e);
    // one copy into pinned memory: the step, then the two parts of the prediction and the flags
    CHK(hipMemcpyAsync(ctx->h_step, ctx->sv_step, sizeof(double) * (P + 3), hipMemcpyDeviceToHost, ctx->stream));
    CHK(hipStreamSynchronize(ctx->stream));
    const int flags = (int)ctx->h_step[P + 2];
    if (flags)
        return cfail(ctx, SVICP_ERR_SINGULAR,
                     flags & 2 ? "svcal_solve: the shared system is not positive definite"
                               : "svcal_solve: a view's 6 x 6 block is not positive definite");
    std::memcpy(step_out, ctx->h_step, sizeof(double) * P);
    predicted[0] = ctx->h_step[P];
    predicted[1] = ctx->h_step[P + 1];
    return SVICP_OK;
}
}  // namespace

int svcal_set_update_intrinsics(svcal_ctx *ctx, int update_intrinsics)
{
    if (!ctx) return SVICP_ERR_ARG;
    ctx->freeze_intrinsics = update_intrinsics ? 0 : 1;
    return SVICP_OK;
}

int svcal_solve_damped(svcal_ctx *ctx, double lam, double *step_out, double *predicted_out)
{
    if (!ctx || !step_out) return cfail(ctx, SVICP_ERR_ARG, "svcal_solve_damped: NULL argument");
    if (ctx->imported)
        return cfail(ctx, SVICP_ERR_ARG,
                     "svcal_solve_damped: the shared sums cover every rank's views (svcal_shard_import) but "
                     "only this rank's views are here; use svcal_solve_begin / svcal_solve_end");
    int rc = solve_first_half(ctx, lam, "svcal_solve_damped");
    if (rc) return rc;
    double pred[2];
    rc = solve_second_half(ctx, lam, step_out, pred, "svcal_solve_damped");
    if (rc) return rc;
    if (predicted_out) *predicted_out = pred[0] + pred[1];
    return SVICP_OK;
}

int svcal_lm_refine(svcal_ctx *ctx, double *params, int k_size, int max_iterations,
                    int *iterations_out, double *rms_out)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (!params || max_iterations < 0)
        return cfail(ctx, SVICP_ERR_ARG, "svcal_lm_refine: bad argument");
    if (!ctx->have_obs) return cfail(ctx, SVICP_ERR_ARG, "svcal_lm_refine: set the observations first");
    const int nI = 4 + k_size;
    const size_t P = (size_t)(ctx->stereo ? 2 * nI + 6 : nI) + 6 * (size_t)ctx->V;
    std::vector<double> step(P), trial(P);
    auto evaluate = [&](const double *p, double *sq) -> int {
        int rc = svcal_set_params(ctx, p, k_size);
        if (rc == SVICP_OK) rc = svcal_eval(ctx);
        if (rc == SVICP_OK) rc = svcal_sq_residual(ctx, sq);
        return rc;
    };
    double err2 = 0.0;
    int rc = evaluate(params, &err2);
    if (rc) return rc;
    double lam = 1.0e-3, v = 2.0;
    int it = 0;
    for (; it < max_iterations; ++it) {
        double denom = 0.0;
        rc = svcal_solve_damped(ctx, lam, step.data(), &denom);
        if (rc) return rc;
        double norm2 = 0.0;
        for (size_t i = 0; i < P; ++i) norm2 += step[i] * step[i];
        if (std::sqrt(norm2) < 1.0e-5) break;
        for (size_t i = 0; i < P; ++i) trial[i] = params[i] + step[i];
        double new2 = 0.0;
        rc = evaluate(trial.data(), &new2);
        if (rc) return rc;
        const double r = (err2 - new2) / denom;
        if (r > 0) {
            std::memcpy(params, trial.data(), sizeof(double) * P);
            err2 = new2;
            if (std::sqrt(err2) < 1.0e-3) { ++it; break; }
            const double c = 2.0 * r - 1.0;
            lam *= std::max(1.0 / 3.0, 1.0 - c * c * c);
            v = 2.0;
        } else {
            // the device holds the rejected trial's blocks: the loop needs the old ones back
            double again = 0.0;
            rc = evaluate(params, &again);
            if (rc) return rc;
            lam *= v;
            v *= 2.0;
        }
    }
    if (iterations_out) *iterations_out = it;
    if (rms_out) *rms_out = std::sqrt(err2);
    return SVICP_OK;
}

int svcal_solve_begin(svcal_ctx *ctx, double lam, double *dev_sums90)
{
    if (!ctx || !dev_sums90) return cfail(ctx, SVICP_ERR_ARG, "svcal_solve_begin: NULL argument");
    int rc = solve_first_half(ctx, lam, "svcal_solve_begin");
    if (rc) return rc;
    CHK(hipMemcpyAsync(dev_sums90, ctx->sv_csum, sizeof(double) * kSolveSums, hipMemcpyDeviceToDevice,
                       ctx->stream));
    ctx->solve_begun = true;
    ctx->begun_lam = lam;
    return SVICP_OK;
}

int svcal_solve_end(svcal_ctx *ctx, double lam, const double *dev_sums90, double *step_out,
                    double predicted_out[2])
{
    if (!ctx || !dev_sums90 || !step_out || !predicted_out)
        return cfail(ctx, SVICP_ERR_ARG, "svcal_solve_end: NULL argument");
    if (!ctx->solve_begun) return cfail(ctx, SVICP_ERR_ARG, "svcal_solve_end: call svcal_solve_begin first");
    if (!(lam >= 0.0) || !std::isfinite(lam))
        return cfail(ctx, SVICP_ERR_ARG, "svcal_solve_end: the damping must be finite and >= 0");
    if (lam != ctx->begun_lam)
        return cfail(ctx, SVICP_ERR_ARG, "svcal_solve_end: the damping differs from svcal_solve_begin's");
    CHK(hipSetDevice(ctx->device));
    ctx->solve_begun = false;
    CHK(hipMemcpyAsync(ctx->sv_csum, dev_sums90, sizeof(double) * kSolveSums, hipMemcpyDeviceToDevice,
                       ctx->stream));
    return solve_second_half(ctx, lam, step_out, predicted_out, "svcal_solve_end");
}

int svcal_set_stream(svcal_ctx *ctx, void *hip_stream, int use_own)
{
    if (!ctx) return SVICP_ERR_ARG;
    CHK(hipSetDevice(ctx->device));
    CHK(hipStreamSynchronize(ctx->stream));
    ctx->stream = use_own ? ctx->own_stream : (hipStream_t)hip_stream;
    return SVICP_OK;
}

int svcal_shard_export(svcal_ctx *ctx, double *dev_sums272)
{
    if (!ctx || !dev_sums272) return cfail(ctx, SVICP_ERR_ARG, "svcal_shard_export: NULL argument");
    if (!ctx->evaluated) return cfail(ctx, SVICP_ERR_ARG, "svcal_shard_export: nothing evaluated yet");
    CHK(hipSetDevice(ctx->device));
    CHK(hipMemcpyAsync(dev_sums272, ctx->gram_sum, sizeof(double) * 2 * kGram, hipMemcpyDeviceToDevice,
                       ctx->stream));
    return SVICP_OK;
}

int svcal_shard_import(svcal_ctx *ctx, const double *dev_sums272)
{
    if (!ctx || !dev_sums272) return cfail(ctx, SVICP_ERR_ARG, "svcal_shard_import: NULL argument");
    if (!ctx->evaluated) return cfail(ctx, SVICP_ERR_ARG, "svcal_shard_import: nothing evaluated yet");
    CHK(hipSetDevice(ctx->device));
    CHK(hipMemcpyAsync(ctx->gram_sum, dev_sums272, sizeof(double) * 2 * kGram, hipMemcpyDeviceToDevice,
                       ctx->stream));
    ctx->imported = true;
    return SVICP_OK;
}

namespace {
// general 3x3 inverse, Gaussian elimination with partial pivoting (the reference calls
// Eigen's dynamic-size inverse() per pixel, CStereoCameraRectification.cpp:82, :111)
bool invert3(const double A[9], double out[9])
{
    double M[3][6];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 6; ++c) M[r][c] = c < 3 ? A[3 * r + c] : (c - 3 == r ? 1.0 : 0.0);
    for (int c = 0; c < 3; ++c) {
        int p = c;
        for (int r = c + 1; r < 3; ++r) if (std::fabs(M[r][c]) > std::fabs(M[p][c])) p = r;
        if (M[p][c] == 0.0) return false;
        if (p != c) for (int k = 0; k < 6; ++k) std::swap(M[c][k], M[p][k]);
        for (int r = 0; r < 3; ++r) {
            if (r == c) continue;
            const double f = M[r][c] / M[c][c];
            for (int k = 0; k < 6; ++k) M[r][k] -= f * M[c][k];
        }
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out[3 * r + c] = M[r][3 + c] / M[r][r];
    return true;
}
}  // namespace

namespace {
// 3 x 3 helpers of svcal_stereo_rectify (row-major)
void mat3_mul(const double *a, const double *b, double *c)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
bool mat3_inv(const double *a, double *inv)
{
    const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c0 + a[1] * c1 + a[2] * c2;
    if (!(std::fabs(det) > 1e-300)) return false;
    const double d = 1.0 / det;
    inv[0] = c0 * d; inv[1] = (a[2] * a[7] - a[1] * a[8]) * d; inv[2] = (a[1] * a[5] - a[2] * a[4]) * d;
    inv[3] = c1 * d; inv[4] = (a[0] * a[8] - a[2] * a[6]) * d; inv[5] = (a[2] * a[3] - a[0] * a[5]) * d;
    inv[6] = c2 * d; inv[7] = (a[1] * a[6] - a[0] * a[7]) * d; inv[8] = (a[0] * a[4] - a[1] * a[3]) * d;
    return true;
}
// rotation matrix -> rotation vector (Common/Misc.hpp:36-70).  The reference first replaces
// the matrix by U V^T of its SVD, the nearest orthogonal matrix; the same matrix is the limit
// of X <- (X + X^-T) / 2, which gets there in a few steps from anything near a rotation.
bool rotation_vector(const double *Rin, double *r)
{
    double R[9];
    std::memcpy(R, Rin, sizeof R);
    for (int it = 0; it < 30; ++it) {
        double inv[9];
        if (!mat3_inv(R, inv)) return false;
        double change = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const double x = 0.5 * (R[3 * i + j] + inv[3 * j + i]);
                change = std::max(change, std::fabs(x - R[3 * i + j]));
                R[3 * i + j] = x;
            }
        if (change < 1e-16) break;
    }
    r[0] = R[7] - R[5]; r[1] = R[2] - R[6]; r[2] = R[3] - R[1];
    const double s = std::sqrt((r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : c < -1.0 ? -1.0 : c;
    double theta = std::acos(c);
    if (s < 1.0e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0.0; return true; }
        r[0] = std::sqrt(std::max((R[0] + 1.0) * 0.5, 0.0));
        r[1] = std::sqrt(std::max((R[4] + 1.0) * 0.5, 0.0)) * (R[1] < 0 ? -1.0 : 1.0);
        r[2] = std::sqrt(std::max((R[8] + 1.0) * 0.5, 0.0)) * (R[2] < 0 ? -1.0 : 1.0);
        if (std::fabs(r[0]) < std::fabs(r[1]) && std::fabs(r[0]) < std::fabs(r[2]) && (R[5] > 0) != (r[1] * r[2] > 0))
            r[2] = -r[2];
        theta /= std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        for (int i = 0; i < 3; ++i) r[i] *= theta;
        return true;
    }
    const double vth = theta / (2.0 * s);
    for (int i = 0; i < 3; ++i) r[i] *= vth;
    return true;
}
// rotation vector -> matrix (Common/Misc.hpp:11-32)
void rotation_matrix(const double *r, double *R)
{
    const double theta = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (theta < 1.0e-6) return;
    const double a[3] = { r[0] / theta, r[1] / theta, r[2] / theta };
    const double K[9] = { 0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0 };
    double K2[9];
    mat3_mul(K, K, K2);
    const double sn = std::sin(theta), cs = 1.0 - std::cos(theta);
    for (int i = 0; i < 9; ++i) R[i] += K[i] * sn + K2[i] * cs;
}
}  // namespace

int svcal_stereo_rectify(const double KL[9], const double KR[9], const double R[9], const double T[3],
                         double left_proj[9], double right_proj[9], double Q[16])
{
    if (!KL || !KR || !R || !T || !left_proj || !right_proj || !Q) return SVICP_ERR_ARG;
    double Rinv[9], eul[3], half[9], peer[9];
    if (!mat3_inv(R, Rinv) || !rotation_vector(Rinv, eul)) return SVICP_ERR_SINGULAR;
    for (int i = 0; i < 3; ++i) eul[i] *= -0.5;
    rotation_matrix(eul, half);
    mat3_mul(half, Rinv, peer);
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = -(peer[3 * i] * T[0] + peer[3 * i + 1] * T[1] + peer[3 * i + 2] * T[2]);
    const double n1 = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]), n2 = std::sqrt(t[0] * t[0] + t[1] * t[1]);
    if (!(n1 > 0.0) || !(n2 > 0.0)) return SVICP_ERR_SINGULAR;
    const double e1[3] = { t[0] / n1, t[1] / n1, t[2] / n1 }, e2[3] = { -t[1] / n2, t[0] / n2, 0.0 };
    const double e3[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
    const double rect[9] = { e1[0], e1[1], e1[2], e2[0], e2[1], e2[2], e3[0], e3[1], e3[2] };
    const double f = (KL[0] + KL[4] + KR[0] + KR[4]) * 0.25;
    const double newK[9] = { f, 0, (KL[2] + KR[2]) * 0.5, 0, f, (KL[5] + KR[5]) * 0.5, 0, 0, 1 };
    double a[9];
    mat3_mul(rect, half, a);
    mat3_mul(newK, a, left_proj);
    mat3_mul(rect, peer, a);
    mat3_mul(newK, a, right_proj);
    const double Tx = rect[0] * t[0] + rect[1] * t[1] + rect[2] * t[2];
    for (int i = 0; i < 16; ++i) Q[i] = (i % 5 == 0) ? 1.0 : 0.0;
    Q[3] = -newK[2]; Q[7] = -newK[5]; Q[10] = 0.0; Q[11] = f; Q[14] = -1.0 / Tx; Q[15] = 0.0;
    return SVICP_OK;
}

int svcal_rectify_maps(svcal_ctx *ctx, int width, int height, const double KL[9], const double *distL,
                       const double KR[9], const double *distR, int k_size, const double projL[9],
                       const double projR[9], double *xmap_left, double *ymap_left,
                       double *xmap_right, double *ymap_right, double *kernel_ms)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (width <= 0 || height <= 0 || !(k_size == 2 || k_size == 5) || !KL || !KR || !distL || !distR ||
        !projL || !projR || !xmap_left || !ymap_left || !xmap_right || !ymap_right)
        return cfail(ctx, SVICP_ERR_ARG, "svcal_rectify_maps: bad argument");
    RectifyCam L{}, R{};
    for (int k = 0; k < 9; ++k) { L.K[k] = KL[k]; R.K[k] = KR[k]; }
    for (int k = 0; k < k_size; ++k) { L.d[k] = distL[k]; R.d[k] = distR[k]; }
    if (!invert3(projL, L.Pinv) || !invert3(projR, R.Pinv))
        return cfail(ctx, SVICP_ERR_SINGULAR, "svcal_rectify_maps: a projection matrix is singular");
    CHK(hipSetDevice(ctx->device));
    const size_t px = (size_t)width * height, bytes = sizeof(double) * px;
    ScopedDevice buf;
    CHK(buf.alloc(4 * bytes));
    double *dev = static_cast<double *>(buf.p);
    ScopedEvent e0, e1;
    CHK(e0.create());
    CHK(e1.create());
    const dim3 grid((unsigned)((px + 255) / 256)), block(256);
    CHK(hipEventRecord(e0.ev, ctx->stream));
    if (k_size == 5)
        hipLaunchKernelGGL(rectify_maps_kernel<5>, grid, block, 0, ctx->stream, L, R, width, height,
                           dev, dev + px, dev + 2 * px, dev + 3 * px);
    else
        hipLaunchKernelGGL(rectify_maps_kernel<2>, grid, block, 0, ctx->stream, L, R, width, height,
                           dev, dev + px, dev + 2 * px, dev + 3 * px);
    CHK(hipGetLastError());
    CHK(hipEventRecord(e1.ev, ctx->stream));
    double *outs[4] = { xmap_left, ymap_left, xmap_right, ymap_right };
    hipError_t err = hipSuccess;
    for (int q = 0; q < 4 && err == hipSuccess; ++q)
        err = hipMemcpyAsync(outs[q], dev + q * px, bytes, hipMemcpyDeviceToHost, ctx->stream);
    // the copies already queued read the buffer: wait before it is released, whatever happened
    const hipError_t sync = hipStreamSynchronize(ctx->stream);
    if (err != hipSuccess) return cfail(ctx, SVICP_ERR_HIP, "svcal_rectify_maps: copy", err);
    if (sync != hipSuccess) return cfail(ctx, SVICP_ERR_HIP, "svcal_rectify_maps: sync", sync);
    float ms = 0.f;
    CHK(hipEventElapsedTime(&ms, e0.ev, e1.ev));
    if (kernel_ms) *kernel_ms = ms;
    return SVICP_OK;
}

int svcal_time_eval(svcal_ctx *ctx, int reps, double *avg_ms)
{
    if (!ctx || !avg_ms || reps < 1) return cfail(ctx, SVICP_ERR_ARG, "svcal_time_eval: bad argument");
    int rc = svcal_eval(ctx);   // warm-up
    if (rc) return rc;
    ScopedEvent a, b;
    CHK(a.create());
    CHK(b.create());
    CHK(hipEventRecord(a.ev, ctx->stream));
    for (int r = 0; r < reps; ++r) {
        rc = svcal_eval(ctx);
        if (rc) return rc;
    }
    CHK(hipEventRecord(b.ev, ctx->stream));
    CHK(hipEventSynchronize(b.ev));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, a.ev, b.ev));
    *avg_ms = (double)ms / reps;
    return SVICP_OK;
}

}  // extern "C"
