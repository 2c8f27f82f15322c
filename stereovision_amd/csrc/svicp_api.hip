// Host side of libsvicp.so: the C ABI declared in include/svicp.h.
// Owns device memory, builds the search grid, launches the kernels of
// svicp_device.hpp and replays the iteration loop as one HIP graph.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <dlfcn.h>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enumerators only: the library is bound at run time (rccl_api)
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "svicp.h"
#include "svicp_device.hpp"
#include "svicp_p2p.hpp"

using namespace svicp;

namespace {

thread_local std::string g_create_error = "";

// Device memory handed out in slices from a few large allocations (a batch of hundreds of
// pairs would otherwise make thousands of hipMalloc calls).  Slices are never returned one
// by one: the arena frees everything when it is destroyed.
struct Arena {
    static constexpr size_t kChunk = (size_t)256 << 20;
    std::vector<void *> chunks;
    size_t used = 0, cap = 0;       // of the last chunk
    size_t total = 0;
    hipError_t alloc(size_t bytes, void **out)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        if (chunks.empty() || used + bytes > cap) {
            const size_t want = std::max(bytes, kChunk);
            void *p = nullptr;
            hipError_t e = hipMalloc(&p, want);
            if (e != hipSuccess) return e;
            chunks.push_back(p);
            used = 0; cap = want; total += want;
        }
        *out = static_cast<char *>(chunks.back()) + used;
        used += bytes;
        return hipSuccess;
    }
    void release() { for (void *p : chunks) (void)hipFree(p); chunks.clear(); used = cap = total = 0; }
};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    Arena *arena = nullptr;      // non-null: slices of the arena (a slice that is outgrown is abandoned)
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        size_t want = std::max<size_t>(bytes, 256);
        if (arena) {
            hipError_t e = arena->alloc(want, &p);
            cap = e == hipSuccess ? want : 0;
            return e;
        }
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p && !arena) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Buffers a context needs while it builds (and, for the k-NN facade, after): the caller-order
// copies of the target and its normals, the cell tables the exact search walks, sort keys,
// staging.  A context owns one; the members of an arena batch share their batch's, so on them
// only what the iteration loop needs outlives the set-up calls.
struct Scratch {
    DevBuf tgt_aos, nrm_aos, run2, end1, qrun, big_count;
    DevBuf tmp_aos, keys_a, keys_b, vals_a, vals_b, marks, ranks, block_of_rank, sort_tmp, bbox6;
    DevBuf knn_idx, knn_d2, unresolved, n_unres, q_soa, far_list;
    float *h_small = nullptr;      // pinned: small results on their way back (bounding box, counts)
    void release()
    {
        DevBuf *bufs[] = { &tgt_aos, &nrm_aos, &run2, &end1, &qrun, &big_count, &tmp_aos, &keys_a, &keys_b, &vals_a, &vals_b,
                           &marks, &ranks, &block_of_rank, &sort_tmp, &bbox6, &knn_idx, &knn_d2,
                           &unresolved, &n_unres, &q_soa, &far_list };
        for (DevBuf *b : bufs) b->release();
        if (h_small) (void)hipHostFree(h_small);
        h_small = nullptr;
    }
};

// an event that is destroyed on every path out of the scope that made it
struct ScopedEvent {
    hipEvent_t ev = nullptr;
    hipError_t create() { return hipEventCreate(&ev); }
    ~ScopedEvent() { if (ev) (void)hipEventDestroy(ev); }
    ScopedEvent() = default;
    ScopedEvent(const ScopedEvent &) = delete;
    ScopedEvent &operator=(const ScopedEvent &) = delete;
};

}  // namespace

// ---- the workspace of svicp_run_enqueue: one file-scope array per device ----
// Everything the iteration loop needs beyond the resident clouds and their per-point state:
// the tiles' sums, the running totals and their per-iteration changes, the iteration state and
// history, the tiles' "had work" flags.  Sized for kEnqMaxIters iterations of a source of
// kEnqMaxTiles tiles (1024 points each).
// from this many tiles on (four rounds of blocks on 256 compute units) the solve gets a launch of its own
constexpr int kManyTiles = 4096;     // tiles from which a source counts as large (see launch_iteration)
constexpr int kManyFromIteration = 4;
constexpr int kEnqMaxIters = 128;
constexpr int kEnqMaxTiles = 4194304 / kItTile;
constexpr size_t enq_align(size_t b) { return (b + 255) & ~(size_t)255; }
constexpr size_t kEnqOffState = 0;
constexpr size_t kEnqOffTotals = kEnqOffState + enq_align(sizeof(IcpState));
constexpr size_t kEnqOffDeltas = kEnqOffTotals + enq_align(sizeof(double) * kNumSums * (kEnqMaxIters + 1));
constexpr size_t kEnqOffXhist = kEnqOffDeltas + enq_align(sizeof(double) * kDeltaRow * (kEnqMaxIters + 1));
constexpr size_t kEnqOffIhist = kEnqOffXhist + enq_align(sizeof(double) * 6 * kEnqMaxIters);
constexpr size_t kEnqOffFlags = kEnqOffIhist + enq_align(sizeof(double) * 4 * kEnqMaxIters);
constexpr size_t kFlagBytes = sizeof(unsigned long long) * kFlagRows * kFlagWords + sizeof(float4) * kFlagCap;   // flag rows, then a look-ahead position per tile
constexpr size_t kEnqOffPartials = kEnqOffFlags + enq_align(kFlagBytes);
constexpr size_t kEnqBytes = kEnqOffPartials + enq_align(sizeof(double) * kNumSums * kEnqMaxTiles);
// The array is cut into kEnqSlots workspaces of kEnqBytes (about 1.2 MB each).  A context takes
// one the first time it calls svicp_run_enqueue and keeps it until it is destroyed, so contexts
// that enqueue on streams of their own at the same time never share a workspace; one context
// has one stream, and calls on one stream are ordered.
constexpr int kEnqSlots = 8;
static __device__ __align__(256) unsigned char scratch[kEnqSlots * kEnqBytes];   // about 9.7 MB

namespace {
// Device address of `scratch` on each device (the array's name is a host-side stand-in, not
// an address a kernel can use): asked for once per device, with that device current.
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxDevices] = {};
unsigned g_scratch_taken[kMaxDevices] = {};     // bit s: workspace s of that device is some context's

// A free workspace of `device` (its number, or -1 when every one is taken), and its return.
int take_scratch_slot(int device)
{
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (device < 0 || device >= kMaxDevices) return -1;
    for (int s = 0; s < kEnqSlots; ++s)
        if (!(g_scratch_taken[device] & (1u << s))) { g_scratch_taken[device] |= 1u << s; return s; }
    return -1;
}

void return_scratch_slot(int device, int slot)
{
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (device >= 0 && device < kMaxDevices && slot >= 0 && slot < kEnqSlots)
        g_scratch_taken[device] &= ~(1u << slot);
}

hipError_t scratch_on_current_device(int device, unsigned char **out)
{
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = static_cast<unsigned char *>(p);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
}  // namespace

struct svicp_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t own_stream = nullptr, stream = nullptr;
    svicp_options opt{};
    std::string err;

    // target
    int64_t m = 0;
    bool have_target = false;
    GridDesc g{};
    int nranked = 0;
    int quad_big_records = 0;      // probe records kept in full (they do not fit the packed form)
    DevBuf pts, nrm, dir, quad, quad_big, g_dev;
    Scratch own_scr, *scr = &own_scr;    // (an arena batch points its members at one shared Scratch)
    bool lean() const { return scr != &own_scr; }

    // source
    int64_t n = 0;
    bool have_source = false, sorted = false;
    int cur = 0;   // 0: the pristine copy is current, 1: the work copy
    DevBuf sx0, sy0, sz0, wx, wy, wz, perm;
    DevBuf st_tp, st_nv, st_lb;   // per source point: matched target point, normal, bound
    DevBuf tile_flags;            // [kFlagRows][kFlagWords] 64-bit words: which tiles had work in the last iterations
    DevBuf tile_box;              // sources of kManyTiles tiles or more: where each wave's points lie (PointState::box)
    bool use_tile_box = true;     // (svicp_debug_use_tile_boxes: off for an A/B against launches that read every tile)


    // iteration loop
    DevBuf partials, totals, deltas, state, xhist, ihist;
    // fixed-point exponents of the 30 sums (see update_fixed_point) and what they are made from
    DevBuf fx_exp;
    DevBuf rt_apart;               // 16 doubles: R, T handed from a solve launch to the iteration launch (large sources); 12..13: the two integers the ranks of a sharded run agree on
    int fx_host[kNumSums] = {};
    float tgt_maxabs = 0.f, nrm_maxabs = 1.f;
    int64_t total_points = 0;      // source points over ALL ranks of a sharded run (0: this context's own)
    int64_t explicit_total = 0;    // what svicp_shard_set_total gave for the present source (0: nothing)
    IcpState *h_state = nullptr;   // pinned and mapped: the kernel that ends a run writes it
    IcpState *h_state_dev = nullptr;
    bool h_state_pinned = false;   // h_state came from hipHostMalloc (a batch member's is plain memory)
    int hist_cap = 0, last_iters = 0;
    int nb = 0, chunk = 0;
    hipGraphExec_t gexec = nullptr;
    int g_iters = -1;

    // timing
    std::vector<hipEvent_t> events;
    double fused_ms = 0, solve_ms = 0;
    std::vector<double> fused_it, solve_it;
    int timed_launches = 0;

    int shard_iter = 0;
    // sharded runs inside the library: the communicator over the ranks of one registration
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    hipGraphExec_t gexec_shard = nullptr;
    int g_iters_shard = -1;
    bool shard_capture_failed = false;   // this runtime would not capture the collective: eager from then on
    unsigned char *enq_scratch = nullptr;   // this device's copy of `scratch`
    int enq_slot = -1;                      // the workspace inside it this context has taken (svicp_run_enqueue)

    // point-to-point variant
    DevBuf p2p_sums, rthist;
    hipGraphExec_t gexec_p2p = nullptr;
    int g_iters_p2p = -1;
    bool last_was_p2p = false;
};


// Many independent registrations advanced by one launch per iteration.
struct svicp_batch {
    // svicp_batch_create_from_arrays: the batch owns its pairs' contexts, one arena for all
    // their device memory, the scratch they share and one stream
    bool owns = false;
    Arena arena;
    Scratch shared;
    hipStream_t own_stream = nullptr;
    std::vector<IcpState> host_states;
    double setup_ms = 0.0;
    std::vector<svicp_ctx *> ctxs;
    svicp_ctx *lead = nullptr;          // device, stream and error text of the batch
    DevBuf pairs, block_map, states, rt_all;   // rt_all: R, T of every pair for the next launch
    std::vector<unsigned char> sent_pairs, sent_map;   // what the device tables hold
    IcpState *h_states = nullptr;       // pinned
    int total_blocks = 0;
    int iters_cap = 0;                  // iterations the pairs' buffers hold (see svicp_batch_run)
    bool tables_ready = false;
    hipGraphExec_t gexec = nullptr;
    int g_iters = -1;
    bool g_coherent = true;             // the option the captured graph was built with
    double last_ms = 0.0;
};

namespace {

int fail(svicp_ctx *c, int code, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) return fail(ctx, SVICP_ERR_HIP, #call, e_);         \
    } while (0)

#define LAUNCHCHK(what)                                                           \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) return fail(ctx, SVICP_ERR_HIP, what, e_);          \
    } while (0)

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

void drop_graph(svicp_ctx *ctx)
{
    if (ctx->gexec) { (void)hipGraphExecDestroy(ctx->gexec); ctx->gexec = nullptr; }
    ctx->g_iters = -1;
    if (ctx->gexec_p2p) { (void)hipGraphExecDestroy(ctx->gexec_p2p); ctx->gexec_p2p = nullptr; }
    ctx->g_iters_p2p = -1;
    if (ctx->gexec_shard) { (void)hipGraphExecDestroy(ctx->gexec_shard); ctx->gexec_shard = nullptr; }
    ctx->g_iters_shard = -1;
}

// A pinned page for results of a few bytes: a copy into pageable memory costs tens of
// microseconds more.
constexpr size_t kSmallBytes = 8192;
int ensure_small(svicp_ctx *ctx)
{
    if (!ctx->scr->h_small) HIPCHK(hipHostMalloc((void **)&ctx->scr->h_small, kSmallBytes, 0));
    return SVICP_OK;
}

// bounding box of an n x 3 device array; returns through mn/mx
int device_bbox(svicp_ctx *ctx, const float *aos, int64_t n, float mn[3], float mx[3])
{
    const int nblk = 256;
    HIPCHK(ctx->scr->bbox6.reserve(sizeof(float) * 6 * nblk));
    hipLaunchKernelGGL(bbox_kernel, dim3(nblk), dim3(kBlockThreads), 0, ctx->stream, aos,
                       (long long)n, ctx->scr->bbox6.as<float>());
    LAUNCHCHK("bbox_kernel");
    int rc = ensure_small(ctx);
    if (rc) return rc;
    float *h = ctx->scr->h_small;
    HIPCHK(hipMemcpyAsync(h, ctx->scr->bbox6.p, sizeof(float) * 6 * nblk, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int a = 0; a < 3; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; }
    for (int b = 0; b < nblk; ++b)
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], h[6 * b + a]);
            mx[a] = std::max(mx[a], h[6 * b + 3 + a]);
        }
    return SVICP_OK;
}

// Exponents of the fixed-point sums (svicp_device.hpp, fx_t): sum t is kept as
// round(value * 2^e_t) with e_t = 62 - ceil(log2(bound_t)), bound_t an upper bound of |sum t|:
// points x (largest rotation-column entry)^nrot x (largest normal component)^ntr x
// (largest residual)^nb, from the target's largest coordinate, the largest normal component and
// the threshold (a point contributes only while it lies within sqrt(threshold) of a target
// point).  Every rank of a sharded run must arrive at the same table: the point count is then
// the one agreed over the ranks (ctx->total_points).  Uploaded here, at set-up time; the
// run paths only read it.
int update_fixed_point(svicp_ctx *ctx)
{
    const double thr = ctx->opt.dist_threshold, root = std::sqrt(thr);
    const double smax = (double)ctx->tgt_maxabs + root + 1.0;
    const double nmax = std::isfinite(ctx->nrm_maxabs) && ctx->nrm_maxabs > 0.f ? (double)ctx->nrm_maxabs : 1.0;
    const double A = 2.0 * smax * nmax * 1.01, Nn = nmax * 1.01, B = 3.0 * root * nmax * 1.01;
    const double npts = (double)std::max<int64_t>(ctx->total_points > 0 ? ctx->total_points : ctx->n, 1);
    auto exponent = [&](int nrot, int ntr, int nb) {
        double bound = npts * std::pow(A, nrot) * std::pow(Nn, ntr) * std::pow(B, nb);
        if (!(bound >= 1.0)) bound = 1.0;
        const int e = 62 - (int)std::ceil(std::log2(bound));
        return std::max(-900, std::min(e, 62));
    };
    int t = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) {
            const int nrot = (r < 3) + (c < 3);
            ctx->fx_host[t++] = exponent(nrot, 2 - nrot, 0);
        }
    for (int r = 0; r < 6; ++r) ctx->fx_host[t++] = exponent(r < 3, r >= 3, 1);
    ctx->fx_host[27] = exponent(0, 0, 0);
    ctx->fx_host[28] = exponent(0, 0, 2);
    ctx->fx_host[29] = exponent(0, 0, 0);
    ctx->fx_host[30] = ctx->fx_host[31] = 0;
    HIPCHK(ctx->fx_exp.reserve(sizeof(int) * kNumSums));
    HIPCHK(ctx->rt_apart.reserve(sizeof(double) * 16));
    HIPCHK(hipMemcpyAsync(ctx->fx_exp.p, ctx->fx_host, sizeof(int) * kNumSums, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SVICP_OK;
}

int sort_pairs(svicp_ctx *ctx, int64_t count, int bits)
{
    HIPCHK(ctx->scr->keys_b.reserve(sizeof(unsigned long long) * count));
    HIPCHK(ctx->scr->vals_b.reserve(sizeof(int) * count));
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, ctx->scr->keys_a.as<unsigned long long>(),
                                     ctx->scr->keys_b.as<unsigned long long>(), ctx->scr->vals_a.as<int>(),
                                     ctx->scr->vals_b.as<int>(), (size_t)count, 0u, (unsigned)bits,
                                     ctx->stream));
    HIPCHK(ctx->scr->sort_tmp.reserve(bytes));
    HIPCHK(rocprim::radix_sort_pairs(ctx->scr->sort_tmp.p, bytes, ctx->scr->keys_a.as<unsigned long long>(),
                                     ctx->scr->keys_b.as<unsigned long long>(), ctx->scr->vals_a.as<int>(),
                                     ctx->scr->vals_b.as<int>(), (size_t)count, 0u, (unsigned)bits,
                                     ctx->stream));
    return SVICP_OK;
}

// exact k-NN of n device-resident AoS queries; results to device idx/d2 (n x k).
// k <= kKnnWaveK: a wave per query; the few queries it passes on (too many candidates at one
// distance) go on a list to the thread-per-query search, and what no ring settles to the
// exhaustive pass.  Larger k: the thread-per-query search for every query.
int run_knn(svicp_ctx *ctx, const float *q_aos, int64_t n, int k, int *idx, float *d2)
{
    if (n == 0) return SVICP_OK;
    HIPCHK(ctx->scr->unresolved.reserve(sizeof(int) * 2 * n));
    HIPCHK(ctx->scr->n_unres.reserve(sizeof(int) * 2));
    HIPCHK(hipMemsetAsync(ctx->scr->n_unres.p, 0, sizeof(int) * 2, ctx->stream));
    const int grid = std::min(cdiv(n, kBlockThreads), 65536);
    int *list = ctx->scr->unresolved.as<int>(), *nlist = ctx->scr->n_unres.as<int>();
    int *unres = list + n, *nun = nlist + 1;
    const bool by_wave = k <= kKnnWaveK;
    if (by_wave) {
        const int waves = kBlockThreads / 64;
        hipLaunchKernelGGL(knn_wave_kernel, dim3(std::min(cdiv(n, waves), 8192)), dim3(kBlockThreads), 0,
                           ctx->stream, ctx->g, q_aos, (int)n, k, idx, d2, list, nlist, unres, nun);
        LAUNCHCHK("knn_wave_kernel");
    }
#define KNN_LAUNCH(KM)                                                                           \
    do {                                                                                         \
        if (by_wave)                                                                             \
            hipLaunchKernelGGL((knn_kernel<KM, true>), dim3(std::min(grid, 1024)),               \
                               dim3(kBlockThreads), 0, ctx->stream, ctx->g, q_aos, (int)n, k,    \
                               idx, d2, list, nlist, unres, nun);                                \
        else                                                                                     \
            hipLaunchKernelGGL((knn_kernel<KM, false>), dim3(grid), dim3(kBlockThreads), 0,      \
                               ctx->stream, ctx->g, q_aos, (int)n, k, idx, d2, nullptr, nullptr, \
                               unres, nun);                                                      \
        LAUNCHCHK("knn_kernel");                                                                 \
        hipLaunchKernelGGL((knn_brute_block_kernel<KM>), dim3(2048), dim3(kBlockThreads), 0,     \
                           ctx->stream, ctx->g, q_aos, k, idx, d2, unres, nun);                  \
        LAUNCHCHK("knn_brute_block_kernel");                                                     \
    } while (0)
    if (k <= 8) KNN_LAUNCH(8);
    else if (k <= 32) KNN_LAUNCH(32);
    else KNN_LAUNCH(128);
#undef KNN_LAUNCH
    return SVICP_OK;
}

// 1-NN of n queries given as three arrays (perm: where query i's answer goes, or null) into idx / d2
int run_nn1_on(svicp_ctx *ctx, const float *x, const float *y, const float *z, const int *perm, int64_t n,
               bool exact, int *idx, float *d2)
{
    if (n == 0) return SVICP_OK;
    HIPCHK(ctx->scr->unresolved.reserve(sizeof(int) * n));
    HIPCHK(ctx->scr->n_unres.reserve(sizeof(int) * 2));
    const int grid = std::min(cdiv(n, kBlockThreads), ctx->num_cus * 8);
    if (!exact) {
        hipLaunchKernelGGL((nn1_kernel<false, 4>), dim3(grid), dim3(kBlockThreads), 0, ctx->stream,
                           ctx->g, x, y, z, perm, (int)n, ctx->opt.dist_threshold, idx, d2, (int *)nullptr,
                           (int *)nullptr, (int4 *)nullptr, (int *)nullptr);
        LAUNCHCHK("nn1_kernel");
    } else {
        // The levels of the shared-box pass: {lanes of a wave an entry stands for, largest box in
        // blocks}.  Level 0 takes whole waves of the search kernel; a wave whose box gets too large
        // hands its open lanes on to the next level in parts, down to single lanes (a box around
        // one query always works; what even that leaves goes to the exhaustive pass).
        struct FarLevel { int lanes, max_blocks; };
        static const std::vector<FarLevel> levels = [] {
            std::vector<FarLevel> v;
            const char *e = getenv("SVICP_FAR_LEVELS");          // development: "64:128,4:128,1:128"
            for (const char *p = e; p && *p;) {
                FarLevel L = { atoi(p), 128 };
                if (const char *c = strchr(p, ':')) L.max_blocks = atoi(c + 1);
                v.push_back(L);
                p = strchr(p, ',');
                if (p) ++p;
            }
            if (v.empty() || v[0].lanes != 64) v = { { 64, 128 }, { 4, 128 }, { 1, 128 } };
            return v;
        }();
        const int nlev = (int)levels.size();
        // counters: [0] queries left for the exhaustive pass, [l] entries of level l's list
        HIPCHK(ctx->scr->n_unres.reserve(sizeof(int) * 8));
        HIPCHK(hipMemsetAsync(ctx->scr->n_unres.p, 0, sizeof(int) * 8, ctx->stream));
        const int waves = cdiv(n, 64);
        std::vector<size_t> cap(nlev), off(nlev);
        size_t total = 0;
        for (int l = 0; l < nlev; ++l) {
            // (every open lane of a level can end up as an entry of the next: never more than that)
            cap[l] = l == 0 ? (size_t)waves : std::min<size_t>((size_t)cdiv(n, (int64_t)levels[l].lanes) + 64, (size_t)1 << 22);
            off[l] = total;
            total += cap[l];
        }
        HIPCHK(ctx->scr->far_list.reserve(sizeof(int4) * total));
        int *const n_unres = ctx->scr->n_unres.as<int>();
        int4 *const lists = ctx->scr->far_list.as<int4>();
        // one chunk of 256 queries per block: the far queries, which take long, come in
        // stretches, and the dispatcher spreads them best when the blocks are small
        const int grid_exact = (int)std::min<int64_t>(cdiv(n, kBlockThreads), 1 << 20);
        hipLaunchKernelGGL((nn1_kernel<true, 4>), dim3(grid_exact), dim3(kBlockThreads), 0, ctx->stream,
                           ctx->g, x, y, z, perm, (int)n, ctx->opt.dist_threshold, idx, d2,
                           ctx->scr->unresolved.as<int>(), n_unres, lists, (int *)nullptr);
        LAUNCHCHK("nn1_kernel(exact)");
        // one candidate box per wave, in LDS (a grid that fills the GPU; every wave takes listed
        // entries in turn)
        const int grid_far = (int)std::min<int64_t>(cdiv(waves, kFarThreads / 64), (int64_t)ctx->num_cus * 32);
        const dim3 fb(kFarThreads);
        int *const unres = ctx->scr->unresolved.as<int>();
        for (int l = 0; l < nlev; ++l) {
            const bool last = l + 1 == nlev;
            const int next_lanes = last ? 1 : levels[l + 1].lanes;
            hipLaunchKernelGGL(nn1_far_kernel, dim3(l == 0 ? grid_far : std::min(grid_far, 2048)), fb, 0, ctx->stream,
                               ctx->g, x, y, z, perm, (int)n, lists + off[l], l == 0 ? (const int *)nullptr : n_unres + l,
                               (int)cap[l], last ? (int4 *)nullptr : lists + off[l + 1], last ? (int *)nullptr : n_unres + l + 1,
                               last ? 0 : (int)cap[l + 1], last ? 1 : levels[l].lanes / next_lanes, next_lanes,
                               levels[l].max_blocks, idx, d2, unres, n_unres);
            LAUNCHCHK("nn1_far_kernel");
        }
        hipLaunchKernelGGL(nn1_brute_kernel, dim3(std::min<int64_t>(n, 2048)), dim3(kBlockThreads),
                           0, ctx->stream, ctx->g, x, y, z, perm, ctx->scr->unresolved.as<int>(),
                           n_unres, idx, d2);
        LAUNCHCHK("nn1_brute_kernel");
    }
    return SVICP_OK;
}

// 1-NN of the resident source into knn_idx / knn_d2 (caller order)
int run_nn1(svicp_ctx *ctx, bool exact)
{
    const int64_t n = ctx->n;
    HIPCHK(ctx->scr->knn_idx.reserve(sizeof(int) * std::max<int64_t>(n, 1)));
    HIPCHK(ctx->scr->knn_d2.reserve(sizeof(float) * std::max<int64_t>(n, 1)));
    const float *x = ctx->cur ? ctx->wx.as<float>() : ctx->sx0.as<float>();
    const float *y = ctx->cur ? ctx->wy.as<float>() : ctx->sy0.as<float>();
    const float *z = ctx->cur ? ctx->wz.as<float>() : ctx->sz0.as<float>();
    const int *perm = ctx->sorted ? ctx->perm.as<int>() : nullptr;
    return run_nn1_on(ctx, x, y, z, perm, n, exact, ctx->scr->knn_idx.as<int>(), ctx->scr->knn_d2.as<float>());
}

int ensure_iteration_buffers(svicp_ctx *ctx, int iters)
{
    const int64_t n = ctx->n;
    const int chunk = kItTile;
    const int nb = std::max(1, cdiv(std::max<int64_t>(n, 1), chunk));
    if (nb != ctx->nb || chunk != ctx->chunk) drop_graph(ctx);
    ctx->nb = nb; ctx->chunk = chunk;
    HIPCHK(ctx->partials.reserve(sizeof(double) * kNumSums * nb));
    if (!ctx->tile_flags.p) {
        // (a pair of an arena batch runs in the batch kernel, which gives every block its tile
        // and looks ahead for nothing: the rows alone)
        const size_t rows = sizeof(unsigned long long) * kFlagRows * kFlagWords;
        HIPCHK(ctx->tile_flags.reserve(ctx->lean() ? rows : kFlagBytes));
        HIPCHK(hipMemsetAsync(ctx->tile_flags.p, 0, rows, ctx->stream));
        // (the look-ahead positions behind the rows start as NaN: nothing to look ahead for)
        if (!ctx->lean())
            HIPCHK(hipMemsetAsync(ctx->tile_flags.as<char>() + rows, 0xff, sizeof(float4) * kFlagCap, ctx->stream));
    }
    HIPCHK(ctx->state.reserve(sizeof(IcpState)));
    if (iters > ctx->hist_cap) {
        drop_graph(ctx);
        int cap = std::max(iters, 64);
        HIPCHK(ctx->xhist.reserve(sizeof(double) * 6 * cap));
        HIPCHK(ctx->ihist.reserve(sizeof(double) * 4 * cap));
        HIPCHK(ctx->totals.reserve(sizeof(double) * kNumSums * (cap + 1)));
        HIPCHK(ctx->deltas.reserve(sizeof(double) * kDeltaRow * (cap + 1)));
        // totals row 0 (the sums "before iteration 0") stays zero for good; delta rows are
        // cleared by the kernels themselves from here on
        HIPCHK(hipMemsetAsync(ctx->totals.p, 0, sizeof(double) * kNumSums * (cap + 1), ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->deltas.p, 0, sizeof(double) * kDeltaRow * (cap + 1), ctx->stream));
        ctx->hist_cap = cap;
    }
    if (!ctx->h_state) {
        HIPCHK(hipHostMalloc((void **)&ctx->h_state, sizeof(IcpState), hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void **)&ctx->h_state_dev, ctx->h_state, 0));
        ctx->h_state_pinned = true;
    }
    return SVICP_OK;
}

// The small buffers of one iteration loop: the context's own, or (svicp_run_enqueue) places in
// the device's file-scope workspace.
struct LoopBufs {
    double *partials, *xhist, *ihist;
    fx_t *totals, *deltas;
    IcpState *state;
    unsigned long long *flags;    // [kFlagRows][kFlagWords] or null
};

LoopBufs own_bufs(svicp_ctx *ctx)
{
    LoopBufs b;
    b.partials = ctx->partials.as<double>();
    b.totals = ctx->totals.as<fx_t>();
    b.deltas = ctx->deltas.as<fx_t>();
    b.xhist = ctx->xhist.as<double>();
    b.ihist = ctx->ihist.as<double>();
    b.state = ctx->state.as<IcpState>();
    // (a pair lent by an arena batch keeps the rows only, without the look-ahead positions
    // behind them: it runs with the fixed block order)
    b.flags = ctx->nb <= kFlagCap && !ctx->lean() ? ctx->tile_flags.as<unsigned long long>() : nullptr;
    return b;
}

// launches iteration k's kernel; the tile sums end up in `b.partials`
void launch_iteration(svicp_ctx *ctx, int k, const LoopBufs &b)
{
    const float thr = ctx->opt.dist_threshold;
    CohParams cp;
    cp.thr = thr;
    cp.root_hi = std::sqrt(thr) * 1.00001f;
    cp.cov = std::sqrt(thr) * 1.001f;   // inside the probe radius rr = sqrt(thr) * 1.002 + slack
    PointState ps;
    ps.tp = ctx->st_tp.as<Float3>();
    ps.nv = ctx->st_nv.as<Float3>();
    ps.lb = ctx->st_lb.as<float>();
    ps.active = b.flags;
    ps.box = ctx->use_tile_box ? ctx->tile_box.as<float4>() : nullptr;     // (null below kManyTiles tiles)
    float *wx = ctx->wx.as<float>(), *wy = ctx->wy.as<float>(), *wz = ctx->wz.as<float>();
    const float *x0 = ctx->sx0.as<float>(), *y0 = ctx->sy0.as<float>(), *z0 = ctx->sz0.as<float>();
    IcpState *state = b.state;
    fx_t *tot = b.totals, *del = b.deltas;
    const int *fx = ctx->fx_exp.as<int>();
    double *xh = b.xhist, *ih = b.ihist;
    double *partials = b.partials;
    const dim3 grid(ctx->nb), block(kItThreads);
    const int n = (int)ctx->n;
    // Many thousands of tiles.  The launches that search most of their points (the first
    // kManyFromIteration of a run: iteration 0 has nothing to go by, iteration 1 follows the
    // first and largest step, and a registration from an unregistered pose needs a few more) run
    // a block per tile, with the previous iteration's system solved once, by a launch of its own,
    // instead of by every block of every round.  The later ones run one round of blocks, each
    // working through several tiles (icp_iterate_many_kernel): most tiles have nothing to do
    // then, and every round of blocks would cost the time a block needs to find that out.  Both
    // kernels are right for any launch; the choice is a matter of time only (searching launches
    // of the 16.8 M-point pair: 520/590/290/106 us against 640/725/365/152; settled ones 47
    // against 22).
    static const int many_from = getenv("SVICP_MANY_TILES") ? atoi(getenv("SVICP_MANY_TILES")) : kManyTiles;
    static const int many_from_iter = getenv("SVICP_MANY_FROM_ITER") ? atoi(getenv("SVICP_MANY_FROM_ITER")) : kManyFromIteration;
    const bool large = ctx->nb >= many_from;
    const bool many = large && k >= many_from_iter;
    const bool apart = large && !many && k >= 1;
    unsigned long long *many_flags =
        ctx->tile_box.p ? reinterpret_cast<unsigned long long *>(ctx->tile_box.as<float4>() + (size_t)kBoxRecord * ctx->nb) : nullptr;
    static const int per_cu = getenv("SVICP_MANY_PER_CU") ? std::max(1, atoi(getenv("SVICP_MANY_PER_CU"))) : SVICP_MANY_OCC;
    const int many_blocks = std::max(std::min(ctx->nb, ctx->num_cus * per_cu), cdiv(ctx->nb, kManyRounds));
    double *rt = ctx->rt_apart.as<double>();
    if (apart)
        hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(64), 0, ctx->stream, tot, del, fx, state, xh, ih, k - 1,
                           (IcpState *)nullptr, (double *)nullptr, rt);
#define SVICP_ITERATE(APPLY, CERTIFY, INPLACE, ix, iy, iz)                                        \
    do {                                                                                          \
        if (many)                                                                                 \
            hipLaunchKernelGGL((icp_iterate_many_kernel<APPLY, CERTIFY, 4, INPLACE>), dim3(many_blocks), block, 0, \
                               ctx->stream, tot, del, fx, k, n, ctx->nb, many_flags, ps.tp, ps.nv, ps.lb, ps.box, ix, iy, iz, \
                               wx, wy, wz, state, xh, ih, cp, partials, ctx->g_dev.as<GridDesc>()); \
        else if (large)                                                                           \
            hipLaunchKernelGGL((icp_iterate_kernel<APPLY, CERTIFY, 4, INPLACE, true>), grid, block, 0, ctx->stream, \
                               tot, del, fx, k, n, many_flags, ps.tp, ps.nv, ps.lb, ix, iy, iz, wx, wy, wz, state, \
                               xh, ih, cp, partials, ctx->g_dev.as<GridDesc>(), rt, ps.box);      \
        else                                                                                      \
            hipLaunchKernelGGL((icp_iterate_kernel<APPLY, CERTIFY, 4, INPLACE, false>), grid, block, 0, ctx->stream, \
                               tot, del, fx, k, n, ps.active, ps.tp, ps.nv, ps.lb, ix, iy, iz, wx, wy, wz, state, \
                               xh, ih, cp, partials, ctx->g_dev.as<GridDesc>(), (const double *)nullptr, (float4 *)nullptr); \
    } while (0)
    // iteration 0 reads the source as set and moves nothing, iteration 1 reads it and writes
    // the moving copy, later ones update the moving copy where it is
    if (k == 0) SVICP_ITERATE(false, false, false, x0, y0, z0);
    else if (k == 1 && ctx->opt.coherent) SVICP_ITERATE(true, true, false, x0, y0, z0);
    else if (k == 1) SVICP_ITERATE(true, false, false, x0, y0, z0);
    else if (ctx->opt.coherent) SVICP_ITERATE(true, true, true, wx, wy, wz);
    else SVICP_ITERATE(true, false, true, wx, wy, wz);
#undef SVICP_ITERATE
}

// the whole loop on ctx->stream: `iters` iteration kernels and the solve that ends the run.
// host_copy (mapped host memory) and dev_out (device, kRunOutDoubles doubles) may each be null.
int enqueue_loop(svicp_ctx *ctx, int iters, bool timed, const LoopBufs &b, IcpState *host_copy,
                 double *dev_out)
{
    // no reset and no clearing here: the first solve of a run starts the state afresh,
    // totals row 0 is always zero, delta row k + 1 is cleared by iteration k's kernel and
    // row 0 by the solve kernel that ended the previous run
    for (int k = 0; k < iters; ++k) {
        if (timed) HIPCHK(hipEventRecord(ctx->events[4 * k + 0], ctx->stream));
        launch_iteration(ctx, k, b);
        LAUNCHCHK("icp_iterate_kernel");
        if (timed) HIPCHK(hipEventRecord(ctx->events[4 * k + 1], ctx->stream));
        if (timed) HIPCHK(hipEventRecord(ctx->events[4 * k + 2], ctx->stream));
        if (timed) HIPCHK(hipEventRecord(ctx->events[4 * k + 3], ctx->stream));
    }
    // the last iteration's system is solved here; the others were solved at the
    // start of the following iteration's kernel
    hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(64), 0, ctx->stream, b.totals, b.deltas,
                       ctx->fx_exp.as<int>(), b.state, b.xhist, b.ihist, iters - 1, host_copy, dev_out);
    LAUNCHCHK("solve_kernel");
    return SVICP_OK;
}

void fill_info(const svicp_ctx *ctx, int iters, svicp_info *info)
{
    if (!info) return;
    const IcpState *s = ctx->h_state;
    info->iterations = iters;
    info->status = s->status;
    info->inliers = s->S[27];
    info->weighted_sq_residual = s->S[28];
    info->weight_sum = s->S[29];
    info->step_norm = s->S[30];
}

// RCCL, bound at run time: a process that already holds an RCCL (PyTorch brings its own) must
// use that one, and a caller that never shards needs none.
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *loaded[] = { "librccl.so.1", "librccl.so" };
        for (const char *name : loaded)
            if (!api.handle) api.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        const char *fresh[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char *name : fresh)
            if (!api.handle) api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!api.handle) { api.error = "librccl.so could not be loaded"; return; }
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.handle, "ncclAllReduce"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy || !api.GetErrorString) {
            api.error = "librccl.so lacks an entry point this library needs";
            api.handle = nullptr;
        }
    });
    return &api;
}

int fail_rccl(svicp_ctx *ctx, const char *what, ncclResult_t r)
{
    RcclApi *api = rccl_api();
    std::string msg = std::string(what) + ": " + (api->GetErrorString ? api->GetErrorString(r) : "RCCL error");
    return fail(ctx, SVICP_ERR_COMM, msg.c_str());
}

// The sharded loop on ctx->stream: iteration k's kernel leaves this rank's change of the sums
// in delta row k (kDeltaCopies copies of 32 doubles); one all-reduce sums that row over the
// ranks where it lies; iteration k + 1's kernel (or the closing solve) adds the copies to the
// totals and solves, on every rank from the same numbers.
int enqueue_shard_loop(svicp_ctx *ctx, int iters)
{
    RcclApi *api = rccl_api();
    const LoopBufs b = own_bufs(ctx);
    for (int k = 0; k < iters; ++k) {
        launch_iteration(ctx, k, b);
        LAUNCHCHK("icp_iterate_kernel");
        if (ctx->comm) {
            // 64-bit integers: the sum over the ranks is exact, whatever order RCCL adds them in
            fx_t *row = b.deltas + (size_t)k * kDeltaRow;
            ncclResult_t r = api->AllReduce(row, row, kDeltaRow, ncclInt64, ncclSum, ctx->comm, ctx->stream);
            if (r != ncclSuccess) return fail_rccl(ctx, "ncclAllReduce", r);
        }
    }
    hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(64), 0, ctx->stream, b.totals, b.deltas,
                       ctx->fx_exp.as<int>(), b.state, b.xhist, b.ihist, iters - 1, ctx->h_state_dev,
                       (double *)nullptr);
    LAUNCHCHK("solve_kernel");
    return SVICP_OK;
}

}  // namespace

extern "C" {

const char *svicp_last_error(const svicp_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int svicp_default_options(svicp_options *opt)
{
    if (!opt) return SVICP_ERR_ARG;
    opt->dist_threshold = 5.0f;
    opt->normals_k = 20;
    opt->use_graph = 1;
    opt->time_kernels = 0;
    opt->sort_source = 1;
    opt->coherent = 1;
    opt->centred_normals = 0;
    return SVICP_OK;
}

int svicp_create(int device_id, svicp_ctx **out)
{
    svicp_ctx *ctx = nullptr;
    if (!out) return fail(nullptr, SVICP_ERR_ARG, "svicp_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, SVICP_ERR_NO_DEVICE, "svicp_create: no HIP device is visible");
    if (device_id < 0 || device_id >= count)
        return fail(nullptr, SVICP_ERR_ARG, "svicp_create: device_id out of range");
    e = hipSetDevice(device_id);
    if (e != hipSuccess) return fail(nullptr, SVICP_ERR_HIP, "hipSetDevice", e);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return fail(nullptr, SVICP_ERR_HIP, "hipGetDeviceProperties", e);
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::string msg = std::string("svicp_create: device is ") + prop.gcnArchName +
                          ", the library holds gfx950 code only";
        return fail(nullptr, SVICP_ERR_NO_DEVICE, msg.c_str());
    }
    ctx = new svicp_ctx();
    ctx->device = device_id;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    svicp_default_options(&ctx->opt);
    e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail(nullptr, SVICP_ERR_HIP, "hipStreamCreate", e); }
    ctx->stream = ctx->own_stream;
    e = scratch_on_current_device(device_id, &ctx->enq_scratch);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return fail(nullptr, SVICP_ERR_HIP, "hipGetSymbolAddress(scratch)", e);
    }
    *out = ctx;
    return SVICP_OK;
}

int svicp_destroy(svicp_ctx *ctx)
{
    if (!ctx) return SVICP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    drop_graph(ctx);
    if (ctx->comm) { (void)rccl_api()->CommDestroy(ctx->comm); ctx->comm = nullptr; }
    return_scratch_slot(ctx->device, ctx->enq_slot);
    for (hipEvent_t ev : ctx->events) (void)hipEventDestroy(ev);
    DevBuf *bufs[] = { &ctx->pts, &ctx->nrm, &ctx->dir, &ctx->quad, &ctx->quad_big, &ctx->g_dev, &ctx->sx0, &ctx->sy0, &ctx->sz0, &ctx->wx,
                       &ctx->wy, &ctx->wz, &ctx->st_tp, &ctx->st_nv, &ctx->st_lb, &ctx->tile_flags, &ctx->tile_box, &ctx->perm,
                       &ctx->partials, &ctx->totals, &ctx->deltas, &ctx->state, &ctx->xhist, &ctx->ihist,
                       &ctx->p2p_sums, &ctx->rthist, &ctx->fx_exp, &ctx->rt_apart };
    for (DevBuf *b : bufs) b->release();          // (slices of an arena are the arena's to free)
    ctx->own_scr.release();
    if (ctx->h_state && ctx->h_state_pinned) (void)hipHostFree(ctx->h_state);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return SVICP_OK;
}

int svicp_set_options(svicp_ctx *ctx, const svicp_options *opt)
{
    if (!ctx || !opt) return fail(ctx, SVICP_ERR_ARG, "svicp_set_options: NULL argument");
    if (!(opt->dist_threshold > 0.f) || !std::isfinite(opt->dist_threshold))
        return fail(ctx, SVICP_ERR_ARG, "svicp_set_options: dist_threshold must be positive");
    if (opt->normals_k < 3 || opt->normals_k > 128)
        return fail(ctx, SVICP_ERR_ARG, "svicp_set_options: normals_k must be in [3, 128]");
    if (ctx->have_target && opt->dist_threshold != ctx->opt.dist_threshold)
        return fail(ctx, SVICP_ERR_ARG,
                    "svicp_set_options: dist_threshold sizes the grid; set it before the target");
    ctx->opt = *opt;
    drop_graph(ctx);
    return SVICP_OK;
}

int svicp_get_options(const svicp_ctx *ctx, svicp_options *opt)
{
    if (!ctx || !opt) return SVICP_ERR_ARG;
    *opt = ctx->opt;
    return SVICP_OK;
}

int svicp_grid_info(const svicp_ctx *ctx, int64_t *ranked_blocks, int64_t *record_bytes,
                    int64_t *records_in_full)
{
    if (!ctx || !ctx->have_target) return SVICP_ERR_ARG;
    if (ranked_blocks) *ranked_blocks = ctx->nranked;
    if (record_bytes)
        *record_bytes = (int64_t)sizeof(uint4) * 64 * kFineX * std::max(ctx->nranked, 1) +
                        (int64_t)sizeof(uint4) * 2 * ctx->quad_big_records;
    if (records_in_full) *records_in_full = ctx->quad_big_records;
    return SVICP_OK;
}

int svicp_sizes(const svicp_ctx *ctx, int64_t *m, int64_t *n)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (m) *m = ctx->have_target ? ctx->m : 0;
    if (n) *n = ctx->have_source ? ctx->n : 0;
    return SVICP_OK;
}

int svicp_set_stream(svicp_ctx *ctx, void *hip_stream)
{
    if (!ctx) return SVICP_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;   // NULL is the default stream
    drop_graph(ctx);
    return SVICP_OK;
}

int svicp_synchronize(svicp_ctx *ctx)
{
    if (!ctx) return SVICP_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SVICP_OK;
}

int svicp_use_own_stream(svicp_ctx *ctx)
{
    if (!ctx) return SVICP_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->stream = ctx->own_stream;
    drop_graph(ctx);
    return SVICP_OK;
}

int svicp_set_target(svicp_ctx *ctx, const float *xyz_aos, int64_t m, const float *normals_aos)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (m < 0 || (m > 0 && !xyz_aos))
        return fail(ctx, SVICP_ERR_ARG, "svicp_set_target: bad cloud pointer or size");
    if (m > (int64_t)0x7fffff00)
        return fail(ctx, SVICP_ERR_ARG, "svicp_set_target: more than 2^31 points");
    HIPCHK(hipSetDevice(ctx->device));
    drop_graph(ctx);
    ctx->have_target = false;
    ctx->m = m;
    const int64_t mm = std::max<int64_t>(m, 1);

    HIPCHK(ctx->scr->tgt_aos.reserve(sizeof(float) * 3 * mm));
    HIPCHK(ctx->scr->nrm_aos.reserve(sizeof(float) * 3 * mm));
    HIPCHK(ctx->pts.reserve(sizeof(float4) * mm));
    HIPCHK(ctx->nrm.reserve(sizeof(float) * 3 * mm));
    if (m > 0)
        HIPCHK(hipMemcpyAsync(ctx->scr->tgt_aos.p, xyz_aos, sizeof(float) * 3 * m, hipMemcpyHostToDevice,
                              ctx->stream));

    // ---- grid geometry ----
    float mn[3] = { 0, 0, 0 }, mx[3] = { 0, 0, 0 };
    if (m > 0) {
        int rc = device_bbox(ctx, ctx->scr->tgt_aos.as<float>(), m, mn, mx);
        if (rc) return rc;
        for (int a = 0; a < 3; ++a)
            if (!std::isfinite(mn[a]) || !std::isfinite(mx[a]))
                return fail(ctx, SVICP_ERR_ARG, "svicp_set_target: non-finite coordinates");
    }
    float maxabs = 0.f;
    for (int a = 0; a < 3; ++a) maxabs = std::max(maxabs, std::max(std::fabs(mn[a]), std::fabs(mx[a])));
    ctx->tgt_maxabs = maxabs;
    const float r = std::sqrt(ctx->opt.dist_threshold);
    const float eps_abs = maxabs * (1.0f / 1048576.0f);
    GridDesc g{};
    g.rr = r * 1.002f + eps_abs;
    float h = 2.0f * g.rr * 1.02f + 4.0f * eps_abs;
    long long nbt = 0;
    for (int attempt = 0; attempt < 64; ++attempt) {
        long long nn[3];
        bool fits = true;
        for (int a = 0; a < 3; ++a) {
            float o = mn[a] - 1.5f * h;
            nn[a] = (long long)std::floor((double)(mx[a] - o) / (double)h) + 3;
            if (nn[a] > 32768) fits = false;
        }
        nbt = ((nn[0] + 3) / 4) * ((nn[1] + 3) / 4) * ((nn[2] + 3) / 4);
        if (nbt > (1ll << 28)) fits = false;
        if (fits) {
            g.ox = mn[0] - 1.5f * h; g.oy = mn[1] - 1.5f * h; g.oz = mn[2] - 1.5f * h;
            g.nx = (int)nn[0]; g.ny = (int)nn[1]; g.nz = (int)nn[2];
            break;
        }
        h *= 1.5f;   // coarser cells stay exact, they only hold more candidates
        if (attempt == 63) return fail(ctx, SVICP_ERR_GRID, "svicp_set_target: extent too large for the grid");
    }
    g.h = h; g.inv_h = 1.0f / h;
    g.nbx = (g.nx + 3) / 4; g.nby = (g.ny + 3) / 4; g.nbz = (g.nz + 3) / 4;
    g.fnx1 = (float)(g.nx - 1); g.fny1 = (float)(g.ny - 1); g.fnz1 = (float)(g.nz - 1);
    g.cov_slack = 8.0f * (maxabs + h * (float)std::max(g.nx, std::max(g.ny, g.nz))) * (1.0f / 8388608.0f);
    g.m = (int)m;
    nbt = (long long)g.nbx * g.nby * g.nbz;

    // ---- sort by cell key ----
    HIPCHK(ctx->scr->keys_a.reserve(sizeof(unsigned long long) * mm));
    HIPCHK(ctx->scr->vals_a.reserve(sizeof(int) * mm));
    HIPCHK(ctx->scr->marks.reserve(sizeof(int) * nbt));
    HIPCHK(ctx->scr->ranks.reserve(sizeof(int) * nbt));
    HIPCHK(ctx->dir.reserve(sizeof(int) * nbt));
    HIPCHK(ctx->scr->block_of_rank.reserve(sizeof(int) * std::min<long long>(nbt, 8 * mm)));   // a point marks up to 8 blocks
    HIPCHK(hipMemsetAsync(ctx->scr->marks.p, 0, sizeof(int) * nbt, ctx->stream));
    int nranked = 0;
    if (m > 0) {
        hipLaunchKernelGGL(cell_key_kernel, dim3(cdiv(m, 256)), dim3(256), 0, ctx->stream, g,
                           ctx->scr->tgt_aos.as<float>(), (int)m, ctx->scr->keys_a.as<unsigned long long>(),
                           ctx->scr->vals_a.as<int>());
        LAUNCHCHK("cell_key_kernel");
        unsigned long long ncell = (unsigned long long)g.nx * g.ny * g.nz * kFineX;     // (the keys carry the quarter cell)
        int bits = 1;
        while (bits < 64 && (ncell >> bits)) ++bits;
        int rc = sort_pairs(ctx, m, bits);
        if (rc) return rc;
        hipLaunchKernelGGL(mark_blocks_kernel, dim3(cdiv(m, 256)), dim3(256), 0, ctx->stream, g,
                           ctx->scr->keys_b.as<unsigned long long>(), (int)m, ctx->scr->marks.as<int>());
        LAUNCHCHK("mark_blocks_kernel");
    }
    {
        size_t bytes = 0;
        HIPCHK(rocprim::exclusive_scan(nullptr, bytes, ctx->scr->marks.as<int>(), ctx->scr->ranks.as<int>(), 0,
                                       (size_t)nbt, rocprim::plus<int>(), ctx->stream));
        HIPCHK(ctx->scr->sort_tmp.reserve(bytes));
        HIPCHK(rocprim::exclusive_scan(ctx->scr->sort_tmp.p, bytes, ctx->scr->marks.as<int>(),
                                       ctx->scr->ranks.as<int>(), 0, (size_t)nbt, rocprim::plus<int>(),
                                       ctx->stream));
        int rc = ensure_small(ctx);
        if (rc) return rc;
        int *last = reinterpret_cast<int *>(ctx->scr->h_small);     // {last rank, last mark}
        HIPCHK(hipMemcpyAsync(&last[0], ctx->scr->ranks.as<int>() + (nbt - 1), sizeof(int),
                              hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(&last[1], ctx->scr->marks.as<int>() + (nbt - 1), sizeof(int),
                              hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        nranked = last[0] + last[1];
    }
    ctx->nranked = nranked;
    hipLaunchKernelGGL(finish_dir_kernel, dim3(cdiv(nbt, 256)), dim3(256), 0, ctx->stream,
                       ctx->scr->marks.as<int>(), ctx->scr->ranks.as<int>(), (int)nbt, ctx->dir.as<int>(),
                       ctx->scr->block_of_rank.as<int>());
    LAUNCHCHK("finish_dir_kernel");
    HIPCHK(ctx->scr->run2.reserve(sizeof(uint2) * 64 * std::max(nranked, 1)));
    HIPCHK(ctx->scr->end1.reserve(sizeof(unsigned) * 64 * std::max(nranked, 1)));
    HIPCHK(ctx->quad.reserve(sizeof(uint4) * 64 * kFineX * std::max(nranked, 1)));
    HIPCHK(ctx->scr->big_count.reserve(sizeof(int)));
    HIPCHK(ctx->scr->qrun.reserve(sizeof(uint2) * 64 * kFineX * std::max(nranked, 1)));
    if (nranked > 0) {
        hipLaunchKernelGGL(fill_cells_kernel, dim3(cdiv((long long)nranked * 64, 256)), dim3(256), 0,
                           ctx->stream, g, ctx->scr->keys_b.as<unsigned long long>(), (int)m,
                           ctx->scr->block_of_rank.as<int>(), nranked, ctx->scr->run2.as<uint2>(),
                           ctx->scr->end1.as<unsigned>());
        LAUNCHCHK("fill_cells_kernel");
        g.dir = ctx->dir.as<int>();
        hipLaunchKernelGGL(fill_quarter_runs_kernel, dim3(cdiv((long long)nranked * 64 * kFineX, 256)), dim3(256), 0,
                           ctx->stream, g, ctx->scr->keys_b.as<unsigned long long>(), (int)m,
                           ctx->scr->block_of_rank.as<int>(), nranked, ctx->scr->run2.as<uint2>(),
                           ctx->scr->qrun.as<uint2>());
        LAUNCHCHK("fill_quarter_runs_kernel");
        // the packed probe records; the few that do not fit the packing go to quad_big, whose
        // size is known only once they have been counted: a second pass when the first found
        // more than the buffer holds (none at all for a cloud without piles of duplicates or
        // rows of more than 65535 points)
        for (int pass = 0; pass < 2; ++pass) {
            const int capacity = (int)std::min<size_t>(ctx->quad_big.cap / (2 * sizeof(uint4)), 0x7fffffff);
            HIPCHK(hipMemsetAsync(ctx->scr->big_count.p, 0, sizeof(int), ctx->stream));
            hipLaunchKernelGGL(fill_quads_kernel, dim3(cdiv((long long)nranked * 64 * kFineX, 256)), dim3(256), 0,
                               ctx->stream, g, ctx->scr->block_of_rank.as<int>(), nranked,
                               ctx->scr->qrun.as<uint2>(), ctx->quad.as<uint4>(), ctx->quad_big.as<uint4>(),
                               capacity, ctx->scr->big_count.as<int>());
            LAUNCHCHK("fill_quads_kernel");
            int *count = reinterpret_cast<int *>(ctx->scr->h_small);
            HIPCHK(hipMemcpyAsync(count, ctx->scr->big_count.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            ctx->quad_big_records = *count;
            if (*count <= capacity) break;
            if (pass == 1) return fail(ctx, SVICP_ERR_HIP, "svicp_set_target: the probe records changed between two passes");
            HIPCHK(ctx->quad_big.reserve(sizeof(uint4) * 2 * (size_t)*count));
        }
    } else {
        // no ranked block: probes read record 0 and drop it, so it must be a plain record
        HIPCHK(hipMemsetAsync(ctx->quad.p, 0, sizeof(uint4), ctx->stream));
    }
    g.dir = ctx->dir.as<int>();
    g.run2 = ctx->scr->run2.as<uint2>();
    g.end1 = ctx->scr->end1.as<unsigned>();
    g.quad = ctx->quad.as<uint4>();
    g.quad_big = ctx->quad_big.as<uint4>();
    g.pts = ctx->pts.as<float4>();
    g.nrm = ctx->nrm.as<float>();
    ctx->g = g;
    // the iteration kernel reads the description from device memory
    HIPCHK(ctx->g_dev.reserve(sizeof(GridDesc)));
    HIPCHK(hipMemcpyAsync(ctx->g_dev.p, &ctx->g, sizeof(GridDesc), hipMemcpyHostToDevice, ctx->stream));

    if (m > 0) {
        hipLaunchKernelGGL(gather_target_kernel, dim3(cdiv(m, 256)), dim3(256), 0, ctx->stream,
                           ctx->scr->tgt_aos.as<float>(), (const float *)nullptr, ctx->scr->vals_b.as<int>(),
                           (int)m, ctx->pts.as<float4>(), ctx->nrm.as<float>());
        LAUNCHCHK("gather_target_kernel");
        // ---- normals ----
        if (normals_aos) {
            HIPCHK(hipMemcpyAsync(ctx->scr->nrm_aos.p, normals_aos, sizeof(float) * 3 * m,
                                  hipMemcpyHostToDevice, ctx->stream));
        } else {
            const int k = ctx->opt.normals_k;
            HIPCHK(ctx->scr->knn_idx.reserve(sizeof(int) * m * k));
            HIPCHK(ctx->scr->knn_d2.reserve(sizeof(float) * m * k));
            int rc = run_knn(ctx, ctx->scr->tgt_aos.as<float>(), m, k, ctx->scr->knn_idx.as<int>(),
                             ctx->scr->knn_d2.as<float>());
            if (rc) return rc;
            hipLaunchKernelGGL(normals_kernel, dim3(cdiv(m, kBlockThreads)), dim3(kBlockThreads), 0,
                               ctx->stream, ctx->scr->tgt_aos.as<float>(), ctx->scr->knn_idx.as<int>(), (int)m,
                               k, ctx->opt.centred_normals != 0, ctx->scr->nrm_aos.as<float>());
            LAUNCHCHK("normals_kernel");
        }
        hipLaunchKernelGGL(gather_target_kernel, dim3(cdiv(m, 256)), dim3(256), 0, ctx->stream,
                           ctx->scr->tgt_aos.as<float>(), ctx->scr->nrm_aos.as<float>(), ctx->scr->vals_b.as<int>(),
                           (int)m, ctx->pts.as<float4>(), ctx->nrm.as<float>());
        LAUNCHCHK("gather_target_kernel(normals)");
        // the largest normal component bounds the sums (estimated normals are unit vectors;
        // given ones are taken as they come)
        ctx->nrm_maxabs = 1.f;
        if (normals_aos) {
            float nmn[3], nmx[3];
            int rc = device_bbox(ctx, ctx->scr->nrm_aos.as<float>(), m, nmn, nmx);
            if (rc) return rc;
            float nm = 0.f;
            for (int a = 0; a < 3; ++a) nm = std::max(nm, std::max(std::fabs(nmn[a]), std::fabs(nmx[a])));
            if (!std::isfinite(nm)) return fail(ctx, SVICP_ERR_ARG, "svicp_set_target: non-finite normals");
            ctx->nrm_maxabs = std::max(nm, 1e-30f);
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->have_target = true;
    return update_fixed_point(ctx);
}

int svicp_get_normals(svicp_ctx *ctx, float *normals_aos_out)
{
    if (!ctx || !normals_aos_out) return fail(ctx, SVICP_ERR_ARG, "svicp_get_normals: NULL argument");
    if (!ctx->have_target) return fail(ctx, SVICP_ERR_ARG, "svicp_get_normals: no target set");
    if (ctx->lean()) return fail(ctx, SVICP_ERR_ARG, "svicp_get_normals: not kept for a pair of an arena batch");
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->m > 0)
        HIPCHK(hipMemcpyAsync(normals_aos_out, ctx->scr->nrm_aos.p, sizeof(float) * 3 * ctx->m,
                              hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SVICP_OK;
}

int svicp_set_source(svicp_ctx *ctx, const float *xyz_aos, int64_t n)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (n < 0 || (n > 0 && !xyz_aos))
        return fail(ctx, SVICP_ERR_ARG, "svicp_set_source: bad cloud pointer or size");
    if (n > (int64_t)0x7fffff00)
        return fail(ctx, SVICP_ERR_ARG, "svicp_set_source: more than 2^31 points");
    HIPCHK(hipSetDevice(ctx->device));
    if (n != ctx->n) drop_graph(ctx);
    ctx->have_source = false;
    const int64_t nn = std::max<int64_t>(n, 1);
    DevBuf *soa[] = { &ctx->sx0, &ctx->sy0, &ctx->sz0, &ctx->wx, &ctx->wy, &ctx->wz,
                      &ctx->st_lb };
    for (DevBuf *b : soa) {
        void *before = b->p;
        HIPCHK(b->reserve(sizeof(float) * nn));
        if (b->p != before) drop_graph(ctx);
    }
    {
        // what the loop remembers per point: the matched target point (16 bytes: the
        // point-to-point variant keeps the weight in the fourth word) and its normal (12)
        void *before_tp = ctx->st_tp.p, *before_nv = ctx->st_nv.p;
        HIPCHK(ctx->st_tp.reserve(sizeof(float4) * nn));
        HIPCHK(ctx->st_nv.reserve(sizeof(Float3) * nn));
        if (ctx->st_tp.p != before_tp || ctx->st_nv.p != before_nv) drop_graph(ctx);
        // a large source: per wave of every tile the box of its points, from which a launch
        // late in the registration proves most tiles static without reading them
        const int64_t tiles = cdiv(nn, (int64_t)kItTile);
        if (tiles >= kManyTiles && !ctx->lean()) {
            void *before_box = ctx->tile_box.p;
            // (behind the records: three rows of a flag bit per tile, "had work in that launch")
            const size_t records = sizeof(float4) * kBoxRecord * tiles, rows = sizeof(unsigned long long) * kFlagRows * cdiv(tiles, (int64_t)64);
            HIPCHK(ctx->tile_box.reserve(records + rows));
            if (ctx->tile_box.p != before_box) {
                drop_graph(ctx);
                HIPCHK(hipMemsetAsync(ctx->tile_box.as<char>() + records, 0, rows, ctx->stream));
            }
        } else if (ctx->tile_box.p) {
            ctx->tile_box.release();
            drop_graph(ctx);
        }
    }
    HIPCHK(ctx->scr->tmp_aos.reserve(sizeof(float) * 3 * nn));
    ctx->n = n;
    ctx->cur = 0;
    ctx->sorted = false;
    if (n > 0) {
        HIPCHK(hipMemcpyAsync(ctx->scr->tmp_aos.p, xyz_aos, sizeof(float) * 3 * n, hipMemcpyHostToDevice,
                              ctx->stream));
        const int *perm = nullptr;
        if (ctx->opt.sort_source && n > 1) {
            float mn[3], mx[3];
            int rc = device_bbox(ctx, ctx->scr->tmp_aos.as<float>(), n, mn, mx);
            if (rc) return rc;
            float ext = 0.f;
            for (int a = 0; a < 3; ++a) ext = std::max(ext, mx[a] - mn[a]);
            if (std::isfinite(ext) && ext > 0.f) {
                HIPCHK(ctx->scr->keys_a.reserve(sizeof(unsigned long long) * nn));
                HIPCHK(ctx->scr->vals_a.reserve(sizeof(int) * nn));
                hipLaunchKernelGGL(morton_key_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream,
                                   ctx->scr->tmp_aos.as<float>(), (int)n, mn[0], mn[1], mn[2],
                                   (float)kMortonMax / ext, ctx->scr->keys_a.as<unsigned long long>(),
                                   ctx->scr->vals_a.as<int>());
                LAUNCHCHK("morton_key_kernel");
                rc = sort_pairs(ctx, n, 3 * kMortonBits);
                if (rc) return rc;
                void *before = ctx->perm.p;
                HIPCHK(ctx->perm.reserve(sizeof(int) * nn));
                if (ctx->perm.p != before) drop_graph(ctx);
                hipLaunchKernelGGL(deal_tiles_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream,
                                   ctx->scr->vals_b.as<int>(), (int)n, ctx->perm.as<int>());
                LAUNCHCHK("deal_tiles_kernel");
                perm = ctx->perm.as<int>();
                ctx->sorted = true;
            }
        }
        hipLaunchKernelGGL(aos_to_soa_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream,
                           ctx->scr->tmp_aos.as<float>(), perm, (int)n, ctx->sx0.as<float>(),
                           ctx->sy0.as<float>(), ctx->sz0.as<float>());
        LAUNCHCHK("aos_to_soa_kernel");
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->have_source = true;
    ctx->total_points = 0;          // a sharded run agrees on the total again
    ctx->explicit_total = 0;
    return update_fixed_point(ctx);
}

int svicp_host_alloc(int64_t bytes, void **out)
{
    if (!out || bytes <= 0) return fail(nullptr, SVICP_ERR_ARG, "svicp_host_alloc: bad argument");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) { *out = nullptr; return fail(nullptr, SVICP_ERR_HIP, "hipHostMalloc", e); }
    return SVICP_OK;
}

int svicp_host_free(void *host_ptr)
{
    if (!host_ptr) return SVICP_OK;
    hipError_t e = hipHostFree(host_ptr);
    if (e != hipSuccess) return fail(nullptr, SVICP_ERR_HIP, "hipHostFree", e);
    return SVICP_OK;
}

int svicp_host_register(void *host_ptr, int64_t bytes)
{
    if (!host_ptr || bytes <= 0) return fail(nullptr, SVICP_ERR_ARG, "svicp_host_register: bad argument");
    hipError_t e = hipHostRegister(host_ptr, (size_t)bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        // whole pages: the pages that hold the buffer's first and last bytes are the caller's too
        (void)hipGetLastError();
        const uintptr_t page = 4096, lo = (uintptr_t)host_ptr & ~(page - 1),
                        hi = ((uintptr_t)host_ptr + (uintptr_t)bytes + page - 1) & ~(page - 1);
        e = hipHostRegister((void *)lo, (size_t)(hi - lo), hipHostRegisterDefault);
    }
    if (e != hipSuccess) return fail(nullptr, SVICP_ERR_HIP, "hipHostRegister", e);
    return SVICP_OK;
}

int svicp_host_unregister(void *host_ptr)
{
    if (!host_ptr) return SVICP_OK;
    hipError_t e = hipHostUnregister(host_ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipHostUnregister((void *)((uintptr_t)host_ptr & ~(uintptr_t)4095));
    }
    if (e != hipSuccess) return fail(nullptr, SVICP_ERR_HIP, "hipHostUnregister", e);
    return SVICP_OK;
}

int svicp_get_source(svicp_ctx *ctx, float *xyz_aos_out)
{
    if (!ctx || !xyz_aos_out) return fail(ctx, SVICP_ERR_ARG, "svicp_get_source: NULL argument");
    if (!ctx->have_source) return fail(ctx, SVICP_ERR_ARG, "svicp_get_source: no source set");
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n = ctx->n;
    if (n > 0) {
        const float *x = ctx->cur ? ctx->wx.as<float>() : ctx->sx0.as<float>();
        const float *y = ctx->cur ? ctx->wy.as<float>() : ctx->sy0.as<float>();
        const float *z = ctx->cur ? ctx->wz.as<float>() : ctx->sz0.as<float>();
        hipLaunchKernelGGL(soa_to_aos_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, x, y, z,
                           ctx->sorted ? ctx->perm.as<int>() : (const int *)nullptr, (int)n,
                           ctx->scr->tmp_aos.as<float>());
        LAUNCHCHK("soa_to_aos_kernel");
        HIPCHK(hipMemcpyAsync(xyz_aos_out, ctx->scr->tmp_aos.p, sizeof(float) * 3 * n,
                              hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SVICP_OK;
}

int svicp_run(svicp_ctx *ctx, int iterations, double T_out[16], svicp_info *info)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (iterations < 1 || !T_out)
        return fail(ctx, SVICP_ERR_ARG, "svicp_run: iterations must be >= 1 and T_out non-NULL");
    if (!ctx->have_target || !ctx->have_source)
        return fail(ctx, SVICP_ERR_ARG, "svicp_run: set the target and the source first");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_iteration_buffers(ctx, iterations);
    if (rc) return rc;

    const bool timed = ctx->opt.time_kernels != 0;
    if (timed) {
        while ((int)ctx->events.size() < 4 * iterations) {
            hipEvent_t ev;
            HIPCHK(hipEventCreate(&ev));
            ctx->events.push_back(ev);
        }
    }
    // (the legacy default stream, svicp_set_stream(ctx, NULL), cannot be captured: eager there)
    if (ctx->opt.use_graph && !timed && ctx->stream != nullptr) {
        if (!ctx->gexec || ctx->g_iters != iterations) {
            drop_graph(ctx);
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            rc = enqueue_loop(ctx, iterations, false, own_bufs(ctx), ctx->h_state_dev, nullptr);
            hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            if (e != hipSuccess) return fail(ctx, SVICP_ERR_HIP, "hipStreamEndCapture", e);
            e = hipGraphInstantiate(&ctx->gexec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess) { ctx->gexec = nullptr; return fail(ctx, SVICP_ERR_HIP, "hipGraphInstantiate", e); }
            ctx->g_iters = iterations;
        }
        HIPCHK(hipGraphLaunch(ctx->gexec, ctx->stream));
    } else {
        rc = enqueue_loop(ctx, iterations, timed, own_bufs(ctx), ctx->h_state_dev, nullptr);
        if (rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));   // the last kernel wrote h_state
    ctx->cur = iterations > 1 ? 1 : 0;
    ctx->last_iters = iterations;
    ctx->last_was_p2p = false;
    std::memcpy(T_out, ctx->h_state->Trans, sizeof(double) * 16);
    fill_info(ctx, iterations, info);

    if (timed) {
        double fa = 0, so = 0;
        ctx->fused_it.assign(iterations, 0.0);
        ctx->solve_it.assign(iterations, 0.0);
        for (int k = 0; k < iterations; ++k) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, ctx->events[4 * k + 0], ctx->events[4 * k + 1]));
            fa += ms;
            ctx->fused_it[k] = ms;
            HIPCHK(hipEventElapsedTime(&ms, ctx->events[4 * k + 2], ctx->events[4 * k + 3]));
            so += ms;
            ctx->solve_it[k] = ms;
        }
        ctx->fused_ms = fa / iterations;
        ctx->solve_ms = so / iterations;
        ctx->timed_launches = iterations;
    }
    if (ctx->h_state->status == SVICP_ERR_SINGULAR)
        return fail(ctx, SVICP_ERR_SINGULAR, "svicp_run: normal matrix not positive definite (too few correspondences?)");
    return SVICP_OK;
}

int svicp_run_enqueue(svicp_ctx *ctx, int iterations, double *dev_out20)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (!dev_out20) return fail(ctx, SVICP_ERR_ARG, "svicp_run_enqueue: dev_out20 is NULL");
    if (iterations < 1 || iterations > kEnqMaxIters)
        return fail(ctx, SVICP_ERR_ARG, "svicp_run_enqueue: iterations must be in [1, 128]");
    if (!ctx->have_target || !ctx->have_source)
        return fail(ctx, SVICP_ERR_ARG, "svicp_run_enqueue: set the target and the source first");
    const int nb = std::max(1, cdiv(std::max<int64_t>(ctx->n, 1), kItTile));
    if (nb > kEnqMaxTiles)
        return fail(ctx, SVICP_ERR_ARG,
                    "svicp_run_enqueue: the source exceeds the workspace (4194304 points at most)");
    if (ctx->enq_slot < 0) {
        ctx->enq_slot = take_scratch_slot(ctx->device);
        if (ctx->enq_slot < 0)
            return fail(ctx, SVICP_ERR_ARG,
                        "svicp_run_enqueue: all 8 workspaces of this device belong to other contexts "
                        "(a context keeps the one it took until svicp_destroy)");
    }
    HIPCHK(hipSetDevice(ctx->device));
    if (nb != ctx->nb) drop_graph(ctx);     // (svicp_run's graph bakes the tile count in)
    ctx->nb = nb; ctx->chunk = kItTile;
    // the device address, never the array's name; this context's own workspace inside it
    unsigned char *w = ctx->enq_scratch + (size_t)ctx->enq_slot * kEnqBytes;
    LoopBufs b;
    b.state = reinterpret_cast<IcpState *>(w + kEnqOffState);
    b.totals = reinterpret_cast<fx_t *>(w + kEnqOffTotals);
    b.deltas = reinterpret_cast<fx_t *>(w + kEnqOffDeltas);
    b.xhist = reinterpret_cast<double *>(w + kEnqOffXhist);
    b.ihist = reinterpret_cast<double *>(w + kEnqOffIhist);
    b.flags = nb <= kFlagCap ? reinterpret_cast<unsigned long long *>(w + kEnqOffFlags) : nullptr;
    b.partials = reinterpret_cast<double *>(w + kEnqOffPartials);
    // The array keeps the last call's contents.  What a run takes to be zero when it starts:
    // row 0 of the totals and rows 0 and 1 of the per-iteration changes (every later row is
    // cleared by the launch before the one that adds to it).  Everything else is written
    // before it is read: the state by the first solve, the tiles' sums and flags by
    // iteration 0.
    hipLaunchKernelGGL(clear_run_rows_kernel, dim3(1), dim3(64), 0, ctx->stream, b.totals, b.deltas);
    LAUNCHCHK("clear_run_rows_kernel");
    int rc = enqueue_loop(ctx, iterations, false, b, nullptr, dev_out20);
    if (rc) return rc;
    ctx->cur = iterations > 1 ? 1 : 0;
    ctx->last_iters = 0;                    // the history stays in the workspace
    ctx->last_was_p2p = false;
    return SVICP_OK;
}

int svicp_last_iterations(const svicp_ctx *ctx, int *iterations)
{
    if (!ctx || !iterations) return SVICP_ERR_ARG;
    *iterations = ctx->last_iters;
    return SVICP_OK;
}

int svicp_get_history(svicp_ctx *ctx, int capacity_rows, double *X_out, double *info_out)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (ctx->last_iters <= 0) return fail(ctx, SVICP_ERR_ARG, "svicp_get_history: no run yet");
    if (capacity_rows < ctx->last_iters)
        return fail(ctx, SVICP_ERR_ARG, "svicp_get_history: capacity_rows is below the last run's iterations");
    HIPCHK(hipSetDevice(ctx->device));
    if (X_out)
        HIPCHK(hipMemcpyAsync(X_out, ctx->xhist.p, sizeof(double) * 6 * ctx->last_iters,
                              hipMemcpyDeviceToHost, ctx->stream));
    if (info_out)
        HIPCHK(hipMemcpyAsync(info_out, ctx->ihist.p, sizeof(double) * 4 * ctx->last_iters,
                              hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SVICP_OK;
}

int svicp_search_counts(svicp_ctx *ctx, int capacity_rows, double *counts_out)
{
    if (!ctx || !counts_out) return fail(ctx, SVICP_ERR_ARG, "svicp_search_counts: NULL argument");
    if (ctx->last_iters <= 0 || ctx->last_was_p2p)
        return fail(ctx, SVICP_ERR_ARG, "svicp_search_counts: no point-to-plane run yet");
    if (capacity_rows < ctx->last_iters)
        return fail(ctx, SVICP_ERR_ARG, "svicp_search_counts: capacity_rows is below the last run's iterations");
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<fx_t> rows((size_t)kDeltaRow * ctx->last_iters);
    HIPCHK(hipMemcpyAsync(rows.data(), ctx->deltas.p, sizeof(fx_t) * rows.size(),
                          hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // row 0 was cleared for the next run; iteration 0 searches every point
    counts_out[0] = (double)ctx->n;
    for (int k = 1; k < ctx->last_iters; ++k) {
        double c = 0.0;
        for (int cp = 0; cp < kDeltaCopies; ++cp) c += (double)rows[(size_t)k * kDeltaRow + cp * kNumSums + 31];
        counts_out[k] = c;
    }
    return SVICP_OK;
}

int svicp_tiles_left_unread(svicp_ctx *ctx, int capacity_rows, double *counts_out)
{
    if (!ctx || !counts_out) return fail(ctx, SVICP_ERR_ARG, "svicp_tiles_left_unread: NULL argument");
    if (ctx->last_iters <= 0 || ctx->last_was_p2p)
        return fail(ctx, SVICP_ERR_ARG, "svicp_tiles_left_unread: no point-to-plane run yet");
    if (capacity_rows < ctx->last_iters)
        return fail(ctx, SVICP_ERR_ARG, "svicp_tiles_left_unread: capacity_rows is below the last run's iterations");
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<fx_t> rows((size_t)kDeltaRow * ctx->last_iters);
    HIPCHK(hipMemcpyAsync(rows.data(), ctx->deltas.p, sizeof(fx_t) * rows.size(),
                          hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    counts_out[0] = 0.0;      // (row 0 was cleared for the next run; iteration 0 reads every tile)
    for (int k = 1; k < ctx->last_iters; ++k) {
        double c = 0.0;
        for (int cp = 0; cp < kDeltaCopies; ++cp) c += (double)rows[(size_t)k * kDeltaRow + cp * kNumSums + 30];
        counts_out[k] = c;
    }
    return SVICP_OK;
}

int svicp_debug_use_tile_boxes(svicp_ctx *ctx, int on)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (ctx->use_tile_box != (on != 0)) drop_graph(ctx);
    ctx->use_tile_box = on != 0;
    return SVICP_OK;
}

int svicp_correspondences(svicp_ctx *ctx, int32_t *idx_out, float *d2_out)
{
    if (!ctx || !idx_out || !d2_out)
        return fail(ctx, SVICP_ERR_ARG, "svicp_correspondences: NULL argument");
    if (!ctx->have_target || !ctx->have_source)
        return fail(ctx, SVICP_ERR_ARG, "svicp_correspondences: set the target and the source first");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = run_nn1(ctx, false);
    if (rc) return rc;
    if (ctx->n > 0) {
        HIPCHK(hipMemcpyAsync(idx_out, ctx->scr->knn_idx.p, sizeof(int) * ctx->n, hipMemcpyDeviceToHost,
                              ctx->stream));
        HIPCHK(hipMemcpyAsync(d2_out, ctx->scr->knn_d2.p, sizeof(float) * ctx->n, hipMemcpyDeviceToHost,
                              ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SVICP_OK;
}

int svicp_knn(svicp_ctx *ctx, const float *query_aos, int64_t n, int k, int32_t *idx_out,
              float *d2_out)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (n < 0 || k < 1 || (n > 0 && (!query_aos || !idx_out || !d2_out)))
        return fail(ctx, SVICP_ERR_ARG, "svicp_knn: bad argument");
    if (!ctx->have_target) return fail(ctx, SVICP_ERR_ARG, "svicp_knn: no target set");
    if (ctx->lean()) return fail(ctx, SVICP_ERR_ARG, "svicp_knn: the cell tables are not kept for a pair of an arena batch");
    if (k > 128) k = 128;   // CKNNSearch::setKCount clamps the same way
    if (n == 0) return SVICP_OK;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(ctx->scr->tmp_aos.reserve(sizeof(float) * 3 * n));
    HIPCHK(ctx->scr->knn_idx.reserve(sizeof(int) * n * k));
    HIPCHK(ctx->scr->knn_d2.reserve(sizeof(float) * n * k));
    HIPCHK(hipMemcpyAsync(ctx->scr->tmp_aos.p, query_aos, sizeof(float) * 3 * n, hipMemcpyHostToDevice,
                          ctx->stream));
    int rc;
    if (k == 1 && ctx->m > 0 && n <= 0x7fffff00) {
        // one neighbour (what the reference's loop asks for, CPoint2PlaneICP.cpp:65-67): the
        // thread-per-query exact search, twenty times the rate of the wave-per-query k-NN
        HIPCHK(ctx->scr->q_soa.reserve(sizeof(float) * 3 * n));
        float *qx = ctx->scr->q_soa.as<float>(), *qy = qx + n, *qz = qy + n;
        hipLaunchKernelGGL(aos_to_soa_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream,
                           ctx->scr->tmp_aos.as<float>(), (const int *)nullptr, (int)n, qx, qy, qz);
        LAUNCHCHK("aos_to_soa_kernel");
        rc = run_nn1_on(ctx, qx, qy, qz, nullptr, n, true, ctx->scr->knn_idx.as<int>(),
                        ctx->scr->knn_d2.as<float>());
    } else {
        rc = run_knn(ctx, ctx->scr->tmp_aos.as<float>(), n, k, ctx->scr->knn_idx.as<int>(),
                     ctx->scr->knn_d2.as<float>());
    }
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(idx_out, ctx->scr->knn_idx.p, sizeof(int) * n * k, hipMemcpyDeviceToHost,
                          ctx->stream));
    HIPCHK(hipMemcpyAsync(d2_out, ctx->scr->knn_d2.p, sizeof(float) * n * k, hipMemcpyDeviceToHost,
                          ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SVICP_OK;
}

int svicp_comm_unique_id(unsigned char id_out[128])
{
    if (!id_out) return fail(nullptr, SVICP_ERR_ARG, "svicp_comm_unique_id: NULL argument");
    RcclApi *api = rccl_api();
    if (!api->handle) return fail(nullptr, SVICP_ERR_COMM, api->error.c_str());
    static_assert(sizeof(ncclUniqueId) == 128, "the identifier travels as 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return fail_rccl(nullptr, "ncclGetUniqueId", r);
    std::memcpy(id_out, &id, sizeof id);
    return SVICP_OK;
}

int svicp_comm_init(svicp_ctx *ctx, const unsigned char id[128], int rank, int world_size)
{
    if (!ctx || !id) return fail(ctx, SVICP_ERR_ARG, "svicp_comm_init: NULL argument");
    if (world_size < 1 || rank < 0 || rank >= world_size)
        return fail(ctx, SVICP_ERR_ARG, "svicp_comm_init: rank must be in [0, world_size)");
    if (ctx->comm) return fail(ctx, SVICP_ERR_ARG, "svicp_comm_init: the context already has a communicator");
    RcclApi *api = rccl_api();
    if (!api->handle) return fail(ctx, SVICP_ERR_COMM, api->error.c_str());
    HIPCHK(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    ncclResult_t r = api->CommInitRank(&comm, world_size, uid, rank);
    if (r != ncclSuccess) return fail_rccl(ctx, "ncclCommInitRank", r);
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world_size;
    drop_graph(ctx);
    return SVICP_OK;
}

int svicp_comm_destroy(svicp_ctx *ctx)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (!ctx->comm) return SVICP_OK;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    drop_graph(ctx);
    ncclResult_t r = rccl_api()->CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
    if (r != ncclSuccess) return fail_rccl(ctx, "ncclCommDestroy", r);
    return SVICP_OK;
}

int svicp_shard_run(svicp_ctx *ctx, int iterations, double T_out[16], svicp_info *info)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (iterations < 1 || !T_out)
        return fail(ctx, SVICP_ERR_ARG, "svicp_shard_run: iterations must be >= 1 and T_out non-NULL");
    if (!ctx->have_target || !ctx->have_source)
        return fail(ctx, SVICP_ERR_ARG, "svicp_shard_run: set the target and this rank's part of the source first");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_iteration_buffers(ctx, iterations);
    if (rc) return rc;
    if (ctx->comm && ctx->comm_world > 1) {
        // Every rank must scale its sums alike.  Whether the ranks exchange anything here must
        // not depend on one rank's own history (a rank that set a new part of the source, or
        // called svicp_shard_set_total, while another did not): EVERY call makes the same one
        // blocking exchange of two 64-bit integers, {this rank's points, the total it was
        // told or 0}, maximum over the ranks.  The total is the largest one any rank was told,
        // else the largest part times the number of ranks; the table is uploaded again only
        // when that changes.
        long long *slot = reinterpret_cast<long long *>(ctx->rt_apart.as<double>() + 12);
        long long mine[2] = { (long long)ctx->n, (long long)ctx->explicit_total }, most[2] = { 0, 0 };
        HIPCHK(hipMemcpyAsync(slot, mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
        ncclResult_t r = rccl_api()->AllReduce(slot, slot, 2, ncclInt64, ncclMax, ctx->comm, ctx->stream);
        if (r != ncclSuccess) return fail_rccl(ctx, "ncclAllReduce(max points)", r);
        HIPCHK(hipMemcpyAsync(most, slot, sizeof most, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const int64_t agreed = most[1] > 0 ? std::max<int64_t>(most[1], most[0])
                                           : (int64_t)most[0] * ctx->comm_world;
        if (agreed != ctx->total_points) {
            ctx->total_points = agreed;
            rc = update_fixed_point(ctx);
            if (rc) return rc;
        }
    }
    const bool capture = ctx->opt.use_graph && ctx->stream != nullptr && !ctx->shard_capture_failed;
    bool launched = false;
    std::string capture_error;
    if (capture) {
        if (!ctx->gexec_shard || ctx->g_iters_shard != iterations) {
            if (ctx->gexec_shard) { (void)hipGraphExecDestroy(ctx->gexec_shard); ctx->gexec_shard = nullptr; }
            ctx->g_iters_shard = -1;
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            rc = enqueue_shard_loop(ctx, iterations);
            hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
            if (rc == SVICP_OK && e == hipSuccess && graph)
                e = hipGraphInstantiate(&ctx->gexec_shard, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
            if (rc != SVICP_OK || e != hipSuccess || !ctx->gexec_shard) {
                // a runtime that cannot capture the collective: launch one by one from now on
                // (what the capture failed with is kept: if the launches one by one fail too,
                // both messages are reported)
                capture_error = rc != SVICP_OK ? ctx->err
                                               : std::string("capture: ") + hipGetErrorString(e);
                (void)hipGetLastError();
                ctx->gexec_shard = nullptr;
                ctx->shard_capture_failed = true;
            } else {
                ctx->g_iters_shard = iterations;
            }
        }
        if (ctx->gexec_shard) {
            HIPCHK(hipGraphLaunch(ctx->gexec_shard, ctx->stream));
            launched = true;
        }
    }
    if (!launched) {
        rc = enqueue_shard_loop(ctx, iterations);
        if (rc) {
            if (!capture_error.empty()) ctx->err += " (after the graph capture had failed with: " + capture_error + ")";
            return rc;
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));   // the last kernel wrote h_state
    ctx->cur = iterations > 1 ? 1 : 0;
    ctx->last_iters = iterations;
    ctx->last_was_p2p = false;
    std::memcpy(T_out, ctx->h_state->Trans, sizeof(double) * 16);
    fill_info(ctx, iterations, info);
    if (ctx->h_state->status == SVICP_ERR_SINGULAR)
        return fail(ctx, SVICP_ERR_SINGULAR, "svicp_shard_run: normal matrix not positive definite");
    return SVICP_OK;
}

int svicp_shard_run_mode(const svicp_ctx *ctx, int *captured, int *world_size)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (captured) *captured = ctx->gexec_shard != nullptr ? 1 : 0;
    if (world_size) *world_size = ctx->comm ? ctx->comm_world : 1;
    return SVICP_OK;
}

int svicp_shard_set_total(svicp_ctx *ctx, int64_t total_points)
{
    if (!ctx || total_points < 0) return fail(ctx, SVICP_ERR_ARG, "svicp_shard_set_total: bad argument");
    if (!ctx->have_target || !ctx->have_source)
        return fail(ctx, SVICP_ERR_ARG, "svicp_shard_set_total: set the target and the source first");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->explicit_total = std::max<int64_t>(total_points, ctx->n);
    ctx->total_points = ctx->explicit_total;
    return update_fixed_point(ctx);
}

int svicp_shard_begin(svicp_ctx *ctx)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (!ctx->have_target || !ctx->have_source)
        return fail(ctx, SVICP_ERR_ARG, "svicp_shard_begin: set the target and the source first");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_iteration_buffers(ctx, 1024);   // room for this many sharded iterations
    if (rc) return rc;
    hipLaunchKernelGGL(reset_state_kernel, dim3(1), dim3(64), 0, ctx->stream,
                       ctx->state.as<IcpState>());
    LAUNCHCHK("reset_state_kernel");
    HIPCHK(hipMemsetAsync(ctx->totals.p, 0, sizeof(double) * kNumSums, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->deltas.p, 0, sizeof(double) * kDeltaRow * ctx->hist_cap, ctx->stream));
    ctx->shard_iter = 0;
    ctx->cur = 0;
    return SVICP_OK;
}

int svicp_shard_accumulate(svicp_ctx *ctx, double *dev_partial32)
{
    if (!ctx || !dev_partial32) return fail(ctx, SVICP_ERR_ARG, "svicp_shard_accumulate: NULL argument");
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->shard_iter >= ctx->hist_cap)
        return fail(ctx, SVICP_ERR_ARG, "svicp_shard_accumulate: too many iterations since svicp_shard_begin");
    const int k = ctx->shard_iter;
    launch_iteration(ctx, k, own_bufs(ctx));
    LAUNCHCHK("icp_iterate_kernel");
    // this rank's change of the sums in iteration k
    hipLaunchKernelGGL(fold_delta_kernel, dim3(1), dim3(64), 0, ctx->stream,
                       ctx->deltas.as<fx_t>() + (size_t)k * kDeltaRow, reinterpret_cast<fx_t *>(dev_partial32));
    LAUNCHCHK("fold_delta_kernel");
    if (ctx->shard_iter >= 1) ctx->cur = 1;
    return SVICP_OK;
}

int svicp_shard_update(svicp_ctx *ctx, const double *dev_sum32)
{
    if (!ctx || !dev_sum32) return fail(ctx, SVICP_ERR_ARG, "svicp_shard_update: NULL argument");
    HIPCHK(hipSetDevice(ctx->device));
    // the summed change replaces this rank's own; the next svicp_shard_accumulate
    // (or _finish) adds it to the running totals and solves
    hipLaunchKernelGGL(unfold_delta_kernel, dim3(1), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const fx_t *>(dev_sum32),
                       ctx->deltas.as<fx_t>() + (size_t)ctx->shard_iter * kDeltaRow);
    LAUNCHCHK("unfold_delta_kernel");
    ctx->shard_iter++;
    ctx->last_iters = ctx->shard_iter;
    return SVICP_OK;
}

int svicp_shard_finish(svicp_ctx *ctx, double T_out[16], svicp_info *info)
{
    if (!ctx || !T_out) return fail(ctx, SVICP_ERR_ARG, "svicp_shard_finish: NULL argument");
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->shard_iter < 1) return fail(ctx, SVICP_ERR_ARG, "svicp_shard_finish: no iteration was run");
    hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->totals.as<fx_t>(),
                       ctx->deltas.as<fx_t>(), ctx->fx_exp.as<int>(), ctx->state.as<IcpState>(),
                       ctx->xhist.as<double>(), ctx->ihist.as<double>(), ctx->shard_iter - 1,
                       (IcpState *)nullptr, (double *)nullptr);
    LAUNCHCHK("solve_kernel");
    HIPCHK(hipMemcpyAsync(ctx->h_state, ctx->state.p, sizeof(IcpState), hipMemcpyDeviceToHost,
                          ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::memcpy(T_out, ctx->h_state->Trans, sizeof(double) * 16);
    fill_info(ctx, ctx->shard_iter, info);
    if (ctx->h_state->status == SVICP_ERR_SINGULAR)
        return fail(ctx, SVICP_ERR_SINGULAR, "svicp_shard_finish: normal matrix not positive definite");
    return SVICP_OK;
}

#ifdef SVICP_FAR_STATS
int svicp_debug_far_stats(unsigned long long *out8, int clear)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(svicp::g_far_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (clear) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(svicp::g_far_stats), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif

#ifdef SVICP_TRACE
// development builds only: the phase stamps of the last run (see SVICP_STAMP)
int svicp_debug_trace(unsigned long long *out, size_t count)
{
    const size_t all = (size_t)kTraceIters * kTraceTiles * kTraceStamps;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * std::min(count, all))
               == hipSuccess ? 0 : SVICP_ERR_HIP;
}
int svicp_debug_trace_clear(void)
{
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_trace)) != hipSuccess) return SVICP_ERR_HIP;
    return hipMemset(p, 0, sizeof(unsigned long long) * kTraceIters * kTraceTiles * kTraceStamps)
               == hipSuccess ? 0 : SVICP_ERR_HIP;
}
#endif

int svicp_kernel_times(svicp_ctx *ctx, double *fused_ms, double *solve_ms, int *launches)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (fused_ms) *fused_ms = ctx->fused_ms;
    if (solve_ms) *solve_ms = ctx->solve_ms;
    if (launches) *launches = ctx->timed_launches;
    return SVICP_OK;
}

int svicp_iteration_times(svicp_ctx *ctx, double *fused_ms_out, double *solve_ms_out)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (fused_ms_out) std::copy(ctx->fused_it.begin(), ctx->fused_it.end(), fused_ms_out);
    if (solve_ms_out) std::copy(ctx->solve_it.begin(), ctx->solve_it.end(), solve_ms_out);
    return SVICP_OK;
}

int svicp_time_nn(svicp_ctx *ctx, int exact, int reps, double *avg_ms)
{
    if (!ctx || !avg_ms || reps < 1) return fail(ctx, SVICP_ERR_ARG, "svicp_time_nn: bad argument");
    if (!ctx->have_target || !ctx->have_source)
        return fail(ctx, SVICP_ERR_ARG, "svicp_time_nn: set the target and the source first");
    if (ctx->lean() && exact)
        return fail(ctx, SVICP_ERR_ARG, "svicp_time_nn: the cell tables are not kept for a pair of an arena batch");
    HIPCHK(hipSetDevice(ctx->device));
    ScopedEvent a, b;
    HIPCHK(a.create());
    HIPCHK(b.create());
    int rc = run_nn1(ctx, exact != 0);   // warm-up
    if (rc) return rc;
    HIPCHK(hipEventRecord(a.ev, ctx->stream));
    for (int r = 0; r < reps; ++r) {
        rc = run_nn1(ctx, exact != 0);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(b.ev, ctx->stream));
    HIPCHK(hipEventSynchronize(b.ev));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, a.ev, b.ev));
    *avg_ms = (double)ms / reps;
    return SVICP_OK;
}

int svicp_time_gather(svicp_ctx *ctx, int real_addresses, int per_query, int reps, double *avg_ms,
                      double *wave_instructions)
{
    if (!ctx || !avg_ms || reps < 1 || per_query < 0 || per_query > 4096)
        return fail(ctx, SVICP_ERR_ARG, "svicp_time_gather: bad argument");
    if (!ctx->have_target || !ctx->have_source || ctx->n == 0 || ctx->m == 0)
        return fail(ctx, SVICP_ERR_ARG, "svicp_time_gather: set a non-empty target and source first");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(ctx->scr->n_unres.reserve(sizeof(unsigned long long) * 2));
    const int64_t n = ctx->n;
    const float *x = ctx->cur ? ctx->wx.as<float>() : ctx->sx0.as<float>();
    const float *y = ctx->cur ? ctx->wy.as<float>() : ctx->sy0.as<float>();
    const float *z = ctx->cur ? ctx->wz.as<float>() : ctx->sz0.as<float>();
    const int grid = std::min(cdiv(n, kBlockThreads), ctx->num_cus * 8);
    unsigned long long *cnt = ctx->scr->n_unres.as<unsigned long long>();
    auto launch = [&](unsigned long long *count) {
        if (real_addresses == 12)
            hipLaunchKernelGGL((gather_probe_kernel<true, 12>), dim3(grid), dim3(kBlockThreads), 0, ctx->stream,
                               ctx->g, x, y, z, (int)n, per_query, reinterpret_cast<unsigned *>(cnt + 1), count);
        else if (real_addresses == 8)
            hipLaunchKernelGGL((gather_probe_kernel<true, 8>), dim3(grid), dim3(kBlockThreads), 0, ctx->stream,
                               ctx->g, x, y, z, (int)n, per_query, reinterpret_cast<unsigned *>(cnt + 1), count);
        else if (real_addresses == 4)
            hipLaunchKernelGGL((gather_probe_kernel<true, 4>), dim3(grid), dim3(kBlockThreads), 0, ctx->stream,
                               ctx->g, x, y, z, (int)n, per_query, reinterpret_cast<unsigned *>(cnt + 1), count);
        else if (real_addresses)
            hipLaunchKernelGGL((gather_probe_kernel<true>), dim3(grid), dim3(kBlockThreads), 0, ctx->stream,
                               ctx->g, x, y, z, (int)n, per_query, reinterpret_cast<unsigned *>(cnt + 1), count);
        else
            hipLaunchKernelGGL((gather_probe_kernel<false>), dim3(grid), dim3(kBlockThreads), 0, ctx->stream,
                               ctx->g, x, y, z, (int)n, per_query, reinterpret_cast<unsigned *>(cnt + 1), count);
    };
    ScopedEvent a, b;
    HIPCHK(a.create());
    HIPCHK(b.create());
    HIPCHK(hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * 2, ctx->stream));
    launch(cnt);                                // warm-up; the only launch that counts
    LAUNCHCHK("gather_probe_kernel");
    HIPCHK(hipEventRecord(a.ev, ctx->stream));
    for (int r = 0; r < reps; ++r) launch(nullptr);
    LAUNCHCHK("gather_probe_kernel");
    HIPCHK(hipEventRecord(b.ev, ctx->stream));
    int rc = ensure_small(ctx);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->scr->h_small, cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, a.ev, b.ev));
    *avg_ms = (double)ms / reps;
    if (wave_instructions)
        *wave_instructions = (double)*reinterpret_cast<unsigned long long *>(ctx->scr->h_small);
    return SVICP_OK;
}

/* ---- point-to-point variant --------------------------------------------- */

int svicp_run_point2point(svicp_ctx *ctx, int iterations, double T_out[16], svicp_info *info)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (iterations < 1 || !T_out)
        return fail(ctx, SVICP_ERR_ARG, "svicp_run_point2point: iterations must be >= 1 and T_out non-NULL");
    if (!ctx->have_target || !ctx->have_source)
        return fail(ctx, SVICP_ERR_ARG, "svicp_run_point2point: set the target and the source first");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_iteration_buffers(ctx, iterations);
    if (rc) return rc;
    {
        void *a = ctx->p2p_sums.p, *b = ctx->rthist.p;
        HIPCHK(ctx->p2p_sums.reserve(sizeof(double) * kDeltaRow * ctx->hist_cap));
        HIPCHK(ctx->rthist.reserve(sizeof(double) * 12 * ctx->hist_cap));
        if (a != ctx->p2p_sums.p || b != ctx->rthist.p) drop_graph(ctx);
    }
    const int n = (int)ctx->n;
    const int nb = std::max(1, cdiv(std::max(n, 1), kP2PTile));
    auto enqueue = [&]() -> int {
        float *wx = ctx->wx.as<float>(), *wy = ctx->wy.as<float>(), *wz = ctx->wz.as<float>();
        const float *x0 = ctx->sx0.as<float>(), *y0 = ctx->sy0.as<float>(), *z0 = ctx->sz0.as<float>();
        IcpState *state = ctx->state.as<IcpState>();
        double *sums = ctx->p2p_sums.as<double>();
        float4 *tp = ctx->st_tp.as<float4>();
        const float thr = ctx->opt.dist_threshold;
        hipLaunchKernelGGL(reset_state_kernel, dim3(1), dim3(64), 0, ctx->stream, state);
        LAUNCHCHK("reset_state_kernel");
        HIPCHK(hipMemsetAsync(sums, 0, sizeof(double) * kDeltaRow * iterations, ctx->stream));
        for (int k = 0; k < iterations; ++k) {
            double *sk = sums + (size_t)k * kDeltaRow;
            const float *cx = k <= 1 ? x0 : wx, *cy = k <= 1 ? y0 : wy, *cz = k <= 1 ? z0 : wz;
            if (k == 0)
                hipLaunchKernelGGL((p2p_match_kernel<false>), dim3(nb), dim3(256), 0, ctx->stream,
                                   ctx->g, cx, cy, cz, wx, wy, wz, n, state, thr, tp, sk);
            else
                hipLaunchKernelGGL((p2p_match_kernel<true>), dim3(nb), dim3(256), 0, ctx->stream,
                                   ctx->g, cx, cy, cz, wx, wy, wz, n, state, thr, tp, sk);
            LAUNCHCHK("p2p_match_kernel");
            // the covariance is taken about the positions the match used
            const float *px = k == 0 ? x0 : wx, *py = k == 0 ? y0 : wy, *pz = k == 0 ? z0 : wz;
            hipLaunchKernelGGL(p2p_cov_kernel, dim3(nb), dim3(256), 0, ctx->stream, px, py, pz, n, tp, sk);
            LAUNCHCHK("p2p_cov_kernel");
            hipLaunchKernelGGL(p2p_solve_kernel, dim3(1), dim3(64), 0, ctx->stream, sk, state,
                               ctx->rthist.as<double>(), ctx->ihist.as<double>(), k);
            LAUNCHCHK("p2p_solve_kernel");
        }
        return SVICP_OK;
    };
    if (ctx->opt.use_graph && ctx->stream != nullptr) {
        if (!ctx->gexec_p2p || ctx->g_iters_p2p != iterations) {
            if (ctx->gexec_p2p) { (void)hipGraphExecDestroy(ctx->gexec_p2p); ctx->gexec_p2p = nullptr; }
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            rc = enqueue();
            hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            if (e != hipSuccess) return fail(ctx, SVICP_ERR_HIP, "hipStreamEndCapture", e);
            e = hipGraphInstantiate(&ctx->gexec_p2p, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess) { ctx->gexec_p2p = nullptr; return fail(ctx, SVICP_ERR_HIP, "hipGraphInstantiate", e); }
            ctx->g_iters_p2p = iterations;
        }
        HIPCHK(hipGraphLaunch(ctx->gexec_p2p, ctx->stream));
    } else {
        rc = enqueue();
        if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(ctx->h_state, ctx->state.p, sizeof(IcpState), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->cur = iterations > 1 ? 1 : 0;
    ctx->last_iters = iterations;
    ctx->last_was_p2p = true;
    std::memcpy(T_out, ctx->h_state->Trans, sizeof(double) * 16);
    fill_info(ctx, iterations, info);
    if (ctx->h_state->status == SVICP_ERR_SINGULAR)
        return fail(ctx, SVICP_ERR_SINGULAR, "svicp_run_point2point: no point had a neighbour within the threshold");
    return SVICP_OK;
}

int svicp_get_history_rt(svicp_ctx *ctx, int capacity_rows, double *RT_out, double *info_out)
{
    if (!ctx) return SVICP_ERR_ARG;
    if (ctx->last_iters <= 0 || !ctx->last_was_p2p)
        return fail(ctx, SVICP_ERR_ARG, "svicp_get_history_rt: no point-to-point run yet");
    if (capacity_rows < ctx->last_iters)
        return fail(ctx, SVICP_ERR_ARG, "svicp_get_history_rt: capacity_rows is below the last run's iterations");
    HIPCHK(hipSetDevice(ctx->device));
    if (RT_out)
        HIPCHK(hipMemcpyAsync(RT_out, ctx->rthist.p, sizeof(double) * 12 * ctx->last_iters,
                              hipMemcpyDeviceToHost, ctx->stream));
    if (info_out)
        HIPCHK(hipMemcpyAsync(info_out, ctx->ihist.p, sizeof(double) * 4 * ctx->last_iters,
                              hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SVICP_OK;
}

/* ---- batches of independent pairs -------------------------------------- */

int svicp_batch_create(svicp_ctx **ctxs, int count, svicp_batch **out)
{
    if (!out) return SVICP_ERR_ARG;
    *out = nullptr;
    if (!ctxs || count < 1 || !ctxs[0]) return fail(nullptr, SVICP_ERR_ARG, "svicp_batch_create: bad argument");
    svicp_ctx *ctx = ctxs[0];
    for (int i = 0; i < count; ++i) {
        if (!ctxs[i] || ctxs[i]->device != ctx->device)
            return fail(ctx, SVICP_ERR_ARG, "svicp_batch_create: every context must be on the same device");
        if (!ctxs[i]->have_target || !ctxs[i]->have_source)
            return fail(ctx, SVICP_ERR_ARG, "svicp_batch_create: set every pair's target and source first");
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return fail(ctx, SVICP_ERR_ARG, "svicp_batch_create: a context appears twice");
    }
    svicp_batch *b = new svicp_batch();
    b->ctxs.assign(ctxs, ctxs + count);
    b->lead = ctx;
    *out = b;
    return SVICP_OK;
}

int svicp_batch_destroy(svicp_batch *b)
{
    if (!b) return SVICP_OK;
    if (b->lead) {
        (void)hipSetDevice(b->lead->device);
        (void)hipStreamSynchronize(b->lead->stream);
    }
    if (b->gexec) (void)hipGraphExecDestroy(b->gexec);
    b->pairs.release(); b->block_map.release(); b->states.release(); b->rt_all.release();
    if (b->h_states) (void)hipHostFree(b->h_states);
    if (b->owns) {
        for (svicp_ctx *c : b->ctxs) {
            c->own_stream = nullptr;        // the batch's stream, destroyed below
            (void)svicp_destroy(c);
        }
        b->shared.release();
        b->arena.release();
        if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
    }
    delete b;
    return SVICP_OK;
}

int svicp_batch_create_from_arrays(int device_id, int count, const int64_t *tgt_offsets,
                                   const float *tgt_xyz, const float *tgt_normals,
                                   const int64_t *src_offsets, const float *src_xyz,
                                   const svicp_options *opt, svicp_batch **out)
{
    if (!out) return SVICP_ERR_ARG;
    *out = nullptr;
    if (count < 1 || !tgt_offsets || !tgt_xyz || !src_offsets || !src_xyz)
        return fail(nullptr, SVICP_ERR_ARG, "svicp_batch_create_from_arrays: bad argument");
    for (int i = 0; i < count; ++i)
        if (tgt_offsets[i + 1] < tgt_offsets[i] || src_offsets[i + 1] < src_offsets[i] || tgt_offsets[0] < 0 ||
            src_offsets[0] < 0)
            return fail(nullptr, SVICP_ERR_ARG, "svicp_batch_create_from_arrays: offsets must not decrease");
    // the first pair's context is made the usual way (device checks, the workspace address);
    // the others copy its device facts and share its stream
    svicp_ctx *first = nullptr;
    int rc = svicp_create(device_id, &first);
    if (rc) return rc;
    svicp_batch *b = new svicp_batch();
    b->owns = true;
    b->own_stream = first->own_stream;
    b->host_states.resize(count);
    b->ctxs.reserve(count);
    const auto t_begin = std::chrono::steady_clock::now();
    for (int i = 0; i < count && rc == SVICP_OK; ++i) {
        svicp_ctx *c = first;
        if (i > 0) {
            c = new svicp_ctx();
            c->device = first->device;
            c->num_cus = first->num_cus;
            c->enq_scratch = first->enq_scratch;
            c->own_stream = nullptr;
            c->stream = b->own_stream;
        }
        if (opt) c->opt = *opt; else svicp_default_options(&c->opt);
        c->scr = &b->shared;
        DevBuf *persistent[] = { &c->pts, &c->nrm, &c->dir, &c->quad, &c->quad_big, &c->g_dev, &c->sx0, &c->sy0, &c->sz0, &c->wx, &c->wy,
                                 &c->wz, &c->st_tp, &c->st_nv, &c->st_lb, &c->tile_flags, &c->tile_box, &c->perm, &c->partials,
                                 &c->totals, &c->deltas, &c->state, &c->xhist, &c->ihist, &c->p2p_sums,
                                 &c->rthist, &c->fx_exp, &c->rt_apart };
        for (DevBuf *d : persistent) d->arena = &b->arena;
        c->h_state = &b->host_states[i];
        c->h_state_pinned = false;
        b->ctxs.push_back(c);
        if (!(c->opt.dist_threshold > 0.f) || c->opt.normals_k < 3 || c->opt.normals_k > 128) {
            rc = fail(c, SVICP_ERR_ARG, "svicp_batch_create_from_arrays: bad options");
            break;
        }
        const int64_t m = tgt_offsets[i + 1] - tgt_offsets[i], n = src_offsets[i + 1] - src_offsets[i];
        rc = svicp_set_target(c, tgt_xyz + 3 * tgt_offsets[i], m,
                              tgt_normals ? tgt_normals + 3 * tgt_offsets[i] : nullptr);
        if (rc == SVICP_OK) rc = svicp_set_source(c, src_xyz + 3 * src_offsets[i], n);
        if (rc == SVICP_OK) rc = ensure_iteration_buffers(c, 64);
        if (rc != SVICP_OK) g_create_error = "pair " + std::to_string(i) + ": " + c->err;
    }
    if (rc == SVICP_OK && hipStreamSynchronize(b->own_stream) != hipSuccess)
        rc = fail(nullptr, SVICP_ERR_HIP, "svicp_batch_create_from_arrays: hipStreamSynchronize");
    if (rc != SVICP_OK) {
        b->lead = b->ctxs.empty() ? nullptr : b->ctxs[0];
        const std::string keep = g_create_error;
        (void)svicp_batch_destroy(b);
        g_create_error = keep;
        return rc;
    }
    b->setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    b->lead = b->ctxs[0];
    *out = b;
    return SVICP_OK;
}

int svicp_batch_info(const svicp_batch *b, int *count, double *setup_ms, double *device_bytes)
{
    if (!b) return SVICP_ERR_ARG;
    if (count) *count = (int)b->ctxs.size();
    if (setup_ms) *setup_ms = b->setup_ms;
    if (device_bytes) *device_bytes = (double)b->arena.total;
    return SVICP_OK;
}

int svicp_batch_pair(svicp_batch *b, int index, svicp_ctx **ctx_out)
{
    if (!b || !ctx_out || index < 0 || index >= (int)b->ctxs.size()) return SVICP_ERR_ARG;
    *ctx_out = b->ctxs[index];
    return SVICP_OK;
}

const char *svicp_batch_last_error(const svicp_batch *b)
{
    return b ? b->lead->err.c_str() : g_create_error.c_str();
}

int svicp_batch_run(svicp_batch *b, int iterations, double *T_out, svicp_info *info_out)
{
    if (!b) return SVICP_ERR_ARG;
    svicp_ctx *ctx = b->lead;
    if (iterations < 1 || !T_out) return fail(ctx, SVICP_ERR_ARG, "svicp_batch_run: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    const int count = (int)b->ctxs.size();

    // A batch built from arrays owns its pairs and nobody can give them other clouds or
    // buffers: once its tables are on the device and its buffers hold `iters_cap` iterations
    // a run has nothing to prepare (512 pairs: 0.3 - 3 ms of host time per run otherwise).
    const bool coherent = ctx->opt.coherent != 0;
    const bool prepared = b->owns && b->tables_ready && iterations <= b->iters_cap;
    if (prepared && (b->g_iters != iterations || b->g_coherent != coherent)) {
        if (b->gexec) { (void)hipGraphExecDestroy(b->gexec); b->gexec = nullptr; }
        b->g_iters = -1;
    }
    if (!prepared) {
    // (re)build the device tables when any pair's buffers changed
    std::vector<BatchPair> hp(count);
    std::vector<int2> map;
    for (int i = 0; i < count; ++i) {
        svicp_ctx *c = b->ctxs[i];
        if (!c->have_target || !c->have_source)
            return fail(ctx, SVICP_ERR_ARG, "svicp_batch_run: a pair lost its target or source");
        int rc = ensure_iteration_buffers(c, iterations);
        if (rc) { ctx->err = c->err; return rc; }
        // the pair's set-up work (and any clearing of fresh buffers just queued) ran on its
        // own stream: it must be complete before the batch's stream touches the buffers
        if (c->stream != ctx->stream) HIPCHK(hipStreamSynchronize(c->stream));
        const float thr = c->opt.dist_threshold;
        BatchPair &P = hp[i];
        P.g = c->g;
        P.x0 = c->sx0.as<float>(); P.y0 = c->sy0.as<float>(); P.z0 = c->sz0.as<float>();
        P.wx = c->wx.as<float>(); P.wy = c->wy.as<float>(); P.wz = c->wz.as<float>();
        P.n = (int)c->n; P.nb = c->nb;
        P.totals = c->totals.as<fx_t>(); P.deltas = c->deltas.as<fx_t>();
        P.fx = c->fx_exp.as<int>();
        P.st = c->state.as<IcpState>();
        P.X_hist = c->xhist.as<double>(); P.info_hist = c->ihist.as<double>();
        P.cp.thr = thr;
        P.cp.root_hi = std::sqrt(thr) * 1.00001f;
        P.cp.cov = std::sqrt(thr) * 1.001f;
        P.ps.tp = c->st_tp.as<Float3>(); P.ps.nv = c->st_nv.as<Float3>(); P.ps.lb = c->st_lb.as<float>();
        P.ps.active = nullptr;   // the batch has its own block table
        P.partials = c->partials.as<double>();
        for (int t = 0; t < c->nb; ++t) map.push_back(make_int2(i, t));
    }
    void *pb = b->pairs.p, *mb = b->block_map.p;
    HIPCHK(b->pairs.reserve(sizeof(BatchPair) * count));
    HIPCHK(b->block_map.reserve(sizeof(int2) * map.size()));
    HIPCHK(b->states.reserve(sizeof(IcpState) * count));
    void *rb = b->rt_all.p;
    HIPCHK(b->rt_all.reserve(sizeof(double) * 12 * count));
    if (!b->h_states) HIPCHK(hipHostMalloc((void **)&b->h_states, sizeof(IcpState) * count, 0));
    // the tables go to the device only when they differ from what is there (a run repeated
    // on the same pairs finds them unchanged)
    const unsigned char *hb = reinterpret_cast<const unsigned char *>(hp.data());
    const unsigned char *mbytes = reinterpret_cast<const unsigned char *>(map.data());
    const size_t hsize = sizeof(BatchPair) * count, msize = sizeof(int2) * map.size();
    const bool same = pb == b->pairs.p && mb == b->block_map.p && b->sent_pairs.size() == hsize &&
                      b->sent_map.size() == msize && std::memcmp(b->sent_pairs.data(), hb, hsize) == 0 &&
                      std::memcmp(b->sent_map.data(), mbytes, msize) == 0;
    if (!same) {
        HIPCHK(hipMemcpyAsync(b->pairs.p, hp.data(), hsize, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(b->block_map.p, map.data(), msize, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));   // hp and map go out of scope
        b->sent_pairs.assign(hb, hb + hsize);
        b->sent_map.assign(mbytes, mbytes + msize);
    }
    const int total_now = (int)map.size();
    if (pb != b->pairs.p || mb != b->block_map.p || rb != b->rt_all.p || total_now != b->total_blocks ||
        b->g_iters != iterations || b->g_coherent != coherent) {
        if (b->gexec) { (void)hipGraphExecDestroy(b->gexec); b->gexec = nullptr; }
        b->g_iters = -1;
    }
    b->total_blocks = total_now;
    b->tables_ready = true;
    b->iters_cap = b->ctxs[0]->hist_cap;
    for (svicp_ctx *c : b->ctxs) b->iters_cap = std::min(b->iters_cap, c->hist_cap);
    }
    const int total = b->total_blocks;

    auto enqueue = [&]() -> int {
        const BatchPair *dp = b->pairs.as<BatchPair>();
        const int2 *dm = b->block_map.as<int2>();
        double *rt = b->rt_all.as<double>();
        hipLaunchKernelGGL(batch_reset_kernel, dim3(count), dim3(64), 0, ctx->stream, dp, iterations);
        LAUNCHCHK("batch_reset_kernel");
        for (int k = 0; k < iterations; ++k) {
            if (k > 0) {
                hipLaunchKernelGGL(batch_solve_kernel, dim3(count), dim3(64), 0, ctx->stream, dp, k - 1, rt);
                LAUNCHCHK("batch_solve_kernel");
            }
            if (k == 0)
                hipLaunchKernelGGL((icp_iterate_batch_kernel<false, false, 4>), dim3(total), dim3(kItThreads), 0, ctx->stream, dp, dm, k, rt);
            else if (coherent)
                hipLaunchKernelGGL((icp_iterate_batch_kernel<true, true, 4>), dim3(total), dim3(kItThreads), 0, ctx->stream, dp, dm, k, rt);
            else
                hipLaunchKernelGGL((icp_iterate_batch_kernel<true, false, 4>), dim3(total), dim3(kItThreads), 0, ctx->stream, dp, dm, k, rt);
            LAUNCHCHK("icp_iterate_batch_kernel");
        }
        hipLaunchKernelGGL(batch_solve_kernel, dim3(count), dim3(64), 0, ctx->stream, dp, iterations - 1, rt);
        LAUNCHCHK("batch_solve_kernel");
        const long long words = (long long)count * (sizeof(IcpState) / sizeof(double));
        hipLaunchKernelGGL(batch_collect_kernel, dim3(cdiv(words, 256)), dim3(256), 0, ctx->stream, dp, count, b->states.as<IcpState>());
        LAUNCHCHK("batch_collect_kernel");
        return SVICP_OK;
    };

    ScopedEvent e0, e1;
    HIPCHK(e0.create());
    HIPCHK(e1.create());
    HIPCHK(hipEventRecord(e0.ev, ctx->stream));
    if (ctx->opt.use_graph && ctx->stream != nullptr) {
        if (!b->gexec) {
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            int rc = enqueue();
            hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            if (e != hipSuccess) return fail(ctx, SVICP_ERR_HIP, "hipStreamEndCapture", e);
            e = hipGraphInstantiate(&b->gexec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess) { b->gexec = nullptr; return fail(ctx, SVICP_ERR_HIP, "hipGraphInstantiate", e); }
            b->g_iters = iterations;
            b->g_coherent = coherent;
        }
        HIPCHK(hipGraphLaunch(b->gexec, ctx->stream));
    } else {
        int rc = enqueue();
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(e1.ev, ctx->stream));
    HIPCHK(hipMemcpyAsync(b->h_states, b->states.p, sizeof(IcpState) * count, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0.ev, e1.ev));
    b->last_ms = ms;

    int worst = SVICP_OK;
    for (int i = 0; i < count; ++i) {
        svicp_ctx *c = b->ctxs[i];
        *c->h_state = b->h_states[i];
        c->cur = iterations > 1 ? 1 : 0;
        c->last_iters = iterations;
        c->last_was_p2p = false;
        std::memcpy(T_out + 16 * (size_t)i, c->h_state->Trans, sizeof(double) * 16);
        if (info_out) fill_info(c, iterations, info_out + i);
        if (c->h_state->status == SVICP_ERR_SINGULAR) worst = SVICP_ERR_SINGULAR;
    }
    if (worst != SVICP_OK)
        return fail(ctx, worst, "svicp_batch_run: a pair's normal matrix was not positive definite (see info[i].status)");
    return SVICP_OK;
}

int svicp_batch_last_ms(const svicp_batch *b, double *ms)
{
    if (!b || !ms) return SVICP_ERR_ARG;
    *ms = b->last_ms;
    return SVICP_OK;
}

}  // extern "C"
