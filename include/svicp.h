/*
 * svicp.h - C ABI of libsvicp.so: point-to-plane ICP registration and exact
 * k-nearest-neighbour search on one AMD MI355X (gfx950) per context.
 *
 * Every entry point returns an int status (SVICP_OK == 0).  No exception, C++
 * type or framework type crosses this boundary: plain pointers and sizes only.
 * Unless a parameter says "device", pointers are host pointers owned by the
 * caller; the library copies what it needs during the call.
 *
 * One context is used by one host thread at a time.  Contexts are independent.
 *
 * The entry points replace the public surface of the reference's registration
 * classes (paths relative to the reference repository):
 *   PointCloud/Registration/CPoint2PlaneICP.h:11-17
 *   PointCloud/Registration/CKNNSearch.h:12-17
 * Each declaration cites the member it stands in for.
 */
#ifndef SVICP_H
#define SVICP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svicp_ctx svicp_ctx;

enum {
    SVICP_OK = 0,
    SVICP_ERR_ARG = 1,       /* null pointer, negative size, wrong call order      */
    SVICP_ERR_HIP = 2,       /* a HIP runtime call failed (see svicp_last_error)   */
    SVICP_ERR_GRID = 3,      /* target extent / cell edge too large for the grid   */
    SVICP_ERR_SINGULAR = 4,  /* the 6x6 normal matrix was not positive definite    */
    SVICP_ERR_NO_DEVICE = 5, /* no gfx950 device / library built without a device  */
    SVICP_ERR_COMM = 6       /* RCCL missing or a collective call failed            */
};

/* Tunables.  Defaults are the reference's hard-coded constants. */
typedef struct svicp_options {
    float dist_threshold;  /* squared-distance cut-off of the weight function, 5.0
                              (CPoint2PlaneICP.cpp:68)                                */
    int normals_k;         /* neighbours per normal estimate, 20 (CPoint2PlaneICP.cpp:125) */
    int use_graph;         /* 1: svicp_run replays a captured HIP graph of the whole
                              iteration loop; 0: eager launches                       */
    int time_kernels;      /* 1: eager launches bracketed by HIP events so that
                              svicp_kernel_times reports per-kernel durations         */
    int sort_source;       /* 1: keep the source in Morton order on the device (the
                              caller-visible order is unchanged)                      */
    int coherent;          /* 1: from the second iteration on, a point whose previous
                              correspondence is provably still its nearest neighbour
                              skips the search (same result, bit for bit); 0: search
                              every point in every iteration                          */
    int centred_normals;   /* 0 (the reference): second moments about the ORIGIN, each
                              term rounded to fp32 and divided by k in fp32
                              (CPoint2PlaneICP.cpp:140-146); 1: the covariance about the
                              neighbourhood's centroid, in fp64 -- the textbook estimate,
                              which the reference's is not once a cloud is far from the
                              origin                                                   */
} svicp_options;

/* Per-run summary.  Arrays hold the LAST iteration's figures. */
typedef struct svicp_info {
    int iterations;        /* solves performed (== the requested count)              */
    int status;            /* SVICP_OK or SVICP_ERR_SINGULAR                          */
    double inliers;        /* correspondences with non-zero weight                    */
    double weighted_sq_residual; /* sum w * b^2 over correspondences                  */
    double step_norm;      /* |X| of the last 6-vector                                */
    double weight_sum;     /* sum w                                                   */
} svicp_info;

/* ---- lifetime ------------------------------------------------------- */

/* CPoint2PlaneICP::CPoint2PlaneICP() / CKNNSearch::CKNNSearch()
 * (CPoint2PlaneICP.cpp:3-6, CKNNSearch.cpp:3-8).  device_id is the HIP device
 * ordinal the context lives on. */
int svicp_create(int device_id, svicp_ctx **out);

/* ~CPoint2PlaneICP / ~CKNNSearch (CPoint2PlaneICP.cpp:7-10, CKNNSearch.cpp:10-25). */
int svicp_destroy(svicp_ctx *ctx);

/* Text of the last failure on this context (or of the last failed
 * svicp_create when ctx is NULL).  Never NULL. */
const char *svicp_last_error(const svicp_ctx *ctx);

int svicp_default_options(svicp_options *opt);
int svicp_set_options(svicp_ctx *ctx, const svicp_options *opt);
int svicp_get_options(const svicp_ctx *ctx, svicp_options *opt);

/* ---- target ("reference data") -------------------------------------- */

/* CPoint2PlaneICP::setReferenceData (CPoint2PlaneICP.cpp:11-18) and
 * CKNNSearch::setTreeData (CKNNSearch.cpp:27-35).
 * xyz_aos: m x 3 fp32 row-major (the flann::Matrix<float> layout).
 * normals_aos: m x 3 fp32 unit normals, or NULL to estimate them the way
 * CPoint2PlaneICP::calcNormals does (CPoint2PlaneICP.cpp:122-154).
 * Builds the search grid on the device. */
int svicp_set_target(svicp_ctx *ctx, const float *xyz_aos, int64_t m, const float *normals_aos);

/* The normals in use (m x 3 fp32, caller order): the ones passed in, or the
 * estimated ones (m_pNormals, CPoint2PlaneICP.h:26). */
int svicp_get_normals(svicp_ctx *ctx, float *normals_aos_out);

/* ---- source ("move data") ------------------------------------------- */

/* CPoint2PlaneICP::setMoveData (CPoint2PlaneICP.cpp:19-22) and
 * CKNNSearch::setQueryData (CKNNSearch.cpp:37-54).  n x 3 fp32 row-major.
 * The library keeps its own device copy; the caller's buffer is not retained. */
int svicp_set_source(svicp_ctx *ctx, const float *xyz_aos, int64_t n);

/* The source cloud as the last svicp_run left it: transformed iterations-1
 * times, the state the reference leaves in the caller's buffer
 * (CPoint2PlaneICP.cpp:42-50).  n x 3 fp32, caller order. */
int svicp_get_source(svicp_ctx *ctx, float *xyz_aos_out);

/* Pins a caller's host buffer in place (page-locks it and maps it for the device) so
 * that svicp_set_source, svicp_get_source, svicp_set_target and svicp_knn copy it at the
 * speed of the link instead of through the runtime's staging of pageable memory: worth it
 * for a buffer that is handed over more than once (the move data of CPoint2PlaneICP, which
 * run() reads and writes back on every call, CPoint2PlaneICP.cpp:42-50).  Registration
 * itself costs about as much as one pageable copy.  Unregister before the buffer is freed.
 * Not every system lets a process pin memory it did not allocate through the runtime
 * (SVICP_ERR_HIP then, and the copies take the pageable path as before); memory from
 * svicp_host_alloc is pinned from the start and always works: the drop-in CTxtReader
 * allocates its clouds with it.  Errors: svicp_last_error(NULL). */
int svicp_host_register(void *host_ptr, int64_t bytes);
int svicp_host_alloc(int64_t bytes, void **out);
int svicp_host_free(void *host_ptr);
int svicp_host_unregister(void *host_ptr);

/* ---- registration ---------------------------------------------------- */

/* CPoint2PlaneICP::setMaxIterateCount + run (CPoint2PlaneICP.cpp:27-62).
 * Always runs exactly `iterations` solves from the source as svicp_set_source
 * gave it (the reference has no convergence test: m_eps is never read).
 * T_out: accumulated 4x4 rigid transform, row-major fp64 (the matrix the
 * reference prints at CPoint2PlaneICP.cpp:58).  info may be NULL.
 * All iterations run on the device without a host round trip. */
int svicp_run(svicp_ctx *ctx, int iterations, double T_out[16], svicp_info *info);

/* svicp_run for callers that cannot allocate device memory or wait on the device
 * (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  It has no counterpart in the reference.  The same kernels as
 * svicp_run, launched one by one on the context's stream; nothing is allocated,
 * freed, instantiated or synchronised: the sums, the iteration state and the
 * history live in a file-scope device array (about 9.7 MB per GPU) cut into 8
 * workspaces of 1.2 MB.  A context takes a free workspace at its first call and
 * keeps it until svicp_destroy, so contexts that enqueue on their own streams at
 * the same time do not touch each other's sums; a ninth context on one device is
 * refused (SVICP_ERR_ARG) while eight others hold theirs.  A workspace keeps the
 * last call's contents (what must start at zero is cleared by the call).  What
 * still races: the SAME context's calls (or a recorded graph of them) running at
 * the same time on different streams; a recording replayed after its context was
 * destroyed and the workspace handed to another.
 * Limits, rejected with SVICP_ERR_ARG: iterations <= 128, source <= 4194304 points.
 * The clouds are the resident ones (svicp_set_target / svicp_set_source allocate;
 * call them beforehand).  dev_out20: DEVICE memory, 20 doubles, written by the last
 * kernel: the accumulated 4x4 transform row-major, then {status (SVICP_OK or
 * SVICP_ERR_SINGULAR), inliers, sum w b^2, |X|} of the last iteration.  The call
 * returns once the work is queued; svicp_get_history afterwards reports "no run".
 * Under a stream capture the call adds kernel nodes only; the recording holds the
 * source's tile count and the resident buffers' addresses, so record again after a
 * source of another size. */
int svicp_run_enqueue(svicp_ctx *ctx, int iterations, double *dev_out20);

/* Iterations of the last successful run on this context (svicp_run,
 * svicp_run_point2point, svicp_batch_run, svicp_shard_update): the number of rows
 * the history calls below copy.  0 before any run. */
int svicp_last_iterations(const svicp_ctx *ctx, int *iterations);

/* Per-iteration record of the last svicp_run: X_out iterations x 6 (the
 * solution vector X of CPoint2PlaneICP.cpp:109), info_out iterations x 4
 * {inliers, sum w b^2, |X|, sum w}.  Either may be NULL.  capacity_rows: rows the
 * caller's arrays hold; SVICP_ERR_ARG when fewer than svicp_last_iterations. */
int svicp_get_history(svicp_ctx *ctx, int capacity_rows, double *X_out, double *info_out);

/* 1-NN correspondences of the current source against the target, as used by
 * the last solve (CPoint2PlaneICP.cpp:65-67).  idx_out, d2_out: n entries,
 * caller order.  Only neighbours within dist_threshold matter to the solve;
 * a point with none closer gets idx -1 and d2 +inf. */
int svicp_correspondences(svicp_ctx *ctx, int32_t *idx_out, float *d2_out);

/* ---- exact k-NN ------------------------------------------------------ */

/* CKNNSearch::setQueryData + setKCount + getKNeighbour (CKNNSearch.cpp:37-71).
 * Exact k nearest target points of each query, k clamped to 128 as
 * setKCount does (CKNNSearch.cpp:60-65).  idx_out: n x k int32 row-major,
 * d2_out: n x k SQUARED distances fp32, ascending; equal distances are ordered
 * by ascending index.  Missing neighbours (k > m): idx -1, d2 +inf.  A query with a NaN
 * coordinate (the reference leaves it to FLANN, undefined): k = 1 answers "no neighbour"
 * (-1, +inf); k > 1 answers NaN distances and indices that mean nothing. */
int svicp_knn(svicp_ctx *ctx, const float *query_aos, int64_t n, int k,
              int32_t *idx_out, float *d2_out);

/* ---- sharded registration (one context per GPU) ---------------------- */
/* A large registration shards the source over ranks; every rank holds the
 * whole target.  The library keeps, per rank, running totals of the 32 sums
 * of the normal equations (21 upper-triangle JtJ entries, 6 Jtr entries,
 * {inliers, sum w b^2, sum w}, padding) and per iteration only their CHANGE:
 * points whose correspondence and position did not change contribute zero.
 * Per iteration each rank calls svicp_shard_accumulate, the caller adds the
 * 32-double change blocks of all ranks (RCCL all-reduce), and each rank calls
 * svicp_shard_update with the total.  The reference has no multi-device path;
 * these split CPoint2PlaneICP::singleRun (CPoint2PlaneICP.cpp:63-120) at the
 * one point where ranks exchange data. */

/* Launches on `hip_stream` (a hipStream_t, e.g. the caller framework's
 * current stream) from now on.  NULL is the device's default stream, not
 * "no change"; svicp_use_own_stream goes back to the context's private
 * non-blocking stream (the initial state). */
int svicp_set_stream(svicp_ctx *ctx, void *hip_stream);
int svicp_use_own_stream(svicp_ctx *ctx);

/* Waits until everything queued on the context's stream has completed (what a
 * caller of svicp_run_enqueue does before it reads dev_out20 from the host). */
int svicp_synchronize(svicp_ctx *ctx);

/* The whole sharded registration in one call, the collective included.  Every rank
 * (one process and one context per GPU) holds the whole target and its part of the
 * source and calls svicp_shard_run with the same iteration count.  Per iteration: one
 * launch of the iteration kernel, then one RCCL all-reduce (sum, 128 64-bit integers: the
 * rank's change of the 32 fixed-point sums in the four copies the kernel keeps) on the context's
 * stream, in place; the next launch solves from the summed numbers on every rank
 * alike, so nothing is broadcast.  With options.use_graph the loop is captured in one
 * HIP graph (collectives included) and replayed; a runtime that refuses to capture the
 * collective gets the same launches one by one.  No host round trip per iteration
 * either way.  The reference is single-process (test.cpp:14-29): nothing to cite.
 *
 * svicp_comm_unique_id: rank 0 calls it and sends the 128 bytes to the other ranks by
 * any means (MPI, torch.distributed, a file).  svicp_comm_init: every rank, with the
 * same bytes; collective (returns when all ranks have called).  RCCL is looked up at run
 * time: the copy already loaded in the process (PyTorch's) or librccl.so.1.
 * A context without a communicator runs svicp_shard_run alone (world size 1). */
int svicp_comm_unique_id(unsigned char id_out[128]);
int svicp_comm_init(svicp_ctx *ctx, const unsigned char id[128], int rank, int world_size);
int svicp_comm_destroy(svicp_ctx *ctx);
int svicp_shard_run(svicp_ctx *ctx, int iterations, double T_out[16], svicp_info *info);
/* captured: 1 when the last svicp_shard_run replayed a captured graph; world_size of
 * the communicator (1 without one).  Either pointer may be NULL. */
int svicp_shard_run_mode(const svicp_ctx *ctx, int *captured, int *world_size);

/* The sums are kept in 64-bit fixed point (their totals then do not depend on the order in
 * which tiles or ranks are added up), scaled by bounds that include the number of source
 * points.  Ranks that exchange sums must scale alike: each rank tells its context the size
 * of the WHOLE source, the same number on every rank, after svicp_set_source and before
 * svicp_shard_begin.  svicp_shard_run needs no such call: with more than one rank EVERY call
 * of it starts with one small blocking all-reduce over its communicator (each rank's point
 * count and the total it was told, if any), whatever the ranks did before, so ranks that
 * changed their part of the source and ranks that did not still make the same collective
 * calls; the total is the largest one any rank was told, else the largest part times the
 * number of ranks.  Without either a context scales for its own part only, which is right
 * for one rank. */
int svicp_shard_set_total(svicp_ctx *ctx, int64_t total_points);

/* Resets the accumulated transform and the iteration counter. */
int svicp_shard_begin(svicp_ctx *ctx);

/* Solves the system of the previous iteration (from the totals every rank
 * holds identically), transforms this rank's source shard by it, finds
 * correspondences, and copies this rank's change of the 32 sums in this
 * iteration to device memory `dev_partial32`: 32 x 8 bytes, to be added over the
 * ranks as 64-bit INTEGERS (they are fixed-point numbers; the argument keeps its
 * double * type for the callers that hold the block as raw 8-byte words).  At most
 * 1024 iterations per svicp_shard_begin. */
int svicp_shard_accumulate(svicp_ctx *ctx, double *dev_partial32);

/* Takes the change summed over all ranks (device memory); the next
 * svicp_shard_accumulate, or svicp_shard_finish, adds it to the running totals
 * and solves. */
int svicp_shard_update(svicp_ctx *ctx, const double *dev_sum32);

/* Solves the last iteration's system and copies out the accumulated transform. */
int svicp_shard_finish(svicp_ctx *ctx, double T_out[16], svicp_info *info);

/* ---- point-to-point variant -------------------------------------------- */

/* CPoint2PointICP::run (CPoint2PointICP.cpp:50-81) on the same context: per
 * iteration 1-NN, the weight function of :94-103 with options.dist_threshold
 * as its cut-off (the reference hard-codes 15.0 at :90; set it BEFORE
 * svicp_set_target, it sizes the grid), strict "<" selection (:106), weighted
 * centroids, fp32-centred cross-covariance divided by the selected count
 * (:128-147), R = V diag(1, 1, det) U^T from its SVD and T = ref centroid -
 * R * move centroid (:149-158).  Normals are not used: pass any m x 3 array
 * as normals to svicp_set_target to skip their estimation.  The centroid sums
 * are accumulated in fp64 (the reference: fp32 in point order).
 * info: inliers = selected count, weighted_sq_residual = sum w d^2,
 * step_norm = sqrt(angle^2 + |T|^2) of the last increment.
 * Returns SVICP_ERR_SINGULAR if an iteration selected no point. */
int svicp_run_point2point(svicp_ctx *ctx, int iterations, double T_out[16], svicp_info *info);

/* Per-iteration increments of the last svicp_run_point2point: RT_out
 * iterations x 12 {R row-major, T}; info_out iterations x 4 {selected count,
 * sum w d^2, step norm, sum w}.  Either pointer may be NULL; capacity_rows as for
 * svicp_get_history. */
int svicp_get_history_rt(svicp_ctx *ctx, int capacity_rows, double *RT_out, double *info_out);

/* ---- batches of independent pairs (one launch per iteration for all) ---- */
/* Many small registrations do not fill the GPU one at a time.  A batch takes
 * contexts that already hold their target and source (all on one device) and
 * advances all of them with one kernel launch per iteration; every pair keeps
 * its own grid, state and sums, so each result equals what svicp_run on that
 * context returns.  The reference registers one pair per CPoint2PlaneICP::run
 * call (test.cpp:22-25); a caller with many pairs loops over it.  The batch
 * uses the stream and the use_graph / coherent options of ctxs[0]; contexts
 * must outlive the batch and must not be used by other threads during a run.
 * Afterwards every context is in the state svicp_run leaves it in
 * (svicp_get_source, svicp_get_history work per pair). */
typedef struct svicp_batch svicp_batch;
int svicp_batch_create(svicp_ctx **ctxs, int count, svicp_batch **out);
int svicp_batch_destroy(svicp_batch *batch);
const char *svicp_batch_last_error(const svicp_batch *batch);
/* The same batch from flat arrays, for hundreds of pairs: pair i's target is points
 * tgt_offsets[i] .. tgt_offsets[i + 1] - 1 of tgt_xyz (and of tgt_normals, or NULL to
 * estimate them), its source src_offsets[i] .. src_offsets[i + 1] - 1 of src_xyz (offsets
 * in points, count + 1 entries each).  One stream and one arena of device memory for all
 * pairs (a few large allocations instead of about 25 per pair), the set-up scratch shared.
 * The batch owns its pairs: svicp_batch_pair lends pair i's context for svicp_get_source,
 * svicp_get_history, svicp_correspondences (not svicp_knn / svicp_get_normals: the cell
 * tables of the exact search and the caller-order normals are not kept per pair; do not
 * destroy it or set clouds on it).  opt: NULL for the defaults.  On failure the text is
 * in svicp_last_error(NULL). */
int svicp_batch_create_from_arrays(int device_id, int count, const int64_t *tgt_offsets,
                                   const float *tgt_xyz, const float *tgt_normals,
                                   const int64_t *src_offsets, const float *src_xyz,
                                   const svicp_options *opt, svicp_batch **out);
int svicp_batch_pair(svicp_batch *batch, int index, svicp_ctx **ctx_out);
/* Pairs, host time of svicp_batch_create_from_arrays (ms; 0 for svicp_batch_create) and
 * bytes of device memory its arena holds.  Any pointer may be NULL. */
int svicp_batch_info(const svicp_batch *batch, int *count, double *setup_ms, double *device_bytes);
/* T_out: count x 16 doubles (row-major 4x4 per pair); info_out: count entries or NULL.
 * Returns SVICP_ERR_SINGULAR if any pair's system was singular (that pair's
 * info.status says which). */
int svicp_batch_run(svicp_batch *batch, int iterations, double *T_out, svicp_info *info_out);
/* Device time (ms, HIP events on the batch's stream) of the last svicp_batch_run's loop. */
int svicp_batch_last_ms(const svicp_batch *batch, double *ms);

/* ---- measurement ----------------------------------------------------- */

/* Over the last svicp_run made with options.time_kernels = 1: the average
 * time in milliseconds between the two HIP events recorded around each launch
 * of the iteration kernel (fused_ms), the average time between two events
 * recorded back to back with nothing between them (solve_ms: what an event
 * pair adds to every measured launch; the solve runs inside the iteration
 * kernel), and the number of launches averaged. */
int svicp_kernel_times(svicp_ctx *ctx, double *fused_ms, double *solve_ms, int *launches);

/* Number of source points that went through the neighbour search in each
 * iteration of the last svicp_run (`iterations` entries; capacity_rows as for
 * svicp_get_history); the others kept their correspondence by the proof described
 * at options.coherent. */
int svicp_search_counts(svicp_ctx *ctx, int capacity_rows, double *counts_out);

/* Sources of 4096 tiles (of 1024 points) or more keep, per tile, the box its
 * points lie in and the smallest change of a coordinate that fp32 would
 * register; a launch whose R, T cannot move any point of the box by that much
 * leaves the tile unread (its sums stand, as for a tile that was read and found
 * unmoved).  Number of tiles left unread in each iteration of the last
 * svicp_run (`iterations` entries, capacity_rows as for svicp_get_history; all
 * zero for smaller sources).  No counterpart in the reference: a statistic for
 * tests and tuning. */
int svicp_tiles_left_unread(svicp_ctx *ctx, int capacity_rows, double *counts_out);

/* on = 0: launches read every tile whatever its record says (the results are
 * the same bits either way: for tests and A/B timings).  Default on. */
int svicp_debug_use_tile_boxes(svicp_ctx *ctx, int on);

/* The same two quantities per iteration over the last timed svicp_run:
 * `iterations` entries each.  Either pointer may be NULL. */
int svicp_iteration_times(svicp_ctx *ctx, double *fused_ms_out, double *solve_ms_out);

/* Runs the standalone 1-NN kernel `reps` times over the resident source and
 * reports its average duration.  exact = 0: radius-capped search as the ICP
 * loop uses it; exact = 1: unbounded exact search. */
int svicp_time_nn(svicp_ctx *ctx, int exact, int reps, double *avg_ms);

/* Measurement only: runs, `reps` times over the resident source, a kernel that issues
 * the vector-memory instructions of the capped 1-NN search and nothing else, and
 * reports its average duration and the wave-level vector-memory instructions per run.
 * real_addresses = 1: per query the directory and probe-record look-ups and exactly the
 * candidate loads the search makes (four at a time as the search does, no distance
 * arithmetic, no result written); 12, 8 or 4: the same with candidate records of that
 * many bytes (from the packed array of normals, which is sorted like the points), to
 * see what a narrower gather would cost;
 * real_addresses = 0: per_query 16-byte loads per query from hashed places of the
 * sorted target.  The first is the floor of a per-lane gather search on this GPU for
 * this pair; the second the rate of the gather path without any locality. */
int svicp_time_gather(svicp_ctx *ctx, int real_addresses, int per_query, int reps,
                      double *avg_ms, double *wave_instructions);

/* Sizes of the resident clouds. */
int svicp_sizes(const svicp_ctx *ctx, int64_t *m, int64_t *n);

/* The target's search tables (no counterpart in the reference, whose index is FLANN's
 * kd-tree, CKNNSearch.cpp:27-35): blocks of 4 x 4 x 4 cells that have a rank, bytes of the
 * probe records (16 bytes per quarter cell of a ranked block), and how many of the records
 * are kept in full (32 bytes more each) because they do not fit the packed form -- a run of
 * more than 254 points, or rows of cells next to each other more than 65535 points apart in
 * the sorted target.  Any pointer may be NULL. */
int svicp_grid_info(const svicp_ctx *ctx, int64_t *ranked_blocks, int64_t *record_bytes,
                    int64_t *records_in_full);

#ifdef __cplusplus
}
#endif
#endif /* SVICP_H */
