/*
 * svcal.h - C ABI (in libsvicp.so) of the calibration reprojection path:
 * residuals and the central-difference Jacobian of the reference's stereo and
 * monocular bundle refinement, reduced on the device to the block-sparse normal
 * equations an LM step needs.  Status codes are those of svicp.h.
 *
 * Replaces the evaluation lambdas inside
 *   CStereoCameraCalibration::doStereoCalibrate
 *       value    Calibration/CStereoCameraCalibration.cpp:25-50
 *       residual Calibration/CStereoCameraCalibration.cpp:51-80
 *       jacobian Calibration/CStereoCameraCalibration.cpp:81-247 (+ J^T J, J^T e of :274-276)
 *   CMonocularCameraCalibration::refineAllPara
 *       value2 / residual2 / jacobian2  Calibration/CMonocularCameraCalibration.cpp:737-906
 *   CStereoCameraRectification::doImageRectify
 *       the per-pixel map loop           Calibration/CStereoCameraRectification.cpp:60-143
 * The reference forms the dense Jacobian (rows x (8 + 2k + 6V + 6)); here only its
 * non-zero blocks are evaluated and J^T J is returned block by block.
 *
 * Parameter vector, as composeStereoPara lays it out
 * (CStereoCameraCalibration.cpp:478-518), nI = 4 + k, k = 2 or 5:
 *   [fxL fyL cxL cyL distL(k) | fxR fyR cxR cyR distR(k) | V x (rvecL, tL) | rvecS tS]
 * Monocular (composeCameraPara with zero skew and five distortion terms,
 * CMonocularCameraCalibration.cpp:1045-1082):
 *   [fx fy cx cy dist(5) | V x (rvec, t)]
 */
#ifndef SVCAL_H
#define SVCAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svcal_ctx svcal_ctx;

/* Caller-provided output blocks (host memory, row-major, symmetric blocks
 * full).  Any pointer may be NULL to skip that block.
 *   LL nI x nI, RR nI x nI, RS nI x 6, SS 6 x 6   (R*, *S unused for monocular)
 *   LE V x nI x 6, EE V x 6 x 6                    per view
 *   gL nI, gR nI, gE V x 6, gS 6                   J^T e
 *   residual: 4 V C (stereo: left rows then right rows, projected - observed,
 *             CStereoCameraCalibration.cpp:59-78) or 2 V C (monocular: observed -
 *             projected, CMonocularCameraCalibration.cpp:821-822)               */
typedef struct svcal_blocks {
    double *LL, *RR, *RS, *SS, *LE, *EE;
    double *gL, *gR, *gE, *gS;
    double *residual;
    double sq_residual;     /* out: |e|^2 */
} svcal_blocks;

int svcal_create(int device_id, svcal_ctx **out);
int svcal_destroy(svcal_ctx *ctx);
const char *svcal_last_error(const svcal_ctx *ctx);

/* The board and the detected corners: world_xy C x 2 (z = 0 as in the
 * reference), left_xy / right_xy V x C x 2 pixels.  right_xy NULL: monocular. */
int svcal_set_observations(svcal_ctx *ctx, const double *world_xy, int corners,
                           const double *left_xy, const double *right_xy, int views);

/* Uploads a parameter vector (length 2 nI + 6 V + 6, or 9 + 6 V monocular). */
int svcal_set_params(svcal_ctx *ctx, const double *params, int k_size);

/* Evaluates residuals and the block-sparse normal equations at the uploaded
 * parameters, leaving the results on the device (what an iteration on the
 * device, or a throughput measurement, needs). */
int svcal_eval(svcal_ctx *ctx);

/* Copies the results of the last svcal_eval into the caller's blocks. */
int svcal_fetch(svcal_ctx *ctx, svcal_blocks *out);

/* The damped step of the refinement loop from the blocks of the last svcal_eval, on the
 * device: step = -(A + lam diag(A))^-1 G (CStereoCameraCalibration.cpp:288-291, where A is
 * inverted densely), with the per-view 6 x 6 blocks eliminated first.  step_out: the
 * parameter vector's length and order; predicted_out (may be NULL): step^T (lam step - G), the
 * denominator of the gain ratio (:312-313).  SVICP_ERR_SINGULAR when a block is not positive
 * definite.  Not for a context whose sums were replaced by svcal_shard_import (the views of
 * the other ranks are not here: svcal_solve_begin / svcal_solve_end below). */
int svcal_solve_damped(svcal_ctx *ctx, double lam, double *step_out, double *predicted_out);

/* The `update_intrinsics` argument of doStereoCalibrate (CStereoCameraCalibration.h:24-28).
 * 1 (the default; what Test/demo.cpp:198 passes): every parameter is refined.  0: the
 * Jacobian the reference builds holds only the views' and the pose's columns
 * (CStereoCameraCalibration.cpp:187-245, applied at offset 8 + 2 k, :284-285, :306): the
 * steps of svcal_solve_damped and svcal_solve_end then leave the 2 (4 + k) intrinsic and
 * distortion parameters where they are (their entries of step_out are exactly 0) and solve
 * each view's 6 x 6 block and the pose's 6 x 6 block on their own (the blocks that couple
 * them with the intrinsics drop out of A).  The evaluation itself does not change. */
int svcal_set_update_intrinsics(svcal_ctx *ctx, int update_intrinsics);

/* |e|^2 of the last svcal_eval alone (two doubles come back instead of every block). */
int svcal_sq_residual(svcal_ctx *ctx, double *out);

/* The refinement loop itself (CStereoCameraCalibration.cpp:278-335, the same in
 * CMonocularCameraCalibration::refineAllPara): lambda starts at 1e-3; per iteration the damped
 * step (svcal_solve_damped), one evaluation at the trial parameters, the gain ratio
 * r = (|e|^2 - |e'|^2) / (step^T (lambda step - G)) (:312-313); r > 0: the step is taken and
 * lambda *= max(1/3, 1 - (2 r - 1)^3) (:325), else lambda *= v, v *= 2 (:330-331); stops when
 * |step| < 1e-5 (:293), |e| < 1e-3 (:323) or after max_iterations (100 in the reference, :282).
 * params (the parameter vector as svcal_set_params takes it) is refined in place;
 * iterations_out / rms_out (either may be NULL): iterations performed and the final |e|.
 * Only the step and |e|^2 cross to the host per iteration; the blocks stay on the device. */
int svcal_lm_refine(svcal_ctx *ctx, double *params, int k_size, int max_iterations,
                    int *iterations_out, double *rms_out);

/* Runs the context's work on the caller's HIP stream (use_own = 0; NULL is the
 * default stream) or back on the context's own stream (use_own = 1). */
int svcal_set_stream(svcal_ctx *ctx, void *hip_stream, int use_own);

/* Views split over several contexts (one per GPU; no counterpart in the reference):
 * every context holds the full shared parameters and its own views' extrinsics.
 * After svcal_eval, svcal_shard_export copies the 272 sums over this context's
 * views (the 16 x 16 upper triangles of [J | e]^T [J | e] for the left and the
 * right camera) into a caller's device buffer; the caller adds the buffers of all
 * contexts (an all-reduce) and hands the total back with svcal_shard_import, after
 * which svcal_fetch returns the shared blocks (LL, RR, RS, SS, gL, gR, gS, |e|^2)
 * of the whole problem and the per-view blocks (LE, EE, gE) of this context's views. */
int svcal_shard_export(svcal_ctx *ctx, double *dev_sums272);
int svcal_shard_import(svcal_ctx *ctx, const double *dev_sums272);

/* The damped step with the views split over several contexts (no counterpart in the
 * reference; after svcal_shard_import, so every context holds the sums of the whole
 * problem).  svcal_solve_begin eliminates this context's views and copies their summed
 * contribution to the shared system (SVCAL_SOLVE_SUMS doubles) into a caller's device
 * buffer; the caller adds the buffers of all contexts (an all-reduce) and hands the total to
 * svcal_solve_end, which solves the shared system (every context the same one) and this
 * context's views.  step_out: this context's parameter vector (shared parameters, its own
 * views' six each, the stereo pose); predicted_out[0]: the shared parameters' part of
 * step^T (lam step - G), the same in every context; predicted_out[1]: the part of this
 * context's views (the caller adds [1] over the contexts and [0] once).  With one context,
 * begin followed by end is svcal_solve_damped. */
#define SVCAL_SOLVE_SUMS 90
int svcal_solve_begin(svcal_ctx *ctx, double lam, double *dev_sums90);
int svcal_solve_end(svcal_ctx *ctx, double lam, const double *dev_sums90, double *step_out,
                    double predicted_out[2]);

/* CStereoCameraRectification::doStereoRectify (Calibration/CStereoCameraRectification.cpp:13-58):
 * from the two intrinsic matrices and the pose of the right camera (R, T: 3 x 3 row-major
 * and 3), the rectifying projections of both cameras (3 x 3, row-major: new intrinsics times
 * the rectifying rotation) and the 4 x 4 reprojection matrix Q.  Half of the relative
 * rotation goes to each camera (the rotation vector of R^-1, halved, Common/Misc.hpp:7-75; the
 * nearest rotation to the given matrix is taken first, as the reference's SVD does), the new
 * x axis is the baseline, the new focal length the mean of the four given ones.  Host
 * arithmetic only, no context.  SVICP_ERR_SINGULAR when R cannot be inverted or the baseline
 * has no x / y component to build the frame from. */
int svcal_stereo_rectify(const double KL[9], const double KR[9], const double R[9], const double T[3],
                         double left_proj[9], double right_proj[9], double Q[16]);

/* CStereoCameraRectification::doImageRectify
 * (Calibration/CStereoCameraRectification.cpp:60-143): for every pixel (i, j) of the
 * rectified image, the position in the original left / right image to sample:
 * inverse(proj) * (i, j, 1), normalise, distort (k_size 2 or 5 terms, :86-97), apply the
 * 3x3 intrinsic matrix K (row-major, K[1] is the skew the reference multiplies in at
 * :98); positions outside [0, width-1] x [0, height-1] become INVALID_MAP = 9.99e+04
 * (Common/Utils.hpp:16).  Outputs: four host arrays of width * height doubles, row-major,
 * as the reference allocates them (:70-73).  Needs a context only for its device and
 * stream (no observations).  kernel_ms (may be NULL): device time of the map kernel.
 * SVICP_ERR_ARG on the argument errors of :68-69, SVICP_ERR_SINGULAR for a singular
 * projection. */
int svcal_rectify_maps(svcal_ctx *ctx, int width, int height, const double KL[9],
                       const double *distL, const double KR[9], const double *distR, int k_size,
                       const double projL[9], const double projR[9], double *xmap_left,
                       double *ymap_left, double *xmap_right, double *ymap_right,
                       double *kernel_ms);

/* Average duration (ms) of `reps` back-to-back svcal_eval launches. */
int svcal_time_eval(svcal_ctx *ctx, int reps, double *avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* SVCAL_H */
