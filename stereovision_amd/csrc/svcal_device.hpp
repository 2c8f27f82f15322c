// Device code of the calibration reprojection path (gfx950): per (view, corner)
// residuals and central-difference Jacobian rows, reduced per view to the Gram
// matrix of [J | e], then over views.
//
// Layout: one block per view, one wave per camera.  The first 13 lanes of each wave build
// the 13 extrinsic matrices its camera needs (base and +/- eps of each of the six
// left-extrinsic parameters; pose * left extrinsic for the base pose and +/- eps of each
// of the six pose parameters) in LDS.  Each lane then takes corners; per corner it forms
// two 16-vectors u = [J row (nI + 6 columns, zero padded to 15), e], and the wave sums
// u u^T over the view's corners on the fp64 matrix cores.  Everything is fp64, as the
// reference (Data_Type = double).
#pragma once
#include <hip/hip_runtime.h>

namespace svcal {

constexpr int kThreads = 256;      // cal_eval_kernel: four waves per view
constexpr int kGram = 136;          // upper triangle of a 16 x 16 matrix
constexpr double kEps = 1.0e-6;     // CStereoCameraCalibration.cpp:84
constexpr double kInv2Eps = 1.0 / (2.0 * kEps);

// Common/Misc.hpp:16-32
__device__ inline void rodrigues(const double r[3], double R[9])
{
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < 1.0e-6) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        return;
    }
    double a0 = r[0] / theta, a1 = r[1] / theta, a2 = r[2] / theta;
    double K[9] = { 0, -a2, a1, a2, 0, -a0, -a1, a0, 0 };
    double sn, cs;
    sincos(theta, &sn, &cs);        // (one range reduction for both)
    cs = 1.0 - cs;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double k2 = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) k2 += K[3 * i + k] * K[3 * k + j];
            R[3 * i + j] = ((i == j) ? 1.0 : 0.0) + K[3 * i + j] * sn + k2 * cs;
        }
}

__device__ inline void ext_from(const double p[6], double *E)
{
    double R[9];
    rodrigues(p, R);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        E[4 * r + 0] = R[3 * r + 0]; E[4 * r + 1] = R[3 * r + 1]; E[4 * r + 2] = R[3 * r + 2];
        E[4 * r + 3] = p[3 + r];
    }
}

// CStereoCameraCalibration.cpp:25-50; intr = {fx, fy, cx, cy}, skew 0.
// The projection in two stages, so that perturbations of the intrinsic parameters reuse the
// geometry: the same operations on the same inputs as value(), hence the same bits.
//
// The two quotients c0 / c2, c1 / c2 of the reference (:31-32) are formed as products with
// one reciprocal of the depth, the hardware's estimate refined twice (a relative error of a
// few 2^-53 instead of half of one): an IEEE division is about twenty issue slots on this
// GPU and the evaluation is bound by instruction issue, and thirteen geometric stages per
// corner and camera carried two of them each.
__device__ __forceinline__ double reciprocal(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    return fma(r, fma(-x, r, 1.0), r);
}

__device__ __forceinline__ void project_geom(const double *E, double wx, double wy, double &nx,
                                             double &ny, double &r)
{
    double c0 = E[0] * wx + E[1] * wy + E[2] * 0.0 + E[3] * 1.0;
    double c1 = E[4] * wx + E[5] * wy + E[6] * 0.0 + E[7] * 1.0;
    double c2 = E[8] * wx + E[9] * wy + E[10] * 0.0 + E[11] * 1.0;
    const double rc = reciprocal(c2);
    nx = c0 * rc; ny = c1 * rc;
    r = nx * nx + ny * ny;
}

template <int K>
__device__ __forceinline__ void project_pix(const double intr[4], const double dist[5], double nx,
                                            double ny, double r, double &u, double &v)
{
    double xe, ye;
    if (K == 2) {
        xe = nx * (1.0 + dist[0] * r + dist[1] * r * r);
        ye = ny * (1.0 + dist[0] * r + dist[1] * r * r);
    } else {
        xe = nx * (1.0 + dist[0] * r + dist[1] * r * r + dist[4] * r * r * r) +
             2.0 * dist[2] * nx * ny + dist[3] * (r * r + 2.0 * nx * nx);
        ye = ny * (1.0 + dist[0] * r + dist[1] * r * r + dist[4] * r * r * r) +
             dist[2] * (r * r + 2.0 * ny * ny) + 2.0 * dist[3] * nx * ny;
    }
    u = intr[0] * xe + 0.0 * ye + intr[2];
    v = intr[1] * ye + intr[3];
}

typedef double v4d __attribute__((ext_vector_type(4)));

// One block per view, four waves: camera x half of the view's corners.  The wave adds
// u u^T, u = [J row (nI + 6 columns, zero padded to 15), e], of its corners' two rows to a
// 16 x 16 Gram matrix with the fp64 matrix instruction (16 x 16 x 4: four corners per step).
// The instruction wants component i of corner k in lane 16 k + i, and that is how the work is
// dealt: a lane computes ONE component of one corner's rows, i.e. one central difference
// (CStereoCameraCalibration.cpp:101-108, :122-123: value() at the parameter plus and minus
// eps = 1e-6, quotient by 2 eps -- a product with 1 / (2 eps) here, one more rounding of
// 2^-53) or, for component 15, the residual (projected - observed).  Columns 0..nI-1 perturb
// an intrinsic or distortion parameter, nI..nI+5 project with the +/- eps variants of the
// extrinsic matrices the first lanes have put in LDS.  Every lane runs the same two
// projections on its own numbers: no staging through LDS, no arrays in registers (the
// corner-per-lane version held both rows of a corner, 64 registers, unrolled over 31
// projections: 398 registers, one wave per SIMD), and the 16 lanes of a corner share its
// loads.  params: the parameter vector; gram_view: V x 2 x 136 (upper triangles, left and
// right camera) per view; residual: 4 V C (stereo order) or 2 V C (monocular, sign flipped),
// may be null.
template <int K>
__global__ __launch_bounds__(kThreads, 4) void cal_eval_kernel(
    const double *__restrict__ params, int V, int C, int stereo,
    const double *__restrict__ world_xy, const double *__restrict__ left_xy,
    const double *__restrict__ right_xy, double *__restrict__ gram_view,
    double *__restrict__ residual)
{
    constexpr int nI = 4 + K;
    __shared__ double ext[2][13 * 12];                       // [camera][variant][3x4]
    __shared__ v4d part[2][64];                              // the second half's sums, per camera
    __shared__ double geo[kThreads / 64][4][13][3];          // [wave][corner of the step][variant]{nx, ny, r}
    const int view = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, cam = wave & 1, split = wave >> 1;
    const double *pE = params + (stereo ? 2 * nI : nI) + 6 * view;
    const double *pS = params + 2 * nI + 6 * V;

    // variant 0: base; 1 + 2q / 2 + 2q: parameter q plus / minus eps
    if (t < 13) {
        double p[6];
        for (int q = 0; q < 6; ++q) p[q] = pE[q];
        if (t > 0) p[(t - 1) >> 1] += ((t - 1) & 1) ? -kEps : kEps;
        ext_from(p, &ext[0][12 * t]);
    }
    // (the second wave forms the pose's variants meanwhile: two Rodrigues formulas one after
    // the other were a sixth of the kernel)
    const bool right_lane = stereo && wave == 1 && lane < 13;
    double S[12];
    if (right_lane) {
        double p[6];
        for (int q = 0; q < 6; ++q) p[q] = pS[q];
        if (lane > 0) p[(lane - 1) >> 1] += ((lane - 1) & 1) ? -kEps : kEps;
        ext_from(p, S);
    }
    __syncthreads();
    if (right_lane) {
        // pose * [left extrinsic; 0 0 0 1]  (CStereoCameraCalibration.cpp:573-579)
        const double *L = &ext[0][0];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                double s = 0;
                for (int kk = 0; kk < 3; ++kk) s += S[4 * r + kk] * L[4 * kk + c];
                s += S[4 * r + 3] * (c == 3 ? 1.0 : 0.0);
                ext[1][12 * lane + 4 * r + c] = s;
            }
    }
    __syncthreads();
    const bool active = cam == 0 || stereo;                  // (whole wave)

    const int comp = lane & 15, k = lane >> 4;               // component of u, corner of the step
    const bool resid = comp == 15;
    const bool pose_col = comp >= nI && comp < nI + 6;
    const bool padding = !resid && comp >= nI + 6;           // (K = 2 only)
    const int vp = pose_col ? 1 + 2 * (comp - nI) : 0, vm = pose_col ? 2 + 2 * (comp - nI) : 0;
    // the geometric stage of a step is dealt differently: lane 13 c + g (52 of the 64) projects
    // corner c of the step with variant g of the extrinsic matrix, once, and the components
    // fetch the two results they difference from LDS (dealt by component, every lane would
    // run two geometric stages: 32 per corner where 13 are different)
    const int gc = lane / 13, gv = lane - 13 * gc;           // (gc = 4: no geometric work)
    const double *Eg = &ext[cam][12 * (gc < 4 ? gv : 0)];
    double (*const mygeo)[13][3] = geo[wave];
    const double *ip = params + cam * nI;
    // the lane's parameter sets: its own parameter plus and minus eps, the others as they are
    double intr_p[4], intr_m[4], dist_p[5], dist_m[5];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double v = ip[c];
        intr_p[c] = comp == c ? v + kEps : v;
        intr_m[c] = comp == c ? v - kEps : v;
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const double v = c < K ? ip[4 + c] : 0.0;
        dist_p[c] = (c < K && comp == 4 + c) ? v + kEps : v;
        dist_m[c] = (c < K && comp == 4 + c) ? v - kEps : v;
    }

    const double *obs = cam == 0 ? left_xy : right_xy;
    // the first half of the corners (a multiple of four) to split 0, the rest to split 1
    const int first_half = ((C + 7) / 8) * 4;
    const int j_begin = split == 0 ? 0 : (first_half < C ? first_half : C);
    const int j_end = split == 0 ? (first_half < C ? first_half : C) : C;
    v4d acc = { 0.0, 0.0, 0.0, 0.0 };
    if (active) {
        for (int jb = j_begin; jb < j_end; jb += 4) {
            const int j = jb + k;
            const bool valid = j < j_end;
            const int jc = valid ? j : j_begin;
            const size_t o = 2 * ((size_t)view * C + jc);
            const double ox = obs[o], oy = obs[o + 1];
            double nx, ny, r, a, b, c, d;
            if (gc < 4) {
                const int jg = jb + gc < j_end ? jb + gc : j_begin;
                project_geom(Eg, world_xy[2 * jg], world_xy[2 * jg + 1], nx, ny, r);
                mygeo[gc][gv][0] = nx; mygeo[gc][gv][1] = ny; mygeo[gc][gv][2] = r;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            project_pix<K>(intr_p, dist_p, mygeo[k][vp][0], mygeo[k][vp][1], mygeo[k][vp][2], a, b);
            project_pix<K>(intr_m, dist_m, mygeo[k][vm][0], mygeo[k][vm][1], mygeo[k][vm][2], c, d);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            double u0 = resid ? a - ox : (a - c) * kInv2Eps;
            double u1 = resid ? b - oy : (b - d) * kInv2Eps;
            if (padding || !valid) { u0 = 0.0; u1 = 0.0; }
            if (residual && resid && valid) {
                if (stereo) {
                    residual[(size_t)cam * 2 * V * C + o] = u0;
                    residual[(size_t)cam * 2 * V * C + o + 1] = u1;
                } else {
                    residual[o] = -u0;
                    residual[o + 1] = -u1;
                }
            }
            // A[row = lane & 15][k = lane >> 4] and B[k][col = lane & 15] are the same number
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(u0, u0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(u1, u1, acc, 0, 0, 0);
        }
    }
    if (split == 1) part[cam][lane] = acc;
    __syncthreads();
    if (split == 1 || !active) return;
    {
        const v4d other = part[cam][lane];
        acc += other;
    }
    // result (row = (lane >> 4) + 4 reg, col = lane & 15); keep the upper triangle
    double *out = gram_view + ((size_t)view * 2 + cam) * kGram;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int row = (lane >> 4) + 4 * reg, col = lane & 15;
        if (row <= col) out[row * 16 - (row * (row - 1)) / 2 + (col - row)] = acc[reg];
    }
}

// Column sums of a table of `rows` rows (row stride `stride` doubles): block e adds up
// column e, thread t the rows t, t + 256, ..., then a tree over the threads, so the result is
// the same bits on every run.  Columns from `columns` on (blocks past it) are set to zero.
__device__ __forceinline__ void column_sum(const double *__restrict__ table, int rows, int stride,
                                           int columns, double *__restrict__ out)
{
    __shared__ double red[256];
    const int e = blockIdx.x;
    double s = 0.0;
    if (e < columns)
        for (int r = threadIdx.x; r < rows; r += 256) s += table[(size_t)r * stride + e];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[e] = red[0];
}

// Sums the per-view Gram matrices over the views: out[camera][136].
__global__ __launch_bounds__(256) void cal_reduce_kernel(const double *__restrict__ gram_view, int V,
                                                         double *__restrict__ gram_sum)
{
    column_sum(gram_view, V, 2 * kGram, gridDim.x, gram_sum);   // block = camera * 136 + entry
}

// doImageRectify (CStereoCameraRectification.cpp:60-143): one thread per pixel of the
// rectified image and both cameras; writes where to sample the original images, or
// INVALID_MAP (Common/Utils.hpp:16) outside them.  Pinv = inverse of the 3x3 projection.
struct RectifyCam {
    double K[9];
    double d[5];
    double Pinv[9];
};

template <int K>
__device__ __forceinline__ void rectify_pixel(const RectifyCam &c, int i, int j, int w, int h,
                                              double &mx, double &my)
{
    const double t0 = c.Pinv[0] * i + c.Pinv[1] * j + c.Pinv[2] * 1.0;
    const double t1 = c.Pinv[3] * i + c.Pinv[4] * j + c.Pinv[5] * 1.0;
    const double t2 = c.Pinv[6] * i + c.Pinv[7] * j + c.Pinv[8] * 1.0;
    const double nx = t0 / t2, ny = t1 / t2;
    const double r = nx * nx + ny * ny;
    double xe, ye;
    if (K == 2) {
        xe = nx * (1.0 + c.d[0] * r + c.d[1] * r * r);
        ye = ny * (1.0 + c.d[0] * r + c.d[1] * r * r);
    } else {
        // the order of the two tangential terms of ye is that of :95-96, not that of the
        // calibration's value()
        xe = nx * (1.0 + c.d[0] * r + c.d[1] * r * r + c.d[4] * r * r * r) + 2.0 * c.d[2] * nx * ny +
             c.d[3] * (r * r + 2.0 * nx * nx);
        ye = ny * (1.0 + c.d[0] * r + c.d[1] * r * r + c.d[4] * r * r * r) + 2.0 * c.d[3] * nx * ny +
             c.d[2] * (r * r + 2.0 * ny * ny);
    }
    const double tx = c.K[0] * xe + c.K[1] * ye + c.K[2];
    const double ty = c.K[4] * ye + c.K[5];
    const bool ok = tx >= 0 && ty >= 0 && tx <= w - 1 && ty <= h - 1;
    mx = ok ? tx : 9.99e+04;
    my = ok ? ty : 9.99e+04;
}

template <int K>
__global__ __launch_bounds__(256) void rectify_maps_kernel(RectifyCam L, RectifyCam R, int w, int h,
                                                           double *__restrict__ xl,
                                                           double *__restrict__ yl,
                                                           double *__restrict__ xr,
                                                           double *__restrict__ yr)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long long)w * h) return;
    const int j = (int)(p / w), i = (int)(p - (long long)j * w);
    double a, b;
    rectify_pixel<K>(L, i, j, w, h, a, b);
    xl[p] = a; yl[p] = b;
    rectify_pixel<K>(R, i, j, w, h, a, b);
    xr[p] = a; yr[p] = b;
}

// ------------------------------------------------------------------------
// The damped step of the refinement loop on the device:
//     step = -(A + lam diag(A))^-1 G                (CStereoCameraCalibration.cpp:288-291)
// without forming A.  The per-view 6 x 6 blocks (EE) couple only with the left camera's
// intrinsics (LE: the reference fills a view's extrinsic columns in the left image's rows
// only), so they are eliminated first, view by view (cal_view_eliminate_kernel), which
// leaves a system in the shared parameters (cal_shared_solve_kernel: 2 nI + 6 of them, nI
// monocular), and the views' steps follow from the shared one (cal_view_step_kernel).
// The reference inverts the dense matrix (12 024 x 12 024 at 2000 views).
// ------------------------------------------------------------------------
constexpr int kMaxShared = 24;      // 2 * (4 + 5) + 6

// entry (a, b) of a 16 x 16 upper triangle stored row by row
__device__ __forceinline__ int tri_at(int a, int b)
{
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return lo * 16 - (lo * (lo - 1)) / 2 + (hi - lo);
}

// W[v] = (EE + lam diag EE)^-1 [LE^T | gE]   (6 x (nI + 1), kept for the
// last kernel) and contrib[v] = LE W[v]   (nI x (nI + 1): what the view takes out of the shared
// system).  flags[0] is set when a block is not positive definite.
__global__ void cal_view_eliminate_kernel(const double *__restrict__ gram_view, int V, int nI,
                                          double lam, double *__restrict__ W,
                                          double *__restrict__ contrib, int *__restrict__ flags)
{
    // one thread per (view, column of [LE^T | gE]): the 6 x 6 factorisation is short, and
    // repeated by the nI + 1 threads of a view it keeps ten times as many lanes busy
    const int nc = nI + 1;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = gid / nc, c = gid - v * nc;
    if (v >= V) return;
    const double *G = gram_view + (size_t)v * 2 * kGram;       // the left camera's triangle
    double L[6][6];
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int p = 0; p <= q; ++p) L[q][p] = G[tri_at(nI + q, nI + p)] * (q == p ? 1.0 + lam : 1.0);
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 6; ++c) {                              // Cholesky, lower factor in place
        double d = L[c][c];
#pragma unroll
        for (int k = 0; k < c; ++k) d -= L[c][k] * L[c][k];
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        const double r = 1.0 / sqrt(d);
        L[c][c] = r;                                           // the reciprocal of the diagonal
#pragma unroll
        for (int q = c + 1; q < 6; ++q) {
            double s = L[q][c];
#pragma unroll
            for (int k = 0; k < c; ++k) s -= L[q][k] * L[c][k];
            L[q][c] = s * r;
        }
    }
    if (!ok && c == 0) atomicOr(flags, 1);
    {                                                          // column c of [LE^T | gE]
        double x[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) x[q] = G[tri_at(c < nI ? c : nI + q, c < nI ? nI + q : 15)];
#pragma unroll
        for (int q = 0; q < 6; ++q) {                          // L y = b
            double s = x[q];
#pragma unroll
            for (int k = 0; k < q; ++k) s -= L[q][k] * x[k];
            x[q] = s * L[q][q];
        }
#pragma unroll
        for (int q = 5; q >= 0; --q) {                         // L^T w = y
            double s = x[q];
#pragma unroll
            for (int k = q + 1; k < 6; ++k) s -= L[k][q] * x[k];
            x[q] = s * L[q][q];
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) W[((size_t)v * 6 + q) * nc + c] = x[q];
        for (int a = 0; a < nI; ++a) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < 6; ++q) s += G[tri_at(a, nI + q)] * x[q];
            contrib[(size_t)v * nI * nc + a * nc + c] = s;
        }
    }
}

constexpr int kSolveSums = 9 * 10;   // a view's contribution: nI x (nI + 1) doubles, nI <= 9

// The views' contributions summed in a fixed order into csum[0 .. kSolveSums) (entries past
// nI (nI + 1): zero), one block per entry.  With the views split over several devices this is
// the part of the shared system that has to be added up across them before it is solved.
__global__ __launch_bounds__(256) void cal_contrib_sum_kernel(const double *__restrict__ contrib,
                                                              int V, int nI,
                                                              double *__restrict__ csum)
{
    const int ne = nI * (nI + 1);
    column_sum(contrib, V, ne, ne, csum);
}

// One block.  Assembles the damped shared system from the summed triangles and the summed
// contributions of the views, solves it (Cholesky in LDS) and writes the shared part of the
// step: dc[0 .. nC), followed by dc^T (lam dc - gc) in dc[kMaxShared].
__global__ __launch_bounds__(256) void cal_shared_solve_kernel(
    const double *__restrict__ gram_sum, const double *__restrict__ csum_in, int nI,
    int stereo, int freeze, double lam, double *__restrict__ dc_out, int *__restrict__ flags)
{
    // freeze: the intrinsics keep their values (update_intrinsics = false,
    // CStereoCameraCalibration.cpp:187-245: the Jacobian holds the views' and the pose's columns
    // only).  Their rows and columns become the identity with a zero gradient, so their
    // step is exactly zero and the pose block is solved alone; nothing of the views couples
    // with what is left.
    __shared__ double S[kMaxShared][kMaxShared + 1];           // last column: the gradient
    __shared__ double csum[kSolveSums];
    const int nc = nI + 1, nC = stereo ? 2 * nI + 6 : nI;
    const int t = threadIdx.x;
    if (t < kSolveSums) csum[t] = csum_in[t];
    for (int i = t; i < kMaxShared * (kMaxShared + 1); i += 256) (&S[0][0])[i] = 0.0;
    __syncthreads();
    // ---- the shared system ----
    const double *GL = gram_sum, *GR = gram_sum + kGram;
    for (int i = t; i < nC * nC; i += 256) {
        const int r = i / nC, c = i % nC;
        double a = 0.0;
        if (r < nI && c < nI) a = GL[tri_at(r, c)] - csum[r * nc + c];
        else if (stereo && r >= nI && c >= nI) {
            const int rr = r - nI, cc = c - nI;                // right intrinsics, then the pose
            a = GR[tri_at(rr, cc)];                            // (RR, RS, SS sit where u has them)
        }
        a = r == c ? a + lam * (r < nI ? GL[tri_at(r, r)] : GR[tri_at(r - nI, r - nI)]) : a;
        const int frozen = stereo ? 2 * nI : nI;               // the intrinsic parameters come first
        if (freeze && (r < frozen || c < frozen)) a = r == c ? 1.0 : 0.0;
        S[r][c] = a;
    }
    for (int r = t; r < nC; r += 256) {
        double g = r < nI ? GL[tri_at(r, 15)] - csum[r * nc + nI] : GR[tri_at(r - nI, 15)];
        if (freeze && r < (stereo ? 2 * nI : nI)) g = 0.0;
        S[r][kMaxShared] = g;
    }
    __syncthreads();
    // ---- Cholesky, column by column, the rest of the matrix updated as each column is done
    // (every entry of the rest by its own thread, one barrier per column) and the gradient
    // carried along as one more row, so that L y = g is solved on the way.  Column c is
    // scaled while column c + 1 is at work: the two touch different entries. ----
    __shared__ double x[kMaxShared], term[kMaxShared];
    constexpr int G = kMaxShared;
    for (int c = 0; c < nC; ++c) {
        const double d = S[c][c];
        if (!(d > 0.0) && t == 0) atomicOr(flags, 2);
        const double inv_d = d > 0.0 ? 1.0 / d : 1.0, r = sqrt(inv_d);
        const double gc = S[c][G];
        double col_t = 0.0;
        if (t > c && t < nC) col_t = S[t][c];
        for (int e = t; e < kMaxShared * kMaxShared; e += 256) {
            const int i = c + 1 + e / kMaxShared, j = c + 1 + e % kMaxShared;
            if (j <= i && i < nC) S[i][j] -= S[i][c] * S[j][c] * inv_d;
        }
        if (t > c && t < nC) S[t][G] -= col_t * gc * inv_d;
        __syncthreads();
        if (t > c && t < nC) S[t][c] = col_t * r;              // L[t][c]
        if (t == c) { S[c][c] = r; S[c][G] = gc * r; }         // 1 / L[c][c], y[c]
    }
    __syncthreads();
    // ---- L^T x = y, from the last row up ----
    for (int rr = nC - 1; rr >= 0; --rr) {
        const double xr = S[rr][G] * S[rr][rr];
        if (t < rr) S[t][G] -= S[rr][t] * xr;
        if (t == rr) x[rr] = xr;
        __syncthreads();
    }
    // the gradient of the shared parameters as the caller knows it (before elimination)
    if (t < nC) {
        const double g = t < nI ? GL[tri_at(t, 15)] : GR[tri_at(t - nI, 15)];
        const double d = (freeze && t < (stereo ? 2 * nI : nI)) ? 0.0 : -x[t];
        dc_out[t] = d;
        term[t] = d * (lam * d - g);
    }
    __syncthreads();
    if (t == 0) {
        double pred = 0.0;
        for (int rr = 0; rr < nC; ++rr) pred += term[rr];
        dc_out[kMaxShared] = pred;
    }
}

// One thread per view: de = -(W[:, nI] + W[:, :nI] dc[:nI]) into the step vector (the
// reference's parameter order: shared intrinsics, the views' six each, then the stereo
// pose), and the view's share of step^T (lam step - G) into pred_view[v].
__global__ void cal_view_step_kernel(const double *__restrict__ gram_view, const double *__restrict__ W,
                                     const double *__restrict__ dc, int V, int nI, int stereo,
                                     double lam, double *__restrict__ step,
                                     double *__restrict__ pred_view)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int nc = nI + 1, head = stereo ? 2 * nI : nI;
    if (v == 0) {
        for (int r = 0; r < head; ++r) step[r] = dc[r];
        if (stereo) for (int q = 0; q < 6; ++q) step[head + 6 * (size_t)V + q] = dc[2 * nI + q];
    }
    if (v >= V) return;
    const double *G = gram_view + (size_t)v * 2 * kGram;
    double pred = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const double *w = W + ((size_t)v * 6 + q) * nc;
        double s = w[nI];
        for (int a = 0; a < nI; ++a) s += w[a] * dc[a];
        const double d = -s;
        step[head + 6 * (size_t)v + q] = d;
        pred += d * (lam * d - G[tri_at(nI + q, 15)]);
    }
    pred_view[v] = pred;
}

// One block, after cal_view_step_kernel: what the host reads besides the step, behind it in
// the same buffer (one copy): tail[0] = the shared parameters' part of step^T (lam step - G),
// tail[1] = the views' part, summed in a fixed order, tail[2] = the flags.
__global__ __launch_bounds__(256) void cal_step_tail_kernel(const double *__restrict__ pred_view, int V,
                                                            const double *__restrict__ dc,
                                                            const int *__restrict__ flags,
                                                            double *__restrict__ tail)
{
    column_sum(pred_view, V, 1, 1, tail + 1);
    if (threadIdx.x == 0) { tail[0] = dc[kMaxShared]; tail[2] = (double)*flags; }
}

}  // namespace svcal
