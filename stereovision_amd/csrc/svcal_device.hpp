// Device code of the calibration reprojection path (gfx950): per (view, corner)
// residuals and central-difference Jacobian rows, reduced per view to the Gram
// matrix of [J | e], then over views.
//
// Layout: one block per view.  The first 26 threads build the 13 + 13 extrinsic
// matrices the view needs (base and +/- eps for each of its six left-extrinsic
// parameters; pose * left extrinsic for the base pose and +/- eps of each of the
// six pose parameters) in LDS.  Each thread then takes corners; per corner and
// camera it writes two 16-vectors u = [J row (nI + 6 columns, zero padded to
// 15), e] to LDS, and the block sums the upper triangle of u u^T (136 numbers
// per camera) over the view's corners.  Everything is fp64, as the reference
// (Data_Type = double).
#pragma once
#include <hip/hip_runtime.h>

namespace svcal {

constexpr int kThreads = 128;
constexpr int kGram = 136;          // upper triangle of a 16 x 16 matrix
constexpr double kEps = 1.0e-6;     // CStereoCameraCalibration.cpp:84

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
    double sn = sin(theta), cs = 1.0 - cos(theta);
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

// CStereoCameraCalibration.cpp:25-50; intr = {fx, fy, cx, cy}, skew 0
__device__ __forceinline__ void value(const double intr[4], const double dist[5], int k,
                                      const double *E, double wx, double wy, double &u, double &v)
{
    double c0 = E[0] * wx + E[1] * wy + E[2] * 0.0 + E[3] * 1.0;
    double c1 = E[4] * wx + E[5] * wy + E[6] * 0.0 + E[7] * 1.0;
    double c2 = E[8] * wx + E[9] * wy + E[10] * 0.0 + E[11] * 1.0;
    double nx = c0 / c2, ny = c1 / c2;
    double r = nx * nx + ny * ny;
    double xe, ye;
    if (k == 2) {
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

// One camera's two rows of [J | e] for one corner: columns 0..nI-1 intrinsics
// and distortion, nI..nI+5 the six pose-like parameters whose +/- eps matrices
// are Evar[1 + 2q], Evar[2 + 2q]; the residual (projected - observed) in slot 15.
__device__ inline void rows_for(const double *ip /* nI values */, int k, const double *Evar,
                                double wx, double wy, double ox, double oy, double *u0, double *u1)
{
    const int nI = 4 + k;
    double intr[4] = { ip[0], ip[1], ip[2], ip[3] };
    double dist[5] = { ip[4], ip[5], k == 5 ? ip[6] : 0.0, k == 5 ? ip[7] : 0.0,
                       k == 5 ? ip[8] : 0.0 };
    double a, b, c, d;
    value(intr, dist, k, Evar, wx, wy, a, b);
    for (int q = 0; q < 16; ++q) { u0[q] = 0.0; u1[q] = 0.0; }
    u0[15] = a - ox; u1[15] = b - oy;
    for (int col = 0; col < nI; ++col) {
        double iu[4] = { intr[0], intr[1], intr[2], intr[3] }, il[4] = { intr[0], intr[1], intr[2], intr[3] };
        double du[5] = { dist[0], dist[1], dist[2], dist[3], dist[4] }, dl[5] = { dist[0], dist[1], dist[2], dist[3], dist[4] };
        if (col < 4) { iu[col] += kEps; il[col] -= kEps; }
        else { du[col - 4] += kEps; dl[col - 4] -= kEps; }
        value(iu, du, k, Evar, wx, wy, a, b);
        value(il, dl, k, Evar, wx, wy, c, d);
        u0[col] = (a - c) / (2.0 * kEps);
        u1[col] = (b - d) / (2.0 * kEps);
    }
    for (int q = 0; q < 6; ++q) {
        value(intr, dist, k, Evar + 12 * (1 + 2 * q), wx, wy, a, b);
        value(intr, dist, k, Evar + 12 * (2 + 2 * q), wx, wy, c, d);
        u0[nI + q] = (a - c) / (2.0 * kEps);
        u1[nI + q] = (b - d) / (2.0 * kEps);
    }
}

// params: the parameter vector; gram_view: V x 2 x 136 (left, right) per view;
// residual: 4 V C (stereo order) or 2 V C (monocular, sign flipped), may be null.
__global__ __launch_bounds__(kThreads) void cal_eval_kernel(
    const double *__restrict__ params, int k, int V, int C, int stereo,
    const double *__restrict__ world_xy, const double *__restrict__ left_xy,
    const double *__restrict__ right_xy, double *__restrict__ gram_view,
    double *__restrict__ residual)
{
    __shared__ double ext[2][13 * 12];          // [camera][variant][3x4]
    __shared__ double rows[kThreads][2][2][16]; // [corner in chunk][camera][row][16]
    const int view = blockIdx.x;
    const int nI = 4 + k;
    const int t = threadIdx.x;
    const double *pE = params + (stereo ? 2 * nI : nI) + 6 * view;
    const double *pS = params + 2 * nI + 6 * V;

    // variant 0: base; 1 + 2q / 2 + 2q: parameter q plus / minus eps
    if (t < 13) {
        double p[6];
        for (int q = 0; q < 6; ++q) p[q] = pE[q];
        if (t > 0) p[(t - 1) >> 1] += ((t - 1) & 1) ? -kEps : kEps;
        ext_from(p, &ext[0][12 * t]);
    }
    __syncthreads();
    if (stereo && t < 13) {
        double p[6], S[12];
        for (int q = 0; q < 6; ++q) p[q] = pS[q];
        if (t > 0) p[(t - 1) >> 1] += ((t - 1) & 1) ? -kEps : kEps;
        ext_from(p, S);
        // pose * [left extrinsic; 0 0 0 1]  (CStereoCameraCalibration.cpp:573-579)
        const double *L = &ext[0][0];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                double s = 0;
                for (int kk = 0; kk < 3; ++kk) s += S[4 * r + kk] * L[4 * kk + c];
                s += S[4 * r + 3] * (c == 3 ? 1.0 : 0.0);
                ext[1][12 * t + 4 * r + c] = s;
            }
    }
    __syncthreads();

    // tasks: entry (a, b), a <= b, of the 16 x 16 Gram matrix of each camera
    double sum[3] = { 0.0, 0.0, 0.0 };
    int ta[3], tb[3], tc[3];
    for (int s = 0; s < 3; ++s) {
        int task = t + s * kThreads;
        tc[s] = task / kGram;                  // camera; 2: no task
        int e = task % kGram, a = 0;
        while (e >= 16 - a) { e -= 16 - a; ++a; }
        ta[s] = a; tb[s] = a + e;
        if (tc[s] > (stereo ? 1 : 0)) tc[s] = 2;
    }

    for (int c0 = 0; c0 < C; c0 += kThreads) {
        const int j = c0 + t;
        if (j < C) {
            const double wx = world_xy[2 * j], wy = world_xy[2 * j + 1];
            const size_t o = 2 * ((size_t)view * C + j);
            rows_for(params, k, &ext[0][0], wx, wy, left_xy[o], left_xy[o + 1], rows[t][0][0],
                     rows[t][0][1]);
            if (stereo)
                rows_for(params + nI, k, &ext[1][0], wx, wy, right_xy[o], right_xy[o + 1],
                         rows[t][1][0], rows[t][1][1]);
            if (residual) {
                if (stereo) {
                    residual[o] = rows[t][0][0][15];
                    residual[o + 1] = rows[t][0][1][15];
                    residual[2 * (size_t)V * C + o] = rows[t][1][0][15];
                    residual[2 * (size_t)V * C + o + 1] = rows[t][1][1][15];
                } else {
                    residual[o] = -rows[t][0][0][15];
                    residual[o + 1] = -rows[t][0][1][15];
                }
            }
        }
        __syncthreads();
        const int cnt = min(kThreads, C - c0);
        for (int s = 0; s < 3; ++s) {
            if (tc[s] == 2) continue;
            double acc = sum[s];
            for (int jj = 0; jj < cnt; ++jj) {
                acc += rows[jj][tc[s]][0][ta[s]] * rows[jj][tc[s]][0][tb[s]];
                acc += rows[jj][tc[s]][1][ta[s]] * rows[jj][tc[s]][1][tb[s]];
            }
            sum[s] = acc;
        }
        __syncthreads();
    }
    for (int s = 0; s < 3; ++s)
        if (tc[s] != 2)
            gram_view[((size_t)view * 2 + tc[s]) * kGram + (t + s * kThreads) % kGram] = sum[s];
}

// Sums the per-view Gram matrices over the views: out[camera][136].
__global__ __launch_bounds__(256) void cal_reduce_kernel(const double *__restrict__ gram_view, int V,
                                                         double *__restrict__ gram_sum)
{
    __shared__ double red[256];
    const int task = blockIdx.x;           // camera * 136 + entry
    double s = 0.0;
    for (int v = threadIdx.x; v < V; v += 256) s += gram_view[(size_t)v * 2 * kGram + task];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) gram_sum[task] = red[0];
}

}  // namespace svcal
