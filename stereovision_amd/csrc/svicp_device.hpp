// Device-side data structures and kernels of libsvicp (gfx950 only).
//
// Layout in HBM
//   target   : float4 {x, y, z, original index as int bits}, sorted by linear
//              cell key (z-major, x fastest), so the points of x-adjacent cells
//              are contiguous;  normals packed 3 x fp32 in the same order.
//   grid     : cells of edge h >= 2 * search radius.  Cells are grouped in
//              4x4x4 blocks; `dir` maps a block to its rank among the blocks
//              that hold (or precede in x) a point, -1 otherwise.  Per ranked
//              block 64 entries: run2 = {first point of the cell, end of the
//              NEXT cell in x} and end1 = end of the cell itself.
//   source   : SoA x[], y[], z[] in Morton order (perm[] maps back).
//
// A query probes the 2x2x2 cells that cover its search cube: four 2-cell runs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svicp {

constexpr int kBlockThreads = 256;
constexpr int kNumSums = 32;   // 21 JtJ + 6 Jtr + inliers + sum w b^2 + sum w + 2 pad
constexpr int kMaxRing = 3;    // exact search: rings tried before the brute-force pass

struct GridDesc {
    float ox, oy, oz;        // origin (one empty cell below the target's bounding box)
    float h, inv_h;          // cell edge
    float rr;                // probe radius: sqrt(threshold) plus rounding slack
    float cov_slack;         // absolute slack when a scanned box is declared to cover a radius
    float fnx1, fny1, fnz1;  // (float)(n - 1): upper bound of a valid probe origin
    int nx, ny, nz;          // cells per axis
    int nbx, nby, nbz;       // blocks per axis
    int m;                   // target points
    const int *dir;          // [nbz][nby][nbx]
    const uint2 *run2;       // [ranked blocks * 64]
    const unsigned *end1;    // [ranked blocks * 64]
    const float4 *pts;       // [m] sorted
    const float *nrm;        // [3 m] sorted
};

// Everything the iteration loop keeps on the device between kernels.
struct IcpState {
    double R[9];
    double T[3];
    double Trans[16];
    double S[kNumSums];      // last reduced sums
    int status;
    int pad;
};

__device__ __forceinline__ bool better(float d, int id, float bd, int bid)
{
    return (d < bd) || (d == bd && id < bid);
}

// Squared distance exactly as flann::L2<float> accumulates it for three
// dimensions: ((dx*dx) + dy*dy) + dz*dz, every operation rounded to fp32.
__device__ __forceinline__ float dist2(float qx, float qy, float qz, const float4 &p)
{
    float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    float d = __fmul_rn(dx, dx);
    d = __fadd_rn(d, __fmul_rn(dy, dy));
    d = __fadd_rn(d, __fmul_rn(dz, dz));
    return d;
}

__device__ __forceinline__ int cell_local(int cx, int cy, int cz)
{
    return ((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3);
}

__device__ __forceinline__ int dir_index(const GridDesc &g, int cx, int cy, int cz)
{
    return ((cz >> 2) * g.nby + (cy >> 2)) * g.nbx + (cx >> 2);
}

// Probe origin: the cell that holds q - rr on each axis.  Returns false when
// the 2x2x2 probe would leave the grid, i.e. no target point lies within rr.
__device__ __forceinline__ bool probe_origin(const GridDesc &g, float qx, float qy, float qz,
                                             int &cx, int &cy, int &cz)
{
    float fx = ((qx - g.rr) - g.ox) * g.inv_h;
    float fy = ((qy - g.rr) - g.oy) * g.inv_h;
    float fz = ((qz - g.rr) - g.oz) * g.inv_h;
    bool ok = (fx >= 0.f) && (fy >= 0.f) && (fz >= 0.f) && (fx < g.fnx1) && (fy < g.fny1) &&
              (fz < g.fnz1);
    if (!ok) return false;
    cx = (int)fx; cy = (int)fy; cz = (int)fz;
    return true;
}

// Nearest target point within the probe radius: the 1-NN the ICP loop needs.
// best = +inf, bpos = -1 when the probed cells are empty.
__device__ __forceinline__ void nn_capped(const GridDesc &g, float qx, float qy, float qz,
                                          float &best, int &bpos, int &bidx)
{
    best = INFINITY; bpos = -1; bidx = 0x7fffffff;
    int cx, cy, cz;
    if (!probe_origin(g, qx, qy, qz, cx, cy, cz)) return;

    int rank[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        rank[r] = g.dir[dir_index(g, cx, cy + (r & 1), cz + (r >> 1))];
    uint2 run[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        run[r] = make_uint2(0u, 0u);
        if (rank[r] >= 0)
            run[r] = g.run2[rank[r] * 64 + cell_local(cx, cy + (r & 1), cz + (r >> 1))];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        for (unsigned j = run[r].x; j < run[r].y; ++j) {
            float4 p = g.pts[j];
            float d = dist2(qx, qy, qz, p);
            int id = __float_as_int(p.w);
            if (better(d, id, best, bidx)) { best = d; bpos = (int)j; bidx = id; }
        }
    }
}

// Visits every target point in the cell box [x0,x1] x [y0,y1] x [z0,z1]
// (already clipped to the grid) exactly once.
template <class F>
__device__ __forceinline__ void scan_box(const GridDesc &g, int x0, int x1, int y0, int y1,
                                         int z0, int z1, F &&visit)
{
    for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y)
            for (int bx = x0 >> 2; bx <= (x1 >> 2); ++bx) {
                int rank = g.dir[((z >> 2) * g.nby + (y >> 2)) * g.nbx + bx];
                if (rank < 0) continue;
                int xa = max(x0, bx * 4), xb = min(x1, bx * 4 + 3);
                int base = rank * 64 + (((z & 3) << 4) | ((y & 3) << 2));
                unsigned s = g.run2[base + (xa & 3)].x;
                unsigned e = g.end1[base + (xb & 3)];
                for (unsigned j = s; j < e; ++j) visit(j);
            }
}

// Box of ring `ring` around the probe origin, clipped; returns the radius
// around q that the box is certain to cover (+inf once it spans the grid).
__device__ __forceinline__ float ring_box(const GridDesc &g, float qx, float qy, float qz, int cx,
                                          int cy, int cz, int ring, int &x0, int &x1, int &y0,
                                          int &y1, int &z0, int &z1)
{
    float rho = INFINITY;
    auto axis = [&](float q, float o, int c, int n, int &a0, int &a1) {
        int lo = c - ring, hi = c + 1 + ring;
        if (lo > 0) rho = fminf(rho, q - (o + (float)lo * g.h));
        if (hi < n - 1) rho = fminf(rho, (o + (float)(hi + 1) * g.h) - q);
        a0 = max(lo, 0);
        a1 = min(hi, n - 1);
    };
    axis(qx, g.ox, cx, g.nx, x0, x1);
    axis(qy, g.oy, cy, g.ny, y0, y1);
    axis(qz, g.oz, cz, g.nz, z0, z1);
    if (rho == INFINITY) return INFINITY;
    return rho * (1.0f - 1e-5f) - g.cov_slack;
}

__device__ __forceinline__ void clamped_origin(const GridDesc &g, float qx, float qy, float qz,
                                               int &cx, int &cy, int &cz)
{
    auto cl = [&](float q, float o, int n) {
        float f = ((q - g.rr) - o) * g.inv_h;
        f = fminf(fmaxf(f, 0.f), (float)(n - 2));   // NaN -> 0
        return (int)f;
    };
    cx = cl(qx, g.ox, g.nx); cy = cl(qy, g.oy, g.ny); cz = cl(qz, g.oz, g.nz);
}

// ------------------------------------------------------------------------
// wave / block reductions
// ------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------
// The per-iteration kernel: transform, correspondence search, residual,
// Jacobian row and normal-equation accumulation in one pass over the source.
//
// Restates CPoint2PlaneICP.cpp:42-50 (update, fp64 math rounded to fp32),
// :65-79 (1-NN, weight) and :85-109 (row assembly, weighted normal equations).
// ------------------------------------------------------------------------
template <bool APPLY>
__global__ __launch_bounds__(kBlockThreads) void icp_accumulate_kernel(
    GridDesc g, const float *__restrict__ sx_in, const float *__restrict__ sy_in,
    const float *__restrict__ sz_in, float *__restrict__ sx_out, float *__restrict__ sy_out,
    float *__restrict__ sz_out, int n, int chunk, const IcpState *__restrict__ st, float thr,
    double *__restrict__ partials)
{
    // blocks that share an XCD (blockIdx % 8 equal) take neighbouring chunks,
    // so each XCD's L2 sees one contiguous stretch of the Morton-ordered source
    const int nb = (int)gridDim.x;
    const int q8 = nb >> 3, r8 = nb & 7;
    const int xcd = (int)blockIdx.x & 7, within = (int)blockIdx.x >> 3;
    const int vb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;

    double R0 = 1, R1 = 0, R2 = 0, R3 = 0, R4 = 1, R5 = 0, R6 = 0, R7 = 0, R8 = 1, T0 = 0, T1 = 0,
           T2 = 0;
    if (APPLY) {
        R0 = st->R[0]; R1 = st->R[1]; R2 = st->R[2]; R3 = st->R[3]; R4 = st->R[4]; R5 = st->R[5];
        R6 = st->R[6]; R7 = st->R[7]; R8 = st->R[8]; T0 = st->T[0]; T1 = st->T[1]; T2 = st->T[2];
    }

    double acc[30];
#pragma unroll
    for (int v = 0; v < 30; ++v) acc[v] = 0.0;

    const long long begin = (long long)vb * chunk;
    const long long end = min((long long)n, begin + chunk);
    for (long long i = begin + threadIdx.x; i < end; i += kBlockThreads) {
        float px = sx_in[i], py = sy_in[i], pz = sz_in[i];
        if (APPLY) {
            // ((R0*px + R1*py) + R2*pz) + T0 in fp64, no contraction, then one rounding to fp32
            double dx = (double)px, dy = (double)py, dz = (double)pz;
            float nx = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(R0, dx), __dmul_rn(R1, dy)), __dmul_rn(R2, dz)), T0);
            float ny = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(R3, dx), __dmul_rn(R4, dy)), __dmul_rn(R5, dz)), T1);
            float nz = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(R6, dx), __dmul_rn(R7, dy)), __dmul_rn(R8, dz)), T2);
            px = nx; py = ny; pz = nz;
            sx_out[i] = px; sy_out[i] = py; sz_out[i] = pz;
        }
        float best; int bpos, bidx;
        nn_capped(g, px, py, pz, best, bpos, bidx);
        if (bpos >= 0 && best <= thr) {
            float sl = best / thr;
            float ss = __fmul_rn(sl, sl);
            double om = 1.0 - (double)ss;
            float w = (float)__dmul_rn(om, om);
            float4 t = g.pts[bpos];
            float nx = g.nrm[3 * (size_t)bpos], ny = g.nrm[3 * (size_t)bpos + 1],
                  nz = g.nrm[3 * (size_t)bpos + 2];
            float a0 = __fsub_rn(__fmul_rn(py, nz), __fmul_rn(pz, ny));
            float a1 = __fsub_rn(__fmul_rn(pz, nx), __fmul_rn(px, nz));
            float a2 = __fsub_rn(__fmul_rn(px, ny), __fmul_rn(py, nx));
            float b = __fadd_rn(__fadd_rn(__fmul_rn(__fsub_rn(t.x, px), nx),
                                          __fmul_rn(__fsub_rn(t.y, py), ny)),
                                __fmul_rn(__fsub_rn(t.z, pz), nz));
            double a[6] = { (double)a0, (double)a1, (double)a2, (double)nx, (double)ny, (double)nz };
            double wd = (double)w, bd = (double)b;
            int v = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                double aw = a[r] * wd;
#pragma unroll
                for (int c = r; c < 6; ++c) acc[v++] += aw * a[c];
                acc[21 + r] += aw * bd;
            }
            if (w > 0.f) acc[27] += 1.0;
            acc[28] += wd * bd * bd;
            acc[29] += wd;
        }
    }

    __shared__ double red[kBlockThreads / 64][30];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < 30; ++v) {
        double s = wave_sum(acc[v]);
        if (lane == 0) red[wv][v] = s;
    }
    __syncthreads();
    if (threadIdx.x < kNumSums) {
        double s = 0.0;
        if (threadIdx.x < 30) {
#pragma unroll
            for (int w = 0; w < kBlockThreads / 64; ++w) s += red[w][threadIdx.x];
        }
        partials[(size_t)threadIdx.x * nb + blockIdx.x] = s;
    }
}

// ------------------------------------------------------------------------
// 6x6 solve and transform update on the device.
// Restates CPoint2PlaneICP.cpp:109 (X), :111-119 (R = Rz*Ry*Rx, T) and
// :36-39 / :52-55 (Trans = [R T; 0 1] * Trans).
// ------------------------------------------------------------------------
// Called by every thread of the block; S points to 32 sums in LDS or global.
__device__ inline void solve_and_update(const double *S, IcpState *st, double *X_hist,
                                        double *info_hist, int iter)
{
    __shared__ double shX[6];
    __shared__ double shSC[6];
    __shared__ int shOk;
    if (threadIdx.x == 0) {
        // symmetric 6x6 in registers: every index below is a compile-time constant
        double A[6][6], b[6], d[6], L[6][6], y[6], X[6];
        {
            int v = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = r; c < 6; ++c) { A[r][c] = S[v]; A[c][r] = S[v]; ++v; }
        }
        bool ok = true;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            b[r] = S[21 + r];
            if (!(A[r][r] > 0.0) || !isfinite(A[r][r])) ok = false;
        }
        // Cholesky of the diagonally scaled matrix D A D, D = diag(A)^-1/2
#pragma unroll
        for (int r = 0; r < 6; ++r) d[r] = ok ? 1.0 / sqrt(A[r][r]) : 1.0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                double s = A[r][c] * d[r] * d[c];
#pragma unroll
                for (int k = 0; k < c; ++k) s -= L[r][k] * L[c][k];
                if (r == c) {
                    if (!(s > 1e-14) || !isfinite(s)) { ok = false; s = 1.0; }
                    L[r][r] = sqrt(s);
                } else
                    L[r][c] = s / L[c][c];
            }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double s = b[r] * d[r];
#pragma unroll
            for (int k = 0; k < r; ++k) s -= L[r][k] * y[k];
            y[r] = s / L[r][r];
        }
#pragma unroll
        for (int r = 5; r >= 0; --r) {
            double s = y[r];
#pragma unroll
            for (int k = r + 1; k < 6; ++k) s -= L[k][r] * X[k];
            X[r] = s / L[r][r];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            X[r] *= d[r];
            if (!isfinite(X[r])) ok = false;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) shX[r] = ok ? X[r] : 0.0;
        shOk = ok ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double sn, cs;
        sincos(shX[threadIdx.x], &sn, &cs);
        shSC[2 * threadIdx.x] = sn;
        shSC[2 * threadIdx.x + 1] = cs;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;

    if (!shOk) st->status = 4;   // SVICP_ERR_SINGULAR
    const double sx = shSC[0], cx = shSC[1], sy = shSC[2], cy = shSC[3], sz = shSC[4], cz = shSC[5];
    const double X0 = shX[0], X1 = shX[1], X2 = shX[2], X3 = shX[3], X4 = shX[4], X5 = shX[5];
    // R = Rz(X2) * Ry(X1) * Rx(X0)
    const double R[9] = { cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                          sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                          -sy,     cy * sx,                cy * cx };
#pragma unroll
    for (int k = 0; k < 9; ++k) st->R[k] = R[k];
    st->T[0] = X3; st->T[1] = X4; st->T[2] = X5;

    // Trans = [R T; 0 1] * Trans
    double Tr[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) Tr[k] = st->Trans[k];
    const double Tv[3] = { X3, X4, X5 };
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double s = R[3 * r] * Tr[c] + R[3 * r + 1] * Tr[4 + c] + R[3 * r + 2] * Tr[8 + c];
            if (c == 3) s += Tv[r];
            st->Trans[4 * r + c] = s;
        }

    const double step = sqrt(X0 * X0 + X1 * X1 + X2 * X2 + X3 * X3 + X4 * X4 + X5 * X5);
    st->S[30] = step;
    if (X_hist) {
        X_hist[6 * iter + 0] = X0; X_hist[6 * iter + 1] = X1; X_hist[6 * iter + 2] = X2;
        X_hist[6 * iter + 3] = X3; X_hist[6 * iter + 4] = X4; X_hist[6 * iter + 5] = X5;
    }
    if (info_hist) {
        info_hist[4 * iter + 0] = S[27];
        info_hist[4 * iter + 1] = S[28];
        info_hist[4 * iter + 2] = step;
        info_hist[4 * iter + 3] = S[29];
    }
}

// Sums the per-block partials (value-major [32][nb]) in a fixed order, then
// either solves (sum_out == nullptr) or hands the 32 sums to the caller.
__global__ __launch_bounds__(1024) void reduce_solve_kernel(const double *__restrict__ partials,
                                                            int nb, IcpState *st, double *X_hist,
                                                            double *info_hist, int iter,
                                                            double *sum_out)
{
    // thread t sums value v = t / 32 over blocks (t % 32), (t % 32) + 32, ...:
    // all loads of a thread are independent, the 32 threads of a value read
    // consecutive doubles, and the order of additions is fixed
    __shared__ double fin[kNumSums];
    const int v = threadIdx.x >> 5, sub = threadIdx.x & 31;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const double *row = partials + (size_t)v * nb;
    int b = sub;
    for (; b + 96 < nb; b += 128) {
        s0 += row[b]; s1 += row[b + 32]; s2 += row[b + 64]; s3 += row[b + 96];
    }
    for (; b < nb; b += 32) s0 += row[b];
    double s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (sub == 0) {
        fin[v] = s;
        if (sum_out) sum_out[v] = s; else st->S[v] = s;
    }
    if (sum_out) return;
    __syncthreads();
    solve_and_update(fin, st, X_hist, info_hist, iter);
}

// Solve from sums that were reduced elsewhere (the all-reduced block of a
// sharded registration).
__global__ void solve_from_sums_kernel(const double *__restrict__ sums, IcpState *st,
                                       double *X_hist, double *info_hist, int iter)
{
    __shared__ double fin[kNumSums];
    if (blockIdx.x != 0) return;
    if (threadIdx.x < kNumSums) {
        fin[threadIdx.x] = sums[threadIdx.x];
        st->S[threadIdx.x] = sums[threadIdx.x];
    }
    __syncthreads();
    solve_and_update(fin, st, X_hist, info_hist, iter);
}

__global__ void reset_state_kernel(IcpState *st)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int k = 0; k < 9; ++k) st->R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        for (int k = 0; k < 3; ++k) st->T[k] = 0.0;
        for (int k = 0; k < 16; ++k) st->Trans[k] = (k % 5 == 0) ? 1.0 : 0.0;
        for (int k = 0; k < kNumSums; ++k) st->S[k] = 0.0;
        st->status = 0;
    }
}

// ------------------------------------------------------------------------
// Standalone 1-NN over the resident source (CKNNSearch::getKNeighbour with
// k = 1, CKNNSearch.cpp:67-71).  EXACT = false: neighbours within the probe
// radius only, as the iteration loop searches.  EXACT = true: unbounded;
// queries the ring search cannot settle are queued for the brute-force pass.
// ------------------------------------------------------------------------
template <bool EXACT>
__global__ __launch_bounds__(kBlockThreads) void nn1_kernel(
    GridDesc g, const float *__restrict__ qx_, const float *__restrict__ qy_,
    const float *__restrict__ qz_, const int *__restrict__ perm, int n, float thr,
    int *__restrict__ idx_out, float *__restrict__ d2_out, int *__restrict__ unresolved,
    int *__restrict__ n_unresolved)
{
    for (long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x; i < n;
         i += (long long)gridDim.x * kBlockThreads) {
        float qx = qx_[i], qy = qy_[i], qz = qz_[i];
        float best; int bpos, bidx;
        nn_capped(g, qx, qy, qz, best, bpos, bidx);
        const long long o = perm ? perm[i] : i;
        if (!EXACT) {
            bool hit = bpos >= 0 && best <= thr;
            idx_out[o] = hit ? bidx : -1;
            d2_out[o] = hit ? best : INFINITY;
            continue;
        }
        int cx, cy, cz;
        clamped_origin(g, qx, qy, qz, cx, cy, cz);
        bool done = false;
        for (int ring = 0; ring <= kMaxRing; ++ring) {
            int x0, x1, y0, y1, z0, z1;
            float rho = ring_box(g, qx, qy, qz, cx, cy, cz, ring, x0, x1, y0, y1, z0, z1);
            if (ring > 0 || bpos < 0) {
                // ring 0 was scanned by nn_capped unless the probe left the grid
                scan_box(g, x0, x1, y0, y1, z0, z1, [&](unsigned j) {
                    float4 p = g.pts[j];
                    float d = dist2(qx, qy, qz, p);
                    int id = __float_as_int(p.w);
                    if (better(d, id, best, bidx)) { best = d; bpos = (int)j; bidx = id; }
                });
            }
            if (rho == INFINITY || (bpos >= 0 && rho > 0.f && best <= rho * rho)) { done = true; break; }
        }
        if (done) {
            idx_out[o] = bpos >= 0 ? bidx : -1;
            d2_out[o] = bpos >= 0 ? best : INFINITY;
        } else {
            int slot = atomicAdd(n_unresolved, 1);
            unresolved[slot] = (int)i;
        }
    }
}

// One block per queued query: scan the whole target.
__global__ __launch_bounds__(kBlockThreads) void nn1_brute_kernel(
    GridDesc g, const float *__restrict__ qx_, const float *__restrict__ qy_,
    const float *__restrict__ qz_, const int *__restrict__ perm,
    const int *__restrict__ unresolved, const int *__restrict__ n_unresolved,
    int *__restrict__ idx_out, float *__restrict__ d2_out)
{
    __shared__ float sd[kBlockThreads];
    __shared__ int si[kBlockThreads];
    const int cnt = *n_unresolved;
    for (int u = blockIdx.x; u < cnt; u += gridDim.x) {
        const int i = unresolved[u];
        float qx = qx_[i], qy = qy_[i], qz = qz_[i];
        float best = INFINITY; int bidx = 0x7fffffff;
        for (int j = threadIdx.x; j < g.m; j += kBlockThreads) {
            float4 p = g.pts[j];
            float d = dist2(qx, qy, qz, p);
            int id = __float_as_int(p.w);
            if (better(d, id, best, bidx)) { best = d; bidx = id; }
        }
        sd[threadIdx.x] = best; si[threadIdx.x] = bidx;
        __syncthreads();
        for (int s = kBlockThreads / 2; s > 0; s >>= 1) {
            if (threadIdx.x < s &&
                better(sd[threadIdx.x + s], si[threadIdx.x + s], sd[threadIdx.x], si[threadIdx.x])) {
                sd[threadIdx.x] = sd[threadIdx.x + s];
                si[threadIdx.x] = si[threadIdx.x + s];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const long long o = perm ? perm[i] : i;
            idx_out[o] = g.m > 0 ? si[0] : -1;
            d2_out[o] = g.m > 0 ? sd[0] : INFINITY;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------
// Exact k-NN, 1 <= k <= KMAX (CKNNSearch::getKNeighbour, CKNNSearch.cpp:67-71).
// One thread per query keeps the k best (distance, index) pairs sorted in
// private memory.  BRUTE = true scans the whole target for the queued queries.
// ------------------------------------------------------------------------
template <int KMAX>
struct KList {
    float d[KMAX];
    int id[KMAX];
    int cnt, k;
    __device__ void clear(int k_) { cnt = 0; k = k_; }
    __device__ void insert(float dd, int ii)
    {
        if (cnt == k && !better(dd, ii, d[k - 1], id[k - 1])) return;
        int pos = cnt < k ? cnt : k - 1;
        while (pos > 0 && better(dd, ii, d[pos - 1], id[pos - 1])) {
            d[pos] = d[pos - 1]; id[pos] = id[pos - 1]; --pos;
        }
        d[pos] = dd; id[pos] = ii;
        if (cnt < k) ++cnt;
    }
};

template <int KMAX, bool BRUTE>
__global__ __launch_bounds__(kBlockThreads) void knn_kernel(
    GridDesc g, const float *__restrict__ q_aos, int n, int k, int *__restrict__ idx_out,
    float *__restrict__ d2_out, int *__restrict__ unresolved, int *__restrict__ n_unresolved)
{
    const int total = BRUTE ? *n_unresolved : n;
    for (long long t = (long long)blockIdx.x * kBlockThreads + threadIdx.x; t < total;
         t += (long long)gridDim.x * kBlockThreads) {
        const long long i = BRUTE ? unresolved[t] : t;
        float qx = q_aos[3 * i], qy = q_aos[3 * i + 1], qz = q_aos[3 * i + 2];
        KList<KMAX> L;
        L.clear(k);
        bool done = false;
        if (BRUTE) {
            for (int j = 0; j < g.m; ++j) {
                float4 p = g.pts[j];
                L.insert(dist2(qx, qy, qz, p), __float_as_int(p.w));
            }
            done = true;
        } else {
            int cx, cy, cz;
            clamped_origin(g, qx, qy, qz, cx, cy, cz);
            for (int ring = (k > 1 ? 1 : 0); ring <= kMaxRing; ++ring) {
                int x0, x1, y0, y1, z0, z1;
                float rho = ring_box(g, qx, qy, qz, cx, cy, cz, ring, x0, x1, y0, y1, z0, z1);
                L.clear(k);
                scan_box(g, x0, x1, y0, y1, z0, z1, [&](unsigned j) {
                    float4 p = g.pts[j];
                    L.insert(dist2(qx, qy, qz, p), __float_as_int(p.w));
                });
                if (rho == INFINITY || (L.cnt == k && rho > 0.f && L.d[k - 1] <= rho * rho)) {
                    done = true;
                    break;
                }
            }
        }
        if (done) {
            for (int j = 0; j < k; ++j) {
                idx_out[i * k + j] = j < L.cnt ? L.id[j] : -1;
                d2_out[i * k + j] = j < L.cnt ? L.d[j] : INFINITY;
            }
        } else {
            int slot = atomicAdd(n_unresolved, 1);
            unresolved[slot] = (int)i;
        }
    }
}

// ------------------------------------------------------------------------
// Normal estimation from the k-NN lists.
// Restates CPoint2PlaneICP::calcNormals (CPoint2PlaneICP.cpp:130-152): raw
// second moments about the origin, each term fl32(fl32(pj*pi)/k) added in
// fp64 in neighbour order, eigenvector of the smallest eigenvalue (cyclic
// Jacobi on the 3x3), stored as fp32.  The sign of a normal is not defined by
// the reference's eigen-solver and does not affect the registration.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void normals_kernel(const float *__restrict__ ref_aos,
                                                                const int *__restrict__ knn_idx,
                                                                int m, int k,
                                                                float *__restrict__ normals)
{
    const long long p = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
    if (p >= m) return;
    double W[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
    const float kf = (float)k;
    for (int t = 0; t < k; ++t) {
        int id = knn_idx[p * k + t];
        if (id < 0) continue;
        float c[3] = { ref_aos[3 * (size_t)id], ref_aos[3 * (size_t)id + 1], ref_aos[3 * (size_t)id + 2] };
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float term = __fdiv_rn(__fmul_rn(c[j], c[i]), kf);
                W[j][i] += (double)term;
            }
    }
    double V[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = W[0][1] * W[0][1] + W[0][2] * W[0][2] + W[1][2] * W[1][2];
        double diag = W[0][0] * W[0][0] + W[1][1] * W[1][1] + W[2][2] * W[2][2];
        if (off <= 1e-300 || off <= 1e-34 * diag) break;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = a + 1; b < 3; ++b) {
                double apq = W[a][b];
                if (apq == 0.0) continue;
                double theta = (W[b][b] - W[a][a]) / (2.0 * apq);
                double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double wa = W[r][a], wb = W[r][b];
                    W[r][a] = cs * wa - sn * wb;
                    W[r][b] = sn * wa + cs * wb;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double wa = W[a][r], wb = W[b][r];
                    W[a][r] = cs * wa - sn * wb;
                    W[b][r] = sn * wa + cs * wb;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double va = V[r][a], vb = V[r][b];
                    V[r][a] = cs * va - sn * vb;
                    V[r][b] = sn * va + cs * vb;
                }
            }
    }
    double e0 = W[0][0], e1 = W[1][1], e2 = W[2][2];
    double v0, v1, v2;
    if (e0 <= e1 && e0 <= e2) { v0 = V[0][0]; v1 = V[1][0]; v2 = V[2][0]; }
    else if (e1 <= e2)        { v0 = V[0][1]; v1 = V[1][1]; v2 = V[2][1]; }
    else                      { v0 = V[0][2]; v1 = V[1][2]; v2 = V[2][2]; }
    double nrm = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
    normals[3 * p] = (float)(v0 / nrm);
    normals[3 * p + 1] = (float)(v1 / nrm);
    normals[3 * p + 2] = (float)(v2 / nrm);
}

// ------------------------------------------------------------------------
// grid construction
// ------------------------------------------------------------------------
__global__ void bbox_kernel(const float *__restrict__ aos, long long n, float *__restrict__ out6)
{
    // out6: per block {minx, miny, minz, maxx, maxy, maxz}
    __shared__ float s[6][kBlockThreads];
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x; i < n;
         i += (long long)gridDim.x * kBlockThreads)
        for (int a = 0; a < 3; ++a) {
            float v = aos[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    for (int a = 0; a < 3; ++a) { s[a][threadIdx.x] = mn[a]; s[3 + a][threadIdx.x] = mx[a]; }
    __syncthreads();
    for (int st = kBlockThreads / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st)
            for (int a = 0; a < 3; ++a) {
                s[a][threadIdx.x] = fminf(s[a][threadIdx.x], s[a][threadIdx.x + st]);
                s[3 + a][threadIdx.x] = fmaxf(s[3 + a][threadIdx.x], s[3 + a][threadIdx.x + st]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) out6[6 * blockIdx.x + threadIdx.x] = s[threadIdx.x][0];
}

__device__ __forceinline__ void point_cell(const GridDesc &g, float x, float y, float z, int &cx,
                                           int &cy, int &cz)
{
    cx = min(max((int)floorf((x - g.ox) * g.inv_h), 0), g.nx - 1);
    cy = min(max((int)floorf((y - g.oy) * g.inv_h), 0), g.ny - 1);
    cz = min(max((int)floorf((z - g.oz) * g.inv_h), 0), g.nz - 1);
}

__global__ void cell_key_kernel(GridDesc g, const float *__restrict__ aos, int m,
                                unsigned long long *__restrict__ keys, int *__restrict__ vals)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int cx, cy, cz;
    point_cell(g, aos[3 * (size_t)i], aos[3 * (size_t)i + 1], aos[3 * (size_t)i + 2], cx, cy, cz);
    keys[i] = ((unsigned long long)cz * g.ny + cy) * g.nx + cx;
    vals[i] = i;
}

// marks the block of every occupied cell and of its x-predecessor
__global__ void mark_blocks_kernel(GridDesc g, const unsigned long long *__restrict__ keys, int m,
                                   int *__restrict__ marks)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    unsigned long long key = keys[i];
    if (i > 0 && keys[i - 1] == key) return;
    int cx = (int)(key % g.nx);
    unsigned long long rest = key / g.nx;
    int cy = (int)(rest % g.ny), cz = (int)(rest / g.ny);
    marks[dir_index(g, cx, cy, cz)] = 1;
    if (cx > 0) marks[dir_index(g, cx - 1, cy, cz)] = 1;
}

__global__ void finish_dir_kernel(const int *__restrict__ marks, const int *__restrict__ ranks,
                                  int nblocks_total, int *__restrict__ dir,
                                  int *__restrict__ block_of_rank)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks_total) return;
    if (marks[b]) { dir[b] = ranks[b]; block_of_rank[ranks[b]] = b; }
    else dir[b] = -1;
}

__device__ __forceinline__ unsigned lower_bound_key(const unsigned long long *keys, int m,
                                                    unsigned long long key)
{
    int lo = 0, hi = m;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (unsigned)lo;
}

// one thread per cell of every ranked block
__global__ void fill_cells_kernel(GridDesc g, const unsigned long long *__restrict__ keys, int m,
                                  const int *__restrict__ block_of_rank, int nranked,
                                  uint2 *__restrict__ run2, unsigned *__restrict__ end1)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nranked * 64) return;
    int rank = (int)(t >> 6), local = (int)(t & 63);
    int b = block_of_rank[rank];
    int bx = b % g.nbx, by = (b / g.nbx) % g.nby, bz = b / (g.nbx * g.nby);
    int cx = bx * 4 + (local & 3), cy = by * 4 + ((local >> 2) & 3), cz = bz * 4 + (local >> 4);
    unsigned s = 0, e1 = 0, e2 = 0;
    if (cx < g.nx && cy < g.ny && cz < g.nz) {
        unsigned long long key = ((unsigned long long)cz * g.ny + cy) * g.nx + cx;
        s = lower_bound_key(keys, m, key);
        e1 = lower_bound_key(keys, m, key + 1);
        e2 = (cx + 1 < g.nx) ? lower_bound_key(keys, m, key + 2) : e1;
    }
    run2[t] = make_uint2(s, e2);
    end1[t] = e1;
}

__global__ void gather_target_kernel(const float *__restrict__ aos, const float *__restrict__ nrm_aos,
                                     const int *__restrict__ vals, int m, float4 *__restrict__ pts,
                                     float *__restrict__ nrm)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int o = vals[i];
    pts[i] = make_float4(aos[3 * (size_t)o], aos[3 * (size_t)o + 1], aos[3 * (size_t)o + 2],
                         __int_as_float(o));
    if (nrm_aos) {
        nrm[3 * (size_t)i] = nrm_aos[3 * (size_t)o];
        nrm[3 * (size_t)i + 1] = nrm_aos[3 * (size_t)o + 1];
        nrm[3 * (size_t)i + 2] = nrm_aos[3 * (size_t)o + 2];
    }
}

__device__ __forceinline__ unsigned long long spread21(unsigned v)
{
    unsigned long long x = v & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// Morton key of each source point over the source's own bounding box
__global__ void morton_key_kernel(const float *__restrict__ aos, int n, float ox, float oy,
                                  float oz, float scale, unsigned long long *__restrict__ keys,
                                  int *__restrict__ vals)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto q = [&](float v, float o) {
        float f = (v - o) * scale;
        f = fminf(fmaxf(f, 0.f), 2097151.f);
        return (unsigned)f;
    };
    keys[i] = spread21(q(aos[3 * (size_t)i], ox)) | (spread21(q(aos[3 * (size_t)i + 1], oy)) << 1) |
              (spread21(q(aos[3 * (size_t)i + 2], oz)) << 2);
    vals[i] = i;
}

__global__ void iota_kernel(int *__restrict__ vals, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vals[i] = i;
}

__global__ void aos_to_soa_kernel(const float *__restrict__ aos, const int *__restrict__ perm, int n,
                                  float *__restrict__ x, float *__restrict__ y, float *__restrict__ z)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t o = perm ? (size_t)perm[i] : (size_t)i;
    x[i] = aos[3 * o]; y[i] = aos[3 * o + 1]; z[i] = aos[3 * o + 2];
}

__global__ void soa_to_aos_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                  const float *__restrict__ z, const int *__restrict__ perm, int n,
                                  float *__restrict__ aos)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t o = perm ? (size_t)perm[i] : (size_t)i;
    aos[3 * o] = x[i]; aos[3 * o + 1] = y[i]; aos[3 * o + 2] = z[i];
}

}  // namespace svicp
