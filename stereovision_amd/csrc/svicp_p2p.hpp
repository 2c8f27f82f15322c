// Point-to-point ICP on the kernels of the point-to-plane path
// (PointCloud/Registration/CPoint2PointICP.cpp).  Per iteration:
//   p2p_match_kernel  update the points (:61-69), 1-NN (:85-87), weights (:94-103),
//                     weighted centroid sums, weight sum, selected count (:104-124)
//   p2p_cov_kernel    centre both point sets in fp32 and accumulate the 3x3
//                     cross-covariance (:128-147)
//   p2p_solve_kernel  rotation from the SVD of the covariance, translation (:149-158),
//                     accumulated transform (:55-58, :71-74)
// The reference accumulates the centroid sums in fp32 in point order; a parallel
// reduction cannot reproduce a sequential fp32 sum, so the same fp32 terms are summed
// in fp64 here and the centroid is rounded to fp32 once.
#pragma once
#include "svicp_device.hpp"

namespace svicp {

constexpr int kP2PTile = 1024;      // points per block, 4 per thread
constexpr int kP2PSums = 18;        // move xyz, ref xyz, weight, count, sum w d2, W (9)

// Block sum of NV per-thread doubles; the totals are added to out[0..NV) with one
// fp64 atomic per value by the first NV threads.  Callers spread the blocks over
// kDeltaCopies copies of `out`: same-address atomics serialise in L2.
template <int NV>
__device__ __forceinline__ void block_atomic_add(double (&v)[NV], double *out, double (*red)[32])
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) red[wv][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
        for (int w = 0; w < (int)blockDim.x / 64; ++w) s += red[w][threadIdx.x];
        if (s != 0.0) atomicAdd(out + threadIdx.x, s);
    }
}

// tp: per source point {matched target x, y, z, weight}; weight < 0 marks a point that
// was not selected (no neighbour with d2 < threshold).
template <bool APPLY>
__global__ __launch_bounds__(256, 4) void p2p_match_kernel(
    // the coordinate arrays carry no __restrict__: from the third iteration on the points are
    // updated where they are read (s?_in == s?_out)
    GridDesc g, const float *sx_in, const float *sy_in, const float *sz_in, float *sx_out,
    float *sy_out, float *sz_out, int n, const IcpState *__restrict__ st, float thr,
    float4 *__restrict__ tp, double *__restrict__ sums)
{
    __shared__ double red[4][32];
    const int nb = (int)gridDim.x;
    const long long tile0 = (long long)xcd_contiguous((int)blockIdx.x, nb) * kP2PTile;
    double R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, T[3] = { 0, 0, 0 };
    if (APPLY) {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = st->R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) T[k] = st->T[k];
    }
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    for (int sl = 0; sl < kP2PTile / 256; ++sl) {
        const long long i = tile0 + sl * 256 + (int)threadIdx.x;
        if (i >= n) continue;
        float px = sx_in[i], py = sy_in[i], pz = sz_in[i];
        if (APPLY) {
            // ((R0*x + R1*y) + R2*z) + T0 in fp64, one rounding to fp32 (:63-65)
            const double dx = (double)px, dy = (double)py, dz = (double)pz;
            px = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(R[0], dx), __dmul_rn(R[1], dy)), __dmul_rn(R[2], dz)), T[0]);
            py = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(R[3], dx), __dmul_rn(R[4], dy)), __dmul_rn(R[5], dz)), T[1]);
            pz = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(R[6], dx), __dmul_rn(R[7], dy)), __dmul_rn(R[8], dz)), T[2]);
            sx_out[i] = px; sy_out[i] = py; sz_out[i] = pz;
        }
        float best; int bpos, bidx;
        nn_capped<4, false>(g, px, py, pz, best, bpos, bidx, nullptr);
        float4 rec = make_float4(0.f, 0.f, 0.f, -1.f);
        if (bpos >= 0 && best < thr) {                      // strict, :106
            const float sl2 = best / thr;                   // :98
            const float ss = __fmul_rn(sl2, sl2);
            const float w = (float)((1.0 - (double)ss) * (1.0 - (double)ss));   // :99
            const float4 t = g.pts[bpos];
            rec = make_float4(t.x, t.y, t.z, w);
            acc[0] += (double)__fmul_rn(px, w); acc[1] += (double)__fmul_rn(py, w);   // :111-113
            acc[2] += (double)__fmul_rn(pz, w);
            acc[3] += (double)__fmul_rn(t.x, w); acc[4] += (double)__fmul_rn(t.y, w); // :117-119
            acc[5] += (double)__fmul_rn(t.z, w);
            acc[6] += (double)w;                                                      // :121
            acc[7] += 1.0;                                                            // :122
            acc[8] += (double)w * (double)best;
        }
        tp[i] = rec;
    }
    block_atomic_add<9>(acc, sums + ((int)blockIdx.x & (kDeltaCopies - 1)) * kNumSums, red);
}

// Sum t of an iteration: the kDeltaCopies copies added in a fixed order.
__device__ __forceinline__ double p2p_sum(const double *__restrict__ sums, int t)
{
    double s = 0.0;
#pragma unroll
    for (int cp = 0; cp < kDeltaCopies; ++cp) s += sums[cp * kNumSums + t];
    return s;
}

__global__ __launch_bounds__(256, 4) void p2p_cov_kernel(
    const float *__restrict__ sx, const float *__restrict__ sy, const float *__restrict__ sz, int n,
    const float4 *__restrict__ tp, double *__restrict__ sums)
{
    __shared__ double red[4][32];
    const int nb = (int)gridDim.x;
    const long long tile0 = (long long)xcd_contiguous((int)blockIdx.x, nb) * kP2PTile;
    const double sw = p2p_sum(sums, 6);
    // centroids rounded to fp32 once (:125-126); count as the reference's float divisor (:144)
    const float mmx = (float)(p2p_sum(sums, 0) / sw), mmy = (float)(p2p_sum(sums, 1) / sw),
                mmz = (float)(p2p_sum(sums, 2) / sw);
    const float mrx = (float)(p2p_sum(sums, 3) / sw), mry = (float)(p2p_sum(sums, 4) / sw),
                mrz = (float)(p2p_sum(sums, 5) / sw);
    const float fcount = (float)p2p_sum(sums, 7);
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    for (int sl = 0; sl < kP2PTile / 256; ++sl) {
        const long long i = tile0 + sl * 256 + (int)threadIdx.x;
        if (i >= n) continue;
        const float4 r = tp[i];
        if (r.w < 0.f) continue;
        const float mc[3] = { sx[i] - mmx, sy[i] - mmy, sz[i] - mmz };   // :133-135
        const float rc[3] = { r.x - mrx, r.y - mry, r.z - mrz };         // :130-132
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float prod = __fmul_rn(mc[j], rc[c]);
                const float term = __fdiv_rn(__fmul_rn(r.w, prod), fcount);   // :144, all fp32
                acc[3 * j + c] += (double)term;
            }
    }
    block_atomic_add<9>(acc, sums + ((int)blockIdx.x & (kDeltaCopies - 1)) * kNumSums + 9, red);
}

// One-sided Jacobi SVD of a 3x3 matrix, columns ordered by descending singular
// value; R = V diag(1, 1, sign det(V U^T)) U^T (CPoint2PointICP.cpp:149-154).
__device__ inline void kabsch_rotation(const double W[9], double R[9])
{
    double A[9], V[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int k = 0; k < 9; ++k) A[k] = W[k];
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, gm = 0;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    a += A[3 * r + p] * A[3 * r + p];
                    b += A[3 * r + q] * A[3 * r + q];
                    gm += A[3 * r + p] * A[3 * r + q];
                }
                if (fabs(gm) <= 1e-300) continue;
                const double rel = fabs(gm) / sqrt(a * b > 0 ? a * b : 1.0);
                if (rel > off) off = rel;
                const double zeta = (b - a) / (2.0 * gm);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double ap = A[3 * r + p], aq = A[3 * r + q];
                    A[3 * r + p] = c * ap - s * aq;
                    A[3 * r + q] = s * ap + c * aq;
                    const double vp = V[3 * r + p], vq = V[3 * r + q];
                    V[3 * r + p] = c * vp - s * vq;
                    V[3 * r + q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double nrm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) nrm[c] = sqrt(A[c] * A[c] + A[3 + c] * A[3 + c] + A[6 + c] * A[6 + c]);
    // columns by descending norm: the exchanges of a bubble sort, on the columns themselves
    // (constant indices only: an index array would put A and V in scratch memory)
    auto order = [&](int i, int j) {
        if (nrm[i] < nrm[j]) {
            double t = nrm[i]; nrm[i] = nrm[j]; nrm[j] = t;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                t = A[3 * r + i]; A[3 * r + i] = A[3 * r + j]; A[3 * r + j] = t;
                t = V[3 * r + i]; V[3 * r + i] = V[3 * r + j]; V[3 * r + j] = t;
            }
        }
    };
    order(0, 1); order(1, 2); order(0, 1);
    double U[9], Vs[9];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            Vs[3 * r + c] = V[3 * r + c];
            U[3 * r + c] = nrm[c] > 0 ? A[3 * r + c] / nrm[c] : 0.0;
        }
    if (!(nrm[2] > 0)) {     // complete U to a right-handed basis
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
    double M[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            M[3 * i + j] = Vs[3 * i] * U[3 * j] + Vs[3 * i + 1] * U[3 * j + 1] + Vs[3 * i + 2] * U[3 * j + 2];
    const double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
                       M[2] * (M[3] * M[7] - M[4] * M[6]);
    const double sg = det < 0 ? -1.0 : 1.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = Vs[3 * i] * U[3 * j] + Vs[3 * i + 1] * U[3 * j + 1] + sg * Vs[3 * i + 2] * U[3 * j + 2];
}

// rt_hist: 12 doubles per iteration {R row-major, T}; info_hist: {count, sum w, sum w d2, step}.
__global__ void p2p_solve_kernel(const double *__restrict__ sum_copies, IcpState *st, double *rt_hist,
                                 double *info_hist, int iter)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double sums[kP2PSums];
    for (int t = 0; t < kP2PSums; ++t) sums[t] = p2p_sum(sum_copies, t);
    double R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, T[3] = { 0, 0, 0 };
    const double count = sums[7], sw = sums[6];
    if (count > 0.0 && sw > 0.0) {
        double W[9];
        for (int k = 0; k < 9; ++k) W[k] = sums[9 + k];
        kabsch_rotation(W, R);
        const double mm[3] = { (double)(float)(sums[0] / sw), (double)(float)(sums[1] / sw), (double)(float)(sums[2] / sw) };
        const double mr[3] = { (double)(float)(sums[3] / sw), (double)(float)(sums[4] / sw), (double)(float)(sums[5] / sw) };
        for (int r = 0; r < 3; ++r)     // T = A - R B (:155-158)
            T[r] = mr[r] - (R[3 * r] * mm[0] + R[3 * r + 1] * mm[1] + R[3 * r + 2] * mm[2]);
    } else {
        st->status = 4;                 // nothing selected: the reference divides by zero here
    }
    double Tr[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) {
                const double m = i < 3 ? (k < 3 ? R[3 * i + k] : T[i]) : (k == 3 ? 1.0 : 0.0);
                s += m * st->Trans[4 * k + j];
            }
            Tr[4 * i + j] = s;
        }
    for (int k = 0; k < 16; ++k) st->Trans[k] = Tr[k];
    for (int k = 0; k < 9; ++k) { st->R[k] = R[k]; rt_hist[12 * iter + k] = R[k]; }
    for (int k = 0; k < 3; ++k) { st->T[k] = T[k]; rt_hist[12 * iter + 9 + k] = T[k]; }
    double ct = 0.5 * (R[0] + R[4] + R[8] - 1.0);
    ct = ct > 1.0 ? 1.0 : (ct < -1.0 ? -1.0 : ct);
    const double ang = acos(ct);
    const double step = sqrt(ang * ang + T[0] * T[0] + T[1] * T[1] + T[2] * T[2]);
    info_hist[4 * iter + 0] = count; info_hist[4 * iter + 1] = sums[8];
    info_hist[4 * iter + 2] = step; info_hist[4 * iter + 3] = sw;
    for (int k = 0; k < kNumSums; ++k) st->S[k] = 0.0;
    st->S[27] = count; st->S[28] = sums[8]; st->S[29] = sw; st->S[30] = step;
}

}  // namespace svicp
