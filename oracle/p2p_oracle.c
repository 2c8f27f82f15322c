/*
 * CPU oracle of the point-to-point ICP (PointCloud/Registration/CPoint2PointICP.cpp).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ and by the checking legs of bench.py
 * and __graft_entry__.smoke(); the shipped library never links or calls it.
 *
 * PARITY UNPINNED: the reference holds no recorded output for this path and cannot
 * be built here (FLANN and Eigen are absent).  Eigen::JacobiSVD is replaced by a
 * one-sided Jacobi SVD; the rotation V S U^T it yields is the unique polar factor of
 * a non-singular W, so the choice of SVD algorithm only shows in the last bits.
 *
 * Two accumulation modes:
 *   faithful = 1: the weighted centroids and the weight sum are accumulated in fp32
 *                 in point order, as CPoint2PointICP.cpp:89-126 does.  An fp32 running
 *                 sum over n points carries a rounding error that grows with n and
 *                 depends on the order of the points.
 *   faithful = 0: the same fp32 terms are accumulated in fp64 and the centroid is
 *                 rounded to fp32 once.  This is what the GPU path computes (a
 *                 parallel reduction cannot reproduce a sequential fp32 sum).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_kdtree orc_kdtree;
void orc_kdtree_knn(const orc_kdtree *T, const float *q, int64_t n, int k, int32_t *idx_out,
                    float *d2_out, int threads);

/* One-sided (Hestenes) Jacobi: A V = U S with orthogonal V, columns of A V orthogonal.
 * Columns are ordered by descending norm, as Eigen orders singular values. */
void orc_svd3(const double W[9], double U[9], double S[3], double V[9])
{
    double A[9], Vm[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy(A, W, sizeof(A));
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, g = 0;
                for (int r = 0; r < 3; ++r) {
                    a += A[3 * r + p] * A[3 * r + p];
                    b += A[3 * r + q] * A[3 * r + q];
                    g += A[3 * r + p] * A[3 * r + q];
                }
                if (g == 0.0 || fabs(g) <= 1e-300) continue;
                if (fabs(g) > off) off = fabs(g) / sqrt(a * b > 0 ? a * b : 1.0);
                double zeta = (b - a) / (2.0 * g);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    double ap = A[3 * r + p], aq = A[3 * r + q];
                    A[3 * r + p] = c * ap - s * aq;
                    A[3 * r + q] = s * ap + c * aq;
                    double vp = Vm[3 * r + p], vq = Vm[3 * r + q];
                    Vm[3 * r + p] = c * vp - s * vq;
                    Vm[3 * r + q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double nrm[3];
    int order[3] = { 0, 1, 2 };
    for (int c = 0; c < 3; ++c)
        nrm[c] = sqrt(A[c] * A[c] + A[3 + c] * A[3 + c] + A[6 + c] * A[6 + c]);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (nrm[order[j]] < nrm[order[j + 1]]) { int t = order[j]; order[j] = order[j + 1]; order[j + 1] = t; }
    for (int c = 0; c < 3; ++c) {
        int o = order[c];
        S[c] = nrm[o];
        for (int r = 0; r < 3; ++r) {
            V[3 * r + c] = Vm[3 * r + o];
            U[3 * r + c] = nrm[o] > 0 ? A[3 * r + o] / nrm[o] : 0.0;
        }
    }
    /* a zero singular value leaves its U column undefined: complete it to a right-handed basis */
    if (!(S[2] > 0)) {
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

static double det3(const double M[9])
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* R = V diag(1, 1, sign det(V U^T)) U^T   (CPoint2PointICP.cpp:149-154) */
void orc_kabsch_rotation(const double W[9], double R[9])
{
    double U[9], S[3], V[9], VUt[9];
    orc_svd3(W, U, S, V);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += V[3 * i + k] * U[3 * j + k];
            VUt[3 * i + j] = s;
        }
    double sg = det3(VUt) < 0 ? -1.0 : 1.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = V[3 * i + 0] * U[3 * j + 0] + V[3 * i + 1] * U[3 * j + 1] +
                           sg * V[3 * i + 2] * U[3 * j + 2];
}

/* CPoint2PointICP::singleRun (:83-159).  info: {selected count, weight sum, sum w d2, 0}.
 * Returns 0 when nothing was selected (the reference then divides by zero). */
static int p2p_single_run(const orc_kdtree *T, const float *ref, const float *move, int64_t n,
                          double R[9], double Tv[3], double W9[9], int32_t *idx, float *d2,
                          double info[4], int faithful, int threads)
{
    orc_kdtree_knn(T, move, n, 1, idx, d2, threads);                       /* :85-87 */
    const float thr = 15.0f;                                               /* :90 */
    float *weight = (float *)calloc((size_t)(n > 0 ? n : 1), sizeof(float));
    float *selRef = (float *)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    float *selMove = (float *)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    float *selW = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) {                                      /* :94-103 */
        if (d2[i] <= thr) {
            float sl = d2[i] / thr;
            float ss = sl * sl;
            weight[i] = (float)((1.0 - (double)ss) * (1.0 - (double)ss));
        } else
            weight[i] = 0;
    }
    int count = 0;
    float fsum[7] = { 0, 0, 0, 0, 0, 0, 0 };      /* move xyz, ref xyz, weight: fp32 as :89-92 */
    double dsum[7] = { 0, 0, 0, 0, 0, 0, 0 };
    double wd2 = 0.0;
    for (int64_t i = 0; i < n; ++i) {                                      /* :104-124 */
        if (d2[i] < thr) {
            float term[7];
            for (int a = 0; a < 3; ++a) {
                selMove[count * 3 + a] = move[3 * i + a];
                selRef[count * 3 + a] = ref[3 * (size_t)idx[i] + a];
                term[a] = selMove[count * 3 + a] * weight[i];
                term[3 + a] = selRef[count * 3 + a] * weight[i];
            }
            term[6] = weight[i];
            for (int a = 0; a < 7; ++a) { fsum[a] += term[a]; dsum[a] += (double)term[a]; }
            selW[count] = weight[i];
            wd2 += (double)weight[i] * (double)d2[i];
            count++;
        }
    }
    float mu[6];                                                           /* :125-126 */
    if (faithful)
        for (int a = 0; a < 6; ++a) mu[a] = fsum[a] / fsum[6];
    else
        for (int a = 0; a < 6; ++a) mu[a] = (float)(dsum[a] / dsum[6]);
    double W[9] = { 0 };
    for (int k = 0; k < count; ++k) {                                      /* :128-147 */
        float mc[3], rc[3];
        for (int a = 0; a < 3; ++a) {
            mc[a] = selMove[k * 3 + a] - mu[a];
            rc[a] = selRef[k * 3 + a] - mu[3 + a];
        }
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {
                float prod = mc[j] * rc[i];
                float term = selW[k] * prod / count;                       /* :144, all fp32 */
                W[3 * j + i] += (double)term;
            }
    }
    info[0] = count; info[1] = faithful ? (double)fsum[6] : dsum[6]; info[2] = wd2; info[3] = 0;
    memcpy(W9, W, sizeof(W));
    free(weight); free(selRef); free(selMove); free(selW);
    if (count == 0) {
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        Tv[0] = Tv[1] = Tv[2] = 0.0;
        return 0;
    }
    orc_kabsch_rotation(W, R);                                             /* :149-154 */
    for (int r = 0; r < 3; ++r)                                            /* :155-158 */
        Tv[r] = (double)mu[3 + r] - (R[3 * r] * (double)mu[0] + R[3 * r + 1] * (double)mu[1] +
                                     R[3 * r + 2] * (double)mu[2]);
    return 1;
}

/* CPoint2PointICP::run (:50-81): src is updated in place iters - 1 times.
 * RT_hist: iters x 12 {R row-major, T}; W_hist: iters x 9; info_hist: iters x 4. */
int orc_icp_point2point(const orc_kdtree *T, const float *ref, float *src, int64_t n, int iters,
                        double *Trans16, double *RT_hist, double *W_hist, int32_t *idx_hist,
                        float *d2_hist, double *info_hist, int faithful, int threads)
{
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    float *d2 = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    double R[9], Tv[3], W[9], info[4];
    int bad = 0;
    for (int i = 0; i < 16; ++i) Trans16[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int k = 0; k < iters; ++k) {
        if (k > 0)
            for (int64_t j = 0; j < n; ++j) {                              /* :61-69 */
                float px = src[3 * j], py = src[3 * j + 1], pz = src[3 * j + 2];
                float x = (float)(R[0] * px + R[1] * py + R[2] * pz + Tv[0]);
                float y = (float)(R[3] * px + R[4] * py + R[5] * pz + Tv[1]);
                float z = (float)(R[6] * px + R[7] * py + R[8] * pz + Tv[2]);
                src[3 * j] = x; src[3 * j + 1] = y; src[3 * j + 2] = z;
            }
        if (!p2p_single_run(T, ref, src, n, R, Tv, W, idx, d2, info, faithful, threads)) bad = 1;
        double M[16] = { R[0], R[1], R[2], Tv[0], R[3], R[4], R[5], Tv[1],
                         R[6], R[7], R[8], Tv[2], 0, 0, 0, 1 }, out[16];
        for (int i = 0; i < 4; ++i)                                        /* :55-58, :71-74 */
            for (int j = 0; j < 4; ++j) {
                double s = 0;
                for (int q = 0; q < 4; ++q) s += M[4 * i + q] * Trans16[4 * q + j];
                out[4 * i + j] = s;
            }
        memcpy(Trans16, out, sizeof(out));
        if (RT_hist) { memcpy(RT_hist + 12 * k, R, sizeof(R)); memcpy(RT_hist + 12 * k + 9, Tv, sizeof(Tv)); }
        if (W_hist) memcpy(W_hist + 9 * k, W, sizeof(W));
        if (idx_hist) memcpy(idx_hist + (size_t)k * n, idx, sizeof(int32_t) * (size_t)n);
        if (d2_hist) memcpy(d2_hist + (size_t)k * n, d2, sizeof(float) * (size_t)n);
        if (info_hist) memcpy(info_hist + 4 * k, info, sizeof(info));
    }
    free(idx); free(d2);
    return bad;
}
