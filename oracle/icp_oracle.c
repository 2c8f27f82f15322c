/*
 * CPU oracle for the point-to-plane ICP path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the reference algorithm.  It exists
 * to check the HIP path and to be timed as the CPU baseline; nothing in the
 * shipped library calls it.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or recorded
 * outputs for this path, and it cannot be built here because its FLANN and
 * Eigen dependencies (PointCloud/3rdparty/3rdparty.zip) are absent.  The
 * restatement follows the reference source line by line for everything the
 * reference itself computes and restates the published behaviour of the two
 * third-party calls it makes:
 *   - flann::Index<L2<float>>(KDTREE_SINGLE)::knnSearch (FLANN 1.8/1.9):
 *     exact k-NN, squared L2 in fp32 accumulated x, y, z from zero, results
 *     ascending.  FLANN's order among exactly equal distances depends on its
 *     tree traversal; here ties go to the lowest point index.
 *   - Eigen MatrixXd::inverse() (partial-pivot LU), AngleAxisd products,
 *     SelfAdjointEigenSolver (eigenvector of the smallest eigenvalue).
 * It is cross-checked in tests/ against independent methods (brute force,
 * scipy cKDTree, numpy.linalg).
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction, so fp32 distances are
 * the same bits as the device path's).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* squared L2 distance as flann::L2<float> computes it for dim 3:     */
/* result = 0; result += d0*d0; result += d1*d1; result += d2*d2      */
/* (CKNNSearch.cpp:33 instantiates flann::L2<float>)                  */
/* ------------------------------------------------------------------ */
static inline float dist2_f32(const float *a, const float *b)
{
    float d0 = a[0] - b[0];
    float d1 = a[1] - b[1];
    float d2 = a[2] - b[2];
    float r = d0 * d0;
    r = r + d1 * d1;
    r = r + d2 * d2;
    return r;
}

/* (d2, idx) ordering used everywhere: ascending distance, then ascending index */
static inline int better(float d2, int32_t idx, float bd2, int32_t bidx)
{
    return (d2 < bd2) || (d2 == bd2 && idx < bidx);
}

/* bounded sorted list of the k best (d2, idx) pairs */
typedef struct {
    int k, n;
    float *d2;
    int32_t *idx;
} knn_list;

static inline void knn_list_insert(knn_list *L, float d2, int32_t idx)
{
    if (L->n == L->k && !better(d2, idx, L->d2[L->k - 1], L->idx[L->k - 1]))
        return;
    int pos = (L->n < L->k) ? L->n : L->k - 1;
    while (pos > 0 && better(d2, idx, L->d2[pos - 1], L->idx[pos - 1])) {
        L->d2[pos] = L->d2[pos - 1];
        L->idx[pos] = L->idx[pos - 1];
        --pos;
    }
    L->d2[pos] = d2;
    L->idx[pos] = idx;
    if (L->n < L->k)
        L->n++;
}

/* ------------------------------------------------------------------ */
/* brute-force k-NN: the definition the kd-tree and the GPU must match */
/* ------------------------------------------------------------------ */
void orc_knn_brute(const float *ref, int64_t m, const float *q, int64_t n, int k,
                   int32_t *idx_out, float *d2_out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        knn_list L = { k, 0, d2_out + i * k, idx_out + i * k };
        for (int64_t j = 0; j < m; ++j)
            knn_list_insert(&L, dist2_f32(q + 3 * i, ref + 3 * j), (int32_t)j);
        for (int j = L.n; j < k; ++j) {
            L.d2[j] = INFINITY;
            L.idx[j] = -1;
        }
    }
}

/* ------------------------------------------------------------------ */
/* exact kd-tree (stands in for flann KDTREE_SINGLE, CKNNSearch.cpp:27-35) */
/* ------------------------------------------------------------------ */
#define ORC_LEAF 10

typedef struct {
    int32_t left, right;   /* children, or -1 for a leaf          */
    int32_t lo, hi;        /* leaf: range in the permuted arrays  */
    int32_t dim;
    float divlow, divhigh; /* max of left / min of right along dim */
} kd_node;

typedef struct orc_kdtree {
    int64_t m;
    float *pts;        /* permuted copy, 3 floats per point */
    int32_t *ids;      /* original index of each permuted point */
    kd_node *nodes;
    int32_t n_nodes, cap_nodes;
} orc_kdtree;

static void kd_select(float *pts, int32_t *ids, int32_t lo, int32_t hi, int32_t nth, int dim)
{
    /* quickselect on [lo, hi): afterwards pts[nth] is in sorted position along dim */
    while (hi - lo > 1) {
        int32_t mid = lo + (hi - lo) / 2;
        float a = pts[3 * lo + dim], b = pts[3 * mid + dim], c = pts[3 * (hi - 1) + dim];
        float pivot = (a < b) ? ((b < c) ? b : (a < c ? c : a)) : ((a < c) ? a : (b < c ? c : b));
        int32_t i = lo, j = hi - 1;
        while (i <= j) {
            while (pts[3 * i + dim] < pivot) ++i;
            while (pts[3 * j + dim] > pivot) --j;
            if (i <= j) {
                float t0 = pts[3 * i], t1 = pts[3 * i + 1], t2 = pts[3 * i + 2];
                pts[3 * i] = pts[3 * j]; pts[3 * i + 1] = pts[3 * j + 1]; pts[3 * i + 2] = pts[3 * j + 2];
                pts[3 * j] = t0; pts[3 * j + 1] = t1; pts[3 * j + 2] = t2;
                int32_t t = ids[i]; ids[i] = ids[j]; ids[j] = t;
                ++i; --j;
            }
        }
        if (nth <= j) hi = j + 1;
        else if (nth >= i) lo = i;
        else return;
    }
}

static int32_t kd_build_rec(orc_kdtree *T, int32_t lo, int32_t hi)
{
    if (T->n_nodes == T->cap_nodes) {
        T->cap_nodes *= 2;
        T->nodes = (kd_node *)realloc(T->nodes, sizeof(kd_node) * (size_t)T->cap_nodes);
    }
    int32_t me = T->n_nodes++;
    kd_node nd;
    nd.left = nd.right = -1;
    nd.lo = lo; nd.hi = hi; nd.dim = 0; nd.divlow = nd.divhigh = 0.f;
    if (hi - lo > ORC_LEAF) {
        float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        for (int32_t i = lo; i < hi; ++i)
            for (int d = 0; d < 3; ++d) {
                float v = T->pts[3 * i + d];
                if (v < mn[d]) mn[d] = v;
                if (v > mx[d]) mx[d] = v;
            }
        int dim = 0;
        if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
        if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
        if (mx[dim] > mn[dim]) {
            int32_t mid = lo + (hi - lo) / 2;
            kd_select(T->pts, T->ids, lo, hi, mid, dim);
            float dl = -FLT_MAX, dh = FLT_MAX;
            for (int32_t i = lo; i < mid; ++i) if (T->pts[3 * i + dim] > dl) dl = T->pts[3 * i + dim];
            for (int32_t i = mid; i < hi; ++i) if (T->pts[3 * i + dim] < dh) dh = T->pts[3 * i + dim];
            nd.dim = dim; nd.divlow = dl; nd.divhigh = dh;
            T->nodes[me] = nd;
            int32_t l = kd_build_rec(T, lo, mid);
            int32_t r = kd_build_rec(T, mid, hi);
            T->nodes[me].left = l;
            T->nodes[me].right = r;
            return me;
        }
    }
    T->nodes[me] = nd;
    return me;
}

orc_kdtree *orc_kdtree_build(const float *xyz, int64_t m)
{
    orc_kdtree *T = (orc_kdtree *)calloc(1, sizeof(orc_kdtree));
    T->m = m;
    T->pts = (float *)malloc(sizeof(float) * 3 * (size_t)(m > 0 ? m : 1));
    T->ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m > 0 ? m : 1));
    memcpy(T->pts, xyz, sizeof(float) * 3 * (size_t)m);
    for (int64_t i = 0; i < m; ++i) T->ids[i] = (int32_t)i;
    T->cap_nodes = 1024;
    T->nodes = (kd_node *)malloc(sizeof(kd_node) * (size_t)T->cap_nodes);
    T->n_nodes = 0;
    if (m > 0) kd_build_rec(T, 0, (int32_t)m);
    return T;
}

void orc_kdtree_free(orc_kdtree *T)
{
    if (!T) return;
    free(T->pts); free(T->ids); free(T->nodes); free(T);
}

/* A far subtree is skipped only when its lower bound exceeds the current k-th
 * distance by more than fp32 rounding could explain, so the result equals the
 * brute-force definition above, ties included. */
static void kd_search_rec(const orc_kdtree *T, int32_t ni, const float *q, knn_list *L,
                          double mindist, double *dists)
{
    const kd_node *nd = &T->nodes[ni];
    if (nd->left < 0) {
        for (int32_t i = nd->lo; i < nd->hi; ++i)
            knn_list_insert(L, dist2_f32(q, T->pts + 3 * i), T->ids[i]);
        return;
    }
    int dim = nd->dim;
    double val = q[dim];
    double diff1 = val - (double)nd->divlow;
    double diff2 = val - (double)nd->divhigh;
    int32_t best, other;
    double cut;
    if (diff1 + diff2 < 0) { best = nd->left;  other = nd->right; cut = diff2 * diff2; }
    else                   { best = nd->right; other = nd->left;  cut = diff1 * diff1; }
    kd_search_rec(T, best, q, L, mindist, dists);
    double saved = dists[dim];
    double md = mindist + cut - saved;
    dists[dim] = cut;
    double worst = (L->n == L->k) ? (double)L->d2[L->k - 1] : INFINITY;
    if (md * (1.0 - 1e-5) <= worst)
        kd_search_rec(T, other, q, L, md, dists);
    dists[dim] = saved;
}

/* CKNNSearch::getKNeighbour (CKNNSearch.cpp:67-71): idx N x k int32 row-major,
 * squared distances N x k fp32, ascending. */
void orc_kdtree_knn(const orc_kdtree *T, const float *q, int64_t n, int k,
                    int32_t *idx_out, float *d2_out, int threads)
{
    (void)threads;
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
#endif
    for (int64_t i = 0; i < n; ++i) {
        knn_list L = { k, 0, d2_out + i * k, idx_out + i * k };
        double dists[3] = { 0, 0, 0 };
        if (T->m > 0) kd_search_rec(T, 0, q + 3 * i, &L, 0.0, dists);
        for (int j = L.n; j < k; ++j) {
            L.d2[j] = INFINITY;
            L.idx[j] = -1;
        }
    }
}

/* ------------------------------------------------------------------ */
/* 3x3 symmetric eigen-decomposition (cyclic Jacobi), fp64.           */
/* Stands in for Eigen::SelfAdjointEigenSolver (CPoint2PlaneICP.cpp:148): */
/* returns the unit eigenvector of the smallest eigenvalue.           */
/* ------------------------------------------------------------------ */
void orc_eig3_smallest(const double Win[9], double v_out[3], double *eval_out)
{
    double A[3][3], V[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            A[i][j] = 0.5 * (Win[3 * i + j] + Win[3 * j + i]);
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-300 || off <= 1e-34 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int r = p + 1; r < 3; ++r) {
                if (A[p][r] == 0.0) continue;
                double theta = (A[r][r] - A[p][p]) / (2.0 * A[p][r]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = A[k][p], akr = A[k][r];
                    A[k][p] = c * akp - s * akr;
                    A[k][r] = s * akp + c * akr;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = A[p][k], ark = A[r][k];
                    A[p][k] = c * apk - s * ark;
                    A[r][k] = s * apk + c * ark;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k][p], vkr = V[k][r];
                    V[k][p] = c * vkp - s * vkr;
                    V[k][r] = s * vkp + c * vkr;
                }
            }
    }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    double nrm = sqrt(V[0][m] * V[0][m] + V[1][m] * V[1][m] + V[2][m] * V[2][m]);
    for (int k = 0; k < 3; ++k) v_out[k] = V[k][m] / nrm;
    if (eval_out) *eval_out = A[m][m];
}

/* ------------------------------------------------------------------ */
/* CPoint2PlaneICP::calcNormals (CPoint2PlaneICP.cpp:122-154)          */
/* The reference hard-codes these two numbers; the library exposes them as options,
 * so the checker can be told the same values (defaults: the reference's). */
static float g_dist_threshold = 5.0f;   /* CPoint2PlaneICP.cpp:68  */
static int g_normals_k = 20;            /* CPoint2PlaneICP.cpp:125 */
void orc_set_constants(float dist_threshold, int normals_k)
{
    g_dist_threshold = dist_threshold;
    g_normals_k = normals_k;
}

/* Sensitivity studies only (tests/test_oracle.py): every solve's X is displaced by
 * +-amplitude per component, signs drawn from `seed`, before R and T are formed from it.
 * It stands for the last bits in which Eigen's inverse() (CPoint2PlaneICP.cpp:109) may
 * differ from this file's elimination.  amplitude = 0 (the default) changes nothing. */
static double g_solve_noise = 0.0;
static uint64_t g_solve_noise_state = 0;
void orc_set_solve_perturbation(double amplitude, uint64_t seed)
{
    g_solve_noise = amplitude;
    g_solve_noise_state = seed;
}

/* 20-NN of each reference point including itself; raw second moments  */
/* about the origin, each term fl32(fl32(pj*pi)/20) summed in fp64;    */
/* eigenvector of the smallest eigenvalue stored as fp32.              */
/* ------------------------------------------------------------------ */
void orc_calc_normals(const orc_kdtree *T, const float *ref, int64_t m, float *normals,
                      int32_t *knn_idx_out /* m*20 or NULL */, int threads)
{
    const int K = g_normals_k;                                       /* 20 at :125 */
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)m * K);
    float *d2 = (float *)malloc(sizeof(float) * (size_t)m * K);
    orc_kdtree_knn(T, ref, m, K, idx, d2, threads);
    for (int64_t p = 0; p < m; ++p) {
        double W[9] = { 0 };
        for (int k = 0; k < K; ++k) {
            int32_t id = idx[p * K + k];
            if (id < 0) continue;
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i) {
                    float pw = 1.0f;                                  /* pWeight[k] :135 */
                    float prod = ref[3 * id + j] * ref[3 * id + i];   /* float * float   */
                    float term = pw * prod / K;                       /* :144, all fp32  */
                    W[3 * j + i] += (double)term;
                }
        }
        double v[3];
        orc_eig3_smallest(W, v, NULL);
        normals[3 * p + 0] = (float)v[0];
        normals[3 * p + 1] = (float)v[1];
        normals[3 * p + 2] = (float)v[2];
    }
    if (knn_idx_out) memcpy(knn_idx_out, idx, sizeof(int32_t) * (size_t)m * K);
    free(idx); free(d2);
}

/* ------------------------------------------------------------------ */
/* 6x6 inverse by partial-pivot LU (Eigen MatrixXd::inverse for a      */
/* dynamic 6x6, CPoint2PlaneICP.cpp:109).  Returns 0 on a zero pivot.  */
/* ------------------------------------------------------------------ */
static int inverse6(const double *Ain, double *inv)
{
    double A[6][12];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            A[i][j] = Ain[6 * i + j];
            A[i][6 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        for (int r = c + 1; r < 6; ++r)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (A[piv][c] == 0.0 || !isfinite(A[piv][c])) return 0;
        if (piv != c)
            for (int j = 0; j < 12; ++j) { double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
        for (int r = c + 1; r < 6; ++r) {
            double f = A[r][c] / A[c][c];
            if (f != 0.0)
                for (int j = c; j < 12; ++j) A[r][j] -= f * A[c][j];
        }
    }
    for (int c = 5; c >= 0; --c) {
        for (int j = 6; j < 12; ++j) A[c][j] /= A[c][c];
        for (int r = 0; r < c; ++r) {
            double f = A[r][c];
            if (f != 0.0)
                for (int j = 6; j < 12; ++j) A[r][j] -= f * A[c][j];
        }
    }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) inv[6 * i + j] = A[i][6 + j];
    return 1;
}

/* ------------------------------------------------------------------ */
/* CPoint2PlaneICP::singleRun (CPoint2PlaneICP.cpp:63-120)             */
/* info[4] = { inlier count (weight > 0), sum w*b*b, |X|, sum w }      */
/* faithful != 0 also forms inv*AW (6xN) before multiplying by B, as   */
/* the reference's expression order does; 0 uses inv*(AW*B).           */
/* ------------------------------------------------------------------ */
static int single_run(const orc_kdtree *T, const float *ref, const float *normals,
                      const float *move, int64_t n, double R[9], double Tv[3], double X[6],
                      int32_t *idx, float *d2, double *info, int faithful, int threads)
{
    orc_kdtree_knn(T, move, n, 1, idx, d2, threads);           /* :65-67 */

    const float distthreshold = g_dist_threshold;              /* 5.0f at :68 */
    float *weight = (float *)calloc((size_t)(n > 0 ? n : 1), sizeof(float));
    for (int64_t i = 0; i < n; ++i) {                          /* :70-79 */
        if (d2[i] <= distthreshold) {
            float sl = d2[i] / distthreshold;
            float ss = sl * sl;
            weight[i] = (float)((1.0 - (double)ss) * (1.0 - (double)ss));
        } else
            weight[i] = 0;
    }

    double *A = (double *)malloc(sizeof(double) * 6 * (size_t)(n > 0 ? n : 1));   /* N x 6 */
    double *AW = (double *)malloc(sizeof(double) * 6 * (size_t)(n > 0 ? n : 1));  /* 6 x N */
    double *B = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));

    for (int64_t i = 0; i < n; ++i) {                          /* :85-99 */
        int32_t id = idx[i];
        float nx = normals[3 * id], ny = normals[3 * id + 1], nz = normals[3 * id + 2];
        float sx = move[3 * i], sy = move[3 * i + 1], sz = move[3 * i + 2];
        float dx = ref[3 * id], dy = ref[3 * id + 1], dz = ref[3 * id + 2];
        float a0 = sy * nz - sz * ny;
        float a1 = sz * nx - sx * nz;
        float a2 = sx * ny - sy * nx;
        float b = (dx - sx) * nx + (dy - sy) * ny + (dz - sz) * nz;
        A[6 * i + 0] = a0; A[6 * i + 1] = a1; A[6 * i + 2] = a2;
        A[6 * i + 3] = nx; A[6 * i + 4] = ny; A[6 * i + 5] = nz;
        B[i] = b;
    }
    for (int i = 0; i < 6; ++i)                                /* :101-107 */
        for (int64_t j = 0; j < n; ++j)
            AW[(size_t)i * n + j] = A[6 * j + i] * weight[j];

    double JtJ[36], Jtr[6], inv[36];
    for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 6; ++k) {
            double s = 0;
            for (int64_t j = 0; j < n; ++j) s += AW[(size_t)i * n + j] * A[6 * j + k];
            JtJ[6 * i + k] = s;
        }
    int ok = inverse6(JtJ, inv);                               /* :109 */
    if (ok && faithful) {
        /* (inv * AW) * B, the reference's left-to-right order */
        double *M = (double *)malloc(sizeof(double) * 6 * (size_t)(n > 0 ? n : 1));
        for (int i = 0; i < 6; ++i)
            for (int64_t j = 0; j < n; ++j) {
                double s = 0;
                for (int k = 0; k < 6; ++k) s += inv[6 * i + k] * AW[(size_t)k * n + j];
                M[(size_t)i * n + j] = s;
            }
        for (int i = 0; i < 6; ++i) {
            double s = 0;
            for (int64_t j = 0; j < n; ++j) s += M[(size_t)i * n + j] * B[j];
            X[i] = s;
        }
        free(M);
    } else if (ok) {
        for (int i = 0; i < 6; ++i) {
            double s = 0;
            for (int64_t j = 0; j < n; ++j) s += AW[(size_t)i * n + j] * B[j];
            Jtr[i] = s;
        }
        for (int i = 0; i < 6; ++i) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += inv[6 * i + k] * Jtr[k];
            X[i] = s;
        }
    } else {
        for (int i = 0; i < 6; ++i) X[i] = NAN;
    }

    if (g_solve_noise != 0.0)
        for (int i = 0; i < 6; ++i) {
            g_solve_noise_state = g_solve_noise_state * 6364136223846793005ull + 1442695040888963407ull;
            X[i] += (g_solve_noise_state >> 63) ? g_solve_noise : -g_solve_noise;
        }

    if (info) {
        double cnt = 0, swr = 0, sw = 0;
        for (int64_t j = 0; j < n; ++j) {
            if (weight[j] > 0) cnt += 1;
            swr += (double)weight[j] * B[j] * B[j];
            sw += (double)weight[j];
        }
        info[0] = cnt; info[1] = swr;
        info[2] = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2] + X[3] * X[3] + X[4] * X[4] + X[5] * X[5]);
        info[3] = sw;
    }

    /* R = Rz(X2) * Ry(X1) * Rx(X0)  (yaw * pitch * roll, :111-115) */
    double cx = cos(X[0]), sx = sin(X[0]);
    double cy = cos(X[1]), sy = sin(X[1]);
    double cz = cos(X[2]), sz = sin(X[2]);
    R[0] = cz * cy; R[1] = cz * sy * sx - sz * cx; R[2] = cz * sy * cx + sz * sx;
    R[3] = sz * cy; R[4] = sz * sy * sx + cz * cx; R[5] = sz * sy * cx - cz * sx;
    R[6] = -sy;     R[7] = cy * sx;                R[8] = cy * cx;
    Tv[0] = X[3]; Tv[1] = X[4]; Tv[2] = X[5];                  /* :118-119 */

    free(weight); free(A); free(AW); free(B);
    return ok;
}

static void trans_premul(double *Trans, const double R[9], const double Tv[3])
{
    /* Trans = [R T; 0 1] * Trans   (:36-39, :52-55) */
    double M[16] = { R[0], R[1], R[2], Tv[0], R[3], R[4], R[5], Tv[1],
                     R[6], R[7], R[8], Tv[2], 0, 0, 0, 1 };
    double out[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += M[4 * i + k] * Trans[4 * k + j];
            out[4 * i + j] = s;
        }
    memcpy(Trans, out, sizeof(out));
}

/* ------------------------------------------------------------------ */
/* CPoint2PlaneICP::run (CPoint2PlaneICP.cpp:31-62)                    */
/* src is updated in place (count-1 times), as the reference does.     */
/* X_hist: iters*6, idx_hist/d2_hist: iters*n or NULL, info_hist: iters*4 or NULL */
/* Returns 0 when every solve succeeded.                               */
/* ------------------------------------------------------------------ */
int orc_icp_point2plane(const orc_kdtree *T, const float *ref, const float *normals,
                        float *src, int64_t n, int iters, double *Trans16,
                        double *X_hist, int32_t *idx_hist, float *d2_hist, double *info_hist,
                        int faithful, int threads)
{
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    float *d2 = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    double R[9], Tv[3], X[6], info[4];
    int bad = 0;
    for (int i = 0; i < 16; ++i) Trans16[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int k = 0; k < iters; ++k) {
        if (k > 0) {
            for (int64_t j = 0; j < n; ++j) {                  /* :42-50 */
                float px = src[3 * j], py = src[3 * j + 1], pz = src[3 * j + 2];
                float x = (float)(R[0] * px + R[1] * py + R[2] * pz + Tv[0]);
                float y = (float)(R[3] * px + R[4] * py + R[5] * pz + Tv[1]);
                float z = (float)(R[6] * px + R[7] * py + R[8] * pz + Tv[2]);
                src[3 * j] = x; src[3 * j + 1] = y; src[3 * j + 2] = z;
            }
        }
        if (!single_run(T, ref, normals, src, n, R, Tv, X, idx, d2, info, faithful, threads))
            bad = 1;
        trans_premul(Trans16, R, Tv);
        if (X_hist) memcpy(X_hist + 6 * k, X, sizeof(X));
        if (idx_hist) memcpy(idx_hist + (size_t)k * n, idx, sizeof(int32_t) * (size_t)n);
        if (d2_hist) memcpy(d2_hist + (size_t)k * n, d2, sizeof(float) * (size_t)n);
        if (info_hist) memcpy(info_hist + 4 * k, info, sizeof(info));
    }
    free(idx); free(d2);
    return bad;
}

/* CTxtReader::split + read (CTxtReader.cpp:13-61): space-separated tokens,
 * atof -> float, rows = tokens / 3.  Returns the token count; out may be NULL
 * to size the buffer. */
int64_t orc_parse_txt(const char *text, int64_t len, float *out, int64_t cap)
{
    int64_t count = 0, pos = 0;
    while (pos < len) {
        int64_t eol = pos;
        while (eol < len && text[eol] != '\n') ++eol;
        /* one line: [pos, eol) */
        int64_t off = pos;
        while (off < eol) {
            int64_t sp = off;
            while (sp < eol && text[sp] != ' ') ++sp;
            if (sp > off || sp == eol) {
                if (sp > off) {
                    char buf[128];
                    int64_t l = sp - off;
                    if (l > 127) l = 127;
                    memcpy(buf, text + off, (size_t)l);
                    buf[l] = 0;
                    if (out && count < cap) out[count] = (float)atof(buf);
                    ++count;
                }
            }
            off = sp + 1;
        }
        pos = eol + 1;
    }
    return count;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
