/*
 * CPU oracle for the calibration reprojection path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the residual / central-difference Jacobian evaluation
 * of the reference's stereo and monocular bundle refinement:
 *   Calibration/CStereoCameraCalibration.cpp:25-50   value
 *   Calibration/CStereoCameraCalibration.cpp:51-80   residual
 *   Calibration/CStereoCameraCalibration.cpp:81-247  jacobian (central differences)
 *   Calibration/CStereoCameraCalibration.cpp:478-581 parameter vector layout
 *   Calibration/CMonocularCameraCalibration.cpp:737-906 value2 / residual2 / jacobian2
 *   Common/Misc.hpp:16-32                             Rodrigues vector -> matrix
 *
 * PARITY UNPINNED: the reference records no expected outputs and cannot be
 * built here (it needs Eigen).  The dense functions follow the reference's loops;
 * the block-sparse ones compute the same numbers without forming the dense
 * Jacobian and are checked against the dense ones in tests/.
 *
 * Parameter vector (stereo, CStereoCameraCalibration.cpp:478-518), nI = 4 + k:
 *   [fxL fyL cxL cyL distL(k) | fxR fyR cxR cyR distR(k) | V x (rvec, t) | rvecS tS]
 * Monocular (skew forced to zero, five distortion terms,
 * CMonocularCameraCalibration.cpp:1045-1082):  [fx fy cx cy dist(5) | V x (rvec, t)]
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Common/Misc.hpp:16-32 */
void orc_rodrigues(const double r[3], double R[9])
{
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < 1.0e-6) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        return;
    }
    double a[3] = { r[0] / theta, r[1] / theta, r[2] / theta };
    double K[9] = { 0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0 };
    double K2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += K[3 * i + k] * K[3 * k + j];
            K2[3 * i + j] = s;
        }
    double sn = sin(theta), cs = 1.0 - cos(theta);
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + K[i] * sn + K2[i] * cs;
}

/* 3x4 extrinsic from (rvec, t) */
static void ext_from(const double *p6, double E[12])
{
    double R[9];
    orc_rodrigues(p6, R);
    for (int r = 0; r < 3; ++r) {
        E[4 * r + 0] = R[3 * r + 0]; E[4 * r + 1] = R[3 * r + 1]; E[4 * r + 2] = R[3 * r + 2];
        E[4 * r + 3] = p6[3 + r];
    }
}

/* right = pose(3x4) * [left; 0 0 0 1]   (CStereoCameraCalibration.cpp:573-579) */
static void ext_compose(const double S[12], const double L[12], double out[12])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += S[4 * r + k] * L[4 * k + c];
            s += S[4 * r + 3] * (c == 3 ? 1.0 : 0.0);
            out[4 * r + c] = s;
        }
}

/* CStereoCameraCalibration.cpp:25-50 (skew entry of the intrinsic matrix is 0
 * after deComposeStereoPara, :525-530).  intr = {fx, fy, cx, cy}. */
void orc_cal_value(const double intr[4], const double *dist, int k, const double E[12], double wx,
                   double wy, double out[2])
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
    out[0] = intr[0] * xe + 0.0 * ye + intr[2];
    out[1] = intr[1] * ye + intr[3];
}

typedef struct {
    int k, V;
    const double *intrL, *distL, *intrR, *distR;
    double *extL, *extR;   /* V x 12 each */
    double pose[12];
} stereo_paras;

/* deComposeStereoPara (CStereoCameraCalibration.cpp:520-581) */
static void stereo_decompose(const double *vec, int k, int V, stereo_paras *P)
{
    const int nI = 4 + k;
    P->k = k; P->V = V;
    P->intrL = vec; P->distL = vec + 4; P->intrR = vec + nI; P->distR = vec + nI + 4;
    for (int i = 0; i < V; ++i) ext_from(vec + 2 * nI + 6 * i, P->extL + 12 * i);
    ext_from(vec + 2 * nI + 6 * V, P->pose);
    for (int i = 0; i < V; ++i) ext_compose(P->pose, P->extL + 12 * i, P->extR + 12 * i);
}

/* residual (CStereoCameraCalibration.cpp:51-80): rows [left: view-major, corner,
 * (x, y)] then [right: same]; projected minus observed. */
void orc_stereo_residual(const double *vec, int k, int V, int C, const double *world_xy,
                         const double *left_xy, const double *right_xy, double *res)
{
    stereo_paras P;
    P.extL = (double *)malloc(sizeof(double) * 12 * (size_t)V);
    P.extR = (double *)malloc(sizeof(double) * 12 * (size_t)V);
    stereo_decompose(vec, k, V, &P);
    size_t loop = 0;
    for (int i = 0; i < V; ++i)
        for (int j = 0; j < C; ++j) {
            double v[2];
            orc_cal_value(P.intrL, P.distL, k, P.extL + 12 * i, world_xy[2 * j], world_xy[2 * j + 1], v);
            res[loop] = v[0] - left_xy[2 * ((size_t)i * C + j)];
            res[loop + 1] = v[1] - left_xy[2 * ((size_t)i * C + j) + 1];
            loop += 2;
        }
    for (int i = 0; i < V; ++i)
        for (int j = 0; j < C; ++j) {
            double v[2];
            orc_cal_value(P.intrR, P.distR, k, P.extR + 12 * i, world_xy[2 * j], world_xy[2 * j + 1], v);
            res[loop] = v[0] - right_xy[2 * ((size_t)i * C + j)];
            res[loop + 1] = v[1] - right_xy[2 * ((size_t)i * C + j) + 1];
            loop += 2;
        }
    free(P.extL); free(P.extR);
}

/* jacobian with update_intrinsics = true (CStereoCameraCalibration.cpp:95-186),
 * dense rows x cols, row-major.  Only feasible for small V. */
void orc_stereo_jacobian_dense(const double *vec, int k, int V, int C, const double *world_xy,
                               double *J)
{
    const double eps = 1.0e-6;
    const int nI = 4 + k, cols = 2 * nI + 6 * V + 6;
    const size_t rows = (size_t)V * C * 4;
    memset(J, 0, sizeof(double) * rows * cols);
    double *up = (double *)malloc(sizeof(double) * cols), *lo = (double *)malloc(sizeof(double) * cols);
    stereo_paras U, L;
    U.extL = (double *)malloc(sizeof(double) * 12 * (size_t)V);
    U.extR = (double *)malloc(sizeof(double) * 12 * (size_t)V);
    L.extL = (double *)malloc(sizeof(double) * 12 * (size_t)V);
    L.extR = (double *)malloc(sizeof(double) * 12 * (size_t)V);
    for (int c = 0; c < cols; ++c) {
        memcpy(up, vec, sizeof(double) * cols);
        memcpy(lo, vec, sizeof(double) * cols);
        up[c] += eps; lo[c] -= eps;
        stereo_decompose(up, k, V, &U);
        stereo_decompose(lo, k, V, &L);
        const int left_cols = c < nI || (c >= 2 * nI && c < cols - 6);
        for (int i = 0; i < V; ++i) {
            if (c >= 2 * nI && c < cols - 6 && (c - 2 * nI) / 6 != i) continue;   /* :151 */
            for (int j = 0; j < C; ++j) {
                double a[2], b[2];
                size_t row;
                if (left_cols) {
                    orc_cal_value(U.intrL, U.distL, k, U.extL + 12 * i, world_xy[2 * j], world_xy[2 * j + 1], a);
                    orc_cal_value(L.intrL, L.distL, k, L.extL + 12 * i, world_xy[2 * j], world_xy[2 * j + 1], b);
                    row = 2 * ((size_t)i * C + j);
                } else {
                    orc_cal_value(U.intrR, U.distR, k, U.extR + 12 * i, world_xy[2 * j], world_xy[2 * j + 1], a);
                    orc_cal_value(L.intrR, L.distR, k, L.extR + 12 * i, world_xy[2 * j], world_xy[2 * j + 1], b);
                    row = 2 * ((size_t)V * C) + 2 * ((size_t)i * C + j);
                }
                J[row * cols + c] = (a[0] - b[0]) / (2.0 * eps);
                J[(row + 1) * cols + c] = (a[1] - b[1]) / (2.0 * eps);
            }
        }
    }
    free(up); free(lo); free(U.extL); free(U.extR); free(L.extL); free(L.extR);
}

/* jacobian with update_intrinsics = false (CStereoCameraCalibration.cpp:187-245): only the
 * 6 V + 6 columns of the views' left extrinsics and of the stereo pose, perturbed at
 * offset 8 + 2 k of the parameter vector (:199-200).  Dense rows x (6 V + 6), row-major.
 * A view's columns fill that view's LEFT rows (:207-224), the pose's columns all RIGHT
 * rows (:226-240). */
void orc_stereo_jacobian_dense_fixed_intrinsics(const double *vec, int k, int V, int C,
                                                const double *world_xy, double *J)
{
    const double eps = 1.0e-6;
    const int nI = 4 + k, all = 2 * nI + 6 * V + 6, cols = 6 * V + 6, offset = 2 * nI;
    const size_t rows = (size_t)V * C * 4;
    memset(J, 0, sizeof(double) * rows * cols);
    double *up = (double *)malloc(sizeof(double) * all), *lo = (double *)malloc(sizeof(double) * all);
    stereo_paras U, L;
    U.extL = (double *)malloc(sizeof(double) * 12 * (size_t)V);
    U.extR = (double *)malloc(sizeof(double) * 12 * (size_t)V);
    L.extL = (double *)malloc(sizeof(double) * 12 * (size_t)V);
    L.extR = (double *)malloc(sizeof(double) * 12 * (size_t)V);
    for (int c = 0; c < cols; ++c) {
        memcpy(up, vec, sizeof(double) * all);
        memcpy(lo, vec, sizeof(double) * all);
        up[c + offset] += eps; lo[c + offset] -= eps;
        stereo_decompose(up, k, V, &U);
        stereo_decompose(lo, k, V, &L);
        for (int i = 0; i < V; ++i) {
            if (c < cols - 6 && c / 6 != i) continue;                           /* :210 */
            for (int j = 0; j < C; ++j) {
                double a[2], b[2];
                size_t row;
                if (c < cols - 6) {
                    orc_cal_value(U.intrL, U.distL, k, U.extL + 12 * i, world_xy[2 * j], world_xy[2 * j + 1], a);
                    orc_cal_value(L.intrL, L.distL, k, L.extL + 12 * i, world_xy[2 * j], world_xy[2 * j + 1], b);
                    row = 2 * ((size_t)i * C + j);
                } else {
                    orc_cal_value(U.intrR, U.distR, k, U.extR + 12 * i, world_xy[2 * j], world_xy[2 * j + 1], a);
                    orc_cal_value(L.intrR, L.distR, k, L.extR + 12 * i, world_xy[2 * j], world_xy[2 * j + 1], b);
                    row = 2 * ((size_t)V * C) + 2 * ((size_t)i * C + j);
                }
                J[row * cols + c] = (a[0] - b[0]) / (2.0 * eps);
                J[(row + 1) * cols + c] = (a[1] - b[1]) / (2.0 * eps);
            }
        }
    }
    free(up); free(lo); free(U.extL); free(U.extR); free(L.extL); free(L.extR);
}

/* Block-sparse J^T J and J^T e of the same Jacobian and residual, without the
 * dense matrix.  nI = 4 + k.  Outputs (row-major, all symmetric blocks full):
 *   LL nI x nI, RR nI x nI, RS nI x 6, SS 6 x 6, LE V x nI x 6, EE V x 6 x 6,
 *   gL nI, gR nI, gE V x 6, gS 6, res (4 V C, may be NULL), returns |e|^2. */
double orc_stereo_eval_sparse(const double *vec, int k, int V, int C, const double *world_xy,
                              const double *left_xy, const double *right_xy, double *LL,
                              double *RR, double *RS, double *SS, double *LE, double *EE,
                              double *gL, double *gR, double *gE, double *gS, double *res)
{
    const double eps = 1.0e-6;
    const int nI = 4 + k;
    memset(LL, 0, sizeof(double) * nI * nI); memset(RR, 0, sizeof(double) * nI * nI);
    memset(RS, 0, sizeof(double) * nI * 6); memset(SS, 0, sizeof(double) * 36);
    memset(LE, 0, sizeof(double) * (size_t)V * nI * 6); memset(EE, 0, sizeof(double) * (size_t)V * 36);
    memset(gL, 0, sizeof(double) * nI); memset(gR, 0, sizeof(double) * nI);
    memset(gE, 0, sizeof(double) * (size_t)V * 6); memset(gS, 0, sizeof(double) * 6);
    double ss = 0;
    const double *pS = vec + 2 * nI + 6 * V;
    double poseB[12], poseU[6][12], poseD[6][12];
    ext_from(pS, poseB);
    for (int q = 0; q < 6; ++q) {
        double t[6];
        memcpy(t, pS, sizeof(t)); t[q] += eps; ext_from(t, poseU[q]);
        memcpy(t, pS, sizeof(t)); t[q] -= eps; ext_from(t, poseD[q]);
    }
    for (int i = 0; i < V; ++i) {
        const double *pE = vec + 2 * nI + 6 * i;
        double EB[12], EU[6][12], ED[6][12], RB[12], RU[6][12], RD[6][12];
        ext_from(pE, EB);
        ext_compose(poseB, EB, RB);
        for (int q = 0; q < 6; ++q) {
            double t[6];
            memcpy(t, pE, sizeof(t)); t[q] += eps; ext_from(t, EU[q]);
            memcpy(t, pE, sizeof(t)); t[q] -= eps; ext_from(t, ED[q]);
            ext_compose(poseU[q], EB, RU[q]);
            ext_compose(poseD[q], EB, RD[q]);
        }
        for (int j = 0; j < C; ++j) {
            const double wx = world_xy[2 * j], wy = world_xy[2 * j + 1];
            double JL[2][15], JR[2][15], eL[2], eR[2], a[2], b[2];
            /* left rows: nI intrinsic columns, 6 extrinsic columns */
            orc_cal_value(vec, vec + 4, k, EB, wx, wy, a);
            eL[0] = a[0] - left_xy[2 * ((size_t)i * C + j)];
            eL[1] = a[1] - left_xy[2 * ((size_t)i * C + j) + 1];
            for (int c = 0; c < nI; ++c) {
                double up[9], lo[9];
                memcpy(up, vec, sizeof(double) * nI); memcpy(lo, vec, sizeof(double) * nI);
                up[c] += eps; lo[c] -= eps;
                orc_cal_value(up, up + 4, k, EB, wx, wy, a);
                orc_cal_value(lo, lo + 4, k, EB, wx, wy, b);
                JL[0][c] = (a[0] - b[0]) / (2.0 * eps); JL[1][c] = (a[1] - b[1]) / (2.0 * eps);
            }
            for (int q = 0; q < 6; ++q) {
                orc_cal_value(vec, vec + 4, k, EU[q], wx, wy, a);
                orc_cal_value(vec, vec + 4, k, ED[q], wx, wy, b);
                JL[0][nI + q] = (a[0] - b[0]) / (2.0 * eps); JL[1][nI + q] = (a[1] - b[1]) / (2.0 * eps);
            }
            /* right rows: nI intrinsic columns, 6 pose columns */
            orc_cal_value(vec + nI, vec + nI + 4, k, RB, wx, wy, a);
            eR[0] = a[0] - right_xy[2 * ((size_t)i * C + j)];
            eR[1] = a[1] - right_xy[2 * ((size_t)i * C + j) + 1];
            for (int c = 0; c < nI; ++c) {
                double up[9], lo[9];
                memcpy(up, vec + nI, sizeof(double) * nI); memcpy(lo, vec + nI, sizeof(double) * nI);
                up[c] += eps; lo[c] -= eps;
                orc_cal_value(up, up + 4, k, RB, wx, wy, a);
                orc_cal_value(lo, lo + 4, k, RB, wx, wy, b);
                JR[0][c] = (a[0] - b[0]) / (2.0 * eps); JR[1][c] = (a[1] - b[1]) / (2.0 * eps);
            }
            for (int q = 0; q < 6; ++q) {
                orc_cal_value(vec + nI, vec + nI + 4, k, RU[q], wx, wy, a);
                orc_cal_value(vec + nI, vec + nI + 4, k, RD[q], wx, wy, b);
                JR[0][nI + q] = (a[0] - b[0]) / (2.0 * eps); JR[1][nI + q] = (a[1] - b[1]) / (2.0 * eps);
            }
            for (int r = 0; r < 2; ++r) {
                ss += eL[r] * eL[r] + eR[r] * eR[r];
                for (int c = 0; c < nI; ++c) {
                    gL[c] += JL[r][c] * eL[r];
                    gR[c] += JR[r][c] * eR[r];
                    for (int d = 0; d < nI; ++d) {
                        LL[c * nI + d] += JL[r][c] * JL[r][d];
                        RR[c * nI + d] += JR[r][c] * JR[r][d];
                    }
                    for (int q = 0; q < 6; ++q) {
                        LE[((size_t)i * nI + c) * 6 + q] += JL[r][c] * JL[r][nI + q];
                        RS[c * 6 + q] += JR[r][c] * JR[r][nI + q];
                    }
                }
                for (int q = 0; q < 6; ++q) {
                    gE[(size_t)i * 6 + q] += JL[r][nI + q] * eL[r];
                    gS[q] += JR[r][nI + q] * eR[r];
                    for (int p = 0; p < 6; ++p) {
                        EE[((size_t)i * 6 + q) * 6 + p] += JL[r][nI + q] * JL[r][nI + p];
                        SS[q * 6 + p] += JR[r][nI + q] * JR[r][nI + p];
                    }
                }
            }
            if (res) {
                res[2 * ((size_t)i * C + j)] = eL[0];
                res[2 * ((size_t)i * C + j) + 1] = eL[1];
                res[2 * ((size_t)V * C) + 2 * ((size_t)i * C + j)] = eR[0];
                res[2 * ((size_t)V * C) + 2 * ((size_t)i * C + j) + 1] = eR[1];
            }
        }
    }
    return ss;
}

/* Monocular residual2 / jacobian2 (CMonocularCameraCalibration.cpp:799-906) for the
 * configuration doCalibrate forces (skew 0, five distortion terms: nine intrinsic
 * parameters).  Residual is observed minus projected and the Jacobian carries the
 * matching minus sign.  Dense, row-major rows x (9 + 6V). */
void orc_mono_residual_jacobian_dense(const double *vec, int V, int C, const double *world_xy,
                                      const double *img_xy, double *res, double *J)
{
    const double eps = 1.0e-6;
    const int cols = 9 + 6 * V;
    const size_t rows = (size_t)V * C * 2;
    double *E = (double *)malloc(sizeof(double) * 12 * (size_t)V);
    for (int i = 0; i < V; ++i) ext_from(vec + 9 + 6 * i, E + 12 * i);
    for (int i = 0; i < V; ++i)
        for (int j = 0; j < C; ++j) {
            double v[2];
            orc_cal_value(vec, vec + 4, 5, E + 12 * i, world_xy[2 * j], world_xy[2 * j + 1], v);
            res[2 * ((size_t)i * C + j)] = img_xy[2 * ((size_t)i * C + j)] - v[0];
            res[2 * ((size_t)i * C + j) + 1] = img_xy[2 * ((size_t)i * C + j) + 1] - v[1];
        }
    if (J) {
        memset(J, 0, sizeof(double) * rows * cols);
        double *up = (double *)malloc(sizeof(double) * cols), *lo = (double *)malloc(sizeof(double) * cols);
        double EU[12], EL[12];
        for (int c = 0; c < cols; ++c) {
            memcpy(up, vec, sizeof(double) * cols); memcpy(lo, vec, sizeof(double) * cols);
            up[c] += eps; lo[c] -= eps;
            for (int i = 0; i < V; ++i) {
                if (c >= 9 && (c - 9) / 6 != i) continue;
                ext_from(up + 9 + 6 * i, EU);
                ext_from(lo + 9 + 6 * i, EL);
                for (int j = 0; j < C; ++j) {
                    double a[2], b[2];
                    orc_cal_value(up, up + 4, 5, EU, world_xy[2 * j], world_xy[2 * j + 1], a);
                    orc_cal_value(lo, lo + 4, 5, EL, world_xy[2 * j], world_xy[2 * j + 1], b);
                    size_t row = 2 * ((size_t)i * C + j);
                    J[row * cols + c] = -(a[0] - b[0]) / (2.0 * eps);
                    J[(row + 1) * cols + c] = -(a[1] - b[1]) / (2.0 * eps);
                }
            }
        }
        free(up); free(lo);
    }
    free(E);
}

/* ---- rectification (Calibration/CStereoCameraRectification.cpp) ---------------------- */

/* General 3x3 inverse by Gaussian elimination with partial pivoting (the reference calls
 * Eigen's dynamic-size inverse(), a partial-pivot LU).  Returns 0 if singular. */
static int inv3(const double A[9], double out[9])
{
    double M[3][6];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 6; ++c) M[r][c] = c < 3 ? A[3 * r + c] : (c - 3 == r ? 1.0 : 0.0);
    for (int c = 0; c < 3; ++c) {
        int p = c;
        for (int r = c + 1; r < 3; ++r) if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
        if (M[p][c] == 0.0) return 0;
        if (p != c) for (int k = 0; k < 6; ++k) { double t = M[c][k]; M[c][k] = M[p][k]; M[p][k] = t; }
        for (int r = 0; r < 3; ++r) {
            if (r == c) continue;
            double f = M[r][c] / M[c][c];
            for (int k = 0; k < 6; ++k) M[r][k] -= f * M[c][k];
        }
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out[3 * r + c] = M[r][3 + c] / M[r][r];
    return 1;
}

/* Rotation matrix -> rotation vector, the branch of rodriguesTransform at
 * CStereoCameraRectification.cpp:170-207 (the SVD polish U V^T of :172-173 is the identity
 * map on a matrix that already is a rotation and is left out). */
static void rot_to_vec(const double R[9], double r[3])
{
    r[0] = R[7] - R[5]; r[1] = R[2] - R[6]; r[2] = R[3] - R[1];
    double s = sqrt((r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1.0e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t = (R[0] + 1) * 0.5;
        r[0] = sqrt(t > 0 ? t : 0.);
        t = (R[4] + 1) * 0.5;
        r[1] = sqrt(t > 0 ? t : 0.) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5;
        r[2] = sqrt(t > 0 ? t : 0.) * (R[2] < 0 ? -1. : 1.);
        if (fabs(r[0]) < fabs(r[1]) && fabs(r[0]) < fabs(r[2]) && (R[5] > 0) != (r[1] * r[2] > 0)) r[2] = -r[2];
        theta /= sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        for (int k = 0; k < 3; ++k) r[k] *= theta;
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        for (int k = 0; k < 3; ++k) r[k] *= vth;
    }
}

static void mat3mul(const double A[9], const double B[9], double C[9])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}

/* doStereoRectify (:13-58): half rotations, rectifying basis from the baseline, shared
 * new intrinsics, Q.  All matrices row-major. */
int orc_stereo_rectify(const double KL[9], const double KR[9], const double R_in[9],
                       const double T_in[3], double projL[9], double projR[9], double Q[16])
{
    double Rinv[9], eul[3], half[9], peer[9], t[3], rect[9], tmp[9];
    if (!inv3(R_in, Rinv)) return 0;                                    /* :21 */
    rot_to_vec(Rinv, eul);                                              /* :22 */
    for (int k = 0; k < 3; ++k) eul[k] *= -0.5;                         /* :23 */
    orc_rodrigues(eul, half);                                           /* :24 */
    mat3mul(half, Rinv, peer);                                          /* :25 */
    for (int i = 0; i < 3; ++i)                                         /* :26 */
        t[i] = -(peer[3 * i] * T_in[0] + peer[3 * i + 1] * T_in[1] + peer[3 * i + 2] * T_in[2]);
    double n1 = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    double e1[3] = { t[0] / n1, t[1] / n1, t[2] / n1 };                 /* :28-29 */
    double n2 = sqrt(t[0] * t[0] + t[1] * t[1]);
    double e2[3] = { -t[1] / n2, t[0] / n2, 0 / n2 };                   /* :30-31 */
    double e3[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2],
                     e1[0] * e2[1] - e1[1] * e2[0] };                   /* :32 */
    for (int k = 0; k < 3; ++k) { rect[k] = e1[k]; rect[3 + k] = e2[k]; rect[6 + k] = e3[k]; }
    double Tx = rect[0] * t[0] + rect[1] * t[1] + rect[2] * t[2];       /* :39 */
    double newK[9] = { (KL[0] + KL[4] + KR[0] + KR[4]) * 0.25, 0, (KL[2] + KR[2]) * 0.5,
                       0, 0, (KL[5] + KR[5]) * 0.5, 0, 0, 1 };          /* :41-47 */
    newK[4] = newK[0];
    mat3mul(rect, half, tmp); mat3mul(newK, tmp, projL);                /* :37, :48 */
    mat3mul(rect, peer, tmp); mat3mul(newK, tmp, projR);                /* :38, :49 */
    for (int k = 0; k < 16; ++k) Q[k] = (k % 5 == 0) ? 1.0 : 0.0;       /* :50-56 */
    Q[3] = -newK[2]; Q[7] = -newK[5]; Q[10] = 0; Q[11] = newK[0]; Q[14] = -1.0 / Tx; Q[15] = 0;
    return 1;
}

static void rectify_one(const double K[9], const double *d, int k, const double Pinv[9],
                        int w, int h, double *xmap, double *ymap)
{
    for (int j = 0; j < h; ++j)
        for (int i = 0; i < w; ++i) {
            double t0 = Pinv[0] * i + Pinv[1] * j + Pinv[2] * 1.0;      /* :81-82 */
            double t1 = Pinv[3] * i + Pinv[4] * j + Pinv[5] * 1.0;
            double t2 = Pinv[6] * i + Pinv[7] * j + Pinv[8] * 1.0;
            double nx = t0 / t2, ny = t1 / t2;                          /* :83-84 */
            double r = nx * nx + ny * ny, xe, ye;
            if (k == 2) {                                               /* :86-97 */
                xe = nx * (1.0 + d[0] * r + d[1] * r * r);
                ye = ny * (1.0 + d[0] * r + d[1] * r * r);
            } else {
                xe = nx * (1.0 + d[0] * r + d[1] * r * r + d[4] * r * r * r) + 2.0 * d[2] * nx * ny +
                     d[3] * (r * r + 2.0 * nx * nx);
                ye = ny * (1.0 + d[0] * r + d[1] * r * r + d[4] * r * r * r) + 2.0 * d[3] * nx * ny +
                     d[2] * (r * r + 2.0 * ny * ny);
            }
            double tx = K[0] * xe + K[1] * ye + K[2];                   /* :98-99 */
            double ty = K[4] * ye + K[5];
            int ok = tx >= 0 && ty >= 0 && tx <= w - 1 && ty <= h - 1;  /* :100 */
            xmap[(size_t)j * w + i] = ok ? tx : 9.99e+04;               /* INVALID_MAP, Common/Utils.hpp:16 */
            ymap[(size_t)j * w + i] = ok ? ty : 9.99e+04;
        }
}

/* doImageRectify (:60-143): for every pixel of the rectified image, where to sample the
 * original left / right image.  Returns 0 on the argument errors of :68-69 or a singular
 * projection. */
int orc_rectify_maps(int w, int h, const double KL[9], const double *dL, const double KR[9],
                     const double *dR, int k, const double projL[9], const double projR[9],
                     double *xl, double *yl, double *xr, double *yr)
{
    if (w <= 0 || h <= 0 || !(k == 2 || k == 5)) return 0;
    double PL[9], PR[9];
    if (!inv3(projL, PL) || !inv3(projR, PR)) return 0;
    rectify_one(KL, dL, k, PL, w, h, xl, yl);
    rectify_one(KR, dR, k, PR, w, h, xr, yr);
    return 1;
}
