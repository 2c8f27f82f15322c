/* Sanitizer run of the CPU oracle (test infrastructure; never part of the product).
 *
 * Linked directly against icp_oracle.c, cal_oracle.c and p2p_oracle.c, everything compiled
 * with -fsanitize=address,undefined (make -C oracle asan_driver), so that the restatement is
 * checked for the class of errors the reference itself has on this path: a `delete` of
 * memory that came from `new[]` (CPoint2PlaneICP.cpp:129,153), leaked buffers (:16,
 * CTxtReader.cpp:52), members read before they are set (CPoint2PlaneICP.cpp:3-6).
 *
 * usage: asan_driver ref.f32 m src.f32 n stereo.f64 V C K out.f64
 *   ref, src : raw float32 x y z rows
 *   stereo   : raw float64: vec (2 (4 + K) + 6 V + 6), world (C x 2), left (V x C x 2), right (V x C x 2)
 *   out      : raw float64: Trans (16) after 5 iterations, X (5 x 6), |e|^2, normals checksum,
 *              point-to-point Trans (16) after 3 iterations
 * Runs calcNormals, 5 iterations of the point-to-plane loop (as test.cpp:23), 3 of the
 * point-to-point loop, one block-sparse stereo evaluation with its residual, and the text
 * parser on a few lines, then frees everything (the leak checker runs at exit).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_kdtree orc_kdtree;
orc_kdtree *orc_kdtree_build(const float *xyz, int64_t m);
void orc_kdtree_free(orc_kdtree *T);
void orc_calc_normals(const orc_kdtree *T, const float *ref, int64_t m, float *normals,
                      int32_t *knn_idx_out, int threads);
int orc_icp_point2plane(const orc_kdtree *T, const float *ref, const float *normals, float *src,
                        int64_t n, int iters, double *Trans16, double *X_hist, int32_t *idx_hist,
                        float *d2_hist, double *info_hist, int faithful, int threads);
int orc_icp_point2point(const orc_kdtree *T, const float *ref, float *src, int64_t n, int iters,
                        double *Trans16, double *RT_hist, double *W_hist, int32_t *idx_hist,
                        float *d2_hist, double *info_hist, int faithful, int threads);
double orc_stereo_eval_sparse(const double *vec, int k, int V, int C, const double *world_xy,
                              const double *left_xy, const double *right_xy, double *LL,
                              double *RR, double *RS, double *SS, double *LE, double *EE,
                              double *gL, double *gR, double *gE, double *gS, double *res);
int64_t orc_parse_txt(const char *text, int64_t len, float *out, int64_t cap);

static void *read_all(const char *path, size_t bytes)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    void *p = malloc(bytes ? bytes : 1);
    if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "%s is shorter than %zu bytes\n", path, bytes); exit(2); }
    fclose(f);
    return p;
}

int main(int argc, char **argv)
{
    if (argc != 10) { fprintf(stderr, "usage: see the header of asan_driver.c\n"); return 2; }
    const int64_t m = atoll(argv[2]), n = atoll(argv[4]);
    const int V = atoi(argv[6]), C = atoi(argv[7]), K = atoi(argv[8]);
    float *ref = read_all(argv[1], sizeof(float) * 3 * (size_t)m);
    float *src = read_all(argv[3], sizeof(float) * 3 * (size_t)n);
    const int nI = 4 + K, P = 2 * nI + 6 * V + 6;
    double *st = read_all(argv[5], sizeof(double) * ((size_t)P + 2 * (size_t)C + 4 * (size_t)V * C));
    const double *vec = st, *world = st + P, *left = world + 2 * C, *right = left + 2 * (size_t)V * C;

    double out[16 + 30 + 2 + 16];
    orc_kdtree *T = orc_kdtree_build(ref, m);
    float *normals = malloc(sizeof(float) * 3 * (size_t)m);
    orc_calc_normals(T, ref, m, normals, NULL, 1);
    double cs = 0;
    for (int64_t i = 0; i < 3 * m; ++i) cs += normals[i] < 0 ? -normals[i] : normals[i];

    float *moving = malloc(sizeof(float) * 3 * (size_t)n);
    memcpy(moving, src, sizeof(float) * 3 * (size_t)n);
    double info[20];
    int bad = orc_icp_point2plane(T, ref, normals, moving, n, 5, out, out + 16, NULL, NULL, info, 1, 1);

    memcpy(moving, src, sizeof(float) * 3 * (size_t)n);
    bad |= orc_icp_point2point(T, ref, moving, n, 3, out + 48, NULL, NULL, NULL, NULL, NULL, 1, 1);

    double *LL = malloc(sizeof(double) * nI * nI), *RR = malloc(sizeof(double) * nI * nI),
           *RS = malloc(sizeof(double) * nI * 6), *SS = malloc(sizeof(double) * 36),
           *LE = malloc(sizeof(double) * (size_t)V * nI * 6), *EE = malloc(sizeof(double) * (size_t)V * 36),
           *gL = malloc(sizeof(double) * nI), *gR = malloc(sizeof(double) * nI),
           *gE = malloc(sizeof(double) * (size_t)V * 6), *gS = malloc(sizeof(double) * 6),
           *res = malloc(sizeof(double) * 4 * (size_t)V * C);
    out[46] = orc_stereo_eval_sparse(vec, K, V, C, world, left, right, LL, RR, RS, SS, LE, EE, gL, gR, gE, gS, res);
    out[47] = cs;

    /* the text format of CTxtReader.cpp:13-61: sized first, then parsed */
    const char text[] = "1.5 2.5 -3\n4 5 6 \n  7e-1 8 9";
    const int64_t tokens = orc_parse_txt(text, (int64_t)sizeof text - 1, NULL, 0);
    float *parsed = malloc(sizeof(float) * (size_t)(tokens > 0 ? tokens : 1));
    orc_parse_txt(text, (int64_t)sizeof text - 1, parsed, tokens);
    if (tokens != 9 || parsed[2] != -3.0f || parsed[6] != 0.7f) bad |= 2;

    FILE *f = fopen(argv[9], "wb");
    if (!f || fwrite(out, sizeof(double), 64, f) != 64) { fprintf(stderr, "cannot write %s\n", argv[9]); return 2; }
    fclose(f);

    free(parsed);
    free(LL); free(RR); free(RS); free(SS); free(LE); free(EE); free(gL); free(gR); free(gE); free(gS); free(res);
    free(moving); free(normals);
    orc_kdtree_free(T);
    free(st); free(src); free(ref);
    printf("asan_driver: done, status %d\n", bad);
    return bad ? 1 : 0;
}
