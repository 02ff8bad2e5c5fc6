/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's contact-path
 * arithmetic (muelea/tuch).  Nothing under tuch_amd/ may link, import or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg do, and only as the checker.
 *
 * Each function cites the reference lines it restates.  All arithmetic is
 * float32 in the reference's operation order (compile with -ffp-contract=off);
 * the only deliberate deviation is that long sums are accumulated in double,
 * because torch's CPU sum() is a vectorised pairwise reduction whose error is
 * far below a sequential float loop's -- the double accumulator is the closer
 * restatement.  Pinned against golden vectors produced by importing the
 * reference itself (tests/golden/make_golden.py); see oracle/README.md.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

void oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* tuch/utils/contact.py:23-47 (batch_pairwise_dist, squared=True), one batch
 * element: P[i][j] = xx[i] + yy[j] - 2 * (x_i . y_j), the three bmm's being
 * 3-term dot products. */
void oracle_pairwise_sq(const float *x, int nx, const float *y, int ny, float *P)
{
    float *xx = (float *)malloc(sizeof(float) * (size_t)nx);
    float *yy = (float *)malloc(sizeof(float) * (size_t)ny);
    for (int i = 0; i < nx; ++i)
        xx[i] = x[3 * i] * x[3 * i] + x[3 * i + 1] * x[3 * i + 1] + x[3 * i + 2] * x[3 * i + 2];
    for (int j = 0; j < ny; ++j)
        yy[j] = y[3 * j] * y[3 * j] + y[3 * j + 1] * y[3 * j + 1] + y[3 * j + 2] * y[3 * j + 2];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nx; ++i) {
        const float *xi = x + 3 * i;
        for (int j = 0; j < ny; ++j) {
            const float *yj = y + 3 * j;
            float zz = xi[0] * yj[0] + xi[1] * yj[1] + xi[2] * yj[2];
            P[(size_t)i * ny + j] = xx[i] + yy[j] - 2.0f * zz;
        }
    }
    free(xx);
    free(yy);
}

/* tuch/smplify/losses.py:92-93 and tuch/train/loss.py:269-270: entries with
 * geomask false become +inf, then min / argmin over dim=1 of the [1,V,V]
 * matrix, i.e. over rows j for every column i; first index wins ties, an
 * all-inf column yields index 0.  geomask is the byte-per-entry bool matrix the
 * reference holds (geod > geothres, smplifydc.py:65). */
void oracle_v2v_min_masked(const float *verts, int V, const uint8_t *geomask,
                           float *min_d2, int64_t *argmin)
{
    float *nn = (float *)malloc(sizeof(float) * (size_t)V);
    for (int i = 0; i < V; ++i)
        nn[i] = verts[3 * i] * verts[3 * i] + verts[3 * i + 1] * verts[3 * i + 1] +
                verts[3 * i + 2] * verts[3 * i + 2];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < V; ++i) {
        float best = INFINITY;
        int64_t arg = 0;
        const float *vi = verts + 3 * i;
        for (int j = 0; j < V; ++j) {
            if (!geomask[(size_t)j * V + i]) continue;
            const float *vj = verts + 3 * j;
            float zz = vj[0] * vi[0] + vj[1] * vi[1] + vj[2] * vi[2];
            float p = nn[j] + nn[i] - 2.0f * zz; /* P[j][i] */
            if (p < best) { best = p; arg = j; }
        }
        min_d2[i] = best;
        argmin[i] = arg;
    }
    free(nn);
}

/* tuch/utils/contact.py:79-109: Van Oosterom-Strackee solid angle of triangle
 * (a,b,c) seen from q, 2*atan2(num, den). */
static inline float solid_angle(const float *q, const float *a, const float *b, const float *c)
{
    float A[3] = {a[0] - q[0], a[1] - q[1], a[2] - q[2]};
    float B[3] = {b[0] - q[0], b[1] - q[1], b[2] - q[2]};
    float C[3] = {c[0] - q[0], c[1] - q[1], c[2] - q[2]};
    float nA = sqrtf(A[0] * A[0] + A[1] * A[1] + A[2] * A[2]);
    float nB = sqrtf(B[0] * B[0] + B[1] * B[1] + B[2] * B[2]);
    float nC = sqrtf(C[0] * C[0] + C[1] * C[1] + C[2] * C[2]);
    float cx = B[1] * C[2] - B[2] * C[1];
    float cy = B[2] * C[0] - B[0] * C[2];
    float cz = B[0] * C[1] - B[1] * C[0];
    float num = A[0] * cx + A[1] * cy + A[2] * cz;
    float d01 = A[0] * B[0] + A[1] * B[1] + A[2] * B[2];
    float d12 = B[0] * C[0] + B[1] * C[1] + B[2] * C[2];
    float d02 = A[0] * C[0] + A[1] * C[1] + A[2] * C[2];
    float den = nA * nB * nC + d01 * nC + d02 * nB + d12 * nA;
    return 2.0f * atan2f(num, den);
}

/* contact.py:49-109, materialised [Q,F] (small sizes only). */
void oracle_solid_angles(const float *points, int Q, const float *tris, int F, float *out)
{
#pragma omp parallel for schedule(static)
    for (int qi = 0; qi < Q; ++qi)
        for (int f = 0; f < F; ++f)
            out[(size_t)qi * F + f] =
                solid_angle(points + 3 * qi, tris + 9 * f, tris + 9 * f + 3, tris + 9 * f + 6);
}

/* contact.py:112-147: w = 1/(4 pi) * sum_f solid_angle.  tris is [F][3][3]. */
void oracle_winding(const float *points, int Q, const float *tris, int F, float *w)
{
    const float scale = (float)(1.0 / (4.0 * M_PI));
#pragma omp parallel for schedule(dynamic, 16)
    for (int qi = 0; qi < Q; ++qi) {
        double acc = 0.0;
        for (int f = 0; f < F; ++f)
            acc += (double)solid_angle(points + 3 * qi, tris + 9 * f, tris + 9 * f + 3,
                                       tris + 9 * f + 6);
        w[qi] = scale * (float)acc;
    }
}

/* (verts)[face_tensor[0]] -- losses.py:81, loss.py:260.  nv may exceed the body's
 * V when cap vertices were appended (segmentation.py:74-77). */
void oracle_gather_tris(const float *verts, const int64_t *faces, int F, float *tris)
{
    for (int f = 0; f < F; ++f)
        for (int k = 0; k < 3; ++k)
            memcpy(tris + 9 * f + 3 * k, verts + 3 * faces[3 * f + k], 3 * sizeof(float));
}

/* Minimum of the *direct-difference* squared distance over a region pair, in
 * double: the fp64 truth that the bmm-form values of losses.py:115-116 /
 * train_module.py:88-90 are noisy estimates of. */
void oracle_region_min_f64(const float *verts, const int64_t *r1, int n1, const int64_t *r2,
                           int n2, double *out_min, int64_t *out_i, int64_t *out_j)
{
    double best = INFINITY;
    int64_t bi = 0, bj = 0;
    for (int a = 0; a < n1; ++a)
        for (int b = 0; b < n2; ++b) {
            const float *p = verts + 3 * r1[a], *q = verts + 3 * r2[b];
            double dx = (double)p[0] - q[0], dy = (double)p[1] - q[1], dz = (double)p[2] - q[2];
            double d = dx * dx + dy * dy + dz * dz;
            if (d < best) { best = d; bi = r1[a]; bj = r2[b]; }
        }
    *out_min = best;
    *out_i = bi;
    *out_j = bj;
}
