/*
 * oracle/cfconv_oracle.c -- TEST INFRASTRUCTURE ONLY (parity checker + timed CPU baseline).
 *
 * A plain-C restatement of the reference's single-threaded CPU algorithm for the SchNet
 * continuous-filter convolution and its half neighbour list.  Nothing in the product path
 * may include, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Algorithm followed (reference = /root/reference/src/schnet/CpuCFConv.cpp):
 *   - minimum image (same z,y,x single-round rule as the ANI code) .. ref :30-54
 *   - half list {j>i : r^2 < c^2} with stored distances ............. ref :61-116
 *   - Gaussian centres mu_g = g*c/(G-1); w1 read as [W][G] .......... ref :118-131, :163
 *   - forward: gaussians -> dense1 -> act -> dense2 -> cutoff ->
 *     symmetric accumulate into both atoms .......................... ref :133-188
 *   - backward: forward-mode d/dr through both layers, input grads
 *     and position grads ............................................ ref :190-299
 *   - cosine cutoff / derivative .................................... ref :301-307
 *
 * Note: the reference writes the Gaussian as `exp(-0.5f*x*x)`; under <cmath> with
 * `using namespace std` that resolves to the float overload, so this C file calls expf
 * (pinned bit-for-bit by tests against oracle/_ref).
 *
 * Parity status: PINNED -- against the SchNetPack-generated golden vectors held by the
 * reference's C++ test (src/schnet/TestCFConv.h:142-247, re-encoded in
 * tests/golden/cfconv_water18.npz) and against oracle/_ref on seeded inputs.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ neighbours */
typedef struct {
    int n_atoms;
    float cutoff;
    int periodic, triclinic;
    int *start;       /* [n_atoms+1] CSR offsets of the half list */
    int *other;       /* atom2 of each half pair */
    float *dist;      /* stored distance of each half pair */
    int cap;
} cfconv_oracle_neighbors;

static inline float displacement(int periodic, int triclinic, const float *box, const float *inv_diag,
                                 const float *p1, const float *p2, float d[3]) {
    d[0] = p2[0] - p1[0];
    d[1] = p2[1] - p1[1];
    d[2] = p2[2] - p1[2];
    if (periodic) {
        if (triclinic) {
            float s3 = roundf(d[2] * inv_diag[2]);
            d[0] -= s3 * box[6]; d[1] -= s3 * box[7]; d[2] -= s3 * box[8];
            float s2 = roundf(d[1] * inv_diag[1]);
            d[0] -= s2 * box[3]; d[1] -= s2 * box[4];
            float s1 = roundf(d[0] * inv_diag[0]);
            d[0] -= s1 * box[0];
        } else {
            d[0] -= roundf(d[0] * inv_diag[0]) * box[0];
            d[1] -= roundf(d[1] * inv_diag[1]) * box[4];
            d[2] -= roundf(d[2] * inv_diag[2]) * box[8];
        }
    }
    return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
}

cfconv_oracle_neighbors *cfconv_oracle_neighbors_create(int n_atoms, float cutoff, int periodic) {
    cfconv_oracle_neighbors *nb = (cfconv_oracle_neighbors *)calloc(1, sizeof(*nb));
    nb->n_atoms = n_atoms; nb->cutoff = cutoff; nb->periodic = periodic;
    nb->start = (int *)calloc(n_atoms + 1, sizeof(int));
    return nb;
}

void cfconv_oracle_neighbors_destroy(cfconv_oracle_neighbors *nb) {
    if (!nb) return;
    free(nb->start); free(nb->other); free(nb->dist); free(nb);
}

void cfconv_oracle_neighbors_build(cfconv_oracle_neighbors *nb, const float *pos, const float *box) {
    const int N = nb->n_atoms;
    float inv_diag[3] = {0, 0, 0};
    nb->triclinic = 0;
    if (nb->periodic) {                                    /* ref :64-68, :93-97 */
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                if (a != b && box[3 * a + b] != 0) nb->triclinic = 1;
        inv_diag[0] = 1 / box[0]; inv_diag[1] = 1 / box[4]; inv_diag[2] = 1 / box[8];
    }
    const float c2 = nb->cutoff * nb->cutoff;
    int count = 0;
    for (int i = 0; i < N; i++) {                          /* ref :104-115 */
        nb->start[i] = count;
        for (int j = i + 1; j < N; j++) {
            float d[3];
            float r2 = displacement(nb->periodic, nb->triclinic, box, inv_diag, &pos[3 * i], &pos[3 * j], d);
            if (r2 < c2) {
                if (count == nb->cap) {
                    nb->cap = nb->cap ? 2 * nb->cap : 1024;
                    nb->other = (int *)realloc(nb->other, sizeof(int) * nb->cap);
                    nb->dist = (float *)realloc(nb->dist, sizeof(float) * nb->cap);
                }
                nb->other[count] = j;
                nb->dist[count] = sqrtf(r2);
                count++;
            }
        }
    }
    nb->start[N] = count;
}

int cfconv_oracle_neighbors_num_pairs(const cfconv_oracle_neighbors *nb) { return nb->start[nb->n_atoms]; }
const int *cfconv_oracle_neighbors_start(const cfconv_oracle_neighbors *nb) { return nb->start; }
const int *cfconv_oracle_neighbors_other(const cfconv_oracle_neighbors *nb) { return nb->other; }
const float *cfconv_oracle_neighbors_dist(const cfconv_oracle_neighbors *nb) { return nb->dist; }

/* ------------------------------------------------------------------ convolution */
typedef struct {
    int n_atoms, width, n_gauss;
    float cutoff, sigma;
    int periodic, activation;   /* 0 = shifted softplus, 1 = tanh (ref CFConv.h:114-117) */
    float *mu, *w1, *b1, *w2, *b2;
    float *gauss, *dgauss, *y1, *dy1, *y2, *dy2;
} cfconv_oracle;

cfconv_oracle *cfconv_oracle_create(int n_atoms, int width, int n_gauss, float cutoff, int periodic, float sigma,
                                    int activation, const float *w1, const float *b1, const float *w2, const float *b2) {
    cfconv_oracle *c = (cfconv_oracle *)calloc(1, sizeof(*c));
    c->n_atoms = n_atoms; c->width = width; c->n_gauss = n_gauss;
    c->cutoff = cutoff; c->sigma = sigma; c->periodic = periodic; c->activation = activation;
    c->mu = (float *)malloc(sizeof(float) * n_gauss);
    for (int g = 0; g < n_gauss; g++) c->mu[g] = g * cutoff / (n_gauss - 1);   /* ref :121-122 */
    c->w1 = (float *)malloc(sizeof(float) * n_gauss * width);
    c->w2 = (float *)malloc(sizeof(float) * width * width);
    c->b1 = (float *)malloc(sizeof(float) * width);
    c->b2 = (float *)malloc(sizeof(float) * width);
    memcpy(c->w1, w1, sizeof(float) * n_gauss * width);
    memcpy(c->w2, w2, sizeof(float) * width * width);
    memcpy(c->b1, b1, sizeof(float) * width);
    memcpy(c->b2, b2, sizeof(float) * width);
    int m = n_gauss > width ? n_gauss : width;
    c->gauss = (float *)malloc(sizeof(float) * m); c->dgauss = (float *)malloc(sizeof(float) * m);
    c->y1 = (float *)malloc(sizeof(float) * m);    c->dy1 = (float *)malloc(sizeof(float) * m);
    c->y2 = (float *)malloc(sizeof(float) * m);    c->dy2 = (float *)malloc(sizeof(float) * m);
    return c;
}

void cfconv_oracle_destroy(cfconv_oracle *c) {
    if (!c) return;
    free(c->mu); free(c->w1); free(c->b1); free(c->w2); free(c->b2);
    free(c->gauss); free(c->dgauss); free(c->y1); free(c->dy1); free(c->y2); free(c->dy2);
    free(c);
}

static inline float fcut(const cfconv_oracle *c, float r) { return 0.5f * cosf(M_PI * r / c->cutoff) + 0.5f; }
static inline float fcut_deriv(const cfconv_oracle *c, float r) {
    return -(0.5f * M_PI / c->cutoff) * sinf(M_PI * r / c->cutoff);
}

void cfconv_oracle_forward(cfconv_oracle *c, const cfconv_oracle_neighbors *nb, const float *pos, const float *box,
                           const float *input, float *output) {
    (void)pos; (void)box;   /* forward works from the stored distances only (ref :146-147) */
    const int N = c->n_atoms, W = c->width, G = c->n_gauss;
    memset(output, 0, sizeof(float) * (size_t)N * W);
    for (int i = 0; i < N; i++) {
        for (int p = nb->start[i]; p < nb->start[i + 1]; p++) {
            const int j = nb->other[p];
            const float r = nb->dist[p];
            for (int g = 0; g < G; g++) {                      /* ref :151-154 */
                float x = (r - c->mu[g]) / c->sigma;
                c->gauss[g] = expf(-0.5f * x * x);
            }
            for (int a = 0; a < W; a++) {                      /* ref :158-166 */
                float s = c->b1[a];
                for (int g = 0; g < G; g++) s += c->gauss[g] * c->w1[a * G + g];
                c->y1[a] = c->activation == 0 ? logf(0.5f * expf(s) + 0.5f) : tanhf(s);
            }
            const float fc = fcut(c, r);                       /* ref :170-176 */
            for (int a = 0; a < W; a++) {
                float s = c->b2[a];
                for (int b = 0; b < W; b++) s += c->y1[b] * c->w2[a * W + b];
                c->y2[a] = fc * s;
            }
            for (int a = 0; a < W; a++) {                      /* ref :180-183 */
                output[(size_t)i * W + a] += c->y2[a] * input[(size_t)j * W + a];
                output[(size_t)j * W + a] += c->y2[a] * input[(size_t)i * W + a];
            }
        }
    }
}

void cfconv_oracle_backward(cfconv_oracle *c, const cfconv_oracle_neighbors *nb, const float *pos, const float *box,
                            const float *input, const float *output_grad, float *input_grad, float *pos_grad) {
    const int N = c->n_atoms, W = c->width, G = c->n_gauss;
    float inv_diag[3] = {0, 0, 0};
    if (c->periodic) { inv_diag[0] = 1 / box[0]; inv_diag[1] = 1 / box[4]; inv_diag[2] = 1 / box[8]; }
    memset(input_grad, 0, sizeof(float) * (size_t)N * W);
    memset(pos_grad, 0, sizeof(float) * 3 * N);
    for (int i = 0; i < N; i++) {
        for (int p = nb->start[i]; p < nb->start[i + 1]; p++) {
            const int j = nb->other[p];
            float d[3];
            float r2 = displacement(c->periodic, nb->triclinic, box, inv_diag, &pos[3 * i], &pos[3 * j], d);
            float r = sqrtf(r2), rinv = 1 / r;                 /* ref :233-235 */
            for (int g = 0; g < G; g++) {                      /* ref :239-243 */
                float x = (r - c->mu[g]) / c->sigma;
                c->gauss[g] = expf(-0.5f * x * x);
                c->dgauss[g] = -x * c->gauss[g] / c->sigma;
            }
            for (int a = 0; a < W; a++) {                      /* ref :247-263 */
                float s = c->b1[a], ds = 0;
                for (int g = 0; g < G; g++) {
                    s += c->gauss[g] * c->w1[a * G + g];
                    ds += c->dgauss[g] * c->w1[a * G + g];
                }
                if (c->activation == 0) {
                    float e = expf(s);
                    c->y1[a] = logf(0.5f * e + 0.5f);
                    c->dy1[a] = ds * e / (e + 1);
                } else {
                    float th = tanhf(s);
                    c->y1[a] = th;
                    c->dy1[a] = ds * (1 - th * th);
                }
            }
            const float fc = fcut(c, r), dfc = fcut_deriv(c, r);   /* ref :267-277 */
            for (int a = 0; a < W; a++) {
                float s = c->b2[a], ds = 0;
                for (int b = 0; b < W; b++) {
                    s += c->y1[b] * c->w2[a * W + b];
                    ds += c->dy1[b] * c->w2[a * W + b];
                }
                c->y2[a] = fc * s;
                c->dy2[a] = dfc * s + fc * ds;
            }
            for (int a = 0; a < W; a++) {                      /* ref :281-293 */
                size_t qi = (size_t)i * W + a, qj = (size_t)j * W + a;
                input_grad[qi] += c->y2[a] * output_grad[qj];
                input_grad[qj] += c->y2[a] * output_grad[qi];
                float sc = rinv * c->dy2[a] * (input[qj] * output_grad[qi] + input[qi] * output_grad[qj]);
                for (int k = 0; k < 3; k++) {
                    float t = sc * d[k];
                    pos_grad[3 * i + k] -= t;
                    pos_grad[3 * j + k] += t;
                }
            }
        }
    }
}
