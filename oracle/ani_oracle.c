/*
 * oracle/ani_oracle.c -- TEST INFRASTRUCTURE ONLY (parity checker + timed CPU baseline).
 *
 * A plain-C restatement of the reference's single-threaded CPU algorithm for the ANI
 * atomic-environment-vector symmetry functions.  Nothing in the product path
 * (nnpops_amd/, libnnpops_hip.so, the torch binding) may include, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Algorithm followed (reference = /root/reference/src/ani/CpuANISymmetryFunctions.cpp):
 *   - species-pair bucket table, upper-triangular row-major ........ ref :39-43
 *   - forward: zero outputs, radial pass, angular pass, post-scale .. ref :46-110
 *   - radial pass over i<j pairs (+ angular neighbour lists) ........ ref :112-151
 *   - angular pass over neighbour pairs of every centre atom ........ ref :153-194
 *   - backward: zero, radial i<j pass, angular triple pass .......... ref :196-353
 *   - minimum image, one round() per axis in z,y,x order ............ ref :355-379
 *   - cosine cutoff and its derivative (double-precision pi) ........ ref :381-387
 *   - angle (0.95 damping in TorchANI mode, asin branch otherwise) .. ref :389-408
 *   - angle gradients ............................................... ref :410-433
 *
 * The arithmetic is kept in the same precision and the same association order as the
 * reference (float everywhere, M_PI promoted to double inside the cutoff argument) so that
 * the restatement can be pinned bit-for-bit against oracle/_ref (the reference's own
 * sources compiled in place); see tests/test_oracle_pinning.py.
 *
 * Parity status: PINNED -- against the TorchANI-generated golden vectors held by the
 * reference's C++ test (src/ani/TestANISymmetryFunctions.h:111-252, re-encoded in
 * tests/golden/ani_water18.npz) and against oracle/_ref on seeded inputs.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Precision of the restatement.  The default, float with the float maths functions, is the reference's arithmetic and the
 * build every parity test and the CPU baseline use (bit-for-bit equal to oracle/_ref).  -DORACLE_DOUBLE builds the same
 * algorithm in double precision (liboracle64.so, exported with an ani_oracle64_ prefix): an "exact" answer against which the
 * fp32 reference's OWN rounding error can be measured, for the ill-conditioned corner of paper-mode angle gradients. */
#ifdef ORACLE_DOUBLE
typedef double real;
#define R_ACOS(x) acos(x)
#define R_ASIN(x) asin(x)
#define R_COS(x) cos(x)
#define R_EXP(x) exp(x)
#define R_POW(x, y) pow(x, y)
#define R_ROUND(x) round(x)
#define R_SIN(x) sin(x)
#define R_SQRT(x) sqrt(x)
#define ani_oracle_create ani_oracle64_create
#define ani_oracle_destroy ani_oracle64_destroy
#define ani_oracle_forward ani_oracle64_forward
#define ani_oracle_backward ani_oracle64_backward
#else
typedef float real;
#define R_ACOS(x) acosf(x)
#define R_ASIN(x) asinf(x)
#define R_COS(x) cosf(x)
#define R_EXP(x) expf(x)
#define R_POW(x, y) powf(x, y)
#define R_ROUND(x) roundf(x)
#define R_SIN(x) sinf(x)
#define R_SQRT(x) sqrtf(x)
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
    int n_atoms, n_species, n_radial, n_angular;
    real rc_radial, rc_angular;
    int periodic, torchani;
    int *species;          /* [n_atoms] */
    real *rad_eta, *rad_rs;                       /* [n_radial] */
    real *ang_eta, *ang_rs, *ang_zeta, *ang_ths;  /* [n_angular] */
    int *bucket;           /* [n_species*n_species] species pair -> angular block */
    /* state left by the last forward(), consumed by backward() (ref ANISymmetryFunctions.h:83-84) */
    real *pos;            /* [n_atoms*3] */
    real box[9], inv_diag[3];
    int triclinic;
    int *nbr_start;        /* [n_atoms+1] */
    int *nbr_count;        /* [n_atoms]   */
    int *nbr;              /* flat, capacity grown on demand */
    int nbr_cap;
    int **tmp_lists;       /* per-atom growable lists used while scanning i<j */
    int *tmp_len, *tmp_cap;
} ani_oracle;

void ani_oracle_destroy(ani_oracle *o);

ani_oracle *ani_oracle_create(int n_atoms, int n_species, real rc_radial, real rc_angular,
                              int periodic, const int *species,
                              int n_radial, const real *radial_eta_rs,         /* [n_radial][2]  */
                              int n_angular, const real *angular_eta_rs_zeta_ths, /* [n_angular][4] */
                              int torchani) {
    ani_oracle *o = (ani_oracle *)calloc(1, sizeof(ani_oracle));
    o->n_atoms = n_atoms; o->n_species = n_species;
    o->n_radial = n_radial; o->n_angular = n_angular;
    o->rc_radial = rc_radial; o->rc_angular = rc_angular;
    o->periodic = periodic; o->torchani = torchani;
    o->species = (int *)malloc(sizeof(int) * (n_atoms > 0 ? n_atoms : 1));
    memcpy(o->species, species, sizeof(int) * n_atoms);
    o->rad_eta = (real *)malloc(sizeof(real) * (n_radial + 1));
    o->rad_rs = (real *)malloc(sizeof(real) * (n_radial + 1));
    for (int k = 0; k < n_radial; k++) {
        o->rad_eta[k] = radial_eta_rs[2 * k];
        o->rad_rs[k] = radial_eta_rs[2 * k + 1];
    }
    o->ang_eta = (real *)malloc(sizeof(real) * (n_angular + 1));
    o->ang_rs = (real *)malloc(sizeof(real) * (n_angular + 1));
    o->ang_zeta = (real *)malloc(sizeof(real) * (n_angular + 1));
    o->ang_ths = (real *)malloc(sizeof(real) * (n_angular + 1));
    for (int m = 0; m < n_angular; m++) {
        o->ang_eta[m] = angular_eta_rs_zeta_ths[4 * m];
        o->ang_rs[m] = angular_eta_rs_zeta_ths[4 * m + 1];
        o->ang_zeta[m] = angular_eta_rs_zeta_ths[4 * m + 2];
        o->ang_ths[m] = angular_eta_rs_zeta_ths[4 * m + 3];
    }
    /* ref :39-43 -- blocks numbered along the upper triangle, row by row, mirrored below it */
    o->bucket = (int *)malloc(sizeof(int) * n_species * n_species);
    int next = 0;
    for (int a = 0; a < n_species; a++)
        for (int b = a; b < n_species; b++) {
            o->bucket[a * n_species + b] = next;
            o->bucket[b * n_species + a] = next;
            next++;
        }
    o->pos = (real *)malloc(sizeof(real) * 3 * (n_atoms > 0 ? n_atoms : 1));
    o->nbr_start = (int *)calloc(n_atoms + 1, sizeof(int));
    o->nbr_count = (int *)calloc(n_atoms + 1, sizeof(int));
    o->tmp_lists = (int **)calloc(n_atoms + 1, sizeof(int *));
    o->tmp_len = (int *)calloc(n_atoms + 1, sizeof(int));
    o->tmp_cap = (int *)calloc(n_atoms + 1, sizeof(int));
    return o;
}

void ani_oracle_destroy(ani_oracle *o) {
    if (!o) return;
    for (int i = 0; i < o->n_atoms; i++) free(o->tmp_lists[i]);
    free(o->tmp_lists); free(o->tmp_len); free(o->tmp_cap);
    free(o->species); free(o->rad_eta); free(o->rad_rs);
    free(o->ang_eta); free(o->ang_rs); free(o->ang_zeta); free(o->ang_ths);
    free(o->bucket); free(o->pos); free(o->nbr_start); free(o->nbr_count); free(o->nbr);
    free(o);
}

/* ref :355-379.  d = p2 - p1, then (periodic) at most one lattice translation per axis. */
static inline real displacement(const ani_oracle *o, const real *p1, const real *p2, real d[3]) {
    d[0] = p2[0] - p1[0];
    d[1] = p2[1] - p1[1];
    d[2] = p2[2] - p1[2];
    if (o->periodic) {
        const real *b = o->box;
        if (o->triclinic) {
            real s3 = R_ROUND(d[2] * o->inv_diag[2]);
            d[0] -= s3 * b[6]; d[1] -= s3 * b[7]; d[2] -= s3 * b[8];
            real s2 = R_ROUND(d[1] * o->inv_diag[1]);
            d[0] -= s2 * b[3]; d[1] -= s2 * b[4];
            real s1 = R_ROUND(d[0] * o->inv_diag[0]);
            d[0] -= s1 * b[0];
        } else {
            d[0] -= R_ROUND(d[0] * o->inv_diag[0]) * b[0];
            d[1] -= R_ROUND(d[1] * o->inv_diag[1]) * b[4];
            d[2] -= R_ROUND(d[2] * o->inv_diag[2]) * b[8];
        }
    }
    return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
}

/* ref :381-387 (the argument is evaluated in double, then narrowed for cosf/sinf) */
static inline real fcut(real r, real rc) { return 0.5f * R_COS(M_PI * r / rc) + 0.5f; }
static inline real fcut_deriv(real r, real rc) { return -(0.5f * M_PI / rc) * R_SIN(M_PI * r / rc); }

/* ref :389-408 */
static inline real angle_between(const ani_oracle *o, const real *u, const real *v, real ru, real rv) {
    real dot = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
    if (o->torchani) dot *= 0.95f;
    real c = dot / (ru * rv);
    if (!o->torchani && (c > 0.99f || c < -0.99f)) {
        real x[3] = { u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0] };
        real a = R_ASIN(R_SQRT(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) / (ru * rv));
        if (c < 0) a = M_PI - a;
        return a;
    }
    return R_ACOS(c);
}

/* ref :410-433 */
static inline void angle_gradients(const ani_oracle *o, const real *u, const real *v, real ru, real rv,
                                   real gu[3], real gv[3]) {
    real dot = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
    real iu = 1 / ru, iv = 1 / rv;
    real iprod = iu * iv, iu2 = iu * iu, iv2 = iv * iv;
    real dadd;
    if (o->torchani) {
        real sd = 0.95f * dot * iprod;
        dadd = -0.95f / R_SQRT(1 - sd * sd);
    } else {
        real sd = dot * iprod;
        dadd = -1 / R_SQRT(1 - sd * sd);
    }
    for (int k = 0; k < 3; k++) {
        gu[k] = dadd * iprod * (v[k] - dot * iu2 * u[k]);
        gv[k] = dadd * iprod * (u[k] - dot * iv2 * v[k]);
    }
}

static void tmp_push(ani_oracle *o, int atom, int value) {
    if (o->tmp_len[atom] == o->tmp_cap[atom]) {
        o->tmp_cap[atom] = o->tmp_cap[atom] ? 2 * o->tmp_cap[atom] : 16;
        o->tmp_lists[atom] = (int *)realloc(o->tmp_lists[atom], sizeof(int) * o->tmp_cap[atom]);
    }
    o->tmp_lists[atom][o->tmp_len[atom]++] = value;
}

void ani_oracle_forward(ani_oracle *o, const real *positions, const real *box, real *radial, real *angular) {
    const int N = o->n_atoms, S = o->n_species, nR = o->n_radial, nA = o->n_angular;
    const int nB = S * (S + 1) / 2;
    memcpy(o->pos, positions, sizeof(real) * 3 * N);
    o->triclinic = 0;
    if (o->periodic) {                                       /* ref :49-64 */
        memcpy(o->box, box, sizeof(real) * 9);
        o->inv_diag[0] = 1 / o->box[0];
        o->inv_diag[1] = 1 / o->box[4];
        o->inv_diag[2] = 1 / o->box[8];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                if (a != b && o->box[3 * a + b] != 0) o->triclinic = 1;
    }
    memset(radial, 0, sizeof(real) * (size_t)N * S * nR);    /* ref :68-69 */
    memset(angular, 0, sizeof(real) * (size_t)N * nB * nA);

    /* ---- radial pass, ref :112-151 ---- */
    const real rc2_rad = o->rc_radial * o->rc_radial;
    const real rc2_ang = o->rc_angular * o->rc_angular;
    for (int i = 0; i < N; i++) o->tmp_len[i] = 0;
    for (int i = 0; i < N; i++) {
        for (int j = i + 1; j < N; j++) {
            real d[3];
            real r2 = displacement(o, &o->pos[3 * i], &o->pos[3 * j], d);
            if (!(r2 < rc2_rad)) continue;
            if (r2 < rc2_ang) { tmp_push(o, i, j); tmp_push(o, j, i); }
            real r = R_SQRT(r2);
            real fc = fcut(r, o->rc_radial);
            real *row_i = &radial[((size_t)i * S + o->species[j]) * nR];
            real *row_j = &radial[((size_t)j * S + o->species[i]) * nR];
            for (int k = 0; k < nR; k++) {
                real sh = r - o->rad_rs[k];
                real v = fc * R_EXP(-o->rad_eta[k] * sh * sh);
                row_i[k] += v;
                row_j[k] += v;
            }
        }
    }
    /* freeze the per-atom lists (ascending neighbour index, as the push order yields) */
    int total = 0;
    for (int i = 0; i < N; i++) { o->nbr_start[i] = total; o->nbr_count[i] = o->tmp_len[i]; total += o->tmp_len[i]; }
    o->nbr_start[N] = total;
    if (total > o->nbr_cap) { o->nbr_cap = total; o->nbr = (int *)realloc(o->nbr, sizeof(int) * (total ? total : 1)); }
    for (int i = 0; i < N; i++)
        memcpy(&o->nbr[o->nbr_start[i]], o->tmp_lists[i], sizeof(int) * o->tmp_len[i]);

    /* ---- angular pass, ref :153-194 ---- */
    for (int i = 0; i < N; i++) {
        const int *list = &o->nbr[o->nbr_start[i]];
        const int n = o->nbr_count[i];
        real *out_i = &angular[(size_t)i * nB * nA];
        for (int a = 0; a < n; a++) {
            int j = list[a];
            real dj[3];
            real rj = R_SQRT(displacement(o, &o->pos[3 * i], &o->pos[3 * j], dj));
            real fcj = fcut(rj, o->rc_angular);
            for (int b = a + 1; b < n; b++) {
                int k = list[b];
                real dk[3];
                real rk = R_SQRT(displacement(o, &o->pos[3 * i], &o->pos[3 * k], dk));
                real fck = fcut(rk, o->rc_angular);
                real rmean = 0.5f * (rj + rk);
                real theta = angle_between(o, dj, dk, rj, rk);
                real *blk = &out_i[o->bucket[o->species[j] * S + o->species[k]] * nA];
                for (int m = 0; m < nA; m++) {
                    real cos_term = R_POW(1 + R_COS(theta - o->ang_ths[m]), o->ang_zeta[m]);
                    real sh = rmean - o->ang_rs[m];
                    real exp_term = R_EXP(-o->ang_eta[m] * sh * sh);
                    blk[m] += fcj * fck * cos_term * exp_term;
                }
            }
        }
    }

    /* ---- post-scale, ref :99-109 ---- */
    if (o->torchani) {
        size_t cnt = (size_t)N * S * nR;
        for (size_t q = 0; q < cnt; q++) radial[q] *= 0.25f;
    }
    size_t cnt = (size_t)N * nB * nA;
    for (int m = 0; m < nA; m++) {
        real scale = R_POW(2, 1 - o->ang_zeta[m]);
        for (size_t q = m; q < cnt; q += nA) angular[q] *= scale;
    }
}

void ani_oracle_backward(ani_oracle *o, const real *radial_grad, const real *angular_grad, real *pos_grad) {
    const int N = o->n_atoms, S = o->n_species, nR = o->n_radial, nA = o->n_angular;
    const int nB = S * (S + 1) / 2;
    memset(pos_grad, 0, sizeof(real) * 3 * N);              /* ref :199 */

    /* ---- radial, ref :228-263 ---- */
    const real rc2_rad = o->rc_radial * o->rc_radial;
    const real gscale = o->torchani ? 0.25f : 1.0f;
    for (int i = 0; i < N; i++) {
        for (int j = i + 1; j < N; j++) {
            real d[3];
            real r2 = displacement(o, &o->pos[3 * i], &o->pos[3 * j], d);
            if (!(r2 < rc2_rad)) continue;
            real r = R_SQRT(r2), rinv = 1 / r;
            real fc = fcut(r, o->rc_radial), dfc = fcut_deriv(r, o->rc_radial);
            const real *gi = &radial_grad[((size_t)i * S + o->species[j]) * nR];
            const real *gj = &radial_grad[((size_t)j * S + o->species[i]) * nR];
            for (int k = 0; k < nR; k++) {
                real sh = r - o->rad_rs[k];
                real e = R_EXP(-o->rad_eta[k] * sh * sh);
                real dvdr = dfc * e - fc * 2 * o->rad_eta[k] * sh * e;
                real dedv = gi[k] + gj[k];
                real sc = gscale * dedv * dvdr * rinv;
                for (int c = 0; c < 3; c++) {
                    real t = sc * d[c];
                    pos_grad[3 * i + c] -= t;
                    pos_grad[3 * j + c] += t;
                }
            }
        }
    }

    /* ---- angular, ref :265-353 ---- */
    for (int i = 0; i < N; i++) {
        const int *list = &o->nbr[o->nbr_start[i]];
        const int n = o->nbr_count[i];
        const real *g_i = &angular_grad[(size_t)i * nB * nA];
        for (int a = 0; a < n; a++) {
            int j = list[a];
            real dj[3];
            real rj = R_SQRT(displacement(o, &o->pos[3 * i], &o->pos[3 * j], dj));
            real rinv_j = 1 / rj;
            real fcj = fcut(rj, o->rc_angular), dfcj = fcut_deriv(rj, o->rc_angular);
            for (int b = a + 1; b < n; b++) {
                int k = list[b];
                real dk[3];
                real rk = R_SQRT(displacement(o, &o->pos[3 * i], &o->pos[3 * k], dk));
                real rinv_k = 1 / rk;
                real fck = fcut(rk, o->rc_angular), dfck = fcut_deriv(rk, o->rc_angular);
                real rmean = 0.5f * (rj + rk);
                real theta = angle_between(o, dj, dk, rj, rk);
                real gj[3], gk[3];
                angle_gradients(o, dj, dk, rj, rk, gj, gk);
                const real *g = &g_i[o->bucket[o->species[j] * S + o->species[k]] * nA];
                for (int m = 0; m < nA; m++) {
                    real zeta = o->ang_zeta[m], ths = o->ang_ths[m];
                    real cos_term = R_POW(1 + R_COS(theta - ths), zeta);
                    real sh = rmean - o->ang_rs[m];
                    real exp_term = R_EXP(-o->ang_eta[m] * sh * sh);
                    real dexp = -o->ang_eta[m] * sh * exp_term;   /* half of 2*eta: rmean carries 1/2 (ref :306) */
                    real dedv = g[m];
                    real zscale = R_POW(2, 1 - zeta);
                    {   /* via r_ij, ref :311-320 */
                        real dvdr = dfcj * fck * cos_term * exp_term + fcj * fck * cos_term * dexp;
                        real sc = zscale * dedv * dvdr * rinv_j;
                        for (int c = 0; c < 3; c++) {
                            real t = sc * dj[c];
                            pos_grad[3 * i + c] -= t;
                            pos_grad[3 * j + c] += t;
                        }
                    }
                    {   /* via r_ik, ref :324-332 */
                        real dvdr = fcj * dfck * cos_term * exp_term + fcj * fck * cos_term * dexp;
                        real sc = zscale * dedv * dvdr * rinv_k;
                        for (int c = 0; c < 3; c++) {
                            real t = sc * dk[c];
                            pos_grad[3 * i + c] -= t;
                            pos_grad[3 * k + c] += t;
                        }
                    }
                    {   /* via the angle, ref :336-348 */
                        real dcos = -zeta * R_POW(1 + R_COS(theta - ths), zeta - 1) * R_SIN(theta - ths);
                        real dvda = fcj * fck * dcos * exp_term;
                        real s2 = zscale * dedv * dvda;
                        real s3 = zscale * dedv * dvda;
                        for (int c = 0; c < 3; c++) {
                            real t2 = s2 * gj[c];
                            real t3 = s3 * gk[c];
                            pos_grad[3 * j + c] += t2;
                            pos_grad[3 * k + c] += t3;
                            pos_grad[3 * i + c] -= t2 + t3;
                        }
                    }
                }
            }
        }
    }
}

/* Introspection used by the tests (neighbour counts drive the triple statistics in DESIGN.md). */
int ani_oracle_angular_neighbor_count(const ani_oracle *o, int atom) { return o->nbr_count[atom]; }
const int *ani_oracle_angular_neighbors(const ani_oracle *o, int atom) { return &o->nbr[o->nbr_start[atom]]; }
