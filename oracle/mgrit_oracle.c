/*
 * mgrit_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle + bench.py's cpu_baseline "port").
 *
 * Plain-C restatement of PyMGRIT's MGRIT hot path (reference: /root/reference, v1.0.6):
 *   src/pymgrit/core/mgrit.py:261-549,715-726,742-858   (cycle, relaxation, FAS residual, layout)
 *   src/pymgrit/heat/heat_1d.py:177-217                 (Heat1D backward-Euler step)
 *   src/pymgrit/advection/advection_1d.py:101-143       (Advection1D backward-Euler step)
 *   src/pymgrit/dahlquist/dahlquist.py:88-111           (Dahlquist steps)
 *   src/pymgrit/heat/heat_2d.py:250-366                 (Heat2D theta-scheme step)
 *   examples/example_spatial_coarsening.py:33-82        (full-weighting / linear transfer)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file. The product
 * (pymgrit_amd/ + libmgrit_hip.so) never links, imports or calls it.
 *
 * Third-party arithmetic: the reference's Phi is scipy.sparse.linalg.spsolve -> SuperLU (scipy>=1.4.1,
 * installed 1.15.3), which is not under /root/reference. Two restatements of the same linear solve live here:
 *   variant 0 "natural": textbook Thomas / forward substitution (pinned against the reference's KATs and
 *                        reference-generated fixtures in tests/golden to forward-error tolerance);
 *   variant 1 "spec":    the chunked-scan arithmetic specification of DESIGN.md section 3 (E=16 elements per
 *                        chunk, 64 chunks per group, row-wise Kogge-Stone + two row broadcasts inside a group, serial carry across groups),
 *                        written independently of the HIP kernels, which must reproduce it bit for bit.
 * Parity pinned: yes (tests/test_oracle_golden.py vs the json/npz fixtures in tests/golden).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: every fused multiply-add below is an explicit fma()).
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_E 16      /* elements per chunk (one GPU lane)      */
#define ORC_LANES 64  /* chunks per group   (one GPU wavefront) */
#define ORC_GROUP (ORC_E * ORC_LANES)
#define ORC_MAX_LEVELS 16
#define ORC_MAX_K 4
#define ORC_MAX_G 64       /* groups of 1024 values per state the spec variant covers (n <= 65536) */

/* ================================================================================================
 * Layout  (mgrit.py:742-838)
 * ============================================================================================== */
typedef struct {
    int32_t n_local;      /* len(t[lvl]) including the ghost point (mgrit.py:783-792) */
    int32_t ghost;        /* 1 when a ghost point precedes the owned block             */
    int32_t first_owned;  /* global index of the first owned point, -1 if none         */
    int32_t n_owned;
    int32_t comm_front, comm_back;
    int32_t first_is_c_point, first_is_f_point, last_is_c_point, last_is_f_point;
    int32_t send_to, get_from;
    int32_t n_c, n_f;
    int32_t m;            /* np.diff(cpts)[0] of the global C-point set (mgrit.py:213), 1 on the coarsest */
} orc_layout_info;

/* mgrit.py:829-838 */
void orc_split_into(int number_points, int number_processes, int32_t *out) {
    int q = number_points / number_processes, r = number_points % number_processes;
    for (int p = 0; p < number_processes; ++p) out[p] = (p < r) ? q + 1 : q;
}

static int member_sorted(const double *a, int n, double v) { /* exact float membership (np.in1d) */
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) / 2; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo < n && a[lo] == v;
}

static int searchsorted_left(const double *a, int n, double v) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) / 2; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

/* is_c for the global level-lvl grid: mgrit.py:767-770 */
static void global_is_c(int n_levels, const int32_t *nt, const double *t_all, const int64_t *t_off, int lvl, uint8_t *is_c) {
    const double *t = t_all + t_off[lvl];
    if (lvl == n_levels - 1) { for (int i = 0; i < nt[lvl]; ++i) is_c[i] = 1; return; }
    const double *tc = t_all + t_off[lvl + 1];
    for (int i = 0; i < nt[lvl]; ++i) is_c[i] = (uint8_t)member_sorted(tc, nt[lvl + 1], t[i]);
}

/*
 * Per-(rank, level) layout. t_all holds the level grids back to back (offsets t_off).
 * Outputs (caller-allocated, capacity nt[lvl]+1): cpts (global indices of owned C-points), index_local,
 * index_local_c, index_local_f. index_local_f is emitted in the canonical order "consecutive runs reversed,
 * ascending inside a run" (the reference's exact order is a CPython set-iteration artefact: SURVEY App. A).
 */
int orc_layout(int n_levels, const int32_t *nt, const double *t_all, int lvl, int rank, int size,
               orc_layout_info *info, int64_t *cpts, int64_t *index_local, int64_t *index_local_c,
               int64_t *index_local_f) {
    int64_t t_off[ORC_MAX_LEVELS + 1];
    t_off[0] = 0;
    for (int l = 0; l < n_levels; ++l) t_off[l + 1] = t_off[l] + nt[l];
    const double *t0 = t_all, *t = t_all + t_off[lvl];
    int N = nt[lvl];
    memset(info, 0, sizeof(*info));
    info->send_to = -99; info->get_from = -99; info->first_owned = -1;

    int32_t *split = (int32_t *)malloc(sizeof(int32_t) * (size_t)size);
    orc_split_into(nt[0], size, split);
    /* level-0 block of this rank (mgrit.py:756-762) */
    int64_t first0 = 0;
    for (int p = 0; p < rank; ++p) first0 += split[p];
    if (split[rank] <= 0) { free(split); return -1; } /* more processes than points: reference raises (mgrit.py:127) */
    double int_start = t0[first0], int_stop = t0[first0 + split[rank] - 1];
    /* owned points on this level (mgrit.py:760,764) */
    int a = -1, z = -2;
    if (lvl == 0) { a = (int)first0; z = (int)(first0 + split[rank] - 1); }
    else {
        for (int i = 0; i < N; ++i) if (t[i] >= int_start && t[i] <= int_stop) { if (a < 0) a = i; z = i; }
        if (a < 0) { a = 0; z = -1; }
    }
    int n_owned = z - a + 1;
    uint8_t *is_c = (uint8_t *)malloc((size_t)N + 1);
    global_is_c(n_levels, nt, t_all, t_off, lvl, is_c);
    /* m (mgrit.py:211-219) */
    if (lvl < n_levels - 1) {
        int c0 = -1, c1 = -1;
        for (int i = 0; i < N && c1 < 0; ++i) if (is_c[i]) { if (c0 < 0) c0 = i; else c1 = i; }
        info->m = (c1 >= 0) ? c1 - c0 : 0;
    } else info->m = 1;

    int ghost = (rank != 0 && n_owned > 0) ? 1 : 0; /* mgrit.py:783 */
    info->ghost = ghost; info->n_owned = n_owned; info->n_local = n_owned + ghost;
    info->first_owned = n_owned > 0 ? a : -1;
    int nc = 0, nf = 0;
    for (int i = a; i <= z; ++i) {
        index_local[i - a] = ghost + (i - a);
        if (is_c[i]) { cpts[nc] = i; index_local_c[nc] = ghost + (i - a); ++nc; }
    }
    /* F-points: runs of consecutive owned F indices, runs reversed (mgrit.py:773-776, canonical order) */
    {
        int i = z;
        while (i >= a) {
            if (is_c[i]) { --i; continue; }
            int e = i; while (i - 1 >= a && !is_c[i - 1]) --i;
            for (int j = i; j <= e; ++j) index_local_f[nf++] = ghost + (j - a);
            --i;
        }
    }
    info->n_c = nc; info->n_f = nf;
#define IS_F(i) ((i) >= 0 && (i) < N && !is_c[i])
#define IS_C(i) ((i) >= 0 && (i) < N && is_c[i])
    if (n_owned > 0) {
        int fmin = -1, fmax = -1;
        for (int i = a; i <= z; ++i) if (!is_c[i]) { if (fmin < 0) fmin = i; fmax = i; }
        info->comm_front = (fmin >= 0) && IS_F(fmin - 1);              /* mgrit.py:796 */
        info->comm_back = (fmax >= 0) && IS_F(fmax + 1);               /* mgrit.py:797 */
        info->first_is_c_point = is_c[a] && a != 0 && IS_F(a - 1);     /* mgrit.py:805-806 */
        info->first_is_f_point = !is_c[a] && IS_C(a - 1);              /* mgrit.py:807 */
        info->last_is_c_point = is_c[z] && z != N - 1 && IS_F(z + 1);  /* mgrit.py:808-810 */
        info->last_is_f_point = !is_c[z] && z != N - 1 && IS_C(z + 1); /* mgrit.py:811-813 */
    }
    /* send_to / get_from (mgrit.py:816-827) */
    if (info->n_local > 0) {
        double *ends = (double *)malloc(sizeof(double) * (size_t)size);
        int64_t acc = 0;
        for (int p = 0; p < size; ++p) { acc += split[p]; ends[p] = t0[acc - 1]; }
        if (z != N - 1) info->send_to = searchsorted_left(ends, size, t[z + 1]);
        double tfirst = t[ghost ? a - 1 : a];
        if (ghost || tfirst != t0[0]) info->get_from = searchsorted_left(ends, size, tfirst);
        free(ends);
    }
    free(is_c); free(split);
    return 0;
}

/* ================================================================================================
 * Coefficient sets for the "spec" variant
 * ============================================================================================== */
typedef struct {
    double dt;
    double rho, ik, scal;   /* heat: rho, 1/kappa ; advection: r, 1/D */
    double pw[ORC_E + 1];   /* pw[k] = rho^k, pw[0] = 1, sequential products */
    double sc[6];           /* sc[s] = rho^(E*2^s), repeated squaring of pw[E] */
    double gc;              /* rho^(64E) = sc[5]^2 */
    double gcp[6];          /* gc^(2^s): coefficients of the cross-group Kogge-Stone scans (4 stages cover 16 groups, 6 cover
                               the 64 of a wide state: the stages beyond a state's groups add exact zeros) */
    double lp[ORC_LANES];   /* lp[l] = rho^(E*l), sequential products of pw[E] */
    double *tab;            /* heat: wg[j] (n) ; advection: rp[j] = r^(j+1) (n) */
    double pt_full[ORC_GROUP]; /* heat: local backward scan of rho^(j'+1) over a full group of 1024           */
    double pt_last[ORC_GROUP]; /* same for the last (possibly partial) group, zero beyond its length          */
    /* tables of the overlapped chain, built on first use: [0] full group, [1] last group */
    double *ch;                /* q1[2][GROUP], v1[2][GROUP], v2[2][GROUP], then v3[padded n] */
    double ca1[2], cb1[2], ca2[2], cb2[2], ca3[16], cb3[16];
} orc_cset;

/* Pt_j' = sum_{i>=j'} rho^(i-j') * rho^(i+1) over a group of len elements: serial recurrences in double */
static void cset_pt(const orc_cset *c, int len, double *pt) {
    double q[ORC_GROUP];
    double p = c->rho;
    for (int j = 0; j < ORC_GROUP; ++j) { q[j] = (j < len) ? p : 0.0; p = p * c->rho; }
    double z = q[ORC_GROUP - 1];
    pt[ORC_GROUP - 1] = z;
    for (int j = ORC_GROUP - 2; j >= 0; --j) { z = fma(c->rho, z, q[j]); pt[j] = z; }
}

static void cset_powers(orc_cset *c, double rho) {
    c->rho = rho;
    c->pw[0] = 1.0; c->pw[1] = rho;
    for (int k = 2; k <= ORC_E; ++k) c->pw[k] = c->pw[k - 1] * rho;
    c->sc[0] = c->pw[ORC_E];
    for (int s = 1; s < 6; ++s) c->sc[s] = c->sc[s - 1] * c->sc[s - 1];
    c->gc = c->sc[5] * c->sc[5];
    c->gcp[0] = c->gc;
    for (int s = 1; s < 6; ++s) c->gcp[s] = c->gcp[s - 1] * c->gcp[s - 1];
    c->lp[0] = 1.0;
    for (int l = 1; l < ORC_LANES; ++l) c->lp[l] = c->lp[l - 1] * c->pw[ORC_E];
}

/* r^e, e >= 0: binary powering, bits of e from the lowest up */
static double ipow(double r, int e) {
    double acc = 1.0, sq = r;
    for (; e > 0; e >>= 1) {
        if (e & 1) acc = acc * sq;
        if (e > 1) sq = sq * sq;
    }
    return acc;
}

/* Heat1D: T = tridiag(-beta, D, -beta) = kappa (I - rho S)(I - rho S^T) + kappa rho^2 e0 e0^T */
static void cset_heat1d(orc_cset *c, int n, double fac, double dt) {
    double beta = dt * fac;
    double D = dt * (2.0 * fac) + 1.0;           /* heat_1d.py:213 diag entry: dt*(2*fac) + 1 */
    double s = sqrt((D - 2.0 * beta) * (D + 2.0 * beta));
    double kappa = 0.5 * (D + s);
    double rho = beta / kappa;
    c->dt = dt; c->ik = 1.0 / kappa; c->scal = 0.0;
    cset_powers(c, rho);
    /* rank-one correction table, closed form: wg_j = gamma * (A^{-1} e0)_j with
       (A^{-1} e0)_j = (rho^j - rho^(2n-j)) / (kappa (1 - rho^2)); rho^j and rho^(2n-j) are assembled from the group, lane and
       element powers so that every entry is  fma(-Q_t, rho^(15-k), P_t * rho^k)  for element k of thread t = j / 16 */
    c->tab = (double *)malloc(sizeof(double) * (size_t)n);
    double one_minus_r2 = (1.0 - rho) * (1.0 + rho);
    double kr2 = beta * rho;
    double w0 = ((1.0 - ipow(rho, 2 * n)) * c->ik) / one_minus_r2;
    double gamma = kr2 / (1.0 + kr2 * w0);
    double gprime = (gamma * c->ik) / one_minus_r2;
    double grp[ORC_MAX_G];                           /* gc^g, sequential products */
    grp[0] = 1.0;
    for (int g = 1; g < ORC_MAX_G; ++g) grp[g] = grp[g - 1] * c->gc;
    int last_thread = (n - 1) / ORC_E;               /* thread that holds element n-1 */
    int g_last = last_thread / ORC_LANES, l_last = last_thread % ORC_LANES;
    int e_base = 2 * n - ORC_E * last_thread - (ORC_E - 1);   /* exponent of rho^(2n-j) at the last thread's element 15 */
    double r_base;
    if (e_base >= 0) r_base = ipow(rho, e_base);
    else r_base = (rho >= 1e-12) ? 1.0 / ipow(rho, -e_base) : 0.0;
    double q_base = gprime * r_base;
    for (int j = 0; j < n; ++j) {
        int thread = j / ORC_E, k = j % ORC_E, g = thread / ORC_LANES, l = thread % ORC_LANES;
        double pfac = (gprime * grp[g]) * c->lp[l];
        double qgrp;
        if (l <= l_last) qgrp = (g <= g_last) ? q_base * grp[g_last - g] : 0.0;
        else qgrp = (g < g_last) ? q_base * grp[g_last - g - 1] : 0.0;
        double qfac = qgrp * c->lp[(l_last - l) & (ORC_LANES - 1)];
        c->tab[j] = fma(-qfac, c->pw[ORC_E - 1 - k], pfac * c->pw[k]);
    }
    cset_pt(c, ORC_GROUP, c->pt_full);
    cset_pt(c, n - ((n - 1) / ORC_GROUP) * ORC_GROUP, c->pt_last);
}

/* Advection1D: (1+alpha) x_j - alpha x_{j-1 mod n} = u_j ; r = alpha/D, x_j = y_j + r^(j+1) x_{n-1} */
static void cset_advection1d(orc_cset *c, int n, double fac, double dt) {
    double alpha = dt * fac;
    double D = alpha + 1.0;                      /* advection_1d.py:140 diag: dt*fac + 1 */
    double r = alpha / D;
    c->dt = dt; c->ik = 1.0 / D;
    cset_powers(c, r);
    c->tab = (double *)malloc(sizeof(double) * (size_t)n);
    double p = r;
    for (int j = 0; j < n; ++j) { c->tab[j] = p; p = p * r; }  /* tab[j] = r^(j+1) */
    c->scal = 1.0 / (1.0 - c->tab[n - 1]);                     /* 1/(1 - r^n) */
}

/* ================================================================================================
 * Group-local chunked scans (spec). A group = 64 lanes x 16 elements = 1024 consecutive values (one GPU
 * wavefront). Forward: y_j = rho*y_{j-1} + d_j with ZERO carry into the group; backward mirrored. Returns the group
 * total (last element of the forward scan / first element of the backward scan).
 * ============================================================================================== */
static double group_scan_fwd(const orc_cset *c, double *y) {
    double S[ORC_LANES], T[ORC_LANES];
    for (int l = 0; l < ORC_LANES; ++l) {
        double *b = y + (size_t)l * ORC_E;
        for (int k = 1; k < ORC_E; ++k) b[k] = fma(c->rho, b[k - 1], b[k]);
        S[l] = b[ORC_E - 1];
    }
    /* Kogge-Stone inside each row of 16 lanes (offsets 1,2,4,8), then two row broadcasts */
    for (int s = 0; s < 4; ++s) {
        int off = 1 << s;
        for (int l = 0; l < ORC_LANES; ++l) T[l] = ((l & 15) >= off) ? fma(c->sc[s], S[l - off], S[l]) : S[l];
        memcpy(S, T, sizeof(S));
    }
    for (int l = 0; l < ORC_LANES; ++l) /* rows 1 and 3 take the last lane of the row below */
        T[l] = ((l >> 4) & 1) ? fma(c->lp[(l & 15) + 1], S[(l & ~15) - 1], S[l]) : S[l];
    memcpy(S, T, sizeof(S));
    for (int l = 0; l < ORC_LANES; ++l) /* lanes 32..63 take lane 31 */
        T[l] = (l >= 32) ? fma(c->lp[l - 31], S[31], S[l]) : S[l];
    memcpy(S, T, sizeof(S));
    for (int l = 0; l < ORC_LANES; ++l) {
        double prev = l > 0 ? S[l - 1] : 0.0;
        double *b = y + (size_t)l * ORC_E;
        for (int k = 0; k < ORC_E; ++k) b[k] = fma(c->pw[k + 1], prev, b[k]);
    }
    return S[ORC_LANES - 1];
}

static double group_scan_bwd(const orc_cset *c, double *z) {
    double S[ORC_LANES], T[ORC_LANES];
    for (int l = 0; l < ORC_LANES; ++l) {
        double *b = z + (size_t)l * ORC_E;
        for (int k = ORC_E - 2; k >= 0; --k) b[k] = fma(c->rho, b[k + 1], b[k]);
        S[l] = b[0];
    }
    for (int s = 0; s < 4; ++s) {
        int off = 1 << s;
        for (int l = 0; l < ORC_LANES; ++l) T[l] = ((l & 15) + off < 16) ? fma(c->sc[s], S[l + off], S[l]) : S[l];
        memcpy(S, T, sizeof(S));
    }
    for (int l = 0; l < ORC_LANES; ++l) /* rows 0 and 2 take the first lane of the row above */
        T[l] = (((l >> 4) & 1) == 0) ? fma(c->lp[16 - (l & 15)], S[(l & ~15) + 16], S[l]) : S[l];
    memcpy(S, T, sizeof(S));
    for (int l = 0; l < ORC_LANES; ++l) /* lanes 0..31 take lane 32 */
        T[l] = (l < 32) ? fma(c->lp[32 - l], S[32], S[l]) : S[l];
    memcpy(S, T, sizeof(S));
    for (int l = 0; l < ORC_LANES; ++l) {
        double next = l < ORC_LANES - 1 ? S[l + 1] : 0.0;
        double *b = z + (size_t)l * ORC_E;
        for (int k = 0; k < ORC_E; ++k) b[k] = fma(c->pw[ORC_E - k], next, b[k]);
    }
    return S[0];
}

/* sum of squares with the spec's reduction tree: lane-local fma chain, xor-butterfly over 64 lanes, serial
 * sum over groups (mirrors the HIP kernels; replaces np.linalg.norm's BLAS ddot, vector.norm()). */
double orc_sumsq_spec(const double *r, int n) {
    int G = (n + ORC_GROUP - 1) / ORC_GROUP;
    double tot = 0.0;
    for (int g = 0; g < G; ++g) {
        double a[ORC_LANES], b[ORC_LANES];
        for (int l = 0; l < ORC_LANES; ++l) {
            double acc = 0.0;
            for (int k = 0; k < ORC_E; ++k) {
                int j = (g * ORC_LANES + l) * ORC_E + k;
                if (j < n) acc = fma(r[j], r[j], acc);
            }
            a[l] = acc;
        }
        for (int off = 32; off >= 1; off >>= 1) {
            for (int l = 0; l < ORC_LANES; ++l) b[l] = a[l] + a[l ^ off];
            memcpy(a, b, sizeof(a));
        }
        tot = tot + a[0];
    }
    return tot;
}

/* two-point states [first | second] (vector_heat_1d_2pts.py:66-72, norm of the appended halves): the groups of `first`,
 * then the groups of `second`, one serial sum */
static double sumsq_groups(const double *r, int n, double tot) {
    int G = (n + ORC_GROUP - 1) / ORC_GROUP;
    for (int g = 0; g < G; ++g) {
        int len = n - g * ORC_GROUP < ORC_GROUP ? n - g * ORC_GROUP : ORC_GROUP;
        tot = tot + orc_sumsq_spec(r + (size_t)g * ORC_GROUP, len);
    }
    return tot;
}
double orc_sumsq_spec_2pts(const double *r, int n) { return sumsq_groups(r + n, n, sumsq_groups(r, n, 0.0)); }

/* ================================================================================================
 * Steppers
 * ============================================================================================== */
enum { ORC_DAHLQUIST = 0, ORC_HEAT1D = 1, ORC_ADVECTION1D = 2, ORC_HEAT2D = 3, ORC_HEAT1D_2PTS = 4 };
enum { ORC_BE = 0, ORC_FE = 1, ORC_TR = 2, ORC_MR = 3 };

typedef struct {
    int kind, variant, n;
    double fac;            /* heat: a/dx^2 ; advection: c/dx ; dahlquist: lambda */
    int method;            /* dahlquist */
    int K;                 /* separable forcing terms: b(x,t_i) = sum_k s_k(x)*tau_k(t_i) */
    double *s;             /* [K][n] */
    double *tau;           /* [K][nt] */
    double *tau2;          /* two-point steppers: [K][nt] tau_k(t_i + dtau) */
    double *frows;         /* heat1d, general (non-separable) forcing: [nt][n] rows rhs(x, t_i) * (t_i - t_{i-1}), the very
                              product heat_1d.py:213 adds to u_start (row 0 unused); NULL: forcing given by s / tau */
    double dtau;           /* two-point steppers: spacing inside a pair; method = BDF order (1 or 2) */
    orc_cset *csets; int n_csets, cap_csets;
    double *w1, *w2;       /* work (padded) */
    /* heat2d (heat_2d.py:147-366): full nx x ny grid incl. the rim; interior mi x mj, padded to Mi x Mj (multiples of 64) */
    int nx, ny, mi, mj, Mi, Mj, has_w;   /* Mi = 2*HPx, Mj = 2*HPy: one padded length for the natural and the spectral layout */
    int HPx, HPy;                        /* padded half sizes: even modes in slots [0, HP), odd modes in [HP, 2 HP) */
    double *Fxe, *Fxo, *Fye, *Fyo;       /* folded sine tables [HP][HP]: Fe[k][e] = Q[k][2e], Fo[k][o] = Q[k][2o+1] */
    double fx, fy, theta;
    double *bc, *W, *lx, *ly, *dinv, *X0, *X1;
    double dinv_dt;
} orc_stepper;

static orc_cset *get_cset(orc_stepper *st, double dt) {
    for (int i = 0; i < st->n_csets; ++i)
        if (memcmp(&st->csets[i].dt, &dt, sizeof(double)) == 0) return &st->csets[i];
    if (st->n_csets == st->cap_csets) {
        st->cap_csets = st->cap_csets ? 2 * st->cap_csets : 4;
        st->csets = (orc_cset *)realloc(st->csets, sizeof(orc_cset) * (size_t)st->cap_csets);
    }
    orc_cset *c = &st->csets[st->n_csets++];
    memset(c, 0, sizeof(*c));
    if (st->kind == ORC_ADVECTION1D) cset_advection1d(c, st->n, st->fac, dt);
    else cset_heat1d(c, st->n, st->fac, dt);
    return c;
}

static int padded(int n) { return ((n + ORC_GROUP - 1) / ORC_GROUP) * ORC_GROUP; }

/* heat_1d.py:198-217:  spsolve(dt*L + I, u + rhs(x, t_stop)*dt) */
static void heat1d_rhs(const orc_stepper *st, int nt, int i_stop, double dt, const double *u, double *d) {
    int n = st->n;
    if (st->frows) { for (int j = 0; j < n; ++j) d[j] = u[j] + st->frows[(size_t)i_stop * n + j]; return; }
    if (st->K == 0) { memcpy(d, u, sizeof(double) * (size_t)n); return; }
    for (int j = 0; j < n; ++j) {
        double f = st->s[j] * st->tau[i_stop];
        for (int k = 1; k < st->K; ++k) f = f + st->s[(size_t)k * n + j] * st->tau[(size_t)k * nt + i_stop];
        d[j] = u[j] + f * dt;
    }
}

/* Thomas algorithm on tridiag(-beta, D, -beta); d is overwritten */
static void thomas_toeplitz(int n, double beta, double D, double *d, double *cp, double *out) {
    double piv = D;
    cp[0] = -beta / piv; d[0] = d[0] / piv;
    for (int j = 1; j < n; ++j) {
        piv = D + beta * cp[j - 1];
        cp[j] = -beta / piv;
        d[j] = (d[j] + beta * d[j - 1]) / piv;
    }
    out[n - 1] = d[n - 1];
    for (int j = n - 2; j >= 0; --j) out[j] = d[j] - cp[j] * out[j + 1];
}

/* scratch passed in: the threaded sweeps of the timing path (orc_problem_set_threads) give every thread its own */
static void heat1d_step_natural_ws(const orc_stepper *st, int nt, int i_stop, double dt, const double *u, double *out,
                                   double *w1, double *w2) {
    double beta = dt * st->fac, D = dt * (2.0 * st->fac) + 1.0;
    heat1d_rhs(st, nt, i_stop, dt, u, w1);
    thomas_toeplitz(st->n, beta, D, w1, w2, out);
}

static void heat1d_step_natural(orc_stepper *st, int nt, int i_stop, double dt, const double *u, double *out) {
    heat1d_step_natural_ws(st, nt, i_stop, dt, u, out, st->w1, st->w2);
}

/* Cross-group carries: inclusive Kogge-Stone scans over the group totals with coefficients gc^(2^s):
 * fwd: I_g = sum_{h<=g} gc^(g-h) A_h ; bwd: J_g = sum_{h>=g} gc^(h-g) A_h. The register-resident kernels hold <= 16 groups in
 * one row of 16 lanes (4 stages); a wide state (17..64 groups, csrc/mgrit_hip_wide.inc) takes the same scan over 64 lanes
 * (6 stages). The arrays are ORC_MAX_G wide with zeros beyond the state's groups, so for <= 16 groups the two extra stages and
 * the entries beyond 16 only ever add fma(coefficient, 0, a) = a: one spec for both. */
static void cross_scan(const orc_cset *c, double *a, int backward) {
    double t[ORC_MAX_G];
    for (int s = 0; s < 6; ++s) {
        int off = 1 << s;
        for (int g = 0; g < ORC_MAX_G; ++g) {
            if (!backward) t[g] = (g >= off) ? fma(c->gcp[s], a[g - off], a[g]) : a[g];
            else t[g] = (g + off < ORC_MAX_G) ? fma(c->gcp[s], a[g + off], a[g]) : a[g];
        }
        memcpy(a, t, sizeof(t));
    }
}

/* Spec variant of the Heat1D step: every group scans locally forward and backward, ONE exchange of the
 * group totals (A_g, B_g), two short carry chains, then one pass that adds the carries and the rank-one correction. */
/* x = (I + dt*L)^{-1} d for the coefficient set c; d (padded work array, first n entries filled) is overwritten */
static void heat_solve_spec(const orc_cset *c, int n, double *d, double *out) {
    int NP = padded(n), G = NP / ORC_GROUP;
    double A[ORC_MAX_G] = {0}, B[ORC_MAX_G] = {0}, C[ORC_MAX_G + 1] = {0}, Zf[ORC_MAX_G + 1] = {0};
    for (int j = n; j < NP; ++j) d[j] = 0.0;
    for (int g = 0; g < G; ++g) {
        double *dg = d + (size_t)g * ORC_GROUP;
        A[g] = group_scan_fwd(c, dg);
        for (int j = 0; j < ORC_GROUP; ++j) if (g * ORC_GROUP + j >= n) dg[j] = 0.0;
        B[g] = group_scan_bwd(c, dg);
    }
    cross_scan(c, A, 0);                       /* A[g] := inclusive forward scan */
    C[0] = 0.0;
    for (int g = 1; g < ORC_MAX_G; ++g) C[g] = A[g - 1];
    for (int g = 0; g < ORC_MAX_G; ++g)
        Zf[g] = (g < G) ? fma(C[g], (g == G - 1) ? c->pt_last[0] : c->pt_full[0], B[g]) : 0.0;
    cross_scan(c, Zf, 1);                      /* Zf[g] := true z at the first element of group g */
    Zf[ORC_MAX_G] = 0.0;
    double z0 = Zf[0] * c->ik;
    for (int g = 0; g < G; ++g) {
        const double *pt = (g == G - 1) ? c->pt_last : c->pt_full;
        for (int l = 0; l < ORC_LANES; ++l) {
            double cb = c->lp[ORC_LANES - 1 - l] * Zf[g + 1];
            for (int k = 0; k < ORC_E; ++k) {
                int jj = l * ORC_E + k, j = g * ORC_GROUP + jj;
                if (j >= n) continue;
                double t = fma(C[g], pt[jj], d[j]);
                double z = fma(c->pw[ORC_E - k], cb, t);
                out[j] = fma(-z0, c->tab[j], z * c->ik);
            }
        }
    }
}

/* (d: padded scratch; the threaded sweeps of orc_problem_set_threads give every thread its own) */
static void heat1d_step_spec_ws(orc_stepper *st, int nt, int i_stop, double dt, const double *u, double *out, double *d) {
    int n = st->n;
    orc_cset *c = get_cset(st, dt);
    /* d = u + dt*b(x, t_i) with the forcing folded as fma(s_k, tau_k*dt, .); a general forcing: d = u + (rhs*dt) row */
    for (int j = 0; j < n; ++j) {
        double v = u[j];
        if (st->frows) v = v + st->frows[(size_t)i_stop * n + j];
        for (int k = 0; k < st->K; ++k) v = fma(st->s[(size_t)k * n + j], st->tau[(size_t)k * nt + i_stop] * dt, v);
        d[j] = v;
    }
    heat_solve_spec(c, n, d, out);
}

static void heat1d_step_spec(orc_stepper *st, int nt, int i_stop, double dt, const double *u, double *out) {
    heat1d_step_spec_ws(st, nt, i_stop, dt, u, out, st->w1);
}

/* ------------------------------------------------------------------------------------------------
 * Two-point steppers (heat/heat_1d_2pts_bdf1.py:84-117, heat/heat_1d_2pts_bdf2.py:82-138): a state is the pair
 * (first, second) = values at t and t + dtau, stored [first | second], n = st->n values each.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { double dt_eff, a, nb, fs; } orc_half;  /* spec coefficients of one half-solve */

/* BDF2 coefficients of heat_1d_2pts_bdf2.py:103-110 for (tau_i, tau_im1), rewritten for the scaled system
 * (I + L/c) x = rhs/c : dt_eff = 1/c, a = cm1/c, nb = -cm2/c, forcing scale fs = 1/c */
static orc_half bdf2_half(double tau_i, double tau_im1) {
    orc_half h;
    double r = tau_i / tau_im1;
    double cm2 = (r * r) / (tau_i * (1.0 + r));
    double cm1 = (1.0 + r) / tau_i;
    double c = (1.0 + 2.0 * r) / (tau_i * (1.0 + r));
    double inv = 1.0 / c;
    h.dt_eff = inv; h.a = cm1 * inv; h.nb = -(cm2 * inv); h.fs = inv;
    return h;
}

static void twopts_step_natural(orc_stepper *st, int nt, int i_stop, double t_start, double t_stop, const double *u, double *out) {
    int n = st->n;
    const double *first = u, *second = u + n;
    double *o1 = out, *o2 = out + n, *d = st->w1;
    double dtau = st->dtau;
    for (int half = 0; half < 2; ++half) {
        const double *tau = half ? st->tau2 : st->tau;
        const double *x_old = half ? second : first, *x_new = half ? o1 : second;  /* values two / one point back */
        if (st->method == 1) {  /* spsolve(tau*L + I, x_new + rhs*tau) */
            double tl = half ? dtau : t_stop - t_start - dtau;
            for (int j = 0; j < n; ++j) {
                double f = 0.0;
                if (st->K > 0) { f = st->s[j] * tau[i_stop]; for (int k = 1; k < st->K; ++k) f = f + st->s[(size_t)k * n + j] * tau[(size_t)k * nt + i_stop]; }
                d[j] = st->K > 0 ? x_new[j] + f * tl : x_new[j];
            }
            thomas_toeplitz(n, tl * st->fac, tl * (2.0 * st->fac) + 1.0, d, st->w2, half ? o2 : o1);
        } else {                /* spsolve(L + coeff*I, rhs - coeffm2*x_old + coeffm1*x_new) */
            double tau_i = half ? dtau : t_stop - t_start - dtau, tau_im1 = half ? t_stop - t_start - dtau : dtau;
            double r = tau_i / tau_im1;
            double cm2 = (r * r) / (tau_i * (1.0 + r)), cm1 = (1.0 + r) / tau_i, c = (1.0 + 2.0 * r) / (tau_i * (1.0 + r));
            for (int j = 0; j < n; ++j) {
                double f = 0.0;
                if (st->K > 0) { f = st->s[j] * tau[i_stop]; for (int k = 1; k < st->K; ++k) f = f + st->s[(size_t)k * n + j] * tau[(size_t)k * nt + i_stop]; }
                d[j] = f - cm2 * x_old[j] + cm1 * x_new[j];
            }
            thomas_toeplitz(n, st->fac, 2.0 * st->fac + c, d, st->w2, half ? o2 : o1);
        }
    }
}

/* Spec variant: each half-solve is base -> forcing fma chain -> heat_solve_spec with the coefficient set of dt_eff.
 * BDF1: base = x_new, forcing scale = tau_l (the half's step). BDF2: base = fma(a, x_new, nb*x_old), see bdf2_half. */
static void twopts_step_spec(orc_stepper *st, int nt, int i_stop, double t_start, double t_stop, const double *u, double *out) {
    int n = st->n;
    const double *first = u, *second = u + n;
    double *o1 = out, *o2 = out + n, *d = st->w1;
    double dtau = st->dtau, tl0 = t_stop - t_start - dtau;
    for (int half = 0; half < 2; ++half) {
        const double *tau = half ? st->tau2 : st->tau;
        const double *x_old = half ? second : first, *x_new = half ? o1 : second;
        orc_half h;
        if (st->method == 1) { h.dt_eff = half ? dtau : tl0; h.a = 1.0; h.nb = 0.0; h.fs = h.dt_eff; }
        else h = half ? bdf2_half(dtau, tl0) : bdf2_half(tl0, dtau);
        for (int j = 0; j < n; ++j) {
            double v = st->method == 1 ? x_new[j] : fma(h.a, x_new[j], h.nb * x_old[j]);
            for (int k = 0; k < st->K; ++k) v = fma(st->s[(size_t)k * n + j], tau[(size_t)k * nt + i_stop] * h.fs, v);
            d[j] = v;
        }
        heat_solve_spec(get_cset(st, h.dt_eff), n, d, half ? o2 : o1);
    }
}

/* advection_1d.py:129-143: spsolve(dt*L + I, u), L = (c/dx)(I - S_periodic) */
static void advection1d_step_natural_ws(orc_stepper *st, double dt, const double *u, double *out, double *p, double *q) {
    int n = st->n;
    double alpha = dt * st->fac, D = alpha + 1.0;
    p[0] = u[0] / D; q[0] = alpha / D;
    for (int j = 1; j < n; ++j) { p[j] = (u[j] + alpha * p[j - 1]) / D; q[j] = alpha * q[j - 1] / D; }
    double xl = p[n - 1] / (1.0 - q[n - 1]);
    for (int j = 0; j < n - 1; ++j) out[j] = p[j] + q[j] * xl;
    out[n - 1] = xl;
}

static void advection1d_step_natural(orc_stepper *st, double dt, const double *u, double *out) {
    advection1d_step_natural_ws(st, dt, u, out, st->w1, st->w2);
}

static void advection1d_step_spec_ws(orc_stepper *st, double dt, const double *u, double *out, double *d) {
    int n = st->n, NP = padded(n), G = NP / ORC_GROUP;
    orc_cset *c = get_cset(st, dt);
    double A[ORC_MAX_G] = {0}, C[ORC_MAX_G + 1] = {0};
    for (int j = 0; j < n; ++j) d[j] = u[j] * c->ik;
    for (int j = n; j < NP; ++j) d[j] = 0.0;
    for (int g = 0; g < G; ++g) A[g] = group_scan_fwd(c, d + (size_t)g * ORC_GROUP);
    cross_scan(c, A, 0);
    C[0] = 0.0;
    for (int g = 1; g < ORC_MAX_G; ++g) C[g] = A[g - 1];
    /* periodic closure from the last real element n-1 (lane l*, element k* of group G-1) */
    int jl = n - 1, gl = jl / ORC_GROUP, ll = (jl % ORC_GROUP) / ORC_E, kl = jl % ORC_E;
    double ylast = fma(c->pw[kl + 1], c->lp[ll] * C[gl], d[jl]);
    double xl = ylast * c->scal;
    for (int g = 0; g < G; ++g)
        for (int l = 0; l < ORC_LANES; ++l) {
            double cf = c->lp[l] * C[g];
            for (int k = 0; k < ORC_E; ++k) {
                int j = g * ORC_GROUP + l * ORC_E + k;
                if (j >= n) continue;
                double y = fma(c->pw[k + 1], cf, d[j]);
                out[j] = fma(c->tab[j], xl, y);
            }
        }
}

static void advection1d_step_spec(orc_stepper *st, double dt, const double *u, double *out) {
    advection1d_step_spec_ws(st, dt, u, out, st->w1);
}

/* dahlquist.py:88-111 */
static void dahlquist_step(const orc_stepper *st, double t_start, double t_stop, const double *u, double *out) {
    double z = (t_stop - t_start) * st->fac;
    switch (st->method) {
    case ORC_BE: out[0] = 1.0 / (1.0 - z) * u[0]; break;
    case ORC_FE: out[0] = (1.0 + z) * u[0]; break;
    case ORC_TR: out[0] = (1.0 + z / 2.0) / (1.0 - z / 2.0) * u[0]; break;
    default: { double k1 = -1.0 / (1.0 - z / 2.0) * u[0]; out[0] = u[0] + (t_stop - t_start) * k1; }
    }
}

/* ================================================================================================
 * Heat2D (heat_2d.py:250-366). The reference solves (I + theta*dt*L) u = b with SuperLU on every step; the restatement
 * uses the fast-diagonalisation identity  U = Qx ((Qx B Qy) o D) Qy  on the interior (Qx, Qy: orthogonal symmetric
 * sine-transform matrices; D = 1/(1 + theta*dt*(lx_k + ly_l))). Every product is a dot product accumulated with fma in
 * ascending k from 0 -- exactly what a chain of v_mfma_f64_16x16x4 instructions computes.
 * Layouts: X = Qx.B is stored TRANSPOSED ([j][i']), so all four products read "symmetric matrix times row-major matrix".
 * ============================================================================================== */
/* One even/odd split of the sine transform. Q[i][k] = s sin(pi (i+1)(k+1)/(m+1)) satisfies
 * Q[m-1-i][k] = (-1)^k Q[i][k], so with the folded inputs xe[k] = x[k] + x[m-1-k], xo[k] = x[k] - x[m-1-k] (k < floor(m/2);
 * for odd m the centre element joins xe unchanged) the even modes need only Fe = Q[k][2e] and the odd modes only
 * Fo = Q[k][2o+1], k < ceil(m/2): half the multiplications of the full product. */
static void h2d_axis_tables(int m, int HP, double f, double **Fe, double **Fo, double **lam_spec) {
    int hE = (m + 1) / 2, hO = m / 2;
    double sc = sqrt(2.0 / (m + 1));
    *Fe = (double *)calloc((size_t)HP * HP, sizeof(double));
    *Fo = (double *)calloc((size_t)HP * HP, sizeof(double));
    *lam_spec = (double *)calloc((size_t)2 * HP, sizeof(double));
    for (int k = 0; k < hE; ++k) {
        for (int e = 0; e < hE; ++e) {
            long r = ((long)(k + 1) * (2 * e + 1)) % (2L * (m + 1));
            (*Fe)[(size_t)k * HP + e] = sc * sin(M_PI * (double)r / (double)(m + 1));
        }
        if (k < hO)
            for (int o = 0; o < hO; ++o) {
                long r = ((long)(k + 1) * (2 * o + 2)) % (2L * (m + 1));
                (*Fo)[(size_t)k * HP + o] = sc * sin(M_PI * (double)r / (double)(m + 1));
            }
    }
    for (int e = 0; e < hE; ++e) { double h = sin(M_PI * (double)(2 * e + 1) / (2.0 * (m + 1))); (*lam_spec)[e] = 4.0 * f * h * h; }
    for (int o = 0; o < hO; ++o) { double h = sin(M_PI * (double)(2 * o + 2) / (2.0 * (m + 1))); (*lam_spec)[HP + o] = 4.0 * f * h * h; }
}

static void h2d_tables(orc_stepper *st) {
    int Mi = st->Mi, Mj = st->Mj;
    h2d_axis_tables(st->mi, st->HPx, st->fx, &st->Fxe, &st->Fxo, &st->lx);   /* lx, ly: eigenvalues in spectral slot order */
    h2d_axis_tables(st->mj, st->HPy, st->fy, &st->Fye, &st->Fyo, &st->ly);
    st->dinv = (double *)calloc((size_t)Mi * Mj, sizeof(double));
    st->X0 = (double *)calloc((size_t)Mi * Mj, sizeof(double));
    st->X1 = (double *)calloc((size_t)Mi * Mj, sizeof(double));
    st->dinv_dt = -1.0;
}

/* The two half transforms below are loops over INDEPENDENT columns n, each output a dot product accumulated by fma in ascending k
 * from 0 (the spec). Round 5: the column is gathered (and folded) once into a contiguous buffer, the forward
 * transform walks k in the outer loop over an array of accumulators -- every accumulator still sees its own products in ascending k,
 * so the bits are those of the plain triple loop --, and the columns are spread over orc_h2d_threads OpenMP threads
 * (orc_set_h2d_threads; default 1): a 512 x 512 step falls from 0.5 s to ~10 ms, which is what lets the GPU tests compare BASELINE
 * config 4 with the oracle AT its size (tests/test_hip_heat2d.py). */
static int orc_h2d_threads = 1;
int orc_set_h2d_threads(int threads) {
#ifdef _OPENMP
    orc_h2d_threads = threads > 1 ? threads : 1;
#else
    (void)threads;
#endif
    return orc_h2d_threads;
}

/* forward half transform along the rows of B (natural index k, m real rows, 2 HP stored): out[n][slot], slot = e or HP + o */
static void h2d_fwd(const double *Fe, const double *Fo, int m, int HP, const double *B, int N, double *out) {
    int hE = (m + 1) / 2, hO = m / 2, P = 2 * HP;
#pragma omp parallel num_threads(orc_h2d_threads) if (orc_h2d_threads > 1 && N >= 32)
    {
        double *xe = (double *)malloc(sizeof(double) * 2 * (size_t)(hE + 1)), *xo = xe + hE + 1;
#pragma omp for schedule(static)
        for (int n = 0; n < N; ++n) {
            for (int k = 0; k < hE; ++k) {
                xe[k] = k < hO ? B[(size_t)k * N + n] + B[(size_t)(m - 1 - k) * N + n] : B[(size_t)k * N + n];
                if (k < hO) xo[k] = B[(size_t)k * N + n] - B[(size_t)(m - 1 - k) * N + n];
            }
            double *ce = out + (size_t)n * P, *co = ce + HP;
            for (int e = 0; e < P; ++e) ce[e] = 0.0;
            for (int k = 0; k < hE; ++k) {
                const double *fe = Fe + (size_t)k * HP; double x = xe[k];
                for (int e = 0; e < HP; ++e) ce[e] = fma(fe[e], x, ce[e]);
            }
            for (int k = 0; k < hO; ++k) {
                const double *fo = Fo + (size_t)k * HP; double x = xo[k];
                for (int o = 0; o < HP; ++o) co[o] = fma(fo[o], x, co[o]);
            }
        }
        free(xe);
    }
}

/* inverse half transform along the rows of B (spectral slots): out[n][i] = Pe + Po, out[n][m-1-i] = Pe - Po */
static void h2d_inv(const double *Fe, const double *Fo, int m, int HP, const double *B, int N, double *out) {
    int hE = (m + 1) / 2, hO = m / 2, P = 2 * HP;
#pragma omp parallel num_threads(orc_h2d_threads) if (orc_h2d_threads > 1 && N >= 32)
    {
        double *col = (double *)malloc(sizeof(double) * (size_t)P);
#pragma omp for schedule(static)
        for (int n = 0; n < N; ++n) {
            for (int e = 0; e < P; ++e) col[e] = B[(size_t)e * N + n];
            for (int i = 0; i < P; ++i) out[(size_t)n * P + i] = 0.0;
            for (int i = 0; i < hE; ++i) {
                double pe = 0.0, po = 0.0;
                const double *fe = Fe + (size_t)i * HP, *fo = Fo + (size_t)i * HP;
                for (int e = 0; e < hE; ++e) pe = fma(fe[e], col[e], pe);
                for (int o = 0; o < hO; ++o) po = fma(fo[o], col[HP + o], po);
                out[(size_t)n * P + i] = pe + po;
                if (i < hO) out[(size_t)n * P + (m - 1 - i)] = pe - po;
            }
        }
        free(col);
    }
}

static double h2d_lap(const orc_stepper *st, const double *u, int gi, int gj) {
    int ny = st->ny;
    double acc = (2.0 * (st->fx + st->fy)) * u[(size_t)gi * ny + gj];
    acc = fma(-st->fx, u[(size_t)(gi - 1) * ny + gj], acc);
    acc = fma(-st->fx, u[(size_t)(gi + 1) * ny + gj], acc);
    acc = fma(-st->fy, u[(size_t)gi * ny + gj - 1], acc);
    acc = fma(-st->fy, u[(size_t)gi * ny + gj + 1], acc);
    return acc;
}

static void heat2d_step(orc_stepper *st, int nt, int i_stop, double t_start, double t_stop, const double *u, double *out) {
    int nx = st->nx, ny = st->ny, mi = st->mi, mj = st->mj, Mi = st->Mi, Mj = st->Mj;
    double dt = t_stop - t_start, th = st->theta;
    if (th == 0.0) { /* FE: heat_2d.py:346-356; boundary = BC + old boundary (quirk of the reference) */
        for (int gi = 0; gi < nx; ++gi)
            for (int gj = 0; gj < ny; ++gj) {
                size_t p = (size_t)gi * ny + gj;
                if (gi == 0 || gj == 0 || gi == nx - 1 || gj == ny - 1) { out[p] = st->bc[p] + u[p]; continue; }
                double v = fma(-dt, h2d_lap(st, u, gi, gj), u[p]);
                for (int k = 0; k < st->K; ++k)
                    v = fma(st->s[((size_t)k * mi + (gi - 1)) * mj + (gj - 1)], dt * st->tau[(size_t)k * nt + i_stop - 1], v);
                if (st->frows) v = fma(st->frows[((size_t)(i_stop - 1) * mi + (gi - 1)) * mj + (gj - 1)], dt, v);
                out[p] = v;
            }
        return;
    }
    double thdt = th * dt, thdt1 = (1.0 - th) * dt;
    if (st->dinv_dt != dt) {   /* spectral slot order on both axes; slots without a mode stay 0 */
        int hxe = (mi + 1) / 2, hxo = mi / 2, hye = (mj + 1) / 2, hyo = mj / 2;
        memset(st->dinv, 0, sizeof(double) * (size_t)Mi * Mj);
        for (int a = 0; a < Mi; ++a) {
            if (!((a < hxe) || (a >= st->HPx && a < st->HPx + hxo))) continue;
            for (int b = 0; b < Mj; ++b)
                if ((b < hye) || (b >= st->HPy && b < st->HPy + hyo))
                    st->dinv[(size_t)a * Mj + b] = 1.0 / (1.0 + thdt * (st->lx[a] + st->ly[b]));
        }
        st->dinv_dt = dt;
    }
    double *B = st->X0, *X = st->X1;
    memset(B, 0, sizeof(double) * (size_t)Mi * Mj);
    for (int a = 0; a < mi; ++a)
        for (int b = 0; b < mj; ++b) {
            size_t p = (size_t)(a + 1) * ny + (b + 1);
            double v;
            if (th == 1.0) {
                v = u[p];
                for (int k = 0; k < st->K; ++k) v = fma(st->s[((size_t)k * mi + a) * mj + b], st->tau[(size_t)k * nt + i_stop] * dt, v);
                if (st->frows) v = fma(st->frows[((size_t)i_stop * mi + a) * mj + b], dt, v);
            } else {
                v = fma(-thdt, h2d_lap(st, u, a + 1, b + 1), u[p]);
                for (int k = 0; k < st->K; ++k)
                    v = fma(st->s[((size_t)k * mi + a) * mj + b],
                            thdt * st->tau[(size_t)k * nt + i_stop] + thdt1 * st->tau[(size_t)k * nt + i_stop - 1], v);
                if (st->frows) {
                    v = fma(st->frows[((size_t)i_stop * mi + a) * mj + b], thdt, v);
                    v = fma(st->frows[((size_t)(i_stop - 1) * mi + a) * mj + b], thdt1, v);
                }
            }
            if (st->has_w) v = fma(thdt, st->W[(size_t)a * mj + b], v);
            B[(size_t)a * Mj + b] = v;
        }
    h2d_fwd(st->Fxe, st->Fxo, mi, st->HPx, B, Mj, X);                   /* X[j][i'] : x to spectral slots, transposed */
    h2d_fwd(st->Fye, st->Fyo, mj, st->HPy, X, Mi, B);                   /* B[i'][j']: y to spectral slots             */
    for (size_t q = 0; q < (size_t)Mi * Mj; ++q) B[q] = B[q] * st->dinv[q];
    h2d_inv(st->Fxe, st->Fxo, mi, st->HPx, B, Mj, X);                   /* X[j'][i] : x back                          */
    h2d_inv(st->Fye, st->Fyo, mj, st->HPy, X, Mi, B);                   /* B[i][j]  = U                               */
    for (int gi = 0; gi < nx; ++gi)
        for (int gj = 0; gj < ny; ++gj) {
            size_t p = (size_t)gi * ny + gj;
            out[p] = (gi == 0 || gj == 0 || gi == nx - 1 || gj == ny - 1) ? st->bc[p] : B[(size_t)(gi - 1) * Mj + (gj - 1)];
        }
}

/* sum of squares, 2-D reduction tree of the spec: per grid row an fma chain over its ny values, then the rows in order */
double orc_sumsq_rows(const double *r, int nx, int ny) {
    double tot = 0.0;
    for (int i = 0; i < nx; ++i) {
        double acc = 0.0;
        for (int j = 0; j < ny; ++j) acc = fma(r[(size_t)i * ny + j], r[(size_t)i * ny + j], acc);
        tot = tot + acc;
    }
    return tot;
}

/* ================================================================================================
 * Problem / solver state (single rank: the reference is bit-identical for every P)
 * ============================================================================================== */
typedef struct {
    int nt, n;              /* time points, DOFs per vector */
    double *t;
    orc_stepper st;
    double *u, *v, *g;      /* slabs [nt][n] ; v,g NULL on level 0 */
    double *u0;             /* vector_t_start of this level */
    uint8_t *is_c;
    int transfer;           /* to next coarser level: 0 copy, 1 heat full-weighting/linear */
} orc_level;

typedef struct {
    int n_levels;
    orc_level L[ORC_MAX_LEVELS];
    double weight_c;
    int cf_iter[ORC_MAX_LEVELS];
    int cycle_type;         /* 0 V, 1 F */
    int nested, t_norm, conv_crit, max_iter, norm_spec;
    double tol;
    double *save_last;      /* conv_crit 1: clone of u[0] */
    double *tmp1, *tmp2, *tmp3;
    int64_t phi_count[ORC_MAX_LEVELS];
    int at_k;               /* > 0: AT-MGRIT (core/at_mgrit.py): truncated local solves of distance k on the coarsest level */
    int no_block_solve;     /* 1: forward_solve never takes the time-parallel form */
    int threads;            /* > 1: the independent loops of the sweeps run on that many OpenMP threads (same arithmetic, same
                               results). Heat1D / Advection1D levels. */
} orc_problem;

orc_problem *orc_problem_create(int n_levels) {
    orc_problem *p = (orc_problem *)calloc(1, sizeof(orc_problem));
    p->n_levels = n_levels; p->weight_c = 1.0; p->nested = 1; p->t_norm = 2; p->max_iter = 100; p->tol = 1e-7;
    p->norm_spec = 1; p->threads = 1;
    for (int l = 0; l < ORC_MAX_LEVELS; ++l) p->cf_iter[l] = 1;
    return p;
}

static void free_stepper(orc_stepper *st) {
    for (int i = 0; i < st->n_csets; ++i) { free(st->csets[i].tab); free(st->csets[i].ch); }
    free(st->csets); free(st->s); free(st->tau); free(st->tau2); free(st->w1); free(st->w2); free(st->frows);
    free(st->bc); free(st->W); free(st->Fxe); free(st->Fxo); free(st->Fye); free(st->Fyo); free(st->lx); free(st->ly); free(st->dinv); free(st->X0); free(st->X1);
}

void orc_problem_destroy(orc_problem *p) {
    if (!p) return;
    for (int l = 0; l < p->n_levels; ++l) {
        orc_level *L = &p->L[l];
        free(L->t); free(L->u); free(L->v); free(L->g); free(L->u0); free(L->is_c);
        free_stepper(&L->st);
    }
    free(p->save_last); free(p->tmp1); free(p->tmp2); free(p->tmp3);
    free(p);
}

static void level_common(orc_problem *p, int lvl, int nt, const double *t, int n, const double *u0) {
    orc_level *L = &p->L[lvl];
    L->nt = nt; L->n = n;
    L->t = (double *)malloc(sizeof(double) * (size_t)nt);
    memcpy(L->t, t, sizeof(double) * (size_t)nt);
    L->u0 = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(L->u0, u0, sizeof(double) * (size_t)n);
    L->st.n = n;
    L->st.w1 = (double *)calloc((size_t)padded(n), sizeof(double));
    L->st.w2 = (double *)calloc((size_t)padded(n), sizeof(double));
}

/* kind-specific level setup. variant: 0 natural, 1 spec. s: [K][n], tau: [K][nt] (may be NULL when K=0) */
void orc_problem_set_level_heat1d(orc_problem *p, int lvl, int nt, const double *t, int n, double fac, int K,
                                  const double *s, const double *tau, const double *u0, int variant) {
    level_common(p, lvl, nt, t, n, u0);
    orc_stepper *st = &p->L[lvl].st;
    st->kind = ORC_HEAT1D; st->variant = variant; st->fac = fac; st->K = K;
    if (K > 0) {
        st->s = (double *)malloc(sizeof(double) * (size_t)K * n);
        st->tau = (double *)malloc(sizeof(double) * (size_t)K * nt);
        memcpy(st->s, s, sizeof(double) * (size_t)K * n);
        memcpy(st->tau, tau, sizeof(double) * (size_t)K * nt);
    }
}

/* general forcing of a heat1d level (after orc_problem_set_level_heat1d with K = 0): rows[i][j] = rhs(x_j, t_i) * (t_i - t_{i-1}) */
void orc_problem_set_forcing_rows(orc_problem *p, int lvl, const double *rows) {
    orc_level *L = &p->L[lvl];
    orc_stepper *st = &L->st;
    free(st->frows);
    st->frows = (double *)malloc(sizeof(double) * (size_t)L->nt * st->n);
    memcpy(st->frows, rows, sizeof(double) * (size_t)L->nt * st->n);
}

/* general forcing of a heat2d level (described with K = 0): rows[i][a][b] = rhs(x_a, y_b, t_i) on the mi x mj interior, NOT yet
 * multiplied by a step size -- the theta-scheme weighs the two ends of a step itself (heat_2d.py:289-320, 346-356) */
void orc_problem_set_forcing_rows_2d(orc_problem *p, int lvl, const double *rows) {
    orc_level *L = &p->L[lvl];
    orc_stepper *st = &L->st;
    size_t per = (size_t)st->mi * st->mj;
    free(st->frows);
    st->frows = (double *)malloc(sizeof(double) * (size_t)L->nt * per);
    memcpy(st->frows, rows, sizeof(double) * (size_t)L->nt * per);
}

/* two-point heat stepper: n values per time point of the pair (state = 2n), order = 1 (BDF1) or 2 (BDF2);
 * tau: [K][nt] tau_k(t_i), tau2: [K][nt] tau_k(t_i + dtau); u0: [2n] */
void orc_problem_set_level_heat1d_2pts(orc_problem *p, int lvl, int nt, const double *t, int n, double fac, double dtau,
                                       int order, int K, const double *s, const double *tau, const double *tau2,
                                       const double *u0, int variant) {
    level_common(p, lvl, nt, t, 2 * n, u0);
    orc_stepper *st = &p->L[lvl].st;
    st->kind = ORC_HEAT1D_2PTS; st->variant = variant; st->fac = fac; st->K = K; st->n = n; st->dtau = dtau; st->method = order;
    if (K > 0) {
        st->s = (double *)malloc(sizeof(double) * (size_t)K * n);
        st->tau = (double *)malloc(sizeof(double) * (size_t)K * nt);
        st->tau2 = (double *)malloc(sizeof(double) * (size_t)K * nt);
        memcpy(st->s, s, sizeof(double) * (size_t)K * n);
        memcpy(st->tau, tau, sizeof(double) * (size_t)K * nt);
        memcpy(st->tau2, tau2, sizeof(double) * (size_t)K * nt);
    }
}

void orc_problem_set_level_advection1d(orc_problem *p, int lvl, int nt, const double *t, int n, double fac,
                                       const double *u0, int variant) {
    level_common(p, lvl, nt, t, n, u0);
    orc_stepper *st = &p->L[lvl].st;
    st->kind = ORC_ADVECTION1D; st->variant = variant; st->fac = fac;
}

void orc_problem_set_level_dahlquist(orc_problem *p, int lvl, int nt, const double *t, double lambda, int method,
                                     double u0) {
    level_common(p, lvl, nt, t, 1, &u0);
    orc_stepper *st = &p->L[lvl].st;
    st->kind = ORC_DAHLQUIST; st->fac = lambda; st->method = method;
}

/* bc: nx*ny boundary values (zero inside); S: [K][mi][mj] forcing space factors on the interior; tau: [K][nt] */
void orc_problem_set_level_heat2d(orc_problem *p, int lvl, int nt, const double *t, int nx, int ny, double fx, double fy,
                                  double theta, const double *bc, int K, const double *S, const double *tau,
                                  const double *u0) {
    level_common(p, lvl, nt, t, nx * ny, u0);
    orc_stepper *st = &p->L[lvl].st;
    st->kind = ORC_HEAT2D; st->variant = 1; st->K = K;
    st->nx = nx; st->ny = ny; st->mi = nx - 2; st->mj = ny - 2;
    st->HPx = (((st->mi + 1) / 2 + 63) / 64) * 64; st->HPy = (((st->mj + 1) / 2 + 63) / 64) * 64;
    st->Mi = 2 * st->HPx; st->Mj = 2 * st->HPy;
    st->fx = fx; st->fy = fy; st->theta = theta;
    st->bc = (double *)malloc(sizeof(double) * (size_t)nx * ny);
    memcpy(st->bc, bc, sizeof(double) * (size_t)nx * ny);
    st->W = (double *)calloc((size_t)st->mi * st->mj, sizeof(double));
    st->has_w = 0;
    for (int a = 0; a < st->mi; ++a)
        for (int b = 0; b < st->mj; ++b) {
            double w = 0.0;
            if (a == 0) w += fx * bc[(size_t)0 * ny + b + 1];
            if (a == st->mi - 1) w += fx * bc[(size_t)(nx - 1) * ny + b + 1];
            if (b == 0) w += fy * bc[(size_t)(a + 1) * ny + 0];
            if (b == st->mj - 1) w += fy * bc[(size_t)(a + 1) * ny + ny - 1];
            st->W[(size_t)a * st->mj + b] = w;
            if (w != 0.0) st->has_w = 1;
        }
    if (K > 0) {
        size_t ns = (size_t)K * st->mi * st->mj;
        st->s = (double *)malloc(sizeof(double) * ns);
        st->tau = (double *)malloc(sizeof(double) * (size_t)K * nt);
        memcpy(st->s, S, sizeof(double) * ns);
        memcpy(st->tau, tau, sizeof(double) * (size_t)K * nt);
    }
    h2d_tables(st);
}

void orc_problem_set_transfer(orc_problem *p, int lvl, int kind) { p->L[lvl].transfer = kind; }

/* Threaded sweeps for the CPU baseline of bench.py: the F-intervals / C-points of a sweep are independent -- the same
 * independence the reference's mpi4py path exploits across ranks (mgrit.py:313-331) -- so they are split over OpenMP
 * threads. Returns the thread count in effect (1 when a level is neither Heat1D nor Advection1D, or the library was built
 * without OpenMP). Both arithmetic variants (round 5: the full-size GPU parity tests run the spec
 * variant at config 3's size, 72 s single-threaded). */
int orc_problem_set_threads(orc_problem *p, int threads) {
    p->threads = 1;
#ifdef _OPENMP
    int ok = threads > 1;
    for (int l = 0; l < p->n_levels; ++l) {
        orc_level *L = &p->L[l];
        if (L->st.kind != ORC_HEAT1D && L->st.kind != ORC_ADVECTION1D) ok = 0;
    }
    if (ok) {
        /* the spec steps look their coefficient set up by step size and build it on first use: every set exists before the
           first parallel loop, so the threads only ever read the table */
        for (int l = 0; l < p->n_levels; ++l) {
            orc_level *L = &p->L[l];
            if (L->st.variant) for (int i = 1; i < L->nt; ++i) (void)get_cset(&L->st, L->t[i] - L->t[i - 1]);
        }
        p->threads = threads;
    }
#else
    (void)threads;
#endif
    return p->threads;
}

void orc_problem_set_options(orc_problem *p, double weight_c, const int32_t *cf_iter, int cycle_type, int nested,
                             int t_norm, int conv_crit, int max_iter, double tol, int norm_spec) {
    p->weight_c = weight_c; p->cycle_type = cycle_type; p->nested = nested; p->t_norm = t_norm;
    p->conv_crit = conv_crit; p->max_iter = max_iter; p->tol = tol; p->norm_spec = norm_spec;
    for (int l = 0; l < p->n_levels; ++l) p->cf_iter[l] = cf_iter[l];
}

/* Phi on level lvl from point i-1 to point i (Application.step, application.py:98-107) */
static void phi(orc_problem *p, int lvl, int i, const double *u_in, double *out) {
    orc_level *L = &p->L[lvl];
    double t_start = L->t[i - 1], t_stop = L->t[i], dt = t_stop - t_start;
    p->phi_count[lvl]++;
    switch (L->st.kind) {
    case ORC_HEAT1D:
        if (L->st.variant) heat1d_step_spec(&L->st, L->nt, i, dt, u_in, out);
        else heat1d_step_natural(&L->st, L->nt, i, dt, u_in, out);
        break;
    case ORC_ADVECTION1D:
        if (L->st.variant) advection1d_step_spec(&L->st, dt, u_in, out);
        else advection1d_step_natural(&L->st, dt, u_in, out);
        break;
    case ORC_HEAT2D: heat2d_step(&L->st, L->nt, i, t_start, t_stop, u_in, out); break;
    case ORC_HEAT1D_2PTS:
        if (L->st.variant) twopts_step_spec(&L->st, L->nt, i, t_start, t_stop, u_in, out);
        else twopts_step_natural(&L->st, L->nt, i, t_start, t_stop, u_in, out);
        break;
    default: dahlquist_step(&L->st, t_start, t_stop, u_in, out);
    }
}

/* single Phi application for tests: out = step(u, t[i-1] -> t[i]) on level lvl */
void orc_phi(orc_problem *p, int lvl, int i, const double *u_in, double *out) { phi(p, lvl, i, u_in, out); }

/* examples/example_spatial_coarsening.py:33-55 / :58-82 ; kind 0: GridTransferCopy (grid_transfer_copy.py:23-47) */
static void restrict_vec(int kind, const double *f, int nf, double *c, int nc) {
    if (kind == 0) { memcpy(c, f, sizeof(double) * (size_t)nc); return; }
    if (kind == 2) { /* periodic full weighting, nf = 2*nc (pymgrit_amd/advection/grid_transfer_advection.py) */
        for (int i = 0; i < nc; ++i)
            c[i] = f[(2 * i - 1 + nf) % nf] * 1.0 / 4.0 + f[2 * i] * 1.0 / 2.0 + f[(2 * i + 1) % nf] * 1.0 / 4.0;
        return;
    }
    for (int i = 0; i < nc; ++i) c[i] = f[2 * i] * 1.0 / 4.0 + f[2 * i + 1] * 1.0 / 2.0 + f[2 * i + 2] * 1.0 / 4.0;
}

static void interp_vec(int kind, const double *c, int nc, double *f, int nf) {
    if (kind == 0) { memcpy(f, c, sizeof(double) * (size_t)nf); return; }
    if (kind == 2) { /* periodic linear interpolation */
        for (int i = 0; i < nc; ++i) {
            f[2 * i] = 0.0 + c[i];
            f[2 * i + 1] = (0.0 + 1.0 / 2.0 * c[i]) + 1.0 / 2.0 * c[(i + 1) % nc];
        }
        return;
    }
    for (int j = 0; j < nf; ++j) f[j] = 0.0;
    for (int i = 0; i < nc; ++i) {
        f[2 * i] += 1.0 / 2.0 * c[i];
        f[2 * i + 1] += c[i];
        f[2 * i + 2] += 1.0 / 2.0 * c[i];
    }
}

#define ROW(a, L, i) ((a) + (size_t)(i) * (size_t)(L)->n)

/* mgrit.py:840-858 (+ level C/F sets, mgrit.py:767-770) */
void orc_problem_init_state(orc_problem *p) {
    int32_t nt[ORC_MAX_LEVELS];
    int64_t off[ORC_MAX_LEVELS + 1];
    int maxn = 1;
    off[0] = 0;
    for (int l = 0; l < p->n_levels; ++l) { nt[l] = p->L[l].nt; off[l + 1] = off[l] + nt[l]; if (p->L[l].n > maxn) maxn = p->L[l].n; }
    double *t_all = (double *)malloc(sizeof(double) * (size_t)off[p->n_levels]);
    for (int l = 0; l < p->n_levels; ++l) memcpy(t_all + off[l], p->L[l].t, sizeof(double) * (size_t)nt[l]);
    for (int l = 0; l < p->n_levels; ++l) {
        orc_level *L = &p->L[l];
        size_t sz = (size_t)L->nt * (size_t)L->n;
        free(L->u); free(L->v); free(L->g); free(L->is_c);
        L->u = (double *)calloc(sz, sizeof(double));
        L->v = l ? (double *)calloc(sz, sizeof(double)) : NULL;
        L->g = l ? (double *)calloc(sz, sizeof(double)) : NULL;
        memcpy(L->u, L->u0, sizeof(double) * (size_t)L->n);      /* u[lvl][0] = vector_t_start.clone() */
        L->is_c = (uint8_t *)malloc((size_t)L->nt);
        global_is_c(p->n_levels, nt, t_all, off, l, L->is_c);
    }
    free(t_all);
    free(p->tmp1); free(p->tmp2); free(p->tmp3);
    p->tmp1 = (double *)calloc((size_t)maxn, sizeof(double));
    p->tmp2 = (double *)calloc((size_t)maxn, sizeof(double));
    p->tmp3 = (double *)calloc((size_t)maxn, sizeof(double));
    memset(p->phi_count, 0, sizeof(p->phi_count));
}

double *orc_state_ptr(orc_problem *p, int which, int lvl) {
    orc_level *L = &p->L[lvl];
    return which == 0 ? L->u : which == 1 ? L->v : L->g;
}

int64_t orc_phi_count(orc_problem *p, int lvl) { return p->phi_count[lvl]; }

#ifdef _OPENMP
#include <omp.h>
/* ---- threaded variants (timing path): identical arithmetic, one interval / C-point per loop trip ---- */
static void phi_ws(orc_problem *p, int lvl, int i, const double *u_in, double *out, double *w1, double *w2) {
    orc_level *L = &p->L[lvl];
    double dt = L->t[i] - L->t[i - 1];
    if (L->st.kind == ORC_ADVECTION1D) {
        if (L->st.variant) advection1d_step_spec_ws(&L->st, dt, u_in, out, w1);
        else advection1d_step_natural_ws(&L->st, dt, u_in, out, w1, w2);
    } else if (L->st.variant) heat1d_step_spec_ws(&L->st, L->nt, i, dt, u_in, out, w1);
    else heat1d_step_natural_ws(&L->st, L->nt, i, dt, u_in, out, w1, w2);
}

static int has_adjacent_c(const orc_level *L) {
    for (int i = 1; i < L->nt; ++i) if (L->is_c[i] && L->is_c[i - 1] && i - 1 > 0) return 1;
    return 0;
}

static void f_relax_mt(orc_problem *p, int lvl) {
    orc_level *L = &p->L[lvl];
    int n = L->n, np = padded(n);
#pragma omp parallel num_threads(p->threads)
    {
        double *w = (double *)malloc(sizeof(double) * 3 * (size_t)np), *w1 = w, *w2 = w + np, *tmp = w + 2 * np;
#pragma omp for schedule(static)
        for (int i = 1; i < L->nt; ++i) {
            if (L->is_c[i] || !L->is_c[i - 1]) continue;       /* i = first F-point of an interval */
            for (int k = i; k < L->nt && !L->is_c[k]; ++k) {
                if (lvl == 0) phi_ws(p, lvl, k, ROW(L->u, L, k - 1), ROW(L->u, L, k), w1, w2);
                else {
                    phi_ws(p, lvl, k, ROW(L->u, L, k - 1), tmp, w1, w2);
                    double *uk = ROW(L->u, L, k); const double *gk = ROW(L->g, L, k);
                    for (int j = 0; j < n; ++j) uk[j] = gk[j] + tmp[j];
                }
            }
        }
        free(w);
    }
}

static void c_relax_mt(orc_problem *p, int lvl) {
    orc_level *L = &p->L[lvl];
    int n = L->n, np = padded(n);
    double wt = p->weight_c, w1c = 1.0 - p->weight_c;
#pragma omp parallel num_threads(p->threads)
    {
        double *w = (double *)malloc(sizeof(double) * 3 * (size_t)np), *w1 = w, *w2 = w + np, *tmp = w + 2 * np;
#pragma omp for schedule(static)
        for (int i = 1; i < L->nt; ++i) {
            if (!L->is_c[i]) continue;
            phi_ws(p, lvl, i, ROW(L->u, L, i - 1), tmp, w1, w2);
            double *ui = ROW(L->u, L, i);
            if (lvl == 0) for (int j = 0; j < n; ++j) ui[j] = tmp[j] * wt + ui[j] * w1c;
            else { const double *gi = ROW(L->g, L, i); for (int j = 0; j < n; ++j) ui[j] = (gi[j] + tmp[j]) * wt + ui[j] * w1c; }
        }
        free(w);
    }
}

static void fas_residual_mt(orc_problem *p, int lvl) {
    orc_level *L = &p->L[lvl], *C = &p->L[lvl + 1];
    int n = L->n, nc_ = C->n, np = padded(n);
    int *cidx = (int *)malloc(sizeof(int) * (size_t)C->nt), nc = 0;
    for (int i = 0; i < L->nt; ++i) if (L->is_c[i]) cidx[nc++] = i;
#pragma omp parallel for schedule(static) num_threads(p->threads)
    for (int j = 0; j < nc; ++j) {
        restrict_vec(L->transfer, ROW(L->u, L, cidx[j]), n, ROW(C->u, C, j), nc_);
        memcpy(ROW(C->v, C, j), ROW(C->u, C, j), sizeof(double) * (size_t)nc_);
    }
#pragma omp parallel num_threads(p->threads)
    {
        double *w = (double *)malloc(sizeof(double) * 5 * (size_t)np), *w1 = w, *w2 = w + np, *a = w + 2 * np, *b = w + 3 * np, *r = w + 4 * np;
#pragma omp for schedule(static)
        for (int j = 1; j < nc; ++j) {
            int i = cidx[j];
            phi_ws(p, lvl, i, ROW(L->u, L, i - 1), a, w1, w2);
            const double *ui = ROW(L->u, L, i);
            if (lvl == 0) for (int k = 0; k < n; ++k) a[k] = a[k] - ui[k];
            else { const double *gi = ROW(L->g, L, i); for (int k = 0; k < n; ++k) a[k] = gi[k] - ui[k] + a[k]; }
            restrict_vec(L->transfer, a, n, r, nc_);
            phi_ws(p, lvl + 1, j, ROW(C->v, C, j - 1), b, w1, w2);
            double *gj = ROW(C->g, C, j); const double *vj = ROW(C->v, C, j);
            for (int k = 0; k < nc_; ++k) gj[k] = r[k] + vj[k] - b[k];
        }
        free(w);
    }
    free(cidx);
}

static void error_correction_mt(orc_problem *p, int lvl) {
    orc_level *L = &p->L[lvl], *C = &p->L[lvl + 1];
    int n = L->n, nc_ = C->n, np = padded(n);
    int *cidx = (int *)malloc(sizeof(int) * (size_t)C->nt), nc = 0;
    for (int i = 0; i < L->nt; ++i) if (L->is_c[i]) cidx[nc++] = i;
#pragma omp parallel num_threads(p->threads)
    {
        double *w = (double *)malloc(sizeof(double) * 2 * (size_t)np), *e = w, *f = w + np;
#pragma omp for schedule(static)
        for (int j = 1; j < nc; ++j) {
            const double *uj = ROW(C->u, C, j), *vj = ROW(C->v, C, j);
            double *ui = ROW(L->u, L, cidx[j]);
            for (int k = 0; k < nc_; ++k) e[k] = uj[k] - vj[k];
            interp_vec(L->transfer, e, nc_, f, n);
            for (int k = 0; k < n; ++k) ui[k] = ui[k] + f[k];
        }
        free(w);
    }
    free(cidx);
}

static int compute_residual_mt(orc_problem *p, double *r_norm) {
    orc_level *L = &p->L[0];
    int n = L->n, np = padded(n);
    int *cidx = (int *)malloc(sizeof(int) * (size_t)L->nt), nc = 0;
    for (int i = 1; i < L->nt; ++i) if (L->is_c[i]) cidx[nc++] = i;
#pragma omp parallel num_threads(p->threads)
    {
        double *w = (double *)malloc(sizeof(double) * 3 * (size_t)np), *w1 = w, *w2 = w + np, *tmp = w + 2 * np;
#pragma omp for schedule(static)
        for (int c = 0; c < nc; ++c) {
            int i = cidx[c];
            phi_ws(p, 0, i, ROW(L->u, L, i - 1), tmp, w1, w2);
            const double *ui = ROW(L->u, L, i);
            double ss = 0.0;
            if (p->norm_spec) {
                for (int j = 0; j < n; ++j) tmp[j] = tmp[j] - ui[j];
                ss = orc_sumsq_spec(tmp, n);
            } else
                for (int j = 0; j < n; ++j) { double r = tmp[j] - ui[j]; ss += r * r; }
            r_norm[c] = sqrt(ss);
        }
        free(w);
    }
    free(cidx);
    return nc;
}
#endif

/* mgrit.py:292-333 (single rank: intervals are independent, ascending inside an interval) */
void orc_f_relax(orc_problem *p, int lvl) {
    orc_level *L = &p->L[lvl];
#ifdef _OPENMP
    if (p->threads > 1) { f_relax_mt(p, lvl); return; }
#endif
    for (int i = 1; i < L->nt; ++i) {
        if (L->is_c[i]) continue;
        if (lvl == 0) phi(p, lvl, i, ROW(L->u, L, i - 1), ROW(L->u, L, i));
        else {
            phi(p, lvl, i, ROW(L->u, L, i - 1), p->tmp1);
            double *ui = ROW(L->u, L, i); const double *gi = ROW(L->g, L, i);
            for (int j = 0; j < L->n; ++j) ui[j] = gi[j] + p->tmp1[j];
        }
    }
}

/* mgrit.py:335-370 */
void orc_c_relax(orc_problem *p, int lvl) {
    orc_level *L = &p->L[lvl];
    double w = p->weight_c, w1 = 1.0 - p->weight_c;
#ifdef _OPENMP
    if (p->threads > 1 && !has_adjacent_c(L)) { c_relax_mt(p, lvl); return; }
#endif
    for (int i = 1; i < L->nt; ++i) {
        if (!L->is_c[i]) continue;
        phi(p, lvl, i, ROW(L->u, L, i - 1), p->tmp1);
        double *ui = ROW(L->u, L, i);
        if (lvl == 0) for (int j = 0; j < L->n; ++j) ui[j] = p->tmp1[j] * w + ui[j] * w1;
        else { const double *gi = ROW(L->g, L, i); for (int j = 0; j < L->n; ++j) ui[j] = (gi[j] + p->tmp1[j]) * w + ui[j] * w1; }
    }
}

/* mgrit.py:459-486 */

/* ================================================================================================
 * Overlapped chain (spec): forward_solve on a Heat1D level whose steps all share one dt and whose vectors
 * span several groups. The step is the same solve as 3.3, rearranged so that the group-local scans of step i+1 do not wait
 * for the carries of step i: with the solve result of step i written as  w_i + cm*Q1 + cb*pw - z0*wg  (w_i = zh_i*ik + g_i,
 * the part that needs no carries), the scans of step i+1 run on  fma(s, tau, w_i)  alone and the images of the three carry
 * terms under the (linear) group-local scans are added afterwards from tables V1, V2, V3 and their totals.
 * ============================================================================================== */
/* group-local image of src (len valid entries, zero beyond): serial forward recurrence, zero padding, serial backward
 * recurrence. a = forward total (last element), b = backward total (first element). */
static void chain_local(const orc_cset *c, const double *src, int len, double *V, double *a, double *b) {
    double y[ORC_GROUP];
    y[0] = src[0];
    for (int j = 1; j < ORC_GROUP; ++j) y[j] = fma(c->rho, y[j - 1], src[j]);
    *a = y[ORC_GROUP - 1];
    for (int j = len; j < ORC_GROUP; ++j) y[j] = 0.0;
    double z = y[ORC_GROUP - 1];
    V[ORC_GROUP - 1] = z;
    for (int j = ORC_GROUP - 2; j >= 0; --j) { z = fma(c->rho, z, y[j]); V[j] = z; }
    *b = V[0];
}

static void cset_chain_tables(orc_cset *c, int n) {
    if (c->ch) return;
    int NP = padded(n), G = NP / ORC_GROUP, last_len = n - (G - 1) * ORC_GROUP;
    c->ch = (double *)calloc((size_t)6 * ORC_GROUP + (size_t)NP, sizeof(double));
    double src[ORC_GROUP];
    for (int var = 0; var < 2; ++var) {
        int len = var ? last_len : ORC_GROUP;
        const double *pt = var ? c->pt_last : c->pt_full;
        double *q1 = c->ch + (size_t)var * ORC_GROUP, *v1 = c->ch + (size_t)(2 + var) * ORC_GROUP, *v2 = c->ch + (size_t)(4 + var) * ORC_GROUP;
        for (int j = 0; j < ORC_GROUP; ++j) q1[j] = (j < len) ? c->ik * pt[j] : 0.0;
        chain_local(c, q1, len, v1, &c->ca1[var], &c->cb1[var]);
        for (int j = 0; j < ORC_GROUP; ++j) {
            int l = j / ORC_E, k = j % ORC_E;
            src[j] = (j < len) ? (c->lp[ORC_LANES - 1 - l] * c->ik) * c->pw[ORC_E - k] : 0.0;
        }
        chain_local(c, src, len, v2, &c->ca2[var], &c->cb2[var]);
    }
    double *v3 = c->ch + (size_t)6 * ORC_GROUP;
    for (int g = 0; g < G; ++g) {
        int len = (g == G - 1) ? last_len : ORC_GROUP;
        for (int j = 0; j < ORC_GROUP; ++j) src[j] = (j < len) ? c->tab[(size_t)g * ORC_GROUP + j] : 0.0;
        chain_local(c, src, len, v3 + (size_t)g * ORC_GROUP, &c->ca3[g], &c->cb3[g]);
    }
}

/* does forward_solve on this level use the overlapped chain? (the HIP engine applies the same rule to its local points) */
static int chain_overlapped(const orc_level *L, int lvl) {
    if (lvl == 0) return 0;   /* a one-level hierarchy is plain time stepping: its residual must vanish exactly */
    const orc_stepper *st = &L->st;
    if (st->kind != ORC_HEAT1D || !st->variant || st->n <= ORC_GROUP || st->n > 16 * ORC_GROUP || st->K > 1 || st->frows || L->nt < 2) return 0;
    double dt0 = L->t[1] - L->t[0];
    for (int i = 2; i < L->nt; ++i) {
        double dt = L->t[i] - L->t[i - 1];
        if (memcmp(&dt, &dt0, sizeof(double)) != 0) return 0;
    }
    return 1;
}

static void heat1d_chain_spec(orc_problem *p, int lvl) {
    orc_level *L = &p->L[lvl];
    orc_stepper *st = &L->st;
    int n = st->n, NP = padded(n), G = NP / ORC_GROUP, nt = L->nt, use_g = lvl > 0;
    double dt = L->t[1] - L->t[0];
    orc_cset *c = get_cset(st, dt);
    cset_chain_tables(c, n);
    const double *v3 = c->ch + (size_t)6 * ORC_GROUP;
    double *w = (double *)calloc((size_t)NP, sizeof(double)), *zh = (double *)calloc((size_t)NP, sizeof(double));
    double A[16] = {0}, B[16] = {0}, cm[16] = {0}, zin[17] = {0}, z0 = 0.0;
    memcpy(w, ROW(L->u, L, 0), sizeof(double) * (size_t)n);
    for (int i = 1; i < nt; ++i) {
        p->phi_count[lvl]++;
        /* group-local scans of step i on the carry-free part of the previous result */
        for (int j = 0; j < n; ++j) zh[j] = st->K ? fma(st->s[j], st->tau[i] * dt, w[j]) : w[j];
        for (int j = n; j < NP; ++j) zh[j] = 0.0;
        for (int g = 0; g < G; ++g) {
            double *dg = zh + (size_t)g * ORC_GROUP;
            int var = (g == G - 1);
            double a = group_scan_fwd(c, dg);
            for (int j = 0; j < ORC_GROUP; ++j) if (g * ORC_GROUP + j >= n) dg[j] = 0.0;
            double b = group_scan_bwd(c, dg);
            /* images of the carry terms of step i-1 (all zero in front of the first step) */
            const double *v1 = c->ch + (size_t)(2 + var) * ORC_GROUP, *v2 = c->ch + (size_t)(4 + var) * ORC_GROUP;
            A[g] = fma(-z0, c->ca3[g], fma(zin[g + 1], c->ca2[var], fma(cm[g], c->ca1[var], a)));
            B[g] = fma(-z0, c->cb3[g], fma(zin[g + 1], c->cb2[var], fma(cm[g], c->cb1[var], b)));
            for (int j = 0; j < ORC_GROUP; ++j)
                dg[j] = fma(-z0, v3[(size_t)g * ORC_GROUP + j], fma(zin[g + 1], v2[j], fma(cm[g], v1[j], dg[j])));
        }
        /* the one exchange of the step: carries from the totals of all groups (3.3 step 3) */
        double I[ORC_MAX_G] = {0}, Zf[ORC_MAX_G + 1] = {0};   /* (the overlapped chain itself covers <= 16 groups) */
        for (int g = 0; g < 16; ++g) I[g] = (g < G) ? A[g] : 0.0;
        cross_scan(c, I, 0);
        cm[0] = 0.0;
        for (int g = 1; g < 16; ++g) cm[g] = I[g - 1];
        for (int g = 0; g < 16; ++g)
            Zf[g] = (g < G) ? fma(cm[g], (g == G - 1) ? c->pt_last[0] : c->pt_full[0], B[g]) : 0.0;
        cross_scan(c, Zf, 1);
        Zf[16] = 0.0;
        z0 = Zf[0] * c->ik;
        for (int g = 0; g < 16; ++g) zin[g + 1] = Zf[g + 1];
        /* result of step i: carry-free part w, then the three carry terms (3.3 step 4 with ik folded into the tables) */
        double *ui = ROW(L->u, L, i);
        const double *gi = use_g ? ROW(L->g, L, i) : NULL;
        for (int g = 0; g < G; ++g) {
            const double *q1 = c->ch + (size_t)((g == G - 1) ? 1 : 0) * ORC_GROUP;
            for (int l = 0; l < ORC_LANES; ++l) {
                double cb = (c->lp[ORC_LANES - 1 - l] * c->ik) * zin[g + 1];
                for (int k = 0; k < ORC_E; ++k) {
                    int jj = l * ORC_E + k, j = g * ORC_GROUP + jj;
                    if (j >= n) continue;
                    double base = zh[j] * c->ik;
                    w[j] = use_g ? gi[j] + base : base;
                    ui[j] = fma(-z0, c->tab[j], fma(c->pw[ORC_E - k], cb, fma(cm[g], q1[jj], w[j])));
                }
            }
        }
    }
    free(w); free(zh);
}

/* ================================================================================================
 * Time-parallel forward solve of a Heat1D level (spec): the steps are cut into blocks of ORC_BLK_K; every block
 * is stepped from a ZERO state (independent of all other blocks), the states at the block ends are then put right by a
 * recurrence over the blocks in the r lowest sine modes -- all the steps of a block share the eigenvectors
 * q_k(j) = sqrt(2/(n+1)) sin(pi (k+1)(j+1)/(n+1)), and the block's propagator prod_i (I + dt_i L)^{-1} has the eigenvalues
 * D_b(k) = prod_i 1/(1 + dt_i fac 4 sin^2(theta_k/2)), which fall below 2^-60 from mode r on (the rule that makes a level
 * eligible: r <= ORC_BLK_RMAX) --, and the interiors of the blocks are stepped again from the corrected block starts.
 * The scheme works on the DEFECT of the level's current values u (the injected fine values of the FAS cycle): with the error e of u,
 * e_i = r_i + Phi_lin(e_{i-1}), r_i = g_i + Phi(u_{i-1}) - u_i, every block propagates its own defects from a zero error,
 *   phase 1   every block: x = 0; over the block's steps x = (g_i + Phi(u_{i-1} + x)) - u_i  (= r_i + Phi_lin(x) up to rounding; the
 *             first step takes u_{i-1} itself), W_b = x kept aside. Values u that already satisfy u_i = g_i + Phi(u_{i-1}) bit for
 *             bit give x = 0 throughout: the solve then changes nothing -- the property that lets the MGRIT iteration converge to
 *             rounding level (a solve that re-derived u from scratch would differ from it by eps cond(Phi) in every cycle)
 *   phase 2a  what_b(k) = <q_k, W_b>, b = 0 .. B-2 (blk_dot: one fma chain per chunk of 1024 values in row storage order, chunks in order)
 *   phase 2b  Uh = what_0; for b = 1 .. B-1: c_b = D_b * Uh (product), Uh = what_b + c_b
 *   phase 2c  u[e_b] = u[e_b] + E_b, E_0 = W_0, E_b = W_b + sum_k q_k c_b(k): x = fma(q_k(j), c_b(k), x) for k ascending from x = W_b
 *   phase 3   every block: x = u[e_{b-1}] (u_0 for the first); u_i = g_i + Phi(x) for the points strictly inside the block
 * The same solve as the sequential one up to rounding and the truncation (2^-60 relative per block); every Phi is the step
 * of 3.3. The last block takes the remainder (K .. 2K-1 steps). A level with fewer than 4 K steps, a level 0 (a one-level
 * hierarchy is plain time stepping) and a level whose decay is too slow for ORC_BLK_RMAX modes are solved step by step.
 * ============================================================================================== */
#define ORC_BLK_K 16
#define ORC_BLK_RMAX 256
#define ORC_BLK_THR 8.673617379884035e-19   /* 2^-60 */

/* 4 sin^2(theta_k / 2), theta_k = pi (k+1)/(n+1) */
static double blk_lam4(int n, int k) {
    double h = sin(M_PI * (double)(k + 1) / (2.0 * (double)(n + 1)));
    return 4.0 * h * h;
}

/* blocks of the level's steps 1 .. nt-1: block b = steps K b + 1 .. K (b+1), the last one up to nt-1. Returns B (0: too short) */
static int blk_count(int nt) { int N = nt - 1; return N >= 4 * ORC_BLK_K ? N / ORC_BLK_K : 0; }
static int blk_end(int nt, int B, int b) { return b == B - 1 ? nt - 1 : ORC_BLK_K * (b + 1); }

/* D[b][k] for k < ORC_BLK_RMAX + 1 and the rank r (0: the level is not eligible); fac = a/dx^2, t = the level's time grid */
int orc_block_solve_rank(int n, double fac, int nt, const double *t, double *D /* [B][ORC_BLK_RMAX + 1] or NULL */) {
    int B = blk_count(nt), r = 0;
    if (B == 0 || n < 2) return 0;
    int KM = ORC_BLK_RMAX + 1 < n ? ORC_BLK_RMAX + 1 : n;
    for (int b = 0; b < B; ++b) {
        int first = ORC_BLK_K * b + 1, last = blk_end(nt, B, b), rb = -1;
        for (int k = 0; k < KM; ++k) {
            double lam = blk_lam4(n, k), d = 1.0;
            for (int i = first; i <= last; ++i) d = d * (1.0 / (1.0 + ((t[i] - t[i - 1]) * fac) * lam));
            if (D) D[(size_t)b * (ORC_BLK_RMAX + 1) + k] = d;
            if (rb < 0 && d < ORC_BLK_THR) rb = k;
        }
        if (b >= 1) {   /* block 0 propagates nothing (its start value is given) */
            if (rb < 0 || rb > ORC_BLK_RMAX) return 0;
            if (rb > r) r = rb;
        }
    }
    return r < 1 ? 1 : r;
}

static int block_solve_on(const orc_problem *p, const orc_level *L, int lvl) {
    const orc_stepper *st = &L->st;
    if (p->no_block_solve || lvl == 0 || st->kind != ORC_HEAT1D || !st->variant || st->n > 16 * ORC_GROUP) return 0;
    return orc_block_solve_rank(st->n, st->fac, L->nt, L->t, NULL);
}

/* <q, x> as the device's matrix cores accumulate it: the vector in the engine's row storage order -- inside a
 * chunk of 1024 values: for q = 0..7, for lane = 0..63, for r = 0..1 the value 16 lane + 2 q + r --, one fma chain per chunk in that
 * order (a K loop of v_mfma_f64_16x16x4 accumulates its products sequentially), the chunks' sums added in chunk order */
static double blk_dot(const double *x, const double *q, int n) {
    int G = (n + ORC_GROUP - 1) / ORC_GROUP;
    double tot = 0.0;
    for (int c = 0; c < G; ++c) {
        double acc = 0.0;
        for (int qq = 0; qq < 8; ++qq)
            for (int l = 0; l < ORC_LANES; ++l)
                for (int r = 0; r < 2; ++r) {
                    int j = c * ORC_GROUP + 16 * l + 2 * qq + r;
                    if (j < n) acc = fma(x[j], q[j], acc);
                }
        tot = tot + acc;
    }
    return tot;
}

static void heat1d_block_solve_spec(orc_problem *p, int lvl, int r) {
    orc_level *L = &p->L[lvl];
    int n = L->n, nt = L->nt, B = blk_count(nt), RM = ORC_BLK_RMAX + 1;
    double *D = (double *)malloc(sizeof(double) * (size_t)B * RM);
    orc_block_solve_rank(n, L->st.fac, nt, L->t, D);
    double *Q = (double *)malloc(sizeof(double) * (size_t)r * n);
    double sc = sqrt(2.0 / (double)(n + 1));
    for (int k = 0; k < r; ++k)
        for (int j = 0; j < n; ++j) {
            long m = ((long)(k + 1) * (long)(j + 1)) % (2L * (n + 1));
            Q[(size_t)k * n + j] = sc * sin(M_PI * (double)m / (double)(n + 1));
        }
    double *x = (double *)calloc((size_t)n, sizeof(double)), *y = (double *)calloc((size_t)n, sizeof(double));
    double *what = (double *)calloc((size_t)B * r, sizeof(double)), *Ws = (double *)calloc((size_t)B * n, sizeof(double));
    double *Uh = (double *)calloc((size_t)r, sizeof(double)), *c = (double *)calloc((size_t)r, sizeof(double));
    /* phase 1: the defect of the level's CURRENT values, propagated through every block from zero */
    for (int b = 0; b < B; ++b) {
        int first = ORC_BLK_K * b + 1, last = blk_end(nt, B, b);
        for (int i = first; i <= last; ++i) {
            const double *up = ROW(L->u, L, i - 1), *ui = ROW(L->u, L, i), *gi = ROW(L->g, L, i);
            if (i == first) memcpy(y, up, sizeof(double) * (size_t)n);
            else for (int j = 0; j < n; ++j) y[j] = up[j] + x[j];
            phi(p, lvl, i, y, p->tmp1);
            for (int j = 0; j < n; ++j) x[j] = (gi[j] + p->tmp1[j]) - ui[j];
        }
        memcpy(Ws + (size_t)b * n, x, sizeof(double) * (size_t)n);
        if (b < B - 1)
            for (int k = 0; k < r; ++k) what[(size_t)b * r + k] = blk_dot(x, Q + (size_t)k * n, n);
    }
    /* phases 2b, 2c */
    memcpy(Uh, what, sizeof(double) * (size_t)r);
    for (int b = 0; b < B; ++b) {
        double *ue = ROW(L->u, L, blk_end(nt, B, b));
        const double *wb = Ws + (size_t)b * n;
        if (b == 0) { for (int j = 0; j < n; ++j) ue[j] = ue[j] + wb[j]; continue; }
        for (int k = 0; k < r; ++k) c[k] = D[(size_t)b * RM + k] * Uh[k];
        if (b < B - 1) for (int k = 0; k < r; ++k) Uh[k] = what[(size_t)b * r + k] + c[k];
        for (int j = 0; j < n; ++j) {
            double v = wb[j];
            for (int k = 0; k < r; ++k) v = fma(Q[(size_t)k * n + j], c[k], v);
            ue[j] = ue[j] + v;
        }
    }
    /* phase 3 */
    for (int b = 0; b < B; ++b) {
        int first = ORC_BLK_K * b + 1, last = blk_end(nt, B, b);
        memcpy(x, ROW(L->u, L, first - 1), sizeof(double) * (size_t)n);
        for (int i = first; i < last; ++i) {
            phi(p, lvl, i, x, p->tmp1);
            const double *gi = ROW(L->g, L, i);
            for (int j = 0; j < n; ++j) x[j] = gi[j] + p->tmp1[j];
            memcpy(ROW(L->u, L, i), x, sizeof(double) * (size_t)n);
        }
    }
    free(D); free(Q); free(x); free(y); free(what); free(Ws); free(Uh); free(c);
}

/* ------------------------------------------------------------------------------------------------
 * The same scheme for Advection1D (periodic upwind, advection_1d.py:101-143): its steps are circulant, so their
 * common eigenvectors are the Fourier modes and NO mode may be dropped (transport does not decay): the states at the block ends
 * are transformed by a radix-2 FFT (n a power of two, 64 .. 8192), the recurrence over the blocks runs on all n complex
 * amplitudes, the propagated part comes back through the inverse transform.
 *   forward   X_k = sum_j x_j e^{-2 pi i jk/n}: bit-reversed input order, stages s = 1 .. log2 n on pairs (a, b) half = 2^(s-1)
 *             apart with the twiddle w = W[j n / 2^s] of position j inside the pair's block: t = w b with
 *             t.re = fma(-w.im, b.im, w.re * b.re), t.im = fma(w.im, b.re, w.re * b.im); a' = a + t, b' = a - t.
 *             W[t] = (cos(2 pi t/n), -sin(2 pi t/n)); inverse: the conjugate twiddles, then the real parts times 1/n
 *   step      (1 + alpha) x_j - alpha x_{j-1} = u_j has the eigenvalues mu_k = (1 + alpha - alpha cos th_k) + i alpha sin th_k,
 *             th_k = 2 pi k/n; d = 1/mu = (mu.re / |mu|^2, -mu.im / |mu|^2); D_b(k) = the complex product of the block's d in
 *             step order (plain products and sums)
 *   recurrence c_b = D_b Uh with c.re = fma(-D.im, Uh.im, D.re * Uh.re), c.im = fma(D.im, Uh.re, D.re * Uh.im); Uh = what_b + c_b
 *   block end u[e_b] = W_b + Re(IFFT(c_b)) / n
 * ---------------------------------------------------------------------------------------------- */
/* cos and sin of one angle as two separate libm calls: a compiler that merges them into sincos() gets, on glibc, a last bit
 * that differs from cos() / sin() for a few arguments (9 of the 16320 twiddle angles up to n = 8192), and the engine's tables
 * must hold the same bits as these */
static double __attribute__((noinline)) sep_cos(double x) { return cos(x); }
static double __attribute__((noinline)) sep_sin(double x) { return sin(x); }

static int adv_block_solve_on(const orc_problem *p, const orc_level *L, int lvl) {
    const orc_stepper *st = &L->st;
    int n = st->n;
    if (p->no_block_solve || lvl == 0 || st->kind != ORC_ADVECTION1D || !st->variant) return 0;
    if (n < 64 || n > 8192) return 0;      /* (n a power of two: radix-2 transforms; any other n: the plain sums, orc_dft_*) */
    return blk_count(L->nt) > 0;
}

/* in-place radix-2 transform of n complex values (re, im interleaved); inverse: conjugate twiddles, no scaling */
void orc_fft_spec(double *x, int n, const double *W, int inverse) {
    int lg = 0;
    while ((1 << lg) < n) ++lg;
    for (int i = 0; i < n; ++i) {
        int r = 0;
        for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
        if (r > i) {
            double tr = x[2 * i], ti = x[2 * i + 1];
            x[2 * i] = x[2 * r]; x[2 * i + 1] = x[2 * r + 1];
            x[2 * r] = tr; x[2 * r + 1] = ti;
        }
    }
    for (int s = 1; s <= lg; ++s) {
        int m = 1 << s, half = m >> 1, stride = n / m;
        for (int base = 0; base < n; base += m)
            for (int j = 0; j < half; ++j) {
                double wr = W[2 * (j * stride)], wi = inverse ? -W[2 * (j * stride) + 1] : W[2 * (j * stride) + 1];
                double *a = x + 2 * (base + j), *b = x + 2 * (base + j + half);
                double tr = fma(-wi, b[1], wr * b[0]), ti = fma(wi, b[0], wr * b[1]);
                double ar = a[0], ai = a[1];
                a[0] = ar + tr; a[1] = ai + ti;
                b[0] = ar - tr; b[1] = ai - ti;
            }
    }
}

void orc_fft_twiddles(int n, double *W /* [n/2][2] */) {
    for (int t = 0; t < n / 2; ++t) {
        double ang = 2.0 * M_PI * (double)t / (double)n;
        W[2 * t] = sep_cos(ang); W[2 * t + 1] = -sep_sin(ang);
    }
}

/* n NOT a power of two (round 5): the transforms as the plain sums over a table of all n roots of unity, T[m] = (cos, -sin)(2 pi m/n),
 * every sum one fma chain in ascending index order from 0.0 (what a K loop of v_mfma_f64_16x16x4 computes):
 *   forward  what(k) = ( sum_j x_j T[(j k) mod n].re , sum_j x_j T[(j k) mod n].im )
 *   inverse  y_j = sum_k ( c(k).re T[(j k) mod n].re , then c(k).im T[(j k) mod n].im )   = Re(sum_k c(k) e^{+i th}), unscaled */
void orc_dft_table(int n, double *T /* [n][2] */) {
    for (int m = 0; m < n; ++m) {
        double ang = 2.0 * M_PI * (double)m / (double)n;
        T[2 * m] = sep_cos(ang); T[2 * m + 1] = -sep_sin(ang);
    }
}

void orc_dft_fwd_spec(const double *x, int n, const double *T, double *out /* [n][2] */) {
#pragma omp parallel for schedule(static) num_threads(orc_h2d_threads) if (orc_h2d_threads > 1 && n >= 256)     /* (independent sums) */
    for (int k = 0; k < n; ++k) {
        double re = 0.0, im = 0.0;
        long m = 0;
        for (int j = 0; j < n; ++j) {
            re = fma(x[j], T[2 * m], re);
            im = fma(x[j], T[2 * m + 1], im);
            m += k; if (m >= n) m -= n;
        }
        out[2 * k] = re; out[2 * k + 1] = im;
    }
}

void orc_dft_inv_real_spec(const double *c /* [n][2] */, int n, const double *T, double *y /* [n] */) {
#pragma omp parallel for schedule(static) num_threads(orc_h2d_threads) if (orc_h2d_threads > 1 && n >= 256)
    for (int j = 0; j < n; ++j) {
        double acc = 0.0;
        long m = 0;
        for (int k = 0; k < n; ++k) {
            acc = fma(c[2 * k], T[2 * m], acc);
            acc = fma(c[2 * k + 1], T[2 * m + 1], acc);
            m += j; if (m >= n) m -= n;
        }
        y[j] = acc;
    }
}

/* D[b][k] complex, k < n, of block b (steps K b + 1 .. its end) */
void orc_adv_block_propagators(int n, double fac, int nt, const double *t, double *D /* [B][n][2] */) {
    int B = blk_count(nt);
    for (int b = 0; b < B; ++b) {
        int first = ORC_BLK_K * b + 1, last = blk_end(nt, B, b);
        for (int k = 0; k < n; ++k) {
            double th = 2.0 * M_PI * (double)k / (double)n, cs = sep_cos(th), sn = sep_sin(th);
            double pr = 1.0, pi = 0.0;
            for (int i = first; i <= last; ++i) {
                double alpha = (t[i] - t[i - 1]) * fac;
                double mr = (1.0 + alpha) - alpha * cs, mi = alpha * sn, den = mr * mr + mi * mi;
                double dr = mr / den, di = -mi / den;
                double qr = pr * dr - pi * di, qi = pr * di + pi * dr;
                pr = qr; pi = qi;
            }
            D[((size_t)b * n + k) * 2] = pr; D[((size_t)b * n + k) * 2 + 1] = pi;
        }
    }
}

static void advection_block_solve_spec(orc_problem *p, int lvl) {
    orc_level *L = &p->L[lvl];
    int n = L->n, nt = L->nt, B = blk_count(nt);
    const int pow2 = (n & (n - 1)) == 0;
    double *W = (double *)malloc(sizeof(double) * (size_t)2 * n), *D = (double *)malloc(sizeof(double) * (size_t)B * n * 2);
    double *yd = (double *)malloc(sizeof(double) * (size_t)n);
    if (pow2) orc_fft_twiddles(n, W);
    else orc_dft_table(n, W);
    orc_adv_block_propagators(n, L->st.fac, nt, L->t, D);
    double *x = (double *)calloc((size_t)n, sizeof(double)), *y = (double *)calloc((size_t)n, sizeof(double));
    double *what = (double *)calloc((size_t)B * n * 2, sizeof(double)), *Ws = (double *)calloc((size_t)B * n, sizeof(double));
    double *Uh = (double *)calloc((size_t)2 * n, sizeof(double)), *c = (double *)calloc((size_t)2 * n, sizeof(double));
    double inv_n = 1.0 / (double)n;
    for (int b = 0; b < B; ++b) {   /* phase 1 */
        int first = ORC_BLK_K * b + 1, last = blk_end(nt, B, b);
        for (int i = first; i <= last; ++i) {
            const double *up = ROW(L->u, L, i - 1), *ui = ROW(L->u, L, i), *gi = ROW(L->g, L, i);
            if (i == first) memcpy(y, up, sizeof(double) * (size_t)n);
            else for (int j = 0; j < n; ++j) y[j] = up[j] + x[j];
            phi(p, lvl, i, y, p->tmp1);
            for (int j = 0; j < n; ++j) x[j] = (gi[j] + p->tmp1[j]) - ui[j];
        }
        memcpy(Ws + (size_t)b * n, x, sizeof(double) * (size_t)n);
        if (b < B - 1) {
            double *wb = what + (size_t)b * n * 2;
            if (pow2) {
                for (int j = 0; j < n; ++j) { wb[2 * j] = x[j]; wb[2 * j + 1] = 0.0; }
                orc_fft_spec(wb, n, W, 0);
            } else orc_dft_fwd_spec(x, n, W, wb);
        }
    }
    memcpy(Uh, what, sizeof(double) * (size_t)2 * n);
    for (int b = 0; b < B; ++b) {   /* recurrence over the blocks + block ends */
        double *ue = ROW(L->u, L, blk_end(nt, B, b));
        const double *ws = Ws + (size_t)b * n;
        if (b == 0) { for (int j = 0; j < n; ++j) ue[j] = ue[j] + ws[j]; continue; }
        const double *Db = D + (size_t)b * n * 2, *wb = what + (size_t)b * n * 2;
        for (int k = 0; k < n; ++k) {
            double dr = Db[2 * k], di = Db[2 * k + 1], ur = Uh[2 * k], ui = Uh[2 * k + 1];
            c[2 * k] = fma(-di, ui, dr * ur);
            c[2 * k + 1] = fma(di, ur, dr * ui);
        }
        if (b < B - 1) for (int k = 0; k < 2 * n; ++k) Uh[k] = wb[k] + c[k];
        if (pow2) {
            orc_fft_spec(c, n, W, 1);
            for (int j = 0; j < n; ++j) ue[j] = ue[j] + (ws[j] + c[2 * j] * inv_n);
        } else {
            orc_dft_inv_real_spec(c, n, W, yd);
            for (int j = 0; j < n; ++j) ue[j] = ue[j] + (ws[j] + yd[j] * inv_n);
        }
    }
    for (int b = 0; b < B; ++b) {   /* phase 3 */
        int first = ORC_BLK_K * b + 1, last = blk_end(nt, B, b);
        memcpy(x, ROW(L->u, L, first - 1), sizeof(double) * (size_t)n);
        for (int i = first; i < last; ++i) {
            phi(p, lvl, i, x, p->tmp1);
            const double *gi = ROW(L->g, L, i);
            for (int j = 0; j < n; ++j) x[j] = gi[j] + p->tmp1[j];
            memcpy(ROW(L->u, L, i), x, sizeof(double) * (size_t)n);
        }
    }
    free(W); free(D); free(x); free(y); free(what); free(Ws); free(Uh); free(c); free(yd);
}

/* ------------------------------------------------------------------------------------------------
 * ... and for Heat2D with backward Euler (theta = 1; heat_2d.py:322-366): the implicit solve of 3.5 IS a
 * diagonalisation, U = Qx ((Qx B Qy) o D) Qy, so the propagator of a block is the elementwise product of the steps' D tables and
 * the recurrence over the blocks runs on the FULL sine spectrum of the interior (the transforms are those of 3.5, unscaled):
 *   what_b = fwd_y(fwd_x(interior of u[e_b]));  D_b = product of dinv(dt_i) over the block's steps in step order;
 *   c_b = D_b o Uh, Uh = what_b + c_b;  interior of u[e_b] += inv_y(inv_x(c_b)).  The rim of every state holds the boundary values.
 * (theta < 1 reads the old rim through the explicit half of the stencil: not covered, solved step by step.)
 * ---------------------------------------------------------------------------------------------- */
static int h2d_block_solve_on(const orc_problem *p, const orc_level *L, int lvl) {
    const orc_stepper *st = &L->st;
    if (p->no_block_solve || lvl == 0 || st->kind != ORC_HEAT2D || !(st->theta == 1.0 || st->theta == 0.5)) return 0;
    return blk_count(L->nt) > 0;
}

static void h2d_dinv_table(const orc_stepper *st, double dt, double *tab) {
    int mi = st->mi, mj = st->mj, Mi = st->Mi, Mj = st->Mj;
    int hxe = (mi + 1) / 2, hxo = mi / 2, hye = (mj + 1) / 2, hyo = mj / 2;
    double thdt = st->theta * dt;
    memset(tab, 0, sizeof(double) * (size_t)Mi * Mj);
    for (int a = 0; a < Mi; ++a) {
        if (!((a < hxe) || (a >= st->HPx && a < st->HPx + hxo))) continue;
        for (int b = 0; b < Mj; ++b)
            if ((b < hye) || (b >= st->HPy && b < st->HPy + hyo))
                tab[(size_t)a * Mj + b] = 1.0 / (1.0 + thdt * (st->lx[a] + st->ly[b]));
    }
}

/* the propagator of one step in mode (a, b): backward Euler 1 / (1 + dt lam); Crank-Nicolson (round 5) (1 - theta dt lam) * (1 / (1 +
 * theta dt lam)) -- the explicit half of heat_2d.py:309 carries theta as well. For an error whose RIM is zero the explicit stencil
 * reads interior values only and is diagonal in the sine basis; heat2d_block_solve_spec checks exactly that. */
static void h2d_prop_table(const orc_stepper *st, double dt, double *tab) {
    h2d_dinv_table(st, dt, tab);
    if (st->theta == 1.0) return;
    int mi = st->mi, mj = st->mj, Mi = st->Mi, Mj = st->Mj;
    int hxe = (mi + 1) / 2, hxo = mi / 2, hye = (mj + 1) / 2, hyo = mj / 2;
    double thdt = st->theta * dt;
    for (int a = 0; a < Mi; ++a) {
        if (!((a < hxe) || (a >= st->HPx && a < st->HPx + hxo))) continue;
        for (int b = 0; b < Mj; ++b)
            if ((b < hye) || (b >= st->HPy && b < st->HPy + hyo))
                tab[(size_t)a * Mj + b] = (1.0 - thdt * (st->lx[a] + st->ly[b])) * tab[(size_t)a * Mj + b];
    }
}

/* returns 0 when the level has to be solved step by step after all (theta < 1 and an error with a non-zero rim): nothing was changed */
static int heat2d_block_solve_spec(orc_problem *p, int lvl) {
    orc_level *L = &p->L[lvl];
    orc_stepper *st = &L->st;
    int n = L->n, nt = L->nt, B = blk_count(nt), ny = st->ny, mi = st->mi, mj = st->mj, Mi = st->Mi, Mj = st->Mj;
    size_t per = (size_t)Mi * Mj;
    double *x = (double *)calloc((size_t)n, sizeof(double)), *y = (double *)calloc((size_t)n, sizeof(double));
    double *what = (double *)calloc((size_t)B * per, sizeof(double)), *Ws = (double *)calloc((size_t)B * n, sizeof(double));
    double *D = (double *)malloc(sizeof(double) * per), *dv = (double *)malloc(sizeof(double) * per);
    double *Uh = (double *)calloc(per, sizeof(double)), *c = (double *)calloc(per, sizeof(double));
    double *P = (double *)calloc(per, sizeof(double)), *X = (double *)calloc(per, sizeof(double));
    for (int b = 0; b < B; ++b) {   /* phase 1 */
        int first = ORC_BLK_K * b + 1, last = blk_end(nt, B, b);
        for (int i = first; i <= last; ++i) {
            const double *up = ROW(L->u, L, i - 1), *ui = ROW(L->u, L, i), *gi = ROW(L->g, L, i);
            if (i == first) memcpy(y, up, sizeof(double) * (size_t)n);
            else for (int j = 0; j < n; ++j) y[j] = up[j] + x[j];
            phi(p, lvl, i, y, p->tmp1);
            for (int j = 0; j < n; ++j) x[j] = (gi[j] + p->tmp1[j]) - ui[j];
        }
        memcpy(Ws + (size_t)b * n, x, sizeof(double) * (size_t)n);
        if (b < B - 1) {
            memset(P, 0, sizeof(double) * per);
            for (int a = 0; a < mi; ++a)
                for (int q = 0; q < mj; ++q) P[(size_t)a * Mj + q] = x[(size_t)(a + 1) * ny + (q + 1)];
            h2d_fwd(st->Fxe, st->Fxo, mi, st->HPx, P, Mj, X);
            h2d_fwd(st->Fye, st->Fyo, mj, st->HPy, X, Mi, what + (size_t)b * per);
        }
    }
    if (st->theta != 1.0) {
        /* theta < 1: the explicit half of a step reads the rim of its input. An error with a zero rim -- every state of the level
         * carries the boundary values there, the usual case -- propagates through interior values alone, which is what the modes
         * describe; otherwise (a C-point never relaxed that still holds an initial guess with another rim) the level is stepped */
        int rim = 0;
        for (int b = 0; b < B - 1 && !rim; ++b) {
            const double *ws = Ws + (size_t)b * n;
            for (int a = 0; a < st->nx && !rim; ++a)
                for (int q = 0; q < ny; ++q)
                    if ((a == 0 || a == st->nx - 1 || q == 0 || q == ny - 1) && ws[(size_t)a * ny + q] != 0.0) { rim = 1; break; }
        }
        if (rim) {
            st->dinv_dt = -1.0;
            free(x); free(y); free(what); free(Ws); free(D); free(dv); free(Uh); free(c); free(P); free(X);
            return 0;
        }
    }
    memcpy(Uh, what, sizeof(double) * per);
    for (int b = 0; b < B; ++b) {   /* recurrence over the blocks + block ends: u[e_b] = (u[e_b] + W_b) + the propagated part (interior) */
        int first = ORC_BLK_K * b + 1, last = blk_end(nt, B, b);
        double *ue = ROW(L->u, L, last);
        const double *ws = Ws + (size_t)b * n;
        for (int j = 0; j < n; ++j) ue[j] = ue[j] + ws[j];
        if (b == 0) continue;
        for (int i = first; i <= last; ++i) {
            h2d_prop_table(st, L->t[i] - L->t[i - 1], dv);
            if (i == first) memcpy(D, dv, sizeof(double) * per);
            else for (size_t q = 0; q < per; ++q) D[q] = D[q] * dv[q];
        }
        for (size_t q = 0; q < per; ++q) c[q] = D[q] * Uh[q];
        if (b < B - 1) for (size_t q = 0; q < per; ++q) Uh[q] = what[(size_t)b * per + q] + c[q];
        h2d_inv(st->Fxe, st->Fxo, mi, st->HPx, c, Mj, X);
        h2d_inv(st->Fye, st->Fyo, mj, st->HPy, X, Mi, P);
        for (int a = 0; a < mi; ++a)
            for (int q = 0; q < mj; ++q) {
                size_t g = (size_t)(a + 1) * ny + (q + 1);
                ue[g] = ue[g] + P[(size_t)a * Mj + q];
            }
    }
    for (int b = 0; b < B; ++b) {   /* phase 3 */
        int first = ORC_BLK_K * b + 1, last = blk_end(nt, B, b);
        memcpy(x, ROW(L->u, L, first - 1), sizeof(double) * (size_t)n);
        for (int i = first; i < last; ++i) {
            phi(p, lvl, i, x, p->tmp1);
            const double *gi = ROW(L->g, L, i);
            for (int j = 0; j < n; ++j) x[j] = gi[j] + p->tmp1[j];
            memcpy(ROW(L->u, L, i), x, sizeof(double) * (size_t)n);
        }
    }
    st->dinv_dt = -1.0;
    free(x); free(y); free(what); free(Ws); free(D); free(dv); free(Uh); free(c); free(P); free(X);
    return 1;
}

void orc_problem_set_block_solve(orc_problem *p, int on) { p->no_block_solve = !on; }

void orc_forward_solve(orc_problem *p, int lvl) {
    orc_level *L = &p->L[lvl];
    { int r = block_solve_on(p, L, lvl); if (r > 0) { heat1d_block_solve_spec(p, lvl, r); return; } }
    if (adv_block_solve_on(p, L, lvl)) { advection_block_solve_spec(p, lvl); return; }
    if (h2d_block_solve_on(p, L, lvl) && heat2d_block_solve_spec(p, lvl)) return;
    if (chain_overlapped(L, lvl)) { heat1d_chain_spec(p, lvl); return; }
    for (int i = 1; i < L->nt; ++i) {
        if (lvl == 0) phi(p, lvl, i, ROW(L->u, L, i - 1), ROW(L->u, L, i));
        else {
            phi(p, lvl, i, ROW(L->u, L, i - 1), p->tmp1);
            double *ui = ROW(L->u, L, i); const double *gi = ROW(L->g, L, i);
            for (int j = 0; j < L->n; ++j) ui[j] = gi[j] + p->tmp1[j];
        }
    }
}

/* AtMgrit.forward_solve, one rank (at_mgrit.py:79-87): every coarsest-level point p is recomputed from the OLD value k-1
 * points back by k-1 steps with the FAS right-hand side; the points are independent of each other. One level: nothing
 * happens (at_mgrit.py:45). */
void orc_at_forward_solve(orc_problem *p, int lvl, int k) {
    orc_level *L = &p->L[lvl];
    if (p->n_levels == 1) return;
    size_t sz = (size_t)L->nt * (size_t)L->n;
    double *old = (double *)malloc(sizeof(double) * sz), *tmp = (double *)malloc(sizeof(double) * (size_t)L->n);
    memcpy(old, L->u, sizeof(double) * sz);
    for (int pt = 0; pt < L->nt; ++pt) {
        int s0 = pt - k + 1 > 0 ? pt - k + 1 : 0;
        double *cur = ROW(L->u, L, pt);
        memcpy(cur, old + (size_t)s0 * (size_t)L->n, sizeof(double) * (size_t)L->n);
        for (int i = (pt - k + 2 > 1 ? pt - k + 2 : 1); i <= pt; ++i) {
            phi(p, lvl, i, cur, tmp);
            const double *gi = ROW(L->g, L, i);
            for (int j = 0; j < L->n; ++j) cur[j] = gi[j] + tmp[j];
        }
    }
    free(old); free(tmp);
}

void orc_problem_set_at(orc_problem *p, int k) { p->at_k = k; }

/* mgrit.py:488-549 */
void orc_fas_residual(orc_problem *p, int lvl) {
    orc_level *L = &p->L[lvl], *C = &p->L[lvl + 1];
    int j = 0;
#ifdef _OPENMP
    if (p->threads > 1) { fas_residual_mt(p, lvl); return; }
#endif
    for (int i = 0; i < L->nt; ++i) if (L->is_c[i]) { restrict_vec(L->transfer, ROW(L->u, L, i), L->n, ROW(C->u, C, j), C->n); ++j; }
    memcpy(C->v, C->u, sizeof(double) * (size_t)C->nt * (size_t)C->n);
    j = 0;
    for (int i = 0; i < L->nt; ++i) {
        if (!L->is_c[i]) continue;
        if (j != 0) {
            double *a = p->tmp1, *r = p->tmp2, *b = p->tmp3;
            phi(p, lvl, i, ROW(L->u, L, i - 1), a);
            const double *ui = ROW(L->u, L, i);
            if (lvl == 0) for (int k = 0; k < L->n; ++k) a[k] = a[k] - ui[k];
            else { const double *gi = ROW(L->g, L, i); for (int k = 0; k < L->n; ++k) a[k] = gi[k] - ui[k] + a[k]; }
            restrict_vec(L->transfer, a, L->n, r, C->n);
            phi(p, lvl + 1, j, ROW(C->v, C, j - 1), b);
            double *gj = ROW(C->g, C, j); const double *vj = ROW(C->v, C, j);
            for (int k = 0; k < C->n; ++k) gj[k] = r[k] + vj[k] - b[k];
        }
        ++j;
    }
}

/* mgrit.py:715-726 */
void orc_error_correction(orc_problem *p, int lvl) {
    orc_level *L = &p->L[lvl], *C = &p->L[lvl + 1];
    int j = 0;
#ifdef _OPENMP
    if (p->threads > 1) { error_correction_mt(p, lvl); return; }
#endif
    for (int i = 0; i < L->nt; ++i) {
        if (!L->is_c[i]) continue;
        if (j != 0) {
            const double *uj = ROW(C->u, C, j), *vj = ROW(C->v, C, j);
            for (int k = 0; k < C->n; ++k) p->tmp1[k] = uj[k] - vj[k];
            interp_vec(L->transfer, p->tmp1, C->n, p->tmp2, L->n);
            double *ui = ROW(L->u, L, i);
            for (int k = 0; k < L->n; ++k) ui[k] = ui[k] + p->tmp2[k];
        }
        ++j;
    }
}

/* mgrit.py:261-290 */
void orc_iteration(orc_problem *p, int lvl, int cycle_type, int iteration, int first_f) {
    if (lvl == p->n_levels - 1) { if (p->at_k > 0) orc_at_forward_solve(p, lvl, p->at_k); else orc_forward_solve(p, lvl); return; }
    if ((lvl > 0 || (iteration == 0 && lvl == 0)) && first_f) orc_f_relax(p, lvl);
    for (int k = 0; k < p->cf_iter[lvl]; ++k) { orc_c_relax(p, lvl); orc_f_relax(p, lvl); }
    orc_fas_residual(p, lvl);
    orc_iteration(p, lvl + 1, cycle_type, iteration, 1);
    orc_error_correction(p, lvl);
    orc_f_relax(p, lvl);
    if (lvl != 0 && cycle_type == 1) orc_iteration(p, lvl, 0, iteration, 0);
}

/* mgrit.py:551-566 */
void orc_nested_iteration(orc_problem *p) {
    if (p->at_k > 0) orc_at_forward_solve(p, p->n_levels - 1, p->at_k); else orc_forward_solve(p, p->n_levels - 1);
    for (int lvl = p->n_levels - 2; lvl >= 0; --lvl) {
        orc_level *L = &p->L[lvl], *C = &p->L[lvl + 1];
        int j = 0;
        for (int i = 0; i < L->nt; ++i) {
            if (!L->is_c[i]) continue;
            if (j != 0) interp_vec(L->transfer, ROW(C->u, C, j), C->n, ROW(L->u, L, i), L->n);
            ++j;
        }
        if (lvl > 0) orc_iteration(p, lvl, 0, 0, 1);
    }
}

static double vec_norm(const orc_problem *p, const double *r, int n) {
    if (p->norm_spec && p->L[0].st.kind == ORC_HEAT2D) return sqrt(orc_sumsq_rows(r, p->L[0].st.nx, p->L[0].st.ny));
    if (p->norm_spec && p->L[0].st.kind == ORC_HEAT1D_2PTS) return sqrt(orc_sumsq_spec_2pts(r, n / 2));
    if (p->norm_spec) return sqrt(orc_sumsq_spec(r, n));
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += r[j] * r[j];
    return sqrt(s);
}

/* mgrit.py:387-413 -> per-C-point residual norms (count returned) */
int orc_compute_residual(orc_problem *p, double *r_norm) {
    orc_level *L = &p->L[0];
    int cnt = 0;
#ifdef _OPENMP
    if (p->threads > 1 && p->L[0].st.kind != ORC_HEAT1D_2PTS) return compute_residual_mt(p, r_norm);
#endif
    for (int i = 1; i < L->nt; ++i) {
        if (!L->is_c[i]) continue;
        phi(p, 0, i, ROW(L->u, L, i - 1), p->tmp1);
        const double *ui = ROW(L->u, L, i);
        for (int j = 0; j < L->n; ++j) p->tmp1[j] = p->tmp1[j] - ui[j];
        r_norm[cnt++] = vec_norm(p, p->tmp1, L->n);
    }
    return cnt;
}

/* mgrit.py:372-385 */
static int compute_jump(orc_problem *p, double *r_norm) {
    orc_level *L = &p->L[0];
    int cnt = 0;
    for (int i = 1; i < L->nt; ++i) {
        if (!L->is_c[i]) continue;
        const double *ui = ROW(L->u, L, i), *si = ROW(p->save_last, L, i);
        for (int j = 0; j < L->n; ++j) p->tmp1[j] = ui[j] - si[j];
        r_norm[cnt++] = vec_norm(p, p->tmp1, L->n);
    }
    memcpy(p->save_last, L->u, sizeof(double) * (size_t)L->nt * (size_t)L->n);
    return cnt;
}

/* np.linalg.norm(list, ord) with ord in {1, None(2), inf}  (mgrit.py:182,430) */
double orc_time_norm(const double *v, int n, int t_norm) {
    double s = 0.0;
    if (t_norm == 1) { for (int i = 0; i < n; ++i) s += fabs(v[i]); return s; }
    if (t_norm == 3) { for (int i = 0; i < n; ++i) if (fabs(v[i]) > s) s = fabs(v[i]); return s; }
    for (int i = 0; i < n; ++i) s += v[i] * v[i];
    return sqrt(s);
}

/* Mgrit.__init__ state part + nested iteration (mgrit.py:206-235). keep_u0: keep caller-provided level-0 guess */
void orc_setup(orc_problem *p) {
    if (p->nested) orc_nested_iteration(p);
    if (p->conv_crit == 1) {
        orc_level *L = &p->L[0];
        size_t sz = (size_t)L->nt * (size_t)L->n;
        free(p->save_last);
        p->save_last = (double *)malloc(sizeof(double) * sz);
        memcpy(p->save_last, L->u, sizeof(double) * sz);
    }
}

/* Mgrit.solve (mgrit.py:590-646). conv_out capacity max_iter. Returns number of iterations performed. */
int orc_solve(orc_problem *p, double *conv_out) {
    orc_level *L = &p->L[0];
    double *r_norm = (double *)malloc(sizeof(double) * (size_t)L->nt);
    int it;
    for (it = 0; it < p->max_iter; ++it) {
        orc_iteration(p, 0, p->cycle_type, it, 1);
        int cnt = (p->conv_crit == 0) ? orc_compute_residual(p, r_norm) : compute_jump(p, r_norm);
        conv_out[it] = orc_time_norm(r_norm, cnt, p->t_norm);
        if (conv_out[it] < p->tol || it == p->max_iter - 1) { ++it; break; }
    }
    free(r_norm);
    return it;
}

/* stand-alone helpers for unit tests ---------------------------------------------------------- */
void orc_restrict(int kind, const double *f, int nf, double *c, int nc) { restrict_vec(kind, f, nf, c, nc); }
void orc_interp(int kind, const double *c, int nc, double *f, int nf) { interp_vec(kind, c, nc, f, nf); }

/* coefficient-set introspection (tests compare the product's host-built tables with these) */
int orc_cset_dump(orc_problem *p, int lvl, double dt, double *scalars /*[4+17+6+1+64]*/, double *tab) {
    orc_stepper *st = &p->L[lvl].st;
    if (st->kind == ORC_DAHLQUIST) return -1;
    orc_cset *c = get_cset(st, dt);
    int o = 0;
    scalars[o++] = c->rho; scalars[o++] = c->ik; scalars[o++] = c->scal; scalars[o++] = c->gc;
    for (int k = 0; k <= ORC_E; ++k) scalars[o++] = c->pw[k];
    for (int s = 0; s < 6; ++s) scalars[o++] = c->sc[s];
    for (int l = 0; l < ORC_LANES; ++l) scalars[o++] = c->lp[l];
    memcpy(tab, c->tab, sizeof(double) * (size_t)st->n);
    return o;
}
