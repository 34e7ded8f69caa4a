/*
 * ff_oracle.c -- CPU restatement of the reference's QP re-sampling step st.finer_fit (st.py:584-723).
 *
 * TEST INFRASTRUCTURE ONLY (same rule as st_oracle.c): only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product never does.
 *
 * Parity status, in two parts:
 *  (1) QP CONSTRUCTION -- PINNED.  Fine-grid length, scipy/numpy linear interpolation and the matrices q, G, h, A, b
 *      are checked bit-for-bit against what the reference's own finer_fit hands to cvxopt.solvers.qp
 *      (tests/golden/golden_qp.npz, recorded by tests/golden/make_golden_qp.py with a recording stand-in for the
 *      solver).
 *  (2) QP SOLVE -- PARITY UNPINNED.  The reference solves with cvxopt.solvers.qp (third-party, version not pinned:
 *      requirements.txt:5 says just "cvxopt"; st.py:16-17 sets show_progress=False, maxiters=10).  cvxopt is not
 *      installed in the build image and cannot be, so no output of the reference's solve exists to compare with.
 *      ff_coneqp() below restates cvxopt's published algorithm for this problem class (coneqp with only the
 *      componentwise cone: a Mehrotra predictor-corrector path-following method with Nesterov-Todd scaling, which for
 *      the nonnegative orthant reduces to the classic primal-dual scaling; L. Vandenberghe, "The CVXOPT linear and
 *      quadratic cone program solvers", 2010, sections 5-7), including its default starting point, step rule
 *      (0.99 of the way to the boundary), centering exponent 3, tolerances (abstol 1e-7, reltol 1e-6, feastol 1e-7)
 *      and the reference's iteration cap of 10 after which the current iterate is returned as is.  It is validated
 *      by optimality conditions and against an independent dense solve (tests/test_finer_fit.py), not against
 *      cvxopt's bits.
 *
 * The arithmetic below is organised exactly like the GPU kernel (one problem per 64-lane wavefront, lane i owning
 * variable i and the constraint rows that start at i, sums across lanes done by an xor-butterfly) so that the two
 * agree bit-for-bit; lanes are array indices here.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define FF_NMAX 64
enum { F_V1 = 0, F_V2, F_A3, F_A4, F_J5, F_J6, F_LO, F_HI, F_NFAM };

typedef struct {
    int n;                       /* number of fine samples = variables, 2..64 */
    double b[FF_NMAX];           /* interpolated coarse path (q = -2 b), st.py:597-607 */
    double beq;                  /* s_sequence[0], st.py:708-711 */
    double cv;                   /* 1 / delta_t                     st.py:613-614 */
    double ca1, ca2;             /* 1 / dt^2, 2 / dt^2              st.py:627-633 */
    double cj1, cj2, cj3;        /* 1 / dt^3, 2 / dt^3, 3 / dt^3    st.py:645-658 */
    double hV2;                  /* MAX_SPEED                       st.py:621-622 */
    double hA3_0, hA3, hA4_0, hA4;                  /* st.py:629,634,639-641 */
    double hJ5_0, hJ5_1, hJ5, hJ6_0, hJ6_1, hJ6;    /* st.py:648,653,659,664-668 */
    int32_t has_lo[FF_NMAX], has_hi[FF_NMAX];       /* position bounds C_7, st.py:670-702 */
    double lo_h[FF_NMAX], hi_h[FF_NMAX];
} ff_qp;

typedef struct { double v_max, a_max, a_min, j_max, j_min, car_length; } ff_settings;

/* libm pow through a volatile pointer (Python's float ** int), as in st_oracle.c */
static double (*volatile ff_pow)(double, double) = pow;

/* st.py:590-595: t = arange(len)*coarse_dt; sub_length = int(round(t[-1]/dt + 1)); minus one if it overshoots */
int ff_sub_length(int len, double dt, double cdt)
{
    double t_last = (double)(len - 1) * cdt;
    int sub = (int)rint(t_last / dt + 1.0);          /* np.round: half to even */
    if ((double)(sub - 1) * dt > t_last) sub -= 1;
    return sub;
}

/* scipy.interpolate.interp1d(kind='linear') on float64 1-D data delegates to numpy.interp (st.py:597-598) */
static double ff_interp(const double *xp, const double *fp, int np_, double x)
{
    int j = 0;                                       /* xp[j] <= x < xp[j+1] */
    if (x >= xp[np_ - 1]) return fp[np_ - 1];
    while (j + 1 < np_ && xp[j + 1] <= x) j++;
    if (xp[j] == x) return fp[j];
    double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    return slope * (x - xp[j]) + fp[j];
}

/* Build the QP of st.finer_fit. bac = {before_s, before_speed, after_s, after_speed} or NULL. Returns n, or <0. */
int ff_build(const double *s_seq, int len, double dt, double cdt, double v0, double a0, const double *bac,
             const ff_settings *S, ff_qp *q)
{
    memset(q, 0, sizeof *q);
    if (len < 2) return -1;
    int n = ff_sub_length(len, dt, cdt);
    if (n < 2 || n > FF_NMAX) return -2;
    q->n = n;
    double t[FF_NMAX * 4];
    if (len > FF_NMAX * 4) return -3;
    for (int i = 0; i < len; i++) t[i] = (double)i * cdt;
    for (int i = 0; i < n; i++) q->b[i] = ff_interp(t, s_seq, len, (double)i * dt);
    q->beq = s_seq[0];
    double dt2 = ff_pow(dt, 2.0), dt3 = ff_pow(dt, 3.0);
    q->cv = 1.0 / dt;
    q->ca1 = 1.0 / dt2; q->ca2 = 2.0 / dt2;
    q->cj1 = 1.0 / dt3; q->cj2 = 2.0 / dt3; q->cj3 = 3.0 / dt3;
    q->hV2 = S->v_max;
    q->hA3_0 = S->a_max + v0 / dt; q->hA3 = S->a_max;
    q->hA4_0 = -S->a_min - v0 / dt; q->hA4 = -S->a_min;
    q->hJ5_0 = S->j_max + a0 / dt + v0 / dt2; q->hJ5_1 = S->j_max - v0 / dt2; q->hJ5 = S->j_max;
    q->hJ6_0 = -S->j_min - a0 / dt - v0 / dt2; q->hJ6_1 = -S->j_min + v0 / dt2; q->hJ6 = -S->j_min;
    if (bac) {
        double before_s = bac[0], before_v = bac[1], after_s = bac[2], after_v = bac[3];
        for (int i = 0; i < n; i++) {
            double ti = (double)i * dt;
            if (!isinf(before_s)) {
                double pr = before_s + ti * before_v;
                if (!(pr < -S->car_length)) { q->has_lo[i] = 1; q->lo_h[i] = -pr - S->car_length; }
            }
            if (!isinf(after_s)) {
                double pr = after_s + ti * after_v;
                if (!(pr < -S->car_length)) { q->has_hi[i] = 1; q->hi_h[i] = pr - S->car_length; }
            }
        }
    }
    return n;
}

/* coefficients of the "+" row r of the V / A / J family on columns r-2 .. r+1 (st.py:609-660) */
static void ff_coef(const ff_qp *q, int r, double cV[4], double cA[4], double cJ[4])
{
    cV[0] = 0.0; cV[1] = 0.0; cV[2] = q->cv; cV[3] = -q->cv;
    if (r == 0) { cA[0] = 0.0; cA[1] = 0.0; cA[2] = -q->ca1; cA[3] = q->ca1; }
    else        { cA[0] = 0.0; cA[1] = q->ca1; cA[2] = -q->ca2; cA[3] = q->ca1; }
    if (r == 0)      { cJ[0] = 0.0; cJ[1] = 0.0; cJ[2] = -q->cj1; cJ[3] = q->cj1; }
    else if (r == 1) { cJ[0] = 0.0; cJ[1] = q->cj2; cJ[2] = -q->cj3; cJ[3] = q->cj1; }
    else             { cJ[0] = -q->cj1; cJ[1] = q->cj3; cJ[2] = -q->cj3; cJ[3] = q->cj1; }
}

static double ff_h(const ff_qp *q, int f, int r)
{
    switch (f) {
    case F_V1: return 0.0;
    case F_V2: return q->hV2;
    case F_A3: return r == 0 ? q->hA3_0 : q->hA3;
    case F_A4: return r == 0 ? q->hA4_0 : q->hA4;
    case F_J5: return r == 0 ? q->hJ5_0 : (r == 1 ? q->hJ5_1 : q->hJ5);
    case F_J6: return r == 0 ? q->hJ6_0 : (r == 1 ? q->hJ6_1 : q->hJ6);
    case F_LO: return q->lo_h[r];
    default:   return q->hi_h[r];
    }
}

static int ff_active(const ff_qp *q, int f, int r)
{
    if (f <= F_J6) return r < q->n - 1;
    if (f == F_LO) return r < q->n && q->has_lo[r];
    return r < q->n && q->has_hi[r];
}

/* number of inequality rows, and the dense matrices in the reference's row order (st.py:715-719) -- golden check */
int ff_rows(const ff_qp *q)
{
    int m = 6 * (q->n - 1);
    for (int i = 0; i < q->n; i++) m += (q->has_lo[i] ? 1 : 0) + (q->has_hi[i] ? 1 : 0);
    return m;
}

void ff_dense(const ff_qp *q, double *G /* [m][n] */, double *h /* [m] */, double *qv /* [n] */)
{
    int n = q->n, m = ff_rows(q), row = 0;
    memset(G, 0, sizeof(double) * (size_t)m * n);
    for (int f = F_V1; f <= F_J6; f++) {
        for (int r = 0; r < n - 1; r++, row++) {
            double cV[4], cA[4], cJ[4];
            ff_coef(q, r, cV, cA, cJ);
            const double *c = (f <= F_V2) ? cV : (f <= F_A4 ? cA : cJ);
            double sgn = (f & 1) ? -1.0 : 1.0;
            for (int k = 0; k < 4; k++) {
                int col = r - 2 + k;
                if (col >= 0 && col < n && c[k] != 0.0) G[(size_t)row * n + col] = sgn * c[k];
            }
            h[row] = ff_h(q, f, r);
        }
    }
    for (int i = 0; i < n; i++) if (q->has_lo[i]) { G[(size_t)row * n + i] = -1.0; h[row++] = q->lo_h[i]; }
    for (int i = 0; i < n; i++) if (q->has_hi[i]) { G[(size_t)row * n + i] = 1.0; h[row++] = q->hi_h[i]; }
    for (int i = 0; i < n; i++) qv[i] = -2.0 * q->b[i];
}

/* ---- emulated wavefront reductions: xor butterfly, every lane ends with the same value ---- */
static double wave_sum(const double *v)
{
    double a[64], b[64];
    memcpy(a, v, sizeof a);
    for (int off = 32; off >= 1; off >>= 1) {
        for (int i = 0; i < 64; i++) b[i] = a[i] + a[i ^ off];
        memcpy(a, b, sizeof a);
    }
    return a[0];
}
static double wave_max(const double *v)
{
    double m = v[0];
    for (int i = 1; i < 64; i++) m = v[i] > m ? v[i] : m;
    return m;
}

typedef struct {
    double cV[64][4], cA[64][4], cJ[64][4];
    double h[F_NFAM][64];
    int act[F_NFAM][64];
    double l1[64], l2[64], l3[64], invd[64];      /* banded L D L' factor of P + G' D G */
} ff_work;

static inline double at(const double *x, int i) { return (i >= 0 && i < 64) ? x[i] : 0.0; }

/* (G x) for the three "+" families at row r */
static void ff_gx(const ff_work *w, const double *x, int r, double *gV, double *gA, double *gJ)
{
    double xm2 = at(x, r - 2), xm1 = at(x, r - 1), x0 = at(x, r), xp1 = at(x, r + 1);
    *gV = ((w->cV[r][0] * xm2 + w->cV[r][1] * xm1) + w->cV[r][2] * x0) + w->cV[r][3] * xp1;
    *gA = ((w->cA[r][0] * xm2 + w->cA[r][1] * xm1) + w->cA[r][2] * x0) + w->cA[r][3] * xp1;
    *gJ = ((w->cJ[r][0] * xm2 + w->cJ[r][1] * xm1) + w->cJ[r][2] * x0) + w->cJ[r][3] * xp1;
}

/* out = G' u for per-row values u[f][r] (inactive rows must hold 0) */
static void ff_gt(const ff_work *w, double u[F_NFAM][64], double *out)
{
    double t[64][4];
    for (int r = 0; r < 64; r++) {
        double wV = u[F_V1][r] - u[F_V2][r], wA = u[F_A3][r] - u[F_A4][r], wJ = u[F_J5][r] - u[F_J6][r];
        for (int k = 0; k < 4; k++) t[r][k] = (w->cV[r][k] * wV + w->cA[r][k] * wA) + w->cJ[r][k] * wJ;
    }
    for (int i = 0; i < 64; i++) {
        double a0 = (i + 2 < 64) ? t[i + 2][0] : 0.0, a1 = (i + 1 < 64) ? t[i + 1][1] : 0.0, a2 = t[i][2];
        double a3 = (i - 1 >= 0) ? t[i - 1][3] : 0.0;
        out[i] = (((a0 + a1) + a2) + a3) + (u[F_HI][i] - u[F_LO][i]);
    }
}

/* factor S = 2 I + G' diag(D) G (7 diagonals) as L diag(d) L', rows in order; inactive lanes get d = 2 */
static void ff_factor(ff_work *w, double D[F_NFAM][64])
{
    double T[64][4][4];
    for (int r = 0; r < 64; r++) {
        double DV = D[F_V1][r] + D[F_V2][r], DA = D[F_A3][r] + D[F_A4][r], DJ = D[F_J5][r] + D[F_J6][r];
        for (int k = 0; k < 4; k++)
            for (int l = 0; l <= k; l++)
                T[r][k][l] = (((DV * w->cV[r][k]) * w->cV[r][l]) + ((DA * w->cA[r][k]) * w->cA[r][l])) + ((DJ * w->cJ[r][k]) * w->cJ[r][l]);
    }
    for (int i = 0; i < 64; i++) {
        double S[4];
        for (int dl = 0; dl < 4; dl++) {
            double acc = 0.0;
            for (int k = dl; k < 4; k++) {
                int r = i + 2 - k;
                acc = acc + ((r >= 0 && r < 64) ? T[r][k][k - dl] : 0.0);
            }
            S[dl] = acc;
        }
        S[0] = (2.0 + S[0]) + (D[F_LO][i] + D[F_HI][i]);
        double e3 = S[3];
        double e2 = S[2] - e3 * at(w->l1, i - 2);
        double e1 = (S[1] - e3 * at(w->l2, i - 1)) - e2 * at(w->l1, i - 1);
        double l3 = e3 * at(w->invd, i - 3), l2 = e2 * at(w->invd, i - 2), l1 = e1 * at(w->invd, i - 1);
        double d = ((S[0] - e3 * l3) - e2 * l2) - e1 * l1;
        w->l1[i] = l1; w->l2[i] = l2; w->l3[i] = l3; w->invd[i] = 1.0 / d;
    }
}

static void ff_solve(const ff_work *w, const double *rhs, double *u)
{
    double y[64];
    for (int i = 0; i < 64; i++)
        y[i] = ((rhs[i] - w->l1[i] * at(y, i - 1)) - w->l2[i] * at(y, i - 2)) - w->l3[i] * at(y, i - 3);
    for (int i = 0; i < 64; i++) y[i] = y[i] * w->invd[i];
    for (int i = 63; i >= 0; i--)
        u[i] = ((y[i] - at(w->l1, i + 1) * at(u, i + 1)) - at(w->l2, i + 2) * at(u, i + 2)) - at(w->l3, i + 3) * at(u, i + 3);
}

/*
 * cvxopt coneqp restated for: minimise x'x - 2 b'x  s.t.  G x + s = h, s >= 0, x_0 = beq.
 * Returns the number of iterations done; *status = 0 converged ("optimal"), 1 iteration cap reached ("unknown").
 */
int ff_coneqp_tol(const ff_qp *q, int maxiters, const double *tol /* abstol, reltol, feastol or NULL = cvxopt defaults */,
                  double *x_out, int *status)
{
    const double STEP = 0.99;
    const double ABSTOL = tol ? tol[0] : 1e-7, RELTOL = tol ? tol[1] : 1e-6, FEASTOL = tol ? tol[2] : 1e-7;
    static const double ZERO64[64] = {0};
    ff_work w;
    memset(&w, 0, sizeof w);
    const int n = q->n;
    int m = 0;
    for (int r = 0; r < 64; r++) {
        if (r < n - 1) ff_coef(q, r, w.cV[r], w.cA[r], w.cJ[r]);
        for (int f = 0; f < F_NFAM; f++) {
            w.act[f][r] = ff_active(q, f, r);
            w.h[f][r] = w.act[f][r] ? ff_h(q, f, r) : 0.0;
            m += w.act[f][r];
        }
    }
    double x[64] = {0}, y = 0.0, s[F_NFAM][64], z[F_NFAM][64], tmp[64], e0[64] = {0}, v[64], u[64];
    double qv[64] = {0};
    for (int i = 0; i < n; i++) qv[i] = -2.0 * q->b[i];
    e0[0] = 1.0;

    /* norms of the data for the relative residuals */
    for (int i = 0; i < 64; i++) tmp[i] = qv[i] * qv[i];
    double resx0 = sqrt(wave_sum(tmp)); resx0 = resx0 > 1.0 ? resx0 : 1.0;
    double resy0 = fabs(q->beq) > 1.0 ? fabs(q->beq) : 1.0;
    for (int i = 0; i < 64; i++) { double a = 0.0; for (int f = 0; f < F_NFAM; f++) a = a + w.h[f][i] * w.h[f][i]; tmp[i] = a; }
    double resz0 = sqrt(wave_sum(tmp)); resz0 = resz0 > 1.0 ? resz0 : 1.0;

    /* starting point: [P A' G'; A 0 0; G 0 -I] [x; y; z] = [-q; b; h], s = -z, then shifted into the cone */
    double D[F_NFAM][64], uu[F_NFAM][64];
    for (int f = 0; f < F_NFAM; f++) for (int i = 0; i < 64; i++) { D[f][i] = w.act[f][i] ? 1.0 : 0.0; uu[f][i] = w.h[f][i]; }
    ff_factor(&w, D);
    ff_gt(&w, uu, tmp);
    for (int i = 0; i < 64; i++) tmp[i] = (i < n) ? (-qv[i] + tmp[i]) : 0.0;
    ff_solve(&w, tmp, u);
    ff_solve(&w, e0, v);
    y = (u[0] - q->beq) / v[0];
    for (int i = 0; i < 64; i++) x[i] = (i < n) ? u[i] - v[i] * y : 0.0;
    {
        double mxs[64], mxz[64], ns[64];
        for (int r = 0; r < 64; r++) {
            double gV, gA, gJ; ff_gx(&w, x, r, &gV, &gA, &gJ);
            double g[F_NFAM] = {gV, -gV, gA, -gA, gJ, -gJ, -x[r], x[r]};
            double a = 0.0, ms = -INFINITY, mz = -INFINITY;
            for (int f = 0; f < F_NFAM; f++) {
                if (w.act[f][r]) {
                    z[f][r] = g[f] - w.h[f][r]; s[f][r] = -z[f][r];
                    a = a + s[f][r] * s[f][r];
                    ms = -s[f][r] > ms ? -s[f][r] : ms; mz = -z[f][r] > mz ? -z[f][r] : mz;
                } else { s[f][r] = 1.0; z[f][r] = 0.0; }
            }
            ns[r] = a; mxs[r] = ms; mxz[r] = mz;
        }
        double nrm = sqrt(wave_sum(ns));          /* |s| = |z| */
        double ts = wave_max(mxs), tz = wave_max(mxz);
        double thr = -1e-8 * (nrm > 1.0 ? nrm : 1.0);
        for (int f = 0; f < F_NFAM; f++) for (int r = 0; r < 64; r++) if (w.act[f][r]) {
            if (ts >= thr) s[f][r] = s[f][r] + (1.0 + ts);
            if (tz >= thr) z[f][r] = z[f][r] + (1.0 + tz);
        }
    }
    for (int i = 0; i < 64; i++) { double a = 0.0; for (int f = 0; f < F_NFAM; f++) if (w.act[f][i]) a = a + s[f][i] * z[f][i]; tmp[i] = a; }
    double gap = wave_sum(tmp);

    int iters;
    *status = 1;
    for (iters = 0; iters <= maxiters; iters++) {
        /* residuals: rx = P x + q + A' y + G' z, ry = A x - b, rz = s + G x - h */
        double rx[64], rz[F_NFAM][64], ry;
        for (int f = 0; f < F_NFAM; f++) for (int r = 0; r < 64; r++) uu[f][r] = w.act[f][r] ? z[f][r] : 0.0;
        ff_gt(&w, uu, tmp);
        double f0p[64], rzz[64], rzn[64];
        for (int i = 0; i < 64; i++) {
            double px = 2.0 * x[i] + qv[i];
            f0p[i] = x[i] * px + x[i] * qv[i];
            rx[i] = (i < n) ? (px + (i == 0 ? y : 0.0)) + tmp[i] : 0.0;
        }
        double f0 = 0.5 * wave_sum(f0p);
        for (int i = 0; i < 64; i++) tmp[i] = rx[i] * rx[i];
        double resx = sqrt(wave_sum(tmp));
        ry = x[0] - q->beq;
        double resy = fabs(ry);
        for (int r = 0; r < 64; r++) {
            double gV, gA, gJ; ff_gx(&w, x, r, &gV, &gA, &gJ);
            double g[F_NFAM] = {gV, -gV, gA, -gA, gJ, -gJ, -x[r], x[r]};
            double a = 0.0, c = 0.0;
            for (int f = 0; f < F_NFAM; f++) {
                rz[f][r] = w.act[f][r] ? (s[f][r] - w.h[f][r]) + g[f] : 0.0;
                a = a + rz[f][r] * rz[f][r];
                c = c + (w.act[f][r] ? z[f][r] * rz[f][r] : 0.0);
            }
            rzn[r] = a; rzz[r] = c;
        }
        double resz = sqrt(wave_sum(rzn));
        double pcost = f0, dcost = ((f0 + y * ry) + wave_sum(rzz)) - gap;
        int have_rel = 0; double relgap = 0.0;
        if (pcost < 0.0) { relgap = gap / -pcost; have_rel = 1; }
        else if (dcost > 0.0) { relgap = gap / dcost; have_rel = 1; }
        double pres = resy / resy0 > resz / resz0 ? resy / resy0 : resz / resz0;
        double dres = resx / resx0;
        if (pres <= FEASTOL && dres <= FEASTOL && (gap <= ABSTOL || (have_rel && relgap <= RELTOL))) { *status = 0; break; }
        if (iters == maxiters) break;

        /* scaling W'W = diag(s/z): S = P + G' diag(z/s) G */
        for (int f = 0; f < F_NFAM; f++) for (int r = 0; r < 64; r++) D[f][r] = w.act[f][r] ? z[f][r] / s[f][r] : 0.0;
        ff_factor(&w, D);
        ff_solve(&w, e0, v);
        const double mu = gap / (double)m;
        double sigma = 0.0, step = 1.0;
        double ds[F_NFAM][64], dz[F_NFAM][64], dx[64], dy = 0.0, dsa_dza[F_NFAM][64];
        for (int pass = 0; pass < 2; pass++) {
            /* lambda o (dz + ds) = -lambda o lambda [- ds_aff o dz_aff] + sigma mu e, in unscaled form z ds + s dz = bs */
            for (int f = 0; f < F_NFAM; f++) for (int r = 0; r < 64; r++) {
                if (!w.act[f][r]) { uu[f][r] = 0.0; continue; }
                double bs = -(s[f][r] * z[f][r]);
                if (pass == 1) bs = (bs - dsa_dza[f][r]) + sigma * mu;
                uu[f][r] = (bs + z[f][r] * rz[f][r]) / s[f][r];           /* tau */
            }
            ff_gt(&w, uu, tmp);
            for (int i = 0; i < 64; i++) tmp[i] = (i < n) ? -rx[i] - tmp[i] : 0.0;
            ff_solve(&w, tmp, u);
            dy = (u[0] + ry) / v[0];
            for (int i = 0; i < 64; i++) dx[i] = (i < n) ? u[i] - v[i] * dy : 0.0;
            double pd[64], mxs[64], mxz[64];
            for (int r = 0; r < 64; r++) {
                double gV, gA, gJ; ff_gx(&w, dx, r, &gV, &gA, &gJ);
                double g[F_NFAM] = {gV, -gV, gA, -gA, gJ, -gJ, -dx[r], dx[r]};
                double a = 0.0, ms = -INFINITY, mz = -INFINITY;
                for (int f = 0; f < F_NFAM; f++) {
                    if (!w.act[f][r]) { ds[f][r] = 0.0; dz[f][r] = 0.0; continue; }
                    dz[f][r] = uu[f][r] + D[f][r] * g[f];
                    ds[f][r] = -rz[f][r] - g[f];
                    a = a + ds[f][r] * dz[f][r];
                    double qs = -ds[f][r] / s[f][r], qz = -dz[f][r] / z[f][r];
                    ms = qs > ms ? qs : ms; mz = qz > mz ? qz : mz;
                }
                pd[r] = a; mxs[r] = ms; mxz[r] = mz;
            }
            double dsdz = wave_sum(pd);
            double t = wave_max(mxs), tz = wave_max(mxz);
            t = t > tz ? t : tz; t = t > 0.0 ? t : 0.0;
            if (t == 0.0) step = 1.0;
            else if (pass == 0) step = 1.0 / t < 1.0 ? 1.0 / t : 1.0;
            else step = STEP / t < 1.0 ? STEP / t : 1.0;
            if (pass == 0) {
                double c = (1.0 - step) + (dsdz / gap) * (step * step);
                c = c > 0.0 ? c : 0.0; c = c < 1.0 ? c : 1.0;
                sigma = (c * c) * c;
                for (int f = 0; f < F_NFAM; f++) for (int r = 0; r < 64; r++) dsa_dza[f][r] = ds[f][r] * dz[f][r];
            }
        }
        for (int i = 0; i < 64; i++) x[i] = x[i] + step * dx[i];
        y = y + step * dy;
        for (int r = 0; r < 64; r++) {
            double a = 0.0;
            for (int f = 0; f < F_NFAM; f++) if (w.act[f][r]) {
                s[f][r] = s[f][r] + step * ds[f][r];
                z[f][r] = z[f][r] + step * dz[f][r];
                a = a + s[f][r] * z[f][r];
            }
            tmp[r] = a;
        }
        gap = wave_sum(tmp);
    }
    (void)ZERO64;
    for (int i = 0; i < n; i++) x_out[i] = x[i];
    return iters;
}

int ff_coneqp(const ff_qp *q, int maxiters, double *x_out, int *status) { return ff_coneqp_tol(q, maxiters, 0, x_out, status); }

/* st.finer_fit end to end (st.py:584-723). out must hold FF_NMAX doubles. Returns the output length (>= 1) or < 0. */
int ff_finer_fit(const double *s_seq, int len, double dt, double cdt, double v0, double a0, const double *bac,
                 const ff_settings *S, int maxiters, double *out, int *iters, int *status)
{
    if (len == 1) { out[0] = s_seq[0]; if (iters) *iters = 0; if (status) *status = 0; return 1; }   /* st.py:587-588 */
    ff_qp q;
    int n = ff_build(s_seq, len, dt, cdt, v0, a0, bac, S, &q);
    if (n < 0) return n;
    int st = 0;
    int it = ff_coneqp(&q, maxiters, out, &st);
    if (iters) *iters = it;
    if (status) *status = st;
    return n;
}
