/*
 * stm_oracle.c -- CPU restatement (fp64) of the strutopy STM E-step.
 * TEST INFRASTRUCTURE ONLY -- see stm_oracle.h for the rules and the list of
 * reference files followed.  Each function cites the file:line it restates.
 *
 * Written from the reference's behaviour, in C; the only "vectorisation" is
 * OpenMP over documents (the reference loop stm.py:519 is serial and documents
 * are independent), used for the reported CPU baseline.
 */
#include "stm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static char g_err[256];
const char *stm_oracle_last_error(void) { return g_err; }
int stm_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ */
/* Python / numpy scalar semantics the line searches rely on           */
/* ------------------------------------------------------------------ */
/* builtin max(a, b[, c]) / min(a, b): keeps the first unless a later one compares greater/less
 * (so a NaN in first position sticks, later NaNs are ignored). */
static double py_max2(double a, double b) { return (b > a) ? b : a; }
static double py_max3(double a, double b, double c) { return py_max2(py_max2(a, b), c); }
static double py_min2(double a, double b) { return (b < a) ? b : a; }
/* np.clip on scalars: minimum(maximum(x, lo), hi), NaN-propagating */
static double np_clip(double x, double lo, double hi) {
    if (isnan(x)) return x;
    double r = x < lo ? lo : x;
    return r > hi ? hi : r;
}
static double np_sign(double x) {
    if (isnan(x)) return x;
    return (x > 0) - (x < 0);
}

/* ------------------------------------------------------------------ */
/* Per-document problem                                                */
/* ------------------------------------------------------------------ */
typedef struct {
    int K, n, Nd;
    const double *mu;     /* n */
    const double *counts; /* Nd */
    const double *betad;  /* K x Nd row-major (stm.py:617 gather) */
    const double *siginv; /* n x n */
    double Ndoc;          /* int(np.sum(word_count)), stm.py:933 */
    double *g0;           /* K: beta_doc @ (c / colsum(beta_doc)), stm.py:954 -- eta independent */
    double *w_e;          /* K scratch */
    double *w_d;          /* n scratch */
    int nfev, njev;
    /* scipy's ScalarFunction (optimize/_differentiable_functions.py) keeps the last x and
     * re-uses f / g when asked again at a bitwise-identical x; emulated so nfev/njev are
     * comparable (values are unaffected: f and df are pure). */
    double *sf_x, *sf_g, sf_f;
    int sf_have_x, sf_f_ok, sf_g_ok;
} doc_t;

static void sf_update_x(doc_t *d, const double *x) {
    int same = d->sf_have_x;
    for (int i = 0; same && i < d->n; ++i) same = (x[i] == d->sf_x[i]);
    if (!same) {
        memcpy(d->sf_x, x, sizeof(double) * (size_t)d->n);
        d->sf_have_x = 1; d->sf_f_ok = 0; d->sf_g_ok = 0;
    }
}

/* scipy.special.logsumexp as installed (scipy/special/_logsumexp.py:_logsumexp):
 * the maximal elements are taken out of the sum, out = log1p(s/m) + log(m) + a_max. */
static double scipy_logsumexp(const double *a, int len) {
    double amax = a[0];
    for (int i = 1; i < len; ++i)
        if (a[i] > amax || isnan(a[i])) amax = a[i]; /* np.max propagates NaN */
    double m = 0.0, s = 0.0;
    double shift = isfinite(amax) ? amax : 0.0;
    for (int i = 0; i < len; ++i) {
        if (a[i] == amax) m += 1.0;
        else s += exp(a[i] - shift);
    }
    if (s != 0.0) s = s / m;
    return log1p(s) + log(m) + amax;
}

/* stm.py:920-944  f(eta) */
static double obj_f_raw(doc_t *d, const double *eta);
static double obj_f(doc_t *d, const double *eta) {
    sf_update_x(d, eta);
    if (!d->sf_f_ok) { d->sf_f = obj_f_raw(d, eta); d->sf_f_ok = 1; }
    return d->sf_f;
}
static double obj_f_raw(doc_t *d, const double *eta) {
    const int K = d->K, n = d->n, Nd = d->Nd;
    d->nfev++;
    /* eta_ = np.insert(eta, K-1, 0) ; m = eta_.max() */
    double m = 0.0;
    for (int k = 0; k < n; ++k)
        if (eta[k] > m || isnan(eta[k])) m = eta[k];
    double *e = d->w_e;
    for (int k = 0; k < n; ++k) e[k] = exp(eta[k] - m);
    e[K - 1] = exp(0.0 - m);
    /* np.dot(word_count, m + np.log(np.exp(eta_ - m) @ beta_doc)) */
    double part = 0.0;
    for (int v = 0; v < Nd; ++v) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += e[k] * d->betad[(size_t)k * Nd + v];
        part += d->counts[v] * (m + log(s));
    }
    /* Ndoc * logsumexp(eta_) */
    double *et = d->w_e; /* reuse: rebuild eta_ */
    for (int k = 0; k < n; ++k) et[k] = eta[k];
    et[K - 1] = 0.0;
    double lse = scipy_logsumexp(et, K);
    /* 0.5 * (eta-mu).T @ siginv @ (eta-mu) */
    double *df = d->w_d;
    for (int k = 0; k < n; ++k) df[k] = eta[k] - d->mu[k];
    double quad = 0.0;
    for (int i = 0; i < n; ++i) {
        /* (diff.T @ siginv)[i] then dotted with diff */
        double t = 0.0;
        for (int j = 0; j < n; ++j) t += df[j] * d->siginv[(size_t)j * n + i];
        quad += t * df[i];
    }
    quad *= 0.5;
    return quad - (part - d->Ndoc * lse);
}

/* stm.py:946-958  df(eta).  NB the data term has no eta dependence (g0). */
static void obj_df_raw(doc_t *d, const double *eta, double *g);
static void obj_df(doc_t *d, const double *eta, double *g) {
    sf_update_x(d, eta);
    if (!d->sf_g_ok) { obj_df_raw(d, eta, d->sf_g); d->sf_g_ok = 1; }
    memcpy(g, d->sf_g, sizeof(double) * (size_t)d->n);
}
static void obj_df_raw(doc_t *d, const double *eta, double *g) {
    const int n = d->n;
    d->njev++;
    double sumexp = 0.0;
    for (int k = 0; k < n; ++k) sumexp += exp(eta[k]);
    sumexp += exp(0.0);
    /* np.sum(word_count) / np.sum(np.exp(eta_)) */
    double scale = d->Ndoc / sumexp;
    for (int i = 0; i < n; ++i) {
        double t = 0.0;
        for (int j = 0; j < n; ++j) t += d->siginv[(size_t)i * n + j] * (eta[j] - d->mu[j]);
        g[i] = t - (d->g0[i] - scale * exp(eta[i]));
    }
}

static void doc_prepare_g0(doc_t *d) {
    const int K = d->K, Nd = d->Nd;
    for (int k = 0; k < K; ++k) d->g0[k] = 0.0;
    for (int v = 0; v < Nd; ++v) {
        double cs = 0.0;
        for (int k = 0; k < K; ++k) cs += d->betad[(size_t)k * Nd + v];
        double w = d->counts[v] / cs;
        for (int k = 0; k < K; ++k) d->g0[k] += d->betad[(size_t)k * Nd + v] * w;
    }
}

/* ------------------------------------------------------------------ */
/* Line-search plumbing: phi(s) = f(xk + s*pk), derphi(s) = df(.)·pk  */
/* (scipy/optimize/_linesearch.py:77-85, 293-303)                     */
/* ------------------------------------------------------------------ */
typedef struct {
    doc_t *d;
    const double *xk, *pk;
    double *xt;   /* trial point */
    double *gval; /* gradient at the last derphi() call */
} ls_t;

static double ls_phi(ls_t *L, double s) {
    const int n = L->d->n;
    for (int i = 0; i < n; ++i) L->xt[i] = L->xk[i] + s * L->pk[i];
    return obj_f(L->d, L->xt);
}
static double ls_derphi(ls_t *L, double s) {
    const int n = L->d->n;
    for (int i = 0; i < n; ++i) L->xt[i] = L->xk[i] + s * L->pk[i];
    obj_df(L->d, L->xt, L->gval);
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += L->gval[i] * L->pk[i];
    return r;
}

/* scipy/optimize/_dcsrch.py:502-728  dcstep */
static void dcstep(double *stx, double *fx, double *dx, double *sty, double *fy, double *dy,
                   double *stp, double fp, double dp, int *brackt, double stpmin, double stpmax) {
    double sgnd = np_sign(dp) * np_sign(*dx);
    double stpf, stpc, stpq, theta, s, gamma, p, q, r;
    if (fp > *fx) {
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = py_max3(fabs(theta), fabs(*dx), fabs(dp));
        gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp < *stx) gamma *= -1;
        p = (gamma - *dx) + theta;
        q = ((gamma - *dx) + gamma) + dp;
        r = p / q;
        stpc = *stx + r * (*stp - *stx);
        stpq = *stx + ((*dx / ((*fx - fp) / (*stp - *stx) + *dx)) / 2.0) * (*stp - *stx);
        if (fabs(stpc - *stx) <= fabs(stpq - *stx)) stpf = stpc;
        else stpf = stpc + (stpq - stpc) / 2.0;
        *brackt = 1;
    } else if (sgnd < 0.0) {
        theta = 3 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = py_max3(fabs(theta), fabs(*dx), fabs(dp));
        gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp > *stx) gamma *= -1;
        p = (gamma - dp) + theta;
        q = ((gamma - dp) + gamma) + *dx;
        r = p / q;
        stpc = *stp + r * (*stx - *stp);
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (fabs(stpc - *stp) > fabs(stpq - *stp)) stpf = stpc;
        else stpf = stpq;
        *brackt = 1;
    } else if (fabs(dp) < fabs(*dx)) {
        theta = 3 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = py_max3(fabs(theta), fabs(*dx), fabs(dp));
        gamma = s * sqrt(py_max2(0.0, (theta / s) * (theta / s) - (*dx / s) * (dp / s)));
        if (*stp > *stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        q = (gamma + (*dx - dp)) + gamma;
        r = p / q;
        if (r < 0 && gamma != 0) stpc = *stp + r * (*stx - *stp);
        else if (*stp > *stx) stpc = stpmax;
        else stpc = stpmin;
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (*brackt) {
            if (fabs(stpc - *stp) < fabs(stpq - *stp)) stpf = stpc;
            else stpf = stpq;
            if (*stp > *stx) stpf = py_min2(*stp + 0.66 * (*sty - *stp), stpf);
            else stpf = py_max2(*stp + 0.66 * (*sty - *stp), stpf);
        } else {
            if (fabs(stpc - *stp) > fabs(stpq - *stp)) stpf = stpc;
            else stpf = stpq;
            stpf = np_clip(stpf, stpmin, stpmax);
        }
    } else {
        if (*brackt) {
            theta = 3.0 * (fp - *fy) / (*sty - *stp) + *dy + dp;
            s = py_max3(fabs(theta), fabs(*dy), fabs(dp));
            gamma = s * sqrt((theta / s) * (theta / s) - (*dy / s) * (dp / s));
            if (*stp > *sty) gamma = -gamma;
            p = (gamma - dp) + theta;
            q = ((gamma - dp) + gamma) + *dy;
            r = p / q;
            stpc = *stp + r * (*sty - *stp);
            stpf = stpc;
        } else if (*stp > *stx) stpf = stpmax;
        else stpf = stpmin;
    }
    if (fp > *fx) {
        *sty = *stp; *fy = fp; *dy = dp;
    } else {
        if (sgnd < 0) { *sty = *stx; *fy = *fx; *dy = *dx; }
        *stx = *stp; *fx = fp; *dx = dp;
    }
    *stp = stpf;
}

/* scipy/optimize/_linesearch.py:93-165 scalar_search_wolfe1 +
 * scipy/optimize/_dcsrch.py:201-500 DCSRCH.__call__/_iterate.
 * Returns 1 and (*stp_out, *phi1_out) on CONVERGENCE, 0 when alpha is None. */
static int search_wolfe1(ls_t *L, double phi0, double old_phi0, double derphi0, double c1,
                         double c2, double amax, double amin, double xtol, double *stp_out,
                         double *phi1_out) {
    double alpha1;
    if (derphi0 != 0) {
        alpha1 = py_min2(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0);
        if (alpha1 < 0) alpha1 = 1.0;
    } else alpha1 = 1.0;

    const double ftol = c1, gtol = c2, stpmin = amin, stpmax = amax;
    const double p5 = 0.5, p66 = 0.66, xtrapl = 1.1, xtrapu = 4.0;
    int brackt = 0, stage = 1;
    double finit = 0, ginit = 0, gtest = 0, width = 0, width1 = 0;
    double stx = 0, fx = 0, gx = 0, sty = 0, fy = 0, gy = 0, stmin = 0, stmax = 0;
    double stp = alpha1, f = phi0, g = derphi0;
    int started = 0;
    for (int it = 0; it < 100; ++it) {
        /* ---- _iterate ---- */
        int task; /* 0 FG, 1 CONV, 2 WARN, 3 ERROR */
        if (!started) {
            started = 1;
            int err = 0;
            if (stp < stpmin) err = 1;
            if (stp > stpmax) err = 1;
            if (g >= 0) err = 1;
            if (err) return 0; /* ERROR -> alpha None */
            brackt = 0; stage = 1; finit = f; ginit = g; gtest = ftol * ginit;
            width = stpmax - stpmin; width1 = width / p5;
            stx = 0.0; fx = finit; gx = ginit; sty = 0.0; fy = finit; gy = ginit;
            stmin = 0; stmax = stp + xtrapu * stp;
            task = 0;
        } else {
            double ftest = finit + stp * gtest;
            if (stage == 1 && f <= ftest && g >= 0) stage = 2;
            task = 0;
            if (brackt && (stp <= stmin || stp >= stmax)) task = 2;
            if (brackt && stmax - stmin <= xtol * stmax) task = 2;
            if (stp == stpmax && f <= ftest && g <= gtest) task = 2;
            if (stp == stpmin && (f > ftest || g >= gtest)) task = 2;
            if (f <= ftest && fabs(g) <= gtol * -ginit) task = 1;
            if (task == 0) {
                if (stage == 1 && f <= fx && f > ftest) {
                    double fm = f - stp * gtest, fxm = fx - stx * gtest, fym = fy - sty * gtest;
                    double gm = g - gtest, gxm = gx - gtest, gym = gy - gtest;
                    dcstep(&stx, &fxm, &gxm, &sty, &fym, &gym, &stp, fm, gm, &brackt, stmin, stmax);
                    fx = fxm + stx * gtest; fy = fym + sty * gtest;
                    gx = gxm + gtest; gy = gym + gtest;
                } else {
                    dcstep(&stx, &fx, &gx, &sty, &fy, &gy, &stp, f, g, &brackt, stmin, stmax);
                }
                if (brackt) {
                    if (fabs(sty - stx) >= p66 * width1) stp = stx + p5 * (sty - stx);
                    width1 = width;
                    width = fabs(sty - stx);
                }
                if (brackt) {
                    stmin = py_min2(stx, sty);
                    stmax = py_max2(stx, sty);
                } else {
                    stmin = stp + xtrapl * (stp - stx);
                    stmax = stp + xtrapu * (stp - stx);
                }
                stp = np_clip(stp, stpmin, stpmax);
                if ((brackt && (stp <= stmin || stp >= stmax)) ||
                    (brackt && stmax - stmin <= xtol * stmax))
                    stp = stx;
            }
        }
        /* ---- __call__ body ---- */
        if (!isfinite(stp)) return 0;
        if (task == 0) {
            f = ls_phi(L, stp);
            g = ls_derphi(L, stp);
        } else {
            if (task == 1) { *stp_out = stp; *phi1_out = f; return 1; }
            return 0;
        }
    }
    return 0; /* maxiter reached */
}

/* scipy/optimize/_linesearch.py:477-508 _cubicmin.  Returns 0 for None. */
static int cubicmin(double a, double fa, double fpa, double b, double fb, double c, double fc,
                    double *xmin) {
    double C = fpa, db = b - a, dc = c - a;
    double t = db * dc;
    double denom = (t * t) * (db - dc);
    double d00 = dc * dc, d01 = -(db * db), d10 = -(dc * dc * dc), d11 = db * db * db;
    double v0 = fb - fa - C * db, v1 = fc - fa - C * dc;
    double A = d00 * v0 + d01 * v1, B = d10 * v0 + d11 * v1;
    if (!isfinite(denom) || !isfinite(A) || !isfinite(B) || denom == 0.0) return 0;
    A /= denom; B /= denom;
    double radical = B * B - 3 * A * C;
    if (!isfinite(A) || !isfinite(B) || !isfinite(radical) || radical < 0) return 0;
    double den2 = 3 * A;
    if (den2 == 0.0 || !isfinite(den2)) return 0;
    double x = a + (-B + sqrt(radical)) / den2;
    if (!isfinite(x)) return 0;
    *xmin = x;
    return 1;
}
/* scipy/optimize/_linesearch.py:511-529 _quadmin */
static int quadmin(double a, double fa, double fpa, double b, double fb, double *xmin) {
    double D = fa, C = fpa, db = b - a * 1.0;
    double den = db * db;
    double num = fb - D - C * db;
    if (den == 0.0 || !isfinite(den) || !isfinite(num)) return 0;
    double B = num / den;
    double den2 = 2.0 * B;
    if (den2 == 0.0 || !isfinite(den2)) return 0;
    double x = a - C / den2;
    if (!isfinite(x)) return 0;
    *xmin = x;
    return 1;
}

/* scipy/optimize/_linesearch.py:532-621 _zoom.  Returns 1 on success. */
static int zoom(ls_t *L, double a_lo, double a_hi, double phi_lo, double phi_hi, double derphi_lo,
                double phi0, double derphi0, double c1, double c2, double *a_star,
                double *val_star) {
    const int maxiter = 10;
    int i = 0;
    const double delta1 = 0.2, delta2 = 0.1;
    double phi_rec = phi0, a_rec = 0;
    for (;;) {
        double dalpha = a_hi - a_lo, a, b;
        if (dalpha < 0) { a = a_hi; b = a_lo; } else { a = a_lo; b = a_hi; }
        double a_j = 0, cchk = 0;
        int have = 0;
        if (i > 0) {
            cchk = delta1 * dalpha;
            have = cubicmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_rec, phi_rec, &a_j);
        }
        if (i == 0 || !have || a_j > b - cchk || a_j < a + cchk) {
            double qchk = delta2 * dalpha;
            have = quadmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, &a_j);
            if (!have || a_j > b - qchk || a_j < a + qchk) a_j = a_lo + 0.5 * dalpha;
        }
        double phi_aj = ls_phi(L, a_j);
        if (phi_aj > phi0 + c1 * a_j * derphi0 || phi_aj >= phi_lo) {
            phi_rec = phi_hi; a_rec = a_hi; a_hi = a_j; phi_hi = phi_aj;
        } else {
            double derphi_aj = ls_derphi(L, a_j);
            if (fabs(derphi_aj) <= -c2 * derphi0) {
                *a_star = a_j; *val_star = phi_aj;
                return 1;
            }
            if (derphi_aj * (a_hi - a_lo) >= 0) {
                phi_rec = phi_hi; a_rec = a_hi; a_hi = a_lo; phi_hi = phi_lo;
            } else {
                phi_rec = phi_lo; a_rec = a_lo;
            }
            a_lo = a_j; phi_lo = phi_aj; derphi_lo = derphi_aj;
        }
        i += 1;
        if (i > maxiter) return 0;
    }
}

/* scipy/optimize/_linesearch.py:341-474 scalar_search_wolfe2 (maxiter=10, no extra_condition).
 * Returns 0: alpha None; 1: accepted with gradient (L->gval valid at alpha);
 * 2: bracketing loop exhausted -> alpha returned, gradient None. */
static int search_wolfe2(ls_t *L, double phi0, double old_phi0, double derphi0, double c1,
                         double c2, double amax, double *alpha_star, double *phi_star) {
    double alpha0 = 0, alpha1;
    if (derphi0 != 0) alpha1 = py_min2(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0);
    else alpha1 = 1.0;
    if (alpha1 < 0) alpha1 = 1.0;
    alpha1 = py_min2(alpha1, amax);
    double phi_a1 = ls_phi(L, alpha1);
    double phi_a0 = phi0, derphi_a0 = derphi0;
    for (int i = 0; i < 10; ++i) {
        if (alpha1 == 0 || alpha0 > amax) return 0;
        if (phi_a1 > phi0 + c1 * alpha1 * derphi0 || (phi_a1 >= phi_a0 && i > 0))
            return zoom(L, alpha0, alpha1, phi_a0, phi_a1, derphi_a0, phi0, derphi0, c1, c2,
                        alpha_star, phi_star);
        double derphi_a1 = ls_derphi(L, alpha1);
        if (fabs(derphi_a1) <= -c2 * derphi0) {
            *alpha_star = alpha1; *phi_star = phi_a1;
            return 1;
        }
        if (derphi_a1 >= 0)
            return zoom(L, alpha1, alpha0, phi_a1, phi_a0, derphi_a1, phi0, derphi0, c1, c2,
                        alpha_star, phi_star);
        double alpha2 = py_min2(2 * alpha1, amax);
        alpha0 = alpha1; alpha1 = alpha2;
        phi_a0 = phi_a1; phi_a1 = ls_phi(L, alpha1);
        derphi_a0 = derphi_a1;
    }
    *alpha_star = alpha1; *phi_star = phi_a1;
    return 2;
}

/* scipy/optimize/_optimize.py:1328-1502 _minimize_bfgs with the defaults the
 * reference uses (stm.py:960-962): gtol=1e-5, norm=inf, maxiter=200*n, c1=1e-4,
 * c2=0.9, xrtol=0, H0=I.  x is updated in place; returns status. */
static int bfgs_minimize(doc_t *d, double *x, int *nit_out, double *fun_out) {
    const int n = d->n;
    const double gtol = 1e-5, c1 = 1e-4, c2 = 0.9;
    const int maxiter = n * 200;
    double *buf = (double *)malloc(sizeof(double) * ((size_t)3 * n * n + (size_t)8 * n));
    double *H = buf, *T1 = H + (size_t)n * n, *T2 = T1 + (size_t)n * n;
    double *g = T2 + (size_t)n * n, *p = g + n, *xt = p + n, *gv = xt + n, *s = gv + n,
           *y = s + n, *gnew = y + n, *xnew = gnew + n;
    for (size_t i = 0; i < (size_t)n * n; ++i) H[i] = 0.0;
    for (int i = 0; i < n; ++i) H[(size_t)i * n + i] = 1.0;

    double old_fval = obj_f(d, x);
    obj_df(d, x, g);
    int k = 0, warnflag = 0;
    double nrm2 = 0.0;
    for (int i = 0; i < n; ++i) nrm2 += g[i] * g[i];
    double old_old_fval = old_fval + sqrt(nrm2) / 2;
    double gnorm = 0.0;
    for (int i = 0; i < n; ++i) { double a = fabs(g[i]); if (a > gnorm || isnan(a)) gnorm = a; }

    while (gnorm > gtol && k < maxiter) {
        for (int i = 0; i < n; ++i) {
            double t = 0.0;
            for (int j = 0; j < n; ++j) t += H[(size_t)i * n + j] * g[j];
            p[i] = -t;
        }
        ls_t L = {d, x, p, xt, gv};
        double derphi0 = 0.0;
        for (int i = 0; i < n; ++i) derphi0 += g[i] * p[i];
        double alpha = 0, fnew = 0;
        int have_g = 0;
        /* _line_search_wolfe12: wolfe1, then wolfe2, else _LineSearchError */
        int ok = search_wolfe1(&L, old_fval, old_old_fval, derphi0, c1, c2, 1e100, 1e-100, 1e-14,
                               &alpha, &fnew);
        if (ok) have_g = 1;
        else {
            int r2 = search_wolfe2(&L, old_fval, old_old_fval, derphi0, c1, c2, 1e100, &alpha,
                                   &fnew);
            if (r2 == 0) { warnflag = 2; break; }
            ok = 1;
            have_g = (r2 == 1);
        }
        double new_old_old = old_fval;
        for (int i = 0; i < n; ++i) { s[i] = alpha * p[i]; xnew[i] = x[i] + s[i]; }
        if (have_g) for (int i = 0; i < n; ++i) gnew[i] = gv[i];
        else obj_df(d, xnew, gnew);
        for (int i = 0; i < n; ++i) { y[i] = gnew[i] - g[i]; g[i] = gnew[i]; x[i] = xnew[i]; }
        old_old_fval = new_old_old;
        old_fval = fnew;
        k += 1;
        gnorm = 0.0;
        for (int i = 0; i < n; ++i) { double a = fabs(g[i]); if (a > gnorm || isnan(a)) gnorm = a; }
        if (gnorm <= gtol) break;
        double pn = 0.0;
        for (int i = 0; i < n; ++i) pn += p[i] * p[i];
        if (alpha * sqrt(pn) <= 0.0) break; /* xrtol = 0 */
        if (!isfinite(old_fval)) { warnflag = 2; break; }
        double rhok_inv = 0.0;
        for (int i = 0; i < n; ++i) rhok_inv += y[i] * s[i];
        double rhok = (rhok_inv == 0.0) ? 1000.0 : 1.0 / rhok_inv;
        /* A1 = I - s y^T rho ; A2 = I - y s^T rho ; H = A1 (H A2) + rho s s^T */
        for (int i = 0; i < n; ++i)        /* T1 = H @ A2 */
            for (int j = 0; j < n; ++j) {
                double t = 0.0;
                for (int l = 0; l < n; ++l) {
                    double a2 = ((l == j) ? 1.0 : 0.0) - y[l] * s[j] * rhok;
                    t += H[(size_t)i * n + l] * a2;
                }
                T1[(size_t)i * n + j] = t;
            }
        for (int i = 0; i < n; ++i)        /* T2 = A1 @ T1 + rho s s^T */
            for (int j = 0; j < n; ++j) {
                double t = 0.0;
                for (int l = 0; l < n; ++l) {
                    double a1 = ((i == l) ? 1.0 : 0.0) - s[i] * y[l] * rhok;
                    t += a1 * T1[(size_t)l * n + j];
                }
                T2[(size_t)i * n + j] = t + rhok * s[i] * s[j];
            }
        memcpy(H, T2, sizeof(double) * (size_t)n * n);
    }
    if (warnflag == 2) {
    } else if (k >= maxiter) warnflag = 1;
    else {
        int anynan = isnan(gnorm) || isnan(old_fval);
        for (int i = 0; i < n; ++i) anynan |= isnan(x[i]);
        if (anynan) warnflag = 3;
    }
    *nit_out = k;
    if (fun_out) *fun_out = old_fval;
    free(buf);
    return warnflag;
}

/* ------------------------------------------------------------------ */
/* Hessian, PD fix, Cholesky, nu, bound, phi                           */
/* ------------------------------------------------------------------ */
#define STM_PIVOT_TOL (32.0 * 2.220446049250313e-16)
/* Smallest |pivot| / diagonal entry over every pivot test since it was last reset (all rungs of the PD ladder, the
 * failing pivot of a failed attempt included): how far the document stays from the STM_PIVOT_TOL band, where this
 * restatement (and the kernels) would deviate from np.linalg.cholesky's "pivot <= 0". */
static double g_pivot_margin = 1e300;
#ifdef _OPENMP
#pragma omp threadprivate(g_pivot_margin)
#endif
/* np.linalg.cholesky (lower).  Returns 0 on success, 1 when not PD. L's upper part is zeroed. */
static int chol_lower(int n, const double *A, double *L) {
    for (size_t i = 0; i < (size_t)n * n; ++i) L[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        double dsum = A[(size_t)j * n + j];
        for (int l = 0; l < j; ++l) dsum -= L[(size_t)j * n + l] * L[(size_t)j * n + l];
        /* A pivot that is nothing but the rounding left over from cancelling A[j][j] counts as failed.  make_pd can
         * produce an EXACTLY singular matrix (n = 2: [[|o|, o], [o, |o|]], every time both diagonals are raised), and
         * there the sign of the pivot -- like the sign of the smallest eigenvalue the reference tests -- is decided
         * by the last bit of the input; "failed" leads to the + 1e-5 branch instead of a factor with a 1e-8 pivot. */
        {
            const double ratio = fabs(dsum) / fabs(A[(size_t)j * n + j]);   /* NaN / 0-diagonal: recorded as 0 */
            if (!(ratio >= g_pivot_margin)) g_pivot_margin = (ratio == ratio) ? ratio : 0.0;
        }
        if (!(dsum > STM_PIVOT_TOL * A[(size_t)j * n + j])) return 1;
        double ljj = sqrt(dsum);
        L[(size_t)j * n + j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double t = A[(size_t)i * n + j];
            for (int l = 0; l < j; ++l) t -= L[(size_t)i * n + l] * L[(size_t)j * n + l];
            L[(size_t)i * n + j] = t / ljj;
        }
    }
    return 0;
}

/* stm.py:964-984 make_pd: diag <- where(diag < sum_j!=i |M_ij|, that sum, diag) */
void stm_oracle_make_pd(int n, double *M) {
    for (int i = 0; i < n; ++i) {
        double dv = M[(size_t)i * n + i];
        double mag = 0.0;
        for (int j = 0; j < n; ++j) mag += fabs(M[(size_t)i * n + j]);
        mag -= fabs(dv);
        if (dv < mag) M[(size_t)i * n + i] = mag;
    }
}

/* stm.py:905-909 */
static void stable_softmax(const double *x, int len, double *out) {
    double m = x[0];
    for (int i = 1; i < len; ++i) if (x[i] > m || isnan(x[i])) m = x[i];
    double s = 0.0;
    for (int i = 0; i < len; ++i) { out[i] = exp(x[i] - m); s += out[i]; }
    for (int i = 0; i < len; ++i) out[i] /= s;
}

/* stm.py:986-1026 hessian.  Hout is n x n.  The reference's PD test is
 * np.all(np.linalg.eigvals(f) > 0) (a general nonsymmetric eigen-solve on an exactly
 * symmetric matrix); restated as "Cholesky succeeds", which agreed with it on every
 * probed document and is validated against the pd_path goldens. */
static int hessian_pd(const doc_t *d, const double *eta, double *Hout, double *scratchL,
                      double *wK /* K*Nd + 4K */) {
    const int K = d->K, n = d->n, Nd = d->Nd;
    double *b = wK, *eta_ = b + (size_t)K * Nd, *theta = eta_ + K, *ex = theta + K,
           *rowc = ex + K;
    for (int k = 0; k < n; ++k) eta_[k] = eta[k];
    eta_[K - 1] = 0.0;
    stable_softmax(eta_, K, theta);
    for (int k = 0; k < K; ++k) { ex[k] = exp(eta_[k]); rowc[k] = 0.0; }
    for (int v = 0; v < Nd; ++v) {
        double S = 0.0;
        for (int k = 0; k < K; ++k) S += d->betad[(size_t)k * Nd + v] * ex[k];
        double sq = sqrt(d->counts[v]);
        for (int k = 0; k < K; ++k) {
            double a = d->betad[(size_t)k * Nd + v] * ex[k];
            double bb = a * sq / S;
            b[(size_t)k * Nd + v] = bb;
            rowc[k] += bb * sq;
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double t = 0.0;
            for (int v = 0; v < Nd; ++v) t += b[(size_t)i * Nd + v] * b[(size_t)j * Nd + v];
            double h = t - d->Ndoc * (theta[i] * theta[j]);
            if (i == j) h = h - rowc[i] + d->Ndoc * theta[i];
            Hout[(size_t)i * n + j] = h + d->siginv[(size_t)i * n + j];
        }
    int path = 0;
    if (chol_lower(n, Hout, scratchL)) {
        stm_oracle_make_pd(n, Hout);
        path = 1;
        if (chol_lower(n, Hout, scratchL)) {
            for (int i = 0; i < n; ++i) Hout[(size_t)i * n + i] += 1e-5;
            path = 2;
        }
    }
    return path;
}

/* stm.py:1031-1050 decompose_hessian followed by stm.py:1052-1066 optimize_nu */
int stm_oracle_decompose(int n, double *H, double *L, double *nu) {
    int path = 0;
    int upper = 0;
    if (chol_lower(n, H, L)) {
        stm_oracle_make_pd(n, H);
        path = 1;
        if (chol_lower(n, H, L)) {
            /* sp.linalg.cholesky(make_pd(hess) + 1e-5 I) -> UPPER factor U (U^T U = M) */
            stm_oracle_make_pd(n, H);
            double *M = (double *)malloc(sizeof(double) * (size_t)n * n);
            memcpy(M, H, sizeof(double) * (size_t)n * n);
            for (int i = 0; i < n; ++i) M[(size_t)i * n + i] += 1e-5;
            int bad = chol_lower(n, M, L);
            free(M);
            if (bad) return -1;
            /* transpose in place: "L" is now upper */
            for (int i = 0; i < n; ++i)
                for (int j = i + 1; j < n; ++j) {
                    L[(size_t)i * n + j] = L[(size_t)j * n + i];
                    L[(size_t)j * n + i] = 0.0;
                }
            upper = 1;
            path = 2;
        }
    }
    /* nu = inv(triu(L.T)) @ inv(triu(L.T)).T */
    double *R = (double *)malloc(sizeof(double) * (size_t)n * n); /* R = inv(triu(L^T)), upper */
    for (size_t i = 0; i < (size_t)n * n; ++i) R[i] = 0.0;
    if (upper) {
        /* L is upper, L^T lower, triu(L^T) = diag(L) */
        for (int i = 0; i < n; ++i) R[(size_t)i * n + i] = 1.0 / L[(size_t)i * n + i];
    } else {
        /* U = L^T upper; solve U R = I by back substitution, column by column */
        for (int c = 0; c < n; ++c)
            for (int i = c; i >= 0; --i) {
                double t = (i == c) ? 1.0 : 0.0;
                for (int l = i + 1; l <= c; ++l) t -= L[(size_t)l * n + i] * R[(size_t)l * n + c];
                R[(size_t)i * n + c] = t / L[(size_t)i * n + i];
            }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double t = 0.0;
            int l0 = i > j ? i : j;
            for (int l = l0; l < n; ++l) t += R[(size_t)i * n + l] * R[(size_t)j * n + l];
            nu[(size_t)i * n + j] = t;
        }
    free(R);
    return path;
}

/* stm.py:1068-1101 lower_bound */
static double lower_bound(const doc_t *d, const double *L, const double *eta, double sigmaentropy,
                          double *wK /* 3K */) {
    const int K = d->K, n = d->n, Nd = d->Nd;
    double *eta_ = wK, *theta = eta_ + K, *w = theta + K;
    for (int k = 0; k < n; ++k) eta_[k] = eta[k];
    eta_[K - 1] = 0.0;
    stable_softmax(eta_, K, theta);
    for (int k = 0; k < K; ++k) w[k] = exp(eta_[k]);
    double det = 0.0;
    for (int i = 0; i < n; ++i) det += log(L[(size_t)i * n + i]);
    double ll = 0.0;
    for (int v = 0; v < Nd; ++v) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += theta[k] * (d->betad[(size_t)k * Nd + v] * w[k]);
        ll += log(s) * d->counts[v];
    }
    double quad = 0.0;
    for (int i = 0; i < n; ++i) {
        double t = 0.0;
        for (int j = 0; j < n; ++j) t += (eta[j] - d->mu[j]) * d->siginv[(size_t)j * n + i];
        quad += t * (eta[i] - d->mu[i]);
    }
    return ll + (-det) - 0.5 * quad - sigmaentropy;
}

/* stm.py:1103-1118 update_z: phi (K x Nd) */
static void update_z(const doc_t *d, const double *eta, double *phi, double *wK /* K */) {
    const int K = d->K, n = d->n, Nd = d->Nd;
    double *ex = wK;
    for (int k = 0; k < n; ++k) ex[k] = exp(eta[k]);
    ex[K - 1] = exp(0.0);
    for (int v = 0; v < Nd; ++v) {
        double S = 0.0;
        for (int k = 0; k < K; ++k) S += d->betad[(size_t)k * Nd + v] * ex[k];
        double sq = sqrt(d->counts[v]);
        double w = sq / S;
        for (int k = 0; k < K; ++k) phi[(size_t)k * Nd + v] = (d->betad[(size_t)k * Nd + v] * ex[k]) * w * sq;
    }
}

/* ------------------------------------------------------------------ */
/* Single-function entry points for parity tests                       */
/* ------------------------------------------------------------------ */
static void doc_init(doc_t *d, int K, int Nd, const double *mu, const double *counts,
                     const double *betad, const double *siginv, double *work /* 5K */) {
    d->K = K; d->n = K - 1; d->Nd = Nd; d->mu = mu; d->counts = counts; d->betad = betad;
    d->siginv = siginv;
    double tot = 0.0;
    for (int v = 0; v < Nd; ++v) tot += counts[v];
    d->Ndoc = (double)(long long)tot;
    d->g0 = work; d->w_e = work + K; d->w_d = work + 2 * K;
    d->sf_x = work + 3 * K; d->sf_g = work + 4 * K;
    d->sf_have_x = d->sf_f_ok = d->sf_g_ok = 0; d->sf_f = 0.0;
    d->nfev = d->njev = 0;
    doc_prepare_g0(d);
}

double stm_oracle_f(int K, int Nd, const double *eta, const double *mu, const double *counts,
                    const double *betad, const double *siginv) {
    double *work = (double *)malloc(sizeof(double) * (size_t)(5 * K));
    doc_t d;
    doc_init(&d, K, Nd, mu, counts, betad, siginv, work);
    double r = obj_f(&d, eta);
    free(work);
    return r;
}
void stm_oracle_df(int K, int Nd, const double *eta, const double *mu, const double *counts,
                   const double *betad, const double *siginv, double *g) {
    double *work = (double *)malloc(sizeof(double) * (size_t)(5 * K));
    doc_t d;
    doc_init(&d, K, Nd, mu, counts, betad, siginv, work);
    obj_df(&d, eta, g);
    free(work);
}
int stm_oracle_bfgs(int K, int Nd, double *eta, const double *mu, const double *counts,
                    const double *betad, const double *siginv, int32_t *nit, int32_t *nfev,
                    int32_t *njev, double *fun) {
    double *work = (double *)malloc(sizeof(double) * (size_t)(5 * K));
    doc_t d;
    doc_init(&d, K, Nd, mu, counts, betad, siginv, work);
    int it = 0;
    int st = bfgs_minimize(&d, eta, &it, fun);
    if (nit) *nit = it;
    if (nfev) *nfev = d.nfev;
    if (njev) *njev = d.njev;
    free(work);
    return st;
}

/* ------------------------------------------------------------------ */
/* E-step driver: stm.py:489-597                                       */
/* ------------------------------------------------------------------ */
int stm_oracle_estep(const stm_oracle_args *a, int nthreads) {
    const int K = a->K, n = K - 1, V = a->V, A = a->A > 0 ? a->A : 1;
    const int64_t N = a->N;
    if (K < 2 || V < 1 || N < 0) { snprintf(g_err, sizeof g_err, "bad shape"); return 1; }
    g_err[0] = 0;
    int maxNd = 0;
    for (int64_t i = 0; i < N; ++i) {
        int64_t nd = a->indptr[i + 1] - a->indptr[i];
        if (nd > maxNd) maxNd = (int)nd;
    }
    const size_t KV = (size_t)K * V * A;
    for (size_t i = 0; i < (size_t)n * n; ++i) a->sigma_ss[i] = 0.0;
    for (size_t i = 0; i < KV; ++i) a->beta_ss[i] = 0.0;
    double *bound = a->bound;
    double *bound_own = NULL;
    if (!bound) { bound_own = (double *)calloc((size_t)(N > 0 ? N : 1), sizeof(double)); bound = bound_own; }
    int failed = 0;
#ifdef _OPENMP
    int nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    int nt = 1;
    (void)nthreads;
#endif
    if (nt < 1) nt = 1;
    /* sigma_ss: per-thread partials ((K-1)^2 each), reduced in thread order afterwards.
     * beta_ss: ONE word-major accumulator [A][V][K] shared by the threads (a document touches Nd runs of K
     * doubles; atomic adds when nt > 1), transposed into beta_ss at the end -- per-thread K x V replicas cost
     * more than the E-step itself once there are a few dozen threads.  beta is gathered from a word-major copy
     * for the same reason (stm.py:614-617 gathers K strided columns per document). */
    double **sss = (double **)calloc((size_t)nt, sizeof(double *));
    sss[0] = a->sigma_ss;
    for (int t = 1; t < nt; ++t) sss[t] = (double *)calloc((size_t)n * n, sizeof(double));
    double *bssT = (double *)calloc(KV, sizeof(double));
    double *betaT = (double *)malloc(sizeof(double) * KV);
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static)
#endif
    for (int64_t av = 0; av < (int64_t)A * V; ++av) {
        const size_t lev = (size_t)(av / V), w = (size_t)(av % V);
        for (int k = 0; k < K; ++k) betaT[(size_t)av * K + k] = a->beta[lev * K * V + (size_t)k * V + w];
    }
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
    {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        size_t mNd = (size_t)(maxNd > 0 ? maxNd : 1);
        double *betad = (double *)malloc(sizeof(double) * (size_t)K * mNd);
        double *work = (double *)malloc(sizeof(double) * (size_t)(5 * K));
        double *wK = (double *)malloc(sizeof(double) * ((size_t)K * mNd + 8 * (size_t)K));
        double *phi = (double *)malloc(sizeof(double) * (size_t)K * mNd);
        double *Hm = (double *)malloc(sizeof(double) * (size_t)n * n * 3);
        double *Lm = Hm + (size_t)n * n, *nu = Lm + (size_t)n * n;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int64_t i = 0; i < N; ++i) {
            if (failed) continue;
            const int64_t p0 = a->indptr[i];
            const int Nd = (int)(a->indptr[i + 1] - p0);
            const int asp = a->aspect ? a->aspect[i] : 0;
            const double *betaT_a = betaT + (size_t)asp * K * V;
            /* get_beta: beta[:, idx] (stm.py:614-617) + assert beta >= 0 (stm.py:534) */
            int bad = 0;
            for (int v = 0; v < Nd; ++v) {
                const double *row = betaT_a + (size_t)a->indices[p0 + v] * K;
                for (int k = 0; k < K; ++k) {
                    double bv = row[k];
                    if (!(bv >= 0)) bad = 1;
                    betad[(size_t)k * Nd + v] = bv;
                }
            }
            if (bad) {
#ifdef _OPENMP
#pragma omp critical
#endif
                { failed = 2; snprintf(g_err, sizeof g_err, "Some entries of beta are negative or nan."); }
                continue;
            }
            doc_t d;
            double *eta = a->eta + (size_t)i * n;
            doc_init(&d, K, Nd, a->mu + (size_t)i * n, a->counts + p0, betad, a->siginv, work);
            int nit = 0;
            int st = bfgs_minimize(&d, eta, &nit, NULL); /* stm.py:538-546 */
            if (a->status) a->status[i] = st;
            if (a->nit) a->nit[i] = nit;
            if (a->nfev) a->nfev[i] = d.nfev;
            if (a->njev) a->njev[i] = d.njev;
            /* theta (unshifted softmax), stm.py:547-549 */
            {
                double *th = a->theta + (size_t)i * K;
                double s = 0.0;
                for (int k = 0; k < n; ++k) { th[k] = exp(eta[k]); s += th[k]; }
                th[K - 1] = exp(0.0); s += th[K - 1];
                for (int k = 0; k < K; ++k) th[k] /= s;
            }
            g_pivot_margin = 1e300;
            int path = hessian_pd(&d, eta, Hm, Lm, wK); /* stm.py:553 */
            if (a->pd_path) a->pd_path[i] = path;
            if (a->hess_out) memcpy(a->hess_out + (size_t)i * n * n, Hm, sizeof(double) * (size_t)n * n);
            int dp = stm_oracle_decompose(n, Hm, Lm, nu); /* stm.py:556, 568 */
            if (a->pivot_margin) a->pivot_margin[i] = g_pivot_margin;
            if (dp < 0) {
#ifdef _OPENMP
#pragma omp critical
#endif
                { failed = 3; snprintf(g_err, sizeof g_err, "Cholesky failed for document %lld", (long long)i); }
                continue;
            }
            if (a->chol_out) memcpy(a->chol_out + (size_t)i * n * n, Lm, sizeof(double) * (size_t)n * n);
            if (a->nu_out) memcpy(a->nu_out + (size_t)i * n * n, nu, sizeof(double) * (size_t)n * n);
            bound[i] = lower_bound(&d, Lm, eta, a->sigmaentropy, wK); /* stm.py:559 */
            update_z(&d, eta, phi, wK);                               /* stm.py:572 */
            if (a->phi_last && i == N - 1) memcpy(a->phi_last, phi, sizeof(double) * (size_t)K * Nd);
            double *ss = sss[tid];
            for (size_t q = 0; q < (size_t)n * n; ++q) ss[q] += nu[q]; /* stm.py:582 */
            double *bs = bssT + (size_t)asp * K * V;                    /* stm.py:584-588 */
            for (int v = 0; v < Nd; ++v) {
                double *row = bs + (size_t)a->indices[p0 + v] * K;
                if (nt > 1) {
                    for (int k = 0; k < K; ++k) {
#ifdef _OPENMP
#pragma omp atomic update
#endif
                        row[k] += phi[(size_t)k * Nd + v];
                    }
                } else {
                    for (int k = 0; k < K; ++k) row[k] += phi[(size_t)k * Nd + v];
                }
            }
        }
        free(betad); free(work); free(wK); free(phi); free(Hm);
    }
    /* word-major accumulator -> beta_ss [A][K][V] */
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static)
#endif
    for (int64_t ak = 0; ak < (int64_t)A * K; ++ak) {
        const size_t lev = (size_t)(ak / K), k = (size_t)(ak % K);
        for (int w = 0; w < V; ++w) a->beta_ss[(size_t)ak * V + w] = bssT[(lev * V + (size_t)w) * K + k];
    }
    free(bssT); free(betaT);
    /* reduce the per-thread sigma_ss partials in thread order */
    for (int t = 1; t < nt; ++t) {
        for (size_t q = 0; q < (size_t)n * n; ++q) a->sigma_ss[q] += sss[t][q];
        free(sss[t]);
    }
    free(sss);
    double tot = 0.0;
    for (int64_t i = 0; i < N; ++i) tot += bound[i]; /* stm.py:592 */
    if (a->bound_total) *a->bound_total = tot;
    free(bound_own);
    return failed;
}
