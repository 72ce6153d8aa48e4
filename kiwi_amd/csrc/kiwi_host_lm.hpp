// Levenberg-Marquardt driver of `minimize_lm` (minimizer_engine.f90:728-803) for the batched engine.
//
// The reference calls `lmdif` of its single-precision MINPACK (sminpack/lmdif.f with fdjac2.f, qrfac.f, lmpar.f,
// qrsolv.f, enorm.f, spmpar.f).  This is that algorithm in fp32 with every expression rounded in the same order, so
// that the same residuals give the same iterates, except for ONE structural change: the forward-difference Jacobian
// asks for its n perturbed points in one call (`Fcn` takes a batch), which the engine evaluates as one device launch
// instead of n consecutive forward steps.  Built with -ffp-contract=off like the rest of the host side.
#pragma once
#include <cmath>
#include <functional>
#include <vector>

namespace kiwi {
namespace lm {

constexpr float kEpsMch = 1.192091E-07f;       // spmpar(1) as sminpack/spmpar.f has it (not 2^-23)
constexpr float kDwarf = 1.175495E-38f;        // spmpar(2)

// k points xs[k][n] (may be modified in place: the engine clamps to the parameter limits) -> fv[k][m]; < 0 aborts
using Fcn = std::function<int(int k, float *xs, float *fv)>;

inline float sq(float v) { return v * v; }

// sminpack/enorm.f: scaled sums for large / intermediate / small components
inline float enorm(int n, const float *x)
{
    const float rdwarf = 3.834e-20f, rgiant = 1.304e19f;
    float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, x1max = 0.0f, x3max = 0.0f;
    const float agiant = rgiant / (float)n;
    for (int i = 0; i < n; i++) {
        const float xabs = fabsf(x[i]);
        if (xabs > rdwarf && xabs < agiant) {
            s2 = s2 + sq(xabs);
        } else if (xabs <= rdwarf) {
            if (xabs > x3max) {
                s3 = 1.0f + s3 * sq(x3max / xabs);
                x3max = xabs;
            } else if (xabs != 0.0f) {
                s3 = s3 + sq(xabs / x3max);
            }
        } else {
            if (xabs > x1max) {
                s1 = 1.0f + s1 * sq(x1max / xabs);
                x1max = xabs;
            } else {
                s1 = s1 + sq(xabs / x1max);
            }
        }
    }
    if (s1 != 0.0f) return x1max * sqrtf(s1 + (s2 / x1max) / x1max);
    if (s2 != 0.0f) {
        if (s2 >= x3max) return sqrtf(s2 * (1.0f + (x3max / s2) * (x3max * s3)));
        return sqrtf(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * sqrtf(s3);
}

// column-major m x n matrix as MINPACK addresses it
struct Mat {
    float *p;
    int ld;
    float &operator()(int i, int j) const { return p[(size_t)j * ld + i]; }
};

// sminpack/qrfac.f with column pivoting: Householder vectors below the diagonal of a, R above it, diag(R) in rdiag
inline void qrfac(int m, int n, Mat a, int *ipvt, float *rdiag, float *acnorm, float *wa)
{
    for (int j = 0; j < n; j++) {
        acnorm[j] = enorm(m, &a(0, j));
        rdiag[j] = acnorm[j];
        wa[j] = rdiag[j];
        ipvt[j] = j;
    }
    const int minmn = m < n ? m : n;
    for (int j = 0; j < minmn; j++) {
        int kmax = j;
        for (int k = j; k < n; k++)
            if (rdiag[k] > rdiag[kmax]) kmax = k;
        if (kmax != j) {
            for (int i = 0; i < m; i++) {
                const float t = a(i, j);
                a(i, j) = a(i, kmax);
                a(i, kmax) = t;
            }
            rdiag[kmax] = rdiag[j];
            wa[kmax] = wa[j];
            const int k = ipvt[j];
            ipvt[j] = ipvt[kmax];
            ipvt[kmax] = k;
        }
        float ajnorm = enorm(m - j, &a(j, j));
        if (ajnorm != 0.0f) {
            if (a(j, j) < 0.0f) ajnorm = -ajnorm;
            for (int i = j; i < m; i++) a(i, j) = a(i, j) / ajnorm;
            a(j, j) = a(j, j) + 1.0f;
            for (int k = j + 1; k < n; k++) {
                float sum = 0.0f;
                for (int i = j; i < m; i++) sum = sum + a(i, j) * a(i, k);
                float temp = sum / a(j, j);
                for (int i = j; i < m; i++) a(i, k) = a(i, k) - temp * a(i, j);
                if (rdiag[k] != 0.0f) {
                    temp = a(j, k) / rdiag[k];
                    rdiag[k] = rdiag[k] * sqrtf(fmaxf(0.0f, 1.0f - sq(temp)));
                    if (!(0.05f * sq(rdiag[k] / wa[k]) > kEpsMch)) {
                        rdiag[k] = enorm(m - j - 1, j + 1 < m ? &a(j + 1, k) : nullptr);
                        wa[k] = rdiag[k];
                    }
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
}

// sminpack/qrsolv.f: least squares of [R; D] by Givens rotations; the strict lower triangle of r holds S^T afterwards
inline void qrsolv(int n, Mat r, const int *ipvt, const float *diag, const float *qtb, float *x, float *sdiag, float *wa)
{
    for (int j = 0; j < n; j++) {
        for (int i = j; i < n; i++) r(i, j) = r(j, i);
        x[j] = r(j, j);
        wa[j] = qtb[j];
    }
    for (int j = 0; j < n; j++) {
        const int l = ipvt[j];
        if (diag[l] != 0.0f) {
            for (int k = j; k < n; k++) sdiag[k] = 0.0f;
            sdiag[j] = diag[l];
            float qtbpj = 0.0f;
            for (int k = j; k < n; k++) {
                if (sdiag[k] == 0.0f) continue;
                float cs, sn;
                if (fabsf(r(k, k)) < fabsf(sdiag[k])) {
                    const float cotan = r(k, k) / sdiag[k];
                    sn = 0.5f / sqrtf(0.25f + 0.25f * sq(cotan));
                    cs = sn * cotan;
                } else {
                    const float tn = sdiag[k] / r(k, k);
                    cs = 0.5f / sqrtf(0.25f + 0.25f * sq(tn));
                    sn = cs * tn;
                }
                r(k, k) = cs * r(k, k) + sn * sdiag[k];
                const float temp = cs * wa[k] + sn * qtbpj;
                qtbpj = -sn * wa[k] + cs * qtbpj;
                wa[k] = temp;
                for (int i = k + 1; i < n; i++) {
                    const float t = cs * r(i, k) + sn * sdiag[i];
                    sdiag[i] = -sn * r(i, k) + cs * sdiag[i];
                    r(i, k) = t;
                }
            }
        }
        sdiag[j] = r(j, j);
        r(j, j) = x[j];
    }
    int nsing = n;
    for (int j = 0; j < n; j++) {
        if (sdiag[j] == 0.0f && nsing == n) nsing = j;
        if (nsing < n) wa[j] = 0.0f;
    }
    for (int j = nsing - 1; j >= 0; j--) {
        float sum = 0.0f;
        for (int i = j + 1; i < nsing; i++) sum = sum + r(i, j) * wa[i];
        wa[j] = (wa[j] - sum) / sdiag[j];
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa[j];
}

// sminpack/lmpar.f: the Levenberg-Marquardt parameter for the trust radius delta
inline void lmpar(int n, Mat r, const int *ipvt, const float *diag, const float *qtb, float delta, float &par, float *x,
                  float *sdiag, float *wa1, float *wa2)
{
    int nsing = n;
    for (int j = 0; j < n; j++) {
        wa1[j] = qtb[j];
        if (r(j, j) == 0.0f && nsing == n) nsing = j;
        if (nsing < n) wa1[j] = 0.0f;
    }
    for (int j = nsing - 1; j >= 0; j--) {
        wa1[j] = wa1[j] / r(j, j);
        const float temp = wa1[j];
        for (int i = 0; i < j; i++) wa1[i] = wa1[i] - r(i, j) * temp;
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa1[j];
    int iter = 0;
    for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
    float dxnorm = enorm(n, wa2);
    float fp = dxnorm - delta;
    if (fp <= 0.1f * delta) {
        par = 0.0f;                                  // iter == 0
        return;
    }
    float parl = 0.0f;
    if (nsing >= n) {
        for (int j = 0; j < n; j++) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < n; j++) {
            float sum = 0.0f;
            for (int i = 0; i < j; i++) sum = sum + r(i, j) * wa1[i];
            wa1[j] = (wa1[j] - sum) / r(j, j);
        }
        const float temp = enorm(n, wa1);
        parl = ((fp / delta) / temp) / temp;
    }
    for (int j = 0; j < n; j++) {
        float sum = 0.0f;
        for (int i = 0; i <= j; i++) sum = sum + r(i, j) * qtb[i];
        wa1[j] = sum / diag[ipvt[j]];
    }
    const float gnorm = enorm(n, wa1);
    float paru = gnorm / delta;
    if (paru == 0.0f) paru = kDwarf / fminf(delta, 0.1f);
    par = fmaxf(par, parl);
    par = fminf(par, paru);
    if (par == 0.0f) par = gnorm / dxnorm;
    for (;;) {
        iter++;
        if (par == 0.0f) par = fmaxf(kDwarf, 0.001f * paru);
        float temp = sqrtf(par);
        for (int j = 0; j < n; j++) wa1[j] = temp * diag[j];
        qrsolv(n, r, ipvt, wa1, qtb, x, sdiag, wa2);
        for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
        dxnorm = enorm(n, wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (fabsf(fp) <= 0.1f * delta || (parl == 0.0f && fp <= temp && temp < 0.0f) || iter == 10) break;
        for (int j = 0; j < n; j++) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < n; j++) {
            wa1[j] = wa1[j] / sdiag[j];
            const float t = wa1[j];
            for (int i = j + 1; i < n; i++) wa1[i] = wa1[i] - r(i, j) * t;
        }
        temp = enorm(n, wa1);
        const float parc = ((fp / delta) / temp) / temp;
        if (fp > 0.0f) parl = fmaxf(parl, par);
        if (fp < 0.0f) paru = fminf(paru, par);
        par = fmaxf(parl, par + parc);
    }
}

// sminpack/lmdif.f (nprint = 0) with fdjac2.f's forward differences requested as one batch.  diag is used as given
// when mode == 2.  Returns info (negative: the value fcn aborted with); nfev counts single forward evaluations.
inline int lmdif(const Fcn &fcn, int m, int n, float *x, float *fvec, float ftol, float xtol, float gtol, int maxfev,
                 float epsfcn, float *diag, int mode, float factor, int &nfev)
{
    int info = 0;
    nfev = 0;
    if (n <= 0 || m < n || ftol < 0.0f || xtol < 0.0f || gtol < 0.0f || maxfev <= 0 || factor <= 0.0f) return 0;
    if (mode == 2)
        for (int j = 0; j < n; j++)
            if (diag[j] <= 0.0f) return 0;
    std::vector<float> fjac_((size_t)m * n), qtf(n), wa1(n), wa2(n), wa3(n), wa4(m), xs((size_t)n * n), fs((size_t)n * m), h(n);
    std::vector<int> ipvt(n);
    Mat fjac{fjac_.data(), m};

    int iflag = fcn(1, x, fvec);
    nfev = 1;
    if (iflag < 0) return iflag;
    float fnorm = enorm(m, fvec);
    float par = 0.0f, delta = 0.0f, xnorm = 0.0f, gnorm = 0.0f;
    int iter = 1;
    const float eps = sqrtf(fmaxf(epsfcn, kEpsMch));
    for (;;) {
        // fdjac2: column j from x with x_j + h_j; the n points go to the engine together
        for (int j = 0; j < n; j++) {
            for (int i = 0; i < n; i++) xs[(size_t)j * n + i] = x[i];
            h[j] = eps * fabsf(x[j]);
            if (h[j] == 0.0f) h[j] = eps;
            xs[(size_t)j * n + j] = x[j] + h[j];
        }
        iflag = fcn(n, xs.data(), fs.data());
        nfev += n;
        if (iflag < 0) return iflag;
        for (int j = 0; j < n; j++)
            for (int i = 0; i < m; i++) fjac(i, j) = (fs[(size_t)j * m + i] - fvec[i]) / h[j];

        qrfac(m, n, fjac, ipvt.data(), wa1.data(), wa2.data(), wa3.data());
        if (iter == 1) {
            if (mode != 2)
                for (int j = 0; j < n; j++) diag[j] = wa2[j] == 0.0f ? 1.0f : wa2[j];
            for (int j = 0; j < n; j++) wa3[j] = diag[j] * x[j];
            xnorm = enorm(n, wa3.data());
            delta = factor * xnorm;
            if (delta == 0.0f) delta = factor;
        }
        // (Q^T f) in qtf, R's diagonal restored into fjac
        for (int i = 0; i < m; i++) wa4[i] = fvec[i];
        for (int j = 0; j < n; j++) {
            if (fjac(j, j) != 0.0f) {
                float sum = 0.0f;
                for (int i = j; i < m; i++) sum = sum + fjac(i, j) * wa4[i];
                const float temp = -sum / fjac(j, j);
                for (int i = j; i < m; i++) wa4[i] = wa4[i] + fjac(i, j) * temp;
            }
            fjac(j, j) = wa1[j];
            qtf[j] = wa4[j];
        }
        gnorm = 0.0f;
        if (fnorm != 0.0f)
            for (int j = 0; j < n; j++) {
                const int l = ipvt[j];
                if (wa2[l] == 0.0f) continue;
                float sum = 0.0f;
                for (int i = 0; i <= j; i++) sum = sum + fjac(i, j) * (qtf[i] / fnorm);
                gnorm = fmaxf(gnorm, fabsf(sum / wa2[l]));
            }
        if (gnorm <= gtol) return 4;
        if (mode != 2)
            for (int j = 0; j < n; j++) diag[j] = fmaxf(diag[j], wa2[j]);

        float ratio;
        do {                                        // inner loop: until a step is accepted
            lmpar(n, fjac, ipvt.data(), diag, qtf.data(), delta, par, wa1.data(), wa2.data(), wa3.data(), wa4.data());
            for (int j = 0; j < n; j++) {
                wa1[j] = -wa1[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            const float pnorm = enorm(n, wa3.data());
            if (iter == 1) delta = fminf(delta, pnorm);
            iflag = fcn(1, wa2.data(), wa4.data());
            nfev++;
            if (iflag < 0) return iflag;
            const float fnorm1 = enorm(m, wa4.data());
            float actred = -1.0f;
            if (0.1f * fnorm1 < fnorm) actred = 1.0f - sq(fnorm1 / fnorm);
            for (int j = 0; j < n; j++) {
                wa3[j] = 0.0f;
                const float temp = wa1[ipvt[j]];
                for (int i = 0; i <= j; i++) wa3[i] = wa3[i] + fjac(i, j) * temp;
            }
            const float temp1 = enorm(n, wa3.data()) / fnorm;
            const float temp2 = (sqrtf(par) * pnorm) / fnorm;
            const float prered = sq(temp1) + sq(temp2) / 0.5f;
            const float dirder = -(sq(temp1) + sq(temp2));
            ratio = 0.0f;
            if (prered != 0.0f) ratio = actred / prered;
            if (ratio <= 0.25f) {
                float temp = 0.0f;
                if (actred >= 0.0f) temp = 0.5f;
                if (actred < 0.0f) temp = 0.5f * dirder / (dirder + 0.5f * actred);
                if (0.1f * fnorm1 >= fnorm || temp < 0.1f) temp = 0.1f;
                delta = temp * fminf(delta, pnorm / 0.1f);
                par = par / temp;
            } else if (par == 0.0f || ratio >= 0.75f) {
                delta = pnorm / 0.5f;
                par = 0.5f * par;
            }
            if (ratio >= 0.0001f) {
                for (int j = 0; j < n; j++) {
                    x[j] = wa2[j];
                    wa2[j] = diag[j] * x[j];
                }
                for (int i = 0; i < m; i++) fvec[i] = wa4[i];
                xnorm = enorm(n, wa2.data());
                fnorm = fnorm1;
                iter++;
            }
            const bool small = fabsf(actred) <= ftol && prered <= ftol && 0.5f * ratio <= 1.0f;
            if (small) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (small && info == 2) info = 3;
            if (info != 0) return info;
            if (nfev >= maxfev) info = 5;
            if (fabsf(actred) <= kEpsMch && prered <= kEpsMch && 0.5f * ratio <= 1.0f) info = 6;
            if (delta <= kEpsMch * xnorm) info = 7;
            if (gnorm <= kEpsMch) info = 8;
            if (info != 0) return info;
        } while (ratio < 0.0001f);
    }
}

}  // namespace lm
}  // namespace kiwi
