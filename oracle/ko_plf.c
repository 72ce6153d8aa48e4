/*
 * ko_plf.c -- CPU ORACLE (test infrastructure, not product code; see ko.h).
 * Restates piecewise_linear_function.f90 (STF binning, tapers, filters) and
 * comparator.f90:1157-1169 discrete_plf_span.
 */
#include "ko.h"
#include <math.h>

static const float PI_F = 3.14159265358979f;   /* constants.f90:21 */

/* piecewise_linear_function.f90:302-306 */
static float ip_linear(float x0, float y0, float x1, float y1, float xi)
{
    return y0 + (y1 - y0) / (x1 - x0) * (xi - x0);
}
/* :308-316 */
static float ip_cos(float x0, float y0, float x1, float y1, float xi)
{
    if (y1 != y0) return y0 + (y1 - y0) * (0.5f - 0.5f * cosf((xi - x0) / (x1 - x0) * PI_F));
    return y0;
}
/* :318-327 */
static float ip_zero_one(float x0, float y0, float x1, float y1, float xi)
{
    if (y0 == 0.f && y1 == 0.f) return 0.f + 0.f * (x0 + x1 + xi);
    return 1.f;
}
static float ip_eval(int ip, float x0, float y0, float x1, float y1, float xi)
{
    if (ip == 0) return ip_cos(x0, y0, x1, y1, xi);
    if (ip == 1) return ip_linear(x0, y0, x1, y1, xi);
    return ip_zero_one(x0, y0, x1, y1, xi);
}

/* :296-300 */
static float trapezoid_area(float x0, float y0, float x1, float y1)
{
    return (y0 + y1) * (x1 - x0) / 2.f;
}
/* :285-294 */
static float trapezoid_centroid(float x0, float y0, float x1, float y1)
{
    if (y0 + y1 == 0.f) return (x0 + x1) / 2.f;
    return (x0 * (2.f * y0 + y1) + x1 * (y0 + 2.f * y1)) / (3.f * (y0 + y1));
}

/* :133-163 */
float ko_plf_integrate(const ko_plf *s, float a, float b)
{
    float area = 0.f;
    if (s->n == 0) return area;
    if (b <= s->x[0]) return area;
    if (a >= s->x[s->n - 1]) return area;
    for (int i = 0; i < s->n - 1; i++) {
        if (a >= s->x[i + 1]) continue;
        if (b <= s->x[i]) return area;
        float x0 = fmaxf(a, s->x[i]);
        float x1 = fminf(b, s->x[i + 1]);
        float y0 = s->y[i];
        if (x0 != s->x[i]) y0 = ip_linear(s->x[i], s->y[i], s->x[i + 1], s->y[i + 1], a);
        float y1 = s->y[i + 1];
        if (x1 != s->x[i + 1]) y1 = ip_linear(s->x[i], s->y[i], s->x[i + 1], s->y[i + 1], b);
        area = area + trapezoid_area(x0, y0, x1, y1);
    }
    return area;
}

/* :165-193 */
void ko_plf_integrate_and_centroid(const ko_plf *s, float a, float b, float *area_, float *centroid)
{
    float area = 0.f, c = 0.f;
    *area_ = 0.f;
    *centroid = (a + b) / 2.f;
    if (s->n == 0) return;
    if (b <= s->x[0]) return;
    if (a >= s->x[s->n - 1]) return;
    for (int i = 0; i < s->n - 1; i++) {
        if (a >= s->x[i + 1]) continue;
        if (b <= s->x[i]) break;
        float x0 = fmaxf(a, s->x[i]);
        float x1 = fminf(b, s->x[i + 1]);
        float y0 = s->y[i];
        if (x0 != s->x[i]) y0 = ip_linear(s->x[i], s->y[i], s->x[i + 1], s->y[i + 1], a);
        float y1 = s->y[i + 1];
        if (x1 != s->x[i + 1]) y1 = ip_linear(s->x[i], s->y[i], s->x[i + 1], s->y[i + 1], b);
        float areathis = trapezoid_area(x0, y0, x1, y1);
        c = c + areathis * trapezoid_centroid(x0, y0, x1, y1);
        area = area + areathis;
    }
    *area_ = area;
    *centroid = c / area;
}

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* :195-237; array index lo..hi; sample j sits at abscissa j*dx (:225) */
void ko_plf_taper_array_r(const ko_plf *s, float *array, int lo, int hi, float dx, int ip)
{
#define A(j) array[(j) - lo]
    int ibeg = (int)floorf(s->x[0] / dx);
    if (lo <= ibeg) for (int j = lo; j <= imin(ibeg, hi); j++) A(j) = 0.f;
    int ibegatleast = lo;
    for (int i = 0; i < s->n - 1; i++) {
        ibeg = imax(imax((int)floorf(s->x[i] / dx) + 1, lo), ibegatleast);
        int iend = imin((int)floorf(s->x[i + 1] / dx), hi);
        if (ibeg <= iend)
            for (int j = ibeg; j <= iend; j++)
                A(j) = A(j) * ip_eval(ip, s->x[i], s->y[i], s->x[i + 1], s->y[i + 1], (float)j * dx);
        ibegatleast = iend + 1;
    }
    int iend = (int)floorf(s->x[s->n - 1] / dx) + 1;
    if (hi >= iend) for (int j = imax(iend, lo); j <= hi; j++) A(j) = 0.f;
#undef A
}

/* :239-282 complex variant (re,im interleaved); complex*real scales both parts */
void ko_plf_taper_array_c(const ko_plf *s, float *array, int lo, int hi, float dx, int ip)
{
#define RE(j) array[2 * ((j) - lo)]
#define IM(j) array[2 * ((j) - lo) + 1]
    int ibeg = (int)floorf(s->x[0] / dx);
    if (lo <= ibeg) for (int j = lo; j <= imin(ibeg, hi); j++) { RE(j) = 0.f; IM(j) = 0.f; }
    int ibegatleast = lo;
    for (int i = 0; i < s->n - 1; i++) {
        ibeg = imax(imax((int)floorf(s->x[i] / dx) + 1, lo), ibegatleast);
        int iend = imin((int)floorf(s->x[i + 1] / dx), hi);
        if (ibeg <= iend)
            for (int j = ibeg; j <= iend; j++) {
                float w = ip_eval(ip, s->x[i], s->y[i], s->x[i + 1], s->y[i + 1], (float)j * dx);
                RE(j) = RE(j) * w; IM(j) = IM(j) * w;
            }
        ibegatleast = iend + 1;
    }
    int iend = (int)floorf(s->x[s->n - 1] / dx) + 1;
    if (hi >= iend) for (int j = imax(iend, lo); j <= hi; j++) { RE(j) = 0.f; IM(j) = 0.f; }
#undef RE
#undef IM
}

/* comparator.f90:1157-1169 with plf_span (piecewise_linear_function.f90:120-131) */
void ko_discrete_plf_span(const ko_plf *s, float dt, int out[2])
{
    float r0 = 0.f, r1 = -1.f;
    if (s->n > 0) { r0 = s->x[0]; r1 = s->x[s->n - 1]; }
    out[0] = (int)ceilf(r0 / dt);
    out[1] = (int)floorf(r1 / dt);
}
