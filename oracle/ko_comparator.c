/*
 * ko_comparator.c -- CPU ORACLE (test infrastructure, not product code; see ko.h).
 * Restates comparator.f90: t_probe (padded power-of-two window, zeros to the
 * left of the data, last value repeated to the right), tapering, the time
 * domain norms with fp64 accumulation and the spectral norms.
 *
 * comparator.f90 itself cannot be compiled here (include 'fftw3.f', libfftw3f
 * absent), so this file is pinned by the reference's test_comparator.f90 KATs
 * (tests/test_oracle_kats.py), not by a reference build.  The r2c/c2r
 * transforms are FFTW's in the reference (unnormalised, comparator.f90:1201-
 * 1209,1244-1249); here they are a double-precision radix-2 FFT rounded to
 * fp32 ("exact DFT"), so ampspec norms are pinned to ~1e-6 relative only.
 */
#include "ko.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int slen(const int s[2]) { return s[1] - s[0] + 1; }
static void span_union(const int a[2], const int b[2], int c[2]) { c[0] = imin(a[0], b[0]); c[1] = imax(a[1], b[1]); }
static void span_isect(const int a[2], const int b[2], int c[2]) { c[0] = imax(a[0], b[0]); c[1] = imin(a[1], b[1]); }

/* comparator.f90:1111-1118: 2**ceiling(log(real(n))/log(2.)).  The fp32 log
 * quotient is libm dependent at exact powers of two (SURVEY hard part 4); the
 * flang build of the reference returns n itself for n = 2^1..2^16, which is
 * what integer arithmetic gives, so integer arithmetic is used. */
int ko_next_power_of_two(int n)
{
    int m = 1;
    while (m < n) m *= 2;
    return m;
}

/* comparator.f90:1092-1109 */
void ko_allowed_span(const int span[2], int minlength, int out[2])
{
    int length = slen(span);
    if (length < minlength) length = minlength;
    int lengthp = ko_next_power_of_two(length);
    out[0] = span[0] - (int)floorf((float)(lengthp - slen(span)) / 2.f);
    out[1] = out[0] + lengthp - 1;
}

/* comparator.f90:1120-1129 */
static int containing(const int outer[2], const int inner[2])
{
    int c[2]; span_isect(outer, inner, c);
    return c[0] == inner[0] && c[1] == inner[1];
}

static void dirtyfy_array(ko_probe *p)   /* :1309-1337 chain */
{
    p->array_tapered_dirty = 1; p->spectrum_dirty = 1;
    p->spectrum_filtered_dirty = 1; p->array_filtered_dirty = 1;
}

/* comparator.f90:179-194 */
void ko_probe_init(ko_probe *p, float dt)
{
    memset(p, 0, sizeof(*p));
    p->dt = dt;
    p->span[0] = p->span[1] = 0;
    p->dataspan[0] = p->dataspan[1] = 0;
    dirtyfy_array(p);
    p->paddingfactor = 2.f;
    p->factor = 1.f;
}

void ko_probe_destroy(ko_probe *p)
{
    free(p->array); free(p->array_tapered); free(p->spectrum); free(p->spectrum_filtered);
    free(p->amp_spectrum); free(p->amp_spectrum_filtered); free(p->array_filtered);
    float dt = p->dt;
    ko_probe_init(p, dt);
}

static void probe_resize(ko_probe *p, const int newspan[2])
{
    size_t n = (size_t)slen(newspan);
    free(p->array); free(p->array_tapered);
    p->array = (float *)malloc(sizeof(float) * n);
    p->array_tapered = (float *)malloc(sizeof(float) * n);
}
#define ARR(p, i) (p)->array[(i) - (p)->span[0]]

/* comparator.f90:222-271 (allow_shrink_ = .false., no span_hint) */
void ko_probe_set_array(ko_probe *p, const ko_strip *strip, float factor)
{
    int sspan[2] = { strip->lo, strip->lo + strip->n - 1 };
    int newspan[2];
    p->dataspan[0] = sspan[0]; p->dataspan[1] = sspan[1];
    if (!p->array) { newspan[0] = sspan[0]; newspan[1] = sspan[1]; }
    else span_union(sspan, p->span, newspan);
    int datalength = slen(p->dataspan);
    int tmp[2] = { newspan[0], newspan[1] };
    ko_allowed_span(tmp, (int)ceilf((float)datalength * p->paddingfactor), newspan);
    if (!p->array || newspan[0] != p->span[0] || newspan[1] != p->span[1]) probe_resize(p, newspan);
    p->span[0] = newspan[0]; p->span[1] = newspan[1];
    if (p->span[0] <= p->dataspan[0] - 1)
        for (int i = p->span[0]; i <= p->dataspan[0] - 1; i++) ARR(p, i) = 0.f;
    for (int i = p->dataspan[0]; i <= p->dataspan[1]; i++) ARR(p, i) = strip->d[i - strip->lo] * factor;
    if (p->dataspan[1] + 1 <= p->span[1])
        for (int i = p->dataspan[1] + 1; i <= p->span[1]; i++) ARR(p, i) = ARR(p, p->dataspan[1]);
    dirtyfy_array(p);
}

/* comparator.f90:273-288 */
void ko_probe_shift(ko_probe *p, int ishift)
{
    if (!p->array) return;
    ko_strip s = { NULL, 1, 0 };
    ko_strip_init(&s, p->dataspan[0] + ishift, p->dataspan[1] + ishift, &ARR(p, p->dataspan[0]));
    ko_probe_set_array(p, &s, 1.f);
    ko_strip_destroy(&s);
}

/* comparator.f90:291-330 */
static void probe_extend_span(ko_probe *p, const int span[2])
{
    int newspan[2];
    if (!p->array) {   /* :302-308 (allocates exactly 'span', leaves self%span untouched) */
        size_t n = (size_t)slen(span);
        p->array = (float *)calloc(n, sizeof(float));
        p->array_tapered = (float *)malloc(sizeof(float) * n);
        /* the reference does not update self%span here; set it so indexing stays valid */
        p->span[0] = span[0]; p->span[1] = span[1];
        return;
    }
    int u[2]; span_union(span, p->dataspan, u);
    ko_allowed_span(u, 0, newspan);
    if (p->span[0] == newspan[0] && p->span[1] == newspan[1]) return;
    int nd = slen(p->dataspan);
    float *temp = (float *)malloc(sizeof(float) * (size_t)nd);
    memcpy(temp, &ARR(p, p->dataspan[0]), sizeof(float) * (size_t)nd);
    probe_resize(p, newspan);
    p->span[0] = newspan[0]; p->span[1] = newspan[1];
    if (p->span[0] <= p->dataspan[0] - 1)
        for (int i = p->span[0]; i <= p->dataspan[0] - 1; i++) ARR(p, i) = 0.f;
    for (int i = 0; i < nd; i++) ARR(p, p->dataspan[0] + i) = temp[i];
    if (p->dataspan[1] + 1 <= p->span[1])
        for (int i = p->dataspan[1] + 1; i <= p->span[1]; i++) ARR(p, i) = ARR(p, p->dataspan[1]);
    free(temp);
    dirtyfy_array(p);
}

/* comparator.f90:435-453 */
void ko_probe_set_taper(ko_probe *p, const ko_plf *plf)
{
    p->taper = *plf;
    p->array_tapered_dirty = 1; p->spectrum_dirty = 1; p->spectrum_filtered_dirty = 1; p->array_filtered_dirty = 1;
}
void ko_probe_set_filter(ko_probe *p, const ko_plf *plf)
{
    p->filter = *plf;
    p->spectrum_filtered_dirty = 1; p->array_filtered_dirty = 1;
}

/* comparator.f90:464-486 */
static void probes_adjust_spans(ko_probe *a, ko_probe *b)
{
    int newspan[2], u[2];
    span_union(a->dataspan, b->dataspan, u);
    int minlength = imax((int)ceilf((float)slen(a->dataspan) * a->paddingfactor),
                         (int)ceilf((float)slen(b->dataspan) * b->paddingfactor));
    ko_allowed_span(u, minlength, newspan);
    if (a->span[0] == b->span[0] && a->span[1] == b->span[1] && slen(a->span) == slen(newspan) &&
        containing(a->span, b->dataspan) && containing(b->span, a->dataspan)) return;
    probe_extend_span(a, newspan);
    probe_extend_span(b, newspan);
}

/* comparator.f90:1173-1184 */
static void make_array_tapered(ko_probe *p)
{
    if (p->taper.n > 0) {
        memcpy(p->array_tapered, p->array, sizeof(float) * (size_t)slen(p->span));
        ko_plf_taper_array_r(&p->taper, p->array_tapered + (p->dataspan[0] - p->span[0]),
                             p->dataspan[0], p->span[1], p->dt, 0);
    }
}

/* in-place iterative radix-2 complex FFT in double; sign = -1 forward, +1 backward */
static void fft_c(double *re, double *im, int n, int sign)
{
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        double ang = sign * 2.0 * M_PI / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; k++) {
                double wr = cos(ang * k), wi = sin(ang * k);
                int u = i + k, v = i + k + len / 2;
                double xr = re[v] * wr - im[v] * wi, xi = re[v] * wi + im[v] * wr;
                re[v] = re[u] - xr; im[v] = im[u] - xi;
                re[u] += xr; im[u] += xi;
            }
    }
}

/* The same radix-2 algorithm with every operation in fp32 (twiddles: cos / sin in double, rounded once): a textbook fp32 FFT,
 * the SECOND checker of the spectral norms (tests/test_oracle_fft32.py): what it changes in a misfit is the round-off any
 * fp32 transform of that length brings -- the device's in-LDS radix-4 and hipFFT are such transforms, FFTW's single-precision
 * library in the reference is another --, and tests/common.py fft_roundoff_bound has to hold it.  ko_set_fft_precision(32). */
static int g_fft_bits = 64;
void ko_set_fft_precision(int bits) { g_fft_bits = (bits == 32) ? 32 : 64; }
int ko_get_fft_precision(void) { return g_fft_bits; }

static void fft_c32(float *re, float *im, int n, int sign)
{
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { float t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        double ang = sign * 2.0 * M_PI / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; k++) {
                const float wr = (float)cos(ang * k), wi = (float)sin(ang * k);
                int u = i + k, v = i + k + len / 2;
                const float a = re[v] * wr, b = im[v] * wi, c = re[v] * wi, d = im[v] * wr;
                const float xr = a - b, xi = c + d;
                re[v] = re[u] - xr; im[v] = im[u] - xi;
                re[u] = re[u] + xr; im[u] = im[u] + xi;
            }
    }
}

/* transform of n doubles held in re / im at the precision in force */
static void fft_any(double *re, double *im, int n, int sign)
{
    if (g_fft_bits == 64) { fft_c(re, im, n, sign); return; }
    float *fr = (float *)malloc(sizeof(float) * 2 * (size_t)n), *fi = fr + n;
    for (int i = 0; i < n; i++) { fr[i] = (float)re[i]; fi[i] = (float)im[i]; }
    fft_c32(fr, fi, n, sign);
    for (int i = 0; i < n; i++) { re[i] = fr[i]; im[i] = fi[i]; }
    free(fr);
}

/* comparator.f90:1186-1216 */
static void make_spectrum(ko_probe *p)
{
    int ntrans = slen(p->span);
    int ns = ntrans / 2 + 1;
    if (p->nspec != ns) {
        free(p->spectrum); free(p->spectrum_filtered); free(p->amp_spectrum); free(p->amp_spectrum_filtered);
        p->spectrum = (float *)malloc(sizeof(float) * 2 * (size_t)ns);
        p->spectrum_filtered = (float *)malloc(sizeof(float) * 2 * (size_t)ns);
        p->amp_spectrum = (float *)malloc(sizeof(float) * (size_t)ns);
        p->amp_spectrum_filtered = (float *)malloc(sizeof(float) * (size_t)ns);
        p->nspec = ns;
    }
    const float *src = (p->taper.n > 0) ? p->array_tapered : p->array;
    double *re = (double *)malloc(sizeof(double) * 2 * (size_t)ntrans), *im = re + ntrans;
    for (int i = 0; i < ntrans; i++) { re[i] = src[i]; im[i] = 0.0; }
    fft_any(re, im, ntrans, -1);
    for (int k = 0; k < ns; k++) {
        p->spectrum[2 * k] = (float)re[k]; p->spectrum[2 * k + 1] = (float)im[k];
        p->amp_spectrum[k] = hypotf(p->spectrum[2 * k], p->spectrum[2 * k + 1]);
    }
    free(re);
    p->df = 1.f / ((float)ntrans * p->dt);
}

/* comparator.f90:1218-1231: filter abscissa of bin j (0-based) is j*df */
static void make_spectrum_filtered(ko_probe *p)
{
    if (p->filter.n > 0) {
        memcpy(p->amp_spectrum_filtered, p->amp_spectrum, sizeof(float) * (size_t)p->nspec);
        memcpy(p->spectrum_filtered, p->spectrum, sizeof(float) * 2 * (size_t)p->nspec);
        ko_plf_taper_array_c(&p->filter, p->spectrum_filtered, 0, p->nspec - 1, p->df, 0);
        ko_plf_taper_array_r(&p->filter, p->amp_spectrum_filtered, 0, p->nspec - 1, p->df, 0);
    }
}

/* comparator.f90:1233-1263 */
static void make_array_filtered(ko_probe *p)
{
    if (p->filter.n == 0) return;
    int ntrans = slen(p->span);
    free(p->array_filtered);
    p->array_filtered = (float *)malloc(sizeof(float) * (size_t)ntrans);
    double *re = (double *)malloc(sizeof(double) * 2 * (size_t)ntrans), *im = re + ntrans;
    for (int k = 0; k < p->nspec; k++) { re[k] = p->spectrum_filtered[2 * k]; im[k] = p->spectrum_filtered[2 * k + 1]; }
    im[0] = 0.0; im[ntrans / 2] = 0.0;                 /* c2r ignores them */
    for (int k = 1; k < ntrans / 2; k++) { re[ntrans - k] = re[k]; im[ntrans - k] = -im[k]; }
    fft_any(re, im, ntrans, +1);
    for (int i = 0; i < ntrans; i++) p->array_filtered[i] = (float)re[i];
    free(re);
    for (int i = 0; i < ntrans; i++) p->array_filtered[i] = p->array_filtered[i] / (float)ntrans;
    if (p->taper.n > 0)
        ko_plf_taper_array_r(&p->taper, p->array_filtered, p->span[0], p->span[1], p->dt, 2);
}

/* comparator.f90:1267-1305 update chain */
static void update_array_tapered(ko_probe *p)
{
    if (p->array_tapered_dirty) make_array_tapered(p);
    p->array_tapered_dirty = 0;
}
static void update_spectrum(ko_probe *p)
{
    update_array_tapered(p);
    if (p->spectrum_dirty) make_spectrum(p);
    p->spectrum_dirty = 0;
}
static void update_spectrum_filtered(ko_probe *p)
{
    update_spectrum(p);
    if (p->spectrum_filtered_dirty) make_spectrum_filtered(p);
    p->spectrum_filtered_dirty = 0;
}
static void update_array_filtered(ko_probe *p)
{
    update_spectrum_filtered(p);
    if (p->array_filtered_dirty) make_array_filtered(p);
    p->array_filtered_dirty = 0;
}

/* ---- norm kernels, comparator.f90:619-707; fp64 accumulation ---- */
static float norm2_apply(int method, const float *a, const float *b, int n, float dt, float fa, float fb)
{
    double sum = 0.0;
    switch (method) {
    case KO_L2NORM: case KO_AMPSPEC_L2NORM:   /* l2norm_func :650-659 */
        if (fa == 1.f && fb == 1.f) for (int i = 0; i < n; i++) { double d = (double)(a[i] - b[i]); sum += d * d; }
        else for (int i = 0; i < n; i++) { double d = (double)(fa * a[i] - fb * b[i]); sum += d * d; }
        return (float)sqrt((double)dt * sum);
    case KO_L1NORM: case KO_AMPSPEC_L1NORM:   /* l1norm_func :639-648 */
        if (fa == 1.f && fb == 1.f) for (int i = 0; i < n; i++) sum += (double)fabsf(a[i] - b[i]);
        else for (int i = 0; i < n; i++) sum += (double)fabsf(fa * a[i] - fb * b[i]);
        return (float)((double)dt * sum);
    case KO_SCALAR_PRODUCT:                   /* scalar_product_2 :627-637 */
        if (fa == 1.f && fb == 1.f) for (int i = 0; i < n; i++) sum += (double)(a[i] * b[i]);
        else for (int i = 0; i < n; i++) sum += (double)(a[i] * fa * b[i] * fb);
        return (float)sum;
    case KO_PEAK: {                           /* maxabs_func :661-667 */
        double m = -HUGE_VAL;
        for (int i = 0; i < n; i++) {
            double x = (double)(fa * a[i]), y = (double)(fb * b[i]);
            double v = sqrt(x * x + y * y);
            if (v > m) m = v;
        }
        return (float)m; }
    }
    return 0.f;
}

static float norm1_apply(int method, const float *a, int n, float dt, float fa)
{
    double sum = 0.0;
    switch (method) {
    case KO_L2NORM: case KO_AMPSPEC_L2NORM:   /* l2norm_func_1 :684-689 */
        for (int i = 0; i < n; i++) { double d = (double)a[i]; sum += d * d; }
        return fa * (float)sqrt((double)dt * sum);
    case KO_L1NORM: case KO_AMPSPEC_L1NORM:   /* l1norm_func_1 :677-682 */
        for (int i = 0; i < n; i++) sum += (double)fabsf(a[i]);
        return fa * (float)((double)dt * sum);
    case KO_SCALAR_PRODUCT:                   /* scalar_product_1 :669-675 */
        for (int i = 0; i < n; i++) sum += (double)(a[i] * a[i]);
        return (fa * fa) * (float)sum;
    case KO_PEAK: {                           /* maxabs_func_1 :691-697 */
        float m = -HUGE_VALF;
        for (int i = 0; i < n; i++) if (fabsf(a[i]) > m) m = fabsf(a[i]);
        return fa * m; }
    }
    return 0.f;
}

/* comparator.f90:770-822 */
static float probes_norm_timedomain(ko_probe *a, ko_probe *b, int method)
{
    int span[2], at[2], bt[2], ps[2];
    probes_adjust_spans(a, b);
    int both_taper = a->taper.n > 0 && b->taper.n > 0;
    if (both_taper) {
        ko_discrete_plf_span(&a->taper, a->dt, ps); span_isect(ps, a->span, at);
        ko_discrete_plf_span(&b->taper, b->dt, ps); span_isect(ps, b->span, bt);
        if (at[0] > at[1]) { span[0] = bt[0]; span[1] = bt[1]; }
        else if (bt[0] > bt[1]) { span[0] = at[0]; span[1] = at[1]; }
        else span_union(at, bt, span);
    } else {
        probes_adjust_spans(a, b);
        span_union(a->dataspan, b->dataspan, span);
    }
    if (span[0] > span[1]) return 0.f;   /* 'applying timedomain norm to empty region' */
    int n = slen(span);
    if (a->filter.n > 0 && b->filter.n > 0) {
        update_array_filtered(a); update_array_filtered(b);
        return norm2_apply(method, a->array_filtered + (span[0] - a->span[0]),
                           b->array_filtered + (span[0] - b->span[0]), n, a->dt, a->factor, b->factor);
    } else if (both_taper) {
        update_array_tapered(a); update_array_tapered(b);
        return norm2_apply(method, a->array_tapered + (span[0] - a->span[0]),
                           b->array_tapered + (span[0] - b->span[0]), n, a->dt, a->factor, b->factor);
    }
    return norm2_apply(method, a->array + (span[0] - a->span[0]), b->array + (span[0] - b->span[0]),
                       n, a->dt, a->factor, b->factor);
}

/* comparator.f90:824-859 */
static float probe_norm_timedomain(ko_probe *a, int method)
{
    int span[2], ps[2];
    if (a->taper.n > 0) { ko_discrete_plf_span(&a->taper, a->dt, ps); span_isect(ps, a->span, span); }
    else { span[0] = a->dataspan[0]; span[1] = a->dataspan[1]; }
    int n = slen(span);
    if (n <= 0) return norm1_apply(method, a->array, 0, a->dt, a->factor);
    if (a->filter.n > 0) {
        update_array_filtered(a);
        return norm1_apply(method, a->array_filtered + (span[0] - a->span[0]), n, a->dt, a->factor);
    } else if (a->taper.n > 0) {
        update_array_tapered(a);
        return norm1_apply(method, a->array_tapered + (span[0] - a->span[0]), n, a->dt, a->factor);
    }
    return norm1_apply(method, a->array + (span[0] - a->span[0]), n, a->dt, a->factor);
}

/* comparator.f90:861-886 */
static float probes_norm_frequencydomain(ko_probe *a, ko_probe *b, int method)
{
    probes_adjust_spans(a, b);
    if (a->filter.n > 0 && b->filter.n > 0) {
        update_spectrum_filtered(a); update_spectrum_filtered(b);
        return norm2_apply(method, a->amp_spectrum_filtered, b->amp_spectrum_filtered, a->nspec, a->df, a->factor, b->factor);
    }
    update_spectrum(a); update_spectrum(b);
    return norm2_apply(method, a->amp_spectrum, b->amp_spectrum, a->nspec, a->df, a->factor, b->factor);
}

/* comparator.f90:888-909 */
static float probe_norm_frequencydomain(ko_probe *a, int method)
{
    if (a->filter.n > 0) {
        update_spectrum_filtered(a);
        return norm1_apply(method, a->amp_spectrum_filtered, a->nspec, a->df, a->factor);
    }
    update_spectrum(a);
    return norm1_apply(method, a->amp_spectrum, a->nspec, a->df, a->factor);
}

/* comparator.f90:911-952 */
float ko_probes_norm(ko_probe *a, ko_probe *b, int method)
{
    switch (method) {
    case KO_L2NORM: case KO_L1NORM: case KO_SCALAR_PRODUCT: case KO_PEAK:
        return probes_norm_timedomain(a, b, method);
    case KO_AMPSPEC_L2NORM: case KO_AMPSPEC_L1NORM:
        return probes_norm_frequencydomain(a, b, method);
    }
    fprintf(stderr, "ko_probes_norm: unknown norm method %d\n", method);
    abort();
}

/* comparator.f90:954-996 */
float ko_probe_norm(ko_probe *a, int method)
{
    switch (method) {
    case KO_L2NORM: case KO_L1NORM: case KO_SCALAR_PRODUCT: case KO_PEAK:
        return probe_norm_timedomain(a, method);
    case KO_AMPSPEC_L2NORM: case KO_AMPSPEC_L1NORM:
        return probe_norm_frequencydomain(a, method);
    }
    fprintf(stderr, "ko_probe_norm: unknown norm method %d\n", method);
    abort();
}

/* comparator.f90:1061-1090 */
void ko_probes_windowed_cross_corr(ko_probe *a, ko_probe *b, int shift_lo, int shift_hi, float *cc)
{
    int ishift = shift_lo;
    for (int i = 0; i < shift_hi - shift_lo + 1; i++) {
        ko_probe_shift(b, ishift);
        ishift = 1;
        cc[i] = probes_norm_timedomain(a, b, KO_SCALAR_PRODUCT);
    }
    ko_probe_shift(b, -shift_hi);
}

/* probes_adjust_spans_3, comparator.f90:488-516 */
static void probes_adjust_spans_3(ko_probe *a, ko_probe *b, ko_probe *c)
{
    int t[2], u[2], newspan[2];
    span_union(a->dataspan, b->dataspan, t);
    span_union(t, c->dataspan, u);
    int minlength = imax(imax((int)ceilf((float)slen(a->dataspan) * a->paddingfactor),
                              (int)ceilf((float)slen(b->dataspan) * b->paddingfactor)),
                         (int)ceilf((float)slen(c->dataspan) * c->paddingfactor));
    ko_allowed_span(u, minlength, newspan);
    if (a->span[0] == b->span[0] && a->span[1] == b->span[1] && a->span[0] == c->span[0] && a->span[1] == c->span[1] &&
        slen(a->span) == slen(newspan) &&
        containing(a->span, b->dataspan) && containing(b->span, a->dataspan) &&
        containing(a->span, c->dataspan) && containing(c->span, a->dataspan) &&
        containing(b->span, c->dataspan) && containing(c->span, b->dataspan)) return;
    probe_extend_span(a, newspan);
    probe_extend_span(b, newspan);
    probe_extend_span(c, newspan);
}

/* max_vecnorm_d1_{1,2,3} (kind 1), max_vecnorm_d2_{1,2,3} (kind 2), arias_intensity_{1,2,3} (kind 3), comparator.f90:519-625 */
static float shake_apply(int kind, const float *const *x, const float *f, int np, int n, float dt)
{
    const float pi = 3.14159265358979f;                           /* constants.f90:21 */
    const int m = kind == 1 ? n - 1 : n - 2;
    double mx = -HUGE_VAL, sum = 0.0;
    for (int i = 0; i < m; i++) {
        double v = 0.0;
        for (int k = 0; k < np; k++) {
            const float *a = x[k];
            const float d = kind == 1 ? a[i] - a[i + 1] : a[i] - 2.0f * a[i + 1] + a[i + 2];
            const double t = (double)(f[k] * f[k]) * ((double)d * (double)d);
            v = k == 0 ? t : v + t;
        }
        if (v > mx) mx = v;
        sum += v;
    }
    if (kind == 1) return (float)(sqrt(mx) / (double)dt);
    if (kind == 2) return (float)(sqrt(mx) / (double)(dt * dt));
    return (float)((double)(pi / (2.f * 9.81f) * dt) * sum / (double)(dt * dt));
}

/* probes_max_vecnorm_{1,2,3} / probes_arias_intensity_{1,2,3} (comparator.f90:1012-1058) through
 * probe_norm_timedomain (:824-859), probes_norm_timedomain (:770-822) and probes_norm_timedomain_3 (:700-766) */
float ko_probes_shake(ko_probe **p, int np, int kind)
{
    int span[2], ps[2], t[3][2];
    int all_taper = 1, all_filter = 1;
    for (int k = 0; k < np; k++) { if (p[k]->taper.n <= 0) all_taper = 0; if (p[k]->filter.n <= 0) all_filter = 0; }
    if (np == 1) {
        if (all_taper) { ko_discrete_plf_span(&p[0]->taper, p[0]->dt, ps); span_isect(ps, p[0]->span, span); }
        else { span[0] = p[0]->dataspan[0]; span[1] = p[0]->dataspan[1]; }
    } else {
        if (np == 2) probes_adjust_spans(p[0], p[1]); else probes_adjust_spans_3(p[0], p[1], p[2]);
        if (all_taper) {
            for (int k = 0; k < np; k++) { ko_discrete_plf_span(&p[k]->taper, p[k]->dt, ps); span_isect(ps, p[k]->span, t[k]); }
            if (t[0][0] > t[0][1]) { span[0] = t[1][0]; span[1] = t[1][1]; }
            else if (t[1][0] > t[1][1]) { span[0] = t[0][0]; span[1] = t[0][1]; }
            else if (np == 3 && t[2][0] > t[2][1]) { span[0] = t[2][0]; span[1] = t[2][1]; }     /* as the reference has it */
            else { span_union(t[0], t[1], span); if (np == 3) { int u[2] = { span[0], span[1] }; span_union(u, t[2], span); } }
        } else {
            span_union(p[0]->dataspan, p[1]->dataspan, span);
            if (np == 3) { int u[2] = { span[0], span[1] }; span_union(u, p[2]->dataspan, span); }
        }
        if (span[0] > span[1]) return 0.f;
    }
    const float *x[3];
    float f[3];
    for (int k = 0; k < np; k++) {
        const float *src;
        if (all_filter) { update_array_filtered(p[k]); src = p[k]->array_filtered; }
        else if (all_taper) { update_array_tapered(p[k]); src = p[k]->array_tapered; }
        else src = p[k]->array;
        x[k] = src + (span[0] - p[k]->span[0]);
        f[k] = p[k]->factor;
    }
    return shake_apply(kind, x, f, np, slen(span), p[0]->dt);
}

/* probe_get_amp_spectrum, comparator.f90:333-354: returns the number of bins */
int ko_probe_get_amp_spectrum(ko_probe *p, int filtered, float *df, float *out, int maxn)
{
    if (!p->array) return 0;
    update_spectrum_filtered(p);
    *df = p->df;
    const float *src = (p->filter.n > 0 && filtered) ? p->amp_spectrum_filtered : p->amp_spectrum;
    const int n = p->nspec < maxn ? p->nspec : maxn;
    memcpy(out, src, sizeof(float) * (size_t)n);
    return p->nspec;
}

/* probe_get_plain/tapered/filtered, comparator.f90:350-420 */
int ko_probe_get(ko_probe *p, int which, int *lo, float *out, int maxn)
{
    int span[2], ps[2];
    const float *src;
    if (!p->array) { *lo = 1; if (maxn > 0) out[0] = 0.f; return 1; }
    if (which == 3 && p->filter.n > 0) {
        update_array_filtered(p);
        if (p->taper.n > 0) {
            ko_discrete_plf_span(&p->taper, p->dt, ps); span_isect(ps, p->span, span);
            if (span[0] > span[1]) { span[0] = p->dataspan[0]; span[1] = p->dataspan[1]; }
        } else { span[0] = p->dataspan[0]; span[1] = p->dataspan[1]; }
        src = p->array_filtered;
    } else if (which >= 2 && p->taper.n > 0) {
        update_array_tapered(p);
        ko_discrete_plf_span(&p->taper, p->dt, ps); span_isect(ps, p->dataspan, span);
        if (span[0] > span[1]) { span[0] = p->dataspan[0]; span[1] = p->dataspan[1]; }
        src = p->array_tapered;
    } else {
        span[0] = p->dataspan[0]; span[1] = p->dataspan[1];
        src = p->array;
    }
    int n = slen(span);
    *lo = span[0];
    for (int i = 0; i < n && i < maxn; i++) out[i] = src[span[0] - p->span[0] + i];
    return n;
}
