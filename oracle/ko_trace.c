/*
 * ko_trace.c -- CPU ORACLE (test infrastructure, not product code; see ko.h).
 * Restates sparse_trace.f90: t_strip / t_trace and their arithmetic.
 * Every Fortran array statement is kept as its own loop, in source order, so
 * each fp32 multiply and add rounds exactly where the reference's does.
 */
#include "ko.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>

#define MAXGAP 5   /* sparse_trace.f90:25 */

/* util.f90:339-357 resize_r: (re)allocate, contents undefined */
static void strip_resize(ko_strip *s, int lo, int n)
{
    if (s->d == NULL) {
        if (n == 0) return;
        s->d = (float *)malloc(sizeof(float) * (size_t)n);
        s->lo = lo; s->n = n;
        return;
    }
    if (s->n != n || s->lo != lo) {
        free(s->d); s->d = NULL; s->n = 0;
        if (n != 0) {
            s->d = (float *)malloc(sizeof(float) * (size_t)n);
            s->lo = lo; s->n = n;
        }
    }
}

static inline int s_lb(const ko_strip *s) { return s->lo; }
static inline int s_ub(const ko_strip *s) { return s->lo + s->n - 1; }
#define AT(s, i) ((s)->d[(i) - (s)->lo])

/* sparse_trace.f90:73-89 */
void ko_strip_init(ko_strip *s, int lo, int hi, const float *data)
{
    int n = hi - lo + 1;
    strip_resize(s, lo, n);
    if (n > 0) memcpy(s->d, data, sizeof(float) * (size_t)n);
}

/* sparse_trace.f90:102-105 */
void ko_strip_destroy(ko_strip *s)
{
    if (s->d) free(s->d);
    s->d = NULL; s->n = 0; s->lo = 1;
}

/* sparse_trace.f90:206-216 */
void ko_strip_copy(const ko_strip *src, ko_strip *dst)
{
    strip_resize(dst, src->lo, src->n);
    if (src->n > 0) memcpy(dst->d, src->d, sizeof(float) * (size_t)src->n);
}

/* sparse_trace.f90:316-345: zeros to the left, LAST VALUE repeated to the right */
void ko_strip_extend(ko_strip *s, int lo, int hi)
{
    int n = hi - lo + 1;
    if (s->d) {
        int r1 = s_lb(s), r2 = s_ub(s);
        size_t on = (size_t)s->n;
        float *temp = (float *)malloc(sizeof(float) * on);
        memcpy(temp, s->d, sizeof(float) * on);
        strip_resize(s, lo, n);
        if (lo < r1) for (int i = lo; i <= r1 - 1; i++) AT(s, i) = 0.f;
        if (hi > r2) for (int i = r2 + 1; i <= hi; i++) AT(s, i) = temp[on - 1];
        for (int i = r1; i <= r2; i++) AT(s, i) = temp[i - r1];
        free(temp);
    } else {
        strip_resize(s, lo, n);
        for (int i = 0; i < s->n; i++) s->d[i] = 0.f;
    }
}

/* sparse_trace.f90:347-377 */
void ko_strip_dataspan(const ko_strip *s, int out[2])
{
    if (s->n == 0) { out[0] = 0; out[1] = -1; return; }
    int lo = s_lb(s), hi = s_ub(s);
    out[0] = lo; out[1] = hi;
    float firstvalue = 0.f;
    for (int i = lo; i <= hi; i++) {
        out[0] = i;
        if (AT(s, i) != firstvalue) break;
    }
    float lastvalue = AT(s, hi);
    for (int i = hi; i >= lo; i--) {
        if (AT(s, i) != lastvalue) break;
        out[1] = i;
    }
}

/* sparse_trace.f90:897-915 */
void ko_trace_destroy(ko_trace *t)
{
    if (t->strips) {
        for (int i = 0; i < t->nstrips; i++) ko_strip_destroy(&t->strips[i]);
        free(t->strips);
        t->strips = NULL;
    }
    t->nstrips = 0; t->span[0] = 0; t->span[1] = 0;
}

/* sparse_trace.f90:404-418 */
void ko_trace_create_simple(ko_trace *t, const float *data, int lo, int hi)
{
    ko_trace_destroy(t);
    t->strips = (ko_strip *)calloc(1, sizeof(ko_strip));
    t->nstrips = 1;
    ko_strip_init(&t->strips[0], lo, hi, data);
    t->span[0] = lo; t->span[1] = hi;
}

/* sparse_trace.f90:443-555 (without the last_span_as_hint option) */
void ko_trace_pack(const ko_strip *strip, ko_trace *trace)
{
    int lo = s_lb(strip), hi = s_ub(strip);
    int gap = 0, interest = 0, istrip = 0;
    for (int i = lo; i <= hi; i++) {
        if (AT(strip, i) != 0.f) {
            if (!interest) { interest = 1; istrip++; }
            gap = 0;
        } else if (interest) {
            gap++;
            if (gap > MAXGAP) interest = 0;
        }
    }
    int nstrips = istrip;
    ko_trace_destroy(trace);
    if (nstrips == 0) {   /* :493-511 single zero at the start of the strip */
        trace->strips = (ko_strip *)calloc(1, sizeof(ko_strip));
        float z = 0.f;
        ko_strip_init(&trace->strips[0], lo, lo, &z);
        trace->nstrips = 1;
        trace->span[0] = lo; trace->span[1] = lo;
        return;
    }
    trace->strips = (ko_strip *)calloc((size_t)nstrips, sizeof(ko_strip));
    trace->nstrips = nstrips;
    gap = 0; interest = 0; istrip = 0;
    int ibeg = 0, iend = 0;
    for (int i = lo; i <= hi; i++) {
        if (AT(strip, i) != 0.f) {
            if (!interest) { interest = 1; ibeg = i; istrip++; }
            gap = 0;
            iend = i;
        } else if (interest) {
            gap++;
            if (gap > MAXGAP) {   /* keep one of the zeros */
                ko_strip_init(&trace->strips[istrip - 1], ibeg, iend + 1, &AT(strip, ibeg));
                interest = 0;
            }
        }
    }
    if (interest) {
        if (gap > 0) ko_strip_init(&trace->strips[istrip - 1], ibeg, iend + 1, &AT(strip, ibeg));
        else         ko_strip_init(&trace->strips[istrip - 1], ibeg, iend, &AT(strip, ibeg));
    }
    trace->span[0] = s_lb(&trace->strips[0]);
    trace->span[1] = s_ub(&trace->strips[nstrips - 1]);
}

/* sparse_trace.f90:557-580 */
void ko_trace_unpack(const ko_trace *trace, ko_strip *strip)
{
    int length = trace->span[1] - trace->span[0] + 1;
    strip_resize(strip, trace->span[0], length);
    for (int i = 0; i < strip->n; i++) strip->d[i] = 0.f;
    for (int k = 0; k < trace->nstrips; k++) {
        const ko_strip *ts = &trace->strips[k];
        for (int i = s_lb(ts); i <= s_ub(ts); i++) AT(strip, i) = AT(ts, i);
    }
}

/* sparse_trace.f90:849-878 (poffsets are 1-based positions into packed) */
void ko_trace_from_storable(ko_trace *t, const float *packed, int npacked,
                            const int *poffsets, const int *offsets, int nstrips)
{
    ko_trace_destroy(t);
    t->nstrips = nstrips;
    t->strips = (ko_strip *)calloc((size_t)nstrips, sizeof(ko_strip));
    for (int k = 0; k < nstrips; k++) {
        int n = (k != nstrips - 1) ? poffsets[k + 1] - poffsets[k] : npacked - poffsets[k] + 1;
        ko_strip_init(&t->strips[k], offsets[k], offsets[k] + n - 1, packed + poffsets[k] - 1);
    }
    t->span[0] = s_lb(&t->strips[0]);
    t->span[1] = s_ub(&t->strips[nstrips - 1]);
}

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* sparse_trace.f90:597-707 */
void ko_trace_multiply_add(const ko_trace *t, ko_strip *s, float factor,
                           int mode, int ishift, float rshift)
{
    int itraceshift = 0;
    float weight_right = 0.f, weight_left = 0.f;
    int has_r = (mode == 2);
    if (mode == 1) itraceshift = ishift;
    if (has_r) {
        itraceshift = (int)floorf(rshift);                 /* :640 */
        weight_right = rshift - (float)itraceshift;        /* :642 */
        weight_left = 1.f - weight_right;
        weight_right = weight_right * factor;
        weight_left = weight_left * factor;
    }
    int span[2] = { t->span[0] + itraceshift, t->span[1] + itraceshift };
    int need[2] = { span[0], span[1] };
    if (has_r) need[1] += 1;                               /* :655 */

    if (s->d) {                                            /* :658-668 */
        int c0 = imin(need[0], s_lb(s)), c1 = imax(need[1], s_ub(s));
        if (c0 != s_lb(s) || c1 != s_ub(s)) ko_strip_extend(s, c0, c1);
    } else {
        strip_resize(s, need[0], need[1] - need[0] + 1);
        for (int i = 0; i < s->n; i++) s->d[i] = 0.f;
    }

    for (int k = 0; k < t->nstrips; k++) {                 /* :671-705 */
        const ko_strip *ts = &t->strips[k];
        int ss0 = s_lb(ts) + itraceshift, ss1 = s_ub(ts) + itraceshift;
        if (ss1 < span[0]) continue;
        if (ss0 > span[1]) break;
        int r0 = imax(ss0, span[0]), r1 = imin(ss1, span[1]);
        int last = (k == t->nstrips - 1);
        if (!has_r) {
            for (int i = r0; i <= r1; i++)
                AT(s, i) = AT(s, i) + factor * AT(ts, i - itraceshift);
        } else {
            for (int i = r0; i <= r1; i++)
                AT(s, i) = AT(s, i) + weight_left * AT(ts, i - itraceshift);
            if (last) {
                for (int i = r0 + 1; i <= r1; i++)
                    AT(s, i) = AT(s, i) + weight_right * AT(ts, i - 1 - itraceshift);
            } else {
                for (int i = r0 + 1; i <= r1 + 1; i++)
                    AT(s, i) = AT(s, i) + weight_right * AT(ts, i - 1 - itraceshift);
            }
        }
        if (last && r1 + 1 <= s_ub(s)) {                   /* :698-703 repeat end point */
            float lastval = AT(ts, s_ub(ts));
            if (lastval != 0.f)
                for (int i = r1 + 1; i <= s_ub(s); i++)
                    AT(s, i) = AT(s, i) + factor * lastval;
        }
    }
}

/* sparse_trace.f90:710-792 */
void ko_trace_multiply_add_nogrow(const ko_trace *t, float *array, int alo, int ahi,
                                  float factor, int mode, int ishift, float rshift)
{
#define A(i) array[(i) - alo]
    int itraceshift = 0;
    float weight_right = 0.f, weight_left = 0.f;
    int has_r = (mode == 2);
    if (mode == 1) itraceshift = ishift;
    if (has_r) {
        itraceshift = (int)floorf(rshift);
        weight_right = rshift - (float)itraceshift;
        weight_left = 1.f - weight_right;
        weight_right = weight_right * factor;
        weight_left = weight_left * factor;
    }
    int span[2] = { imax(alo, t->span[0] + itraceshift), imin(ahi, t->span[1] + itraceshift) };
    if (span[1] < span[0]) return;
    for (int k = 0; k < t->nstrips; k++) {
        const ko_strip *ts = &t->strips[k];
        int ss0 = s_lb(ts) + itraceshift, ss1 = s_ub(ts) + itraceshift;
        if (ss1 < span[0]) continue;
        if (ss0 > span[1]) break;
        int r0 = imax(ss0, span[0]), r1 = imin(ss1, span[1]);
        int last = (k == t->nstrips - 1);
        if (!has_r) {
            for (int i = r0; i <= r1; i++) A(i) = A(i) + factor * AT(ts, i - itraceshift);
        } else {
            for (int i = r0; i <= r1; i++) A(i) = A(i) + weight_left * AT(ts, i - itraceshift);
            if (last || r1 + 1 > ahi) {
                for (int i = r0 + 1; i <= r1; i++) A(i) = A(i) + weight_right * AT(ts, i - 1 - itraceshift);
            } else {
                for (int i = r0 + 1; i <= r1 + 1; i++) A(i) = A(i) + weight_right * AT(ts, i - 1 - itraceshift);
            }
        }
        if (last && r1 + 1 <= ahi) {
            float lastval = AT(ts, s_ub(ts));
            if (lastval != 0.f)
                for (int i = r1 + 1; i <= ahi; i++) A(i) = A(i) + factor * lastval;
        }
    }
#undef A
}

/* sparse_trace.f90:379-402 */
void ko_strip_fold(ko_strip *s, int nshifts, const float *shifts, const float *amplitudes)
{
    int ds[2];
    ko_strip_dataspan(s, ds);
    if (ds[1] < ds[0]) return;
    ko_trace t = { 0, { 0, 0 }, NULL };
    ko_trace_create_simple(&t, &AT(s, ds[0]), ds[0], ds[1]);
    for (int i = 0; i < s->n; i++) s->d[i] = 0.f;
    for (int i = 0; i < nshifts; i++)
        ko_trace_multiply_add(&t, s, amplitudes[i], 2, 0, shifts[i]);
    ko_trace_destroy(&t);
}
