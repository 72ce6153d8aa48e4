/*
 * ko_engine.c -- CPU ORACLE (test infrastructure, not product code; see ko.h).
 * Restates the per-trial path of the reference engine:
 *   seismogram.f90:36-336          make_seismogram, rotate, make_weights
 *   receiver.f90:140-437,853-904   receiver_init, component lookup, scale/fold, misfits
 *   minimizer_engine.f90:885-945   calculate_seismograms / scale_seismograms / calculate_misfits
 *   minimizer_engine.f90:1130-1172 get_misfits
 * Loop structure is the reference's: OpenMP parallel-do over receivers
 * (minimizer_engine.f90:893-903), sequential over centroids, one
 * trace_multiply_add per Green's function component.
 */
#include "ko.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <omp.h>

const ko_trace *ko_gfdb_get_trace_bilin(ko_gfdb *db, const int ix[2], const int iz[2], int ig,
                                        float dix, float diz);

static const float  PI_F = 3.14159265358979f;     /* constants.f90:21 */

#define C_AWAY 1
#define C_RIGHT 2
#define C_DOWN 3
#define C_NORTH 4
#define C_EAST 5

/* receiver.f90:294-307 character_to_id over component_names(-5:5) = w s u l c ? a r d n e */
static int character_to_id(char ch)
{
    static const char names[] = "wsulc?ardne";
    for (int i = 0; i < 11; i++) if (names[i] == ch) return i - 5;
    return 0;
}

/* receiver.f90:309-329 (1-based index, 0 if absent) */
static int component_index(const ko_receiver *r, int comp)
{
    for (int i = 0; i < r->ncomponents; i++) if (abs(r->components[i]) == abs(comp)) return i + 1;
    return 0;
}
/* receiver.f90:331-351 */
static float component_sign(const ko_receiver *r, int comp)
{
    for (int i = 0; i < r->ncomponents; i++)
        if (abs(r->components[i]) == abs(comp)) return r->components[i] < 0 ? -1.f : 1.f;
    return 0.f;
}

static void receiver_destroy(ko_receiver *r)
{
    for (int i = 0; i < 5; i++) {
        ko_strip_destroy(&r->displacement[i]);
        ko_probe_destroy(&r->ref_probes[i]);
        ko_probe_destroy(&r->syn_probes[i]);
    }
}

ko_engine *ko_engine_create(ko_gfdb *db)
{
    ko_engine *e = (ko_engine *)calloc(1, sizeof(ko_engine));
    e->db = db;
    e->effective_dt = 1.f;            /* minimizer_engine.f90:79 */
    e->xundersample = e->zundersample = 1;
    e->misfit_method = KO_L2NORM;
    e->nthreads = 1;
    e->psm.moment = 1.f;
    return e;
}

void ko_engine_destroy(ko_engine *e)
{
    if (!e) return;
    for (int i = 0; i < e->nreceivers; i++) receiver_destroy(&e->receivers[i]);
    free(e->receivers);
    free(e->centroids);
    free(e);
}

void ko_engine_set_nthreads(ko_engine *e, int n) { e->nthreads = n > 0 ? n : 1; }

/* minimizer_engine.f90:236-262 + receiver_init receiver.f90:140-211 */
int ko_engine_set_receivers(ko_engine *e, int n, const double *lat_deg, const double *lon_deg,
                            const float *depth, const char *const *comps)
{
    for (int i = 0; i < e->nreceivers; i++) receiver_destroy(&e->receivers[i]);
    free(e->receivers);
    e->receivers = (ko_receiver *)calloc((size_t)n, sizeof(ko_receiver));
    e->nreceivers = n;
    for (int i = 0; i < n; i++) {
        ko_receiver *r = &e->receivers[i];
        r->enabled = 1;
        r->dt = e->db->dt;
        r->origin.lat = ko_d2r_d(lat_deg[i]);      /* d2r(origin), orthodrome.f90:293-301 */
        r->origin.lon = ko_d2r_d(lon_deg[i]);
        r->depth = depth ? depth[i] : 0.f;
        int nc = (int)strlen(comps[i]);
        if (nc > 5) return -1;
        if (nc == 0) r->enabled = 0;
        for (int k = 0; k < nc; k++) {
            int id = character_to_id(comps[i][k]);
            if (id == 0) return -1;
            for (int j = 0; j < k; j++) if (abs(r->components[j]) == abs(id)) return -1;
            r->components[k] = id;
        }
        r->ncomponents = nc;
        for (int k = 0; k < 5; k++) {
            ko_probe_init(&r->ref_probes[k], r->dt);
            ko_probe_init(&r->syn_probes[k], r->dt);
        }
    }
    return 0;
}

/* minimizer_engine.f90:288-309 + receiver_set_enabled receiver.f90:274-292 */
void ko_engine_switch_receiver(ko_engine *e, int irec1, int state)
{
    ko_receiver *r = &e->receivers[irec1 - 1];
    if (!state)
        for (int k = 0; k < r->ncomponents; k++)
            for (int i = 0; i < r->displacement[k].n; i++) r->displacement[k].d[i] = 0.f;
    r->enabled = state;
}

/* minimizer.f90:485-517 (lat, lon parsed as default real, d2r in fp32) + minimizer_engine.f90:453-467 */
void ko_engine_set_source_location(ko_engine *e, float lat_deg, float lon_deg, double ref_time)
{
    e->origin.lat = (double)ko_d2r_r(lat_deg);
    e->origin.lon = (double)ko_d2r_r(lon_deg);
    e->ref_time = ref_time;
}

void ko_engine_set_effective_dt(ko_engine *e, float dt) { e->effective_dt = dt; }
void ko_engine_set_interpolation(ko_engine *e, int bilinear, int xus, int zus)
{
    e->interpolate = bilinear; e->xundersample = xus; e->zundersample = zus;
}

/* minimizer_engine.f90:500-523 + discretize_source :876-883 */
int ko_engine_set_source_params(ko_engine *e, int sourcetype, const float *params)
{
    int omc = ko_psm_set(&e->psm, sourcetype, params);
    if (omc < 0) return -1;
    if (omc && e->centroids) return 0;       /* only re-scale (:516-517) */
    free(e->centroids); e->centroids = NULL;
    e->ncentroids = ko_psm_to_tdsm(&e->psm, e->effective_dt, &e->centroids);
    return e->ncentroids < 0 ? -1 : 0;
}

void ko_engine_set_centroids(ko_engine *e, int n, const ko_centroid *c, float moment, float risetime)
{
    free(e->centroids);
    e->centroids = (ko_centroid *)malloc(sizeof(ko_centroid) * (size_t)(n > 0 ? n : 1));
    memcpy(e->centroids, c, sizeof(ko_centroid) * (size_t)n);
    e->ncentroids = n;
    e->psm.moment = moment;
    e->psm.risetime = risetime;
}

/* receiver.f90:746-851: the strip spans (ibeg+1 .. ibeg+n); here 'first' = ibeg+1 */
void ko_engine_set_reference(ko_engine *e, int irec1, int icomp1, int first, int n, const float *data)
{
    ko_receiver *r = &e->receivers[irec1 - 1];
    ko_strip s = { NULL, 1, 0 };
    ko_strip_init(&s, first, first + n - 1, data);
    ko_probe_set_array(&r->ref_probes[icomp1 - 1], &s, 1.f);
    ko_strip_destroy(&s);
}

static void mkplf(ko_plf *p, int npts, const float *x, const float *y)
{
    p->n = npts;
    for (int i = 0; i < npts; i++) { p->x[i] = x[i]; p->y[i] = y[i]; }
}

/* receiver.f90:372-389 */
void ko_engine_set_taper(ko_engine *e, int irec1, int npts, const float *x, const float *y)
{
    ko_receiver *r = &e->receivers[irec1 - 1];
    ko_plf p; mkplf(&p, npts, x, y);
    for (int k = 0; k < r->ncomponents; k++) {
        ko_probe_set_taper(&r->ref_probes[k], &p);
        ko_probe_set_taper(&r->syn_probes[k], &p);
    }
}
/* receiver.f90:355-370 */
void ko_engine_set_filter(ko_engine *e, int irec1, int npts, const float *x, const float *y)
{
    ko_receiver *r = &e->receivers[irec1 - 1];
    ko_plf p; mkplf(&p, npts, x, y);
    for (int k = 0; k < r->ncomponents; k++) {
        ko_probe_set_filter(&r->ref_probes[k], &p);
        ko_probe_set_filter(&r->syn_probes[k], &p);
    }
}
void ko_engine_set_misfit_method(ko_engine *e, int method) { e->misfit_method = method; }
/* receiver.f90:391-405 */
void ko_engine_set_synthetics_factor(ko_engine *e, float f)
{
    for (int i = 0; i < e->nreceivers; i++)
        for (int k = 0; k < e->receivers[i].ncomponents; k++) e->receivers[i].syn_probes[k].factor = f;
}
void ko_engine_set_floating_shiftrange(ko_engine *e, int irec1, int lo, int hi)
{
    e->receivers[irec1 - 1].floating_shiftrange[0] = lo;
    e->receivers[irec1 - 1].floating_shiftrange[1] = hi;
}

/* probe bookkeeping for diagnostics: out = span(1:2), dataspan(1:2) of the reference (which = 0) or synthetic probe */
void ko_engine_probe_spans(ko_engine *e, int irec1, int icomp1, int which, int out[4])
{
    ko_receiver *r = &e->receivers[irec1 - 1];
    ko_probe *p = which ? &r->syn_probes[icomp1 - 1] : &r->ref_probes[icomp1 - 1];
    out[0] = p->span[0]; out[1] = p->span[1]; out[2] = p->dataspan[0]; out[3] = p->dataspan[1];
}

/* get_component_ids, receiver.f90:512-542 (1-based component indices, 0 = none) */
static void get_component_ids(const ko_receiver *r, int *iver, int *ihor1, int *ihor2)
{
    *ihor1 = 0; *ihor2 = 0; *iver = 0;
    for (int k = 0; k < r->ncomponents; k++) {
        const int ict = abs(r->components[k]);
        if (ict == 1) *ihor1 = k + 1;
        if (ict == 2) *ihor2 = k + 1;
        if (ict == 3) *iver = k + 1;
    }
    if (*ihor1 == 0 || *ihor2 == 0)
        for (int k = 0; k < r->ncomponents; k++) {
            const int ict = abs(r->components[k]);
            if (ict == 4) *ihor1 = k + 1;
            if (ict == 5) *ihor2 = k + 1;
        }
    if (*ihor1 == 0 || *ihor2 == 0) { *ihor1 = 0; *ihor2 = 0; }
}

/* get_peak_amplitudes (minimizer_engine.f90:1174-1212, receiver_get_maxabs receiver.f90:544-574) for differentiate = 1, 2
 * and get_arias_intensities (:1214-1246, receiver_get_arias_intensity receiver.f90:576-594) for differentiate = 0:
 * one value per ENABLED receiver; the synthetic probes must be current.  Returns the count. */
int ko_engine_shake(ko_engine *e, int differentiate, float *out)
{
    int n = 0;
    for (int ir = 0; ir < e->nreceivers; ir++) {
        ko_receiver *r = &e->receivers[ir];
        if (!r->enabled) continue;
        int iver, ih1, ih2;
        get_component_ids(r, &iver, &ih1, &ih2);
        ko_probe *p[3];
        int np = 0;
        float val = 0.f;
        if (differentiate) {
            const int ic[3] = { iver, ih1, ih2 };
            for (int i = 0; i < 3; i++) if (ic[i]) p[np++] = &r->syn_probes[ic[i] - 1];
            if (np) val = ko_probes_shake(p, np, differentiate);
        } else {
            if (iver && ih1 && ih2) { p[0] = &r->syn_probes[iver - 1]; p[1] = &r->syn_probes[ih1 - 1]; p[2] = &r->syn_probes[ih2 - 1]; np = 3; }
            else if (ih1 && ih2) { p[0] = &r->syn_probes[ih1 - 1]; p[1] = &r->syn_probes[ih2 - 1]; np = 2; }
            else if (iver) { p[0] = &r->syn_probes[iver - 1]; np = 1; }
            if (np) val = ko_probes_shake(p, np, 3);
        }
        out[n++] = val;
    }
    return n;
}

/* receiver_shift_ref_seismogram, receiver.f90:800-814 (shift in samples) */
void ko_engine_shift_ref_seismogram(ko_engine *e, int irec1, int ishift)
{
    ko_receiver *r = &e->receivers[irec1 - 1];
    for (int k = 0; k < r->ncomponents; k++) ko_probe_shift(&r->ref_probes[k], ishift);
}

/* receiver_output_seismogram_spectra, receiver.f90:666-708 (one probe; the data instead of a file) */
int ko_engine_amp_spectrum(ko_engine *e, int irec1, int icomp1, int synthetic, int filtered, float *df, float *out, int maxn)
{
    ko_receiver *r = &e->receivers[irec1 - 1];
    ko_probe *p = synthetic ? &r->syn_probes[icomp1 - 1] : &r->ref_probes[icomp1 - 1];
    return ko_probe_get_amp_spectrum(p, filtered, df, out, maxn);
}

/* receiver_calculate_cross_correlations, receiver.f90:597-616: cc[k][q] for shifts lo..hi (samples); returns ncomponents
 * (0 for a disabled receiver).  The synthetic probes must be current. */
int ko_engine_cross_correlations(ko_engine *e, int irec1, int lo, int hi, float *cc)
{
    ko_receiver *r = &e->receivers[irec1 - 1];
    if (!r->enabled) return 0;
    const int ns = hi - lo + 1;
    for (int k = 0; k < r->ncomponents; k++)
        ko_probes_windowed_cross_corr(&r->syn_probes[k], &r->ref_probes[k], lo, hi, cc + (size_t)k * ns);
    return r->ncomponents;
}

/* receiver_autoshift_ref_seismogram, receiver.f90:816-832 (range in samples; synthetics must be current).
 * Returns the shift applied. */
int ko_engine_autoshift_ref_seismogram(ko_engine *e, int irec1, int lo, int hi)
{
    ko_receiver *r = &e->receivers[irec1 - 1];
    if (!r->enabled || r->ncomponents == 0) return 0;
    const int ns = hi - lo + 1, nc = r->ncomponents;
    float *cc = (float *)malloc(sizeof(float) * (size_t)(ns * nc));
    for (int k = 0; k < nc; k++)                     /* receiver_calculate_cross_correlations, :597-616 */
        ko_probes_windowed_cross_corr(&r->syn_probes[k], &r->ref_probes[k], lo, hi, cc + (size_t)k * ns);
    float mx = cc[0];
    for (int i = 1; i < ns * nc; i++) if (cc[i] > mx) mx = cc[i];
    const float den = mx > 1.f ? mx : 1.f;
    int imax = 0; float best = 0.f;
    for (int q = 0; q < ns; q++) {
        float s = 0.f;
        for (int k = 0; k < nc; k++) { float x = cc[(size_t)k * ns + q] / den; if (x < 0.f) x = 0.f; s = s + x * x; }
        if (q == 0 || s > best) { best = s; imax = q; }
    }
    free(cc);
    const int ishift = imax + lo;
    ko_engine_shift_ref_seismogram(e, irec1, ishift);
    return ishift;
}

/* receiver%floating_shift after calculate_misfits with a floating norm (receiver.f90:498), in samples */
int ko_engine_get_floating_shift(ko_engine *e, int irec1) { return e->receivers[irec1 - 1].floating_shift; }

/* seismogram.f90:316-336 */
static void make_weights(float azimuth, const float m[6], float f[6])
{
    float sa = sinf(azimuth), ca = cosf(azimuth);
    float s2a = sinf(2.f * azimuth), c2a = cosf(2.f * azimuth);
    f[0] = m[0] * (ca * ca) + m[1] * (sa * sa) + m[3] * s2a;
    f[1] = m[4] * ca + m[5] * sa;
    f[2] = m[2];
    f[3] = 0.5f * (m[1] - m[0]) * s2a + m[3] * c2a;
    f[4] = m[5] * ca - m[4] * sa;
    f[5] = m[0] * (sa * sa) + m[1] * (ca * ca) - m[3] * s2a;
}

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* strip_extend_to_same_span_N, sparse_trace.f90:218-314 */
static void extend_to_same_span(ko_strip **s, int n)
{
    int lo = 2147483647, hi = -2147483647;
    for (int i = 0; i < n; i++)
        if (s[i]->d) { lo = imin(lo, s[i]->lo); hi = imax(hi, s[i]->lo + s[i]->n - 1); }
    if (lo < hi) for (int i = 0; i < n; i++) ko_strip_extend(s[i], lo, hi);
}

void ko_engine_receiver_geometry(ko_engine *e, int irec1, double *azi, double *bazi, double *dist)
{
    ko_receiver *r = &e->receivers[irec1 - 1];
    ko_azibazi(e->origin, r->origin, azi, bazi);                  /* seismogram.f90:99 */
    *dist = ko_distance_accurate50m(e->origin, r->origin);        /* seismogram.f90:100 */
}


/* The per (receiver, centroid) quantities make_seismogram derives before touching any trace
 * (seismogram.f90:133-165 + gfdb.f90:781-815 + sparse_trace.f90:640-645), packed like the
 * product's GeoRec (20 x 4 bytes) so that tests can compare the device geometry kernel field by
 * field: int row[4] (0-based first GF row of the 4 nodes, -1 = centroid skipped), float w[4],
 * int ishift, float wfrac, float f[6], float cl, sl, int flags (bit0 direct, bit1 rotate), pad. */
void ko_engine_centroid_geometry(ko_engine *e, int irec1, int icent0, void *out20)
{
    ko_receiver *rec = &e->receivers[irec1 - 1];
    ko_gfdb *db = e->db;
    const ko_centroid *c = &e->centroids[icent0];
    int *oi = (int *)out20; float *of = (float *)out20;
    double azi_orig, bazi_orig, dist_orig, azi, bazi, dist;
    ko_azibazi(e->origin, rec->origin, &azi_orig, &bazi_orig);
    dist_orig = ko_distance_accurate50m(e->origin, rec->origin);
    ko_approx_differential_azidist(c->north, c->east, azi_orig, bazi_orig, dist_orig, &azi, &bazi, &dist);
    float f[6];
    make_weights((float)azi, c->m, f);
    int ix[2], iz[2]; float dix = 0.f, diz = 0.f;
    if (e->interpolate) {
        ko_gfdb_get_indices_bilin(db, (float)dist, c->depth - rec->depth, e->xundersample, e->zundersample, ix, iz, &dix, &diz);
    } else {
        ko_gfdb_get_indices(db, (float)dist, c->depth - rec->depth, &ix[0], &iz[0]);
        ix[1] = ix[0] + 1; iz[1] = iz[0] + 1;
    }
    int direct = (dix == 0.f && diz == 0.f);
    int need_h = component_index(rec, C_AWAY) || component_index(rec, C_RIGHT) || component_index(rec, C_NORTH) || component_index(rec, C_EAST);
    int has_d = component_index(rec, C_DOWN) != 0;
    /* row = -1: nothing of this centroid is added -- the first trace make_seismogram asks for (ig 1, or 6 for a
     * vertical-only receiver) is missing at one of the nodes, or the horizontals of the rotate branch are dropped and with
     * them the vertical block (seismogram.f90:171-250) */
    double lambda0 = bazi - bazi_orig;
    static const int seq_h10[6] = { 1, 2, 3, 9, 4, 5 }, seq_h8[5] = { 1, 2, 3, 4, 5 }, seq_d[4] = { 6, 7, 8, 10 };
    int nH = db->ng == 10 ? 6 : 5, nD = db->ng == 10 ? 4 : 3, nlh = 0, nld = 0, hfull = 1;
#define STORED(ig, res) do { res = 1; for (int a = 0; a < (direct ? 1 : 2); a++) for (int b = 0; b < (direct ? 1 : 2); b++) { \
        int sp[2]; if (!ko_gfdb_trace_span(db, ix[a], iz[b], (ig), sp)) res = 0; } } while (0)
    if (need_h) {
        int k = 0, st;
        for (; k < nH; k++) { STORED(db->ng == 10 ? seq_h10[k] : seq_h8[k], st); if (!st) break; }
        hfull = (k == nH);
        nlh = hfull ? nH : (lambda0 != 0. ? 0 : k);
    }
    if (has_d && hfull) {
        int k = 0, st;
        for (; k < nD; k++) { STORED(seq_d[k], st); if (!st) break; }
        nld = k;
    }
#undef STORED
    int ok = (nlh > 0 || nld > 0);
#define ROW(a, b) (((ix[a] - 1) * db->nz + (iz[b] - 1)) * db->ng)
    if (!ok) { oi[0] = oi[1] = oi[2] = oi[3] = -1; }
    else if (direct) { oi[0] = oi[1] = oi[2] = oi[3] = ROW(0, 0); }
    else { oi[0] = ROW(0, 0); oi[1] = ROW(0, 1); oi[2] = ROW(1, 0); oi[3] = ROW(1, 1); }
#undef ROW
    of[4] = (1.f - dix) * (1.f - diz); of[5] = (1.f - dix) * diz; of[6] = dix * (1.f - diz); of[7] = dix * diz;
    float rshift = c->time / db->dt;
    int its = (int)floorf(rshift);
    oi[8] = its; of[9] = rshift - (float)its;
    for (int k = 0; k < 6; k++) of[10 + k] = f[k];
    double lambda = bazi - bazi_orig;
    of[16] = (float)cos(lambda); of[17] = (float)sin(lambda);
    oi[18] = (direct ? 1 : 0) | (lambda != 0. ? 2 : 0);
    oi[19] = 0;
}

/* seismogram.f90:36-301 */
static void make_seismogram(ko_engine *e, ko_receiver *rec)
{
    ko_gfdb *db = e->db;
    ko_strip temp[2] = { { NULL, 1, 0 }, { NULL, 1, 0 } };
    ko_strip ar[2] = { { NULL, 1, 0 }, { NULL, 1, 0 } };
    int ja = component_index(rec, C_AWAY), jr = component_index(rec, C_RIGHT);
    int jd = component_index(rec, C_DOWN), jn = component_index(rec, C_NORTH), je = component_index(rec, C_EAST);
    float sa = component_sign(rec, C_AWAY), sr = component_sign(rec, C_RIGHT);
    float sd = component_sign(rec, C_DOWN), sn = component_sign(rec, C_NORTH), se = component_sign(rec, C_EAST);
    int need_horizontal = ja || jr || jn || je;
    double azi_orig, bazi_orig, dist_orig;
    ko_azibazi(e->origin, rec->origin, &azi_orig, &bazi_orig);
    dist_orig = ko_distance_accurate50m(e->origin, rec->origin);

    for (int i = 0; i < rec->ncomponents; i++)                    /* :102-106 zero, keep span */
        for (int k = 0; k < rec->displacement[i].n; k++) rec->displacement[i].d[k] = 0.f;

    if (need_horizontal) {                                        /* :109-130 */
        int js[4] = { ja, jr, jn, je };
        for (int q = 0; q < 4; q++) if (js[q]) {
            ko_strip *five[5] = { &rec->displacement[js[q] - 1], &ar[0], &temp[0], &ar[1], &temp[1] };
            extend_to_same_span(five, 5);
        }
        for (int i = 0; i < 2; i++) for (int k = 0; k < ar[i].n; k++) ar[i].d[k] = 0.f;
    }

    for (int ic = 0; ic < e->ncentroids; ic++) {                  /* :131-254 */
        const ko_centroid *c = &e->centroids[ic];
        float dnorth = c->north, deast = c->east, depth = c->depth, time = c->time;
        float f[6];
        double azi, bazi, dist;
        int ix[2], iz[2];
        float dix, diz;
        const ko_trace *tp;
        float rshift = time / db->dt;                             /* :139 */
        ko_approx_differential_azidist(dnorth, deast, azi_orig, bazi_orig, dist_orig, &azi, &bazi, &dist);
        make_weights((float)azi, c->m, f);
        if (e->interpolate) {
            ko_gfdb_get_indices_bilin(db, (float)dist, depth - rec->depth, e->xundersample, e->zundersample,
                                      ix, iz, &dix, &diz);
        } else {
            ko_gfdb_get_indices(db, (float)dist, depth - rec->depth, &ix[0], &iz[0]);
            ix[1] = ix[0] + 1; iz[1] = iz[0] + 1; dix = 0.f; diz = 0.f;
        }
#define GET(ig) tp = ko_gfdb_get_trace_bilin(db, ix, iz, (ig), dix, diz); if (!tp) continue;
        if (need_horizontal) {
            double lambda = bazi - bazi_orig;                     /* :159 */
            if (lambda != 0.) {
                float cl = (float)cos(lambda), sl = (float)sin(lambda);
                for (int k = 0; k < temp[0].n; k++) temp[0].d[k] = 0.f;
                GET(1) ko_trace_multiply_add(tp, &temp[0], f[0], 2, 0, rshift);
                GET(2) ko_trace_multiply_add(tp, &temp[0], f[1], 2, 0, rshift);
                GET(3) ko_trace_multiply_add(tp, &temp[0], f[2], 2, 0, rshift);
                if (db->ng == 10) { GET(9) ko_trace_multiply_add(tp, &temp[0], f[5], 2, 0, rshift); }
                for (int k = 0; k < temp[1].n; k++) temp[1].d[k] = 0.f;
                GET(4) ko_trace_multiply_add(tp, &temp[1], f[3], 2, 0, rshift);
                GET(5) ko_trace_multiply_add(tp, &temp[1], f[4], 2, 0, rshift);
                ko_strip *four[4] = { &temp[0], &temp[1], &ar[0], &ar[1] };
                extend_to_same_span(four, 4);                     /* :196-197 */
                for (int k = 0; k < ar[0].n; k++)                  /* :200-201 */
                    ar[0].d[k] = ar[0].d[k] + cl * temp[0].d[k] - sl * temp[1].d[k];
                for (int k = 0; k < ar[1].n; k++)                  /* :202-203 */
                    ar[1].d[k] = ar[1].d[k] + cl * temp[1].d[k] + sl * temp[0].d[k];
            } else {
                GET(1) ko_trace_multiply_add(tp, &ar[0], f[0], 2, 0, rshift);
                GET(2) ko_trace_multiply_add(tp, &ar[0], f[1], 2, 0, rshift);
                GET(3) ko_trace_multiply_add(tp, &ar[0], f[2], 2, 0, rshift);
                if (db->ng == 10) { GET(9) ko_trace_multiply_add(tp, &ar[0], f[5], 2, 0, rshift); }
                GET(4) ko_trace_multiply_add(tp, &ar[1], f[3], 2, 0, rshift);
                GET(5) ko_trace_multiply_add(tp, &ar[1], f[4], 2, 0, rshift);
            }
        }
        if (jd) {                                                 /* :236-253 */
            ko_strip *dz = &rec->displacement[jd - 1];
            GET(6) ko_trace_multiply_add(tp, dz, f[0] * sd, 2, 0, rshift);
            GET(7) ko_trace_multiply_add(tp, dz, f[1] * sd, 2, 0, rshift);
            GET(8) ko_trace_multiply_add(tp, dz, f[2] * sd, 2, 0, rshift);
            if (db->ng == 10) { GET(10) ko_trace_multiply_add(tp, dz, f[5] * sd, 2, 0, rshift); }
        }
#undef GET
    }

    if (need_horizontal) {                                        /* :256-289 */
        if (ja) {
            ko_strip *two[2] = { &rec->displacement[ja - 1], &ar[0] };
            extend_to_same_span(two, 2);
            for (int k = 0; k < ar[0].n; k++) rec->displacement[ja - 1].d[k] = ar[0].d[k] * sa;
        }
        if (jr) {
            ko_strip *two[2] = { &rec->displacement[jr - 1], &ar[1] };
            extend_to_same_span(two, 2);
            for (int k = 0; k < ar[1].n; k++) rec->displacement[jr - 1].d[k] = ar[1].d[k] * sr;
        }
        if (jn || je) {
            float cl = (float)cos(bazi_orig + (double)PI_F);      /* :270 'pi' is the default-real constant */
            float sl = (float)sin(bazi_orig + (double)PI_F);
            ko_strip *two[2] = { &ar[0], &ar[1] };
            extend_to_same_span(two, 2);
            for (int k = 0; k < ar[0].n; k++) {                   /* rotate :303-314 */
                float a = ar[0].d[k], b = ar[1].d[k];
                float aa = cl * a - sl * b;
                b = cl * b + sl * a;
                ar[0].d[k] = aa; ar[1].d[k] = b;
            }
            if (jn) {
                ko_strip *t2[2] = { &rec->displacement[jn - 1], &ar[0] };
                extend_to_same_span(t2, 2);
                for (int k = 0; k < ar[0].n; k++) rec->displacement[jn - 1].d[k] = ar[0].d[k] * sn;
            }
            if (je) {
                ko_strip *t2[2] = { &rec->displacement[je - 1], &ar[1] };
                extend_to_same_span(t2, 2);
                for (int k = 0; k < ar[1].n; k++) rec->displacement[je - 1].d[k] = ar[1].d[k] * se;
            }
        }
        ko_strip_destroy(&temp[0]); ko_strip_destroy(&temp[1]);
        ko_strip_destroy(&ar[0]); ko_strip_destroy(&ar[1]);
    }
}

/* minimizer_engine.f90:885-907 */
void ko_engine_calculate_seismograms(ko_engine *e)
{
    int n = e->nreceivers;
#pragma omp parallel for schedule(dynamic) num_threads(e->nthreads)
    for (int i = 0; i < n; i++)
        if (e->receivers[i].enabled) make_seismogram(e, &e->receivers[i]);
}

static int nint_f(float x) { return (int)roundf(x); }

/* receiver.f90:853-904 */
static void scaled_seismograms_to_probes(ko_receiver *rec, float risetime, float moment)
{
    if (!rec->enabled) return;
    int nshifts = 0;
    float *weights = NULL, *shifts = NULL;
    if (risetime > 0.f) {
        float rrise[2] = { -risetime / 2.f, +risetime / 2.f };
        nshifts = 1 + 2 * nint_f(0.5f * risetime / rec->dt);
        weights = (float *)malloc(sizeof(float) * 2 * (size_t)nshifts);
        shifts = weights + nshifts;
        for (int is = 1; is <= nshifts; is++) {
            float ts = ((float)(is - 1) - 0.5f * (float)(nshifts - 1)) * rec->dt;
            float rsamp[2] = { ts - rec->dt / 2.f, ts + rec->dt / 2.f };
            float ro0 = fmaxf(rrise[0], rsamp[0]), ro1 = fminf(rrise[1], rsamp[1]);
            weights[is - 1] = fmaxf(0.f, ro1 - ro0);
            shifts[is - 1] = ts / rec->dt;
        }
        float sum = 0.f;
        for (int i = 0; i < nshifts; i++) sum = sum + weights[i];
        for (int i = 0; i < nshifts; i++) weights[i] = weights[i] / sum;
    }
    ko_strip tmp = { NULL, 1, 0 };
    for (int k = 0; k < rec->ncomponents; k++) {
        if (!rec->displacement[k].d) continue;
        ko_strip_copy(&rec->displacement[k], &tmp);
        if (risetime > 0.f) ko_strip_fold(&tmp, nshifts, shifts, weights);
        ko_probe_set_array(&rec->syn_probes[k], &tmp, moment);
    }
    free(weights);
    ko_strip_destroy(&tmp);
}

/* minimizer_engine.f90:909-921 */
void ko_engine_scale_seismograms(ko_engine *e)
{
    for (int i = 0; i < e->nreceivers; i++)
        scaled_seismograms_to_probes(&e->receivers[i], e->psm.risetime, e->psm.moment);
}

/* receiver.f90:439-510 */
static void calculate_floating_misfits(ko_receiver *r, int misfit_method)
{
    int eval = (misfit_method == KO_FLOATING_L1NORM) ? KO_L1NORM : KO_L2NORM;
    int nc = r->ncomponents;
    if (nc == 0) return;
    if (!r->enabled) { for (int k = 0; k < nc; k++) { r->misfits[k] = 0.f; r->misfits_norm_factors[k] = 0.f; } return; }
    int lo = r->floating_shiftrange[0], hi = r->floating_shiftrange[1];
    int ns = hi - lo + 1;
    float *mis = (float *)malloc(sizeof(float) * 2 * (size_t)(nc * ns)), *nrm = mis + nc * ns;
    int ishift = lo;
    for (int i = 0; i < ns; i++) {
        for (int k = 0; k < nc; k++) {
            ko_probe_shift(&r->ref_probes[k], ishift);
            mis[k * ns + i] = ko_probes_norm(&r->ref_probes[k], &r->syn_probes[k], eval);
            nrm[k * ns + i] = ko_probe_norm(&r->ref_probes[k], eval);
        }
        ishift = 1;
    }
    int iloc = 0; float best = 0.f;
    for (int i = 0; i < ns; i++) {
        float s = 0.f;
        for (int k = 0; k < nc; k++) s = s + (eval == KO_L1NORM ? mis[k * ns + i] : mis[k * ns + i] * mis[k * ns + i]);
        if (i == 0 || s < best) { best = s; iloc = i; }
    }
    r->floating_shift = lo + iloc;
    for (int k = 0; k < nc; k++) {
        r->misfits[k] = mis[k * ns + iloc];
        float s = 0.f;
        for (int i = 0; i < ns; i++) s = s + nrm[k * ns + i];
        r->misfits_norm_factors[k] = s / (float)ns;
    }
    for (int k = 0; k < nc; k++) ko_probe_shift(&r->ref_probes[k], -hi);
    free(mis);
}

/* receiver.f90:407-437 */
static void receiver_calculate_misfits(ko_receiver *r, int misfit_method)
{
    if (misfit_method == KO_FLOATING_L1NORM || misfit_method == KO_FLOATING_L2NORM) {
        calculate_floating_misfits(r, misfit_method);
        return;
    }
    for (int k = 0; k < r->ncomponents; k++) {
        if (r->enabled) {
            r->misfits[k] = ko_probes_norm(&r->ref_probes[k], &r->syn_probes[k], misfit_method);
            r->misfits_norm_factors[k] = ko_probe_norm(&r->ref_probes[k], misfit_method);
        } else {
            r->misfits[k] = 0.f; r->misfits_norm_factors[k] = 0.f;
        }
    }
}

/* minimizer_engine.f90:924-945 */
void ko_engine_calculate_misfits(ko_engine *e)
{
    float misfit = 0.f, nf = 0.f;
    for (int i = 0; i < e->nreceivers; i++) {
        ko_receiver *r = &e->receivers[i];
        receiver_calculate_misfits(r, e->misfit_method);
        float s = 0.f, t = 0.f;
        for (int k = 0; k < r->ncomponents; k++) s = s + r->misfits[k] * r->misfits[k];
        for (int k = 0; k < r->ncomponents; k++) t = t + r->misfits_norm_factors[k] * r->misfits_norm_factors[k];
        misfit = misfit + s;
        nf = nf + t;
    }
    e->misfit = sqrtf(misfit) / sqrtf(nf);
}

/* minimizer_engine.f90:1130-1172 */
int ko_engine_get_misfits(ko_engine *e, float *m, float *n, int maxn)
{
    ko_engine_calculate_seismograms(e);
    ko_engine_scale_seismograms(e);
    ko_engine_calculate_misfits(e);
    int im = 0;
    for (int i = 0; i < e->nreceivers; i++) {
        ko_receiver *r = &e->receivers[i];
        if (!r->enabled) continue;
        for (int k = 0; k < r->ncomponents; k++) {
            if (im < maxn) { m[im] = r->misfits[k]; n[im] = r->misfits_norm_factors[k]; }
            im++;
        }
    }
    return im;
}

float ko_engine_get_global_misfit(ko_engine *e) { return e->misfit; }

int ko_engine_get_displacement(ko_engine *e, int irec1, int icomp1, int *lo, float *out, int maxn)
{
    ko_strip *s = &e->receivers[irec1 - 1].displacement[icomp1 - 1];
    *lo = s->lo;
    for (int i = 0; i < s->n && i < maxn; i++) out[i] = s->d[i];
    return s->n;
}

int ko_engine_get_synthetic(ko_engine *e, int irec1, int icomp1, int which, int *lo, float *out, int maxn)
{
    return ko_probe_get(&e->receivers[irec1 - 1].syn_probes[icomp1 - 1], which, lo, out, maxn);
}

/* the reference probe the same way (receiver_output_seismogram with which_probe = REFERENCES, receiver.f90:618-680) */
int ko_engine_get_reference(ko_engine *e, int irec1, int icomp1, int which, int *lo, float *out, int maxn)
{
    return ko_probe_get(&e->receivers[irec1 - 1].ref_probes[icomp1 - 1], which, lo, out, maxn);
}
