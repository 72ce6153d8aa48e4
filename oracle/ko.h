/*
 * ko.h -- CPU ORACLE for the Kiwi trial-source -> synthetics -> misfit path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / timed CPU baseline.  The
 * product (kiwi_amd/, include/kiwi_hip.h) never links or calls it.
 *
 * This is a plain-C restatement of the reference's Fortran algorithm, keeping
 * the reference's data structures (growable t_strip, gap-compressed t_trace,
 * t_probe) and its operation order, so that every fp32 rounding happens where
 * the reference's happens.  Each function cites the reference file:line
 * (paths relative to the reference checkout) it restates.
 *
 * Parity pins (see DESIGN.md "Oracle"):
 *   - leaf arithmetic (sparse_trace, orthodrome, euler, piecewise_linear_function,
 *     discrete sources) is checked bit-for-bit against the reference's own
 *     modules compiled unmodified by oracle/Makefile into oracle/_ref/
 *     (tests/test_oracle_vs_ref.py) and against the committed vectors in
 *     tests/golden/ generated from that build;
 *   - the reference's own unit-test KATs (test_sparse_trace, test_comparator,
 *     test_piecewise_linear_function, test_source_bilat, test_orthodrome) are
 *     re-expressed as data in tests/test_oracle_kats.py;
 *   - gfdb.f90 / seismogram.f90 / receiver.f90 / comparator.f90 themselves are
 *     UNBUILDABLE here (HDF5 Fortran module, FFTW3 include, libmseed absent):
 *     their composition is restated from source and pinned only through the
 *     primitives they call and the reference KATs above.
 */
#ifndef KO_H
#define KO_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- sparse_trace.f90 ---------------- */

/* t_strip (sparse_trace.f90:29-33): dense data with arbitrary lower bound. */
typedef struct {
    float *d;   /* d[0] is sample index lo; NULL when unallocated */
    int lo;     /* lbound */
    int n;      /* size; 0 == not allocated */
} ko_strip;

/* t_trace (sparse_trace.f90:35-50) */
typedef struct {
    int nstrips;
    int span[2];
    ko_strip *strips;   /* NULL == trace_is_empty() */
} ko_trace;

void ko_strip_init(ko_strip *s, int lo, int hi, const float *data);
void ko_strip_destroy(ko_strip *s);
void ko_strip_copy(const ko_strip *src, ko_strip *dst);
void ko_strip_extend(ko_strip *s, int lo, int hi);
void ko_strip_dataspan(const ko_strip *s, int out[2]);
void ko_strip_fold(ko_strip *s, int nshifts, const float *shifts, const float *amplitudes);
void ko_trace_destroy(ko_trace *t);
void ko_trace_pack(const ko_strip *strip, ko_trace *trace);
void ko_trace_unpack(const ko_trace *trace, ko_strip *strip);
void ko_trace_from_storable(ko_trace *t, const float *packed, int npacked,
                            const int *poffsets, const int *offsets, int nstrips);
void ko_trace_create_simple(ko_trace *t, const float *data, int lo, int hi);
/* mode: 0 = no shift args, 1 = itraceshift, 2 = rtraceshift */
void ko_trace_multiply_add(const ko_trace *t, ko_strip *s, float factor,
                           int mode, int ishift, float rshift);
void ko_trace_multiply_add_nogrow(const ko_trace *t, float *array, int alo, int ahi,
                                  float factor, int mode, int ishift, float rshift);

/* ---------------- orthodrome.f90 / euler.f90 ---------------- */
typedef struct { double lat, lon; } ko_geo;   /* radians */

double ko_d2r_d(double deg);      /* orthodrome.f90:331-338 */
float  ko_d2r_r(float deg);       /* orthodrome.f90:313-320 */
void   ko_azibazi(ko_geo a, ko_geo b, double *azi, double *bazi);
double ko_distance_accurate50m(ko_geo a, ko_geo b);
void   ko_approx_differential_azidist(float dx, float dy, double azi, double bazi, double dist,
                                      double *nazi, double *nbazi, double *ndist);
void   ko_init_euler(float alpha, float beta, float gamma, float mat[3][3]); /* mat[row][col] */

/* ---------------- piecewise_linear_function.f90 ---------------- */
#define KO_PLF_MAX 64
typedef struct { int n; float x[KO_PLF_MAX], y[KO_PLF_MAX]; } ko_plf;  /* n==0: undefined */

void  ko_plf_integrate_and_centroid(const ko_plf *s, float a, float b, float *area, float *centroid);
float ko_plf_integrate(const ko_plf *s, float a, float b);
/* ip: 0 = ip_cos, 1 = ip_linear, 2 = ip_zero_one */
void  ko_plf_taper_array_r(const ko_plf *s, float *array, int lo, int hi, float dx, int ip);
void  ko_plf_taper_array_c(const ko_plf *s, float *array_reim, int lo, int hi, float dx, int ip);
void  ko_discrete_plf_span(const ko_plf *s, float dt, int out[2]);

/* ---------------- discrete_source.f90 + source_*.f90 ---------------- */
typedef struct { float north, east, depth, time, m[6]; } ko_centroid;   /* discrete_source.f90:27-30 */

#define KO_SRC_BILAT 1
#define KO_SRC_CIRCULAR 2
#define KO_SRC_POINT_LP 3
#define KO_SRC_MOMENT_TENSOR 6

typedef struct {
    int sourcetype;
    int nparams;
    float params[24];
    float moment;       /* psm%moment   (parameterized_source.f90:70) */
    float risetime;     /* psm%risetime (parameterized_source.f90:71) */
    float rotmat_rup[3][3], rotmat_slip[3][3];
    int grid_size[3];
    int inited;
} ko_psm;

int ko_psm_nparams(int sourcetype);
/* returns only_moment_changed */
int ko_psm_set(ko_psm *psm, int sourcetype, const float *params);
/* discretise; *out is malloc'ed (caller frees), returns ncentroids (<0 on error) */
int ko_psm_to_tdsm(ko_psm *psm, float shortest_doi, ko_centroid **out);
/* psm%pax, psm%tax of a bilateral source (get_principal_axes) */
void ko_principal_axes_bilat(const float *params, float pax[2], float tax[2]);

/* ---------------- source_eikonal.f90 / source_mt_eikonal.f90 (+ eikonal, heap, geometry) ---------------- */
#define KO_SRC_EIKONAL 4
#define KO_SRC_MT_EIKONAL 5
/* t_crust2x2_1d_profile (crust2x2.f90:45-50): 7 layers + the mantle below (index 7) */
typedef struct { float vp[8], vs[8], rho[8], thickness[7]; } ko_crust_profile;
void ko_crust_profile_averages(const ko_crust_profile *p, float *vvp, float *vvs, float *vrho, float *vthi);
void ko_eikonal_solver_fmm(const float *speed, int nx, int ny, const float origin[2], const float delta[2],
                           const float initialpoint[2], float *times);
/* constraints: ncon half spaces (point, normal) as psm_set_default_constraints / set_source_constraints give them */
int ko_psm_to_tdsm_eikonal(int sourcetype, const float *params, float shortest_doi,
                           const ko_crust_profile *prof_speed, int ncon, const float *con_points,
                           const float *con_normals, ko_centroid **out, float *moment, float *risetime,
                           int grid_size[2]);

/* ---------------- gfdb.f90 (read side, in-memory) ---------------- */
typedef struct {
    float dt, dx, dz, firstx, firstz;
    int nx, nz, ng;
    ko_trace *traces;       /* [(ix*nz+iz)*ng+ig], 0-based; empty trace == not stored */
    ko_trace *scratch;      /* per-thread blend buffers (gfdb.f90:913-931) */
    int nscratch;
} ko_gfdb;

ko_gfdb *ko_gfdb_create(int nx, int nz, int ng, float dt, float dx, float dz, float firstx, float firstz);
void ko_gfdb_destroy(ko_gfdb *db);
/* store one trace given as dense samples [lo..hi]; it is trace_pack'ed (gap rule) like gfdb_build does */
void ko_gfdb_set_trace_dense(ko_gfdb *db, int ix, int iz, int ig, int lo, int hi, const float *data);
/* query packed span of stored trace (1-based ix,iz,ig as in the reference); returns 0 if absent */
int  ko_gfdb_trace_span(const ko_gfdb *db, int ix, int iz, int ig, int span[2]);
/* dense copy of a stored trace over its span (trace_unpack) into out[span2-span1+1] */
void ko_gfdb_trace_unpack(const ko_gfdb *db, int ix, int iz, int ig, float *out);
void ko_gfdb_get_indices(const ko_gfdb *db, float x, float z, int *ix, int *iz);
void ko_gfdb_get_indices_bilin(const ko_gfdb *db, float x, float z, int xus, int zus,
                               int ix[2], int iz[2], float *dix, float *diz);

/* ---------------- comparator.f90 ---------------- */
enum { KO_L2NORM = 1, KO_L1NORM = 2, KO_AMPSPEC_L2NORM = 3, KO_AMPSPEC_L1NORM = 4,
       KO_SCALAR_PRODUCT = 5, KO_PEAK = 6, KO_FLOATING_L2NORM = 7, KO_FLOATING_L1NORM = 8 };

typedef struct {
    float dt, df;
    int span[2], dataspan[2];
    float *array, *array_tapered;    /* index span[0]..span[1]; NULL when unallocated */
    float *spectrum;                 /* (ntrans/2+1) complex, re/im interleaved */
    float *spectrum_filtered;
    float *amp_spectrum, *amp_spectrum_filtered;
    float *array_filtered;
    int nspec;
    int array_tapered_dirty, spectrum_dirty, spectrum_filtered_dirty, array_filtered_dirty;
    float paddingfactor;
    ko_plf taper, filter;
    float factor;
} ko_probe;

void  ko_probe_init(ko_probe *p, float dt);
void  ko_probe_destroy(ko_probe *p);
void  ko_probe_set_array(ko_probe *p, const ko_strip *strip, float factor);
void  ko_probe_shift(ko_probe *p, int ishift);
void  ko_probe_set_taper(ko_probe *p, const ko_plf *plf);
void  ko_probe_set_filter(ko_probe *p, const ko_plf *plf);
float ko_probes_norm(ko_probe *a, ko_probe *b, int method);
float ko_probe_norm(ko_probe *a, int method);
int   ko_probe_get_amp_spectrum(ko_probe *p, int filtered, float *df, float *out, int maxn);
float ko_probes_shake(ko_probe **p, int np, int kind);   /* 1 peak velocity, 2 peak acceleration, 3 Arias intensity */
void  ko_probes_windowed_cross_corr(ko_probe *a, ko_probe *b, int shift_lo, int shift_hi, float *cc);
int   ko_next_power_of_two(int n);
void  ko_allowed_span(const int span[2], int minlength, int out[2]);
/* copy a processed probe array: which 1=plain,2=tapered,3=filtered; returns n, sets *lo */
int   ko_probe_get(ko_probe *p, int which, int *lo, float *out, int maxn);

/* ---------------- receiver.f90 + seismogram.f90 + minimizer_engine.f90 ---------------- */
typedef struct {
    int enabled;
    float dt;
    ko_geo origin;      /* radians */
    float depth;
    int ncomponents;
    int components[5];  /* +-1 a/c, +-2 r/l, +-3 d/u, +-4 n/s, +-5 e/w (receiver.f90:35-48) */
    ko_strip displacement[5];
    float misfits[5], misfits_norm_factors[5];
    ko_probe ref_probes[5], syn_probes[5];
    int floating_shiftrange[2];
    int floating_shift;
} ko_receiver;

typedef struct {
    ko_gfdb *db;             /* borrowed */
    int nreceivers;
    ko_receiver *receivers;
    ko_geo origin;           /* source origin, radians */
    double ref_time;
    ko_psm psm;
    int ncentroids;
    ko_centroid *centroids;
    float effective_dt;
    int interpolate, xundersample, zundersample;
    int misfit_method;
    float misfit;            /* global */
    int nthreads;            /* OpenMP threads over receivers (minimizer_engine.f90:893-903) */
} ko_engine;

ko_engine *ko_engine_create(ko_gfdb *db);
void ko_engine_destroy(ko_engine *e);
/* lat/lon in degrees as in the receivers file (minimizer_engine.f90:236-262); comps e.g. "ned" */
int  ko_engine_set_receivers(ko_engine *e, int n, const double *lat_deg, const double *lon_deg,
                             const float *depth, const char *const *comps);
void ko_engine_switch_receiver(ko_engine *e, int irec1, int state);
void ko_engine_set_source_location(ko_engine *e, float lat_deg, float lon_deg, double ref_time);
void ko_engine_set_effective_dt(ko_engine *e, float dt);
void ko_engine_set_interpolation(ko_engine *e, int bilinear, int xus, int zus);
int  ko_engine_set_source_params(ko_engine *e, int sourcetype, const float *params);
/* bypass the discretiser: give the centroid table, moment and risetime directly */
void ko_engine_set_centroids(ko_engine *e, int n, const ko_centroid *c, float moment, float risetime);
/* reference trace for (irec1, icomp1): samples at indices first..first+n-1 */
void ko_engine_set_reference(ko_engine *e, int irec1, int icomp1, int first, int n, const float *data);
void ko_engine_set_taper(ko_engine *e, int irec1, int npts, const float *x, const float *y);
void ko_engine_set_filter(ko_engine *e, int irec1, int npts, const float *x, const float *y);
void ko_engine_set_misfit_method(ko_engine *e, int method);
void ko_engine_set_synthetics_factor(ko_engine *e, float f);
void ko_engine_set_floating_shiftrange(ko_engine *e, int irec1, int lo, int hi);
int ko_engine_get_floating_shift(ko_engine *e, int irec1);
int ko_engine_get_reference(ko_engine *e, int irec1, int icomp1, int which, int *lo, float *out, int maxn);
void ko_engine_probe_spans(ko_engine *e, int irec1, int icomp1, int which, int out[4]);
void ko_engine_shift_ref_seismogram(ko_engine *e, int irec1, int ishift);
int ko_engine_autoshift_ref_seismogram(ko_engine *e, int irec1, int lo, int hi);
int ko_engine_cross_correlations(ko_engine *e, int irec1, int lo, int hi, float *cc);
int ko_engine_amp_spectrum(ko_engine *e, int irec1, int icomp1, int synthetic, int filtered, float *df, float *out, int maxn);
int ko_engine_shake(ko_engine *e, int differentiate, float *out);   /* get_peak_amplitudes (1, 2) / get_arias_intensities (0) */
void ko_engine_set_nthreads(ko_engine *e, int n);
/* precision of the comparator's transforms: 64 (default: an exact DFT rounded once) or 32 (a textbook fp32 radix-2 FFT, the second
 * checker of the spectral tolerances); process-wide, set before the probes are evaluated */
void ko_set_fft_precision(int bits);
int ko_get_fft_precision(void);
/* the three private engine steps (minimizer_engine.f90:885-945) */
void ko_engine_calculate_seismograms(ko_engine *e);
void ko_engine_scale_seismograms(ko_engine *e);
void ko_engine_calculate_misfits(ko_engine *e);
/* update_misfits + get_misfits (minimizer_engine.f90:1130-1172): fills m,n pairs for enabled receivers */
int  ko_engine_get_misfits(ko_engine *e, float *m, float *n, int maxn);
float ko_engine_get_global_misfit(ko_engine *e);
/* raw displacement strip of (irec1, icomp1) after calculate_seismograms */
int  ko_engine_get_displacement(ko_engine *e, int irec1, int icomp1, int *lo, float *out, int maxn);
/* synthetic probe contents after scale_seismograms: which 1 plain 2 tapered 3 filtered */
int  ko_engine_get_synthetic(ko_engine *e, int irec1, int icomp1, int which, int *lo, float *out, int maxn);
/* per (receiver, centroid) geometry record, 20 x 4 bytes, see ko_engine.c */
void ko_engine_centroid_geometry(ko_engine *e, int irec1, int icent0, void *out20);
/* per receiver geometry as make_seismogram computes it (seismogram.f90:99-100) */
void ko_engine_receiver_geometry(ko_engine *e, int irec1, double *azi, double *bazi, double *dist);

#ifdef __cplusplus
}
#endif
#endif
