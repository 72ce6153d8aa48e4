/*
 * kiwi_hip.h -- C-ABI of the MI355X forward-modelling + misfit engine for Kiwi's inner
 * inversion loop (trial source -> synthetic seismograms -> misfits).
 *
 * This is the drop-in boundary: plain C types only, callable from Fortran through
 * iso_c_binding (kiwi_amd/fortran/kiwi_hip_binding.f90), from Python through ctypes
 * (kiwi_amd/engine.py) or from C/C++.  Every function returns 0 on success and a
 * non-zero code on failure; the message is retrieved with kiwi_hip_last_error() and maps
 * onto the reference's recoverable-error convention ("<cmd>: nok >" + message,
 * minimizer.f90:1689-1696).  No function aborts the process.  All arrays are caller-owned,
 * contiguous, and copied before the call returns.  Indices irec/icomp are 1-based like
 * the reference's wire protocol (switch_receiver, minimizer.f90:273-312).
 *
 * The three private engine steps this library replaces are
 *     calculate_seismograms()  minimizer_engine.f90:885-907   (-> make_seismogram, seismogram.f90:36)
 *     scale_seismograms()      minimizer_engine.f90:909-921   (-> receiver.f90:853)
 *     calculate_misfits()      minimizer_engine.f90:924-945   (-> receiver.f90:407, comparator.f90:911,954)
 * batched over many trial sources per call (kiwi_hip_eval); the setters mirror the
 * engine's public state setters, cited one by one below.
 *
 * Sample index convention: all 'first' arguments are indices in the reference's strip
 * index space (t_strip lower bounds, sparse_trace.f90:29-33): a Green's function trace
 * sample j added with shift s lands on seismogram sample j+s (sparse_trace.f90:605), and a
 * reference seismogram read from a file starting at time t0 has first = nint((t0 -
 * reftime)/dt) + 1 (receiver.f90:834-851).
 */
#ifndef KIWI_HIP_H
#define KIWI_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kiwi_hip_ctx kiwi_hip_ctx;

/* misfit method ids = comparator.f90:35-42 */
#define KIWI_L2NORM 1
#define KIWI_L1NORM 2
#define KIWI_AMPSPEC_L2NORM 3
#define KIWI_AMPSPEC_L1NORM 4
#define KIWI_SCALAR_PRODUCT 5
#define KIWI_PEAK 6
#define KIWI_FLOATING_L2NORM 7
#define KIWI_FLOATING_L1NORM 8

/* source type ids = parameterized_source.f90:45-50 */
#define KIWI_SRC_BILAT 1
#define KIWI_SRC_CIRCULAR 2
#define KIWI_SRC_POINT_LP 3
#define KIWI_SRC_EIKONAL 4
#define KIWI_SRC_MT_EIKONAL 5
#define KIWI_SRC_MOMENT_TENSOR 6

/* arithmetic contract of the accumulate kernels (kiwi_hip_set_arithmetic).
 * EXACT (default): every fp32 multiply and add of the superposition is rounded on its own, in the reference's order
 *   (gfdb.f90:944-949, sparse_trace.f90:684-703, seismogram.f90:171-250 on an x86-64 host without fused operations):
 *   synthetics and misfits are bit-identical to the CPU restatement of the reference given equal geometry records.
 * FUSED: the same operations in the same order, each multiply contracted with the add that consumes it into one fused
 *   multiply-add (one rounding instead of two; half the vector instructions): tolerance class -- misfits within 1e-6 of the
 *   slot's norm factor, synthetics within 2e-6 of the trace maximum (BASELINE.json north_star: "misfits within 1e-6 relative").
 * Geometry (rows, sample shifts, weights), the comparator's fp64 sums and the host discretisers are the same in both. */
#define KIWI_ARITH_EXACT 0
#define KIWI_ARITH_FUSED 1

/* ---- lifetime: program start / cleanup_minimizer (minimizer_engine.f90:1057-1067) ---- */
int kiwi_hip_init(int device, kiwi_hip_ctx **ctx);
/* SURVEY 8b's `kiwi_hip_init(int ndev_wanted, void** ctx)`: ONE context over ndev_wanted devices of this process
 * (<= 0: every visible device; more than are visible is an error unless KIWI_HIP_MULTI_OVERSUBSCRIBE=1 stacks the contexts
 * on the devices there are -- for tests on a one-GPU box).  The returned context is the first device's and owns the others:
 * every setter called on it is repeated on them (Green's function tensor, receivers, references replicated: SURVEY 8e), and
 * kiwi_hip_misfits_for_params cuts its trial list into contiguous shards in list order (the order of Source.grid,
 * python/tunguska/source.py:119-164), one per device, each evaluated on its device by a thread of its own straight into its slice
 * of the caller's arrays -- no collective.  Everything else (kiwi_hip_eval, getters, kiwi_hip_minimize_lm) works on the first
 * device as with kiwi_hip_init.  Counterpart for the Fortran host of the process pool of python/tunguska/seismosizer.py:785-827;
 * results do not depend on the number of devices (a source's evaluation does not depend on its batch). */
int kiwi_hip_init_multi(int ndev_wanted, kiwi_hip_ctx **ctx);
int kiwi_hip_ndevices(kiwi_hip_ctx *ctx, int *n);
/* KIWI_ARITH_EXACT / KIWI_ARITH_FUSED (above); takes effect at the next kiwi_hip_eval.  The environment variable
 * KIWI_HIP_ARITH=exact|fused sets the initial value of every context (for the unmodified Fortran protocol host). */
int kiwi_hip_set_arithmetic(kiwi_hip_ctx *ctx, int mode);
int kiwi_hip_get_arithmetic(kiwi_hip_ctx *ctx, int *mode);
int kiwi_hip_destroy(kiwi_hip_ctx *ctx);
/* copies the last error message (NUL terminated, truncated to buflen); ctx may be NULL for init errors */
int kiwi_hip_last_error(kiwi_hip_ctx *ctx, char *buf, int buflen);

/* ---- set_database (minimizer_engine.f90:114-139; gfdb.f90:163-264) ----
 * One-time dense upload replacing the lazy chunk cache (gfdb.f90:952-1031).
 * G[((ix*nz+iz)*ng+ig)*L + l], l < nsamp[...] : samples of trace (ix,iz,ig) starting at
 * strip index first[...] (== trace%span(1), sparse_trace.f90:46); interior gaps are zeros;
 * samples at l >= nsamp are ignored (the last valid sample is the repeated end value,
 * sparse_trace.f90:696-703).  nsamp == 0 marks a trace that is not stored (gfdb.f90:1003). */
int kiwi_hip_set_gfdb(kiwi_hip_ctx *ctx, int nx, int nz, int ng, int L,
                      float dt, float dx, float dz, float firstx, float firstz,
                      const float *G, const int *first, const int *nsamp);

/* set_local_interpolation + set_spacial_undersampling (minimizer_engine.f90:141-163; minimizer.f90:155-207) */
int kiwi_hip_set_interp(kiwi_hip_ctx *ctx, int bilinear, int xundersample, int zundersample);

/* set_effective_dt (minimizer_engine.f90:612-620): shortest duration of interest for the discretisers */
int kiwi_hip_set_effective_dt(kiwi_hip_ctx *ctx, float effective_dt);

/* set_source_location lat lon reftime (minimizer.f90:485-517; degrees, parsed as default real) */
int kiwi_hip_set_source_location(kiwi_hip_ctx *ctx, float lat_deg, float lon_deg, double ref_time);

/* set_receivers (minimizer_engine.f90:165-286): per receiver lat lon [depth] components;
 * components is a string over "acrlduesnw" (receiver.f90:35-56), at most 5, no axis twice */
int kiwi_hip_set_receivers(kiwi_hip_ctx *ctx, int nrec, const double *lat_deg, const double *lon_deg,
                           const float *depth, const char *const *components);
/* switch_receiver (minimizer_engine.f90:288-309) */
int kiwi_hip_switch_receiver(kiwi_hip_ctx *ctx, int irec, int enabled);

/* set_ref_seismograms (minimizer_engine.f90:313-352; receiver.f90:746-851): one trace per receiver component */
int kiwi_hip_set_reference(kiwi_hip_ctx *ctx, int irec, int icomp, int first, int n, const float *data);
/* set_misfit_taper / set_misfit_filter (minimizer_engine.f90:632-698; minimizer.f90:875-1016; receiver.f90:355-389):
 * piecewise linear function control points; npts == 0 removes it; for the filter irec == 0 means every receiver (:646-661) */
int kiwi_hip_set_taper(kiwi_hip_ctx *ctx, int irec, int npts, const float *x, const float *y);
int kiwi_hip_set_filter(kiwi_hip_ctx *ctx, int irec, int npts, const float *x, const float *y);
/* set_misfit_method (minimizer_engine.f90:622-630) */
int kiwi_hip_set_misfit_method(kiwi_hip_ctx *ctx, int method);
/* set_floating_shiftrange ireceiver min-shift max-shift (minimizer_engine.f90:421-451; minimizer.f90:388-419): range in
 * seconds for floating_l1norm / floating_l2norm (receiver.f90:439-510), ireceiver 0 = all; and get_floating_shifts
 * (minimizer_engine.f90:1095-1128): the winning shift in seconds per source and ENABLED receiver, shifts[nsrc][n_enabled] */
int kiwi_hip_set_floating_shiftrange(kiwi_hip_ctx *ctx, int irec, float min_shift, float max_shift);
int kiwi_hip_get_floating_shifts(kiwi_hip_ctx *ctx, int isrc0, int nsrc, float *shifts);
/* shift_ref_seismogram ireceiver shift (minimizer_engine.f90:354-378): move a receiver's reference traces by nint(shift/dt)
 * samples.  autoshift_ref_seismogram ireceiver min-shift max-shift (:380-419; receiver.f90:816-832;
 * comparator.f90:1061-1090): cross-correlate the tapered synthetics of uploaded source `isrc` with the references over
 * the integer shifts of the range, move the references of receiver `irec` (0 = all) to the best shift and return the
 * shifts applied in seconds (shifts[nrec] for irec == 0, else shifts[1]).  Setup-time operations, computed on the host. */
int kiwi_hip_shift_ref_seismogram(kiwi_hip_ctx *ctx, int irec, float shift);
int kiwi_hip_autoshift_ref_seismogram(kiwi_hip_ctx *ctx, int irec, float min_shift, float max_shift, int isrc, float *shifts);
/* set_synthetics_factor (minimizer_engine.f90:700-727; receiver.f90:391-405) */
int kiwi_hip_set_synthetics_factor(kiwi_hip_ctx *ctx, float factor);

/* ---- trial sources ----
 * psm_set + psm_to_tdsm on the host (source_all.f90:216-261,431-465): number of parameters
 * of a source type (<0: unsupported), and one discretisation into a centroid table
 * cent[ncent][10] = north east depth time mxx myy mzz mxy mxz myz (discrete_source.f90:27-30). */
int kiwi_hip_source_nparams(int sourcetype);
int kiwi_hip_discretize(int sourcetype, const float *params, int nparams, float effective_dt,
                        float *cent, int maxcent, int *ncent, float *moment, float *risetime);

/* ---- variable-rupture-speed sources `eikonal` (type 4, 15 params) and `mt_eikonal` (type 5, 20 params)
 * (source_eikonal.f90, source_mt_eikonal.f90).  A crust profile is 31 floats: vp[8] vs[8] rho[8] thickness[7]
 * (t_crust2x2_1d_profile, crust2x2.f90:45-50; layer 8 = below the crust).  The CRUST2.0 tables stay with the
 * caller, which does the two look-ups set_source_location triggers in the reference:
 *   rupture_profile = crust2x2_get_profile(psm%origin)       -- rupture speeds, source_eikonal.f90:472
 *                     (the reference passes the origin in RADIANS there; a drop-in caller does the same)
 *   origin_profile  = crust2x2_get_profile(r2d(psm%origin))  -- crustal thickness, parameterized_source.f90:215
 * set_source_crust and set_source_crustal_thickness_limit (minimizer_engine.f90:479-486) both re-install the
 * default constraints (surface at 1500 m, bottom of the crust; parameterized_source.f90:127-145);
 * set_source_constraints (minimizer_engine.f90:469-477) replaces them: points[n][3], normals[n][3] (ned). */
int kiwi_hip_set_source_crust(kiwi_hip_ctx *ctx, const float *rupture_profile, const float *origin_profile);
int kiwi_hip_set_source_crustal_thickness_limit(kiwi_hip_ctx *ctx, float limit);
int kiwi_hip_get_source_crustal_thickness(kiwi_hip_ctx *ctx, float *thickness);   /* minimizer_engine.f90:488-498 */
int kiwi_hip_set_source_constraints(kiwi_hip_ctx *ctx, int n, const float *points, const float *normals);
/* stateless psm_set + psm_to_tdsm of the two types (needs no GPU); returns 5 for "Empty rupture area"
 * (source_eikonal.f90:284), 6 for a nucleation point outside of the rupture region (:427) */
int kiwi_hip_discretize_eikonal(int sourcetype, const float *params, int nparams, float effective_dt,
                                const float *rupture_profile, int ncon, const float *points, const float *normals,
                                float *cent, int maxcent, int *ncent, float *moment, float *risetime);

/* upload a batch of discretised trial sources: cent_ofs[nsrc+1] row offsets into cent[][10];
 * moment / risetime = psm%moment / psm%risetime per source (parameterized_source.f90:70-71) */
int kiwi_hip_set_sources(kiwi_hip_ctx *ctx, int nsrc, const int *cent_ofs, const float *cent,
                         const float *moment, const float *risetime);
/* the centroid table of uploaded source isrc as the engine holds it (output_source_model writes it to
 * <base>-dsm.table, minimizer_engine.f90:947-977); cent == NULL or maxcent <= 0 only returns the count */
int kiwi_hip_get_source_centroids(kiwi_hip_ctx *ctx, int isrc, int maxcent, int *ncent, float *cent);
/* set_source_params for a whole batch (minimizer_engine.f90:500-523): params[nsrc][nparams] in
 * wire order; discretised on the host with the current effective dt, then uploaded.
 * A trial source the discretiser rejects ("Empty rupture area", source_eikonal.f90:286; nucleation point outside of the
 * rupture region, :428) does not fail the batch: it is recorded (kiwi_hip_get_source_status) and skipped, its misfits,
 * norm factors and global misfit read as zeros -- python/tunguska/seismosizer.py:703-720 (`failings`).  The call
 * returns non-zero only when NO source of the batch could be discretised (a batch of one: the reference's
 * `set_source_params: nok > Empty rupture area`); the statuses stay readable then too. */
int kiwi_hip_set_sources_params(kiwi_hip_ctx *ctx, int sourcetype, int nsrc, const float *params);
/* status[nsrc] of uploaded sources isrc0 ..: 0 discretised, 5 "Empty rupture area", 6 "position of nucleation point is
 * outside of rupture region" (the codes kiwi_hip_discretize_eikonal returns); message text of a code */
int kiwi_hip_get_source_status(kiwi_hip_ctx *ctx, int isrc0, int nsrc, int *status);
int kiwi_hip_source_status_message(int code, char *buf, int buflen);

/* make_misfits_for_sources for a whole trial list in ONE call (python/tunguska/seismosizer.py:682-722), host and device
 * overlapped: the list is cut into pieces of `piece` sources (<= 0: 128 for the eikonal types, 2048 otherwise); while the
 * device evaluates one piece a second host thread discretises the next.  Per piece this IS kiwi_hip_set_sources_params +
 * kiwi_hip_eval + kiwi_hip_get_misfits + kiwi_hip_get_source_status, so misfit[nsrc][nmis], norm[nsrc][nmis], global[nsrc]
 * and status[nsrc] (any may be NULL) are bit for bit what those calls return for any piece size; a piece none of whose
 * sources could be discretised is all failings (zeros), not an error.  Pieces are taken from the end of the list, so the
 * context is left with its HEAD (sources 0 .. piece - 1, evaluated), as after kiwi_hip_set_sources_params + kiwi_hip_eval
 * of those.  (Eikonal types, lists of two pieces or more: the last piece of the list -- the first worked on, whose
 * discretisation nothing hides -- is taken as an eighth, an eighth, a quarter and half of it, and the discretiser runs up to
 * three pieces ahead of the device.)  For the eikonal source types the host discretiser (a fast-marching solve per trial source, eikonal.f90:29-199)
 * costs as much as the device evaluation; overlapped, a sweep runs at the slower of the two instead of their sum. */
int kiwi_hip_misfits_for_params(kiwi_hip_ctx *ctx, int sourcetype, int nsrc, const float *params, int piece,
                                float *misfit, float *norm, float *global, int *status);
/* CPUs the host side may keep busy: allowed hardware threads cut to the cgroup CPU quota (the discretiser's thread count) */
int kiwi_hip_effective_cpus(void);
/* The eikonal discretiser keeps its fast-marching solves (eikonal.f90:29-199) by their complete inputs -- speed grid, grid
 * spacing, start cell (source_mt_eikonal.f90:467-519 builds them in rupture coordinates) -- and returns the stored arrival
 * times when ALL of them recur (hash, then comparison in full: bit-identical centroid tables).  North / east / time shifts
 * and moment-tensor changes of a rupture leave the inputs alone: a location grid at fixed depth costs one solve.  Counters
 * since the library was loaded (either pointer may be NULL); reset != 0 clears them, reset & 2 also drops the stored solves.
 * Process-wide, shared by all contexts.
 * KIWI_HIP_EIK_CACHE=0 in the environment switches the cache off. */
int kiwi_hip_eikonal_cache_stats(long long *hits, long long *misses, int reset);
/* The fast-marching solve on its own (eikonal_solver_fmm, eikonal.f90:29-199; needs no GPU): speed[ny][nx] and times[ny][nx]
 * with x fastest, origin / delta / start as there.  `discard`: nodes of exactly this speed may be left undone once every other
 * node is accepted (what the discretiser passes for the points outside of the rupture; NaN = solve all).  plain != 0 runs the
 * reference's statements one by one, 0 the layout-optimised march the discretiser uses (kiwi_host_fmm.hpp) -- same bits;
 * *fallbacks (may be NULL) = how often, since the library was loaded, the optimised march handed a solve to the plain one. */
int kiwi_hip_fast_marching(const float *speed, int nx, int ny, const float *origin, const float *delta, const float *start,
                           float discard, int plain, float *times, long long *fallbacks);

/* minimize_lm (minimizer_engine.f90:728-874; sminpack/lmdif.f in fp32 with the reference's settings: ftol = xtol =
 * sqrt(spmpar(1)), gtol = 0, maxfev = 500 (n + 1), mode 2 with diag = 1, factor 0.01) over the parameters with
 * mask[i] != 0 (set_source_params_mask), starting at params[nparams].  mins / maxs: limits of the FREE parameters in
 * physical units or both NULL (set_source_subparams_limits, :580-611; :820-842).  Each forward-difference Jacobian is
 * ONE batched evaluation of n sources.  On return params = the source of the LAST forward step -- what the reference
 * leaves in psm and reports through get_source_subparams --, misfit = its global misfit, iterations = forward steps,
 * info as lmdif (8 reported as 4, :796); best (may be NULL) = lmdif's accepted iterate in physical units. */
int kiwi_hip_minimize_lm(kiwi_hip_ctx *ctx, int sourcetype, float *params, const int *mask, const float *mins,
                         const float *maxs, int *info, int *iterations, float *misfit, float *best);

/* The optimiser underneath, usable with any residual function: sminpack/lmdif.f in fp32, the n forward differences of
 * a Jacobian requested as ONE call.  fcn gets k points xs[k][n] (it may modify them in place, as lm_forward_step clamps
 * its argument) and fills fv[k][m]; a negative return aborts and becomes *info.  No device involved. */
typedef int (*kiwi_hip_residual_fn)(void *user, int k, int m, int n, float *xs, float *fv);
int kiwi_hip_lmdif(kiwi_hip_residual_fn fcn, void *user, int m, int n, float *x, float *fvec, float ftol, float xtol,
                   float gtol, int maxfev, float epsfcn, float *diag, int mode, float factor, int *info, int *nfev);

/* ---- the hot path: calculate_seismograms + scale_seismograms + calculate_misfits for
 * sources [isrc0, isrc0+nsrc) of the uploaded batch.  Asynchronous on the context's HIP
 * stream; results stay on the device until fetched. */
int kiwi_hip_eval(kiwi_hip_ctx *ctx, int isrc0, int nsrc);
int kiwi_hip_sync(kiwi_hip_ctx *ctx);

/* get_misfits (minimizer_engine.f90:1130-1172): nmis = sum of components over ENABLED receivers,
 * receiver-major, component-minor */
int kiwi_hip_nmisfits(kiwi_hip_ctx *ctx, int *nmis);
/* misfit[nsrc][nmis], norm[nsrc][nmis] (misfits_norm_factors), global[nsrc] =
 * sqrt(sum m^2)/sqrt(sum n^2) (minimizer_engine.f90:939-942); any pointer may be NULL.  Synchronises. */
int kiwi_hip_get_misfits(kiwi_hip_ctx *ctx, int isrc0, int nsrc, float *misfit, float *norm, float *global);
/* The global misfits of evaluated sources isrc0 .. isrc0 + nsrc - 1 WHERE THEY LIE: a device pointer (fp32, contiguous) on the
 * context's device, valid until the next kiwi_hip_eval / kiwi_hip_set_sources* on the context.  For the multi-GPU exchange
 * (one all-gather of per-source misfit scalars): a collective library takes the shard straight from device memory,
 * no staging through the host.  Synchronises the context's stream (the values are final when the call returns). */
int kiwi_hip_get_global_misfits_device(kiwi_hip_ctx *ctx, int isrc0, int nsrc, const float **device_ptr);

/* output_seismograms (minimizer_engine.f90:980-1010): synthetic of one source of the LAST
 * kiwi_hip_eval range, over the receiver's misfit window.  which: 1 plain (scaled by moment,
 * rise-time folded), 2 tapered.  Returns first sample index and count. */
/* keep the processed synthetics of every evaluated chunk on the device (0 off, 1 plain, 2 tapered) so
 * that kiwi_hip_get_synthetics copies them instead of re-evaluating the source */
int kiwi_hip_set_keep_synthetics(kiwi_hip_ctx *ctx, int which);
int kiwi_hip_get_synthetics(kiwi_hip_ctx *ctx, int isrc, int irec, int icomp, int which,
                            int *first, int *n, float *out, int maxn);
/* the reference probe the same way (output_seismograms ... references plain|tapered|filtered, receiver.f90:618-680):
 * plain = the data as set, tapered / filtered = over the comparator window */
int kiwi_hip_get_reference(kiwi_hip_ctx *ctx, int irec, int icomp, int which, int *first, int *n, float *out, int maxn);

/* ---- measurement / inspection ---- */
/* output_seismogram_spectra (minimizer_engine.f90:1012-1039; probe_get_amp_spectrum, comparator.f90:333-354): amplitude
 * spectrum of the reference (which_probe 0) or of the synthetic of source isrc (1) of one receiver component:
 * |r2c| of the tapered window, n = ntrans / 2 + 1 bins at spacing df, transform length as the spectral comparator sizes
 * the reference / synthetic pair (in the reference it follows the probes' span history); filtered != 0: times the
 * frequency filter where the receiver has one.  Needs references and tapers. */
int kiwi_hip_get_amp_spectrum(kiwi_hip_ctx *ctx, int isrc, int irec, int icomp, int which_probe, int filtered,
                              float *df, int *n, float *out, int maxn);
/* get_principal_axes (minimizer_engine.f90:1248-1258): P and T axis (azimuth, polar angle in degrees, lower hemisphere)
 * of a bilateral source as psm_update_dep_params_bilat derives them (source_bilat.f90:216-239); the reference sets them
 * for no other source type: returns -1 for those.  Host only. */
int kiwi_hip_principal_axes(int sourcetype, const float *params, float *pax, float *tax);
/* output_cross_correlations (minimizer_engine.f90:1283-1306; receiver.f90:597-616; comparator.f90:1061-1090): for one
 * receiver cc[component][shift] = scalar product of the tapered synthetic of source isrc with the reference pulled
 * through its fixed taper, for the integer shifts nint(min/dt) .. nint(max/dt) (first_shift, nshift; nshift = 0 for a
 * disabled receiver).  Set-up-time helper like autoshift: evaluated on the host from one device evaluation. */
int kiwi_hip_get_cross_correlations(kiwi_hip_ctx *ctx, int isrc, int irec, float min_shift, float max_shift,
                                    int *first_shift, int *nshift, float *cc, int maxn);
/* get_peak_amplitudes (minimizer_engine.f90:1174-1212; receiver.f90:544-574; comparator.f90:519-589): per ENABLED
 * receiver the maximum over the misfit window (taper span; without taper the union of the synthetic strips' data spans)
 * of the vector norm of the once (differentiate = 1, velocity) or twice (2, acceleration) differenced synthetics of
 * uploaded source isrc -- vertical + the horizontal pair a/c + r/l or else n/s + e/w, whatever of them the receiver has.
 * get_arias_intensities (:1214-1246; receiver.f90:576-594; comparator.f90:591-625) likewise.  Tapered synthetics where
 * the receiver has a taper; not available while a misfit filter is set.  Without a taper the span is that of THIS
 * source's strips (the reference's strips remember earlier sources).  out[number of enabled receivers]. */
int kiwi_hip_get_peak_amplitudes(kiwi_hip_ctx *ctx, int isrc, int differentiate, float *out);
int kiwi_hip_get_arias_intensities(kiwi_hip_ctx *ctx, int isrc, float *out);

/* HIP-event durations [ms] of the last kiwi_hip_eval on the context stream:
 * ms[0] geometry kernel, ms[1] accumulate kernel(s), ms[2] misfit kernels, ms[3] whole eval;
 * launches[0..2] = number of launches of each in that eval.  Synchronises. */
int kiwi_hip_get_kernel_ms(kiwi_hip_ctx *ctx, float ms[4], int launches[3]);
/* per (source, receiver, centroid) geometry record of the last eval, 20 floats/ints each
 * (layout in kiwi_amd/csrc/kiwi_kernels.hpp); for parity tests */
int kiwi_hip_get_geometry(kiwi_hip_ctx *ctx, int isrc, int irec, int maxcent, int *ncent, void *records);
/* receiver constants computed at set time: azimuth, back-azimuth [rad], distance [m]
 * (seismogram.f90:99-100), as output_distances prints them (minimizer.f90:1404-1440) */
int kiwi_hip_get_receiver_geometry(kiwi_hip_ctx *ctx, int irec, double *azi, double *bazi, double *dist);
/* device memory currently held [bytes] */
int kiwi_hip_get_device_bytes(kiwi_hip_ctx *ctx, long long *bytes);
/* the extra compiler flags the library was built with (`make EXTRA=...`; empty for the default build): profiles are matched to a
 * build by its kernel sources AND these */
int kiwi_hip_build_flags(char *buf, int buflen);
/* diagnostics: the rate [GB/s, 1e9 bytes] of a pure read of `bytes` bytes of device memory (16 bytes per lane, contiguous
 * slices; choose bytes >> 256 MiB Infinity Cache), `reps` passes timed with HIP events on the context's stream -- the ceiling
 * bench.py holds the accumulate kernels' HBM-regime figure against, measured in the same run */
int kiwi_hip_measure_read_bandwidth(kiwi_hip_ctx *ctx, long long bytes, int reps, double *gbs);

#ifdef __cplusplus
}
#endif
#endif
