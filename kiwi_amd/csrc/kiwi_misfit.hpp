// kiwi_misfit.hpp -- comparator kernels: scale by moment, rise-time fold (receiver.f90:853-904), taper, time-domain norms with
// fp64 accumulation (comparator.f90:619-697), amplitude-spectrum norms and frequency filter (comparator.f90:861-886,1186-1263)
// with in-LDS transforms, floating norms (receiver.f90:439-510), global misfit (minimizer_engine.f90:924-945), shake-map
// diagnostics.  Built with -ffp-contract=off (the in-LDS transforms use explicit fused multiply-adds where stated).
#pragma once
#include "kiwi_common.hpp"

namespace kiwi {

// ------------------------------------------------------------------------------------------------
// get_peak_amplitudes / get_arias_intensities: block per enabled receiver over the processed synthetics of ONE source
// (moment, fold and taper applied by misfit_kernel).  kind 1: max_vecnorm_d1 (comparator.f90:519-553), 2: max_vecnorm_d2
// (:555-589), 3: arias_intensity (:591-625); fp32 differences, fp64 squares and sums as there.
__global__ __launch_bounds__(256) void shake_kernel(const float *__restrict__ proc, const CompDev *__restrict__ comps,
                                                    const ShakeRec *__restrict__ recs, const int *__restrict__ spansrc,
                                                    int kind, float dt, float factor, int fold_grow, float *__restrict__ out)
{
    const ShakeRec sr = recs[blockIdx.x];
    __shared__ double red[256];
    if (sr.np == 0) { if (threadIdx.x == 0) out[blockIdx.x] = 0.f; return; }
    const CompDev c0 = comps[sr.slot[0]];
    int i_lo = 0, i_hi = c0.wlen - 1;
    if (sr.untapered) {
        int lo = 0x7fffffff, hi = -0x7fffffff;
        for (int k = 0; k < sr.np; k++) {
            int s0, s1;
            strip_span(spansrc + (size_t)sr.rec * kSpanInts, comps[sr.slot[k]].spankind, s0, s1);
            if (s1 >= s0) { lo = min(lo, s0 - fold_grow); hi = max(hi, s1 + (fold_grow ? fold_grow + 1 : 0)); }
        }
        i_lo = max(lo - c0.w0, 0); i_hi = min(hi - c0.w0, c0.wlen - 1);
        if (hi < lo) { if (threadIdx.x == 0) out[blockIdx.x] = 0.f; return; }
    }
    const float *x[3];
    for (int k = 0; k < 3; k++) { const CompDev cd = comps[sr.slot[k < sr.np ? k : 0]]; x[k] = proc + cd.synofs + cd.halo; }
    const int n = i_hi - i_lo + 1, m = kind == 1 ? n - 1 : n - 2;
    const double f2 = (double)(factor * factor);
    double acc = kind == 3 ? 0.0 : -HUGE_VAL;
    for (int i = threadIdx.x; i < m; i += 256) {
        double v = 0.0;
        for (int k = 0; k < sr.np; k++) {
            const float *a = x[k] + i_lo + i;
            const float d = kind == 1 ? a[0] - a[1] : a[0] - 2.0f * a[1] + a[2];
            const double t = f2 * ((double)d * (double)d);
            v = k == 0 ? t : v + t;
        }
        if (kind == 3) acc += v; else acc = fmax(acc, v);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] = kind == 3 ? red[threadIdx.x] + red[threadIdx.x + st] : fmax(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float pi = 3.14159265358979f;                         // constants.f90:21
        float res;
        if (kind == 1) res = (float)(sqrt(red[0]) / (double)dt);
        else if (kind == 2) res = (float)(sqrt(red[0]) / (double)(dt * dt));
        else res = (float)((double)(pi / (2.f * 9.81f) * dt) * red[0] / (double)(dt * dt));
        out[blockIdx.x] = res;
    }
}

// ------------------------------------------------------------------------------------------------
// misfit

constexpr int kMaxFold = 129;
constexpr int kMaxFloatShifts = 513;   // integer shifts tried by a floating norm at most

struct MisfitParams {
    int method;          // comparator.f90:35-40 (time-domain ones)
    float dt;
    float syn_factor;    // b%factor (receiver.f90:391-405); a%factor == 1
    int nmis;
    int isrc0;
    int write_tapered;   // keep scaled+folded (+tapered) synthetics for get_synthetics
    int fft_mode;        // bit0: write the tapered synthetic zero-padded to ntrans into fftbuf, no norm; bit1: every slot (spectral norms);
                         // bit2: the rows are not needed (spec_fft_filter_norm_kernel reads the synthetics)
    int chunk_nsrc;      // sources in this launch (row stride of the FFT groups)
    int skip_norm;       // floating norms: only produce the tapered synthetics (vt_out), norms follow in floating_norm_kernel
    int lds_lo = 1, lds_hi = 0;      // transform lengths the in-LDS kernels take (SpecParams); bit2 of fft_mode applies to those pairs only
};

// rise-time fold of a source (receiver.f90:868-886): weights, integer shifts and fractions of the 1 + 2 nint(rise / 2 dt) taps
__device__ __forceinline__ int fold_setup(float rise, float dt, float *fw, int *fs, float *fr)
{
    int n = 0;
    if (rise > 0.f) {
        const float rr0 = -rise / 2.f, rr1 = +rise / 2.f;
        n = 1 + 2 * (int)roundf(0.5f * rise / dt);
        if (n > kMaxFold) n = kMaxFold;       // guarded on the host (set_sources)
        float sum = 0.f;
        for (int is = 1; is <= n; is++) {
            const float ts = ((float)(is - 1) - 0.5f * (float)(n - 1)) * dt;
            const float lo = fmaxf(rr0, ts - dt / 2.f), hi = fminf(rr1, ts + dt / 2.f);
            fw[is - 1] = fmaxf(0.f, hi - lo);
            const float sh = ts / dt;
            const float fl = floorf(sh);
            fs[is - 1] = (int)fl;
            fr[is - 1] = sh - (float)(int)fl;
            sum = sum + fw[is - 1];
        }
        for (int i = 0; i < n; i++) fw[i] = fw[i] / sum;
    }
    return n;
}

// window sample i of a synthetic, folded (strip_fold, sparse_trace.f90:379-402) and scaled by the moment
// (probe_set_array(..., factor_=moment), comparator.f90:264); sy[i] = plain synthetic at window sample i
__device__ __forceinline__ float folded_scaled_sample(const float *__restrict__ sy, int i, int nf, const float *fw, const int *fs,
                                                      const float *fr, float mom)
{
    float v;
    if (nf > 0) {
        v = 0.f;
        for (int k = 0; k < nf; k++) {
            float wr = fr[k];
            float wl = 1.f - wr;
            wr = wr * fw[k]; wl = wl * fw[k];
            v = v + wl * sy[i - fs[k]];
            v = v + wr * sy[i - fs[k] - 1];
        }
    } else {
        v = sy[i];
    }
    return v * mom;
}

// Data span of the synthetic PROBE of every (chunk source, un-tapered slot) as a fresh engine sets it (receiver.f90:853-904):
// the strip's extent [n0, n1] (geometry_kernel's union over the centroids; one zero at sample 0 when no centroid reached it) and,
// with a rise time, what strip_fold makes of it (sparse_trace.f90:379-402): the strip is cut to its DATA span first --
// strip_dataspan, :347-376: from the first sample that is not zero to the first sample of the trailing run of equal values (a
// Green's function with a static end value repeats it behind every centroid's trace) --, folded, and only ever grows:
// [min(n0, d1 - h), max(n1, d2 + h + 1)], h = half width of the taps.  The cut depends on the DATA, so this runs behind the accumulate
// kernel.  synspan[(s * nmis + m) * 2 ...] = (first, last); slots with a taper are left alone.
__global__ __launch_bounds__(256) void synspan_kernel(
    const float *__restrict__ syn, size_t syn_stride, const CompDev *__restrict__ comps, const float *__restrict__ risetime /* of the chunk's sources */,
    float dt, int nmis, const int *__restrict__ spansrc, int nrec, const int *__restrict__ synrow, int *__restrict__ synspan)
{
    const int m = blockIdx.x, s = blockIdx.y;
    const CompDev cd = comps[m];
    if (!cd.untapered) return;
    int n0, n1;
    strip_span(spansrc + ((size_t)(synrow ? synrow[s] : s) * nrec + cd.rec) * kSpanInts, cd.spankind, n0, n1);
    const bool empty = n1 < n0;
    if (empty) { n0 = 0; n1 = 0; }
    const float rise = risetime[s];
    int h = 0;
    if (rise > 0.f) h = ((1 + 2 * (int)roundf(0.5f * rise / dt)) - 1) / 2;
    int lo = n0, hi = n1;
    if (h > 0 && !empty) {                            // (a strip nothing was added to has no length: strip_fold leaves it alone, :360-363)
        int d1 = n1, dl = n0 - 1;                     // first sample that is not zero (else the last one); last sample that differs from the end value
        {
            const float *__restrict__ sy = syn + (size_t)(synrow ? synrow[s] : s) * syn_stride + cd.synofs + cd.halo;   // sy[i] = sample w0 + i
            const float last = sy[n1 - cd.w0];
            for (int t = n0 + (int)threadIdx.x; t <= n1; t += 256) {
                const float v = sy[t - cd.w0];
                if (v != 0.f) d1 = min(d1, t);
                if (v != last) dl = max(dl, t);
            }
        }
        __shared__ int r1[256], r2[256];
        r1[threadIdx.x] = d1; r2[threadIdx.x] = dl;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) { r1[threadIdx.x] = min(r1[threadIdx.x], r1[threadIdx.x + st]); r2[threadIdx.x] = max(r2[threadIdx.x], r2[threadIdx.x + st]); }
            __syncthreads();
        }
        d1 = r1[0];
        const int d2 = r2[0] + 1;
        if (d2 >= d1) { lo = min(n0, d1 - h); hi = max(n1, d2 + h + 1); }      // (an all-zero strip longer than a sample: strip_fold returns at once)
    }
    if (threadIdx.x == 0) { synspan[((size_t)s * nmis + m) * 2] = lo; synspan[((size_t)s * nmis + m) * 2 + 1] = hi; }
}

__global__ __launch_bounds__(256) void misfit_kernel(
    const float *__restrict__ syn, size_t syn_stride, const CompDev *__restrict__ comps,
    const float *__restrict__ reft, const float *__restrict__ tw,
    const float *__restrict__ moment, const float *__restrict__ risetime, MisfitParams mp,
    float *__restrict__ misfit_out, float *__restrict__ proc /* optional [src][stride] processed synthetics */,
    float *__restrict__ fftbuf, float *__restrict__ vt_out /* optional [src][stride] tapered synthetics */,
    const int *__restrict__ synspan /* data spans of the synthetic probes (synspan_kernel), un-tapered receivers only */,
    const FftPair *__restrict__ pairs /* [source][slot], fft_mode only */,
    const int *__restrict__ synrow /* optional [source]: read the synthetics of that source -- sources whose centroid tables are
                                      identical differ only in moment / rise time, which are applied here (the reference
                                      re-scales without re-synthesising then, minimizer_engine.f90:516-521) */)
{
    const int m = blockIdx.x, s = blockIdx.y;
    const CompDev cd = comps[m];
    // window samples that take part in the norm: all of them with a taper; without, the union of the reference's data
    // span and the data span of this source's synthetic strip (probes_norm_timedomain, comparator.f90:798-800)
    int i_lo = 0, i_hi = cd.wlen - 1;
    if (cd.untapered) {
        const int s0 = synspan[((size_t)s * mp.nmis + m) * 2], s1 = synspan[((size_t)s * mp.nmis + m) * 2 + 1];
        const int lo = min(cd.rf0, s0), hi = max(cd.rf1, s1);
        i_lo = max(lo - cd.w0, 0); i_hi = min(hi - cd.w0, cd.wlen - 1);      // (outside the window both arrays are zero)
    }
    const float mom = moment[mp.isrc0 + s];
    const float rise = risetime[mp.isrc0 + s];
    const float *__restrict__ sy = syn + (size_t)(synrow ? synrow[s] : s) * syn_stride + cd.synofs + cd.halo;   // sy[i] = sample w0 + i
    const float *__restrict__ rt = reft + cd.refofs;
    const float *__restrict__ tp = tw + cd.refofs;

    __shared__ float fw[kMaxFold];
    __shared__ int fs[kMaxFold];
    __shared__ float fr[kMaxFold];
    __shared__ int nfold;
    __shared__ double red[256];
    if (threadIdx.x == 0) nfold = fold_setup(rise, mp.dt, fw, fs, fr);
    __syncthreads();
    const int nf = nfold;
    const bool unit = (mp.syn_factor == 1.f);
    double acc = 0.0;
    double peak = 0.0;
    float *__restrict__ frow = nullptr;
    // fft_mode bit0: the tapered synthetic goes, zero-padded to the pair's transform length, into the FFT buffer; a slot
    // without a frequency filter under a time-domain method is compared right here (probes_norm_timedomain takes the
    // plain tapered arrays then, comparator.f90:806-813)
    const bool to_fft = mp.fft_mode && (cd.has_filter || (mp.fft_mode & 2));
    if (to_fft && (mp.fft_mode & 4) && !proc) {            // the in-LDS transform kernel takes the plain synthetics itself (workgroup-uniform)
        const int nt = pairs[(size_t)s * mp.nmis + m].ntrans;
        if (nt >= mp.lds_lo && nt <= mp.lds_hi) return;
    }
    const bool own_row = to_fft && cd.untapered;           // un-tapered: the row is the padded probe array over the PAIR's span (untapered_rows_kernel)
    if (to_fft && !own_row) {
        const FftPair pr = pairs[(size_t)s * mp.nmis + m];
        frow = fftbuf + pr.fft_ofs;
        for (int i = cd.wlen + threadIdx.x; i < pr.ntrans; i += 256) frow[i] = 0.f;      // zero padding
    }
    for (int i = threadIdx.x; i < cd.wlen; i += 256) {
        const float v = folded_scaled_sample(sy, i, nf, fw, fs, fr, mom);
        const float vt = v * tp[i];               // make_array_tapered, comparator.f90:1173-1184
        if (proc) proc[(size_t)s * syn_stride + cd.synofs + cd.halo + i] = mp.write_tapered == 2 ? vt : v;
        if (frow) { frow[i] = vt; continue; }
        if (own_row) continue;
        if (mp.skip_norm) { vt_out[(size_t)s * syn_stride + cd.synofs + cd.halo + i] = vt; continue; }
        if (i < i_lo || i > i_hi) continue;
        const float a = rt[i];
        switch (mp.method) {
        case 1: {                                 // l2norm_func, comparator.f90:650-659
            const float d = unit ? (a - vt) : (1.f * a - mp.syn_factor * vt);
            acc = sq_acc(acc, d); break; }
        case 2: {                                 // l1norm_func, :639-648
            const float d = unit ? fabsf(a - vt) : fabsf(1.f * a - mp.syn_factor * vt);
            acc += (double)d; break; }
        case 5:                                   // scalar_product_2, :627-637
            acc += unit ? (double)(a * vt) : (double)(a * 1.f * vt * mp.syn_factor); break;
        default: {                                // maxabs_func, :661-667
            const double x = (double)(1.f * a), y = (double)(mp.syn_factor * vt);
            peak = fmax(peak, sqrt(x * x + y * y)); break; }
        }
    }
    if (to_fft || mp.skip_norm) return;
    red[threadIdx.x] = (mp.method == 6) ? peak : acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            if (mp.method == 6) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + st]);
            else red[threadIdx.x] += red[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double tot = red[0];
        float res;
        switch (mp.method) {
        case 1: res = (float)sqrt((double)mp.dt * tot); break;
        case 2: res = (float)((double)mp.dt * tot); break;
        case 5: res = (float)tot; break;
        default: res = (float)tot; break;
        }
        misfit_out[(size_t)(mp.isrc0 + s) * mp.nmis + m] = res;
    }
}

// ------------------------------------------------------------------------------------------------
// spectral comparator (comparator.f90:861-886,1186-1231): amplitude spectra from a batched hipFFT r2c
// of the tapered, zero-padded synthetics; optional cosine-PLF frequency filter; fp64 accumulation.

struct SpecParams {
    int method;          // 3 ampspec_l2norm, 4 ampspec_l1norm; or a time-domain id when filtering (1,2,5,6)
    float dt;
    float syn_factor;
    int nmis, isrc0;
    int has_filter;
    // un-tapered slots (comparator.f90:798-800, 861-886): the reference's padded array follows the pair's span, so its amplitude
    // spectrum / filtered trace belongs to the PAIR: refpair[pair.spec_ofs + k], reffiltpair[pair.fft_ofs + n] (pair_span below)
    const float *refpair = nullptr, *reffiltpair = nullptr;
    // Which PAIRS go through the in-LDS transforms (spec_fft_*_kernel): those with lds_lo <= ntrans <= lds_hi; the others through the
    // library transforms (spec_norm_kernel, spec_filter_kernel, filtered_norm_kernel).  Decided per pair, not per batch: what a pair's
    // result is must not depend on the lengths of its neighbours in the batch.  Default: none in LDS.
    int lds_lo = 1, lds_hi = 0;
    __host__ __device__ bool in_lds(int ntrans) const { return ntrans >= lds_lo && ntrans <= lds_hi; }
};

__device__ __forceinline__ double block_sum(double v, double *red)
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    return red[0];
}

__global__ __launch_bounds__(256) void spec_norm_kernel(
    const float2 *__restrict__ spec, const FftPair *__restrict__ pairs, const float *__restrict__ refamp,
    const float *__restrict__ filtw, SpecParams sp, float *__restrict__ misfit_out, const CompDev *__restrict__ comps)
{
    __shared__ double red[256];
    const int m = blockIdx.x, s = blockIdx.y;
    const FftPair pr = pairs[(size_t)s * sp.nmis + m];
    if (sp.in_lds(pr.ntrans)) return;                     // (that pair went through spec_fft_norm_kernel)
    const bool untapered = comps[m].untapered != 0;
    const int nb = pr.ntrans / 2 + 1;
    const float2 *__restrict__ row = spec + pr.spec_ofs;
    const float *__restrict__ ra = untapered ? sp.refpair + pr.spec_ofs : refamp + pr.specofs;
    const float *__restrict__ fw = filtw + pr.specofs;
    const bool unit = (sp.syn_factor == 1.f);
    double acc = 0.0;
    for (int k = threadIdx.x; k < nb; k += 256) {
        const float2 z = row[k];
        float b = hypotf(z.x, z.y);                       // amp_spectrum = abs(spectrum), comparator.f90:1213
        if (sp.has_filter) b = b * fw[k];                 // make_spectrum_filtered, :1226-1228
        const float a = ra[k];                            // reference, already filtered
        if (sp.method == 3) {                             // l2norm_func on amplitude spectra
            const float d = unit ? (a - b) : (1.f * a - sp.syn_factor * b);
            acc = sq_acc(acc, d);
        } else {
            const float d = unit ? fabsf(a - b) : fabsf(1.f * a - sp.syn_factor * b);
            acc += (double)d;
        }
    }
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) {
        const float df = 1.f / ((float)pr.ntrans * sp.dt);               // comparator.f90:1215
        misfit_out[(size_t)(sp.isrc0 + s) * sp.nmis + m] =
            (sp.method == 3) ? (float)sqrt((double)df * tot) : (float)((double)df * tot);
    }
}

// ---- amplitude-spectrum norms without the library transform ---------------------------------------------------------
// ampspec_l2norm / ampspec_l1norm need |X[k]| of the tapered, zero-padded trace only to compare it with the reference's
// (comparator.f90:861-886,1186-1231): the spectrum itself is never used again.  hipFFT's r2c is two kernels (a complex
// transform of half the length and a post-processing pass) that write and re-read the whole spectrum, and spec_norm_kernel
// reads it once more -- four passes over 2.5 GB at cfg5.  Here one workgroup per (slot, source) pair transforms its row in
// LDS and reduces it to the one number that leaves the chip:
//   z[n] = x[2n] + i x[2n+1], n < M = ntrans / 2;  Z = DFT_M(z) by decimation in frequency, radix 4 (one radix-2 stage
//   at the end when log2 M is odd), in place -- stage `len` turns each block of len points into four blocks of len / 4
//   whose transforms are the outputs 4k', 4k'+1, 4k'+2, 4k'+3, so frequency k ends at position
//   sum_j digit_j(k) * len_j / 4 (digit-reversed; the norm needs every bin once, in no particular order);
//   X[k] = E[k] + exp(-2 pi i k / ntrans) O[k],  E = (Z[k] + conj Z[M-k]) / 2,  O = (Z[k] - conj Z[M-k]) / (2i),  k = 0 .. M.
// Twiddle factors come from a table per length made on the host in double precision: per stage three runs of len / 4
// factors w^pos, w^2pos, w^3pos (read with unit stride), then exp(-2 pi i k / ntrans) for k = 0 .. M.
constexpr int kFusedFftMinLog2 = 6, kFusedFftMaxLog2 = 15;       // 64 .. 32768 samples (M * 8 B of LDS: up to 128 KB)
struct FusedFftTables { const float2 *tab[kFusedFftMaxLog2 + 1]; };

__host__ __device__ inline size_t fused_fft_table_size(int ntrans)
{
    const int M = ntrans / 2;
    size_t n = 0;
    for (int len = M; len >= 4; len >>= 2) n += 3 * (size_t)(len >> 2);
    return n + (size_t)M + 1;
}

// complex product and product-sum with fused multiply-adds: these transforms are compared with the reference's to a
// tolerance (its FFTW rounds differently anyway), so the fewer roundings the better -- unlike the accumulate path, which
// must round every operation as the reference does
__device__ __forceinline__ float2 cmulf(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
// e + w o
__device__ __forceinline__ float2 cmaddf(float2 e, float2 w, float2 o)
{
    return make_float2(fmaf(w.x, o.x, fmaf(-w.y, o.y, e.x)), fmaf(w.x, o.y, fmaf(w.y, o.x, e.y)));
}

// one radix-4 decimation-in-frequency butterfly: a, b, c, d a quarter block apart, w1..w3 = w^pos, w^2pos, w^3pos
__device__ __forceinline__ void fused_fft_r4(float2 &a, float2 &b, float2 &c, float2 &d, float2 w1, float2 w2, float2 w3)
{
    const float2 t0 = make_float2(a.x + c.x, a.y + c.y), t1 = make_float2(a.x - c.x, a.y - c.y);
    const float2 t2 = make_float2(b.x + d.x, b.y + d.y), t3 = make_float2(b.y - d.y, d.x - b.x);     // -i (b - d)
    a = make_float2(t0.x + t2.x, t0.y + t2.y);
    b = cmulf(make_float2(t1.x + t3.x, t1.y + t3.y), w1);
    c = cmulf(make_float2(t0.x - t2.x, t0.y - t2.y), w2);
    d = cmulf(make_float2(t1.x - t3.x, t1.y - t3.y), w3);
}

// position of frequency k (0 <= k < M) after the in-place stages: the base-4 digits of k in reverse order (bit reversal
// with the two bits of every digit swapped back), the odd top bit of k -- the radix-2 stage -- at the bottom
__device__ __forceinline__ int fused_fft_pos(int k, int lgM)
{
    const int nd = lgM & ~1;                                             // bits taken by the radix-4 digits
    unsigned r = __brev((unsigned)k << (32 - nd));                       // low nd bits of k, reversed
    r = ((r & 0xaaaaaaaau) >> 1) | ((r & 0x55555555u) << 1);
    return (lgM & 1) ? (int)((r << 1) | ((unsigned)k >> nd)) : (int)r;
}

// LDS index of point p: the five bits that select the bank pair are mixed with higher bits, so that every access pattern of
// the kernel -- consecutive points, the stages' strides of len / 4, the digit-reversed reads at the end (64 lanes on ONE
// bank pair without it) -- spreads over all banks (at most 3 lanes per bank pair, 2 is the floor for 8-byte accesses;
// found by search over xor masks; a bijection of [0, M) for M >= 32)
__device__ __forceinline__ int fused_fft_lds(int p) { return p ^ (((p >> 2) ^ (p >> 5) ^ (p >> 10)) & 31); }

// |x + i y| without overflow or underflow of the squares (the scale is a power of two: exact)
__device__ __forceinline__ float amp2f(float x, float y)
{
    const float m = fmaxf(fabsf(x), fabsf(y));
    if (!(m > 0.f) || m > 3.0e38f) return m != m ? m : fabsf(x) + fabsf(y);     // 0, inf, nan
    const int e = __builtin_amdgcn_frexp_expf(m);
    const float sx = ldexpf(x, -e), sy = ldexpf(y, -e);
    return ldexpf(__builtin_amdgcn_sqrtf(sx * sx + sy * sy), e);          // argument in [1/4, 2): the hardware root (1 ulp) needs no fix-ups
}

// In-place forward transform of the M points in `zf` (decimation in frequency; frequency k ends at fused_fft_pos(k)); `tw`:
// the stage tables of this length; returns the table that follows them (exp(-2 pi i k / ntrans)).  Ends with a barrier.
__device__ __forceinline__ const float2 *fused_fft_forward(float2 *zf, const float2 *__restrict__ tw, int M, int tid)
{
    int len = M;
    // two radix-4 stages at a time while the block length allows: the 16 points base + a len/4 + b len/16 stay in registers
    // between the stage over a and the stage over b (same operations as two single stages, half the LDS round trips)
    for (; len >= 16; len >>= 4) {
        const int q1 = len >> 2, q2 = len >> 4;
        const float2 *__restrict__ tw2 = tw + 3 * q1;
        // the swizzle is linear over xor and base, a q1, b q2 occupy different bits: index = lds(base) ^ lds(a q1) ^ lds(b q2),
        // the last two the same for every lane
        int sa[4], sb[4];
#pragma unroll
        for (int a = 0; a < 4; a++) { sa[a] = fused_fft_lds(a * q1); sb[a] = fused_fft_lds(a * q2); }
        for (int j = tid; j < (M >> 4); j += 256) {
            const int pos = j & (q2 - 1), base = fused_fft_lds(((j - pos) << 4) + pos);
            float2 v[4][4];
            float2 w1[4][3], w2[3];
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int r = 0; r < 3; r++) w1[b][r] = tw[r * q1 + pos + b * q2];
#pragma unroll
            for (int r = 0; r < 3; r++) w2[r] = tw2[r * q2 + pos];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) v[a][b] = zf[base ^ sa[a] ^ sb[b]];
#pragma unroll
            for (int b = 0; b < 4; b++) fused_fft_r4(v[0][b], v[1][b], v[2][b], v[3][b], w1[b][0], w1[b][1], w1[b][2]);
#pragma unroll
            for (int a = 0; a < 4; a++) fused_fft_r4(v[a][0], v[a][1], v[a][2], v[a][3], w2[0], w2[1], w2[2]);
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) zf[base ^ sa[a] ^ sb[b]] = v[a][b];
        }
        tw += 3 * q1 + 3 * q2;
        __syncthreads();
    }
    if (len >= 4) {
        const int q = len >> 2;
#pragma unroll 4
        for (int j = tid; j < (M >> 2); j += 256) {
            const int pos = j & (q - 1), base = ((j - pos) << 2) + pos;
            const int ia = fused_fft_lds(base), ib = fused_fft_lds(base + q), ic = fused_fft_lds(base + 2 * q), id = fused_fft_lds(base + 3 * q);
            float2 a = zf[ia], b = zf[ib], c = zf[ic], d = zf[id];
            fused_fft_r4(a, b, c, d, tw[pos], tw[q + pos], tw[2 * q + pos]);
            zf[ia] = a; zf[ib] = b; zf[ic] = c; zf[id] = d;
        }
        tw += 3 * q;
        len >>= 2;
        __syncthreads();
    }
    if (len == 2) {
        for (int j = tid; j < (M >> 1); j += 256) {
            const int ia = fused_fft_lds(2 * j), ib = fused_fft_lds(2 * j + 1);
            const float2 a = zf[ia], b = zf[ib];
            zf[ia] = make_float2(a.x + b.x, a.y + b.y);
            zf[ib] = make_float2(a.x - b.x, a.y - b.y);
        }
        __syncthreads();
    }
    return tw;
}

// transposed butterfly for the way back: twiddles (conjugated) first, then the 4-point inverse transform across the quarters
__device__ __forceinline__ void fused_fft_r4_inv(float2 &a, float2 &b, float2 &c, float2 &d, float2 w1, float2 w2, float2 w3)
{
    w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y;
    b = cmulf(b, w1); c = cmulf(c, w2); d = cmulf(d, w3);
    const float2 t0 = make_float2(a.x + c.x, a.y + c.y), t1 = make_float2(a.x - c.x, a.y - c.y);
    const float2 t2 = make_float2(b.x + d.x, b.y + d.y), t3 = make_float2(d.y - b.y, b.x - d.x);     // +i (b - d)
    a = make_float2(t0.x + t2.x, t0.y + t2.y);
    b = make_float2(t1.x + t3.x, t1.y + t3.y);
    c = make_float2(t0.x - t2.x, t0.y - t2.y);
    d = make_float2(t1.x - t3.x, t1.y - t3.y);
}

// In-place inverse (unnormalised: M times the inverse transform) of M points that sit where fused_fft_forward leaves them
// (frequency k at fused_fft_pos(k)); the result is in natural order.  The forward stages transposed, last stage first:
// the DFT matrix is symmetric, so (stages)^T applied to the digit-reversed arrangement is the transform itself;
// conjugated twiddles and +i make it the inverse.  `tab`: start of the length's stage tables (stage `len` at tab + M - len).
__device__ __forceinline__ void fused_fft_inverse(float2 *zf, const float2 *__restrict__ tab, int M, int tid)
{
    int rem = M;
    while (rem >= 16) rem >>= 4;                       // what the forward pass had left after its double stages: 1, 2, 4 or 8
    int len = 1;
    if (rem == 2 || rem == 8) {                        // the radix-2 stage
        for (int j = tid; j < (M >> 1); j += 256) {
            const int ia = fused_fft_lds(2 * j), ib = fused_fft_lds(2 * j + 1);
            const float2 a = zf[ia], b = zf[ib];
            zf[ia] = make_float2(a.x + b.x, a.y + b.y);
            zf[ib] = make_float2(a.x - b.x, a.y - b.y);
        }
        len = 2;
        __syncthreads();
    }
    if (rem >= 4) {                                    // the single radix-4 stage (block length 4 or 8)
        len <<= 2;
        const int q = len >> 2;
        const float2 *__restrict__ tw = tab + (M - len);
#pragma unroll 4
        for (int j = tid; j < (M >> 2); j += 256) {
            const int pos = j & (q - 1), base = ((j - pos) << 2) + pos;
            const int ia = fused_fft_lds(base), ib = fused_fft_lds(base + q), ic = fused_fft_lds(base + 2 * q), id = fused_fft_lds(base + 3 * q);
            float2 a = zf[ia], b = zf[ib], c = zf[ic], d = zf[id];
            fused_fft_r4_inv(a, b, c, d, tw[pos], tw[q + pos], tw[2 * q + pos]);
            zf[ia] = a; zf[ib] = b; zf[ic] = c; zf[id] = d;
        }
        __syncthreads();
    }
    while (len < M) {                                  // double stages, small block length first: stage len / 4 (over b), then len (over a)
        len <<= 4;
        const int q1 = len >> 2, q2 = len >> 4;
        const float2 *__restrict__ tw = tab + (M - len), *__restrict__ tw2 = tw + 3 * q1;
        int sa[4], sb[4];
#pragma unroll
        for (int a = 0; a < 4; a++) { sa[a] = fused_fft_lds(a * q1); sb[a] = fused_fft_lds(a * q2); }
        for (int j = tid; j < (M >> 4); j += 256) {
            const int pos = j & (q2 - 1), base = fused_fft_lds(((j - pos) << 4) + pos);
            float2 v[4][4];
            float2 w1[4][3], w2[3];
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int r = 0; r < 3; r++) w1[b][r] = tw[r * q1 + pos + b * q2];
#pragma unroll
            for (int r = 0; r < 3; r++) w2[r] = tw2[r * q2 + pos];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) v[a][b] = zf[base ^ sa[a] ^ sb[b]];
#pragma unroll
            for (int a = 0; a < 4; a++) fused_fft_r4_inv(v[a][0], v[a][1], v[a][2], v[a][3], w2[0], w2[1], w2[2]);
#pragma unroll
            for (int b = 0; b < 4; b++) fused_fft_r4_inv(v[0][b], v[1][b], v[2][b], v[3][b], w1[b][0], w1[b][1], w1[b][2]);
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) zf[base ^ sa[a] ^ sb[b]] = v[a][b];
        }
        __syncthreads();
    }
}

// where the rows come from when the kernel takes the plain synthetics itself (mode 2): what misfit_kernel is given
struct SynRows {
    const float *syn; size_t syn_stride;
    const CompDev *comps;
    const float *taper;                    // taper weights per window sample, [refofs + i]
    const float *moment, *risetime;        // per uploaded source
    const int *synrow;                     // optional: the source whose synthetics this one shares
};

// mode 0: misfit of pair (source s = blockIdx.x, slot m = blockIdx.y) of `pairs[s * nmis + m]` into misfit_out; the row is
//         the tapered, zero-padded synthetic misfit_kernel left in fftbuf
// mode 2: the same from the PLAIN synthetics: rise-time fold, moment and taper (what misfit_kernel does per sample) are
//         applied while the row is brought into LDS, the zero padding is never stored anywhere
// mode 1: rows of reference variants, pairs[blockIdx.x]: amp_out[specofs + k] = |X[k]| * filtw[specofs + k]
template <int MODE>
__global__ __launch_bounds__(256) void spec_fft_norm_kernel(
    const float *__restrict__ fftbuf, const FftPair *__restrict__ pairs, FusedFftTables tabs,
    const float *__restrict__ refamp, const float *__restrict__ filtw, SpecParams sp, float *__restrict__ misfit_out,
    float *__restrict__ amp_out, SynRows sr)
{
    extern __shared__ __attribute__((aligned(16))) float2 zf[];
    __shared__ double red[256];
    const int tid = threadIdx.x;
    // (modes 0, 2: source index fastest -- the workgroups in flight share the reference and filter rows of a few slots)
    const int m = MODE != 1 ? (int)blockIdx.y : 0, s = MODE != 1 ? (int)blockIdx.x : 0;
    const FftPair pr = MODE != 1 ? pairs[(size_t)s * sp.nmis + m] : pairs[blockIdx.x];
    if (!sp.in_lds(pr.ntrans)) return;                    // (library transforms for that pair; workgroup-uniform)
    const int N = pr.ntrans, M = N >> 1;
    const float2 *__restrict__ tw = tabs.tab[31 - __clz(N)];
    if constexpr (MODE == 2) {
        __shared__ float fw[kMaxFold];
        __shared__ int fs[kMaxFold];
        __shared__ float fr[kMaxFold];
        __shared__ int nfold;
        const CompDev cd = sr.comps[m];
        const float mom = sr.moment[sp.isrc0 + s];
        if (tid == 0) nfold = fold_setup(sr.risetime[sp.isrc0 + s], sp.dt, fw, fs, fr);
        __syncthreads();
        const int nf = nfold;
        const float *__restrict__ sy = sr.syn + (size_t)(sr.synrow ? sr.synrow[s] : s) * sr.syn_stride + cd.synofs + cd.halo;
        const float *__restrict__ tp = sr.taper + cd.refofs;
#pragma unroll 4
        for (int n = tid; n < M; n += 256) {
            const int i = 2 * n;
            float2 x = make_float2(0.f, 0.f);
            if (i < cd.wlen) x.x = folded_scaled_sample(sy, i, nf, fw, fs, fr, mom) * tp[i];           // make_array_tapered, comparator.f90:1173-1184
            if (i + 1 < cd.wlen) x.y = folded_scaled_sample(sy, i + 1, nf, fw, fs, fr, mom) * tp[i + 1];
            zf[fused_fft_lds(n)] = x;
        }
    } else {
        const float2 *__restrict__ row = reinterpret_cast<const float2 *>(fftbuf + pr.fft_ofs);
#pragma unroll 8
        for (int n = tid; n < M; n += 256) zf[fused_fft_lds(n)] = row[n];
    }
    __syncthreads();
    tw = fused_fft_forward(zf, tw, M, tid);
    const float *__restrict__ ra = MODE != 1 ? refamp + pr.specofs : nullptr;
    const float *__restrict__ fw = filtw + pr.specofs;
    const bool unit = (sp.syn_factor == 1.f);
    const int lgM = 31 - __clz(M);
    double acc = 0.0;
    auto bin = [&](int k, float re, float im) {
        float b = amp2f(re, im);                                         // amp_spectrum = abs(spectrum), comparator.f90:1213
        if (MODE == 1) { amp_out[pr.specofs + k] = b * fw[k]; return; }
        if (sp.has_filter) b = b * fw[k];                                // make_spectrum_filtered, :1226-1228
        const float a = ra[k];                                           // reference, already filtered
        if (sp.method == 3) {                                            // l2norm_func on amplitude spectra
            const float d = unit ? (a - b) : (1.f * a - sp.syn_factor * b);
            acc = sq_acc(acc, d);
        } else {
            const float d = unit ? fabsf(a - b) : fabsf(1.f * a - sp.syn_factor * b);
            acc += (double)d;
        }
    };
    // bins k and M - k come from the same two points: X[k] = E + w O, X[M-k] = conj(E - w O), w = exp(-2 pi i k / ntrans)
    // (k = 0 gives bins 0 and M, k = M / 2 one bin)
#pragma unroll 2
    for (int k = tid; k <= (M >> 1); k += 256) {
        const float2 zk = zf[fused_fft_lds(fused_fft_pos(k, lgM))];
        float2 zm = zf[fused_fft_lds(fused_fft_pos((M - k) & (M - 1), lgM))];
        zm.y = -zm.y;                                                    // conj Z[M - k]
        const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y + zm.y));
        const float2 o = make_float2(0.5f * (zk.y - zm.y), -0.5f * (zk.x - zm.x));
        const float2 w = tw[k];
        const float2 xp = cmaddf(e, w, o), xm = cmaddf(e, make_float2(-w.x, -w.y), o);
        bin(k, xp.x, xp.y);
        if (2 * k != M) bin(M - k, xm.x, xm.y);
    }
    if (MODE == 1) return;
    const double tot = block_sum(acc, red);
    if (tid == 0) {
        const float df = 1.f / ((float)N * sp.dt);                       // comparator.f90:1215
        misfit_out[(size_t)(sp.isrc0 + s) * sp.nmis + m] =
            (sp.method == 3) ? (float)sqrt((double)df * tot) : (float)((double)df * tot);
    }
}

// ---- time-domain norms on frequency-filtered traces without the library transforms --------------------------------
// comparator.f90:810-813,1224-1263: spectrum * filter(j df), back to the time domain, / ntrans, zero where the taper is zero,
// then the time-domain norm against the reference processed the same way.  With hipFFT that is r2c (two kernels),
// spec_filter_kernel, c2r (two kernels) and filtered_norm_kernel -- six passes over the padded rows; here the row goes
// forward and back inside LDS:
//   forward as in spec_fft_norm_kernel; per point pair (Z[k], Z[M-k]): X[k] = E + w O and conj X[M-k] = E - w O are
//   multiplied by their filter weights and packed again for the way back, Z''[k] = A + i B, Z''[M-k] = conj A + i conj B with
//   A = Y[k] + conj Y[M-k], B = (Y[k] - conj Y[M-k]) conj w  (twice the spectra of the even / odd samples: the factor makes
//   the unnormalised inverse ntrans times the filtered trace, what c2r delivers);
//   inverse of M points (fused_fft_inverse) -> y[2n] + i y[2n+1] in natural order.
// mode 0: trial source rows from the plain synthetics (fold, moment, taper on the way in) -> misfit of the pair
// mode 1: reference variants, rows (tapered reference, zero padded) from fftbuf -> ref_filt[filtofs + i]
template <int MODE>
__global__ __launch_bounds__(256) void spec_fft_filter_norm_kernel(
    const float *__restrict__ fftbuf, const FftPair *__restrict__ pairs, FusedFftTables tabs, const CompDev *__restrict__ comps,
    const float *__restrict__ filtw, const float *__restrict__ ref_filt, const float *__restrict__ zmask, SpecParams sp,
    float *__restrict__ misfit_out, float *__restrict__ filt_out, SynRows sr)
{
    extern __shared__ __attribute__((aligned(16))) float2 zf[];
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int m = MODE == 0 ? (int)blockIdx.y : 0, s = MODE == 0 ? (int)blockIdx.x : 0;
    const FftPair pr = MODE == 0 ? pairs[(size_t)s * sp.nmis + m] : pairs[blockIdx.x];
    const CompDev cd = comps[MODE == 0 ? m : pr.slot];
    if (!cd.has_filter) return;                           // compared by misfit_kernel on the plain tapered arrays
    if (!sp.in_lds(pr.ntrans)) return;                    // (library transforms for that pair)
    const int N = pr.ntrans, M = N >> 1;
    const float2 *__restrict__ tab = tabs.tab[31 - __clz(N)];
    if constexpr (MODE == 0) {
        __shared__ float fw[kMaxFold];
        __shared__ int fs[kMaxFold];
        __shared__ float fr[kMaxFold];
        __shared__ int nfold;
        const float mom = sr.moment[sp.isrc0 + s];
        if (tid == 0) nfold = fold_setup(sr.risetime[sp.isrc0 + s], sp.dt, fw, fs, fr);
        __syncthreads();
        const int nf = nfold;
        const float *__restrict__ sy = sr.syn + (size_t)(sr.synrow ? sr.synrow[s] : s) * sr.syn_stride + cd.synofs + cd.halo;
        const float *__restrict__ tp = sr.taper + cd.refofs;
#pragma unroll 4
        for (int n = tid; n < M; n += 256) {
            const int i = 2 * n;
            float2 x = make_float2(0.f, 0.f);
            if (i < cd.wlen) x.x = folded_scaled_sample(sy, i, nf, fw, fs, fr, mom) * tp[i];
            if (i + 1 < cd.wlen) x.y = folded_scaled_sample(sy, i + 1, nf, fw, fs, fr, mom) * tp[i + 1];
            zf[fused_fft_lds(n)] = x;
        }
    } else {
        const float2 *__restrict__ row = reinterpret_cast<const float2 *>(fftbuf + pr.fft_ofs);
#pragma unroll 8
        for (int n = tid; n < M; n += 256) zf[fused_fft_lds(n)] = row[n];
    }
    __syncthreads();
    const float2 *__restrict__ tw = fused_fft_forward(zf, tab, M, tid);
    const float *__restrict__ fwt = filtw + pr.specofs;
    const int lgM = 31 - __clz(M);
#pragma unroll 2
    for (int k = tid; k <= (M >> 1); k += 256) {
        const int pk = fused_fft_lds(fused_fft_pos(k, lgM)), pm = fused_fft_lds(fused_fft_pos((M - k) & (M - 1), lgM));
        const float2 zk = zf[pk];
        float2 zm = zf[pm];
        zm.y = -zm.y;                                                    // conj Z[M - k]
        const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y + zm.y));
        const float2 o = make_float2(0.5f * (zk.y - zm.y), -0.5f * (zk.x - zm.x));
        const float2 w = tw[k];
        const float2 xp = cmaddf(e, w, o), xm = cmaddf(e, make_float2(-w.x, -w.y), o);
        const float fk = fwt[k], fm = fwt[M - k];
        const float2 yk = make_float2(xp.x * fk, xp.y * fk);                          // spectrum * filter, comparator.f90:1224-1225
        const float2 ym = make_float2(xm.x * fm, xm.y * fm);                          // conj of bin M - k, filtered
        const float2 A = make_float2(yk.x + ym.x, yk.y + ym.y);
        const float2 B = cmulf(make_float2(yk.x - ym.x, yk.y - ym.y), make_float2(w.x, -w.y));
        zf[pk] = make_float2(A.x - B.y, A.y + B.x);                                   // A + i B
        if (pm != pk) zf[pm] = make_float2(A.x + B.y, B.x - A.y);                     // conj A + i conj B
    }
    __syncthreads();
    fused_fft_inverse(zf, tab, M, tid);
    const float *__restrict__ zm_ = zmask + cd.refofs;
    if constexpr (MODE == 1) {
        for (int i = tid; i < cd.wlen; i += 256) {
            const float2 z = zf[fused_fft_lds(i >> 1)];
            const float v = ((i & 1) ? z.y : z.x) / (float)N;
            filt_out[pr.filtofs + i] = v * zm_[i];
        }
        return;
    }
    const float *__restrict__ rf = ref_filt + pr.filtofs;
    const bool unit = (sp.syn_factor == 1.f);
    double acc = 0.0, peak = 0.0;
    for (int i = tid; i < cd.wlen; i += 256) {
        const float2 z = zf[fused_fft_lds(i >> 1)];
        float v = ((i & 1) ? z.y : z.x) / (float)N;                      // normalize result, comparator.f90:1251
        v = v * zm_[i];                                                  // :1254-1258
        const float a = rf[i];
        switch (sp.method) {
        case 1: { const float d = unit ? (a - v) : (1.f * a - sp.syn_factor * v); acc = sq_acc(acc, d); break; }
        case 2: { const float d = unit ? fabsf(a - v) : fabsf(1.f * a - sp.syn_factor * v); acc += (double)d; break; }
        case 5: acc += unit ? (double)(a * v) : (double)(a * 1.f * v * sp.syn_factor); break;
        default: { const double x = (double)(1.f * a), y = (double)(sp.syn_factor * v); peak = fmax(peak, sqrt(x * x + y * y)); break; }
        }
    }
    double tot;
    if (sp.method == 6) {
        red[tid] = peak;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) red[tid] = fmax(red[tid], red[tid + st]);
            __syncthreads();
        }
        tot = red[0];
    } else {
        tot = block_sum(acc, red);
    }
    if (tid == 0) {
        float res;
        switch (sp.method) {
        case 1: res = (float)sqrt((double)sp.dt * tot); break;
        case 2: res = (float)((double)sp.dt * tot); break;
        default: res = (float)tot; break;
        }
        misfit_out[(size_t)(sp.isrc0 + s) * sp.nmis + m] = res;
    }
}

// spectrum_filtered = spectrum * filter(j df) (comparator.f90:1224-1225), in place, before the c2r; block per entry of
// `pairs` (trial-source pairs of a chunk, or reference variants)
__global__ __launch_bounds__(256) void spec_filter_kernel(float2 *__restrict__ spec, const FftPair *__restrict__ pairs,
                                                          const CompDev *__restrict__ comps, const float *__restrict__ filtw, int lds_lo, int lds_hi)
{
    const FftPair pr = pairs[blockIdx.x];
    if (!comps[pr.slot].has_filter) return;
    if (pr.ntrans >= lds_lo && pr.ntrans <= lds_hi) return;      // (that pair goes through spec_fft_filter_norm_kernel)
    const int nb = pr.ntrans / 2 + 1;
    float2 *__restrict__ row = spec + pr.spec_ofs;
    const float *__restrict__ fw = filtw + pr.specofs;
    for (int k = threadIdx.x; k < nb; k += 256) {
        float2 z = row[k];
        z.x = z.x * fw[k]; z.y = z.y * fw[k];
        row[k] = z;
    }
}

// Data span [s0, s1] of the synthetic probe of pair (chunk source s, slot) as a fresh engine sets it (strip span grown by the
// taps of THIS source's rise time) and, for a transform length N, where the pair's common span starts:
// allowed_span(union of the data spans, .) = union(1) - floor((N - slen(union)) / 2.)  (comparator.f90:1092-1109).
__device__ __forceinline__ void pair_span(const int *__restrict__ synspan, const CompDev &cd, int s, int m, int nmis, int N,
                                          int &s0, int &s1, int &span0)
{
    s0 = synspan[((size_t)s * nmis + m) * 2]; s1 = synspan[((size_t)s * nmis + m) * 2 + 1];      // (synspan_kernel)
    const int u0 = min(cd.rf0, s0), u1 = max(cd.rf1, s1);
    span0 = u0 - (N - (u1 - u0 + 1)) / 2;
}

// time-domain norms on the filtered traces (comparator.f90:810-813,1233-1263): c2r output / ntrans, zeroed
// where the taper is zero (ip_zero_one mask), against the reference processed the same way
__global__ __launch_bounds__(256) void filtered_norm_kernel(
    const float *__restrict__ fftbuf, const CompDev *__restrict__ comps, const FftPair *__restrict__ pairs,
    const float *__restrict__ ref_filt, const float *__restrict__ zmask, SpecParams sp, float *__restrict__ misfit_out,
    float *__restrict__ proc, size_t syn_stride, const int *__restrict__ synspan)
{
    __shared__ double red[256];
    const int m = blockIdx.x, s = blockIdx.y;
    const CompDev cd = comps[m];
    if (!cd.has_filter) return;                           // compared by misfit_kernel on the plain tapered arrays
    const FftPair pr = pairs[(size_t)s * sp.nmis + m];
    if (sp.in_lds(pr.ntrans)) return;                     // (that pair went through spec_fft_filter_norm_kernel)
    const float *__restrict__ row = fftbuf + pr.fft_ofs;
    const float *__restrict__ rf = ref_filt + pr.filtofs;
    const float *__restrict__ zm = zmask + cd.refofs;
    const bool unit = (sp.syn_factor == 1.f);
    double acc = 0.0, peak = 0.0;
    // without a taper: the row is the filtered probe array over the pair's span, the norm runs over the union of the two data
    // spans (probes_norm_timedomain, comparator.f90:798-800) against the pair's filtered reference, nothing is zeroed
    int i_lo = 0, i_hi = cd.wlen - 1, shift = 0;
    if (cd.untapered) {
        int s0, s1, span0;
        pair_span(synspan, cd, s, m, sp.nmis, pr.ntrans, s0, s1, span0);
        shift = cd.w0 - span0;                            // row index of window sample i: i + shift
        i_lo = min(cd.rf0, s0) - cd.w0; i_hi = max(cd.rf1, s1) - cd.w0;
        rf = sp.reffiltpair + pr.fft_ofs + shift;
        if (proc)
            for (int i = threadIdx.x; i < cd.wlen; i += 256)
                if (i < i_lo || i > i_hi) proc[(size_t)s * syn_stride + cd.synofs + cd.halo + i] = 0.f;
    }
    for (int i = i_lo + threadIdx.x; i <= i_hi; i += 256) {
        float v = row[i + shift] / (float)pr.ntrans;      // normalize result, comparator.f90:1251
        if (!cd.untapered) v = v * zm[i];                 // :1254-1258
        if (proc && i >= 0 && i < cd.wlen) proc[(size_t)s * syn_stride + cd.synofs + cd.halo + i] = v;
        const float a = rf[i];
        switch (sp.method) {
        case 1: { const float d = unit ? (a - v) : (1.f * a - sp.syn_factor * v); acc = sq_acc(acc, d); break; }
        case 2: { const float d = unit ? fabsf(a - v) : fabsf(1.f * a - sp.syn_factor * v); acc += (double)d; break; }
        case 5: acc += unit ? (double)(a * v) : (double)(a * 1.f * v * sp.syn_factor); break;
        default: { const double x = (double)(1.f * a), y = (double)(sp.syn_factor * v); peak = fmax(peak, sqrt(x * x + y * y)); break; }
        }
    }
    double tot;
    if (sp.method == 6) {
        red[threadIdx.x] = peak;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + st]);
            __syncthreads();
        }
        tot = red[0];
    } else {
        tot = block_sum(acc, red);
    }
    if (threadIdx.x == 0) {
        float res;
        switch (sp.method) {
        case 1: res = (float)sqrt((double)sp.dt * tot); break;
        case 2: res = (float)((double)sp.dt * tot); break;
        default: res = (float)tot; break;
        }
        misfit_out[(size_t)(sp.isrc0 + s) * sp.nmis + m] = res;
    }
}

// Rows of the un-tapered slots for the transforms: the probe array over the pair's span -- zeros before the data span, the data,
// the last value repeated behind it (probe_set_array / probe_extend_span, comparator.f90:259-265,320-324) -- of the REFERENCE
// (REF) or of the trial source's synthetic (folded, scaled by the moment).  reft holds the un-tapered reference over the window,
// which contains every data span of the batch.
template <bool REF>
__global__ __launch_bounds__(256) void untapered_rows_kernel(
    const float *__restrict__ syn, size_t syn_stride, const CompDev *__restrict__ comps, const float *__restrict__ reft,
    const float *__restrict__ moment, const float *__restrict__ risetime, int isrc0, float dt, int nmis,
    const int *__restrict__ spansrc, int nrec, const int *__restrict__ synspan, const FftPair *__restrict__ pairs, float *__restrict__ fftbuf)
{
    const int m = blockIdx.x, s = blockIdx.y;
    const CompDev cd = comps[m];
    if (!cd.untapered) return;
    const FftPair pr = pairs[(size_t)s * nmis + m];
    const int N = pr.ntrans;
    const float rise = risetime[isrc0 + s];
    int s0, s1, span0;
    pair_span(synspan, cd, s, m, nmis, N, s0, s1, span0);
    bool empty;                                              // (an empty strip: zeros, wherever the window lies)
    { int e0, e1; strip_span(spansrc + ((size_t)s * nrec + cd.rec) * kSpanInts, cd.spankind, e0, e1); empty = e1 < e0; }
    float *__restrict__ frow = fftbuf + pr.fft_ofs;
    if constexpr (REF) {
        const float *__restrict__ rt = reft + cd.refofs;
        for (int n = threadIdx.x; n < N; n += 256) {
            const int t = span0 + n;
            frow[n] = t < cd.rf0 ? 0.f : rt[min(t, cd.rf1) - cd.w0];
        }
    } else {
        __shared__ float fw[kMaxFold];
        __shared__ int fs[kMaxFold];
        __shared__ float fr[kMaxFold];
        __shared__ int nfold;
        if (threadIdx.x == 0) nfold = fold_setup(rise, dt, fw, fs, fr);
        __syncthreads();
        const int nf = nfold;
        const float mom = moment[isrc0 + s];
        const float *__restrict__ sy = syn + (size_t)s * syn_stride + cd.synofs + cd.halo;       // sy[i] = sample w0 + i
        for (int n = threadIdx.x; n < N; n += 256) {
            const int t = span0 + n;
            frow[n] = (t < s0 || empty) ? 0.f : folded_scaled_sample(sy, min(t, s1) - cd.w0, nf, fw, fs, fr, mom);
        }
    }
}

// Reference side of the un-tapered amplitude-spectrum norms, per pair: |X[k]| (x filter weight) of the transformed reference
// row -> refpair, and the pair's norm factor probe_norm_frequencydomain (comparator.f90:888-910) -> normsrc
__global__ __launch_bounds__(256) void pair_refamp_kernel(
    const float2 *__restrict__ spec, const FftPair *__restrict__ pairs, const CompDev *__restrict__ comps,
    const float *__restrict__ filtw, SpecParams sp, float *__restrict__ refpair, float *__restrict__ normsrc)
{
    __shared__ double red[256];
    const int m = blockIdx.x, s = blockIdx.y;
    if (!comps[m].untapered) return;
    const FftPair pr = pairs[(size_t)s * sp.nmis + m];
    const int nb = pr.ntrans / 2 + 1;
    const float2 *__restrict__ row = spec + pr.spec_ofs;
    const float *__restrict__ fw = filtw + pr.specofs;
    double acc = 0.0;
    for (int k = threadIdx.x; k < nb; k += 256) {
        const float2 z = row[k];
        float a = hypotf(z.x, z.y);
        if (sp.has_filter) a = a * fw[k];
        refpair[pr.spec_ofs + k] = a;
        acc += (sp.method == 3) ? (double)a * (double)a : (double)fabsf(a);
    }
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) {
        const float df = 1.f / ((float)pr.ntrans * sp.dt);
        normsrc[(size_t)(sp.isrc0 + s) * sp.nmis + m] = (sp.method == 3) ? 1.f * (float)sqrt((double)df * tot) : 1.f * (float)((double)df * tot);
    }
}

// ... and of the time-domain norms on filtered traces: the filtered reference of the pair (c2r output / ntrans; no taper, so
// nothing is zeroed: comparator.f90:1251-1258) -> reffiltpair, its norm over the reference's data span
// (probe_norm_timedomain, :843-845) -> normsrc
__global__ __launch_bounds__(256) void pair_reffilt_kernel(
    const float *__restrict__ fftbuf, const FftPair *__restrict__ pairs, const CompDev *__restrict__ comps, SpecParams sp,
    const int *__restrict__ synspan,
    float *__restrict__ reffiltpair, float *__restrict__ normsrc, float *__restrict__ reffilt_win /* of the chunk's first source, over
    the window at [pair.filtofs + i]: what get_reference(filtered) hands out */)
{
    __shared__ double red[256];
    const int m = blockIdx.x, s = blockIdx.y;
    const CompDev cd = comps[m];
    if (!cd.untapered || !cd.has_filter) return;
    const FftPair pr = pairs[(size_t)s * sp.nmis + m];
    const int N = pr.ntrans;
    int s0, s1, span0;
    pair_span(synspan, cd, s, m, sp.nmis, N, s0, s1, span0);
    const float *__restrict__ row = fftbuf + pr.fft_ofs;
    double acc = 0.0, peak = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float a = row[n] / (float)N;
        reffiltpair[pr.fft_ofs + n] = a;
        const int t = span0 + n;
        if (s == 0 && t >= cd.w0 && t < cd.w0 + cd.wlen) reffilt_win[pr.filtofs + (t - cd.w0)] = a;
        if (t < cd.rf0 || t > cd.rf1) continue;
        switch (sp.method) {
        case 1: acc += (double)a * (double)a; break;
        case 2: acc += (double)fabsf(a); break;
        case 5: acc += (double)(a * a); break;
        default: peak = fmax(peak, (double)fabsf(a)); break;
        }
    }
    double tot;
    if (sp.method == 6) {
        red[threadIdx.x] = peak;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + st]);
            __syncthreads();
        }
        tot = red[0];
    } else {
        tot = block_sum(acc, red);
    }
    if (threadIdx.x == 0) {
        float nf;
        switch (sp.method) {
        case 1: nf = 1.f * (float)sqrt((double)sp.dt * tot); break;
        case 2: nf = 1.f * (float)((double)sp.dt * tot); break;
        case 5: nf = (1.f * 1.f) * (float)tot; break;
        default: nf = 1.f * (float)tot; break;
        }
        normsrc[(size_t)(sp.isrc0 + s) * sp.nmis + m] = nf;
    }
}

// Transform length of every (trial source, slot) pair of a chunk, as a FRESH reference engine sizes it for this source:
// the synthetic probe is set from the source's own strip (probe_set_array, comparator.f90:222-271: data span = strip
// span, padded to a power of two of at least twice the data length), then probes_adjust_spans (:464-486) gives both
// probes the span allowed_span(union of the two data spans, max of the two minimum lengths) (:1092-1109) -- so
// ntrans = next_power_of_two(max(length of the union, 2 len_ref, 2 len_syn)).  spansrc: per (source, receiver) data spans
// of the horizontal / vertical strips, reduced by geometry_kernel; fold_grow: strip_fold's growth (sparse_trace.f90:379-402).
// (Un-tapered slots: the spans follow the synthetics' VALUES where a rise time folds them -- synspan_kernel, behind the accumulate
// kernel --; with such slots in the batch the sizing runs behind it as well and takes their spans from there.)
__global__ void fft_size_kernel(const int *__restrict__ spansrc, const CompDev *__restrict__ comps, int nmis, int nsrc, int nrec,
                                const float *__restrict__ risetime /* of the chunk's sources */, float dt, int *__restrict__ ntr_out,
                                const int *__restrict__ synspan /* or null: spans of the un-tapered slots' synthetic probes (synspan_kernel) */)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nsrc * nmis) return;
    const int s = idx / nmis, m = idx - s * nmis;
    const CompDev cd = comps[m];
    // strip_fold grows a strip by the taps of THIS source's rise time (receiver.f90:868-897, sparse_trace.f90:379-402), not by
    // the batch's longest: a pair's transform length must not depend on the batch it is evaluated in
    int fold_grow = 0;
    {
        const float rise = risetime[s];
        if (rise > 0.f) fold_grow = ((1 + 2 * (int)roundf(0.5f * rise / dt)) - 1) / 2;
    }
    int s0, s1;
    strip_span(spansrc + ((size_t)s * nrec + cd.rec) * kSpanInts, cd.spankind, s0, s1);
    if (s1 < s0) { s0 = 0; s1 = 0; }                                 // no centroid reached this strip: the reference's empty strip is one zero at sample 0, and no rise time grows it
    else if (fold_grow > 0) { s0 -= fold_grow; s1 += fold_grow + 1; }
    if (synspan && cd.untapered) { s0 = synspan[(size_t)idx * 2]; s1 = synspan[(size_t)idx * 2 + 1]; }
    const int len_ref = cd.rf1 - cd.rf0 + 1, len_syn = s1 - s0 + 1;
    const int len_u = max(cd.rf1, s1) - min(cd.rf0, s0) + 1;
    const int minlength = max((int)ceilf((float)len_ref * 2.f), (int)ceilf((float)len_syn * 2.f));
    int need = max(len_u, minlength);
    if (!cd.untapered) need = max(need, cd.wlen);                    // (without a taper the window is the batch's, not the pair's)
    int n = 1;
    while (n < need) n *= 2;                                         // next_power_of_two, comparator.f90:1111-1118
    ntr_out[idx] = n;
}

// empty spans for the per-source reduction of geometry_kernel
__global__ void span_init_kernel(int *__restrict__ spansrc, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // n4 = number of int4s = pairs of (lo, hi) pairs
    if (i < n4) reinterpret_cast<int4 *>(spansrc)[i] = make_int4(0x7fffffff, -0x7fffffff, 0x7fffffff, -0x7fffffff);
}

// second half of the fused comparator: sum (peak: max) the per-tile, per-wave partials of a slot in a fixed order
__global__ void misfit_finish_kernel(const double *__restrict__ partial, const CompDev *__restrict__ comps, int nmis, int nparts,
                                     int waves_per_tile, int tile_len, int method, float dt, int isrc0, int nsrc,
                                     float *__restrict__ misfit_out)
{
    // The 256 slots of a workgroup are 256 x nparts consecutive doubles: they come in through LDS with every load instruction reading
    // consecutive bytes (a thread walking its own nparts values touches a cache line of its own per load: 0.52 ms per 12 960 cfg2
    // sources, 4 % of a step); each thread then sums ITS values in the same fixed order as before.
    constexpr int kStageParts = 16;                      // (256 x 16 doubles = 32 KB)
    __shared__ double stage[256 * (kStageParts + 1)];    // (a slot's values nparts + 1 doubles apart: no two lanes on one bank pair)
    const int idx0 = blockIdx.x * blockDim.x, idx = idx0 + threadIdx.x;
    const int total = nsrc * nmis;
    const bool staged = nparts <= kStageParts;           // (uniform)
    if (staged) {
        const int nvalid = min(256, total - idx0) * nparts;
        const double *src = partial + (size_t)idx0 * nparts;
        for (int i = threadIdx.x; i < nvalid; i += 256) stage[(i / nparts) * (nparts + 1) + i % nparts] = src[i];
        __syncthreads();
    }
    if (idx >= total) return;
    const int s = idx / nmis, m = idx - s * nmis;
    // tiles this slot's window spans; tile_len == 0: every entry (the buffer was cleared; two kernels with different tilings)
    const int np = tile_len > 0 ? ((comps[m].wlen + tile_len - 1) / tile_len) * waves_per_tile : nparts;
    const double *p = staged ? stage + (size_t)threadIdx.x * (nparts + 1) : partial + (size_t)idx * nparts;
    double tot = 0.0;
    for (int q = 0; q < np; q++) tot = (method == 6) ? fmax(tot, p[q]) : tot + p[q];
    float res;
    switch (method) {
    case 1: res = (float)sqrt((double)dt * tot); break;
    case 2: res = (float)((double)dt * tot); break;
    default: res = (float)tot; break;
    }
    misfit_out[(size_t)(isrc0 + s) * nmis + m] = res;
}

// ------------------------------------------------------------------------------------------------
// floating norms (receiver.f90:439-510): the reference is tried at every integer shift of the receiver's range
// (probe_shift, comparator.f90:273-288: the data move, the taper stays), each time against the same tapered
// synthetic; the shift with the smallest sum over the components (of the misfits, or of their squares) wins.
// partial[(s * nmis + m) * maxns + q] = misfit of slot m at shift fl_lo + q.
__global__ __launch_bounds__(256) void floating_norm_kernel(
    const float *__restrict__ vt, size_t syn_stride, const CompDev *__restrict__ comps,
    const float *__restrict__ refx, const float *__restrict__ tw, int method /* 1 l2, 2 l1 */, float dt,
    float syn_factor, int nmis, int maxns, float *__restrict__ partial, const int *__restrict__ synspan)
{
    __shared__ double red[256];
    const int m = blockIdx.x, s = blockIdx.y;
    const CompDev cd = comps[m];
    int s_lo = 0x7fffffff, s_hi = -0x7fffffff;          // data span of this source's synthetic strip (un-tapered only)
    if (cd.untapered) { s_lo = synspan[((size_t)s * nmis + m) * 2]; s_hi = synspan[((size_t)s * nmis + m) * 2 + 1]; }
    const float *__restrict__ sy = vt + (size_t)s * syn_stride + cd.synofs + cd.halo;
    const float *__restrict__ rx = refx + cd.refxofs;
    const float *__restrict__ tp = tw + cd.refofs;
    const bool unit = (syn_factor == 1.f);
    for (int q = 0; q < cd.fl_ns; q++) {
        // reference value at window sample i for shift fl_lo + q: un-tapered reference at w0 + i - (fl_lo + q)
        const float *__restrict__ rq = rx + (cd.fl_ns - 1 - q);
        int i_lo = 0, i_hi = cd.wlen - 1;
        if (cd.untapered) {                            // union of the SHIFTED reference's data span and the strip's
            const int sh = cd.fl_lo + q;
            i_lo = max(min(cd.rf0 + sh, s_lo) - cd.w0, 0);
            i_hi = min(max(cd.rf1 + sh, s_hi) - cd.w0, cd.wlen - 1);
        }
        double acc = 0.0;
        for (int i = i_lo + threadIdx.x; i <= i_hi; i += 256) {
            const float a = rq[i] * tp[i];                 // make_array_tapered, comparator.f90:1173-1184
            const float b = sy[i];
            if (method == 1) {
                const float d = unit ? (a - b) : (1.f * a - syn_factor * b);
                acc = sq_acc(acc, d);
            } else {
                const float d = unit ? fabsf(a - b) : fabsf(1.f * a - syn_factor * b);
                acc += (double)d;
            }
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0)
            partial[((size_t)s * nmis + m) * maxns + q] = (method == 1) ? (float)sqrt((double)dt * red[0])
                                                                        : (float)((double)dt * red[0]);
        __syncthreads();
    }
}

// minloc over the shifts of sum_k misfit (l1) or sum_k misfit^2 (l2), fp32, first minimum (receiver.f90:490-500)
__global__ void floating_select_kernel(const float *__restrict__ partial, const CompDev *__restrict__ comps,
                                       const int *__restrict__ rec_first, int nrec_en, int nmis, int maxns, int method,
                                       int isrc0, int nsrc, float *__restrict__ misfit_out, int *__restrict__ shift_out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nsrc * nrec_en) return;
    const int s = idx / nrec_en, r = idx - s * nrec_en;
    const int k0 = rec_first[r], k1 = rec_first[r + 1];
    const CompDev cd = comps[k0];
    const float *p = partial + (size_t)s * nmis * maxns;
    int iloc = 0;
    float best = 0.f;
    for (int q = 0; q < cd.fl_ns; q++) {
        float sum = 0.f;
        for (int k = k0; k < k1; k++) {
            const float v = p[(size_t)k * maxns + q];
            sum = sum + (method == 2 ? v : v * v);
        }
        if (q == 0 || sum < best) { best = sum; iloc = q; }
    }
    for (int k = k0; k < k1; k++) misfit_out[(size_t)(isrc0 + s) * nmis + k] = p[(size_t)k * maxns + iloc];
    shift_out[(size_t)(isrc0 + s) * nrec_en + r] = cd.fl_lo + iloc;
}

// minimizer_engine.f90:936-942: per receiver sum of squares in fp32, receivers in order
__global__ void global_kernel(float *misfit, const float *__restrict__ norm,
                              const int *__restrict__ rec_first /*[nrec_en+1]*/, int nrec_en, int nmis,
                              int isrc0, int nsrc, float *__restrict__ global_out, const int *__restrict__ status,
                              const float *__restrict__ norm_src /* [source][slot] when the norm factors follow the pair's transform length */)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsrc) return;
    if (status && status[isrc0 + s]) {               // a trial source the discretiser rejected: zeros (seismosizer.py:703-720)
        float *mz = misfit + (size_t)(isrc0 + s) * nmis;
        for (int k = 0; k < nmis; k++) mz[k] = 0.f;
        global_out[isrc0 + s] = 0.f;
        return;
    }
    const float *m = misfit + (size_t)(isrc0 + s) * nmis;
    if (norm_src) norm = norm_src + (size_t)(isrc0 + s) * nmis;
    float mis = 0.f, nf = 0.f;
    for (int r = 0; r < nrec_en; r++) {
        float a = 0.f, b = 0.f;
        for (int k = rec_first[r]; k < rec_first[r + 1]; k++) a = a + m[k] * m[k];
        for (int k = rec_first[r]; k < rec_first[r + 1]; k++) b = b + norm[k] * norm[k];
        mis = mis + a;
        nf = nf + b;
    }
    global_out[isrc0 + s] = sqrtf(mis) / sqrtf(nf);
}

} // namespace kiwi
