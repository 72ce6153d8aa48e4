// kiwi_kernels.hpp -- CDNA4 (gfx950) kernels of the trial-source hot path.
//
//   geometry_kernel    one thread per (source, receiver, centroid): spherical azimuth/distance
//                      update (orthodrome.f90:77-156), moment-tensor weights (seismogram.f90:316-336),
//                      GF grid indices and bilinear weights (gfdb.f90:781-815), time shift split
//                      (sparse_trace.f90:640-645).  fp64 trig; writes one 80-byte GeoRec.
//   accumulate_kernel  the bandwidth-bound "convolution": for one (source, receiver, 1024-sample
//                      time tile) walk the centroids IN TABLE ORDER and, per Green's function
//                      component, stream the 4 neighbour traces, blend them (gfdb.f90:944-949) and
//                      shift-multiply-add them (sparse_trace.f90:684-703) into register accumulators;
//                      rotate per centroid (seismogram.f90:158-204) and to north/east at the end
//                      (seismogram.f90:256-283).  Every output sample is owned by one thread, so the
//                      fp32 summation order is exactly the reference's.
//   misfit_kernel      scale by moment, optional rise-time fold (receiver.f90:853-904), taper, and
//                      the time-domain norms with fp64 accumulation (comparator.f90:619-697).
//   global_kernel      sqrt(sum m^2)/sqrt(sum n^2) per source (minimizer_engine.f90:924-945).
//
// Built with -ffp-contract=off: the reference rounds every multiply and add separately.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "kiwi_libm32.hpp"

namespace kiwi {

constexpr int kRowPad = 32;       // zeros in front of every GF row (>= 5, see load5)
constexpr int kTile = 1024;       // samples per workgroup: 256 threads x 4 consecutive samples
constexpr int kMaxComp = 5;
constexpr int kHalo = 64;         // grouped accumulate: LDS tile = kTile + kHalo samples
constexpr int kMaxGroup = 64;     // centroids per group at most

// per (source, receiver, centroid) record, written by geometry_kernel, read (wave-uniformly,
// through scalar loads) by accumulate_kernel.  20 x 4 B.
struct GeoRec {
    int   row[4];     // first GF row (ig = 0) of nodes (ix0,iz0) (ix0,iz1) (ix1,iz0) (ix1,iz1); row[0] < 0: skipped
    float w[4];       // (1-dix)(1-diz), (1-dix)diz, dix(1-diz), dix diz       gfdb.f90:946-949
    int   ishift;     // floor(time/dt)                                        sparse_trace.f90:640
    float wfrac;      // time/dt - ishift                                      sparse_trace.f90:642
    float f[6];       // make_weights                                          seismogram.f90:329-334
    float cl, sl;     // cos / sin (bazi - bazi_orig)                          seismogram.f90:164-165
    int   flags;      // bit0: exactly on a node -> no blend (gfdb.f90:890-893); bit1: rotate (seismogram.f90:160);
                      // bit2: same position as the previous centroid; bit3: some needed trace is not stored and the
                      // reference `cycle`s there (seismogram.f90:171-250): only the first (flags >> 8 & 15) horizontal and
                      // (flags >> 12 & 15) vertical components IN APPLICATION ORDER are added, see geometry_kernel
    int   pad;        // group hint: len | (smax-ishift)<<8 | (ishift-smin)<<16, see geometry_kernel
};
static_assert(sizeof(GeoRec) == 80, "GeoRec layout");

struct GfMeta {
    int nx, nz, ng;
    int pitch;                 // floats per row (kRowPad zeros | samples | repeated end value)
    float dt, dx, dz, firstx, firstz;
};

// per receiver constants
struct RecvDev {
    double azi0, bazi0, dist0;    // seismogram.f90:99-100
    float depth;
    float cl0, sl0;               // cos/sin(bazi_orig + pi)             seismogram.f90:270-271
    int   enabled;
    int   ncomp;
    int   comp[kMaxComp];         // |id| 1 away 2 right 3 down 4 north 5 east (receiver.f90:35-48)
    float sign[kMaxComp];
    int   need_h, has_d;
    float sd;                     // sign of the down/up component
    int   wbeg, wlen;             // synthetic window [wbeg, wbeg+wlen) incl. fold halo
    int   synofs[kMaxComp];       // float offset of each component inside one source's synthetic block
    int   slot0;                  // first misfit slot of this receiver (its components follow)
    int   refofs[kMaxComp];       // offset of each component's tapered reference / taper weights (= CompDev::refofs)
};

// fused comparator of the grouped accumulate kernel (time-domain norms without rise-time fold): the synthetics are
// compared with the references where they are produced and never written to memory
struct FuseParams {
    const float *reft, *tw, *moment;      // tapered references, taper weights, moment per source
    double *partial;                      // [source][slot][part] partial sums (peak: maxima); part = tile * waves + wave
    int method;                           // 1 l2norm, 2 l1norm, 5 scalar_product, 6 peak
    float syn_factor;
    int nmis, nparts, isrc0;
};

// per misfit slot (enabled receiver component)
struct CompDev {
    int synofs;      // offset of the component's synthetic (incl. halo) inside a source block
    int halo;
    int w0, wlen;    // misfit window first sample, length
    int refofs;      // offset into reft / tw arrays
    int rec;
    // spectral / filtered comparator (comparator.f90:1186-1263): the transform length belongs to the (trial source, slot)
    // PAIR (FftPair below); per slot only whether the receiver has a frequency filter and the longest transform any source
    // of the batch can need (buffer capacity)
    int has_filter;
    int ntrans_max;
    // floating norms (receiver.f90:439-510): integer shift range of the receiver and where the un-tapered
    // reference over [w0 - fl_hi, w0 + wlen - 1 - fl_lo] lives
    int fl_lo, fl_ns, refxofs;
    // un-tapered comparator (comparator.f90:798-800): norms run over the union of the two data spans
    int untapered, rf0, rf1, vertical;      // reference data span [rf0, rf1]; which strip span applies to the synthetic:
    // vertical: 0 away / right ... see spankind; kept for the shake diagnostics
    int spankind;    // 0: the radial strip (components a / c), 1: the transverse strip (r / l), 2: both made equal (n / e,
                     // seismogram.f90:268-283), 3: the vertical strip -- see strip_span()
};

// Spectral / filtered comparator, one record per (trial source of the chunk, misfit slot) -- or per reference variant
// when the reference probes are pushed through the same pipeline.  The transform length of a probe pair is what a fresh
// reference engine gives THIS source (comparator.f90:222-271,464-486, see fft_size_kernel); pairs of equal length form
// one batched hipFFT plan, their rows are contiguous in the FFT buffers.
struct FftPair {
    long long fft_ofs;     // float offset of the pair's row (ntrans reals) in the real buffer
    long long spec_ofs;    // complex offset of its row (ntrans / 2 + 1 bins) in the spectrum buffer
    int ntrans;
    int specofs;           // reference amplitude spectrum / filter weights of (slot, ntrans): offset into refamp / filtw
    int filtofs;           // filtered reference of (slot, ntrans) over the window: offset into ref_filt
    int slot;
};
static_assert(sizeof(FftPair) == 32, "FftPair layout");

// Shake-map diagnostics of one source (get_peak_amplitudes / get_arias_intensities): the components a receiver's value
// is made of, in the reference's order (receiver.f90:544-594)
struct ShakeRec {
    int slot[3];     // misfit slots (CompDev indices) of the probes, np of them used
    int np;
    int untapered;   // no taper: the norm runs over the union of the synthetic strips' data spans (comparator.f90:733-736)
    int rec;
};

// Data spans of one (source, receiver)'s synthetic strips, 8 ints: [lo, hi] of the radial sum displacement_ar(1), of the
// transverse sum displacement_ar(2), of the vertical strip, 2 unused; lo > hi = empty.  The two horizontal sums are
// separate strips in the reference: a centroid that leaves at a missing trace in the plain (non-rotating) branch may have
// extended one and not the other (seismogram.f90:205-231), the rotating branch makes them equal before it adds
// (strip_extend_to_same_span_4, :196-197), and so does the rotation to north / east at the end (:268-283), AFTER the away /
// right components have taken theirs (:256-267).
constexpr int kSpanInts = 8;
__device__ __forceinline__ void strip_span(const int *__restrict__ sp, int kind, int &lo, int &hi)
{
    if (kind == 3) { lo = sp[4]; hi = sp[5]; return; }
    if (kind == 0) { lo = sp[0]; hi = sp[1]; return; }
    if (kind == 1) { lo = sp[2]; hi = sp[3]; return; }
    lo = min(sp[0], sp[2]); hi = max(sp[1], sp[3]);          // (an empty span is (+inf, -inf): the union is the other one)
}

// ------------------------------------------------------------------------------------------------
// geometry

__device__ __forceinline__ double clipd(double x, double mi, double ma) { return fmin(fmax(mi, x), ma); }
__device__ __forceinline__ double wrapd(double x, double mi, double ma) { return x - floor((x - mi) / (ma - mi)) * (ma - mi); }

// The default-real libm calls of the reference host (sin, cos, atan2 -> glibc sinf, cosf, atan2f)
// are reproduced bit for bit by kiwi_libm32.hpp; the real*8 ones (sin, cos, acos, asin) use the
// device's fp64 libm, which agrees with glibc to an ulp of fp64 ("tolerances").
__device__ __forceinline__ float sin32(float x) { return libm32::sinf_glibc(x); }
__device__ __forceinline__ float cos32(float x) { return libm32::cosf_glibc(x); }

struct EvalParams {
    int bilinear, xus, zus;
    int nrec;
    int isrc0;
    int cellmode;      // groups = runs of centroids in the same 4-node GF cell (cellgroup_kernel marks them), not same-point runs
};

// Length of the centroid group that accumulate_grouped_kernel forms when a group STARTS at centroid c of a source: the
// following centroids at this same point whose integer shifts stay within the LDS halo (also returns the shift range).
__device__ __forceinline__ int group_len(const float *__restrict__ cent, int c0, int nc, int c, float dt, int &smin, int &smax)
{
    const float *ce = cent + (size_t)(c0 + c) * 10;
    const float dnorth = ce[0], deast = ce[1], depth = ce[2];
    int len = 1;
    smin = smax = (int)floorf(ce[3] / dt);
    for (int k = c + 1; k < nc && len < kMaxGroup; k++) {
        const float *ne = cent + (size_t)(c0 + k) * 10;
        if (!(ne[0] == dnorth && ne[1] == deast && ne[2] == depth)) break;
        const int sh = (int)floorf(ne[3] / dt);
        const int nmin = min(smin, sh), nmax = max(smax, sh);
        if (nmax - nmin > kHalo - 10) break;
        smin = nmin; smax = nmax; len++;
    }
    return len;
}

// Does the grouped kernel start a group at centroid c?  Groups never span two points, so the first centroid of a
// same-point run always starts one; inside a run the starts follow from the group lengths (all centroids of a run share
// their GF rows, so they are stored or missing together and the kernel's skipping of missing ones does not interfere).
__device__ __forceinline__ bool starts_group(const float *__restrict__ cent, int c0, int nc, int c, float dt)
{
    const float *ce = cent + (size_t)(c0 + c) * 10;
    int r0 = c;
    while (r0 > 0) {
        const float *pe = cent + (size_t)(c0 + r0 - 1) * 10;
        if (!(pe[0] == ce[0] && pe[1] == ce[1] && pe[2] == ce[2])) break;
        r0--;
    }
    int pos = r0, lo, hi;
    while (pos < c) pos += group_len(cent, c0, nc, pos, dt, lo, hi);
    return pos == c;
}

// Load descriptors for accumulate_grouped_kernel, 128 ints per record, laid out so that one coalesced load per wave
// brings them in lane-distributed: for component ig and node k
//   tab[4*ig + k]      = (row - row0)*pitch + kRowPad - first   (float index of trace sample 0 relative to the group base
//                                                        G + row0*pitch, row0 = first row of the cell's first node)
//   tab[64 + 4*ig + k] = (row - row0)*pitch             (clamp floor; ceiling = floor + pitch - 4)
//   tab[40 + ig]       = last stored sample of the blended trace (max over the nodes)
//   tab[50], tab[51]   = minimum of those over the horizontal (1-5, 9) / vertical (6-8, 10) components
//   tab[52], tab[53]   = minimum / maximum over the cell's rows of (kRowPad - first): where trace sample 0 sits inside its row
//                        (accumulate_cell_kernel: a tile that stays inside every row is loaded without clamps)
//   tab[54]            = 1 when the components of each node start at the same sample: the sample-0 positions of a node's rows are
//                        then `pitch` apart and the kernel takes four descriptors instead of forty
//   tab[64 + 40 + 2*i], [.. + 1] = wl, wr: per-component interpolation coefficients of THIS centroid
//       (sparse_trace.f90:643-647 with the factors of seismogram.f90:171-250): wl = (1 - w) * factor, wr = w * factor,
//       each rounded on its own, for the i-th component in application order 0 1 2 8 | 3 4 | 5 6 7 9 (ng = 8:
//       0 1 2 | 3 4 | 5 6 7).  Read by the grouped kernel with scalar loads, which takes them off the vector pipe.
// Every thread owns one 512-byte row.  All span look-ups come first, then each of the row's four 128-byte lines goes
// out as consecutive 16-byte stores, so that a line is complete in L2 before it leaves it (interleaving the stores
// with the look-ups left every line open for microseconds: partial-line write-backs, 0.32 ms per 1.3 M records).
template <int NG>
__device__ __forceinline__ bool write_tab(int *__restrict__ tb, const GeoRec &g, const int2 *__restrict__ span, int pitch, float sd,
                                          bool full, const unsigned char *__restrict__ endz)
{
    bool all_endzero = true;          // returned: (full rows) every row of the cell ends in an exact zero -- no tail rule for this group
    // full: this centroid starts a group and the kernel reads the whole row; otherwise only its coefficients (they
    // share the row's last 128-byte line with the clamp floors of components 9 and 10, which are then not needed)
    const int nn = (g.flags & 1) ? 1 : 4;
    int bases[NG][4], floors[NG][4], jend[12];
    int jmin_h = 0x7fffffff, jmin_d = 0x7fffffff;
    int amin = 0x7fffffff, amax = -0x7fffffff;      // range of (trace sample 0 inside its row) over the cell's rows
    bool uni = true;                                // the components of every node start at the same sample (the usual database)
    if (full) {
    // Rows whose stored trace ends in an exact zero (the reference's trace_pack keeps one of the zeros that follow the last
    // non-zero sample, sparse_trace.f90:535,545, so this is the normal case for traces that die out inside the database's
    // time range): the repeated end value is 0 and the rule "factor * last after the span" (sparse_trace.f90:698-703) adds
    // the same signed zeros as the interpolation formula does -- the kernel may then skip its tail variant for this group.
    // endz[row]: the row's end value is zero (set by kiwi_hip_set_gfdb).
    bool endzero = true;
#pragma unroll
    for (int ig = 0; ig < NG; ig++) {
        int je = -0x7fffffff;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int row = g.row[k < nn ? k : 0] + ig;
            const int2 sp = span[row];
            // offsets are relative to the first row of the cell's first node (g.row[0]): the kernels add them to a 64-bit
            // base, so a tensor of any size works as long as one cell spans less than 2^31 floats (checked by the host)
            bases[ig][k] = (row - g.row[0]) * pitch + kRowPad - sp.x;
            floors[ig][k] = (row - g.row[0]) * pitch;
            amin = min(amin, kRowPad - sp.x); amax = max(amax, kRowPad - sp.x);
            uni = uni && bases[ig][k] - floors[ig][k] == bases[0][k] - floors[0][k];
            if (k < nn) { je = max(je, sp.y); endzero = endzero && endz[row]; }
        }
        jend[ig] = je;
        const bool horiz = (ig <= 4) || (ig == 8);
        if (horiz) jmin_h = min(jmin_h, je); else jmin_d = min(jmin_d, je);
    }
#pragma unroll
    for (int ig = NG; ig < 10; ig++) jend[ig] = 0;
    jend[10] = endzero ? 0x7fffffff : jmin_h;
    jend[11] = endzero ? 0x7fffffff : jmin_d;
    all_endzero = endzero;
    }
    float cf[20];
    {
        const float wr0 = g.wfrac, wl0 = 1.f - g.wfrac;
        const float fd[4] = { g.f[0] * sd, g.f[1] * sd, g.f[2] * sd, g.f[5] * sd };
        int i = 0;
        if (NG == 10) {
            const float fh[6] = { g.f[0], g.f[1], g.f[2], g.f[5], g.f[3], g.f[4] };
#pragma unroll
            for (int q = 0; q < 6; q++, i++) { cf[2 * i] = wl0 * fh[q]; cf[2 * i + 1] = wr0 * fh[q]; }
#pragma unroll
            for (int q = 0; q < 4; q++, i++) { cf[2 * i] = wl0 * fd[q]; cf[2 * i + 1] = wr0 * fd[q]; }
        } else {
            const float fh8[5] = { g.f[0], g.f[1], g.f[2], g.f[3], g.f[4] };
#pragma unroll
            for (int q = 0; q < 5; q++, i++) { cf[2 * i] = wl0 * fh8[q]; cf[2 * i + 1] = wr0 * fh8[q]; }
#pragma unroll
            for (int q = 0; q < 3; q++, i++) { cf[2 * i] = wl0 * fd[q]; cf[2 * i + 1] = wr0 * fd[q]; }
#pragma unroll
            for (; i < 10; i++) { cf[2 * i] = 0.f; cf[2 * i + 1] = 0.f; }
        }
    }
    int4 *t4 = reinterpret_cast<int4 *>(tb);
    if (full) {
#pragma unroll
        for (int ig = 0; ig < NG; ig++) t4[ig] = make_int4(bases[ig][0], bases[ig][1], bases[ig][2], bases[ig][3]);
#pragma unroll
        for (int q = 0; q < 3; q++) t4[10 + q] = make_int4(jend[4 * q], jend[4 * q + 1], jend[4 * q + 2], jend[4 * q + 3]);
        t4[13] = make_int4(amin, amax, uni ? 1 : 0, 0);
#pragma unroll
        for (int ig = 0; ig < NG; ig++) t4[16 + ig] = make_int4(floors[ig][0], floors[ig][1], floors[ig][2], floors[ig][3]);
    }
    float4 *f4 = reinterpret_cast<float4 *>(tb);
#pragma unroll
    for (int q = 0; q < (NG == 10 ? 5 : 4); q++) f4[26 + q] = make_float4(cf[4 * q], cf[4 * q + 1], cf[4 * q + 2], cf[4 * q + 3]);
    return all_endzero;
}

__global__ __launch_bounds__(256) void geometry_kernel(
    const float *__restrict__ cent, const int *__restrict__ cent_ofs, EvalParams ep, GfMeta gm,
    const int2 *__restrict__ span, const RecvDev *__restrict__ recv, GeoRec *__restrict__ out,
    int *__restrict__ tab, int *__restrict__ spanbuf, int *__restrict__ spansrc /* optional [source][receiver][kSpanInts] */,
    int *__restrict__ pairflag /* optional [source][receiver]: bit 0 some centroid of the pair is added in part (a trace is missing),
                                  bit 1 some centroid is left out (a trace missing or outside the database), bit 2 some group's rows do
                                  not all end in zero (the tail rule can apply); see cell_pair(), multi_taken() */,
    const unsigned char *__restrict__ endz /* per GF row: its end value is zero (write_tab) */,
    const int *__restrict__ synrow /* optional [source]: source whose synthetics this one shares; != own index: nothing to do */)
{
    const int s = blockIdx.y;
    if (synrow && synrow[s] != s) return;
    const int c0 = cent_ofs[ep.isrc0 + s], nc = cent_ofs[ep.isrc0 + s + 1] - c0;
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= nc * ep.nrec) return;                    // (whole workgroup)
    // lanes past the end of the source's records stay in the wave -- the span reduction below is a wave operation --
    // as copies of its last record that neither store nor count
    const bool live = idx < nc * ep.nrec;
    if (!live) { if (!spanbuf && !spansrc) return; idx = nc * ep.nrec - 1; }
    const int r = idx / nc, c = idx - r * nc;
    const RecvDev &rv = recv[r];
    const float *ce = cent + (size_t)(c0 + c) * 10;
    GeoRec g;
    const float dnorth = ce[0], deast = ce[1], depth = ce[2], time = ce[3];
    const float pi_f = 3.14159265358979f;           // constants.f90:21
    const double pi_d = (double)pi_f;               // constants.f90:22
    const float earthradius = 6371.f * 1000.f;      // constants.f90:23

    // ---- approx_differential_azidist, orthodrome.f90:77-156 (exact branch; r == 0 -> const azimuth)
    double azi, bazi, dist;
    {
        const double azimuth = rv.azi0, backazimuth = rv.bazi0, d0 = rv.dist0;
        const float rr = sqrtf(dnorth * dnorth + deast * deast);
        const double rd = (double)rr;
        if (d0 / rd > 1.79769313486231570815e308) {
            azi = azimuth; bazi = backazimuth;
            dist = d0 - ((double)dnorth * cos(azimuth) + (double)deast * sin(azimuth));
        } else {
            const double a = rd / (double)earthradius;
            const double b = d0 / (double)earthradius;
            const double lambda = (double)libm32::atan2f_glibc(deast, dnorth);
            const double gamma = azimuth - lambda;
            const double ca = cos(a), cb = cos(b), sa = sin(a), sb = sin(b), sg = sin(gamma);
            const double cc = acos(clipd(ca * cb + sa * sb * cos(gamma), -1., 1.));
            const double sc = sin(cc), cosc = cos(cc);
            double alpha = asin(clipd(sa * sg / sc, -1., 1.));
            double beta = asin(clipd(sb * sg / sc, -1., 1.));
            if (ca - cb * cosc < 0) alpha = (alpha > 0) ? pi_d - alpha : -pi_d - alpha;
            if (cb - ca * cosc < 0) beta = (beta > 0) ? pi_d - beta : -pi_d - beta;
            dist = cc * (double)earthradius;
            bazi = wrapd(backazimuth + alpha, -pi_d, pi_d);
            azi = wrapd(lambda - pi_d - beta, -pi_d, pi_d);
        }
    }

    // ---- make_weights, seismogram.f90:316-336
    {
        const float azf = (float)azi;
        const float sa = sin32(azf), ca = cos32(azf), s2a = sin32(2.f * azf), c2a = cos32(2.f * azf);
        const float *m = ce + 4;
        g.f[0] = m[0] * (ca * ca) + m[1] * (sa * sa) + m[3] * s2a;
        g.f[1] = m[4] * ca + m[5] * sa;
        g.f[2] = m[2];
        g.f[3] = 0.5f * (m[1] - m[0]) * s2a + m[3] * c2a;
        g.f[4] = m[5] * ca - m[4] * sa;
        g.f[5] = m[0] * (sa * sa) + m[1] * (ca * ca) - m[3] * s2a;
    }

    // ---- time shift, seismogram.f90:139 + sparse_trace.f90:640-642
    {
        const float rshift = time / gm.dt;
        const float fl = floorf(rshift);
        g.ishift = (int)fl;
        g.wfrac = rshift - (float)g.ishift;
    }

    // ---- rotation of horizontals, seismogram.f90:159-165
    {
        const double lambda = bazi - rv.bazi0;
        g.flags = (lambda != 0.) ? 2 : 0;
        g.cl = (float)cos(lambda);
        g.sl = (float)sin(lambda);
    }

    // ---- GF indices, gfdb.f90:781-815
    const float x = (float)dist, z = depth - rv.depth;
    int ix0, iz0, ix1, iz1;
    float dix = 0.f, diz = 0.f;
    if (ep.bilinear) {
        ix0 = (int)floorf((x - gm.firstx) / (gm.dx * (float)ep.xus)) * ep.xus + 1;
        iz0 = (int)floorf((z - gm.firstz) / (gm.dz * (float)ep.zus)) * ep.zus + 1;
        ix1 = ix0 + ep.xus; iz1 = iz0 + ep.zus;
        dix = (x - gm.firstx - (float)(ix0 - 1) * gm.dx) / (gm.dx * (float)ep.xus);
        diz = (z - gm.firstz - (float)(iz0 - 1) * gm.dz) / (gm.dz * (float)ep.zus);
    } else {
        ix0 = (int)roundf((x - gm.firstx) / gm.dx) + 1;       // nint
        iz0 = (int)roundf((z - gm.firstz) / gm.dz) + 1;
        ix1 = ix0 + 1; iz1 = iz0 + 1;
    }
    const bool direct = (dix == 0.f && diz == 0.f);            // gfdb.f90:890
    if (direct) g.flags |= 1;
    // bit2: this centroid sits at exactly the same point as its predecessor in the table (the nt
    // time steps of one sub-fault, source_bilat.f90:443-457): same azimuth, distance, GF nodes and
    // blend weights, so the blended traces can be reused (accumulate_grouped_kernel)
    if (c > 0) {
        const float *pe = ce - 10;
        if (pe[0] == dnorth && pe[1] == deast && pe[2] == depth) g.flags |= 4;
    }
    g.w[0] = (1.f - dix) * (1.f - diz);
    g.w[1] = (1.f - dix) * diz;
    g.w[2] = dix * (1.f - diz);
    g.w[3] = dix * diz;
    auto inrange = [&](int ix, int iz) { return ix >= 1 && ix <= gm.nx && iz >= 1 && iz <= gm.nz; };
    auto rowof = [&](int ix, int iz) { return ((ix - 1) * gm.nz + (iz - 1)) * gm.ng; };
    bool ok = inrange(ix0, iz0);
    if (!direct) ok = ok && inrange(ix0, iz1) && inrange(ix1, iz0) && inrange(ix1, iz1);
    g.row[0] = g.row[1] = g.row[2] = g.row[3] = -1;
    int nlim_h = 0, nlim_d = 0;                    // components of the horizontal / vertical block that are added
    if (ok) {
        g.row[0] = rowof(ix0, iz0);
        if (!direct) { g.row[1] = rowof(ix0, iz1); g.row[2] = rowof(ix1, iz0); g.row[3] = rowof(ix1, iz1); }
        else { g.row[1] = g.row[2] = g.row[3] = g.row[0]; }
        // A trace that is not stored makes gfdb_get_trace[_bilin] return null (gfdb.f90:899-903,1003) and the reference
        // leaves the centroid AT THAT COMPONENT (`if (.not. associated(tracep)) cycle`, seismogram.f90:171-250): what was
        // added before stays, the rest -- including the vertical block when the gap is in the horizontal one -- is not
        // added.  In the rotate branch the horizontals are collected in temporaries that are only added after the last of
        // them (:196-203), so there a gap drops all of them.  Components in application order: 1 2 3 [9] 4 5 | 6 7 8 [10].
        const int nn = direct ? 1 : 4;
        const int nH = gm.ng == 10 ? 6 : 5, nD = gm.ng == 10 ? 4 : 3;
        auto stored = [&](int ig) {
            for (int k = 0; k < nn; k++) { const int2 sp = span[g.row[k] + ig]; if (sp.y < sp.x) return false; }
            return true;
        };
        bool hfull = true;
        if (rv.need_h) {
            int k = 0;
            for (; k < nH; k++) if (!stored(gm.ng == 10 ? (k < 3 ? k : (k == 3 ? 8 : k - 1)) : k)) break;
            hfull = (k == nH);
            nlim_h = hfull ? nH : ((g.flags & 2) ? 0 : k);
        }
        if (rv.has_d && hfull) {
            int k = 0;
            for (; k < nD; k++) if (!stored(k < 3 ? 5 + k : 9)) break;
            nlim_d = k;
        }
        const bool complete = (!rv.need_h || nlim_h == nH) && (!rv.has_d || nlim_d == nD);
        if (!complete) {
            if (nlim_h == 0 && nlim_d == 0) ok = false;                  // nothing of this centroid is added
            else { g.flags |= 8 | (nlim_h << 8) | (nlim_d << 12); if (pairflag) atomicOr(&pairflag[(size_t)s * ep.nrec + r], 1); }
        }
        if (!ok) g.row[0] = -1;
    }
    if (pairflag && live && g.row[0] < 0) atomicOr(&pairflag[(size_t)s * ep.nrec + r], 2);
    // group hint (used by accumulate_grouped_kernel when a group STARTS at this centroid): how many
    // following centroids sit at this same point with their integer shifts within the LDS halo, and
    // the spread of those shifts.  pad = len | (smax - ishift) << 8 | (ishift - smin) << 16
    {
        int len = 1, smin = g.ishift, smax = g.ishift;
        if (g.row[0] >= 0 && !ep.cellmode) len = group_len(cent, c0, nc, c, gm.dt, smin, smax);
        g.pad = len | ((smax - g.ishift) << 8) | ((g.ishift - smin) << 16);
    }
    // natural span of the synthetic strips (seismogram.f90:102-130 + sparse_trace.f90:648-668): union over
    // centroids of [first + shift, last + shift + 1], horizontals (all share one span, :196-197) and vertical
    // separately; only needed to size the comparator's FFT (comparator.f90:464-486)
    if (spanbuf || spansrc) {
        int lo_1 = 0x7fffffff, hi_1 = -0x7fffffff, lo_2 = 0x7fffffff, hi_2 = -0x7fffffff, lo_d = 0x7fffffff, hi_d = -0x7fffffff;
        if (live && g.row[0] >= 0) {
            const int nn = (g.flags & 1) ? 1 : 4;
            const int nH1 = gm.ng == 10 ? 4 : 3;          // components of the radial sum
            for (int i = 0; i < nlim_h + nlim_d; i++) {   // the components that are added (all of the needed ones, normally)
                const bool horiz = i < nlim_h;
                const int q = horiz ? i : i - nlim_h;
                const int ig = horiz ? (gm.ng == 10 ? (q < 3 ? q : (q == 3 ? 8 : q - 1)) : q) : (q < 3 ? 5 + q : 9);
                int lo = 0x7fffffff, hi = -0x7fffffff;
                for (int k = 0; k < nn; k++) { const int2 sp = span[g.row[k] + ig]; lo = min(lo, sp.x); hi = max(hi, sp.y); }
                if (!horiz) { lo_d = min(lo_d, lo); hi_d = max(hi_d, hi); }
                else if (q < nH1) { lo_1 = min(lo_1, lo); hi_1 = max(hi_1, hi); }
                else { lo_2 = min(lo_2, lo); hi_2 = max(hi_2, hi); }
            }
            if ((g.flags & 2) && nlim_h) {                // rotating branch: both sums get the union before the rotated add
                lo_1 = lo_2 = min(lo_1, lo_2); hi_1 = hi_2 = max(hi_1, hi_2);
            }
            // strip spans of this centroid's contribution: [first + shift, last + shift + 1]; empty stays (+inf, -inf)
            if (hi_1 >= lo_1) { lo_1 += g.ishift; hi_1 += g.ishift + 1; }
            if (hi_2 >= lo_2) { lo_2 += g.ishift; hi_2 += g.ishift + 1; }
            if (nlim_d && hi_d >= lo_d) { lo_d += g.ishift; hi_d += g.ishift + 1; } else { lo_d = 0x7fffffff; hi_d = -0x7fffffff; }
        }
        // Union over the centroids of one (source, receiver): the lanes of a wave that belong to the same receiver are
        // consecutive (idx = r * nc + c), so a segmented suffix reduction leaves the union of each run in its first lane
        // and only that lane goes to memory -- a 64th of the atomics (135 centroids updating the same six words made this
        // kernel 4 x slower per record at cfg5 than at cfg3).
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int kr = __shfl_down(r, off, 64);
            const int a1 = __shfl_down(lo_1, off, 64), b1 = __shfl_down(hi_1, off, 64);
            const int a2 = __shfl_down(lo_2, off, 64), b2 = __shfl_down(hi_2, off, 64);
            const int ad = __shfl_down(lo_d, off, 64), bd = __shfl_down(hi_d, off, 64);
            if (lane + off < 64 && kr == r) {
                lo_1 = min(lo_1, a1); hi_1 = max(hi_1, b1);
                lo_2 = min(lo_2, a2); hi_2 = max(hi_2, b2);
                lo_d = min(lo_d, ad); hi_d = max(hi_d, bd);
            }
        }
        const int rprev = __shfl_up(r, 1, 64);
        if (lane == 0 || rprev != r) {
            if (spanbuf) {                                // per receiver over all sources: [horizontal lo, hi, vertical lo, hi]
                if (max(hi_1, hi_2) >= min(lo_1, lo_2)) { atomicMin(&spanbuf[4 * r + 0], min(lo_1, lo_2)); atomicMax(&spanbuf[4 * r + 1], max(hi_1, hi_2)); }
                if (hi_d >= lo_d) { atomicMin(&spanbuf[4 * r + 2], lo_d); atomicMax(&spanbuf[4 * r + 3], hi_d); }
            }
            if (spansrc) {                               // the same per trial source and strip: data spans of ITS synthetic strips
                int *sp = spansrc + ((size_t)s * ep.nrec + r) * kSpanInts;
                if (hi_1 >= lo_1) { atomicMin(&sp[0], lo_1); atomicMax(&sp[1], hi_1); }
                if (hi_2 >= lo_2) { atomicMin(&sp[2], lo_2); atomicMax(&sp[3], hi_2); }
                if (hi_d >= lo_d) { atomicMin(&sp[4], lo_d); atomicMax(&sp[5], hi_d); }
            }
        }
    }
    if (!live) return;
    if (!out) return;
    const size_t base = (size_t)(c0 - cent_ofs[ep.isrc0]) * ep.nrec + (size_t)r * nc + c;
    out[base] = g;
    if (tab && g.row[0] >= 0) {
        // cell mode: only the coefficient line here, cellgroup_kernel completes the rows of the group starts it finds
        const bool full = !ep.cellmode && (!(g.flags & 4) || starts_group(cent, c0, nc, c, gm.dt));
        bool ez;
        if (gm.ng == 10) ez = write_tab<10>(tab + base * 128, g, span, gm.pitch, rv.sd, full, endz);
        else ez = write_tab<8>(tab + base * 128, g, span, gm.pitch, rv.sd, full, endz);
        if (pairflag && !ez) atomicOr(&pairflag[(size_t)s * ep.nrec + r], 4);
    }
}

// Second geometry pass of the cell mode: consecutive centroids (table order) of one (source, receiver) whose four GF nodes
// are the same -- neighbouring sub-faults of a rupture are hundreds of metres apart, the nodes kilometres -- form a
// group: accumulate_cell_kernel fetches the raw node traces ONCE per group and blends them per centroid with that
// centroid's weights.  Groups are cut greedily from the start of a same-cell run (length <= kMaxGroup, integer shifts
// within the LDS halo).  Pairs accumulate_cell_kernel does not take (see cell_pair) get same-point groups.  Thread per record: a thread
// that finds itself at a group start leaves the hint in its record and completes its descriptor row.
__device__ __forceinline__ bool same_cell(const GeoRec *__restrict__ a, const int (&row)[4])
{
    const int4 r = *reinterpret_cast<const int4 *>(a->row);
    return r.x == row[0] && r.y == row[1] && r.z == row[2] && r.w == row[3];
}

// is (source s, receiver r) evaluated by accumulate_cell_kernel?  Receivers with horizontal AND vertical components whose
// centroids all find ALL their traces (pairflag bits 0 and 1: none partial, none skipped); every other pair keeps same-point
// groups and goes through accumulate_grouped_kernel
__device__ __forceinline__ bool cell_pair(const RecvDev &rv, const int *__restrict__ pairflag, int s, int nrec, int r)
{
    return rv.need_h && rv.has_d && !(pairflag[(size_t)s * nrec + r] & 3);
}

// (the (group of NS sources, receiver) combinations accumulate_multi_kernel takes; the same rule as multi_taken() further down)
template <int NS>
__device__ __forceinline__ bool multi_taken_fwd(const RecvDev &rv, const int *__restrict__ pairflag, const int *__restrict__ mate,
                                                int s, int nrec, int r)
{
    if (!mate) return false;
    const int a = s - s % NS;
    if (!mate[a / NS] || !rv.need_h || !rv.has_d) return false;
    int f = 0;
#pragma unroll
    for (int i = 0; i < NS; i++) f |= pairflag[(size_t)(a + i) * nrec + r];
    return f == 0;
}

__global__ __launch_bounds__(256) void cellgroup_kernel(const int *__restrict__ cent_ofs, EvalParams ep, GfMeta gm,
                                                        const int2 *__restrict__ span, const RecvDev *__restrict__ recv,
                                                        GeoRec *__restrict__ recs, int *__restrict__ tab,
                                                        const int *__restrict__ pairflag, const unsigned char *__restrict__ endz,
                                                        const int *__restrict__ synrow, int cell_range /* largest shift range of a cell group */)
{
    const int s = blockIdx.y;
    if (synrow && synrow[s] != s) return;
    const int cb = cent_ofs[ep.isrc0], c0 = cent_ofs[ep.isrc0 + s], nc = cent_ofs[ep.isrc0 + s + 1] - c0;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nc * ep.nrec) return;
    const int r = idx / nc, c = idx - r * nc;
    const size_t base0 = (size_t)(c0 - cb) * ep.nrec + (size_t)r * nc;
    GeoRec *__restrict__ rr = recs + base0;
    const GeoRec me = rr[c];
    if (me.row[0] < 0) return;
    int len = 1, smin = me.ishift, smax = me.ishift;
    {
        // cell pairs: a group runs on while the four nodes stay the same; other pairs: while the POINT stays the same
        // (flags bit 2 of the follower), which is what accumulate_grouped_kernel's shared blended tile needs
        const bool cellp = cell_pair(recv[r], pairflag, s, ep.nrec, r);
        int r0 = c;
        while (r0 > 0 && same_cell(rr + r0 - 1, me.row) && (cellp || (rr[r0].flags & 4))) r0--;
        int pos = r0;
        for (;;) {
            len = 1;
            smin = smax = rr[pos].ishift;
            for (int k = pos + 1; k < nc && len < kMaxGroup; k++) {
                if (!same_cell(rr + k, me.row) || !(cellp || (rr[k].flags & 4))) break;
                const int sh = rr[k].ishift;
                const int nmin = min(smin, sh), nmax = max(smax, sh);
                if (nmax - nmin > (cellp ? cell_range : kHalo - 10)) break;
                smin = nmin; smax = nmax; len++;
            }
            if (pos == c) break;                  // this centroid starts a group
            if (pos + len > c) return;            // inside a group: its coefficient line is all the kernel reads of its row
            pos += len;
        }
    }
    rr[c].pad = len | ((smax - me.ishift) << 8) | ((me.ishift - smin) << 16);
    if (gm.ng == 10) write_tab<10>(tab + (base0 + c) * 128, me, span, gm.pitch, recv[r].sd, true, endz);
    else write_tab<8>(tab + (base0 + c) * 128, me, span, gm.pitch, recv[r].sd, true, endz);
}

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
// accumulate

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte load

// Raw buffer loads for the rows of one group: the descriptor holds the group base (64 bit), the row's start comes as the
// instruction's SCALAR offset and the lane's position as its 32-bit vector offset -- no per-load vector address arithmetic
// (a global_load needs a 64-bit VGPR address per load: one v_lshl_add_u64 each, 40 per group).
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gf_rsrc(const float *base)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, -1, 0x00020000);      // raw buffer, 32-bit data, no range limit below 4 GB
}
__device__ __forceinline__ f4u buf_load4(__amdgpu_buffer_rsrc_t r, int voff_floats, int soff_floats)
{
    // byte offsets as unsigned 32-bit values: the host keeps a cell below 2^30 floats (kiwi_hip_set_gfdb / kiwi_hip_set_interp),
    // so neither wraps
    const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(4u * (unsigned)voff_floats), (int)(4u * (unsigned)soff_floats), 0);
    return f4u{ __int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w) };
}

// five consecutive samples l-1 .. l+3 of one GF row (l relative to the row's first sample).
// The row is stored as [kRowPad zeros | n samples | end value repeated up to pitch], so clamping
// the start position implements "zero before the span, last value repeated after it"
// (sparse_trace.f90:337-338,696-703) without branches.
__device__ __forceinline__ void load5(const float *__restrict__ rowp, int l, int pitch, float (&v)[5])
{
    int q = kRowPad + l - 1;
    q = min(max(q, 0), pitch - 5);
    const float *p = rowp + q;
    v[0] = p[0];
    const f4u t = *(const f4u *)(p + 1);
    v[1] = t.x; v[2] = t.y; v[3] = t.z; v[4] = t.w;
}

// one Green's function component of one centroid onto 4 consecutive output samples.
// sparse_trace.f90:640-703 (shift, linear sub-sample interpolation, repeated end point) on top
// of gfdb.f90:944-949 (4-neighbour blend, summed in the order t00,t01,t10,t11).
template <bool BLEND>
__device__ __forceinline__ void gf_add(float (&out)[4], const float *__restrict__ G, const int2 *__restrict__ span,
                                       int pitch, const GeoRec &g, int ig, float factor, int j0)
{
    float b[5];
    int jend;
    const int r0 = g.row[0] + ig;
    const int2 s0 = span[r0];
    if (BLEND) {
        const int r1 = g.row[1] + ig, r2 = g.row[2] + ig, r3 = g.row[3] + ig;
        const int2 s1 = span[r1], s2 = span[r2], s3 = span[r3];
        jend = max(max(s0.y, s1.y), max(s2.y, s3.y));
        float v0[5], v1[5], v2[5], v3[5];
        load5(G + (size_t)r0 * pitch, j0 - s0.x, pitch, v0);
        load5(G + (size_t)r1 * pitch, j0 - s1.x, pitch, v1);
        load5(G + (size_t)r2 * pitch, j0 - s2.x, pitch, v2);
        load5(G + (size_t)r3 * pitch, j0 - s3.x, pitch, v3);
#pragma unroll
        for (int i = 0; i < 5; i++) {
            float acc = g.w[0] * v0[i];
            acc = acc + g.w[1] * v1[i];
            acc = acc + g.w[2] * v2[i];
            acc = acc + g.w[3] * v3[i];
            b[i] = acc;
        }
    } else {
        jend = s0.y;
        load5(G + (size_t)r0 * pitch, j0 - s0.x, pitch, b);
    }
    float wr = g.wfrac;
    float wl = 1.f - wr;
    wr = wr * factor;
    wl = wl * factor;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const bool tail = (j0 + i) > jend;            // sparse_trace.f90:698-703
        const float c1 = tail ? factor : wl;
        const float c2 = tail ? 0.f : wr;
        out[i] = out[i] + c1 * b[i + 1];
        out[i] = out[i] + c2 * b[i];
    }
}

template <int NG, bool BLEND>
__device__ __forceinline__ void centroid_add(float (&ar1)[4], float (&ar2)[4], float (&dz)[4],
                                             const float *__restrict__ G, const int2 *__restrict__ span, int pitch,
                                             const GeoRec &g, const RecvDev &rv, int j0)
{
    // components that are added, in application order (all of them unless a trace is missing: geometry_kernel)
    const int nlh = (g.flags & 8) ? (g.flags >> 8) & 15 : 15, nld = (g.flags & 8) ? (g.flags >> 12) & 15 : 15;
    constexpr int o9 = (NG == 10) ? 1 : 0;       // position shift behind the optional near-field component
#define GA(acc, pos, lim, ig, fac) do { if ((pos) < (lim)) gf_add<BLEND>(acc, G, span, pitch, g, ig, fac, j0); } while (0)
    if (rv.need_h && nlh > 0) {
        if (g.flags & 2) {                       // seismogram.f90:160-203 (a missing trace leaves before the sums are added)
            float t1[4] = { 0.f, 0.f, 0.f, 0.f }, t2[4] = { 0.f, 0.f, 0.f, 0.f };
            gf_add<BLEND>(t1, G, span, pitch, g, 0, g.f[0], j0);
            gf_add<BLEND>(t1, G, span, pitch, g, 1, g.f[1], j0);
            gf_add<BLEND>(t1, G, span, pitch, g, 2, g.f[2], j0);
            if (NG == 10) gf_add<BLEND>(t1, G, span, pitch, g, 8, g.f[5], j0);
            gf_add<BLEND>(t2, G, span, pitch, g, 3, g.f[3], j0);
            gf_add<BLEND>(t2, G, span, pitch, g, 4, g.f[4], j0);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ar1[i] = ar1[i] + g.cl * t1[i] - g.sl * t2[i];
                ar2[i] = ar2[i] + g.cl * t2[i] + g.sl * t1[i];
            }
        } else {                                 // seismogram.f90:205-231
            GA(ar1, 0, nlh, 0, g.f[0]);
            GA(ar1, 1, nlh, 1, g.f[1]);
            GA(ar1, 2, nlh, 2, g.f[2]);
            if (NG == 10) GA(ar1, 3, nlh, 8, g.f[5]);
            GA(ar2, 3 + o9, nlh, 3, g.f[3]);
            GA(ar2, 4 + o9, nlh, 4, g.f[4]);
        }
    }
    if (rv.has_d) {                              // seismogram.f90:236-253
        GA(dz, 0, nld, 5, g.f[0] * rv.sd);
        GA(dz, 1, nld, 6, g.f[1] * rv.sd);
        GA(dz, 2, nld, 7, g.f[2] * rv.sd);
        if (NG == 10) GA(dz, 3, nld, 9, g.f[5] * rv.sd);
    }
#undef GA
}

template <int NG>
__global__ __launch_bounds__(256) void accumulate_kernel(
    const float *__restrict__ G, const int2 *__restrict__ span, int pitch,
    const GeoRec *__restrict__ recs, const int *__restrict__ cent_ofs, int isrc0, int nrec,
    const RecvDev *__restrict__ recv, float *__restrict__ syn, size_t syn_stride, const int *__restrict__ synrow)
{
    const int tile = blockIdx.x, r = blockIdx.y, s = blockIdx.z;
    if (synrow && synrow[s] != s) return;
    const RecvDev &rv = recv[r];
    if (!rv.enabled) return;
    if (tile * kTile >= rv.wlen) return;
    const int tl = tile * kTile + threadIdx.x * 4;
    const int t0 = rv.wbeg + tl;
    const int cb = cent_ofs[isrc0], c0 = cent_ofs[isrc0 + s], nc = cent_ofs[isrc0 + s + 1] - c0;
    const GeoRec *__restrict__ rc = recs + ((size_t)(c0 - cb) * nrec + (size_t)r * nc);

    float ar1[4] = { 0.f, 0.f, 0.f, 0.f }, ar2[4] = { 0.f, 0.f, 0.f, 0.f }, dz[4] = { 0.f, 0.f, 0.f, 0.f };
    for (int c = 0; c < nc; c++) {               // seismogram.f90:131, table order
        const GeoRec &g = rc[c];
        if (g.row[0] < 0) continue;              // 'cycle' on a missing trace
        const int j0 = t0 - g.ishift;            // strip(x) += trace(x - itraceshift), sparse_trace.f90:605
        if (g.flags & 1) centroid_add<NG, false>(ar1, ar2, dz, G, span, pitch, g, rv, j0);
        else             centroid_add<NG, true>(ar1, ar2, dz, G, span, pitch, g, rv, j0);
    }

    if (tl >= rv.wlen) return;
    float *__restrict__ so = syn + (size_t)s * syn_stride + tl;
    for (int k = 0; k < rv.ncomp; k++) {         // seismogram.f90:256-283
        float4 o;
        const float sg = rv.sign[k];
        switch (rv.comp[k]) {
        case 1: o = make_float4(ar1[0] * sg, ar1[1] * sg, ar1[2] * sg, ar1[3] * sg); break;
        case 2: o = make_float4(ar2[0] * sg, ar2[1] * sg, ar2[2] * sg, ar2[3] * sg); break;
        case 3: o = make_float4(dz[0], dz[1], dz[2], dz[3]); break;       // sign already in the factors (:239)
        case 4: {
            float a[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = (rv.cl0 * ar1[i] - rv.sl0 * ar2[i]) * sg;
            o = make_float4(a[0], a[1], a[2], a[3]); break; }
        default: {
            float a[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = (rv.cl0 * ar2[i] + rv.sl0 * ar1[i]) * sg;
            o = make_float4(a[0], a[1], a[2], a[3]); break; }
        }
        *(float4 *)(so + rv.synofs[k]) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// accumulate, grouped: blended GF tiles staged in LDS and reused by every centroid of a sub-fault
//
// Consecutive centroids that sit at the same point (the nt source-time-function steps of one
// sub-fault) need the SAME blended traces, only shifted by a different number of samples and
// weighted differently.  The workgroup therefore blends each needed GF component ONCE per group
// into an LDS tile (coalesced 16-byte global loads, 16-byte LDS stores) that covers the tile plus
// the spread of the group's integer shifts, and then every centroid of the group does its
// shift-interpolate-accumulate from LDS (conflict-free: lane l owns samples l, l+256, l+512, l+768
// of the tile).  Global traffic drops by the group size (5 for the benchmark's bilateral source);
// per-sample operation order is unchanged, so results are bit-identical to accumulate_kernel.


// A GeoRec travels through the grouped kernel "lane-distributed": lane i (< 20) of every wave
// holds dword i of the record in ONE VGPR (a single coalesced 80-byte load that can be issued a
// whole centroid ahead), and fields are broadcast to SGPRs with v_readlane when needed.  Compared
// with scalar loads this removes ~16 serialised SMEM round trips per centroid.
__device__ __forceinline__ int rec_load(const GeoRec *__restrict__ rc, int c, int nc, int lane)
{
    int v = 0;
    if (c < nc && lane < 20) v = ((const int *)(rc + c))[lane];
    return v;
}
#define REC_I(v, k) __builtin_amdgcn_readlane((v), (k))
#define REC_F(v, k) __int_as_float(__builtin_amdgcn_readlane((v), (k)))

__device__ __forceinline__ void rec_head(int v, int o, GeoRec &g)
{
    g.row[0] = REC_I(v, o + 0); g.row[1] = REC_I(v, o + 1); g.row[2] = REC_I(v, o + 2); g.row[3] = REC_I(v, o + 3);
    g.w[0] = REC_F(v, o + 4); g.w[1] = REC_F(v, o + 5); g.w[2] = REC_F(v, o + 6); g.w[3] = REC_F(v, o + 7);
    g.ishift = REC_I(v, o + 8);
    g.flags = REC_I(v, o + 18);
    g.pad = REC_I(v, o + 19);
}

__device__ __forceinline__ f4u load4(const float *__restrict__ rowp, int l, int pitch)
{
    int q = kRowPad + l;
    q = min(max(q, 0), pitch - 4);           // zeros before the span, end value after it (see load5)
    return *(const f4u *)(rowp + q);
}

// Load descriptors (see geometry_kernel): ta / tb are the lane-distributed halves of the group's descriptor row;
// the float index of sample j of node k's trace of component ig is
// clamp(ta[4ig+k] + j, tb[4ig+k], tb[4ig+k] + pitch - 4), which implements "zero before the span, end value
// repeated after it" on the padded row.

// Build the main chunk (LDS positions [4*tid, 4*tid+4)) of SEVERAL components at once: all 4*N
// loads are issued before the first blend so that one L2 round trip is paid per batch, not per
// component (left to itself the compiler serialises load -> blend -> ds_write per component).
// FAST: every row of the group covers the whole tile without clamping (checked once per group), so a
// load address is a wave-uniform row base (SGPR pair) plus the lane's position -- no per-load VALU
// address arithmetic.
template <bool BLEND, bool FAST, int N>
__device__ __forceinline__ void build_batch(float *__restrict__ tile0, int lds_tile, const int (&igs)[N], int p, int jb,
                                            const float *__restrict__ G, int pitch, int ta, int tb, const GeoRec &g)
{
    const int j = jb + p;
    f4u v[N][BLEND ? 4 : 1];
#pragma unroll
    for (int q = 0; q < N; q++) {
#pragma unroll
        for (int k = 0; k < (BLEND ? 4 : 1); k++) {
            const int base = REC_I(ta, 4 * igs[q] + k);
            if constexpr (FAST) {
                v[q][k] = buf_load4(gf_rsrc(G), p, base + jb);       // scalar row offset + the lane's position
            } else {
                const int lo = REC_I(tb, 4 * igs[q] + k);
                const int idx = min(max(base + j, lo), lo + pitch - 4);
                v[q][k] = *(const f4u *)(G + (size_t)(unsigned)idx);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < N; q++) {
        f4u b;
        if constexpr (BLEND) {
            b = g.w[0] * v[q][0];             // gfdb.f90:946-949, summed in this order
            b = b + g.w[1] * v[q][1];
            b = b + g.w[2] * v[q][2];
            b = b + g.w[3] * v[q][3];
        } else {
            b = v[q][0];
        }
        *(float4 *)(tile0 + igs[q] * lds_tile + p) = make_float4(b.x, b.y, b.z, b.w);
    }
}

// Streamed form of build_batch for ALL components: sub-batches of two components (8 loads); two sub-batches are
// kept in flight, the blend of sub-batch i is followed by the issue of sub-batch i + 2.  The whole group then costs
// about one L2 round trip plus the issue time instead of one round trip per batch, with fewer registers in flight.
template <bool BLEND, bool FAST>
__device__ __forceinline__ void pair_issue(f4u (&v)[2][BLEND ? 4 : 1], int ig0, int p, int jb, const float *__restrict__ G,
                                           int pitch, int ta, int tb)
{
    const int j = jb + p;
#pragma unroll
    for (int q = 0; q < 2; q++) {
#pragma unroll
        for (int k = 0; k < (BLEND ? 4 : 1); k++) {
            const int base = REC_I(ta, 4 * (ig0 + q) + k);
            if constexpr (FAST) {
                v[q][k] = buf_load4(gf_rsrc(G), p, base + jb);
            } else {
                const int lo = REC_I(tb, 4 * (ig0 + q) + k);
                const int idx = min(max(base + j, lo), lo + pitch - 4);
                v[q][k] = *(const f4u *)(G + (size_t)(unsigned)idx);
            }
        }
    }
}

template <bool BLEND>
__device__ __forceinline__ void pair_finish(const f4u (&v)[2][BLEND ? 4 : 1], float *__restrict__ tile0, int lds_tile, int ig0, int p,
                                            const GeoRec &g)
{
#pragma unroll
    for (int q = 0; q < 2; q++) {
        f4u b;
        if constexpr (BLEND) {
            b = g.w[0] * v[q][0];             // gfdb.f90:946-949, summed in this order
            b = b + g.w[1] * v[q][1];
            b = b + g.w[2] * v[q][2];
            b = b + g.w[3] * v[q][3];
        } else {
            b = v[q][0];
        }
        *(float4 *)(tile0 + (ig0 + q) * lds_tile + p) = make_float4(b.x, b.y, b.z, b.w);
    }
}

// The halo (LDS positions TILE .. npos, at most kHalo / 4 = 16 chunks of 4 samples per component) is built by
// one lane per (component, chunk) pair so that its loads travel in the same L2 round trip as the second main batch
// instead of costing a round trip of their own.  Per-lane descriptors come from the lane-distributed row by
// ds_bpermute.  halo_issue only loads; halo_finish blends and stores.
struct HaloRegs { f4u v[4]; };

template <bool BLEND, bool FAST>
__device__ __forceinline__ HaloRegs halo_issue(bool active, int ig, int ph, int jb, const float *__restrict__ G, int pitch,
                                               int ta, int tb)
{
    HaloRegs h;
#pragma unroll
    for (int k = 0; k < (BLEND ? 4 : 1); k++) {
        const int base = __shfl(ta, 4 * ig + k, 64);
        int idx = base + jb + ph;
        if constexpr (!FAST) {
            const int lo = __shfl(tb, 4 * ig + k, 64);
            idx = min(max(idx, lo), lo + pitch - 4);
        }
        h.v[k] = f4u{ 0.f, 0.f, 0.f, 0.f };
        if (active) h.v[k] = *(const f4u *)(G + (size_t)(unsigned)idx);
    }
    return h;
}

template <bool BLEND>
__device__ __forceinline__ void halo_finish(bool active, const HaloRegs &h, float *__restrict__ tile0, int lds_tile, int ig, int ph,
                                            const GeoRec &g)
{
    f4u b;
    if constexpr (BLEND) {
        b = g.w[0] * h.v[0];                  // gfdb.f90:946-949, summed in this order
        b = b + g.w[1] * h.v[1];
        b = b + g.w[2] * h.v[2];
        b = b + g.w[3] * h.v[3];
    } else {
        b = h.v[0];
    }
    if (active) *(float4 *)(tile0 + ig * lds_tile + ph) = make_float4(b.x, b.y, b.z, b.w);
}

typedef float f2v __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) float *lds_cfp;

// Apply-phase mapping of samples to lanes: lane l of wave w owns the four tile samples 256 w + l + 64 q, q = 0..3 (stride
// 64, not four consecutive ones).  The two blended samples an output needs, b[j-1] and b[j], then sit at LDS dwords
// base + 64 q and base + 64 q + 1 of the lane: consecutive lanes read consecutive dwords (no bank conflicts whatever the
// shift), the compiler fuses the reads of two q into one ds_read2st64_b32 whose result IS the aligned register pair a
// packed multiply wants -- for every shift residue alike, so no v_pk_mov assemblies and no per-residue code variants.
// LDS_TILE is a multiple of 64 dwords, which puts all components and q of a centroid within the instruction's offsets.
// NP register pairs per lane: 2 (four outputs per lane: q0,q1 | q2,q3) or 1 (two outputs per lane, accumulate_cell_kernel)
template <int NP = 2> struct TileRegsN { f2v lo[NP], hi[NP]; };   // b[j-1] and b[j] of the lane's outputs
typedef TileRegsN<2> TileRegs;

// The lane's two LDS base addresses of a centroid: b[j-1] and b[j] of its output q = 0 in component 0.  The second is
// the first plus one dword, but hidden from the compiler: it would otherwise pair the adjacent dwords (b[j-1], b[j]) of
// ONE output into a ds_read2_b32 and then shuffle registers, instead of pairing the same quantity of TWO outputs.
struct TileBase { const __attribute__((address_space(3))) float *lo, *hi; };

__device__ __forceinline__ TileBase tile_base(const float *__restrict__ p)
{
    typedef const __attribute__((address_space(3))) float *lds_fp;
    TileBase b;
    b.lo = (lds_fp)p;
    unsigned a = (unsigned)(size_t)(b.lo + 1);
    asm volatile("" : "+v"(a));
    b.hi = (lds_fp)(size_t)a;
    return b;
}

template <int NP = 2>
__device__ __forceinline__ TileRegsN<NP> tile_load(const TileBase &b, int ofs)
{
    TileRegsN<NP> t;
#pragma unroll
    for (int h = 0; h < NP; h++) {
        t.lo[h] = f2v{ b.lo[ofs + 128 * h], b.lo[ofs + 128 * h + 64] };
        t.hi[h] = f2v{ b.hi[ofs + 128 * h], b.hi[ofs + 128 * h + 64] };
    }
    return t;
}

// The lane's 4 output samples are held as two register pairs so that the multiplies and adds are
// v_pk_mul_f32 / v_pk_add_f32 (two IEEE fp32 operations per lane and instruction, each rounded
// separately exactly like the scalar form; no FMA).
template <bool TAIL, int NP = 2>
__device__ __forceinline__ void tile_fma(f2v (&out)[NP], const TileRegsN<NP> &t, int jl, int jend, float factor, float wl, float wr)
{
#pragma unroll
    for (int h = 0; h < NP; h++) {
        f2v c1 = { wl, wl }, c2 = { wr, wr };
        if (TAIL) {                               // sparse_trace.f90:698-703, jl = trace index of b[j-1] of the lane's output q = 0
            const bool t0 = (jl + 128 * h + 1) > jend, t1 = (jl + 128 * h + 64 + 1) > jend;
            c1.x = t0 ? factor : wl; c2.x = t0 ? 0.f : wr;
            c1.y = t1 ? factor : wl; c2.y = t1 ? 0.f : wr;
        }
        out[h] = out[h] + c1 * t.hi[h];
        out[h] = out[h] + c2 * t.lo[h];
    }
}

template <bool TAIL, int NP = 2>
__device__ __forceinline__ void tile_add(f2v (&out)[NP], const TileBase &b, int ofs, int jl, int jend,
                                         float factor, float wfrac)
{
    const TileRegsN<NP> t = tile_load<NP>(b, ofs);
    float wr = wfrac;
    float wl = 1.f - wr;
    wr = wr * factor;
    wl = wl * factor;
    tile_fma<TAIL, NP>(out, t, jl, jend, factor, wl, wr);
}

// all GF components of one centroid (reference order)
template <int NG, int LDS_TILE, bool TAIL, int NP = 2>
__device__ __forceinline__ void centroid_apply(f2v (&ar1)[NP], f2v (&ar2)[NP], f2v (&dz)[NP],
                                               const TileBase &chunk0, int jl, const int (&jend)[NG],
                                               bool need_h, bool has_d, int flags, float wfrac, float sd,
                                               float f0, float f1, float f2, float f3, float f4, float f5,
                                               float cl, float sl)
{
    // components that are added, in application order (all of them unless a trace is missing: geometry_kernel)
    const int nlh = (flags & 8) ? (flags >> 8) & 15 : 15, nld = (flags & 8) ? (flags >> 12) & 15 : 15;
    constexpr int o9 = (NG == 10) ? 1 : 0;
#define TADD(acc, ig, fac) tile_add<TAIL, NP>(acc, chunk0, (ig) * LDS_TILE, jl, jend[ig], fac, wfrac)
#define TADDL(acc, pos, lim, ig, fac) do { if ((pos) < (lim)) TADD(acc, ig, fac); } while (0)
    if (need_h && nlh > 0) {
        if (flags & 2) {                         // seismogram.f90:160-203
            f2v t1[NP], t2[NP];
#pragma unroll
            for (int i = 0; i < NP; i++) { t1[i] = f2v{ 0.f, 0.f }; t2[i] = f2v{ 0.f, 0.f }; }
            TADD(t1, 0, f0); TADD(t1, 1, f1); TADD(t1, 2, f2);
            if constexpr (NG == 10) TADD(t1, 8, f5);
            TADD(t2, 3, f3); TADD(t2, 4, f4);
#pragma unroll
            for (int i = 0; i < NP; i++) {
                ar1[i] = ar1[i] + cl * t1[i] - sl * t2[i];
                ar2[i] = ar2[i] + cl * t2[i] + sl * t1[i];
            }
        } else {                                 // seismogram.f90:205-231
            TADDL(ar1, 0, nlh, 0, f0); TADDL(ar1, 1, nlh, 1, f1); TADDL(ar1, 2, nlh, 2, f2);
            if constexpr (NG == 10) TADDL(ar1, 3, nlh, 8, f5);
            TADDL(ar2, 3 + o9, nlh, 3, f3); TADDL(ar2, 4 + o9, nlh, 4, f4);
        }
    }
    if (has_d) {                                 // seismogram.f90:236-253
        TADDL(dz, 0, nld, 5, f0 * sd); TADDL(dz, 1, nld, 6, f1 * sd); TADDL(dz, 2, nld, 7, f2 * sd);
        if constexpr (NG == 10) TADDL(dz, 3, nld, 9, f5 * sd);
    }
#undef TADDL
#undef TADD
}

// The common case (receiver with horizontal and vertical components), software-pipelined: the LDS reads
// of component i + kAhead are issued before the arithmetic of component i, so that a lone wave is not
// stalled for a full LDS round trip per component (a wave can only issue every 4th cycle; with 3 waves per
// SIMD exposed latency is what bounds this kernel).  Same operations in the same order as centroid_apply.
template <int NG, int LDS_TILE, bool TAIL, bool SCOEF, int NP = 2>
__device__ __forceinline__ void centroid_apply_hd(f2v (&ar1)[NP], f2v (&ar2)[NP], f2v (&dz)[NP],
                                                  const TileBase &chunk0, int jl, const int (&jend)[NG],
                                                  int flags, const float *__restrict__ coef, int rec, float sd,
                                                  float cl, float sl)
{
    // rec: the lane-distributed record of the centroid; its weights and interpolation fraction are fetched only where
    // the variant needs them (the tail rule's `factor`, or coefficients computed in registers)
    float wfrac = 0.f, f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f, f4 = 0.f, f5 = 0.f;
    if constexpr (TAIL || !SCOEF) {
        wfrac = REC_F(rec, 9);
        f0 = REC_F(rec, 10); f1 = REC_F(rec, 11); f2 = REC_F(rec, 12); f3 = REC_F(rec, 13); f4 = REC_F(rec, 14); f5 = REC_F(rec, 15);
    }
    // coef: the centroid's 2 * NG interpolation coefficients (wl, wr per component in application order), computed
    // by geometry_kernel; the pointer is wave-uniform, so these are scalar loads and the coefficients reach the
    // packed multiplies as SGPR operands -- no vector instructions spent on them
    constexpr int kAhead = 2;
    constexpr int seq10[10] = { 0, 1, 2, 8, 3, 4, 5, 6, 7, 9 }, seq8[8] = { 0, 1, 2, 3, 4, 5, 6, 7 };
    constexpr int nH1 = (NG == 10) ? 4 : 3;      // components summed into the radial trace
    const float fac10[10] = { f0, f1, f2, f5, f3, f4, f0 * sd, f1 * sd, f2 * sd, f5 * sd };
    const float fac8[8] = { f0, f1, f2, f3, f4, f0 * sd, f1 * sd, f2 * sd };
    const bool rot = (flags & 2) != 0;           // seismogram.f90:160-203 vs :205-231
    f2v t1[NP], t2[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) { t1[k] = rot ? f2v{ 0.f, 0.f } : ar1[k]; t2[k] = rot ? f2v{ 0.f, 0.f } : ar2[k]; }
    float cw[2 * NG];
    if constexpr (SCOEF) {                        // scalar loads, issued together with the first LDS reads
#pragma unroll
        for (int i = 0; i < 2 * NG; i++) cw[i] = coef[i];
    } else {                                      // short sources (fused variant): computed here, as centroid_apply does
        const float wr0 = wfrac, wl0 = 1.f - wfrac;
#pragma unroll
        for (int i = 0; i < NG; i++) { const float f = (NG == 10) ? fac10[i] : fac8[i]; cw[2 * i] = wl0 * f; cw[2 * i + 1] = wr0 * f; }
    }
    TileRegsN<NP> tr[NG];
#pragma unroll
    for (int i = 0; i < kAhead; i++) tr[i] = tile_load<NP>(chunk0, ((NG == 10) ? seq10[i] : seq8[i]) * LDS_TILE);
#pragma unroll
    for (int i = 0; i < NG; i++) {
        if (i + kAhead < NG) tr[i + kAhead] = tile_load<NP>(chunk0, ((NG == 10) ? seq10[i + kAhead] : seq8[i + kAhead]) * LDS_TILE);
        __builtin_amdgcn_sched_barrier(0);
        const int ig = (NG == 10) ? seq10[i] : seq8[i];
        const float fac = (NG == 10) ? fac10[i] : fac8[i];
        const float wl = cw[2 * i], wr = cw[2 * i + 1];
        if (i < nH1) tile_fma<TAIL, NP>(t1, tr[i], jl, jend[ig], fac, wl, wr);
        else if (i < nH1 + 2) tile_fma<TAIL, NP>(t2, tr[i], jl, jend[ig], fac, wl, wr);
        else tile_fma<TAIL, NP>(dz, tr[i], jl, jend[ig], fac, wl, wr);
        if (i == nH1 + 1) {
            if (rot) {
#pragma unroll
                for (int k = 0; k < NP; k++) {
                    ar1[k] = ar1[k] + cl * t1[k] - sl * t2[k];
                    ar2[k] = ar2[k] + cl * t2[k] + sl * t1[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < NP; k++) { ar1[k] = t1[k]; ar2[k] = t2[k]; }
            }
        }
    }
}

// acc + d * d in fp64: the square of an fp32 value is exact in fp64 (48 significant bits), so the fused form rounds once,
// exactly like the exact product followed by the add (comparator.f90:650-659 accumulates in real*8)
__device__ __forceinline__ double sq_acc(double acc, float d) { return fma((double)d, (double)d, acc); }

// Sum (or maximum of non-negative values) over the 64 lanes of a wave in fp64, through DPP moves instead of LDS permutes:
// inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8; lanes shifted in from outside the row read 0), then the
// row totals travel up (row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3).  The result is valid in lane 63;
// the order of the additions is fixed.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <bool IS_MAX>
__device__ __forceinline__ double wave_reduce_f64_t(double v)
{
#define KIWI_STEP(CTRL, MASK) do { const double o = dpp_f64<CTRL, MASK>(v); v = IS_MAX ? fmax(v, o) : v + o; } while (0)
    KIWI_STEP(0x111, 0xf);      // row_shr:1
    KIWI_STEP(0x112, 0xf);      // row_shr:2
    KIWI_STEP(0x114, 0xf);      // row_shr:4
    KIWI_STEP(0x118, 0xf);      // row_shr:8
    KIWI_STEP(0x142, 0xa);      // row_bcast:15 -> rows 1, 3
    KIWI_STEP(0x143, 0xc);      // row_bcast:31 -> rows 2, 3
#undef KIWI_STEP
    return v;
}
__device__ __forceinline__ double wave_reduce_f64(double v, bool is_max)      // (is_max is wave-uniform: one branch, not a select per step)
{
    return is_max ? wave_reduce_f64_t<true>(v) : wave_reduce_f64_t<false>(v);
}

// ---- epilogue of the accumulate kernels with four outputs per lane (samples tl + 64 q, q = 0 .. 3, held as two register pairs)
// one output trace of the receiver from the accumulators: rotation to N/E and sign (seismogram.f90:256-283), packed
__device__ __forceinline__ void out4(int comp, float sg, float cl0, float sl0, const f2v (&ar1)[2], const f2v (&ar2)[2], const f2v (&dz)[2],
                                     f2v &o01, f2v &o23)
{
    const f2v s2 = { sg, sg }, c2 = { cl0, cl0 }, n2 = { sl0, sl0 };
    switch (comp) {
    case 1: o01 = ar1[0] * s2; o23 = ar1[1] * s2; break;
    case 2: o01 = ar2[0] * s2; o23 = ar2[1] * s2; break;
    case 3: o01 = dz[0]; o23 = dz[1]; break;
    case 4: o01 = (c2 * ar1[0] - n2 * ar2[0]) * s2; o23 = (c2 * ar1[1] - n2 * ar2[1]) * s2; break;
    default: o01 = (c2 * ar2[0] + n2 * ar1[0]) * s2; o23 = (c2 * ar2[1] + n2 * ar1[1]) * s2; break;
    }
}
// the fused comparator over the lane's four samples (what misfit_kernel does per sample: comparator.f90:264,1173-1184,627-667);
// `whole`: all of them lie inside the window (workgroup-uniform).  l2norm with unit factor, the norm of every grid search, has a
// straight-line packed form -- the same operations in the same order as the general loop
__device__ __forceinline__ double fused_acc4(const FuseParams &fp, f2v o01, f2v o23, float mom, const float *__restrict__ rt,
                                             const float *__restrict__ tp, int tl, int wlen, bool whole)
{
    const bool unit = (fp.syn_factor == 1.f);
    double acc = 0.0;
    if (whole && unit && fp.method == 1) {
        const f2v m2 = { mom, mom };
        const f2v t01 = { tp[0], tp[64] }, t23 = { tp[128], tp[192] }, r01 = { rt[0], rt[64] }, r23 = { rt[128], rt[192] };
        const f2v d01 = r01 - (o01 * m2) * t01, d23 = r23 - (o23 * m2) * t23;
        acc = sq_acc(acc, d01.x); acc = sq_acc(acc, d01.y); acc = sq_acc(acc, d23.x); acc = sq_acc(acc, d23.y);
        return acc;
    }
    const float o[4] = { o01.x, o01.y, o23.x, o23.y };
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (tl + 64 * i >= wlen) break;
        const float v = o[i] * mom;
        const float vt = v * tp[64 * i];
        const float a = rt[64 * i];
        switch (fp.method) {
        case 1: { const float d = unit ? (a - vt) : (1.f * a - fp.syn_factor * vt); acc = sq_acc(acc, d); break; }
        case 2: { const float d = unit ? fabsf(a - vt) : fabsf(1.f * a - fp.syn_factor * vt); acc += (double)d; break; }
        case 5: acc += unit ? (double)(a * vt) : (double)(a * 1.f * vt * fp.syn_factor); break;
        default: { const double x = (double)(1.f * a), y = (double)(fp.syn_factor * vt); acc = fmax(acc, sqrt(x * x + y * y)); break; }
        }
    }
    return acc;
}

#ifndef KIWI_GROUPED_WAVES
#define KIWI_GROUPED_WAVES 3
#endif
#ifndef KIWI_X_FULL
#define KIWI_X_FULL(x) (x)       // measurement switch: -DKIWI_X_FULL(x)=0x7fff makes every centroid read both register sets
#endif
// ------------------------------------------------------------------------------------------------
// Register pairs carried between the centroids of a group (round 3), accumulate_grouped_kernel with 256 threads.
//
// The centroids of a group are the time steps of ONE sub-fault: the same blended traces, read at an integer shift that
// usually moves by exactly one sample from step to step (effective_dt == dt of the database: source_bilat.f90:443-457 with
// seismogram.f90:139).  With the stride-64 lane mapping b[j] of step k+1 IS b[j-1] of step k -- the register pairs the lane
// already holds.  The apply keeps the pairs of ALL components of the previous centroid in registers (two sets that swap
// roles at every centroid) and reads only what the new shift needs:
//     shift + 1 -> the old b[j-1] set serves as b[j], only b[j-1] is read        (20 reads instead of 40)
//     shift - 1 -> the old b[j] set serves as b[j-1], only b[j] is read
//     anything else -> both are read.
// The reads are ONE asm statement per set with the pairs as read-modify-write operands: a set that is not read keeps its
// registers, and the compiler sees no merge of a loaded and a kept value (which it answers with a copy per pair, or --
// with the sets as arrays -- by promoting a whole set to one 32-register tuple that is copied and spilled as a whole).
// The compiler's wait-count bookkeeping does not see these reads: set2_wait / set2_dep make the pairs operands of the
// wait, so that the arithmetic cannot be scheduled in front of it.
// ---- (generated text) carried register sets of accumulate_grouped_kernel, 4 outputs per lane, component stride 17 x 64 dwords
struct Set2_10 { f2v &a0, &a1, &a2, &a3, &a4, &a5, &a6, &a7, &a8, &a9, &b0, &b1, &b2, &b3, &b4, &b5, &b6, &b7, &b8, &b9; };
template <int SKIP> __device__ __forceinline__ void set2_read_10(int d, unsigned a, const Set2_10 &S)
{
    asm volatile("s_cmp_eq_u32 %21, %22\n\ts_cbranch_scc1 .Lkiwi_skip%=\n\t"
                 "ds_read2st64_b32 %0, %20 offset1:1\n\t"
                 "ds_read2st64_b32 %1, %20 offset0:17 offset1:18\n\t"
                 "ds_read2st64_b32 %2, %20 offset0:34 offset1:35\n\t"
                 "ds_read2st64_b32 %3, %20 offset0:136 offset1:137\n\t"
                 "ds_read2st64_b32 %4, %20 offset0:51 offset1:52\n\t"
                 "ds_read2st64_b32 %5, %20 offset0:68 offset1:69\n\t"
                 "ds_read2st64_b32 %6, %20 offset0:85 offset1:86\n\t"
                 "ds_read2st64_b32 %7, %20 offset0:102 offset1:103\n\t"
                 "ds_read2st64_b32 %8, %20 offset0:119 offset1:120\n\t"
                 "ds_read2st64_b32 %9, %20 offset0:153 offset1:154\n\t"
                 "ds_read2st64_b32 %10, %20 offset0:2 offset1:3\n\t"
                 "ds_read2st64_b32 %11, %20 offset0:19 offset1:20\n\t"
                 "ds_read2st64_b32 %12, %20 offset0:36 offset1:37\n\t"
                 "ds_read2st64_b32 %13, %20 offset0:138 offset1:139\n\t"
                 "ds_read2st64_b32 %14, %20 offset0:53 offset1:54\n\t"
                 "ds_read2st64_b32 %15, %20 offset0:70 offset1:71\n\t"
                 "ds_read2st64_b32 %16, %20 offset0:87 offset1:88\n\t"
                 "ds_read2st64_b32 %17, %20 offset0:104 offset1:105\n\t"
                 "ds_read2st64_b32 %18, %20 offset0:121 offset1:122\n\t"
                 "ds_read2st64_b32 %19, %20 offset0:155 offset1:156\n\t"
                 "\n.Lkiwi_skip%=:"
                 : "+v"(S.a0), "+v"(S.a1), "+v"(S.a2), "+v"(S.a3), "+v"(S.a4), "+v"(S.a5), "+v"(S.a6), "+v"(S.a7), "+v"(S.a8), "+v"(S.a9), "+v"(S.b0), "+v"(S.b1), "+v"(S.b2), "+v"(S.b3), "+v"(S.b4), "+v"(S.b5), "+v"(S.b6), "+v"(S.b7), "+v"(S.b8), "+v"(S.b9) : "v"(a), "s"(d), "i"(SKIP) : "memory", "scc");
}
__device__ __forceinline__ void set2_wait_10(const Set2_10 &S)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(S.a0), "+v"(S.a1), "+v"(S.a2), "+v"(S.a3), "+v"(S.a4), "+v"(S.a5), "+v"(S.a6), "+v"(S.a7), "+v"(S.a8), "+v"(S.a9), "+v"(S.b0), "+v"(S.b1), "+v"(S.b2), "+v"(S.b3), "+v"(S.b4), "+v"(S.b5), "+v"(S.b6), "+v"(S.b7), "+v"(S.b8), "+v"(S.b9) :: "memory");
}
__device__ __forceinline__ void set2_dep_10(const Set2_10 &S)
{
    asm volatile("" : "+v"(S.a0), "+v"(S.a1), "+v"(S.a2), "+v"(S.a3), "+v"(S.a4), "+v"(S.a5), "+v"(S.a6), "+v"(S.a7), "+v"(S.a8), "+v"(S.a9), "+v"(S.b0), "+v"(S.b1), "+v"(S.b2), "+v"(S.b3), "+v"(S.b4), "+v"(S.b5), "+v"(S.b6), "+v"(S.b7), "+v"(S.b8), "+v"(S.b9) :: "memory");
}
__device__ __forceinline__ void set2_dead_10(const Set2_10 &S)
{
    asm volatile("" : "=v"(S.a0), "=v"(S.a1), "=v"(S.a2), "=v"(S.a3), "=v"(S.a4), "=v"(S.a5), "=v"(S.a6), "=v"(S.a7), "=v"(S.a8), "=v"(S.a9), "=v"(S.b0), "=v"(S.b1), "=v"(S.b2), "=v"(S.b3), "=v"(S.b4), "=v"(S.b5), "=v"(S.b6), "=v"(S.b7), "=v"(S.b8), "=v"(S.b9));
}
template <int I> __device__ __forceinline__ f2v &s2a(const Set2_10 &s) { if constexpr (I == 0) return s.a0; else if constexpr (I == 1) return s.a1; else if constexpr (I == 2) return s.a2; else if constexpr (I == 3) return s.a3; else if constexpr (I == 4) return s.a4; else if constexpr (I == 5) return s.a5; else if constexpr (I == 6) return s.a6; else if constexpr (I == 7) return s.a7; else if constexpr (I == 8) return s.a8; else return s.a9; }
template <int I> __device__ __forceinline__ f2v &s2b(const Set2_10 &s) { if constexpr (I == 0) return s.b0; else if constexpr (I == 1) return s.b1; else if constexpr (I == 2) return s.b2; else if constexpr (I == 3) return s.b3; else if constexpr (I == 4) return s.b4; else if constexpr (I == 5) return s.b5; else if constexpr (I == 6) return s.b6; else if constexpr (I == 7) return s.b7; else if constexpr (I == 8) return s.b8; else return s.b9; }
struct Set2_8 { f2v &a0, &a1, &a2, &a3, &a4, &a5, &a6, &a7, &b0, &b1, &b2, &b3, &b4, &b5, &b6, &b7; };
template <int SKIP> __device__ __forceinline__ void set2_read_8(int d, unsigned a, const Set2_8 &S)
{
    asm volatile("s_cmp_eq_u32 %17, %18\n\ts_cbranch_scc1 .Lkiwi_skip%=\n\t"
                 "ds_read2st64_b32 %0, %16 offset1:1\n\t"
                 "ds_read2st64_b32 %1, %16 offset0:17 offset1:18\n\t"
                 "ds_read2st64_b32 %2, %16 offset0:34 offset1:35\n\t"
                 "ds_read2st64_b32 %3, %16 offset0:51 offset1:52\n\t"
                 "ds_read2st64_b32 %4, %16 offset0:68 offset1:69\n\t"
                 "ds_read2st64_b32 %5, %16 offset0:85 offset1:86\n\t"
                 "ds_read2st64_b32 %6, %16 offset0:102 offset1:103\n\t"
                 "ds_read2st64_b32 %7, %16 offset0:119 offset1:120\n\t"
                 "ds_read2st64_b32 %8, %16 offset0:2 offset1:3\n\t"
                 "ds_read2st64_b32 %9, %16 offset0:19 offset1:20\n\t"
                 "ds_read2st64_b32 %10, %16 offset0:36 offset1:37\n\t"
                 "ds_read2st64_b32 %11, %16 offset0:53 offset1:54\n\t"
                 "ds_read2st64_b32 %12, %16 offset0:70 offset1:71\n\t"
                 "ds_read2st64_b32 %13, %16 offset0:87 offset1:88\n\t"
                 "ds_read2st64_b32 %14, %16 offset0:104 offset1:105\n\t"
                 "ds_read2st64_b32 %15, %16 offset0:121 offset1:122\n\t"
                 "\n.Lkiwi_skip%=:"
                 : "+v"(S.a0), "+v"(S.a1), "+v"(S.a2), "+v"(S.a3), "+v"(S.a4), "+v"(S.a5), "+v"(S.a6), "+v"(S.a7), "+v"(S.b0), "+v"(S.b1), "+v"(S.b2), "+v"(S.b3), "+v"(S.b4), "+v"(S.b5), "+v"(S.b6), "+v"(S.b7) : "v"(a), "s"(d), "i"(SKIP) : "memory", "scc");
}
__device__ __forceinline__ void set2_wait_8(const Set2_8 &S)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(S.a0), "+v"(S.a1), "+v"(S.a2), "+v"(S.a3), "+v"(S.a4), "+v"(S.a5), "+v"(S.a6), "+v"(S.a7), "+v"(S.b0), "+v"(S.b1), "+v"(S.b2), "+v"(S.b3), "+v"(S.b4), "+v"(S.b5), "+v"(S.b6), "+v"(S.b7) :: "memory");
}
__device__ __forceinline__ void set2_dep_8(const Set2_8 &S)
{
    asm volatile("" : "+v"(S.a0), "+v"(S.a1), "+v"(S.a2), "+v"(S.a3), "+v"(S.a4), "+v"(S.a5), "+v"(S.a6), "+v"(S.a7), "+v"(S.b0), "+v"(S.b1), "+v"(S.b2), "+v"(S.b3), "+v"(S.b4), "+v"(S.b5), "+v"(S.b6), "+v"(S.b7) :: "memory");
}
__device__ __forceinline__ void set2_dead_8(const Set2_8 &S)
{
    asm volatile("" : "=v"(S.a0), "=v"(S.a1), "=v"(S.a2), "=v"(S.a3), "=v"(S.a4), "=v"(S.a5), "=v"(S.a6), "=v"(S.a7), "=v"(S.b0), "=v"(S.b1), "=v"(S.b2), "=v"(S.b3), "=v"(S.b4), "=v"(S.b5), "=v"(S.b6), "=v"(S.b7));
}
template <int I> __device__ __forceinline__ f2v &s2a(const Set2_8 &s) { if constexpr (I == 0) return s.a0; else if constexpr (I == 1) return s.a1; else if constexpr (I == 2) return s.a2; else if constexpr (I == 3) return s.a3; else if constexpr (I == 4) return s.a4; else if constexpr (I == 5) return s.a5; else if constexpr (I == 6) return s.a6; else return s.a7; }
template <int I> __device__ __forceinline__ f2v &s2b(const Set2_8 &s) { if constexpr (I == 0) return s.b0; else if constexpr (I == 1) return s.b1; else if constexpr (I == 2) return s.b2; else if constexpr (I == 3) return s.b3; else if constexpr (I == 4) return s.b4; else if constexpr (I == 5) return s.b5; else if constexpr (I == 6) return s.b6; else return s.b7; }

// ---- the same for a component stride of 9 x 64 dwords (512-sample tiles: two sources per workgroup, accumulate_multi_kernel)
template <int SKIP> __device__ __forceinline__ void set2_read_10_k9(int d, unsigned a, const Set2_10 &S)
{
    asm volatile("s_cmp_eq_u32 %21, %22\n\ts_cbranch_scc1 .Lkiwi_skip%=\n\t"
                 "ds_read2st64_b32 %0, %20 offset1:1\n\t"
                 "ds_read2st64_b32 %1, %20 offset0:9 offset1:10\n\t"
                 "ds_read2st64_b32 %2, %20 offset0:18 offset1:19\n\t"
                 "ds_read2st64_b32 %3, %20 offset0:72 offset1:73\n\t"
                 "ds_read2st64_b32 %4, %20 offset0:27 offset1:28\n\t"
                 "ds_read2st64_b32 %5, %20 offset0:36 offset1:37\n\t"
                 "ds_read2st64_b32 %6, %20 offset0:45 offset1:46\n\t"
                 "ds_read2st64_b32 %7, %20 offset0:54 offset1:55\n\t"
                 "ds_read2st64_b32 %8, %20 offset0:63 offset1:64\n\t"
                 "ds_read2st64_b32 %9, %20 offset0:81 offset1:82\n\t"
                 "ds_read2st64_b32 %10, %20 offset0:2 offset1:3\n\t"
                 "ds_read2st64_b32 %11, %20 offset0:11 offset1:12\n\t"
                 "ds_read2st64_b32 %12, %20 offset0:20 offset1:21\n\t"
                 "ds_read2st64_b32 %13, %20 offset0:74 offset1:75\n\t"
                 "ds_read2st64_b32 %14, %20 offset0:29 offset1:30\n\t"
                 "ds_read2st64_b32 %15, %20 offset0:38 offset1:39\n\t"
                 "ds_read2st64_b32 %16, %20 offset0:47 offset1:48\n\t"
                 "ds_read2st64_b32 %17, %20 offset0:56 offset1:57\n\t"
                 "ds_read2st64_b32 %18, %20 offset0:65 offset1:66\n\t"
                 "ds_read2st64_b32 %19, %20 offset0:83 offset1:84\n\t"
                 "\n.Lkiwi_skip%=:"
                 : "+v"(S.a0), "+v"(S.a1), "+v"(S.a2), "+v"(S.a3), "+v"(S.a4), "+v"(S.a5), "+v"(S.a6), "+v"(S.a7), "+v"(S.a8), "+v"(S.a9), "+v"(S.b0), "+v"(S.b1), "+v"(S.b2), "+v"(S.b3), "+v"(S.b4), "+v"(S.b5), "+v"(S.b6), "+v"(S.b7), "+v"(S.b8), "+v"(S.b9) : "v"(a), "s"(d), "i"(SKIP) : "memory", "scc");
}
template <int SKIP> __device__ __forceinline__ void set2_read_8_k9(int d, unsigned a, const Set2_8 &S)
{
    asm volatile("s_cmp_eq_u32 %17, %18\n\ts_cbranch_scc1 .Lkiwi_skip%=\n\t"
                 "ds_read2st64_b32 %0, %16 offset1:1\n\t"
                 "ds_read2st64_b32 %1, %16 offset0:9 offset1:10\n\t"
                 "ds_read2st64_b32 %2, %16 offset0:18 offset1:19\n\t"
                 "ds_read2st64_b32 %3, %16 offset0:27 offset1:28\n\t"
                 "ds_read2st64_b32 %4, %16 offset0:36 offset1:37\n\t"
                 "ds_read2st64_b32 %5, %16 offset0:45 offset1:46\n\t"
                 "ds_read2st64_b32 %6, %16 offset0:54 offset1:55\n\t"
                 "ds_read2st64_b32 %7, %16 offset0:63 offset1:64\n\t"
                 "ds_read2st64_b32 %8, %16 offset0:2 offset1:3\n\t"
                 "ds_read2st64_b32 %9, %16 offset0:11 offset1:12\n\t"
                 "ds_read2st64_b32 %10, %16 offset0:20 offset1:21\n\t"
                 "ds_read2st64_b32 %11, %16 offset0:29 offset1:30\n\t"
                 "ds_read2st64_b32 %12, %16 offset0:38 offset1:39\n\t"
                 "ds_read2st64_b32 %13, %16 offset0:47 offset1:48\n\t"
                 "ds_read2st64_b32 %14, %16 offset0:56 offset1:57\n\t"
                 "ds_read2st64_b32 %15, %16 offset0:65 offset1:66\n\t"
                 "\n.Lkiwi_skip%=:"
                 : "+v"(S.a0), "+v"(S.a1), "+v"(S.a2), "+v"(S.a3), "+v"(S.a4), "+v"(S.a5), "+v"(S.a6), "+v"(S.a7), "+v"(S.b0), "+v"(S.b1), "+v"(S.b2), "+v"(S.b3), "+v"(S.b4), "+v"(S.b5), "+v"(S.b6), "+v"(S.b7) : "v"(a), "s"(d), "i"(SKIP) : "memory", "scc");
}
// ---- ... and of 5 x 64 dwords (256-sample tiles: four sources per workgroup)
template <int SKIP> __device__ __forceinline__ void set2_read_10_k5(int d, unsigned a, const Set2_10 &S)
{
    asm volatile("s_cmp_eq_u32 %21, %22\n\ts_cbranch_scc1 .Lkiwi_skip%=\n\t"
                 "ds_read2st64_b32 %0, %20 offset1:1\n\t"
                 "ds_read2st64_b32 %1, %20 offset0:5 offset1:6\n\t"
                 "ds_read2st64_b32 %2, %20 offset0:10 offset1:11\n\t"
                 "ds_read2st64_b32 %3, %20 offset0:40 offset1:41\n\t"
                 "ds_read2st64_b32 %4, %20 offset0:15 offset1:16\n\t"
                 "ds_read2st64_b32 %5, %20 offset0:20 offset1:21\n\t"
                 "ds_read2st64_b32 %6, %20 offset0:25 offset1:26\n\t"
                 "ds_read2st64_b32 %7, %20 offset0:30 offset1:31\n\t"
                 "ds_read2st64_b32 %8, %20 offset0:35 offset1:36\n\t"
                 "ds_read2st64_b32 %9, %20 offset0:45 offset1:46\n\t"
                 "ds_read2st64_b32 %10, %20 offset0:2 offset1:3\n\t"
                 "ds_read2st64_b32 %11, %20 offset0:7 offset1:8\n\t"
                 "ds_read2st64_b32 %12, %20 offset0:12 offset1:13\n\t"
                 "ds_read2st64_b32 %13, %20 offset0:42 offset1:43\n\t"
                 "ds_read2st64_b32 %14, %20 offset0:17 offset1:18\n\t"
                 "ds_read2st64_b32 %15, %20 offset0:22 offset1:23\n\t"
                 "ds_read2st64_b32 %16, %20 offset0:27 offset1:28\n\t"
                 "ds_read2st64_b32 %17, %20 offset0:32 offset1:33\n\t"
                 "ds_read2st64_b32 %18, %20 offset0:37 offset1:38\n\t"
                 "ds_read2st64_b32 %19, %20 offset0:47 offset1:48\n\t"
                 "\n.Lkiwi_skip%=:"
                 : "+v"(S.a0), "+v"(S.a1), "+v"(S.a2), "+v"(S.a3), "+v"(S.a4), "+v"(S.a5), "+v"(S.a6), "+v"(S.a7), "+v"(S.a8), "+v"(S.a9), "+v"(S.b0), "+v"(S.b1), "+v"(S.b2), "+v"(S.b3), "+v"(S.b4), "+v"(S.b5), "+v"(S.b6), "+v"(S.b7), "+v"(S.b8), "+v"(S.b9) : "v"(a), "s"(d), "i"(SKIP) : "memory", "scc");
}
template <int SKIP> __device__ __forceinline__ void set2_read_8_k5(int d, unsigned a, const Set2_8 &S)
{
    asm volatile("s_cmp_eq_u32 %17, %18\n\ts_cbranch_scc1 .Lkiwi_skip%=\n\t"
                 "ds_read2st64_b32 %0, %16 offset1:1\n\t"
                 "ds_read2st64_b32 %1, %16 offset0:5 offset1:6\n\t"
                 "ds_read2st64_b32 %2, %16 offset0:10 offset1:11\n\t"
                 "ds_read2st64_b32 %3, %16 offset0:15 offset1:16\n\t"
                 "ds_read2st64_b32 %4, %16 offset0:20 offset1:21\n\t"
                 "ds_read2st64_b32 %5, %16 offset0:25 offset1:26\n\t"
                 "ds_read2st64_b32 %6, %16 offset0:30 offset1:31\n\t"
                 "ds_read2st64_b32 %7, %16 offset0:35 offset1:36\n\t"
                 "ds_read2st64_b32 %8, %16 offset0:2 offset1:3\n\t"
                 "ds_read2st64_b32 %9, %16 offset0:7 offset1:8\n\t"
                 "ds_read2st64_b32 %10, %16 offset0:12 offset1:13\n\t"
                 "ds_read2st64_b32 %11, %16 offset0:17 offset1:18\n\t"
                 "ds_read2st64_b32 %12, %16 offset0:22 offset1:23\n\t"
                 "ds_read2st64_b32 %13, %16 offset0:27 offset1:28\n\t"
                 "ds_read2st64_b32 %14, %16 offset0:32 offset1:33\n\t"
                 "ds_read2st64_b32 %15, %16 offset0:37 offset1:38\n\t"
                 "\n.Lkiwi_skip%=:"
                 : "+v"(S.a0), "+v"(S.a1), "+v"(S.a2), "+v"(S.a3), "+v"(S.a4), "+v"(S.a5), "+v"(S.a6), "+v"(S.a7), "+v"(S.b0), "+v"(S.b1), "+v"(S.b2), "+v"(S.b3), "+v"(S.b4), "+v"(S.b5), "+v"(S.b6), "+v"(S.b7) : "v"(a), "s"(d), "i"(SKIP) : "memory", "scc");
}
template <int NG> struct Set2Sel;
template <> struct Set2Sel<10> { typedef Set2_10 type; };
template <> struct Set2Sel<8> { typedef Set2_8 type; };
// (the set is NOT read when d == SKIP: the comparison happens inside the asm statement on the scalar d -- a flag computed
// outside reaches an "s" operand through a vector register)
template <int K, int SKIP> __device__ __forceinline__ void set2_read(int d, unsigned a, const Set2_10 &S)
{
    if constexpr (K == 17) set2_read_10<SKIP>(d, a, S); else if constexpr (K == 9) set2_read_10_k9<SKIP>(d, a, S); else set2_read_10_k5<SKIP>(d, a, S);
}
template <int K, int SKIP> __device__ __forceinline__ void set2_read(int d, unsigned a, const Set2_8 &S)
{
    if constexpr (K == 17) set2_read_8<SKIP>(d, a, S); else if constexpr (K == 9) set2_read_8_k9<SKIP>(d, a, S); else set2_read_8_k5<SKIP>(d, a, S);
}
__device__ __forceinline__ void set2_wait(const Set2_10 &S) { set2_wait_10(S); }
__device__ __forceinline__ void set2_wait(const Set2_8 &S) { set2_wait_8(S); }
__device__ __forceinline__ void set2_dep(const Set2_10 &S) { set2_dep_10(S); }
__device__ __forceinline__ void set2_dep(const Set2_8 &S) { set2_dep_8(S); }
__device__ __forceinline__ void set2_dead(const Set2_10 &S) { set2_dead_10(S); }
__device__ __forceinline__ void set2_dead(const Set2_8 &S) { set2_dead_8(S); }

template <int N, int I = 0, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<N, I + 1>(f); }
}

// All GF components of one centroid from the register sets L (b[j-1]) and H (b[j]), reference order (centroid_apply_hd's
// operations); a: LDS byte address of b[j-1] of the lane's first output in component 0; coef: the centroid's coefficient line
// (wave-uniform pointer: scalar loads, SGPR operands of the packed multiplies).
template <int NG, bool ROT, int K = 17>
__device__ __forceinline__ void carry2_apply(f2v (&ar1)[2], f2v (&ar2)[2], f2v (&dz)[2], const typename Set2Sel<NG>::type &L,
                                             const typename Set2Sel<NG>::type &H, unsigned a,
                                             int dh, int dl /* previous shift position minus this one (none: 0x7fff): 1 keeps H, -1 keeps L */,
                                             const float *__restrict__ coef, float cl, float sl)
{
    constexpr int nH1 = (NG == 10) ? 4 : 3;      // components summed into the radial trace
    float cw[2 * NG];
#pragma unroll
    for (int i = 0; i < 2 * NG; i++) cw[i] = coef[i];
    static_assert(K == 17 || K == 9 || K == 5, "component stride in units of 64 dwords");
    set2_read<K, 1>(dh, a + 4, H);            // shift + 1 (d == 1): the b[j] set is in place
    set2_read<K, -1>(dl, a, L);               // shift - 1: the b[j-1] set is
    set2_wait(H);
    set2_dep(L);
    f2v t1[2], t2[2];
#pragma unroll
    for (int k = 0; k < 2; k++) { t1[k] = ROT ? f2v{ 0.f, 0.f } : ar1[k]; t2[k] = ROT ? f2v{ 0.f, 0.f } : ar2[k]; }
    static_for<NG>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        const f2v c1 = { cw[2 * i], cw[2 * i] }, c2 = { cw[2 * i + 1], cw[2 * i + 1] };
        if constexpr (i < nH1) {
            t1[0] = t1[0] + c1 * s2a<i>(H); t1[0] = t1[0] + c2 * s2a<i>(L);
            t1[1] = t1[1] + c1 * s2b<i>(H); t1[1] = t1[1] + c2 * s2b<i>(L);
        } else if constexpr (i < nH1 + 2) {
            t2[0] = t2[0] + c1 * s2a<i>(H); t2[0] = t2[0] + c2 * s2a<i>(L);
            t2[1] = t2[1] + c1 * s2b<i>(H); t2[1] = t2[1] + c2 * s2b<i>(L);
        } else {
            dz[0] = dz[0] + c1 * s2a<i>(H); dz[0] = dz[0] + c2 * s2a<i>(L);
            dz[1] = dz[1] + c1 * s2b<i>(H); dz[1] = dz[1] + c2 * s2b<i>(L);
        }
        if constexpr (i == nH1 + 1) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (ROT) {
                    ar1[k] = ar1[k] + cl * t1[k] - sl * t2[k];
                    ar2[k] = ar2[k] + cl * t2[k] + sl * t1[k];
                } else {
                    ar1[k] = t1[k]; ar2[k] = t2[k];
                }
            }
        }
    });
}

template <int NG>
__device__ __forceinline__ typename Set2Sel<NG>::type make_set2(f2v &a0, f2v &a1, f2v &a2, f2v &a3, f2v &a4, f2v &a5, f2v &a6, f2v &a7, f2v &a8, f2v &a9,
                                                                f2v &b0, f2v &b1, f2v &b2, f2v &b3, f2v &b4, f2v &b5, f2v &b6, f2v &b7, f2v &b8, f2v &b9)
{
    if constexpr (NG == 10) return Set2_10{ a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, b0, b1, b2, b3, b4, b5, b6, b7, b8, b9 };
    else return Set2_8{ a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7 };
}

template <int NG, int T, bool FUSE, bool RUNS>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(KIWI_GROUPED_WAVES))) void accumulate_grouped_kernel(
    const float *__restrict__ G, const int2 *__restrict__ span, int pitch,
    const GeoRec *__restrict__ recs, const int *__restrict__ cent_ofs, int isrc0, int nrec,
    const RecvDev *__restrict__ recv, float *__restrict__ syn, size_t syn_stride, int ntiles,
    const int *__restrict__ tab, const int *__restrict__ run_first, FuseParams fp,
    const int *__restrict__ pairflag /* see geometry_kernel */,
    int pairsel /* 0 all pairs, 1 not the cell kernel's, 3 not the duo kernel's */,
    const int *__restrict__ mate, const int *__restrict__ mate4 /* pairsel 3: groups of two / four sources accumulate_multi_kernel takes */,
    const int *__restrict__ synrow /* optional: sources that share another source's synthetics are not synthesised */,
    const int *__restrict__ fam_ofs, const int *__restrict__ fam_list /* FUSE with synrow: the sources that share source s's
                                          synthetics, fam_list[fam_ofs[s] .. fam_ofs[s + 1]): compared here with their moments */)
{
    // run_first != nullptr: blockIdx.x indexes RUNS of consecutive trial sources [run_first[b], run_first[b+1]) that the
    // host found to have identical centroid geometry (same points and times: only the moment tensors differ, e.g. a
    // strike/dip/rake grid at a fixed location) and to consist of a single centroid group each.  The blended tiles
    // are then built ONCE and every source of the run is applied from them -- the sharing across trial sources
    // SURVEY.md 8d asks to report separately.  Per-source operations and their order are unchanged.
    constexpr int TILE = 4 * T;                          // samples per workgroup, 4 consecutive per thread
    constexpr int LDS_TILE = TILE + kHalo;
    __shared__ __attribute__((aligned(16))) float tiles[NG][LDS_TILE];
    // SOURCE index fastest in dispatch order: the workgroups resident at any moment are the same
    // (tile, receiver) of many neighbouring trial sources, which read (nearly) the same GF rows at
    // the same time, and blocks b, b+8, ... share an XCD and therefore its L2 (dispatch is
    // round-robin over the 8 XCDs; speed only, never correctness).  Measured alternative: giving each XCD a
    // contiguous eighth of this order (all sources of one (tile, receiver) on ONE XCD, eight different
    // (tile, receiver) sets in flight) is 4 % slower at cfg3 and 6 % at cfg4 -- eight L2s holding the same few rows
    // cost nothing, eight times as many distinct rows in flight load the Infinity Fabric.
    const int s = RUNS ? run_first[blockIdx.x] : (int)blockIdx.x;               // first (or only) source of this workgroup
    const int s_end = RUNS ? run_first[blockIdx.x + 1] : s + 1;
    const bool multi = RUNS && s_end - s > 1;
    const int tile = blockIdx.y % ntiles, r = blockIdx.y / ntiles;
    const RecvDev &rv = recv[r];
    if (!rv.enabled) return;
    if (tile * TILE >= rv.wlen) return;
    if (pairsel == 1 && cell_pair(rv, pairflag, s, nrec, r)) return;
    if (pairsel == 3 && (multi_taken_fwd<4>(rv, pairflag, mate4, s, nrec, r) || multi_taken_fwd<2>(rv, pairflag, mate, s, nrec, r))) return;
    if (synrow && !multi && synrow[s] != s) return;  // (in a run the sources that share synthetics are left out one by one)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int t_tile0 = rv.wbeg + tile * TILE;
    const int cb = cent_ofs[isrc0], c0 = cent_ofs[isrc0 + s], nc = cent_ofs[isrc0 + s + 1] - c0;
    const GeoRec *__restrict__ rc = recs + ((size_t)(c0 - cb) * nrec + (size_t)r * nc);
    const int *__restrict__ tc = tab + ((size_t)(c0 - cb) * nrec + (size_t)r * nc) * 128;
    const bool need_h = rv.need_h != 0, has_d = rv.has_d != 0;
    const float sd = rv.sd;

    f2v ar1[2] = { { 0.f, 0.f }, { 0.f, 0.f } }, ar2[2] = { { 0.f, 0.f }, { 0.f, 0.f } }, dz[2] = { { 0.f, 0.f }, { 0.f, 0.f } };
    // rotation to N/E, signs and store of one source's accumulators (seismogram.f90:256-283)
    auto store = [&](int js) {
        const int tl = tile * TILE + 4 * (tid & ~63) + lane;     // window sample of the lane's output q = 0; q-th: + 64 q
        if (!FUSE && tl >= rv.wlen) return;
        float *__restrict__ so = syn + (size_t)js * syn_stride + tl;
        float mom = 0.f;
        if constexpr (FUSE) mom = fp.moment[fp.isrc0 + js];
        const bool whole = tile * TILE + TILE <= rv.wlen;
        for (int k = 0; k < rv.ncomp; k++) {
            f2v o01, o23;
            out4(rv.comp[k], rv.sign[k], rv.cl0, rv.sl0, ar1, ar2, dz, o01, o23);
            if constexpr (!FUSE) {
                const float o[4] = { o01.x, o01.y, o23.x, o23.y };
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (tl + 64 * i < rv.wlen) so[rv.synofs[k] + 64 * i] = o[i];      // 256 contiguous bytes per wave and store
                continue;
            }
            // ---- fused comparator, then a wave reduction; the partial of (source, slot, tile, wave) is summed by
            // misfit_finish_kernel in a fixed order
            double acc = fused_acc4(fp, o01, o23, mom, fp.reft + rv.refofs[k] + tl, fp.tw + rv.refofs[k] + tl, tl, rv.wlen, whole);
            acc = wave_reduce_f64(acc, fp.method == 6);                 // total in lane 63
            if (lane == 63)
                fp.partial[((size_t)js * fp.nmis + rv.slot0 + k) * fp.nparts + tile * (T / 64) + (tid >> 6)] = acc;
        }
    };
    // a source and, with the fused comparator, the sources that share its synthetics (other moments; minimizer_engine.f90:516-521)
    auto store_family = [&](int js) {
        __builtin_amdgcn_s_setprio(1);                   // (the epilogue's loads of reference and taper first: cfg2 12.62 -> 12.48 ms)
        store(js);
        __builtin_amdgcn_s_setprio(0);
        if constexpr (FUSE) {
            if (fam_ofs)
                for (int q = fam_ofs[js]; q < fam_ofs[js + 1]; q++) store(fam_list[q]);
        }
    };
    bool stored = false;
    int c = 0;
    int cur = rec_load(rc, 0, nc, lane);                 // record c, lane-distributed
    int ta = 0, tb = 0;                                  // load descriptors of record c (a rejected trial source has no centroids, no rows)
    if (nc > 0) { ta = tc[lane]; tb = tc[64 + lane]; }
    // carried register sets (see set2_read): the common case of 256-thread workgroups
    constexpr bool kCarry = (T == 256);
    f2v xa0, xa1, xa2, xa3, xa4, xa5, xa6, xa7, xa8, xa9, xb0, xb1, xb2, xb3, xb4, xb5, xb6, xb7, xb8, xb9;
    f2v ya0, ya1, ya2, ya3, ya4, ya5, ya6, ya7, ya8, ya9, yb0, yb1, yb2, yb3, yb4, yb5, yb6, yb7, yb8, yb9;
    typedef typename Set2Sel<NG>::type SetT;
    const SetT X = make_set2<NG>(xa0, xa1, xa2, xa3, xa4, xa5, xa6, xa7, xa8, xa9, xb0, xb1, xb2, xb3, xb4, xb5, xb6, xb7, xb8, xb9);
    const SetT Y = make_set2<NG>(ya0, ya1, ya2, ya3, ya4, ya5, ya6, ya7, ya8, ya9, yb0, yb1, yb2, yb3, yb4, yb5, yb6, yb7, yb8, yb9);
    while (c < nc) {
        GeoRec g0;
        rec_head(cur, 0, g0);
        if (g0.row[0] < 0) {                             // 'cycle' on a missing trace
            c++;
            cur = rec_load(rc, c, nc, lane);
            if (c < nc) { ta = tc[(size_t)c * 128 + lane]; tb = tc[(size_t)c * 128 + 64 + lane]; }
            continue;
        }
        // ---- the group starting here (hint computed by geometry_kernel)
        const int cend = c + (g0.pad & 0xff);
        const int smax = g0.ishift + ((g0.pad >> 8) & 0xff), smin = g0.ishift - ((g0.pad >> 16) & 0xff);
        // LDS position p holds blended trace sample jb + p
        const int jb = t_tile0 - smax - 1;
        const int npos = TILE + (smax - smin) + 8;       // positions read by the group (<= LDS_TILE)
        const bool direct = (g0.flags & 1) != 0;
        // end indices of the blended traces (tail rule); their minima over the horizontal / vertical components
        // come precomputed from geometry_kernel so that no per-component branching is needed here
        int jend[NG];
#pragma unroll
        for (int ig = 0; ig < NG; ig++) jend[ig] = REC_I(ta, 40 + ig);
        const int jend_h = REC_I(ta, 50), jend_d = REC_I(ta, 51);
        const int jend_min = min(need_h ? jend_h : 0x7fffffff, has_d ? jend_d : 0x7fffffff);
        // carried register sets: receivers with horizontal and vertical components, every component added, no tail rule
        // anywhere in the tile for any shift of the group; the group's cos / sin of the back-azimuth change and its flags are
        // those of its head record (same point, same receiver), its integer shifts come in one load (lane k = centroid c + k)
        bool carry_grp = false;
        int ishv = 0, cur_next = 0;
        float gcl = 0.f, gsl = 0.f;
        if constexpr (kCarry) {
            carry_grp = need_h && has_d && !(g0.flags & 8) && !((jb + (smax - smin) + TILE) > jend_min);
            if (carry_grp) {
                if (lane < (g0.pad & 0xff)) ishv = rc[c + lane].ishift;
                cur_next = rec_load(rc, cend, nc, lane);
                gcl = REC_F(cur, 16); gsl = REC_F(cur, 17);
            }
        }
        // the build phase first (see accumulate_multi_kernel: cfg3-scatter 38.9 -> 37.6 ms); not in runs of sources, where one
        // build serves many applies (cfg2: 12.46 -> 12.60 with it)
        if constexpr (!RUNS) __builtin_amdgcn_s_setprio(1);
        {
            float *tile0 = &tiles[0][0];
            // descriptors are relative to the first row of the group's cell (64-bit base, see write_tab)
            const float *__restrict__ Gg = G + (size_t)g0.row[0] * (size_t)pitch;
            // all 40 rows of the group cover [jb, jb + LDS_TILE) inside their padded storage?  (lane l < 40 holds
            // row l's descriptors; workgroup-uniform because every wave holds the same table)
            const bool lane_ok = lane >= 4 * NG || (ta + jb >= tb && ta + jb + LDS_TILE <= tb + pitch);
            const bool fast = __builtin_amdgcn_ballot_w64(lane_ok) == ~0ull;
#define BUILD_B(IGS, P) do { \
                if (fast) { if (direct) build_batch<false, true>(tile0, LDS_TILE, IGS, P, jb, Gg, pitch, ta, tb, g0); \
                            else        build_batch<true, true>(tile0, LDS_TILE, IGS, P, jb, Gg, pitch, ta, tb, g0); } \
                else      { if (direct) build_batch<false, false>(tile0, LDS_TILE, IGS, P, jb, Gg, pitch, ta, tb, g0); \
                            else        build_batch<true, false>(tile0, LDS_TILE, IGS, P, jb, Gg, pitch, ta, tb, g0); } \
            } while (0)
            constexpr int H1 = NG / 2;
            int igA[H1], igB[NG - H1];
#pragma unroll
            for (int q = 0; q < H1; q++) igA[q] = q;
#pragma unroll
            for (int q = 0; q < NG - H1; q++) igB[q] = H1 + q;
            const int igH10[6] = { 0, 1, 2, 3, 4, 8 }, igD10[4] = { 5, 6, 7, 9 };      // horizontals / vertical only
            const int igH8[5] = { 0, 1, 2, 3, 4 }, igD8[3] = { 5, 6, 7 };
            // halo: pair q = (component slot, chunk) -> lanes; kHaloIter passes cover 16 * (number of components) pairs
            constexpr int kHaloIter = (16 * NG + T - 1) / T;
            const int ncmp = (need_h && has_d) ? NG : (need_h ? (NG == 10 ? 6 : 5) : (NG == 10 ? 4 : 3));
            bool hact[kHaloIter];
            int hig[kHaloIter], hph[kHaloIter];
            HaloRegs hr[kHaloIter];
#pragma unroll
            for (int it = 0; it < kHaloIter; it++) {
                const int q = tid + it * T, slot = q >> 4;
                int ig = slot;                                            // all components: slot = component
                if (!(need_h && has_d)) {
                    if (need_h) ig = (slot == 5) ? 8 : slot;              // 0 1 2 3 4 8
                    else        ig = (NG == 10 && slot == 3) ? 9 : 5 + slot;   // 5 6 7 9
                }
                hig[it] = min(ig, NG - 1);
                hph[it] = TILE + 4 * (q & 15);
                hact[it] = slot < ncmp && hph[it] < npos;
            }
#define HALO_ISSUE() do { _Pragma("unroll") for (int it = 0; it < kHaloIter; it++) { \
                if (fast) { if (direct) hr[it] = halo_issue<false, true>(hact[it], hig[it], hph[it], jb, Gg, pitch, ta, tb); \
                            else        hr[it] = halo_issue<true, true>(hact[it], hig[it], hph[it], jb, Gg, pitch, ta, tb); } \
                else      { if (direct) hr[it] = halo_issue<false, false>(hact[it], hig[it], hph[it], jb, Gg, pitch, ta, tb); \
                            else        hr[it] = halo_issue<true, false>(hact[it], hig[it], hph[it], jb, Gg, pitch, ta, tb); } } } while (0)
#define HALO_FINISH() do { _Pragma("unroll") for (int it = 0; it < kHaloIter; it++) { \
                if (direct) halo_finish<false>(hact[it], hr[it], tile0, LDS_TILE, hig[it], hph[it], g0); \
                else        halo_finish<true>(hact[it], hr[it], tile0, LDS_TILE, hig[it], hph[it], g0); } } while (0)
#ifdef KIWI_X_NOBUILD
            if (false) {
#else
            if (need_h && has_d) {
#endif
                // streamed: NG / 2 sub-batches of two components, two in flight (see pair_issue)
                constexpr int NP = NG / 2;
#define STREAM(BL, FA) do { \
                    f4u va[2][BL ? 4 : 1], vb[2][BL ? 4 : 1]; \
                    pair_issue<BL, FA>(va, 0, 4 * tid, jb, Gg, pitch, ta, tb); \
                    pair_issue<BL, FA>(vb, 2, 4 * tid, jb, Gg, pitch, ta, tb); \
                    __syncthreads();        /* the tiles are free: every wave has applied the previous group (see multi_build) */ \
                    _Pragma("unroll") for (int i = 0; i < NP; i++) { \
                        __builtin_amdgcn_sched_barrier(0); \
                        if ((i & 1) == 0) { pair_finish<BL>(va, tile0, LDS_TILE, 2 * i, 4 * tid, g0); \
                                            if (i + 2 < NP) pair_issue<BL, FA>(va, 2 * (i + 2), 4 * tid, jb, Gg, pitch, ta, tb); } \
                        else              { pair_finish<BL>(vb, tile0, LDS_TILE, 2 * i, 4 * tid, g0); \
                                            if (i + 2 < NP) pair_issue<BL, FA>(vb, 2 * (i + 2), 4 * tid, jb, Gg, pitch, ta, tb); } \
                        if (i == NP - 3) HALO_ISSUE(); \
                    } } while (0)
                if (fast) { if (direct) STREAM(false, true); else STREAM(true, true); }
                else      { if (direct) STREAM(false, false); else STREAM(true, false); }
#undef STREAM
                HALO_FINISH();
#ifdef KIWI_X_NOBUILD
            } else if (false) {
#else
            } else if (need_h) {
#endif
                HALO_ISSUE();
                __syncthreads();
                if constexpr (NG == 10) BUILD_B(igH10, 4 * tid); else BUILD_B(igH8, 4 * tid);
                HALO_FINISH();
#ifdef KIWI_X_NOBUILD
            } else if (false) {
#else
            } else {
#endif
                HALO_ISSUE();
                __syncthreads();
                if constexpr (NG == 10) BUILD_B(igD10, 4 * tid); else BUILD_B(igD8, 4 * tid);
                HALO_FINISH();
            }
#undef HALO_ISSUE
#undef HALO_FINISH
#undef BUILD_B
        }
        // descriptors of the NEXT group: in flight while this group is applied
        if (cend < nc) { ta = tc[(size_t)cend * 128 + lane]; tb = tc[(size_t)cend * 128 + 64 + lane]; }
        __syncthreads();
        if constexpr (!RUNS) __builtin_amdgcn_s_setprio(0);
#ifdef KIWI_X_NOAPPLY
        if (kCarry && carry_grp) { cur = cur_next; } else
#endif
        if (kCarry && carry_grp) {
            // ---- apply with carried register sets: the two sets swap roles at EVERY centroid (pairs of centroids, roles
            // static): after a centroid L holds its b[j-1] and H its b[j]; with the roles swapped, shift + 1 finds b[j] in place
            // and reads b[j-1], shift - 1 finds b[j-1] in place and reads b[j]; every other step reads both.  The rotating /
            // plain branch (seismogram.f90:160-203 / :205-231) is the same for the whole group: two copies of the loop.
            // In a run (RUNS: geometry-identical sources sharing these tiles) source after source, the carried sets passing from
            // the last centroid of one to the first of the next (shift - k + 1 steps back: for two time steps per source the set
            // that is in place again); accumulators and coefficient rows per source.
            const unsigned abase = (unsigned)(size_t)(lds_cfp)&tiles[0][4 * (tid & ~63) + lane];
            int eprev = 0;
            bool have = false;
#define KIWI_C2STEP(LL, HH, RV, CC) do { \
                const int e = smax - __builtin_amdgcn_readlane(ishv, (CC) - c);      /* LDS position of b[j-1] of the tile's first sample */ \
                const int d = have ? eprev - e : 0x7fff; \
                /* `same`: the sets are in the roles of the previous step (first centroid of a source behind a source with an odd \
                   number of centroids): both are in place when the shift has not moved, none otherwise */ \
                const int dh = same ? (d == 0 ? 1 : 0x7fff) : d, dl = same ? (d == 0 ? -1 : 0x7fff) : d; \
                same = false; \
                carry2_apply<NG, RV>(ar1, ar2, dz, LL, HH, abase + 4u * (unsigned)e, KIWI_X_FULL(dh), KIWI_X_FULL(dl), coef_grp + (size_t)((CC) - c) * 128, gcl, gsl); \
                have = true; eprev = e; } while (0)
#define KIWI_C2SOURCES(RV) do { \
                set2_dead(X); set2_dead(Y);       /* nothing is carried into a group, its first centroid reads both sets: tells the register allocator so */ \
                for (int js = s; js < s_end; js++) { \
                    if (multi) { \
                        ar1[0] = ar1[1] = ar2[0] = ar2[1] = dz[0] = dz[1] = f2v{ 0.f, 0.f }; \
                        if (synrow && synrow[js] != js) continue;       /* evaluated with the source it shares synthetics with */ \
                    } \
                    const size_t crow = ((size_t)(cent_ofs[isrc0 + js] - cb) * nrec + (size_t)r * nc + c) * 128 + 64 + 40; \
                    const unsigned clo = __builtin_amdgcn_readfirstlane((unsigned)crow), chi = __builtin_amdgcn_readfirstlane((unsigned)(crow >> 32)); \
                    const float *__restrict__ coef_grp = (const float *)(tab + (((size_t)chi << 32) | clo)); \
                    int cc = c; \
                    /* odd count in a run: the sets end in the roles they started in, and the next source's first centroid finds \
                       them so (point sources of ONE centroid: no LDS read at all behind the first source of the run) */ \
                    bool same = RUNS && multi && have && ((cend - c) & 1); \
                    for (; cc + 1 < cend; cc += 2) { KIWI_C2STEP(X, Y, RV, cc); KIWI_C2STEP(Y, X, RV, cc + 1); } \
                    if (cc < cend) KIWI_C2STEP(X, Y, RV, cc); \
                    if (multi) store_family(js); \
                } } while (0)
            if (g0.flags & 2) KIWI_C2SOURCES(true); else KIWI_C2SOURCES(false);
#undef KIWI_C2SOURCES
#undef KIWI_C2STEP
            cur = cur_next;
        } else
        // ---- apply: every centroid of the group, in table order (seismogram.f90:131); in a run, source after source
        for (int js = s; js < s_end; js++) {
        const GeoRec *__restrict__ rcj = multi ? recs + ((size_t)(cent_ofs[isrc0 + js] - cb) * nrec + (size_t)r * nc) : rc;
        // first record of the NEXT source of the run: in flight while this one is applied
        int cur_next_src = 0;
        if (multi) {
            if (js + 1 < s_end) cur_next_src = rec_load(recs + ((size_t)(cent_ofs[isrc0 + js + 1] - cb) * nrec + (size_t)r * nc), c, nc, lane);
            ar1[0] = ar1[1] = ar2[0] = ar2[1] = dz[0] = dz[1] = f2v{ 0.f, 0.f };
            if (synrow && synrow[js] != js) { cur = cur_next_src; continue; }      // evaluated with the source it shares synthetics with
        }
        // coefficient rows of the group's centroids: wave-uniform by construction; readfirstlane tells the compiler so
        // (-> scalar address arithmetic and scalar loads of the coefficients)
        const size_t crow = ((size_t)(cent_ofs[isrc0 + js] - cb) * nrec + (size_t)r * nc + c) * 128 + 64 + 40;
        const unsigned clo = __builtin_amdgcn_readfirstlane((unsigned)crow), chi = __builtin_amdgcn_readfirstlane((unsigned)(crow >> 32));
        const float *__restrict__ coef_grp = (const float *)(tab + (((size_t)chi << 32) | clo));
        for (int cc = c; cc < cend; cc++) {
            const int nxt = rec_load(rcj, cc + 1, nc, lane);     // prefetch the next record
            constexpr int ro = 0;
            const int ishift = REC_I(cur, ro + 8);
            // (the interpolation fraction and the six weights are read from the record only by the variants that use them:
            // with scalar-loaded coefficients and no tail rule they are not needed at all)
            const float cl = REC_F(cur, ro + 16), sl = REC_F(cur, ro + 17);
            const int flags = REC_I(cur, ro + 18);
            const float *__restrict__ coef = coef_grp + (size_t)(cc - c) * 128;
            const int e = smax - ishift;                 // LDS position of b[j-1] of the tile's first sample
            const int u0 = 4 * (tid & ~63) + lane;       // this lane's first tile sample (the others: + 64 q)
            const TileBase chunk0 = tile_base(&tiles[0][e + u0]);
            const int jl = jb + e + u0;                  // trace index of b[j-1] of the lane's output q = 0
            const bool tail = (jb + e + TILE) > jend_min;        // workgroup-uniform
#define APPLY(TV) do { \
                if (need_h && has_d && !(flags & 8)) centroid_apply_hd<NG, LDS_TILE, TV, !(FUSE && RUNS)>(ar1, ar2, dz, chunk0, jl, jend, flags, coef, cur, sd, cl, sl); \
                else centroid_apply<NG, LDS_TILE, TV>(ar1, ar2, dz, chunk0, jl, jend, need_h, has_d, flags, \
                                                      REC_F(cur, 9), sd, REC_F(cur, 10), REC_F(cur, 11), REC_F(cur, 12), REC_F(cur, 13), \
                                                      REC_F(cur, 14), REC_F(cur, 15), cl, sl); } while (0)
            if (!tail) APPLY(false); else APPLY(true);
#undef APPLY
            cur = nxt;
        }
        if (multi) { store_family(js); cur = cur_next_src; }
        }
        stored = multi;
#ifdef KIWI_X_NOBUILD
        __syncthreads();
#endif
        c = cend;                                        // (the barrier in front of the next group's LDS writes stands in its build)
    }
    if (!multi) store_family(s);
    else if (!stored) {                                  // every centroid skipped: the run's synthetics are zero
        ar1[0] = ar1[1] = ar2[0] = ar2[1] = dz[0] = dz[1] = f2v{ 0.f, 0.f };
        for (int js = s; js < s_end; js++)
            if (!(synrow && synrow[js] != js)) store_family(js);
    }
}

template <bool BLEND, bool FAST>
__device__ __forceinline__ void one_issue(f4u (&v)[BLEND ? 4 : 1], int ig, int p, int jb, const float *__restrict__ G,
                                          int pitch, int ta, int tb)
{
    const int j = jb + p;
#pragma unroll
    for (int k = 0; k < (BLEND ? 4 : 1); k++) {
        const int base = REC_I(ta, 4 * ig + k);
        if constexpr (FAST) {
            v[k] = buf_load4(gf_rsrc(G), p, base + jb);
        } else {
            const int lo = REC_I(tb, 4 * ig + k);
            const int idx = min(max(base + j, lo), lo + pitch - 4);
            v[k] = *(const f4u *)(G + (size_t)(unsigned)idx);
        }
    }
}

template <bool BLEND>
__device__ __forceinline__ void one_finish(const f4u (&v)[BLEND ? 4 : 1], float *__restrict__ tile0, int lds_tile, int ig, int p,
                                           const GeoRec &g)
{
    f4u b;
    if constexpr (BLEND) {
        b = g.w[0] * v[0];                    // gfdb.f90:946-949, summed in this order
        b = b + g.w[1] * v[1];
        b = b + g.w[2] * v[2];
        b = b + g.w[3] * v[3];
    } else {
        b = v[0];
    }
    *(float4 *)(tile0 + ig * lds_tile + p) = make_float4(b.x, b.y, b.z, b.w);
}

// Build of accumulate_multi_kernel: NS tile sets of TILE = 1024 / NS samples (+ halo).  A wave-task = one component's 64 main
// chunks (chunk = 4 samples) of one 256-sample slab; NS = 2: wave w takes components 2 i + (w >> 1), slab w & 1; NS = 4: wave w
// components w + 4 i (one slab); the halo chunks go to the LAST threads, chunk-major (thread 255 - q: chunk q / NG of component
// q % NG): a group reads only the first few chunks behind the tile (its shift range + 8 samples), so the halo is the work of the
// last wave alone -- the one with the fewest main tasks (NS = 4: 3, 3, 2, 2) --, `whalo` says whether this wave has any.  Every chunk's four node rows are loaded ONCE and
// blended with the weights of each source into its tile set (gfdb.f90:946-949, summed in this order) -- `only` >= 0: into that
// source's set alone.  A centroid exactly on a node carries the weights (1, 0, 0, 0) over four copies of its row:
// 1 v + 0 v + 0 v + 0 v is v bit for bit, so there is no unblended variant.
template <int NG, bool FAST, int NS>
__device__ __forceinline__ void multi_build(float *__restrict__ tile0, int wv, int lane, int tid, int jb, const float *__restrict__ G,
                                            int pitch, int ta, int tb, const GeoRec (&gw)[NS], int only, bool hact, bool whalo, int hig, int hph)
{
    constexpr int TILE = 1024 / NS, LDS_TILE = TILE + kHalo, DEPTH = 3;
    constexpr int N = (NS == 2) ? NG / 2 : (NG + 3) / 4;           // wave-tasks per wave at most
    const int p = (NS == 2) ? 4 * (64 * (wv & 1) + lane) : 4 * lane;
    auto comp = [&](int i) { return (NS == 2) ? 2 * i + (wv >> 1) : wv + 4 * i; };
    f4u v[N][4];
    HaloRegs hv;
    // (a wave without a halo lane skips the halo; in the others inactive lanes load a valid chunk too: no merge of registers)
    if (whalo) hv = halo_issue<true, FAST>(true, hig, hph, jb, G, pitch, ta, tb);
#pragma unroll
    for (int i = 0; i < DEPTH && i < N; i++) if (comp(i) < NG) one_issue<true, FAST>(v[i], comp(i), p, jb, G, pitch, ta, tb);
    // The barrier that frees the tile sets (every wave has applied the previous group) stands HERE, behind the first loads of
    // this group and in front of its first LDS write: a wave that is done applying has its loads in flight while it waits
    // for the others.  (Workgroup-uniform call: every wave takes the same variant of this function.)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; i++) {
        __builtin_amdgcn_sched_barrier(0);
        if (comp(i) < NG) {
#pragma unroll
            for (int s = 0; s < NS; s++)
                if (only < 0 || only == s) one_finish<true>(v[i], tile0 + s * NG * LDS_TILE, LDS_TILE, comp(i), p, gw[s]);
        }
        if (i + DEPTH < N && comp(i + DEPTH) < NG) one_issue<true, FAST>(v[i + DEPTH], comp(i + DEPTH), p, jb, G, pitch, ta, tb);
    }
    if (whalo) {
#pragma unroll
        for (int s = 0; s < NS; s++)
            if (only < 0 || only == s) halo_finish<true>(hact, hv, tile0 + s * NG * LDS_TILE, LDS_TILE, hig, hph, gw[s]);
    }
}

// ------------------------------------------------------------------------------------------------
// accumulate, several trial sources per workgroup (round 3)
//
// Measured on accumulate_grouped_kernel (cfg3, 1024 sources per launch): build phase alone 22.8 ms, apply phase alone
// 22.7 ms, together 38.6 ms -- and the build alone moves 713 GB from the L2s to the CUs (40 node rows of 1088 samples per
// group and workgroup) in those 22.8 ms: 31 TB/s of the ~34.5 TB/s the L2s deliver.  The build is bound by L2 bandwidth,
// the apply by vector issue, and a workgroup alternates between the two.  What shrinks the first: NEIGHBOURING trial sources
// of a grid search put their sub-faults into the same cells of the Green's function grid (a strike step of 0.1 degree moves a
// sub-fault by metres, the nodes are kilometres apart): the 40 node rows a group needs are the same for all of them, only
// the four blend weights differ.  This kernel gives a workgroup the same (receiver, tile) of NS = 2 or 4 CONSECUTIVE trial
// sources, tiles of 1024 / NS samples: every group's node rows are loaded ONCE and blended NS times, into a tile set per
// source (NS x (1024 / NS + 64) x NG floats of LDS: 46 / 51 KB, three workgroups per CU); each wave then applies the
// centroids of ONE of the sources from that source's tile set (carry2_apply, as accumulate_grouped_kernel).  Per output
// sample: 1 / NS of the L2 traffic and of the build's addresses, descriptors and loads (the blend arithmetic stays), the
// apply as before.  Where the sources' groups do NOT all sit in the same cell with the same tile origin, the workgroup
// builds the tile sets one after the other.
//
// Grouping (host, eval): aligned groups of NS sources of a chunk whose centroid tables have the same STRUCTURE -- same
// number of centroids, same boundaries of the centroid groups (group_len) -- so that they walk their groups in lockstep;
// `mate[k]` says so for group k.  Of those the kernel takes the (group, receiver) combinations where every source is "clean"
// for the receiver (horizontal and vertical components, no missing trace, no tail rule: pairflag of geometry_kernel);
// accumulate_grouped_kernel runs behind it for everything else (pairsel 3).
// maximum / minimum of wave-uniform integers on the scalar unit (left to itself the compiler moves them to the vector pipe
// for its three-operand forms)
__device__ __forceinline__ int smax_u(int a, int b) { int r; asm("s_max_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
__device__ __forceinline__ int smin_u(int a, int b) { int r; asm("s_min_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }

template <int NS>
__device__ __forceinline__ bool multi_taken(const RecvDev &rv, const int *__restrict__ pairflag, const int *__restrict__ mate,
                                            int s, int nrec, int r)
{
    if (!mate) return false;
    const int a = s - s % NS;
    if (!mate[a / NS] || !rv.need_h || !rv.has_d) return false;
    int f = 0;
#pragma unroll
    for (int i = 0; i < NS; i++) f |= pairflag[(size_t)(a + i) * nrec + r];
    return f == 0;
}

template <int NG, bool FUSE, int NS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void accumulate_multi_kernel(
    const float *__restrict__ G, const int2 *__restrict__ span, int pitch,
    const GeoRec *__restrict__ recs, const int *__restrict__ cent_ofs, int isrc0, int nrec,
    const RecvDev *__restrict__ recv, float *__restrict__ syn, size_t syn_stride, int ntiles,
    const int *__restrict__ tab, FuseParams fp, const int *__restrict__ pairflag, const int *__restrict__ mate,
    const int *__restrict__ mate_wider /* NS = 2: the groups of four the wider launch has taken (or null) */)
{
    static_assert(NS == 2 || NS == 4, "sources per workgroup");
    constexpr int TILE = 1024 / NS, LDS_TILE = TILE + kHalo, K = LDS_TILE / 64, WPS = 4 / NS;       // WPS: waves per source
    __shared__ __attribute__((aligned(16))) float tiles[NS][NG][LDS_TILE];
    const int s0 = NS * (int)blockIdx.x;                 // chunk-local sources s0 .. s0 + NS - 1
    const int tile = blockIdx.y % ntiles, r = blockIdx.y / ntiles;
    const RecvDev &rv = recv[r];
    if (!rv.enabled) return;
    if (tile * TILE >= rv.wlen) return;
    if (!multi_taken<NS>(rv, pairflag, mate, s0, nrec, r)) return;
    if constexpr (NS == 2) { if (multi_taken<4>(rv, pairflag, mate_wider, s0, nrec, r)) return; }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sh = wv / WPS;                             // the source this wave applies
    const int t_tile0 = rv.wbeg + tile * TILE;
    const int cb = cent_ofs[isrc0];
    const int nc = cent_ofs[isrc0 + s0 + 1] - cent_ofs[isrc0 + s0];      // (every source of the group has nc centroids)
    const size_t base_me = (size_t)(cent_ofs[isrc0 + s0 + sh] - cb) * nrec + (size_t)r * nc;
    const GeoRec *__restrict__ rc = recs + base_me;      // this wave's source
    const float sd = rv.sd;
    (void)sd;
    const int u0 = 256 * (wv % WPS) + lane;              // the lane's first sample of its source's tile (the others: + 64 q)

    f2v ar1[2] = { { 0.f, 0.f }, { 0.f, 0.f } }, ar2[2] = { { 0.f, 0.f }, { 0.f, 0.f } }, dz[2] = { { 0.f, 0.f }, { 0.f, 0.f } };
    f2v xa0, xa1, xa2, xa3, xa4, xa5, xa6, xa7, xa8, xa9, xb0, xb1, xb2, xb3, xb4, xb5, xb6, xb7, xb8, xb9;
    f2v ya0, ya1, ya2, ya3, ya4, ya5, ya6, ya7, ya8, ya9, yb0, yb1, yb2, yb3, yb4, yb5, yb6, yb7, yb8, yb9;
    typedef typename Set2Sel<NG>::type SetT;
    const SetT X = make_set2<NG>(xa0, xa1, xa2, xa3, xa4, xa5, xa6, xa7, xa8, xa9, xb0, xb1, xb2, xb3, xb4, xb5, xb6, xb7, xb8, xb9);
    const SetT Y = make_set2<NG>(ya0, ya1, ya2, ya3, ya4, ya5, ya6, ya7, ya8, ya9, yb0, yb1, yb2, yb3, yb4, yb5, yb6, yb7, yb8, yb9);

    // head records of the group starting at c of EVERY source, load descriptors of the first one, lane-distributed
    auto recs_of = [&](int i) { return recs + ((size_t)(cent_ofs[isrc0 + s0 + i] - cb) * nrec + (size_t)r * nc); };
    auto tab_of = [&](int i) { return tab + ((size_t)(cent_ofs[isrc0 + s0 + i] - cb) * nrec + (size_t)r * nc) * 128; };
    int c = 0;
    int cur[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) cur[i] = rec_load(recs_of(i), 0, nc, lane);
    int ta = tab_of(0)[lane], tb = tab_of(0)[64 + lane];
    // halo: one lane per (4-sample chunk, component), chunk-major from the last thread down (see multi_build)
    const int hq = 255 - tid, hch = min(hq / NG, 15), hig = hq % NG, hph = TILE + 4 * hch;
    while (c < nc) {
        GeoRec g[NS];
        int smaxs[NS], smins[NS], npos = 0;
        bool shared = true, same_rows = true;
#pragma unroll
        for (int i = 0; i < NS; i++) {
            // of the head record: blend weights, integer shift and group hint of every source; the four node rows of the first
            // one only -- the others' are compared where they lie (lanes 0 .. 3 of the lane-distributed records)
            g[i].w[0] = REC_F(cur[i], 4); g[i].w[1] = REC_F(cur[i], 5); g[i].w[2] = REC_F(cur[i], 6); g[i].w[3] = REC_F(cur[i], 7);
            g[i].ishift = REC_I(cur[i], 8);
            g[i].pad = REC_I(cur[i], 19);
            smaxs[i] = g[i].ishift + ((g[i].pad >> 8) & 0xff);
            smins[i] = g[i].ishift - ((g[i].pad >> 16) & 0xff);
            npos = max(npos, TILE + (smaxs[i] - smins[i]) + 8);
            // the node rows once for all -- if all sit in the same cell and read it from the same tile origin
            if (i == 0) g[0].row[0] = REC_I(cur[0], 0);
            else {
                same_rows = same_rows && (__builtin_amdgcn_ballot_w64(cur[i] != cur[0]) & 0xfull) == 0ull;
                shared = shared && smaxs[i] == smaxs[0];
            }
        }
        shared = shared && same_rows;
        if (same_rows && !shared) {
            // the sources of a time sweep: same cell, shifts a few samples apart.  ONE tile origin all the same -- that of the
            // largest shift -- when the shifts of all of them still fit the tile's halo; every source counts its positions from it
            int smax_c = smaxs[0], smin_c = smins[0];
#pragma unroll
            for (int i = 1; i < NS; i++) { smax_c = smax_u(smax_c, smaxs[i]); smin_c = smin_u(smin_c, smins[i]); }
            if (TILE + (smax_c - smin_c) + 8 <= LDS_TILE) {
                shared = true;
                npos = TILE + (smax_c - smin_c) + 8;
#pragma unroll
                for (int i = 0; i < NS; i++) smaxs[i] = smax_c;
            }
        }
        const int glen = g[0].pad & 0xff;                // (same structure: equal for all)
        const int cend = c + glen;
        // this wave's source: what its apply needs (equal for all centroids of the group: same point, same receiver)
        int curme = cur[0], smax = smaxs[0];
#pragma unroll
        for (int i = 1; i < NS; i++) if (sh == i) { curme = cur[i]; smax = smaxs[i]; }
        const int flags = REC_I(curme, 18);
        const float gcl = REC_F(curme, 16), gsl = REC_F(curme, 17);
        int ishv = 0;
        if (lane < glen) ishv = rc[c + lane].ishift;
        // ---- build
#define KIWI_MULTI_BUILD(TA, TB, I0, ONLY, NPOS) do { \
            const int jb_ = t_tile0 - smaxs[I0] - 1;      /* LDS position p of a tile set holds its source's blended trace sample jb + p */ \
            const float *__restrict__ Gg = G + (size_t)g[I0].row[0] * (size_t)pitch; \
            const bool lane_ok = lane >= 4 * NG || ((TA) + jb_ >= (TB) && (TA) + jb_ + LDS_TILE <= (TB) + pitch); \
            const bool fast = __builtin_amdgcn_ballot_w64(lane_ok) == ~0ull; \
            const bool hact = hq < 16 * NG && hph < (NPOS); \
            const bool whalo = __builtin_amdgcn_ballot_w64(hact) != 0ull; \
            if (fast) multi_build<NG, true, NS>(&tiles[0][0][0], wv, lane, tid, jb_, Gg, pitch, TA, TB, g, ONLY, hact, whalo, hig, hph); \
            else      multi_build<NG, false, NS>(&tiles[0][0][0], wv, lane, tid, jb_, Gg, pitch, TA, TB, g, ONLY, hact, whalo, hig, hph); } while (0)
        // A workgroup in its build phase is waiting for memory most of the time: its waves go first whenever they have an
        // instruction to issue (loads out early, the blend done as soon as the rows arrive), the workgroups that are applying
        // fill the rest of the issue slots.  Measured: cfg3 133.5 -> 127.2 ms per 4096 sources, cfg3-100pt 55.2 -> 51.3 (the
        // other way round -- apply first -- 137.8 / 57.5; priority 1, 2 and 3 alike).
        __builtin_amdgcn_s_setprio(1);
#ifndef KIWI_X_NOBUILD
        if (shared) KIWI_MULTI_BUILD(ta, tb, 0, -1, npos);
        else {
            KIWI_MULTI_BUILD(ta, tb, 0, 0, npos);
#pragma unroll
            for (int i = 1; i < NS; i++) {               // (descriptors of the others only here: not kept in registers)
                const int tai = tab_of(i)[(size_t)c * 128 + lane], tbi = tab_of(i)[(size_t)c * 128 + 64 + lane];
                g[i].row[0] = REC_I(cur[i], 0);
                KIWI_MULTI_BUILD(tai, tbi, i, i, npos);
            }
        }
#else
        (void)shared;
#endif
#undef KIWI_MULTI_BUILD
        // head records and descriptors of the NEXT group: in flight while this group is applied
#pragma unroll
        for (int i = 0; i < NS; i++) cur[i] = rec_load(recs_of(i), cend, nc, lane);
        if (cend < nc) { ta = tab_of(0)[(size_t)cend * 128 + lane]; tb = tab_of(0)[(size_t)cend * 128 + 64 + lane]; }
        __syncthreads();
        __builtin_amdgcn_s_setprio(0);
        // ---- apply: this wave's source from its tile set.  (No tail rule and no partly added centroid here: multi_taken()
        // admits only pairs whose rows all end in zero and whose centroids all find all their traces.)
        {
            const size_t crow = (base_me + c) * 128 + 64 + 40;
            const unsigned clo = __builtin_amdgcn_readfirstlane((unsigned)crow), chi = __builtin_amdgcn_readfirstlane((unsigned)(crow >> 32));
            const float *__restrict__ coef_grp = (const float *)(tab + (((size_t)chi << 32) | clo));
            const unsigned abase = (unsigned)(size_t)(lds_cfp)&tiles[sh][0][u0];
            int cc = c, eprev = 0;
            bool have = false;
#ifdef KIWI_X_NOAPPLY
            cc = cend;
#endif
#define KIWI_C2STEP(LL, HH, RV, CC) do { \
                const int e = smax - __builtin_amdgcn_readlane(ishv, (CC) - c);       /* LDS position of b[j-1] of the tile's first sample */ \
                const int d = have ? eprev - e : 0x7fff; \
                carry2_apply<NG, RV, K>(ar1, ar2, dz, LL, HH, abase + 4u * (unsigned)e, KIWI_X_FULL(d), KIWI_X_FULL(d), coef_grp + (size_t)((CC) - c) * 128, gcl, gsl); \
                have = true; eprev = e; } while (0)
            if (flags & 2) {
                set2_dead(X); set2_dead(Y);
                for (; cc + 1 < cend; cc += 2) { KIWI_C2STEP(X, Y, true, cc); KIWI_C2STEP(Y, X, true, cc + 1); }
                if (cc < cend) KIWI_C2STEP(X, Y, true, cc);
            } else {
                set2_dead(X); set2_dead(Y);
                for (; cc + 1 < cend; cc += 2) { KIWI_C2STEP(X, Y, false, cc); KIWI_C2STEP(Y, X, false, cc + 1); }
                if (cc < cend) KIWI_C2STEP(X, Y, false, cc);
            }
#undef KIWI_C2STEP
        }
#ifdef KIWI_X_NOBUILD
        __syncthreads();
#endif
        c = cend;                                        // (the barrier in front of the next group's LDS writes: multi_build)
    }
    // ---- rotation to N/E, signs and store (or fused comparison) of this wave's source (seismogram.f90:256-283)
    {
        const int js = s0 + sh;
        const int tl = tile * TILE + u0;                 // window sample of the lane's output q = 0; q-th: + 64 q
        if (!FUSE && tl >= rv.wlen) return;
        float *__restrict__ so = syn + (size_t)js * syn_stride + tl;
        float mom = 0.f;
        if constexpr (FUSE) mom = fp.moment[fp.isrc0 + js];
        const bool whole = tile * TILE + TILE <= rv.wlen;
        for (int k = 0; k < rv.ncomp; k++) {
            f2v o01, o23;
            out4(rv.comp[k], rv.sign[k], rv.cl0, rv.sl0, ar1, ar2, dz, o01, o23);
            if constexpr (!FUSE) {
                const float o[4] = { o01.x, o01.y, o23.x, o23.y };
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (tl + 64 * i < rv.wlen) so[rv.synofs[k] + 64 * i] = o[i];
                continue;
            }
            double acc = fused_acc4(fp, o01, o23, mom, fp.reft + rv.refofs[k] + tl, fp.tw + rv.refofs[k] + tl, tl, rv.wlen, whole);
            acc = wave_reduce_f64(acc, fp.method == 6);                 // total in lane 63
            if (lane == 63)
                fp.partial[((size_t)js * fp.nmis + rv.slot0 + k) * fp.nparts + tile * WPS + (wv % WPS)] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// accumulate, cell groups: the raw node traces stay in registers across the centroids of one GF cell
//
// Sources whose sub-faults are all different points (eikonal ruptures: one centroid per cell of the rupture grid) give
// accumulate_grouped_kernel groups of ONE centroid: every centroid re-fetches its 4 x NG node rows although its
// neighbours in the table sit in the same 4-node cell of the Green's function grid and need the very same rows -- only
// the four blend weights change.  Here a group is a run of consecutive centroids in the same cell (cellgroup_kernel):
// the workgroup loads the raw rows of the tile ONCE per group into registers (4 x NG dwordx4 per lane) and, per centroid,
// blends them with that centroid's weights (gfdb.f90:944-949, same order) into the LDS tile the apply phase reads, exactly
// as the grouped kernel does after its loads.  Per output sample the operations and their order are those of
// accumulate_kernel, so the results are bit-identical.  The halo (positions beyond the tile that the shifts reach) is held
// the same way: one 4-sample chunk of one component per halo lane, 16 more registers.
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));   // dword-aligned 8-byte load
template <int SPL> struct RawVec;
template <> struct RawVec<4> { typedef f4u type; };
template <> struct RawVec<2> { typedef f2u type; };

// raw rows of one cell over the tile's main chunk (SPL samples per lane at LDS position p): always the clamped form of the
// address (no separate clamp-free variant: the loads run once per GROUP) and always all four nodes: a centroid exactly on
// a node carries the weights (1, 0, 0, 0) and four times the same row, and 1 v + 0 v + 0 v + 0 v is v bit for bit (also
// for -0), so no separate unblended form is needed (gfdb.f90:890-893 vs :944-949)
// The components one workgroup of accumulate_cell_kernel works on, in APPLICATION order (seismogram.f90:171-250):
// PART 0 = all of them; PART 1 = the horizontal block 1 2 3 [9] 4 5 (radial and transverse sums, rotated into the north /
// east or away / right traces), PART 2 = the vertical block 6 7 8 [10].  Two workgroups per (source, tile, receiver) halve
// the raw rows a lane keeps in registers (more waves per SIMD) at the price of doing the per-centroid bookkeeping twice.
template <int NG, int PART> struct CellPart;
template <> struct CellPart<10, 0> { static constexpr int n = 10; static constexpr int first = 0; __device__ static constexpr int ig(int i) { return i < 3 ? i : (i == 3 ? 8 : (i < 9 ? i - 1 : 9)); } };
template <> struct CellPart<8, 0>  { static constexpr int n = 8; static constexpr int first = 0; __device__ static constexpr int ig(int i) { return i; } };
template <> struct CellPart<10, 1> { static constexpr int n = 6; static constexpr int first = 0; __device__ static constexpr int ig(int i) { return i < 3 ? i : (i == 3 ? 8 : i - 1); } };
template <> struct CellPart<10, 2> { static constexpr int n = 4; static constexpr int first = 6; __device__ static constexpr int ig(int i) { return i < 3 ? 5 + i : 9; } };
template <> struct CellPart<8, 1>  { static constexpr int n = 5; static constexpr int first = 0; __device__ static constexpr int ig(int i) { return i; } };
template <> struct CellPart<8, 2>  { static constexpr int n = 3; static constexpr int first = 5; __device__ static constexpr int ig(int i) { return 5 + i; } };

template <int NG, int PART, int SPL> using RawArr = typename RawVec<SPL>::type[CellPart<NG, PART>::n][4];

// FAST: the whole tile lies inside every row of the cell (tab[52], tab[53]; the caller's test): no clamp -- the lane's
// position is the same vector offset for all loads and the row's sample-0 position goes into the load's scalar offset: one
// scalar instruction per load and no vector one
// UNI (with FAST; tab[54]): the rows of a node are `pitch` apart AND start at the same sample: the four descriptors of component 0
// give all forty positions
template <int NG, int PART, int SPL, bool FAST = false, bool UNI = false>
__device__ __forceinline__ void raw_issue(RawArr<NG, PART, SPL> &v, int p, int jb,
                                          const float *__restrict__ G, int pitch, int ta, int tb)
{
    typedef typename RawVec<SPL>::type RV;
    typedef CellPart<NG, PART> P;
    __builtin_assume(pitch >= 8);
    const int p4 = 4 * p, hi4 = 4 * (pitch - 4);
    if constexpr (FAST) {
        const unsigned jb4 = 4u * (unsigned)jb;
        unsigned node0[4] = { 0u, 0u, 0u, 0u };
        if constexpr (UNI) {
#pragma unroll
            for (int k = 0; k < 4; k++) node0[k] = 4u * (unsigned)REC_I(ta, k) + jb4;
        }
#pragma unroll
        for (int i = 0; i < P::n; i++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // byte position of tile sample 0 in the cell
                const unsigned so = UNI ? node0[k] + (unsigned)P::ig(i) * (4u * (unsigned)pitch)
                                        : 4u * (unsigned)REC_I(ta, 4 * P::ig(i) + k) + jb4;
                if constexpr (SPL == 4) {
                    const v4i_t w = __builtin_amdgcn_raw_buffer_load_b128(gf_rsrc(G), p4, (int)so, 0);
                    v[i][k] = RV{ __int_as_float(w.x), __int_as_float(w.y), __int_as_float(w.z), __int_as_float(w.w) };
                } else {
                    const v2i_t w = __builtin_amdgcn_raw_buffer_load_b64(gf_rsrc(G), p4, (int)so, 0);
                    v[i][k] = RV{ __int_as_float(w.x), __int_as_float(w.y) };
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < P::n; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // row start (wave-uniform: scalar base address) + this lane's clamped position inside the row (32-bit offset):
            // one address register per load instead of a 64-bit pair, and nothing here overflows for tensors beyond 2^31 bytes
            const int base = REC_I(ta, 4 * P::ig(i) + k), lo = REC_I(tb, 4 * P::ig(i) + k);
            // lane position inside the row, clamped (one v_add + one v_med3 per load); the row start is the load's scalar offset
            // byte offset of LDS position 0 inside the row: scalar arithmetic (kept there by hand -- left to itself the
            // compiler re-associates it into the lanes), then one add and one v_med3 per load for the lane's clamped position
            int u4, q4;
            asm("s_sub_i32 %0, %1, %2\n\ts_add_i32 %0, %0, %3\n\ts_lshl_b32 %0, %0, 2" : "=&s"(u4) : "s"(base), "s"(lo), "s"(jb) : "scc");
            asm("v_add_u32 %0, %1, %2\n\tv_med3_i32 %0, %0, 0, %3" : "=&v"(q4) : "s"(u4), "v"(p4), "s"(hi4));
            if constexpr (SPL == 4) {
                const v4i_t w = __builtin_amdgcn_raw_buffer_load_b128(gf_rsrc(G), q4, (int)(4u * (unsigned)lo), 0);
                v[i][k] = RV{ __int_as_float(w.x), __int_as_float(w.y), __int_as_float(w.z), __int_as_float(w.w) };
            } else {
                const v2i_t w = __builtin_amdgcn_raw_buffer_load_b64(gf_rsrc(G), q4, (int)(4u * (unsigned)lo), 0);
                v[i][k] = RV{ __int_as_float(w.x), __int_as_float(w.y) };
            }
        }
    }
}

template <int NG, int PART, int SPL>
__device__ __forceinline__ void raw_blend_store(const RawArr<NG, PART, SPL> &v, float *__restrict__ tile0,
                                                int lds_tile, int p, float w0, float w1, float w2, float w3)
{
    typedef typename RawVec<SPL>::type RV;
#pragma unroll
    for (int i = 0; i < CellPart<NG, PART>::n; i++) {
        RV b = w0 * v[i][0];                      // gfdb.f90:946-949, summed in this order
        b = b + w1 * v[i][1];
        b = b + w2 * v[i][2];
        b = b + w3 * v[i][3];
        if constexpr (SPL == 4) *(float4 *)(tile0 + i * lds_tile + p) = make_float4(b.x, b.y, b.z, b.w);
        else                    *(float2 *)(tile0 + i * lds_tile + p) = make_float2(b.x, b.y);
    }
}

// One centroid's components of PART from the LDS tiles (tile i = i-th component of the part in application order); the same
// operations in the same order as centroid_apply_hd performs for these components.
template <int NG, int PART, int LDS_TILE, bool TAIL, int NP>
__device__ __forceinline__ void cell_apply(f2v (&ar1)[NP], f2v (&ar2)[NP], f2v (&dz)[NP], const TileBase &chunk0, int jl,
                                           const int (&jend)[CellPart<NG, PART>::n], int flags,
                                           const float (&cw)[2 * CellPart<NG, PART>::n] /* (wl, wr) per component in application order */,
                                           int rec, float sd, float cl, float sl)
{
    typedef CellPart<NG, PART> P;
    constexpr int nH1 = (NG == 10) ? 4 : 3;      // components summed into the radial trace
    constexpr int nH = (NG == 10) ? 6 : 5;       // horizontal block
    constexpr int g0 = P::first;                 // application-order index of the part's first component
    float fac[P::n];
#pragma unroll
    for (int i = 0; i < P::n; i++) fac[i] = 0.f;
    if constexpr (TAIL) {                         // the tail rule needs the plain factors (sparse_trace.f90:698-703)
        const float f0 = REC_F(rec, 10), f1 = REC_F(rec, 11), f2 = REC_F(rec, 12), f3 = REC_F(rec, 13), f4 = REC_F(rec, 14), f5 = REC_F(rec, 15);
        const float all10[10] = { f0, f1, f2, f5, f3, f4, f0 * sd, f1 * sd, f2 * sd, f5 * sd };
        const float all8[8] = { f0, f1, f2, f3, f4, f0 * sd, f1 * sd, f2 * sd };
#pragma unroll
        for (int i = 0; i < P::n; i++) fac[i] = (NG == 10) ? all10[g0 + i] : all8[g0 + i];
    }
    constexpr int kAhead = 2;
    TileRegsN<NP> tr[P::n];
#pragma unroll
    for (int i = 0; i < kAhead && i < P::n; i++) tr[i] = tile_load<NP>(chunk0, i * LDS_TILE);
    const bool rot = (flags & 2) != 0;           // seismogram.f90:160-203 vs :205-231
    f2v t1[NP], t2[NP];
    if constexpr (PART != 2) {
#pragma unroll
        for (int k = 0; k < NP; k++) { t1[k] = rot ? f2v{ 0.f, 0.f } : ar1[k]; t2[k] = rot ? f2v{ 0.f, 0.f } : ar2[k]; }
    }
#pragma unroll
    for (int i = 0; i < P::n; i++) {
        if (i + kAhead < P::n) tr[i + kAhead] = tile_load<NP>(chunk0, (i + kAhead) * LDS_TILE);
        __builtin_amdgcn_sched_barrier(0);
        const int a = g0 + i;                    // application-order index
        if (a < nH1)     tile_fma<TAIL, NP>(t1, tr[i], jl, jend[i], fac[i], cw[2 * i], cw[2 * i + 1]);
        else if (a < nH) tile_fma<TAIL, NP>(t2, tr[i], jl, jend[i], fac[i], cw[2 * i], cw[2 * i + 1]);
        else             tile_fma<TAIL, NP>(dz, tr[i], jl, jend[i], fac[i], cw[2 * i], cw[2 * i + 1]);
        if (a == nH - 1) {
            if (rot) {
#pragma unroll
                for (int k = 0; k < NP; k++) {
                    ar1[k] = ar1[k] + cl * t1[k] - sl * t2[k];
                    ar2[k] = ar2[k] + cl * t2[k] + sl * t1[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < NP; k++) { ar1[k] = t1[k]; ar2[k] = t2[k]; }
            }
        }
    }
}

// waves per SIMD the register allocation aims at: the whole component set (PART 0) keeps 80 + 16 raw registers per lane and
// runs best at two, a half set (PART 1 / 2) at three
#ifndef KIWI_CELL_WAVES
#define KIWI_CELL_WAVES(PART) ((PART) == 0 ? 2 : 3)
#endif
// measurement switches (wrong results; phase timings in DESIGN.md): -DKIWI_XC_NOAPPLY / NOBLEND / NOBARRIER / NOLOAD
#ifndef KIWI_XC_NOAPPLY
#define KIWI_XC_NOAPPLY 0
#endif
#ifndef KIWI_XC_NOBLEND
#define KIWI_XC_NOBLEND 0
#endif
#ifndef KIWI_XC_NOBARRIER
#define KIWI_XC_NOBARRIER 0
#endif
#ifndef KIWI_XC_NOLOAD
#define KIWI_XC_NOLOAD 0
#endif
// SPL: output samples per lane (tile = SPL * T samples).  With four the raw rows of a group take 160 registers per lane and
// leave room for one wave per SIMD only; with two they take 80 and the kernel keeps the grouped kernel's three waves.
//
// Takes the (source, receiver) pairs of cell_pair(): receivers with horizontal and vertical components, no centroid with a
// missing trace; accumulate_grouped_kernel is launched behind it for the other pairs.
//
// Schedule.  Two LDS tile sets: while centroid k is applied from one, centroid k + 1 of the group is blended into the
// other (ONE barrier per centroid, and the blend's arithmetic fills the issue slots the apply's LDS reads leave).  During
// the last apply of a group the raw registers are free again, so the loads of the NEXT group's rows are issued there and
// their L2 round trip is hidden behind that apply.
template <int NG, int T, int SPL, int PART, bool FUSE>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(KIWI_CELL_WAVES(PART)))) void accumulate_cell_kernel(
    const float *__restrict__ G, const int2 *__restrict__ span, int pitch,
    const GeoRec *__restrict__ recs, const int *__restrict__ cent_ofs, int isrc0, int nrec,
    const RecvDev *__restrict__ recv, float *__restrict__ syn, size_t syn_stride, int ntiles,
    const int *__restrict__ tab, FuseParams fp, const int *__restrict__ pairflag, const int *__restrict__ synrow,
    const int *__restrict__ fam_ofs, const int *__restrict__ fam_list)
{
    constexpr int NP = SPL / 2;
    constexpr int TILE = SPL * T;
    constexpr int LDS_TILE = TILE + kHalo;
    static_assert(LDS_TILE % 64 == 0, "ds_read2st64 offsets");
    typedef CellPart<NG, PART> P;
    static_assert(16 * P::n <= T, "one halo chunk per lane");
    __shared__ __attribute__((aligned(16))) float tiles[2][P::n][LDS_TILE];
    const int s = (int)blockIdx.x;                        // source index fastest (see accumulate_grouped_kernel)
    const int tile = blockIdx.y % ntiles, r = blockIdx.y / ntiles;
    const RecvDev &rv = recv[r];
    if (!rv.enabled) return;
    if (tile * TILE >= rv.wlen) return;
    if (!cell_pair(rv, pairflag, s, nrec, r)) return;
    if (synrow && synrow[s] != s) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int t_tile0 = rv.wbeg + tile * TILE;
    const int cb = cent_ofs[isrc0], c0 = cent_ofs[isrc0 + s], nc = cent_ofs[isrc0 + s + 1] - c0;
    const GeoRec *__restrict__ rc = recs + ((size_t)(c0 - cb) * nrec + (size_t)r * nc);
    const int *__restrict__ tc = tab + ((size_t)(c0 - cb) * nrec + (size_t)r * nc) * 128;
    const float sd = rv.sd;
    const int u0 = SPL * (tid & ~63) + lane;             // this lane's first tile sample (the others: + 64 q)
    // halo: lane q < 16 n owns the 4-sample chunk q / n of the part's component q % n beyond the tile.  Chunk-major: a
    // group needs the first few chunks only (its shift range plus 8 samples), so the active lanes are the first ones and
    // the waves behind them skip the halo's blend
    const int hloc = tid % P::n, hph = TILE + 4 * (tid / P::n);
    int hig = P::ig(0);
#pragma unroll
    for (int i = 1; i < P::n; i++) if (hloc == i) hig = P::ig(i);
    const bool hslot = tid < 16 * P::n;

    f2v ar1[NP], ar2[NP], dz[NP];
#pragma unroll
    for (int h = 0; h < NP; h++) { ar1[h] = f2v{ 0.f, 0.f }; ar2[h] = f2v{ 0.f, 0.f }; dz[h] = f2v{ 0.f, 0.f }; }
    typename RawVec<SPL>::type raw[P::n][4];             // raw rows of the current group over the tile's main chunk
    HaloRegs hraw;                                       // ... and over the halo chunk of this lane

    // issue the loads of the group with head fields (pad, ishift) and descriptors ta_ / tb_
#define CELL_LOAD(head_row0, head_pad, head_ishift, ta_, tb_) do { \
        const int smax_ = (head_ishift) + (((head_pad) >> 8) & 0xff), smin_ = (head_ishift) - (((head_pad) >> 16) & 0xff); \
        const int jb_ = t_tile0 - smax_ - 1, npos_ = TILE + (smax_ - smin_) + 8; \
        const float *__restrict__ Gg_ = G + (size_t)(head_row0) * (size_t)pitch;     /* descriptors are relative to it (write_tab) */ \
        /* every row of the cell holds the whole tile: rows start at or before it and end behind it */ \
        const bool inside_ = (REC_I(ta_, 52) + jb_ >= 0) && (REC_I(ta_, 53) + jb_ + SPL * (T - 1) <= pitch - 4); \
        if (inside_) raw_issue<NG, PART, SPL, true>(raw, SPL * tid, jb_, Gg_, pitch, ta_, tb_); \
        else         raw_issue<NG, PART, SPL, false>(raw, SPL * tid, jb_, Gg_, pitch, ta_, tb_); \
        hraw = halo_issue<true, false>(hslot && hph < npos_, hig, hph, jb_, Gg_, pitch, ta_, tb_); \
    } while (0)
    // blend the registers with the weights in lanes 4..7 of record `rec_` into tile set `buf_`
#define CELL_BLEND(rec_, buf_, npos_) do { \
        GeoRec gw_; \
        gw_.w[0] = REC_F(rec_, 4); gw_.w[1] = REC_F(rec_, 5); gw_.w[2] = REC_F(rec_, 6); gw_.w[3] = REC_F(rec_, 7); \
        float *tile0_ = &tiles[buf_][0][0]; \
        raw_blend_store<NG, PART, SPL>(raw, tile0_, LDS_TILE, SPL * tid, gw_.w[0], gw_.w[1], gw_.w[2], gw_.w[3]); \
        halo_finish<true>(hslot && hph < (npos_), hraw, tile0_, LDS_TILE, hloc, hph, gw_); \
    } while (0)

    // One load site inside the loop (the NEXT group's rows, issued in front of the last apply of the group, when the raw
    // registers are free: their L2 round trip hides behind that apply) and one in front of it for the first group.  The last
    // centroid of a group is peeled off the loop over its centroids so that this redefinition of the raw registers is
    // straight-line code: as a conditional inside the loop it cost ~130 register copies per group.  (Pairs with skipped
    // centroids, row < 0, are not cell pairs: accumulate_grouped_kernel has them.)
    int c = 0;
    int cur = rec_load(rc, 0, nc, lane);                 // record c, lane-distributed; nx1: record c + 1
    int nx1 = rec_load(rc, 1, nc, lane);
    int ta = 0, tb = 0;                                  // load descriptors of record c (none for a source without centroids)
    if (nc > 0) {
        ta = tc[lane]; tb = tc[64 + lane];
        CELL_LOAD(REC_I(cur, 0), REC_I(cur, 19), REC_I(cur, 8), ta, tb);
    }
    // apply centroid cc_ of the group from tile set bsel (as accumulate_grouped_kernel); its record is `cur`
#define CELL_APPLY(cc_) do { \
        const int flags = REC_I(cur, 18); \
        const int ishift = REC_I(cur, 8); \
        const float cl = REC_F(cur, 16), sl = REC_F(cur, 17); \
        const int e = smax - ishift; \
        const TileBase chunk0 = tile_base(&tiles[bsel][0][e + u0]); \
        const int jl = jb + e + u0; \
        const bool tail = (jb + e + TILE) > jend_min; \
        if (KIWI_XC_NOAPPLY) { ar1[0].x += (float)(jl + (int)tail) * cl + cw[0]; } \
        else if (!tail) { const int nojend[P::n] = {}; cell_apply<NG, PART, LDS_TILE, false, NP>(ar1, ar2, dz, chunk0, jl, nojend, flags, cw, cur, sd, cl, sl); } \
        else { \
            int jend[P::n];                     /* end indices: the tail rule only (kept out of the scalar registers otherwise) */ \
            _Pragma("unroll") for (int i = 0; i < P::n; i++) jend[i] = REC_I(tg, 40 + P::ig(i)); \
            cell_apply<NG, PART, LDS_TILE, true, NP>(ar1, ar2, dz, chunk0, jl, jend, flags, cw, cur, sd, cl, sl); \
        } \
    } while (0)
    // the coefficient line of centroid cc_ (scalar loads), issued at the top of the centroid's step: in front of the blend of
    // the next centroid, whose arithmetic covers its round trip (the table comes from HBM).  Not a step earlier: scalar loads
    // share the counter of the LDS operations, and the drain in front of the barrier would wait for them
#define CELL_COEF(cc_) do { \
        const float *__restrict__ coef_ = coef_grp + (size_t)((cc_) - c) * 128 + 2 * P::first; \
        _Pragma("unroll") for (int i = 0; i < 2 * P::n; i++) cw[i] = coef_[i]; \
    } while (0)
    while (c < nc) {
        const int pad0 = REC_I(cur, 19), ishift0 = REC_I(cur, 8);
        const int cend = c + (pad0 & 0xff);
        const int smax = ishift0 + ((pad0 >> 8) & 0xff), smin = ishift0 - ((pad0 >> 16) & 0xff);
        const int jb = t_tile0 - smax - 1;               // LDS position p holds blended trace sample jb + p
        const int npos = TILE + (smax - smin) + 8;
        const int tg = ta;                               // this group's descriptor row (end indices for the tail rule)
        const int jend_min = PART == 0 ? min(REC_I(ta, 50), REC_I(ta, 51)) : REC_I(ta, PART == 1 ? 50 : 51);
        // descriptors of the NEXT group
        int ta_n = 0, tb_n = 0;
        if (cend < nc) { ta_n = tc[(size_t)cend * 128 + lane]; tb_n = tc[(size_t)cend * 128 + 64 + lane]; }
        const size_t crow = ((size_t)(c0 - cb) * nrec + (size_t)r * nc + c) * 128 + 64 + 40;
        const unsigned clo = __builtin_amdgcn_readfirstlane((unsigned)crow), chi = __builtin_amdgcn_readfirstlane((unsigned)(crow >> 32));
        const float *__restrict__ coef_grp = (const float *)(tab + (((size_t)chi << 32) | clo));
        float cw[2 * P::n];
        // ---- first centroid of the group: its tile goes into set 0 (every set is free after the barrier that ended the last group)
        if (!KIWI_XC_NOBLEND) CELL_BLEND(cur, 0, npos);
        if (!KIWI_XC_NOBARRIER) __syncthreads();
        int bsel = 0;
        for (int cc = c; cc + 1 < cend; cc++) {
            const int nx2 = rec_load(rc, cc + 2, nc, lane);      // two records ahead: the next one is needed for its weights now
            // ---- centroid cc + 1 of the group into the other tile set (a centroid at the point of its predecessor keeps the tile)
            const bool blend_next = !(REC_I(nx1, 18) & 4);
            CELL_COEF(cc);
            if (blend_next && !KIWI_XC_NOBLEND) CELL_BLEND(nx1, bsel ^ 1, npos);
            CELL_APPLY(cc);
            if (!KIWI_XC_NOBARRIER) __syncthreads();     // set bsel may be overwritten, set bsel ^ 1 is complete
            if (blend_next) bsel ^= 1;
            cur = nx1; nx1 = nx2;
        }
        {
            // ---- last centroid of the group: the raw registers are free, the next group's rows can be on their way
            const int nx2 = rec_load(rc, cend + 1, nc, lane);
            CELL_COEF(cend - 1);
            if (cend < nc && !(KIWI_XC_NOLOAD)) CELL_LOAD(REC_I(nx1, 0), REC_I(nx1, 19), REC_I(nx1, 8), ta_n, tb_n);
            CELL_APPLY(cend - 1);
            if (!KIWI_XC_NOBARRIER) __syncthreads();
            cur = nx1; nx1 = nx2;
        }
        ta = ta_n; tb = tb_n;
        c = cend;
    }
#undef CELL_APPLY
#undef CELL_COEF
#undef CELL_LOAD
#undef CELL_BLEND
    // ---- rotation to N/E, signs, store or fused comparator (seismogram.f90:256-283), as accumulate_grouped_kernel; with the
    // fused comparator also for the sources that share these synthetics (fam_list), each with its own moment
    const int nfam = (FUSE && fam_ofs) ? fam_ofs[s + 1] - fam_ofs[s] : 0;
    for (int qf = -1; qf < nfam; qf++) {
        const int js = qf < 0 ? s : fam_list[fam_ofs[s] + qf];
        const int tl = tile * TILE + u0;
        if (!FUSE && tl >= rv.wlen) return;
        float *__restrict__ so = syn + (size_t)js * syn_stride + tl;
        float a1[SPL], a2[SPL], ad[SPL];
#pragma unroll
        for (int h = 0; h < NP; h++) {
            a1[2 * h] = ar1[h].x; a1[2 * h + 1] = ar1[h].y; a2[2 * h] = ar2[h].x; a2[2 * h + 1] = ar2[h].y;
            ad[2 * h] = dz[h].x; ad[2 * h + 1] = dz[h].y;
        }
        float mom = 0.f;
        if constexpr (FUSE) mom = fp.moment[fp.isrc0 + js];
        const bool unit = (fp.syn_factor == 1.f);
        for (int k = 0; k < rv.ncomp; k++) {
            if (PART != 0 && (rv.comp[k] == 3) != (PART == 2)) continue;      // the vertical trace belongs to the PART 2 workgroup
            const float sg = rv.sign[k];
            float o[SPL];
#pragma unroll
            for (int i = 0; i < SPL; i++) {
                switch (rv.comp[k]) {
                case 1: o[i] = a1[i] * sg; break;
                case 2: o[i] = a2[i] * sg; break;
                case 3: o[i] = ad[i]; break;
                case 4: o[i] = (rv.cl0 * a1[i] - rv.sl0 * a2[i]) * sg; break;
                default: o[i] = (rv.cl0 * a2[i] + rv.sl0 * a1[i]) * sg; break;
                }
            }
            if constexpr (!FUSE) {
#pragma unroll
                for (int i = 0; i < SPL; i++)
                    if (tl + 64 * i < rv.wlen) so[rv.synofs[k] + 64 * i] = o[i];
                continue;
            }
            double acc = 0.0;
            const float *__restrict__ rt = fp.reft + rv.refofs[k] + tl, *__restrict__ tp = fp.tw + rv.refofs[k] + tl;
#pragma unroll
            for (int i = 0; i < SPL; i++) {
                if (tl + 64 * i >= rv.wlen) break;
                const float v = o[i] * mom;
                const float vt = v * tp[64 * i];
                const float a = rt[64 * i];
                switch (fp.method) {
                case 1: { const float d = unit ? (a - vt) : (1.f * a - fp.syn_factor * vt); acc = sq_acc(acc, d); break; }
                case 2: { const float d = unit ? fabsf(a - vt) : fabsf(1.f * a - fp.syn_factor * vt); acc += (double)d; break; }
                case 5: acc += unit ? (double)(a * vt) : (double)(a * 1.f * vt * fp.syn_factor); break;
                default: { const double x = (double)(1.f * a), y = (double)(fp.syn_factor * vt); acc = fmax(acc, sqrt(x * x + y * y)); break; }
                }
            }
            acc = wave_reduce_f64(acc, fp.method == 6);                 // total in lane 63
            if (lane == 63)
                fp.partial[((size_t)js * fp.nmis + rv.slot0 + k) * fp.nparts + tile * (T / 64) + (tid >> 6)] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// accumulate, cell groups, a tile per WAVE (r03)
//
// accumulate_cell_kernel shares its tile among the four waves of the workgroup: a barrier per centroid, at which three
// waves wait for the slowest (measured: 10 % of the kernel), and every wave in the same phase at the same time.  Here a wave
// owns the 128 output samples it computes AND the blended samples they need: it blends positions 0 .. 127 of its tile from
// the raw rows in its registers plus the few positions behind them that the group's shifts reach (one 4-sample chunk of one
// component per lane, lanes 0 .. 59: groups are cut at a shift range of kCellwRange), and reads them back itself.  LDS
// operations of one wave execute in order, so there is no barrier anywhere in the loop, and ONE tile set is enough: per
// component the step reads what the apply of centroid k needs, then writes the blend of centroid k + 1 over it, then does
// the arithmetic of both.  Waves drift apart and fill each other's stalls (loads of the next group's rows, the first blend of
// a group, LDS round trips).  Per output sample the operations and their order are those of accumulate_cell_kernel.
constexpr int kCellwRow = 192;      // floats per component row of a wave's tile: 128 + halo, a multiple of 64 (ds_read2st64 offsets)
constexpr int kCellwRange = 23;     // largest shift range of a group: 24 positions behind the 128 = six chunks x 10 components = 60 lanes

// One step of a wave: apply centroid k (record `rec`, coefficients cw, tile position `pos` = e + lane) and, BLEND, put centroid
// k + 1 (weights w0 .. w3) into the tile -- component by component, the write of a component's row behind the read of it.
template <int NG, bool TAIL, bool BLEND>
__device__ __forceinline__ void cellw_step(f2v (&ar1)[1], f2v (&ar2)[1], f2v (&dz)[1], float *__restrict__ wt, int pos,
                                           const RawArr<NG, 0, 2> &raw, float w0, float w1, float w2, float w3, int pl,
                                           bool hact, const HaloRegs &hraw, int hloc, int hph,
                                           int jl, const int (&jend)[CellPart<NG, 0>::n], int flags,
                                           const float (&cw)[2 * CellPart<NG, 0>::n], int rec, float sd, float cl, float sl)
{
    typedef CellPart<NG, 0> P;
    constexpr int nH1 = (NG == 10) ? 4 : 3;      // components summed into the radial trace
    constexpr int nH = (NG == 10) ? 6 : 5;       // horizontal block
    float fac[P::n];
#pragma unroll
    for (int i = 0; i < P::n; i++) fac[i] = 0.f;
    if constexpr (TAIL) {                         // the tail rule needs the plain factors (sparse_trace.f90:698-703)
        const float f0 = REC_F(rec, 10), f1 = REC_F(rec, 11), f2 = REC_F(rec, 12), f3 = REC_F(rec, 13), f4 = REC_F(rec, 14), f5 = REC_F(rec, 15);
        const float all10[10] = { f0, f1, f2, f5, f3, f4, f0 * sd, f1 * sd, f2 * sd, f5 * sd };
        const float all8[8] = { f0, f1, f2, f3, f4, f0 * sd, f1 * sd, f2 * sd };
#pragma unroll
        for (int i = 0; i < P::n; i++) fac[i] = (NG == 10) ? all10[i] : all8[i];
    }
    const TileBase chunk0 = tile_base(wt + pos);
    constexpr int kAhead = 2;
    TileRegsN<1> tr[P::n];
#pragma unroll
    for (int i = 0; i < kAhead && i < P::n; i++) tr[i] = tile_load<1>(chunk0, i * kCellwRow);
    const bool rot = (flags & 2) != 0;           // seismogram.f90:160-203 vs :205-231
    f2v t1[1], t2[1];
    t1[0] = rot ? f2v{ 0.f, 0.f } : ar1[0]; t2[0] = rot ? f2v{ 0.f, 0.f } : ar2[0];
#pragma unroll
    for (int i = 0; i < P::n; i++) {
        if (i + kAhead < P::n) tr[i + kAhead] = tile_load<1>(chunk0, (i + kAhead) * kCellwRow);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BLEND) {                    // row i has been read (kAhead iterations ago): its next content may go in
            f2u b = w0 * raw[i][0];               // gfdb.f90:946-949, summed in this order
            b = b + w1 * raw[i][1];
            b = b + w2 * raw[i][2];
            b = b + w3 * raw[i][3];
            *(float2 *)(wt + i * kCellwRow + pl) = make_float2(b.x, b.y);
        }
        if (i < nH1)     tile_fma<TAIL, 1>(t1, tr[i], jl, jend[i], fac[i], cw[2 * i], cw[2 * i + 1]);
        else if (i < nH) tile_fma<TAIL, 1>(t2, tr[i], jl, jend[i], fac[i], cw[2 * i], cw[2 * i + 1]);
        else             tile_fma<TAIL, 1>(dz, tr[i], jl, jend[i], fac[i], cw[2 * i], cw[2 * i + 1]);
        if (i == nH - 1) {
            if (rot) {
                ar1[0] = ar1[0] + cl * t1[0] - sl * t2[0];
                ar2[0] = ar2[0] + cl * t2[0] + sl * t1[0];
            } else {
                ar1[0] = t1[0]; ar2[0] = t2[0];
            }
        }
    }
    if constexpr (BLEND) {                        // behind every read of this step
        GeoRec gw;
        gw.w[0] = w0; gw.w[1] = w1; gw.w[2] = w2; gw.w[3] = w3;
        halo_finish<true>(hact, hraw, wt, kCellwRow, hloc, hph, gw);
    }
}

template <int NG, bool FUSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void accumulate_cellw_kernel(
    const float *__restrict__ G, const int2 *__restrict__ span, int pitch,
    const GeoRec *__restrict__ recs, const int *__restrict__ cent_ofs, int isrc0, int nrec,
    const RecvDev *__restrict__ recv, float *__restrict__ syn, size_t syn_stride, int ntiles,
    const int *__restrict__ tab, FuseParams fp, const int *__restrict__ pairflag, const int *__restrict__ synrow,
    const int *__restrict__ fam_ofs, const int *__restrict__ fam_list)
{
    constexpr int T = 256, SPL = 2, TILE = SPL * T;
    typedef CellPart<NG, 0> P;
    __shared__ __attribute__((aligned(16))) float tiles[T / 64][P::n][kCellwRow];
    const int s = (int)blockIdx.x;                        // source index fastest (see accumulate_grouped_kernel)
    const int tile = blockIdx.y % ntiles, r = blockIdx.y / ntiles;
    const RecvDev &rv = recv[r];
    if (!rv.enabled) return;
    if (tile * TILE >= rv.wlen) return;
    if (!cell_pair(rv, pairflag, s, nrec, r)) return;
    if (synrow && synrow[s] != s) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    float *__restrict__ wt = &tiles[wv][0][0];            // this wave's tile
    const int t_tile0 = rv.wbeg + tile * TILE;
    const int cb = cent_ofs[isrc0], c0 = cent_ofs[isrc0 + s], nc = cent_ofs[isrc0 + s + 1] - c0;
    const GeoRec *__restrict__ rc = recs + ((size_t)(c0 - cb) * nrec + (size_t)r * nc);
    const int *__restrict__ tc = tab + ((size_t)(c0 - cb) * nrec + (size_t)r * nc) * 128;
    const float sd = rv.sd;
    const int w0s = 128 * wv;                             // first sample of the wave inside the workgroup's 512
    const int u0 = w0s + lane;                            // this lane's first output sample (the other: + 64)
    const int pl = 2 * lane;                              // the two tile positions this lane blends
    // halo: lane q owns the 4-sample chunk q / n of component q % n behind the wave's 128 positions
    const int hloc = lane % P::n, hch = lane / P::n, hph = 128 + 4 * hch;
    int hig = P::ig(0);
#pragma unroll
    for (int i = 1; i < P::n; i++) if (hloc == i) hig = P::ig(i);

    f2v ar1[1], ar2[1], dz[1];
    ar1[0] = f2v{ 0.f, 0.f }; ar2[0] = f2v{ 0.f, 0.f }; dz[0] = f2v{ 0.f, 0.f };
    f2u raw[P::n][4];                                     // raw rows of the current group over the wave's 128 positions
    HaloRegs hraw;                                        // ... and over the halo chunk of this lane

    // issue the loads of the group with head fields (pad, ishift) and descriptors ta_ / tb_
#define CELLW_LOAD(head_row0, head_pad, head_ishift, ta_, tb_) do { \
        const int smax_ = (head_ishift) + (((head_pad) >> 8) & 0xff), smin_ = (head_ishift) - (((head_pad) >> 16) & 0xff); \
        const int jb_ = t_tile0 - smax_ - 1; \
        const float *__restrict__ Gg_ = G + (size_t)(head_row0) * (size_t)pitch;     /* descriptors are relative to it (write_tab) */ \
        /* every row of the cell holds the whole tile: rows start at or before it and end behind it */ \
        const bool inside_ = (REC_I(ta_, 52) + jb_ >= 0) && (REC_I(ta_, 53) + jb_ + SPL * (T - 1) <= pitch - 4); \
        if (inside_ && REC_I(ta_, 54)) raw_issue<NG, 0, SPL, true, true>(raw, SPL * tid, jb_, Gg_, pitch, ta_, tb_); \
        else if (inside_) raw_issue<NG, 0, SPL, true>(raw, SPL * tid, jb_, Gg_, pitch, ta_, tb_); \
        else         raw_issue<NG, 0, SPL, false>(raw, SPL * tid, jb_, Gg_, pitch, ta_, tb_); \
        hraw = halo_issue<true, false>(4 * hch <= smax_ - smin_, hig, w0s + hph, jb_, Gg_, pitch, ta_, tb_); \
    } while (0)
#define CELLW_COEF(cc_) do { \
        const float *__restrict__ coef_ = coef_grp + (size_t)((cc_) - c) * 128; \
        _Pragma("unroll") for (int i = 0; i < 2 * P::n; i++) cw[i] = coef_[i]; \
    } while (0)
    // one step: apply centroid `cur` from the tile and (blend_) put centroid nx1 into it
#define CELLW_STEP(blend_) do { \
        const int flags = REC_I(cur, 18); \
        const int ishift = REC_I(cur, 8); \
        const float cl = REC_F(cur, 16), sl = REC_F(cur, 17); \
        const int e = smax - ishift; \
        const int jl = jb + e + u0; \
        const bool tail = (jb + e + w0s + 128) > jend_min; \
        const float bw0 = REC_F(nx1, 4), bw1 = REC_F(nx1, 5), bw2 = REC_F(nx1, 6), bw3 = REC_F(nx1, 7); \
        if (!tail) { \
            const int nojend[P::n] = {}; \
            if (blend_) cellw_step<NG, false, true>(ar1, ar2, dz, wt, e + lane, raw, bw0, bw1, bw2, bw3, pl, hact, hraw, hloc, hph, jl, nojend, flags, cw, cur, sd, cl, sl); \
            else        cellw_step<NG, false, false>(ar1, ar2, dz, wt, e + lane, raw, bw0, bw1, bw2, bw3, pl, hact, hraw, hloc, hph, jl, nojend, flags, cw, cur, sd, cl, sl); \
        } else { \
            int jend[P::n];                     /* end indices: the tail rule only (kept out of the scalar registers otherwise) */ \
            _Pragma("unroll") for (int i = 0; i < P::n; i++) jend[i] = REC_I(tg, 40 + P::ig(i)); \
            if (blend_) cellw_step<NG, true, true>(ar1, ar2, dz, wt, e + lane, raw, bw0, bw1, bw2, bw3, pl, hact, hraw, hloc, hph, jl, jend, flags, cw, cur, sd, cl, sl); \
            else        cellw_step<NG, true, false>(ar1, ar2, dz, wt, e + lane, raw, bw0, bw1, bw2, bw3, pl, hact, hraw, hloc, hph, jl, jend, flags, cw, cur, sd, cl, sl); \
        } \
    } while (0)

    int c = 0;
    int cur = rec_load(rc, 0, nc, lane);                 // record c, lane-distributed; nx1: record c + 1
    int nx1 = rec_load(rc, 1, nc, lane);
    int ta = 0, tb = 0;                                  // load descriptors of record c (none for a source without centroids)
    if (nc > 0) {
        ta = tc[lane]; tb = tc[64 + lane];
        CELLW_LOAD(REC_I(cur, 0), REC_I(cur, 19), REC_I(cur, 8), ta, tb);
    }
    while (c < nc) {
        const int pad0 = REC_I(cur, 19), ishift0 = REC_I(cur, 8);
        const int cend = c + (pad0 & 0xff);
        const int smax = ishift0 + ((pad0 >> 8) & 0xff), smin = ishift0 - ((pad0 >> 16) & 0xff);
        const int jb = t_tile0 - smax - 1;               // tile position q of this wave holds blended trace sample jb + 128 wv + q
        const bool hact = 4 * hch <= smax - smin;        // positions 128 .. 128 + (smax - smin) are read
        const int tg = ta;                               // this group's descriptor row (end indices for the tail rule)
        const int jend_min = min(REC_I(ta, 50), REC_I(ta, 51));
        // descriptors of the NEXT group
        int ta_n = 0, tb_n = 0;
        if (cend < nc) { ta_n = tc[(size_t)cend * 128 + lane]; tb_n = tc[(size_t)cend * 128 + 64 + lane]; }
        const size_t crow = ((size_t)(c0 - cb) * nrec + (size_t)r * nc + c) * 128 + 64 + 40;
        const unsigned clo = __builtin_amdgcn_readfirstlane((unsigned)crow), chi = __builtin_amdgcn_readfirstlane((unsigned)(crow >> 32));
        const float *__restrict__ coef_grp = (const float *)(tab + (((size_t)chi << 32) | clo));
        float cw[2 * P::n];
        {   // ---- first centroid of the group into the tile (behind the reads of the last step: LDS operations of a wave run in order)
            GeoRec gw;
            gw.w[0] = REC_F(cur, 4); gw.w[1] = REC_F(cur, 5); gw.w[2] = REC_F(cur, 6); gw.w[3] = REC_F(cur, 7);
            raw_blend_store<NG, 0, SPL>(raw, wt, kCellwRow, pl, gw.w[0], gw.w[1], gw.w[2], gw.w[3]);
            halo_finish<true>(hact, hraw, wt, kCellwRow, hloc, hph, gw);
        }
        __builtin_amdgcn_s_setprio(0);
        for (int cc = c; cc + 1 < cend; cc++) {
            const int nx2 = rec_load(rc, cc + 2, nc, lane);      // two records ahead: the next one is needed for its weights now
            const bool blend_next = !(REC_I(nx1, 18) & 4);       // (a centroid at the point of its predecessor keeps the tile)
            CELLW_COEF(cc);
            CELLW_STEP(blend_next);
            cur = nx1; nx1 = nx2;
        }
        {
            // ---- last centroid of the group: the raw registers are free, the next group's rows can be on their way
            const int nx2 = rec_load(rc, cend + 1, nc, lane);
            CELLW_COEF(cend - 1);
            // from here to the first blend of the next run this wave waits for memory: it goes first whenever it can issue
            // (accumulate_multi_kernel's rule; cfg4 171.5 -> 168.6 ms)
            __builtin_amdgcn_s_setprio(1);
            if (cend < nc) CELLW_LOAD(REC_I(nx1, 0), REC_I(nx1, 19), REC_I(nx1, 8), ta_n, tb_n);
            CELLW_STEP(false);
            cur = nx1; nx1 = nx2;
        }
        ta = ta_n; tb = tb_n;
        c = cend;
    }
#undef CELLW_LOAD
#undef CELLW_COEF
#undef CELLW_STEP
    // ---- rotation to N/E, signs, store or fused comparator (seismogram.f90:256-283), as accumulate_cell_kernel
    const int nfam = (FUSE && fam_ofs) ? fam_ofs[s + 1] - fam_ofs[s] : 0;
    for (int qf = -1; qf < nfam; qf++) {
        const int js = qf < 0 ? s : fam_list[fam_ofs[s] + qf];
        const int tl = tile * TILE + u0;
        if (!FUSE && tl >= rv.wlen) return;
        float *__restrict__ so = syn + (size_t)js * syn_stride + tl;
        const float a1[SPL] = { ar1[0].x, ar1[0].y }, a2[SPL] = { ar2[0].x, ar2[0].y }, ad[SPL] = { dz[0].x, dz[0].y };
        float mom = 0.f;
        if constexpr (FUSE) mom = fp.moment[fp.isrc0 + js];
        const bool unit = (fp.syn_factor == 1.f);
        for (int k = 0; k < rv.ncomp; k++) {
            const float sg = rv.sign[k];
            float o[SPL];
#pragma unroll
            for (int i = 0; i < SPL; i++) {
                switch (rv.comp[k]) {
                case 1: o[i] = a1[i] * sg; break;
                case 2: o[i] = a2[i] * sg; break;
                case 3: o[i] = ad[i]; break;
                case 4: o[i] = (rv.cl0 * a1[i] - rv.sl0 * a2[i]) * sg; break;
                default: o[i] = (rv.cl0 * a2[i] + rv.sl0 * a1[i]) * sg; break;
                }
            }
            if constexpr (!FUSE) {
#pragma unroll
                for (int i = 0; i < SPL; i++)
                    if (tl + 64 * i < rv.wlen) so[rv.synofs[k] + 64 * i] = o[i];
                continue;
            }
            double acc = 0.0;
            const float *__restrict__ rt = fp.reft + rv.refofs[k] + tl, *__restrict__ tp = fp.tw + rv.refofs[k] + tl;
#pragma unroll
            for (int i = 0; i < SPL; i++) {
                if (tl + 64 * i >= rv.wlen) break;
                const float v = o[i] * mom;
                const float vt = v * tp[64 * i];
                const float a = rt[64 * i];
                switch (fp.method) {
                case 1: { const float d = unit ? (a - vt) : (1.f * a - fp.syn_factor * vt); acc = sq_acc(acc, d); break; }
                case 2: { const float d = unit ? fabsf(a - vt) : fabsf(1.f * a - fp.syn_factor * vt); acc += (double)d; break; }
                case 5: acc += unit ? (double)(a * vt) : (double)(a * 1.f * vt * fp.syn_factor); break;
                default: { const double x = (double)(1.f * a), y = (double)(fp.syn_factor * vt); acc = fmax(acc, sqrt(x * x + y * y)); break; }
                }
            }
            acc = wave_reduce_f64(acc, fp.method == 6);                 // total in lane 63
            if (lane == 63)
                fp.partial[((size_t)js * fp.nmis + rv.slot0 + k) * fp.nparts + tile * (T / 64) + wv] = acc;
        }
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

__global__ __launch_bounds__(256) void misfit_kernel(
    const float *__restrict__ syn, size_t syn_stride, const CompDev *__restrict__ comps,
    const float *__restrict__ reft, const float *__restrict__ tw,
    const float *__restrict__ moment, const float *__restrict__ risetime, MisfitParams mp,
    float *__restrict__ misfit_out, float *__restrict__ proc /* optional [src][stride] processed synthetics */,
    float *__restrict__ fftbuf, float *__restrict__ vt_out /* optional [src][stride] tapered synthetics */,
    const int *__restrict__ spansrc /* per-source strip spans, un-tapered receivers only */, int nrec, int fold_grow,
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
        int s0, s1;
        strip_span(spansrc + ((size_t)(synrow ? synrow[s] : s) * nrec + cd.rec) * kSpanInts, cd.spankind, s0, s1);
        int lo = cd.rf0, hi = cd.rf1;
        if (s1 >= s0) { lo = min(lo, s0 - fold_grow); hi = max(hi, s1 + (fold_grow ? fold_grow + 1 : 0)); }
        i_lo = max(lo - cd.w0, 0); i_hi = min(hi - cd.w0, cd.wlen - 1);
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
    if (to_fft && (mp.fft_mode & 4) && !proc) return;      // the transform kernel takes the plain synthetics itself (workgroup-uniform)
    if (to_fft) {
        const FftPair pr = pairs[(size_t)s * mp.nmis + m];
        frow = fftbuf + pr.fft_ofs;
        for (int i = cd.wlen + threadIdx.x; i < pr.ntrans; i += 256) frow[i] = 0.f;      // zero padding
    }
    for (int i = threadIdx.x; i < cd.wlen; i += 256) {
        const float v = folded_scaled_sample(sy, i, nf, fw, fs, fr, mom);
        const float vt = v * tp[i];               // make_array_tapered, comparator.f90:1173-1184
        if (proc) proc[(size_t)s * syn_stride + cd.synofs + cd.halo + i] = mp.write_tapered == 2 ? vt : v;
        if (frow) { frow[i] = vt; continue; }
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
    const float *__restrict__ filtw, SpecParams sp, float *__restrict__ misfit_out)
{
    __shared__ double red[256];
    const int m = blockIdx.x, s = blockIdx.y;
    const FftPair pr = pairs[(size_t)s * sp.nmis + m];
    const int nb = pr.ntrans / 2 + 1;
    const float2 *__restrict__ row = spec + pr.spec_ofs;
    const float *__restrict__ ra = refamp + pr.specofs;
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
                                                          const CompDev *__restrict__ comps, const float *__restrict__ filtw)
{
    const FftPair pr = pairs[blockIdx.x];
    if (!comps[pr.slot].has_filter) return;
    const int nb = pr.ntrans / 2 + 1;
    float2 *__restrict__ row = spec + pr.spec_ofs;
    const float *__restrict__ fw = filtw + pr.specofs;
    for (int k = threadIdx.x; k < nb; k += 256) {
        float2 z = row[k];
        z.x = z.x * fw[k]; z.y = z.y * fw[k];
        row[k] = z;
    }
}

// time-domain norms on the filtered traces (comparator.f90:810-813,1233-1263): c2r output / ntrans, zeroed
// where the taper is zero (ip_zero_one mask), against the reference processed the same way
__global__ __launch_bounds__(256) void filtered_norm_kernel(
    const float *__restrict__ fftbuf, const CompDev *__restrict__ comps, const FftPair *__restrict__ pairs,
    const float *__restrict__ ref_filt, const float *__restrict__ zmask, SpecParams sp, float *__restrict__ misfit_out,
    float *__restrict__ proc, size_t syn_stride)
{
    __shared__ double red[256];
    const int m = blockIdx.x, s = blockIdx.y;
    const CompDev cd = comps[m];
    if (!cd.has_filter) return;                           // compared by misfit_kernel on the plain tapered arrays
    const FftPair pr = pairs[(size_t)s * sp.nmis + m];
    const float *__restrict__ row = fftbuf + pr.fft_ofs;
    const float *__restrict__ rf = ref_filt + pr.filtofs;
    const float *__restrict__ zm = zmask + cd.refofs;
    const bool unit = (sp.syn_factor == 1.f);
    double acc = 0.0, peak = 0.0;
    for (int i = threadIdx.x; i < cd.wlen; i += 256) {
        float v = row[i] / (float)pr.ntrans;              // normalize result, comparator.f90:1251
        v = v * zm[i];                                    // :1254-1258
        if (proc) proc[(size_t)s * syn_stride + cd.synofs + cd.halo + i] = v;
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

// Transform length of every (trial source, slot) pair of a chunk, as a FRESH reference engine sizes it for this source:
// the synthetic probe is set from the source's own strip (probe_set_array, comparator.f90:222-271: data span = strip
// span, padded to a power of two of at least twice the data length), then probes_adjust_spans (:464-486) gives both
// probes the span allowed_span(union of the two data spans, max of the two minimum lengths) (:1092-1109) -- so
// ntrans = next_power_of_two(max(length of the union, 2 len_ref, 2 len_syn)).  spansrc: per (source, receiver) data spans
// of the horizontal / vertical strips, reduced by geometry_kernel; fold_grow: strip_fold's growth (sparse_trace.f90:379-402).
__global__ void fft_size_kernel(const int *__restrict__ spansrc, const CompDev *__restrict__ comps, int nmis, int nsrc, int nrec,
                                const float *__restrict__ risetime /* of the chunk's sources */, float dt, int *__restrict__ ntr_out)
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
    if (s1 < s0) { s0 = cd.rf0; s1 = cd.rf0; }                       // no centroid reached this strip
    if (fold_grow > 0) { s0 -= fold_grow; s1 += fold_grow + 1; }
    const int len_ref = cd.rf1 - cd.rf0 + 1, len_syn = s1 - s0 + 1;
    const int len_u = max(cd.rf1, s1) - min(cd.rf0, s0) + 1;
    const int minlength = max((int)ceilf((float)len_ref * 2.f), (int)ceilf((float)len_syn * 2.f));
    int need = max(max(len_u, minlength), cd.wlen);
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
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nsrc * nmis) return;
    const int s = idx / nmis, m = idx - s * nmis;
    // tiles this slot's window spans; tile_len == 0: every entry (the buffer was cleared; two kernels with different tilings)
    const int np = tile_len > 0 ? ((comps[m].wlen + tile_len - 1) / tile_len) * waves_per_tile : nparts;
    const double *p = partial + (size_t)idx * nparts;
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
    float syn_factor, int nmis, int maxns, float *__restrict__ partial, const int *__restrict__ spansrc, int nrec, int fold_grow)
{
    __shared__ double red[256];
    const int m = blockIdx.x, s = blockIdx.y;
    const CompDev cd = comps[m];
    int s_lo = 0x7fffffff, s_hi = -0x7fffffff;          // data span of this source's synthetic strip (un-tapered only)
    if (cd.untapered) {
        int s0, s1;
        strip_span(spansrc + ((size_t)s * nrec + cd.rec) * kSpanInts, cd.spankind, s0, s1);
        if (s1 >= s0) { s_lo = s0 - fold_grow; s_hi = s1 + (fold_grow ? fold_grow + 1 : 0); }
    }
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
