// kiwi_common.hpp -- data structures shared by the host library and every device translation unit of the trial-source hot
// path (geometry -> accumulate -> misfit), plus the few device helpers more than one of them uses.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

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

// A centroid's interpolation-coefficient line (written by geometry_kernel's write_tab, read by the accumulate kernels with scalar
// loads): wl = (1 - w) * factor and wr = w * factor of every GF component (sparse_trace.f90:643-647 with the factors of
// seismogram.f90:171-250), each rounded on its own, for the a-th component in APPLICATION order (ng = 10: 1 2 3 9 | 4 5 | 6 7 8 10,
// ng = 8: 1 2 3 | 4 5 | 6 7 8).  One line of kCoefLine floats per record, lines of consecutive centroids consecutive in memory:
// the time steps of a sub-fault share cache lines and DRAM bursts (until round 3 each line sat alone in its record's 512-byte
// descriptor row: 2.6 GB per cfg3 launch from HBM, and a third of geometry_kernel's writes).
constexpr int kCoefLine = 20;      // floats per record in the coefficient array (ng = 8 uses the first 16)
template <int NG> __host__ __device__ constexpr int coef_wl(int a) { return 2 * a; }
template <int NG> __host__ __device__ constexpr int coef_wr(int a) { return 2 * a + 1; }

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

struct EvalParams {
    int bilinear, xus, zus;
    int nrec;
    int isrc0;
    int cellmode;      // groups = runs of centroids in the same 4-node GF cell (cellgroup_kernel marks them), not same-point runs
    int nogaps;        // every trace of the database is stored (kiwi_hip_set_gfdb): no centroid leaves at a missing trace, geometry_kernel
                       // need not look at the spans of its forty rows to find that out
};

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

} // namespace kiwi
