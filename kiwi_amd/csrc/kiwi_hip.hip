// kiwi_hip.hip -- context + C-ABI (include/kiwi_hip.h) of the MI355X engine.
//
// State mirrors what minimizer_engine.f90:78-108 keeps in module variables (database, receivers,
// reference probes, misfit setup, source), plus a batch of discretised trial sources; the
// dirty-flag chain of minimizer_engine.f90:1340-1511 collapses to one "prepared" flag that is
// cleared by every setter the reference routes through dirtyfy_*.
#include "../../include/kiwi_hip.h"
#include "kiwi_host.hpp"
#include "kiwi_host_eikonal.hpp"
#include "kiwi_host_lm.hpp"
#include "kiwi_geometry.hpp"
#include "kiwi_misfit.hpp"
#include "kiwi_accum_api.hpp"

#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <algorithm>
#include <map>
#include <tuple>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>
#include <omp.h>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <future>
#include <mutex>
#include <thread>
#include <limits>

using namespace kiwi;

namespace {

std::string g_init_error;

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };

#define HIPCHECK(expr)                                                                             \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            throw HipError(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
    } while (0)

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    long long *tally = nullptr;
    void alloc(size_t count, long long *t)
    {
        release();
        tally = t;
        if (count == 0) return;
        HIPCHECK(hipMalloc((void **)&p, count * sizeof(T)));
        n = count;
        if (tally) *tally += (long long)(n * sizeof(T));
    }
    void ensure(size_t count, long long *t) { if (count > n) alloc(count, t); }
    void release()
    {
        if (p) { (void)hipFree(p); if (tally) *tally -= (long long)(n * sizeof(T)); }
        p = nullptr; n = 0;
    }
    ~DevBuf() { release(); }
};

struct Receiver {
    GeoCoords origin;                // radians
    float depth = 0.f;
    bool enabled = true;
    int ncomp = 0;
    int comp[kMaxComp] = { 0 };      // signed ids, receiver.f90:35-48
    struct Ref { int first = 0; std::vector<float> data; } ref[kMaxComp];
    Plf taper, filter;
    int float_lo = 0, float_hi = 0;  // floating_shiftrange in samples (receiver.f90:94)
    double azi0 = 0, bazi0 = 0, dist0 = 0;
};

struct EventPair { hipEvent_t a, b; int kind; };

} // namespace

struct kiwi_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    long long dev_bytes = 0;

    // database
    GfMeta gm{};
    bool have_db = false;
    DevBuf<float> G;
    DevBuf<int2> span;
    DevBuf<unsigned char> endz;       // per GF row: the stored trace ends in an exact zero (its repeated end value is 0)

    // setup
    int bilinear = 0, xus = 1, zus = 1;
    float effective_dt = 1.f;                 // minimizer_engine.f90:79
    GeoCoords src_origin; bool have_origin = false; double ref_time = 0;
    std::vector<Receiver> recv;
    int method = KIWI_L2NORM;
    float syn_factor = 1.f;
    // variable-rupture-speed sources (parameterized_source.f90:89-90, source_eikonal.f90:472)
    bool have_crust = false;
    CrustProfile rupture_profile{}, origin_profile{};
    float crustal_thickness_limit = 0.f;
    std::vector<HalfSpace> constraints;

    // prepared (derived) state
    bool prepared = false;
    bool synth_only = false;              // some enabled receiver has no reference yet: synthetics only
    bool any_untapered = false;           // some enabled receiver has references but no taper (comparator.f90:798-800)
    DevBuf<int> spansrc_d;                // per-source strip spans of the current chunk (un-tapered receivers)
    bool want_spansrc = false;            // a diagnostic needs them although no misfit does (shake_impl)
    int nmis = 0, nrec_en = 0;
    int halo = 0;
    size_t syn_stride = 0;
    std::vector<CompDev> comps;
    std::vector<float> norm_h;
    std::vector<float> reft_h;            // tapered references over the windows (host copy)
    DevBuf<RecvDev> recv_d;
    DevBuf<CompDev> comps_d;
    DevBuf<float> reft_d, tw_d, norm_d;
    DevBuf<int> recfirst_d;
    int max_wlen = 0;

    // sources
    int nsrc = 0;
    std::vector<int> cent_ofs;
    float max_risetime = 0.f;
    DevBuf<float> cent_d, moment_d, risetime_d;
    DevBuf<int> centofs_d;
    // per uploaded source: 0 = discretised, 5 = "Empty rupture area", 6 = "position of nucleation point is outside of
    // rupture region" (source_eikonal.f90:286,428).  A failed trial source stays in the batch with no centroids, its
    // misfits, norm factors and global misfit read as zeros (seismosizer.py:703-720: skipped, arrays keep zeros)
    std::vector<int> src_status;
    bool any_failed = false;
    DevBuf<int> status_d;
    // runs of consecutive sources with identical centroid geometry (points and times), single group each: the
    // grouped kernel builds their blended tiles once (env KIWI_HIP_RUNS=0 switches the sharing off)
    // sources whose whole centroid table (points, times AND moment tensors) equals an earlier one's: only moment and/or
    // rise time differ, which act after the synthesis -- evaluated by re-scaling the earlier source's synthetics
    // (minimizer_engine.f90:516-521, source_bilat.f90:206, source_mt_eikonal.f90:234-239).  same_as[s] = that source, or s
    std::vector<int> same_as;
    bool any_same = false;
    DevBuf<int> synrow_d, famofs_d, famlist_d;
    int dedupe_enabled = 1;           // env KIWI_HIP_DEDUPE=0 switches it off
    std::vector<unsigned long long> geo_hash;
    // multi-device context (kiwi_hip_init_multi): the contexts of the other devices, owned by this one; setters are
    // repeated on them, kiwi_hip_misfits_for_params shards the trial list over all of them
    std::vector<kiwi_hip_ctx *> mates;
    int cpu_share = 1;                // contexts that discretise at the same time: divides the discretiser's thread team
    std::vector<unsigned long long> struct_hash;   // per source: number of centroids and boundaries of its centroid groups (accumulate_multi_kernel's grouping)
    std::vector<unsigned char> group_lens;          // per centroid: length of the centroid group that starts there, 0 inside a group (what struct_hash hashes)
    std::vector<int> first_shift;                  // per source: integer shift of its first centroid (groups of four: within 16 samples of each other,
                                                   // so that their groups can share a tile origin)
    std::vector<float> src_ends;                   // per source: position (north, east, depth) of its first and of its last centroid
    DevBuf<int> mate_d, mate4_d;
    int duo = 4;                      // accumulate_multi_kernel: up to this many consecutive sources of equal structure per workgroup
                                      // (4, 2, or 0 = off); env KIWI_HIP_DUO
    std::vector<char> single_group;
    DevBuf<int> runfirst_d;
    int share_runs = 1, max_run = 64;
    // fused comparator (time-domain norms, no fold, nothing kept): env KIWI_HIP_FUSE=0 switches it off
    int fuse_enabled = 1;
    bool fuse_now = false;
    DevBuf<double> fusepart_d;

    // results + workspace
    DevBuf<float> misfit_d, global_d;
    DevBuf<GeoRec> recs_d;
    DevBuf<int> tab_d;                // grouped kernel load descriptors, 128 ints per GeoRec (written at group starts only)
    DevBuf<int> off4_d;               // ... or their compact form, 4 ints per GeoRec (db_simple databases; geometry_kernel's off4)
    DevBuf<float> coef_d;             // interpolation coefficients, kCoefLine floats per GeoRec, consecutive (kiwi_common.hpp coef_wl)
    DevBuf<int> pairflag_d;           // cell mode: per (source of the chunk, receiver) "some centroid misses a trace"
    DevBuf<float> syn_d, proc_d;
    // floating norms
    bool floating = false;
    int max_ns = 1;
    DevBuf<float> refx_d, vt_d, partial_d;
    DevBuf<int> fshift_d;
    int last_isrc0 = 0, last_nsrc = 0, last_chunk0 = 0, last_chunkn = 0;
    std::vector<char> evaluated;      // per uploaded source: misfit_d / global_d hold its results (cleared by set_sources and prepare)
    int last_proc_which = 0;
    int group_threads_env = 0;
    int group_threads = 128;          // workgroup size of the grouped kernel (tile = 4x); env KIWI_HIP_GROUP_THREADS
    int accum_mode = 0;               // 0 grouped (LDS-staged), 1 direct; env KIWI_HIP_ACCUM
    // cell groups (accumulate_cell_kernel: raw node traces fetched once per run of centroids in the same GF cell):
    // -1 decided per batch -- sources whose centroids are mostly different points --, 0 off, 1 on; env KIWI_HIP_CELL
    int cell_mode = -1;
    int cell_wave = 1;                // 1: accumulate_cellw_kernel (a tile per wave, no barriers); 0: accumulate_cell_kernel; env KIWI_HIP_CELL_WAVE
    int arith = KIWI_ARITH_EXACT;     // arithmetic contract of the accumulate kernels (kiwi_hip_set_arithmetic; env KIWI_HIP_ARITH=exact|fused)
    double points_per_centroid = 0.0; // of the uploaded batch: distinct consecutive points / centroids
    int keep_which = 0;               // kiwi_hip_set_keep_synthetics
    int proc_chunk0 = 0, proc_chunkn = 0, proc_which_held = 0;   // what proc_d currently holds
    size_t chunk_bytes_limit = (size_t)16 << 30;      // workspace per launch; the device has 288 GB

    // spectral / filtered comparator (hipFFT).  The transform length belongs to the (trial source, slot) pair
    // (fft_size_kernel); the reference-side data that depend on it -- amplitude spectrum, filter weights per bin,
    // filtered reference, norm factor -- are kept per (slot, ntrans) VARIANT and made when a length first occurs.
    bool fft_needed = false, fft_ready = false, any_filter = false;
    int fft_cap = 0;                        // sources per chunk the FFT buffers are sized for
    size_t fft_floats_per_src = 0, spec_cplx_per_src = 0;      // capacity per source: every slot at its longest transform
    struct FftVariant { int ntrans, specofs, filtofs; float norm; };
    std::map<std::pair<int, int>, FftVariant> variants;         // (slot, ntrans) ->
    std::vector<float> refamp_h, filtw_h, reffilt_h;            // host mirrors of the variant tables (re-uploaded when they grow)
    std::vector<int> slot_has_filter;
    DevBuf<int> spanbuf_d, ntr_d;
    DevBuf<float> fft_d, refamp_d, filtw_d, reffilt_d, zmask_d, normsrc_d;
    DevBuf<int> lmax_d;                                          // [chunk source][receiver]: last centroid that reaches the rotated add (geometry_kernel, databases with gaps)
    bool db_gaps = false;                                        // some trace of the database is not stored
    bool db_simple = false;           // every node's components start at the same sample and every row ends in an exact zero: the load
                                      // descriptors of a group follow from four numbers (compact descriptors, geometry_kernel's off4)
    int compact = 1;                  // use them where the database allows it; env KIWI_HIP_COMPACT=0: the 512-byte rows throughout
    DevBuf<int> synspan_d;                                       // data spans of the synthetic probes of the chunk's un-tapered slots (synspan_kernel)
    DevBuf<float> refpair_d, reffiltpair_d;                      // un-tapered slots: reference spectrum / filtered reference per PAIR (SpecParams)
    bool untapered_fft = false;                                  // some un-tapered slot goes through the transforms
    DevBuf<float2> spec_d;
    DevBuf<FftPair> pairs_d;
    int *ntr_pin = nullptr; size_t ntr_pin_n = 0;               // pinned staging: transform lengths down, pair table up
    int *mate_pin = nullptr; size_t mate_pin_n = 0;             // pinned staging of the mate flags of a chunk (groups of four, then pairs)
    hipEvent_t mate_event = nullptr;                            // their upload has been read
    FftPair *pairs_pin = nullptr; size_t pairs_pin_n = 0;
    std::vector<FftPair> last_pairs;                             // pair table of the last chunk (diagnostic getters)
    std::vector<float> norm_src_h;                               // norm factors per (uploaded source, slot)
    struct FftBucket { int ntrans; long long count, fft_base, spec_base; };
    std::vector<FftBucket> buckets;                              // of the chunk being evaluated
    hipEvent_t size_event = nullptr;
    std::map<std::tuple<int, int, int>, hipfftHandle> plans;     // (ntrans, batch, type) -> plan
    DevBuf<float2> fused_tab[kFusedFftMaxLog2 + 1];              // twiddle tables of spec_fft_norm_kernel, by log2(ntrans)
    bool fused_fft = true;                                       // KIWI_HIP_FUSED_FFT=0: amplitude spectra through hipFFT
    bool fused_fft_attr = false;

    std::vector<EventPair> events;
    std::vector<hipEvent_t> event_pool;

    hipEvent_t get_event()
    {
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e; HIPCHECK(hipEventCreate(&e)); return e;
    }
};

namespace {

int fail(kiwi_hip_ctx *ctx, const std::string &msg)
{
    if (ctx) ctx->err = msg; else g_init_error = msg;
    return 1;
}

#define GUARD_BEGIN try {
// entry points of a context that touch its device: the context's device is the calling thread's current one from here on
// (a multi-device context forwards its setters to the other devices, kiwi_hip_init_multi: the caller's current device is
// not something an entry point may rely on)
#define GUARD_BEGIN_DEV(ctx) try { HIPCHECK(hipSetDevice((ctx)->device));
#define GUARD_END(ctx)                                                                             \
    } catch (const std::exception &e) { return fail(ctx, e.what()); }                              \
      catch (...) { return fail(ctx, "unknown error"); }

// repeat a setter on the other devices of a multi-device context (kiwi_hip_init_multi)
// (a mate's setter may make its device current for the calling thread: the owner's device is current again afterwards)
template <class F> static int forward(kiwi_hip_ctx *c, F &&f)
{
    if (c->mates.empty()) return 0;
    int rc = 0;
    for (kiwi_hip_ctx *m : c->mates)
        if ((rc = f(m))) { c->err = "device " + std::to_string(m->device) + ": " + m->err; break; }
    if (hipSetDevice(c->device) != hipSuccess && !rc) { c->err = "hipSetDevice failed"; rc = 1; }
    return rc;
}

int component_id(char ch)     // receiver.f90:294-307
{
    static const char names[] = "wsulc?ardne";
    for (int i = 0; i < 11; i++) if (names[i] == ch) return i - 5;
    return 0;
}

void update_receiver_geometry(kiwi_hip_ctx *c)
{
    if (!c->have_origin) return;
    for (auto &r : c->recv) {
        azibazi(c->src_origin, r.origin, r.azi0, r.bazi0);                 // seismogram.f90:99
        r.dist0 = distance_accurate50m(c->src_origin, r.origin);          // seismogram.f90:100
    }
}

int fold_halfwidth(float risetime, float dt)
{
    if (!(risetime > 0.f)) return 0;
    const int n = 1 + 2 * (int)std::round(0.5f * risetime / dt);          // receiver.f90:872
    return (n - 1) / 2;
}

// derive windows, tapered references, norm factors and device tables
// Natural spans of the synthetic strips over all uploaded sources (seismogram.f90:102-130 with
// sparse_trace.f90:648-668): per receiver [lo_h, hi_h, lo_d, hi_d], reduced on the device by geometry_kernel.
// Needs recv_d (geometry part) uploaded.
void natural_spans(kiwi_hip_ctx *c, std::vector<int> &sb)
{
    const int nrec = (int)c->recv.size();
    sb.assign((size_t)nrec * 4, 0);
    for (int r = 0; r < nrec; r++) { sb[4 * r] = sb[4 * r + 2] = 0x7fffffff; sb[4 * r + 1] = sb[4 * r + 3] = -0x7fffffff; }
    c->spanbuf_d.ensure(sb.size(), &c->dev_bytes);
    HIPCHECK(hipMemcpyAsync(c->spanbuf_d.p, sb.data(), sb.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    for (int s0 = 0; s0 < c->nsrc; s0 += 32768) {
        const int n = std::min(32768, c->nsrc - s0);
        int maxnc = 0;
        for (int s = s0; s < s0 + n; s++) maxnc = std::max(maxnc, c->cent_ofs[s + 1] - c->cent_ofs[s]);
        if (maxnc == 0) continue;
        EvalParams ep{ c->bilinear, c->xus, c->zus, nrec, s0, 0, c->db_gaps ? 0 : 1 };
        dim3 grid((unsigned)((maxnc * nrec + 255) / 256), (unsigned)n);
        int *lmax = nullptr;
        if (c->db_gaps) {      // (see geometry_kernel's span reduction: which partly added centroids count follows the centroids' order)
            c->lmax_d.ensure((size_t)n * nrec, &c->dev_bytes);
            HIPCHECK(hipMemsetAsync(c->lmax_d.p, 0xff, (size_t)n * nrec * sizeof(int), c->stream));
            lmax = c->lmax_d.p;
            hipLaunchKernelGGL(geometry_kernel, grid, dim3(256), 0, c->stream, c->cent_d.p, c->centofs_d.p, ep, c->gm,
                               c->span.p, c->recv_d.p, (GeoRec *)nullptr, (int *)nullptr, (float *)nullptr, (int *)nullptr, (int *)nullptr, (int *)nullptr, c->endz.p, (const int *)nullptr,
                               lmax, (const int *)nullptr, (int4 *)nullptr);
        }
        hipLaunchKernelGGL(geometry_kernel, grid, dim3(256), 0, c->stream, c->cent_d.p, c->centofs_d.p, ep, c->gm,
                           c->span.p, c->recv_d.p, (GeoRec *)nullptr, (int *)nullptr, (float *)nullptr, c->spanbuf_d.p, (int *)nullptr, (int *)nullptr, c->endz.p, (const int *)nullptr,
                           (int *)nullptr, (const int *)lmax, (int4 *)nullptr);
    }
    HIPCHECK(hipMemcpyAsync(sb.data(), c->spanbuf_d.p, sb.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
}

void prepare(kiwi_hip_ctx *c)
{
    if (c->prepared) return;
    c->proc_which_held = 0;
    c->evaluated.assign((size_t)c->nsrc, 0);
    if (!c->have_db) throw std::runtime_error("no database set");
    if (!c->have_origin) throw std::runtime_error("no source location set");
    if (c->recv.empty()) throw std::runtime_error("no receivers set");
    if (c->method < KIWI_L2NORM || c->method > KIWI_FLOATING_L1NORM) throw std::runtime_error("unknown misfit method");
    c->floating = (c->method == KIWI_FLOATING_L2NORM || c->method == KIWI_FLOATING_L1NORM);
    // the norm evaluated inside a floating norm (receiver.f90:452-458)
    const int eval_method = c->method == KIWI_FLOATING_L2NORM ? KIWI_L2NORM : (c->method == KIWI_FLOATING_L1NORM ? KIWI_L1NORM : c->method);
    c->max_ns = 1;
    std::vector<float> refx;
    c->fft_ready = false;
    const float dt = c->gm.dt;
    const int hs = fold_halfwidth(c->max_risetime, dt);
    c->halo = hs > 0 ? hs + 2 : 0;
    const int nrec = (int)c->recv.size();
    std::vector<RecvDev> rd(nrec);
    c->comps.clear();
    c->norm_h.clear();
    // Before any reference or taper is set the engine can still synthesise (output_seismograms is how a
    // "synthetic reference" is made, minimizer.f90:1296-1380): then every window is the natural span of the
    // synthetic strips and misfits are not available.
    c->synth_only = false;
    c->any_untapered = false;
    for (auto &r : c->recv) {
        if (!r.enabled || r.ncomp == 0) continue;
        if (!r.taper.defined()) c->any_untapered = true;
        for (int k = 0; k < r.ncomp; k++) if (r.ref[k].data.empty()) c->synth_only = true;
    }
    if (c->synth_only) c->any_untapered = false;
    std::vector<int> nat;
    if (c->synth_only || c->any_untapered) {
        if (c->nsrc == 0) throw std::runtime_error("no source set");
        if (c->synth_only) c->floating = false;
        for (int ir = 0; ir < nrec; ir++) {                 // geometry part of the receiver records only
            Receiver &r = c->recv[ir];
            RecvDev &d = rd[ir];
            std::memset(&d, 0, sizeof(d));
            d.azi0 = r.azi0; d.bazi0 = r.bazi0; d.dist0 = r.dist0; d.depth = r.depth;
            d.enabled = r.enabled && r.ncomp > 0;
            for (int k = 0; k < r.ncomp; k++) { if (std::abs(r.comp[k]) == 3) d.has_d = 1; else d.need_h = 1; }
        }
        c->recv_d.ensure(rd.size(), &c->dev_bytes);
        HIPCHECK(hipMemcpyAsync(c->recv_d.p, rd.data(), rd.size() * sizeof(RecvDev), hipMemcpyHostToDevice, c->stream));
        natural_spans(c, nat);
    }
    std::vector<float> reft, tw;
    std::vector<int> recfirst;
    size_t synofs = 0;
    c->max_wlen = 0;
    for (int ir = 0; ir < nrec; ir++) {
        Receiver &r = c->recv[ir];
        RecvDev &d = rd[ir];
        std::memset(&d, 0, sizeof(d));
        d.azi0 = r.azi0; d.bazi0 = r.bazi0; d.dist0 = r.dist0;
        d.depth = r.depth;
        d.cl0 = (float)std::cos(r.bazi0 + (double)kPi);                    // seismogram.f90:270-271
        d.sl0 = (float)std::sin(r.bazi0 + (double)kPi);
        d.enabled = r.enabled && r.ncomp > 0;
        d.ncomp = r.ncomp;
        d.sd = 0.f;
        for (int k = 0; k < r.ncomp; k++) {
            d.comp[k] = std::abs(r.comp[k]);
            d.sign[k] = r.comp[k] < 0 ? -1.f : 1.f;                        // receiver.f90:331-351
            if (d.comp[k] == 3) { d.has_d = 1; d.sd = d.sign[k]; } else d.need_h = 1;
        }
        if (!d.enabled) continue;
        int w[2];
        const bool untapered = !c->synth_only && !r.taper.defined();
        if (c->synth_only || untapered) {
            // no taper: the window must hold every strip of the uploaded sources and, for the comparator, the
            // references at every floating shift; the norms themselves are restricted per source in the kernels
            w[0] = std::min(nat[4 * ir], nat[4 * ir + 2]);
            w[1] = std::max(nat[4 * ir + 1], nat[4 * ir + 3]);
            if (w[1] < w[0]) { w[0] = 0; w[1] = 0; }                        // no centroid reached this receiver
            else if (hs > 0) { w[0] -= hs; w[1] += hs + 1; }                // strip_fold grows the strip, sparse_trace.f90:379-402
            if (untapered) {
                const int flo = c->floating ? r.float_lo : 0, fhi = c->floating ? r.float_hi : 0;
                bool first = (nat[4 * ir + 1] < nat[4 * ir] && nat[4 * ir + 3] < nat[4 * ir + 2]);
                for (int k = 0; k < r.ncomp; k++) {
                    const int f0 = r.ref[k].first, f1 = f0 + (int)r.ref[k].data.size() - 1;
                    if (first) { w[0] = f0 + flo; w[1] = f1 + fhi; first = false; }
                    w[0] = std::min(w[0], f0 + flo); w[1] = std::max(w[1], f1 + fhi);
                }
            }
        } else {
            discrete_plf_span(r.taper, dt, w);                             // comparator.f90:1157-1169
            if (w[1] < w[0]) throw std::runtime_error("receiver " + std::to_string(ir + 1) + ": empty taper span");
        }
        const int wlen = w[1] - w[0] + 1;
        d.wbeg = w[0] - c->halo;
        d.wlen = wlen + 2 * c->halo;
        c->max_wlen = std::max(c->max_wlen, d.wlen);
        // taper weights: plf_taper_array applied to ones (piecewise_linear_function.f90:195-237)
        std::vector<float> tww(wlen, 1.f);
        if (!c->synth_only && !untapered) plf_taper_array(r.taper, tww.data(), w[0], w[1], dt, IP_COS);
        recfirst.push_back((int)c->comps.size());
        d.slot0 = (int)c->comps.size();
        for (int k = 0; k < r.ncomp; k++) {
            static const Receiver::Ref no_ref = { 0, std::vector<float>(1, 0.f) };
            const auto &rf = c->synth_only ? no_ref : r.ref[k];
            CompDev cd;
            std::memset(&cd, 0, sizeof(cd));
            cd.synofs = (int)synofs; cd.halo = c->halo; cd.w0 = w[0]; cd.wlen = wlen;
            cd.refofs = (int)reft.size(); cd.rec = ir;
            cd.fl_lo = 0; cd.fl_ns = 1; cd.refxofs = 0;
            cd.untapered = untapered ? 1 : 0;
            cd.vertical = std::abs(r.comp[k]) == 3 ? 1 : 0;
            { const int a = std::abs(r.comp[k]); cd.spankind = a == 3 ? 3 : (a == 1 ? 0 : (a == 2 ? 1 : 2)); }   // receiver.f90:35-48: 1 away 2 right 3 down 4 north 5 east
            d.synofs[k] = (int)synofs;
            d.refofs[k] = cd.refofs;
            synofs += ((size_t)d.wlen + 3) / 4 * 4;
            // reference probe contents over the window: zeros before the data, last value repeated
            // after it (probe_set_array, comparator.f90:259-265), then tapered (:1173-1184)
            const int f0 = rf.first, f1 = rf.first + (int)rf.data.size() - 1;
            cd.rf0 = f0; cd.rf1 = f1;
            double sum = 0.0, pk = 0.0;
            for (int t = w[0]; t <= w[1]; t++) {
                float v = 0.f;
                if (t >= f0) v = rf.data[std::min(t, f1) - f0] * 1.f;
                if (t >= f0) v = v * tww[t - w[0]];                         // taper acts from dataspan(1) on
                reft.push_back(v);
                if (untapered && (t < f0 || t > f1)) continue;             // probe_norm without taper: the data span only, :843-845
                switch (eval_method) {                                     // probe_norm, comparator.f90:669-697
                case KIWI_L2NORM: sum += (double)v * (double)v; break;
                case KIWI_L1NORM: sum += (double)std::fabs(v); break;
                case KIWI_SCALAR_PRODUCT: sum += (double)(v * v); break;
                default: pk = std::max(pk, (double)std::fabs(v)); break;
                }
            }
            tw.insert(tw.end(), tww.begin(), tww.end());
            float nf;
            switch (eval_method) {
            case KIWI_L2NORM: nf = 1.f * (float)std::sqrt((double)dt * sum); break;
            case KIWI_L1NORM: nf = 1.f * (float)((double)dt * sum); break;
            case KIWI_SCALAR_PRODUCT: nf = (1.f * 1.f) * (float)sum; break;
            default: nf = 1.f * (float)pk; break;
            }
            if (c->floating) {
                // receiver_calculate_floating_misfits, receiver.f90:439-510: the reference DATA are moved by every
                // integer shift of the range (probe_shift, comparator.f90:273-288), the taper stays; the norm factor
                // is the mean over the shifts of probe_norm of the shifted reference (:502)
                const int lo = r.float_lo, hi = r.float_hi, ns = hi - lo + 1;
                if (ns < 1) throw std::runtime_error("receiver " + std::to_string(ir + 1) + ": empty floating shift range");
                if (ns > kMaxFloatShifts) throw std::runtime_error("receiver " + std::to_string(ir + 1) + ": floating shift range too long");
                c->max_ns = std::max(c->max_ns, ns);
                cd.fl_lo = lo; cd.fl_ns = ns; cd.refxofs = (int)refx.size();
                // un-tapered reference at t = w0 - hi ... w1 - lo (zeros before the data, end value after)
                for (int t = w[0] - hi; t <= w[1] - lo; t++) refx.push_back(t >= f0 ? rf.data[std::min(t, f1) - f0] * 1.f : 0.f);
                float nsum = 0.f;
                for (int q = 0; q < ns; q++) {
                    const int sh = lo + q;
                    double acc = 0.0;
                    for (int t = w[0]; t <= w[1]; t++) {
                        const int ts = t - sh;
                        if (untapered && (ts < f0 || ts > f1)) continue;
                        float v = 0.f;
                        if (ts >= f0) v = (rf.data[std::min(ts, f1) - f0] * 1.f) * tww[t - w[0]];
                        acc += eval_method == KIWI_L2NORM ? (double)v * (double)v : (double)std::fabs(v);
                    }
                    const float nq = eval_method == KIWI_L2NORM ? 1.f * (float)std::sqrt((double)dt * acc) : 1.f * (float)((double)dt * acc);
                    nsum = nsum + nq;
                }
                nf = nsum / (float)ns;
            }
            c->norm_h.push_back(nf);
            c->comps.push_back(cd);
        }
    }
    recfirst.push_back((int)c->comps.size());
    c->reft_h = reft;
    c->nmis = (int)c->comps.size();
    c->nrec_en = (int)recfirst.size() - 1;
    c->syn_stride = synofs;
    if (c->nmis == 0) throw std::runtime_error("no enabled receiver components");

    c->recv_d.ensure(rd.size(), &c->dev_bytes);
    c->comps_d.ensure(c->comps.size(), &c->dev_bytes);
    c->reft_d.ensure(reft.size(), &c->dev_bytes);
    c->tw_d.ensure(tw.size(), &c->dev_bytes);
    c->norm_d.ensure(c->norm_h.size(), &c->dev_bytes);
    c->recfirst_d.ensure(recfirst.size(), &c->dev_bytes);
    HIPCHECK(hipMemcpyAsync(c->recv_d.p, rd.data(), rd.size() * sizeof(RecvDev), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(c->comps_d.p, c->comps.data(), c->comps.size() * sizeof(CompDev), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(c->reft_d.p, reft.data(), reft.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(c->tw_d.p, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(c->norm_d.p, c->norm_h.data(), c->norm_h.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(c->recfirst_d.p, recfirst.data(), recfirst.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));      // host vectors go out of scope
    if (c->nsrc > 0) {
        c->misfit_d.ensure((size_t)c->nsrc * c->nmis, &c->dev_bytes);
        c->global_d.ensure((size_t)c->nsrc, &c->dev_bytes);
    }
    if (c->floating) {
        c->refx_d.ensure(std::max<size_t>(refx.size(), 1), &c->dev_bytes);
        HIPCHECK(hipMemcpy(c->refx_d.p, refx.data(), refx.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    c->any_filter = false;
    for (auto &r : c->recv) if (r.enabled && r.ncomp > 0 && r.filter.defined()) c->any_filter = true;
    if (c->floating && c->any_filter) throw std::runtime_error("floating norms with a misfit filter are not supported by the device comparator");
    c->fft_needed = !c->synth_only && (c->method == KIWI_AMPSPEC_L2NORM || c->method == KIWI_AMPSPEC_L1NORM || c->any_filter);
    // Spectral norms / filters on a receiver without a taper (comparator.f90:861-886 over the whole padded probes): the probes' common
    // span follows the PAIR (fresh-engine semantics as for the un-tapered time-domain norms), so the reference side is transformed per
    // pair as well -- through the library transforms, see run_chunk
    c->untapered_fft = c->fft_needed && c->any_untapered;
    c->prepared = true;
}

void record(kiwi_hip_ctx *c, int kind, hipEvent_t &a)
{
    a = c->get_event();
    HIPCHECK(hipEventRecord(a, c->stream));
    (void)kind;
}


#define FFTCHECK(expr)                                                                             \
    do {                                                                                           \
        hipfftResult r_ = (expr);                                                                  \
        if (r_ != HIPFFT_SUCCESS) throw HipError(std::string(#expr) + ": hipfft error " + std::to_string((int)r_)); \
    } while (0)

hipfftHandle get_plan(kiwi_hip_ctx *c, int ntrans, int batch, hipfftType type)
{
    auto key = std::make_tuple(ntrans, batch, (int)type);
    auto it = c->plans.find(key);
    if (it != c->plans.end()) return it->second;
    hipfftHandle h;
    int n[1] = { ntrans };
    FFTCHECK(hipfftPlanMany(&h, 1, n, nullptr, 1, ntrans, nullptr, 1, ntrans / 2 + 1, type, batch));
    FFTCHECK(hipfftSetStream(h, c->stream));
    c->plans[key] = h;
    return h;
}

// Batched transforms of `count` contiguous rows, issued as plans of power-of-two batch sizes: the bucket sizes change
// from chunk to chunk, the set of plans stays small (log2 of the largest bucket per length).
void fft_rows(kiwi_hip_ctx *c, int ntrans, long long count, long long fft_base, long long spec_base, bool forward)
{
    const long long nb = ntrans / 2 + 1;
    long long done = 0;
    for (int bit = 30; bit >= 0; bit--) {
        const long long b = 1ll << bit;
        if (!(count & b)) continue;
        if (forward)
            FFTCHECK(hipfftExecR2C(get_plan(c, ntrans, (int)b, HIPFFT_R2C), c->fft_d.p + fft_base + done * ntrans,
                                   (hipfftComplex *)(c->spec_d.p + spec_base + done * nb)));
        else
            FFTCHECK(hipfftExecC2R(get_plan(c, ntrans, (int)b, HIPFFT_C2R), (hipfftComplex *)(c->spec_d.p + spec_base + done * nb),
                                   c->fft_d.p + fft_base + done * ntrans));
        done += b;
    }
}

void fft_buckets(kiwi_hip_ctx *c, bool forward, int lds_lo = 1, int lds_hi = 0)
{
    for (auto &b : c->buckets)
        if (b.ntrans < lds_lo || b.ntrans > lds_hi) fft_rows(c, b.ntrans, b.count, b.fft_base, b.spec_base, forward);      // (the other lengths: in-LDS kernels)
}

int next_pow2(int n) { int m = 1; while (m < n) m *= 2; return m; }      // comparator.f90:1111-1118 (integer form)

// the lengths whose pairs go through the in-LDS transforms in this context (SpecParams::lds_lo / lds_hi): decided per PAIR
void lds_fft_range(const kiwi_hip_ctx *c, int &lo, int &hi)
{
    if (c->fused_fft && !c->untapered_fft) { lo = 1 << kFusedFftMinLog2; hi = 1 << kFusedFftMaxLog2; }
    else { lo = 1; hi = 0; }          // none: library transforms throughout (KIWI_HIP_FUSED_FFT=0; un-tapered slots, see run_chunk)
}

// twiddle table of one length (layout: fused_fft_table_size / spec_fft_norm_kernel), made in double precision
void fused_fft_table(kiwi_hip_ctx *c, int ntrans)
{
    int lg = 0;
    while ((1 << lg) < ntrans) lg++;
    if (c->fused_tab[lg].p) return;
    const int M = ntrans / 2;
    std::vector<float2> t;
    t.reserve(fused_fft_table_size(ntrans));
    const double twopi = 6.283185307179586476925286766559;
    for (int len = M; len >= 4; len >>= 2) {
        const int q = len >> 2;
        for (int r = 1; r <= 3; r++)
            for (int pos = 0; pos < q; pos++) {
                const double a = -twopi * (double)r * (double)pos / (double)len;
                t.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
            }
    }
    for (int k = 0; k <= M; k++) {
        const double a = -twopi * (double)k / (double)ntrans;
        t.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
    }
    c->fused_tab[lg].alloc(t.size(), &c->dev_bytes);
    HIPCHECK(hipMemcpy(c->fused_tab[lg].p, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice));
}

FusedFftTables fused_fft_tables(kiwi_hip_ctx *c)
{
    FusedFftTables ft;
    for (int i = 0; i <= kFusedFftMaxLog2; i++) ft.tab[i] = c->fused_tab[i].p;
    if (!c->fused_fft_attr) {          // more than 64 KB of dynamic LDS per workgroup has to be asked for
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&spec_fft_norm_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 << kFusedFftMaxLog2));
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&spec_fft_norm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 << kFusedFftMaxLog2));
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&spec_fft_norm_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 << kFusedFftMaxLog2));
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&spec_fft_filter_norm_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 << kFusedFftMaxLog2));
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&spec_fft_filter_norm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 << kFusedFftMaxLog2));
        c->fused_fft_attr = true;
    }
    return ft;
}

__global__ void ref_amp_kernel(const float2 *__restrict__ spec, const FftPair *__restrict__ pairs,
                               const float *__restrict__ filtw, float *__restrict__ refamp, int lds_lo, int lds_hi)
{
    const FftPair pr = pairs[blockIdx.x];
    if (pr.ntrans >= lds_lo && pr.ntrans <= lds_hi) return;
    const int nb = pr.ntrans / 2 + 1;
    const float2 *row = spec + pr.spec_ofs;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        const float2 z = row[k];
        refamp[pr.specofs + k] = hypotf(z.x, z.y) * filtw[pr.specofs + k];
    }
}

__global__ void ref_filt_kernel(const float *__restrict__ fftbuf, const FftPair *__restrict__ pairs, const CompDev *__restrict__ comps,
                                const float *__restrict__ zmask, float *__restrict__ ref_filt, int lds_lo, int lds_hi)
{
    const FftPair pr = pairs[blockIdx.x];
    if (pr.ntrans >= lds_lo && pr.ntrans <= lds_hi) return;
    const CompDev cd = comps[pr.slot];
    if (cd.untapered) return;                  // (the window of an un-tapered slot may be longer than the row; its filtered reference is the pair's)
    const float *row = fftbuf + pr.fft_ofs;
    for (int i = threadIdx.x; i < cd.wlen; i += blockDim.x)
        ref_filt[pr.filtofs + i] = (row[i] / (float)pr.ntrans) * zmask[cd.refofs + i];
}

template <class T> void pin_ensure(T *&p, size_t &have, size_t want)
{
    if (want <= have) return;
    (void)hipDeviceSynchronize();                        // a copy from / to the old block may still be in flight (rare: growth only)
    if (p) (void)hipHostFree(p);
    p = nullptr; have = 0;
    HIPCHECK(hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault));
    have = want;
}

// Capacity of the FFT buffers and the slot-level tables.  The longest transform a slot can need in this batch follows
// from the natural spans of the synthetic strips over ALL uploaded sources (a source's own span lies inside); the
// lengths actually used are per source (fft_size_kernel).
void prepare_fft(kiwi_hip_ctx *c, const std::vector<float> &reft_host)
{
    const float dt = c->gm.dt;
    std::vector<int> sb;
    natural_spans(c, sb);
    const int hs = fold_halfwidth(c->max_risetime, dt);
    c->fft_floats_per_src = 0; c->spec_cplx_per_src = 0;
    c->slot_has_filter.assign(c->comps.size(), 0);
    for (size_t m = 0; m < c->comps.size(); m++) {
        CompDev &cd = c->comps[m];
        const int f0 = cd.rf0, f1 = cd.rf1;
        int s0 = sb[4 * cd.rec + (cd.vertical ? 2 : 0)], s1 = sb[4 * cd.rec + (cd.vertical ? 3 : 1)];
        if (s1 < s0) { s0 = 0; s1 = 0; }                            // no centroid contributed: the reference's empty strip is one zero at sample 0
        else if (hs > 0) { s0 -= hs; s1 += hs + 1; }                // strip_fold grows the strip
        const int len_ref = f1 - f0 + 1, len_syn = s1 - s0 + 1;
        const int len_u = std::max(f1, s1) - std::min(f0, s0) + 1;
        const int minlength = std::max((int)std::ceil(len_ref * 2.f), (int)std::ceil(len_syn * 2.f));
        cd.ntrans_max = next_pow2(std::max(std::max(len_u, minlength), cd.wlen));
        cd.has_filter = c->recv[cd.rec].filter.defined() ? 1 : 0;
        c->slot_has_filter[m] = cd.has_filter;
        c->fft_floats_per_src += (size_t)cd.ntrans_max;
        c->spec_cplx_per_src += (size_t)cd.ntrans_max / 2 + 1;
    }
    const size_t per_src = c->fft_floats_per_src * 4 + c->spec_cplx_per_src * 8;
    c->fft_cap = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(c->nsrc, 1), c->chunk_bytes_limit / per_src));
    c->fft_d.ensure((size_t)c->fft_cap * c->fft_floats_per_src, &c->dev_bytes);
    c->spec_d.ensure((size_t)c->fft_cap * c->spec_cplx_per_src, &c->dev_bytes);
    // zero/one mask of the taper over the window (comparator.f90:1254-1258)
    std::vector<float> zm(reft_host.size(), 1.f);
    for (size_t m = 0; m < c->comps.size(); m++) {
        const CompDev &cd = c->comps[m];
        if (c->recv[cd.rec].taper.defined()) plf_taper_array(c->recv[cd.rec].taper, zm.data() + cd.refofs, cd.w0, cd.w0 + cd.wlen - 1, dt, IP_ZERO_ONE);
    }
    c->zmask_d.ensure(zm.size(), &c->dev_bytes);
    HIPCHECK(hipMemcpyAsync(c->zmask_d.p, zm.data(), zm.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(c->comps_d.p, c->comps.data(), c->comps.size() * sizeof(CompDev), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    c->variants.clear();
    c->refamp_h.clear(); c->filtw_h.clear(); c->reffilt_h.clear();
    c->norm_src_h.assign((size_t)c->nsrc * c->nmis, 0.f);
    c->normsrc_d.ensure(std::max<size_t>(1, (size_t)c->nsrc * c->nmis), &c->dev_bytes);
    if (!c->size_event) HIPCHECK(hipEventCreateWithFlags(&c->size_event, hipEventDisableTiming));
    c->fft_ready = true;
}

// The reference probes of the (slot, ntrans) pairs in `want` that have not been seen yet go through the device
// pipeline (r2c -> |.| x filter; filter -> c2r / ntrans -> zero outside the taper), their norm factors (probe_norm,
// comparator.f90:954-996) are taken on the host.  Uses the FFT buffers as scratch: stream-ordered before the chunk's rows.
void make_variants(kiwi_hip_ctx *c, const std::vector<std::pair<int, int>> &want)
{
    std::vector<std::pair<int, int>> fresh;
    for (auto &w : want) if (!c->variants.count(w)) { bool dup = false; for (auto &f : fresh) if (f == w) dup = true; if (!dup) fresh.push_back(w); }
    if (fresh.empty()) return;
    const float dt = c->gm.dt;
    const bool spectral = (c->method == KIWI_AMPSPEC_L2NORM || c->method == KIWI_AMPSPEC_L1NORM);
    std::vector<FftPair> prs;
    long long fofs = 0, sofs = 0;
    std::sort(fresh.begin(), fresh.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.second != b.second ? a.second < b.second : a.first < b.first; });
    for (auto &f : fresh) {
        const CompDev &cd = c->comps[f.first];
        const int ntr = f.second, nb = ntr / 2 + 1;
        FftPair pr;
        pr.fft_ofs = fofs; pr.spec_ofs = sofs; pr.ntrans = ntr; pr.slot = f.first;
        pr.specofs = (int)c->refamp_h.size(); pr.filtofs = (int)c->reffilt_h.size();
        c->refamp_h.resize(c->refamp_h.size() + nb, 0.f);
        c->filtw_h.resize(c->filtw_h.size() + nb, 1.f);
        c->reffilt_h.resize(c->reffilt_h.size() + cd.wlen, 0.f);
        // filter weights per bin: plf_taper_array on ones, abscissa j * df (comparator.f90:1224-1228)
        const Receiver &r = c->recv[cd.rec];
        if (r.filter.defined()) {
            const float df = 1.f / ((float)ntr * dt);
            plf_taper_array(r.filter, c->filtw_h.data() + pr.specofs, 0, nb - 1, df, IP_COS);
        }
        prs.push_back(pr);
        fofs += ntr; sofs += nb;
    }
    c->fft_d.ensure((size_t)fofs, &c->dev_bytes);
    c->spec_d.ensure((size_t)sofs, &c->dev_bytes);
    c->refamp_d.ensure(c->refamp_h.size(), &c->dev_bytes);
    c->filtw_d.ensure(c->filtw_h.size(), &c->dev_bytes);
    c->reffilt_d.ensure(c->reffilt_h.size(), &c->dev_bytes);
    DevBuf<FftPair> prs_d;
    prs_d.ensure(prs.size(), nullptr);
    HIPCHECK(hipMemcpyAsync(prs_d.p, prs.data(), prs.size() * sizeof(FftPair), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(c->filtw_d.p, c->filtw_h.data(), c->filtw_h.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    // FFT input rows: the tapered reference over the window, zero padded
    {
        std::vector<float> rows((size_t)fofs, 0.f);
        for (auto &pr : prs) {
            const CompDev &cd = c->comps[pr.slot];
            if (cd.untapered) continue;            // the reference array of an un-tapered slot follows the pair's span: made per pair on the device (run_chunk); only the filter weights of (slot, ntrans) are kept here
            std::memcpy(rows.data() + pr.fft_ofs, c->reft_h.data() + cd.refofs, (size_t)cd.wlen * sizeof(float));
        }
        HIPCHECK(hipMemcpyAsync(c->fft_d.p, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIPCHECK(hipStreamSynchronize(c->stream));
    }
    int lds_lo, lds_hi;
    lds_fft_range(c, lds_lo, lds_hi);
    auto in_lds = [&](int n) { return n >= lds_lo && n <= lds_hi; };
    // rows of equal length are contiguous (sorted): one batched transform per length; lib_only: the lengths the in-LDS kernels leave
    auto for_each_length = [&](bool forward, bool lib_only) {
        size_t i = 0;
        while (i < prs.size()) {
            size_t j = i;
            while (j < prs.size() && prs[j].ntrans == prs[i].ntrans) j++;
            if (!(lib_only && in_lds(prs[i].ntrans))) fft_rows(c, prs[i].ntrans, (long long)(j - i), prs[i].fft_ofs, prs[i].spec_ofs, forward);
            i = j;
        }
    };
    // Every variant by the transform its trial sources go through -- in LDS or the library's, by its LENGTH, whatever else is in the
    // batch (a pair's result must not depend on its neighbours): a trial source that reproduces the reference trace bit for bit then
    // has the misfit 0 exactly.
    bool any_lds = false, any_lib = false;
    int longest = 0;
    for (auto &pr : prs) { if (in_lds(pr.ntrans)) { any_lds = true; longest = std::max(longest, pr.ntrans); fused_fft_table(c, pr.ntrans); } else any_lib = true; }
    SpecParams sp{ c->method, dt, c->syn_factor, c->nmis, 0, c->any_filter ? 1 : 0 };
    sp.lds_lo = lds_lo; sp.lds_hi = lds_hi;
    if (spectral) {
        if (any_lds)
            hipLaunchKernelGGL(spec_fft_norm_kernel<1>, dim3((unsigned)prs.size()), dim3(256), (size_t)longest * 4, c->stream, c->fft_d.p, prs_d.p, fused_fft_tables(c),
                               (const float *)nullptr, c->filtw_d.p, sp, (float *)nullptr, c->refamp_d.p, SynRows{});
        if (any_lib) {
            for_each_length(true, true);
            hipLaunchKernelGGL(ref_amp_kernel, dim3((unsigned)prs.size()), dim3(256), 0, c->stream, c->spec_d.p, prs_d.p, c->filtw_d.p, c->refamp_d.p, lds_lo, lds_hi);
        }
    } else {
        for_each_length(true, false);
        hipLaunchKernelGGL(ref_amp_kernel, dim3((unsigned)prs.size()), dim3(256), 0, c->stream, c->spec_d.p, prs_d.p, c->filtw_d.p, c->refamp_d.p, 1, 0);
    }
    if (c->any_filter && !spectral) {
        sp.has_filter = 1;
        if (any_lds)      // filtered references by the transform pair the trial sources go through (spec_fft_filter_norm_kernel); the rows in
                          // fft_d are untouched by the library transform above (out of place)
            hipLaunchKernelGGL(spec_fft_filter_norm_kernel<1>, dim3((unsigned)prs.size()), dim3(256), (size_t)longest * 4, c->stream, c->fft_d.p, prs_d.p,
                               fused_fft_tables(c), c->comps_d.p, c->filtw_d.p, (const float *)nullptr, c->zmask_d.p, sp, (float *)nullptr,
                               c->reffilt_d.p, SynRows{});
        if (any_lib) {
            hipLaunchKernelGGL(spec_filter_kernel, dim3((unsigned)prs.size()), dim3(256), 0, c->stream, c->spec_d.p, prs_d.p, c->comps_d.p, c->filtw_d.p, lds_lo, lds_hi);
            for_each_length(false, true);
            hipLaunchKernelGGL(ref_filt_kernel, dim3((unsigned)prs.size()), dim3(256), 0, c->stream, c->fft_d.p, prs_d.p, c->comps_d.p,
                               c->zmask_d.p, c->reffilt_d.p, lds_lo, lds_hi);
        }
    }
    HIPCHECK(hipGetLastError());
    // results of the fresh variants down to the host mirrors (the device arrays were re-allocated if they grew: the
    // mirrors are the master copy and go up again as a whole)
    for (auto &pr : prs) {
        const CompDev &cd = c->comps[pr.slot];
        const int nb = pr.ntrans / 2 + 1;
        HIPCHECK(hipMemcpyAsync(c->refamp_h.data() + pr.specofs, c->refamp_d.p + pr.specofs, nb * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        if (c->any_filter && !spectral)
            HIPCHECK(hipMemcpyAsync(c->reffilt_h.data() + pr.filtofs, c->reffilt_d.p + pr.filtofs, cd.wlen * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHECK(hipStreamSynchronize(c->stream));
    HIPCHECK(hipMemcpyAsync(c->refamp_d.p, c->refamp_h.data(), c->refamp_h.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(c->reffilt_d.p, c->reffilt_h.data(), c->reffilt_h.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    // norm factors of the reference (probe_norm, comparator.f90:954-996)
    for (auto &pr : prs) {
        const CompDev &cd = c->comps[pr.slot];
        double sum = 0.0, pk = 0.0;
        float nf = c->norm_h[pr.slot];                        // time-domain norm of a slot without filter: unchanged
        if (spectral) {
            const int nb = pr.ntrans / 2 + 1;
            const float df = 1.f / ((float)pr.ntrans * dt);
            for (int k = 0; k < nb; k++) {
                const float a = c->refamp_h[pr.specofs + k];
                sum += (c->method == KIWI_AMPSPEC_L2NORM) ? (double)a * (double)a : (double)std::fabs(a);
            }
            nf = (c->method == KIWI_AMPSPEC_L2NORM) ? 1.f * (float)std::sqrt((double)df * sum) : 1.f * (float)((double)df * sum);
        } else if (c->slot_has_filter[pr.slot]) {
            for (int i = 0; i < cd.wlen; i++) {
                const float a = c->reffilt_h[pr.filtofs + i];
                switch (c->method) {
                case KIWI_L2NORM: sum += (double)a * (double)a; break;
                case KIWI_L1NORM: sum += (double)std::fabs(a); break;
                case KIWI_SCALAR_PRODUCT: sum += (double)(a * a); break;
                default: pk = std::max(pk, (double)std::fabs(a)); break;
                }
            }
            switch (c->method) {
            case KIWI_L2NORM: nf = 1.f * (float)std::sqrt((double)dt * sum); break;
            case KIWI_L1NORM: nf = 1.f * (float)((double)dt * sum); break;
            case KIWI_SCALAR_PRODUCT: nf = (1.f * 1.f) * (float)sum; break;
            default: nf = 1.f * (float)pk; break;
            }
        }
        c->variants[std::make_pair(pr.slot, pr.ntrans)] = kiwi_hip_ctx::FftVariant{ pr.ntrans, pr.specofs, pr.filtofs, nf };
    }
}

// Host half of the per-pair sizing: transform lengths of the chunk (already on their way down) -> buckets of equal
// length with contiguous rows, reference variants, the pair table for the kernels, norm factors per (source, slot).
void layout_fft_chunk(kiwi_hip_ctx *c, int isrc0, int nsrc)
{
    HIPCHECK(hipEventSynchronize(c->size_event));
    const size_t np = (size_t)nsrc * c->nmis;
    const int *ntr = c->ntr_pin;
    std::map<int, long long> count;
    std::vector<std::pair<int, int>> want;
    {
        std::vector<int> seen((size_t)c->nmis, 0);               // last length seen per slot: most pairs repeat it
        for (size_t i = 0; i < np; i++) {
            const int m = (int)(i % c->nmis);
            count[ntr[i]]++;
            if (seen[m] != ntr[i]) { seen[m] = ntr[i]; want.emplace_back(m, ntr[i]); }
        }
        std::sort(want.begin(), want.end());
        want.erase(std::unique(want.begin(), want.end()), want.end());
    }
    make_variants(c, want);
    c->buckets.clear();
    std::map<int, size_t> bucket_of;
    long long fb = 0, sbase = 0;
    for (auto &kv : count) {
        bucket_of[kv.first] = c->buckets.size();
        c->buckets.push_back({ kv.first, 0, fb, sbase });
        fb += kv.second * kv.first;
        sbase += kv.second * (kv.first / 2 + 1);
    }
    c->fft_d.ensure((size_t)fb, &c->dev_bytes);                 // within the capacity prepare_fft laid out; grows only if not
    c->spec_d.ensure((size_t)sbase, &c->dev_bytes);
    if (c->untapered_fft) {
        c->refpair_d.ensure((size_t)sbase, &c->dev_bytes);
        c->reffiltpair_d.ensure((size_t)fb, &c->dev_bytes);
    }
    pin_ensure(c->pairs_pin, c->pairs_pin_n, np);
    c->norm_src_h.resize((size_t)c->nsrc * c->nmis, 0.f);
    int last_ntr = -1; size_t last_b = 0;
    for (size_t i = 0; i < np; i++) {
        const int m = (int)(i % c->nmis);
        if (ntr[i] != last_ntr) { last_ntr = ntr[i]; last_b = bucket_of[last_ntr]; }
        auto &b = c->buckets[last_b];
        const auto &v = c->variants[std::make_pair(m, ntr[i])];
        FftPair &pr = c->pairs_pin[i];
        pr.fft_ofs = b.fft_base + b.count * b.ntrans;
        pr.spec_ofs = b.spec_base + b.count * (b.ntrans / 2 + 1);
        pr.ntrans = b.ntrans; pr.specofs = v.specofs; pr.filtofs = v.filtofs; pr.slot = m;
        b.count++;
        c->norm_src_h[(size_t)isrc0 * c->nmis + i] = v.norm;
    }
    c->pairs_d.ensure(np, &c->dev_bytes);
    HIPCHECK(hipMemcpyAsync(c->pairs_d.p, c->pairs_pin, np * sizeof(FftPair), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(c->normsrc_d.p + (size_t)isrc0 * c->nmis, c->norm_src_h.data() + (size_t)isrc0 * c->nmis, np * sizeof(float),
                            hipMemcpyHostToDevice, c->stream));
    c->last_pairs.assign(c->pairs_pin, c->pairs_pin + np);
}

// may this evaluation compare inside the accumulate kernel (no synthetics in memory)?
bool can_fuse(const kiwi_hip_ctx *c, int proc_which, int isrc0, int nsrc)
{
    // whenever nothing needs the synthetics themselves: saves writing and re-reading them (a third of the time for point
    // sources; 0.2 ms of 12 per 256 sources of 100 centroids, where the epilogue's registers cost the main loop 1 %)
    (void)isrc0; (void)nsrc;
    return c->fuse_enabled && c->accum_mode == 0 && proc_which == 0 && !c->fft_needed && !c->floating && !c->synth_only && !c->any_untapered &&
           c->halo == 0 && (c->method == KIWI_L2NORM || c->method == KIWI_L1NORM || c->method == KIWI_SCALAR_PRODUCT || c->method == KIWI_PEAK);
}

// groups of four trial sources per workgroup: largest difference of their origin times in samples (experiment switch)
static int kiwi_quad_shift_span()
{
    static const int v = [] { const char *m = std::getenv("KIWI_HIP_QUAD_SPAN"); return m ? std::atoi(m) : 16; }();
    return v;
}

void run_chunk(kiwi_hip_ctx *c, int isrc0, int nsrc, int proc_which)
{
    const int nrec = (int)c->recv.size();
    const int cbeg = c->cent_ofs[isrc0], cend = c->cent_ofs[isrc0 + nsrc];
    int maxnc = 0;
    for (int s = isrc0; s < isrc0 + nsrc; s++) maxnc = std::max(maxnc, c->cent_ofs[s + 1] - c->cent_ofs[s]);
    c->recs_d.ensure((size_t)(cend - cbeg) * nrec, &c->dev_bytes);
    int *tab = nullptr;
    int *off4 = nullptr;              // compact descriptors: the rows of `tab` are then address space only, nothing writes or reads them
    if (c->accum_mode == 0) {
        c->tab_d.ensure((size_t)(cend - cbeg) * nrec * 128, &c->dev_bytes); tab = c->tab_d.p;
        if (c->db_simple && c->compact) {
            c->off4_d.ensure((size_t)(cend - cbeg) * nrec * 4, &c->dev_bytes); off4 = c->off4_d.p;
            if (std::getenv("KIWI_HIP_POISON")) HIPCHECK(hipMemsetAsync(off4, 0x7f, (size_t)(cend - cbeg) * nrec * 4 * sizeof(int), c->stream));
        }
        // KIWI_HIP_POISON=1 (tests): rows keep nothing from earlier evaluations -- a descriptor line the kernel reads but
        // geometry_kernel did not write shows as a wild address instead of passing by accident
        if (std::getenv("KIWI_HIP_POISON")) HIPCHECK(hipMemsetAsync(tab, 0x7f, (size_t)(cend - cbeg) * nrec * 128 * sizeof(int), c->stream));
        c->coef_d.ensure((size_t)(cend - cbeg) * nrec * kCoefLine + 160, &c->dev_bytes);   // (+ 640 bytes: accumulate_multi_kernel warms 512 bytes from a group's first line)
    }
    // ---- sources of this chunk that can take an earlier source's synthetics (same centroid table, same chunk): not
    // synthesised, compared from that source's row with their own moment and rise time.  Plain time-domain comparator only.
    // (Their geometry records are still made: a run of geometry-identical sources takes its group structure from its first
    // member, whichever that is; the geometry kernel is a few per cent of a step.)
    const int *synrow = nullptr, *famofs = nullptr, *famlist = nullptr;
    // (point sources -- a couple of centroids each -- cost less to synthesise than to look up: left alone unless forced, = 2)
    const bool heavy = c->dedupe_enabled == 2 || (size_t)(cend - cbeg) >= (size_t)8 * (size_t)nsrc;
    if (c->dedupe_enabled && heavy && c->any_same && !c->fft_needed && !c->floating && !c->any_untapered && !c->want_spansrc && !c->synth_only &&
        proc_which == 0) {
        std::vector<int> sr((size_t)nsrc);
        int ndup = 0;
        for (int s = 0; s < nsrc; s++) {
            const int f = c->same_as[(size_t)isrc0 + s];
            sr[s] = (f >= isrc0 && f != isrc0 + s) ? f - isrc0 : s;
            ndup += sr[s] != s;
        }
        // Worth it where repeats are what the batch is made of (moment / rise-time sweeps: most sources repeat an earlier table).  A
        // few stray repeats -- a strike sweep that passes 360 degrees -- are synthesised like everybody else: shared synthetics rule
        // out the kernels with several sources per workgroup for the WHOLE chunk (measured, cfg3-w600: a chunk with 28 % repeats
        // 51 ms instead of 19)
        const bool any = c->dedupe_enabled == 2 ? ndup > 0 : 2 * ndup >= nsrc;
        if (any) {
            c->synrow_d.ensure((size_t)nsrc, &c->dev_bytes);
            HIPCHECK(hipMemcpyAsync(c->synrow_d.p, sr.data(), (size_t)nsrc * sizeof(int), hipMemcpyHostToDevice, c->stream));
            synrow = c->synrow_d.p;
            if (c->fuse_now) {
                // fused comparator: the workgroup that synthesised a source compares its followers too (their moments differ)
                std::vector<int> ofs((size_t)nsrc + 1, 0), lst;
                for (int s = 0; s < nsrc; s++) if (sr[s] != s) ofs[(size_t)sr[s] + 1]++;
                for (int s = 0; s < nsrc; s++) ofs[(size_t)s + 1] += ofs[s];
                lst.resize((size_t)ofs[nsrc]);
                std::vector<int> fill(ofs.begin(), ofs.end() - 1);
                for (int s = 0; s < nsrc; s++) if (sr[s] != s) lst[(size_t)fill[sr[s]]++] = s;
                c->famofs_d.ensure(ofs.size(), &c->dev_bytes);
                c->famlist_d.ensure(std::max<size_t>(lst.size(), 1), &c->dev_bytes);
                HIPCHECK(hipMemcpyAsync(c->famofs_d.p, ofs.data(), ofs.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
                HIPCHECK(hipMemcpyAsync(c->famlist_d.p, lst.data(), lst.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
                HIPCHECK(hipStreamSynchronize(c->stream));            // ofs / lst go out of scope
                famofs = c->famofs_d.p; famlist = c->famlist_d.p;
            }
            HIPCHECK(hipStreamSynchronize(c->stream));
        }
    }
    const bool fuse = c->fuse_now;
    if (!fuse) c->syn_d.ensure((size_t)nsrc * c->syn_stride, &c->dev_bytes);
    float *proc = nullptr;
    if (proc_which) { c->proc_d.ensure((size_t)nsrc * c->syn_stride, &c->dev_bytes); proc = c->proc_d.p; }

    // cell groups pay where most centroids are points of their own (no blended tile to share between time steps)
    // (with nearest-neighbour interpolation there is nothing to blend: same-point groups do)
    const bool cell = c->accum_mode == 0 && (c->cell_mode == 1 || (c->cell_mode < 0 && c->bilinear && c->points_per_centroid > 0.5));
    // several sources per workgroup (accumulate_multi_kernel); decided below, once the runs and the shared synthetics are known
    const bool duo_maybe = c->accum_mode == 0 && c->duo && !cell && c->max_wlen >= 128 && !c->group_threads_env && nsrc >= 2;      // (four sources: 256-sample tiles)
    EvalParams ep{ c->bilinear, c->xus, c->zus, nrec, isrc0, cell ? 1 : 0, c->db_gaps ? 0 : 1 };
    int *spansrc = nullptr;
    if (c->any_untapered || c->want_spansrc || c->fft_needed) {     // per-source strip spans, initialised empty
        const size_t n = (size_t)nsrc * nrec;
        c->spansrc_d.ensure(n * kSpanInts, &c->dev_bytes);
        hipLaunchKernelGGL(span_init_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, c->stream, c->spansrc_d.p, 2 * n);
        spansrc = c->spansrc_d.p;
    }
    hipEvent_t e0, e1, e2, e3;
    record(c, 0, e0);
    if (maxnc > 0) {
        dim3 grid((unsigned)((maxnc * nrec + 255) / 256), (unsigned)nsrc);
        if (cell || duo_maybe) {
            c->pairflag_d.ensure((size_t)nsrc * nrec, &c->dev_bytes);
            HIPCHECK(hipMemsetAsync(c->pairflag_d.p, 0, (size_t)nsrc * nrec * sizeof(int), c->stream));
        }
        int *lmax = nullptr;
        if (c->db_gaps && spansrc) {      // (a pass in front for the order-dependent part of the strips' spans, see geometry_kernel)
            c->lmax_d.ensure((size_t)nsrc * nrec, &c->dev_bytes);
            HIPCHECK(hipMemsetAsync(c->lmax_d.p, 0xff, (size_t)nsrc * nrec * sizeof(int), c->stream));
            lmax = c->lmax_d.p;
            hipLaunchKernelGGL(geometry_kernel, grid, dim3(256), 0, c->stream, c->cent_d.p, c->centofs_d.p, ep, c->gm,
                               c->span.p, c->recv_d.p, (GeoRec *)nullptr, (int *)nullptr, (float *)nullptr, (int *)nullptr, (int *)nullptr, (int *)nullptr, c->endz.p, (const int *)nullptr,
                               lmax, (const int *)nullptr, (int4 *)nullptr);
        }
        hipLaunchKernelGGL(geometry_kernel, grid, dim3(256), 0, c->stream, c->cent_d.p, c->centofs_d.p, ep, c->gm,
                           c->span.p, c->recv_d.p, c->recs_d.p, tab, c->coef_d.p, (int *)nullptr, spansrc, (cell || duo_maybe) ? c->pairflag_d.p : (int *)nullptr, c->endz.p, (const int *)nullptr,
                           (int *)nullptr, (const int *)lmax, (int4 *)off4);
        if (cell)
            hipLaunchKernelGGL(cellgroup_kernel, grid, dim3(256), 0, c->stream, c->centofs_d.p, ep, c->gm, c->span.p, c->recv_d.p,
                               c->recs_d.p, tab, c->coef_d.p, c->pairflag_d.p, c->endz.p, (const int *)nullptr,
                               c->cell_wave ? exact::cellw_range() : kHalo - 10, off4 ? 1 : 0);
    }
    if (c->fft_needed && !c->untapered_fft) {
        // transform length of every (source, slot) pair from the source's own strip spans; the lengths travel to the host
        // while the accumulate kernel runs.  (With un-tapered slots among them the spans follow the synthetics' values: sized
        // behind the accumulate kernel, below.)
        const size_t np = (size_t)nsrc * c->nmis;
        c->ntr_d.ensure(np, &c->dev_bytes);
        pin_ensure(c->ntr_pin, c->ntr_pin_n, np);
        hipLaunchKernelGGL(fft_size_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, c->stream, spansrc, c->comps_d.p, c->nmis,
                           nsrc, nrec, c->risetime_d.p + isrc0, c->gm.dt, c->ntr_d.p, (const int *)nullptr);
        HIPCHECK(hipMemcpyAsync(c->ntr_pin, c->ntr_d.p, np * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(hipEventRecord(c->size_event, c->stream));
    }
    record(c, 0, e1);
    int fuse_T = 0, fuse_tile = 0, fuse_ntiles = 0, fuse_nparts = 0;
    bool fuse_all = false;
    {
        dim3 grid((unsigned)((c->max_wlen + kTile - 1) / kTile), (unsigned)nrec, (unsigned)nsrc);
        // the kernels of the arithmetic contract in force (kiwi_accum.inc compiled twice: kiwi::exact, kiwi::fused)
        const bool fusedar = c->arith == KIWI_ARITH_FUSED;
        AccumArgs aa{ c->stream, c->gm.ng, c->fuse_now, c->G.p, c->span.p, c->gm.pitch, c->recs_d.p, c->centofs_d.p, isrc0, nrec, c->recv_d.p,
                      c->syn_d.p, c->syn_stride, off4 ? off4 : c->tab_d.p, c->coef_d.p, FuseParams{ nullptr, nullptr, nullptr, nullptr, 0, 1.f, 0, 0, 0 },
                      nullptr, synrow, famofs, famlist, off4 ? 1 : 0 };
        if (c->accum_mode == 1) {            // KIWI_HIP_ACCUM=direct: A/B reference kernel, no LDS staging
            if (fusedar) fused::launch_direct(aa, grid); else exact::launch_direct(aa, grid);
        } else {
            // workgroup size: env override, else by window length (halo overhead vs tile fit)
            const int T = c->group_threads_env ? c->group_threads : (c->max_wlen >= 2048 ? 256 : (c->max_wlen >= 384 ? 128 : 64));
            const int ntiles = (c->max_wlen + 4 * T - 1) / (4 * T);
            const int ntiles_p = (c->max_wlen + 511) / 512;              // accumulate_multi_kernel: 512 samples per source with two of them
            // cell mode: accumulate_cell_kernel (256 threads, tile = spl x 256 samples) takes the pairs of cell_pair(), the
            // grouped kernel behind it the others
            const int spl = 2, Tc = 256;      // cell kernels: 256 threads, two output samples per lane
            const int ntiles_c = (c->max_wlen + spl * Tc - 1) / (spl * Tc);
            // runs of geometry-identical single-group sources (chunk-local indices); singletons otherwise
            int *runs = nullptr;
            unsigned gx = (unsigned)nsrc;
            if (c->share_runs && !cell && (!synrow || fuse)) {     // (runs + shared synthetics: only with the fused comparator)
                std::vector<int> rf;
                // keep enough workgroups in flight: no run longer than nsrc / 1024 rounded up, nor than max_run
                const int cap = std::max(1, std::min(c->max_run, (nsrc * ntiles * nrec) / 8192));
                int s = 0;
                while (s < nsrc) {
                    int e = s + 1;
                    const int g = isrc0 + s;
                    if (c->single_group[g])
                        while (e < nsrc && e - s < cap && c->single_group[isrc0 + e] && c->geo_hash[isrc0 + e] == c->geo_hash[g] &&
                               c->cent_ofs[isrc0 + e + 1] - c->cent_ofs[isrc0 + e] == c->cent_ofs[g + 1] - c->cent_ofs[g]) e++;
                    rf.push_back(s);
                    s = e;
                }
                if ((int)rf.size() < nsrc) {
                    rf.push_back(nsrc);
                    c->runfirst_d.ensure(rf.size(), &c->dev_bytes);
                    HIPCHECK(hipMemcpyAsync(c->runfirst_d.p, rf.data(), rf.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
                    HIPCHECK(hipStreamSynchronize(c->stream));         // rf goes out of scope
                    runs = c->runfirst_d.p;
                    gx = (unsigned)rf.size() - 1;
                }
            }
            // aligned groups of four / two consecutive sources of equal structure (see accumulate_multi_kernel); four only where
            // their origin times are within 16 samples of each other (their groups then share a tile origin: that of the largest
            // shift; further apart the four tile sets would be built one after the other -- two at a time then)
            bool duo = duo_maybe && !runs && !synrow && maxnc > 0;
            bool any4 = false, any2 = false;
            if (duo) {
                // (flags staged in pinned memory the context owns: the upload needs no stream synchronisation -- the host only waits,
                // before it rewrites them for the next chunk, until the previous upload has been read)
                const size_t n4 = (size_t)(nsrc + 3) / 4, n2 = (size_t)(nsrc + 1) / 2;
                pin_ensure(c->mate_pin, c->mate_pin_n, n4 + n2);
                if (!c->mate_event) HIPCHECK(hipEventCreateWithFlags(&c->mate_event, hipEventDisableTiming));
                else HIPCHECK(hipEventSynchronize(c->mate_event));
                int *m4 = c->mate_pin, *m2 = c->mate_pin + n4;
                std::fill(m4, m4 + n4 + n2, 0);
                // ... and only sources that are NEIGHBOURS in space (first and last centroid within a quarter of the database's node
                // spacing): their groups then sit in the same cells for most receivers.  Sources further apart would have their tile
                // sets built one after the other, from shorter tiles -- slower than the grouped kernel (measured: a shuffled
                // location grid 53 instead of 42 ms per 1024 sources).
                const float near_h = 0.25f * c->gm.dx * (float)c->xus, near_z = 0.25f * c->gm.dz * (float)c->zus;
                auto same = [&](int a, int b, bool shifts) {
                    const int na = c->cent_ofs[a + 1] - c->cent_ofs[a], nb = c->cent_ofs[b + 1] - c->cent_ofs[b];
                    if (!(na > 0 && na == nb && c->struct_hash[a] == c->struct_hash[b] && (!shifts || std::abs(c->first_shift[a] - c->first_shift[b]) <= kiwi_quad_shift_span()))) return false;
                    // (the hash only sorts out; the kernel walks all sources of a group with the group lengths of the first: compared in full)
                    if (std::memcmp(c->group_lens.data() + c->cent_ofs[a], c->group_lens.data() + c->cent_ofs[b], (size_t)na) != 0) return false;
                    const float *p = c->src_ends.data() + (size_t)a * 6, *q = c->src_ends.data() + (size_t)b * 6;
                    for (int k = 0; k < 6; k++)
                        if (std::fabs(p[k] - q[k]) > (k % 3 == 2 ? near_z : near_h)) return false;
                    return true;
                };
                if (c->duo >= 4)
                    for (int k = 0; 4 * k + 3 < nsrc; k++) {
                        const int a = isrc0 + 4 * k;
                        m4[k] = (same(a, a + 1, true) && same(a, a + 2, true) && same(a, a + 3, true)) ? 1 : 0;
                        any4 = any4 || m4[k];
                    }
                // (pairs work on 512-sample tiles: not for windows of one 256-sample tile, where half of their lanes would idle)
                for (int k = 0; c->max_wlen > 256 && 2 * k + 1 < nsrc; k++) {
                    const int a = isrc0 + 2 * k;
                    m2[k] = (!m4[(size_t)k / 2] && same(a, a + 1, false)) ? 1 : 0;
                    any2 = any2 || m2[k];
                }
                duo = any4 || any2;
                if (std::getenv("KIWI_HIP_DEBUG")) {
                    int c4 = 0, c2 = 0; for (size_t i = 0; i < n4; i++) c4 += m4[i]; for (size_t i = 0; i < n2; i++) c2 += m2[i];
                    std::fprintf(stderr, "[kiwi_hip] chunk of %d sources: %d groups of four, %d pairs of equal structure\n", nsrc, c4, c2);
                }
                if (duo) {
                    c->mate4_d.ensure(n4, &c->dev_bytes);
                    c->mate_d.ensure(n2, &c->dev_bytes);
                    HIPCHECK(hipMemcpyAsync(c->mate4_d.p, m4, n4 * sizeof(int), hipMemcpyHostToDevice, c->stream));
                    HIPCHECK(hipMemcpyAsync(c->mate_d.p, m2, n2 * sizeof(int), hipMemcpyHostToDevice, c->stream));
                    HIPCHECK(hipEventRecord(c->mate_event, c->stream));
                }
            }
            const int ntiles_q = (c->max_wlen + 255) / 256;              // ... 256 samples per source with four of them
            dim3 ggrid(gx, (unsigned)(ntiles * nrec));                   // source index fastest (L2 sharing)
            dim3 dgrid((unsigned)((nsrc + 1) / 2), (unsigned)(ntiles_p * nrec)), qgrid((unsigned)((nsrc + 3) / 4), (unsigned)(ntiles_q * nrec));
            dim3 cgrid((unsigned)nsrc, (unsigned)(ntiles_c * nrec));
            if (fuse) {
                // partial sums per (source, slot): [tile][wave] of the kernel that evaluated the pair.  In cell mode two
                // kernels with different tilings share the buffer: it is cleared and misfit_finish_kernel sums all of it
                const int nparts = cell ? std::max(ntiles * (T / 64), ntiles_c * (Tc / 64)) : (duo ? std::max({ ntiles * (T / 64), ntiles_p * 2, ntiles_q }) : ntiles * (T / 64));
                c->fusepart_d.ensure((size_t)nsrc * c->nmis * nparts, &c->dev_bytes);
                if (cell || duo) HIPCHECK(hipMemsetAsync(c->fusepart_d.p, 0, (size_t)nsrc * c->nmis * nparts * sizeof(double), c->stream));
                aa.fp = FuseParams{ c->reft_d.p, c->tw_d.p, c->moment_d.p, c->fusepart_d.p, c->method, c->syn_factor, c->nmis, nparts, isrc0 };
                fuse_nparts = nparts;
            }
            fuse_T = T; fuse_tile = 4 * T; fuse_ntiles = ntiles; fuse_all = cell || duo;
            aa.pairflag = (cell || duo) ? c->pairflag_d.p : (const int *)nullptr;
            // The per-wave cell kernel has a fused build since round 6 (accum_4_fused.o: compiled WITHOUT contraction like the exact
            // one, its multiply-adds fused explicitly in the source, kiwi_accum.inc KIWI_EXPLICIT_FMA; zero scratch).  Until then
            // the cell kernels ran uncontracted under either contract: the contracting compile needs 175-189 registers for the
            // kernel's 168 (cfg4 246 ms per 128 sources against 169).  The shared-tile kernel (KIWI_HIP_CELL_WAVE=0) still does
            // -- bit-identical results are inside any tolerance.  KIWI_HIP_CELL_FUSED=0: the exact build under `fused` too (A/B).
            if (cell) {
                static const bool cell_fused = [] { const char *e = std::getenv("KIWI_HIP_CELL_FUSED"); return !e || std::atoi(e) != 0; }();
                if (c->cell_wave) { if (fusedar && cell_fused) fused::launch_cellw(aa, cgrid, ntiles_c); else exact::launch_cellw(aa, cgrid, ntiles_c); }
                else exact::launch_cell(aa, cgrid, ntiles_c);
            }
            // the (group of sources, receiver) combinations accumulate_multi_kernel takes; the grouped kernel behind it returns at once for those
            const int *m2p = any2 ? c->mate_d.p : (const int *)nullptr, *m4p = any4 ? c->mate4_d.p : (const int *)nullptr;
            if (any4) { if (fusedar) fused::launch_multi(aa, qgrid, 4, ntiles_q, m4p, nullptr); else exact::launch_multi(aa, qgrid, 4, ntiles_q, m4p, nullptr); }
            if (any2) { if (fusedar) fused::launch_multi(aa, dgrid, 2, ntiles_p, m2p, m4p); else exact::launch_multi(aa, dgrid, 2, ntiles_p, m2p, m4p); }
            const int pairsel = cell ? 1 : (duo ? 3 : 0);
            if (fusedar) fused::launch_grouped(aa, ggrid, T, ntiles, runs, pairsel, m2p, m4p);
            else         exact::launch_grouped(aa, ggrid, T, ntiles, runs, pairsel, m2p, m4p);
        }
    }
    record(c, 1, e2);
    if (c->any_untapered) {      // data spans of the synthetic probes: they follow the synthetics' values where a rise time folds them
        c->synspan_d.ensure((size_t)nsrc * c->nmis * 2, &c->dev_bytes);
        hipLaunchKernelGGL(synspan_kernel, dim3((unsigned)c->nmis, (unsigned)nsrc), dim3(256), 0, c->stream, c->syn_d.p, c->syn_stride, c->comps_d.p,
                           c->risetime_d.p + isrc0, c->gm.dt, c->nmis, spansrc, nrec, synrow, c->synspan_d.p);
    }
    if (c->untapered_fft) {
        const size_t np = (size_t)nsrc * c->nmis;
        c->ntr_d.ensure(np, &c->dev_bytes);
        pin_ensure(c->ntr_pin, c->ntr_pin_n, np);
        hipLaunchKernelGGL(fft_size_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, c->stream, spansrc, c->comps_d.p, c->nmis,
                           nsrc, nrec, c->risetime_d.p + isrc0, c->gm.dt, c->ntr_d.p, (const int *)c->synspan_d.p);
        HIPCHECK(hipMemcpyAsync(c->ntr_pin, c->ntr_d.p, np * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(hipEventRecord(c->size_event, c->stream));
    }
    if (c->fft_needed) layout_fft_chunk(c, isrc0, nsrc);
    {
        const bool spectral = (c->method == KIWI_AMPSPEC_L2NORM || c->method == KIWI_AMPSPEC_L1NORM);
        const int fl_method = c->method == KIWI_FLOATING_L2NORM ? KIWI_L2NORM : KIWI_L1NORM;
        const int td_method = spectral ? KIWI_L2NORM : (c->floating ? fl_method : c->method);
        const int fft_mode = !c->fft_needed ? 0 : (spectral ? 3 : 1);        // bit0 write FFT input, bit1 skip the norm
        // kept synthetics "filtered" (3): slots without a filter keep their tapered trace, the others are overwritten by
        // filtered_norm_kernel
        MisfitParams mp{ td_method, c->gm.dt, c->syn_factor, c->nmis, isrc0, proc_which == 3 ? 2 : proc_which, fft_mode, nsrc,
                         c->floating ? 1 : 0 };
        if (c->floating) {
            c->vt_d.ensure((size_t)nsrc * c->syn_stride, &c->dev_bytes);
            c->partial_d.ensure((size_t)nsrc * c->nmis * c->max_ns, &c->dev_bytes);
            c->fshift_d.ensure((size_t)c->nsrc * c->nrec_en, &c->dev_bytes);
        }
        if (fuse) {
            const int nth = nsrc * c->nmis;
            hipLaunchKernelGGL(misfit_finish_kernel, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, c->stream,
                               c->fusepart_d.p, c->comps_d.p, c->nmis, fuse_all ? fuse_nparts : fuse_ntiles * (fuse_T / 64), fuse_T / 64,
                               fuse_all ? 0 : fuse_tile,
                               c->method, c->gm.dt, isrc0, nsrc, c->misfit_d.p);
        }
        // Transforms: a pair whose length fits goes through the in-LDS kernels (spec_fft_norm_kernel / spec_fft_filter_norm_kernel), the
        // others through the library's -- decided per PAIR by its length, so that a pair's result does not depend on what else is in
        // the batch.  The in-LDS kernels take the plain synthetics themselves (fold, moment, taper while the row goes into LDS)
        // unless the processed synthetics are to be kept (spectral norms: rows from misfit_kernel then; filtered time-domain norms:
        // the library pair).
        int lds_lo = 1, lds_hi = 0;
        if (c->fft_needed) lds_fft_range(c, lds_lo, lds_hi);
        const bool direct_ok = !proc && !fuse;
        if (!spectral && !direct_ok) { lds_lo = 1; lds_hi = 0; }
        bool any_lds = false, any_lib = false;
        int lds_longest = 0;
        if (c->fft_needed)
            for (auto &b : c->buckets) { if (b.ntrans >= lds_lo && b.ntrans <= lds_hi) { any_lds = true; lds_longest = std::max(lds_longest, b.ntrans); } else any_lib = true; }
        const bool spec_direct_all = spectral && c->fft_needed && direct_ok && !any_lib;      // no row is needed: misfit_kernel has nothing to do
        if (c->fft_needed && direct_ok) { mp.fft_mode |= 4; mp.lds_lo = lds_lo; mp.lds_hi = lds_hi; }
        SpecParams sp{ c->method, c->gm.dt, c->syn_factor, c->nmis, isrc0, c->any_filter ? 1 : 0 };
        sp.lds_lo = lds_lo; sp.lds_hi = lds_hi;
        if (c->untapered_fft) {
            // Un-tapered slots, reference side first: the reference's padded array over every PAIR's span through the same
            // transforms -> amplitude spectrum (x filter) or filtered trace per pair + the pair's norm factor.  (Before
            // misfit_kernel writes the tapered slots' rows: the way back of the filter overwrites every row.)  Behind it the
            // synthetics' arrays take the un-tapered slots' rows and everything goes through the transforms together.
            sp.refpair = c->refpair_d.p; sp.reffiltpair = c->reffiltpair_d.p;
            const dim3 pg((unsigned)c->nmis, (unsigned)nsrc);
            hipLaunchKernelGGL(untapered_rows_kernel<true>, pg, dim3(256), 0, c->stream, c->syn_d.p, c->syn_stride, c->comps_d.p, c->reft_d.p,
                               c->moment_d.p, c->risetime_d.p, isrc0, c->gm.dt, c->nmis, spansrc, nrec, c->synspan_d.p, c->pairs_d.p, c->fft_d.p);
            fft_buckets(c, true);
            if (spectral) {
                hipLaunchKernelGGL(pair_refamp_kernel, pg, dim3(256), 0, c->stream, c->spec_d.p, c->pairs_d.p, c->comps_d.p, c->filtw_d.p, sp,
                                   c->refpair_d.p, c->normsrc_d.p);
            } else {
                hipLaunchKernelGGL(spec_filter_kernel, dim3((unsigned)(c->nmis * nsrc)), dim3(256), 0, c->stream,
                                   c->spec_d.p, c->pairs_d.p, c->comps_d.p, c->filtw_d.p, 1, 0);
                fft_buckets(c, false);
                hipLaunchKernelGGL(pair_reffilt_kernel, pg, dim3(256), 0, c->stream, c->fft_d.p, c->pairs_d.p, c->comps_d.p, sp, c->synspan_d.p,
                                   c->reffiltpair_d.p, c->normsrc_d.p, c->reffilt_d.p);
            }
        }
        if (!fuse && !spec_direct_all)
        hipLaunchKernelGGL(misfit_kernel, dim3((unsigned)c->nmis, (unsigned)nsrc), dim3(256), 0, c->stream,
                           c->syn_d.p, c->syn_stride, c->comps_d.p, c->reft_d.p, c->tw_d.p, c->moment_d.p,
                           c->risetime_d.p, mp, c->misfit_d.p, proc, c->fft_d.p, c->vt_d.p,
                           c->synspan_d.p, c->pairs_d.p, synrow);
        if (c->floating) {
            hipLaunchKernelGGL(floating_norm_kernel, dim3((unsigned)c->nmis, (unsigned)nsrc), dim3(256), 0, c->stream,
                               c->vt_d.p, c->syn_stride, c->comps_d.p, c->refx_d.p, c->tw_d.p, fl_method, c->gm.dt,
                               c->syn_factor, c->nmis, c->max_ns, c->partial_d.p, c->synspan_d.p);
            const int nth = nsrc * c->nrec_en;
            hipLaunchKernelGGL(floating_select_kernel, dim3((unsigned)((nth + 127) / 128)), dim3(128), 0, c->stream,
                               c->partial_d.p, c->comps_d.p, c->recfirst_d.p, c->nrec_en, c->nmis, c->max_ns, fl_method,
                               isrc0, nsrc, c->misfit_d.p, c->fshift_d.p);
        }
        if (c->fft_needed) {
            if (c->untapered_fft) {
                const dim3 pg((unsigned)c->nmis, (unsigned)nsrc);
                hipLaunchKernelGGL(untapered_rows_kernel<false>, pg, dim3(256), 0, c->stream, c->syn_d.p, c->syn_stride, c->comps_d.p, c->reft_d.p,
                                   c->moment_d.p, c->risetime_d.p, isrc0, c->gm.dt, c->nmis, spansrc, nrec, c->synspan_d.p, c->pairs_d.p, c->fft_d.p);
            }
            const SynRows sr{ c->syn_d.p, c->syn_stride, c->comps_d.p, c->tw_d.p, c->moment_d.p, c->risetime_d.p, synrow };
            if (any_lds) for (auto &b : c->buckets) if (sp.in_lds(b.ntrans)) fused_fft_table(c, b.ntrans);
            if (spectral) {
                if (any_lds) {
                    // transform, amplitude, filter and norm of a (slot, source) row in one pass through LDS
                    if (direct_ok)
                        hipLaunchKernelGGL(spec_fft_norm_kernel<2>, dim3((unsigned)nsrc, (unsigned)c->nmis), dim3(256), (size_t)lds_longest * 4, c->stream,
                                           (const float *)nullptr, c->pairs_d.p, fused_fft_tables(c), c->refamp_d.p, c->filtw_d.p, sp, c->misfit_d.p, (float *)nullptr, sr);
                    else
                        hipLaunchKernelGGL(spec_fft_norm_kernel<0>, dim3((unsigned)nsrc, (unsigned)c->nmis), dim3(256), (size_t)lds_longest * 4, c->stream,
                                           c->fft_d.p, c->pairs_d.p, fused_fft_tables(c), c->refamp_d.p, c->filtw_d.p, sp, c->misfit_d.p, (float *)nullptr, sr);
                }
                if (any_lib) {
                    fft_buckets(c, true, lds_lo, lds_hi);
                    hipLaunchKernelGGL(spec_norm_kernel, dim3((unsigned)c->nmis, (unsigned)nsrc), dim3(256), 0, c->stream,
                                       c->spec_d.p, c->pairs_d.p, c->refamp_d.p, c->filtw_d.p, sp, c->misfit_d.p, c->comps_d.p);
                }
            } else {
                if (any_lds)
                    hipLaunchKernelGGL(spec_fft_filter_norm_kernel<0>, dim3((unsigned)nsrc, (unsigned)c->nmis), dim3(256), (size_t)lds_longest * 4, c->stream,
                                       (const float *)nullptr, c->pairs_d.p, fused_fft_tables(c), c->comps_d.p, c->filtw_d.p, c->reffilt_d.p, c->zmask_d.p, sp,
                                       c->misfit_d.p, (float *)nullptr, sr);
                if (any_lib) {
                    fft_buckets(c, true, lds_lo, lds_hi);
                    hipLaunchKernelGGL(spec_filter_kernel, dim3((unsigned)(c->nmis * nsrc)), dim3(256), 0, c->stream,
                                       c->spec_d.p, c->pairs_d.p, c->comps_d.p, c->filtw_d.p, lds_lo, lds_hi);
                    fft_buckets(c, false, lds_lo, lds_hi);
                    hipLaunchKernelGGL(filtered_norm_kernel, dim3((unsigned)c->nmis, (unsigned)nsrc), dim3(256), 0, c->stream,
                                       c->fft_d.p, c->comps_d.p, c->pairs_d.p, c->reffilt_d.p, c->zmask_d.p, sp, c->misfit_d.p,
                                       proc_which == 3 ? proc : nullptr, c->syn_stride, c->synspan_d.p);
                }
            }
        }
        hipLaunchKernelGGL(global_kernel, dim3((unsigned)((nsrc + 127) / 128)), dim3(128), 0, c->stream,
                           c->misfit_d.p, c->norm_d.p, c->recfirst_d.p, c->nrec_en, c->nmis, isrc0, nsrc, c->global_d.p,
                           c->any_failed ? c->status_d.p : (const int *)nullptr,
                           c->fft_needed ? c->normsrc_d.p : (const float *)nullptr);
    }
    record(c, 2, e3);
    HIPCHECK(hipGetLastError());
    if (c->events.size() >= 3 * 2048) {               // nobody reads the timings: recycle instead of growing without bound
        HIPCHECK(hipStreamSynchronize(c->stream));
        for (auto &ev : c->events) { if (ev.kind == 0) c->event_pool.push_back(ev.a); c->event_pool.push_back(ev.b); }
        c->events.clear();
    }
    c->events.push_back({ e0, e1, 0 });
    c->events.push_back({ e1, e2, 1 });
    c->events.push_back({ e2, e3, 2 });
    c->last_chunk0 = isrc0; c->last_chunkn = nsrc;
    if (proc_which) { c->proc_chunk0 = isrc0; c->proc_chunkn = nsrc; c->proc_which_held = proc_which; }
    else c->proc_which_held = c->proc_which_held;   // proc_d untouched
}

int eval_impl(kiwi_hip_ctx *c, int isrc0, int nsrc, int proc_which)
{
    if (isrc0 < 0 || nsrc < 0 || isrc0 + nsrc > c->nsrc) throw std::runtime_error("source range out of bounds");
    HIPCHECK(hipSetDevice(c->device));
    prepare(c);
    c->misfit_d.ensure((size_t)c->nsrc * c->nmis, &c->dev_bytes);
    c->global_d.ensure((size_t)c->nsrc, &c->dev_bytes);
    if (c->fft_needed && !c->fft_ready) prepare_fft(c, c->reft_h);
    const int nrec = (int)c->recv.size();
    c->fuse_now = can_fuse(c, proc_which, isrc0, nsrc);
    int s = isrc0;
    while (s < isrc0 + nsrc) {
        // greedy chunk bounded by workspace bytes
        size_t bytes = 0;
        int n = 0;
        while (s + n < isrc0 + nsrc) {
            const size_t nc = (size_t)(c->cent_ofs[s + n + 1] - c->cent_ofs[s + n]);
            const size_t syn_bytes = c->fuse_now ? (size_t)c->nmis * 64 * sizeof(double)
                                                             : c->syn_stride * sizeof(float) * ((proc_which ? 2 : 1) + (c->floating ? 1 : 0));
            const size_t add = nc * nrec * (sizeof(GeoRec) + (c->accum_mode == 0 ? 512 + kCoefLine * sizeof(float) : 0)) + syn_bytes;
            if (n > 0 && (bytes + add > c->chunk_bytes_limit || n >= 65535)) break;
            if (c->fft_needed && n >= c->fft_cap) break;
            bytes += add; n++;
        }
        run_chunk(c, s, n, proc_which);
        s += n;
    }
    c->last_isrc0 = isrc0; c->last_nsrc = nsrc; c->last_proc_which = proc_which;
    c->evaluated.resize((size_t)c->nsrc, 0);
    std::fill(c->evaluated.begin() + isrc0, c->evaluated.begin() + isrc0 + nsrc, 1);
    return 0;
}

} // namespace

// ================================================================================================
// pure-read microbenchmark kernels (kiwi_hip_measure_read_bandwidth)
typedef unsigned int read_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read_fill_kernel(read_u4 *p, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        p[i] = read_u4{ (unsigned)i, 1u, 2u, 3u };
}
__global__ __launch_bounds__(256) void read_in_order_kernel(const read_u4 *__restrict__ p, size_t n16, unsigned *__restrict__ sink)
{
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t a = per * blockIdx.x, b = a + per < n16 ? a + per : n16;
    unsigned x = 0u, y = 0u, z = 0u, w = 0u;
    size_t i = a + threadIdx.x;
    for (; i + 7 * 256 < b; i += 8 * 256) {
        read_u4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(p + i + 256 * k);
#pragma unroll
        for (int k = 0; k < 8; k++) { x ^= v[k].x; y ^= v[k].y; z ^= v[k].z; w ^= v[k].w; }
    }
    for (; i < b; i += 256) { const read_u4 v = p[i]; x ^= v.x; y ^= v.y; z ^= v.z; w ^= v.w; }
    if ((x ^ y ^ z ^ w) == 0x9e3779b9u) sink[blockIdx.x] = x;        // (never true for the fill pattern: keeps the loads)
}

extern "C" {

int kiwi_hip_init(int device, kiwi_hip_ctx **out)
{
    kmp_set_blocktime(0);          // LLVM OpenMP runtime: workers sleep right after a parallel region instead of spinning
    kiwi_hip_ctx *c = nullptr;
    try {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0)
            return fail(nullptr, std::string("no HIP device available: ") + hipGetErrorString(e));
        if (device < 0 || device >= ndev) return fail(nullptr, "device index out of range");
        HIPCHECK(hipSetDevice(device));
        c = new kiwi_hip_ctx();
        c->device = device;
        HIPCHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        if (const char *m = std::getenv("KIWI_HIP_GROUP_THREADS")) {
            const int v = std::atoi(m);
            if (v == 64 || v == 128 || v == 256) { c->group_threads = v; c->group_threads_env = 1; }
        }
        if (const char *m = std::getenv("KIWI_HIP_ACCUM")) c->accum_mode = (std::strcmp(m, "direct") == 0) ? 1 : 0;
        if (const char *m = std::getenv("KIWI_HIP_FUSE")) c->fuse_enabled = std::atoi(m);
        if (const char *m = std::getenv("KIWI_HIP_DUO")) { const int v = std::atoi(m); c->duo = v >= 4 ? 4 : (v >= 1 ? 2 : 0); }
        if (const char *m = std::getenv("KIWI_HIP_CELL")) c->cell_mode = std::atoi(m) ? 1 : 0;
        if (const char *m = std::getenv("KIWI_HIP_DEDUPE")) c->dedupe_enabled = std::atoi(m);      // 0 off, 1 default, 2 also for point sources
        if (const char *m = std::getenv("KIWI_HIP_FUSED_FFT")) c->fused_fft = std::atoi(m) != 0;   // 0: amplitude spectra through hipFFT
        if (const char *m = std::getenv("KIWI_HIP_ARITH")) {
            if (std::strcmp(m, "fused") == 0 || std::strcmp(m, "fma") == 0) c->arith = KIWI_ARITH_FUSED;
            else if (std::strcmp(m, "exact") != 0) throw std::runtime_error("KIWI_HIP_ARITH: exact or fused");
        }
        if (const char *m = std::getenv("KIWI_HIP_CELL_WAVE")) c->cell_wave = std::atoi(m) ? 1 : 0;
        if (const char *m = std::getenv("KIWI_HIP_COMPACT")) c->compact = std::atoi(m) ? 1 : 0;
        if (const char *m = std::getenv("KIWI_HIP_CHUNK_MB")) {      // workspace bound per launch (default 16 GiB); tests use it
            const long v = std::atol(m);
            if (v > 0) c->chunk_bytes_limit = (size_t)v << 20;
        }
        if (const char *m = std::getenv("KIWI_HIP_RUNS")) {          // 0: no tile sharing across sources; n > 1: longest run
            const int v = std::atoi(m);
            c->share_runs = v != 0;
            if (v > 1) c->max_run = v;
        }
        *out = c;
        return 0;
    } catch (const std::exception &e) {
        delete c;
        return fail(nullptr, e.what());
    }
}

// One context per device inside ONE process (the Fortran host and its machine of 8 GPUs; counterpart of the process pool of
// python/tunguska/seismosizer.py:785-827): the returned context is the one of the first device and owns the others.  Every
// setter called on it is repeated on them (the Green's function tensor is replicated); kiwi_hip_misfits_for_params
// cuts the trial list into contiguous shards in list order, one per device, each evaluated by a thread of its own into its
// slice of the caller's arrays -- no collective, the results are where the caller wants them.
int kiwi_hip_init_multi(int ndev_wanted, kiwi_hip_ctx **out)
{
    if (!out) return fail(nullptr, "null argument");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, std::string("no HIP device available: ") + hipGetErrorString(e));
    int n = ndev_wanted <= 0 ? ndev : ndev_wanted;
    // more contexts than devices: only on request (tests on a one-GPU box; KIWI_HIP_MULTI_OVERSUBSCRIBE=1)
    const char *over = std::getenv("KIWI_HIP_MULTI_OVERSUBSCRIBE");
    if (n > ndev && !(over && std::atoi(over) != 0))
        return fail(nullptr, "kiwi_hip_init_multi: " + std::to_string(n) + " devices wanted, " + std::to_string(ndev) + " visible");
    kiwi_hip_ctx *c = nullptr;
    if (int rc = kiwi_hip_init(0, &c)) return rc;
    for (int i = 1; i < n; i++) {
        kiwi_hip_ctx *m = nullptr;
        if (int rc = kiwi_hip_init(i % ndev, &m)) { kiwi_hip_destroy(c); return rc; }
        c->mates.push_back(m);
    }
    c->cpu_share = n;
    for (kiwi_hip_ctx *m : c->mates) m->cpu_share = n;
    if (hipSetDevice(c->device) != hipSuccess) { kiwi_hip_destroy(c); return fail(nullptr, "hipSetDevice failed"); }
    *out = c;
    return 0;
}

int kiwi_hip_ndevices(kiwi_hip_ctx *c, int *n)
{
    if (!c || !n) return fail(c, "null argument");
    *n = 1 + (int)c->mates.size();
    return 0;
}

int kiwi_hip_destroy(kiwi_hip_ctx *c)
{
    if (!c) return 0;
    for (kiwi_hip_ctx *m : c->mates) kiwi_hip_destroy(m);
    c->mates.clear();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &ev : c->events) {
        if (ev.kind == 0) (void)hipEventDestroy(ev.a);
        (void)hipEventDestroy(ev.b);
    }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto &kv : c->plans) (void)hipfftDestroy(kv.second);
    if (c->ntr_pin) (void)hipHostFree(c->ntr_pin);
    if (c->mate_pin) (void)hipHostFree(c->mate_pin);
    if (c->mate_event) (void)hipEventDestroy(c->mate_event);
    if (c->pairs_pin) (void)hipHostFree(c->pairs_pin);
    if (c->size_event) (void)hipEventDestroy(c->size_event);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int kiwi_hip_last_error(kiwi_hip_ctx *c, char *buf, int buflen)
{
    const std::string &m = c ? c->err : g_init_error;
    if (buf && buflen > 0) { std::snprintf(buf, (size_t)buflen, "%s", m.c_str()); }
    return 0;
}

int kiwi_hip_set_gfdb(kiwi_hip_ctx *c, int nx, int nz, int ng, int L, float dt, float dx, float dz,
                      float firstx, float firstz, const float *G, const int *first, const int *nsamp)
{
    GUARD_BEGIN
    if (nx < 1 || nz < 1 || L < 1) throw std::runtime_error("bad database dimensions");
    if (ng != 8 && ng != 10) throw std::runtime_error("ng must be 8 or 10 (gfdb.f90:57)");
    if (!(dt > 0.f) || !(dx > 0.f) || !(dz > 0.f)) throw std::runtime_error("dt, dx, dz must be positive");
    HIPCHECK(hipSetDevice(c->device));
    const size_t nrows = (size_t)nx * nz * ng;
    int lmax = 1;
    for (size_t i = 0; i < nrows; i++) {
        if (nsamp[i] < 0 || nsamp[i] > L) throw std::runtime_error("nsamp out of range");
        lmax = std::max(lmax, nsamp[i]);
    }
    // >= pad + n + 5 for load5's clamp; the extra halo of repeated end values lets the grouped kernel's last
    // tile (which overhangs the window by up to kHalo samples) take its clamp-free path
    const int pitch = (kRowPad + lmax + kHalo + 32 + 3) / 4 * 4;
    // The LDS-staged kernels address a group's rows relative to the first row of its cell with 32-bit BYTE offsets (write_tab
    // holds float offsets, the buffer loads take 4 x that as their unsigned scalar offset): the tensor itself may have any
    // size, one cell (two neighbouring distances, all depths between, all components) must stay below 2^30 floats = 4 GB.
    // set_interp checks the same with the undersampling factors.
    if (((size_t)nz + 2) * (size_t)ng * (size_t)pitch >= ((size_t)1 << 30))
        throw std::runtime_error("database rows too long: one distance step of the grid exceeds 2^30 samples (4 GB)");
    // host staging in slabs of rows: [kRowPad zeros | samples | repeated end value]
    c->G.alloc(nrows * (size_t)pitch, &c->dev_bytes);
    c->span.alloc(nrows, &c->dev_bytes);
    std::vector<int2> sp(nrows);
    bool gaps = false;
    std::vector<unsigned char> ez(nrows);
    c->endz.alloc(nrows, &c->dev_bytes);
    const size_t slab = std::max<size_t>(1, ((size_t)64 << 20) / ((size_t)pitch * sizeof(float)));
    std::vector<float> stage(slab * (size_t)pitch);
    for (size_t r0 = 0; r0 < nrows; r0 += slab) {
        const size_t nr = std::min(slab, nrows - r0);
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < (long long)nr; i++) {
            const size_t row = r0 + (size_t)i;
            float *d = stage.data() + (size_t)i * pitch;
            const int n = nsamp[row];
            const float *src = G + row * (size_t)L;
            for (int k = 0; k < kRowPad; k++) d[k] = 0.f;
            for (int k = 0; k < n; k++) d[kRowPad + k] = src[k];
            const float tail = n > 0 ? src[n - 1] : 0.f;
            for (int k = kRowPad + n; k < pitch; k++) d[k] = tail;
            sp[row] = make_int2(first[row], first[row] + n - 1);       // n == 0 -> empty span = not stored
            if (n <= 0) gaps = true;
            ez[row] = (tail == 0.f) ? 1 : 0;
        }
        HIPCHECK(hipMemcpy(c->G.p + r0 * (size_t)pitch, stage.data(), nr * (size_t)pitch * sizeof(float), hipMemcpyHostToDevice));
    }
    HIPCHECK(hipMemcpy(c->span.p, sp.data(), nrows * sizeof(int2), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(c->endz.p, ez.data(), nrows, hipMemcpyHostToDevice));
    c->gm = GfMeta{ nx, nz, ng, pitch, dt, dx, dz, firstx, firstz };
    c->db_gaps = gaps;
    {
        bool simple = !gaps;
        for (size_t node = 0; simple && node < (size_t)nx * nz; node++)
            for (int ig = 0; ig < ng; ig++)
                if (sp[node * ng + ig].x != sp[node * ng].x || !ez[node * ng + ig]) { simple = false; break; }
        c->db_simple = simple;
    }
    c->have_db = true;
    c->prepared = false;            // dirtyfy_database, minimizer_engine.f90:1483
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_gfdb(m, nx, nz, ng, L, dt, dx, dz, firstx, firstz, G, first, nsamp); });
    GUARD_END(c)
}

int kiwi_hip_set_interp(kiwi_hip_ctx *c, int bilinear, int xus, int zus)
{
    if (!c) return fail(nullptr, "null context");
    if (xus < 1 || zus < 1) return fail(c, "undersampling must be >= 1");
    if (c->have_db && ((size_t)xus * c->gm.nz + zus + 1) * (size_t)c->gm.ng * (size_t)c->gm.pitch >= ((size_t)1 << 30))
        return fail(c, "undersampling too coarse for this database: one interpolation cell exceeds 2^30 samples (4 GB)");
    c->bilinear = bilinear ? 1 : 0; c->xus = xus; c->zus = zus;
    // set_local_interpolation / set_spacial_undersampling dirty the seismograms (minimizer_engine.f90:1483-1493):
    // natural-span windows, transform lengths and any kept synthetics are stale
    c->prepared = false;
    c->proc_which_held = 0;
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_interp(m, bilinear, xus, zus); });
}

int kiwi_hip_set_effective_dt(kiwi_hip_ctx *c, float edt)
{
    if (!c) return fail(nullptr, "null context");
    if (!(edt > 0.f)) return fail(c, "effective dt must be positive");
    c->effective_dt = edt;
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_effective_dt(m, edt); });
}

int kiwi_hip_set_source_location(kiwi_hip_ctx *c, float lat_deg, float lon_deg, double ref_time)
{
    c->src_origin.lat = (double)d2r(lat_deg);      // minimizer.f90:517: d2r on default reals
    c->src_origin.lon = (double)d2r(lon_deg);
    c->ref_time = ref_time;
    c->have_origin = true;
    update_receiver_geometry(c);
    c->prepared = false;                           // dirtyfy_source_location
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_source_location(m, lat_deg, lon_deg, ref_time); });
}

int kiwi_hip_set_receivers(kiwi_hip_ctx *c, int nrec, const double *lat_deg, const double *lon_deg,
                           const float *depth, const char *const *components)
{
    GUARD_BEGIN
    if (nrec < 1) throw std::runtime_error("need at least one receiver");
    std::vector<Receiver> rs((size_t)nrec);
    for (int i = 0; i < nrec; i++) {
        Receiver &r = rs[i];
        r.origin.lat = d2r(lat_deg[i]);            // d2r(origin), minimizer_engine.f90:262
        r.origin.lon = d2r(lon_deg[i]);
        r.depth = depth ? depth[i] : 0.f;
        const char *cs = components[i];
        const int nc = (int)std::strlen(cs);
        if (nc > kMaxComp) throw std::runtime_error("too many components at receiver " + std::to_string(i + 1));
        for (int k = 0; k < nc; k++) {             // receiver_init, receiver.f90:168-189
            const int id = component_id(cs[k]);
            bool bad = (id == 0);
            for (int j = 0; j < k; j++) if (std::abs(r.comp[j]) == std::abs(id)) bad = true;
            if (bad)
                throw std::runtime_error("initializing receiver failed: possibly a forbidden combination of receiver "
                                         "components has been given at receiver no. " + std::to_string(i + 1));
            r.comp[k] = id;
        }
        r.ncomp = nc;
        r.enabled = nc > 0;
    }
    c->recv.swap(rs);
    update_receiver_geometry(c);
    c->prepared = false;                           // dirtyfy_receivers
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_receivers(m, nrec, lat_deg, lon_deg, depth, components); });
    GUARD_END(c)
}

int kiwi_hip_switch_receiver(kiwi_hip_ctx *c, int irec, int enabled)
{
    if (irec < 1 || irec > (int)c->recv.size()) return fail(c, "receiver index out of range");
    c->recv[irec - 1].enabled = enabled != 0;
    c->prepared = false;
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_switch_receiver(m, irec, enabled); });
}

int kiwi_hip_set_reference(kiwi_hip_ctx *c, int irec, int icomp, int first, int n, const float *data)
{
    if (irec < 1 || irec > (int)c->recv.size()) return fail(c, "receiver index out of range");
    Receiver &r = c->recv[irec - 1];
    if (icomp < 1 || icomp > r.ncomp) return fail(c, "component index out of range");
    if (n < 1) return fail(c, "empty reference seismogram");
    r.ref[icomp - 1].first = first;
    r.ref[icomp - 1].data.assign(data, data + n);
    c->prepared = false;                           // dirtyfy_ref_probes
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_reference(m, irec, icomp, first, n, data); });
}

static int set_plf(kiwi_hip_ctx *c, int irec, int npts, const float *x, const float *y, bool taper)
{
    // filters: ireceiver 0 = every receiver (minimizer_engine.f90:646-661); tapers are per receiver (:684)
    if (irec < (taper ? 1 : 0) || irec > (int)c->recv.size()) return fail(c, "receiver index out of range");
    if (npts == 1) return fail(c, "need at least two control points");
    Plf p;
    if (npts > 0) { p.x.assign(x, x + npts); p.y.assign(y, y + npts); }
    for (int ir = (irec == 0 ? 1 : irec); ir <= (irec == 0 ? (int)c->recv.size() : irec); ir++)
        (taper ? c->recv[ir - 1].taper : c->recv[ir - 1].filter) = p;
    c->prepared = false;
    return 0;
}

int kiwi_hip_set_taper(kiwi_hip_ctx *c, int irec, int npts, const float *x, const float *y)
{
    if (int rc = set_plf(c, irec, npts, x, y, true)) return rc;
    return forward(c, [&](kiwi_hip_ctx *m) { return set_plf(m, irec, npts, x, y, true); });
}

int kiwi_hip_set_filter(kiwi_hip_ctx *c, int irec, int npts, const float *x, const float *y)
{
    if (int rc = set_plf(c, irec, npts, x, y, false)) return rc;
    return forward(c, [&](kiwi_hip_ctx *m) { return set_plf(m, irec, npts, x, y, false); });
}

int kiwi_hip_set_misfit_method(kiwi_hip_ctx *c, int method)
{
    if (method < 1 || method > 8) return fail(c, "unknown misfit method");
    c->method = method;
    c->prepared = false;
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_misfit_method(m, method); });
}

int kiwi_hip_set_floating_shiftrange(kiwi_hip_ctx *c, int irec, float min_shift, float max_shift)
{
    GUARD_BEGIN
    if (!c->have_db) throw std::runtime_error("set the database first (shifts are converted with its sampling interval)");
    const int lo = (int)std::lround(min_shift / c->gm.dt), hi = (int)std::lround(max_shift / c->gm.dt);   // nint, minimizer_engine.f90:432
    if (irec == 0) {
        for (auto &r : c->recv) { r.float_lo = lo; r.float_hi = hi; }
    } else {
        if (irec < 1 || irec > (int)c->recv.size()) throw std::runtime_error("receiver index out of range");
        c->recv[irec - 1].float_lo = lo; c->recv[irec - 1].float_hi = hi;
    }
    c->prepared = false;
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_floating_shiftrange(m, irec, min_shift, max_shift); });
    GUARD_END(c)
}

int kiwi_hip_get_floating_shifts(kiwi_hip_ctx *c, int isrc0, int nsrc, float *shifts)
{
    GUARD_BEGIN_DEV(c)
    if (!c->prepared || !c->floating) throw std::runtime_error("no floating norm evaluated");
    if (isrc0 < 0 || nsrc < 0 || isrc0 + nsrc > c->nsrc) throw std::runtime_error("source range out of bounds");
    HIPCHECK(hipStreamSynchronize(c->stream));
    std::vector<int> sh((size_t)nsrc * c->nrec_en);
    HIPCHECK(hipMemcpy(sh.data(), c->fshift_d.p + (size_t)isrc0 * c->nrec_en, sh.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < sh.size(); i++) shifts[i] = (float)sh[i] * c->gm.dt;          // minimizer_engine.f90:1122
    return 0;
    GUARD_END(c)
}

int kiwi_hip_set_synthetics_factor(kiwi_hip_ctx *c, float factor)
{
    c->syn_factor = factor;
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_synthetics_factor(m, factor); });
}

static int nparams_any(int sourcetype)
{
    const int n = source_nparams(sourcetype);
    return n > 0 ? n : source_nparams_eikonal(sourcetype);
}

int kiwi_hip_source_nparams(int sourcetype) { return nparams_any(sourcetype); }

static CrustProfile unpack_profile(const float *p)
{
    CrustProfile c;
    std::memcpy(c.vp, p, 8 * sizeof(float));
    std::memcpy(c.vs, p + 8, 8 * sizeof(float));
    std::memcpy(c.rho, p + 16, 8 * sizeof(float));
    std::memcpy(c.thickness, p + 24, 7 * sizeof(float));
    return c;
}

// psm_get_crustal_thickness, parameterized_source.f90:207-222
static float limited_thickness(const kiwi_hip_ctx *c)
{
    float t = crust_thickness(c->origin_profile);
    if (c->crustal_thickness_limit > 0.f) t = std::min(c->crustal_thickness_limit, t);
    return t;
}

// psm_set_default_constraints, parameterized_source.f90:127-145
static void default_constraints(kiwi_hip_ctx *c)
{
    c->constraints.assign(2, HalfSpace{ { 0.f, 0.f, 1500.f }, { 0.f, 0.f, -1.f } });
    c->constraints[1] = HalfSpace{ { 0.f, 0.f, limited_thickness(c) }, { 0.f, 0.f, 1.f } };
}

int kiwi_hip_set_source_crust(kiwi_hip_ctx *c, const float *rupture_profile, const float *origin_profile)
{
    GUARD_BEGIN
    if (!rupture_profile || !origin_profile) throw std::runtime_error("crust profiles missing");
    c->rupture_profile = unpack_profile(rupture_profile);
    c->origin_profile = unpack_profile(origin_profile);
    c->have_crust = true;
    default_constraints(c);
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_source_crust(m, rupture_profile, origin_profile); });
    GUARD_END(c)
}

int kiwi_hip_set_source_crustal_thickness_limit(kiwi_hip_ctx *c, float limit)
{
    GUARD_BEGIN
    c->crustal_thickness_limit = limit;
    if (c->have_crust) default_constraints(c);
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_source_crustal_thickness_limit(m, limit); });
    GUARD_END(c)
}

int kiwi_hip_get_source_crustal_thickness(kiwi_hip_ctx *c, float *thickness)
{
    GUARD_BEGIN
    if (!c->have_crust) throw std::runtime_error("no crust profile set");
    *thickness = limited_thickness(c);
    return 0;
    GUARD_END(c)
}

int kiwi_hip_set_source_constraints(kiwi_hip_ctx *c, int n, const float *points, const float *normals)
{
    GUARD_BEGIN
    if (n < 0) throw std::runtime_error("negative number of constraints");
    c->constraints.resize((size_t)n);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { c->constraints[i].point[k] = points[3 * i + k]; c->constraints[i].normal[k] = normals[3 * i + k]; }
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_source_constraints(m, n, points, normals); });
    GUARD_END(c)
}

int kiwi_hip_discretize_eikonal(int sourcetype, const float *params, int nparams, float effective_dt,
                                const float *rupture_profile, int ncon, const float *points, const float *normals,
                                float *cent, int maxcent, int *ncent, float *moment, float *risetime)
{
    try {
        if (source_nparams_eikonal(sourcetype) != nparams) return 2;
        std::vector<HalfSpace> cons((size_t)std::max(ncon, 0));
        for (int i = 0; i < ncon; i++)
            for (int k = 0; k < 3; k++) { cons[i].point[k] = points[3 * i + k]; cons[i].normal[k] = normals[3 * i + k]; }
        DiscreteSource ds;
        const std::string err = discretize_eikonal(sourcetype, params, effective_dt, unpack_profile(rupture_profile), cons, ds);
        if (!err.empty()) return err[0] == 'E' ? 5 : 6;      // 5: empty rupture area, 6: nucleation point outside
        *ncent = (int)ds.centroids.size();
        if (moment) *moment = ds.moment;
        if (risetime) *risetime = ds.risetime;
        if (cent) {
            if (*ncent > maxcent) return 4;
            std::memcpy(cent, ds.centroids.data(), ds.centroids.size() * sizeof(Centroid));
        }
        return 0;
    } catch (...) { return 1; }
}

int kiwi_hip_discretize(int sourcetype, const float *params, int nparams, float effective_dt,
                        float *cent, int maxcent, int *ncent, float *moment, float *risetime)
{
    try {
        if (source_nparams(sourcetype) != nparams) return 2;
        DiscreteSource ds;
        if (!discretize(sourcetype, params, effective_dt, ds)) return 3;
        *ncent = (int)ds.centroids.size();
        if (moment) *moment = ds.moment;
        if (risetime) *risetime = ds.risetime;
        if (cent) {
            if (*ncent > maxcent) return 4;
            std::memcpy(cent, ds.centroids.data(), ds.centroids.size() * sizeof(Centroid));
        }
        return 0;
    } catch (...) { return 1; }
}

int kiwi_hip_set_sources(kiwi_hip_ctx *c, int nsrc, const int *cent_ofs, const float *cent,
                         const float *moment, const float *risetime)
{
    GUARD_BEGIN
    if (nsrc < 1) throw std::runtime_error("need at least one source");
    if (!c->have_db) throw std::runtime_error("set the database before the sources");
    HIPCHECK(hipSetDevice(c->device));
    if (cent_ofs[0] != 0) throw std::runtime_error("cent_ofs[0] must be 0");
    for (int s = 0; s < nsrc; s++) if (cent_ofs[s + 1] < cent_ofs[s]) throw std::runtime_error("cent_ofs not monotone");
    const size_t ntot = (size_t)cent_ofs[nsrc];
    float maxrise = 0.f;
    for (int s = 0; s < nsrc; s++) maxrise = std::max(maxrise, risetime[s]);
    if (2 * fold_halfwidth(maxrise, c->gm.dt) + 1 > kMaxFold) throw std::runtime_error("rise time too long for the fold kernel");
    c->cent_ofs.assign(cent_ofs, cent_ofs + nsrc + 1);
    c->geo_hash.assign((size_t)nsrc, 0ull);
    c->struct_hash.assign((size_t)nsrc, 0ull);
    c->group_lens.assign(ntot, 0);
    c->first_shift.assign((size_t)nsrc, 0);
    c->src_ends.assign((size_t)nsrc * 6, 0.f);
    c->single_group.assign((size_t)nsrc, 0);
    {
        const float dt = c->gm.dt;
#pragma omp parallel for schedule(static) num_threads(std::max(1, std::min(8, nsrc / 512)))
        for (int s = 0; s < nsrc; s++) {
            const float *ce = cent + (size_t)cent_ofs[s] * 10;
            const int nc = cent_ofs[s + 1] - cent_ofs[s];
            unsigned long long h = 1469598103934665603ull ^ (unsigned long long)nc;        // FNV-1a over the bit patterns of (north, east, depth, time)
            unsigned long long hs = 1469598103934665603ull ^ (unsigned long long)nc;       // ... over what decides the centroid groups (group_len)
            bool one = nc >= 1 && nc <= kMaxGroup;
            int smin = 0, smax = 0;
            for (int k = 0; k < nc; k++) {
                unsigned int w[4];
                std::memcpy(w, ce + (size_t)k * 10, sizeof(w));
                for (int q = 0; q < 4; q++) { h ^= w[q]; h *= 1099511628211ull; }

                if (one) {
                    const float *p = ce + (size_t)k * 10;
                    if (!(p[0] == ce[0] && p[1] == ce[1] && p[2] == ce[2])) one = false;
                    const int sh = (int)std::floor(p[3] / dt);                              // as geometry_kernel's group hint
                    if (k == 0) smin = smax = sh; else { smin = std::min(smin, sh); smax = std::max(smax, sh); }
                    if (smax - smin > kHalo - 10) one = false;
                }
            }
            // boundaries of the centroid groups as geometry_kernel's group_len / starts_group cut them: runs of centroids at the
            // same point (compared with the group's FIRST centroid), integer shifts within the LDS halo, at most kMaxGroup
            for (int k = 0; k < nc;) {
                const float *g0 = ce + (size_t)k * 10;
                int len = 1, lo = (int)std::floor(g0[3] / dt), hi = lo;
                for (int q = k + 1; q < nc && len < kMaxGroup; q++) {
                    const float *p = ce + (size_t)q * 10;
                    if (!(p[0] == g0[0] && p[1] == g0[1] && p[2] == g0[2])) break;
                    const int sh = (int)std::floor(p[3] / dt);
                    const int nlo = std::min(lo, sh), nhi = std::max(hi, sh);
                    if (nhi - nlo > kHalo - 10) break;
                    lo = nlo; hi = nhi; len++;
                }
                hs ^= (unsigned)len; hs *= 1099511628211ull;
                c->group_lens[(size_t)cent_ofs[s] + k] = (unsigned char)len;
                k += len;
            }
            c->geo_hash[s] = h;
            c->struct_hash[s] = hs;
            c->first_shift[s] = nc > 0 ? (int)std::floor(ce[3] / dt) : 0;
            if (nc > 0)
                for (int q = 0; q < 3; q++) { c->src_ends[(size_t)s * 6 + q] = ce[q]; c->src_ends[(size_t)s * 6 + 3 + q] = ce[(size_t)(nc - 1) * 10 + q]; }
            c->single_group[s] = one ? 1 : 0;
        }
        // identical tables (hash over all ten columns, confirmed by comparison)
        c->same_as.assign((size_t)nsrc, 0);
        c->any_same = false;
        {
            std::vector<unsigned long long> th((size_t)nsrc);
#pragma omp parallel for schedule(static) num_threads(std::max(1, std::min(8, nsrc / 512)))
            for (int s = 0; s < nsrc; s++) {
                const unsigned int *w = reinterpret_cast<const unsigned int *>(cent + (size_t)cent_ofs[s] * 10);
                const size_t n = (size_t)(cent_ofs[s + 1] - cent_ofs[s]) * 10;
                unsigned long long h = 1469598103934665603ull ^ (unsigned long long)n;
                for (size_t q = 0; q < n; q++) { h ^= w[q]; h *= 1099511628211ull; }
                th[s] = h;
            }
            std::map<unsigned long long, int> first_of;
            for (int s = 0; s < nsrc; s++) {
                c->same_as[s] = s;
                const int nc = cent_ofs[s + 1] - cent_ofs[s];
                if (nc == 0) continue;
                auto it = first_of.find(th[s]);
                if (it == first_of.end()) { first_of[th[s]] = s; continue; }
                const int f = it->second;
                if (cent_ofs[f + 1] - cent_ofs[f] == nc &&
                    std::memcmp(cent + (size_t)cent_ofs[f] * 10, cent + (size_t)cent_ofs[s] * 10, (size_t)nc * 10 * sizeof(float)) == 0) {
                    c->same_as[s] = f;
                    c->any_same = true;
                }
            }
        }
        // how many of the centroids start a new point (sampled): decides between same-point and same-cell groups
        long long npts = 0, ncen = 0;
        const int stride = std::max(1, nsrc / 64);
        for (int s = 0; s < nsrc; s += stride) {
            const float *ce = cent + (size_t)cent_ofs[s] * 10;
            const int nc = cent_ofs[s + 1] - cent_ofs[s];
            for (int k = 0; k < nc; k++) {
                const float *p = ce + (size_t)k * 10, *q = p - 10;
                if (k == 0 || !(p[0] == q[0] && p[1] == q[1] && p[2] == q[2])) npts++;
            }
            ncen += nc;
        }
        c->points_per_centroid = ncen > 0 ? (double)npts / (double)ncen : 0.0;
    }
    c->cent_d.ensure(std::max<size_t>(ntot, 1) * 10, &c->dev_bytes);
    c->centofs_d.ensure((size_t)nsrc + 1, &c->dev_bytes);
    c->moment_d.ensure((size_t)nsrc, &c->dev_bytes);
    c->risetime_d.ensure((size_t)nsrc, &c->dev_bytes);
    HIPCHECK(hipMemcpy(c->cent_d.p, cent, ntot * 10 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(c->centofs_d.p, cent_ofs, ((size_t)nsrc + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(c->moment_d.p, moment, (size_t)nsrc * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(c->risetime_d.p, risetime, (size_t)nsrc * sizeof(float), hipMemcpyHostToDevice));
    c->nsrc = nsrc;
    c->src_status.assign((size_t)nsrc, 0);
    c->any_failed = false;
    c->evaluated.assign((size_t)nsrc, 0);
    if (c->prepared) {                 // results of the new batch have their place before anything reads them
        c->misfit_d.ensure((size_t)nsrc * c->nmis, &c->dev_bytes);
        c->global_d.ensure((size_t)nsrc, &c->dev_bytes);
    }
    if (fold_halfwidth(maxrise, c->gm.dt) != fold_halfwidth(c->max_risetime, c->gm.dt)) c->prepared = false;
    c->max_risetime = maxrise;
    if (c->synth_only || c->any_untapered) c->prepared = false;   // windows follow the natural spans of the uploaded sources
    c->last_nsrc = 0;
    c->proc_which_held = 0;
    c->fft_ready = false;
    return 0;
    GUARD_END(c)
}

// CPUs this process may actually keep busy: the hardware threads it is allowed on, cut to the cgroup's CPU quota (a
// container sees every hardware thread of the machine but is throttled at its quota -- on the GPU boxes 16 CPUs of 256:
// more discretiser threads than that do not finish sooner, they are stopped for the rest of each scheduling period)
static int effective_cpus()
{
    static const int n = [] {
        int hw = std::max(1, omp_get_num_procs());
        double quota = 0.0;
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota|max> <period>"
            char q[64]; long long per = 0;
            if (std::fscanf(f, "%63s %lld", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) quota = std::atof(q) / (double)per;
            std::fclose(f);
        } else {
            long long q = -1, per = 0;
            if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lld", &q) != 1) q = -1; std::fclose(g); }
            if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lld", &per) != 1) per = 0; std::fclose(g); }
            if (q > 0 && per > 0) quota = (double)q / (double)per;
        }
        if (quota > 0.0) hw = std::min(hw, std::max(1, (int)std::floor(quota + 0.5)));
        return hw;
    }();
    return n;
}

int kiwi_hip_effective_cpus(void) { return effective_cpus(); }

int kiwi_hip_eikonal_cache_stats(long long *hits, long long *misses, int reset)
{
    eik::SolveCache &sc = eik::SolveCache::get();
    if (hits) *hits = sc.hits.load();
    if (misses) *misses = sc.misses.load();
    if (reset) { sc.hits = 0; sc.misses = 0; sc.miss_streak = 0; sc.probe = 0; }
    if (reset & 2) { std::lock_guard<std::mutex> lk(sc.mu); sc.slots.clear(); }
    return 0;
}

int kiwi_hip_fast_marching(const float *speed, int nx, int ny, const float *origin, const float *delta, const float *start,
                           float discard, int plain, float *times, long long *fallbacks)
{
    try {
        if (!speed || !origin || !delta || !start || !times || nx < 1 || ny < 1) return 2;
        std::vector<float> t;
        if (plain) eik::fast_marching_plain(speed, nx, ny, origin, delta, start, t, discard);
        else eik::fast_marching(speed, nx, ny, origin, delta, start, t, discard);
        std::memcpy(times, t.data(), t.size() * sizeof(float));
        if (fallbacks) *fallbacks = eik::fmm_fallbacks().load();
        return 0;
    } catch (...) { return 1; }
}

// One batch of trial sources after the host discretiser, before anything touches the device
struct HostBatch {
    int nsrc = 0, nbad = 0, bad = -1;
    std::string why;
    std::vector<int> ofs, status;
    std::vector<float> cent, mom, rise;
};

// psm_set + psm_to_tdsm of every source of the batch (minimizer_engine.f90:500-523), host only: reads the context's
// discretiser settings (effective dt, crust profile, constraints), writes nothing of it -- so the next piece of a trial list
// can be discretised while the device evaluates the present one (kiwi_hip_misfits_for_params).  `spare` = CPUs left to
// the caller's other threads.
static void discretise_batch(const kiwi_hip_ctx *c, int sourcetype, int nsrc, const float *params, int spare, HostBatch &hb)
{
    const int np = nparams_any(sourcetype);
    if (np < 0) throw std::runtime_error("source type not supported by the host discretiser");
    if (nsrc < 1) throw std::runtime_error("need at least one source");
    const bool eikonal = source_nparams_eikonal(sourcetype) > 0;
    if (eikonal && !c->have_crust) throw std::runtime_error("eikonal sources need the crust profiles (kiwi_hip_set_source_crust)");
    std::vector<DiscreteSource> ds((size_t)nsrc);
    hb.nsrc = nsrc;
    hb.status.assign((size_t)nsrc, 0);
    hb.nbad = 0; hb.bad = -1;
    hb.why = "source discretisation failed";
    // a few threads only: the discretisers take microseconds per source, and idle OpenMP workers spin for their
    // block time after the loop, competing with the HIP runtime's own threads for the caller's next calls.
    // The eikonal discretisers run a fast-marching solve per source (cfg4: a 25 m fine grid of 1200 x 360 points, 43 ms per
    // solve on a core of the GPU box): one thread per source pays, up to the CPUs this process really has -- measured
    // there, 128 solves: 5.5 s on one thread, 0.70 s on 8, 0.37 s on 16 = the container's CPU quota, 0.43-0.54 s on 32-128.
    int ecap = std::max(1, (effective_cpus() - std::max(0, spare)) / std::max(1, c->cpu_share));
    if (const char *m = std::getenv("KIWI_HIP_DISC_THREADS")) ecap = std::max(1, std::atoi(m));
    const int nthreads = std::max(1, eikonal ? std::min({ omp_get_max_threads(), nsrc, ecap })
                                             : std::min({ omp_get_max_threads(), (nsrc + 31) / 32, 16, ecap }));
    (void)nthreads;
    // Eikonal types: trial sources that differ from an earlier one of the batch only in moment (factor) and rise time have
    // the same rupture -- psm%moment and psm%risetime do not enter psm_to_tdsm (source_eikonal.f90:228-229,
    // source_mt_eikonal.f90:243-244) -- so the fast-marching solve is done once per distinct rupture and the table copied.
    std::vector<int> solve_of((size_t)nsrc);
    for (int s = 0; s < nsrc; s++) solve_of[s] = s;
    if (eikonal) {
        const int imom = 4, irise = sourcetype == 5 ? 19 : 14;
        std::map<std::vector<unsigned int>, int> seen;
        std::vector<unsigned int> key((size_t)np);
        for (int s = 0; s < nsrc; s++) {
            std::memcpy(key.data(), params + (size_t)s * np, (size_t)np * sizeof(float));
            key[imom] = 0; key[irise] = 0;
            auto it = seen.find(key);
            if (it == seen.end()) seen.emplace(key, s); else solve_of[s] = it->second;
        }
    }
    const float edt = c->effective_dt;
    int nbad = 0, bad = -1;
    std::string why = hb.why;
    std::vector<int> &status = hb.status;
    if (eikonal) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
        for (int s = 0; s < nsrc; s++) {
            if (solve_of[s] != s) continue;
            const std::string err = discretize_eikonal(sourcetype, params + (size_t)s * np, edt, c->rupture_profile, c->constraints, ds[s]);
            if (!err.empty()) {
                status[s] = err[0] == 'E' ? 5 : 6;
                ds[s].centroids.clear(); ds[s].moment = 0.f; ds[s].risetime = 0.f;
#pragma omp critical
                { nbad++; if (bad < 0 || s < bad) { bad = s; why = err; } }
            }
        }
    } else {
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
        for (int s = 0; s < nsrc; s++) {
            if (!discretize(sourcetype, params + (size_t)s * np, edt, ds[s])) {
                status[s] = 3;
                ds[s].centroids.clear(); ds[s].moment = 0.f; ds[s].risetime = 0.f;
#pragma omp critical
                { nbad++; if (bad < 0 || s < bad) bad = s; }
            }
        }
    }
    for (int s = 0; s < nsrc; s++)
        if (solve_of[s] != s) {                  // same rupture as an earlier source: its table, own moment and rise time
            const int f = solve_of[s];
            status[s] = status[f];               // (its table is taken from ds[f] when the batch is packed: no copy per duplicate)
            if (status[f] == 0) { ds[s].moment = params[(size_t)s * np + 4]; ds[s].risetime = params[(size_t)s * np + (sourcetype == 5 ? 19 : 14)]; }
            else { ds[s].moment = 0.f; ds[s].risetime = 0.f; nbad++; }
        }
    hb.nbad = nbad; hb.bad = bad; hb.why = why;
    // wrong type / parameter count is the caller's error for the whole batch; a source the discretiser rejects
    // ("Empty rupture area", ...) is recorded and skipped like seismosizer.py:703-720 does (failings)
    if (bad >= 0 && !eikonal) throw std::runtime_error(why + " (source " + std::to_string(bad + 1) + ")");
    hb.ofs.assign((size_t)nsrc + 1, 0);
    auto table_of = [&](int s) -> const std::vector<Centroid> & { return ds[solve_of[s]].centroids; };
    for (int s = 0; s < nsrc; s++) hb.ofs[s + 1] = hb.ofs[s] + (int)table_of(s).size();
    hb.cent.resize((size_t)hb.ofs[nsrc] * 10); hb.mom.resize((size_t)nsrc); hb.rise.resize((size_t)nsrc);
    for (int s = 0; s < nsrc; s++) {
        std::memcpy(hb.cent.data() + (size_t)hb.ofs[s] * 10, table_of(s).data(), table_of(s).size() * sizeof(Centroid));
        hb.mom[s] = ds[s].moment; hb.rise[s] = ds[s].risetime;
    }
}

// the discretised batch becomes the context's uploaded sources; throws when no source of it could be discretised
static void upload_batch(kiwi_hip_ctx *c, const HostBatch &hb)
{
    const int nsrc = hb.nsrc;
    if (kiwi_hip_set_sources(c, nsrc, hb.ofs.data(), hb.cent.data(), hb.mom.data(), hb.rise.data())) throw std::runtime_error(c->err);
    if (hb.nbad > 0) {
        c->src_status = hb.status;
        c->any_failed = true;
        c->status_d.ensure((size_t)nsrc, &c->dev_bytes);
        HIPCHECK(hipMemcpy(c->status_d.p, hb.status.data(), (size_t)nsrc * sizeof(int), hipMemcpyHostToDevice));
        // no source of the batch could be discretised (a batch of one: the reference's `set_source_params: nok >`)
        if (hb.nbad == nsrc) throw std::runtime_error(hb.why + " (source " + std::to_string(hb.bad + 1) + ")");
    }
}

int kiwi_hip_set_sources_params(kiwi_hip_ctx *c, int sourcetype, int nsrc, const float *params)
{
    GUARD_BEGIN
    HostBatch hb;
    discretise_batch(c, sourcetype, nsrc, params, 0, hb);
    upload_batch(c, hb);
    return 0;
    GUARD_END(c)
}

static const char *status_message(int code)
{
    switch (code) {
    case 0: return "";
    case 5: return "Empty rupture area";                                              // source_eikonal.f90:286
    case 6: return "position of nucleation point is outside of rupture region";       // source_eikonal.f90:428
    default: return "source discretisation failed";
    }
}

int kiwi_hip_get_source_status(kiwi_hip_ctx *c, int isrc0, int nsrc, int *status)
{
    GUARD_BEGIN
    if (!c) return fail(nullptr, "null context");
    if (!status) throw std::runtime_error("null argument");
    if (isrc0 < 0 || nsrc < 0 || isrc0 + nsrc > c->nsrc) throw std::runtime_error("source range out of bounds");
    for (int s = 0; s < nsrc; s++) status[s] = c->src_status[(size_t)isrc0 + s];
    return 0;
    GUARD_END(c)
}

int kiwi_hip_source_status_message(int code, char *buf, int buflen)
{
    if (!buf || buflen <= 0) return 1;
    std::snprintf(buf, (size_t)buflen, "%s", status_message(code));
    return 0;
}

// psm_params_norm_* (source_bilat.f90:45-46, source_circular.f90:44-45, source_point_lp.f90:54-55, source_eikonal.f90:48-49,
// source_mt_eikonal.f90:48-50, source_moment_tensor.f90:42-43)
static const std::vector<float> &params_norm(int sourcetype)
{
    static const std::vector<float> none, norm[7] = {
        {},
        { 1.f, 10000.f, 10000.f, 10000.f, 7e18f, 360.f, 90.f, 360.f, 360.f, 10000.f, 10000.f, 10000.f, 3000.f, 1.f },
        { 1.f, 10000.f, 10000.f, 10000.f, 7e18f, 360.f, 90.f, 360.f, 10000.f, 3000.f, 1.f },
        { 1.f, 10000.f, 10000.f, 10000.f, 7e18f, 1.f, 0.f, -1.f, 1.f, 1.f, 1.f, 20.f, 1.f },
        { 1.f, 10000.f, 10000.f, 10000.f, 7e18f, 360.f, 90.f, 360.f, 10000.f, 10000.f, 10000.f, 360.f, 10000.f, 1.f, 1.f },
        { 1.f, 10000.f, 10000.f, 10000.f, 7e18f, 360.f, 90.f, 10000.f, 10000.f, 10000.f, 360.f, 10000.f, 1.f, 7e18f, 7e18f,
          7e18f, 7e18f, 7e18f, 7e18f, 1.f },
        { 1.f, 10000.f, 10000.f, 10000.f, 7e18f, 7e18f, 7e18f, 7e18f, 7e18f, 7e18f, 1.f },
    };
    return sourcetype >= 1 && sourcetype <= 6 ? norm[sourcetype] : none;
}

int kiwi_hip_lmdif(kiwi_hip_residual_fn fcn, void *user, int m, int n, float *x, float *fvec, float ftol, float xtol, float gtol,
                   int maxfev, float epsfcn, float *diag, int mode, float factor, int *info, int *nfev)
{
    if (!fcn || !x || !fvec || !diag || !info || !nfev) return -1;
    try {
        lm::Fcn f = [&](int k, float *xs, float *fv) { return fcn(user, k, m, n, xs, fv); };
        *info = lm::lmdif(f, m, n, x, fvec, ftol, xtol, gtol, maxfev, epsfcn, diag, mode, factor, *nfev);
        return 0;
    } catch (...) { return -1; }
}

// minimize_lm (minimizer_engine.f90:728-874): lmdif over the masked, normalised parameters; every forward step is
// lm_forward_step (:806-872): clamp to the limits with a penalty factor on the residuals, psm_set_subparams through the
// normalised copy of ALL parameters (source_all.f90:377-425, so unmasked ones go through (p / norm) * norm as well),
// update_misfits.  The n forward steps of a Jacobian are one batch on the device.
int kiwi_hip_minimize_lm(kiwi_hip_ctx *c, int sourcetype, float *params, const int *mask, const float *mins, const float *maxs,
                         int *info, int *iterations, float *misfit, float *best)
{
    GUARD_BEGIN
    if (!params || !mask || !info || !iterations || !misfit) throw std::runtime_error("null argument");
    const int np = nparams_any(sourcetype);
    if (np < 0) throw std::runtime_error("source type not supported by the host discretiser");
    const std::vector<float> &norm = params_norm(sourcetype);
    std::vector<int> idx;
    for (int i = 0; i < np; i++) if (mask[i]) idx.push_back(i);
    const int n = (int)idx.size();
    int m = 0;
    if (int rc = kiwi_hip_nmisfits(c, &m)) return rc;
    if (n <= 0 || m < n) throw std::runtime_error("minimize_lm needs at least one free parameter and at least as many misfits");
    if ((mins == nullptr) != (maxs == nullptr)) throw std::runtime_error("parameter limits need both minima and maxima");
    std::vector<float> cur(params, params + np), x(n), fvec(m), diag(n, 1.0f), rows, mis, glob;
    int nsteps = 0;
    float last_global = 0.0f;
    lm::Fcn fcn = [&](int k, float *xs, float *fv) -> int {
        rows.resize((size_t)k * np); mis.resize((size_t)k * m); glob.resize(k);
        std::vector<float> factor(k);
        for (int s = 0; s < k; s++) {
            float *sub = xs + (size_t)s * n;
            float penalty = 0.0f;
            if (mins)
                for (int i = 0; i < n; i++) {
                    const float nrm = norm[idx[i]];
                    if (sub[i] * nrm < mins[i]) {
                        penalty = penalty + fabsf(sub[i] * nrm - mins[i]) / fabsf(maxs[i] - mins[i]);
                        sub[i] = mins[i] / nrm;
                    }
                    if (sub[i] * nrm > maxs[i]) {
                        penalty = penalty + fabsf(sub[i] * nrm - maxs[i]) / fabsf(maxs[i] - mins[i]);
                        sub[i] = maxs[i] / nrm;
                    }
                }
            factor[s] = 1.0f + penalty;
            std::vector<float> copy(np);
            for (int i = 0; i < np; i++) copy[i] = cur[i] / norm[i];
            for (int i = 0; i < n; i++) copy[idx[i]] = sub[i];
            for (int i = 0; i < np; i++) cur[i] = copy[i] * norm[i];
            std::copy(cur.begin(), cur.end(), rows.begin() + (size_t)s * np);
        }
        if (kiwi_hip_set_sources_params(c, sourcetype, k, rows.data())) return -2;
        for (int s = 0; s < k; s++)                 // update_misfits( ok ) false -> iflag = -2, minimizer_engine.f90:853-857
            if (c->src_status[s]) { c->err = std::string(status_message(c->src_status[s])) + " (forward step " + std::to_string(nsteps + s + 1) + ")"; return -2; }
        if (kiwi_hip_eval(c, 0, k)) return -2;
        if (kiwi_hip_get_misfits(c, 0, k, mis.data(), nullptr, glob.data())) return -2;
        for (int s = 0; s < k; s++)
            for (int i = 0; i < m; i++) fv[(size_t)s * m + i] = mis[(size_t)s * m + i] * factor[s];
        nsteps += k;
        last_global = glob[k - 1];
        return 0;
    };
    for (int i = 0; i < n; i++) x[i] = cur[idx[i]] / norm[idx[i]];
    const float tol = sqrtf(lm::kEpsMch);
    int nfev = 0;
    int rc = lm::lmdif(fcn, m, n, x.data(), fvec.data(), tol, tol, 0.0f, 500 * (n + 1), 0.0f, diag.data(), 2, 0.01f, nfev);
    if (rc == -2) return -1;                         // the engine call has set the error text
    if (rc == 8) rc = 4;
    *info = rc; *iterations = nsteps; *misfit = last_global;
    std::copy(cur.begin(), cur.end(), params);       // the source the engine is left with: the LAST forward step
    if (best) {
        std::vector<float> copy(np);
        for (int i = 0; i < np; i++) copy[i] = cur[i] / norm[i];
        for (int i = 0; i < n; i++) copy[idx[i]] = x[i];
        for (int i = 0; i < np; i++) best[i] = copy[i] * norm[i];
    }
    return 0;
    GUARD_END(c)
}

int kiwi_hip_eval(kiwi_hip_ctx *c, int isrc0, int nsrc)
{
    GUARD_BEGIN
    return eval_impl(c, isrc0, nsrc, c->keep_which);
    GUARD_END(c)
}

int kiwi_hip_set_keep_synthetics(kiwi_hip_ctx *c, int which)
{
    if (which < 0 || which > 3) return fail(c, "which must be 0 (off), 1 (plain), 2 (tapered) or 3 (filtered)");
    c->keep_which = which;
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_keep_synthetics(m, which); });
}

int kiwi_hip_set_arithmetic(kiwi_hip_ctx *c, int mode)
{
    if (mode != KIWI_ARITH_EXACT && mode != KIWI_ARITH_FUSED) return fail(c, "arithmetic: 0 (exact) or 1 (fused)");
    c->arith = mode;
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_set_arithmetic(m, mode); });
}

int kiwi_hip_get_arithmetic(kiwi_hip_ctx *c, int *mode)
{
    *mode = c->arith;
    return 0;
}

int kiwi_hip_sync(kiwi_hip_ctx *c)
{
    GUARD_BEGIN_DEV(c)
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
    GUARD_END(c)
}

int kiwi_hip_nmisfits(kiwi_hip_ctx *c, int *nmis)
{
    GUARD_BEGIN
    HIPCHECK(hipSetDevice(c->device));
    prepare(c);
    *nmis = c->nmis;
    return 0;
    GUARD_END(c)
}

int kiwi_hip_get_misfits(kiwi_hip_ctx *c, int isrc0, int nsrc, float *misfit, float *norm, float *global)
{
    GUARD_BEGIN_DEV(c)
    if (!c->prepared) throw std::runtime_error("nothing evaluated yet");
    if (c->synth_only) throw std::runtime_error("misfits need a reference seismogram and a misfit taper for every enabled receiver component "
                                                "(the device comparator evaluates norms over the taper span, comparator.f90:782-792)");
    if (isrc0 < 0 || nsrc < 0 || isrc0 + nsrc > c->nsrc) throw std::runtime_error("source range out of bounds");
    for (int s = isrc0; s < isrc0 + nsrc; s++)
        if ((size_t)s >= c->evaluated.size() || !c->evaluated[s])
            throw std::runtime_error("nothing evaluated yet for source " + std::to_string(s + 1) + " of the uploaded batch");
    HIPCHECK(hipStreamSynchronize(c->stream));
    if (misfit)
        HIPCHECK(hipMemcpy(misfit, c->misfit_d.p + (size_t)isrc0 * c->nmis, (size_t)nsrc * c->nmis * sizeof(float), hipMemcpyDeviceToHost));
    if (norm && c->untapered_fft)        // (the norm factors of un-tapered pairs are made on the device: pair_refamp_kernel / pair_reffilt_kernel)
        HIPCHECK(hipMemcpy(c->norm_src_h.data() + (size_t)isrc0 * c->nmis, c->normsrc_d.p + (size_t)isrc0 * c->nmis, (size_t)nsrc * c->nmis * sizeof(float),
                           hipMemcpyDeviceToHost));
    if (norm)
        for (int s = 0; s < nsrc; s++) {
            if (c->src_status[(size_t)isrc0 + s]) std::memset(norm + (size_t)s * c->nmis, 0, (size_t)c->nmis * sizeof(float));
            else if (c->fft_needed)      // spectral norms / filtered traces: the norm factor follows the pair's transform length
                std::memcpy(norm + (size_t)s * c->nmis, c->norm_src_h.data() + ((size_t)isrc0 + s) * c->nmis, (size_t)c->nmis * sizeof(float));
            else std::memcpy(norm + (size_t)s * c->nmis, c->norm_h.data(), (size_t)c->nmis * sizeof(float));
        }
    if (global)
        HIPCHECK(hipMemcpy(global, c->global_d.p + isrc0, (size_t)nsrc * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
    GUARD_END(c)
}

int kiwi_hip_get_global_misfits_device(kiwi_hip_ctx *c, int isrc0, int nsrc, const float **device_ptr)
{
    GUARD_BEGIN_DEV(c)
    if (!device_ptr) throw std::runtime_error("null argument");
    if (!c->prepared || c->synth_only) throw std::runtime_error("nothing evaluated yet");
    if (isrc0 < 0 || nsrc < 0 || isrc0 + nsrc > c->nsrc) throw std::runtime_error("source range out of bounds");
    for (int s = isrc0; s < isrc0 + nsrc; s++)
        if ((size_t)s >= c->evaluated.size() || !c->evaluated[s])
            throw std::runtime_error("nothing evaluated yet for source " + std::to_string(s + 1) + " of the uploaded batch");
    HIPCHECK(hipStreamSynchronize(c->stream));
    *device_ptr = c->global_d.p + isrc0;
    return 0;
    GUARD_END(c)
}

// make_misfits_for_sources for a whole trial list in one call (seismosizer.py:682-722), host and device overlapped: the
// list is cut into pieces; while the device evaluates one piece, a second host thread discretises the next.  Per piece
// the calls are exactly kiwi_hip_set_sources_params + kiwi_hip_eval + kiwi_hip_get_misfits + kiwi_hip_get_source_status,
// so results do not depend on `piece` (a source's evaluation does not depend on its batch: tests).
int kiwi_hip_misfits_for_params(kiwi_hip_ctx *c, int sourcetype, int nsrc, const float *params, int piece,
                                float *misfit, float *norm, float *global, int *status)
{
    GUARD_BEGIN
    const int np = nparams_any(sourcetype);
    if (np < 0) throw std::runtime_error("source type not supported by the host discretiser");
    if (nsrc < 1) throw std::runtime_error("need at least one source");
    HIPCHECK(hipSetDevice(c->device));
    prepare(c);
    if (c->synth_only) throw std::runtime_error("misfits need a reference seismogram and a misfit taper for every enabled receiver component "
                                                "(the device comparator evaluates norms over the taper span, comparator.f90:782-792)");
    const size_t nmis = (size_t)c->nmis;
    if (!c->mates.empty() && nsrc >= 2) {
        // ---- multi-device context: contiguous shards in list order (Source.grid order), shard 0 here -- this context keeps the
        // HEAD of the list as in the one-device case --, the others each in a thread of their own on their device
        const int ndev = std::min(nsrc, 1 + (int)c->mates.size());
        // (futures: their destructors join -- a std::thread that is still joinable when an exception unwinds this frame, e.g.
        // from the next emplace_back, would end the process in std::terminate)
        std::vector<std::future<int>> th;
        std::vector<int> rc((size_t)ndev, 0);
        auto bound = [&](int i) { return (int)((long long)nsrc * i / ndev); };
        for (int i = 1; i < ndev; i++) {
            kiwi_hip_ctx *m = c->mates[(size_t)i - 1];
            const int s0 = bound(i), n = bound(i + 1) - s0;
            th.push_back(std::async(std::launch::async, [=] {
                return kiwi_hip_misfits_for_params(m, sourcetype, n, params + (size_t)s0 * np, piece, misfit ? misfit + (size_t)s0 * nmis : nullptr,
                                                   norm ? norm + (size_t)s0 * nmis : nullptr, global ? global + s0 : nullptr, status ? status + s0 : nullptr);
            }));
        }
        std::vector<kiwi_hip_ctx *> keep;
        keep.swap(c->mates);                                  // (shard 0 through the one-device path of this very function)
        rc[0] = kiwi_hip_misfits_for_params(c, sourcetype, bound(1), params, piece, misfit, norm, global, status);
        keep.swap(c->mates);
        for (int i = 1; i < ndev; i++) rc[(size_t)i] = th[(size_t)i - 1].get();
        HIPCHECK(hipSetDevice(c->device));
        for (int i = 0; i < ndev; i++)
            if (rc[(size_t)i]) {
                if (i > 0) c->err = "device " + std::to_string(c->mates[(size_t)i - 1]->device) + ": " + c->mates[(size_t)i - 1]->err;
                return rc[(size_t)i];
            }
        return 0;
    }
    // (default piece: 128 eikonal solves keep the discretiser team busy for one device evaluation; for the closed-form source types the
    // host is a few per cent of a piece and larger launches fill the device better -- cfg3, 4096 trials: 31.1 k evals/s at 1024, 32.5 k at 2048)
    if (piece <= 0) piece = source_nparams_eikonal(sourcetype) > 0 ? 128 : 2048;
    // pieces [first, first + count) in list order.  Eikonal types: the LAST piece of the list -- the first one worked on, the one
    // whose discretisation nothing hides -- is cut into a ramp of an eighth, an eighth, a quarter and half a piece, so that the
    // device starts after an eighth of a piece's fast-marching solves (one per discretiser thread at the default 128 on 16 CPUs)
    // instead of a whole one (512 cfg4-nukl trials: 137 ms of the call's 1066 were that wait).
    std::vector<std::pair<int, int>> pieces;
    for (int s0 = 0; s0 < nsrc; s0 += piece) pieces.emplace_back(s0, std::min(piece, nsrc - s0));
    if (source_nparams_eikonal(sourcetype) > 0 && pieces.size() >= 2 && pieces.back().second >= 8) {
        const std::pair<int, int> last = pieces.back();
        pieces.pop_back();
        const int e = last.second / 8, q = last.second / 4, h = last.second - 2 * e - q;
        pieces.emplace_back(last.first, h);
        pieces.emplace_back(last.first + h, q);
        pieces.emplace_back(last.first + h + q, e);
        pieces.emplace_back(last.first + h + q + e, e);
    }
    const int npieces = (int)pieces.size();
    auto work = [c, sourcetype, np, params, npieces, &pieces](int k) {
        HostBatch hb;
        discretise_batch(c, sourcetype, pieces[(size_t)k].second, params + (size_t)pieces[(size_t)k].first * np, npieces > 1 ? 1 : 0, hb);
        return hb;
    };
    // last piece first: the context is left with the HEAD of the list (sources 0 .. piece - 1), its source 0 the list's.
    // The discretiser runs AHEAD of the device by up to kAhead pieces (round 6; until then it started a piece when the device
    // started the one before, and idled once it was done: with the eikonal types' solves at 0.6 of a piece's device time the
    // device still waited for the host after every short piece of the ramp).
    struct Ahead {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<HostBatch> q;
        std::exception_ptr err;
        bool stop = false;
    } ahead;
    constexpr size_t kAhead = 3;
    std::future<void> producer = std::async(std::launch::async, [&] {
        try {
            for (int k = npieces - 1; k >= 0; k--) {
                HostBatch hb = work(k);
                std::unique_lock<std::mutex> lk(ahead.mu);
                ahead.cv.wait(lk, [&] { return ahead.q.size() < kAhead || ahead.stop; });
                if (ahead.stop) return;
                ahead.q.push_back(std::move(hb));
                ahead.cv.notify_all();
            }
        } catch (...) {
            std::lock_guard<std::mutex> lk(ahead.mu);
            ahead.err = std::current_exception();
            ahead.cv.notify_all();
        }
    });
    struct StopProducer {                          // (an exception below must not leave the producer waiting for room in the queue)
        Ahead &a; std::future<void> &f;
        ~StopProducer() { { std::lock_guard<std::mutex> lk(a.mu); a.stop = true; } a.cv.notify_all(); if (f.valid()) f.wait(); }
    } stop_producer{ ahead, producer };
    // KIWI_HIP_TRACE_PIECES=1: per piece on stderr -- ms waited for its discretisation, ms of upload, ms of evaluation + download
    static const bool trace = [] { const char *e = std::getenv("KIWI_HIP_TRACE_PIECES"); return e && std::atoi(e) != 0; }();
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int k = npieces - 1; k >= 0; k--) {
        const double t_wait = now();
        HostBatch hb;
        {
            std::unique_lock<std::mutex> lk(ahead.mu);
            ahead.cv.wait(lk, [&] { return !ahead.q.empty() || ahead.err; });
            if (ahead.q.empty()) std::rethrow_exception(ahead.err);
            hb = std::move(ahead.q.front());
            ahead.q.pop_front();
            ahead.cv.notify_all();
        }
        const double t_got = now();
        const int s0 = pieces[(size_t)k].first, n = hb.nsrc;
        if (status) std::memcpy(status + s0, hb.status.data(), (size_t)n * sizeof(int));
        if (hb.nbad == n) {                       // nothing of this piece to evaluate: every trial of it is a failing
            if (misfit) std::memset(misfit + (size_t)s0 * nmis, 0, (size_t)n * nmis * sizeof(float));
            if (norm) std::memset(norm + (size_t)s0 * nmis, 0, (size_t)n * nmis * sizeof(float));
            if (global) std::memset(global + s0, 0, (size_t)n * sizeof(float));
            continue;
        }
        upload_batch(c, hb);
        const double t_up = now();
        eval_impl(c, 0, n, c->keep_which);
        if (kiwi_hip_get_misfits(c, 0, n, misfit ? misfit + (size_t)s0 * nmis : nullptr, norm ? norm + (size_t)s0 * nmis : nullptr,
                                 global ? global + s0 : nullptr)) throw std::runtime_error(c->err);
        if (trace) std::fprintf(stderr, "kiwi_hip piece [%d, %d): waited %.1f ms for the discretiser, upload %.1f ms, evaluation + download %.1f ms\n",
                                s0, s0 + n, t_got - t_wait, t_up - t_got, now() - t_up);
    }
    return 0;
    GUARD_END(c)
}

int kiwi_hip_get_synthetics(kiwi_hip_ctx *c, int isrc, int irec, int icomp, int which, int *first, int *n,
                            float *out, int maxn)
{
    GUARD_BEGIN_DEV(c)
    if (which < 1 || which > 3) throw std::runtime_error("which must be 1 (plain), 2 (tapered) or 3 (filtered)");
    if (which == 3) { prepare(c); if (!c->any_filter || c->method == KIWI_AMPSPEC_L2NORM || c->method == KIWI_AMPSPEC_L1NORM) throw std::runtime_error("filtered synthetics need a misfit filter and a time-domain norm"); }
    if (isrc < 0 || isrc >= c->nsrc) throw std::runtime_error("source index out of range");
    if (irec < 1 || irec > (int)c->recv.size()) throw std::runtime_error("receiver index out of range");
    // served from the retained chunk when kiwi_hip_set_keep_synthetics(which) was on during the
    // last eval of this source; otherwise this one source is re-evaluated keeping them
    size_t srcofs;
    if (c->prepared && c->proc_which_held == which && isrc >= c->proc_chunk0 && isrc < c->proc_chunk0 + c->proc_chunkn) {
        srcofs = (size_t)(isrc - c->proc_chunk0) * c->syn_stride;
    } else {
        eval_impl(c, isrc, 1, which);
        srcofs = 0;
    }
    HIPCHECK(hipStreamSynchronize(c->stream));
    int slot = -1, k = 0;
    for (size_t i = 0; i < c->comps.size(); i++) {
        if (c->comps[i].rec == irec - 1) { if (k == icomp - 1) { slot = (int)i; break; } k++; }
    }
    if (slot < 0) throw std::runtime_error("receiver disabled or component index out of range");
    const CompDev &cd = c->comps[slot];
    *first = cd.w0; *n = cd.wlen;
    const int m = std::min(cd.wlen, maxn);
    HIPCHECK(hipMemcpy(out, c->proc_d.p + srcofs + cd.synofs + cd.halo, (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
    GUARD_END(c)
}

// get_peak_amplitudes (minimizer_engine.f90:1174-1212) / get_arias_intensities (:1214-1246) of uploaded source isrc
static int shake_impl(kiwi_hip_ctx *c, int isrc, int kind, float *out)
{
    if (!out) throw std::runtime_error("null argument");
    if (isrc < 0 || isrc >= c->nsrc) throw std::runtime_error("source index out of range");
    prepare(c);
    if (c->any_filter) throw std::runtime_error("peak amplitudes / Arias intensities are not available with a misfit filter set");
    std::vector<ShakeRec> recs;
    bool spans = false;
    for (size_t ir = 0; ir < c->recv.size(); ir++) {
        const Receiver &r = c->recv[ir];
        if (!r.enabled) continue;
        int iver = 0, ih1 = 0, ih2 = 0;                           // get_component_ids, receiver.f90:512-542
        for (int k = 0; k < r.ncomp; k++) {
            const int ict = std::abs(r.comp[k]);
            if (ict == 1) ih1 = k + 1;
            if (ict == 2) ih2 = k + 1;
            if (ict == 3) iver = k + 1;
        }
        if (ih1 == 0 || ih2 == 0)
            for (int k = 0; k < r.ncomp; k++) {
                const int ict = std::abs(r.comp[k]);
                if (ict == 4) ih1 = k + 1;
                if (ict == 5) ih2 = k + 1;
            }
        if (ih1 == 0 || ih2 == 0) { ih1 = 0; ih2 = 0; }
        ShakeRec sr;
        std::memset(&sr, 0, sizeof(sr));
        sr.rec = (int)ir;
        sr.untapered = r.taper.defined() ? 0 : 1;
        int slot0 = -1;
        for (size_t i = 0; i < c->comps.size(); i++) if (c->comps[i].rec == (int)ir) { slot0 = (int)i; break; }
        auto use = [&](int k1) { if (k1 && slot0 >= 0) sr.slot[sr.np++] = slot0 + k1 - 1; };
        if (kind != 3) { use(iver); use(ih1); use(ih2); }        // receiver_get_maxabs, receiver.f90:544-574
        else if (iver && ih1 && ih2) { use(iver); use(ih1); use(ih2); }     // receiver_get_arias_intensity, :576-594
        else if (ih1 && ih2) { use(ih1); use(ih2); }
        else if (iver) use(iver);
        if (sr.untapered && sr.np) spans = true;
        recs.push_back(sr);
    }
    if (recs.empty()) return 0;
    c->want_spansrc = spans;
    try { eval_impl(c, isrc, 1, 2); } catch (...) { c->want_spansrc = false; throw; }
    c->want_spansrc = false;
    DevBuf<ShakeRec> recs_d;
    DevBuf<float> out_d;
    recs_d.ensure(recs.size(), nullptr);
    out_d.ensure(recs.size(), nullptr);
    HIPCHECK(hipMemcpyAsync(recs_d.p, recs.data(), recs.size() * sizeof(ShakeRec), hipMemcpyHostToDevice, c->stream));
    std::vector<float> rise(1);
    HIPCHECK(hipMemcpyAsync(rise.data(), c->risetime_d.p + isrc, sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    hipLaunchKernelGGL(shake_kernel, dim3((unsigned)recs.size()), dim3(256), 0, c->stream, c->proc_d.p, c->comps_d.p, recs_d.p,
                       spans ? c->spansrc_d.p : (const int *)nullptr, kind, c->gm.dt, c->syn_factor, fold_halfwidth(rise[0], c->gm.dt), out_d.p);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(out, out_d.p, recs.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// output_seismogram_spectra (minimizer_engine.f90:1012-1039; probe_get_amp_spectrum, comparator.f90:333-354): the amplitude
// spectrum of one probe as the spectral comparator sees it -- |r2c| of the tapered window, transform length as
// probes_adjust_spans sizes the reference / synthetic pair; filtered = times the frequency filter where one is set.
// Runs the engine's own spectral pipeline once with the method switched to ampspec_l2norm for the duration of the call.
int kiwi_hip_get_amp_spectrum(kiwi_hip_ctx *c, int isrc, int irec, int icomp, int which_probe, int filtered, float *df, int *n,
                              float *out, int maxn)
{
    GUARD_BEGIN_DEV(c)
    if (!df || !n || !out) throw std::runtime_error("null argument");
    if (irec < 1 || irec > (int)c->recv.size()) throw std::runtime_error("receiver index out of range");
    if (which_probe && (isrc < 0 || isrc >= c->nsrc)) throw std::runtime_error("source index out of range");
    HIPCHECK(hipSetDevice(c->device));
    const int method0 = c->method;
    const Plf filter0 = c->recv[irec - 1].filter;
    c->method = KIWI_AMPSPEC_L2NORM;
    if (!filtered) c->recv[irec - 1].filter = Plf();              // plain: as if this receiver had no filter
    c->prepared = false;
    const bool fused0 = c->fused_fft;
    c->fused_fft = false;                                         // the spectrum itself is wanted: library transform into spec_d
    std::string err;
    try {
        prepare(c);
        if (c->synth_only) throw std::runtime_error("spectra need reference seismograms and misfit tapers");
        int slot = -1, k = 0;
        for (size_t i = 0; i < c->comps.size(); i++)
            if (c->comps[i].rec == irec - 1) { if (k == icomp - 1) { slot = (int)i; break; } k++; }
        if (slot < 0) throw std::runtime_error("receiver disabled or component index out of range");
        // the probe pair is sized for uploaded source isrc (the pair's transform length follows ITS strips); without a
        // source the reference probe keeps the span probe_set_array gave it: twice its data length, padded to a power of two
        FftPair pr;
        if (isrc >= 0 && isrc < c->nsrc) {
            eval_impl(c, isrc, 1, 0);
            pr = c->last_pairs[(size_t)slot];
        } else {
            if (c->comps[slot].untapered)
                throw std::runtime_error("the spectrum of an un-tapered reference follows the probe pair's span: name a source (set_source_params first)");
            if (!c->fft_ready) prepare_fft(c, c->reft_h);
            const CompDev &cd0 = c->comps[slot];
            const int ntr = next_pow2(std::max(2 * (cd0.rf1 - cd0.rf0 + 1), cd0.wlen));
            make_variants(c, { std::make_pair(slot, ntr) });
            const auto &v = c->variants[std::make_pair(slot, ntr)];
            pr = FftPair{ 0, 0, ntr, v.specofs, v.filtofs, slot };
        }
        HIPCHECK(hipStreamSynchronize(c->stream));
        const int nb = pr.ntrans / 2 + 1;
        *df = 1.f / ((float)pr.ntrans * c->gm.dt);
        *n = nb;
        if (nb > maxn) throw std::runtime_error("spectrum buffer too small");
        const bool has_filter = c->recv[irec - 1].filter.defined();
        if (which_probe) {
            std::vector<float2> z(nb);
            HIPCHECK(hipMemcpy(z.data(), c->spec_d.p + pr.spec_ofs, nb * sizeof(float2), hipMemcpyDeviceToHost));
            for (int i = 0; i < nb; i++) out[i] = hypotf(z[i].x, z[i].y) * ((filtered && has_filter) ? c->filtw_h[pr.specofs + i] : 1.f);
        } else if (c->comps[slot].untapered) {                   // the pair's reference spectrum (pair_refamp_kernel): |spec| x filter weights
            HIPCHECK(hipMemcpy(out, c->refpair_d.p + pr.spec_ofs, nb * sizeof(float), hipMemcpyDeviceToHost));
        } else {
            std::memcpy(out, c->refamp_h.data() + pr.specofs, nb * sizeof(float));     // |spec| x filter weights
        }
    } catch (const std::exception &e) { err = e.what(); }
    c->method = method0;
    c->recv[irec - 1].filter = filter0;
    c->fused_fft = fused0;
    c->prepared = false;
    if (!err.empty()) throw std::runtime_error(err);
    return 0;
    GUARD_END(c)
}

int kiwi_hip_principal_axes(int sourcetype, const float *params, float *pax, float *tax)
{
    if (sourcetype != KIWI_SRC_BILAT || !params || !pax || !tax) return -1;          // only psm_update_dep_params_bilat sets them (source_bilat.f90:233-237)
    principal_axes_bilat(params, pax, tax);
    return 0;
}

int kiwi_hip_get_peak_amplitudes(kiwi_hip_ctx *c, int isrc, int differentiate, float *out)
{
    GUARD_BEGIN
    if (differentiate != 1 && differentiate != 2)
        throw std::runtime_error("differentiate argument must be 1 for velocity or 2 for acceleration");
    return shake_impl(c, isrc, differentiate, out);
    GUARD_END(c)
}

int kiwi_hip_get_arias_intensities(kiwi_hip_ctx *c, int isrc, float *out)
{
    GUARD_BEGIN
    return shake_impl(c, isrc, 3, out);
    GUARD_END(c)
}

int kiwi_hip_shift_ref_seismogram(kiwi_hip_ctx *c, int irec, float shift)
{
    GUARD_BEGIN
    if (!c->have_db) throw std::runtime_error("no database set");
    if (irec < 1 || irec > (int)c->recv.size()) throw std::runtime_error("receiver index out of range");
    const int ishift = (int)std::lround(shift / c->gm.dt);                        // minimizer_engine.f90:373
    Receiver &r = c->recv[irec - 1];
    for (int k = 0; k < r.ncomp; k++) r.ref[k].first += ishift;                   // probe_shift, comparator.f90:273-288
    c->prepared = false;
    return forward(c, [&](kiwi_hip_ctx *m) { return kiwi_hip_shift_ref_seismogram(m, irec, shift); });
    GUARD_END(c)
}

// receiver_calculate_cross_correlations (receiver.f90:597-616): cc[k][q] = scalar product of the tapered synthetic of
// component k (host copy `syn` of one evaluated source block) with the reference shifted by lo + q samples inside its
// fixed taper (probes_windowed_cross_corr, comparator.f90:1061-1090; scalar_product_2, :627-637)
static void receiver_cross_correlations(kiwi_hip_ctx *c, int ir, const std::vector<float> &syn, int lo, int ns, std::vector<float> &cc)
{
    const Receiver &r = c->recv[ir];
    const float dt = c->gm.dt;
    int w[2];
    discrete_plf_span(r.taper, dt, w);
    const int wlen = w[1] - w[0] + 1;
    std::vector<float> tww(wlen, 1.f);
    plf_taper_array(r.taper, tww.data(), w[0], w[1], dt, IP_COS);
    cc.assign((size_t)ns * r.ncomp, 0.f);
    int slot0 = -1;
    for (size_t i = 0; i < c->comps.size(); i++) if (c->comps[i].rec == ir) { slot0 = (int)i; break; }
    for (int k = 0; k < r.ncomp; k++) {
        const CompDev &cd = c->comps[slot0 + k];
        const float *a = syn.data() + cd.synofs + cd.halo;               // tapered synthetic over the window
        const auto &rf = r.ref[k];
        const int f0 = rf.first, f1 = f0 + (int)rf.data.size() - 1;
        for (int q = 0; q < ns; q++) {
            const int sh = lo + q;
            double acc = 0.0;
            for (int t = w[0]; t <= w[1]; t++) {
                const int ts = t - sh;
                float b = 0.f;
                if (ts >= f0) b = (rf.data[std::min(ts, f1) - f0] * 1.f) * tww[t - w[0]];
                const float av = a[t - w[0]];
                acc += (c->syn_factor == 1.f) ? (double)(av * b) : (double)(av * c->syn_factor * b * 1.f);
            }
            cc[(size_t)k * ns + q] = (float)acc;
        }
    }
}

// output_cross_correlations (minimizer_engine.f90:1283-1306): the table of one receiver
int kiwi_hip_get_cross_correlations(kiwi_hip_ctx *c, int isrc, int irec, float min_shift, float max_shift, int *first_shift,
                                    int *nshift, float *cc_out, int maxn)
{
    GUARD_BEGIN_DEV(c)
    if (!first_shift || !nshift || !cc_out) throw std::runtime_error("null argument");
    if (irec < 1 || irec > (int)c->recv.size()) throw std::runtime_error("receiver index out of range");
    if (isrc < 0 || isrc >= c->nsrc) throw std::runtime_error("source index out of range");
    HIPCHECK(hipSetDevice(c->device));
    prepare(c);
    const Receiver &r = c->recv[irec - 1];
    if (c->synth_only || !r.taper.defined()) throw std::runtime_error("cross-correlations need reference seismograms and misfit tapers");
    const float dt = c->gm.dt;
    const int lo = (int)std::lround(min_shift / dt), hi = (int)std::lround(max_shift / dt), ns = hi - lo + 1;      // :1298
    if (ns < 1) throw std::runtime_error("empty shift range");
    *first_shift = lo; *nshift = ns;
    if (!r.enabled || r.ncomp == 0) { *nshift = 0; return 0; }
    if ((long long)ns * r.ncomp > maxn) throw std::runtime_error("cross-correlation buffer too small");
    eval_impl(c, isrc, 1, 2);
    HIPCHECK(hipStreamSynchronize(c->stream));
    std::vector<float> syn(c->syn_stride), cc;
    HIPCHECK(hipMemcpy(syn.data(), c->proc_d.p, c->syn_stride * sizeof(float), hipMemcpyDeviceToHost));
    receiver_cross_correlations(c, irec - 1, syn, lo, ns, cc);
    std::copy(cc.begin(), cc.end(), cc_out);
    return 0;
    GUARD_END(c)
}

int kiwi_hip_autoshift_ref_seismogram(kiwi_hip_ctx *c, int irec, float min_shift, float max_shift, int isrc, float *shifts)
{
    GUARD_BEGIN
    if (irec < 0 || irec > (int)c->recv.size()) throw std::runtime_error("receiver index out of range");
    if (isrc < 0 || isrc >= c->nsrc) throw std::runtime_error("source index out of range");
    HIPCHECK(hipSetDevice(c->device));
    prepare(c);
    if (c->synth_only) throw std::runtime_error("autoshift needs reference seismograms and misfit tapers");
    const float dt = c->gm.dt;
    const int lo = (int)std::lround(min_shift / dt), hi = (int)std::lround(max_shift / dt), ns = hi - lo + 1;   // :397
    if (ns < 1) throw std::runtime_error("empty shift range");
    // tapered, scaled synthetics of the current source (update_misfits, minimizer_engine.f90:393)
    eval_impl(c, isrc, 1, 2);
    HIPCHECK(hipStreamSynchronize(c->stream));
    std::vector<float> syn(c->syn_stride);
    HIPCHECK(hipMemcpy(syn.data(), c->proc_d.p, c->syn_stride * sizeof(float), hipMemcpyDeviceToHost));
    const int r0 = irec == 0 ? 0 : irec - 1, r1 = irec == 0 ? (int)c->recv.size() : irec;
    std::vector<int> applied;
    for (int ir = r0; ir < r1; ir++) {
        Receiver &r = c->recv[ir];
        int ishift = 0;
        if (r.enabled && r.ncomp > 0) {                                          // receiver.f90:816-832
            std::vector<float> cc;
            receiver_cross_correlations(c, ir, syn, lo, ns, cc);
            float ccmax = -std::numeric_limits<float>::infinity();
            for (float v : cc) ccmax = std::max(ccmax, v);
            // imax = maxloc( sum( max(cc / max(1, maxval(cc)), 0)**2, over components ) ), first maximum
            const float den = std::max(1.f, ccmax);
            int imax = 0;
            float best = 0.f;
            for (int q = 0; q < ns; q++) {
                float sum = 0.f;
                for (int k = 0; k < r.ncomp; k++) { const float x = std::max(cc[(size_t)k * ns + q] / den, 0.f); sum = sum + x * x; }
                if (q == 0 || sum > best) { best = sum; imax = q; }
            }
            ishift = lo + imax;
        }
        applied.push_back(ishift);
        shifts[ir - r0] = (float)ishift * dt;
    }
    for (int ir = r0; ir < r1; ir++) {
        Receiver &r = c->recv[ir];
        for (int k = 0; k < r.ncomp; k++) r.ref[k].first += applied[ir - r0];
    }
    c->prepared = false;
    // the other devices of a multi-device context take the shifts found here
    return forward(c, [&](kiwi_hip_ctx *m) {
        for (int ir = r0; ir < r1; ir++) {
            Receiver &r = m->recv[ir];
            for (int k = 0; k < r.ncomp; k++) r.ref[k].first += applied[ir - r0];
        }
        m->prepared = false;
        return 0;
    });
    GUARD_END(c)
}

int kiwi_hip_get_source_centroids(kiwi_hip_ctx *c, int isrc, int maxcent, int *ncent, float *cent)
{
    GUARD_BEGIN
    if (!ncent) throw std::runtime_error("null argument");
    if (isrc < 0 || isrc >= c->nsrc) throw std::runtime_error("source index out of range");
    HIPCHECK(hipSetDevice(c->device));
    const int c0 = c->cent_ofs[isrc], nc = c->cent_ofs[isrc + 1] - c0;
    *ncent = nc;
    if (cent && maxcent > 0) {
        if (nc > maxcent) throw std::runtime_error("centroid buffer too small");
        HIPCHECK(hipStreamSynchronize(c->stream));
        HIPCHECK(hipMemcpy(cent, c->cent_d.p + (size_t)c0 * 10, (size_t)nc * 10 * sizeof(float), hipMemcpyDeviceToHost));
    }
    return 0;
    GUARD_END(c)
}

int kiwi_hip_get_reference(kiwi_hip_ctx *c, int irec, int icomp, int which, int *first, int *n, float *out, int maxn)
{
    GUARD_BEGIN_DEV(c)
    if (which < 1 || which > 3) throw std::runtime_error("which must be 1 (plain), 2 (tapered) or 3 (filtered)");
    if (irec < 1 || irec > (int)c->recv.size()) throw std::runtime_error("receiver index out of range");
    const Receiver &r = c->recv[irec - 1];
    if (icomp < 1 || icomp > r.ncomp) throw std::runtime_error("component index out of range");
    if (which == 1) {                                  // the data as set (and shifted), comparator.f90:350-372
        const auto &rf = r.ref[icomp - 1];
        if (rf.data.empty()) throw std::runtime_error("no reference seismogram set");
        *first = rf.first; *n = (int)rf.data.size();
        std::memcpy(out, rf.data.data(), (size_t)std::min(*n, maxn) * sizeof(float));
        return 0;
    }
    HIPCHECK(hipSetDevice(c->device));
    prepare(c);
    if (c->synth_only) throw std::runtime_error("no reference seismogram set");
    int slot = -1, k = 0;
    for (size_t i = 0; i < c->comps.size(); i++) if (c->comps[i].rec == irec - 1) { if (k == icomp - 1) { slot = (int)i; break; } k++; }
    if (slot < 0) throw std::runtime_error("receiver disabled");
    const CompDev &cd = c->comps[slot];
    *first = cd.w0; *n = cd.wlen;
    const int m = std::min(cd.wlen, maxn);
    if (which == 2) {                                  // over the comparator window, tapered (:1173-1184)
        std::memcpy(out, c->reft_h.data() + cd.refofs, (size_t)m * sizeof(float));
    } else {                                           // filtered and cut to the taper (:1233-1263): device pipeline output
        if (!c->any_filter || c->method == KIWI_AMPSPEC_L2NORM || c->method == KIWI_AMPSPEC_L1NORM)
            throw std::runtime_error("filtered references need a misfit filter and a time-domain norm");
        if (c->nsrc == 0) throw std::runtime_error("no source set (the transform length follows the synthetics)");
        if (!c->slot_has_filter.empty() && c->fft_ready && !c->slot_has_filter[slot]) {
            std::memcpy(out, c->reft_h.data() + cd.refofs, (size_t)m * sizeof(float));      // no filter at this receiver
            return 0;
        }
        eval_impl(c, 0, 1, 0);             // the probe pair of the current source (source 0 of the batch) sizes the transform
        HIPCHECK(hipStreamSynchronize(c->stream));
        const FftPair pr = c->last_pairs[(size_t)slot];
        if (!c->slot_has_filter[slot]) std::memcpy(out, c->reft_h.data() + cd.refofs, (size_t)m * sizeof(float));
        else if (cd.untapered)             // the filtered reference of the PAIR (current source, slot), the part inside the window (pair_reffilt_kernel)
            HIPCHECK(hipMemcpy(out, c->reffilt_d.p + pr.filtofs, (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
        else std::memcpy(out, c->reffilt_h.data() + pr.filtofs, (size_t)m * sizeof(float));
    }
    return 0;
    GUARD_END(c)
}

int kiwi_hip_get_kernel_ms(kiwi_hip_ctx *c, float ms[4], int launches[3])
{
    GUARD_BEGIN_DEV(c)
    HIPCHECK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 4; i++) ms[i] = 0.f;
    for (int i = 0; i < 3; i++) launches[i] = 0;
    for (auto &ev : c->events) {
        float t = 0.f;
        HIPCHECK(hipEventElapsedTime(&t, ev.a, ev.b));
        ms[ev.kind] += t; ms[3] += t; launches[ev.kind]++;
    }
    // events are shared between neighbouring pairs: e0,e1,e2,e3 per chunk
    for (size_t i = 0; i < c->events.size(); i++) {
        if (c->events[i].kind == 0) c->event_pool.push_back(c->events[i].a);
        c->event_pool.push_back(c->events[i].b);
    }
    c->events.clear();
    return 0;
    GUARD_END(c)
}

int kiwi_hip_get_geometry(kiwi_hip_ctx *c, int isrc, int irec, int maxcent, int *ncent, void *records)
{
    GUARD_BEGIN_DEV(c)
    HIPCHECK(hipStreamSynchronize(c->stream));
    if (isrc < c->last_chunk0 || isrc >= c->last_chunk0 + c->last_chunkn)
        throw std::runtime_error("source not in the last evaluated chunk");
    if (irec < 1 || irec > (int)c->recv.size()) throw std::runtime_error("receiver index out of range");
    const int nrec = (int)c->recv.size();
    const int c0 = c->cent_ofs[isrc], nc = c->cent_ofs[isrc + 1] - c0, cb = c->cent_ofs[c->last_chunk0];
    *ncent = nc;
    const int m = std::min(nc, maxcent);
    const size_t base = (size_t)(c0 - cb) * nrec + (size_t)(irec - 1) * nc;
    HIPCHECK(hipMemcpy(records, c->recs_d.p + base, (size_t)m * sizeof(GeoRec), hipMemcpyDeviceToHost));
    return 0;
    GUARD_END(c)
}

int kiwi_hip_get_receiver_geometry(kiwi_hip_ctx *c, int irec, double *azi, double *bazi, double *dist)
{
    if (irec < 1 || irec > (int)c->recv.size()) return fail(c, "receiver index out of range");
    if (!c->have_origin) return fail(c, "no source location set");
    const Receiver &r = c->recv[irec - 1];
    *azi = r.azi0; *bazi = r.bazi0; *dist = r.dist0;
    return 0;
}

#ifndef KIWI_BUILD_EXTRA
#define KIWI_BUILD_EXTRA ""
#endif
// the extra compiler flags this library was built with (make EXTRA=...; "" for the default build)
int kiwi_hip_build_flags(char *buf, int buflen)
{
    if (!buf || buflen < 1) return 1;
    std::snprintf(buf, (size_t)buflen, "%s", KIWI_BUILD_EXTRA);
    return 0;
}

int kiwi_hip_get_device_bytes(kiwi_hip_ctx *c, long long *bytes)
{
    *bytes = c->dev_bytes;
    return 0;
}

// What a pure read reaches on this device (the ceiling the accumulate kernels' HBM-regime figure is held against: bench.py
// `also_hbm`): every lane 16 bytes per load, eight loads in flight, a workgroup walks its contiguous slice of a buffer far larger
// than the Infinity Cache; HIP events around `reps` passes on the context's stream.  The buffer is allocated and freed here.
int kiwi_hip_measure_read_bandwidth(kiwi_hip_ctx *c, long long bytes, int reps, double *gbs)
{
    GUARD_BEGIN_DEV(c)
    if (!gbs || bytes < (1ll << 20) || reps < 1) throw std::runtime_error("bad arguments");
    const size_t n16 = (size_t)bytes / 16;
    read_u4 *buf = nullptr;
    unsigned *sink = nullptr;
    HIPCHECK(hipMalloc(&buf, n16 * 16));
    hipError_t e = hipMalloc(&sink, 4096 * sizeof(unsigned));
    if (e != hipSuccess) { (void)hipFree(buf); HIPCHECK(e); }
    hipEvent_t a, b;
    HIPCHECK(hipEventCreate(&a)); HIPCHECK(hipEventCreate(&b));
    const int grid = 4096;
    hipLaunchKernelGGL(read_fill_kernel, dim3(grid), dim3(256), 0, c->stream, buf, n16);
    hipLaunchKernelGGL(read_in_order_kernel, dim3(grid), dim3(256), 0, c->stream, buf, n16, sink);      // warm-up
    HIPCHECK(hipEventRecord(a, c->stream));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(read_in_order_kernel, dim3(grid), dim3(256), 0, c->stream, buf, n16, sink);
    HIPCHECK(hipEventRecord(b, c->stream));
    HIPCHECK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(buf); (void)hipFree(sink);
    *gbs = ms > 0.f ? (double)n16 * 16.0 * reps / (ms * 1e-3) / 1e9 : 0.0;
    return 0;
    GUARD_END(c)
}

} // extern "C"
