// kiwi_host_eikonal.hpp -- host discretiser of the variable-rupture-speed sources `eikonal` and
// `mt_eikonal` (source_eikonal.f90:205-316,435-710; source_mt_eikonal.f90:200-323,442-762):
//   bounding circle clipped by the constraint half-spaces (geometry.f90:173-256) -> fine grid of
//   rupture speeds from a 1-D crustal profile (crust2x2.f90:170-195) -> fast-marching arrival
//   times (eikonal.f90:29-199 with the index heap of heap.f90: kiwi_host_fmm.hpp) -> coarse cells (mean time,
//   harmonic-mean speed, weight, duration) -> centroid table; the constant rise time is applied
//   after synthesis (psm%risetime).
// Sequential, default-real arithmetic in the reference's order; stays on the host.
#pragma once
#include "kiwi_host.hpp"
#include "kiwi_host_fmm.hpp"
#include <array>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <emmintrin.h>

namespace kiwi {

struct CrustProfile {                 // t_crust2x2_1d_profile, crust2x2.f90:45-50 (index 7 = below the crust)
    float vp[8], vs[8], rho[8], thickness[7];
};

struct HalfSpace { float point[3], normal[3]; };      // geometry.f90:25-28

// crust2x2_get_profile_averages (crust2x2.f90:146-168): total crustal thickness (ice .. lower crust)
inline float crust_thickness(const CrustProfile &p)
{
    float thi = 0.f;
    for (int i = 1; i < 7; i++) thi = thi + p.thickness[i];
    return thi;
}

namespace eik {

using V3 = std::array<float, 3>;

inline float dot(const float *a, const float *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
inline V3 mul(const float m[3][3], const V3 &v)
{
    V3 o;
    for (int i = 0; i < 3; i++) o[i] = (m[i][0] * v[0] + m[i][1] * v[1]) + m[i][2] * v[2];
    return o;
}
inline V3 mulT(const float m[3][3], const V3 &v)
{
    V3 o;
    for (int i = 0; i < 3; i++) o[i] = (m[0][i] * v[0] + m[1][i] * v[1]) + m[2][i] * v[2];
    return o;
}

inline bool inside(const V3 &p, const HalfSpace &h)                 // point_in_halfspace, geometry.f90:55-64
{
    const float d[3] = { h.point[0] - p[0], h.point[1] - p[1], h.point[2] - p[2] };
    return dot(h.normal, d) >= 0.f;
}

// get_piercingpoint, geometry.f90:66-118
struct Pierce { V3 point; bool between, a_inside; };
inline Pierce pierce(const V3 &a, const V3 &b, const HalfSpace &h)
{
    Pierce r;
    const float ab[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
    const float da[3] = { h.point[0] - a[0], h.point[1] - a[1], h.point[2] - a[2] };
    const float db[3] = { h.point[0] - b[0], h.point[1] - b[1], h.point[2] - b[2] };
    const float la = dot(h.normal, da), lb = dot(h.normal, db), lab = dot(h.normal, ab);
    const bool ain = la >= 0.f, bin = lb >= 0.f;
    r.a_inside = ain;
    r.between = (ain && !bin) || (bin && !ain);
    const bool parallel = lab * lab < dot(ab, ab) / 16777216.f;     // 2**digits(real)
    if (parallel && r.between) r.point = (std::fabs(la) <= std::fabs(lb)) ? a : b;
    else if (parallel) r.point = { 0.f, 0.f, 0.f };
    else for (int i = 0; i < 3; i++) r.point[i] = a[i] + ab[i] * la / lab;
    return r;
}

// trim_polygon (one half-space, then all of them), geometry.f90:192-256
inline std::vector<V3> clip(const std::vector<V3> &poly, const HalfSpace &h)
{
    const size_t n = poly.size();
    std::vector<Pierce> pr(n);
    for (size_t i = 0; i < n; i++) pr[i] = pierce(poly[i], poly[(i + 1) % n], h);
    std::vector<V3> out;
    for (size_t i = 0; i < n; i++) {
        if (pr[i].a_inside) out.push_back(poly[i]);
        if (pr[i].between) out.push_back(pr[i].point);
    }
    return out;
}

// ---- solves kept by their real inputs ------------------------------------------------------------------------------
// The arrival times are a pure function of (speed grid, its dimensions and spacing, the start cell): north / east / time
// shifts of a rupture and changes of its moment tensor leave all of these alone, a depth change alters the speed grid only
// through the layer boundaries it crosses.  A location grid search therefore repeats a handful of solves over and over.
// The cache compares the COMPLETE inputs (hash first, then memcmp of the speed grid), so a hit returns exactly the
// array the solver would produce: bit-identical centroid tables, whatever the hit rate.  Shared by the discretiser
// threads; a few entries (a solve's grids are a few MB each).  KIWI_HIP_EIK_CACHE=0 switches it off.
struct SolveCache {
    struct Entry {
        unsigned long long hash = 0;
        int nx = 0, ny = 0, ix = 0, iy = 0;
        float dx = 0.f, dy = 0.f;
        unsigned discard = 0;            // bits of the early-termination speed the solve was made with (a solve that stops early
                                         // leaves the nodes of that speed undone: another `discard` is another result)
        std::vector<float> speed, times;
        unsigned long long stamp = 0;
    };
    static constexpr int kEntries = 24;
    std::vector<std::shared_ptr<Entry>> slots;
    std::mutex mu;
    unsigned long long clock = 0;
    std::atomic<long long> hits{ 0 }, misses{ 0 };
    bool enabled = true;
    SolveCache() { if (const char *m = std::getenv("KIWI_HIP_EIK_CACHE")) enabled = std::atoi(m) != 0; }
    static SolveCache &get() { static SolveCache c; return c; }
    static unsigned long long hash_of(const std::vector<float> &speed, int nx, int ny, int ix, int iy, float dx, float dy)
    {
        unsigned long long h = 1469598103934665603ull;
        auto mix = [&h](unsigned long long v) { h ^= v; h *= 1099511628211ull; h ^= h >> 29; };
        mix((unsigned long long)nx << 32 | (unsigned)ny);
        mix((unsigned long long)ix << 32 | (unsigned)iy);
        unsigned a, b;
        std::memcpy(&a, &dx, 4); std::memcpy(&b, &dy, 4);
        mix((unsigned long long)a << 32 | b);
        const size_t n = speed.size();
        size_t k = 0;
        for (; k + 2 <= n; k += 2) { unsigned long long w; std::memcpy(&w, &speed[k], 8); mix(w); }
        if (k < n) { unsigned w; std::memcpy(&w, &speed[k], 4); mix(w); }
        return h;
    }
    std::shared_ptr<Entry> find(unsigned long long h, const std::vector<float> &speed, int nx, int ny, int ix, int iy, float dx, float dy, unsigned discard)
    {
        std::vector<std::shared_ptr<Entry>> cand;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (auto &e : slots)
                if (e && e->hash == h && e->nx == nx && e->ny == ny && e->ix == ix && e->iy == iy && e->dx == dx && e->dy == dy && e->discard == discard) {
                    e->stamp = ++clock;
                    cand.push_back(e);
                }
        }
        for (auto &e : cand)                              // (outside the lock: entries are immutable once published)
            if (e->speed.size() == speed.size() && std::memcmp(e->speed.data(), speed.data(), speed.size() * sizeof(float)) == 0) return e;
        return nullptr;
    }
    // An entry object to fill: the oldest one when the table is full and nobody else holds it (its two grids keep their
    // memory: a fresh 3.4 MB per stored solve is a thousand page faults), otherwise none (the caller allocates).
    std::shared_ptr<Entry> take_victim()
    {
        std::lock_guard<std::mutex> lk(mu);
        if ((int)slots.size() < kEntries) return nullptr;
        int old = -1;
        for (size_t i = 0; i < slots.size(); i++) if (slots[i] && (old < 0 || slots[i]->stamp < slots[(size_t)old]->stamp)) old = (int)i;
        if (old < 0 || slots[(size_t)old].use_count() != 1) return nullptr;
        return std::move(slots[(size_t)old]);                                  // (the slot stays empty until put())
    }
    void put(std::shared_ptr<Entry> e)
    {
        std::lock_guard<std::mutex> lk(mu);
        e->stamp = ++clock;
        for (auto &sl : slots) if (!sl) { sl = std::move(e); return; }
        if ((int)slots.size() < kEntries) { slots.push_back(std::move(e)); return; }
        size_t old = 0;
        for (size_t i = 1; i < slots.size(); i++) if (slots[i]->stamp < slots[old]->stamp) old = i;
        slots[old] = std::move(e);
    }
    // could an entry match at all?  (dimensions, spacing, start cell, discard speed: no need to hash 1.7 MB to know it cannot)
    bool candidates(int nx, int ny, int ix, int iy, float dx, float dy, unsigned discard)
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &e : slots)
            if (e && e->nx == nx && e->ny == ny && e->ix == ix && e->iy == iy && e->dx == dx && e->dy == dy && e->discard == discard) return true;
        return false;
    }
    // A sweep over the rupture's shape (nucleation point, rupture velocity: BASELINE config 4's source type inverted for what it
    // is made for) never repeats a solve: hashing and storing 3.4 MB per trial bought nothing and cost a fifth of a discretisation.
    // After kStreak misses in a row only every kProbe-th solve is stored, until something hits again.
    static constexpr int kStreak = 32, kProbe = 8;
    std::atomic<int> miss_streak{ 0 };
    std::atomic<unsigned> probe{ 0 };
};

inline void fast_marching_cached(const std::vector<float> &speed, int nx, int ny, const float origin[2], const float delta[2],
                                 const float start[2], std::vector<float> &times, float discard)
{
    SolveCache &sc = SolveCache::get();
    if (!sc.enabled) { fast_marching(speed.data(), nx, ny, origin, delta, start, times, discard); return; }
    // the start cell exactly as fast_marching computes it: all it takes from `origin` and `start`
    int ix = (int)((start[0] - origin[0]) / delta[0]) + 1, iy = (int)((start[1] - origin[1]) / delta[1]) + 1;
    ix = std::min(std::max(ix, 1), nx);
    iy = std::min(std::max(iy, 1), ny);
    unsigned dbits;
    std::memcpy(&dbits, &discard, 4);
    unsigned long long h = 0;
    bool hashed = false;
    if (sc.candidates(nx, ny, ix, iy, delta[0], delta[1], dbits)) {
        h = SolveCache::hash_of(speed, nx, ny, ix, iy, delta[0], delta[1]);
        hashed = true;
        if (auto e = sc.find(h, speed, nx, ny, ix, iy, delta[0], delta[1], dbits)) {
            times = e->times;
            sc.hits++;
            sc.miss_streak = 0;
            return;
        }
    }
    fast_marching(speed.data(), nx, ny, origin, delta, start, times, discard);
    sc.misses++;
    if (sc.miss_streak.fetch_add(1) >= SolveCache::kStreak && sc.probe.fetch_add(1) % SolveCache::kProbe != 0) return;
    if (!hashed) h = SolveCache::hash_of(speed, nx, ny, ix, iy, delta[0], delta[1]);
    auto e = sc.take_victim();
    if (!e) e = std::make_shared<SolveCache::Entry>();
    e->hash = h; e->nx = nx; e->ny = ny; e->ix = ix; e->iy = iy; e->dx = delta[0]; e->dy = delta[1]; e->discard = dbits;
    e->speed.assign(speed.begin(), speed.end());
    e->times.assign(times.begin(), times.end());
    sc.put(std::move(e));
}

// ---- the two passes over the fine grid, four points at a time (round 6) -------------------------------------------------------
// With the march at half its former cost, building the speed grid and binning the arrival times were a third of a
// discretisation.  Both are element-wise in the fine grid's points -- the same multiplies, adds, divides and square roots per
// point, in the order of the scalar statements in discretize_eikonal below (which stay as the KIWI_HIP_EIK_PLAIN=1 path and are
// what these are tested against, bit for bit) -- so four points go through one SSE2 lane each.  The point coordinates are not kept
// (12 bytes per point, written and read twice): the second pass recomputes them.
struct FineGrid {
    float R[3][3];                 // rupture coordinates -> ned (init_euler(dip, strike, 0))
    float shift[3], center[3];
    float brad;
    const HalfSpace *cons; int ncons;
    float lo[2], fd[2];
    int fx, fy;
};

// x of the points of a row, and the ned coordinates of four neighbouring points of row `yv` (rc_to_ned of (x, y, 0))
inline void fine_xs(const FineGrid &g, std::vector<float> &xs)
{
    xs.resize((size_t)g.fx + 4);
    for (int ix = 1; ix <= g.fx; ix++) xs[ix - 1] = g.lo[0] + ((float)ix - 0.5f) * g.fd[0];
    for (int q = 0; q < 4; q++) xs[(size_t)g.fx + q] = xs[(size_t)g.fx - 1];
}
struct RowTerms { __m128 m0[3], c1[3], z[3], sh[3]; };
inline RowTerms row_terms(const FineGrid &g, float y)
{
    RowTerms t;
    for (int i = 0; i < 3; i++) {
        t.m0[i] = _mm_set1_ps(g.R[i][0]);
        t.c1[i] = _mm_set1_ps(g.R[i][1] * y);
        t.z[i] = _mm_set1_ps(g.R[i][2] * 0.f);
        t.sh[i] = _mm_set1_ps(g.shift[i]);
    }
    return t;
}
inline void ned4(const RowTerms &t, __m128 X, __m128 P[3])
{
    for (int i = 0; i < 3; i++)
        P[i] = _mm_add_ps(_mm_add_ps(_mm_add_ps(_mm_mul_ps(t.m0[i], X), t.c1[i]), t.z[i]), t.sh[i]);
}
inline __m128 sel(__m128 mask, __m128 a, __m128 b) { return _mm_or_ps(_mm_and_ps(mask, a), _mm_andnot_ps(mask, b)); }

// psm_make_*_grid (source_mt_eikonal.f90:467-519): speed[fy][fx] = vs(depth) * relv inside of the rupture, 0 outside;
// returns the smallest speed inside
inline float speed_grid_simd(const FineGrid &g, const CrustProfile &prof, float relv, std::vector<float> &speed)
{
    const int fx = g.fx, fy = g.fy;
    speed.resize((size_t)fx * fy + 4);
    static thread_local std::vector<float> xs;
    fine_xs(g, xs);
    float accl[7], sv[8];
    { float acc = 0.f; for (int l = 2; l < 7; l++) { acc = acc + prof.thickness[l]; accl[l] = acc; } }      // crust2x2_get_at_depth
    for (int l = 0; l < 8; l++) sv[l] = prof.vs[l] * relv;
    const __m128 brad = _mm_set1_ps(g.brad), zero = _mm_setzero_ps();
    __m128 vmin = _mm_set1_ps(std::numeric_limits<float>::max());
    for (int iy = 1; iy <= fy; iy++) {
        const RowTerms t = row_terms(g, g.lo[1] + ((float)iy - 0.5f) * g.fd[1]);
        float *row = &speed[(size_t)(iy - 1) * fx];
        for (int ix = 0; ix < fx; ix += 4) {
            __m128 P[3];
            ned4(t, _mm_loadu_ps(&xs[ix]), P);
            const __m128 d0 = _mm_sub_ps(P[0], _mm_set1_ps(g.center[0])), d1 = _mm_sub_ps(P[1], _mm_set1_ps(g.center[1])),
                         d2 = _mm_sub_ps(P[2], _mm_set1_ps(g.center[2]));
            const __m128 dd = _mm_add_ps(_mm_add_ps(_mm_mul_ps(d0, d0), _mm_mul_ps(d1, d1)), _mm_mul_ps(d2, d2));
            __m128 in = _mm_cmpngt_ps(_mm_sqrt_ps(dd), brad);                // inside: NOT (distance > radius), as the scalar test reads
            for (int c = 0; c < g.ncons; c++) {                               // point_in_halfspace, geometry.f90:55-64
                const HalfSpace &h = g.cons[c];
                const __m128 e0 = _mm_sub_ps(_mm_set1_ps(h.point[0]), P[0]), e1 = _mm_sub_ps(_mm_set1_ps(h.point[1]), P[1]),
                             e2 = _mm_sub_ps(_mm_set1_ps(h.point[2]), P[2]);
                const __m128 lam = _mm_add_ps(_mm_add_ps(_mm_mul_ps(_mm_set1_ps(h.normal[0]), e0), _mm_mul_ps(_mm_set1_ps(h.normal[1]), e1)),
                                              _mm_mul_ps(_mm_set1_ps(h.normal[2]), e2));
                in = _mm_and_ps(in, _mm_cmpge_ps(lam, zero));
            }
            __m128 v = _mm_set1_ps(sv[7]);
            for (int l = 6; l >= 2; l--) v = sel(_mm_cmpge_ps(_mm_set1_ps(accl[l]), P[2]), _mm_set1_ps(sv[l]), v);   // first layer reaching the depth
            const __m128 sp = _mm_and_ps(in, v);
            if (ix + 4 <= fx) {
                _mm_storeu_ps(row + ix, sp);
                vmin = _mm_min_ps(vmin, sel(in, v, vmin));
            } else {
                float tmp[4], tin[4], tv[4];
                _mm_storeu_ps(tmp, sp); _mm_storeu_ps(tin, in); _mm_storeu_ps(tv, v);
                float lane_min[4];
                _mm_storeu_ps(lane_min, vmin);
                for (int q = 0; ix + q < fx; q++) {
                    row[ix + q] = tmp[q];
                    uint32_t b; std::memcpy(&b, &tin[q], 4);
                    if (b) lane_min[q] = std::min(tv[q], lane_min[q]);
                }
                vmin = _mm_loadu_ps(lane_min);
            }
        }
    }
    float lm[4];
    _mm_storeu_ps(lm, vmin);
    speed.resize((size_t)fx * fy);
    return std::min(std::min(lm[0], lm[1]), std::min(lm[2], lm[3]));
}

// psm_downsample_grid's first pass (source_mt_eikonal.f90:526-608): the coarse cell of every fine point that has an arrival time,
// and the sums per cell in the order of the fine points
struct CoarseSums { std::vector<float> cnt, ct, cs; std::vector<std::array<float, 3>> cp; int npf = 0; };
// (`invalid`: the speed given to the points outside of the rupture -- they have no arrival time, source_mt_eikonal.f90:519)
inline void coarse_sums_simd(const FineGrid &g, const std::vector<float> &speed, const std::vector<float> &ftimes, float invalid, const float cd[2],
                             int nxc, int nyc, std::vector<int> &cellof, CoarseSums &o)
{
    const int fx = g.fx, fy = g.fy;
    static thread_local std::vector<float> xs, pb;
    static thread_local std::vector<int> icb;
    fine_xs(g, xs);
    pb.resize(3 * ((size_t)fx + 4));
    icb.resize((size_t)fx + 4);
    float *p0 = pb.data(), *p1 = p0 + fx + 4, *p2 = p1 + fx + 4;
    cellof.resize((size_t)fx * fy);
    const __m128 lo0 = _mm_set1_ps(g.lo[0]), lo1 = _mm_set1_ps(g.lo[1]), cd0 = _mm_set1_ps(cd[0]), cd1 = _mm_set1_ps(cd[1]);
    const __m128i one = _mm_set1_epi32(1), vnxc = _mm_set1_epi32(nxc), vnyc = _mm_set1_epi32(nyc);
    auto ifloor4 = [](__m128 v) {                                            // floor() inside the int range, as integers
        const __m128i i = _mm_cvttps_epi32(v);
        return _mm_add_epi32(i, _mm_castps_si128(_mm_cmpgt_ps(_mm_cvtepi32_ps(i), v)));     // (mask = -1 where (float)i > v)
    };
    float last_speed = 0.f, last_inv = 0.f;
    bool have_inv = false;
    int cur = -1;
    float a_cnt = 0.f, a_ct = 0.f, a_cs = 0.f, a_p0 = 0.f, a_p1 = 0.f, a_p2 = 0.f;
    for (int iy = 1; iy <= fy; iy++) {
        const RowTerms t = row_terms(g, g.lo[1] + ((float)iy - 0.5f) * g.fd[1]);
        for (int ix = 0; ix < fx; ix += 4) {
            __m128 P[3];
            ned4(t, _mm_loadu_ps(&xs[ix]), P);
            _mm_storeu_ps(p0 + ix, P[0]); _mm_storeu_ps(p1 + ix, P[1]); _mm_storeu_ps(p2 + ix, P[2]);
            const __m128 v0 = _mm_sub_ps(P[0], t.sh[0]), v1 = _mm_sub_ps(P[1], t.sh[1]), v2 = _mm_sub_ps(P[2], t.sh[2]);     // ned_to_rc
            const __m128 r0 = _mm_add_ps(_mm_add_ps(_mm_mul_ps(_mm_set1_ps(g.R[0][0]), v0), _mm_mul_ps(_mm_set1_ps(g.R[1][0]), v1)), _mm_mul_ps(_mm_set1_ps(g.R[2][0]), v2));
            const __m128 r1 = _mm_add_ps(_mm_add_ps(_mm_mul_ps(_mm_set1_ps(g.R[0][1]), v0), _mm_mul_ps(_mm_set1_ps(g.R[1][1]), v1)), _mm_mul_ps(_mm_set1_ps(g.R[2][1]), v2));
            const __m128i ixc = _mm_add_epi32(ifloor4(_mm_div_ps(_mm_sub_ps(r0, lo0), cd0)), one);
            const __m128i iyc = _mm_add_epi32(ifloor4(_mm_div_ps(_mm_sub_ps(r1, lo1), cd1)), one);
            // orphaned point: ixc < 1 || iyc < 1 || ixc > nxc || iyc > nyc
            const __m128i bad = _mm_or_si128(_mm_or_si128(_mm_cmplt_epi32(ixc, one), _mm_cmplt_epi32(iyc, one)),
                                             _mm_or_si128(_mm_cmpgt_epi32(ixc, vnxc), _mm_cmpgt_epi32(iyc, vnyc)));
            int a[4], b[4], c[4];
            _mm_storeu_si128((__m128i *)a, ixc); _mm_storeu_si128((__m128i *)b, iyc); _mm_storeu_si128((__m128i *)c, bad);
            for (int q = 0; q < 4; q++) icb[ix + q] = c[q] ? -1 : (b[q] - 1) * nxc + a[q] - 1;
        }
        // the sums of a cell stay in registers while consecutive points fall into it (same additions, same order per cell)
        const size_t k0 = (size_t)(iy - 1) * fx;
        for (int ix = 0; ix < fx; ix++) {
            const size_t k = k0 + ix;
            const float s = speed[k];
            int ic = -1;
            if (!(s == invalid) && !(ftimes[k] < 0.f)) ic = icb[ix];                       // (points outside: no arrival time)
            cellof[k] = ic;
            if (ic < 0) continue;
            if (ic != cur) {
                if (cur >= 0) { o.cnt[cur] = a_cnt; o.ct[cur] = a_ct; o.cs[cur] = a_cs; o.cp[cur] = { a_p0, a_p1, a_p2 }; }
                cur = ic;
                a_cnt = o.cnt[ic]; a_ct = o.ct[ic]; a_cs = o.cs[ic]; a_p0 = o.cp[ic][0]; a_p1 = o.cp[ic][1]; a_p2 = o.cp[ic][2];
            }
            a_cnt = a_cnt + 1.f;
            if (a_ct == -1.f) a_ct = 0.f;
            a_ct = a_ct + ftimes[k];
            if (!have_inv || std::memcmp(&s, &last_speed, 4) != 0) { last_speed = s; last_inv = 1.f / s; have_inv = true; }
            a_cs = a_cs + last_inv;
            a_p0 = a_p0 + p0[ix]; a_p1 = a_p1 + p1[ix]; a_p2 = a_p2 + p2[ix];
            o.npf++;
        }
    }
    if (cur >= 0) { o.cnt[cur] = a_cnt; o.ct[cur] = a_ct; o.cs[cur] = a_cs; o.cp[cur] = { a_p0, a_p1, a_p2 }; }
}

// its second pass: the sum of |t - mean t of the cell| per cell, in the order of the fine points
inline void coarse_durations(const std::vector<int> &cellof, const std::vector<float> &ftimes, const std::vector<float> &ct, std::vector<float> &cdur)
{
    int cur = -1;
    float acc = 0.f, mean = 0.f;
    const size_t n = cellof.size();
    for (size_t k = 0; k < n; k++) {
        const int ic = cellof[k];
        if (ic < 0) continue;
        if (ic != cur) {
            if (cur >= 0) cdur[cur] = acc;
            cur = ic; acc = cdur[ic]; mean = ct[ic];
        }
        acc = acc + std::fabs(ftimes[k] - mean);
    }
    if (cur >= 0) cdur[cur] = acc;
}

} // namespace eik

inline int source_nparams_eikonal(int type) { return type == 4 ? 15 : (type == 5 ? 20 : -1); }

// returns "" on success, otherwise the reference's error text
inline std::string discretize_eikonal(int type, const float *P, float doi, const CrustProfile &prof,
                                      const std::vector<HalfSpace> &cons, DiscreteSource &out)
{
    using namespace eik;
    const bool mt = (type == 5);
    const int o = mt ? 0 : 1;                                  // `eikonal` has slip-rake at position 8
    const float bsx = P[7 + o], bsy = P[8 + o], brad = P[9 + o], nux = P[10 + o], nuy = P[11 + o], relv = P[12 + o];
    float Rrup[3][3], Rslip[3][3];
    init_euler(d2r(P[6]), d2r(P[5]), 0.f, Rrup);
    if (!mt) init_euler(d2r(P[6]), d2r(P[5]), -d2r(P[7]), Rslip);
    const V3 shift = { P[1], P[2], P[3] };
    auto rc_to_ned = [&](const V3 &rc) { V3 p = mul(Rrup, rc); for (int k = 0; k < 3; k++) p[k] = p[k] + shift[k]; return p; };
    auto ned_to_rc = [&](const V3 &p) { return mulT(Rrup, V3{ p[0] - shift[0], p[1] - shift[1], p[2] - shift[2] }); };
    auto allowed = [&](const V3 &p) { for (auto &h : cons) if (!inside(p, h)) return false; return true; };

    // bounding circle as a 180-gon, clipped by every constraint (psm_borderline_*)
    const V3 center = rc_to_ned({ bsx, bsy, 0.f });
    float tr[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tr[i][j] = -Rrup[i][j] * brad;
    const int ninit = (brad == 0.f) ? 1 : 180;
    std::vector<V3> poly((size_t)ninit);
    for (int i = 1; i <= ninit; i++) {
        const float ang = (float)i * 2.f * kPi / (float)ninit;
        const V3 q = mul(tr, V3{ std::cos(ang), std::sin(ang), 0.f });
        poly[i - 1] = { q[0] + center[0], q[1] + center[1], q[2] + center[2] };
    }
    if (cons.empty()) poly.clear();
    for (auto &h : cons) poly = clip(poly, h);
    if (poly.empty()) return "Empty rupture area";
    float lo[2] = { std::numeric_limits<float>::max(), std::numeric_limits<float>::max() }, hi[2] = { -lo[0], -lo[1] };
    for (auto &p : poly) {
        const V3 rc = ned_to_rc(p);
        for (int k = 0; k < 2; k++) { lo[k] = std::min(lo[k], rc[k]); hi[k] = std::max(hi[k], rc[k]); }
    }

    // fine grid of rupture speeds (psm_make_*_grid)
    const float dgrid = std::min(100.f * doi / 2.f, 4000.f);
    const float ext[2] = { hi[0] - lo[0], hi[1] - lo[1] };
    int nf[2] = { (int)std::ceil(ext[0] / dgrid), (int)std::ceil(ext[1] / dgrid) };
    if (nf[0] == 0) nf[0] = 1;
    if (nf[1] == 0) nf[1] = 1;
    const float fd[2] = { ext[0] / (float)nf[0], ext[1] / (float)nf[1] };
    {
        const float nukl = std::sqrt(nux * nux + nuy * nuy);
        if (!allowed(rc_to_ned({ nux, nuy, 0.f })) || nukl > brad)
            return "position of nucleation point is outside of rupture region";
    }
    const int fx = nf[0], fy = nf[1];
    static thread_local std::vector<float> speed, ftimes;       // per-thread work arrays, see fast_marching
    static thread_local std::vector<V3> fpt;
    const bool plain = fmm_mode() == 1;                         // KIWI_HIP_EIK_PLAIN=1: the scalar statements (tests compare the two)
    FineGrid fg;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) fg.R[i][j] = Rrup[i][j]; fg.shift[i] = shift[i]; fg.center[i] = center[i]; }
    fg.brad = brad; fg.cons = cons.data(); fg.ncons = (int)cons.size();
    fg.lo[0] = lo[0]; fg.lo[1] = lo[1]; fg.fd[0] = fd[0]; fg.fd[1] = fd[1]; fg.fx = fx; fg.fy = fy;
    float minspeed = std::numeric_limits<float>::max();
    if (plain) {
        speed.assign((size_t)fx * fy, 0.f);
        fpt.resize((size_t)fx * fy);
        for (int iy = 1; iy <= fy; iy++)
            for (int ix = 1; ix <= fx; ix++) {
                const size_t k = (size_t)(iy - 1) * fx + ix - 1;
                const V3 p = rc_to_ned({ lo[0] + ((float)ix - 0.5f) * fd[0], lo[1] + ((float)iy - 0.5f) * fd[1], 0.f });
                fpt[k] = p;
                const float d[3] = { p[0] - center[0], p[1] - center[1], p[2] - center[2] };
                if (std::sqrt(dot(d, d)) > brad || !allowed(p)) { speed[k] = 0.f; continue; }
                float vs = prof.vs[7], acc = 0.f;                          // crust2x2_get_at_depth
                for (int l = 2; l < 7; l++) { acc = acc + prof.thickness[l]; if (acc >= p[2]) { vs = prof.vs[l]; break; } }
                speed[k] = vs * relv;
                minspeed = std::min(speed[k], minspeed);
            }
    } else minspeed = speed_grid_simd(fg, prof, relv, speed);
    const float invalid = minspeed * 0.5f;
    for (auto &v : speed) if (v == 0.f) v = invalid;
    const float start[2] = { nux, nuy };
    fast_marching_cached(speed, fx, fy, lo, fd, start, ftimes, invalid);      // (exact: hit = same inputs, compared in full)
    if (plain) for (size_t k = 0; k < speed.size(); k++) if (speed[k] == invalid) ftimes[k] = -1.f;

    // coarse grid (psm_to_tdsm_size_*, psm_downsample_grid)
    const float maxd = 0.5f * doi * minspeed;
    auto count = [](float size, float maxstep) { int n = (int)std::floor(size / maxstep) + 1; if (n <= 1) n = 2; if (size == 0.f) n = 1; return n; };
    const int nxc = count(ext[0], maxd), nyc = count(ext[1], maxd);
    float cd[2] = { (hi[0] - lo[0]) / (float)nxc, (hi[1] - lo[1]) / (float)nyc };
    if (cd[0] == 0.f) cd[0] = 1.f;
    if (cd[1] == 0.f) cd[1] = 1.f;
    const size_t nc = (size_t)nxc * nyc;
    CoarseSums sums;
    sums.cnt.assign(nc, 0.f); sums.ct.assign(nc, -1.f); sums.cs.assign(nc, 0.f); sums.cp.assign(nc, V3{ 0.f, 0.f, 0.f });
    std::vector<float> &cnt = sums.cnt, &ct = sums.ct, &cs = sums.cs;
    std::vector<V3> &cp = sums.cp;
    std::vector<float> cdur(nc, 0.f), cw(nc, 0.f);
    static thread_local std::vector<int> cellof;      // coarse cell of every fine point (-1: none), for the second pass
    int npf = 0;
    if (plain) {
        // floor() of a value inside the int range, as an integer (std::floor is a library call without SSE4.1)
        auto ifloor = [](float v) { int i = (int)v; return i - ((float)i > v ? 1 : 0); };
        auto cell = [&](size_t k) -> int {
            const V3 rc = ned_to_rc(fpt[k]);
            const int ixc = ifloor((rc[0] - lo[0]) / cd[0]) + 1, iyc = ifloor((rc[1] - lo[1]) / cd[1]) + 1;
            if (ixc < 1 || iyc < 1 || ixc > nxc || iyc > nyc) return -1;          // "orphaned point"
            return (iyc - 1) * nxc + ixc - 1;
        };
        cellof.resize(speed.size());
        for (size_t k = 0; k < speed.size(); k++) {
            cellof[k] = -1;
            if (ftimes[k] < 0.f) continue;
            const int ic = cell(k);
            if (ic < 0) continue;
            cellof[k] = ic;
            cnt[ic] = cnt[ic] + 1.f;
            if (ct[ic] == -1.f) ct[ic] = 0.f;
            ct[ic] = ct[ic] + ftimes[k];
            cs[ic] = cs[ic] + 1.f / speed[k];
            for (int q = 0; q < 3; q++) cp[ic][q] = cp[ic][q] + fpt[k][q];
            npf++;
        }
    } else {
        coarse_sums_simd(fg, speed, ftimes, invalid, cd, nxc, nyc, cellof, sums);
        npf = sums.npf;
    }
    for (size_t ic = 0; ic < nc; ic++) if (cnt[ic] > 0.f) {
        ct[ic] = 1.f / cnt[ic] * ct[ic];
        cs[ic] = 1.f / (1.f / cnt[ic] * cs[ic]);
        for (int q = 0; q < 3; q++) cp[ic][q] = 1.f / cnt[ic] * cp[ic][q];
    }
    for (size_t ic = 0; ic < nc; ic++) cw[ic] = cnt[ic] / (float)npf;
    if (plain) {
        for (size_t k = 0; k < speed.size(); k++) {
            const int ic = cellof[k];
            if (ic < 0) continue;
            cdur[ic] = cdur[ic] + std::fabs(ftimes[k] - ct[ic]);
        }
    } else coarse_durations(cellof, ftimes, ct, cdur);
    for (size_t ic = 0; ic < nc; ic++) if (cnt[ic] > 0.f) cdur[ic] = 4.f / cnt[ic] * cdur[ic];

    // centroid table (psm_to_tdsm_table_*); rise time deferred to the fold
    float centertime = 0.f;
    for (size_t ic = 0; ic < nc; ic++) if (ct[ic] >= 0.f) centertime = centertime + ct[ic] * cw[ic];
    float m6[6];
    if (mt) {
        for (int k = 0; k < 6; k++) m6[k] = P[13 + k];
    } else {
        float mr[3][3];
        detail::double_couple(Rslip, 1, mr);               // R m_unrot R^T (np = 1: not divided)
        m6[0] = mr[0][0]; m6[1] = mr[1][1]; m6[2] = mr[2][2]; m6[3] = mr[0][1]; m6[4] = mr[0][2]; m6[5] = mr[1][2];
    }
    out.centroids.clear();
    std::vector<float> tw, to;
    for (size_t ic = 0; ic < nc; ic++) {
        if (ct[ic] < 0.f) continue;
        const float dur = cdur[ic];
        const int nt = (int)std::floor((dur + 0.f) / doi) + 1;          // discretize_subfault_time, risetime 0
        if (nt == 1) { tw.assign(1, 1.f); to.assign(1, 0.f); }
        else detail::bin_stf(detail::trapezoid_stf(dur, 0.f), dur + 0.f, nt, tw, to);
        for (int it = 0; it < nt; it++) {
            Centroid c;
            c.north = cp[ic][0]; c.east = cp[ic][1]; c.depth = cp[ic][2];
            c.time = ct[ic] + to[it] + P[0] - centertime;
            for (int k = 0; k < 6; k++) c.m[k] = m6[k] * tw[it] * cw[ic];
            out.centroids.push_back(c);
        }
    }
    out.moment = P[4];
    out.risetime = mt ? P[19] : P[14];
    return "";
}

} // namespace kiwi
