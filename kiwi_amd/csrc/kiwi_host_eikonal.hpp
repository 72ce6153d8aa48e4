// kiwi_host_eikonal.hpp -- host discretiser of the variable-rupture-speed sources `eikonal` and
// `mt_eikonal` (source_eikonal.f90:205-316,435-710; source_mt_eikonal.f90:200-323,442-762):
//   bounding circle clipped by the constraint half-spaces (geometry.f90:173-256) -> fine grid of
//   rupture speeds from a 1-D crustal profile (crust2x2.f90:170-195) -> fast-marching arrival
//   times (eikonal.f90:29-199 with the index heap of heap.f90) -> coarse cells (mean time,
//   harmonic-mean speed, weight, duration) -> centroid table; the constant rise time is applied
//   after synthesis (psm%risetime).
// Sequential, default-real arithmetic in the reference's order; stays on the host.
#pragma once
#include "kiwi_host.hpp"
#include <array>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>

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

// index heap keyed by an external array, with back pointers (heap.f90); indices 1-based as there.  The key of an entry
// is kept next to its index (one load per comparison instead of two dependent ones); the comparisons, swaps and with
// them the order among equal keys are those of heap.f90.
struct HeapEntry { float key; int idx; };
struct FmmNode { float t; int bp; };
struct IndexHeap {
    std::vector<HeapEntry> &h;     // h[1..n]; storage lent by the caller (reused from solve to solve)
    int n = 0;
    FmmNode *nodes;
    IndexHeap(int cap, FmmNode *nd, std::vector<HeapEntry> &store) : h(store), nodes(nd) { h.resize((size_t)cap + 2); }
    void place(int pos, HeapEntry e) { h[pos] = e; nodes[e.idx - 1].bp = pos; }
    void up(int v)                                   // upheap :205-229
    {
        const HeapEntry e = h[v];
        while (v > 1) {
            const int u = (v - 2) / 2 + 1;
            if (h[u].key <= e.key) break;
            place(v, h[u]);
            v = u;
        }
        place(v, e);
    }
    void down(int v)                                 // downheap :172-203
    {
        const HeapEntry e = h[v];
        int w = 2 * (v - 1) + 2;
        while (w <= n) {
            if (w + 1 <= n && h[w + 1].key < h[w].key) w++;
            if (e.key <= h[w].key) break;
            place(v, h[w]);
            v = w;
            w = 2 * (v - 1) + 2;
        }
        place(v, e);
    }
    void push(int idx) { n++; h[n] = HeapEntry{ nodes[idx - 1].t, idx }; nodes[idx - 1].bp = n; up(n); }   // pushheap :76-101
    void rekey(int pos, float key) { h[pos].key = key; }
    int pop()                                                                // popheap :103-131
    {
        if (n == 0) return 0;
        const HeapEntry top = h[1];
        h[1] = h[n];
        nodes[top.idx - 1].bp = 0;
        n--;
        if (n >= 1) down(1);
        return top.idx;
    }
};

// eikonal_solver_fmm, eikonal.f90:29-199; arrays (nx,ny) with x fastest
// `discard`: nodes of exactly this speed are points outside of the rupture whose times the caller throws away (psm_make_*_grid
// gives them half the slowest speed, source_mt_eikonal.f90:501-517, and overwrites their times with -1 behind the solve).  The march
// ends when the last node that is NOT one of them has been accepted: an accepted node is never touched again (update_neighbor
// returns at once for it, eikonal.f90:131), so every time that is kept is the reference's bit for bit; what is left undone is the
// tail of slow outside nodes (up to a fifth of the grid for an unclipped circle in its bounding box).
inline void fast_marching(const std::vector<float> &speed, int nx, int ny, const float origin[2], const float delta[2],
                          const float start[2], std::vector<float> &times, float discard = std::numeric_limits<float>::quiet_NaN())
{
    constexpr int FARAWAY = -1, ALIVE = 0;
    const float inf = std::numeric_limits<float>::max() * 0.1f;
    const float dx = delta[0], dy = delta[1];
    // The work arrays of a solve (a few MB on the 25 m grid of a 30 km rupture) are kept per thread: a fresh allocation per
    // trial source means a million page faults per batch, which serialise in the kernel when every core discretises at once.
    static thread_local std::vector<FmmNode> nodes;
    static thread_local std::vector<HeapEntry> heap_store;
    nodes.assign((size_t)nx * ny, FmmNode{ inf, FARAWAY });
    auto id = [nx](int x, int y) { return (y - 1) * nx + x; };
    int ix = (int)((start[0] - origin[0]) / dx) + 1, iy = (int)((start[1] - origin[1]) / dy) + 1;
    ix = std::min(std::max(ix, 1), nx);
    iy = std::min(std::max(iy, 1), ny);
    auto finish = [&] { times.resize((size_t)nx * ny); for (size_t k = 0; k < times.size(); k++) times[k] = nodes[k].t; };
    nodes[id(ix, iy) - 1].t = 0.f;
    if (nx == 1 && ny == 1) { finish(); return; }
    nodes[id(ix, iy) - 1].bp = ALIVE;
    int nalive = 1;
    long long wanted = 0;
    for (size_t k = 0; k < speed.size(); k++) wanted += speed[k] != discard;
    if (speed[id(ix, iy) - 1] != discard) wanted--;
    IndexHeap heap(nx * ny, nodes.data(), heap_store);
    auto T = [&](int x, int y) -> float & { return nodes[id(x, y) - 1].t; };
    auto S = [&](int x, int y) { return speed[id(x, y) - 1]; };
    if (1 < ix) T(ix - 1, iy) = dx / S(ix - 1, iy);
    if (ix < nx) T(ix + 1, iy) = dx / S(ix + 1, iy);
    if (1 < iy) T(ix, iy - 1) = dy / S(ix, iy - 1);
    if (iy < ny) T(ix, iy + 1) = dy / S(ix, iy + 1);
    if (1 < ix) heap.push(id(ix - 1, iy));
    if (ix < nx) heap.push(id(ix + 1, iy));
    if (1 < iy) heap.push(id(ix, iy - 1));
    if (iy < ny) heap.push(id(ix, iy + 1));
    const float dx2 = dx * dx, dy2 = dy * dy, dxy2 = dx2 * dy2, dsum = dx2 + dy2;
    auto update = [&](int x, int y) {                // update_neighbor :121-186
        const int i = id(x, y);
        FmmNode &nd = nodes[i - 1];
        if (nd.bp == ALIVE) return;
        if (nd.bp == FARAWAY) heap.push(i);
        float a = inf, b = inf, c = inf, d = inf;
        const float told = nd.t, sp = speed[i - 1];
        if (1 < x) a = nodes[i - 2].t;
        if (x < nx) b = nodes[i].t;
        if (1 < y) c = nodes[i - 1 - nx].t;
        if (y < ny) d = nodes[i - 1 + nx].t;
        float t = 0.f;
        const float aa = std::min(a, b), cc = std::min(c, d);
        if (std::max(aa, cc) != inf) {
            const float q = (aa - cc) * sp;
            const float s = dxy2 * (dsum - q * q);
            if (s >= 0.f) t = std::max(t, ((aa * dy2 + cc * dx2) * sp + std::sqrt(s)) / (sp * dsum));
        }
        if (cc == inf) {
            if (a < inf) t = std::max(t, a + dx / sp);
            if (b < inf) t = std::max(t, b + dx / sp);
        }
        if (aa == inf) {
            if (c < inf) t = std::max(t, c + dy / sp);
            if (d < inf) t = std::max(t, d + dy / sp);
        }
        if (t == 0.f) {
            t = inf;
            if (a < inf) t = std::min(t, a + dx / sp);
            if (b < inf) t = std::min(t, b + dx / sp);
            if (c < inf) t = std::min(t, c + dy / sp);
            if (d < inf) t = std::min(t, d + dy / sp);
        }
        if (t != 0.f && told != t) {                 // updateheap, heap.f90:133-156
            nd.t = t;
            heap.rekey(nd.bp, t);
            if (t < told) heap.up(nd.bp);
            if (t > told) heap.down(nd.bp);
        }
    };
    while (nalive <= nx * ny) {
        const int imin = heap.pop();
        if (imin == 0) break;
        ix = (imin - 1) % nx + 1;
        iy = (imin - 1) / nx + 1;
        nodes[imin - 1].bp = ALIVE;
        nalive++;
        if (speed[imin - 1] != discard && --wanted == 0) break;
        if (1 < ix) update(ix - 1, iy);
        if (ix < nx) update(ix + 1, iy);
        if (1 < iy) update(ix, iy - 1);
        if (iy < ny) update(ix, iy + 1);
    }
    finish();
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
    std::shared_ptr<Entry> find(unsigned long long h, const std::vector<float> &speed, int nx, int ny, int ix, int iy, float dx, float dy)
    {
        std::vector<std::shared_ptr<Entry>> cand;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (auto &e : slots)
                if (e && e->hash == h && e->nx == nx && e->ny == ny && e->ix == ix && e->iy == iy && e->dx == dx && e->dy == dy) {
                    e->stamp = ++clock;
                    cand.push_back(e);
                }
        }
        for (auto &e : cand)                              // (outside the lock: entries are immutable once published)
            if (e->speed.size() == speed.size() && std::memcmp(e->speed.data(), speed.data(), speed.size() * sizeof(float)) == 0) return e;
        return nullptr;
    }
    void put(std::shared_ptr<Entry> e)
    {
        std::lock_guard<std::mutex> lk(mu);
        e->stamp = ++clock;
        if ((int)slots.size() < kEntries) { slots.push_back(std::move(e)); return; }
        size_t old = 0;
        for (size_t i = 1; i < slots.size(); i++) if (slots[i]->stamp < slots[old]->stamp) old = i;
        slots[old] = std::move(e);
    }
};

inline void fast_marching_cached(const std::vector<float> &speed, int nx, int ny, const float origin[2], const float delta[2],
                                 const float start[2], std::vector<float> &times, float discard)
{
    SolveCache &sc = SolveCache::get();
    if (!sc.enabled) { fast_marching(speed, nx, ny, origin, delta, start, times, discard); return; }
    // the start cell exactly as fast_marching computes it: all it takes from `origin` and `start`
    int ix = (int)((start[0] - origin[0]) / delta[0]) + 1, iy = (int)((start[1] - origin[1]) / delta[1]) + 1;
    ix = std::min(std::max(ix, 1), nx);
    iy = std::min(std::max(iy, 1), ny);
    const unsigned long long h = SolveCache::hash_of(speed, nx, ny, ix, iy, delta[0], delta[1]);
    if (auto e = sc.find(h, speed, nx, ny, ix, iy, delta[0], delta[1])) {
        times = e->times;
        sc.hits++;
        return;
    }
    fast_marching(speed, nx, ny, origin, delta, start, times, discard);
    sc.misses++;
    auto e = std::make_shared<SolveCache::Entry>();
    e->hash = h; e->nx = nx; e->ny = ny; e->ix = ix; e->iy = iy; e->dx = delta[0]; e->dy = delta[1];
    e->speed = speed; e->times = times;
    sc.put(std::move(e));
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
    speed.assign((size_t)fx * fy, 0.f);
    fpt.resize((size_t)fx * fy);
    float minspeed = std::numeric_limits<float>::max();
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
    const float invalid = minspeed * 0.5f;
    for (auto &v : speed) if (v == 0.f) v = invalid;
    const float start[2] = { nux, nuy };
    fast_marching_cached(speed, fx, fy, lo, fd, start, ftimes, invalid);      // (exact: hit = same inputs, compared in full)
    for (size_t k = 0; k < speed.size(); k++) if (speed[k] == invalid) ftimes[k] = -1.f;

    // coarse grid (psm_to_tdsm_size_*, psm_downsample_grid)
    const float maxd = 0.5f * doi * minspeed;
    auto count = [](float size, float maxstep) { int n = (int)std::floor(size / maxstep) + 1; if (n <= 1) n = 2; if (size == 0.f) n = 1; return n; };
    const int nxc = count(ext[0], maxd), nyc = count(ext[1], maxd);
    float cd[2] = { (hi[0] - lo[0]) / (float)nxc, (hi[1] - lo[1]) / (float)nyc };
    if (cd[0] == 0.f) cd[0] = 1.f;
    if (cd[1] == 0.f) cd[1] = 1.f;
    const size_t nc = (size_t)nxc * nyc;
    std::vector<float> cnt(nc, 0.f), ct(nc, -1.f), cs(nc, 0.f), cdur(nc, 0.f), cw(nc, 0.f);
    std::vector<V3> cp(nc, V3{ 0.f, 0.f, 0.f });
    // floor() of a value inside the int range, as an integer (std::floor is a library call without SSE4.1)
    auto ifloor = [](float v) { int i = (int)v; return i - ((float)i > v ? 1 : 0); };
    auto cell = [&](size_t k) -> int {
        const V3 rc = ned_to_rc(fpt[k]);
        const int ixc = ifloor((rc[0] - lo[0]) / cd[0]) + 1, iyc = ifloor((rc[1] - lo[1]) / cd[1]) + 1;
        if (ixc < 1 || iyc < 1 || ixc > nxc || iyc > nyc) return -1;          // "orphaned point"
        return (iyc - 1) * nxc + ixc - 1;
    };
    static thread_local std::vector<int> cellof;      // coarse cell of every fine point (-1: none), for the second pass
    cellof.resize(speed.size());
    int npf = 0;
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
    for (size_t ic = 0; ic < nc; ic++) if (cnt[ic] > 0.f) {
        ct[ic] = 1.f / cnt[ic] * ct[ic];
        cs[ic] = 1.f / (1.f / cnt[ic] * cs[ic]);
        for (int q = 0; q < 3; q++) cp[ic][q] = 1.f / cnt[ic] * cp[ic][q];
    }
    for (size_t ic = 0; ic < nc; ic++) cw[ic] = cnt[ic] / (float)npf;
    for (size_t k = 0; k < speed.size(); k++) {
        const int ic = cellof[k];
        if (ic < 0) continue;
        cdur[ic] = cdur[ic] + std::fabs(ftimes[k] - ct[ic]);
    }
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
