// kiwi_host_fmm.hpp -- the fast-marching solve of the eikonal sources on the host (eikonal.f90:29-199 with the index heap of
// heap.f90), bit for bit: the arrival times depend on the ORDER in which the reference accepts nodes -- update_neighbor reads
// neighbours that are not final yet (eikonal.f90:143-146) and overwrites with `told /= t` (:181) -- and, among equal keys, that
// order is whatever heap.f90's binary heap makes of it.  Two routines:
//
//   fast_marching_plain   the reference's statements one by one (heap.f90's comparisons and swaps; its order among equal keys):
//                         what the golden vectors pinned first, kept as the fall-back and as the in-library check of the other;
//   fast_marching         the same march -- every comparison of keys, every swap of heap entries, every arithmetic statement of
//                         update_neighbor -- on a layout made for it (round 6: a rupture-shape sweep was host-bound by this
//                         routine, 86 % of a discretisation):
//     * the grid padded by a border of accepted nodes at `infinity` and stored with its SHORTER side fastest (the front of a long
//       rupture spans the short side: a column of it is a few cache lines, not one line per row): no bounds tests, no div / mod
//       per accepted node, neighbours at fixed offsets;
//     * the node state in the time array itself: sign bit = accepted (times are >= 0; readers take |t|), `infinity` = far away
//       (a node enters the heap at `infinity` and gets its first time in the same update; the solve falls back to the plain
//       routine should a time ever come out >= `infinity`, negative or NaN);
//     * heap keys and node indices in separate arrays, keys beyond the heap's end held at FLT_MAX, so that removing the top
//       walks THREE levels of smaller children per step (8 + 4 + 2 keys compared at once, SSE2) and stops by index alone;
//     * the removal itself bottom-up: heap.f90's downheap follows the smaller children from the root until the moved element
//       fits; keys do not decrease along that path, so "first level where it fits" found from the bottom is the same level -- one
//       comparison per level on the way down instead of two, and the element (the youngest, almost always the largest) is
//       compared once or twice;
//     * back pointers written only where an element is PLACED (push, the element a sift moves, the parents an upheap displaces):
//       entries a removal shifts up by one level keep a stale pointer, and a node's entry is found at its pointer or one of its
//       ancestors (pointer >> 1, >> 2, ...) -- needed only when a time changes, which on a layered speed field happens once per
//       node.
//   Measured on the GPU box (EPYC 9575F, cfg4's 1200 x 360 grid): 25.6 -> 14.9 ms per solve and thread.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>
#include <emmintrin.h>

namespace kiwi {
namespace eik {

// ---- the reference's statements ----------------------------------------------------------------------------------------------
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
inline void fast_marching_plain(const float *speed, int nx, int ny, const float origin[2], const float delta[2],
                                const float start[2], std::vector<float> &times, float discard = std::numeric_limits<float>::quiet_NaN())
{
    constexpr int FARAWAY = -1, ALIVE = 0;
    const float inf = std::numeric_limits<float>::max() * 0.1f;
    const float dx = delta[0], dy = delta[1];
    const size_t nn = (size_t)nx * ny;
    // The work arrays of a solve (a few MB on the 25 m grid of a 30 km rupture) are kept per thread: a fresh allocation per
    // trial source means a million page faults per batch, which serialise in the kernel when every core discretises at once.
    static thread_local std::vector<FmmNode> nodes;
    static thread_local std::vector<HeapEntry> heap_store;
    nodes.assign(nn, FmmNode{ inf, FARAWAY });
    auto id = [nx](int x, int y) { return (y - 1) * nx + x; };
    int ix = (int)((start[0] - origin[0]) / dx) + 1, iy = (int)((start[1] - origin[1]) / dy) + 1;
    ix = std::min(std::max(ix, 1), nx);
    iy = std::min(std::max(iy, 1), ny);
    auto finish = [&] { times.resize(nn); for (size_t k = 0; k < nn; k++) times[k] = nodes[k].t; };
    nodes[id(ix, iy) - 1].t = 0.f;
    if (nx == 1 && ny == 1) { finish(); return; }
    nodes[id(ix, iy) - 1].bp = ALIVE;
    int nalive = 1;
    long long wanted = 0;
    for (size_t k = 0; k < nn; k++) wanted += speed[k] != discard;
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

// ---- the same march on its own layout ------------------------------------------------------------------------------------------
namespace fm {

constexpr float kInf = std::numeric_limits<float>::max() * 0.1f;      // `infinity` of eikonal.f90:56
constexpr float kSent = std::numeric_limits<float>::max();            // key of the slots beyond the heap's end

inline float mag(float v) { uint32_t b; std::memcpy(&b, &v, 4); b &= 0x7fffffffu; std::memcpy(&v, &b, 4); return v; }
inline bool accepted(float v) { uint32_t b; std::memcpy(&b, &v, 4); return (b >> 31) != 0; }

struct March {
    float *T;                // padded times: sign bit = accepted, kInf = far away
    const float *S;          // padded speeds
    int *BP;                 // last position an entry of the node was PLACED at
    float *hk; int *hi;      // heap keys (32-byte aligned at slot 0) and node indices, slots 1..n
    int sx, sy;              // index steps to the x and y neighbours
    float dx, dy, discard;
    int hiwater;             // out: highest heap slot written
    long long wanted;        // nodes of a speed other than `discard` still to accept
    bool ok;                 // out: false = an assumption of the encoding failed, the caller solves again with the plain routine
};

// The whole march in one function with its state in locals (the arrays are written through float* / int*: state kept in a
// structure would be reloaded after every store).
__attribute__((noinline)) inline void run(March &mm, int i0, bool l, bool r, bool u, bool d)
{
    float *const T = mm.T; const float *const S = mm.S;
    int *const BP = mm.BP; float *const hk = mm.hk; int *const hi = mm.hi;
    const int sx = mm.sx, sy = mm.sy;
    const float dx = mm.dx, dy = mm.dy, discard = mm.discard;
    const float dx2 = dx * dx, dy2 = dy * dy, dxy2 = dx2 * dy2, dsum = dx2 + dy2;
    int n = 0, hiwater = 0;
    long long wanted = mm.wanted;
    bool ok = true;
    mm.ok = false; mm.hiwater = 0;

    auto place = [&](int pos, float key, int idx) __attribute__((always_inline)) { hk[pos] = key; hi[pos] = idx; BP[idx] = pos; };
    auto up = [&](int v) __attribute__((always_inline)) {               // upheap, heap.f90:205-229
        const float ek = hk[v]; const int ei = hi[v];
        while (v > 1) {
            const int p = v >> 1;
            if (hk[p] <= ek) break;
            place(v, hk[p], hi[p]);
            v = p;
        }
        place(v, ek, ei);
    };
    auto down = [&](int v) {                                            // downheap, heap.f90:172-203 (a key raised in place)
        const float ek = hk[v]; const int ei = hi[v];
        int w = 2 * v;
        while (w <= n) {
            w += hk[w + 1] < hk[w];                                     // slot n + 1 holds kSent: never the smaller one
            if (ek <= hk[w]) break;
            hk[v] = hk[w]; hi[v] = hi[w];                               // (moves to the parent slot: pointer left stale)
            v = w;
            w = 2 * v;
        }
        place(v, ek, ei);
    };
    auto push = [&](int idx, float key) __attribute__((always_inline)) {   // pushheap, heap.f90:76-101
        n++;
        hiwater = std::max(hiwater, n);
        hk[n] = key; hi[n] = idx; BP[idx] = n;
        up(n);
    };
    // popheap, heap.f90:103-131: the last entry goes to the root and sifts down along the smaller children
    auto pop = [&]() __attribute__((always_inline)) -> int {
        if (n == 0) return -1;
        const int topi = hi[1];
        const float ek = hk[n]; const int ei = hi[n];
        hk[n] = kSent;
        n--;
        if (n == 0) return topi;
        int path[40];
        int K = 0, v = 1;
        path[0] = 1;
        while (8 * v <= n) {                                            // children, grandchildren, great-grandchildren of v at once
            const __m128 g0 = _mm_load_ps(hk + 8 * v), g1 = _mm_load_ps(hk + 8 * v + 4);
            const __m128 q = _mm_load_ps(hk + 4 * v);
            const int b1 = hk[2 * v + 1] < hk[2 * v];
            const int m3 = _mm_movemask_ps(_mm_cmplt_ps(_mm_shuffle_ps(g0, g1, 0xDD), _mm_shuffle_ps(g0, g1, 0x88)));   // right < left, per pair
            const int m2 = _mm_movemask_ps(_mm_cmplt_ps(_mm_shuffle_ps(q, q, 0xDD), _mm_shuffle_ps(q, q, 0x88)));
            const int c2 = 2 * b1 + ((m2 >> b1) & 1);
            const int c3 = 2 * c2 + ((m3 >> c2) & 1);
            path[K + 1] = 2 * v + b1;
            path[K + 2] = 4 * v + c2;
            v = 8 * v + c3;
            path[K + 3] = v;
            K += 3;
        }
        while (2 * v <= n) {
            const int w = 2 * v;
            v = w + (hk[w + 1] < hk[w]);
            path[++K] = v;
        }
        int j = K;
        while (j >= 1 && ek <= hk[path[j]]) j--;                        // deepest level whose key is below the element's
        for (int k = 1; k <= j; k++) { hk[path[k - 1]] = hk[path[k]]; hi[path[k - 1]] = hi[path[k]]; }
        place(path[j], ek, ei);
        return topi;
    };
    // update_neighbor, eikonal.f90:121-186
    auto update = [&](int i) __attribute__((always_inline)) {
        const float told = T[i];
        if (accepted(told)) return;                                     // :131 (the border too)
        const bool fresh = told == kInf;
        if (fresh) push(i, kInf);                                       // :132-134
        const float sp = S[i];
        const float a = mag(T[i - sx]), b = mag(T[i + sx]), c = mag(T[i - sy]), d = mag(T[i + sy]);
        float t = 0.f;
        const float aa = std::min(a, b), cc = std::min(c, d);
        if (std::max(aa, cc) != kInf) {
            const float q = (aa - cc) * sp;
            const float s = dxy2 * (dsum - q * q);
            if (s >= 0.f) t = std::max(t, ((aa * dy2 + cc * dx2) * sp + std::sqrt(s)) / (sp * dsum));
        }
        if (cc == kInf) {
            if (a < kInf) t = std::max(t, a + dx / sp);
            if (b < kInf) t = std::max(t, b + dx / sp);
        }
        if (aa == kInf) {
            if (c < kInf) t = std::max(t, c + dy / sp);
            if (d < kInf) t = std::max(t, d + dy / sp);
        }
        if (t == 0.f) {
            t = kInf;
            if (a < kInf) t = std::min(t, a + dx / sp);
            if (b < kInf) t = std::min(t, b + dx / sp);
            if (c < kInf) t = std::min(t, c + dy / sp);
            if (d < kInf) t = std::min(t, d + dy / sp);
        }
        if (t != 0.f && told != t) {                                    // updateheap, heap.f90:133-156
            if (!(t > 0.f && t < kInf)) { ok = false; return; }         // not representable in the state encoding
            T[i] = t;
            int p = BP[i];
            while (p > 0 && hi[p] != i) p >>= 1;                        // the entry sits at its pointer or at an ancestor of it
            if (p == 0) { ok = false; return; }
            hk[p] = t;
            if (t < told) up(p);
            if (t > told) down(p);
        } else if (fresh) ok = false;                                   // a node would stay in the heap at `infinity`
    };

    if (l) T[i0 - sx] = dx / S[i0 - sx];                                // eikonal.f90:92-95
    if (r) T[i0 + sx] = dx / S[i0 + sx];
    if (u) T[i0 - sy] = dy / S[i0 - sy];
    if (d) T[i0 + sy] = dy / S[i0 + sy];
    auto good = [](float v) { return v >= 0.f && v < kInf; };
    if ((l && !good(T[i0 - sx])) || (r && !good(T[i0 + sx])) || (u && !good(T[i0 - sy])) || (d && !good(T[i0 + sy]))) return;
    if (l) push(i0 - sx, T[i0 - sx]);                                   // :97-100
    if (r) push(i0 + sx, T[i0 + sx]);
    if (u) push(i0 - sy, T[i0 - sy]);
    if (d) push(i0 + sy, T[i0 + sy]);
    // (`nalive <= nx*ny` of :104 cannot end the loop before the heap is empty: every accepted node was far away once)
    for (;;) {
        const int imin = pop();
        if (imin < 0) break;
        uint32_t bits; std::memcpy(&bits, &T[imin], 4); bits |= 0x80000000u; std::memcpy(&T[imin], &bits, 4);
        if (S[imin] != discard && --wanted == 0) break;
        update(imin - sx);
        update(imin + sx);
        update(imin - sy);
        update(imin + sy);
        if (!ok) break;
    }
    mm.ok = ok; mm.hiwater = hiwater;
}

struct Work { std::vector<float> T, S, hk; std::vector<int> BP, hi; };

} // namespace fm

inline std::atomic<long long> &fmm_fallbacks() { static std::atomic<long long> n{ 0 }; return n; }
inline int &fmm_mode() { static int mode = [] { const char *e = std::getenv("KIWI_HIP_EIK_PLAIN"); return e ? std::atoi(e) : 0; }(); return mode; }

inline void fast_marching(const float *speed, int nx, int ny, const float origin[2], const float delta[2],
                          const float start[2], std::vector<float> &times, float discard = std::numeric_limits<float>::quiet_NaN())
{
    using namespace fm;
    if (fmm_mode() == 1 || (long long)nx * ny > (1ll << 30)) { fast_marching_plain(speed, nx, ny, origin, delta, start, times, discard); return; }
    const float dx = delta[0], dy = delta[1];
    if (nx == 1 && ny == 1) { times.assign(1, 0.f); return; }          // eikonal.f90:85
    static thread_local Work w;
    const bool tr = ny < nx;                                           // the shorter side runs fastest
    const int W = (tr ? ny : nx) + 2, H = (tr ? nx : ny) + 2;
    const int sx = tr ? W : 1, sy = tr ? 1 : W;
    const size_t NP = (size_t)W * H;
    w.T.resize(NP); w.S.resize(NP); w.BP.resize(NP);
    float *T = w.T.data(), *S = w.S.data();
    const float border = -kInf;
    long long wanted = 0;
    for (int q = 0; q < W; q++) { T[q] = border; T[(size_t)(H - 1) * W + q] = border; S[q] = 1.f; S[(size_t)(H - 1) * W + q] = 1.f; }
    if (!tr) {
        for (int y = 1; y <= ny; y++) {
            float *trow = T + (size_t)y * W, *srow = S + (size_t)y * W;
            const float *in = speed + (size_t)(y - 1) * nx;
            trow[0] = border; trow[W - 1] = border; srow[0] = 1.f; srow[W - 1] = 1.f;
            for (int x = 0; x < nx; x++) { trow[x + 1] = kInf; srow[x + 1] = in[x]; wanted += in[x] != discard; }
        }
    } else {
        for (int x = 1; x <= nx; x++) {
            float *trow = T + (size_t)x * W, *srow = S + (size_t)x * W;
            trow[0] = border; trow[W - 1] = border; srow[0] = 1.f; srow[W - 1] = 1.f;
            for (int y = 0; y < ny; y++) trow[y + 1] = kInf;
        }
        for (int y0 = 0; y0 < ny; y0 += 32)                            // transposed in strips of 32 rows
            for (int x = 0; x < nx; x++) {
                float *srow = S + (size_t)(x + 1) * W + 1;
                const int y1 = std::min(ny, y0 + 32);
                for (int y = y0; y < y1; y++) { const float v = speed[(size_t)y * nx + x]; srow[y] = v; wanted += v != discard; }
            }
    }
    int ix = (int)((start[0] - origin[0]) / dx) + 1, iy = (int)((start[1] - origin[1]) / dy) + 1;     // eikonal.f90:70-76
    ix = std::min(std::max(ix, 1), nx);
    iy = std::min(std::max(iy, 1), ny);
    const int i0 = ix * sx + iy * sy;
    { const float z = -0.f; T[i0] = z; }                               // accepted at time 0
    if (S[i0] != discard) wanted--;
    const size_t cap = (size_t)nx * ny + 32;
    if (w.hk.size() < cap + 8) { w.hk.assign(cap + 8, kSent); w.hi.resize(cap + 8); }
    March m;
    m.T = T; m.S = S; m.BP = w.BP.data(); m.hi = w.hi.data();
    m.hk = w.hk.data(); { const size_t mis = ((uintptr_t)m.hk & 31) / 4; if (mis) m.hk += 8 - mis; }
    m.sx = sx; m.sy = sy; m.dx = dx; m.dy = dy; m.discard = discard; m.wanted = wanted; m.hiwater = 0; m.ok = false;
    run(m, i0, 1 < ix, ix < nx, 1 < iy, iy < ny);
    for (int k = 1; k <= m.hiwater; k++) m.hk[k] = kSent;              // the key array goes back clean
    if (!m.ok) {
        fmm_fallbacks()++;
        fast_marching_plain(speed, nx, ny, origin, delta, start, times, discard);
        return;
    }
    times.resize((size_t)nx * ny);
    if (!tr) {
        for (int y = 1; y <= ny; y++) { const float *trow = T + (size_t)y * W + 1; float *o = &times[(size_t)(y - 1) * nx]; for (int x = 0; x < nx; x++) o[x] = mag(trow[x]); }
    } else {
        for (int y0 = 0; y0 < ny; y0 += 32)
            for (int x = 0; x < nx; x++) {
                const float *trow = T + (size_t)(x + 1) * W + 1;
                const int y1 = std::min(ny, y0 + 32);
                for (int y = y0; y < y1; y++) times[(size_t)y * nx + x] = mag(trow[y]);
            }
    }
}

} // namespace eik
} // namespace kiwi
