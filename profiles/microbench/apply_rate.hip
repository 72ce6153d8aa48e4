// Apply-phase microbenchmark (round 3): cycles per wave and centroid of a carried-register apply step (two outputs per
// lane, as the experimental pipelined kernel of that round had it) in isolation -- LDS tile filled once, no global loads but the coefficient lines -- by variant and
// by waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize
//        -I../../kiwi_amd/csrc apply_rate.hip -o apply_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kiwi_kernels.hpp"
using namespace kiwi;

// ---- the compiler-scheduled form of the carried apply step this file measures (two outputs per lane; the product's
// carry2_apply has four and reads its sets through one asm statement per set -- DESIGN.md section 3 says why)
// A carried register set: one register pair per GF component (application order), each a variable of its own.  (As an
// array the compiler's scalar-replacement pass promotes the whole set to ONE <20 x float> value -- a 32-register tuple
// that is copied and spilled as a whole at every conditional load.)
struct CarrySet { f2v &c0, &c1, &c2, &c3, &c4, &c5, &c6, &c7, &c8, &c9; };
template <int I> __device__ __forceinline__ f2v &cs_get(const CarrySet &s)
{
    if constexpr (I == 0) return s.c0; else if constexpr (I == 1) return s.c1; else if constexpr (I == 2) return s.c2;
    else if constexpr (I == 3) return s.c3; else if constexpr (I == 4) return s.c4; else if constexpr (I == 5) return s.c5;
    else if constexpr (I == 6) return s.c6; else if constexpr (I == 7) return s.c7; else if constexpr (I == 8) return s.c8;
    else return s.c9;
}

// All GF components of one centroid from the two register sets L (b[j-1]) and H (b[j]), reference order.  load_lo /
// load_hi: which of the sets this centroid's shift makes it read (see the head of this section).  coef: the centroid's
// 2 NG interpolation coefficients (wave-uniform pointer: scalar loads, SGPR operands of the packed multiplies).
// TAIL: the `factor * last` rule needs the factors and the fraction; they are read from the record then (rare).
template <int NG, int LDS_TILE, bool TAIL, bool ROT>
__device__ __forceinline__ void carry_apply(f2v &ar1, f2v &ar2, f2v &dz, const CarrySet &L, const CarrySet &H,
                                            const TileBase &cb, bool load_lo, bool load_hi, const float *__restrict__ coef,
                                            int jl, const int *__restrict__ jendp, const GeoRec *__restrict__ rec, float sd,
                                            float cl, float sl)
{
    constexpr int seq10[10] = { 0, 1, 2, 8, 3, 4, 5, 6, 7, 9 }, seq8[8] = { 0, 1, 2, 3, 4, 5, 6, 7 };
    constexpr int nH1 = (NG == 10) ? 4 : 3;      // components summed into the radial trace
    if (load_hi)
        static_for<NG>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value, o = ((NG == 10) ? seq10[i] : seq8[i]) * LDS_TILE;
            cs_get<i>(H) = f2v{ cb.hi[o], cb.hi[o + 64] };
        });
    if (load_lo)
        static_for<NG>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value, o = ((NG == 10) ? seq10[i] : seq8[i]) * LDS_TILE;
            cs_get<i>(L) = f2v{ cb.lo[o], cb.lo[o + 64] };
        });
    float cw[2 * NG];
#pragma unroll
    for (int i = 0; i < 2 * NG; i++) cw[i] = coef[i];
    float fac[NG];
    int jend[NG];
    if constexpr (TAIL) {
        const float f0 = rec->f[0], f1 = rec->f[1], f2 = rec->f[2], f3 = rec->f[3], f4 = rec->f[4], f5 = rec->f[5];
        const float fac10[10] = { f0, f1, f2, f5, f3, f4, f0 * sd, f1 * sd, f2 * sd, f5 * sd };
        const float fac8[8] = { f0, f1, f2, f3, f4, f0 * sd, f1 * sd, f2 * sd };
#pragma unroll
        for (int i = 0; i < NG; i++) { fac[i] = (NG == 10) ? fac10[i] : fac8[i]; jend[i] = jendp[(NG == 10) ? seq10[i] : seq8[i]]; }
    } else {
#pragma unroll
        for (int i = 0; i < NG; i++) { fac[i] = 0.f; jend[i] = 0; }
    }
    f2v t1[1], t2[1], dd[1];
    t1[0] = ROT ? f2v{ 0.f, 0.f } : ar1; t2[0] = ROT ? f2v{ 0.f, 0.f } : ar2; dd[0] = dz;
    static_for<NG>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        TileRegsN<1> tr;
        tr.lo[0] = cs_get<i>(L); tr.hi[0] = cs_get<i>(H);
        const float wl = cw[2 * i], wr = cw[2 * i + 1];
        if constexpr (i < nH1) tile_fma<TAIL, 1>(t1, tr, jl, jend[i], fac[i], wl, wr);
        else if constexpr (i < nH1 + 2) tile_fma<TAIL, 1>(t2, tr, jl, jend[i], fac[i], wl, wr);
        else tile_fma<TAIL, 1>(dd, tr, jl, jend[i], fac[i], wl, wr);
        if constexpr (i == nH1 + 1) {
            if (ROT) {
                ar1 = ar1 + cl * t1[0] - sl * t2[0];
                ar2 = ar2 + cl * t2[0] + sl * t1[0];
            } else {
                ar1 = t1[0]; ar2 = t2[0];
            }
        }
    });
    dz = dd[0];
}


// MODE bit0: coefficient line of every centroid from memory (scalar loads) / 0: one line for all (cache hits)
//      bit1: carry (shift pattern +1 per step, groups of 5) / 0: every step reads both sets
//      bit2: no LDS reads at all (registers keep what they have)
template <int MODE, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES))) void k(const float *__restrict__ coefs, float *out, int ncent, int pad_lds)
{
    constexpr int NG = 10, LDS_TILE = 576;
    extern __shared__ float dyn[];
    float (*tiles)[LDS_TILE] = (float (*)[LDS_TILE])dyn;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < NG * LDS_TILE; i += 256) dyn[i] = 1e-3f * (float)((i * 7 + blockIdx.x) % 97);
    __syncthreads();
    const int u0 = 2 * (tid & ~63) + lane;
    f2v ar1 = { 0.f, 0.f }, ar2 = { 0.f, 0.f }, dz = { 0.f, 0.f };
    f2v x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, y0, y1, y2, y3, y4, y5, y6, y7, y8, y9;
    x0 = x1 = x2 = x3 = x4 = x5 = x6 = x7 = x8 = x9 = y0 = y1 = y2 = y3 = y4 = y5 = y6 = y7 = y8 = y9 = f2v{ 1.f, 2.f };
    const CarrySet X{ x0, x1, x2, x3, x4, x5, x6, x7, x8, x9 }, Y{ y0, y1, y2, y3, y4, y5, y6, y7, y8, y9 };
    // coefficient lines: 128 floats apart like the descriptor rows; workgroups b, b + 1, b + 2, b + 3 share theirs (the four
    // tiles of a receiver)
    const float *__restrict__ cbase = coefs + (size_t)(blockIdx.x / 4) * (size_t)ncent * 128;
    int eprev = 0;
    bool have = false;
#define STEP(LL, HH, CC) do { \
        const int e = 4 - ((CC) % 5); \
        const float *__restrict__ coef = cbase + ((MODE & 1) ? (size_t)(CC) * 128 : 0) + 104; \
        const TileBase chunk0 = tile_base(&tiles[0][e + u0]); \
        const int d = ((MODE & 2) && have) ? eprev - e : 0x7fff; \
        const bool load_lo = !(MODE & 4) && d != -1, load_hi = !(MODE & 4) && d != 1; \
        carry_apply<NG, LDS_TILE, false, true>(ar1, ar2, dz, LL, HH, chunk0, load_lo, load_hi, coef, 0, nullptr, nullptr, 1.f, 0.8f, 0.6f); \
        have = true; eprev = e; if ((CC) % 5 == 4) have = false; \
        asm volatile("; step" ::: "memory"); } while (0)
    for (int cc = 0; cc + 1 < ncent; cc += 2) { STEP(X, Y, cc); STEP(Y, X, cc + 1); }
    out[(size_t)blockIdx.x * 256 + tid] = ar1.x + ar1.y + ar2.x + ar2.y + dz.x + dz.y;
}

template <int MODE, int WAVES>
static void run(const char *name, const float *coefs, float *out, int ncent)
{
    // WAVES per SIMD = workgroups (4 waves) per CU; LDS padding keeps more from being resident
    const int wg_per_cu = WAVES;
    const size_t lds = std::max<size_t>(10 * 576 * 4, (size_t)(160 * 1024 / wg_per_cu) - 1024);
    hipFuncSetAttribute((const void *)k<MODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        k<MODE, WAVES><<<blocks, 256, lds>>>(coefs, out, ncent, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = std::min(best, ms);
    }
    // per wave and centroid, in ns and in issue cycles of the SIMD at 2.4 GHz: all WAVES waves of a SIMD run ncent centroids
    const double ns_per = best * 1e6 / ncent;            // wall time per centroid (every wave advances one)
    printf("%-34s waves/SIMD %d: %8.3f ms  %7.1f ns per centroid step  = %6.1f cycles@2.4GHz per wave-centroid and SIMD slot  (useful: 48 packed ops = 192 cycles per wave)\n",
           name, WAVES, best, ns_per, ns_per * 2.4 / WAVES);
}

int main()
{
    const int ncent = 4000;
    float *coefs, *out;
    const size_t ncoef = (size_t)(256 * 8 / 4 + 1) * ncent * 128;
    hipMalloc(&coefs, ncoef * sizeof(float));
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    std::vector<float> h(ncoef);
    for (size_t i = 0; i < ncoef; i++) h[i] = 1e-3f * (float)(i % 89);
    hipMemcpy(coefs, h.data(), ncoef * sizeof(float), hipMemcpyHostToDevice);
#define RUNW(M, NAME) run<M, 2>(NAME, coefs, out, ncent); run<M, 3>(NAME, coefs, out, ncent); run<M, 4>(NAME, coefs, out, ncent);
    RUNW(3, "carry, coef lines from memory");
    RUNW(2, "carry, one coef line (cache hit)");
    RUNW(1, "both sets read, coefs from memory");
    RUNW(0, "both sets read, one coef line");
    RUNW(6, "no LDS reads, one coef line");
    RUNW(7, "no LDS reads, coefs from memory");
    return 0;
}
