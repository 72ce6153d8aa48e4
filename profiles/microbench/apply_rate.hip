// Apply-phase microbenchmark (round 3): cycles per wave and centroid of the carried-register apply step of
// accumulate_pipe_kernel in isolation -- LDS tile filled once, no global loads but the coefficient lines -- by variant and
// by waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize
//        -I../../kiwi_amd/csrc apply_rate.hip -o apply_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kiwi_kernels.hpp"
using namespace kiwi;

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
