// mfma_product.hip -- can the matrix pipe take the PRODUCTS of the 4-neighbour blend under the exact contract?
//
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o mfma_product mfma_product.hip && ./mfma_product
//
// v_mfma_f32_4x4x1_16b_f32 computes, in each of 16 blocks, D[i][j] = A[i] * B[j] + C[i][j] (K = 1: one product, one rounding:
// MI355X_MICROARCH.md "FP32-input MFMA": the result is an fmaf chain).  With C = -0 that is fma(a, b, -0), which equals the
// separately rounded product a * b for EVERY input: a non-zero product is unchanged by adding -0, +0 + -0 = +0, -0 + -0 = -0.
// Operand layout (checked below, not assumed): lane l = 4 block + i supplies A[block][i]; lane l = 4 block + j supplies
// B[block][j]; result register r of lane l = 4 block + j holds D[block][i = r][j].  So with A = the blend weights of the four
// trial sources of a workgroup (lane l: w[l & 3]) and B = one register of samples (one per lane), ONE instruction yields the four
// sources' products for 64 samples: D_r[l] = w[r] * b[l].
// Prints the number of inputs on which D_r[l] differs IN BITS from the vector unit's w[r] * b[l] (random bit patterns: zeros of both
// signs, denormals, infinities and NaNs included; NaN payloads are compared as "both NaN").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k(const float *__restrict__ w, const float *__restrict__ b, unsigned *__restrict__ out, unsigned *__restrict__ ref, int n)
{
    const int lane = threadIdx.x & 63;
    for (int it = blockIdx.x; it < n; it += gridDim.x) {
        const float a = w[it * 4 + (lane & 3)];
        const float x = b[(size_t)it * 64 + lane];
        f4 c = { -0.f, -0.f, -0.f, -0.f };
        const f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x, c, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float wr = w[it * 4 + r];
            float p;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(wr), "v"(x));
            out[((size_t)it * 4 + r) * 64 + lane] = __float_as_uint(d[r]);
            ref[((size_t)it * 4 + r) * 64 + lane] = __float_as_uint(p);
        }
    }
}

int main()
{
    const int n = 1 << 16;
    std::vector<float> w((size_t)n * 4), b((size_t)n * 64);
    std::mt19937_64 rng(20261005);
    auto rnd = [&](int mode) {
        unsigned u = (unsigned)rng();
        if (mode == 1) u &= 0x807fffffu;                                  // denormal / zero
        if (mode == 2) u = (u & 0x80000000u) | 0x7f800000u | ((u >> 9) & (rng() & 1 ? 0x7fffffu : 0u));   // inf / NaN
        if (mode == 3) u = (u & 0x80000000u);                               // +-0
        if (mode == 4) u = (u & 0x807fffffu) | ((unsigned)(20 + rng() % 40) << 23);   // tiny magnitudes: products underflow
        float f; memcpy(&f, &u, 4); return f;
    };
    for (auto &v : w) v = rnd(rng() % 16 < 11 ? 0 : (int)(rng() % 5));
    for (auto &v : b) v = rnd(rng() % 16 < 11 ? 0 : (int)(rng() % 5));
    float *dw, *db; unsigned *dout, *dref;
    hipMalloc(&dw, w.size() * 4); hipMalloc(&db, b.size() * 4);
    hipMalloc(&dout, (size_t)n * 256 * 4); hipMalloc(&dref, (size_t)n * 256 * 4);
    hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    k<<<1024, 64>>>(dw, db, dout, dref, n);
    std::vector<unsigned> o((size_t)n * 256), r((size_t)n * 256);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), dref, r.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, nanboth = 0, den = 0, zeros = 0;
    for (size_t i = 0; i < o.size(); i++) {
        const bool on = (o[i] & 0x7fffffffu) > 0x7f800000u, rn = (r[i] & 0x7fffffffu) > 0x7f800000u;
        if (on && rn) { nanboth++; continue; }
        if ((r[i] & 0x7f800000u) == 0 && (r[i] & 0x7fffffu)) den++;
        if ((r[i] & 0x7fffffffu) == 0) zeros++;
        if (o[i] != r[i]) { if (bad < 5) printf("  differs: mfma %08x  v_mul %08x\n", o[i], r[i]); bad++; }
    }
    printf("{\"products\": %zu, \"differing_bits\": %zu, \"both_nan\": %zu, \"denormal_results\": %zu, \"zero_results\": %zu}\n", o.size(), bad, nanboth, den, zeros);
    return bad != 0;
}
