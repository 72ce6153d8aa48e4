// Issue-rate microbenchmark on gfx950: v_mul_f32 / v_add_f32 vs v_pk_mul_f32 / v_pk_add_f32 (no FMA contraction),
// at 1, 2 and 8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x[16];
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 0.001f + i;
    f2 y[8];
    for (int i = 0; i < 8; i++) y[i] = f2{ x[2 * i], x[2 * i + 1] };
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) { x[i] = x[i] * a; x[i] = x[i] + b; }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) { y[i] = y[i] * a; y[i] = y[i] + b; }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += x[i];
    for (int i = 0; i < 8; i++) s += y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    const int iters = 2000;
    for (int blocks : { 256, 512, 256 * 8 }) {
    printf("--- %d workgroups of 256 threads (%d wave(s) per SIMD)\n", blocks, blocks / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            else k<1><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // scalar mode: 8*16*2 = 256 VALU instr / iter / wave ; packed: 8*8*2 = 128 instr for the same 256 flop/lane
            const double flops = (double)blocks * 256 * iters * 256;
            printf("mode %s rep %d: %.3f ms  %.1f Tflop/s (%d instr/iter)\n", mode ? "packed" : "scalar", rep, ms, flops / ms * 1e-9, mode ? 128 : 256);
        }
    }
    }
    return 0;
}
