// LDS read throughput on gfx950 by access width, all conflict-free, 8 reads in flight per wave, 3 waves per SIMD
// (the occupancy of accumulate_grouped_kernel):   hipcc --offload-arch=gfx950 -O3 lds_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define RD8(INSTR, TYPE, MUL, EXTRA)                                                              \
    for (int it = 0; it < iters; it++) {                                                          \
        TYPE r0, r1, r2, r3, r4, r5, r6, r7;                                                      \
        asm volatile(INSTR " %0, %8" EXTRA "\n\t" INSTR " %1, %9" EXTRA "\n\t" INSTR " %2, %10" EXTRA "\n\t" INSTR " %3, %11" EXTRA "\n\t" \
                     INSTR " %4, %8" EXTRA "\n\t" INSTR " %5, %9" EXTRA "\n\t" INSTR " %6, %10" EXTRA "\n\t" INSTR " %7, %11" EXTRA "\n\t" \
                     "s_waitcnt lgkmcnt(0)"                                                       \
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) \
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");                            \
    }

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void kern(float *out, int iters)
{
    __shared__ float tile[10240];
    for (int i = threadIdx.x; i < 10240; i += 256) tile[i] = (float)i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)tile;
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (MODE == 0) { const unsigned a0 = base + 16 * lane + 4096 * w, a1 = a0 + 1024, a2 = a0 + 2048, a3 = a0 + 3072; RD8("ds_read_b128", f4, 16, "") }
    if (MODE == 1) { const unsigned a0 = base + 8 * lane + 4096 * w, a1 = a0 + 512, a2 = a0 + 1024, a3 = a0 + 1536; RD8("ds_read_b64", f2, 8, "") }
    if (MODE == 2) { const unsigned a0 = base + 8 * lane + 4096 * w + 4, a1 = a0 + 512, a2 = a0 + 1024, a3 = a0 + 1536; RD8("ds_read2_b32", f2, 8, " offset1:1") }
    if (MODE == 3) { const unsigned a0 = base + 4 * lane + 4096 * w, a1 = a0 + 512, a2 = a0 + 1024, a3 = a0 + 1536; RD8("ds_read2st64_b32", f2, 8, " offset1:1") }
    if (MODE == 4) { const unsigned a0 = base + 4 * lane + 4096 * w, a1 = a0 + 256, a2 = a0 + 512, a3 = a0 + 768; RD8("ds_read_b32", float, 4, "") }
    if (iters < 0) out[0] = 1.f;
}

template <int MODE> void run(float *d, const char *name, int bytes)
{
    const int iters = 4000, blocks = 256 * 3;              // 3 workgroups of 4 waves per CU = 3 waves per SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<MODE><<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(e0);
    kern<MODE><<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ninstr = (double)blocks * 4 * iters * 8;      // wave instructions
    const double per_cu_cycles = ms * 1e-3 * 2.4e9 / (ninstr / 256);
    printf("%-34s %7.3f ms  %5.2f cycles per wave instruction per CU  %6.1f TB/s\n", name, ms, per_cu_cycles,
           ninstr * 64 * bytes / ms / 1e9);
}

int main()
{
    float *d; (void)hipMalloc(&d, 1024);
    run<0>(d, "ds_read_b128 (16 B lane stride)", 16);
    run<1>(d, "ds_read_b64 (8 B lane stride)", 8);
    run<2>(d, "ds_read2_b32 ofs 0,1 (8 B stride)", 8);
    run<3>(d, "ds_read2st64_b32 (4 B stride)", 8);
    run<4>(d, "ds_read_b32 (4 B lane stride)", 4);
    return 0;
}
