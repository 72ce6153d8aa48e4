// hbm_read.hip -- what a PURE read reaches on this chip, next to the accumulate kernel's HBM-regime figure (cfg3-bigdb4).
//
//   hipcc --offload-arch=gfx950 -O3 -o hbm_read hbm_read.hip
//   ./hbm_read [GiB = 4] [reps = 20]          -> one JSON line: wall-clock GB/s of
//       in_order     every lane 16 bytes (global_load_dwordx4), a workgroup walks 4 KB at a time through its contiguous slice
//       random_rows  the buffer as rows of 16 896 bytes (the Green's function tensor's pitch at L = 4096: 4224 floats), rows
//                    visited in a shuffled order, a workgroup reads one row per step -- the accumulate kernel's access pattern
//                    in the HBM regime (a node row per load task, rows scattered over a 4.2 GB tensor)
//       random_rows_4B   the same rows with 4-byte loads per lane (the narrow form FETCH_SIZE tallies at full size)
//   Under rocprofv3 (counters only: `rocprofv3 --pmc FETCH_SIZE -- ./hbm_read`, `--pmc TCC_MISS_sum TCC_HIT_sum`, `--pmc
//   TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum`) the same kernels give the bytes the counters see for a KNOWN number of bytes read:
//   profiles/README.md has the table (FETCH_SIZE raw, x 2, TCC_MISS x 128 against the buffer size).
//
// The buffer is larger than the 256 MiB Infinity Cache by 16x or more and every pass touches all of it once, so at most 1/16 of a
// pass can be cache hits left over from the pass before.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int kRowBytes = 16896;          // 4224 floats

__global__ __launch_bounds__(256) void fill_kernel(u4 *p, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        p[i] = u4{ (unsigned)i, 1u, 2u, 3u };
}

// contiguous slice per workgroup, 8 loads of 16 bytes in flight per lane
__global__ __launch_bounds__(256) void in_order_kernel(const u4 *__restrict__ p, size_t n16, unsigned *__restrict__ sink)
{
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t a = per * blockIdx.x, b = a + per < n16 ? a + per : n16;
    u4 acc = { 0u, 0u, 0u, 0u };
    size_t i = a + threadIdx.x;
    for (; i + 7 * 256 < b; i += 8 * 256) {
        u4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(p + i + 256 * k);
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k];
    }
    for (; i < b; i += 256) acc ^= p[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[blockIdx.x] = acc.x;        // (never true for the fill pattern: keeps the loads)
}

// a row per workgroup and step, rows in the order of `perm`; 16 896 bytes = 1056 x 16: 4 full sweeps of 256 lanes + 32 lanes
template <int BYTES>
__global__ __launch_bounds__(256) void rows_kernel(const unsigned char *__restrict__ base, const int *__restrict__ perm, int nrows,
                                                   unsigned *__restrict__ sink)
{
    unsigned accs = 0u;
    u4 acc = { 0u, 0u, 0u, 0u };
    for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
        const unsigned char *row = base + (size_t)perm[r] * kRowBytes;
        if constexpr (BYTES == 16) {
            const u4 *q = (const u4 *)row;
            u4 v[5];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = __builtin_nontemporal_load(q + threadIdx.x + 256 * k);
            v[4] = threadIdx.x < 32 ? __builtin_nontemporal_load(q + 1024 + threadIdx.x) : u4{ 0u, 0u, 0u, 0u };
#pragma unroll
            for (int k = 0; k < 5; k++) acc ^= v[k];
        } else {
            const unsigned *q = (const unsigned *)row;
            unsigned v[17];
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(q + threadIdx.x + 256 * k);
            v[16] = threadIdx.x < 128 ? __builtin_nontemporal_load(q + 4096 + threadIdx.x) : 0u;
#pragma unroll
            for (int k = 0; k < 17; k++) accs ^= v[k];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w ^ accs) == 0x9e3779b9u) sink[blockIdx.x] = acc.x;
}

template <class F>
static double timed(F launch, int reps, double bytes)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return bytes * reps / (ms * 1e-3) / 1e9;
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int nrows = (int)(gib * 1073741824.0 / kRowBytes);
    const size_t bytes = (size_t)nrows * kRowBytes, n16 = bytes / 16;
    unsigned char *buf;
    unsigned *sink;
    int *perm;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 1 << 20));
    CK(hipMalloc(&perm, sizeof(int) * (size_t)nrows));
    fill_kernel<<<4096, 256>>>((u4 *)buf, n16);
    std::vector<int> h(nrows);
    for (int i = 0; i < nrows; i++) h[i] = i;
    std::mt19937 rng(20261005);
    std::shuffle(h.begin(), h.end(), rng);
    CK(hipMemcpy(perm, h.data(), sizeof(int) * (size_t)nrows, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    // grids: 256 CUs x 8 workgroups (in order); for the rows three workgroups per CU and more (the accumulate kernel holds 3)
    const double g_in = timed([&] { in_order_kernel<<<2048, 256>>>((const u4 *)buf, n16, sink); }, reps, (double)bytes);
    double g_rows = 0.0;
    int best_grid = 0;
    for (int grid : { 768, 1536, 3072, 6144 }) {
        const double g = timed([&] { rows_kernel<16><<<grid, 256>>>(buf, perm, nrows, sink); }, reps, (double)bytes);
        if (g > g_rows) { g_rows = g; best_grid = grid; }
    }
    const double g_rows768 = timed([&] { rows_kernel<16><<<768, 256>>>(buf, perm, nrows, sink); }, reps, (double)bytes);
    const double g_rows4 = timed([&] { rows_kernel<4><<<3072, 256>>>(buf, perm, nrows, sink); }, reps, (double)bytes);
    printf("{\"buffer_bytes\": %zu, \"row_bytes\": %d, \"rows\": %d, \"reps\": %d, \"in_order_gbs\": %.1f, \"random_rows_gbs\": %.1f, "
           "\"random_rows_best_grid\": %d, \"random_rows_3wg_per_cu_gbs\": %.1f, \"random_rows_4B_gbs\": %.1f, "
           "\"note\": \"wall-clock (HIP events) bytes read per second, 16 B per lane unless stated; buffer >= 16 x Infinity Cache\"}\n",
           bytes, kRowBytes, nrows, reps, g_in, g_rows, best_grid, g_rows768, g_rows4);
    return 0;
}
