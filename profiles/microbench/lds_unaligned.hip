// Does ds_read_b128 accept a 4-byte-aligned LDS address on gfx950, and at what rate?
// (accumulate_grouped_kernel assembles straddling sample pairs with v_pk_mov_b32; reading the tile a second time one
// float further would trade those VALU slots for LDS slots.)   hipcc --offload-arch=gfx950 -O3 lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 lds_read128(unsigned addr)
{
    f4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

template <int OFS>
__global__ __launch_bounds__(256) void kern(float *out, int iters, int check)
{
    __shared__ float tile[256 * 4 + 64];
    for (int i = threadIdx.x; i < 256 * 4 + 64; i += 256) tile[i] = (float)i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)tile;
    f4 acc = { 0.f, 0.f, 0.f, 0.f };
    for (int it = 0; it < iters; it++) {
        const unsigned a = base + 4u * (threadIdx.x * 4 + OFS + ((it & 3) << 2));
        acc += lds_read128(a);
    }
    if (check) {
        const f4 v = lds_read128(base + 4u * (threadIdx.x * 4 + OFS));
        out[(blockIdx.x * 256 + threadIdx.x) * 4 + 0] = v.x; out[(blockIdx.x * 256 + threadIdx.x) * 4 + 1] = v.y;
        out[(blockIdx.x * 256 + threadIdx.x) * 4 + 2] = v.z; out[(blockIdx.x * 256 + threadIdx.x) * 4 + 3] = v.w;
    } else if (acc.x == -1.f) out[0] = acc.x + acc.y + acc.z + acc.w;
}

template <int OFS> void run(float *d, const char *name)
{
    std::vector<float> h(256 * 4);
    kern<OFS><<<1, 256>>>(d, 0, 1);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; t++) for (int k = 0; k < 4; k++) if (h[t * 4 + k] != (float)(t * 4 + OFS + k)) bad++;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 12;
    kern<OFS><<<blocks, 256>>>(d, iters, 0);
    hipEventRecord(e0);
    kern<OFS><<<blocks, 256>>>(d, iters, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 256 * iters * 16;
    printf("%s: wrong values %d; %.2f ms, %.1f TB/s LDS read\n", name, bad, ms, bytes / ms / 1e9);
}

int main()
{
    float *d; hipMalloc(&d, 1 << 20);
    run<0>(d, "offset 0 (16-byte aligned)");
    run<1>(d, "offset 1 float");
    run<2>(d, "offset 2 floats (8-byte aligned)");
    run<3>(d, "offset 3 floats");
    return 0;
}
