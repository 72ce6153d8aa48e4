// Check and timing of spec_fft_norm_kernel (kiwi_kernels.hpp) outside the engine: amplitude spectrum of a zero-padded row
// against a double-precision DFT, and the time for `rows` rows of one length (all rows share one reference / filter row).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I../../kiwi_amd/csrc -I../../include \
//         -o spec_fft spec_fft.hip && ./spec_fft 8192 65535
// MI355X (r02): max |amp - dft| / peak = 4e-8 .. 1.1e-7 for 64 .. 32768 samples; 65535 rows of 8192 samples: 1.4 ms
// (hipFFT r2c + spec_norm_kernel on the same rows: 3.3 ms); 80 % of the vector issue slots busy, LDS pipe 17 %.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "kiwi_kernels.hpp"
using namespace kiwi;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static std::vector<float2> make_table(int ntrans)
{
    const int M = ntrans / 2;
    std::vector<float2> t;
    const double twopi = 6.283185307179586476925286766559;
    for (int len = M; len >= 4; len >>= 2) {
        const int q = len >> 2;
        for (int r = 1; r <= 3; r++)
            for (int pos = 0; pos < q; pos++) { const double a = -twopi * r * pos / len; t.push_back(make_float2((float)cos(a), (float)sin(a))); }
    }
    for (int k = 0; k <= M; k++) { const double a = -twopi * k / ntrans; t.push_back(make_float2((float)cos(a), (float)sin(a))); }
    return t;
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 8192, rows = argc > 2 ? atoi(argv[2]) : 76800;
    const int nb = N / 2 + 1;
    std::vector<float> x((size_t)N * 4);
    srand(1);
    for (int r = 0; r < 4; r++)
        for (int i = 0; i < N; i++) x[(size_t)r * N + i] = i < N / 2 ? (float)(sin(0.01 * i * (r + 1)) * exp(-i / 900.0) + 0.3 * (rand() / (double)RAND_MAX - 0.5)) : 0.f;
    float *buf; CK(hipMalloc(&buf, (size_t)rows * N * 4));
    for (int r = 0; r < rows; r++) CK(hipMemcpyAsync(buf + (size_t)r * N, x.data() + (size_t)(r & 3) * N, (size_t)N * 4, hipMemcpyHostToDevice, 0));
    std::vector<FftPair> pr(rows);
    for (int r = 0; r < rows; r++) { pr[r].fft_ofs = (long long)r * N; pr[r].spec_ofs = 0; pr[r].ntrans = N; pr[r].specofs = 0; pr[r].filtofs = 0; pr[r].slot = 0; }
    FftPair *prd; CK(hipMalloc(&prd, rows * sizeof(FftPair))); CK(hipMemcpy(prd, pr.data(), rows * sizeof(FftPair), hipMemcpyHostToDevice));
    auto tab = make_table(N);
    float2 *tabd; CK(hipMalloc(&tabd, tab.size() * 8)); CK(hipMemcpy(tabd, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    FusedFftTables ft; for (int i = 0; i <= kFusedFftMaxLog2; i++) ft.tab[i] = tabd;
    float *refamp, *filtw, *mis, *amp;
    CK(hipMalloc(&refamp, nb * 4)); CK(hipMalloc(&filtw, nb * 4)); CK(hipMalloc(&mis, rows * 4)); CK(hipMalloc(&amp, nb * 4));
    std::vector<float> ones(nb, 1.f), zeros(nb, 0.f);
    CK(hipMemcpy(filtw, ones.data(), nb * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(refamp, zeros.data(), nb * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&spec_fft_norm_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&spec_fft_norm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    SpecParams sp{ 3, 0.5f, 1.f, rows, 0, 1 };
    // check: amplitudes of row 1 against a double-precision DFT
    hipLaunchKernelGGL(spec_fft_norm_kernel<1>, dim3(2), dim3(256), (size_t)N * 4, 0, buf, prd, ft, (const float *)nullptr, filtw, sp, (float *)nullptr, amp, SynRows{});
    CK(hipDeviceSynchronize());
    std::vector<float> got(nb);
    CK(hipMemcpy(got.data(), amp, nb * 4, hipMemcpyDeviceToHost));        // (both blocks write the same array: row 1 last or first; rows 0 and 1 differ)
    hipLaunchKernelGGL(spec_fft_norm_kernel<1>, dim3(1), dim3(256), (size_t)N * 4, 0, buf, prd, ft, (const float *)nullptr, filtw, sp, (float *)nullptr, amp, SynRows{});
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), amp, nb * 4, hipMemcpyDeviceToHost));
    double worst = 0, amax = 0, l2 = 0;
    const double twopi = 6.283185307179586476925286766559;
    const int kstep = N > 4096 ? 37 : 1;
    for (int k = 0; k < nb; k += kstep) {
        double re = 0, im = 0;
        for (int n = 0; n < N / 2; n++) { const double a = -twopi * (double)((long long)k * n % N) / N; re += x[n] * cos(a); im += x[n] * sin(a); }
        const double want = sqrt(re * re + im * im);
        worst = fmax(worst, fabs(want - got[k])); amax = fmax(amax, want);
    }
    for (int k = 0; k < nb; k++) l2 += (double)got[k] * got[k];
    printf("N %d: max |amp - dft| = %.3e of peak %.3e (rel %.2e)\n", N, worst, amax, worst / amax);
    // misfit against zero reference = sqrt(df * sum amp^2)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(spec_fft_norm_kernel<0>, dim3(1, rows), dim3(256), (size_t)N * 4, 0, buf, prd, ft, refamp, filtw, sp, mis, (float *)nullptr, SynRows{});
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("N %d rows %d: %.3f ms  (%.2f TB/s of input)\n", N, rows, ms, (double)rows * N * 4 / ms / 1e9);
    }
    float m0; CK(hipMemcpy(&m0, mis, 4, hipMemcpyDeviceToHost));
    printf("misfit row 0: %.6e, from amplitudes %.6e\n", m0, sqrt(1.0 / (N * 0.5) * l2));
    return 0;
}
