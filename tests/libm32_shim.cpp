// host build of kiwi_amd/csrc/kiwi_libm32.hpp for tests/test_libm32.py
#include "../kiwi_amd/csrc/kiwi_libm32.hpp"
extern "C" {
void shim_sinf(const float *x, float *y, int n) { for (int i = 0; i < n; i++) y[i] = kiwi::libm32::sinf_glibc(x[i]); }
void shim_cosf(const float *x, float *y, int n) { for (int i = 0; i < n; i++) y[i] = kiwi::libm32::cosf_glibc(x[i]); }
void shim_atan2f(const float *a, const float *b, float *y, int n) { for (int i = 0; i < n; i++) y[i] = kiwi::libm32::atan2f_glibc(a[i], b[i]); }
void libm_sinf(const float *x, float *y, int n) { for (int i = 0; i < n; i++) y[i] = sinf(x[i]); }
void libm_cosf(const float *x, float *y, int n) { for (int i = 0; i < n; i++) y[i] = cosf(x[i]); }
void libm_atan2f(const float *a, const float *b, float *y, int n) { for (int i = 0; i < n; i++) y[i] = atan2f(a[i], b[i]); }
}
