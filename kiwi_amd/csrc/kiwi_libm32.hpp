// kiwi_libm32.hpp -- the three default-real libm functions on the reference's per-centroid path
// (sin, cos in make_weights, seismogram.f90:324-327; atan2 in approx_differential_azidist,
// orthodrome.f90:122), reproduced bit for bit as the reference's host libm (glibc 2.35, what a
// flang/gfortran build on this image links) evaluates them, so that the device geometry kernel
// gets the same fp32 weights as the Fortran host.  Usable from host (g++) and device (hipcc).
//
//   sinf / cosf : glibc sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, sincosf_data.c
//                 (Szabolcs Nagy, ARM optimized-routines; double polynomial, fast path |x| < 120;
//                 outside that range we fall back to the correctly rounded double function)
//   atan2f/atanf: glibc sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c (fdlibm):
//                 "Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.  Developed at
//                 SunPro, a Sun Microsystems, Inc. business.  Permission to use, copy, modify, and
//                 distribute this software is freely granted, provided that this notice is preserved."
//
// tests/test_libm32.py compiles this header for the host and checks it against the running
// libm on millions of arguments.  Must be built with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define KIWI_HD __host__ __device__ __forceinline__
#else
#define KIWI_HD inline
#endif

namespace kiwi {
namespace libm32 {

KIWI_HD uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
KIWI_HD float asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
KIWI_HD uint32_t abstop12(float x) { return (asuint(x) >> 20) & 0x7ff; }

// sincosf_data.c: c0..c4 cosine, s1..s3 sine polynomial coefficients
struct SinCosPoly { double c0, c1, s1, c2, s2, c3, s3, c4; };

// sincosf.h sinf_poly: sine for even n, cosine for odd n
KIWI_HD float sinf_poly(double x, double x2, const SinCosPoly &p, int n)
{
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = p.s2 + x2 * p.s3;
        const double x7 = x3 * x2;
        const double s = x + x3 * p.s1;
        return (float)(s + x7 * s1);
    } else {
        const double x4 = x2 * x2;
        const double c2 = p.c3 + x2 * p.c4;
        const double c1 = p.c0 + x2 * p.c1;
        const double x6 = x4 * x2;
        const double c = c1 + x4 * p.c2;
        return (float)(c + x6 * c2);
    }
}

// which = 0: sinf, 1: cosf
KIWI_HD float sincos_core(float y, int which)
{
    const SinCosPoly P0 = { 0x1p0, -0x1.ffffffd0c621cp-2, -0x1.555545995a603p-3, 0x1.55553e1068f19p-5,
                            0x1.1107605230bc4p-7, -0x1.6c087e89a359dp-10, -0x1.994eb3774cf24p-13,
                            0x1.99343027bf8c3p-16 };
    const SinCosPoly P1 = { -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.555545995a603p-3, -0x1.55553e1068f19p-5,
                            0x1.1107605230bc4p-7, 0x1.6c087e89a359dp-10, -0x1.994eb3774cf24p-13,
                            -0x1.99343027bf8c3p-16 };
    const double hpi_inv = 0x1.45F306DC9C883p+23;     // 2/pi * 2^24
    const double hpi = 0x1.921FB54442D18p0;
    double x = (double)y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {      // |y| < pi/4
        const double x2 = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return which ? 1.0f : y;
        return sinf_poly(x, x2, P0, which);
    }
    if (abstop12(y) < abstop12(120.0f)) {
        // reduce_fast: quadrant in bits 24..31 of the scaled product
        const double r = x * hpi_inv;
        const int n = ((int32_t)r + 0x800000) >> 24;
        x = x - (double)n * hpi;
        const double sgn = ((n & 3) == 0 || (n & 3) == 3) ? 1.0 : -1.0;     // sign[] = {1,-1,-1,1}
        const SinCosPoly &p = (n & 2) ? P1 : P0;
        return sinf_poly(x * sgn, x * x, p, n ^ which);
    }
    return which ? (float)cos((double)y) : (float)sin((double)y);
}

KIWI_HD float sinf_glibc(float y) { return sincos_core(y, 0); }
KIWI_HD float cosf_glibc(float y) { return sincos_core(y, 1); }

// s_atanf.c
KIWI_HD float atanf_glibc(float x)
{
    const float atanhi[4] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
    const float atanlo[4] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
    const float aT[11] = { 3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                           9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                           4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f };
    const float one = 1.0f;
    const int32_t hx = (int32_t)asuint(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                 // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;  // NaN
        if (hx > 0) return atanhi[3] + atanlo[3];
        return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                  // |x| < 0.4375
        if (ix < 0x31000000) return x;      // |x| < 2^-29
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {              // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }
            else                 { id = 1; x = (x - one) / (x + one); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }
            else                 { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -z : z;
}

// e_atan2f.c
KIWI_HD float atan2f_glibc(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
                pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)asuint(x), hy = (int32_t)asuint(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;      // NaN
    if (hx == 0x3f800000) return atanf_glibc(y);               // x = 1.0
    const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);      // 2*sign(x) + sign(y)
    if (iy == 0) {
        switch (m) {
        case 0: case 1: return y;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    float z;
    const int32_t k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = atanf_glibc(fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return asfloat(asuint(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

} // namespace libm32
} // namespace kiwi
