// kiwi_host.hpp -- host side of the engine: everything the reference does ONCE per setup or
// once per trial source on the CPU and that is not worth a kernel: receiver geometry
// (orthodrome.f90), piecewise linear tapers (piecewise_linear_function.f90), and the source
// discretisers (source_moment_tensor.f90, source_bilat.f90, source_circular.f90).
//
// All of it is default-real (fp32) / real*8 arithmetic in the reference; the operation
// order is kept so that, with the same libm, results are identical to the Fortran host.
// Compiled with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <string>
#include <algorithm>

namespace kiwi {

// constants.f90:21-25.  pi_ (real*8) is initialised from a default-real literal, so it holds
// the fp32 value; earth_oblateness likewise is a default-real quotient.
constexpr float  kPi      = 3.14159265358979f;
constexpr double kPi8     = (double)3.14159265358979f;
constexpr float  kEarthRadius = 6371.f * 1000.f;
inline float  earthradius_equator() { volatile float a = 6378.14f, b = 1000.f; return a * b; }
inline double earth_oblateness() { volatile float a = 1.f, b = 298.257223563f; return (double)(a / b); }

struct GeoCoords { double lat = 0, lon = 0; };   // orthodrome.f90:32-34, radians

// d2r for real*8 / real (orthodrome.f90:313-338): the factor 2./360.*pi is a default-real constant
inline double d2r(double deg) { return (double)((2.f / 360.f) * kPi) * deg; }
inline float  d2r(float deg)  { return ((2.f / 360.f) * kPi) * deg; }
inline float  r2d(float rad)  { return ((360.f / 2.f) / kPi) * rad; }          // r2d_r, orthodrome.f90:319-326

// cosdelta, orthodrome.f90:284-293
inline double cosdelta(const GeoCoords &a, const GeoCoords &b)
{
    return std::sin(a.lat) * std::sin(b.lat) + std::cos(a.lat) * std::cos(b.lat) * std::cos(b.lon - a.lon);
}

// azibazi, orthodrome.f90:245-265
inline void azibazi(const GeoCoords &a, const GeoCoords &b, double &azi, double &bazi)
{
    const double t = std::cos(a.lat) * std::cos(b.lat) * std::sin(b.lon - a.lon);
    const double sb = std::sin(b.lat), sa = std::sin(a.lat);
    const double cd = cosdelta(a, b);
    azi = std::atan2(t, sb - sa * cd);
    bazi = std::atan2(-t, sa - sb * cd);
}

// distance_accurate50m, orthodrome.f90:193-229
inline double distance_accurate50m(const GeoCoords &a, const GeoCoords &b)
{
    const double f = (a.lat + b.lat) / 2., g = (a.lat - b.lat) / 2., l = (a.lon - b.lon) / 2.;
    const double sg2 = std::sin(g) * std::sin(g), cg2 = std::cos(g) * std::cos(g);
    const double sl2 = std::sin(l) * std::sin(l), cl2 = std::cos(l) * std::cos(l);
    const double sf2 = std::sin(f) * std::sin(f), cf2 = std::cos(f) * std::cos(f);
    const double s = sg2 * cl2 + cf2 * sl2;
    const double c = cg2 * cl2 + sf2 * sl2;
    const double w = std::atan(std::sqrt(s / c));
    const double r = std::sqrt(s * c) / w;
    const double d = 2. * w * (double)earthradius_equator();
    const double h1 = (3. * r - 1.) / (2. * c);
    const double h2 = (3. * r + 1.) / (2. * s);
    const double eo = earth_oblateness();
    return d * (1. + eo * h1 * sf2 * cg2 - eo * h2 * cf2 * sg2);
}

// init_euler, euler.f90:28-67; m[row][col]
inline void init_euler(float alpha, float beta, float gamma, float m[3][3])
{
    const float ca = std::cos(alpha), cb = std::cos(beta), cg = std::cos(gamma);
    const float sa = std::sin(alpha), sb = std::sin(beta), sg = std::sin(gamma);
    m[0][0] = cb * cg - ca * sb * sg;  m[0][1] = -cb * sg - ca * sb * cg;  m[0][2] = sa * sb;
    m[1][0] = sb * cg + ca * cb * sg;  m[1][1] = -sb * sg + ca * cb * cg;  m[1][2] = -sa * cb;
    m[2][0] = sa * sg;                 m[2][1] = sa * cg;                  m[2][2] = ca;
}

// ------------------------------------------------------------------ piecewise linear functions
struct Plf {                     // piecewise_linear_function.f90:27-35
    std::vector<float> x, y;
    bool defined() const { return !x.empty(); }
    int n() const { return (int)x.size(); }
};

enum Interp { IP_COS = 0, IP_LINEAR = 1, IP_ZERO_ONE = 2 };

inline float ip_linear(float x0, float y0, float x1, float y1, float xi) { return y0 + (y1 - y0) / (x1 - x0) * (xi - x0); }
inline float ip_cos(float x0, float y0, float x1, float y1, float xi)
{
    if (y1 != y0) return y0 + (y1 - y0) * (0.5f - 0.5f * std::cos((xi - x0) / (x1 - x0) * kPi));
    return y0;
}
inline float ip_zero_one(float x0, float y0, float x1, float y1, float xi)
{
    if (y0 == 0.f && y1 == 0.f) return 0.f + 0.f * (x0 + x1 + xi);
    return 1.f;
}

// plf_integrate_and_centroid, piecewise_linear_function.f90:165-193 (bins an STF into weights/offsets)
inline void plf_integrate_and_centroid(const Plf &s, float a, float b, float &area, float &centroid)
{
    area = 0.f;
    centroid = (a + b) / 2.f;
    float c = 0.f;
    const int n = s.n();
    if (n == 0 || b <= s.x[0] || a >= s.x[n - 1]) return;
    for (int i = 0; i + 1 < n; i++) {
        if (a >= s.x[i + 1]) continue;
        if (b <= s.x[i]) break;
        const float x0 = std::max(a, s.x[i]), x1 = std::min(b, s.x[i + 1]);
        float y0 = s.y[i], y1 = s.y[i + 1];
        if (x0 != s.x[i]) y0 = ip_linear(s.x[i], s.y[i], s.x[i + 1], s.y[i + 1], a);
        if (x1 != s.x[i + 1]) y1 = ip_linear(s.x[i], s.y[i], s.x[i + 1], s.y[i + 1], b);
        const float areathis = (y0 + y1) * (x1 - x0) / 2.f;                       // trapezoid_area :296
        float tc;                                                                 // trapezoid_centroid :285
        if (y0 + y1 == 0.f) tc = (x0 + x1) / 2.f;
        else tc = (x0 * (2.f * y0 + y1) + x1 * (y0 + 2.f * y1)) / (3.f * (y0 + y1));
        c = c + areathis * tc;
        area = area + areathis;
    }
    centroid = c / area;
}

// plf_taper_array (real), piecewise_linear_function.f90:195-237; sample j is at abscissa j*dx
inline void plf_taper_array(const Plf &s, float *array, int lo, int hi, float dx, Interp ip)
{
    auto A = [&](int j) -> float & { return array[j - lo]; };
    auto F = [&](int i, float xi) {
        switch (ip) {
        case IP_COS: return ip_cos(s.x[i], s.y[i], s.x[i + 1], s.y[i + 1], xi);
        case IP_LINEAR: return ip_linear(s.x[i], s.y[i], s.x[i + 1], s.y[i + 1], xi);
        default: return ip_zero_one(s.x[i], s.y[i], s.x[i + 1], s.y[i + 1], xi);
        }
    };
    const int n = s.n();
    int ibeg = (int)std::floor(s.x[0] / dx);
    if (lo <= ibeg) for (int j = lo; j <= std::min(ibeg, hi); j++) A(j) = 0.f;
    int ibegatleast = lo;
    for (int i = 0; i + 1 < n; i++) {
        ibeg = std::max(std::max((int)std::floor(s.x[i] / dx) + 1, lo), ibegatleast);
        const int iend = std::min((int)std::floor(s.x[i + 1] / dx), hi);
        for (int j = ibeg; j <= iend; j++) A(j) = A(j) * F(i, (float)j * dx);
        ibegatleast = iend + 1;
    }
    const int iend = (int)std::floor(s.x[n - 1] / dx) + 1;
    if (hi >= iend) for (int j = std::max(iend, lo); j <= hi; j++) A(j) = 0.f;
}

// discrete_plf_span, comparator.f90:1157-1169
inline void discrete_plf_span(const Plf &s, float dt, int span[2])
{
    float r0 = 0.f, r1 = -1.f;
    if (s.defined()) { r0 = s.x.front(); r1 = s.x.back(); }
    span[0] = (int)std::ceil(r0 / dt);
    span[1] = (int)std::floor(r1 / dt);
}

// ------------------------------------------------------------------ source discretisers
struct Centroid { float north, east, depth, time, m[6]; };   // discrete_source.f90:27-30

struct DiscreteSource {
    std::vector<Centroid> centroids;
    float moment = 1.f;      // psm%moment,   parameterized_source.f90:70
    float risetime = 0.f;    // psm%risetime, parameterized_source.f90:71
};

inline int source_nparams(int type)
{
    switch (type) {
    case 1: return 14;   // bilateral, source_bilat.f90:32
    case 2: return 11;   // circular, source_circular.f90:32
    case 3: return 13;   // point_lp, source_point_lp.f90:43
    case 6: return 11;   // moment_tensor, source_moment_tensor.f90:34
    }
    return -1;
}

namespace detail {

inline Plf plf4(float x1, float y1, float x2, float y2, float x3, float y3, float x4, float y4)
{
    Plf p; p.x = { x1, x2, x3, x4 }; p.y = { y1, y2, y3, y4 }; return p;
}

// weights and time offsets of nt equal bins over the STF (source_bilat.f90:405-416)
inline void bin_stf(const Plf &stf, float duration, int nt, std::vector<float> &wt, std::vector<float> &toff)
{
    wt.resize(nt); toff.resize(nt);
    const float tbeg = stf.x[0];
    const float dt = duration / (float)nt;
    for (int it = 1; it <= nt; it++)
        plf_integrate_and_centroid(stf, tbeg + dt * (float)(it - 1), tbeg + dt * (float)it, wt[it - 1], toff[it - 1]);
}

// box(risetime) * box(dursf) trapezoid, unit area (source_bilat.f90:386-401)
inline Plf trapezoid_stf(float dursf, float risetime)
{
    if (risetime < dursf)
        return plf4((-dursf - risetime) / 2.f, 0.f, (-dursf + risetime) / 2.f, 1.f / dursf,
                    (dursf - risetime) / 2.f, 1.f / dursf, (dursf + risetime) / 2.f, 0.f);
    return plf4((-risetime - dursf) / 2.f, 0.f, (-risetime + dursf) / 2.f, 1.f / risetime,
                (risetime - dursf) / 2.f, 1.f / risetime, (risetime + dursf) / 2.f, 0.f);
}

inline float dot3(const float r[3], float a, float b, float c) { return (r[0] * a + r[1] * b) + r[2] * c; }

// m_rot = R m_unrot R^T / np with m_unrot = -(e1 e3^T + e3 e1^T)   (source_bilat.f90:347,424-438)
inline void double_couple(const float R[3][3], int np, float mr[3][3])
{
    float inner[3][3];   // m_unrot * R^T : row0 = -R^T row2, row1 = 0, row2 = -R^T row0
    for (int j = 0; j < 3; j++) {
        inner[0][j] = (0.f * R[j][0] + 0.f * R[j][1]) + (-1.f) * R[j][2];
        inner[1][j] = (0.f * R[j][0] + 0.f * R[j][1]) + 0.f * R[j][2];
        inner[2][j] = ((-1.f) * R[j][0] + 0.f * R[j][1]) + 0.f * R[j][2];
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            mr[i][j] = dot3(R[i], inner[0][j], inner[1][j], inner[2][j]) / (float)np;
}

inline void emit(DiscreteSource &out, const std::vector<float> &grid, const std::vector<float> &tshift,
                 const std::vector<float> &wt, const std::vector<float> &toff, const float mr[3][3])
{
    const int np = (int)tshift.size(), nt = (int)wt.size();
    out.centroids.resize((size_t)np * nt);
    size_t id = 0;
    for (int ip = 0; ip < np; ip++)
        for (int it = 0; it < nt; it++, id++) {
            Centroid &c = out.centroids[id];
            c.north = grid[3 * ip]; c.east = grid[3 * ip + 1]; c.depth = grid[3 * ip + 2];
            c.time = tshift[ip] + toff[it];
            c.m[0] = mr[0][0] * wt[it]; c.m[1] = mr[1][1] * wt[it]; c.m[2] = mr[2][2] * wt[it];
            c.m[3] = mr[0][1] * wt[it]; c.m[4] = mr[0][2] * wt[it]; c.m[5] = mr[1][2] * wt[it];
        }
}

inline int grid_count(float extent, float maxd)
{
    int n = (int)std::floor(extent / maxd) + 1;
    if (n <= 1) n = 2;
    if (extent == 0.f) n = 1;
    return n;
}

} // namespace detail

// psm_to_tdsm_moment_tensor, source_moment_tensor.f90:205-267 (psm%moment = 1, :201)
inline bool discretize_moment_tensor(const float *p, float doi, DiscreteSource &out)
{
    const float risetime = p[10], time = p[0];
    int nt = (int)std::floor(risetime / doi) + 1;
    if (nt <= 1) nt = 2;
    const Plf stf = detail::plf4((-risetime) / 2.f, 0.f, (-risetime) / 2.f, 1.f / risetime,
                                 (risetime) / 2.f, 1.f / risetime, (risetime) / 2.f, 0.f);
    std::vector<float> wt, toff;
    detail::bin_stf(stf, risetime, nt, wt, toff);
    out.centroids.resize(nt);
    for (int it = 0; it < nt; it++) {
        Centroid &c = out.centroids[it];
        c.north = p[1]; c.east = p[2]; c.depth = p[3];
        c.time = toff[it] + time;
        for (int k = 0; k < 6; k++) c.m[k] = p[4 + k] * wt[it];
    }
    out.moment = 1.f; out.risetime = 0.f;
    return true;
}

// P and T axes of a bilateral source (psm_update_dep_params_bilat, source_bilat.f90:216-239, with polar / domeshot / wrap
// :565-593): (azimuth, polar angle) in degrees of R_slip (+-sqrt 2, 0, -sqrt 2), folded onto the lower hemisphere
inline void principal_axes_bilat(const float *p, float pax[2], float tax[2])
{
    float R[3][3];
    const float strike = d2r(p[5]), dip = d2r(p[6]), rake = d2r(p[7]);
    init_euler(dip, strike, -rake, R);
    const float s2 = std::sqrt(2.f);
    auto wrap = [](float x, float mi, float ma) { return x - std::floor((x - mi) / (ma - mi)) * (ma - mi); };
    auto axis = [&](float vx, float out[2]) {
        float xyz[3], pol[3];
        for (int i = 0; i < 3; i++) xyz[i] = detail::dot3(R[i], vx, 0.f, -s2);
        pol[0] = std::sqrt((xyz[0] * xyz[0] + xyz[1] * xyz[1]) + xyz[2] * xyz[2]);
        pol[1] = std::atan2(xyz[1], xyz[0]);
        pol[2] = std::acos(xyz[2] / pol[0]);
        float d1 = wrap(pol[1], kPi, -kPi), d2 = wrap(pol[2], kPi, -kPi);
        if (d2 > kPi / 2.f) { d1 = wrap(d1 + kPi, -kPi, kPi); d2 = kPi - d2; }
        out[0] = r2d(d1); out[1] = r2d(d2);
    };
    axis(s2, pax);
    axis(-s2, tax);
}

// psm_set_bilat + psm_to_tdsm_bilat, source_bilat.f90:173-459
inline bool discretize_bilat(const float *p, float doi, DiscreteSource &out)
{
    float Rrup[3][3], Rslip[3][3];
    const float strike = d2r(p[5]), dip = d2r(p[6]), rake = d2r(p[7]), rupdir = d2r(p[8]);
    init_euler(dip, strike, -rupdir, Rrup);
    init_euler(dip, strike, -rake, Rslip);
    const float la = p[9], lb = p[10], width = p[11], rupvel = p[12], risetime = p[13];
    const float length = la + lb;
    const int nx = detail::grid_count(length, 0.5f * doi * rupvel);
    const int ny = detail::grid_count(width, doi * rupvel);
    const float dursf = length / (float)nx / rupvel;
    int nt = (int)std::floor((risetime + dursf) / doi) + 1;
    if (nt <= 1) nt = 2;
    const int np = nx * ny;
    std::vector<float> grid(3 * (size_t)np), tshift(np);
    int ip = 0;
    for (int ix = 1; ix <= nx; ix++)
        for (int iy = 1; iy <= ny; iy++, ip++) {
            const float gx = (2.f * ((float)ix - 1.f) - (float)nx + 1.f) / (2.f * (float)nx) * length;
            const float gy = (2.f * ((float)iy - 1.f) - (float)ny + 1.f) / (2.f * (float)ny) * width;
            tshift[ip] = std::fabs(length / 2.f - lb + gx) / rupvel + p[0] - std::max(la, lb) / 2.f / rupvel;
            grid[3 * ip] = detail::dot3(Rrup[0], gx, gy, 0.f) + p[1];
            grid[3 * ip + 1] = detail::dot3(Rrup[1], gx, gy, 0.f) + p[2];
            grid[3 * ip + 2] = detail::dot3(Rrup[2], gx, gy, 0.f) + p[3];
        }
    std::vector<float> wt, toff;
    detail::bin_stf(detail::trapezoid_stf(dursf, risetime), dursf + risetime, nt, wt, toff);
    float mr[3][3];
    detail::double_couple(Rslip, np, mr);
    detail::emit(out, grid, tshift, wt, toff, mr);
    out.moment = p[4]; out.risetime = 0.f;
    return true;
}

// psm_set_circular + psm_to_tdsm_circular, source_circular.f90:165-444.
// (:221 feeds params(9), the radius, into the rupture-direction Euler angle; kept.)
inline bool discretize_circular(const float *p, float doi, DiscreteSource &out)
{
    float Rrup[3][3], Rslip[3][3];
    const float strike = d2r(p[5]), dip = d2r(p[6]), rake = d2r(p[7]), rupdir = d2r(p[8]);
    init_euler(dip, strike, -rupdir, Rrup);
    init_euler(dip, strike, -rake, Rslip);
    const float radius = p[8], rupvel = p[9], risetime = p[10];
    const float length = radius * 2.f;
    const int nx = detail::grid_count(length, 0.5f * doi * rupvel), ny = nx;
    const float dursf = length / (float)nx / rupvel;
    int nt = (int)std::floor((risetime + dursf) / doi) + 1;
    if (nt <= 1) nt = 2;
    std::vector<float> grid, tshift;
    for (int ix = 1; ix <= nx; ix++)
        for (int iy = 1; iy <= ny; iy++) {
            const float x = (2.f * ((float)ix - 1.f) - (float)nx + 1.f) / (2.f * (float)nx) * length;
            const float y = (2.f * ((float)iy - 1.f) - (float)ny + 1.f) / (2.f * (float)ny) * length;
            const float r = std::sqrt(x * x + y * y);
            if (r <= radius) {
                grid.push_back(detail::dot3(Rrup[0], x, y, 0.f) + p[1]);
                grid.push_back(detail::dot3(Rrup[1], x, y, 0.f) + p[2]);
                grid.push_back(detail::dot3(Rrup[2], x, y, 0.f) + p[3]);
                tshift.push_back(r / rupvel + p[0]);
            }
        }
    const int np = (int)tshift.size();
    std::vector<float> wt, toff;
    detail::bin_stf(detail::trapezoid_stf(dursf, risetime), dursf + risetime, nt, wt, toff);
    float mr[3][3];
    detail::double_couple(Rslip, np, mr);
    detail::emit(out, grid, tshift, wt, toff, mr);
    out.moment = p[4]; out.risetime = 0.f;
    return true;
}

// psm_set_point_lp + psm_to_tdsm_point_lp (source_point_lp.f90:192-337): a point source whose moment tensor follows
// a band-limited source time function stf(t) (:408-419, default-real exp and sin), sampled every effective dt
inline bool discretize_point_lp(const float *p, float doi, DiscreteSource &out)
{
    const float dur_exc = p[11], prd = p[12];
    int nt = (int)std::floor(dur_exc / doi) + 1;
    if (nt <= 1) nt = 2;
    out.centroids.resize(nt);
    const float t1 = 2.f, t2 = t1 + dur_exc - 5.f, t3 = t2 / 4.f;
    for (int it = 1; it <= nt; it++) {
        const float rel = (float)(it - 1) * doi, d = rel - t3;
        const float tf = std::exp(-(d * d) / (2.f * kPi * dur_exc)) * 1.f / (1.f + std::exp(-2.f * (rel - t1))) * 1.f /
                         (1.f + std::exp(0.5f * (rel - t2))) * std::sin(2.f * kPi / prd * rel);
        Centroid &c = out.centroids[it - 1];
        c.north = p[1]; c.east = p[2]; c.depth = p[3];
        c.time = p[0] + (float)it * doi;
        for (int k = 0; k < 6; k++) c.m[k] = p[5 + k] * tf;
    }
    out.moment = p[4]; out.risetime = 0.f;
    return true;
}

// psm_to_tdsm dispatch, source_all.f90:431-465
inline bool discretize(int type, const float *params, float doi, DiscreteSource &out)
{
    switch (type) {
    case 1: return discretize_bilat(params, doi, out);
    case 2: return discretize_circular(params, doi, out);
    case 3: return discretize_point_lp(params, doi, out);
    case 6: return discretize_moment_tensor(params, doi, out);
    }
    return false;
}

} // namespace kiwi
