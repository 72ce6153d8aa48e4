// AddressSanitizer / UBSan run of the round-6 host code (CPU build only): the optimised march and grid passes against the plain ones
#include "kiwi_host_eikonal.hpp"
#include <cstdio>
#include <random>
using namespace kiwi;
int main()
{
    std::mt19937 rng(12345);
    auto U = [&](float a, float b) { return a + (b - a) * (float)(rng() % 100000) / 100000.f; };
    // 1. the march on random grids (both routines, with and without early termination)
    int bad = 0;
    for (int c = 0; c < 300; c++) {
        const int nx = 1 + rng() % 70, ny = 1 + rng() % 50;
        std::vector<float> sp((size_t)nx * ny);
        const int kind = c % 3;
        for (auto &v : sp) v = kind == 0 ? 3000.f : (kind == 1 ? 1000.f + 500.f * (rng() % 6) : (rng() % 9 == 0 ? 900.f : 2500.f));
        const float mn = *std::min_element(sp.begin(), sp.end());
        const float origin[2] = { U(-5000, 0), U(-5000, 0) }, delta[2] = { U(20, 900), U(20, 900) };
        const float start[2] = { origin[0] + U(-0.2f, 1.2f) * delta[0] * nx, origin[1] + U(-0.2f, 1.2f) * delta[1] * ny };
        for (int dis = 0; dis < 2; dis++) {
            const float discard = dis ? mn : std::numeric_limits<float>::quiet_NaN();
            std::vector<float> a, b;
            eik::fast_marching_plain(sp.data(), nx, ny, origin, delta, start, a, discard);
            eik::fast_marching(sp.data(), nx, ny, origin, delta, start, b, discard);
            for (size_t k = 0; k < a.size(); k++) if (!(dis && sp[k] == discard) && std::memcmp(&a[k], &b[k], 4)) { bad++; break; }
        }
    }
    // 2. the discretiser on random ruptures (optimised path; the plain one through the mode switch)
    float crust[31] = { 1500., 3810., 2500., 4000., 6000., 6400., 6900., 8100., 0., 1940., 1200., 2300., 3500., 3700., 3900., 4600., 1020., 920., 2100., 2400., 2750., 2850., 3000., 3350., 0., 0., 1000., 1000., 10000., 10000., 10000. };
    CrustProfile pr; std::memcpy(&pr, crust, sizeof pr);
    std::vector<HalfSpace> cons(2);
    cons[0] = { { 0, 0, 1500.f }, { 0, 0, -1.f } }; cons[1] = { { 0, 0, 31000.f }, { 0, 0, 1.f } };
    int nok = 0, nrej = 0;
    for (int c = 0; c < 120; c++) {
        const int st = 4 + c % 2;
        float P[20] = { 0 };
        P[0] = U(-1, 1); P[1] = U(-3e3f, 3e3f); P[2] = U(-3e3f, 3e3f); P[3] = U(2e3f, 3e4f); P[4] = 1.f; P[5] = U(-180, 180); P[6] = U(0, 90);
        const int o = st == 5 ? 0 : 1;
        if (st == 4) P[7] = U(-180, 180);
        P[7 + o] = U(-2e3f, 2e3f); P[8 + o] = U(-2e3f, 2e3f); P[9 + o] = U(5e2f, 7e3f);
        P[10 + o] = U(-1, 1) * 0.7f * P[9 + o]; P[11 + o] = U(-1, 1) * 0.7f * P[9 + o]; P[12 + o] = U(0.5f, 1.f);
        if (st == 5) { for (int k = 0; k < 6; k++) P[13 + k] = U(-1, 1) * 1e18f; P[19] = U(0, 3); } else P[14] = U(0, 3);
        const float edt = (c % 3 == 0) ? 0.5f : (c % 3 == 1 ? 1.f : 2.f);
        DiscreteSource a, b;
        eik::fmm_mode() = 0;
        const std::string ea = discretize_eikonal(st, P, edt, pr, cons, a);
        eik::fmm_mode() = 1;
        const std::string eb = discretize_eikonal(st, P, edt, pr, cons, b);
        if (ea != eb) { bad++; continue; }
        if (!ea.empty()) { nrej++; continue; }
        nok++;
        if (a.centroids.size() != b.centroids.size() || std::memcmp(a.centroids.data(), b.centroids.data(), a.centroids.size() * sizeof(Centroid))) bad++;
    }
    std::printf("asan host run: %d bad, %d ruptures discretised, %d rejected, fallbacks %lld\n", bad, nok, nrej, eik::fmm_fallbacks().load());
    return bad != 0;
}
