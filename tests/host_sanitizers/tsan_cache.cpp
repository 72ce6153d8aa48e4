// ThreadSanitizer run of the solve cache (hits, misses, eviction with storage reuse, miss-streak bypass) under 8 threads
#include "kiwi_host_eikonal.hpp"
#include <cstdio>
#include <thread>
using namespace kiwi;
int main()
{
    float crust[31] = { 1500., 3810., 2500., 4000., 6000., 6400., 6900., 8100., 0., 1940., 1200., 2300., 3500., 3700., 3900., 4600., 1020., 920., 2100., 2400., 2750., 2850., 3000., 3350., 0., 0., 1000., 1000., 10000., 10000., 10000. };
    CrustProfile pr; std::memcpy(&pr, crust, sizeof pr);
    std::vector<HalfSpace> cons(2);
    cons[0] = { { 0, 0, 6500.f }, { 0, 0, -1.f } }; cons[1] = { { 0, 0, 15500.f }, { 0, 0, 1.f } };
    // reference tables without the cache
    const int NP = 40;
    std::vector<std::vector<Centroid>> want(NP);
    auto params = [](int k, float *P) {
        const float base[20] = { 0, 0, 0, 11000.f, 1.f, 91.f, 90.f, 0, 0, 3000.f, 400.f, 0, 0.9f, 1e20f, 2e19f, -3e19f, 1e19f, 5e19f, -2e19f, 1.0f };
        std::memcpy(P, base, sizeof base);
        P[10] = 400.f + 60.f * (k % NP);          // NP distinct nucleation points -> NP distinct solves (more than the cache holds)
    };
    eik::SolveCache::get().enabled = false;
    for (int k = 0; k < NP; k++) { float P[20]; params(k, P); DiscreteSource d; discretize_eikonal(5, P, 2.0f, pr, cons, d); want[k] = d.centroids; }
    eik::SolveCache::get().enabled = true;
    std::atomic<int> bad{ 0 };
    std::vector<std::thread> th;
    for (int t = 0; t < 8; t++)
        th.emplace_back([&, t] {
            for (int it = 0; it < 120; it++) {
                const int k = (it * 7 + t * 3) % NP;
                float P[20]; params(k, P);
                P[1] = 100.f * t;                 // north shift: same solve, another table
                DiscreteSource d;
                discretize_eikonal(5, P, 2.0f, pr, cons, d);
                if (d.centroids.size() != want[k].size()) { bad++; continue; }
                for (size_t i = 0; i < d.centroids.size(); i++) if (std::memcmp(d.centroids[i].m, want[k][i].m, sizeof d.centroids[i].m)) { bad++; break; }
            }
        });
    for (auto &x : th) x.join();
    auto &sc = eik::SolveCache::get();
    std::printf("tsan cache run: %d bad, hits %lld misses %lld\n", bad.load(), sc.hits.load(), sc.misses.load());
    return bad != 0;
}
