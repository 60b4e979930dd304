// The unpatched drop-in path in its own language: MatchingBase::pairwise_match_lowres +
// pairwise_match per pair, called through the C ABI from an OpenMP loop the way
// sfm::bundler::Matching::compute does (bundler_matching.cc:86-88,149,162), serial and with
// 16 threads.  Views: noisy copies of one set of landmark descriptors (every pair matches).
//   ./drop_in_bench [views] [features] [threads]
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "osfm_hip.h"

int main(int argc, char **argv)
{
    const int V = argc > 1 ? std::atoi(argv[1]) : 24, F = argc > 2 ? std::atoi(argv[2]) : 20000;
    std::mt19937 rng(5);
    std::normal_distribution<float> nd;
    std::vector<float> land((size_t)F * 128);
    for (int i = 0; i < F; ++i) {
        double n2 = 0;
        for (int k = 0; k < 128; ++k) { float v = std::fabs(nd(rng)); land[(size_t)i * 128 + k] = v; n2 += v * v; }
        const float s = (float)(1.0 / std::sqrt(n2));
        for (int k = 0; k < 128; ++k) land[(size_t)i * 128 + k] *= s;
    }
    osfm_match_options o;
    osfm_match_options_default(&o);
    osfm_matcher *m = nullptr;
    if (osfm_match_create(0, V, &o, &m) != OSFM_OK) { std::fprintf(stderr, "create: %s\n", osfm_last_error()); return 1; }
    std::vector<float> view((size_t)F * 128);
    std::vector<int> perm(F);
    for (int v = 0; v < V; ++v) {
        for (int i = 0; i < F; ++i) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), rng);
        for (int i = 0; i < F; ++i) {
            double n2 = 0;
            float *d = &view[(size_t)i * 128];
            for (int k = 0; k < 128; ++k) { d[k] = std::fabs(land[(size_t)perm[i] * 128 + k] + 0.02f * nd(rng)); n2 += d[k] * d[k]; }
            const float s = (float)(1.0 / std::sqrt(n2));
            for (int k = 0; k < 128; ++k) d[k] *= s;
        }
        if (osfm_match_set_view_float(m, v, view.data(), F, nullptr, 0) != OSFM_OK) { std::fprintf(stderr, "set_view: %s\n", osfm_last_error()); return 1; }
    }
    const int P = V * (V - 1) / 2;
    long long total = 0;
    const int many = argc > 3 ? std::atoi(argv[3]) : 16;
    for (int threads : {1, many}) {
        long long matches = 0;
        int failed = 0;
        const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic) num_threads(threads) reduction(+ : matches, failed)
        for (int i = 0; i < P; ++i) {
            const int v1 = (int)(0.5 + std::sqrt(0.25 + 2.0 * i)), v2 = i - v1 * (v1 - 1) / 2;
            std::vector<int32_t> m12(F), m21(F);
            int32_t low = 0, l12 = 0, l21 = 0;
            if (osfm_match_pair_lowres(m, v1, v2, 500, &low) != OSFM_OK) { failed++; continue; }
            if (low < 5) continue;
            if (osfm_match_pair(m, v1, v2, m12.data(), &l12, m21.data(), &l21) != OSFM_OK) { failed++; continue; }
            for (int k = 0; k < l12; ++k) matches += m12[k] >= 0;
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (failed) { std::fprintf(stderr, "%d calls failed: %s\n", failed, osfm_last_error()); return 1; }
        if (threads == 1) total = matches;
        else if (matches != total) { std::fprintf(stderr, "MISMATCH: %lld matches with the thread team, %lld serial\n", matches, total); return 1; }
        std::printf("{\"threads\": %d, \"pairs\": %d, \"features\": %d, \"seconds\": %.4f, \"pairs_per_s\": %.1f, \"mutual_matches\": %lld}\n",
            threads, P, F, dt, P / dt, matches);
    }
    osfm_match_destroy(m);
    return 0;
}
