/*
 * adapter_check.cc -- TEST INFRASTRUCTURE.  Compiled only where the
 * reference sources exist (this container) into oracle/_ref/adapter_check;
 * the binary travels to the GPU box.  It runs the product's MVE adapter
 * (orthosfm_amd/host/mve_hip_matching.h, an sfm::MatchingBase subclass) and
 * the REFERENCE's own sfm::ExhaustiveMatching -- and sfm::CascadeHashing, the
 * application's default -- side by side through the same virtual interface on
 * the same random viewports and requires identical Matching::Result lists and
 * low-res counts.
 */
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <exception>
#include <memory>
#include <random>
#include <vector>

#include "sfm/exhaustive_matching.h"
#include "sfm/cascade_hashing.h"
#include "mve_hip_matching.h"

static void fill_views(sfm::bundler::ViewportList* vl, unsigned seed)
{
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    const int L = 900;
    std::vector<std::vector<float>> bs(L, std::vector<float>(128)), bu(L, std::vector<float>(64));
    for (auto& d : bs) { float n = 0; for (auto& x : d) { x = std::fabs(nd(rng)); n += x * x; } n = std::sqrt(n); for (auto& x : d) x /= n; }
    for (auto& d : bu) { float n = 0; for (auto& x : d) { x = nd(rng); n += x * x; } n = std::sqrt(n); for (auto& x : d) x /= n; }
    const int counts[5][2] = { { 700, 300 }, { 650, 0 }, { 0, 280 }, { 810, 333 }, { 3, 2 } };
    vl->resize(5);
    for (int v = 0; v < 5; ++v) {
        sfm::FeatureSet& fs = (*vl)[v].features;
        fs.sift_descriptors.resize(counts[v][0]);
        for (auto& d : fs.sift_descriptors) {
            const int id = rng() % L; float n = 0;
            for (int k = 0; k < 128; ++k) { float x = std::fabs(bs[id][k] + 0.01f * nd(rng)); d.data[k] = x; n += x * x; }
            n = std::sqrt(n); for (int k = 0; k < 128; ++k) d.data[k] /= n;
        }
        fs.surf_descriptors.resize(counts[v][1]);
        for (auto& d : fs.surf_descriptors) {
            const int id = rng() % L; float n = 0;
            for (int k = 0; k < 64; ++k) { float x = bu[id][k] + 0.03f * nd(rng); d.data[k] = x; n += x * x; }
            n = std::sqrt(n); for (int k = 0; k < 64; ++k) d.data[k] /= n;
        }
        fs.positions.resize(counts[v][0] + counts[v][1]);
    }
}

int main()
{
    sfm::bundler::ViewportList va, vb;
    fill_views(&va, 7);
    fill_views(&vb, 7);
    std::unique_ptr<sfm::MatchingBase> ref(new sfm::ExhaustiveMatching());
    std::unique_ptr<sfm::MatchingBase> hip(new osfm_adapter::HipMatching(0));
    ref->init(&va);
    hip->init(&vb);
    int checked = 0, valid = 0;
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b) {
            if (a == b) continue;
            sfm::Matching::Result r1, r2;
            ref->pairwise_match(a, b, &r1);
            hip->pairwise_match(a, b, &r2);
            if (r1.matches_1_2 != r2.matches_1_2 || r1.matches_2_1 != r2.matches_2_1) {
                std::fprintf(stderr, "MISMATCH pairwise_match(%d,%d)\n", a, b);
                return 1;
            }
            for (int nf : { 100, 500 }) {
                const int c1 = ref->pairwise_match_lowres(a, b, nf), c2 = hip->pairwise_match_lowres(a, b, nf);
                if (c1 != c2) { std::fprintf(stderr, "MISMATCH lowres(%d,%d,%d): %d vs %d\n", a, b, nf, c1, c2); return 1; }
            }
            for (int m : r1.matches_1_2) valid += m >= 0;
            checked++;
        }
    /* cascade hashing: the adapter's default layout is sfm::CascadeHashing's own (a type
     * that either view lacks contributes no entries, cascade_hashing.h:341-342), so the
     * Result vectors are compared as they are */
    sfm::bundler::ViewportList vc, vd;
    fill_views(&vc, 7);
    fill_views(&vd, 7);
    std::unique_ptr<sfm::MatchingBase> cref(new sfm::CascadeHashing());
    std::unique_ptr<sfm::MatchingBase> chip(new osfm_adapter::HipMatching(0, OSFM_MATCHER_CASCADE_HASHING));
    cref->init(&vc);
    chip->init(&vd);
    int cchecked = 0, cvalid = 0;
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b) {
            if (a == b) continue;
            sfm::Matching::Result r1, r2;
            cref->pairwise_match(a, b, &r1);
            chip->pairwise_match(a, b, &r2);
            if (r1.matches_1_2 != r2.matches_1_2 || r1.matches_2_1 != r2.matches_2_1) {
                std::fprintf(stderr, "MISMATCH cascade pairwise_match(%d,%d)\n", a, b);
                return 1;
            }
            const int c1 = cref->pairwise_match_lowres(a, b, 500), c2 = chip->pairwise_match_lowres(a, b, 500);
            if (c1 != c2) { std::fprintf(stderr, "MISMATCH cascade lowres(%d,%d): %d vs %d\n", a, b, c1, c2); return 1; }
            for (int m : r1.matches_1_2) cvalid += m >= 0;
            cchecked++;
        }
    std::printf("adapter_check ok: %d pairs identical to sfm::CascadeHashing, %d valid matches\n", cchecked, cvalid);

    /* The caller's loop (bundler::Matching::compute, bundler_matching.cc:74-79,149,162):
     * pairwise_match / pairwise_match_lowres invoked concurrently from an OpenMP team on
     * one const matcher.  Every thread's results must equal the serial ones. */
    {
        const int num_pairs = 5 * 4 / 2;
        std::vector<sfm::Matching::Result> serial(num_pairs), par(num_pairs);
        std::vector<int> low_serial(num_pairs), low_par(num_pairs);
        for (int i = 0; i < num_pairs; ++i) {
            const int v1 = (int)(0.5 + std::sqrt(0.25 + 2.0 * i)), v2 = i - v1 * (v1 - 1) / 2;
            hip->pairwise_match(v1, v2, &serial[i]);
            low_serial[i] = hip->pairwise_match_lowres(v1, v2, 500);
        }
        int failed = 0;
        for (int round = 0; round < 3; ++round) {
#pragma omp parallel for schedule(dynamic) num_threads(8)
            for (int i = 0; i < num_pairs; ++i) {
                const int v1 = (int)(0.5 + std::sqrt(0.25 + 2.0 * i)), v2 = i - v1 * (v1 - 1) / 2;
                try {
                    low_par[i] = hip->pairwise_match_lowres(v1, v2, 500);
                    hip->pairwise_match(v1, v2, &par[i]);
                } catch (std::exception const& e) {
#pragma omp critical
                    { std::fprintf(stderr, "exception in thread: %s\n", e.what()); failed++; }
                }
            }
            for (int i = 0; i < num_pairs; ++i)
                if (par[i].matches_1_2 != serial[i].matches_1_2 || par[i].matches_2_1 != serial[i].matches_2_1 ||
                    low_par[i] != low_serial[i]) failed++;
        }
        if (failed) { std::fprintf(stderr, "MISMATCH: %d concurrent calls differ from the serial results\n", failed); return 1; }
        std::printf("adapter_check ok: %d pairs x 3 rounds from 8 OpenMP threads identical to the serial calls\n", num_pairs);
    }

    /* Two and three logical devices behind the same virtual interface (osfm_match_create_multi;
     * on a one-GPU box device 0 several times): identical to sfm::ExhaustiveMatching from the
     * serial loop and from the OpenMP team. */
    for (int nd : { 2, 3 }) {
        sfm::bundler::ViewportList ve;
        fill_views(&ve, 7);
        std::unique_ptr<sfm::MatchingBase> multi(new osfm_adapter::HipMatching(std::vector<int>(nd, 0)));
        multi->init(&ve);
        int bad = 0;
        const int num_pairs = 5 * 4;
#pragma omp parallel for schedule(dynamic) num_threads(8) reduction(+:bad)
        for (int i = 0; i < num_pairs; ++i) {
            const int a = i / 4, b = (i % 4) + ((i % 4) >= a ? 1 : 0);
            sfm::Matching::Result r1, r2;
            ref->pairwise_match(a, b, &r1);
            multi->pairwise_match(a, b, &r2);
            if (r1.matches_1_2 != r2.matches_1_2 || r1.matches_2_1 != r2.matches_2_1) bad++;
            if (ref->pairwise_match_lowres(a, b, 500) != multi->pairwise_match_lowres(a, b, 500)) bad++;
        }
        if (bad) { std::fprintf(stderr, "MISMATCH: %d calls on %d logical devices differ from sfm::ExhaustiveMatching\n", bad, nd); return 1; }
        std::printf("adapter_check ok: %d pairs on %d logical devices identical to sfm::ExhaustiveMatching\n", num_pairs, nd);
    }

    bool threw = false;
    try { hip->init(nullptr); } catch (std::invalid_argument const&) { threw = true; }
    if (!threw) { std::fprintf(stderr, "init(nullptr) did not throw\n"); return 1; }
    std::printf("adapter_check ok: %d pairs identical to sfm::ExhaustiveMatching, %d valid matches\n", checked, valid);
    return 0;
}
