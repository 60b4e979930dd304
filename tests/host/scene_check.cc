/*
 * scene_check.cc -- TEST INFRASTRUCTURE (built by tests/host/Makefile, run on the GPU box by
 * tests/test_e2e_gpu.py).  A C++ caller of the device-resident scene (include/osfm_hip.h, osfm_scene_*): the
 * steps runPoseEstimation takes for its first two camera groups (src/sfm/reconstruct.cpp:193-281) -- local
 * adjustment of a group, its cameras into the scene, triangulation, a second group with one new view, the
 * incremental triangulation checked against a full pass, the global adjustment, both filters, the state back.
 * Reads the track table and the start poses from a binary file the test wrote, prints cameras, flags and points
 * with 17 significant digits; the test runs the same steps through orthosfm_amd/scene.py and compares: bit for bit.
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "osfm_hip.h"

#define CHECK(call) do { int st_ = (call); if (st_ != OSFM_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, st_, osfm_last_error()); return 1; } } while (0)

template <typename T> static bool rd(FILE *f, std::vector<T> &v, size_t n) { v.resize(n); return n == 0 || fread(v.data(), sizeof(T), n, f) == n; }

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[6];     // model, views, tracks, width, height, group size
    if (fread(hdr, 4, 6, f) != 6) return 2;
    const int model = hdr[0], V = hdr[1], T = hdr[2], W = hdr[3], H = hdr[4], G = hdr[5];
    std::vector<int64_t> offsets;
    std::vector<int32_t> view, g1, g2;
    std::vector<float> xy;
    std::vector<double> p1, p2;
    std::vector<uint8_t> c1, c2;
    if (!rd(f, offsets, (size_t)T + 1)) return 2;
    const size_t F = (size_t)offsets[T];
    if (!rd(f, view, F) || !rd(f, xy, 2 * F) || !rd(f, g1, (size_t)G) || !rd(f, p1, 7 * (size_t)G) || !rd(f, c1, 7 * (size_t)G) ||
        !rd(f, g2, (size_t)G) || !rd(f, p2, 7 * (size_t)G) || !rd(f, c2, 7 * (size_t)G)) return 2;
    fclose(f);

    std::vector<int32_t> w(V, W), h(V, H);
    osfm_scene *sc = nullptr;
    CHECK(osfm_scene_create(0, model, V, w.data(), h.data(), T, offsets.data(), view.data(), xy.data(), &sc));
    osfm_ba_options o, ol;
    CHECK(osfm_ba_options_default(&o));
    ol = o; ol.retriangulate_points = 1;
    osfm_ba_summary s;
    int32_t M = 0, O = 0, bad = 0, killed = 0;
    // group 1: local adjustment, its cameras join, every track triangulated
    CHECK(osfm_scene_local_adjustment(sc, G, g1.data(), p1.data(), c1.data(), 1.5, &ol, &s, &M, &O));
    printf("local1 %d %d %d\n", M, O, s.num_iterations);
    CHECK(osfm_scene_align_views(sc, G, g1.data(), p1.data(), c1.data()));
    CHECK(osfm_scene_triangulate(sc, 0, nullptr, 0, nullptr));
    // group 2: its last view is new; the cameras it shares with group 1 start from the scene's
    std::vector<int32_t> cv(V);
    std::vector<double> cp(7 * (size_t)V);
    int32_t nc = 0;
    CHECK(osfm_scene_get_cameras(sc, V, cv.data(), cp.data(), &nc));
    for (int i = 0; i < G; ++i)
        for (int c = 0; c < nc; ++c)
            if (cv[c] == g2[i]) for (int k = 0; k < 7; ++k) p2[7 * i + k] = cp[7 * c + k];
    CHECK(osfm_scene_local_adjustment(sc, G, g2.data(), p2.data(), c2.data(), 1.5, &ol, &s, &M, &O));
    printf("local2 %d %d %d\n", M, O, s.num_iterations);
    CHECK(osfm_scene_align_views(sc, 1, &g2[G - 1], &p2[7 * (G - 1)], &c2[7 * (G - 1)]));
    CHECK(osfm_scene_triangulate(sc, 1, &g2[G - 1], 1, &bad));
    printf("incremental_mismatches %d\n", bad);
    CHECK(osfm_scene_global_adjustment(sc, &o, &s, &M, &O));
    printf("global %d %d %d %.17g\n", M, O, s.num_iterations, s.final_cost);
    CHECK(osfm_scene_filter_outliers(sc, nullptr, &killed));
    printf("outliers %d\n", killed);
    CHECK(osfm_scene_filter_reprojection(sc, 1.5));
    CHECK(osfm_scene_get_cameras(sc, V, cv.data(), cp.data(), &nc));
    for (int c = 0; c < nc; ++c) {
        printf("cam %d", cv[c]);
        for (int k = 0; k < 7; ++k) printf(" %.17g", cp[7 * c + k]);
        printf("\n");
    }
    std::vector<uint8_t> at(T), af(F), hp(T);
    std::vector<double> pt(4 * (size_t)T);
    CHECK(osfm_scene_download(sc, at.data(), af.data(), hp.data(), pt.data()));
    long na = 0, nf = 0, np = 0;
    for (int t = 0; t < T; ++t) { na += at[t]; np += at[t] && hp[t]; }
    for (size_t i = 0; i < F; ++i) nf += af[i];
    printf("alive %ld %ld %ld\n", na, nf, np);
    for (int t = 0; t < T; ++t)
        if (at[t] && hp[t]) printf("pt %d %.17g %.17g %.17g %.17g\n", t, pt[4 * t], pt[4 * t + 1], pt[4 * t + 2], pt[4 * t + 3]);
    CHECK(osfm_scene_destroy(sc));
    return 0;
}
