/*
 * ref_shim.cc -- TEST INFRASTRUCTURE.  extern "C" shim over the REFERENCE's
 * own matcher so the oracle restatement (match_oracle.c) can be pinned
 * against it.  It is compiled by oracle/Makefile ONLY when /root/reference
 * is present, together with the reference's
 *   src/mve/sfm/{matching,nearest_neighbor,exhaustive_matching}.cc
 * taken where they lie (nothing is copied into this repo); the output
 * goes to oracle/_ref/libref_match.so (git-ignored).
 *
 * The shim contains no algorithm of its own: it only marshals flat arrays
 * into the reference's types and calls the reference's functions.
 */
#include <cstdint>
#include <cstring>
#include <vector>

#include "sfm/matching.h"
#include "sfm/nearest_neighbor.h"
#include "sfm/exhaustive_matching.h"
#include "sfm/bundler_common.h"
#include "util/aligned_memory.h"

namespace {

template <typename T>
std::vector<T, util::AlignedAllocator<T, 16>>
aligned_copy(const T *p, std::size_t n)
{
    std::vector<T, util::AlignedAllocator<T, 16>> v(n + 8);
    if (n) std::memcpy(v.data(), p, n * sizeof(T));
    return v;
}

template <typename T>
void
twoway(const T *s1, int n1, const T *s2, int n2, int dim, float lowe,
    float dist, int *m12, int *m21)
{
    auto a = aligned_copy(s1, (std::size_t)n1 * dim);
    auto b = aligned_copy(s2, (std::size_t)n2 * dim);
    sfm::Matching::Options o;
    o.descriptor_length = dim;
    o.lowe_ratio_threshold = lowe;
    o.distance_threshold = dist;
    sfm::Matching::Result r;
    sfm::Matching::twoway_match(o, a.data(), n1, b.data(), n2, &r);
    for (std::size_t i = 0; i < r.matches_1_2.size(); ++i) m12[i] = r.matches_1_2[i];
    for (std::size_t i = 0; i < r.matches_2_1.size(); ++i) m21[i] = r.matches_2_1[i];
}

struct RefMatcher
{
    sfm::bundler::ViewportList viewports;
    sfm::ExhaustiveMatching matcher;
};

}  // namespace

extern "C" {

void
ref_nn_find_u16(const uint16_t *q, const uint16_t *el, int n, int dim, int *out4)
{
    auto a = aligned_copy(q, (std::size_t)dim);
    auto b = aligned_copy(el, (std::size_t)n * dim);
    sfm::NearestNeighbor<unsigned short> nn;
    nn.set_elements(b.data());
    nn.set_num_elements(n);
    nn.set_element_dimensions(dim);
    sfm::NearestNeighbor<unsigned short>::Result r;
    nn.find(a.data(), &r);
    out4[0] = r.dist_1st_best; out4[1] = r.dist_2nd_best;
    out4[2] = r.index_1st_best; out4[3] = r.index_2nd_best;
}

void
ref_nn_find_s16(const int16_t *q, const int16_t *el, int n, int dim, int *out4)
{
    auto a = aligned_copy(q, (std::size_t)dim);
    auto b = aligned_copy(el, (std::size_t)n * dim);
    sfm::NearestNeighbor<short> nn;
    nn.set_elements(b.data());
    nn.set_num_elements(n);
    nn.set_element_dimensions(dim);
    sfm::NearestNeighbor<short>::Result r;
    nn.find(a.data(), &r);
    out4[0] = r.dist_1st_best; out4[1] = r.dist_2nd_best;
    out4[2] = r.index_1st_best; out4[3] = r.index_2nd_best;
}

void
ref_twoway_match_u16(const uint16_t *s1, int n1, const uint16_t *s2, int n2,
    int dim, float lowe, float dist, int *m12, int *m21)
{
    twoway<unsigned short>(s1, n1, s2, n2, dim, lowe, dist, m12, m21);
}

void
ref_twoway_match_s16(const int16_t *s1, int n1, const int16_t *s2, int n2,
    int dim, float lowe, float dist, int *m12, int *m21)
{
    twoway<short>(s1, n1, s2, n2, dim, lowe, dist, m12, m21);
}

void
ref_remove_inconsistent(int *m12, int n1, int *m21, int n2)
{
    sfm::Matching::Result r;
    r.matches_1_2.assign(m12, m12 + n1);
    r.matches_2_1.assign(m21, m21 + n2);
    sfm::Matching::remove_inconsistent_matches(&r);
    std::memcpy(m12, r.matches_1_2.data(), sizeof(int) * n1);
    std::memcpy(m21, r.matches_2_1.data(), sizeof(int) * n2);
}

int
ref_count_consistent(const int *m12, int n1, const int *m21, int n2)
{
    sfm::Matching::Result r;
    r.matches_1_2.assign(m12, m12 + n1);
    r.matches_2_1.assign(m21, m21 + n2);
    return sfm::Matching::count_consistent_matches(r);
}

void
ref_combine_results(const int *sift12, int ns1, const int *sift21, int ns2,
    const int *surf12, int nu1, const int *surf21, int nu2,
    int *out12, int *out21)
{
    sfm::Matching::Result a, b, c;
    a.matches_1_2.assign(sift12, sift12 + ns1);
    a.matches_2_1.assign(sift21, sift21 + ns2);
    b.matches_1_2.assign(surf12, surf12 + nu1);
    b.matches_2_1.assign(surf21, surf21 + nu2);
    sfm::Matching::combine_results(a, b, &c);
    std::memcpy(out12, c.matches_1_2.data(), sizeof(int) * c.matches_1_2.size());
    std::memcpy(out21, c.matches_2_1.data(), sizeof(int) * c.matches_2_1.size());
}

/* --- ExhaustiveMatching (init incl. the float -> u16/s16 quantisation) --- */

void *
ref_matcher_create(int num_views)
{
    RefMatcher *m = new RefMatcher();
    m->viewports.resize(num_views);
    return m;
}

void
ref_matcher_set_view(void *h, int view, const float *sift, int n_sift,
    const float *surf, int n_surf)
{
    RefMatcher *m = static_cast<RefMatcher *>(h);
    sfm::FeatureSet &fs = m->viewports[view].features;
    fs.sift_descriptors.resize(n_sift);
    for (int i = 0; i < n_sift; ++i)
        for (int k = 0; k < 128; ++k)
            fs.sift_descriptors[i].data[k] = sift[(std::size_t)i * 128 + k];
    fs.surf_descriptors.resize(n_surf);
    for (int i = 0; i < n_surf; ++i)
        for (int k = 0; k < 64; ++k)
            fs.surf_descriptors[i].data[k] = surf[(std::size_t)i * 64 + k];
}

void
ref_matcher_set_options(void *h, float sift_lowe, float sift_dist,
    float surf_lowe, float surf_dist)
{
    RefMatcher *m = static_cast<RefMatcher *>(h);
    m->matcher.opts.sift_matching_opts.lowe_ratio_threshold = sift_lowe;
    m->matcher.opts.sift_matching_opts.distance_threshold = sift_dist;
    m->matcher.opts.surf_matching_opts.lowe_ratio_threshold = surf_lowe;
    m->matcher.opts.surf_matching_opts.distance_threshold = surf_dist;
}

void
ref_matcher_init(void *h)
{
    RefMatcher *m = static_cast<RefMatcher *>(h);
    m->matcher.init(&m->viewports);
}

/* out12/out21 must hold n_sift+n_surf ints of view 1 / view 2. */
void
ref_matcher_pairwise_match(void *h, int v1, int v2, int *out12, int *len12,
    int *out21, int *len21)
{
    RefMatcher *m = static_cast<RefMatcher *>(h);
    sfm::Matching::Result r;
    m->matcher.pairwise_match(v1, v2, &r);
    *len12 = (int)r.matches_1_2.size();
    *len21 = (int)r.matches_2_1.size();
    std::memcpy(out12, r.matches_1_2.data(), sizeof(int) * r.matches_1_2.size());
    std::memcpy(out21, r.matches_2_1.data(), sizeof(int) * r.matches_2_1.size());
}

int
ref_matcher_pairwise_match_lowres(void *h, int v1, int v2, int num_features)
{
    RefMatcher *m = static_cast<RefMatcher *>(h);
    return m->matcher.pairwise_match_lowres(v1, v2, (std::size_t)num_features);
}

void
ref_matcher_destroy(void *h)
{
    delete static_cast<RefMatcher *>(h);
}

}  // extern "C"
