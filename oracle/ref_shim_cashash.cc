/*
 * ref_shim_cashash.cc -- TEST INFRASTRUCTURE.  extern "C" shim over the
 * REFERENCE's own cascade-hashing matcher (src/mve/sfm/cascade_hashing.{h,cc},
 * compiled where they lie by oracle/Makefile into
 * oracle/_ref/libref_cashash.so) so that cashash_oracle.c can be pinned against
 * it, stage by stage: projection matrices, hashes, bucket ids, match results.
 * No algorithm here.  The private members are read through the usual test
 * back door (private -> public for this translation unit only).
 */
#include <cstdint>
#include <cstring>
#include <vector>

/* everything cascade_hashing.h pulls in first, so that only its own class is
 * affected by the access override */
#include <iostream>
#include <random>
#include <sstream>
#include "math/functions.h"
#include "math/vector.h"
#include "sfm/defines.h"
#include "sfm/exhaustive_matching.h"
#include "sfm/matching.h"
#include "sfm/sift.h"
#include "sfm/surf.h"
#include "sfm/bundler_common.h"
#include "util/system.h"
#include "util/timer.h"
#define private public
#include "sfm/cascade_hashing.h"
#undef private

namespace {
struct RefCasHash
{
    sfm::bundler::ViewportList viewports;
    sfm::CascadeHashing matcher;
};
}

#define SHIM_API extern "C" __attribute__((visibility("default")))

SHIM_API void *
ref_cashash_create(int num_views)
{
    RefCasHash *m = new RefCasHash();
    m->viewports.resize(num_views);
    return m;
}

SHIM_API void
ref_cashash_destroy(void *h) { delete static_cast<RefCasHash *>(h); }

SHIM_API void
ref_cashash_set_view(void *h, int view, const float *sift, int n_sift, const float *surf, int n_surf)
{
    RefCasHash *m = static_cast<RefCasHash *>(h);
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

SHIM_API void
ref_cashash_init(void *h)
{
    RefCasHash *m = static_cast<RefCasHash *>(h);
    m->matcher.init(&m->viewports);
}

/* prim [dim][dim], sec [groups][bits][dim]; type 0 = SIFT (dim 128), 1 = SURF (dim 64) */
SHIM_API void
ref_cashash_get_proj(void *h, int type, float *prim, float *sec)
{
    RefCasHash *m = static_cast<RefCasHash *>(h);
    if (type == 0) {
        auto const &p = m->matcher.global_data.sift;
        for (std::size_t i = 0; i < p.prim_proj_mat.size(); ++i)
            for (int k = 0; k < 128; ++k) prim[i * 128 + k] = p.prim_proj_mat[i][k];
        std::size_t o = 0;
        for (auto const &g : p.sec_proj_mats) for (auto const &v : g) for (int k = 0; k < 128; ++k) sec[o++] = v[k];
    } else {
        auto const &p = m->matcher.global_data.surf;
        for (std::size_t i = 0; i < p.prim_proj_mat.size(); ++i)
            for (int k = 0; k < 64; ++k) prim[i * 64 + k] = p.prim_proj_mat[i][k];
        std::size_t o = 0;
        for (auto const &g : p.sec_proj_mats) for (auto const &v : g) for (int k = 0; k < 64; ++k) sec[o++] = v[k];
    }
}

/* hashes [n][dim/64] u64, bucket_ids [groups][n] u16 of one view */
SHIM_API void
ref_cashash_get_local(void *h, int type, int view, uint64_t *hashes, uint16_t *bucket_ids)
{
    RefCasHash *m = static_cast<RefCasHash *>(h);
    auto const &ld = type == 0 ? m->matcher.local_data_sift[view] : m->matcher.local_data_surf[view];
    std::memcpy(hashes, ld.comp_hash_data.data(), ld.comp_hash_data.size() * sizeof(uint64_t));
    std::size_t o = 0;
    for (auto const &g : ld.bucket_grps_bucket_ids) for (uint16_t b : g) bucket_ids[o++] = b;
}

SHIM_API void
ref_cashash_pairwise_match(void *h, int v1, int v2, int *out12, int *len12, int *out21, int *len21)
{
    RefCasHash *m = static_cast<RefCasHash *>(h);
    sfm::Matching::Result r;
    m->matcher.pairwise_match(v1, v2, &r);
    *len12 = (int)r.matches_1_2.size();
    *len21 = (int)r.matches_2_1.size();
    std::memcpy(out12, r.matches_1_2.data(), sizeof(int) * r.matches_1_2.size());
    std::memcpy(out21, r.matches_2_1.data(), sizeof(int) * r.matches_2_1.size());
}
