// Drop-in matcher plug-in for OrthoSfM / MVE: an sfm::MatchingBase
// implementation (src/mve/sfm/matching_base.h:22-55) that forwards to the
// MI355X backend through the C ABI of include/osfm_hip.h.
//
// This header is compiled INSIDE the OrthoSfM source tree (it includes MVE's
// own headers by their in-tree names); it adds no algorithm of its own.  To
// use it, add one enumerator and one switch case next to MATCHER_EXHAUSTIVE
// (src/mve/sfm/bundler_matching.h:52-56, bundler_matching.cc:31-41) -- see
// INTEGRATION.md.
//
// Error behaviour mirrors MVE: invalid arguments / backend failures throw
// (std::invalid_argument for a null viewport list as bundler_matching.cc:47-48,
// std::runtime_error otherwise); unsuccessful matches are -1.
#ifndef OSFM_MVE_HIP_MATCHING_HEADER
#define OSFM_MVE_HIP_MATCHING_HEADER

#include <stdexcept>
#include <string>
#include <vector>

#include "sfm/bundler_common.h"
#include "sfm/matching.h"
#include "sfm/matching_base.h"

#include "osfm_hip.h"

namespace osfm_adapter {

class HipMatching : public sfm::MatchingBase
{
public:
    /* matcher_type: OSFM_MATCHER_EXHAUSTIVE (sfm::ExhaustiveMatching) or
     * OSFM_MATCHER_CASCADE_HASHING (sfm::CascadeHashing, the application's default) */
    explicit HipMatching (int device = 0, int matcher_type = OSFM_MATCHER_EXHAUSTIVE)
        : devices(1, device), matcher_type(matcher_type), handle(nullptr) {}
    /* Several devices of the node behind the one matcher the single-process caller holds
     * (osfm_match_create_multi): the OpenMP team of bundler::Matching::compute
     * (bundler_matching.cc:86-88) is spread over them call by call. */
    explicit HipMatching (std::vector<int> const& device_ids, int matcher_type = OSFM_MATCHER_EXHAUSTIVE)
        : devices(device_ids), matcher_type(matcher_type), handle(nullptr) {}
    HipMatching (HipMatching const&) = delete;
    HipMatching& operator= (HipMatching const&) = delete;

    ~HipMatching (void) override
    {
        if (this->handle != nullptr)
            osfm_match_destroy(this->handle);
    }

    /* ExhaustiveMatching::init: quantise and upload every view's descriptors. */
    void init (sfm::bundler::ViewportList* viewports) override
    {
        if (viewports == nullptr)
            throw std::invalid_argument("Viewports must not be null");
        if (this->handle != nullptr)
        {
            osfm_match_destroy(this->handle);
            this->handle = nullptr;
        }
        // the library fills public structs in full: one built from another header must not run
        if (osfm_version() != OSFM_ABI_VERSION)
            throw std::runtime_error("osfm: libosfm_hip.so has ABI version " + std::to_string(osfm_version()) +
                ", this adapter was built against " + std::to_string(OSFM_ABI_VERSION));
        osfm_match_options o;
        osfm_match_options_default(&o);
        o.sift_lowe_ratio = this->opts.sift_matching_opts.lowe_ratio_threshold;
        o.sift_distance_threshold = this->opts.sift_matching_opts.distance_threshold;
        o.surf_lowe_ratio = this->opts.surf_matching_opts.lowe_ratio_threshold;
        o.surf_distance_threshold = this->opts.surf_matching_opts.distance_threshold;
        o.matcher_type = this->matcher_type;
        if (this->devices.size() == 1)
            check(osfm_match_create(this->devices[0], (int)viewports->size(), &o, &this->handle));
        else
            check(osfm_match_create_multi(this->devices.data(), (int)this->devices.size(),
                (int)viewports->size(), &o, &this->handle));

        std::vector<float> sift, surf;
        for (std::size_t v = 0; v < viewports->size(); ++v)
        {
            sfm::FeatureSet const& fs = (*viewports)[v].features;
            sift.resize(fs.sift_descriptors.size() * 128);
            for (std::size_t i = 0; i < fs.sift_descriptors.size(); ++i)
                for (int k = 0; k < 128; ++k)
                    sift[i * 128 + k] = fs.sift_descriptors[i].data[k];
            surf.resize(fs.surf_descriptors.size() * 64);
            for (std::size_t i = 0; i < fs.surf_descriptors.size(); ++i)
                for (int k = 0; k < 64; ++k)
                    surf[i * 64 + k] = fs.surf_descriptors[i].data[k];
            check(osfm_match_set_view_float(this->handle, (int)v,
                sift.data(), (int)fs.sift_descriptors.size(),
                surf.data(), (int)fs.surf_descriptors.size()));
        }
    }

    void pairwise_match (int view_1_id, int view_2_id,
        sfm::Matching::Result* result) const override
    {
        int ns1 = 0, nu1 = 0, ns2 = 0, nu2 = 0;
        check(osfm_match_view_size(this->handle, view_1_id, &ns1, &nu1));
        check(osfm_match_view_size(this->handle, view_2_id, &ns2, &nu2));
        result->matches_1_2.assign(ns1 + nu1 + 1, -1);
        result->matches_2_1.assign(ns2 + nu2 + 1, -1);
        int32_t len12 = 0, len21 = 0;
        check(osfm_match_pair(this->handle, view_1_id, view_2_id,
            result->matches_1_2.data(), &len12, result->matches_2_1.data(), &len21));
        result->matches_1_2.resize(len12);
        result->matches_2_1.resize(len21);
    }

    int pairwise_match_lowres (int view_1_id, int view_2_id,
        std::size_t num_features) const override
    {
        int32_t count = 0;
        check(osfm_match_pair_lowres(this->handle, view_1_id, view_2_id,
            (int)num_features, &count));
        return count;
    }

    osfm_matcher* native (void) const { return this->handle; }

private:
    static void check (int status)
    {
        if (status != OSFM_OK)
            throw std::runtime_error(std::string("osfm: ") + osfm_last_error());
    }

    std::vector<int> devices;
    int matcher_type;
    osfm_matcher* handle;
};

}  // namespace osfm_adapter

#endif /* OSFM_MVE_HIP_MATCHING_HEADER */
