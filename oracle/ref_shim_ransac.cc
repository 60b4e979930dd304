/*
 * ref_shim_ransac.cc -- TEST INFRASTRUCTURE.  extern "C" shim over the
 * REFERENCE's geometric verification (src/mve/sfm/fundamental.cc,
 * ransac_fundamental.cc, compiled where they lie into oracle/_ref/
 * libref_ransac.so).  No algorithm of its own.
 */
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sfm/correspondence.h"
#include "sfm/fundamental.h"
#include "sfm/ransac_fundamental.h"

extern "C" {

double
ref_sampson_distance(const double *F, const double *p1, const double *p2)
{
    sfm::FundamentalMatrix Fm;
    for (int i = 0; i < 9; ++i) Fm[i] = F[i];
    sfm::Correspondence2D2D m;
    m.p1[0] = p1[0]; m.p1[1] = p1[1]; m.p2[0] = p2[0]; m.p2[1] = p2[1];
    return sfm::sampson_distance(Fm, m);
}

/* fundamental_8_point followed by enforce_fundamental_constraints, as
 * RansacFundamental::estimate_8_point does (ransac_fundamental.cc:92-94) */
void
ref_fundamental_8_point(const double *p1_8x2, const double *p2_8x2, double *F)
{
    sfm::Eight2DPoints a, b;
    for (int i = 0; i < 8; ++i) {
        a(0, i) = p1_8x2[2 * i]; a(1, i) = p1_8x2[2 * i + 1]; a(2, i) = 1.0;
        b(0, i) = p2_8x2[2 * i]; b(1, i) = p2_8x2[2 * i + 1]; b(2, i) = 1.0;
    }
    sfm::FundamentalMatrix Fm;
    sfm::fundamental_8_point(a, b, &Fm);
    sfm::enforce_fundamental_constraints(&Fm);
    for (int i = 0; i < 9; ++i) F[i] = Fm[i];
}

/* RansacFundamental::estimate with std::srand(seed) first; returns the inlier
 * count and writes the inlier ids. */
int
ref_ransac_fundamental(const double *p1, const double *p2, int k, int max_iterations,
    double threshold, unsigned seed, int *inliers, double *F)
{
    sfm::Correspondences2D2D matches(k);
    for (int i = 0; i < k; ++i) {
        matches[i].p1[0] = p1[2 * i]; matches[i].p1[1] = p1[2 * i + 1];
        matches[i].p2[0] = p2[2 * i]; matches[i].p2[1] = p2[2 * i + 1];
    }
    sfm::RansacFundamental::Options o;
    o.max_iterations = max_iterations;
    o.threshold = threshold;
    o.verbose_output = false;
    std::srand(seed);
    sfm::RansacFundamental r(o);
    sfm::RansacFundamental::Result res;
    r.estimate(matches, &res);
    for (std::size_t i = 0; i < res.inliers.size(); ++i) inliers[i] = res.inliers[i];
    for (int i = 0; i < 9; ++i) F[i] = res.fundamental[i];
    return (int)res.inliers.size();
}

}  // extern "C"
