// Elimination order of the Schur-reduced camera system (host code).
//
// The reference asks Ceres for SPARSE_SCHUR with SUITE_SPARSE (bundle_adjustment.cpp:126-133): the reduced
// camera matrix is factored by CHOLMOD behind a fill-reducing ordering.  This backend factors it with a dense
// blocked Cholesky whose critical path is one diagonal block after the other (ba_cholesky.hip: POTRF(j) ->
// TRSM(j + 1, j) -> SYRK -> POTRF(j + 1), ~6 us per 32-unknown block column).  Where the cameras form a ring or
// a strip -- every track is seen by a short run of neighbouring views, as on the reference's turntable sets --
// the matrix is block banded (with two corner blocks for a closed ring), and the chain need not be that long:
// cut the ring into K arcs by K separators of w cameras (w = the longest run a track spans), lay the arcs out
// first, each from a block boundary on, the separators behind them.  No block of one arc is coupled to a block
// of another, so the K arcs are factored side by side and only the separators' rows wait for all of them: the
// chain is one arc plus the separators instead of the whole ring.
//
// What the kernels need of it: the cameras' offsets (interior padding included: an arc ends at a block boundary,
// the unknowns that fill it up are identity rows), the block pattern of the factor (symbolic fill included) so
// that structurally zero tiles are neither waited for nor multiplied, and the list of the tiles outside the
// band the D workgroups own.  Everything else -- pair pass, camera update -- addresses unknowns through cam_off
// and does not care where they are.
#include <algorithm>
#include <cstdlib>

#include "ba_solve.h"

namespace osfm {

namespace {

struct BlockPattern {
    int nblk = 0;
    std::vector<unsigned long long> nz;      // [(nblk + 1)][kNzWords], lower triangle + the right-hand side's row
    bool get(int i, int k) const { return (nz[(size_t)i * kNzWords + (k >> 6)] >> (k & 63)) & 1ull; }
    void set(int i, int k) { nz[(size_t)i * kNzWords + (k >> 6)] |= 1ull << (k & 63); }
};

// pattern of the lower triangle from the camera pairs under the layout `off`, then the fill of the factorisation
void block_pattern(int nblk, const std::vector<std::pair<int, int>> &pairs, const int32_t *ldim, const std::vector<int32_t> &off, BlockPattern *P)
{
    P->nblk = nblk;
    P->nz.assign((size_t)(nblk + 1) * kNzWords, 0ull);
    for (int k = 0; k < nblk; ++k) { P->set(k, k); P->set(nblk, k); }
    for (const auto &pr : pairs) {
        const int a = pr.first, b = pr.second;
        if (ldim[a] == 0 || ldim[b] == 0) continue;
        const int ra0 = off[a] / 32, ra1 = (off[a] + ldim[a] - 1) / 32, rb0 = off[b] / 32, rb1 = (off[b] + ldim[b] - 1) / 32;
        for (int x = ra0; x <= ra1; ++x)
            for (int y = rb0; y <= rb1; ++y) P->set(std::max(x, y), std::min(x, y));
    }
    // right-looking symbolic factorisation: eliminating column k couples every pair of rows below it
    std::vector<int> rows;
    for (int k = 0; k < nblk; ++k) {
        rows.clear();
        for (int i = k + 1; i < nblk; ++i) if (P->get(i, k)) rows.push_back(i);
        for (size_t x = 0; x < rows.size(); ++x)
            for (size_t y = 0; y < x; ++y) P->set(rows[x], rows[y]);
    }
}

// length of the longest chain of diagonal blocks: block j cannot be factored before every block k < j with a tile (j, k)
int chain_length(const BlockPattern &P)
{
    std::vector<int> depth(P.nblk, 1);
    int best = 0;
    for (int j = 0; j < P.nblk; ++j) {
        for (int k = 0; k < j; ++k) if (P.get(j, k)) depth[j] = std::max(depth[j], depth[k] + 1);
        best = std::max(best, depth[j]);
    }
    return best;
}

}  // namespace

bool choose_reduced_order(int C, const int32_t *ldim, const std::vector<std::pair<int, int>> &pairs, ReducedOrder *out)
{
    out->active = false;
    int nc = 0, free_cams = 0;
    for (int c = 0; c < C; ++c) { nc += ldim[c]; free_cams += ldim[c] != 0; }
    const int nblk0 = (nc + 31) / 32;
    out->chain_natural = nblk0;
    if (nblk0 < 8 || free_cams < 16 || nblk0 + 8 > kFlowOrderMaxBlocks) return false;
    // the longest run of views a track spans, on the ring of camera indices
    int w = 0;
    for (const auto &pr : pairs) {
        if (ldim[pr.first] == 0 || ldim[pr.second] == 0) continue;
        const int d = std::abs(pr.first - pr.second);
        w = std::max(w, std::min(d, C - d));
    }
    if (w == 0 || 4 * w > C) return false;                 // not a band: the dense order stands
    std::vector<int32_t> natural(C, 0);
    for (int c = 0, t = 0; c < C; ++c) { natural[c] = t; t += ldim[c]; }
    {
        BlockPattern P;
        block_pattern(nblk0, pairs, ldim, natural, &P);
        out->chain_natural = chain_length(P);
    }
    ReducedOrder best;
    const int forced = getenv("OSFM_BA_ORDER_ARCS") ? atoi(getenv("OSFM_BA_ORDER_ARCS")) : 0;     // experiments: this K or none
    for (int K = 2; K <= 8; ++K) {
        if (forced && K != forced) continue;
        if ((C - K * w) / K < w) break;                    // arcs shorter than a separator: no use
        ReducedOrder o;
        o.cam_off.assign(C, 0);
        o.arcs = K; o.sep_cams = w;
        // separator t: the w cameras from p_t on; arc t: the cameras between separator t and separator t + 1
        std::vector<int> p(K + 1);
        for (int t = 0; t <= K; ++t) p[t] = (int)((long long)t * C / K);
        int offv = 0;
        for (int t = 0; t < K; ++t) {
            for (int c = p[t] + w; c < p[t + 1]; ++c) { o.cam_off[c] = offv; offv += ldim[c]; }
            while (offv % 32) o.pad.push_back(offv++);     // the next arc starts a block of its own
        }
        for (int t = 0; t < K; ++t)
            for (int c = p[t]; c < p[t] + w; ++c) { o.cam_off[c] = offv; offv += ldim[c]; }
        o.span = offv;
        o.nblk = (offv + 31) / 32;
        if (o.nblk > kFlowOrderMaxBlocks) continue;
        BlockPattern P;
        block_pattern(o.nblk, pairs, ldim, o.cam_off, &P);
        o.chain_ordered = chain_length(P);
        // (ties go to the larger K: with the same chain of diagonal blocks the arcs are shorter, and the separators'
        //  rows -- tiles that follow an arc through the slower workgroup-to-workgroup hand-off -- finish sooner behind
        //  them: config 4 with K = 3 / 4 / 5, all 14 blocks: 145 / 138 / 134 us per factorisation and solve)
        if (!best.active || o.chain_ordered <= best.chain_ordered) {
            o.nz = P.nz;
            o.active = true;
            best = std::move(o);
        }
    }
    if (!best.active || 4 * best.chain_ordered > 3 * out->chain_natural) return false;     // less than a quarter shorter: not worth the padding
    best.chain_natural = out->chain_natural;
    // the tiles below the band the D workgroups own (rows more than kFlowBand under the diagonal), column-major
    BlockPattern P;
    P.nblk = best.nblk; P.nz = best.nz;
    for (int j = 0; j < best.nblk; ++j)
        for (int i = j + kFlowBand + 1; i <= best.nblk; ++i)
            if (P.get(i, j)) best.ptiles.push_back(i << 16 | j);
    *out = std::move(best);
    return true;
}

}  // namespace osfm
