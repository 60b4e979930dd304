// Stand-alone check of the one-launch Cholesky with a block pattern (ba_order.hip): ring-banded systems (C cameras
// of 5 unknowns, coupled within w positions on the ring) solved in the cameras' own order and in the ordered
// layout (arcs + separators, interior padding), against a host residual; every solve repeated and compared
// bit for bit (a wait that was skipped where it was needed shows up as run-to-run differences).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o chol_order_test.bin chol_order_test.hip
#ifdef BASE_KERNEL
#include "/tmp/base_r4/orthosfm_amd/csrc/ba_cholesky.hip"
#include <utility>
namespace osfm { struct FlowPattern { const unsigned long long *nz = nullptr; const int32_t *ptiles = nullptr; int num_ptiles = 0; };
struct ReducedOrder { bool active = false; int span = 0, nblk = 0, arcs = 0, sep_cams = 0, chain_natural = 0, chain_ordered = 0; std::vector<int32_t> cam_off, pad, ptiles; std::vector<unsigned long long> nz; };
static bool choose_reduced_order(int, const int32_t *, const std::vector<std::pair<int, int>> &, ReducedOrder *) { return false; } }
#else
#include "../../orthosfm_amd/csrc/ba_cholesky.hip"
#include "../../orthosfm_amd/csrc/ba_order.hip"
#endif
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
namespace osfm { void set_error(const char *, ...) {} }
using namespace osfm;

static int solve_reps(const std::vector<double> &S, int n, int N, FlowPattern pat, std::vector<double> *x_out, float *best_ms, bool *same)
{
    double *dS, *dL, *dD, *dx, *dbox; int *dinfo, *dflags;
    hipMalloc(&dS, S.size() * 8); hipMalloc(&dL, S.size() * 8); hipMalloc(&dD, (size_t)N * 32 * 8); hipMalloc(&dx, N * 8); hipMalloc(&dinfo, 16);
    hipMalloc(&dflags, (size_t)chol_flow_flag_count(n) * 4); hipMemset(dflags, 0, (size_t)chol_flow_flag_count(n) * 4);
    hipMalloc(&dbox, chol_flow_mailbox_bytes(n));
    hipMemset(dL, 0xff, S.size() * 8);        // NaNs: a tile read before it was written shows
    int epoch = 0, used = -1, info = 0;
    *best_ms = 1e9f; *same = true;
    std::vector<double> x(n), first;
    for (int rep = 0; rep < 12; ++rep) {
        hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 16); hipMemset(dx, 0, N * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        #ifdef BASE_KERNEL
        used = launch_cholesky_solve(dS, dL, n, dD, dx, dinfo, nullptr, 0, dflags, ++epoch, dbox);
#else
        used = launch_cholesky_solve(dS, dL, n, dD, dx, dinfo, nullptr, 0, dflags, ++epoch, dbox, pat);
#endif
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); exit(2); }
        float ms; hipEventElapsedTime(&ms, e0, e1); *best_ms = std::min(*best_ms, ms);
        hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
        if (rep == 0) first = x; else if (memcmp(first.data(), x.data(), n * 8) != 0) *same = false;
    }
    *x_out = x;
    hipFree(dS); hipFree(dL); hipFree(dD); hipFree(dx); hipFree(dinfo); hipFree(dflags); hipFree(dbox);
    return used ? info : -1;
}

int main(int argc, char **argv)
{
    struct Case { int C, w; } cases[] = {{200, 11}, {120, 11}, {500, 11}, {64, 5}, {300, 20}};
    int failures = 0;
    for (const Case &cs : cases) {
        const int C = cs.C, w = cs.w;
        std::vector<int32_t> ldim(C, 5);
        ldim[0] = 0;
        std::vector<std::pair<int, int>> pairs;
        for (int a = 0; a < C; ++a)
            for (int b = 0; b <= a; ++b) { const int d = a - b; if (std::min(d, C - d) <= w) pairs.push_back({a, b}); }
        ReducedOrder ord;
        const bool act = choose_reduced_order(C, ldim.data(), pairs, &ord) && getenv("NATURAL_ONLY") == nullptr;
        std::vector<int32_t> nat(C, 0);
        int nc = 0;
        for (int c = 0; c < C; ++c) { nat[c] = nc; nc += ldim[c]; }
        printf("C=%d w=%d: unknowns %d, order %s: %d arcs, span %d (%d blocks), chain %d -> %d, %zu P tiles\n", C, w, nc, act ? "on" : "off", ord.arcs, ord.span,
            ord.nblk, ord.chain_natural, ord.chain_ordered, ord.ptiles.size());
        // camera-block matrix: M = sum over "tracks" (runs of w + 1 consecutive cameras) of g g^T + I, in camera coordinates
        std::mt19937_64 rng(C * 131 + w);
        std::normal_distribution<double> nd;
        std::vector<double> dense((size_t)nc * nc, 0.0), b(nc);
        for (int t = 0; t < 3 * C; ++t) {
            const int start = (int)(rng() % C), len = 2 + (int)(rng() % w);
            std::vector<int> idx;
            for (int q = 0; q < len; ++q) { const int c = (start + q) % C; for (int e = 0; e < ldim[c]; ++e) idx.push_back(nat[c] + e); }
            std::vector<double> g(idx.size());
            for (auto &v : g) v = nd(rng);
            for (size_t x = 0; x < idx.size(); ++x) for (size_t y = 0; y < idx.size(); ++y) dense[(size_t)idx[x] * nc + idx[y]] += g[x] * g[y];
        }
        for (int i = 0; i < nc; ++i) { dense[(size_t)i * nc + i] += 1.0; b[i] = nd(rng); }
        std::vector<double> xs[2];
        for (int form = 0; form < (act ? 2 : 1); ++form) {
            const std::vector<int32_t> &off = form ? ord.cam_off : nat;
            const int n = form ? ord.span : nc, N = cholesky_padded_dim(n);
            std::vector<int> pos(nc);           // natural unknown -> laid-out unknown
            for (int c = 0; c < C; ++c) for (int e = 0; e < ldim[c]; ++e) pos[nat[c] + e] = off[c] + e;
            std::vector<double> S((size_t)(N + 32) * N, 0.0);
            for (int i = 0; i < N; ++i) S[(size_t)i * N + i] = 1.0;            // padding (interior and tail): identity
            for (int i = 0; i < nc; ++i)
                for (int j = 0; j < nc; ++j) {
                    const int pi = pos[i], pj = pos[j];
                    if (pi >= pj) S[(size_t)pi * N + pj] = dense[(size_t)i * nc + j];
                }
            for (int i = 0; i < nc; ++i) S[(size_t)N * N + pos[i]] = b[i];

            unsigned long long *dnz = nullptr; int32_t *dpt = nullptr;
            FlowPattern pat;
            if (form) {
                hipMalloc(&dnz, ord.nz.size() * 8); hipMemcpy(dnz, ord.nz.data(), ord.nz.size() * 8, hipMemcpyHostToDevice);
                hipMalloc(&dpt, std::max<size_t>(ord.ptiles.size(), 1) * 4); hipMemcpy(dpt, ord.ptiles.data(), ord.ptiles.size() * 4, hipMemcpyHostToDevice);
                pat.nz = dnz; pat.ptiles = dpt; pat.num_ptiles = (int)ord.ptiles.size();
            }
            std::vector<double> xl; float ms; bool same;
            const int info = solve_reps(S, n, N, pat, &xl, &ms, &same);
            std::vector<double> x(nc);
            for (int i = 0; i < nc; ++i) x[i] = xl[pos[i]];
            double rn = 0, bn = 0;
            for (int i = 0; i < nc; ++i) { double a = 0; for (int j = 0; j < nc; ++j) a += dense[(size_t)i * nc + j] * x[j]; rn += (a - b[i]) * (a - b[i]); bn += b[i] * b[i]; }
            const double res = std::sqrt(rn / bn);
            const bool ok = info == 0 && res < 1e-11 && same;
            if (!ok) ++failures;
            printf("   %-8s info=%d residual %.2e  %.3f ms  12 solves bit-identical: %s%s\n", form ? "ordered" : "natural", info, res, ms, same ? "yes" : "NO", ok ? "" : "   <-- FAIL");
            fflush(stdout);
            xs[form] = x;
            if (dnz) hipFree(dnz); if (dpt) hipFree(dpt);
        }
        if (act) { double dm = 0; for (int i = 0; i < nc; ++i) dm = std::max(dm, std::fabs(xs[0][i] - xs[1][i])); printf("   natural vs ordered: max |dx| = %.2e\n", dm); }
    }
    printf(failures ? "FAILED: %d\n" : "all ok\n", failures);
    return failures ? 1 : 0;
}
