// Stand-alone check of the dense Cholesky solve (orthosfm_amd/csrc/ba_cholesky.hip) against a
// host residual on random SPD systems, with timing: the one-launch flow form and the launch-per-column
// form on the same systems.   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
#include "../../orthosfm_amd/csrc/ba_cholesky.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
namespace osfm { void set_error(const char *, ...) {} }
using namespace osfm;
int main(int argc, char **argv)
{
    std::vector<int> sizes = {20, 32, 45, 96, 250, 300, 995, 2495, 2800, 4090};
    if (argc > 1) { sizes.clear(); for (int a = 1; a < argc; ++a) sizes.push_back(atoi(argv[a])); }
    int failures = 0;
    for (int n : sizes) {
        const int N = cholesky_padded_dim(n);
        std::mt19937_64 rng(n);
        std::normal_distribution<double> nd;
        // S = G G^T / m + I with m = 64 random columns (rank-64 + identity: SPD, O(n^2 m) to build)
        const int m = 64;
        std::vector<double> G((size_t)n * m), S((size_t)(N + 32) * N, 0.0), b(n), x(n);
        for (auto &v : G) v = nd(rng);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                double a = 0; for (int k = 0; k < m; ++k) a += G[(size_t)i * m + k] * G[(size_t)j * m + k];
                S[(size_t)i * N + j] = a / m + (i == j ? 1.0 : 0.0);
            }
        for (int i = n; i < N; ++i) S[(size_t)i * N + i] = 1.0;
        for (int i = 0; i < n; ++i) { b[i] = nd(rng); S[(size_t)N * N + i] = b[i]; }
        double *dS, *dL, *dD, *dx, *dbox; int *dinfo, *dflags;
        hipMalloc(&dS, S.size() * 8); hipMalloc(&dL, S.size() * 8); hipMalloc(&dD, (size_t)N * 32 * 8); hipMalloc(&dx, N * 8); hipMalloc(&dinfo, 16);
        hipMalloc(&dflags, (size_t)chol_flow_flag_count(n) * 4); hipMemset(dflags, 0, (size_t)chol_flow_flag_count(n) * 4);
        hipMalloc(&dbox, chol_flow_mailbox_bytes(n));
        int epoch = 0;
        std::vector<double> xs[2];
        for (int form = 0; form < 2; ++form) {        // 0: flow, 1: steps
            float best = 1e9f; int used = -1;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 16); hipMemset(dx, 0, N * 8);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                used = launch_cholesky_solve(dS, dL, n, dD, dx, dinfo, nullptr, 0, form == 0 ? dflags : nullptr, ++epoch, form == 0 ? dbox : nullptr);
                hipEventRecord(e1);
                if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); return 2; }
                float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
            }
            hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost);
            xs[form] = x;
            int info; hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
            // residual |S x - b| / |b| with the full symmetric matrix
            double rn = 0, bn = 0;
            for (int i = 0; i < n; ++i) {
                double a = 0;
                for (int j = 0; j < n; ++j) a += (j <= i ? S[(size_t)i * N + j] : S[(size_t)j * N + i]) * x[j];
                rn += (a - b[i]) * (a - b[i]); bn += b[i] * b[i];
            }
            const double res = std::sqrt(rn / bn);
            const bool ok = info == 0 && res < 1e-12;
            if (!ok) ++failures;
            printf("n=%4d nblk=%3d %-5s info=%d  relative residual %.3e  %.3f ms (%.2f us per block column)%s\n", n, N / 32,
                used ? "flow" : "steps", info, res, best, 1e3 * best / (N / 32), ok ? "" : "   <-- FAIL");
            fflush(stdout);
        }
        double dmax = 0, xmax = 0;
        for (int i = 0; i < n; ++i) { dmax = std::max(dmax, std::fabs(xs[0][i] - xs[1][i])); xmax = std::max(xmax, std::fabs(xs[1][i])); }
        printf("        flow vs steps: max |dx| / max |x| = %.2e\n", dmax / xmax);
        if (n == 32) {
            // block 0: compare inv(L) with a host factorisation, entry by entry
            std::vector<double> Lh((size_t)32 * 32, 0.0), inv((size_t)32 * 32, 0.0), got((size_t)32 * 32);
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j <= i; ++j) {
                    double a = S[(size_t)i * N + j];
                    for (int k = 0; k < j; ++k) a -= Lh[i * 32 + k] * Lh[j * 32 + k];
                    Lh[i * 32 + j] = i == j ? std::sqrt(a) : a / Lh[j * 32 + j];
                }
            for (int c = 0; c < 32; ++c)
                for (int i = 0; i < 32; ++i) {
                    double a = i == c ? 1.0 : 0.0;
                    for (int k = 0; k < i; ++k) a -= Lh[i * 32 + k] * inv[k * 32 + c];
                    inv[i * 32 + c] = a / Lh[i * 32 + i];
                }
            hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(chol_first_kernel, dim3(1), dim3(128), 0, 0, dS, N, dD, dinfo, nullptr);
            hipMemcpy(got.data(), dD, 32 * 32 * 8, hipMemcpyDeviceToHost);
            double worst = 0; int nz_upper = 0;
            for (int i = 0; i < 32; ++i)
                for (int c = 0; c < 32; ++c) {
                    worst = std::max(worst, std::fabs(got[i * 32 + c] - inv[i * 32 + c]));
                    if (c > i && got[i * 32 + c] != 0.0) ++nz_upper;
                }
            printf("        inv(L) of one block: max error %.2e, non-zeros above the diagonal %d%s\n", worst, nz_upper,
                worst < 1e-13 && nz_upper == 0 ? "" : "   <-- FAIL");
            if (!(worst < 1e-13) || nz_upper) ++failures;
        }
        hipFree(dS); hipFree(dL); hipFree(dD); hipFree(dx); hipFree(dinfo); hipFree(dflags); hipFree(dbox);
    }
    printf(failures ? "FAILED: %d\n" : "all ok\n", failures);
    return failures ? 1 : 0;
}
