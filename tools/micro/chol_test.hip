// Stand-alone check of the dense Cholesky solve (orthosfm_amd/csrc/ba_cholesky.hip) against a
// host solve on random SPD systems, with timing.   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
#include "../../orthosfm_amd/csrc/ba_cholesky.hip"
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
namespace osfm { void set_error(const char *, ...) {} }
using namespace osfm;
__global__ void dbg_factor(const double *A, int ld, double *Ldiag, int *info, double *Lout)
{
    __shared__ __attribute__((aligned(16))) double M[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) double LsT[NB][NB];
    const int lane = threadIdx.x, r = lane & 31;
    if (lane < NB) for (int c = 0; c < NB; ++c) M[r][c] = A[(size_t)r * ld + c];
    __syncthreads();
    factor_diag_block(&M[0][0], NB + 1, LsT, 0, Ldiag, info);
    __syncthreads();
    if (lane < NB) for (int c = 0; c < NB; ++c) Lout[r * NB + c] = LsT[c][r];
}
int main()
{
    for (int n : {20, 32, 45, 96, 250, 995}) {
        const int N = cholesky_padded_dim(n);
        std::mt19937_64 rng(n);
        std::normal_distribution<double> nd;
        std::vector<double> G((size_t)n * n), S((size_t)(N + 32) * N, 0.0), b(n), x(n);
        for (auto &v : G) v = nd(rng);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                double a = 0; for (int k = 0; k < n; ++k) a += G[(size_t)i * n + k] * G[(size_t)j * n + k];
                S[(size_t)i * N + j] = a / n + (i == j ? 1.0 : 0.0);
            }
        for (int i = n; i < N; ++i) S[(size_t)i * N + i] = 1.0;
        for (int i = 0; i < n; ++i) { b[i] = nd(rng); S[(size_t)N * N + i] = b[i]; }
        double *dS, *dL, *dD, *dx; int *dinfo;
        hipMalloc(&dS, S.size() * 8); hipMalloc(&dL, S.size() * 8); hipMalloc(&dD, (size_t)N * 32 * 8); hipMalloc(&dx, N * 8); hipMalloc(&dinfo, 16);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 16);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            launch_cholesky_solve(dS, dL, n, dD, dx, dinfo, nullptr, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        }
        hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost);
        int info; hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
        // residual |S x - b| / |b| with the full symmetric matrix
        double rn = 0, bn = 0;
        for (int i = 0; i < n; ++i) {
            double a = 0;
            for (int j = 0; j < n; ++j) a += (j <= i ? S[(size_t)i * N + j] : S[(size_t)j * N + i]) * x[j];
            rn += (a - b[i]) * (a - b[i]); bn += b[i] * b[i];
        }
        printf("n=%4d N=%4d info=%d  relative residual %.3e  %.3f ms\n", n, N, info, std::sqrt(rn / bn), best);
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
            {
                double *dLo; hipMalloc(&dLo, 32 * 32 * 8);
                hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(dbg_factor, dim3(1), dim3(64), 0, 0, dS, N, dD, dinfo, dLo);
                std::vector<double> Lg(32 * 32);
                hipMemcpy(Lg.data(), dLo, 32 * 32 * 8, hipMemcpyDeviceToHost);
                int shown = 0;
                for (int i = 0; i < 32 && shown < 12; ++i)
                    for (int c = 0; c <= i && shown < 12; ++c)
                        if (std::fabs(Lg[i * 32 + c] - Lh[i * 32 + c]) > 1e-12) { printf("  L[%d][%d] got %.6f want %.6f\n", i, c, Lg[i * 32 + c], Lh[i * 32 + c]); ++shown; }
            }
            hipMemcpy(got.data(), dD, 32 * 32 * 8, hipMemcpyDeviceToHost);
            for (int i = 0; i < 32; ++i) {
                double e = 0; int worst = -1;
                for (int c = 0; c < 32; ++c) { const double d_ = std::fabs(got[i * 32 + c] - inv[i * 32 + c]); if (d_ > e) { e = d_; worst = c; } }
                printf("  inv row %2d: max err %.2e at col %d (got %.5f want %.5f)\n", i, e, worst, worst >= 0 ? got[i * 32 + worst] : 0.0, worst >= 0 ? inv[i * 32 + worst] : 0.0);
            }
        }
        hipFree(dS); hipFree(dL); hipFree(dD); hipFree(dx); hipFree(dinfo);
    }
    return 0;
}
