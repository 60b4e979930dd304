// gfx950 kernels of the bundle-adjustment path (hot path B).
//
// One Levenberg-Marquardt iteration of the problem runBundleAdjustment builds
// (src/bundle_adjustment/bundle_adjustment.cpp:61-145) with the Jacobian never
// written to memory: every consumer re-derives the 2x(6+3) block of an
// observation from (camera, point, pixel) in registers.
//
//   point_pass   thread per track   V_j = sum Jp^T Jp (+D^2) -> V_j^-1, g_j, cost
//   pair_pass    wave per camera pair (c1 >= c2) that shares a track:
//                S[c1][c2] = [c1==c2](sum Jc^T Jc + D^2) - sum Z_a W_b^T,
//                rhs_c1    = sum Jc^T r - Z_a g_j          (Schur complement)
//   (dense Cholesky of S: ba_api.hip)
//   back_pass    thread per track   y_p = V^-1 (g - sum W^T y_c), candidate
//                point = Plus(P, -scale*y_p), model cost change, step norms
//   cost_pass    thread per track   cost at the candidate
// All sums that cross threads use fixed-order two-level reductions (no
// floating-point atomics): the solve is bit-reproducible run to run.
#include "ba_kernels.h"
#include <algorithm>
#include <cstdlib>

namespace osfm {

// ---- deterministic block reductions ---------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
    return v;
}

// ---- the 54 sums of a pair-pass wave ---------------------------------------------
// Every lane holds a partial of each of the 54 quantities; lane v ends up with the wave's total of
// quantity v.  Recursive halving: at the level with lane mask m a lane keeps the half of its live
// quantities whose index has bit m like its own lane number, hands the other half to lane ^ m and adds
// what it receives -- 32 + 16 + 8 + 4 + 2 + 1 exchanges in all, where folding every quantity over the
// whole wave on its own (six xor-shuffles each) was 324 additions and 648 ds_bpermute instructions per
// wave: the LDS crossbar, not the gathers, set the pace of the pair pass.  The two upper levels are
// v_permlane32_swap / v_permlane16_swap (register halves / odd and even rows trade places: no select, no
// LDS); the order of the additions is fixed, so repeats stay bit-identical.
__device__ __forceinline__ double swap_add(double a, double b, bool level32)
{
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    double x0, x1;
    if (level32) {
        // a' = { a[0..31], b[0..31] }, b' = { a[32..63], b[32..63] }
        const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        x0 = __hiloint2double((int)hi[0], (int)lo[0]); x1 = __hiloint2double((int)hi[1], (int)lo[1]);
    } else {
        // rows of 16 lanes: a' = { a.r0, b.r0, a.r2, b.r2 }, b' = { a.r1, b.r1, a.r3, b.r3 }
        const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        x0 = __hiloint2double((int)hi[0], (int)lo[0]); x1 = __hiloint2double((int)hi[1], (int)lo[1]);
    }
    return x0 + x1;
}

// s[0..53] -> the total of quantity `lane` (0 for lanes 54..63)
__device__ __forceinline__ double wave_reduce_54(const double (&s)[54], int lane)
{
    double w[32], x[16], y[8], z[4], q[2];
#pragma unroll
    for (int v = 0; v < 32; ++v) w[v] = swap_add(s[v], v + 32 < 54 ? s[v + 32] : 0.0, true);
#pragma unroll
    for (int v = 0; v < 16; ++v) x[v] = swap_add(w[v], w[v + 16], false);
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
#pragma unroll
    for (int v = 0; v < 8; ++v) y[v] = (b3 ? x[v + 8] : x[v]) + __shfl_xor(b3 ? x[v] : x[v + 8], 8);
#pragma unroll
    for (int v = 0; v < 4; ++v) z[v] = (b2 ? y[v + 4] : y[v]) + __shfl_xor(b2 ? y[v] : y[v + 4], 4);
#pragma unroll
    for (int v = 0; v < 2; ++v) q[v] = (b1 ? z[v + 2] : z[v]) + __shfl_xor(b1 ? z[v] : z[v + 2], 2);
    return (b0 ? q[1] : q[0]) + __shfl_xor(b0 ? q[0] : q[1], 1);
}

// per-block partial -> partials[slot * count + index] (index: the workgroup's window)
template <bool IS_MAX, bool SC1 = false>
__device__ __forceinline__ void block_partial(double v, double *partials, int slot, double *sh, int index, int count)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = IS_MAX ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = IS_MAX ? fmax(t, sh[w]) : t + sh[w];
        if (SC1) store_sc1(&partials[(size_t)slot * count + index], t);
        else partials[(size_t)slot * count + index] = t;
    }
}

// The workgroup that takes the last ticket of a launch (every workgroup calls this once, behind its last
// write-through store): MI355X_MICROARCH.md, "Valid forms" -- every storing wave drains its stores, one lane adds
// to the counter behind the workgroup's barrier, the workgroup whose add came last loads (sc1) behind a barrier
// that lane joins.  The counter is sharded: one word of all workgroups was 1.5k - 6k device-scope adds to ONE
// address from eight XCDs, 25 - 30 us of a launch; here a workgroup adds to the shard blockIdx % kTicketShards (a
// cache line each), the last of a shard to the top word, the last of those is the last of the launch.  All words
// are left at zero for the next launch.
constexpr int kTicketShards = 64, kTicketStride = 32;        // ints: a shard per 128-byte line, the top word behind them
__device__ __forceinline__ bool last_workgroup(int32_t *ticket, int *lds_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int shard = (int)(blockIdx.x % kTicketShards);
        const int members = ((int)gridDim.x - shard + kTicketShards - 1) / kTicketShards;
        const int live = min((int)gridDim.x, kTicketShards);
        int32_t *word = ticket + shard * kTicketStride, *top = ticket + kTicketShards * kTicketStride;
        int last = 0;
        if (__hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            __hip_atomic_store(word, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(top, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == live - 1) {
                __hip_atomic_store(top, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        *lds_flag = last;
    }
    __syncthreads();
    return *lds_flag != 0;
}
size_t lm_ticket_bytes() { return (size_t)(kTicketShards + 1) * kTicketStride * 4; }

// ---------------------------------------------------------------------------
// LM control on the device.  Fixed-order reductions of the per-block partials, then the
// decisions of TrustRegionMinimizer / LevenbergMarquardtStrategy by thread 0.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double block_reduce_256(double v, bool is_max, double *sh)
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st)
            sh[threadIdx.x] = is_max ? fmax(sh[threadIdx.x], sh[threadIdx.x + st]) : sh[threadIdx.x] + sh[threadIdx.x + st];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// SC1: the values were stored write-through by other workgroups of THIS launch (sc1 loads, see last_workgroup)
template <bool SC1 = false>
__device__ __forceinline__ double strided_reduce(const double *p, int n, int stride, bool is_max, double *sh)
{
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double x = SC1 ? load_sc1(p + (size_t)i * stride) : p[(size_t)i * stride];
        v = is_max ? fmax(v, x) : v + x;
    }
    return block_reduce_256(v, is_max, sh);
}

// After the candidate of an iteration has been evaluated: step validity, the parameter /
// function tolerance tests, accept or reject, the new radius, and what the linearisation
// that follows has to do.
// state -> the host's slot for this iteration (page-locked memory the device writes directly:
// a copy engine transfer per iteration cost 4 us plus a 6 us bubble before the next kernel)
__device__ __forceinline__ void publish_state(const LmDev *lm, LmDev *host_out)
{
    if (host_out) *host_out = *lm;
}

__device__ __forceinline__ void
lm_decide_logic(LmDev *lm, const LmParams &prm, const LmScratch &sc, bool solved, double mcc, double sn, double xn,
    double cand)
{
    const int info = *sc.chol_info;
    *sc.chol_info = 0;
    const double step_norm = sqrt(sn), x_norm = sqrt(xn);
    lm->model_cost_change = mcc; lm->step_norm = step_norm; lm->x_norm = x_norm; lm->cand_cost = cand;
    const bool solve_ok = solved && info == 0 && isfinite(mcc) && isfinite(step_norm);
    const bool step_valid = solve_ok && mcc > 0.0;
    if (!step_valid) {
        // HandleInvalidStep -> LevenbergMarquardtStrategy::StepIsInvalid
        if (++lm->invalid_steps >= prm.max_invalid_steps) { lm->term = OSFM_BA_FAILURE; lm->stop = 1; return; }
        lm->radius = lm->radius / lm->decrease_factor; lm->decrease_factor *= 2.0;
        lm->num_unsuccess++;
        lm->update_diag = 0; lm->want_gradient = 0;
        return;
    }
    lm->invalid_steps = 0;
    if (!isfinite(cand)) cand = 1.7976931348623157e308;
    // ParameterToleranceReached / FunctionToleranceReached
    if (step_norm <= prm.parameter_tolerance * (x_norm + prm.parameter_tolerance)) { lm->term = OSFM_BA_CONVERGENCE_PARAMETER; lm->stop = 1; return; }
    const double cost_change = lm->x_cost - cand;
    if (fabs(cost_change) <= prm.function_tolerance * lm->x_cost) { lm->term = OSFM_BA_CONVERGENCE_FUNCTION; lm->stop = 1; return; }
    const double relative_decrease = cost_change / mcc;
    if (relative_decrease > prm.min_relative_decrease) {
        // HandleSuccessfulStep + LevenbergMarquardtStrategy::StepAccepted
        lm->cur ^= 1;
        const double t = 2.0 * relative_decrease - 1.0;
        lm->radius = fmin(prm.max_radius, lm->radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
        lm->decrease_factor = 2.0;
        lm->num_success++;
        lm->last_successful = 1;
        lm->update_diag = 1; lm->want_gradient = 1;
    } else {
        // LevenbergMarquardtStrategy::StepRejected
        lm->radius = lm->radius / lm->decrease_factor; lm->decrease_factor *= 2.0;
        lm->num_unsuccess++;
        lm->update_diag = 0; lm->want_gradient = 0;
    }
}

// One workgroup of 256 threads (a kernel of its own, or the last workgroup of the back pass: SC1)
template <bool SC1>
__device__ __forceinline__ void
lm_decide_block(LmDev *lm, const LmParams &prm, const LmScratch &sc, LmDev *host_out, double *sh)
{
    if (lm->stop || lm->flow_aborted) { if (threadIdx.x == 0) publish_state(lm, host_out); return; }
    if (*sc.chol_info >= kFlowAborted) {
        // not a numerical failure: the factorisation's launch was given up (ba_cholesky.hip).  Nothing is decided;
        // the host repeats the iteration in the launch-per-column form (osfm_ba_solve)
        if (threadIdx.x == 0) { *sc.chol_info = 0; lm->flow_aborted = 1; publish_state(lm, host_out); }
        return;
    }
    const bool solved = !lm->lin_failed;
    double mcc = 0.0, sn = 0.0, xn = 0.0, cand = 0.0;
    if (solved) {
        mcc = strided_reduce<SC1>(sc.partB, sc.blocksM, 1, false, sh);
        sn = strided_reduce<SC1>(sc.partB + sc.blocksM, sc.blocksM, 1, false, sh) + strided_reduce<SC1>(sc.part_cam, sc.C, 2, false, sh);
        xn = strided_reduce<SC1>(sc.partB + 2 * (size_t)sc.blocksM, sc.blocksM, 1, false, sh) + strided_reduce<SC1>(sc.part_cam + 1, sc.C, 2, false, sh);
        cand = strided_reduce<SC1>(sc.partC, sc.blocksM, 1, false, sh);
    }
    // a one-block system was consumed by the solve at the head of this iteration: clear it
    // for the linearisation that follows (saves that launch its own reset kernel)
    if (sc.reset_S) {
        const int N = sc.reset_N, n = sc.reset_n;
        for (int i = threadIdx.x; i < (N + 32) * N; i += blockDim.x) {
            const int row = i / N, col = i - row * N;
            sc.reset_S[i] = (row == col && row >= n && row < N) ? 1.0 : 0.0;
        }
    }
    if (threadIdx.x != 0) return;
    lm_decide_logic(lm, prm, sc, solved, mcc, sn, xn, cand);
    publish_state(lm, host_out);
}

__global__ __launch_bounds__(256) void
ba_lm_decide_kernel(LmDev *lm, LmParams prm, LmScratch sc, LmDev *host_out)
{
    __shared__ double sh[256];
    lm_decide_block<false>(lm, prm, sc, host_out, sh);
}

// After a linearisation: cost / gradient norm of a new iterate, the not-positive-definite
// flag, then FinalizeIterationAndCheckIfMinimizerCanContinue for the iteration that follows.
__device__ __forceinline__ void
lm_post_logic(LmDev *lm, const LmParams &prm, int initial, double cost, double gp, double bad, double gc)
{
    if (lm->want_gradient) { lm->x_cost = cost; lm->grad_max = fmax(gp, gc); }
    lm->lin_failed = bad != 0.0 ? 1 : 0;
    if (initial) {
        lm->initial_cost = cost;
        if (!isfinite(cost)) { lm->nonfinite = 1; lm->term = OSFM_BA_FAILURE; lm->stop = 1; return; }
        if (lm->grad_max <= prm.gradient_tolerance) { lm->term = OSFM_BA_CONVERGENCE_GRADIENT; lm->stop = 1; return; }
    }
    // the LM diagonal is refreshed by the linearisation that follows an accepted step only
    lm->update_diag = 0; lm->want_gradient = 0;
    if (lm->iteration >= prm.max_iterations) { lm->term = OSFM_BA_NO_CONVERGENCE; lm->stop = 1; return; }
    if (lm->last_successful && lm->grad_max <= prm.gradient_tolerance) { lm->term = OSFM_BA_CONVERGENCE_GRADIENT; lm->stop = 1; return; }
    if (lm->radius <= prm.min_radius) { lm->term = OSFM_BA_CONVERGENCE_TRUST_REGION; lm->stop = 1; return; }
    lm->iteration++;
    lm->last_successful = 0;
}

// One workgroup of 256 threads (a kernel of its own, or the last workgroup of the pair pass: GMAX_SC1 -- the
// camera gradient norms come from that launch, the point pass's partials from the one before it)
template <bool GMAX_SC1>
__device__ __forceinline__ void
lm_post_block(LmDev *lm, const LmParams &prm, const LmScratch &sc, int initial, LmDev *host_out, double *sh)
{
    if (lm->stop || lm->flow_aborted) { if (threadIdx.x == 0) publish_state(lm, host_out); return; }
    const double cost = strided_reduce(sc.partA, sc.blocksM, 1, false, sh);
    const double gp = strided_reduce(sc.partA + sc.blocksM, sc.blocksM, 1, true, sh);
    const double bad = strided_reduce(sc.partA + 2 * (size_t)sc.blocksM, sc.blocksM, 1, true, sh);
    const double gc = strided_reduce<GMAX_SC1>(sc.gmax_cam, sc.C, 1, true, sh);
    if (threadIdx.x != 0) return;
    lm_post_logic(lm, prm, initial, cost, gp, bad, gc);
    publish_state(lm, host_out);
}

__global__ __launch_bounds__(256) void
ba_lm_post_kernel(LmDev *lm, LmParams prm, LmScratch sc, int initial, LmDev *host_out)
{
    __shared__ double sh[256];
    lm_post_block<false>(lm, prm, sc, initial, host_out, sh);
}

void launch_lm_decide(LmDev *lm, const LmParams &prm, const LmScratch &sc, LmDev *host_out, hipStream_t s)
{
    hipLaunchKernelGGL(ba_lm_decide_kernel, dim3(1), dim3(256), 0, s, lm, prm, sc, host_out);
}

__global__ void ba_lm_clear_abort_kernel(LmDev *lm) { lm->flow_aborted = 0; }
void launch_lm_clear_abort(LmDev *lm, hipStream_t s) { hipLaunchKernelGGL(ba_lm_clear_abort_kernel, dim3(1), dim3(1), 0, s, lm); }

void launch_lm_post(LmDev *lm, const LmParams &prm, const LmScratch &sc, int initial, LmDev *host_out, hipStream_t s)
{
    hipLaunchKernelGGL(ba_lm_post_kernel, dim3(1), dim3(256), 0, s, lm, prm, sc, initial, host_out);
}

// ---------------------------------------------------------------------------
// Observation windows.  The per-point passes (point, back, cost) give every OBSERVATION a lane: a workgroup takes
// the tracks whose first observation falls into a window of kWinObs consecutive observations (observations are
// grouped by track), which are at most 256 observations while no track runs more than 256 - kWinObs past the
// window's end.  One round per wave -- where four lanes per track walked the track's observations three at a
// time, each step a chain of dependent loads, with 62 % of the lanes busy (tracks of 3..12) -- and a track's sums
// are added in observation order from LDS by every lane of the track, so they do not depend on the packing.
// Windows that hold more (a long track at their end) are listed once per solve and taken by the *_over kernels:
// the four-lanes-per-track form, any length.  The window layout is the same in every iteration of a solve.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void
ba_windows_kernel(BaDev d, int num, WinDesc *desc, int32_t *over_list, int32_t *ok_list, int32_t *counts)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= num) return;
    const int64_t k0 = (int64_t)b * kWinObs, k1 = k0 + kWinObs;
    WinDesc w;
    // the first track that starts at or behind observation k0: the one behind the track of observation k0 - 1
    w.jf = k0 == 0 ? 0 : d.obs_pt[k0 - 1] + 1;
    w.jn = k1 >= d.O ? d.M : d.obs_pt[k1 - 1] + 1;        // the last window also takes the empty tracks at the end
    w.ka = d.pt_start[w.jf];
    w.kb = d.pt_start[w.jn];
    desc[b] = w;
    // (the order inside the two lists is whatever the atomics make it: a window's partials have the window's slot)
    if (w.kb - w.ka > 256) over_list[atomicAdd(counts, 1)] = b;
    else ok_list[atomicAdd(counts + 1, 1)] = b;
}

// obs_lay[k] = cam_off | cam_ldim << 24 of the observation's camera: one coalesced load instead of two gathers
// behind the camera index
__global__ __launch_bounds__(256) void
ba_obs_lay_kernel(BaDev d, int32_t *obs_lay)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= d.O) return;
    const int c = d.obs_cam[k];
    obs_lay[k] = d.cam_off[c] | (d.cam_ldim[c] << 24);
}

int obs_windows_count(int O) { return std::max(1, (O + kWinObs - 1) / kWinObs); }

void launch_obs_windows(const BaDev &d, int num, WinDesc *desc, int32_t *over_list, int32_t *ok_list, int32_t *counts, int32_t *obs_lay,
    hipStream_t s)
{
    hipLaunchKernelGGL(ba_windows_kernel, dim3((num + 255) / 256), dim3(256), 0, s, d, num, desc, over_list, ok_list, counts);
    if (d.O > 0) hipLaunchKernelGGL(ba_obs_lay_kernel, dim3((d.O + 255) / 256), dim3(256), 0, s, d, obs_lay);
}

// ---------------------------------------------------------------------------
// point pass
// ---------------------------------------------------------------------------
// what a track's lanes do once its sums are complete (V lower triangle, g): the LM diagonal, (V + D^2 / radius)^-1,
// the stores of the track's first lane (`leader`), the point's share of the gradient norm.  Returns false when V is
// not positive definite.
// |Plus(x, -g) - x|_inf of a point with the UNSCALED gradient g / scale
__device__ __forceinline__ double point_gradient_norm(const BaDev &d, int j, const double (&g)[3])
{
    double dl[3], out[4], m = 0.0;
    for (int x = 0; x < 3; ++x) dl[x] = -g[x] / d.scale_p[3 * j + x];
    homog_plus(d.points + 4 * j, dl, out);
    for (int x = 0; x < 4; ++x) m = fmax(m, fabs(d.points[4 * j + x] - out[x]));
    return m;
}

__device__ __forceinline__ bool
point_track_finish(const BaDev &d, const PointPassArgs &a, int j, bool leader, bool empty, double (&V)[3][3],
    const double (&g)[3], double (&Vi)[3][3], double &gmax, bool with_gmax = true)
{
    bool ok = true;
    V[0][1] = V[1][0]; V[0][2] = V[2][0]; V[1][2] = V[2][1];
    if (a.mode == kPassScaleInit) {
        // Jacobi scaling, computed once from the unscaled column norms
        // (TrustRegionMinimizer::EvaluateGradientAndJacobian, iteration 0)
        if (leader)
            for (int x = 0; x < 3; ++x) a.scale_p_out[3 * j + x] = 1.0 / (1.0 + sqrt(V[x][x]));
        return true;
    }
    for (int x = 0; x < 3; ++x) {
        // every lane derives the diagonal itself (the leader's store is not read back)
        const double dp = a.update_diag ? fmin(fmax(V[x][x], a.min_diag), a.max_diag) : a.diag_p[3 * j + x];
        if (a.update_diag && leader) a.diag_p[3 * j + x] = dp;
        V[x][x] += dp / a.radius;
    }
    if (!empty) {
        ok = inv3_spd(V, Vi);
    } else {
        for (int x = 0; x < 3; ++x)
            for (int y = 0; y < 3; ++y) Vi[x][y] = x == y ? 1.0 / V[x][x] : 0.0;
    }
    if (leader)
        for (int x = 0; x < 3; ++x) {
            a.ge[3 * j + x] = g[x];
            for (int y = 0; y < 3; ++y) a.vinv[9 * j + 3 * x + y] = ok ? Vi[x][y] : 0.0;
        }
    if (a.want_gradient && leader && with_gmax) gmax = fmax(gmax, point_gradient_norm(d, j, g));
    return ok;
}

// the record of one observation (16-byte aligned: 26 doubles) in 16-byte stores
__device__ __forceinline__ void
store_record(double *obsrec, int k, const ObsLin &o, const double (&qv)[6])
{
    double w[kObsRec];
#pragma unroll
    for (int x = 0; x < 6; ++x) { w[kRecJc + x] = o.Jc[0][x]; w[kRecJc + 6 + x] = o.Jc[1][x]; w[kRecQ + x] = qv[x]; }
#pragma unroll
    for (int x = 0; x < 3; ++x) { w[kRecJp + x] = o.Jp[0][x]; w[kRecJp + 3 + x] = o.Jp[1][x]; }
    w[kRecR] = o.r[0]; w[kRecR + 1] = o.r[1];
    double2 *dst = reinterpret_cast<double2 *>(obsrec + (size_t)k * kObsRec);
#pragma unroll
    for (int i = 0; i < kObsRec / 2; ++i) dst[i] = make_double2(w[2 * i], w[2 * i + 1]);
}

// The records of a wave's 64 observations are 13 KB of consecutive memory, but a lane storing its own record makes every
// store instruction touch 64 different cache lines (13 of them: 32 of the point pass's 83 us; the same bytes stored
// coalesced cost 10).  So the wave turns them through LDS in two rounds -- pieces 0 .. 6, then 7 .. 12 of every record,
// a lane writing its own and reading piece (64 i + lane) of the round's stream: consecutive lanes, consecutive 16 bytes,
// a handful of lines per instruction.  stage: kRecStage double2 of LDS of the wave's own (one wave's LDS operations
// execute in order: no barrier); nrec: records of the wave that exist (the others' lanes store nothing).
constexpr int kRecStage = 64 * 7;
__device__ __forceinline__ void
store_records_staged(double *obsrec, int k_wave, int nrec, int lane, const ObsLin &o, const double (&qv)[6], double2 *stage)
{
    double w[kObsRec];
#pragma unroll
    for (int x = 0; x < 6; ++x) { w[kRecJc + x] = o.Jc[0][x]; w[kRecJc + 6 + x] = o.Jc[1][x]; w[kRecQ + x] = qv[x]; }
#pragma unroll
    for (int x = 0; x < 3; ++x) { w[kRecJp + x] = o.Jp[0][x]; w[kRecJp + 3 + x] = o.Jp[1][x]; }
    w[kRecR] = o.r[0]; w[kRecR + 1] = o.r[1];
    double2 *dst = reinterpret_cast<double2 *>(obsrec + (size_t)k_wave * kObsRec);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int np = round == 0 ? 7 : 6, p0 = round == 0 ? 0 : 7;           // pieces of this round
#pragma unroll
        for (int i = 0; i < 7; ++i)
            if (i < np) stage[lane * np + i] = make_double2(w[2 * (p0 + i)], w[2 * (p0 + i) + 1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            if (i >= np) continue;
            const int q = i * 64 + lane, rec = q / np, pc = q - rec * np;
            const double2 v = stage[q];
            if (rec < nrec) dst[rec * (kObsRec / 2) + p0 + pc] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// the tracks of a window without observations (point or back pass: lane per track)
template <typename F>
__device__ __forceinline__ void for_empty_tracks(const BaDev &d, const WinDesc &wd, F f)
{
    for (int j = wd.jf + (int)threadIdx.x; j < wd.jn; j += 256)
        if (d.pt_start[j + 1] == d.pt_start[j]) f(j);
}

__global__ __launch_bounds__(256) void
ba_point_win_kernel(BaDev d, PointPassArgs a, ObsWindows w)
{
    // what an observation adds to its track's sums -- Jp^T Jp (lower triangle: 6) and Jp^T r (3) -- piece i of
    // lane t at [i][t]: the lanes' 16-byte accesses are consecutive
    // (the same LDS serves the staged store of the records afterwards: 4 waves x kRecStage pieces)
    __shared__ double2 lds_buf[4 * kRecStage];
    double2 (*obs_lds)[256] = reinterpret_cast<double2 (*)[256]>(lds_buf);
    static_assert(4 * kRecStage >= 5 * 256, "the products of 256 observations fit the staging space");
    __shared__ double sh[4];
    if (!lm_resolve(d)) return;
    if (d.lm && a.mode == kPassNormal) {
        // the LM state decides what this linearisation is for
        a.radius = d.lm->radius; a.update_diag = d.lm->update_diag; a.want_gradient = d.lm->want_gradient;
    }
    const int b = w.ok_list[blockIdx.x], tid = threadIdx.x;          // (the other windows: ba_point_over_kernel)
    const WinDesc wd = w.desc[b];
    const int k = wd.ka + tid;
    const bool live = k < wd.kb;
    double cost = 0.0, gmax = 0.0;
    int bad = 0;
    int j = 0, ks = 0, ke = 0;
    ObsLin o;
    if (live) {
        j = d.obs_pt[k];
        ks = d.pt_start[j]; ke = d.pt_start[j + 1];
        PointDer pd;
        point_der(d, j, d.points + 4 * j, true, pd);
        linearize_obs(d, k, d.obs_cam[k], w.obs_lay[k], d.camder, pd, o);
        cost = 0.5 * o.rho0;
        double pr[10];
        int q = 0;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
#pragma unroll
            for (int y = 0; y <= x; ++y) pr[q++] = o.Jp[0][x] * o.Jp[0][y] + o.Jp[1][x] * o.Jp[1][y];
        }
#pragma unroll
        for (int x = 0; x < 3; ++x) pr[6 + x] = o.Jp[0][x] * o.r[0] + o.Jp[1][x] * o.r[1];
        pr[9] = 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) obs_lds[i][tid] = make_double2(pr[2 * i], pr[2 * i + 1]);
    }
    __syncthreads();
    double qv[6] = { 0, 0, 0, 0, 0, 0 };
    if (live) {
        if (d.pdim) {
            // the track's sums, in observation order, in every lane of the track
            double sm[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            for (int i = ks - wd.ka; i < ke - wd.ka; ++i) {
#pragma unroll
                for (int u = 0; u < 5; ++u) { const double2 v = obs_lds[u][i]; sm[2 * u] += v.x; sm[2 * u + 1] += v.y; }
            }
            double V[3][3], Vi[3][3];
            V[0][0] = sm[0]; V[1][0] = sm[1]; V[1][1] = sm[2]; V[2][0] = sm[3]; V[2][1] = sm[4]; V[2][2] = sm[5];
            const double g[3] = { sm[6], sm[7], sm[8] };
            // (the points' gradient norm: by a lane per track, below)
            if (!point_track_finish(d, a, j, k == ks, false, V, g, Vi, gmax, false)) bad = 1;
            if (a.mode != kPassScaleInit) {
                // Q = Jp V^-1
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        qv[3 * rr + t] = bad ? 0.0 : o.Jp[rr][0] * Vi[0][t] + o.Jp[rr][1] * Vi[1][t] + o.Jp[rr][2] * Vi[2][t];
            }
        }
    }
    // every wave has its track sums: the LDS is free for the records (and the gradients below are in memory)
    __syncthreads();
    {
        const int wave = tid >> 6, k_wave = wd.ka + wave * 64;
        if (k_wave < wd.kb) store_records_staged(a.obsrec, k_wave, min(64, wd.kb - k_wave), tid & 63, o, qv, lds_buf + wave * kRecStage);
    }
    if (d.pdim) {
        // a lane per TRACK: the tracks without observations, and every point's share of the gradient norm (sine,
        // cosine and two roots that all four waves went through for their few first lanes) from the gradient its
        // first lane has stored
        const bool with_gmax = a.mode != kPassScaleInit && a.want_gradient;
        for (int je = wd.jf + tid; je < wd.jn; je += 256) {
            if (d.pt_start[je + 1] == d.pt_start[je]) {
                double V[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } }, Vi[3][3];
                const double g[3] = { 0, 0, 0 };
                point_track_finish(d, a, je, true, true, V, g, Vi, gmax, false);
            } else if (with_gmax) {
                const double g[3] = { a.ge[3 * je], a.ge[3 * je + 1], a.ge[3 * je + 2] };
                gmax = fmax(gmax, point_gradient_norm(d, je, g));
            }
        }
    }
    block_partial<false>(cost, a.partials, 0, sh, b, w.num);
    block_partial<true>(gmax, a.partials, 1, sh, b, w.num);
    block_partial<true>((double)bad, a.partials, 2, sh, b, w.num);
}

// the windows of more than 256 observations (a long track at their end, or nothing but long tracks): a WAVE per
// track, its lanes striding over the track's observations, the track's sums by a fixed-order butterfly
__global__ __launch_bounds__(256) void
ba_point_over_kernel(BaDev d, PointPassArgs a, ObsWindows w)
{
    __shared__ double sh[4];
    if (!lm_resolve(d)) return;
    if (d.lm && a.mode == kPassNormal) {
        a.radius = d.lm->radius; a.update_diag = d.lm->update_diag; a.want_gradient = d.lm->want_gradient;
    }
    const int b = w.over_list[blockIdx.x];
    const WinDesc wd = w.desc[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double cost = 0.0, gmax = 0.0;
    int bad = 0;
    for (int j = wd.jf + wave; j < wd.jn; j += 4) {
        const int k0 = d.pt_start[j], k1 = d.pt_start[j + 1];
        double V[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } }, g[3] = { 0, 0, 0 };
        PointDer pd;
        point_der(d, j, d.points + 4 * j, true, pd);
        const double zero6[6] = { 0, 0, 0, 0, 0, 0 };
        for (int k = k0 + lane; k < k1; k += 64) {
            ObsLin o;
            linearize_obs(d, k, d.obs_cam[k], w.obs_lay[k], d.camder, pd, o);
            cost += 0.5 * o.rho0;
            store_record(a.obsrec, k, o, zero6);
            if (d.pdim) {
                for (int x = 0; x < 3; ++x) {
                    g[x] += o.Jp[0][x] * o.r[0] + o.Jp[1][x] * o.r[1];
                    for (int y = 0; y <= x; ++y) V[x][y] += o.Jp[0][x] * o.Jp[0][y] + o.Jp[1][x] * o.Jp[1][y];
                }
            }
        }
        if (!d.pdim) continue;
        // the track's sums, identical in all lanes from here on
        for (int x = 0; x < 3; ++x) {
            g[x] = wave_sum(g[x]);
            for (int y = 0; y <= x; ++y) V[x][y] = wave_sum(V[x][y]);
        }
        double Vi[3][3];
        const bool ok = point_track_finish(d, a, j, lane == 0, k1 == k0, V, g, Vi, gmax);
        if (!ok) bad = 1;
        if (a.mode == kPassScaleInit) continue;
        // Q = Jp V^-1 for every observation of the track (each lane its own records)
        for (int k = k0 + lane; k < k1; k += 64) {
            double2 *rec2 = reinterpret_cast<double2 *>(a.obsrec + (size_t)k * kObsRec);
            double jp[6], qv[6];
#pragma unroll
            for (int i = 0; i < 3; ++i) { const double2 v = rec2[kRecJp / 2 + i]; jp[2 * i] = v.x; jp[2 * i + 1] = v.y; }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    qv[3 * rr + t] = !ok ? 0.0 : jp[3 * rr] * Vi[0][t] + jp[3 * rr + 1] * Vi[1][t] + jp[3 * rr + 2] * Vi[2][t];
#pragma unroll
            for (int i = 0; i < 3; ++i) rec2[kRecQ / 2 + i] = make_double2(qv[2 * i], qv[2 * i + 1]);
        }
    }
    block_partial<false>(cost, a.partials, 0, sh, b, w.num);
    block_partial<true>(gmax, a.partials, 1, sh, b, w.num);
    block_partial<true>((double)bad, a.partials, 2, sh, b, w.num);
}

void launch_point_pass(const BaDev &d, const PointPassArgs &a, const ObsWindows &w, hipStream_t s)
{
    if (w.num > w.num_over) hipLaunchKernelGGL(ba_point_win_kernel, dim3(w.num - w.num_over), dim3(256), 0, s, d, a, w);
    if (w.num_over > 0) hipLaunchKernelGGL(ba_point_over_kernel, dim3(w.num_over), dim3(256), 0, s, d, a, w);
}

// ---------------------------------------------------------------------------
// pair pass: one wave per CHUNK of a camera pair's entry list (PairPassArgs::chunk entries).
// A pair with one chunk is finished by its wave.  The chunks of a longer list -- the
// diagonal pair of a camera holds all its observations, and a 3-camera local problem
// has six pairs with every track in each -- leave their sums in a scratch row each, and
// the chunk that arrives last (a ticket per pair) adds the rows in chunk order.
// The camera part of the gradient-norm test rides along on the diagonal pairs:
// g_c = sum_a Jc_a^T r_a (unscaled), |Plus(x, -g) - x|_inf per camera.
// ---------------------------------------------------------------------------
struct PairChunk { int pi, e0, e1, chunk, nchunks; };

// Bijective block -> logical block map that hands every XCD a contiguous range (workgroups are
// dealt to the 8 XCDs round robin).
__device__ __forceinline__ int xcd_remap_blocks(int bid, int total)
{
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

__device__ __forceinline__ double lane_value(double v, int src)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffu), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// The sums of a camera pair, complete: block of S, LM diagonal, reduced rhs, and for a
// diagonal pair the camera's gradient norm.  One lane.
__device__ __forceinline__ void
pair_finish(const BaDev &d, const PairPassArgs &a, int c1, int c2, int n1, int n2, int o1, int o2,
    double (&acc)[6][6], const double (&U)[6], const double (&rhs)[6], const double (&g)[6])
{
    // n1, n2, o1, o2: cam_ldim / cam_off of the two cameras, loaded by the caller in front of its sums
    const bool diag_pair = c1 == c2;
    // fully unrolled with predicates: a loop bounded by n1 / n2 would index the
    // accumulators dynamically and push all 36 of them into scratch memory
    if (a.mode == kPassScaleInit) {
        if (diag_pair) {
#pragma unroll
            for (int x = 0; x < 6; ++x)
                if (x < n1) a.scale_c_out[o1 + x] = 1.0 / (1.0 + sqrt(U[x]));
        }
        return;
    }
    if (diag_pair) {
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            if (x >= n1) continue;
            if (a.update_diag) a.diag_c[o1 + x] = fmin(fmax(U[x], a.min_diag), a.max_diag);
            acc[x][x] += a.diag_c[o1 + x] / a.radius;
            a.rhs[o1 + x] = rhs[x];
        }
    }
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int y = 0; y < 6; ++y)
            if (x < n1 && y < n2) {
                // (an ordered layout -- ba_order.hip -- can put camera c1 >= c2 in front of c2: the block then lies
                //  above the diagonal and is stored transposed; the factorisation reads the lower triangle)
                double *dst = o1 >= o2 ? a.S + (size_t)(o1 + x) * a.ldS + (o2 + y) : a.S + (size_t)(o2 + y) * a.ldS + (o1 + x);
                *dst = a.dense ? *dst + acc[x][y] : acc[x][y];
            }
    if (diag_pair && a.gmax_out && a.want_gradient) {
        double dl[6];
#pragma unroll
        for (int x = 0; x < 6; ++x) dl[x] = x < n1 ? -g[x] / d.scale_c[o1 + x] : 0.0;
        const double *cam = d.cams + 7 * c1;
        double out[7];
        for (int i = 0; i < 7; ++i) out[i] = cam[i];
        int t = 0;
        if (d.model == kModelQuat && n1 > 0 && d.cam_colmap[6 * c1] == 0) { quat_plus(cam, dl, out); t = 3; }
        for (; t < n1; ++t) {
            const int f = d.cam_colmap[6 * c1 + t];
            const int slot = d.model == kModelQuat ? f + 1 : f;     // full col 3,4,5 -> slots 4,5,6
            double dv = 0.0;
#pragma unroll
            for (int x = 0; x < 6; ++x) dv = t == x ? dl[x] : dv;
            out[slot] = cam[slot] + dv;
        }
        double gm = 0.0;
        for (int i = 0; i < 7; ++i) gm = fmax(gm, fabs(cam[i] - out[i]));
        store_sc1(&a.gmax_out[c1], gm);        // read by the last workgroup of this launch (ba_lm_post in its tail)
    }
}

// one wave's chunk (ent_lds / rec_lds: the kernel's LDS blocks, one per wave)
__device__ __forceinline__ void
pair_pass_wave(const BaDev &d, const PairPassArgs &a, int2 (*ent_lds)[64], double (*rec_lds)[2][64 * 18])
{
    const int lane = threadIdx.x & 63;
    // consecutive chunks (camera pairs sorted by c1, then c2) on the same XCD: the records of
    // camera c1 serve ~200 pairs in a row and stay in that XCD's L2 (round robin over the XCDs
    // put 15 cameras' worth of them in front of every L2: 25 % hits)
    const int wave = xcd_remap_blocks((int)blockIdx.x, (int)gridDim.x) * 4 + (int)(threadIdx.x >> 6);
    if (wave >= a.max_chunks) return;
    // A wave's life is a chain of memory latencies -- descriptor, entries, records, camera layout -- a few
    // times over; everything that can be asked for early is: the descriptor is one load, the camera layout
    // is requested right behind it, the entries of the next round while the records of this one are awaited.
    const PairChunkDesc dsc = a.chunk_desc[wave];
    if (dsc.nchunks == 0) return;
    PairChunk w;
    w.pi = dsc.pi; w.e0 = dsc.e0; w.e1 = dsc.e1; w.nchunks = dsc.nchunks; w.chunk = 0;
    const int c1 = dsc.c1, c2 = dsc.c2;
    const int e0 = w.e0, e1 = w.e1;
    const bool diag_pair = c1 == c2;
    if (a.mode == kPassScaleInit && !diag_pair) return;
    const int n1 = d.cam_ldim[c1], n2 = d.cam_ldim[c2], o1 = d.cam_off[c1], o2 = d.cam_off[c2];

    double acc[6][6];      // - sum Z_a W_b^T  (+ sum Jc^T Jc on the diagonal pair)
    double U[6];           // diagonal of sum Jc^T Jc (for the LM diagonal)
    double rhs[6];
    double g[6];           // sum Jc^T r (diagonal pairs: the camera's gradient)
#pragma unroll
    for (int x = 0; x < 6; ++x) {
        U[x] = 0.0; rhs[x] = 0.0; g[x] = 0.0;
#pragma unroll
        for (int y = 0; y < 6; ++y) acc[x][y] = 0.0;
    }
    // The records of a round of 64 entries come through LDS: a lane loading its own record with nine 16-byte
    // loads makes every load instruction touch 64 different cache lines, and the pass ran at the rate the
    // vector cache looks up tags (36 such instructions per round), not at any bandwidth.  Here the 64 records
    // of a side are 576 pieces of 16 bytes that the lanes fetch in piece order -- seven records, about
    // fourteen lines per instruction -- by DMA (global_load_lds: no registers in between) into a block of
    // the wave's own, record after record (144 B apart: the lanes' 16-byte reads of their own records then
    // fall on different banks).  No barrier: one wave's LDS operations execute in order.
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const int wv = (int)(threadIdx.x >> 6);
    const int wv_s = __builtin_amdgcn_readfirstlane(wv);
    uint64_t ent_next = a.entries[min(e0 + lane, e1 - 1)];
    for (int eb = e0; eb < e1; eb += 64) {
        const int e = eb + lane;
        const bool valid = e < e1;                    // lanes past the end repeat the last entry and add nothing
        const uint64_t ent = ent_next;
        if (eb + 64 < e1) ent_next = a.entries[min(e + 64, e1 - 1)];
        const int ka = (int)(ent >> 32), kb = (int)(ent & 0xffffffffu);
        const bool with_b = d.pdim && a.mode != kPassScaleInit && !a.dense;
        ent_lds[wv][lane] = make_int2(ka, kb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // of record a only Jc and Q (18 doubles at [kRecJc, kRecR)) -- the residual is read by the diagonal
        // entries alone; of record b only Jp and Jc (18 doubles at [0, kRecQ)).  Piece p = i * 64 + lane is
        // chunk p % 9 of the record of lane p / 9.
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int pz = i * 64 + lane, rec = (pz * 7282) >> 16, ch = pz - 9 * rec;      // pz / 9 for pz < 576
            const int2 kk = ent_lds[wv][rec];
            __builtin_amdgcn_global_load_lds((glb_void *)(a.obsrec + (size_t)kk.x * kObsRec + kRecJc + 2 * ch),
                (lds_void *)(uintptr_t)(&rec_lds[wv_s][0][i * 128]), 16, 0, 0);
            if (with_b)
                __builtin_amdgcn_global_load_lds((glb_void *)(a.obsrec + (size_t)kk.y * kObsRec + 2 * ch),
                    (lds_void *)(uintptr_t)(&rec_lds[wv_s][1][i * 128]), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        double ra[kObsRec];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double2 v = *reinterpret_cast<const double2 *>(&rec_lds[wv][0][lane * 18 + 2 * i]);
            ra[kRecJc + 2 * i] = v.x; ra[kRecJc + 2 * i + 1] = v.y;
        }
        // record b stays where it is and is read as it is used (36 registers less across the products below);
        // the next round's DMA cannot overtake these reads: they are consumed in this round
        const double *rb = &rec_lds[wv][1][lane * 18];
        if (!valid) continue;
        double Ja[2][6];
#pragma unroll
        for (int x = 0; x < 6; ++x) { Ja[0][x] = ra[kRecJc + x]; Ja[1][x] = ra[kRecJc + 6 + x]; }
        if (ka == kb) {
            const double2 rv = reinterpret_cast<const double2 *>(a.obsrec + (size_t)ka * kObsRec)[kRecR / 2];
            double rr0 = rv.x, rr1 = rv.y;
#pragma unroll
            for (int x = 0; x < 6; ++x) g[x] += Ja[0][x] * rr0 + Ja[1][x] * rr1;
            if (d.pdim && a.mode != kPassScaleInit) {
                // rhs = Jc^T (r - Q g):  Z g = Jc^T (Q g)
                const double *gp = a.ge + 3 * d.obs_pt[ka];
                rr0 -= ra[kRecQ] * gp[0] + ra[kRecQ + 1] * gp[1] + ra[kRecQ + 2] * gp[2];
                rr1 -= ra[kRecQ + 3] * gp[0] + ra[kRecQ + 4] * gp[1] + ra[kRecQ + 5] * gp[2];
            }
#pragma unroll
            for (int x = 0; x < 6; ++x) {
                rhs[x] += Ja[0][x] * rr0 + Ja[1][x] * rr1;
                U[x] += Ja[0][x] * Ja[0][x] + Ja[1][x] * Ja[1][x];
#pragma unroll
                for (int y = 0; y < 6; ++y) acc[x][y] += Ja[0][x] * Ja[0][y] + Ja[1][x] * Ja[1][y];
            }
        }
        if (with_b) {
            // Z_a W_b^T = Jc_a^T (Q_a Jp_b^T) Jc_b; of record b only Jc and Jp (18 doubles)
            double M[2][2];
#pragma unroll
            for (int r1 = 0; r1 < 2; ++r1)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
                    M[r1][r2] = ra[kRecQ + 3 * r1] * rb[kRecJp + 3 * r2] + ra[kRecQ + 3 * r1 + 1] * rb[kRecJp + 3 * r2 + 1] +
                                ra[kRecQ + 3 * r1 + 2] * rb[kRecJp + 3 * r2 + 2];
#pragma unroll
            for (int y = 0; y < 6; ++y) {
                const double jb0 = rb[kRecJc + y], jb1 = rb[kRecJc + 6 + y];
                const double t0 = M[0][0] * jb0 + M[0][1] * jb1;
                const double t1 = M[1][0] * jb0 + M[1][1] * jb1;
#pragma unroll
                for (int x = 0; x < 6; ++x) acc[x][y] -= Ja[0][x] * t0 + Ja[1][x] * t1;
            }
        }
    }
    // fixed-order wave reduction: lane v ends up with sum v (36 block entries, 6 diagonal sums, 6 rhs entries,
    // 6 gradient entries)
    double mine;
    {
        double s[kPairSums];
#pragma unroll
        for (int x = 0; x < 6; ++x) {
#pragma unroll
            for (int y = 0; y < 6; ++y) s[x * 6 + y] = acc[x][y];
            s[36 + x] = U[x]; s[42 + x] = rhs[x]; s[48 + x] = g[x];
        }
        mine = wave_reduce_54(s, lane);
    }
    if (w.nchunks > 1) {
        // A longer list: every chunk leaves its sums in a row of its own (write-through stores: the reader is
        // on another CU, maybe another XCD) and takes a ticket; the chunk that takes the last one adds the rows
        // -- in chunk order, whichever chunk it is -- and finishes the pair.  (This was a second launch, 17-25 us
        // of a 0.8 ms iteration for two hundred waves of work.)
        if (lane < kPairSums) store_sc1(a.chunk_partials + (size_t)wave * kPairSums + lane, mine);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(a.pair_ticket + w.pi, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != w.nchunks - 1) return;
        if (lane == 0) __hip_atomic_store(a.pair_ticket + w.pi, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
        const double *row = a.chunk_partials + (size_t)dsc.first * kPairSums + lane;
        double total = 0.0;
        if (lane < kPairSums) {
            // four rows per round, loads first (a plain loop is one memory latency per row)
            int c = 0;
            for (; c + 4 <= w.nchunks; c += 4) {
                const double v0 = load_sc1(row + (size_t)c * kPairSums), v1 = load_sc1(row + (size_t)(c + 1) * kPairSums);
                const double v2 = load_sc1(row + (size_t)(c + 2) * kPairSums), v3 = load_sc1(row + (size_t)(c + 3) * kPairSums);
                total += v0; total += v1; total += v2; total += v3;
            }
            for (; c < w.nchunks; ++c) total += load_sc1(row + (size_t)c * kPairSums);
        }
        mine = total;
    }
    // a pair with one chunk is finished here: the sums back into every lane (v_readlane: scalar registers)
#pragma unroll
    for (int x = 0; x < 6; ++x) {
#pragma unroll
        for (int y = 0; y < 6; ++y) acc[x][y] = lane_value(mine, x * 6 + y);
        U[x] = lane_value(mine, 36 + x); rhs[x] = lane_value(mine, 42 + x); g[x] = lane_value(mine, 48 + x);
    }
    if (lane == 0) pair_finish(d, a, c1, c2, n1, n2, o1, o2, acc, U, rhs, g);
}

__global__ __launch_bounds__(256, 2) void
ba_pair_pass_kernel(BaDev d, PairPassArgs a)
{
    __shared__ int lds_last;
    __shared__ double sh[256];
    __shared__ int2 ent_lds[4][64];
    __shared__ __attribute__((aligned(16))) double rec_lds[4][2][64 * 18];
    if (!lm_resolve(d)) {
        // a stopped solve: the state still goes to the host's slot (ba_lm_post did that)
        if (a.post.enabled && blockIdx.x == 0 && threadIdx.x == 0) publish_state(a.post.lm, a.post.host_out);
        return;
    }
    if (d.lm && a.mode == kPassNormal) {
        a.radius = d.lm->radius; a.update_diag = d.lm->update_diag; a.want_gradient = d.lm->want_gradient;
    }
    pair_pass_wave(d, a, ent_lds, rec_lds);
    if (!a.post.enabled) return;
    // ---- ba_lm_post in the tail of the workgroup that finishes last (the camera gradient norms of the diagonal
    //      pairs were stored write-through) ----
    if (!last_workgroup(a.post.ticket, &lds_last)) return;
    lm_post_block<true>(a.post.lm, a.post.prm, a.post.sc, a.post.initial, a.post.host_out, sh);
}

void launch_pair_pass(const BaDev &d, const PairPassArgs &a, hipStream_t s)
{
    if (a.num_pairs <= 0) return;
    const int blocks = (a.max_chunks + 3) / 4;
    hipLaunchKernelGGL(ba_pair_pass_kernel, dim3(blocks), dim3(256), 0, s, d, a);
}

// ---------------------------------------------------------------------------
// candidate cameras: x+ = Plus(x, scale * step), step = -y
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void
ba_cam_update_kernel(BaDev d, const double *y_c, double *cams_out, double *table_out, double *partials_cam)
{
    if (!lm_resolve(d)) return;
    if (d.lm) {
        if (d.lm->lin_failed) return;
        cams_out = d.lm->cur ? d.cams2[0] : d.cams2[1];
        table_out = d.lm->cur ? d.camder2[0] : d.camder2[1];
    }
    // four lanes per camera: each makes the candidate, and a quarter of its table
    const int q = blockIdx.x * blockDim.x + threadIdx.x, c = q >> 2;
    if (c >= d.C) return;
    cam_update_part(d, y_c, c, q & 3, cams_out, partials_cam, table_out, kCamDer);
}

void launch_cam_update(const BaDev &d, const double *y_c, double *cams_out, double *table_out, double *partials_cam, hipStream_t s)
{
    if (d.C > 0) hipLaunchKernelGGL(ba_cam_update_kernel, dim3((4 * d.C + 63) / 64), dim3(64), 0, s, d, y_c, cams_out, table_out, partials_cam);
}

// the table rows of cameras given as they are (the first iterate of a solve)
__global__ __launch_bounds__(64) void
ba_cam_derive_kernel(BaDev d, const double *cams, double *table_out)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x, c = q >> 2;
    if (c >= d.C) return;
    double cam[7];
    for (int i = 0; i < 7; ++i) cam[i] = cams[7 * c + i];
    cam_derive_part(d.model, cam, (double)d.img_w[c], (double)d.img_h[c], d.cam_colmap + 6 * c, d.cam_ldim[c], q & 3, true,
        table_out + (size_t)kCamDer * c);
}

void launch_cam_derive(const BaDev &d, const double *cams, double *table_out, hipStream_t s)
{
    if (d.C > 0) hipLaunchKernelGGL(ba_cam_derive_kernel, dim3((4 * d.C + 63) / 64), dim3(64), 0, s, d, cams, table_out);
}

// ---------------------------------------------------------------------------
// back substitution + model cost change + candidate points (+ fused: the candidate's cost and the LM decision)
// ---------------------------------------------------------------------------
// u = Jc y of one observation (lay: obs_lay, rec: its record's [0, kRecQ) part)
__device__ __forceinline__ void
obs_camera_step(const double *y_c, int lay, const double *rec, double &u0, double &u1)
{
    const int off = lay & 0xffffff, n = lay >> 24;
    u0 = 0.0; u1 = 0.0;
#pragma unroll
    for (int x = 0; x < 6; ++x)
        if (x < n) { const double yx = y_c[off + x]; u0 += rec[kRecJc + x] * yx; u1 += rec[kRecJc + 6 + x] * yx; }
}

// the state every back-pass kernel starts from; false: nothing to do (a stopped solve, or no step)
__device__ __forceinline__ bool
back_pass_begin(BaDev &d, BackPassArgs &a, const double *&cand, double *sh)
{
    const bool fused = a.fused != 0 && a.decide.enabled;
    if (!lm_resolve(d)) {
        if (fused && blockIdx.x == 0) lm_decide_block<true>(a.decide.lm, a.decide.prm, a.decide.sc, a.decide.host_out, sh);   // publishes the stopped state
        return false;
    }
    if (d.lm && d.lm->lin_failed) {
        // the linearisation failed: there is no step; the decision (an invalid step) is one workgroup's work
        if (fused && blockIdx.x == 0) lm_decide_block<true>(a.decide.lm, a.decide.prm, a.decide.sc, a.decide.host_out, sh);
        return false;
    }
    // fused: the table rows of the CANDIDATE cameras (ba_cam_update / chol_small made them just before)
    cand = nullptr;
    if (d.lm) { a.points_out = d.lm->cur ? d.points2[0] : d.points2[1]; cand = d.lm->cur ? d.camder2[0] : d.camder2[1]; }
    return true;
}

// behind a back-pass kernel's sums: the partials, and ba_lm_decide in the tail of the workgroup that finishes last
__device__ __forceinline__ void
back_pass_end(const BackPassArgs &a, double mcc, double sn, double xn, double ccost, int index, int count, double *sh, int *lds_last)
{
    if (!a.fused) {
        block_partial<false>(mcc, a.partials, 0, sh, index, count);
        block_partial<false>(sn, a.partials, 1, sh, index, count);
        block_partial<false>(xn, a.partials, 2, sh, index, count);
        return;
    }
    block_partial<false, true>(mcc, a.partials, 0, sh, index, count);
    block_partial<false, true>(sn, a.partials, 1, sh, index, count);
    block_partial<false, true>(xn, a.partials, 2, sh, index, count);
    block_partial<false, true>(ccost, a.cost_partials, 0, sh, index, count);
    if (!a.decide.enabled) return;           // (the launch behind this one carries the decision)
    if (!last_workgroup(a.decide.ticket, lds_last)) return;
    lm_decide_block<true>(a.decide.lm, a.decide.prm, a.decide.sc, a.decide.host_out, sh);
}

__global__ __launch_bounds__(256) void
ba_back_win_kernel(BaDev d, BackPassArgs a, ObsWindows w)
{
    __shared__ double t_lds[3][256];       // - Jp^T (Jc y) of the window's observations
    __shared__ double sh[256];
    __shared__ int lds_last;
    const double *cand;
    if (!back_pass_begin(d, a, cand, sh)) return;
    const bool fused = a.fused != 0;
    const int b = w.ok_list[blockIdx.x], tid = threadIdx.x;          // (the other windows: ba_back_over_kernel)
    const WinDesc wd = w.desc[b];
    const int k = wd.ka + tid;
    const bool live = k < wd.kb;
    double mcc = 0.0, sn = 0.0, xn = 0.0, ccost = 0.0;
    int j = 0, ks = 0, ke = 0;
    double r[kObsRec], u0 = 0.0, u1 = 0.0, cd[kCamCost];
    double2 xy = make_double2(0.0, 0.0);
    if (live) {
        // Jc, Jp and the residual of the record, the loads issued together (brought in stream order through LDS like
        // the point pass stores them, the pass took 91 us instead of 87: its scattered loads are not what it waits for)
        const double2 *src = reinterpret_cast<const double2 *>(a.obsrec + (size_t)k * kObsRec);
#pragma unroll
        for (int i = 0; i < kRecQ / 2; ++i) { const double2 v = src[i]; r[2 * i] = v.x; r[2 * i + 1] = v.y; }
        { const double2 v = src[kRecR / 2]; r[kRecR] = v.x; r[kRecR + 1] = v.y; }
        j = d.obs_pt[k];
        if (fused) {
            const double2 *q = reinterpret_cast<const double2 *>(cand + (size_t)kCamDer * d.obs_cam[k]);
#pragma unroll
            for (int i = 0; i < kCamCost / 2; ++i) { const double2 v = q[i]; cd[2 * i] = v.x; cd[2 * i + 1] = v.y; }
            xy = reinterpret_cast<const double2 *>(d.obs_xy)[k];
        }
        ks = d.pt_start[j]; ke = d.pt_start[j + 1];
        obs_camera_step(a.y_c, w.obs_lay[k], r, u0, u1);
#pragma unroll
        for (int t = 0; t < 3; ++t) t_lds[t][tid] = -(r[kRecJp + t] * u0 + r[kRecJp + 3 + t] * u1);
    }
    __syncthreads();
    if (live) {
        double step_p[3] = { 0, 0, 0 };
        if (d.pdim) {
            double t3[3] = { 0, 0, 0 };
            for (int i = ks - wd.ka; i < ke - wd.ka; ++i)
                for (int t = 0; t < 3; ++t) t3[t] += t_lds[t][i];
            for (int t = 0; t < 3; ++t) t3[t] = a.ge[3 * j + t] + t3[t];
            const double *Vi = a.vinv + 9 * j;
            for (int x = 0; x < 3; ++x) step_p[x] = -(Vi[3 * x] * t3[0] + Vi[3 * x + 1] * t3[1] + Vi[3 * x + 2] * t3[2]);
        }
        // model_cost_change = -(J step)^T (r + J step / 2)  (scaled J, scaled step)
        double m0 = -u0, m1 = -u1;
#pragma unroll
        for (int t = 0; t < 3; ++t) { m0 += r[kRecJp + t] * step_p[t]; m1 += r[kRecJp + 3 + t] * step_p[t]; }
        mcc = -(m0 * (r[kRecR] + m0 / 2.0) + m1 * (r[kRecR + 1] + m1 / 2.0));
        // the candidate point, in every lane of the track; its first lane stores it
        const double *P = d.points + 4 * j;
        double out[4] = { P[0], P[1], P[2], P[3] };
        const bool leader = k == ks;
        if (d.pdim) {
            double dl[3];
            for (int x = 0; x < 3; ++x) dl[x] = step_p[x] * d.scale_p[3 * j + x];
            homog_plus(P, dl, out);
            if (leader)
                for (int x = 0; x < 4; ++x) { sn += (P[x] - out[x]) * (P[x] - out[x]); xn += P[x] * P[x]; }
        }
        if (leader)
            for (int x = 0; x < 4; ++x) a.points_out[4 * j + x] = out[x];
        if (fused) {
            const double pc[3] = { out[0] / out[3], out[1] / out[3], out[2] / out[3] };
            ccost = 0.5 * row_cost(cd, pc, xy.x, xy.y, d.huber);
        }
    }
    for_empty_tracks(d, wd, [&](int je) {
            // no observations: no step (ge = 0)
            const double *P = d.points + 4 * je;
            for (int x = 0; x < 4; ++x) { a.points_out[4 * je + x] = P[x]; if (d.pdim) xn += P[x] * P[x]; }
        });
    back_pass_end(a, mcc, sn, xn, ccost, b, w.num, sh, &lds_last);
}

__global__ __launch_bounds__(256) void
ba_back_over_kernel(BaDev d, BackPassArgs a, ObsWindows w)
{
    __shared__ double sh[256];
    __shared__ int lds_last;
    const double *cand;
    if (!back_pass_begin(d, a, cand, sh)) return;
    const bool fused = a.fused != 0;
    const int b = w.over_list[blockIdx.x];
    const WinDesc wd = w.desc[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double mcc = 0.0, sn = 0.0, xn = 0.0, ccost = 0.0;
    // a wave per track (ba_point_over_kernel)
    for (int j = wd.jf + wave; j < wd.jn; j += 4) {
        const int k0 = d.pt_start[j], k1 = d.pt_start[j + 1];
        double step_p[3] = { 0, 0, 0 };
        if (d.pdim) {
            double t3[3] = { 0, 0, 0 };
            for (int k = k0 + lane; k < k1; k += 64) {
                double r[kRecQ];
                {
                    const double2 *src = reinterpret_cast<const double2 *>(a.obsrec + (size_t)k * kObsRec);
#pragma unroll
                    for (int i = 0; i < kRecQ / 2; ++i) { const double2 v = src[i]; r[2 * i] = v.x; r[2 * i + 1] = v.y; }
                }
                double u0, u1;
                obs_camera_step(a.y_c, w.obs_lay[k], r, u0, u1);
#pragma unroll
                for (int t = 0; t < 3; ++t) t3[t] -= r[kRecJp + t] * u0 + r[kRecJp + 3 + t] * u1;
            }
            for (int t = 0; t < 3; ++t) t3[t] = a.ge[3 * j + t] + wave_sum(t3[t]);
            const double *Vi = a.vinv + 9 * j;
            for (int x = 0; x < 3; ++x) step_p[x] = -(Vi[3 * x] * t3[0] + Vi[3 * x + 1] * t3[1] + Vi[3 * x + 2] * t3[2]);
        }
        for (int k = k0 + lane; k < k1; k += 64) {
            double r[kObsRec];
            {
                const double2 *src = reinterpret_cast<const double2 *>(a.obsrec + (size_t)k * kObsRec);
#pragma unroll
                for (int i = 0; i < kRecQ / 2; ++i) { const double2 v = src[i]; r[2 * i] = v.x; r[2 * i + 1] = v.y; }
                const double2 v = src[kRecR / 2]; r[kRecR] = v.x; r[kRecR + 1] = v.y;
            }
            double u0, u1;
            obs_camera_step(a.y_c, w.obs_lay[k], r, u0, u1);
            double m0 = -u0, m1 = -u1;
#pragma unroll
            for (int t = 0; t < 3; ++t) { m0 += r[kRecJp + t] * step_p[t]; m1 += r[kRecJp + 3 + t] * step_p[t]; }
            mcc -= m0 * (r[kRecR] + m0 / 2.0) + m1 * (r[kRecR + 1] + m1 / 2.0);
        }
        const double *P = d.points + 4 * j;
        double out[4] = { P[0], P[1], P[2], P[3] };
        if (d.pdim) {
            double dl[3];
            for (int x = 0; x < 3; ++x) dl[x] = step_p[x] * d.scale_p[3 * j + x];
            homog_plus(P, dl, out);
            if (lane == 0)
                for (int x = 0; x < 4; ++x) { sn += (P[x] - out[x]) * (P[x] - out[x]); xn += P[x] * P[x]; }
        }
        if (lane == 0)
            for (int x = 0; x < 4; ++x) a.points_out[4 * j + x] = out[x];
        if (fused) {
            const double pc[3] = { out[0] / out[3], out[1] / out[3], out[2] / out[3] };
            for (int k = k0 + lane; k < k1; k += 64) ccost += 0.5 * obs_cost_at(d, k, cand, kCamDer, pc);
        }
    }
    back_pass_end(a, mcc, sn, xn, ccost, b, w.num, sh, &lds_last);
}

void launch_back_pass(const BaDev &d, const BackPassArgs &a, const ObsWindows &w, hipStream_t s)
{
    // the decision rides in the tail of the launch that comes last
    BackPassArgs first = a;
    if (w.num_over > 0) first.decide.enabled = 0;
    if (w.num > w.num_over) hipLaunchKernelGGL(ba_back_win_kernel, dim3(w.num - w.num_over), dim3(256), 0, s, d, first, w);
    if (w.num_over > 0) hipLaunchKernelGGL(ba_back_over_kernel, dim3(w.num_over), dim3(256), 0, s, d, a, w);
}

// ---------------------------------------------------------------------------
// cost at (cameras with table rows `table`, points) given explicitly: a lane per observation
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void
ba_cost_pass_kernel(BaDev d, const double *table, const double *points, double *partials, ObsWindows w)
{
    __shared__ double sh[4];
    if (d.lm) {
        // LM solve: the cost of the CANDIDATE (the iterate buffer that is not current)
        if (d.lm->stop || d.lm->lin_failed || d.lm->flow_aborted) return;
        table = d.lm->cur ? d.camder2[0] : d.camder2[1]; points = d.lm->cur ? d.points2[0] : d.points2[1];
    }
    const WinDesc wd = w.desc[blockIdx.x];
    double cost = 0.0;
    for (int k = wd.ka + (int)threadIdx.x; k < wd.kb; k += 256) {
        const double *P = points + 4 * d.obs_pt[k];
        const double p[3] = { P[0] / P[3], P[1] / P[3], P[2] / P[3] };
        cost += 0.5 * obs_cost_at(d, k, table, kCamDer, p);
    }
    block_partial<false>(cost, partials, 0, sh, blockIdx.x, w.num);
}

// table: the derived table rows of the cameras (ba_cam_derive / ba_cam_update)
void launch_cost_pass(const BaDev &d, const double *table, const double *points, double *partials,
    const ObsWindows &w, hipStream_t s)
{
    hipLaunchKernelGGL(ba_cost_pass_kernel, dim3(w.num), dim3(256), 0, s, d, table, points, partials, w);
}

// ---------------------------------------------------------------------------
// final fixed-order reduction of the per-block partials into scalars
//   out[slot] = reduce(partials[slot][0..n))
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void
ba_reduce_kernel(const double *partials, int n, int num_slots, unsigned max_mask, double *out,
    const double *extra, int extra_n, int extra_stride, int extra_slots)
{
    __shared__ double sh[256];
    for (int slot = 0; slot < num_slots + extra_slots; ++slot) {
        const bool is_extra = slot >= num_slots;
        const bool is_max = !is_extra && ((max_mask >> slot) & 1u);
        double v = 0.0;
        if (!is_extra) {
            for (int i = threadIdx.x; i < n; i += 256) {
                const double x = partials[(size_t)slot * n + i];
                v = is_max ? fmax(v, x) : v + x;
            }
        } else {
            const int es = slot - num_slots;
            for (int i = threadIdx.x; i < extra_n; i += 256) v += extra[(size_t)i * extra_stride + es];
        }
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int st = 128; st >= 1; st >>= 1) {
            if ((int)threadIdx.x < st)
                sh[threadIdx.x] = is_max ? fmax(sh[threadIdx.x], sh[threadIdx.x + st]) : sh[threadIdx.x] + sh[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[slot] = sh[0];
        __syncthreads();
    }
}

void launch_reduce(const double *partials, int n, int num_slots, unsigned max_mask, double *out,
    const double *extra, int extra_n, int extra_stride, int extra_slots, hipStream_t s)
{
    hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, s, partials, n, num_slots, max_mask, out,
        extra, extra_n, extra_stride, extra_slots);
}

__global__ void
ba_max_reduce_kernel(const double *v, int n, double *out)
{
    __shared__ double sh[256];
    double m = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) m = fmax(m, v[i]);
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

// zero, with the identity on the padding diagonal
__global__ __launch_bounds__(256) void
ba_reset_system_kernel(double2 *S2, size_t pairs, double *S, int ld, int n, int N)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // element (r, r) of the padding diagonal, r in [n, N): 1, everything else 0
    if (i < pairs) {
        double2 v = {0.0, 0.0};
        const size_t e = 2 * i, row = e / (size_t)ld, col = e - row * (size_t)ld;     // ld is even (a multiple of 32)
        if (row >= (size_t)n && row < (size_t)N) {
            if (col == row) v.x = 1.0;
            if (col + 1 == row) v.y = 1.0;
        }
        S2[i] = v;
    }
}

void launch_reset_system(double *S, size_t elems, int ld, int n, int N, hipStream_t s)
{
    const size_t pairs = elems / 2;
    hipLaunchKernelGGL(ba_reset_system_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s,
        reinterpret_cast<double2 *>(S), pairs, S, ld, n, N);
}

// v[i] = value (the unit Jacobi scales before their first estimate)
__global__ void
ba_fill_kernel(double *v, size_t n, double value)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = value;
}

void launch_fill(double *v, size_t n, double value, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(ba_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, n, value);
}

// obs_pt[k] = j for pt_start[j] <= k < pt_start[j + 1] (observations are grouped by
// point: the per-observation point index is redundant with the CSR and is expanded
// here instead of crossing PCIe)
__global__ void
ba_expand_points_kernel(const int32_t *__restrict__ pt_start, int M, int32_t *__restrict__ obs_pt)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    for (int k = pt_start[j]; k < pt_start[j + 1]; ++k) obs_pt[k] = j;
}

void launch_expand_points(const int32_t *pt_start, int M, int32_t *obs_pt, hipStream_t s)
{
    if (M > 0) hipLaunchKernelGGL(ba_expand_points_kernel, dim3((M + 255) / 256), dim3(256), 0, s, pt_start, M, obs_pt);
}

void launch_max_reduce(const double *v, int n, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(ba_max_reduce_kernel, dim3(1), dim3(256), 0, s, v, n, out);
}

// ---------------------------------------------------------------------------
// B7: batched evaluateReprojectionError (OrthoQuaternionRecoAlgorithm.cpp:175-194)
// ---------------------------------------------------------------------------
__global__ void
ba_reproj_kernel(BaDev d, double *err, double *residuals)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= d.O) return;
    const int c = d.obs_cam[k], j = d.obs_pt[k];
    ObsFull e;
    if (d.model == kModelQuat)
        eval_quat(d.cams + 7 * c, d.points + 4 * j, (double)d.img_w[c], (double)d.img_h[c],
            d.obs_xy[2 * k], d.obs_xy[2 * k + 1], false, e);
    else
        eval_euler(d.cams + 7 * c, d.points + 4 * j, (double)d.img_w[c], (double)d.img_h[c],
            d.obs_xy[2 * k], d.obs_xy[2 * k + 1], false, e);
    if (residuals) { residuals[2 * k] = e.r[0]; residuals[2 * k + 1] = e.r[1]; }
    if (err) err[k] = sqrt(e.r[0] * e.r[0] + e.r[1] * e.r[1]);
}

void launch_reproj(const BaDev &d, double *err, double *residuals, hipStream_t s)
{
    if (d.O <= 0) return;
    hipLaunchKernelGGL(ba_reproj_kernel, dim3((d.O + 255) / 256), dim3(256), 0, s, d, err, residuals);
}

// ---------------------------------------------------------------------------
// B8: triangulateOrthographicTracks + intersectRays
// (src/triangulation/triangulation.cpp:11-93; camera accessors
// OrthoQuaternionCamera.cpp:45-59, OrthographicCamera.cpp:63-95,187-193)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void quat_rot(const double *q, const double v[3], double out[3])
{
    const double u[3] = { q[0], q[1], q[2] };
    double t[3];
    cross3(u, v, t);
    t[0] *= 2.0; t[1] *= 2.0; t[2] *= 2.0;
    double ut[3];
    cross3(u, t, ut);
    out[0] = v[0] + q[3] * t[0] + ut[0];
    out[1] = v[1] + q[3] * t[1] + ut[1];
    out[2] = v[2] + q[3] * t[2] + ut[2];
}

__device__ void camera_ray(const BaDev &d, int c, double x, double y, double origin[3], double dir[3])
{
    const double *cam = d.cams + 7 * c;
    const double W = (double)d.img_w[c], H = (double)d.img_h[c];
    if (d.model == kModelQuat) {
        const double xn = -2.0 * ((x / W) - 0.5) + cam[4];
        const double yn = -2.0 * ((y / H) - 0.5) + cam[5];
        const double loc[3] = { cam[6] * xn, cam[6] * yn, -10.0 };
        const double z[3] = { 0.0, 0.0, 1.0 };
        quat_rot(cam, loc, origin);
        quat_rot(cam, z, dir);
    } else {
        const double om = cam[1] + 0.5 * 3.14159265358979323846, ph = cam[0], ro = cam[2];
        const double Ry[3][3] = { { cos(ro), -sin(ro), 0 }, { sin(ro), cos(ro), 0 }, { 0, 0, 1 } };
        const double Rx[3][3] = { { 1, 0, 0 }, { 0, cos(om), -sin(om) }, { 0, sin(om), cos(om) } };
        const double Rz[3][3] = { { cos(ph), -sin(ph), 0 }, { sin(ph), cos(ph), 0 }, { 0, 0, 1 } };
        double A[3][3], S[3][3];
        mat3_mul(Rz, Rx, A);
        mat3_mul(A, Ry, S);
        const double xn = -2.0 * ((x / W) - 0.5) + cam[3];
        const double yn = -2.0 * ((y / H) - 0.5) + cam[4];
        // toCameraSpace(v) = T^T S v = (s0, s2, -s1)
        auto tcs = [&](double v0, double v1, double v2, double o[3]) {
            const double s0 = S[0][0] * v0 + S[0][1] * v1 + S[0][2] * v2;
            const double s1 = S[1][0] * v0 + S[1][1] * v1 + S[1][2] * v2;
            const double s2 = S[2][0] * v0 + S[2][1] * v1 + S[2][2] * v2;
            o[0] = s0; o[1] = s2; o[2] = -s1;
        };
        double ax[3], ay[3], org[3];
        tcs(1, 0, 0, ax);
        tcs(0, 1, 0, ay);
        tcs(0, 0, -10.0, org);
        tcs(0, 0, 1, dir);
        for (int i = 0; i < 3; ++i) origin[i] = org[i] + xn * ax[i] * cam[5] + yn * ay[i] * cam[5];
    }
}

__device__ void eig3_jacobi(double A[3][3], double V[3][3], double w[3])
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int pp = 0; pp < 2; ++pp)
            for (int q = pp + 1; q < 3; ++q) {
                if (fabs(A[pp][q]) < 1e-300) continue;
                const double th = (A[q][q] - A[pp][pp]) / (2.0 * A[pp][q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][pp], akq = A[k][q];
                    A[k][pp] = cs * akp - sn * akq; A[k][q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[pp][k], aqk = A[q][k];
                    A[pp][k] = cs * apk - sn * aqk; A[q][k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][pp], vkq = V[k][q];
                    V[k][pp] = cs * vkp - sn * vkq; V[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}

__global__ void
ba_triangulate_kernel(BaDev d, double *points_out, uint8_t *valid)
{
    // eight neighbouring lanes per track: a thread per track walked up to C observations (200 - 250 in the global
    // adjustments of the end-to-end jobs) on 2500 threads; the eight partial sums are folded in a fixed order
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = gt >> 3, sub = gt & 7;
    if (j >= d.M) return;                       // (an octet never straddles j < M)
    const int k0 = d.pt_start[j], k1 = d.pt_start[j + 1];
    if (k1 - k0 < 2) { if (valid && sub == 0) valid[j] = 0; return; }
    double R[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } }, q[3] = { 0, 0, 0 };
    for (int k = k0 + sub; k < k1; k += 8) {
        double o3[3], dir[3];
        camera_ray(d, d.obs_cam[k], d.obs_xy[2 * k], d.obs_xy[2 * k + 1], o3, dir);
        const double n = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        dir[0] /= n; dir[1] /= n; dir[2] /= n;
        for (int a = 0; a < 3; ++a) {
            double row[3];
            for (int b = 0; b < 3; ++b) { row[b] = (a == b ? 1.0 : 0.0) - dir[a] * dir[b]; R[a][b] += row[b]; }
            q[a] += row[0] * o3[0] + row[1] * o3[1] + row[2] * o3[2];
        }
    }
    for (int a = 0; a < 3; ++a) {
        for (int m = 1; m < 8; m <<= 1) q[a] += __shfl_xor(q[a], m);
        for (int b = 0; b < 3; ++b)
            for (int m = 1; m < 8; m <<= 1) R[a][b] += __shfl_xor(R[a][b], m);
    }
    double A[3][3], V[3][3], w[3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) A[a][b] = R[a][b];
    eig3_jacobi(A, V, w);
    const double wmax = fmax(fabs(w[0]), fmax(fabs(w[1]), fabs(w[2])));
    const double thr = fmax(wmax * 3.0 * 2.220446049250313e-16, 2.2250738585072014e-308);
    double x[3] = { 0, 0, 0 };
    for (int i = 0; i < 3; ++i) {
        if (!(fabs(w[i]) > thr)) continue;
        const double c = (V[0][i] * q[0] + V[1][i] * q[1] + V[2][i] * q[2]) / w[i];
        x[0] += c * V[0][i]; x[1] += c * V[1][i]; x[2] += c * V[2][i];
    }
    if (sub != 0) return;
    points_out[4 * j] = x[0]; points_out[4 * j + 1] = x[1]; points_out[4 * j + 2] = x[2]; points_out[4 * j + 3] = 1.0;
    if (valid) valid[j] = 1;
}

void launch_triangulate(const BaDev &d, double *points_out, uint8_t *valid, hipStream_t s)
{
    if (d.M <= 0) return;
    hipLaunchKernelGGL(ba_triangulate_kernel, dim3((unsigned)(((int64_t)d.M * 8 + 127) / 128)), dim3(128), 0, s, d, points_out, valid);
}

}  // namespace osfm
