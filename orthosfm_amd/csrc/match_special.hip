// Special descriptors of the matching path (hot path A): SIFT descriptors with a byte > 127.
//
// MVE renormalises a SIFT descriptor after clamping it at 0.2 (src/mve/sfm/sift.cc:830-839),
// so a descriptor whose energy sits in a few bins comes out with entries of 0.5-0.7, i.e.
// bytes 128..180 after convert_descriptor (exhaustive_matching.cc:17-27).  Such a descriptor
// does not fit the raw int8 operand of the correction-free tile kernel.  Real images hold a
// handful of them per view, so they must not decide which kernel the other 20000 descriptors
// of the view take: the tile kernel always runs the correction-free form with these
// descriptors blanked (zero rows / columns: every score 0, which cannot change a result
// because the reference's running state starts at (0, 0)), and this kernel scores the few
// special descriptors of a view against ALL descriptors of the other view exactly:
//
//   rows    = one unit of 32 special descriptors (value - 128 form, correction through the
//             MFMA C operand), resident in registers;
//   columns = the other view's full value - 128 bank, streamed from L2 / HBM straight into
//             MFMA fragments (no LDS staging: 1-2 % of the tile kernel's work);
// and reduces them the way the tile kernel does -- maxima only, the finish kernel re-scores
// what decides a result (with exact top-2 reductions in both directions this kernel spent six
// vector operations per score, and 200 special rows per view cost 6.3 ms per 1225 pairs):
//   row direction    -> per special descriptor and chunk of 4096 candidates the largest
//             inner product of each (wave, lane) STREAM of 32 candidates (columns
//             base + 128 j), merged to (best stream, its maximum, second largest stream
//             maximum); the finish kernel re-scores the winning stream of a query that passes
//             the ratio test against that lower bound of the second best;
//   column direction -> for every streamed descriptor the largest inner product over the
//             special descriptors (one int32, units folded by the lane that owns the column);
//             the finish kernel needs more only when that value beats every ordinary
//             candidate AND passes the ratio test against the best of them -- then it scans
//             the other view's special descriptors itself (rare: a landmark whose descriptor is
//             special in one view and ordinary in the other).
// match_finish_kernel merges both with the tile kernel's partials (match_kernels.hip).
#include <climits>

#include "match_kernels.h"

namespace osfm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

namespace {

// (best, index, second) <- candidate (ip, idx) with its own second; larger index wins ties
__device__ __forceinline__ void fold_top2(int &bip, int &bidx, int &sec, int ip, int idx, int ip2)
{
    sec = max(max(sec, ip2), min(bip, ip));
    if (ip > bip || (ip == bip && idx > bidx)) { bip = ip; bidx = idx; }
}

}  // namespace

// One workgroup = one chunk of 4096 streamed descriptors x up to kSpSlots UNITS of 32 special rows that all
// want that stream: the kernel is bound by streaming the other view out of L2 (8 TB/s of 16-byte
// fragment loads: 0.8 ms per 1225 pairs and unit with one unit per pass), not by its vector work.
//   count > 0 : `count` different (problem, side) entries of one unit each that stream the same view
//               (the common case: a handful of special rows per view) -- one pass, every entry writes
//               its own column results;
//   count == 0: ONE entry with many units -- passes of kSpSlots units; the column results of a pass are
//               folded in registers and across passes by the lane that owns the column.
__global__ __launch_bounds__(256, 2) void
match_special_kernel(const MatchProblem *__restrict__ problems, const SpecialJob *__restrict__ jobs,
    RowPart *__restrict__ sp, int32_t *__restrict__ sp_col)
{
    constexpr int NS = kSpSlots;               // units per pass (with four the kernel spills: 448 B of scratch in the step loop)
    __shared__ int sk[4][16][64];
    __shared__ RowPart wres[4][32];

    const SpecialJob job = jobs[blockIdx.x];
    const bool many = job.count == 0;
    // the streamed operand: the other set of entry 0 (the same view for every entry of the job)
    const MatchProblem &p0 = problems[job.problem[0]];
    const int8_t *__restrict__ O = job.side[0] == 0 ? p0.B : p0.A;
    const int32_t *__restrict__ corrO = job.side[0] == 0 ? p0.corrB : p0.corrA;
    const int no = job.side[0] == 0 ? p0.n2 : p0.n1;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int col0 = job.chunk * kSpChunk;
    const int nsteps = (min(no, col0 + kSpChunk) - col0 + 31) / 32;      // <= 128: 32 per wave
    const int ns0 = job.side[0] == 0 ? p0.nsA : p0.nsB;
    const int npass = many ? (ns0 + 32 * NS - 1) / (32 * NS) : 1;

    for (int pass = 0; pass < npass; ++pass) {
        // the slots of this pass: (problem, side, unit)
        const int8_t *S[NS];
        RowPart *rowres[NS];
        int32_t *colres[NS];
        v16i ra[NS];
        int nslots = 0;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int e = many ? 0 : min(t, job.count - 1);
            const MatchProblem &pd = problems[job.problem[e]];
            const int side = job.side[e];
            const int ns = side == 0 ? pd.nsA : pd.nsB;
            const int unit = many ? pass * NS + t : 0;
            const bool live = many ? unit * 32 < ns : t < job.count;
            if (live) nslots = t + 1;
            const int u = live ? unit : 0;
            const int ns_pad = (ns + 31) & ~31;
            S[t] = (side == 0 ? pd.A_special : pd.B_special) + (size_t)u * 32 * 128;
            rowres[t] = sp + pd.sp_row_off[side] + (int64_t)job.chunk * ns_pad + u * 32;
            colres[t] = sp_col + pd.sp_col_off[side];
            const int32_t *corrS = (side == 0 ? pd.corrA_special : pd.corrB_special) + u * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) ra[t][r] = corrS[(r & 3) + 8 * (r >> 2) + 4 * lh];
        }
        // resident A fragments of the slots' rows
        v4i a[NS][4];
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int8_t *srow = S[t] + (size_t)lr * 128 + lh * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[t][ks] = *reinterpret_cast<const v4i *>(srow + ks * 32);
        }
        v16i kmax[NS];                                 // largest exact inner product of this lane's stream, per slot and row
#pragma unroll
        for (int t = 0; t < NS; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) kmax[t][r] = INT_MIN;

        // One step in flight beside the one being reduced (measured: a third stage is slower, 0.72 against 0.60 ms per
        // 1225 pairs with one unit per view): operands come straight from L2 / HBM.
        struct Stage { v4i b[4]; int cb; int old; };
        auto fetch = [&](Stage &st, int step) {
            const int col = col0 + min(step, nsteps - 1) * 32 + lr;     // below the bank's 256-row padding
            const int8_t *p = O + (size_t)col * 128 + lh * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) st.b[ks] = *reinterpret_cast<const v4i *>(p + ks * 32);
            st.cb = corrO[col];
            // the fold of the earlier passes (same lane wrote it: program order)
            st.old = INT_MIN;
            if (many && pass > 0 && lh == 0) st.old = colres[0][col];
        };
        auto process = [&](const Stage &st, int step) {
            const int cb = st.cb;
            const int col = col0 + step * 32 + lr;
            int gall = INT_MIN;                        // folded across the slots of a many-unit entry (cb added at the end)
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                if (t >= nslots) break;
                v16i acc = ra[t];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t][ks], st.b[ks], acc, 0, 0, 0);
                // row direction: exact ip = acc + cb (acc carries the row correction), maxima only
#pragma unroll
                for (int r = 0; r < 16; ++r) kmax[t][r] = max(kmax[t][r], acc[r] + cb);
                // column direction: the largest of this lane's 16 rows of column `col`
                int g = max(acc[0], acc[1]);
#pragma unroll
                for (int r = 2; r < 16; r += 2) g = max(max(g, acc[r]), acc[r + 1]);
                g = max(g, __shfl_xor(g, 32));
                if (many) gall = max(gall, g);
                else if (lh == 0) colres[t][col] = g + cb;
            }
            if (many && lh == 0) colres[0][col] = max(st.old, gall + cb);
        };
        Stage s0, s1;
        int step = wave;
        if (step < nsteps) { fetch(s0, step); fetch(s1, step + 4); }
        while (step < nsteps) {
            process(s0, step); fetch(s0, step + 8); step += 4;
            if (step >= nsteps) break;
            process(s1, step); fetch(s1, step + 8); step += 4;
        }

        // ---- row direction: the 32 lane streams of a half-wave, then the four waves ----
        // (best stream = its first candidate column, col0 + wave * 32 + lane; its candidates follow
        //  at a stride of 128 columns)
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            if (t >= nslots) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) sk[wave][r][lane] = kmax[t][r];
            __syncthreads();
            {
                const int r = lane & 15, h = (lane >> 4) & 1, part = lane >> 5;
                int bip = INT_MIN, bcol = -1, sec = INT_MIN;
                for (int i = 0; i < 16; ++i) {
                    const int l = part * 16 + ((i + r) & 15);             // skewed: bank-conflict free
                    const int kb = sk[wave][r][h * 32 + l];
                    // SIFT inner products are >= 0: a negative maximum is "no step" or padding columns only
                    if (kb >= 0) fold_top2(bip, bcol, sec, kb, col0 + wave * 32 + l, INT_MIN);
                }
                const int obip = __shfl_xor(bip, 32), obcol = __shfl_xor(bcol, 32), osec = __shfl_xor(sec, 32);
                fold_top2(bip, bcol, sec, obip, obcol, osec);
                if (part == 0) {
                    RowPart out;
                    out.ip_best = bip; out.idx_best = bcol; out.ip_second = sec; out.pad = 0;
                    wres[wave][(r & 3) + 8 * (r >> 2) + 4 * h] = out;
                }
            }
            __syncthreads();
            if (tid < 32) {
                int bip = INT_MIN, bcol = -1, sec = INT_MIN;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const RowPart p = wres[w][tid];
                    fold_top2(bip, bcol, sec, p.ip_best, p.idx_best, p.ip_second);
                }
                RowPart out;
                out.ip_best = bip; out.idx_best = max(bcol, 0); out.ip_second = sec; out.pad = 3;
                rowres[t][tid] = out;
            }
            __syncthreads();
        }
    }
}

void launch_match_special(const MatchProblem *d_problems, const SpecialJob *d_jobs, int num_jobs,
    RowPart *sp_parts, int32_t *sp_col, hipStream_t s)
{
    if (num_jobs <= 0) return;
    hipLaunchKernelGGL(match_special_kernel, dim3(num_jobs), dim3(256), 0, s, d_problems, d_jobs, sp_parts, sp_col);
}

}  // namespace osfm
