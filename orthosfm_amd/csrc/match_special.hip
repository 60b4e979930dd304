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
//   row direction    -> (best, second, index) of every special descriptor: exact running
//             top-2 per (lane, register) on keys  ip << 8 | step, merged across lanes and
//             waves once per unit;
//   column direction -> for every streamed descriptor its exact (best, second, index) over
//             the special descriptors: top-2 of the lane's 16 scores on keys ip << 5 | row,
//             the two half-waves exchanged, units folded in order by the wave that owns the
//             columns (no atomics).
// match_finish_kernel merges both with the tile kernel's partials (match_kernels.hip).
#include <climits>

#include "match_kernels.h"

namespace osfm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ int med3s(int a, int b, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// (best, index, second) <- candidate (ip, idx) with its own second; larger index wins ties
__device__ __forceinline__ void fold_top2(int &bip, int &bidx, int &sec, int ip, int idx, int ip2)
{
    sec = max(max(sec, ip2), min(bip, ip));
    if (ip > bip || (ip == bip && idx > bidx)) { bip = ip; bidx = idx; }
}

}  // namespace

__global__ __launch_bounds__(256) void
match_special_kernel(const MatchProblem *__restrict__ problems, const SpecialJob *__restrict__ jobs,
    RowPart *__restrict__ sp)
{
    __shared__ int sk[4][16][64];
    __shared__ int ss[4][16][64];
    __shared__ RowPart wres[4][32];

    const SpecialJob job = jobs[blockIdx.x];
    const MatchProblem &pd = problems[job.problem];
    const int side = job.side;
    const int8_t *__restrict__ S = side == 0 ? pd.A_special : pd.B_special;
    const int32_t *__restrict__ corrS = side == 0 ? pd.corrA_special : pd.corrB_special;
    const int ns = side == 0 ? pd.nsA : pd.nsB;
    const int8_t *__restrict__ O = side == 0 ? pd.B : pd.A;
    const int32_t *__restrict__ corrO = side == 0 ? pd.corrB : pd.corrA;
    const int no = side == 0 ? pd.n2 : pd.n1;
    const int ns_pad = (ns + 31) & ~31;
    RowPart *rowres = sp + pd.sp_row_off[side] + (int64_t)job.chunk * ns_pad;
    RowPart *colres = sp + pd.sp_col_off[side];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int col0 = job.chunk * kSpChunk;
    const int nsteps = (min(no, col0 + kSpChunk) - col0 + 31) / 32;      // <= 128: 32 per wave, the key holds 8 bits

    for (int u = 0; u * 32 < ns; ++u) {
        // resident fragment of the unit's 32 rows and their corrections (C operand layout)
        v4i a[4];
        v16i ra;
        {
            const int8_t *srow = S + (size_t)(u * 32 + lr) * 128 + lh * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[ks] = *reinterpret_cast<const v4i *>(srow + ks * 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) ra[r] = corrS[u * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
        }
        v16i kbest, ksec;
#pragma unroll
        for (int r = 0; r < 16; ++r) { kbest[r] = kKeyNone; ksec[r] = kKeyNone; }

        // Three steps in flight per wave: a step's operands come straight from L2 / HBM (no
        // other wave shares them), so the loads of steps j+1 .. j+3 are issued before step j is
        // reduced -- with one step ahead the kernel was bound by that latency (1.05 ms per
        // 1225 pairs with one unit per view; the MFMA + reduction work is a quarter of it).
        struct Stage { v4i b[4]; int cb; RowPart old; };
        auto fetch = [&](Stage &st, int step) {
            const int col = col0 + min(step, nsteps - 1) * 32 + lr;     // below the bank's 256-row padding
            const int8_t *p = O + (size_t)col * 128 + lh * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) st.b[ks] = *reinterpret_cast<const v4i *>(p + ks * 32);
            st.cb = corrO[col];
            // the fold of the earlier units (same lane wrote it: program order)
            st.old.ip_best = INT_MIN; st.old.idx_best = -1; st.old.ip_second = INT_MIN; st.old.pad = 0;
            if (u > 0 && lh == 0) st.old = colres[col];
        };
        auto process = [&](const Stage &st, int step, int j) {
            const int cb = st.cb;
            const int col = col0 + step * 32 + lr;
            v16i acc = ra;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks], st.b[ks], acc, 0, 0, 0);

            // row direction: exact ip = acc + cb (acc carries the row correction)
            const unsigned cjt = ((unsigned)cb << 8) + (unsigned)j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = (int)(((unsigned)acc[r] << 8) + cjt);
                ksec[r] = med3s(kbest[r], ksec[r], key);
                kbest[r] = max(kbest[r], key);
            }
            // column direction: this lane's 16 rows of column `col`, cb added after the maximum
            int cbst = INT_MIN, csec = INT_MIN;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = (int)(((unsigned)acc[r] << 5) | (unsigned)r);
                csec = med3s(cbst, csec, k);
                cbst = max(cbst, k);
            }
            cbst |= lh << 4; csec |= lh << 4;
            const int o1 = __shfl_xor(cbst, 32), o2 = __shfl_xor(csec, 32);
            const int nb = max(cbst, o1), nsec = max(min(cbst, o1), max(csec, o2));
            if (lh == 0) {
                const int fr = nb & 31;
                const int slot = u * 32 + (fr & 3) + 8 * ((fr >> 2) & 3) + 4 * (fr >> 4);
                int bip = st.old.ip_best, bidx = st.old.idx_best, sec = st.old.ip_second;
                fold_top2(bip, bidx, sec, (nb >> 5) + cb, slot, (nsec >> 5) + cb);
                RowPart out;
                out.ip_best = bip; out.idx_best = bidx; out.ip_second = sec; out.pad = 0;
                colres[col] = out;          // idx_best: slot among the special descriptors (mapped by the finish kernel)
            }
        };
        Stage s0, s1, s2;
        int step = wave, j = 0;
        if (step < nsteps) { fetch(s0, step); fetch(s1, step + 4); fetch(s2, step + 8); }
        while (step < nsteps) {
            process(s0, step, j); fetch(s0, step + 12); step += 4; ++j;
            if (step >= nsteps) break;
            process(s1, step, j); fetch(s1, step + 12); step += 4; ++j;
            if (step >= nsteps) break;
            process(s2, step, j); fetch(s2, step + 12); step += 4; ++j;
        }

        // ---- row direction: merge the 32 lanes of a half-wave, then the four waves ----
#pragma unroll
        for (int r = 0; r < 16; ++r) { sk[wave][r][lane] = kbest[r]; ss[wave][r][lane] = ksec[r]; }
        __syncthreads();
        {
            const int r = lane & 15, h = (lane >> 4) & 1, part = lane >> 5;
            int bip = INT_MIN, bcol = -1, sec = INT_MIN;
            for (int i = 0; i < 16; ++i) {
                const int l = part * 16 + ((i + r) & 15);             // skewed: bank-conflict free
                const int kb = sk[wave][r][h * 32 + l], k2 = ss[wave][r][h * 32 + l];
                // SIFT inner products are >= 0: a negative key is "none" or a padding column
                if (kb >= 0)
                    fold_top2(bip, bcol, sec, kb >> 8, col0 + ((kb & 255) * 4 + wave) * 32 + l, k2 >= 0 ? (k2 >> 8) : INT_MIN);
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
            out.ip_best = bip; out.idx_best = max(bcol, 0); out.ip_second = sec; out.pad = 0;
            rowres[u * 32 + tid] = out;
        }
    }
}

void launch_match_special(const MatchProblem *d_problems, const SpecialJob *d_jobs, int num_jobs,
    RowPart *sp_parts, hipStream_t s)
{
    if (num_jobs <= 0) return;
    hipLaunchKernelGGL(match_special_kernel, dim3(num_jobs), dim3(256), 0, s, d_problems, d_jobs, sp_parts);
}

}  // namespace osfm
