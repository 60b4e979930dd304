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
    __shared__ __attribute__((aligned(16))) char bst[4][2][32 * 128];     // per wave: two staged steps of 32 descriptors

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
    // DMA source offsets of this lane: LDS chunk q = c * 64 + lane holds chunk (q % 8) ^ swz(d) of descriptor d = q / 8
    unsigned lane_off[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int q = c * 64 + lane, d = q >> 3;
        lane_off[c] = (unsigned)(d * 128 + (((q & 7) ^ ((d >> 1) & 7)) * 16));
    }
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
        // 1225 pairs with one unit per view).  The 32 streamed descriptors of a step come through LDS by DMA
        // (global_load_lds_dwordx4: 64 lanes x 16 B = eight whole cache lines per instruction), into a buffer that
        // belongs to this wave alone: loaded straight into MFMA fragments (lane = descriptor, 16 B of it) every
        // instruction touched 32 lines for a quarter of each, and the kernel was bound by the address unit of the
        // vector cache, not by the L2 (200 special rows per view: 4.5 ms per 1225 pairs whether a pass carried
        // two units or four).  The chunks of a descriptor are XOR-swizzled by the SOURCE address (the DMA writes
        // LDS linearly), so that the fragment reads are conflict-free, as in the tile kernel.
        struct Stage { int cb; int old; };
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void glb_void;
        auto fetch = [&](Stage &st, int step, int buf) {
            const int cbase = col0 + min(step, nsteps - 1) * 32;         // below the bank's 256-row padding
            const int8_t *src = O + (size_t)cbase * 128;
            char *dst = &bst[wave][buf][0];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                __builtin_amdgcn_global_load_lds((glb_void *)(src + lane_off[c]), (lds_void *)(uintptr_t)(dst + c * 1024), 16, 0, 0);
            // the loads below are issued behind the DMA: vector memory loads return in order, so the wait the
            // compiler puts in front of the first use of `cb` also covers the four DMA instructions
            asm volatile("" ::: "memory");
            st.cb = corrO[cbase + lr];
            // the fold of the earlier passes (same lane wrote it: program order)
            st.old = INT_MIN;
            if (many && pass > 0 && lh == 0) st.old = colres[0][cbase + lr];
        };
        auto process = [&](const Stage &st, int step, int buf) {
            int cb = st.cb;
            // `cb` has arrived, hence the DMA in front of it; nothing below may be moved above this point
            asm volatile("" : "+v"(cb) :: "memory");
            v4i b[4];
            {
                const int swz = (lr >> 1) & 7;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    b[ks] = *reinterpret_cast<const v4i *>(&bst[wave][buf][lr * 128 + (((ks * 2 + lh) ^ swz) * 16)]);
            }
            const int col = col0 + step * 32 + lr;
            int gall = INT_MIN;                        // folded across the slots of a many-unit entry (cb added at the end)
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                if (t >= nslots) break;
                v16i acc = ra[t];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t][ks], b[ks], acc, 0, 0, 0);
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
        if (step < nsteps) { fetch(s0, step, 0); fetch(s1, step + 4, 1); }
        while (step < nsteps) {
            process(s0, step, 0); fetch(s0, step + 8, 0); step += 4;
            if (step >= nsteps) break;
            process(s1, step, 1); fetch(s1, step + 8, 1); step += 4;
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

// Many special descriptors on one side (more than kSpSlots units, i.e. more than 64): the kernel above would
// stream the other view once per two units, and it is bound by exactly that stream (200 special rows per view:
// 3.7 ms per 1225 pairs for 25 GB out of the L2).  Here one workgroup takes a chunk of 1024 candidates, stages
// it through LDS ONCE (every wave fetches a quarter of each 32-descriptor step by DMA, three steps in flight)
// and all four waves read it, each for two units of its own: eight units per pass.  A lane's stream is then
// the 32 candidates of the chunk at a stride of 32 (RowPart.pad = 4 tells the finish kernel); the column
// maxima of the four waves meet in LDS and are folded by the wave whose turn it is.
__global__ __launch_bounds__(256, 2) void
match_special_wide_kernel(const MatchProblem *__restrict__ problems, const SpecialJob *__restrict__ jobs,
    RowPart *__restrict__ sp, int32_t *__restrict__ sp_col)
{
    constexpr int NS = 2;                                                  // units per wave
    __shared__ int sk[4][16][64];
    constexpr int NB = 6;                                                  // staged steps in the ring (NB - 1 in flight)
    __shared__ __attribute__((aligned(16))) char bst[NB][32 * 128];
    __shared__ int colmax[2][4][32];                                       // [step parity][wave][column of the step]
    __shared__ int cbs[kSpWideChunk];                                      // column corrections of the chunk
    __shared__ int colfin[kSpWideChunk];                                   // column results of the pass

    // jobs of one view are neighbours in the list: keep them on one XCD (one L2 fetches the view once)
    const SpecialJob job = jobs[xcd_remap(blockIdx.x, gridDim.x)];
    const MatchProblem &pd = problems[job.problem[0]];
    const int side = job.side[0];
    const int8_t *__restrict__ O = side == 0 ? pd.B : pd.A;
    const int32_t *__restrict__ corrO = side == 0 ? pd.corrB : pd.corrA;
    const int no = side == 0 ? pd.n2 : pd.n1;
    const int ns = side == 0 ? pd.nsA : pd.nsB;
    const int ns_pad = (ns + 31) & ~31;
    const int8_t *__restrict__ Sbase = side == 0 ? pd.A_special : pd.B_special;
    const int32_t *__restrict__ corrSbase = side == 0 ? pd.corrA_special : pd.corrB_special;
    int32_t *__restrict__ colres = sp_col + pd.sp_col_off[side];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int col0 = job.chunk * kSpWideChunk;
    const int nsteps = (min(no, col0 + kSpWideChunk) - col0 + 31) / 32;      // 1 .. 32
    const int nunits = ns_pad / 32;
    const int npass = (nunits + kSpWideUnits - 1) / kSpWideUnits;

    // DMA source offset of this lane for its wave's quarter of a step: LDS chunk q = wave * 64 + lane holds
    // chunk (q % 8) ^ swz(d) of descriptor d = q / 8 (the swizzle of the tile kernel: conflict-free fragment reads)
    unsigned lane_off;
    {
        const int q = wave * 64 + lane, d = q >> 3;
        lane_off = (unsigned)(d * 128 + (((q & 7) ^ ((d >> 1) & 7)) * 16));
    }
    // The DMA is inline assembly with m0 written behind the compiler's back, as in the tile kernel (hipcc
    // rejects m0 as a clobber): nothing else in this kernel uses m0 (no compiler-issued LDS DMA, and gfx950 LDS /
    // cross-lane instructions do not read it; tools/check_m0.sh looks at the ISA).  Through the builtin the
    // compiler sees an LDS write it cannot tell apart from the fragment reads and drains the whole vector
    // memory counter in front of every one of them -- no DMA would ever be in flight beside a step.
    auto stage = [&](int step) {
        const int cbase = col0 + min(step, nsteps - 1) * 32;                 // below the bank's 256-row padding
        const int8_t *src = O + (size_t)cbase * 128;                         // uniform
        const unsigned lds_at = (unsigned)(uintptr_t)(&bst[step % NB][0]) + (unsigned)wave_s * 1024u;
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(src), "s"(lds_at) : "memory");
    };

    // the chunk's column corrections, once (the step loop then issues no vector memory operation but its DMA)
    for (int i = tid; i < kSpWideChunk; i += 256) cbs[i] = corrO[col0 + min(i, nsteps * 32 - 1)];

    for (int pass = 0; pass < npass; ++pass) {
        // this wave's units (wave-uniform): dead ones compute nothing and write nothing
        int unit[NS];
        bool live[NS];
        v16i ra[NS];
        v4i a[NS][4];
        v16i kmax[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int u = pass * kSpWideUnits + wave * NS + t;
            live[t] = u < nunits;
            unit[t] = live[t] ? u : 0;
            const int32_t *corrS = corrSbase + unit[t] * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) ra[t][r] = corrS[(r & 3) + 8 * (r >> 2) + 4 * lh];
            const int8_t *srow = Sbase + (size_t)(unit[t] * 32 + lr) * 128 + lh * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[t][ks] = *reinterpret_cast<const v4i *>(srow + ks * 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) kmax[t][r] = INT_MIN;
        }
        const bool any_live = live[0];                                       // units of a wave are consecutive
        // every load above has arrived BEFORE the loop, in the compiler's books as well: left pending, it
        // waits for them inside the loop -- with vmcnt(0), i.e. for the DMA issued a moment ago
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            asm volatile("" : "+v"(ra[t]));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(a[t][ks]));
        }

        __syncthreads();                          // cbs is there; staging buffers, colmax and colfin of the pass before are done with
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // (nothing of the compiler's is in flight: the counts below are the DMAs alone)
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) stage(i);
        for (int step = 0; step < nsteps; ++step) {
            // This wave's quarter of step `step` has landed (those of the four steps after it may be in flight: vmcnt(4)) --
            // then, behind the barrier, everybody's has.  A bare s_barrier: __syncthreads() would drain the
            // vector memory counter.
            static_assert(NB == 6, "the wait below leaves NB - 2 DMAs in flight");
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            v4i b[4];
            {
                const int swz = (lr >> 1) & 7;
                const char *bt = &bst[step % NB][0];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    b[ks] = *reinterpret_cast<const v4i *>(bt + lr * 128 + (((ks * 2 + lh) ^ swz) * 16));
            }
            const int cb = cbs[step * 32 + lr];
            // step + NB - 1 goes where step - 1 was read (every wave is past that: it is past the barrier above)
            stage(step + NB - 1);
            // the column maxima of the step before: the four waves' values are in LDS since the barrier
            if (step > 0 && wave == ((step - 1) & 3) && lh == 0) {
                const int ps = step - 1;
                const int *cm = &colmax[ps & 1][0][0];
                const int g = max(max(cm[lr], cm[32 + lr]), max(cm[64 + lr], cm[96 + lr]));
                colfin[ps * 32 + lr] = g + cbs[ps * 32 + lr];
            }
            int gall = INT_MIN;
            if (any_live) {
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    if (!live[t]) break;
                    v16i acc = ra[t];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t][ks], b[ks], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) kmax[t][r] = max(kmax[t][r], acc[r] + cb);
                    int g = max(acc[0], acc[1]);
#pragma unroll
                    for (int r = 2; r < 16; r += 2) g = max(max(g, acc[r]), acc[r + 1]);
                    gall = max(gall, g);
                }
                gall = max(gall, __shfl_xor(gall, 32));
            }
            if (lh == 0) colmax[step & 1][wave][lr] = gall;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the DMAs issued past the end
        __syncthreads();
        {
            // the last step's column maxima, then the chunk's column results leave (folded with the pass before)
            const int ps = nsteps - 1;
            if (wave == (ps & 3) && lh == 0) {
                const int *cm = &colmax[ps & 1][0][0];
                const int g = max(max(cm[lr], cm[32 + lr]), max(cm[64 + lr], cm[96 + lr]));
                colfin[ps * 32 + lr] = g + cbs[ps * 32 + lr];
            }
            __syncthreads();
            for (int i = tid; i < nsteps * 32; i += 256) {
                int v = colfin[i];
                if (pass > 0) v = max(v, colres[col0 + i]);                  // same thread wrote it in the pass before
                colres[col0 + i] = v;
            }
        }

        // ---- row direction: the 32 lane streams of a half-wave (a wave owns its rows: no merge across waves) ----
#pragma unroll
        for (int t = 0; t < NS; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sk[wave][r][lane] = kmax[t][r];
            __syncthreads();
            {
                const int r = lane & 15, h = (lane >> 4) & 1, part = lane >> 5;
                int bip = INT_MIN, bcol = -1, sec = INT_MIN;
                for (int i = 0; i < 16; ++i) {
                    const int l = part * 16 + ((i + r) & 15);             // skewed: bank-conflict free
                    const int kb = sk[wave][r][h * 32 + l];
                    // SIFT inner products are >= 0: a negative maximum is padding columns only
                    if (kb >= 0) fold_top2(bip, bcol, sec, kb, col0 + l, INT_MIN);
                }
                const int obip = __shfl_xor(bip, 32), obcol = __shfl_xor(bcol, 32), osec = __shfl_xor(sec, 32);
                fold_top2(bip, bcol, sec, obip, obcol, osec);
                if (part == 0 && live[t]) {
                    RowPart out;
                    out.ip_best = bip; out.idx_best = max(bcol, 0); out.ip_second = sec; out.pad = 4;
                    sp[pd.sp_row_off[side] + (int64_t)job.chunk * ns_pad + unit[t] * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = out;
                }
            }
            __syncthreads();
        }
    }
}

void launch_match_special_wide(const MatchProblem *d_problems, const SpecialJob *d_jobs, int num_jobs,
    RowPart *sp_parts, int32_t *sp_col, hipStream_t s)
{
    if (num_jobs <= 0) return;
    hipLaunchKernelGGL(match_special_wide_kernel, dim3(num_jobs), dim3(256), 0, s, d_problems, d_jobs, sp_parts, sp_col);
}

void launch_match_special(const MatchProblem *d_problems, const SpecialJob *d_jobs, int num_jobs,
    RowPart *sp_parts, int32_t *sp_col, hipStream_t s)
{
    if (num_jobs <= 0) return;
    hipLaunchKernelGGL(match_special_kernel, dim3(num_jobs), dim3(256), 0, s, d_problems, d_jobs, sp_parts, sp_col);
}

}  // namespace osfm
