// Hand-written gfx950 kernels of the matching path (hot path A).
//
// What the reference does per view pair (src/mve/sfm/matching.h:114-159,
// nearest_neighbor.cc:60-129,214-268): for every descriptor of set 1 scan all
// descriptors of set 2, keep the largest / second largest integer inner
// product (ties: later index wins; state starts at 0,0), turn them into
// clamped squared distances, apply the Lowe ratio test; then the same with
// the roles swapped (recomputing every inner product), then a cross-check.
//
// What this file does instead: the N1 x N2 score matrix S = A * B^T is a
// dense int8 contraction (K = 128 or 64), so it is computed ONCE on the
// matrix cores (v_mfma_i32_32x32x32_i8, exact in int32) and both directions
// are reduced from the same accumulator tile while it is still in registers:
// row-wise top-2 (set 1 -> set 2) is carried in VGPRs across column tiles,
// column-wise top-2 (set 2 -> set 1) is reduced per tile and merged across
// the workgroup through LDS.  The score matrix never exists in memory.
//
// SIFT values 0..255 do not fit int8: they are stored as a' = a - 128 and the
// exact product is recovered as  sum(a*b) = sum(a'*b') + ra_i + cb_j  with
// per-descriptor corrections ra = 128*sum(a') + 2^20 (the two 2^20 make up
// 128*128*128).  ra enters through the MFMA C operand, cb through the key
// construction, so the correction costs no extra instruction per element.
#include <algorithm>

#include "match_kernels.h"

namespace osfm {


typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));


// ---------------------------------------------------------------------------
// Score-tile kernel.  CH = 16-byte chunks per descriptor (8: SIFT, 4: SURF).
// Workgroup = 256 threads = 4 waves; wave w owns rows [64w, 64w+64) of its
// row block as two 32-row MFMA fragments that stay in registers for the whole
// kernel; the workgroup walks the columns of its segment in 64-column tiles
// staged through LDS (double buffered, XOR-swizzled 16-B chunks so the
// ds_read_b128 fragment reads are bank-conflict free).
//
// Key formats (int32, signed compare):
//   row direction:  (ip << 8) | t        t = 32-column fragment index in segment
//   col direction:  (x  << 5) | fr       x = ip - cb_j, fr = 16*rf + reg  (per lane)
// Both preserve "later index wins on ties" because t / fr grow with the index.
// ---------------------------------------------------------------------------
// Three-input integer max / median as single VALU instructions.  Written as
// asm because hipcc otherwise CSEs max(b, x) between the max3 and med3
// patterns of the same operands and emits 4-5 two-input ops instead of 2.
// The operands are always results of compiler-visible VALU ops (the key
// constructions), never raw MFMA outputs, so no MFMA->VALU wait states are
// hidden from the hazard recogniser.
__device__ __forceinline__ int max3i(int a, int b, int c)
{
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int med3a(int a, int b, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Epilogue (per 32x32 accumulator fragment, everything stays in registers).
// The VALU, not the matrix pipe, bounds this kernel (a wave64 integer op
// occupies its SIMD for 4 cycles, an int8 32x32x32 MFMA for 32), so the
// running top-2 is organised to touch each score as little as possible:
//   * scores are first reduced to GROUP bests with v_max3 (0.5 op / score):
//     row direction  - a (lane, register) slot sees 2 columns per tile; a group
//                      is kGroupTiles tiles = 16 columns of that slot;
//     column direction - a lane sees 32 rows of a column per tile (both row
//                      fragments of its half-wave): that is the group;
//   * only group bests enter the exact (best, second) update (v_med3 + v_max).
// Hence best / index are exact, while "second" is the second largest GROUP
// best: the true second largest can only be larger if it sits in the same
// group as the best.  The finish kernel closes that gap exactly: a query that
// passes the ratio test against this lower bound gets its best group (16
// columns / 32 rows) re-scored and re-tested; one that fails it is rejected for
// good (the test is monotone in the second-best value).
//
// Software pipeline: the wave's 64 x 64 tile is produced and consumed in two
// PHASES of one 32-row fragment (x both 32-column fragments) each.  While the
// VALU reduces the accumulators of one phase, the MFMAs of the next phase are
// already in flight into the other accumulator pair: every epilogue chunk is
// preceded by exactly one MFMA (fenced with sched_barrier so the order
// survives the compiler), so matrix and vector pipes overlap inside one wave
// instead of relying on the other resident wave to fill the gaps.
//
// MASKED = false: rows >= n1 / columns >= n2 are PADDING descriptors whose
// stored bytes and corrections make their inner products come out at
// -2^22 (prepare_*_kernel), so they lose every comparison without a single
// masking instruction.  MASKED = true (low-res / num_features-limited
// matching, where the rows behind the limit are real descriptors): explicit
// per-element masks.
constexpr int kGroupTiles = 8;          // tiles per row-direction group (16 columns per slot), keyed kernels
constexpr int kPipeGroupTiles = 16;     // ... of the correction-free kernel's staggered schedule: every close
                                        // is 3 vector + 2 LDS operations in a loop where ~1 % of the time hangs
                                        // on each vector operation per tile; 8-tile groups were 12 per tile
constexpr int kPipeTileBits = 9;        // tile bits in the correction-free kernel's group keys (segments up to 512 tiles)
constexpr int kValNone = -(1 << 28);    // "no candidate" for un-keyed column scores

template <int V> struct IntC { static constexpr int value = V; };

// A column partial is written once and read once, by the finish kernel, after 5 GB of its kind: stored non-temporally
// it does not push the column bank of the pair out of the XCD's L2 on its way (the 64 workgroups an XCD runs side by
// side all stream the same 2.5 MB of descriptors).
__device__ __forceinline__ void store_colpart(ColPart *dst, const ColPart &v)
{
#ifdef OSFM_COLPART_PLAIN_STORE
    *dst = v;
#else
    __builtin_nontemporal_store(*reinterpret_cast<const unsigned long long *>(&v), reinterpret_cast<unsigned long long *>(dst));
#endif
}

// RAW = true: row operand in raw form (see MatchProblem): the accumulator IS
// the inner product, no key is built per score in either direction (group keys
// only), and the best column is recovered by the group rescan of the finish
// kernel.  RAW = false: value-128 row operand with per-score keys
// (ip << 8 | tile) carrying the column correction.
// C0 (RAW only): both operands in raw form, no correction anywhere, C = 0.
template <int CH, bool MASKED, bool RAW, bool C0>
__device__ __forceinline__ void
tile_body(const MatchProblem &pd, int rb, int seg, RowPart *__restrict__ rowparts,
    ColPart *__restrict__ colparts, char *smem, const int8_t *__restrict__ zero_tile)
{
    static_assert(!C0 || RAW, "the correction-free form needs the raw row operand");
    constexpr int D = CH * 16;
    constexpr int KS = CH / 2;            // MFMA k-steps (K = 32 bytes each)
    constexpr int RPB = 16 / CH;          // descriptor rows per 256-B LDS bank row
    constexpr int TILE_BYTES = kTileCols * D;
    constexpr int CHUNKS = kTileCols * CH;        // 16-B chunks per tile
    constexpr int CPT = CHUNKS / 256;             // chunks per thread (2 or 1)
    constexpr int BBUF_BYTES = 2 * TILE_BYTES > 16384 ? 2 * TILE_BYTES : 16384;
    // RAW group keys: running best << 7 | first tile of the group inside the segment
    // (|ip| < 2^22 for every operand form that takes this path, so the shift is safe)
    // The correction-free form has products below 2^21 (127^2 * 128): nine tile bits fit beside them, so one
    // segment may span 512 tiles (the host chooses seg_cols); the forms with a correction reach 2^23 and keep seven.
    constexpr int kRawShift = C0 ? kPipeTileBits : 7;
    constexpr int kCurNone = RAW ? -(1 << (30 - kRawShift)) : kKeyNone;   // (kCurNone << kRawShift) == kKeyNone
    static_assert(kSegCols / kTileCols <= (1 << 7), "tile index must fit the key");
    // The correction-free kernel folds the group closes and the column merges into
    // the MFMA-paced phases (measured: a vector op between the phases costs about
    // twice what it costs inside one); the others keep them between phases.
    constexpr bool PIPE = C0;
    constexpr int NCHUNK = 2 * KS;        // MFMAs (= epilogue chunks) per phase
    constexpr int RPC = 16 / NCHUNK;      // row-direction registers per chunk
    constexpr int CPC = 16 / KS;          // column-direction registers per chunk

    char *bbuf = smem;                                              // [2][TILE_BYTES] (>= 16 KB)
    int *corrbuf = reinterpret_cast<int *>(smem + BBUF_BYTES);      // [2][64]
    ColPart *colbuf = reinterpret_cast<ColPart *>(smem + BBUF_BYTES + 2 * 64 * 4);  // [8 tiles][4 waves][64]
    int *rsecbuf = reinterpret_cast<int *>(smem + BBUF_BYTES + 2 * 64 * 4 + 8 * 4 * 64 * 8);  // [32][256]

    // problem fields used inside the tile loop, read once (the loop's LDS
    // traffic would otherwise force scalar re-loads and lgkmcnt(0) waits)
    const int n1 = pd.n1, n2 = pd.n2, n2stride = pd.n2stride;
    const int8_t *const Bbase = C0 ? pd.B_raw : pd.B;
    const int32_t *const corrBbase = pd.corrB;
    ColPart *const colout = colparts + pd.colpart_off + (int64_t)rb * pd.n2stride;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int row0 = rb * kRowsPerBlock + wave * 64;      // row slot (partials, validity)
    const int gcode = wave * 2 + lh;                      // column-direction group of this lane

    // --- resident A fragments and row corrections -------------------------
    const int8_t *Abase = pd.A;
    const int32_t *corrA = pd.corrA;
    int arow0 = row0;
    if (RAW) { Abase = pd.A_raw; corrA = pd.corrA_raw; }
    else if (!MASKED && rb >= pd.nrb_main) {
        Abase = pd.A_special; corrA = pd.corrA_special;
        arow0 = (rb - pd.nrb_main) * kRowsPerBlock + wave * 64;
    }
    v4i a[2][KS];
    v16i ra[2];
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) {
        const int8_t *arow = Abase + (size_t)(arow0 + rf * 32 + lr) * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            a[rf][ks] = *reinterpret_cast<const v4i *>(arow + (ks * 2 + lh) * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            ra[rf][r] = C0 ? 0 : corrA[arow0 + rf * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
    }
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned row_valid_bits = 0xffffffffu;
    if (MASKED) {
        row_valid_bits = 0;
#pragma unroll
        for (int fr = 0; fr < 32; ++fr) {
            const int row = row0 + (fr >> 4) * 32 + (fr & 3) + 8 * ((fr >> 2) & 3) + 4 * lh;
            row_valid_bits |= (row < n1 ? 1u : 0u) << fr;
        }
    }

    // row direction state: best key and the running best of the current group
    // in registers, the second-best group key in LDS (touched once per group)
    v16i rbest[2], rcur[2];
#pragma unroll
    for (int rf = 0; rf < 2; ++rf)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            rbest[rf][r] = kKeyNone; rcur[rf][r] = kCurNone;
            rsecbuf[(rf * 16 + r) * 256 + tid] = kKeyNone;
        }

    const int seg_cols = pd.seg_cols;
    const int col_begin = seg * seg_cols;
    const int col_lim = min(n2, col_begin + seg_cols);
    const int ntiles = (col_lim - col_begin + kTileCols - 1) / kTileCols;

    // --- B tile staging: LDS-DMA (global_load_lds, 16 B per lane) ------------
    // The DMA writes LDS linearly (wave base + lane*16), so the XOR swizzle is
    // applied to the per-lane SOURCE address: LDS chunk q holds chunk
    // (q % CH) ^ swz(col) of column col = q / CH.
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    // per-lane byte offset inside a tile (loop invariant, 32 bits) and the wave id
    // as a scalar: the DMA then addresses with a scalar base + vector offset and a
    // scalar LDS destination, no vector arithmetic per tile
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    unsigned lane_off[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int q = (c * 4 + wave) * 64 + lane;
        const int col = q / CH;
        const int ch = (q % CH) ^ ((col / RPB) % CH);
        lane_off[c] = (unsigned)(col * D + ch * 16);
    }
    // (vector-memory operations complete in issue order: behind a tile whose phase stored a merged column partial --
    //  issued after the tile's two DMA pieces -- waiting for all but the youngest is waiting for the pieces; the
    //  store's acknowledgement is not something the next tile needs)
    auto dma_wait = [&](bool store_behind = false) {
        if (!PIPE) return;
#ifndef OSFM_TILE_WAIT_ALL
        if (store_behind) { asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); return; }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto stage_tile = [&](int t, int buf) {
        const int8_t *src = Bbase + (size_t)(col_begin + t * kTileCols) * D;      // uniform
        // PIPE: the tile loop runs whole cycles; the tiles behind the segment's last one are blank
        // (zero descriptors score 0 against everything, which changes nothing: the reference's
        // running state starts at (0, 0) and the correction-free products are >= 0)
        if (PIPE && t >= ntiles) src = zero_tile;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int q0 = (c * 4 + wave_s) * 64;         // first chunk of this wave-instruction
            if (PIPE) {
                // scalar base + 32-bit lane offset: through the builtin the compiler adds the base
                // to a 64-bit lane pointer with a vector instruction per DMA (two per tile in a loop
                // where every vector operation per tile is ~1 % of the time).  The compiler does not
                // see this memory operation: dma_wait() below stands where its s_waitcnt would.
                // m0 is written here behind the compiler's back (hipcc rejects it as a clobber: a
                // reserved register).  That is safe only because NOTHING else in this instantiation
                // uses m0: PIPE implies RAW, so the builtin DMA of the column corrections below does
                // not exist in it (static_assert), and gfx950 LDS / readlane instructions do not read
                // m0.  tools/check_m0.sh greps the ISA of the PIPE kernels for any other m0 use.
                static_assert(!PIPE || RAW, "the PIPE schedule must not contain compiler-managed m0 users");
                const unsigned lds_at = (unsigned)(uintptr_t)(bbuf + buf * TILE_BYTES + q0 * 16);
                asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(lane_off[c]), "s"(src), "s"(lds_at) : "memory");
            } else {
                __builtin_amdgcn_global_load_lds(
                    (glb_void *)(src + lane_off[c]),
                    (lds_void *)(uintptr_t)(bbuf + buf * TILE_BYTES + q0 * 16), 16, 0, 0);
            }
        }
        if (!RAW && wave_s == 0)
            __builtin_amdgcn_global_load_lds(
                (glb_void *)(corrBbase + col_begin + t * kTileCols + lane),
                (lds_void *)(uintptr_t)(corrbuf + buf * 64), 4, 0, 0);
    };

    // B fragments (and column corrections) of one 32-column group, LDS -> registers
    v4i b[2][KS];
    int cb_next[2] = {0, 0};
    auto load_b = [&](int buf, int cf) {
        const int col = cf * 32 + lr;
        const int swz = (col / RPB) % CH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            b[cf][ks] = *reinterpret_cast<const v4i *>(
                bbuf + buf * TILE_BYTES + col * D + (((ks * 2 + lh) ^ swz) * 16));
        if (!RAW) cb_next[cf] = corrbuf[buf * 64 + col];
    };

    // per-tile epilogue inputs (set at the top of each iteration)
    int cbj[2] = {0, 0}, cjt[2] = {0, 0};
    bool col_valid[2] = {true, true};
    int g[2] = {kValNone, kValNone};      // column direction: best of this lane's 32 rows

    v16i acc0[2], acc1[2];                // accumulators of row fragment 0 / 1 (x 2 column groups)

    // One phase: reduce the finished accumulators `cur` (row fragment PH) while
    // the MFMAs of the other fragment are issued into `nxt`, one per chunk.
    // Extras a phase can carry (PIPE): CL >= 0 -- close the group of slot CL
    // of the OTHER row fragment (its running bests are not touched in this phase)
    // with first tile `tile0`; MG -- merge the four waves' column partials of tile tm.
    // Their LDS operands are fetched in front of the first chunk and used from the
    // third on; everything is straight-line code (a branch would split the block).
    // RS >= 0 -- slot RS of THIS fragment was closed by the phase before:
    // its running best restarts from this tile's scores (no reset needed there).
    auto phase = [&](auto ph_c, auto cl_c, auto mg_c, auto rs_c, v16i (&cur)[2], v16i (&nxt)[2], int buf_next,
                     int tile0, int tm) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr int CL = decltype(cl_c)::value;
        constexpr int RS = decltype(rs_c)::value;
        constexpr bool MG = decltype(mg_c)::value != 0;
        constexpr int OF = PH ^ 1;
        int sv = 0;
        ColPart cpm[4];
        if (CL >= 0) sv = rsecbuf[(OF * 16 + CL) * 256 + tid];
        if (MG) {
#pragma unroll
            for (int w = 0; w < 4; ++w) cpm[w] = colbuf[((tm & 7) * 4 + w) * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < NCHUNK; ++i) {
            const int cf = i / KS, ks = i % KS;
            if (ks == 0)
                nxt[cf] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[PH ^ 1][0], b[cf][0], C0 ? zero16 : ra[PH ^ 1], 0, 0, 0);
            else
                nxt[cf] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[PH ^ 1][ks], b[cf][ks], nxt[cf], 0, 0, 0);
            // The B registers of a column group are free once phase 0 (second half
            // of tile t) has issued that group's chain; they take the fragments of
            // tile t+1.  Group 0 is fetched right there, group 1 behind the first
            // MFMA of phase 1, so that the LDS wait in front of phase 1 (the compiler
            // drains the whole counter) never covers reads issued a moment ago.
            if (PH == 0 && i == KS - 1) load_b(buf_next, 0);
            if (PH == 1 && i == 0) {
                __builtin_amdgcn_sched_barrier(0);      // keep the reads behind the MFMA
                load_b(buf_next, 1);
            }

            // row direction: the two column groups feed the same (lane, reg) slot
#pragma unroll
            for (int r = i * RPC; r < (i + 1) * RPC; ++r) {
                if (RAW) {
                    if (RS >= 0 && r == RS) rcur[PH][r] = max(cur[0][r], cur[1][r]);
                    else rcur[PH][r] = max(max(rcur[PH][r], cur[0][r]), cur[1][r]);
                } else {
                    int k0 = (int)(((unsigned)cur[0][r] << 8) + (unsigned)cjt[0]);
                    int k1 = (int)(((unsigned)cur[1][r] << 8) + (unsigned)cjt[1]);
                    if (MASKED) {
                        k0 = col_valid[0] ? k0 : kKeyNone;
                        k1 = col_valid[1] ? k1 : kKeyNone;
                    }
                    rcur[PH][r] = max3i(rcur[PH][r], k0, k1);
                }
            }
            // column direction: un-keyed scores of column group cf, CPC registers
            {
                int x[CPC];
#pragma unroll
                for (int j = 0; j < CPC; ++j) {
                    x[j] = cur[cf][ks * CPC + j];
                    if (MASKED) x[j] = ((row_valid_bits >> (PH * 16 + ks * CPC + j)) & 1u) ? x[j] : kValNone;
                }
                int gg;
                if (PH == 0 && ks == 0) gg = max(x[0], x[1]);
                else gg = max(max(g[cf], x[0]), x[1]);
#pragma unroll
                for (int j = 2; j < CPC; j += 2) gg = max(max(gg, x[j]), x[j + 1]);
                // pinned here: the optimiser would otherwise sink the whole chain to its
                // only use after the loop body, keeping both accumulator pairs alive
                asm volatile("" : "+v"(gg));
                g[cf] = gg;
            }
            if (CL >= 0 && i == 2) {
                const int r = CL;
                const int gk = (int)(((unsigned)rcur[OF][r] << kRawShift) | (unsigned)tile0);
                rsecbuf[(OF * 16 + r) * 256 + tid] = med3a(rbest[OF][r], sv, gk);
                rbest[OF][r] = max(rbest[OF][r], gk);
                // no reset: the next phase that touches this slot restarts it (RS)
            }
            if (MG && (i == 2 || i == 3)) {
                // chunk 2: fold the four partials; chunk 3: store (no branch: every tile
                // merged inside the loop exists, its 64 columns lie below n2stride)
                if (i == 2) {
                    int k1 = cpm[0].key_best, k2 = cpm[0].key_second;      // every key is >= kKeyNone
#pragma unroll
                    for (int w = 1; w < 4; ++w) {
                        k2 = max(max(min(k1, cpm[w].key_best), k2), cpm[w].key_second);
                        k1 = max(k1, cpm[w].key_best);
                    }
                    cpm[0].key_best = k1; cpm[0].key_second = k2;
                } else {
                    store_colpart(colout + col_begin + tm * kTileCols + lane, cpm[0]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // --- prologue: tiles 0 and 1 into LDS, fragments of tile 0, first half of tile 0
    stage_tile(0, 0);
    stage_tile(PIPE || ntiles > 1 ? 1 : 0, 1);
    dma_wait();
    __syncthreads();
    load_b(0, 0);
    load_b(0, 1);
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
        acc0[cf] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0][0], b[cf][0], C0 ? zero16 : ra[0], 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < KS; ++ks)
            acc0[cf] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0][ks], b[cf][ks], acc0[cf], 0, 0, 0);
    }
    __syncthreads();        // every wave holds tile 0 in registers: its LDS buffer may be refilled

    // per-tile steps shared by the two loop forms below
    auto tile_top = [&](int t) {
        // tile t+2 into the buffer tile t was read from (clamped: the extra
        // refills of the last tile are never consumed)
        stage_tile(PIPE ? t + 2 : min(t + 2, ntiles - 1), t & 1);
#pragma unroll
        for (int cf = 0; cf < 2; ++cf) {
            if (!RAW) {
                cbj[cf] = cb_next[cf];
                cjt[cf] = (int)(((unsigned)cbj[cf] << 8) + (unsigned)(t * 2 + cf));
                // opaque to the optimiser: otherwise it re-associates the key into
                // ((acc + cb) << 8) + t, two ops per element instead of one v_lshl_add_u32
                asm volatile("" : "+v"(cjt[cf]));
            }
            col_valid[cf] = (col_begin + t * kTileCols + cf * 32 + lr) < n2;
        }
    };
    // column direction: this lane's group best as (ip << 8 | group code); the two
    // half-waves of a column are exchanged with one v_permlane32_swap so that
    // lane == column within the tile, then the wave's (best, second) goes to LDS.
    // group code = wave * 2 + half-wave (32 rows: both fragments of the half-wave)
    auto tile_bottom = [&](int t, bool store_behind = false) {
        int kk[2];
#pragma unroll
        for (int cf = 0; cf < 2; ++cf) {
            kk[cf] = (int)((unsigned)(g[cf] + cbj[cf]) << 8) | gcode;
            if (MASKED) kk[cf] = g[cf] < -(1 << 27) ? kKeyNone : kk[cf];   // no valid row in this lane
        }
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)kk[0], (unsigned)kk[1], false, false);
        const int x0 = (int)sw[0], x1 = (int)sw[1];
        ColPart cp;
        cp.key_best = max(x0, x1);
        cp.key_second = min(x0, x1);
        colbuf[((t & 7) * 4 + wave) * 64 + lane] = cp;
        dma_wait(store_behind);
        __syncthreads();
    };
    // close the open row-direction group of every slot: fold the group bests into
    // (best, second); first_tile(rf, r) = first tile of that slot's open group.
    // All LDS reads first (one exposed latency instead of 32).
    auto close_all = [&](auto first_tile) {
        int sv[2][16];
#pragma unroll
        for (int rf = 0; rf < 2; ++rf)
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[rf][r] = rsecbuf[(rf * 16 + r) * 256 + tid];
#pragma unroll
        for (int rf = 0; rf < 2; ++rf)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gk = RAW ? (int)(((unsigned)rcur[rf][r] << kRawShift) | (unsigned)first_tile(rf, r)) : rcur[rf][r];
                rsecbuf[(rf * 16 + r) * 256 + tid] = med3a(rbest[rf][r], sv[rf][r], gk);
                rbest[rf][r] = max(rbest[rf][r], gk);
                rcur[rf][r] = kCurNone;
            }
    };
    // this wave's share of a batch of four tiles: merge the four waves' partials of tile tm
    auto merge_tile = [&](int tm) {
        ColPart cp[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) cp[w] = colbuf[((tm & 7) * 4 + w) * 64 + lane];
        int k1 = cp[0].key_best, k2 = cp[0].key_second;                          // every key is >= kKeyNone
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            k2 = max(max(min(k1, cp[w].key_best), k2), cp[w].key_second);
            k1 = max(k1, cp[w].key_best);
        }
        const int col = col_begin + tm * kTileCols + lane;
        if (col < n2stride) {
            ColPart out;
            out.key_best = k1;
            out.key_second = k2;
            store_colpart(colout + col, out);
        }
    };

    int t = 0;
    if (PIPE) {
        // Cycles of sixteen tiles, fully unrolled: every phase carries a fixed share of
        // the work that would otherwise sit between phases, so nothing has to be
        // chosen at run time (a run-time choice between phase bodies costs a
        // register copy of the accumulators at the join).
        //   phase (u, 0) closes slot u of fragment 1 -- its group is the sixteen tiles
        //     t-16 .. t-1;  phase (u, 1) closes slot u of fragment 0 -- tiles t-15 .. t.
        //     Group boundaries are staggered by slot; the key carries each group's first
        //     tile, which is all the finish kernel needs.
        //   phases (1, 0), (5, 0), (9, 0), (13, 0) merge the column partials of the four
        //     tiles before t-1 (for the very first cycle the result of (1, 0) is a
        //     throw-away that (5, 0) overwrites: same wave, same addresses, program order).
        // In the first cycle the closes of not yet started groups fold "nothing"
        // into (best, second): a no-op.
        static_assert(kPipeGroupTiles == 16, "the staggered schedule closes one of 16 slots per phase");
        auto cycle_tile = [&](auto u_c, int t0) {
            constexpr int u = decltype(u_c)::value;
            const int tt = t0 + u;
            tile_top(tt);
            const int bn = (tt & 1) ^ 1;
            const int tm = max(tt - 5, 0) + wave;
            // restarts: the fragment 0 slot closed by phase (u-1, 1), the fragment 1 slot by phase (u, 0)
            phase(IntC<0>(), IntC<u>(), IntC<(u & 3) == 1 ? 1 : 0>(), IntC<(u + 15) & 15>(), acc0, acc1, bn,
                max(tt - 16, 0), tm);
            phase(IntC<1>(), IntC<u>(), IntC<0>(), IntC<u>(), acc1, acc0, bn, max(tt - 15, 0), 0);
            tile_bottom(tt, (u & 3) == 1);       // phase (u, 0) of these tiles stores a merged partial
        };
        // (whole cycles only: blank tiles fill the last one, see stage_tile -- a tail of single tiles costs 2.5x
        //  the vector instructions per tile and a close of all slots)
        const int tcyc = (ntiles + 15) & ~15;
        for (; t < tcyc; t += 16) {
            cycle_tile(IntC<0>(), t); cycle_tile(IntC<1>(), t); cycle_tile(IntC<2>(), t); cycle_tile(IntC<3>(), t);
            cycle_tile(IntC<4>(), t); cycle_tile(IntC<5>(), t); cycle_tile(IntC<6>(), t); cycle_tile(IntC<7>(), t);
            cycle_tile(IntC<8>(), t); cycle_tile(IntC<9>(), t); cycle_tile(IntC<10>(), t); cycle_tile(IntC<11>(), t);
            cycle_tile(IntC<12>(), t); cycle_tile(IntC<13>(), t); cycle_tile(IntC<14>(), t); cycle_tile(IntC<15>(), t);
        }
        merge_tile(t - 4 + wave);                // the batch of the last four tiles of the last cycle
        rcur[0][15] = kCurNone;                  // closed by the last phase, never restarted
        // the open groups of all slots (staggered starts), before the remaining tiles
        // start one common group
        const int tc = t;
        close_all([&](int rf, int r) { return max(tc - (rf ? 16 : 15) + r, 0); });
    }
    // all tiles of the kernels without the staggered schedule
    const int t_tail = t;
    if (!PIPE)
    for (; t < ntiles; ++t) {
        tile_top(t);
        phase(IntC<0>(), IntC<-1>(), IntC<0>(), IntC<-1>(), acc0, acc1, (t & 1) ^ 1, 0, 0);   // reduce (rf 0, t), produce (rf 1, t), fetch B(t+1)
        phase(IntC<1>(), IntC<-1>(), IntC<0>(), IntC<-1>(), acc1, acc0, (t & 1) ^ 1, 0, 0);   // reduce (rf 1, t), produce (rf 0, t+1)
        if (PIPE ? t == ntiles - 1 : ((t % kGroupTiles) == kGroupTiles - 1 || t == ntiles - 1)) {
            const int first = PIPE ? t_tail : (t / kGroupTiles) * kGroupTiles;
            close_all([&](int, int) { return first; });
        }
        tile_bottom(t);
        // every fourth tile all four waves merge one tile each of the last batch
        // (the same work in every wave: nobody is waited for at the next barrier)
        if ((t & 3) == 3 || t == ntiles - 1) {
            const int tm = (t & ~3) + wave;
            if (tm <= t) merge_tile(tm);
        }
    }

    // --- row direction: merge the 32 lanes that hold the same row -------------
    // Transposed through LDS (tile buffers and column ring are free now, 32 KB with the ring's neighbours):
    // every lane takes ONE of the wave's 64 rows and reads the 32 (best key, second key) entries its
    // source lanes hold for it, 8 + 8 ds_read_b128.  Lane L <-> (fragment L >> 5, half-wave (L >> 4) & 1,
    // register L & 15); the 16-byte pieces are read rotated by the register number, which keeps every
    // 16-lane service group of ds_read_b128 on distinct banks.  Key order is (inner product, tile) and
    // the column is tile-major, so among equal keys the larger source lane is the later column.
    // RAW keys carry the tile group only: the column reported is the first one of the winning
    // (lane, group) stream.
    RowPart *rp = rowparts + pd.rowpart_off + (int64_t)seg * ((int64_t)pd.nrb * kRowsPerBlock);
    int *bb = reinterpret_cast<int *>(smem);                        // [32][256], over bbuf / corrbuf / colbuf
    static_assert(BBUF_BYTES + 2 * 64 * 4 + 8 * 4 * 64 * 8 >= 32 * 256 * 4, "the best keys of both fragments must fit in front of rsecbuf");
    __syncthreads();
#pragma unroll
    for (int rf = 0; rf < 2; ++rf)
#pragma unroll
        for (int r = 0; r < 16; ++r) bb[(rf * 16 + r) * 256 + tid] = rbest[rf][r];
    __syncthreads();
    {
        constexpr int SH = RAW ? kRawShift : 8;
        const int r = lane & 15, h = (lane >> 4) & 1, rf = lane >> 5;
        const int base = (rf * 16 + r) * 256 + wave * 64 + h * 32;
        v4i kb[8], ks[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int jj = (j + r) & 7;
            kb[j] = *reinterpret_cast<const v4i *>(bb + base + jj * 4);
            ks[j] = *reinterpret_cast<const v4i *>(rsecbuf + base + jj * 4);
        }
        // (best, second) over the 32 best keys -- med3 keeps a duplicate of the best as second --
        // and the largest of the 32 second keys
        int b1 = kKeyNone, b2 = kKeyNone, smax = kKeyNone;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                b2 = med3a(b1, b2, kb[j][e]);
                b1 = max(b1, kb[j][e]);
            }
#pragma unroll
        for (int j = 0; j < 8; j += 2)
#pragma unroll
            for (int e = 0; e < 4; ++e) smax = max3i(smax, ks[j][e], ks[j + 1][e]);
        // the source lane of the best key, the largest one among equals: a bit per slot (slot 4j + e
        // holds source lane (4j + e + 4r) & 31), rotated into lane order, highest set bit
        unsigned hit = 0;
#pragma unroll
        for (int j = 7; j >= 0; --j)
#pragma unroll
            for (int e = 3; e >= 0; --e) hit = (hit << 1) | (kb[j][e] == b1 ? 1u : 0u);
        const unsigned rot = __builtin_amdgcn_alignbit(hit, hit, (32 - 4 * r) & 31);      // rotate left by 4r
        const int bl = 31 - __builtin_clz(rot | 1u);
        const int sk = max(b2, smax);
        const int row = row0 + rf * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        RowPart out;
        out.ip_best = b1 == kKeyNone ? INT_MIN : (b1 >> SH);
        out.ip_second = sk == kKeyNone ? INT_MIN : (sk >> SH);
        // pad = 1: idx_best is the first column of the winning (lane, group) stream,
        // the group being the tiles from there; 0: the exact best column
        out.idx_best = b1 == kKeyNone ? 0
                     : RAW ? col_begin + (b1 & ((1 << kRawShift) - 1)) * kTileCols + bl
                           : col_begin + (b1 & 255) * 32 + bl;
        out.pad = RAW ? 1 : 0;
        rp[row] = out;
    }
}

// RAW = true runs the row blocks [0, nrb_main) of every problem, RAW = false the
// blocks of gathered special rows behind them (and everything of a MASKED
// launch); C0 selects the problems whose column operand is correction-free.  A
// block of another kind returns at once.  Separate kernels rather than one
// with all bodies: the register budget of each is its own.
template <int CH, bool MASKED, bool RAW, bool C0>
__global__ __launch_bounds__(256, 2) void
match_tile_kernel(const MatchProblem *__restrict__ problems, int num_problems, int total_blocks,
    RowPart *__restrict__ rowparts, ColPart *__restrict__ colparts, unsigned long long *__restrict__ clock_probe,
    const int8_t *__restrict__ zero_tile)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lin = xcd_remap(blockIdx.x, total_blocks);
    // locate the problem: largest p with block_start <= lin
    int lo = 0, hi = num_problems - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (problems[mid].block_start <= lin) lo = mid; else hi = mid - 1;
    }
    const MatchProblem &pd = problems[lo];
    const int local = lin - pd.block_start;
    const int rb = local / pd.nseg;
    const int seg = local - rb * pd.nseg;
    if (!MASKED && (rb < pd.nrb_main) != RAW) return;
    if (RAW && (pd.c0 != 0) != C0) return;
    // Every 1024th workgroup of the correction-free kernel reports how many shader cycles (s_memtime)
    // and how much wall time (the 100 MHz counter) its sweep took: their ratio is the clock the chip
    // actually held under this kernel, which is what the matrix pipes' peak scales with (bench.py:
    // roofline.frac_at_held_clock).  Both counters are scalar reads: no vector register is spent.
    if (C0 && clock_probe != nullptr && (blockIdx.x & 1023) == 0) {
        const unsigned long long c0 = (unsigned long long)clock64(), w0 = (unsigned long long)wall_clock64();
        tile_body<CH, MASKED, RAW, C0>(pd, rb, seg, rowparts, colparts, smem, zero_tile);
        const unsigned long long c1 = (unsigned long long)clock64(), w1 = (unsigned long long)wall_clock64();
        if (threadIdx.x == 0) { atomicAdd(clock_probe, c1 - c0); atomicAdd(clock_probe + 1, w1 - w0); }
        return;
    }
    tile_body<CH, MASKED, RAW, C0>(pd, rb, seg, rowparts, colparts, smem, zero_tile);
}

void launch_match_tiles(int ch, bool masked, bool any_special, bool any_c0, bool any_corrected,
    const MatchProblem *d_problems, int num_problems, int total_blocks, RowPart *rowparts,
    ColPart *colparts, hipStream_t s, unsigned long long *clock_probe, const int8_t *zero_tile)
{
    if (total_blocks <= 0) return;
    const int d = ch * 16;
    const size_t lds = std::max<size_t>(2 * (size_t)kTileCols * d, 16384) + 2 * 64 * 4 + 8 * 4 * 64 * sizeof(ColPart) + 32 * 256 * 4;
    const dim3 grid(total_blocks), block(256);
#define OSFM_LAUNCH_TILES(CHV, MASKEDV, RAWV, C0V) \
    for (unsigned long long *PROBE = ((C0V) ? clock_probe : nullptr), *once_ = (unsigned long long *)1; once_; once_ = nullptr) \
    hipLaunchKernelGGL((match_tile_kernel<CHV, MASKEDV, RAWV, C0V>), grid, block, lds, s, d_problems, num_problems, total_blocks, rowparts, colparts, PROBE, zero_tile)
    if (masked) {
        if (ch == 8) OSFM_LAUNCH_TILES(8, true, false, false); else OSFM_LAUNCH_TILES(4, true, false, false);
        return;
    }
    if (any_c0) { if (ch == 8) OSFM_LAUNCH_TILES(8, false, true, true); else OSFM_LAUNCH_TILES(4, false, true, true); }
    if (any_corrected) { if (ch == 8) OSFM_LAUNCH_TILES(8, false, true, false); else OSFM_LAUNCH_TILES(4, false, true, false); }
    if (any_special) { if (ch == 8) OSFM_LAUNCH_TILES(8, false, false, false); else OSFM_LAUNCH_TILES(4, false, false, false); }
#undef OSFM_LAUNCH_TILES
}

// ---------------------------------------------------------------------------
// Finish: merge partials, apply the reference's clamps and ratio test.
//   u16 (nearest_neighbor.cc:262-267): b = min(65025, ip); d = min(32767, 65025 - b) * 2
//   s16 (nearest_neighbor.cc:234-237): b = clamp(ip, 0, 16129); d = 32258 - 2 b
//   accept unless d1 > dist_thres^2 or float(d1)/float(d2) > lowe^2
//   (matching.h:138-144) -- evaluated through a host-built integer table that
//   encodes exactly those float operations for every (d1, d2).
// The reference's state starts at (0, 0, idx 0): best = max(0, ip1),
// second = max(0, ip2), index = idx1 if ip1 >= 0 else 0.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int
accept_match(int ip1, int ip2, int idx1, const LoweTable &tab)
{
    int b = max(ip1, 0), s = max(ip2, 0);
    const int idx = ip1 >= 0 ? idx1 : 0;
    int d1, d2;
    if (tab.is_signed) {
        b = min(b, 16129); s = min(s, 16129);
        d1 = 32258 - 2 * b; d2 = 32258 - 2 * s;
    } else {
        b = min(b, 65025); s = min(s, 65025);
        d1 = min(32767, 65025 - b) * 2; d2 = min(32767, 65025 - s) * 2;
    }
    if (d1 > tab.max_d1) return -1;
    if (d1 >= tab.reject_from[d2 >> 1]) return -1;
    return idx;
}

// DPP helpers: data movement inside the wave without LDS traffic.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_mov(int v, int old)
{
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // lane i <-> 7 - i inside 8 lanes
constexpr int kDppMirror = 0x140;      // lane i <-> 15 - i inside a row of 16
constexpr int kDppBcast15 = 0x142;     // lane 15 of a row -> the next row
constexpr int kDppBcast31 = 0x143;     // lane 31 -> rows 2 and 3

// maximum over the 64 lanes, returned uniformly
__device__ __forceinline__ int wave_max(int v)
{
    v = max(v, dpp_mov<kDppXor1>(v, INT_MIN));
    v = max(v, dpp_mov<kDppXor2>(v, INT_MIN));
    v = max(v, dpp_mov<kDppHalfMirror>(v, INT_MIN));
    v = max(v, dpp_mov<kDppMirror>(v, INT_MIN));
    v = max(v, dpp_mov<kDppBcast15, 0xa>(v, INT_MIN));
    v = max(v, dpp_mov<kDppBcast31, 0xc>(v, INT_MIN));
    return __builtin_amdgcn_readlane(v, 63);
}

// Re-scores the groups that produced the best matches of TWO queries with the
// whole wave (the two independent chains hide each other's load latency; all
// loads of both are issued before the first use) and returns, uniformly, the
// exact best index / value and the exact second-best value of each group.
//   DIR 0: query = row q, group = 16 columns of one (lane, tile-group) stream
//   DIR 1: query = column q, group = 32 rows of one (wave, half-wave)
// Each candidate descriptor is read by DIM/16 neighbouring lanes, 16 bytes each
// (full cache lines per wave instruction -- one candidate per lane would touch
// 64 lines for the same bytes); v_dot4 partial sums are folded with DPP adds.
// Exact inner product = sum of stored bytes products + both corrections
// (SIFT bytes hold value - 128, see prepare_sift_kernel; SURF corrections are 0).
template <int DIM, int DIR>
__device__ __forceinline__ void
rescan_groups(const MatchProblem &pd, const int (&q)[2], const int (&idx1)[2], const int (&code)[2],
    const int (&kind)[2], int lane, int (&idx_out)[2], int (&best_out)[2], int (&second_out)[2])
{
    constexpr int LPC = DIM / 16;       // lanes per candidate (8: SIFT, 4: SURF)
    constexpr int CPL = 64 / LPC;       // candidates per wave-wide load
    constexpr int NCAND = DIR == 0 ? 2 * kPipeGroupTiles : 32;   // kind 0 (keyed kernels): the first 2 * kGroupTiles of them
    constexpr int J = NCAND / CPL;      // wave-wide loads per query
    const int c = lane % LPC, k = lane / LPC;
    const int8_t *Q = DIR == 0 ? pd.A : pd.B;
    const int8_t *Cm = DIR == 0 ? pd.B : pd.A;
    const int32_t *corrQ = DIR == 0 ? pd.corrA : pd.corrB;
    const int32_t *corrC = DIR == 0 ? pd.corrB : pd.corrA;

    int cand[2][J], corr_c[2][J], corr_q[2];
    int4 qv[2], cv[2][J];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        qv[u] = *reinterpret_cast<const int4 *>(Q + (size_t)q[u] * DIM + c * 16);
        corr_q[u] = corrQ[q[u]];
        int base0 = 0, lr = 0, rbase = 0, lh = 0;
        if (DIR == 0) {
            // first column of the group's first tile, lane slot inside the tiles
            const int seg = idx1[u] / kSegCols, off = idx1[u] - seg * kSegCols;
            lr = off & 31;
            // kind 1 (raw-operand kernels): idx1 is the first column of the group's first
            // tile (groups may start at any tile); kind 0 (keyed kernels): the exact best
            // column, its group is the aligned run of kGroupTiles tiles around it
            const int tile = off >> 6;
            base0 = seg * kSegCols + (kind[u] ? tile : (tile / kGroupTiles) * kGroupTiles) * kTileCols;
        } else {
            // first row of the wave's 64-row strip, half-wave
            rbase = idx1[u] * kRowsPerBlock + (code[u] >> 1) * 64;
            lh = code[u] & 1;
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int n = j * CPL + k;
            int cd;
            if (kind[u] >= 3) {
                // a special query's winning stream: 32 candidates at a stride of 128 (match_special_kernel,
                // kind 3) or of 32 (match_special_wide_kernel, kind 4)
                cd = idx1[u] + (kind[u] == 3 ? 128 : 32) * n;
                if (cd >= (DIR == 0 ? pd.n2 : pd.n1)) cd = -1;
            } else if (DIR == 0) {
                cd = base0 + (n >> 1) * kTileCols + (n & 1) * 32 + lr;
                if (cd >= pd.n2 || (!kind[u] && n >= 2 * kGroupTiles)) cd = -1;
            } else {
                const int r = n & 15;
                cd = rbase + (n >> 4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (idx1[u] < pd.nrb_main && cd >= pd.n1) cd = -1;
            }
            cand[u][j] = cd;
        }
        if (DIR == 1 && kind[u] < 3 && idx1[u] >= pd.nrb_main) {
            // a block of gathered special rows (uniform, rare): back to the original rows,
            // all look-ups of the query together and BEFORE any descriptor load -- a
            // look-up between the loads makes the compiler drain the load counter at
            // every candidate (one memory latency each instead of one for all)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int sidx = cand[u][j] - pd.nrb_main * kRowsPerBlock;
                cand[u][j] = sidx < pd.n_special ? pd.special_map[sidx] : -1;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int cs = max(cand[u][j], 0);     // padding candidates read row 0, masked below
            cv[u][j] = *reinterpret_cast<const int4 *>(Cm + (size_t)cs * DIM + c * 16);
            corr_c[u][j] = corrC[cs];
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        // key = score * 32 + position in the group: one maximum yields value and index.
        // Keys are distinct, so (best, second) of a lane is a max / med3 pair.  Equal
        // SCORES are told apart by position only; the caller defers every accepted
        // tie for best to the sequential-scan kernel, so that order never shows.
        int kbest = INT_MIN, ksec = INT_MIN;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            int acc = __builtin_amdgcn_sdot4(qv[u].x, cv[u][j].x, 0, false);
            acc = __builtin_amdgcn_sdot4(qv[u].y, cv[u][j].y, acc, false);
            acc = __builtin_amdgcn_sdot4(qv[u].z, cv[u][j].z, acc, false);
            acc = __builtin_amdgcn_sdot4(qv[u].w, cv[u][j].w, acc, false);
            acc += dpp_mov<kDppXor1>(acc, 0);
            acc += dpp_mov<kDppXor2>(acc, 0);
            if (LPC == 8) acc += dpp_mov<kDppHalfMirror>(acc, 0);
            const int key = cand[u][j] >= 0 ? (acc + corr_q[u] + corr_c[u][j]) * 32 + (j * CPL + k) : INT_MIN;
            ksec = med3a(kbest, ksec, key);                // kbest >= ksec: the middle one
            kbest = max(kbest, key);
        }
        const int wbest = wave_max(kbest);
        const int wsec = wave_max(kbest == wbest ? ksec : kbest);
        // position -> candidate index: the lane group that holds it broadcasts
        const int pos = wbest & 31;
        int widx = 0;
#pragma unroll
        for (int j = 0; j < J; ++j)
            if ((pos / CPL) == j) widx = __builtin_amdgcn_readlane(cand[u][j], (pos % CPL) * LPC);
        idx_out[u] = max(widx, 0);
        best_out[u] = wbest >> 5;
        second_out[u] = wsec == INT_MIN ? INT_MIN : wsec >> 5;
    }
}

// Exact (best, second, index) of query q of direction dir over the special descriptors of the OTHER set
// (value - 128 form with corrections, as everywhere in this kernel); the whole wave works on one query.
template <int DIM>
__device__ __forceinline__ void
scan_specials(const MatchProblem &pd, int dir, int q, int lane, int &best_out, int &second_out, int &idx_out)
{
    constexpr int LPC = DIM / 16, CPL = 64 / LPC;
    const int c = lane % LPC, k = lane / LPC;
    const int8_t *Q = dir == 0 ? pd.A : pd.B;
    const int8_t *Cm = dir == 0 ? pd.B_special : pd.A_special;
    const int32_t *corrC = dir == 0 ? pd.corrB_special : pd.corrA_special;
    const int32_t *map = dir == 0 ? pd.special_map_B : pd.special_map;
    const int ns = dir == 0 ? pd.nsB : pd.nsA;
    const int4 qv = *reinterpret_cast<const int4 *>(Q + (size_t)q * DIM + c * 16);
    const int corr_q = (dir == 0 ? pd.corrA : pd.corrB)[q];
    int kbest = INT_MIN, ksec = INT_MIN;
    for (int base = 0; base < ns; base += CPL) {
        const int slot = base + k;
        const bool live = slot < ns;
        const int4 cv = *reinterpret_cast<const int4 *>(Cm + (size_t)(live ? slot : 0) * DIM + c * 16);
        int acc = __builtin_amdgcn_sdot4(qv.x, cv.x, 0, false);
        acc = __builtin_amdgcn_sdot4(qv.y, cv.y, acc, false);
        acc = __builtin_amdgcn_sdot4(qv.z, cv.z, acc, false);
        acc = __builtin_amdgcn_sdot4(qv.w, cv.w, acc, false);
        acc += dpp_mov<kDppXor1>(acc, 0);
        acc += dpp_mov<kDppXor2>(acc, 0);
        if (LPC == 8) acc += dpp_mov<kDppHalfMirror>(acc, 0);
        const int key = live ? (acc + corr_q + corrC[live ? slot : 0]) * 512 + slot : INT_MIN;
        ksec = med3a(kbest, ksec, key);
        kbest = max(kbest, key);
    }
    const int wbest = wave_max(kbest);
    const int wsec = wave_max(kbest == wbest ? ksec : kbest);
    best_out = wbest >> 9;
    second_out = wsec == INT_MIN ? INT_MIN : wsec >> 9;
    idx_out = map[wbest & 511];
}

template <int DIM, bool SIGNED>
__global__ __launch_bounds__(128) void
match_finish_kernel(const MatchProblem *__restrict__ problems, const RowPart *__restrict__ rowparts,
    const ColPart *__restrict__ colparts, const RowPart *__restrict__ sp_parts, const int32_t *__restrict__ sp_col,
    LoweTable tab, int force_exact, ExactItem *__restrict__ exact_items, int32_t *__restrict__ exact_count, int exact_cap,
    int blocks_per_dir, int total_blocks)
{
    // every XCD works through a contiguous run of problems, so the descriptors the
    // rescans gather from are fetched into one L2 instead of all eight
    const int lin = xcd_remap(blockIdx.x, total_blocks);
    const int problem = lin / (2 * blocks_per_dir), within = lin - problem * 2 * blocks_per_dir;
    const MatchProblem &pd = problems[problem];
    const int dir = within / blocks_per_dir;
    const int block_x = within - dir * blocks_per_dir;
    const int q = block_x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int nq = dir == 0 ? pd.n1 : pd.n2;
    const int nc = dir == 0 ? pd.n2 : pd.n1;
    if ((int)(block_x * blockDim.x) >= nq) return;    // whole block out of range
    int32_t *out = dir == 0 ? pd.m12 : pd.m21;
    const bool active = q < nq;

    // ip1 / idx1: exact best; ip2: second largest GROUP best (lower bound of the
    // true second); code: group of the best (dir 1), idx1 = its row block there
    // kind1 = 3: a special query: (ip1, idx1) = the largest stream maximum of match_special_kernel and the
    // first column of that stream (its 32 candidates follow at a stride of 128), ip2 the second largest
    // stream maximum
    int ip1 = INT_MIN, ip2 = INT_MIN, idx1 = 0, code = 0, kind1 = 0;
    // Problems whose special descriptors went through match_special_kernel (pd.sp): a special
    // query takes its result from there (one partial per chunk of candidates); any other query
    // merges the tile kernel's partials below and then the largest inner product over the OTHER
    // set's special descriptors, which the tile kernel saw as blanks.
    int sp_slot = -1;
    if (active && nc > 0 && pd.sp) {
        const int ns_mine = dir == 0 ? pd.nsA : pd.nsB;
        if (ns_mine > 0) sp_slot = (dir == 0 ? pd.special_slot : pd.special_slot_B)[q];
        if (sp_slot >= 0) {
            const int ns_pad = (ns_mine + 31) & ~31;
            const int chunk_cols = pd.sp_wide[dir] ? kSpWideChunk : kSpChunk;
            const int nchunk = (nc + chunk_cols - 1) / chunk_cols;
            const RowPart *rp0 = sp_parts + pd.sp_row_off[dir] + sp_slot;
            for (int c = 0; c < nchunk; ++c) {
                const RowPart p = rp0[(int64_t)c * ns_pad];
                ip2 = max(max(ip2, p.ip_second), min(ip1, p.ip_best));
                if (p.ip_best >= ip1 && p.ip_best != INT_MIN) { ip1 = p.ip_best; idx1 = p.idx_best; }   // later chunk wins ties
            }
            kind1 = pd.sp_wide[dir] ? 4 : 3;
        }
    }
    // an ordinary query whose best candidate is (or ties with) a special descriptor of the other set:
    // sp_m1 = the best of the ordinary candidates (exact value), the special ones are scanned below
    bool sp_scan = false;
    int sp_m1 = INT_MIN;
    if (active && nc > 0 && sp_slot < 0) {
        if (dir == 0) {
            const int64_t stride = (int64_t)pd.nrb * kRowsPerBlock;
            // special rows keep their partials behind the main row blocks
            int slot = q;
            if (pd.special_slot && !pd.sp) {
                const int sidx = pd.special_slot[q];
                if (sidx >= 0) slot = pd.nrb_main * kRowsPerBlock + sidx;
            }
            // four segments per round, loads issued together (see the column side below)
            constexpr int kBatch = 4;
            const RowPart *rp0 = rowparts + pd.rowpart_off + slot;
            const int nseg = pd.nseg;
            for (int s0 = 0; s0 < nseg; s0 += kBatch) {
                RowPart p[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) p[u] = rp0[(int64_t)min(s0 + u, nseg - 1) * stride];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const bool live = s0 + u < nseg;
                    const int pb = live ? p[u].ip_best : INT_MIN, ps = live ? p[u].ip_second : INT_MIN;
                    // later segment wins ties (its columns have larger indices)
                    ip2 = max(max(ip2, ps), min(ip1, pb));
                    if (pb >= ip1 && pb != INT_MIN) { ip1 = pb; idx1 = p[u].idx_best; kind1 = p[u].pad; }
                }
            }
        } else {
            // eight row blocks per round, their loads issued together (left as a plain
            // loop the compiler emits load, wait, fold per row block: one memory latency
            // per partial with a single request in flight per lane)
            constexpr int kBatch = 8;
            const ColPart *cp0 = colparts + pd.colpart_off + q;
            const int nrb = pd.nrb;
            const int64_t stride = pd.n2stride;
            for (int rb0 = 0; rb0 < nrb; rb0 += kBatch) {
                ColPart p[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u)
                {
                    // (read once, 5 GB of them per launch: non-temporal, so that they do not push the descriptors the
                    //  rescans gather out of the L2)
                    const unsigned long long raw = __builtin_nontemporal_load(
                        reinterpret_cast<const unsigned long long *>(cp0 + (int64_t)min(rb0 + u, nrb - 1) * stride));     // clamped repeats are not folded
                    p[u].key_best = (int)(raw & 0xffffffffu); p[u].key_second = (int)(raw >> 32);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    // no branch (a branch would let the optimiser sink each load to its use)
                    const bool live = rb0 + u < nrb;
                    const int pb = (!live || p[u].key_best == kKeyNone) ? INT_MIN : (p[u].key_best >> 8);
                    const int ps = (!live || p[u].key_second == kKeyNone) ? INT_MIN : (p[u].key_second >> 8);
                    ip2 = max(max(ip2, ps), min(ip1, pb));
                    if (pb >= ip1 && pb != INT_MIN) { ip1 = pb; idx1 = rb0 + u; code = p[u].key_best & 255; }
                }
            }
        }
        if (pd.sp && (dir == 0 ? pd.nsB : pd.nsA) > 0) {
            // the other set's special descriptors as candidates of this query: their largest inner product
            const int t1 = sp_col[pd.sp_col_off[dir ^ 1] + q];
            if (t1 >= ip1 && t1 != INT_MIN) {
                // the best is a special descriptor (or ties with one): which one, and the second best among
                // them, is found by scanning them -- only if the ratio test can still pass (the best of the
                // ordinary candidates, an exact score or a blank's 0, bounds the second best from below)
                sp_m1 = ip1; ip1 = t1; sp_scan = true;
            } else {
                ip2 = max(ip2, t1);
            }
        }
    }
    const int limit = tab.is_signed ? 32767 : 65535;
    bool exact = active && nc > 0 && (force_exact || pd.force_exact || ip1 > limit);
    bool refine = false;
    int res = -1;
    if (active && nc > 0 && !exact) {
        // optimistic test against the lower bound of the second best
        res = accept_match(ip1, sp_scan ? max(sp_m1, ip2) : ip2, 0, tab);
        if (res >= 0) {
            if (ip1 < 0) { res = 0; sp_scan = false; }   // reference state (0, 0, idx 0): nothing to refine
            else if ((sp_scan ? sp_m1 : ip2) == ip1) { exact = true; sp_scan = false; }   // accepted despite a tie for best
                                                // (NaN accept / ratio >= 1): defer to the sequential-scan kernel
            else if (!sp_scan) refine = true;
        } else {
            sp_scan = false;
        }
    } else {
        sp_scan = false;
    }
    // Queries that pass get their best group re-scored, one query at a time by
    // the whole wave.  The rescan also yields the exact best of the group: it
    // differs from ip1 only when ip1 = 0 came from a padding column (raw path),
    // i.e. when no real candidate reaches 0 -- then it is the value to test.
    unsigned long long todo = __ballot(refine);
    int rbest = 0, rsecond = 0, ridx = 0;
    while (todo) {
        int src[2], qs[2], is[2], cs[2], ks[2], idx[2], found[2], second[2];
        src[0] = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        src[1] = todo ? __ffsll((long long)todo) - 1 : src[0];     // odd count: the last one twice
        todo &= todo - 1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            qs[u] = __builtin_amdgcn_readlane(q, src[u]);
            is[u] = __builtin_amdgcn_readlane(idx1, src[u]);
            cs[u] = __builtin_amdgcn_readlane(code, src[u]);
            ks[u] = __builtin_amdgcn_readlane(kind1, src[u]);
        }
        if (dir == 0) rescan_groups<DIM, 0>(pd, qs, is, cs, ks, lane, idx, found, second);
        else rescan_groups<DIM, 1>(pd, qs, is, cs, ks, lane, idx, found, second);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (lane == src[u]) { rbest = found[u]; rsecond = second[u]; ridx = idx[u]; }
    }
    if (refine) {
        // the table lookups of all rescanned queries of the wave at once
        rsecond = max(rsecond, ip2);
        res = accept_match(rbest, rsecond, ridx, tab);
        // accepted although two candidates tie for best (0/0 distances): the index the
        // reference keeps depends on scan order -- leave it to the sequential scan
        if (res >= 0 && rsecond == rbest) exact = true;
    }
    // Ordinary queries whose best candidate is a special descriptor of the other set (rare): the whole
    // wave scores that set's special descriptors for one query at a time -- eight lanes per candidate,
    // exact top-2 on keys value * 512 + slot (slots < 512 = special_kernel_max; inner products < 2^22 here:
    // the wrap-exact path has taken anything above 65535).
    unsigned long long todo2 = __ballot(sp_scan);
    while (todo2) {
        const int src = __ffsll((long long)todo2) - 1;
        todo2 &= todo2 - 1;
        const int qs = __builtin_amdgcn_readlane(q, src);
        int wb, ws, wi;
        scan_specials<DIM>(pd, dir, qs, lane, wb, ws, wi);
        if (lane == src) { rbest = wb; rsecond = ws; ridx = wi; }
    }
    if (sp_scan) {
        rsecond = max(max(rsecond, sp_m1), ip2);
        res = accept_match(rbest, rsecond, ridx, tab);
        if (res >= 0 && rsecond == rbest) exact = true;
    }
    if (exact) {
        const int slot = atomicAdd(exact_count, 1);
        if (slot < exact_cap) {
            ExactItem it;
            it.problem = problem; it.dir = dir; it.query = q;
            exact_items[slot] = it;
        }
    }
    if (active) out[q] = res;
}

void launch_match_finish(const MatchProblem *d_problems, int num_problems, int max_n,
    const RowPart *rowparts, const ColPart *colparts, const RowPart *sp_parts, const int32_t *sp_col, LoweTable tab, int force_exact,
    ExactItem *exact_items, int32_t *exact_count, int exact_cap, hipStream_t s)
{
    if (num_problems <= 0 || max_n <= 0) return;
    const int blocks_per_dir = (max_n + 127) / 128;
    const int64_t total = (int64_t)blocks_per_dir * 2 * num_problems;
    if (total > INT_MAX) return;    // cannot happen: launches are sized by pairs_per_batch
    const dim3 grid((unsigned)total);
    if (tab.is_signed)
        hipLaunchKernelGGL((match_finish_kernel<64, true>), grid, dim3(128), 0, s, d_problems, rowparts,
            colparts, sp_parts, sp_col, tab, force_exact, exact_items, exact_count, exact_cap, blocks_per_dir, (int)total);
    else
        hipLaunchKernelGGL((match_finish_kernel<128, false>), grid, dim3(128), 0, s, d_problems, rowparts,
            colparts, sp_parts, sp_col, tab, force_exact, exact_items, exact_count, exact_cap, blocks_per_dir, (int)total);
}

// ---------------------------------------------------------------------------
// Wrap-exact scan: bit-for-bit emulation of the reference's SSE2 kernel for
// the (rare) queries whose inner products leave the 16-bit range:
// 8 lanes accumulate elements 8i+l modulo 2^16 (_mm_mullo_epi16 /
// _mm_add_epi16, nearest_neighbor.cc:75-81), are read back as T, summed as
// int (:82-84), compared against and stored into T-typed state (:87-101,
// nearest_neighbor.h:50-56).  One wave per query: the 64 lanes evaluate 64
// candidates, then the state is advanced through them in index order.
// ---------------------------------------------------------------------------
template <int DIM, bool SIGNED>
__global__ __launch_bounds__(256) void
exact_scan_kernel(const MatchProblem *__restrict__ problems, const ExactItem *__restrict__ items,
    const int32_t *__restrict__ count, int exact_cap, LoweTable tab)
{
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int n_items = min(*count, exact_cap);
    for (int it = wave_global; it < n_items; it += nwaves) {
        const ExactItem item = items[it];
        const MatchProblem &pd = problems[item.problem];
        const int8_t *Q = item.dir == 0 ? pd.A : pd.B;
        const int8_t *C = item.dir == 0 ? pd.B : pd.A;
        const int nc = item.dir == 0 ? pd.n2 : pd.n1;
        int32_t *out = item.dir == 0 ? pd.m12 : pd.m21;
        const int8_t *qrow = Q + (size_t)item.query * DIM;
        int best = 0, second = 0, idx = 0;   // values as held in T-typed fields
        for (int base = 0; base < nc; base += 64) {
            const int j = base + lane;
            int ip = 0;
            if (j < nc) {
                const int8_t *crow = C + (size_t)j * DIM;
                unsigned lanes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int k = 0; k < DIM; k += 8) {
#pragma unroll
                    for (int l = 0; l < 8; ++l) {
                        const int qa = SIGNED ? (int)qrow[k + l] : (int)qrow[k + l] + 128;
                        const int ca = SIGNED ? (int)crow[k + l] : (int)crow[k + l] + 128;
                        lanes[l] += (unsigned)(qa * ca);
                    }
                }
#pragma unroll
                for (int l = 0; l < 8; ++l)
                    ip += SIGNED ? (int)(short)(lanes[l] & 0xffffu) : (int)(lanes[l] & 0xffffu);
            }
            const int cnt = min(64, nc - base);
            for (int l = 0; l < cnt; ++l) {
                const int v = __shfl(ip, l);
                if (v >= second) {
                    if (v >= best) {
                        second = best;
                        best = SIGNED ? (int)(short)v : (int)(unsigned short)v;
                        idx = base + l;
                    } else {
                        second = SIGNED ? (int)(short)v : (int)(unsigned short)v;
                    }
                }
            }
        }
        if (lane == 0) {
            int d1, d2;
            if (SIGNED) {
                const int b = min(16129, max(0, best)), s2 = min(16129, max(0, second));
                d1 = (int)(short)(32258 - 2 * b); d2 = (int)(short)(32258 - 2 * s2);
            } else {
                const int b = min(65025, best), s2 = min(65025, second);
                d1 = min(32767, 65025 - b) * 2; d2 = min(32767, 65025 - s2) * 2;
            }
            int res = idx;
            if (d1 > tab.max_d1) res = -1;
            else if (d1 >= tab.reject_from[d2 >> 1]) res = -1;
            out[item.query] = res;
        }
    }
}

void launch_exact_scan(int dim, const MatchProblem *d_problems, const ExactItem *items,
    const int32_t *count, int exact_cap, LoweTable tab, hipStream_t s)
{
    if (exact_cap <= 0) return;
    const dim3 grid(1024), block(256);
    if (dim == 128)
        hipLaunchKernelGGL((exact_scan_kernel<128, false>), grid, block, 0, s, d_problems, items,
            count, exact_cap, tab);
    else
        hipLaunchKernelGGL((exact_scan_kernel<64, true>), grid, block, 0, s, d_problems, items,
            count, exact_cap, tab);
}

// ---------------------------------------------------------------------------
// Cross-check (Matching::remove_inconsistent_matches, matching.cc:18-36) and
// mutual-match count (count_consistent_matches, :38-47).  Two phases because
// both directions must be judged from the UNMODIFIED lists: phase 0 marks,
// phase 1 applies (+ adds the combine_results offsets, matching.cc:74-86).
// ---------------------------------------------------------------------------
// kCheckPerThread queries per thread: a workgroup per 256 queries is mostly launch
// overhead for a kernel this small (0.73 ms for 49 M queries that way).
constexpr int kCheckPerThread = 8;

__global__ __launch_bounds__(256) void
cross_check_mark_kernel(const MatchProblem *__restrict__ problems, uint8_t *__restrict__ keep12,
    uint8_t *__restrict__ keep21, const int64_t *__restrict__ mark_off, int32_t *__restrict__ counts)
{
    const MatchProblem &pd = problems[blockIdx.y];
    const int dir = blockIdx.z;
    const int nq = dir == 0 ? pd.n1 : pd.n2;
    const int n_other = dir == 0 ? pd.n2 : pd.n1;
    const int q0 = blockIdx.x * (256 * kCheckPerThread) + threadIdx.x;
    if ((int)(blockIdx.x * (256 * kCheckPerThread)) >= nq) return;       // whole block out of range
    const int32_t *mine = dir == 0 ? pd.m12 : pd.m21;
    const int32_t *other = dir == 0 ? pd.m21 : pd.m12;
    uint8_t *keep = (dir == 0 ? keep12 : keep21) + mark_off[blockIdx.y * 2 + dir];
    int j[kCheckPerThread], back[kCheckPerThread];
#pragma unroll
    for (int u = 0; u < kCheckPerThread; ++u) {
        const int q = q0 + u * 256;
        j[u] = q < nq ? mine[q] : -1;
    }
#pragma unroll
    for (int u = 0; u < kCheckPerThread; ++u)       // entry 0 for "no match": discarded (uniform guard: empty side)
        back[u] = n_other > 0 ? other[max(j[u], 0)] : -1;
    int n_ok = 0;
#pragma unroll
    for (int u = 0; u < kCheckPerThread; ++u) {
        const int q = q0 + u * 256;
        const int ok = (j[u] >= 0 && back[u] == q) ? 1 : 0;
        if (q < nq) keep[q] = (uint8_t)ok;
        n_ok += ok;
    }
    if (dir == 0 && counts) {           // uniform per block
        __shared__ int wave_ok[4];
        for (int off = 32; off >= 1; off >>= 1) n_ok += __shfl_down(n_ok, off);
        if ((threadIdx.x & 63) == 0) wave_ok[threadIdx.x >> 6] = n_ok;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int t = wave_ok[0] + wave_ok[1] + wave_ok[2] + wave_ok[3];
            if (t) atomicAdd(&counts[blockIdx.y], t);
        }
    }
}

__global__ __launch_bounds__(256) void
cross_check_apply_kernel(const MatchProblem *__restrict__ problems, const uint8_t *__restrict__ keep12,
    const uint8_t *__restrict__ keep21, const int64_t *__restrict__ mark_off)
{
    const MatchProblem &pd = problems[blockIdx.y];
    const int dir = blockIdx.z;
    const int nq = dir == 0 ? pd.n1 : pd.n2;
    const int q0 = blockIdx.x * (256 * kCheckPerThread) + threadIdx.x;
    if ((int)(blockIdx.x * (256 * kCheckPerThread)) >= nq) return;
    int32_t *mine = dir == 0 ? pd.m12 : pd.m21;
    const uint8_t *keep = (dir == 0 ? keep12 : keep21) + mark_off[blockIdx.y * 2 + dir];
    const int off = dir == 0 ? pd.out_off12 : pd.out_off21;
    int v[kCheckPerThread];
    uint8_t k[kCheckPerThread];
#pragma unroll
    for (int u = 0; u < kCheckPerThread; ++u) {
        const int q = min(q0 + u * 256, nq - 1);
        v[u] = mine[q]; k[u] = keep[q];
    }
#pragma unroll
    for (int u = 0; u < kCheckPerThread; ++u) {
        const int q = q0 + u * 256;
        if (q < nq) mine[q] = k[u] ? v[u] + off : -1;
    }
}

void launch_cross_check_mark(const MatchProblem *d_problems, int num_problems, int max_n,
    uint8_t *keep12, uint8_t *keep21, const int64_t *mark_off, int32_t *counts, hipStream_t s)
{
    if (num_problems <= 0 || max_n <= 0) return;
    dim3 grid((max_n + 256 * kCheckPerThread - 1) / (256 * kCheckPerThread), num_problems, 2);
    hipLaunchKernelGGL(cross_check_mark_kernel, grid, dim3(256), 0, s, d_problems, keep12, keep21,
        mark_off, counts);
}

void launch_cross_check_apply(const MatchProblem *d_problems, int num_problems, int max_n,
    const uint8_t *keep12, const uint8_t *keep21, const int64_t *mark_off, hipStream_t s)
{
    if (num_problems <= 0 || max_n <= 0) return;
    dim3 grid((max_n + 256 * kCheckPerThread - 1) / (256 * kCheckPerThread), num_problems, 2);
    hipLaunchKernelGGL(cross_check_apply_kernel, grid, dim3(256), 0, s, d_problems, keep12, keep21,
        mark_off);
}

// ---------------------------------------------------------------------------
// Ordered compaction of the combined matches_1_2 list of each kept pair into
// (feature_1, feature_2) correspondences (bundler_matching.cc:176-192).  One
// workgroup per pair; order by feature_1 is preserved by a running prefix.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void
compact_kernel(const int32_t *__restrict__ m12_all, const int64_t *__restrict__ m12_off,
    const int32_t *__restrict__ len12, const int64_t *__restrict__ corr_off,
    const uint8_t *__restrict__ keep, int32_t *__restrict__ corr)
{
    const int p = blockIdx.x;
    if (!keep[p]) return;
    const int32_t *m12 = m12_all + m12_off[p];
    const int n = len12[p];
    int32_t *dst = corr + 2 * corr_off[p];
    __shared__ int wave_sum[4];
    __shared__ int running;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 256) {
        const int i = base + threadIdx.x;
        const int j = i < n ? m12[i] : -1;
        const unsigned long long bal = __ballot(j >= 0);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_sum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_sum[w];
        const int r0 = running;
        if (j >= 0) {
            const int pos = r0 + woff + before;
            dst[2 * pos] = i;
            dst[2 * pos + 1] = j;
        }
        __syncthreads();
        if (threadIdx.x == 0) running = r0 + wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        __syncthreads();
    }
}

void launch_compact_pairs(int num_pairs, const int32_t *m12_all, const int64_t *m12_off,
    const int32_t *len12, const int64_t *corr_off, const uint8_t *keep, int32_t *corr, hipStream_t s)
{
    if (num_pairs <= 0) return;
    hipLaunchKernelGGL(compact_kernel, dim3(num_pairs), dim3(256), 0, s, m12_all, m12_off, len12,
        corr_off, keep, corr);
}

// inlier correspondences of the verified pairs: dst[j] = corr[src + inl[src + j]]
__global__ void
gather_inliers_kernel(const int32_t *__restrict__ corr, const int32_t *__restrict__ inl,
    const int64_t *__restrict__ src_off, const int64_t *__restrict__ dst_off,
    const int32_t *__restrict__ counts, int32_t *__restrict__ out)
{
    const int g = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= counts[g]) return;
    const int64_t so = src_off[g], d = dst_off[g];
    const int id = inl[so + j];
    out[2 * (d + j)] = corr[2 * (so + id)];
    out[2 * (d + j) + 1] = corr[2 * (so + id) + 1];
}

void launch_gather_inliers(int num, const int32_t *corr, const int32_t *inl, const int64_t *src_off,
    const int64_t *dst_off, const int32_t *counts, int32_t *out, hipStream_t s)
{
    if (num <= 0) return;
    // counts are bounded by the features of a view
    hipLaunchKernelGGL(gather_inliers_kernel, dim3(512, num), dim3(256), 0, s, corr, inl, src_off, dst_off, counts, out);
}

// ---------------------------------------------------------------------------
// View preparation: 16-bit lanes -> int8 storage + per-descriptor correction.
// ---------------------------------------------------------------------------
__global__ void
prepare_sift_kernel(const uint16_t *__restrict__ src, int n, int npad, int8_t *__restrict__ dst,
    int32_t *__restrict__ corr, int8_t *__restrict__ dst_raw, int32_t *__restrict__ corr_raw,
    int32_t *__restrict__ range_err)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= npad) return;
    int sum = 0, vmax = 0, vals[2];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = lane * 2 + k;
        int v = row < n ? (int)src[(size_t)row * 128 + e] : 0;     // pad rows: the zero vector
        if (v > 255) { bad = true; v = 255; }
        vals[k] = v;
        dst[(size_t)row * 128 + e] = (int8_t)(v - 128);
        sum += v;
        vmax = max(vmax, v);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { sum += __shfl_xor(sum, m); vmax = max(vmax, __shfl_xor(vmax, m)); }
    // raw form: only rows whose values all fit int8; the others (and the
    // padding) become zero rows that lose every comparison (ip = -2^22)
    const bool raw_ok = row < n && vmax <= 127;
#pragma unroll
    for (int k = 0; k < 2; ++k) dst_raw[(size_t)row * 128 + lane * 2 + k] = (int8_t)(raw_ok ? vals[k] : 0);
    if (lane == 0) {
        // offset form: 128 * sum(a - 128) + 2^20; padding rows 2^22 lower, so every
        // inner product with a padding row comes out at exactly -2^22
        corr[row] = 128 * (sum - 128 * 128) + (1 << 20) - (row < n ? 0 : (1 << 22));
        corr_raw[row] = raw_ok ? 128 * sum : -(1 << 22);
    }
    if (bad) atomicOr(range_err, 1);
}

// rows map[0..n) of (src, corr) gathered into a compact padded set
__global__ void
gather_rows_kernel(const int8_t *__restrict__ src, const int32_t *__restrict__ corr,
    const int32_t *__restrict__ map, int n, int npad, int dim, int32_t pad_corr, int8_t pad_byte,
    int8_t *__restrict__ dst, int32_t *__restrict__ dst_corr)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= npad) return;
    const int from = row < n ? map[row] : -1;
    for (int e = lane; e < dim; e += 64)
        dst[(size_t)row * dim + e] = from >= 0 ? src[(size_t)from * dim + e] : pad_byte;
    if (lane == 0) dst_corr[row] = from >= 0 ? corr[from] : pad_corr;
}

void launch_gather_rows(const int8_t *src, const int32_t *corr, const int32_t *map, int n, int npad,
    int dim, int32_t pad_corr, int8_t pad_byte, int8_t *dst, int32_t *dst_corr, hipStream_t s)
{
    if (npad <= 0) return;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((npad + 3) / 4), dim3(256), 0, s, src, corr, map, n, npad,
        dim, pad_corr, pad_byte, dst, dst_corr);
}

__global__ void
prepare_surf_kernel(const int16_t *__restrict__ src, int n, int npad, int8_t *__restrict__ dst,
    int32_t *__restrict__ corr, int32_t *__restrict__ norm2max, int32_t *__restrict__ range_err)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= npad) return;
    int v = row < n ? (int)src[(size_t)row * 64 + lane] : 0;
    if (v > 127 || v < -128) { atomicOr(range_err, 1); v = 0; }
    dst[(size_t)row * 64 + lane] = (int8_t)v;
    int sq = v * v;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m);
    if (lane == 0) {
        corr[row] = row < n ? 0 : -(1 << 22);     // padding rows lose every comparison
        atomicMax(norm2max, sq);
    }
}

void launch_prepare_sift(const uint16_t *src, int n, int npad, int8_t *dst, int32_t *corr,
    int8_t *dst_raw, int32_t *corr_raw, int32_t *range_err, hipStream_t s)
{
    if (npad <= 0) return;
    hipLaunchKernelGGL(prepare_sift_kernel, dim3((npad + 3) / 4), dim3(256), 0, s, src, n, npad, dst,
        corr, dst_raw, corr_raw, range_err);
}

void launch_prepare_surf(const int16_t *src, int n, int npad, int8_t *dst, int32_t *corr,
    int32_t *norm2max, int32_t *range_err, hipStream_t s)
{
    if (npad <= 0) return;
    hipLaunchKernelGGL(prepare_surf_kernel, dim3((npad + 3) / 4), dim3(256), 0, s, src, n, npad, dst,
        corr, norm2max, range_err);
}

}  // namespace osfm
