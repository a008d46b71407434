// pcreg_amd/csrc/match_sad16.hip -- certified fast path of the SAD descriptor search.
//
// getMatches.m:51-56 with Metric = 'SAD' (every PCReg driver: completeExperimentFast.m:84) is an
// all-pairs sum of absolute differences over D = 981 features.  In fp64 that is two DP-rate VALU
// ops per element.  gfx950 has V_SAD_U16: one instruction adds |a.lo-b.lo| + |a.hi-b.hi| to an
// accumulator, i.e. TWO elements per (half-rate) issue slot -- 4.3x the element rate of the fp64
// loop (scripts/ubench/op_rates.hip).  So:
//
//   1. quantise   both (already L2-normalised) matrices to u16 on one common range, two features
//                 per dword, K-major like the input ([feature pair][row]: coalesced along rows);
//   2. candidates the same 128 x 64 register-tiled all-pairs kernel as the exact one, on dwords:
//                 32 v_sad_u16 per lane and feature pair; per lane and query a sorted top-4
//                 (integer score, index); the 16 lanes sharing a query merge through LDS;
//   3. finalize   one wave per query: candidates that can still be in the top-2 are re-scored
//                 EXACTLY in fp64 with the oracle's accumulation order (s += |a-b|, ascending
//                 feature index, one lane per candidate), wave-shuffle (dist, idx) top-2, and the
//                 certificate: every row outside the lists has integer score >= G, so its true
//                 distance is >= (G - D - 1) / scale; if that exceeds the exact 2nd-best the
//                 answer is proven identical to the exhaustive fp64 search;
//   4. fallback   unproven queries are redone by the exact fp64 kernel.
// Quantisation error: |f - (fmin + u/scale)| <= 0.5/scale per value => each |a-b| moves by at
// most 1/scale => |SAD - SADq/scale| <= D/scale.
#include "common.hpp"
#include "select.hpp"
#include <cmath>
#include <cfloat>
#include <cstdlib>
#include <cstdio>
#include <algorithm>
#include <vector>

namespace pcreg {

namespace {

constexpr int kBlock = 256;
constexpr int TQ = 8, TM = 4;
constexpr int BQ = 16 * TQ, BM = 16 * TM;     // 128 x 64 tile
constexpr int DK2 = 32;                        // feature PAIRS per LDS slab
constexpr int KC = 4;
constexpr int kMaxSplit = 32;                  // S * KC <= 128: two list entries per lane in the finalize wave

struct Range { double fmin, scale, inv_scale; };
// per-segment element strides of the candidates kernel's arrays (A, B: dwords; P: list entries; live: int32) and the
// segments' row offsets; all zero / null for the one-segment call
struct SegZ { size_t A, B, P, live; const int32_t* seg_off; uint32_t* mat; int ldm; };
constexpr int kSadLists = 0, kSadDry = 1, kSadMatrix = 2;     // what the candidates kernel does with a finished 128 x 64 tile of scores

// ---- range (two-stage, deterministic) --------------------------------------------------------
__global__ __launch_bounds__(kBlock) void minmax_partial_kernel(const double* __restrict__ A, int nA, int lda,
                                                                const double* __restrict__ B, int nB, int ldb, int D,
                                                                double* __restrict__ part, const int32_t* __restrict__ nA_live) {
    // nA_live (device, may be null): only the first *nA_live rows of A hold data (the rest of the capacity is skipped)
    const int live = nA_live ? min(nA, *nA_live) : nA;
    double lo = INFINITY, hi = -INFINITY;
    const size_t eA = (size_t)nA * D, eB = (size_t)nB * D;
    for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < eA + eB; e += (size_t)gridDim.x * kBlock) {
        double v;
        if (e < eA) { if ((int)(e % nA) >= live) continue; v = A[e % nA + (e / nA) * (size_t)lda]; }
        else { size_t f = e - eA; v = B[f % nB + (f / nB) * (size_t)ldb]; }
        lo = fmin(lo, v); hi = fmax(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
    __shared__ double s[8];
    if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6] = lo; s[4 + (threadIdx.x >> 6)] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2] = fmin(fmin(s[0], s[1]), fmin(s[2], s[3]));
        part[blockIdx.x * 2 + 1] = fmax(fmax(s[4], s[5]), fmax(s[6], s[7]));
    }
}
__global__ void range_final_kernel(const double* __restrict__ part, int nparts, Range* __restrict__ r) {
    double lo = INFINITY, hi = -INFINITY;                        // launched with one wave; min / max are order-free
    for (int b = threadIdx.x; b < nparts; b += 64) { lo = fmin(lo, part[2 * b]); hi = fmax(hi, part[2 * b + 1]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
    if (threadIdx.x != 0) return;
    if (!(hi > lo)) hi = lo + 1.0;                 // constant (or empty) input: any scale works
    r->fmin = lo; r->scale = 65535.0 / (hi - lo); r->inv_scale = (hi - lo) / 65535.0;
}
// f (n x D, ld) -> packed u16 pairs [D2p][ldq]; D2p = D2 rounded up to DK2, ldq = n rounded up to BQ;
// padding is zero, so the tile loads of the candidates kernel need no bounds checks.
__global__ __launch_bounds__(kBlock) void quantize_pack_kernel(const double* __restrict__ f, int n, int ld, int D, int D2p, int ldq,
                                                               const Range* __restrict__ rp, uint32_t* __restrict__ out,
                                                               const int32_t* __restrict__ n_live) {
    if (n_live) n = min(n, *n_live);                      // rows past the live count quantise to zero like the padding
    const double f0 = rp->fmin, scale = rp->scale;
    size_t total = (size_t)ldq * D2p;
    for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
        int kk = (int)(e / ldq), i = (int)(e % ldq);
        unsigned q[2] = {0u, 0u};
        if (i < n) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int d = 2 * kk + h;
                double u = d < D ? rint((f[i + (size_t)d * ld] - f0) * scale) : 0.0;
                q[h] = (unsigned)fmin(fmax(u, 0.0), 65535.0);
            }
        }
        out[e] = q[0] | (q[1] << 16);
    }
}

// ---- candidates ------------------------------------------------------------------------------------
struct CandU { unsigned s[KC]; int i[KC]; };
__device__ __forceinline__ void candu_insert(CandU& c, unsigned s, int j) {   // ascending j inside a lane: strict '<'
    if (s < c.s[3]) {
        if (s < c.s[1]) {
            c.s[3] = c.s[2]; c.i[3] = c.i[2]; c.s[2] = c.s[1]; c.i[2] = c.i[1];
            if (s < c.s[0]) { c.s[1] = c.s[0]; c.i[1] = c.i[0]; c.s[0] = s; c.i[0] = j; } else { c.s[1] = s; c.i[1] = j; }
        } else {
            if (s < c.s[2]) { c.s[3] = c.s[2]; c.i[3] = c.i[2]; c.s[2] = s; c.i[2] = j; } else { c.s[3] = s; c.i[3] = j; }
        }
    }
}
__device__ __forceinline__ bool lexu_lt(unsigned sa, int ia, unsigned sb, int ib) { return sa < sb || (sa == sb && (unsigned)ia < (unsigned)ib); }
__device__ __forceinline__ void candu_insert_lex(CandU& c, unsigned s, int j) {
    if (j < 0) return;
    int pos = KC;
#pragma unroll
    for (int k = KC - 1; k >= 0; --k) if (lexu_lt(s, j, c.s[k], c.i[k])) pos = k;
    if (pos == KC) return;
#pragma unroll
    for (int k = KC - 1; k > 0; --k) if (k > pos) { c.s[k] = c.s[k - 1]; c.i[k] = c.i[k - 1]; }
#pragma unroll
    for (int k = 0; k < KC; ++k) if (k == pos) { c.s[k] = s; c.i[k] = j; }
}

// grid = (ceil(nA/BQ), S).  part_* layout [S][nA][KC].  Empty slots: idx -1, score 0xFFFFFFFF.
// Aq / Bq are the padded layouts of quantize_pack_kernel (lda / ldb multiples of BQ, D2p of DK2).
template <int MODE>
__global__ __launch_bounds__(kBlock, 3) void sad16_candidates_kernel(const uint32_t* __restrict__ Aq, int nA, int lda,
                                                                  const uint32_t* __restrict__ Bq, int nB, int ldb, int D2p, int chunk,
                                                                  int32_t* __restrict__ part_idx, uint32_t* __restrict__ part_s,
                                                                  unsigned long long* __restrict__ dbg, const int32_t* __restrict__ nA_live,
                                                                  SegZ sz) {
    // segmented form (blockIdx.z = segment, section "segmented getMatches" below): every array is strided by the segment,
    // a segment's B rows are [seg_off[z], seg_off[z + 1]) of the concatenated lists; one segment (sz all zero): unchanged
    {
        const size_t z = blockIdx.z;
        Aq += z * sz.A; Bq += z * sz.B; part_idx += z * sz.P; part_s += z * sz.P;
        if (nA_live) nA_live += z * sz.live;
        if (sz.seg_off) nB = sz.seg_off[z + 1] - sz.seg_off[z];
    }
    if (nA_live && (int)blockIdx.x * BQ >= *nA_live) return;          // a query tile beyond the live rows: nothing to do
    unsigned long long t_start = 0;
    if (dbg) t_start = __builtin_amdgcn_s_memrealtime();
    // two slab buffers (2 x 24 KiB, filled by LDS-DMA) and the final merge cells (32 KiB) share one array
    constexpr int kSlab = DK2 * BQ + DK2 * BM;
    __shared__ __attribute__((aligned(16))) uint32_t smem[2 * kSlab];
    uint32_t* cells = smem;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int q0 = blockIdx.x * BQ, sidx = blockIdx.y;
    const int b_begin = sidx * chunk, b_end = min(nB, b_begin + chunk);
    CandU best[TQ];
#pragma unroll
    for (int r = 0; r < TQ; ++r)
#pragma unroll
        for (int k = 0; k < KC; ++k) { best[r].s[k] = 0xFFFFFFFFu; best[r].i[k] = -1; }

    // One flat sequence of slabs over (model tile, feature slab).  Slab it+1 is copied global -> LDS
    // by `global_load_lds_dwordx4` (no VGPR staging; a wave instruction lands 64 x 16 B contiguously:
    // two 512-B rows of the A slab or four 256-B rows of the B slab) while slab it is consumed.
    const int n_slabs = D2p / DK2;
    const int n_mt = (b_end - b_begin + BM - 1) / BM;
    const int n_it = n_mt * n_slabs;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // uniform 64-bit base (SGPR pair) + one 32-bit per-lane offset: the saddr form of the DMA load
    const unsigned a_off = (unsigned)((lane >> 5) * lda + (lane & 31) * 4);
    const unsigned b_off = (unsigned)((lane >> 4) * ldb + (lane & 15) * 4);
    const uint32_t* a_base = Aq + q0;
    const uint32_t* b_base = Bq + b_begin;
#define PCREG_SAD_DMA(IT, BUF)                                                                                   \
    {                                                                                                            \
        const int mt_ = (IT) / n_slabs, d0_ = ((IT) - mt_ * n_slabs) * DK2;                                      \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                          \
            const int seg = k * 4 + wave;                                                                        \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + (size_t)(d0_ + seg * 2) * lda + a_off), \
                                             (__attribute__((address_space(3))) void*)(smem + (BUF) * kSlab + seg * 256), 16, 0, 0); \
        }                                                                                                        \
        _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                          \
            const int seg = k * 4 + wave;                                                                        \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_base + ((size_t)(d0_ + seg * 4) * ldb + mt_ * BM) + b_off), \
                                             (__attribute__((address_space(3))) void*)(smem + (BUF) * kSlab + DK2 * BQ + seg * 256), 16, 0, 0); \
        }                                                                                                        \
    }
    unsigned acc[TQ][TM];
#pragma unroll
    for (int r = 0; r < TQ; ++r)
#pragma unroll
        for (int c = 0; c < TM; ++c) acc[r][c] = 0u;
    if (n_it > 0) PCREG_SAD_DMA(0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int slab = 0, m0 = b_begin;
    for (int it = 0; it < n_it; ++it) {
        if (it + 1 < n_it) PCREG_SAD_DMA(it + 1, (it + 1) & 1)
        const uint32_t* As = smem + (it & 1) * kSlab + tx * 4;
        const uint32_t* Bs = smem + (it & 1) * kSlab + DK2 * BQ + ty * TM;
        // register double buffering of the LDS reads: feature pair dd+1 is in flight while dd is summed
        uint4 a0[2], a1[2], b0, b1;
        a0[0] = *(const uint4*)(As); a0[1] = *(const uint4*)(As + 64); b0 = *(const uint4*)(Bs);
#define PCREG_SAD_RANK1(A, B)                                                            \
        {                                                                                \
            const unsigned av[8] = {A[0].x, A[0].y, A[0].z, A[0].w, A[1].x, A[1].y, A[1].z, A[1].w}; \
            const unsigned bv[4] = {B.x, B.y, B.z, B.w};                                 \
            _Pragma("unroll") for (int r = 0; r < TQ; ++r)                               \
                _Pragma("unroll") for (int c = 0; c < TM; ++c) acc[r][c] = __builtin_amdgcn_sad_u16(av[r], bv[c], acc[r][c]); \
        }
#pragma unroll 2
        for (int dd = 0; dd < DK2; dd += 2) {
            a1[0] = *(const uint4*)(As + (dd + 1) * BQ); a1[1] = *(const uint4*)(As + (dd + 1) * BQ + 64); b1 = *(const uint4*)(Bs + (dd + 1) * BM);
            PCREG_SAD_RANK1(a0, b0)
            if (dd + 2 < DK2) { a0[0] = *(const uint4*)(As + (dd + 2) * BQ); a0[1] = *(const uint4*)(As + (dd + 2) * BQ + 64); b0 = *(const uint4*)(Bs + (dd + 2) * BM); }
            PCREG_SAD_RANK1(a1, b1)
        }
        if (++slab == n_slabs) {                    // a model tile is complete: fold it into the lists
            slab = 0;
            if (MODE == kSadMatrix) {
                // the whole tile goes to the score matrix [model row][query] (segmented getMatches: every segment selects from it);
                // a lane's four consecutive queries are one 16-byte store, 16 lanes cover 256 contiguous bytes of a row
#pragma unroll
                for (int c = 0; c < TM; ++c) {
                    const int j = m0 + ty * TM + c;
                    if (j < b_end) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            *(uint4*)(sz.mat + (size_t)j * sz.ldm + q0 + h * 64 + tx * 4) = make_uint4(acc[4 * h][c], acc[4 * h + 1][c], acc[4 * h + 2][c], acc[4 * h + 3][c]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < TQ; ++r) {
                unsigned mn = min(min(acc[r][0], acc[r][1]), min(acc[r][2], acc[r][3]));
                if (MODE == kSadDry) { asm volatile("" :: "v"(mn)); }
                else if (MODE == kSadMatrix) {}
                else if (mn < best[r].s[3]) {
#pragma unroll
                    for (int c = 0; c < TM; ++c) {
                        int j = m0 + ty * TM + c;
                        if (j < b_end) candu_insert(best[r], acc[r][c], j);
                    }
                }
#pragma unroll
                for (int c = 0; c < TM; ++c) acc[r][c] = 0u;
            }
            m0 += BM;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // slab it+1 has landed
        __syncthreads();                                        // ... and nobody still reads slab it
    }
#undef PCREG_SAD_DMA
#undef PCREG_SAD_RANK1
    if (MODE == kSadMatrix) return;
    // merge the 16 ty-lists of every query (two halves of 64 queries)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < TQ; ++r) {
            int ql = (r >> 2) * 64 + tx * 4 + (r & 3);      // the register-tile mapping of the main loop
            if ((r >> 2) == half) {
                uint32_t* cell = cells + ((ql & 63) * 16 + ty) * 8;
#pragma unroll
                for (int k = 0; k < KC; ++k) { cell[k] = best[r].s[k]; cell[4 + k] = (uint32_t)best[r].i[k]; }
            }
        }
        __syncthreads();
        if (tid < 64) {
            int qi = q0 + half * 64 + tid;
            if (qi < nA) {
                CandU t;
#pragma unroll
                for (int k = 0; k < KC; ++k) { t.s[k] = 0xFFFFFFFFu; t.i[k] = -1; }
                for (int y = 0; y < 16; ++y) {
                    const uint32_t* cell = cells + (tid * 16 + y) * 8;
#pragma unroll
                    for (int k = 0; k < KC; ++k) candu_insert_lex(t, cell[k], (int)cell[4 + k]);
                }
                size_t o = ((size_t)sidx * nA + qi) * KC;
#pragma unroll
                for (int k = 0; k < KC; ++k) { part_idx[o + k] = t.i[k]; part_s[o + k] = t.s[k]; }
            }
        }
    }
    if (dbg && tid == 0) {          // PCREG_SAD_TIMELINE: residency timeline of the grid (100 MHz clock, HW_ID, XCC_ID)
        unsigned long long* d = dbg + 4 * (size_t)(blockIdx.y * gridDim.x + blockIdx.x);
        d[0] = t_start; d[1] = __builtin_amdgcn_s_memrealtime();
        d[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); d[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
}

// ---- row-major copies for the re-rank -----------------------------------------------------------
// The inputs are MATLAB-shaped (feature-major: a row's D features lie ld*8 bytes apart, one page
// each), which makes "read row j" 981 translations.  The re-rank reads whole rows, so it gets a
// row-major copy: 64 x 64 tiles through LDS, coalesced on both sides.
__global__ __launch_bounds__(kBlock) void transpose_rows_kernel(const double* __restrict__ f, int n, int ld, int D,
                                                                double* __restrict__ out /* [n][D] */, const int32_t* __restrict__ n_live) {
    __shared__ double tile[64][65];
    const int i0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    if (n_live) { n = min(n, *n_live); if (i0 >= n) return; }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int d = d0 + ty + 4 * k, i = i0 + tx;
        tile[ty + 4 * k][tx] = (d < D && i < n) ? f[i + (size_t)d * ld] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int i = i0 + ty + 4 * k, d = d0 + tx;
        if (i < n && d < D) out[(size_t)i * D + d] = tile[tx][ty + 4 * k];
    }
}

// ---- exact re-rank + certificate: one wave per query, one lane per candidate ----------------------
__device__ __forceinline__ bool lexd_lt(double da, int ia, double db, int ib) { return da < db || (da == db && (unsigned)ia < (unsigned)ib); }

// Entries of a query: S chunks x KC, lane e handles entries e and e + 64 (S <= kMaxSplit = 32).
// The |a - b| terms of a candidate are order-free, so the whole wave computes them in parallel into
// LDS (256 features x up to kNC candidates at a time); only the SUM keeps the oracle's order: lane c
// adds candidate c's terms one by one, s carried from tile to tile.
#ifndef PCREG_SAD_KFT
#define PCREG_SAD_KFT 128
#endif
constexpr int kNC = 4, kFT = PCREG_SAD_KFT, kEPL = 2;      // candidates per re-rank group, features per LDS tile, list entries per lane
__global__ __launch_bounds__(kBlock) void sad16_finalize_kernel(const double* __restrict__ At, int nA,
                                                                const double* __restrict__ Bt, int nB, int D,
                                                                const Range* __restrict__ rp, const int32_t* __restrict__ part_idx,
                                                                const uint32_t* __restrict__ part_s, int S,
                                                                int32_t* __restrict__ idx, double* __restrict__ dist,
                                                                int32_t* __restrict__ flag_list, int32_t* __restrict__ n_flag, int force_unproven,
                                                                const int32_t* __restrict__ nA_live, unsigned long long* __restrict__ stats) {
    __shared__ double s_t[kBlock / 64][kNC][kFT];          // 32 KiB
    __shared__ int s_j[kBlock / 64][64 * kEPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = blockIdx.x * (kBlock / 64) + wave;
    if (qi >= nA || (nA_live && qi >= *nA_live)) return;   // wave-uniform; no block barrier below
    const int total = S * KC;                              // <= 64 * kEPL
    int j[kEPL]; unsigned sq[kEPL];
    unsigned a1 = 0xFFFFFFFFu, a2 = 0xFFFFFFFFu, g = 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < kEPL; ++u) {
        const int e = lane + 64 * u;
        j[u] = -1; sq[u] = 0xFFFFFFFFu;
        if (e < total) { size_t o = ((size_t)(e / KC) * nA + qi) * KC + (e % KC); j[u] = part_idx[o]; sq[u] = part_s[o]; }
        if (j[u] >= 0) {
            if (sq[u] < a1) { a2 = a1; a1 = sq[u]; } else if (sq[u] < a2) a2 = sq[u];
            if ((e % KC) == KC - 1) g = min(g, sq[u]);     // a full list: rows outside it score >= its last entry
        }
    }
    // the two smallest integer scores of the union, and G = the smallest "last entry of a full list"
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned b1 = __shfl_xor(a1, o), b2 = __shfl_xor(a2, o);
        unsigned n1 = min(a1, b1), n2 = min(max(a1, b1), min(a2, b2));
        a1 = n1; a2 = n2;
        g = min(g, (unsigned)__shfl_xor((int)g, o));
    }
    const unsigned slack = 2u * (unsigned)(D + 1) + 2u;
    // compact the candidates that can still be among the two best into s_j
    int n_need = 0;
#pragma unroll
    for (int u = 0; u < kEPL; ++u) {
        const bool need = j[u] >= 0 && (a2 == 0xFFFFFFFFu || a2 > 0xFFFFFFFFu - slack || sq[u] <= a2 + slack);
        const unsigned long long m = __ballot(need);
        if (need) s_j[wave][n_need + __popcll(m & ((1ull << lane) - 1ull))] = j[u];
        n_need += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();

    double d1 = INFINITY, d2 = INFINITY; int i1 = -1, i2 = -1;       // lane-private running top-2
    const double* a = At + (size_t)qi * D;                 // row-major copies: a row is contiguous
    for (int g0 = 0; g0 < n_need; g0 += kNC) {
        const int nc = min(kNC, n_need - g0);
        double sum = 0.0;
        for (int d0 = 0; d0 < D; d0 += kFT) {
            // branch-free, clamped addresses: all (1 + kNC) x 4 scattered loads are in flight together
            double av[kFT / 64], bv[kNC][kFT / 64];
            size_t dof[kFT / 64];
#pragma unroll
            for (int u = 0; u < kFT / 64; ++u) dof[u] = (size_t)min(d0 + lane + 64 * u, D - 1);
#pragma unroll
            for (int u = 0; u < kFT / 64; ++u) av[u] = a[dof[u]];
#pragma unroll
            for (int c = 0; c < kNC; ++c) {
                const double* b = Bt + (size_t)s_j[wave][min(g0 + c, n_need - 1)] * D;
#pragma unroll
                for (int u = 0; u < kFT / 64; ++u) bv[c][u] = b[dof[u]];
            }
#pragma unroll
            for (int c = 0; c < kNC; ++c)
#pragma unroll
                for (int u = 0; u < kFT / 64; ++u)
                    s_t[wave][c][lane + 64 * u] = (d0 + lane + 64 * u < D) ? fabs(av[u] - bv[c][u]) : 0.0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (lane < nc) {
                // the oracle's order; terms past D are +0.0, which leaves a non-negative sum unchanged
                const double* t = s_t[wave][lane];
                for (int k = 0; k < kFT; k += 16) {
                    double v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = t[k + u];
#pragma unroll
                    for (int u = 0; u < 16; ++u) sum += v[u];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
        if (lane < nc) {
            const int jc = s_j[wave][g0 + lane];
            if (lexd_lt(sum, jc, d1, i1)) { d2 = d1; i2 = i1; d1 = sum; i1 = jc; }
            else if (lexd_lt(sum, jc, d2, i2)) { d2 = sum; i2 = jc; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double e1 = __shfl_xor(d1, o), e2 = __shfl_xor(d2, o);
        int k1 = __shfl_xor(i1, o), k2 = __shfl_xor(i2, o);
        // empty slots carry (inf, -1): (unsigned)-1 orders last
        bool fm = lexd_lt(d1, i1, e1, k1);
        double w1 = fm ? d1 : e1; int x1 = fm ? i1 : k1;
        double m2 = fm ? d2 : d1; int y2 = fm ? i2 : i1;
        double o2 = fm ? e1 : e2; int z2 = fm ? k1 : k2;
        bool sm = lexd_lt(m2, y2, o2, z2);
        d1 = w1; i1 = x1; d2 = sm ? m2 : o2; i2 = sm ? y2 : z2;
    }
    bool ok;
    if (g == 0xFFFFFFFFu) ok = true;                      // no chunk filled its list: every row is a candidate
    else {
        double lower = ((double)g - (double)(D + 1)) * rp->inv_scale;
        ok = (i2 >= 0) && (lower * (1.0 - 1e-12) > d2);
    }
    if (force_unproven) ok = false;                        // test hook: exercise the fallback
    if (lane == 0) {
        if (ok) { idx[(size_t)qi * 2] = i1; idx[(size_t)qi * 2 + 1] = i2; dist[(size_t)qi * 2] = d1; dist[(size_t)qi * 2 + 1] = d2; }
        else { int slot = atomicAdd(n_flag, 1); flag_list[slot] = qi; }
        if (stats) {          // pcreg_debug_set("match_stats", 1): what the certificate proved, re-scored, handed on
            atomicAdd(&stats[0], 1ull); atomicAdd(&stats[1], (unsigned long long)n_need);
            if (!ok) { atomicAdd(&stats[2], 1ull); atomicAdd(&stats[3], 1ull); }
        }
    }
}

__global__ void stats_bump_kernel(unsigned long long* stats, int k, unsigned long long v) { atomicAdd(&stats[k], v); }

// ---- fallback: unproven queries, exhaustively, exact ---------------------------------------------
// grid (nf, SC): block = one unproven query x one slice of B; the query row sits in LDS, each thread
// owns rows j = begin + tid + 256 t and accumulates in the oracle's order; B is read coalesced.
constexpr int kFbSlices = 64;
__global__ __launch_bounds__(kBlock) void sad_exact_rows_kernel(const double* __restrict__ A, int lda,
                                                                const double* __restrict__ B, int nB, int ldb, int D,
                                                                const int32_t* __restrict__ list, const int32_t* __restrict__ n_flag,
                                                                int cap, int slice,
                                                                int32_t* __restrict__ part_idx, double* __restrict__ part_dist) {
    extern __shared__ double s_a[];                           // D doubles
    __shared__ double s_d[4][2]; __shared__ int s_i[4][2];
    const int nf = min(*n_flag, cap);                         // device-side count: no host round trip (usually 0 .. 9)
    for (int k = blockIdx.x; k < nf; k += gridDim.x) {
    const int qi = list[k];
    __syncthreads();                                          // the previous trip's readers are done with s_a / s_d
    for (int d = threadIdx.x; d < D; d += kBlock) s_a[d] = A[qi + (size_t)d * lda];
    __syncthreads();
    const int begin = blockIdx.y * slice, end = min(nB, begin + slice);
    double d1 = INFINITY, d2 = INFINITY; int i1 = -1, i2 = -1;
    for (int j = begin + threadIdx.x; j < end; j += kBlock) {
        const double* b = B + j;
        double s = 0.0;
        int d = 0;
        for (; d + 32 <= D; d += 32) {                       // 32 coalesced loads in flight, then the ordered sum
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = b[(size_t)(d + u) * ldb];
#pragma unroll
            for (int u = 0; u < 32; ++u) s += fabs(s_a[d + u] - v[u]);
        }
        for (; d < D; ++d) s += fabs(s_a[d] - b[(size_t)d * ldb]);
        if (s < d1) { d2 = d1; i2 = i1; d1 = s; i1 = j; } else if (s < d2) { d2 = s; i2 = j; }   // ascending j: strict
    }
    auto merge = [&](double e1, int k1, double e2, int k2) {
        bool fm = lexd_lt(d1, i1, e1, k1);
        double w1 = fm ? d1 : e1; int x1 = fm ? i1 : k1;
        double m2 = fm ? d2 : d1; int y2 = fm ? i2 : i1;
        double o2 = fm ? e1 : e2; int z2 = fm ? k1 : k2;
        bool sm = lexd_lt(m2, y2, o2, z2);
        d1 = w1; i1 = x1; d2 = sm ? m2 : o2; i2 = sm ? y2 : z2;
    };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) merge(__shfl_xor(d1, o), __shfl_xor(i1, o), __shfl_xor(d2, o), __shfl_xor(i2, o));
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_d[wv][0] = d1; s_d[wv][1] = d2; s_i[wv][0] = i1; s_i[wv][1] = i2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) merge(s_d[w][0], s_i[w][0], s_d[w][1], s_i[w][1]);
        size_t o = ((size_t)blockIdx.y * cap + k) * 2;       // [slice][cap][2]
        part_idx[o] = i1; part_idx[o + 1] = i2; part_dist[o] = d1; part_dist[o + 1] = d2;
    }
    }
}
// merge the slices' top-2 of every flagged row by (distance, index) and put the result in its place
__global__ void sad_fallback_finish_kernel(const int32_t* __restrict__ list, const int32_t* __restrict__ n_flag, int cap, int slices,
                                           const int32_t* __restrict__ part_idx, const double* __restrict__ part_dist,
                                           int32_t* __restrict__ idx, double* __restrict__ dist) {
    const int nf = min(*n_flag, cap);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nf; k += gridDim.x * blockDim.x) {
        Top2T<double> t{INFINITY, INFINITY, -1, -1};
        for (int sl = 0; sl < slices; ++sl) {
            const size_t o = ((size_t)sl * cap + k) * 2;
            top2_insert_lex_t(t, part_dist[o], part_idx[o]);
            top2_insert_lex_t(t, part_dist[o + 1], part_idx[o + 1]);
        }
        const int qi = list[k];
        idx[(size_t)qi * 2] = t.i1; idx[(size_t)qi * 2 + 1] = t.i2;
        dist[(size_t)qi * 2] = t.d1; dist[(size_t)qi * 2 + 1] = t.d2;
    }
}


// ================================================================================================================
// Segmented getMatches: ONE surface descriptor set against MANY row subsets of one model set in one chain of launches
// (the sphere sweep of completeExperimentFast.m:101-150: getMatches(descSurface, descModel(mask_i, :), par) for every
// sphere i).  blockIdx.z (or .x where a workgroup owns a segment) = segment.
//
// What differs between segments is only (a) which model rows take part and (b) the appended constant
// c_i = norm_factor * mean(vecnorm([descSurface; descCur_i], 1, 2)) (getMatches.m:24-26), hence every row's L2 norm.
// Everything else is computed ONCE for all segments:
//   P    = descriptors .^ metric_factor            (getMatches.m:36-37 on the 980 real columns)
//   l1   = vecnorm(row, 1)                         (ascending feature order, like row_l1_kernel)
//   s2   = sum of P^2 over the real columns        (ascending, fma: the prefix of normalize_rows_kernel's chain)
// and per segment: cc_i = c_i ^ metric_factor, nrm_i(row) = sqrt(fma(cc_i, cc_i, s2(row))) -- the same chain, since
// the constant column comes last.  A normalised value is P / nrm_i(row) (last column cc_i / nrm_i(row)): the SAME IEEE
// operations on the same operands as the one-segment path (preprocess_kernel, normalize_rows_kernel), so the exact
// re-rank sees the same bits and the pairs are those of one pcreg_dev_get_matches call per segment.  No normalised
// double matrix is ever materialised: the re-rank divides the handful of rows it touches on the fly.
//
// The APPROXIMATE scores are shared by all segments.  The segments' constants differ by ~1e-3 relative (the mean of
// ~3500 row lengths), so one u16 quantisation under a REFERENCE constant cc_ref (the middle of the segments' constants) scores
// every (surface row, model row) pair ONCE -- the spheres of the sweep overlap ~8-fold, and the back-search of Unique reads
// the same matrix.  Segment i relates its true distances to the reference ones: with u(r) = 1 / n_ref(r), w(r) = 1 / n_i(r)
// and ONE factor rho_i (the ratio n_ref / n_i at a typical row: the constant dominates every norm, so all rows rescale
// almost alike),
//     SAD_i(a, b) - rho_i SAD_ref(a, b) = sum_k<D0 (|p_ak w_a - p_bk w_b| - rho_i |p_ak u_a - p_bk u_b|) + (|g_a - g_b| ...)
//     |.| <= t(a) + t(b) + (max g - min g),   t(r) = sum|P(r)| |w_r - rho_i u_r|,   g(r) = cc_i w_r - rho_i cc_ref u_r
// (triangle inequality per column; the appended column's two terms differ by at most the spread of g over the rows in play).
// E_i = max_a t + max_b t + spread(g), evaluated exactly per segment: a few units of the quantisation against the D + 1 = 982
// the rounding already costs, so the certificate is as strong as the one-segment path's.
struct SegConst { double cc, fmin, scale, inv_scale, rho; unsigned eunits, slack2; };

constexpr int kPR = 64, kPW = 16, kPT = 64;          // rows per workgroup, per wave; features per tile
// src row-major [n][D] -> P row-major, l1, s2, min / max of P per row.  One LANE per row keeps the oracle's order (980 dependent
// adds per row), so the kernel has only n / 64 waves' worth of arithmetic and lives on latency: each WAVE takes 16 rows (four
// times the waves of one wave per 64 rows, no workgroup barrier anywhere), reads a 16 x 64 tile with 16 coalesced loads that are
// issued before the previous tile's chains run, and hands the tile to its 16 summing lanes through LDS.  (Round 3's form -- a
// workgroup per 64 rows, one of its four waves summing between two barriers per 48 features -- took 0.62 ms for the sweep's
// 60 000 x 980 model, 1 GB of traffic.)
__global__ __launch_bounds__(kBlock) void segp_rows_kernel(const double* __restrict__ src, int n, int D, int change_metric, double factor,
                                                           double* __restrict__ P, double* __restrict__ l1, double* __restrict__ s2,
                                                           double* __restrict__ pmin, double* __restrict__ pmax, double* __restrict__ sp,
                                                           double* __restrict__ odd /* 1: the row holds a value outside {0} U [2^-400, 2^400] */) {
    __shared__ double t_raw[kBlock / 64][kPT][kPW + 1], t_p[kBlock / 64][kPT][kPW + 1];
    // The descriptors are COUNTS (getSpacialHistogramDescriptors: integers, a few units per bin): their powers come from a table
    // of this very pow -- same bits -- that each workgroup fills first (kPowTab fp64 pow calls against the 62 720 of its rows);
    // anything that is not a small integer takes pow itself.
    constexpr int kPowTab = 512;
    __shared__ double s_pow[kPowTab];
    if (change_metric) {
        for (int e = threadIdx.x; e < kPowTab; e += kBlock) s_pow[e] = pow((double)e, factor);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * kPR + wave * kPW, rows = min(kPW, n - r0);
    if (rows <= 0) return;                                     // wave-uniform; no workgroup barrier below
    double (*traw)[kPW + 1] = t_raw[wave], (*tp)[kPW + 1] = t_p[wave];
    double a = 0.0, s = 0.0, lo = INFINITY, hi = -INFINITY, ap = 0.0;
    bool bad = false;
    double x[kPW];
    auto load_tile = [&](int d0) {
#pragma unroll
        for (int i = 0; i < kPW; ++i) {
            const int r = min(i, rows - 1), d = min(d0 + lane, D - 1);          // clamped: the loads are unconditional, the surplus unused
            x[i] = src[(size_t)(r0 + r) * D + d];
        }
    };
    load_tile(0);
    for (int d0 = 0; d0 < D; d0 += kPT) {
        const int dn = min(kPT, D - d0);
#pragma unroll
        for (int i = 0; i < kPW; ++i) {
            double pv = x[i];
            if (change_metric) {
                const int xi = (x[i] >= 0.0 && x[i] < (double)kPowTab) ? (int)x[i] : -1;
                pv = (xi >= 0 && (double)xi == x[i]) ? s_pow[xi] : pow(x[i], factor);
            }
            if (i < rows && lane < dn) P[(size_t)(r0 + i) * D + d0 + lane] = pv;
            traw[lane][i] = x[i]; tp[lane][i] = pv;
        }
        if (d0 + kPT < D) load_tile(d0 + kPT);                 // in flight while the chains below run
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (lane < rows) {
            for (int f = 0; f < dn; ++f) {
                const double xr = traw[f][lane], pv = tp[f][lane];
                a += fabs(xr); s = fma(pv, pv, s); lo = fmin(lo, pv); hi = fmax(hi, pv); ap += fabs(pv);
                bad = bad || !(pv == 0.0 || (fabs(pv) >= 0x1p-400 && fabs(pv) <= 0x1p400));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    if (lane < rows) { const int r = r0 + lane; l1[r] = a; s2[r] = s; pmin[r] = lo; pmax[r] = hi; sp[r] = ap; odd[r] = bad ? 1.0 : 0.0; }
}

struct SegSets {                // the two descriptor sets after segp_rows_kernel, and the segments
    const double *PS, *l1S, *s2S, *pminS, *pmaxS, *spS, *oddS;      // surface: Q rows
    const double *PM, *l1M, *s2M, *pminM, *pmaxM, *spM, *oddM;      // model: every row of the full set (VM)
    const int32_t *seg_rows, *seg_off;                // segment z owns model rows seg_rows[seg_off[z] .. seg_off[z + 1])
    int Q, VM, D0, Dp;                                // Dp = D0 + 1 with the appended column
    size_t rinv_s, rinv_m;                            // the reciprocals of the norms sit this many doubles behind nrmS / nrmM
};

__device__ __forceinline__ double seg_norm(const pcreg_match_opts& o, double cc, double q2) {
    if (o.prenormalized) return 1.0;
    const double nrm = sqrt(o.unnormalize ? fma(cc, cc, q2) : q2);
    return nrm <= (double)FLT_EPSILON ? INFINITY : nrm;        // normalizeX: an effectively-zero row becomes zeros
}
// every segment's appended constant (mean_kernel's order over [l1 of the surface; l1 of the segment's rows]) -> sc[z].cc
__global__ __launch_bounds__(kBlock) void segp_cc_kernel(SegSets S, pcreg_match_opts o, SegConst* __restrict__ sc) {
    __shared__ double s[kBlock];
    const int z = blockIdx.x, off = S.seg_off[z], n = S.seg_off[z + 1] - off, Q = S.Q, tid = threadIdx.x;
    const int32_t* rows = S.seg_rows + off;
    double cc = 0.0;
    if (o.unnormalize) {
        double a = 0;
        for (int i = tid; i < Q + n; i += kBlock) a += i < Q ? S.l1S[i] : S.l1M[rows[i - Q]];
        s[tid] = a;
        __syncthreads();
        for (int w = kBlock / 2; w > 0; w >>= 1) { if (tid < w) s[tid] += s[tid + w]; __syncthreads(); }
        const double c = o.norm_factor * (s[0] / (double)(Q + n));
        cc = o.change_metric ? pow(c, o.metric_factor) : c;
    }
    if (tid == 0) sc[z].cc = cc;
}
// The reference constant = the middle of the segments' constants (any value works: it only sets how much slack the segments
// need), every row's norm under it, and the common quantisation range; ref = sc[n_seg].  Minima and maxima only, so the rows are
// spread over kRefParts workgroups whose partial ranges a second, one-workgroup launch folds (as ONE workgroup over all 62 000
// rows of the sweep -- 242 square roots and divisions in a row per thread -- it was 0.14 ms of every sweep).
constexpr int kRefParts = 1024;
__device__ __forceinline__ void seg_ref_fold(double (&v)[4], double (*s_w)[kBlock / 64]) {      // v = {lo, hi, q_lo, q_hi} -> thread 0 holds the workgroup's
    const int tid = threadIdx.x;
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) {
        v[0] = fmin(v[0], __shfl_xor(v[0], w)); v[1] = fmax(v[1], __shfl_xor(v[1], w));
        v[2] = fmin(v[2], __shfl_xor(v[2], w)); v[3] = fmax(v[3], __shfl_xor(v[3], w));
    }
    if ((tid & 63) == 0) { for (int k = 0; k < 4; ++k) s_w[k][tid >> 6] = v[k]; }
    __syncthreads();
    if (tid == 0)
        for (int w = 1; w < kBlock / 64; ++w) { v[0] = fmin(v[0], s_w[0][w]); v[1] = fmax(v[1], s_w[1][w]); v[2] = fmin(v[2], s_w[2][w]); v[3] = fmax(v[3], s_w[3][w]); }
}
__global__ __launch_bounds__(kBlock) void segp_ref_kernel(SegSets S, pcreg_match_opts o, double* __restrict__ nrefS, double* __restrict__ nrefM,
                                                          const SegConst* __restrict__ sc, int n_seg, double* __restrict__ part /*[gridDim.x][4]*/) {
    __shared__ double s_w[4][kBlock / 64];
    __shared__ double s_cc;
    const int tid = threadIdx.x, n = S.Q + S.VM;
    {   // every workgroup derives the same constant from the segments' (a few hundred values)
        double v[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
        for (int z = tid; z < n_seg; z += kBlock) { v[0] = fmin(v[0], sc[z].cc); v[1] = fmax(v[1], sc[z].cc); }
        seg_ref_fold(v, s_w);
        if (tid == 0) s_cc = 0.5 * (v[0] + v[1]);
        __syncthreads();
    }
    const double cc = o.unnormalize ? s_cc : 0.0;
    double v[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int i = blockIdx.x * kBlock + tid; i < n; i += gridDim.x * kBlock) {
        const bool surf = i < S.Q;
        const int r = surf ? i : i - S.Q;
        const double q2 = surf ? S.s2S[r] : S.s2M[r];
        v[2] = fmin(v[2], q2); v[3] = fmax(v[3], q2);
        const double nrm = seg_norm(o, cc, q2);
        if (surf) nrefS[r] = nrm; else nrefM[r] = nrm;
        const double a = (surf ? S.pminS[r] : S.pminM[r]) / nrm, b = (surf ? S.pmaxS[r] : S.pmaxM[r]) / nrm;
        v[0] = fmin(v[0], fmin(a, b)); v[1] = fmax(v[1], fmax(a, b));
        if (o.unnormalize) { const double c = cc / nrm; v[0] = fmin(v[0], c); v[1] = fmax(v[1], c); }
    }
    seg_ref_fold(v, s_w);
    if (tid == 0) { for (int k = 0; k < 4; ++k) part[(size_t)blockIdx.x * 4 + k] = v[k]; if (blockIdx.x == 0) part[(size_t)gridDim.x * 4] = cc; }
}
__global__ __launch_bounds__(kBlock) void segp_ref_final_kernel(const double* __restrict__ part, int n_parts, SegConst* __restrict__ ref) {
    __shared__ double s_w[4][kBlock / 64];
    double v[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int p = threadIdx.x; p < n_parts; p += kBlock) {
        v[0] = fmin(v[0], part[(size_t)p * 4]); v[1] = fmax(v[1], part[(size_t)p * 4 + 1]);
        v[2] = fmin(v[2], part[(size_t)p * 4 + 2]); v[3] = fmax(v[3], part[(size_t)p * 4 + 3]);
    }
    seg_ref_fold(v, s_w);
    if (threadIdx.x == 0) {
        double lo = v[0], hi = v[1];
        if (!(hi > lo)) hi = lo + 1.0;
        // .rho of the reference slot carries the typical row's sum of squares (the segments derive their factor from it)
        *ref = SegConst{part[(size_t)n_parts * 4], lo, 65535.0 / (hi - lo), (hi - lo) / 65535.0, 0.5 * (v[2] + v[3]), 0u, 0u};
    }
}

// one workgroup per segment: every row's norm under the segment's constant, and E_i in quantisation units (how far the segment's normalised rows are from the reference ones)
__global__ __launch_bounds__(kBlock) void segp_consts_kernel(SegSets S, pcreg_match_opts o, const double* __restrict__ nrefS,
                                                             const double* __restrict__ nrefM, double* __restrict__ nrmS,
                                                             double* __restrict__ nrmM, SegConst* __restrict__ sc, int n_seg) {
    double *rinvS = nrmS + S.rinv_s, *rinvM = nrmM + S.rinv_m;
    const int z = blockIdx.x, off = S.seg_off[z], n = S.seg_off[z + 1] - off, Q = S.Q, tid = threadIdx.x;
    const int32_t* rows = S.seg_rows + off;
    const SegConst ref = sc[n_seg];
    const double cc = sc[z].cc;
    double rho = 1.0;
    if (o.unnormalize && !o.prenormalized) {
        const double nr = sqrt(fma(ref.cc, ref.cc, ref.rho)), ni = sqrt(fma(cc, cc, ref.rho));
        if (nr > 0.0 && ni > 0.0 && nr < INFINITY && ni < INFINITY) rho = nr / ni;
    }
    double ts = 0.0, tm = 0.0, g_lo = INFINITY, g_hi = -INFINITY;       // max t over the surface rows / the segment's model rows; range of g
    for (int i = tid; i < Q + n; i += kBlock) {
        const bool surf = i < Q;
        const int r = surf ? i : rows[i - Q];
        const double nrm = seg_norm(o, cc, surf ? S.s2S[r] : S.s2M[r]);
        if (surf) nrmS[(size_t)z * Q + i] = nrm; else nrmM[off + i - Q] = nrm;
        const double wi = 1.0 / nrm, ur = 1.0 / (surf ? nrefS[r] : nrefM[r]);          // 0 for a row that normalizeX zeroes
        {   // the correctly rounded reciprocal of the norm, for seg_value's division by multiplication -- or 0 when a value of
            // the row, the appended constant or the norm itself is outside the range in which that division is proven exact
            const bool cc_ok = cc == 0.0 || (fabs(cc) >= 0x1p-400 && fabs(cc) <= 0x1p400);
            const bool ok = cc_ok && (surf ? S.oddS[r] : S.oddM[r]) == 0.0 && nrm >= 0x1p-400 && nrm <= 0x1p400;
            if (surf) rinvS[(size_t)z * Q + i] = ok ? wi : 0.0; else rinvM[off + i - Q] = ok ? wi : 0.0;
        }
        const double t = (surf ? S.spS[r] : S.spM[r]) * fabs(wi - rho * ur);
        if (surf) ts = fmax(ts, t); else tm = fmax(tm, t);
        if (o.unnormalize) { const double g = cc * wi - rho * ref.cc * ur; g_lo = fmin(g_lo, g); g_hi = fmax(g_hi, g); }
    }
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) {
        ts = fmax(ts, __shfl_xor(ts, w)); tm = fmax(tm, __shfl_xor(tm, w));
        g_lo = fmin(g_lo, __shfl_xor(g_lo, w)); g_hi = fmax(g_hi, __shfl_xor(g_hi, w));
    }
    __shared__ double s_r[4][kBlock / 64];
    if ((tid & 63) == 0) { s_r[0][tid >> 6] = ts; s_r[1][tid >> 6] = tm; s_r[2][tid >> 6] = g_lo; s_r[3][tid >> 6] = g_hi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kBlock / 64; ++w) { ts = fmax(ts, s_r[0][w]); tm = fmax(tm, s_r[1][w]); g_lo = fmin(g_lo, s_r[2][w]); g_hi = fmax(g_hi, s_r[3][w]); }
        double E = ts + tm + (g_hi > g_lo ? g_hi - g_lo : 0.0);
        // rounding of the fp64 evaluation above is ~1e-16 relative: the factor and the two extra units swamp it.  A NaN / inf
        // bound (non-finite descriptors) becomes the largest slack: every query of the segment then takes the exact path.
        const double units = ceil(E * ref.scale * (1.0 + 1e-9)) + 2.0;
        const unsigned eu = units < 5.0e8 ? (unsigned)units : 500000000u;
        const double s2 = ceil(2.0 * (double)eu / rho * (1.0 + 1e-9)) + 1.0;
        sc[z] = SegConst{cc, ref.fmin, ref.scale, ref.inv_scale, rho, eu, s2 < 2.0e9 ? (unsigned)s2 : 2000000000u};
    }
}

// rows under the REFERENCE constant -> packed u16 pairs [D2p][ldq] (quantize_pack_kernel's layout, zero padding)
__global__ __launch_bounds__(kBlock) void segp_quantize_ref_kernel(const double* __restrict__ P, const double* __restrict__ nref, int n, int D0, int Dp,
                                                                   const SegConst* __restrict__ ref, int D2p, int ldq, uint32_t* __restrict__ out) {
    __shared__ uint16_t tile[64][66];
    const int i0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    const SegConst c = *ref;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int i = i0 + ty + 4 * k, d = d0 + tx;
        unsigned q = 0u;
        if (i < n && d < Dp) {
            const double v = d < D0 ? P[(size_t)i * D0 + d] : c.cc;
            const double u = rint((v / nref[i] - c.fmin) * c.scale);
            q = (unsigned)fmin(fmax(u, 0.0), 65535.0);
        }
        tile[ty + 4 * k][tx] = (uint16_t)q;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int kk = ty + 4 * k, i = i0 + tx;
        if (i < ldq) out[(size_t)(d0 / 2 + kk) * ldq + i] = (uint32_t)tile[tx][2 * kk] | ((uint32_t)tile[tx][2 * kk + 1] << 16);
    }
}

// Candidate lists of every (segment, chunk of its rows, surface row) from the shared score matrix Sc [model row][ldsc]:
// one wave = 64 surface rows (lanes) x one chunk; the row of a model point is read as 256 contiguous bytes.
// part_* [segment][chunk][Q][KC] as sad16_candidates_kernel writes them (sorted, ascending row on ties, -1 / 0xFFFFFFFF empty).
__global__ __launch_bounds__(kBlock) void segp_select_kernel(const uint32_t* __restrict__ Sc, int ldsc, const int32_t* __restrict__ seg_rows,
                                                             const int32_t* __restrict__ seg_off, int Q, int chunk, int splits,
                                                             int32_t* __restrict__ part_idx, uint32_t* __restrict__ part_s) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int z = blockIdx.z, sidx = blockIdx.y * (kBlock / 64) + wave;
    if (sidx >= splits) return;
    const int off = seg_off[z], n = seg_off[z + 1] - off;
    const int32_t* rows = seg_rows + off;
    const int qi = blockIdx.x * 64 + lane;                     // < ldsc: the matrix is padded to whole query tiles
    const int b_begin = sidx * chunk, b_end = min(n, b_begin + chunk);
    CandU t;
#pragma unroll
    for (int k = 0; k < KC; ++k) { t.s[k] = 0xFFFFFFFFu; t.i[k] = -1; }
    int j = b_begin;
    for (; j + 8 <= b_end; j += 8) {
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = Sc[(size_t)rows[j + u] * ldsc + qi];
#pragma unroll
        for (int u = 0; u < 8; ++u) candu_insert(t, v[u], j + u);
    }
    for (; j < b_end; ++j) candu_insert(t, Sc[(size_t)rows[j] * ldsc + qi], j);
    if (qi < Q) {
        const size_t o = (((size_t)z * splits + sidx) * Q + qi) * KC;
#pragma unroll
        for (int k = 0; k < KC; ++k) { part_idx[o + k] = t.i[k]; part_s[o + k] = t.s[k]; }
    }
}

// a row of the segment's normalised matrices, never stored: value(d) = (d < D0 ? p[d] : cc) / nrm
struct SegRow { const double* p; double nrm, rinv; };
struct SegView { const double *PS, *PM, *nrmS, *nrmM, *rinvS, *rinvM; const int32_t* rows; int D0; double cc; };
__device__ __forceinline__ SegView seg_view(const SegSets& S, const double* nrmS, const double* nrmM, const SegConst& c, int z) {
    const int off = S.seg_off[z];
    nrmS += (size_t)z * S.Q; nrmM += off;
    return SegView{S.PS, S.PM, nrmS, nrmM, nrmS + S.rinv_s, nrmM + S.rinv_m, S.seg_rows + off, S.D0, c.cc};
}
__device__ __forceinline__ SegRow seg_surface_row(const SegView& V, int i) { return SegRow{V.PS + (size_t)i * V.D0, V.nrmS[i], V.rinvS[i]}; }
__device__ __forceinline__ SegRow seg_model_row(const SegView& V, int j) { return SegRow{V.PM + (size_t)V.rows[j] * V.D0, V.nrmM[j], V.rinvM[j]}; }
// value(d) = (d < D0 ? p[d] : cc) / nrm, the oracle's correctly rounded division.  An fp64 division is a ~12-instruction
// sequence, and the exact re-rank does two per term: with r = RN(1 / nrm), q = x r, e = fma(-nrm, q, x), q' = fma(e, r, q) is
// that same correctly rounded quotient (Markstein's correction step, what v_div_fmas does at the end of the hardware's own
// sequence; 900 000 random and adversarial cases -- all-ones mantissas, neighbours of powers of two -- against exact rational
// arithmetic without a difference) in three instructions, valid when no intermediate leaves the normal range: x and nrm in
// {0} U [2^-400, 2^400], which segp_rows / segp_consts check per row (rinv = 0 otherwise: the plain division).
__device__ __forceinline__ double seg_div(double x, const SegRow& r) {
    if (r.rinv != 0.0) {
        const double q = x * r.rinv;
        return fma(fma(-r.nrm, q, x), r.rinv, q);
    }
    return x / r.nrm;
}
__device__ __forceinline__ double seg_value(const SegView& V, const SegRow& r, int d) {
    const double v = r.p[min(d, V.D0 - 1)];
    return seg_div(d < V.D0 ? v : V.cc, r);
}
// the same without the test, for a loop that has established once which form all its rows take (a test per value put every
// load of the re-rank's tile behind its own branch: 5.9 ms where the plain division took 4.5)
template <bool FAST>
__device__ __forceinline__ double seg_value_as(const SegView& V, const SegRow& r, int d) {
    const double v = r.p[min(d, V.D0 - 1)];
    const double x = d < V.D0 ? v : V.cc;
    if (FAST) {
        const double q = x * r.rinv;
        return fma(fma(-r.nrm, q, x), r.rinv, q);
    }
    return x / r.nrm;
}

// sad16_finalize_kernel on the segments.  BACK = false: queries = surface rows, candidates = the segment's model rows
// (local numbers j).  BACK = true (the Unique back-search): queries = the candidates' model rows cand_m[k], k < n_cand,
// candidates = surface rows.  All per-query arrays are [segment][Q]...
// The exact re-rank of a query against a list of candidate rows (numbers local to the segment / surface rows), the oracle's
// order per candidate: the |a - b| terms of up to kNC candidates are computed by the whole wave into LDS, 256 features at a time,
// then lane c adds candidate c's terms one by one.  Returns the two smallest (distance, row), wave-uniform.
template <bool BACK>
__device__ __forceinline__ void seg_rerank(const SegView& V, const SegRow& a, const int* __restrict__ sj, int n_need, int D,
                                           double (*st)[kFT], double& d1, int& i1, double& d2, int& i2) {
    const int lane = threadIdx.x & 63;
    d1 = INFINITY; d2 = INFINITY; i1 = -1; i2 = -1;
    for (int g0 = 0; g0 < n_need; g0 += kNC) {
        const int nc = min(kNC, n_need - g0);
        double sum = 0.0;
        SegRow brow[kNC];                          // the group's rows, looked up once (index -> row number -> pointer and norm: two dependent
#pragma unroll                                     // loads that used to sit in front of every feature tile's loads)
        for (int cnd = 0; cnd < kNC; ++cnd) {
            const int jc = sj[min(g0 + cnd, n_need - 1)];
            brow[cnd] = BACK ? seg_surface_row(V, jc) : seg_model_row(V, jc);
        }
        bool all_fast = a.rinv != 0.0;             // every row of the group divides through its reciprocal (wave-uniform: the rows are)
#pragma unroll
        for (int cnd = 0; cnd < kNC; ++cnd) all_fast = all_fast && brow[cnd].rinv != 0.0;
        all_fast = __builtin_amdgcn_readfirstlane((int)all_fast) != 0;
        for (int d0 = 0; d0 < D; d0 += kFT) {
            double av[kFT / 64], bv[kNC][kFT / 64];
            int dof[kFT / 64];
#pragma unroll
            for (int u = 0; u < kFT / 64; ++u) dof[u] = min(d0 + lane + 64 * u, D - 1);
            if (all_fast) {
#pragma unroll
                for (int u = 0; u < kFT / 64; ++u) av[u] = seg_value_as<true>(V, a, dof[u]);
#pragma unroll
                for (int cnd = 0; cnd < kNC; ++cnd) {
#pragma unroll
                    for (int u = 0; u < kFT / 64; ++u) bv[cnd][u] = seg_value_as<true>(V, brow[cnd], dof[u]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < kFT / 64; ++u) av[u] = seg_value(V, a, dof[u]);
#pragma unroll
                for (int cnd = 0; cnd < kNC; ++cnd) {
#pragma unroll
                    for (int u = 0; u < kFT / 64; ++u) bv[cnd][u] = seg_value(V, brow[cnd], dof[u]);
                }
            }
#pragma unroll
            for (int cnd = 0; cnd < kNC; ++cnd)
#pragma unroll
                for (int u = 0; u < kFT / 64; ++u)
                    st[cnd][lane + 64 * u] = (d0 + lane + 64 * u < D) ? fabs(av[u] - bv[cnd][u]) : 0.0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (lane < nc) {
                // the oracle's order; terms past D are +0.0, which leaves a non-negative sum unchanged
                const double* t = st[lane];
                for (int k = 0; k < kFT; k += 16) {
                    double v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = t[k + u];
#pragma unroll
                    for (int u = 0; u < 16; ++u) sum += v[u];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
        {   // selects, not branches: as two conditional blocks the compiler stored through a SELECTED ADDRESS, which put the four
            // results in scratch memory for every kernel that calls this (1.7 GB of scratch writes per sweep in round 3's finalize)
            const int jc = sj[min(g0 + lane, n_need - 1)];
            const bool in = lane < nc, b1 = in && lexd_lt(sum, jc, d1, i1), b2 = in && !b1 && lexd_lt(sum, jc, d2, i2);
            const double nd2 = b1 ? d1 : (b2 ? sum : d2); const int ni2 = b1 ? i1 : (b2 ? jc : i2);
            d1 = b1 ? sum : d1; i1 = b1 ? jc : i1; d2 = nd2; i2 = ni2;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double e1 = __shfl_xor(d1, o), e2 = __shfl_xor(d2, o);
        int k1 = __shfl_xor(i1, o), k2 = __shfl_xor(i2, o);
        bool fm = lexd_lt(d1, i1, e1, k1);
        double w1 = fm ? d1 : e1; int x1 = fm ? i1 : k1;
        double m2 = fm ? d2 : d1; int y2 = fm ? i2 : i1;
        double o2 = fm ? e1 : e2; int z2 = fm ? k1 : k2;
        bool sm = lexd_lt(m2, y2, o2, z2);
        d1 = w1; i1 = x1; d2 = sm ? m2 : o2; i2 = sm ? y2 : z2;
    }
}

template <bool BACK>
__global__ __launch_bounds__(kBlock) void segp_finalize_kernel(SegSets S, const double* __restrict__ nrmS, const double* __restrict__ nrmM,
                                                               const SegConst* __restrict__ sc, const int32_t* __restrict__ cand_m,
                                                               const int32_t* __restrict__ n_live, const int32_t* __restrict__ part_idx,
                                                               const uint32_t* __restrict__ part_s, int splits,
                                                               int32_t* __restrict__ idx, double* __restrict__ dist,
                                                               int32_t* __restrict__ flag_list, int32_t* __restrict__ n_flag, int force_unproven,
                                                               double match_thr, double max_ratio, int32_t* __restrict__ dbg_hist,
                                                               unsigned long long* __restrict__ stats) {
    __shared__ double s_t[kBlock / 64][kNC][kFT];
    __shared__ int s_j[kBlock / 64][64 * kEPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int z = blockIdx.z, nA = S.Q, D = S.Dp;
    const int qi = blockIdx.x * (kBlock / 64) + wave;
    if (qi >= nA || (BACK && qi >= n_live[z])) return;        // wave-uniform; no block barrier below
    const SegConst c = sc[z];
    const SegView V = seg_view(S, nrmS, nrmM, c, z);
    part_idx += (size_t)z * splits * nA * KC; part_s += (size_t)z * splits * nA * KC;
    idx += (size_t)z * nA * 2; dist += (size_t)z * nA * 2; flag_list += (size_t)z * nA;
    const int total = splits * KC;
    int j[kEPL]; unsigned sq[kEPL];
    unsigned a1 = 0xFFFFFFFFu, a2 = 0xFFFFFFFFu, g = 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < kEPL; ++u) {
        const int e = lane + 64 * u;
        j[u] = -1; sq[u] = 0xFFFFFFFFu;
        if (e < total) { size_t o = ((size_t)(e / KC) * nA + qi) * KC + (e % KC); j[u] = part_idx[o]; sq[u] = part_s[o]; }
        if (j[u] >= 0) {
            if (sq[u] < a1) { a2 = a1; a1 = sq[u]; } else if (sq[u] < a2) a2 = sq[u];
            if ((e % KC) == KC - 1) g = min(g, sq[u]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned b1 = __shfl_xor(a1, o), b2 = __shfl_xor(a2, o);
        unsigned n1 = min(a1, b1), n2 = min(max(a1, b1), min(a2, b2));
        a1 = n1; a2 = n2;
        g = min(g, (unsigned)__shfl_xor((int)g, o));
    }
    // A query that cannot pass filter_keep needs no exact sums.  Every row's exact distance lies within
    // [rho (s - (D + 1)) - E, rho (s + (D + 1)) + E] / scale of its reference score s, so with the two smallest scores a1 <= a2:
    // d1 >= lower(a1) and d2 <= upper(a2) (two rows at least are that close).  No pair if lower(a1) > thr (d1 <= thr fails) or if
    // lower(a1) / upper(a2) > maxRatio (the ratio test fails: most queries of a sphere that does not hold their partner -- their
    // nearest and second-nearest rows are random neighbours a fraction of a percent apart).
    if (!BACK && !force_unproven && a1 != 0xFFFFFFFFu) {
        const double lower1 = (c.rho * ((double)a1 - (double)(D + 1)) - (double)c.eunits) * c.inv_scale * (1.0 - 1e-12);
        bool none = lower1 > match_thr;
        if (!none && a2 != 0xFFFFFFFFu && S.seg_off[z + 1] - S.seg_off[z] > 1) {
            const double upper2 = (c.rho * ((double)a2 + (double)(D + 1)) + (double)c.eunits) * c.inv_scale * (1.0 + 1e-12);
            none = upper2 >= 1e-6 && lower1 > max_ratio * upper2 * (1.0 + 1e-12);
        }
        if (none) {                                                                 // wave-uniform
            if (lane == 0) { idx[(size_t)qi * 2] = -1; idx[(size_t)qi * 2 + 1] = -1; dist[(size_t)qi * 2] = INFINITY; dist[(size_t)qi * 2 + 1] = INFINITY; }
            return;
        }
    }
    // the scores were taken under the reference constant: this segment's true distances lie within E_i (c.eunits) of rho_i times them
    const unsigned slack = 2u * (unsigned)(D + 1) + 2u + c.slack2;
    int n_need = 0;
#pragma unroll
    for (int u = 0; u < kEPL; ++u) {
        const bool need = j[u] >= 0 && (a2 == 0xFFFFFFFFu || a2 > 0xFFFFFFFFu - slack || sq[u] <= a2 + slack);
        const unsigned long long m = __ballot(need);
        if (need) s_j[wave][n_need + __popcll(m & ((1ull << lane) - 1ull))] = j[u];
        n_need += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();

#ifdef PCREG_EXPERIMENTS
    if (dbg_hist && lane == 0) atomicAdd(&dbg_hist[min(n_need, 129)], 1);          // PCREG_SEG_DEBUG: candidates re-scored per query
#endif
    const SegRow a = BACK ? seg_model_row(V, cand_m[(size_t)z * nA + qi]) : seg_surface_row(V, qi);
    double d1, d2; int i1, i2;
    if (n_need > kNC) {
        // More candidates than one group holds (5.3 on average at the sweep's shape, inside a slack that must allow the rounding of
        // BOTH distances compared): sum the kNC with the smallest reference scores first.  Their exact second distance d2' bounds the
        // true one from above, and a remaining candidate can only matter if its own lower bound does not exceed d2' -- a one-sided
        // test that usually leaves nothing for a second group.
        int jk[kEPL]; unsigned sk[kEPL];
#pragma unroll
        for (int u = 0; u < kEPL; ++u) {
            const bool need = j[u] >= 0 && (a2 == 0xFFFFFFFFu || a2 > 0xFFFFFFFFu - slack || sq[u] <= a2 + slack);
            jk[u] = need ? j[u] : -1; sk[u] = need ? sq[u] : 0xFFFFFFFFu;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < kNC; ++r) {                       // kNC rounds of a wave-wide (score, entry) minimum
            unsigned long long key = 0xFFFFFFFFFFFFFFFFull;
#pragma unroll
            for (int u = 0; u < kEPL; ++u) if (jk[u] >= 0) { const unsigned long long k2 = ((unsigned long long)sk[u] << 32) | (unsigned)(lane + 64 * u); key = k2 < key ? k2 : key; }
            unsigned long long m = key;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned long long x = __shfl_xor(m, o); m = x < m ? x : m; }
            const int e = (int)(unsigned)(m & 0xFFFFFFFFull);          // the winning entry (lane + 64 u): unique
#pragma unroll
            for (int u = 0; u < kEPL; ++u) if (lane + 64 * u == e && jk[u] >= 0) { s_j[wave][r] = jk[u]; jk[u] = -1; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        seg_rerank<BACK>(V, a, s_j[wave], kNC, D, s_t[wave], d1, i1, d2, i2);
        unsigned smax = 0xFFFFFFFFu;
        if (i2 >= 0 && d2 < INFINITY) {
            const double t = (d2 * c.scale * (1.0 + 1e-12) + (double)c.eunits) / c.rho + (double)(D + 1) + 1.0;
            if (t < 4.0e9) smax = (unsigned)t;
        }
        int n_more = 0;
#pragma unroll
        for (int u = 0; u < kEPL; ++u) {
            const bool more = jk[u] >= 0 && sk[u] <= smax;
            const unsigned long long m = __ballot(more);
            if (more) s_j[wave][n_more + __popcll(m & ((1ull << lane) - 1ull))] = jk[u];
            n_more += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (n_more > 0) {
            double e1, e2; int k1, k2;
            seg_rerank<BACK>(V, a, s_j[wave], n_more, D, s_t[wave], e1, k1, e2, k2);
            Top2T<double> t2{d1, d2, i1, i2};
            top2_insert_lex_t(t2, e1, k1); top2_insert_lex_t(t2, e2, k2);
            d1 = t2.d1; d2 = t2.d2; i1 = t2.i1; i2 = t2.i2;
        }
    } else {
        seg_rerank<BACK>(V, a, s_j[wave], n_need, D, s_t[wave], d1, i1, d2, i2);
    }
    bool ok;
    if (g == 0xFFFFFFFFu) ok = true;
    else {
        double lower = (c.rho * ((double)g - (double)(D + 1)) - (double)c.eunits) * c.inv_scale;
        ok = (i2 >= 0) && (lower * (1.0 - 1e-12) > d2);
    }
    if (force_unproven) ok = false;
    if (lane == 0) {          // an unproven query keeps its provisional pair: segp_refine_kernel starts from it
        idx[(size_t)qi * 2] = i1; idx[(size_t)qi * 2 + 1] = i2; dist[(size_t)qi * 2] = d1; dist[(size_t)qi * 2 + 1] = d2;
        if (!ok) { int slot = atomicAdd(&n_flag[z], 1); flag_list[slot] = qi; }
        if (stats) { atomicAdd(&stats[0], 1ull); atomicAdd(&stats[1], (unsigned long long)n_need); if (!ok) atomicAdd(&stats[2], 1ull); }
    }
}

// ---- the forward re-rank as a dense chain (round 4) -----------------------------------------------------------------------
// segp_finalize_kernel gives a WAVE to each query, and in its summing phase kNC = 4 lanes of the 64 add (the oracle's order is one
// chain of D adds per candidate): 4.2 of the sweep's 11.7 ms, half of the vector cycles spent with 60 lanes masked off.  Here
//   segp_pick_kernel    (a wave per query) reads the query's lists, drops it if no pair can pass the filter, and writes up to
//                       TWO groups of kNC candidates (the 8 smallest reference scores inside the slack): rows, norms, numbers;
//   segp_scan_kernel    (a workgroup per segment) numbers the groups a round sums: slot -> group;
//   segp_rerank_pairs_kernel (a wave per kRQ = 4 groups = 16 (query, candidate) pairs) forms the a - b terms of a 64-feature tile
//                       pair by pair -- lane = feature, the norms in scalar registers -- into LDS [pair][feature], then lane p adds
//                       pair p's 64 |terms| in order: every lane sums;
//   segp_decide_kernel  (a THREAD per query) takes the exact distances, applies the certificate, writes idx / dist / the unproven
//                       list.  TWO rounds of scan / pairs / decide: round 1 sums every query's first group, and the exact second
//                       distance of those four excludes what it can of the second group through the reference scores (round 3's
//                       bound); round 2 sums the second groups that keep a candidate.  The few queries with more than 8
//                       candidates inside the slack are listed for
//   segp_decide_more_kernel (a wave per such query): the remaining candidates that the exact second distance does not exclude
//                       are summed by seg_rerank as before.
// The sums are the same chains of the same terms; the top two of ALL candidates inside the slack equal the top two of
// segp_finalize_kernel's "first four, then whoever the bound leaves" because that bound only drops rows that are strictly
// farther than the second distance.  segp_finalize_kernel<false> stays as the reference ("seg_wave_finalize").
constexpr int kNG = 2;                                                            // groups per query
struct alignas(32) SegGroupRows { int32_t arow, fast, row[kNC], pad[2]; };          // the a-side row, the kNC b-side rows (row 0 where there is no candidate); fast: all six divide through rinv
struct alignas(16) SegGroupNorms { double nrm_a, rinv_a, nrm[kNC], rinv[kNC]; };  // of the surface row and the kNC rows (1, 1 where none)
struct alignas(16) SegQueryInfo {
    int32_t j[kNG * kNC];          // the candidates' local numbers, < 0: none (or excluded by the first group's exact second distance)
    uint32_t g;                    // seg_front's
    int32_t ng, n_need, fast;      // groups that hold candidates
    uint32_t sc2[kNC];             // reference scores of the second group's candidates
    int32_t r2, pad[3];            // the second group is to be summed (set by the first decide launch)
};

struct SegFront { int j[kEPL]; unsigned sq[kEPL]; unsigned a1, a2, g; };
// a query's candidate lists (entry lane + 64 u), the two smallest reference scores and the smallest KC-th score of a chunk
__device__ __forceinline__ SegFront seg_front(const int32_t* __restrict__ part_idx, const uint32_t* __restrict__ part_s, int splits, int nA, int qi, int lane) {
    SegFront F;
    const int total = splits * KC;
    F.a1 = 0xFFFFFFFFu; F.a2 = 0xFFFFFFFFu; F.g = 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < kEPL; ++u) {
        const int e = lane + 64 * u;
        F.j[u] = -1; F.sq[u] = 0xFFFFFFFFu;
        if (e < total) { size_t o = ((size_t)(e / KC) * nA + qi) * KC + (e % KC); F.j[u] = part_idx[o]; F.sq[u] = part_s[o]; }
        if (F.j[u] >= 0) {
            if (F.sq[u] < F.a1) { F.a2 = F.a1; F.a1 = F.sq[u]; } else if (F.sq[u] < F.a2) F.a2 = F.sq[u];
            if ((e % KC) == KC - 1) F.g = min(F.g, F.sq[u]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned b1 = __shfl_xor(F.a1, o), b2 = __shfl_xor(F.a2, o);
        unsigned n1 = min(F.a1, b1), n2 = min(max(F.a1, b1), min(F.a2, b2));
        F.a1 = n1; F.a2 = n2;
        F.g = min(F.g, (unsigned)__shfl_xor((int)F.g, o));
    }
    return F;
}
__device__ __forceinline__ bool seg_front_need(const SegFront& F, int u, unsigned slack) {
    return F.j[u] >= 0 && (F.a2 == 0xFFFFFFFFu || F.a2 > 0xFFFFFFFFu - slack || F.sq[u] <= F.a2 + slack);
}
// kNG kNC rounds of a wave-wide (score, entry) minimum over the entries still in jk: the winners leave jk (-> -1) and, if out is
// given, are written to out[0 .. kNG kNC) in the order found.  Deterministic: the same lists give the same picks in every kernel.
__device__ __forceinline__ void seg_pick_first(int (&jk)[kEPL], const unsigned (&sk)[kEPL], int lane, int* out) {
#pragma unroll
    for (int r = 0; r < kNG * kNC; ++r) {
        unsigned long long key = 0xFFFFFFFFFFFFFFFFull;
#pragma unroll
        for (int u = 0; u < kEPL; ++u) if (jk[u] >= 0) { const unsigned long long k2 = ((unsigned long long)sk[u] << 32) | (unsigned)(lane + 64 * u); key = k2 < key ? k2 : key; }
        unsigned long long m = key;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long x = __shfl_xor(m, o); m = x < m ? x : m; }
        if (m == 0xFFFFFFFFFFFFFFFFull) break;                  // nobody left (wave-uniform)
        const int e = (int)(unsigned)(m & 0xFFFFFFFFull);          // the winning entry (lane + 64 u): unique
#pragma unroll
        for (int u = 0; u < kEPL; ++u) if (lane + 64 * u == e && jk[u] >= 0) { if (out) out[r] = jk[u]; jk[u] = -1; }
    }
}

__global__ __launch_bounds__(kBlock) void segp_pick_kernel(SegSets S, const double* __restrict__ nrmS, const double* __restrict__ nrmM,
                                                           const SegConst* __restrict__ sc, const int32_t* __restrict__ part_idx,
                                                           const uint32_t* __restrict__ part_s, int splits, int32_t* __restrict__ idx,
                                                           double* __restrict__ dist, int force_unproven, double match_thr, double max_ratio,
                                                           SegGroupRows* __restrict__ grows, SegGroupNorms* __restrict__ gnorms,
                                                           SegQueryInfo* __restrict__ qinfo) {
    __shared__ int s_j[kBlock / 64][kNG * kNC];
    __shared__ unsigned s_s[kBlock / 64][kNG * kNC];
    __shared__ unsigned long long s_key[kBlock / 64][64 * kEPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int z = blockIdx.z, nA = S.Q, D = S.Dp;
    const int qi = blockIdx.x * (kBlock / 64) + wave;
    if (qi >= nA) return;                                      // wave-uniform; no block barrier below
    const SegConst c = sc[z];
    const SegView V = seg_view(S, nrmS, nrmM, c, z);
    part_idx += (size_t)z * splits * nA * KC; part_s += (size_t)z * splits * nA * KC;
    idx += (size_t)z * nA * 2; dist += (size_t)z * nA * 2;
    const size_t o = (size_t)z * nA + qi;
    const SegFront F = seg_front(part_idx, part_s, splits, nA, qi, lane);
    // (segp_finalize_kernel's comment) a query that cannot pass filter_keep needs no exact sums; neither does one without candidates
    bool none = F.a1 == 0xFFFFFFFFu;
    if (!force_unproven && !none) {
        const double lower1 = (c.rho * ((double)F.a1 - (double)(D + 1)) - (double)c.eunits) * c.inv_scale * (1.0 - 1e-12);
        none = lower1 > match_thr;
        if (!none && F.a2 != 0xFFFFFFFFu && S.seg_off[z + 1] - S.seg_off[z] > 1) {
            const double upper2 = (c.rho * ((double)F.a2 + (double)(D + 1)) + (double)c.eunits) * c.inv_scale * (1.0 + 1e-12);
            none = upper2 >= 1e-6 && lower1 > max_ratio * upper2 * (1.0 + 1e-12);
        }
    }
    if (none) {                                                                 // wave-uniform
        if (lane == 0) {
            idx[(size_t)qi * 2] = -1; idx[(size_t)qi * 2 + 1] = -1; dist[(size_t)qi * 2] = INFINITY; dist[(size_t)qi * 2 + 1] = INFINITY;
            qinfo[o].ng = 0;
        }
        return;
    }
    const unsigned slack = 2u * (unsigned)(D + 1) + 2u + c.slack2;
    int jk[kEPL]; unsigned sk[kEPL];
    int n_need = 0;
#pragma unroll
    for (int u = 0; u < kEPL; ++u) {
        const bool need = seg_front_need(F, u, slack);
        jk[u] = need ? F.j[u] : -1; sk[u] = need ? F.sq[u] : 0xFFFFFFFFu;
        n_need += __popcll(__ballot(need));
    }
    // The kNG kNC smallest (score, entry) keys in ascending order = seg_pick_first's picks, without its rounds of wave-wide minima
    // (8 x 6 cross-lane steps of 64 bits: two thirds of this kernel's instructions): the needed entries are compacted into LDS
    // (5 of them on average) and each looks up its rank among them.
    if (lane < kNG * kNC) s_j[wave][lane] = -1;
    int pos[kEPL];
    {
        int n0 = 0;
#pragma unroll
        for (int u = 0; u < kEPL; ++u) {
            const unsigned long long m = __ballot(jk[u] >= 0);
            pos[u] = n0 + __popcll(m & ((1ull << lane) - 1ull));
            if (jk[u] >= 0) s_key[wave][pos[u]] = ((unsigned long long)sk[u] << 32) | (unsigned)(lane + 64 * u);
            n0 += __popcll(m);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < kEPL; ++u) {
        if (jk[u] >= 0) {
            const unsigned long long mine = ((unsigned long long)sk[u] << 32) | (unsigned)(lane + 64 * u);
            int rank = 0;
            for (int k = 0; k < n_need; ++k) rank += s_key[wave][k] < mine ? 1 : 0;
            if (rank < kNG * kNC) { s_j[wave][rank] = jk[u]; s_s[wave][rank] = sk[u]; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    const int ng = n_need > kNC ? 2 : 1;
    const SegRow a = seg_surface_row(V, qi);
    bool fast = a.rinv != 0.0;
    if (lane < kNG * kNC) {
        const int jc = s_j[wave][lane], gq = lane / kNC, cq = lane % kNC;
        int row = 0; double nrm = 1.0, rinv = 1.0;              // no candidate: row 0 (readable), nobody looks at its sum
        if (jc >= 0) { row = V.rows[jc]; nrm = V.nrmM[jc]; rinv = V.rinvM[jc]; fast = fast && rinv != 0.0; }
        qinfo[o].j[lane] = jc;
        if (gq == 1) qinfo[o].sc2[cq] = jc >= 0 ? s_s[wave][lane] : 0xFFFFFFFFu;
        if (gq < ng) {
            grows[o * kNG + gq].row[cq] = row; gnorms[o * kNG + gq].nrm[cq] = nrm; gnorms[o * kNG + gq].rinv[cq] = rinv;
            if (cq == 0) { gnorms[o * kNG + gq].nrm_a = a.nrm; gnorms[o * kNG + gq].rinv_a = a.rinv; }
        }
    }
    fast = __ballot(fast) == ~0ull;                            // lanes >= 8 carry the surface row's flag
    if (lane < ng) { grows[o * kNG + lane].arow = qi; grows[o * kNG + lane].fast = fast ? 1 : 0; }
    if (lane == 0) { qinfo[o].g = F.g; qinfo[o].ng = ng; qinfo[o].n_need = n_need; qinfo[o].fast = fast ? 1 : 0; qinfo[o].r2 = 0; }
}

// slot -> group (2 qi + g) for the groups that exist, in ascending order; n_slots[z] = their number.  One workgroup per segment.
// which: 0 = every group, 1 = the first group of every item that has one, 2 = the second groups marked r2
__global__ __launch_bounds__(kBlock) void segp_scan_kernel(const SegQueryInfo* __restrict__ qinfo, int Q, const int32_t* __restrict__ n_items, int which,
                                                           int32_t* __restrict__ map, int32_t* __restrict__ n_slots) {
    __shared__ int s_w[kBlock / 64], s_base;
    const int z = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    qinfo += (size_t)z * Q; map += (size_t)z * Q * kNG;
    const int n = n_items ? min(n_items[z], Q) : Q;            // items 0 .. n of the segment carry an ng
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int q0 = 0; q0 < n; q0 += kBlock) {
        const int qi = q0 + tid;
        int ng = qi < n ? qinfo[qi].ng : 0;
        const int g0 = which == 2 ? 1 : 0;
        if (which == 1) ng = min(ng, 1);
        else if (which == 2) ng = (ng == 2 && qinfo[qi].r2) ? 1 : 0;
        int incl = ng;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int base = s_base;
        for (int w = 0; w < wave; ++w) base += s_w[w];
        const int first = base + incl - ng;
        for (int g = 0; g < ng; ++g) map[first + g] = qi * kNG + g0 + g;
        __syncthreads();
        if (tid == kBlock - 1) s_base = base + incl;
        __syncthreads();
    }
    if (tid == 0) n_slots[z] = s_base;
}

constexpr int kRT = 64;          // features per tile of the pairs kernel
#ifndef PCREG_SEG_RQ
#define PCREG_SEG_RQ 4          // same-box A/B of the forward launch: 8 groups per wave 2.11 ms (2 waves per SIMD), 6: 2.12, 4: 1.98 (3 waves per SIMD)
#endif
#ifndef PCREG_SEG_RW
#define PCREG_SEG_RW 4
#endif
constexpr int kRQ = PCREG_SEG_RQ, kRP = kRQ * kNC, kRW = PCREG_SEG_RW;          // groups, pairs per wave; waves per workgroup
// One wave = kRQ groups of a segment = kRP pairs.  The grid is one-dimensional and XCD-aware: the hardware deals consecutive
// workgroups to the 8 XCDs in turn, so workgroup L works for segment 8 (L / 8 / nb) + L % 8 -- all batches of a segment run on ONE
// XCD, whose L2 then holds that sphere's rows (each is a candidate of ~6 queries) instead of an eighth of eight spheres'.
template <bool FAST>
__device__ __forceinline__ double seg_quot(double x, double nrm, double rinv) {
    if (FAST) { const double q = x * rinv; return fma(fma(-nrm, q, x), rinv, q); }
    return x / nrm;
}
typedef const double __attribute__((address_space(1)))* SegGlobalPtr;          // global, not flat: a flat load counts on both wait counters and
                                                                               // makes the tile wait for ALL its loads before the first is used
// One tile: ALL the tile's loads first (5 per group: the latency of one round trip per tile, not per group -- the LDS tile leaves
// two waves per SIMD, so the loads in flight have to come from the wave itself), then the differences a - b into LDS
// [pair][feature]; the summing lane takes the absolute value (a free operand modifier there).  Every lane keeps the running
// address of its feature in each of the 5 kRQ rows (wave-uniform row, + 512 bytes per tile): one vector add per load and no scalar
// load in front of any of them.  The loads of tile t + 1 are issued as soon as tile t's differences are in LDS -- the registers are
// free then -- so their latency runs under the summing phase's chain of 64 dependent adds.
template <bool TAIL>
__device__ __forceinline__ void seg_pairs_load(SegGlobalPtr (&pa)[kRQ], SegGlobalPtr (&pb)[kRQ][kNC], double (&xa)[kRQ], double (&xb)[kRQ][kNC], int D0, int d0, int lane) {
    const int back = TAIL ? max(d0 + lane - (D0 - 1), 0) : 0;                       // past the row's end: re-read its last value (unused)
#pragma unroll
    for (int q = 0; q < kRQ; ++q) {
        xa[q] = *(pa[q] - back); pa[q] += kRT;
#pragma unroll
        for (int c = 0; c < kNC; ++c) { xb[q][c] = *(pb[q][c] - back); pb[q][c] += kRT; }
    }
}
// the terms of tile d0 from the values in (xa, xb); NEXT = 1 / 2: group by group, as soon as a group's values are used, its
// loads for the tile behind (2: that tile reaches past the row's end) go out into the same registers
template <bool FAST, bool TAIL, int NEXT>
__device__ __forceinline__ void seg_pairs_terms(SegGlobalPtr (&pa)[kRQ], SegGlobalPtr (&pb)[kRQ][kNC], double (&xa)[kRQ], double (&xb)[kRQ][kNC],
                                                const SegGroupNorms* __restrict__ gn, const int (&code)[kRQ],
                                                int D0, int Dp, double cc, int d0, int lane, double (*st)[kRT + 2]) {
    static_assert(kNC == 4, "the pair tile is written for four candidates per group");
    const int d = d0 + lane;
    const bool is_cc = TAIL && d >= D0, valid = !TAIL || d < Dp;
    const int back = NEXT == 2 ? max(d + kRT - (D0 - 1), 0) : 0;
    SegGroupNorms N = gn[code[0]];
#pragma unroll
    for (int q = 0; q < kRQ; ++q) {
        const SegGroupNorms Nn = gn[code[min(q + 1, kRQ - 1)]];                     // the next group's norms are on their way while this one computes
        // (a short batch repeats its last group into LDS rows nobody sums -- straight-line code)
        if (is_cc) { xa[q] = cc; xb[q][0] = cc; xb[q][1] = cc; xb[q][2] = cc; xb[q][3] = cc; }
        const double av = seg_quot<FAST>(xa[q], N.nrm_a, N.rinv_a);
        double t[kNC];
#pragma unroll
        for (int c = 0; c < kNC; ++c) t[c] = av - seg_quot<FAST>(xb[q][c], N.nrm[c], N.rinv[c]);
        if (NEXT) {
            xa[q] = *(pa[q] - back); pa[q] += kRT;
#pragma unroll
            for (int c = 0; c < kNC; ++c) { xb[q][c] = *(pb[q][c] - back); pb[q][c] += kRT; }
        }
#pragma unroll
        for (int c = 0; c < kNC; ++c) st[q * kNC + c][lane] = valid ? t[c] : 0.0;
        N = Nn;
        __builtin_amdgcn_sched_barrier(0);          // the scheduler would hoist all 40 loads to the top of the tile: twice the registers, spills
    }
}
__global__ __launch_bounds__(64 * kRW, kRQ <= 4 ? 3 : 2) void segp_rerank_pairs_kernel(const double* __restrict__ A, const double* __restrict__ B, int nA, int D0, int Dp,
                                                                     const SegConst* __restrict__ sc, const SegGroupRows* __restrict__ grows,
                                                                     const SegGroupNorms* __restrict__ gnorms,
                                                                     const int32_t* __restrict__ map, const int32_t* __restrict__ n_slots,
                                                                     int nb, int n_seg, double* __restrict__ psum) {
    __shared__ __attribute__((aligned(16))) double s_t[kRW][kRP][kRT + 2];            // rows of 528 bytes: b128 reads of 16 lanes cover the 64 banks once
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int L = blockIdx.x, xcd = L & 7, r = L >> 3;
    const int z = (r / nb) * 8 + xcd, bx = r % nb;
    if (z >= n_seg) return;
    const int ns = min(n_slots[z], nA * kNG);                  // nA: items per segment (the arrays hold kNG groups for each)
    const int s0 = (bx * kRW + wave) * kRQ;
    if (s0 >= ns) return;                                      // wave-uniform; no block barrier below
    const int nsb = min(kRQ, ns - s0);
    map += (size_t)z * nA * kNG + s0;
    grows += (size_t)z * nA * kNG; const SegGroupNorms* gn = gnorms + (size_t)z * nA * kNG;
    psum += (size_t)z * nA * kNG * kNC;
    const double cc = sc[z].cc;
    double (*st)[kRT + 2] = s_t[wave];
    bool fast = true;
    int code[kRQ];
    SegGlobalPtr pa[kRQ], pb[kRQ][kNC];
#pragma unroll
    for (int q = 0; q < kRQ; ++q) {
        code[q] = map[min(q, nsb - 1)];                        // a short batch repeats its last group (loads only: nothing is stored for it)
        const SegGroupRows R = grows[code[q]];
        fast = fast && R.fast != 0;
        pa[q] = (SegGlobalPtr)(A + (unsigned long long)(unsigned)R.arow * (unsigned)D0 + lane);
#pragma unroll
        for (int c = 0; c < kNC; ++c) pb[q][c] = (SegGlobalPtr)(B + (unsigned long long)(unsigned)R.row[c] * (unsigned)D0 + lane);
    }
    double sum = 0.0;                                          // lane p: pair p = (group p / kNC, candidate p % kNC)
    double xa[kRQ], xb[kRQ][kNC];
    auto add_tile = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (lane < nsb * kNC) {
            // the oracle's order; terms past D are +0.0, which leaves a non-negative sum unchanged
            const double* t = st[lane];
#pragma unroll 2
            for (int k = 0; k < kRT; k += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = t[k + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) sum += fabs(v[u]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    };
    // tiles 0 .. n_full - 1 lie inside the rows (no clamps, no selects), one more holds the rows' end and the appended constant
    const int n_full = D0 / kRT, has_tail = Dp > n_full * kRT;
    auto run = [&](auto fastc) {
        constexpr bool F = decltype(fastc)::value;
        if (n_full == 0) seg_pairs_load<true>(pa, pb, xa, xb, D0, 0, lane); else seg_pairs_load<false>(pa, pb, xa, xb, D0, 0, lane);
        for (int t = 0; t + 1 < n_full; ++t) { seg_pairs_terms<F, false, 1>(pa, pb, xa, xb, gn, code, D0, Dp, cc, t * kRT, lane, st); add_tile(); }
        if (n_full > 0) {
            if (has_tail) seg_pairs_terms<F, false, 2>(pa, pb, xa, xb, gn, code, D0, Dp, cc, (n_full - 1) * kRT, lane, st);
            else seg_pairs_terms<F, false, 0>(pa, pb, xa, xb, gn, code, D0, Dp, cc, (n_full - 1) * kRT, lane, st);
            add_tile();
        }
        if (has_tail) { seg_pairs_terms<F, true, 0>(pa, pb, xa, xb, gn, code, D0, Dp, cc, n_full * kRT, lane, st); add_tile(); }
    };
    if (fast) run(std::true_type{}); else run(std::false_type{});
    if (lane < nsb * kNC) {
        const int cd = map[lane / kNC];                        // (per-lane: the group this pair belongs to)
        psum[(size_t)cd * kNC + lane % kNC] = sum;
    }
}

// the certificate of a query whose candidates inside the slack have all been summed (seg_front's g: the smallest KC-th score of a chunk)
__device__ __forceinline__ bool seg_certified(const SegConst& c, unsigned g, int D, const Top2T<double>& t2) {
    if (g == 0xFFFFFFFFu) return true;
    const double lower = (c.rho * ((double)g - (double)(D + 1)) - (double)c.eunits) * c.inv_scale;
    return t2.i2 >= 0 && lower * (1.0 - 1e-12) > t2.d2;
}
// round 1 (after the first groups are summed): a query with one group is decided; a query with two has the exact second distance of
// its first four candidates bound the other four -- a candidate whose reference score cannot lie below it is dropped (j <- -1) --
// and, if any is left, waits for round 2 (r2 <- 1: segp_scan_kernel(which = 2) lists its second group).  round 2: the queries with r2.
__global__ __launch_bounds__(kBlock) void segp_decide_kernel(const SegConst* __restrict__ sc, SegQueryInfo* __restrict__ qinfo, const double* __restrict__ psum,
                                                             int Q, int D, int round, int32_t* __restrict__ idx, double* __restrict__ dist,
                                                             int32_t* __restrict__ flag_list, int32_t* __restrict__ n_flag,
                                                             int32_t* __restrict__ more_list, int32_t* __restrict__ n_more, int force_unproven,
                                                             unsigned long long* __restrict__ stats) {
    const int z = blockIdx.z, qi = blockIdx.x * kBlock + threadIdx.x;
    if (qi >= Q) return;
    const size_t o = (size_t)z * Q + qi;
    SegQueryInfo I = qinfo[o];
    if (I.ng == 0) return;                                     // segp_pick_kernel wrote "no pair"
    if (round == 2 && !I.r2) return;                           // decided in round 1
    const SegConst c = sc[z];
    Top2T<double> t2{INFINITY, INFINITY, -1, -1};
#pragma unroll
    for (int r = 0; r < kNC; ++r) top2_insert_lex_t(t2, psum[o * kNG * kNC + r], I.j[r]);          // j < 0: no candidate
    if (round == 1 && I.ng == 2) {
        unsigned smax = 0xFFFFFFFFu;
        if (t2.i2 >= 0 && t2.d2 < INFINITY) {
            const double t = (t2.d2 * c.scale * (1.0 + 1e-12) + (double)c.eunits) / c.rho + (double)(D + 1) + 1.0;
            if (t < 4.0e9) smax = (unsigned)t;
        }
        bool any = false;
#pragma unroll
        for (int r = 0; r < kNC; ++r) {
            const bool keep = I.j[kNC + r] >= 0 && I.sc2[r] <= smax;
            if (!keep) { I.j[kNC + r] = -1; qinfo[o].j[kNC + r] = -1; }
            any = any || keep;
        }
        if (any) { qinfo[o].r2 = 1; return; }
    }
    if (round == 2) {
#pragma unroll
        for (int r = kNC; r < kNG * kNC; ++r) top2_insert_lex_t(t2, psum[o * kNG * kNC + r], I.j[r]);
    }
    if (I.n_need > kNG * kNC) { more_list[(size_t)z * Q + atomicAdd(&n_more[z], 1)] = qi; return; }
    const bool ok = !force_unproven && seg_certified(c, I.g, D, t2);
    idx[o * 2] = t2.i1; idx[o * 2 + 1] = t2.i2; dist[o * 2] = t2.d1; dist[o * 2 + 1] = t2.d2;          // unproven: the provisional pair segp_refine_kernel starts from
    if (!ok) flag_list[(size_t)z * Q + atomicAdd(&n_flag[z], 1)] = qi;
    if (stats) { atomicAdd(&stats[0], 1ull); atomicAdd(&stats[1], (unsigned long long)I.n_need); if (!ok) atomicAdd(&stats[2], 1ull); }
}
// a query with more candidates inside the slack than the two groups hold: the exact second distance of the 8 summed bounds the
// true one from above, and a remaining candidate matters only if its own lower bound does not exceed it
__global__ __launch_bounds__(kBlock) void segp_decide_more_kernel(SegSets S, const double* __restrict__ nrmS, const double* __restrict__ nrmM,
                                                                  const SegConst* __restrict__ sc, const int32_t* __restrict__ part_idx,
                                                                  const uint32_t* __restrict__ part_s, int splits, const SegQueryInfo* __restrict__ qinfo,
                                                                  const double* __restrict__ psum, const int32_t* __restrict__ more_list,
                                                                  const int32_t* __restrict__ n_more, int32_t* __restrict__ idx, double* __restrict__ dist,
                                                                  int32_t* __restrict__ flag_list, int32_t* __restrict__ n_flag, int force_unproven,
                                                                  unsigned long long* __restrict__ stats) {
    __shared__ double s_t[kBlock / 64][kNC][kFT];
    __shared__ int s_j[kBlock / 64][64 * kEPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int z = blockIdx.z, nA = S.Q, D = S.Dp;
    const int nm = min(n_more[z], nA);
    if (nm == 0) return;
    const SegConst c = sc[z];
    const SegView V = seg_view(S, nrmS, nrmM, c, z);
    part_idx += (size_t)z * splits * nA * KC; part_s += (size_t)z * splits * nA * KC;
    idx += (size_t)z * nA * 2; dist += (size_t)z * nA * 2; flag_list += (size_t)z * nA; more_list += (size_t)z * nA;
    for (int k = blockIdx.x * (kBlock / 64) + wave; k < nm; k += gridDim.x * (kBlock / 64)) {       // wave-uniform
        const int qi = more_list[k];
        const size_t o = (size_t)z * nA + qi;
        const SegQueryInfo I = qinfo[o];
        Top2T<double> t2{INFINITY, INFINITY, -1, -1};
#pragma unroll
        for (int r = 0; r < kNG * kNC; ++r) top2_insert_lex_t(t2, psum[o * kNG * kNC + r], I.j[r]);
        const SegFront F = seg_front(part_idx, part_s, splits, nA, qi, lane);
        const unsigned slack = 2u * (unsigned)(D + 1) + 2u + c.slack2;
        int jk[kEPL]; unsigned sk[kEPL];
#pragma unroll
        for (int u = 0; u < kEPL; ++u) {
            const bool need = seg_front_need(F, u, slack);
            jk[u] = need ? F.j[u] : -1; sk[u] = need ? F.sq[u] : 0xFFFFFFFFu;
        }
        seg_pick_first(jk, sk, lane, nullptr);                 // the 8 that were summed leave jk
        unsigned smax = 0xFFFFFFFFu;
        if (t2.i2 >= 0 && t2.d2 < INFINITY) {
            const double t = (t2.d2 * c.scale * (1.0 + 1e-12) + (double)c.eunits) / c.rho + (double)(D + 1) + 1.0;
            if (t < 4.0e9) smax = (unsigned)t;
        }
        int n_left = 0;
#pragma unroll
        for (int u = 0; u < kEPL; ++u) {
            const bool more = jk[u] >= 0 && sk[u] <= smax;
            const unsigned long long m = __ballot(more);
            if (more) s_j[wave][n_left + __popcll(m & ((1ull << lane) - 1ull))] = jk[u];
            n_left += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (n_left > 0) {
            double e1, e2; int k1, k2;
            seg_rerank<false>(V, seg_surface_row(V, qi), s_j[wave], n_left, D, s_t[wave], e1, k1, e2, k2);
            top2_insert_lex_t(t2, e1, k1); top2_insert_lex_t(t2, e2, k2);
        }
        const bool ok = !force_unproven && seg_certified(c, F.g, D, t2);
        if (lane == 0) {
            idx[(size_t)qi * 2] = t2.i1; idx[(size_t)qi * 2 + 1] = t2.i2; dist[(size_t)qi * 2] = t2.d1; dist[(size_t)qi * 2 + 1] = t2.d2;
            if (!ok) { int sl = atomicAdd(&n_flag[z], 1); flag_list[sl] = qi; }
            if (stats) { atomicAdd(&stats[0], 1ull); atomicAdd(&stats[1], (unsigned long long)I.n_need); if (!ok) atomicAdd(&stats[2], 1ull); }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
}

// An unproven query does not need every row: a row whose reference score s has rho (s - (D + 1)) - E > scale * d2 for the
// provisional (exact) second distance d2 cannot enter the pair.  One wave per unproven query reads the query's scores of all
// candidate rows from the score matrix, keeps the rows below that threshold (a few dozen where the exhaustive scan read 1533 rows of
// 7.8 KB each) and re-ranks them exactly; a query with more than 128 rows under the threshold stays for segp_exact_rows_kernel
// (flag2 / n_flag2).
template <bool BACK>
__global__ __launch_bounds__(kBlock) void segp_refine_kernel(SegSets S, const double* __restrict__ nrmS, const double* __restrict__ nrmM,
                                                             const SegConst* __restrict__ sc, const int32_t* __restrict__ cand_m,
                                                             const uint32_t* __restrict__ Sc, int ldsc, const int32_t* __restrict__ flag_list,
                                                             const int32_t* __restrict__ n_flag, int32_t* __restrict__ idx, double* __restrict__ dist,
                                                             int32_t* __restrict__ flag2, int32_t* __restrict__ n_flag2, int skip,
                                                             unsigned long long* __restrict__ stats) {
    __shared__ double s_t[kBlock / 64][kNC][kFT];
    __shared__ int s_j[kBlock / 64][64 * kEPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int z = blockIdx.z, nA = S.Q, D = S.Dp;
    const int nf = min(n_flag[z], nA);
    const SegConst c = sc[z];
    const SegView V = seg_view(S, nrmS, nrmM, c, z);
    flag_list += (size_t)z * nA; flag2 += (size_t)z * nA; idx += (size_t)z * nA * 2; dist += (size_t)z * nA * 2;
    const int off = S.seg_off[z], nB = BACK ? S.Q : S.seg_off[z + 1] - off;
    for (int k = blockIdx.x * (kBlock / 64) + wave; k < nf; k += gridDim.x * (kBlock / 64)) {       // wave-uniform
        const int qi = flag_list[k];
        const double p1 = dist[(size_t)qi * 2], p2 = dist[(size_t)qi * 2 + 1];
        const int q1 = idx[(size_t)qi * 2], q2 = idx[(size_t)qi * 2 + 1];
        // rows that can still matter: reference score <= smax (everything when there is no provisional second yet)
        unsigned smax = 0xFFFFFFFFu;
        if (q2 >= 0 && p2 < INFINITY) {
            const double t = (p2 * c.scale * (1.0 + 1e-12) + (double)c.eunits) / c.rho + (double)(D + 1) + 1.0;
            if (t < 4.0e9) smax = (unsigned)t;
        }
        const int mrow = BACK ? V.rows[cand_m[(size_t)z * nA + qi]] : 0;
        int n_need = 0; bool over = skip != 0;
        for (int j0 = 0; j0 < nB && !over; j0 += 64) {
            const int j = j0 + lane;
            unsigned sv = 0xFFFFFFFFu;
            if (j < nB) sv = BACK ? Sc[(size_t)mrow * ldsc + j] : Sc[(size_t)V.rows[j] * ldsc + qi];
            const bool take = j < nB && sv <= smax;
            const unsigned long long m = __ballot(take);
            const int cnt = __popcll(m);
            if (n_need + cnt > 64 * kEPL) { over = true; break; }
            if (take) s_j[wave][n_need + __popcll(m & ((1ull << lane) - 1ull))] = j;
            n_need += cnt;
        }
        if (over) { if (lane == 0) { const int slot = atomicAdd(&n_flag2[z], 1); flag2[slot] = qi; if (stats) atomicAdd(&stats[3], 1ull); } continue; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        const SegRow a = BACK ? seg_model_row(V, cand_m[(size_t)z * nA + qi]) : seg_surface_row(V, qi);
        double d1, d2; int i1, i2;
        seg_rerank<BACK>(V, a, s_j[wave], n_need, D, s_t[wave], d1, i1, d2, i2);
        // the provisional pair was summed exactly too (its rows are under the threshold, so they were summed again: same bits)
        Top2T<double> t2{d1, d2, i1, i2};
        top2_insert_lex_t(t2, p1, q1); top2_insert_lex_t(t2, p2, q2);
        if (lane == 0) { idx[(size_t)qi * 2] = t2.i1; idx[(size_t)qi * 2 + 1] = t2.i2; dist[(size_t)qi * 2] = t2.d1; dist[(size_t)qi * 2 + 1] = t2.d2; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
}

// The Unique back-check straight from the score matrix.  Candidate k = (surface row q, model row m) asks whether q is the best
// surface row for m.  Its distance d(q, m) is known exactly (the forward result: the same terms in the same order), so only surface
// rows whose reference score allows a distance <= d(q, m) can answer "no": one wave per candidate reads m's row of the matrix,
// keeps those rows (q among them) and re-ranks them exactly -- no candidate lists, no certificate, because nothing is left out.
// A candidate with more than 128 such rows goes to segp_exact_rows_kernel<true> (flag2 / n_flag2).
__global__ __launch_bounds__(kBlock) void segp_back_direct_kernel(SegSets S, const double* __restrict__ nrmS, const double* __restrict__ nrmM,
                                                                  const SegConst* __restrict__ sc, const int32_t* __restrict__ cand_q,
                                                                  const int32_t* __restrict__ cand_m, const int32_t* __restrict__ n_cand,
                                                                  const uint32_t* __restrict__ Sc, int ldsc, const double* __restrict__ fdist,
                                                                  int32_t* __restrict__ bidx, double* __restrict__ bdist,
                                                                  int32_t* __restrict__ flag2, int32_t* __restrict__ n_flag2, int skip,
                                                                  unsigned long long* __restrict__ stats) {
    __shared__ double s_t[kBlock / 64][kNC][kFT];
    __shared__ int s_j[kBlock / 64][64 * kEPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int z = blockIdx.z, nA = S.Q, D = S.Dp;
    const int nc = min(n_cand[z], nA);
    const SegConst c = sc[z];
    const SegView V = seg_view(S, nrmS, nrmM, c, z);
    cand_q += (size_t)z * nA; cand_m += (size_t)z * nA; fdist += (size_t)z * nA * 2;
    bidx += (size_t)z * nA * 2; bdist += (size_t)z * nA * 2; flag2 += (size_t)z * nA;
    for (int k = blockIdx.x * (kBlock / 64) + wave; k < nc; k += gridDim.x * (kBlock / 64)) {       // wave-uniform
        const int qi = cand_q[k], jm = cand_m[k];
        const double dq = fdist[(size_t)qi * 2];
        unsigned smax = 0xFFFFFFFFu;
        { const double t = (dq * c.scale * (1.0 + 1e-12) + (double)c.eunits) / c.rho + (double)(D + 1) + 1.0; if (t < 4.0e9) smax = (unsigned)t; }
        const uint32_t* row = Sc + (size_t)V.rows[jm] * ldsc;
        // the best reference score of the row bounds the best true distance from above: rows more than the two-sided slack above it
        // cannot be the best -- if q is one of them the answer is "no" without any exact work (most weak matches end here)
        unsigned best = 0xFFFFFFFFu;
        for (int a = lane; a < nA; a += 64) best = min(best, row[a]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
        const unsigned slack = 2u * (unsigned)(D + 1) + 2u + c.slack2;
        const unsigned bmax = best > 0xFFFFFFFFu - slack ? 0xFFFFFFFFu : best + slack;
        if (row[qi] > bmax && !skip) {
            if (lane == 0) { bidx[(size_t)k * 2] = -1; bidx[(size_t)k * 2 + 1] = -1; bdist[(size_t)k * 2] = INFINITY; bdist[(size_t)k * 2 + 1] = INFINITY; }
            continue;
        }
        smax = min(smax, bmax);
        int n_need = 0; bool over = skip != 0;
        for (int a0 = 0; a0 < nA && !over; a0 += 64) {
            const int a = a0 + lane;
            const bool take = a < nA && row[a] <= smax;
            const unsigned long long m = __ballot(take);
            const int cnt = __popcll(m);
            if (n_need + cnt > 64 * kEPL) { over = true; break; }
            if (take) s_j[wave][n_need + __popcll(m & ((1ull << lane) - 1ull))] = a;
            n_need += cnt;
        }
        if (stats && lane == 0) atomicAdd(&stats[4], 1ull);
        if (over) { if (lane == 0) { const int slot = atomicAdd(&n_flag2[z], 1); flag2[slot] = k; if (stats) atomicAdd(&stats[5], 1ull); } continue; }
        // q itself always passes (smax bounds its own score from above, and it was not rejected against bmax); every row left out
        // is STRICTLY farther than d(q, m).  If q is alone, it is the best surface row of m: nothing to sum.
        if (n_need == 1) {
            if (lane == 0) { bidx[(size_t)k * 2] = qi; bidx[(size_t)k * 2 + 1] = -1; bdist[(size_t)k * 2] = dq; bdist[(size_t)k * 2 + 1] = INFINITY; }
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        double d1, d2; int i1, i2;
        seg_rerank<true>(V, seg_model_row(V, jm), s_j[wave], n_need, D, s_t[wave], d1, i1, d2, i2);
        if (lane == 0) { bidx[(size_t)k * 2] = i1; bidx[(size_t)k * 2 + 1] = i2; bdist[(size_t)k * 2] = d1; bdist[(size_t)k * 2 + 1] = d2; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
}

// min over the surface rows of every model row's reference scores: the back-check of EVERY sphere compares a candidate's own score
// with it first (most weak matches end there), so it is taken once per call from the shared matrix -- a wave per model row, 0.1 ms --
// instead of once per back-check item (a full pass over the 8 KB row of the matrix per item: 1.5 ms of the sweep on described rows,
// where half the queries reach the back-check).
__global__ __launch_bounds__(kBlock) void segp_rowmin_kernel(const uint32_t* __restrict__ Sc, int ldsc, int nA, int VM, uint32_t* __restrict__ rowmin) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (r >= VM) return;
    const uint32_t* row = Sc + (size_t)r * ldsc;
    unsigned best = 0xFFFFFFFFu;
    const int n4 = (nA + 3) / 4;
#pragma unroll 4
    for (int a4 = lane; a4 < n4; a4 += 64) {
        const uint4 v = *(const uint4*)(row + (size_t)a4 * 4);
        const int a = a4 * 4;
        best = min(best, v.x);
        if (a + 1 < nA) best = min(best, v.y);
        if (a + 2 < nA) best = min(best, v.z);
        if (a + 3 < nA) best = min(best, v.w);
    }
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o_));
    if (lane == 0) rowmin[r] = best;
}

// segp_back_direct_kernel with its exact sums handed to the dense chain: an item whose rows to sum (q among them) fit two groups
// writes them as groups 2 k, 2 k + 1 of item k -- a-side = the model row, b-side = surface rows -- for segp_scan_kernel /
// segp_rerank_pairs_kernel (A = PM, B = PS) / segp_back_decide_kernel; the few with more rows are summed here as before.
__global__ __launch_bounds__(kBlock) void segp_back_pick_kernel(SegSets S, const double* __restrict__ nrmS, const double* __restrict__ nrmM,
                                                                const SegConst* __restrict__ sc, const int32_t* __restrict__ cand_q,
                                                                const int32_t* __restrict__ cand_m, const int32_t* __restrict__ n_cand,
                                                                const uint32_t* __restrict__ Sc, int ldsc, const uint32_t* __restrict__ rowmin,
                                                                const double* __restrict__ fdist,
                                                                int32_t* __restrict__ bidx, double* __restrict__ bdist,
                                                                int32_t* __restrict__ flag2, int32_t* __restrict__ n_flag2, int skip,
                                                                SegGroupRows* __restrict__ grows, SegGroupNorms* __restrict__ gnorms,
                                                                SegQueryInfo* __restrict__ qinfo, unsigned long long* __restrict__ stats) {
    __shared__ double s_t[kBlock / 64][kNC][kFT];
    __shared__ int s_j[kBlock / 64][64 * kEPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int z = blockIdx.z, nA = S.Q, D = S.Dp;
    const int nc = min(n_cand[z], nA);
    const SegConst c = sc[z];
    const SegView V = seg_view(S, nrmS, nrmM, c, z);
    cand_q += (size_t)z * nA; cand_m += (size_t)z * nA; fdist += (size_t)z * nA * 2;
    bidx += (size_t)z * nA * 2; bdist += (size_t)z * nA * 2; flag2 += (size_t)z * nA;
    for (int k = blockIdx.x * (kBlock / 64) + wave; k < nc; k += gridDim.x * (kBlock / 64)) {       // wave-uniform
        const size_t o = (size_t)z * nA + k;
        const int qi = cand_q[k], jm = cand_m[k];
        const double dq = fdist[(size_t)qi * 2];
        unsigned smax = 0xFFFFFFFFu;
        { const double t = (dq * c.scale * (1.0 + 1e-12) + (double)c.eunits) / c.rho + (double)(D + 1) + 1.0; if (t < 4.0e9) smax = (unsigned)t; }
        const uint32_t* row = Sc + (size_t)V.rows[jm] * ldsc;
        // (segp_back_direct_kernel's comments)  The row's minimum comes from segp_rowmin_kernel; the row itself is read -- 16 bytes per lane,
        // four loads in flight -- only by the items that survive the test against it.
        const unsigned best = rowmin[V.rows[jm]];
        const int n4 = (nA + 3) / 4;                              // ldsc is a multiple of 4 and >= nA: the 16-byte loads stay inside the row
        const unsigned slack = 2u * (unsigned)(D + 1) + 2u + c.slack2;
        const unsigned bmax = best > 0xFFFFFFFFu - slack ? 0xFFFFFFFFu : best + slack;
        if (row[qi] > bmax && !skip) {
            if (lane == 0) { bidx[(size_t)k * 2] = -1; bidx[(size_t)k * 2 + 1] = -1; bdist[(size_t)k * 2] = INFINITY; bdist[(size_t)k * 2 + 1] = INFINITY; qinfo[o].ng = 0; }
            continue;
        }
        smax = min(smax, bmax);
        int n_need = 0; bool over = skip != 0;
        // (the list's order is not ascending: no consumer depends on it -- the top two are taken by (distance, row))
        for (int a0 = 0; a0 < n4 && !over; a0 += 64) {
            const int a4 = a0 + lane;
            uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            if (a4 < n4) v = *(const uint4*)(row + (size_t)a4 * 4);
            const unsigned ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = a4 * 4 + e;
                const bool take = a4 < n4 && a < nA && ve[e] <= smax;
                const unsigned long long m = __ballot(take);
                const int cnt = __popcll(m);
                if (cnt && !over) {
                    if (n_need + cnt > 64 * kEPL) over = true;
                    else { if (take) s_j[wave][n_need + __popcll(m & ((1ull << lane) - 1ull))] = a; n_need += cnt; }
                }
            }
        }
        if (stats && lane == 0) atomicAdd(&stats[4], 1ull);
        if (over) { if (lane == 0) { const int slot = atomicAdd(&n_flag2[z], 1); flag2[slot] = k; qinfo[o].ng = 0; if (stats) atomicAdd(&stats[5], 1ull); } continue; }
        if (n_need == 1) {
            if (lane == 0) { bidx[(size_t)k * 2] = qi; bidx[(size_t)k * 2 + 1] = -1; bdist[(size_t)k * 2] = dq; bdist[(size_t)k * 2 + 1] = INFINITY; qinfo[o].ng = 0; }
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        const SegRow am = seg_model_row(V, jm);
        if (n_need <= kNG * kNC) {
            const int ng = (n_need + kNC - 1) / kNC;
            bool fast = am.rinv != 0.0;
            if (lane < kNG * kNC) {
                const int a = lane < n_need ? s_j[wave][lane] : -1, gq = lane / kNC, cq = lane % kNC;
                int rw = 0; double nrm = 1.0, rinv = 1.0;
                if (a >= 0) { rw = a; nrm = V.nrmS[a]; rinv = V.rinvS[a]; fast = fast && rinv != 0.0; }
                qinfo[o].j[lane] = a;
                if (gq < ng) {
                    grows[o * kNG + gq].row[cq] = rw; gnorms[o * kNG + gq].nrm[cq] = nrm; gnorms[o * kNG + gq].rinv[cq] = rinv;
                    if (cq == 0) { gnorms[o * kNG + gq].nrm_a = am.nrm; gnorms[o * kNG + gq].rinv_a = am.rinv; }
                }
            }
            fast = __ballot(fast) == ~0ull;
            if (lane < ng) { grows[o * kNG + lane].arow = V.rows[jm]; grows[o * kNG + lane].fast = fast ? 1 : 0; }
            if (lane == 0) qinfo[o].ng = ng;
        } else {
            double d1, d2; int i1, i2;
            seg_rerank<true>(V, am, s_j[wave], n_need, D, s_t[wave], d1, i1, d2, i2);
            if (lane == 0) { bidx[(size_t)k * 2] = i1; bidx[(size_t)k * 2 + 1] = i2; bdist[(size_t)k * 2] = d1; bdist[(size_t)k * 2 + 1] = d2; qinfo[o].ng = 0; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
}
__global__ __launch_bounds__(kBlock) void segp_back_decide_kernel(const SegQueryInfo* __restrict__ qinfo, const double* __restrict__ psum, const int32_t* __restrict__ n_cand,
                                                                  int Q, int32_t* __restrict__ bidx, double* __restrict__ bdist) {
    const int z = blockIdx.z, k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= min(n_cand[z], Q)) return;
    const size_t o = (size_t)z * Q + k;
    const SegQueryInfo I = qinfo[o];
    if (I.ng == 0) return;                                     // answered by segp_back_pick_kernel
    Top2T<double> t2{INFINITY, INFINITY, -1, -1};
#pragma unroll
    for (int r = 0; r < kNG * kNC; ++r) if (r < I.ng * kNC) top2_insert_lex_t(t2, psum[o * kNG * kNC + r], I.j[r]);
    bidx[o * 2] = t2.i1; bidx[o * 2 + 1] = t2.i2; bdist[o * 2] = t2.d1; bdist[o * 2 + 1] = t2.d2;
}

// unproven queries of every segment, exhaustively and exactly (sad_exact_rows_kernel's order); grid (queries, slices, segments).
// A workgroup takes one unproven query and one slice of the candidate rows, 256 rows at a time: all threads stage the
// |a - b| terms of 256 rows x 16 features in LDS (reads coalesced along the features of the row-major P), then every thread
// adds ITS row's terms in ascending feature order.
constexpr int kSegFbSlices = 8, kXF = 16;
template <bool BACK>
__global__ __launch_bounds__(kBlock) void segp_exact_rows_kernel(SegSets S, const double* __restrict__ nrmS, const double* __restrict__ nrmM,
                                                                 const SegConst* __restrict__ sc, const int32_t* __restrict__ cand_m,
                                                                 const int32_t* __restrict__ flag_list, const int32_t* __restrict__ n_flag,
                                                                 int slice_cap, int32_t* __restrict__ part_idx, double* __restrict__ part_dist) {
    extern __shared__ double s_a[];                           // Dp doubles
    __shared__ double s_t[kXF][kBlock + 1];
    __shared__ double s_d[4][2]; __shared__ int s_i[4][2];
    const int z = blockIdx.z, cap = S.Q, D = S.Dp, tid = threadIdx.x;
    const int nf = min(n_flag[z], cap);
    if (nf == 0) return;
    const SegConst c = sc[z];
    const SegView V = seg_view(S, nrmS, nrmM, c, z);
    const int nB = BACK ? S.Q : S.seg_off[z + 1] - S.seg_off[z];
    const int slice = BACK ? (S.Q + kSegFbSlices - 1) / kSegFbSlices : slice_cap;
    flag_list += (size_t)z * cap;
    part_idx += (size_t)z * kSegFbSlices * cap * 2; part_dist += (size_t)z * kSegFbSlices * cap * 2;
    const int begin = blockIdx.y * slice, end = min(nB, begin + slice);
    const int f = tid & (kXF - 1), rq = tid >> 4;             // staging role: feature f of rows rq, rq + 16, ...
    for (int k = blockIdx.x; k < nf; k += gridDim.x) {
        const int qi = flag_list[k];
        __syncthreads();
        const SegRow a = BACK ? seg_model_row(V, cand_m[(size_t)z * cap + qi]) : seg_surface_row(V, qi);
        for (int d = tid; d < D; d += kBlock) s_a[d] = seg_value(V, a, d);
        __syncthreads();
        double d1 = INFINITY, d2 = INFINITY; int i1 = -1, i2 = -1;
        for (int j0 = begin; j0 < end; j0 += kBlock) {
            SegRow br[kBlock / kXF];
#pragma unroll
            for (int u = 0; u < kBlock / kXF; ++u) {
                const int jr = min(j0 + rq + (kBlock / kXF) * u, end - 1);
                br[u] = BACK ? seg_surface_row(V, jr) : seg_model_row(V, jr);
            }
            double s = 0.0;
            for (int d0 = 0; d0 < D; d0 += kXF) {
                const int d = d0 + f, dc = min(d, D - 1);
                // sixteen unconditional loads in flight, then the arithmetic: written as one conditional expression per row the
                // compiler put every load behind its own branch and wait, and a task took 16 x 62 serial round trips (6.9 ms for a
                // kernel with 141 queries to do)
                double pv[kBlock / kXF];
#pragma unroll
                for (int u = 0; u < kBlock / kXF; ++u) pv[u] = br[u].p[min(dc, V.D0 - 1)];
                const double av = s_a[dc];
#pragma unroll
                for (int u = 0; u < kBlock / kXF; ++u) {
                    const double bvv = seg_div(dc < V.D0 ? pv[u] : V.cc, br[u]);
                    s_t[f][rq + (kBlock / kXF) * u] = d < D ? fabs(av - bvv) : 0.0;
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < kXF; ++e) s += s_t[e][tid];          // terms past D are +0.0: a non-negative sum is unchanged
                __syncthreads();
            }
            const int j = j0 + tid;
            if (j < end) { if (s < d1) { d2 = d1; i2 = i1; d1 = s; i1 = j; } else if (s < d2) { d2 = s; i2 = j; } }   // ascending j: strict
        }
        auto merge = [&](double e1, int k1, double e2, int k2) {
            bool fm = lexd_lt(d1, i1, e1, k1);
            double w1 = fm ? d1 : e1; int x1 = fm ? i1 : k1;
            double m2 = fm ? d2 : d1; int y2 = fm ? i2 : i1;
            double o2 = fm ? e1 : e2; int z2 = fm ? k1 : k2;
            bool sm = lexd_lt(m2, y2, o2, z2);
            d1 = w1; i1 = x1; d2 = sm ? m2 : o2; i2 = sm ? y2 : z2;
        };
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) merge(__shfl_xor(d1, o), __shfl_xor(i1, o), __shfl_xor(d2, o), __shfl_xor(i2, o));
        const int wv = tid >> 6;
        if ((tid & 63) == 0) { s_d[wv][0] = d1; s_d[wv][1] = d2; s_i[wv][0] = i1; s_i[wv][1] = i2; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) merge(s_d[w][0], s_i[w][0], s_d[w][1], s_i[w][1]);
            size_t o = ((size_t)blockIdx.y * cap + k) * 2;
            part_idx[o] = i1; part_idx[o + 1] = i2; part_dist[o] = d1; part_dist[o + 1] = d2;
        }
    }
}
__global__ void segp_fallback_finish_kernel(const int32_t* __restrict__ flag_list, const int32_t* __restrict__ n_flag, int cap,
                                            const int32_t* __restrict__ part_idx, const double* __restrict__ part_dist,
                                            int32_t* __restrict__ idx, double* __restrict__ dist) {
    const int z = blockIdx.z;
    const int nf = min(n_flag[z], cap);
    flag_list += (size_t)z * cap; idx += (size_t)z * cap * 2; dist += (size_t)z * cap * 2;
    part_idx += (size_t)z * kSegFbSlices * cap * 2; part_dist += (size_t)z * kSegFbSlices * cap * 2;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nf; k += gridDim.x * blockDim.x) {
        Top2T<double> t{INFINITY, INFINITY, -1, -1};
        for (int sl = 0; sl < kSegFbSlices; ++sl) {
            const size_t o = ((size_t)sl * cap + k) * 2;
            top2_insert_lex_t(t, part_dist[o], part_idx[o]);
            top2_insert_lex_t(t, part_dist[o + 1], part_idx[o + 1]);
        }
        const int qi = flag_list[k];
        idx[(size_t)qi * 2] = t.i1; idx[(size_t)qi * 2 + 1] = t.i2;
        dist[(size_t)qi * 2] = t.d1; dist[(size_t)qi * 2 + 1] = t.d2;
    }
}

// threshold + ratio + ordered compaction of a segment's surviving queries (filter_compact_kernel, one workgroup per segment);
// also clears the back-search's fallback counter
__global__ __launch_bounds__(kCompactThreads) void segp_filter_kernel(const int32_t* __restrict__ idx, const double* __restrict__ dist, int Q,
                                                                      const int32_t* __restrict__ seg_off, double thr, double ratio,
                                                                      int32_t* __restrict__ cand_q, int32_t* __restrict__ cand_m,
                                                                      int32_t* __restrict__ n_cand, int32_t* __restrict__ n_flag) {
    __shared__ int s_wave[kCompactThreads / 64];
    const int z = blockIdx.x, M_total = seg_off[z + 1] - seg_off[z];
    idx += (size_t)z * Q * 2; dist += (size_t)z * Q * 2; cand_q += (size_t)z * Q; cand_m += (size_t)z * Q;
    const int per = (Q + kCompactThreads - 1) / kCompactThreads;
    const int lo = min(Q, (int)threadIdx.x * per), hi = min(Q, lo + per);
    int cnt = 0;
    for (int qi = lo; qi < hi; ++qi) cnt += filter_keep<double>(idx, dist, qi, M_total, thr, ratio);
    int total;
    int o = block_exclusive_scan_1024(cnt, s_wave, &total);
    for (int qi = lo; qi < hi; ++qi)
        if (filter_keep<double>(idx, dist, qi, M_total, thr, ratio)) { cand_q[o] = qi; cand_m[o] = idx[(size_t)qi * 2]; ++o; }
    if (threadIdx.x == 0) { n_cand[z] = total; n_flag[z] = 0; }
}

// Unique (keep candidate k iff the best surface row of its model row is its own query) + ordered emission of the
// 1-based pairs; one workgroup per segment (unique_flag_kernel + emit_pairs_kernel)
__global__ __launch_bounds__(kBlock) void segp_emit_kernel(const int32_t* __restrict__ cand_q, const int32_t* __restrict__ cand_m,
                                                           const int32_t* __restrict__ n_cand, const int32_t* __restrict__ back_idx, int unique,
                                                           const double* __restrict__ dist, int Q, uint32_t* __restrict__ pairs,
                                                           double* __restrict__ metric, int32_t* __restrict__ n_pairs) {
    __shared__ int s_cnt[kBlock / 64];
    __shared__ int s_base;
    const int z = blockIdx.x, P = n_cand[z];
    cand_q += (size_t)z * Q; cand_m += (size_t)z * Q; back_idx += (size_t)z * Q * 2; dist += (size_t)z * Q * 2;
    pairs += (size_t)z * Q * 2; if (metric) metric += (size_t)z * Q;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int k0 = 0; k0 < P; k0 += kBlock) {
        const int k = k0 + threadIdx.x;
        const bool kp = k < P && (!unique || back_idx[(size_t)k * 2] == cand_q[k]);
        const unsigned long long b = __ballot(kp);
        if (lane == 0) s_cnt[wave] = __popcll(b);
        __syncthreads();
        int base = s_base;
        for (int w = 0; w < wave; ++w) base += s_cnt[w];
        if (kp) {
            const int o = base + __popcll(b & ((1ull << lane) - 1ull));
            pairs[(size_t)o * 2] = (uint32_t)cand_q[k] + 1u;
            pairs[(size_t)o * 2 + 1] = (uint32_t)cand_m[k] + 1u;
            if (metric) metric[o] = dist[(size_t)cand_q[k] * 2];
        }
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < kBlock / 64; ++w) t += s_cnt[w]; s_base += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) n_pairs[z] = s_base;
}

}  // namespace

// out [n][D] (row-major) = f (n x D, feature-major, leading dimension ld).  The reverse direction is
// the same call with the roles swapped: launch_transpose_rows(rowmajor, D, D, n, featmajor).
int launch_transpose_rows(const double* f, int n, int ld, int D, double* out, hipStream_t st) {
    if (n <= 0 || D <= 0) return PCREG_OK;
    hipLaunchKernelGGL(transpose_rows_kernel, dim3((n + 63) / 64, (D + 63) / 64), dim3(kBlock), 0, st, f, n, ld, D, out, (const int32_t*)nullptr);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

// u16 rows [.][D] (the descriptor entry's rows, written once in keypoint order) -> feature-major doubles [D][n]: counts are
// exact in both.  index (or null = identity): row i of the output is rows[index[i]] -- the list of surviving keypoints.
__global__ __launch_bounds__(kBlock) void widen_rows_u16_kernel(const uint16_t* __restrict__ rows, const int32_t* __restrict__ index, int n, int D,
                                                                double* __restrict__ out) {
    __shared__ uint16_t tile[64][66];
    const int i0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = i0 + ty + 4 * k, d = d0 + tx;
        tile[ty + 4 * k][tx] = (i < n && d < D) ? rows[(size_t)(index ? index[i] : i) * D + d] : (uint16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int d = d0 + ty + 4 * k, i = i0 + tx;
        if (i < n && d < D) out[i + (size_t)d * n] = (double)tile[tx][ty + 4 * k];
    }
}
int launch_widen_rows_u16(const uint16_t* rows, const int32_t* index, int n, int D, double* featmajor, hipStream_t st) {
    if (n <= 0) return PCREG_OK;
    hipLaunchKernelGGL(widen_rows_u16_kernel, dim3((n + 63) / 64, (D + 63) / 64), dim3(kBlock), 0, st, rows, index, n, D, featmajor);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

// workspace: Range | n_flag | minmax partials | Aq | Bq | At | Bt | part_idx | part_s | flag_list
//            | fallback: fi | fd | slice partials [kFbSlices][nA][2] (i32, f64)
size_t sad16_workspace_bytes(int nA, int nB, int D) {
    size_t a = (size_t)(nA > 0 ? nA : 1), b = (size_t)(nB > 0 ? nB : 1);
    size_t D2p = align_up((size_t)(D + 1) / 2, DK2);
    return 256 + 256 + align_up(1024 * 2 * 8, 256) + D2p * align_up(a, BQ) * 4 + D2p * align_up(b, BQ) * 4 +
           align_up(a * (size_t)D * 8, 256) + align_up(b * (size_t)D * 8, 256) +
           2 * align_up((size_t)kMaxSplit * a * KC * 4, 256) + align_up(a * 4, 256) +
           align_up(a * 2 * 4, 256) + align_up(a * 2 * 8, 256) +
           align_up((size_t)kFbSlices * a * 2 * 4, 256) + align_up((size_t)kFbSlices * a * 2 * 8, 256);
}

// Top-2 of every row of A against all rows of B under SAD: indices and fp64 distances identical to
// the exhaustive fp64 search (launch_score_top2_exact).  No host round trip.
// nA_live (device int32, may be null): only the first *nA_live of the nA rows of A are real; everything is sized and
// strided by nA, the kernels skip the rest -- lets a caller run on a capacity without reading the count back.
int run_sad16_top2(const double* A, int nA, int lda, const double* B, int nB, int ldb, int D,
                   int32_t* idx, double* dist, void* ws, size_t ws_bytes, hipStream_t st, const int32_t* nA_live) {
    PCREG_ARG(nA >= 1 && nB >= 1 && D >= 1);
    size_t need = sad16_workspace_bytes(nA, nB, D);
    if (ws_bytes < need) { set_error("sad16 workspace too small: %zu < %zu", ws_bytes, need); return PCREG_E_WORKSPACE; }
    const int D2p = (int)align_up((size_t)(D + 1) / 2, DK2);
    const int ldqa = (int)align_up((size_t)nA, BQ), ldqb = (int)align_up((size_t)nB, BQ);
    size_t a = (size_t)nA, b = (size_t)nB;
    char* w = (char*)ws;
    Range* range = (Range*)w;               w += 256;
    int32_t* n_flag = (int32_t*)w;          w += 256;
    double* mpart = (double*)w;             w += align_up(1024 * 2 * 8, 256);
    uint32_t* Aq = (uint32_t*)w;            w += (size_t)D2p * ldqa * 4;
    uint32_t* Bq = (uint32_t*)w;            w += (size_t)D2p * ldqb * 4;
    double* At = (double*)w;                w += align_up(a * (size_t)D * 8, 256);
    double* Bt = (double*)w;                w += align_up(b * (size_t)D * 8, 256);
    int32_t* part_idx = (int32_t*)w;        w += align_up((size_t)kMaxSplit * a * KC * 4, 256);
    uint32_t* part_s = (uint32_t*)w;        w += align_up((size_t)kMaxSplit * a * KC * 4, 256);
    int32_t* flag_list = (int32_t*)w;       w += align_up(a * 4, 256);
    w += align_up(a * 2 * 4, 256) + align_up(a * 2 * 8, 256);      // (formerly the merged fallback rows)
    int32_t* fpi = (int32_t*)w;             w += align_up((size_t)kFbSlices * a * 2 * 4, 256);
    double* fpd = (double*)w;

    PCREG_ARG(lda >= nA && ldb >= nB);
    int nb = (int)std::min<size_t>(1024, ((a + b) * D + kBlock * 8 - 1) / (kBlock * 8)); if (nb < 1) nb = 1;
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(nb), dim3(kBlock), 0, st, A, nA, lda, B, nB, ldb, D, mpart, nA_live);
    hipLaunchKernelGGL(range_final_kernel, dim3(1), dim3(64), 0, st, mpart, nb, range);
    hipLaunchKernelGGL(quantize_pack_kernel, dim3(2048), dim3(kBlock), 0, st, A, nA, lda, D, D2p, ldqa, range, Aq, nA_live);
    hipLaunchKernelGGL(quantize_pack_kernel, dim3(2048), dim3(kBlock), 0, st, B, nB, ldb, D, D2p, ldqb, range, Bq, (const int32_t*)nullptr);
    PCREG_HIP(hipMemsetAsync(n_flag, 0, sizeof(int32_t), st));
    // Split B into S chunks so that the grid loads every CU equally: a CU holds 3 workgroups, all
    // resident at once, so the kernel lasts (workgroups on the fullest CU) x (row tiles per chunk).
    const int n_tiles = (nA + BQ - 1) / BQ, row_tiles = (nB + BM - 1) / BM;
    int S = 1, chunk = 0;
    {
        // cost ~ (row tiles per chunk) x (workgroups on the fullest CU); one workgroup alone on a CU runs
        // at about 2/3 of the per-workgroup speed of a full CU (measured sweep: scripts/sad_split_sweep.sh)
        const int forced = PCREG_EXP_ENV("PCREG_SAD_SPLITS", 0);            // experiment override
        long best_cost = -1;
        for (int s_try = 1; s_try <= kMaxSplit && s_try <= row_tiles; ++s_try) {
            if (forced > 0 && s_try != std::min(forced, std::min(kMaxSplit, row_tiles))) continue;
            int ct = (row_tiles + s_try - 1) / s_try, s_eff = (row_tiles + ct - 1) / ct;
            long per_cu = ((long)n_tiles * s_eff + 255) / 256;
            long cost = std::max(2 * per_cu, 3L) * ct;
            if (best_cost < 0 || cost <= best_cost) { best_cost = cost; S = s_eff; chunk = ct * BM; }   // ties: more chunks
        }
    }
    unsigned long long* dbg = nullptr;
    const char* tl = PCREG_EXP_STR("PCREG_SAD_TIMELINE");      // debug: dump per-block (start, end, HW_ID, XCC_ID) to this file
    if (tl) PCREG_HIP(hipMalloc(&dbg, (size_t)n_tiles * S * 4 * sizeof(unsigned long long)));
#ifdef PCREG_EXPERIMENTS
    if (PCREG_EXP_ENV("PCREG_SAD_DRY", 0))      // timing experiment only: list maintenance compiled out, results invalid
        hipLaunchKernelGGL(sad16_candidates_kernel<kSadDry>, dim3(n_tiles, S), dim3(kBlock), 0, st, Aq, nA, ldqa, Bq, nB, ldqb, D2p, chunk, part_idx, part_s, dbg, nA_live, SegZ{0, 0, 0, 0, nullptr, nullptr, 0});
    else
#endif
        hipLaunchKernelGGL(sad16_candidates_kernel<kSadLists>, dim3(n_tiles, S), dim3(kBlock), 0, st, Aq, nA, ldqa, Bq, nB, ldqb, D2p, chunk, part_idx, part_s, dbg, nA_live, SegZ{0, 0, 0, 0, nullptr, nullptr, 0});
    const int force = debug_flag(kDbgMatchForceFallback) != 0;
    unsigned long long* stats = match_stats_dev();
    if (stats) hipLaunchKernelGGL(stats_bump_kernel, dim3(1), dim3(1), 0, st, stats, 6, 1ull);
    if (dbg) {
        std::vector<unsigned long long> h((size_t)n_tiles * S * 4);
        PCREG_HIP(hipStreamSynchronize(st));
        PCREG_HIP(hipMemcpy(h.data(), dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(dbg);
        static int call = 0;
        char name[512]; snprintf(name, sizeof name, "%s.%d", tl, call++);
        if (FILE* f = fopen(name, "w")) {
            for (size_t b = 0; b < h.size() / 4; ++b) fprintf(f, "%zu %llu %llu %llu %llu\n", b, h[4 * b], h[4 * b + 1], h[4 * b + 2], h[4 * b + 3]);
            fclose(f);
        }
    }
    hipLaunchKernelGGL(transpose_rows_kernel, dim3((nA + 63) / 64, (D + 63) / 64), dim3(kBlock), 0, st, A, nA, lda, D, At, nA_live);
    hipLaunchKernelGGL(transpose_rows_kernel, dim3((nB + 63) / 64, (D + 63) / 64), dim3(kBlock), 0, st, B, nB, ldb, D, Bt, (const int32_t*)nullptr);
    hipLaunchKernelGGL(sad16_finalize_kernel, dim3((nA + 3) / 4), dim3(kBlock), 0, st, At, nA, Bt, nB, D, range,
                       part_idx, part_s, S, idx, dist, flag_list, n_flag, force, nA_live, stats);
    PCREG_HIP(hipGetLastError());
    if (PCREG_EXP_ENV("PCREG_MATCH_DEBUG", 0)) {                       // the only host round trip of the call, debugging only
        int32_t nf = 0;
        PCREG_HIP(hipMemcpyAsync(&nf, n_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        PCREG_HIP(hipStreamSynchronize(st));
        int occ = -1; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, sad16_candidates_kernel<kSadLists>, kBlock, 0);
        fprintf(stderr, "[pcreg] sad16: nA=%d nB=%d D=%d S=%d unproven=%d (candidates kernel: %d blocks/CU)\n", nA, nB, D, S, nf, occ);
    }
    {   // unproven rows: exact fp64 rows, sliced over B; both kernels read the count on the device
        int slices = std::min(kFbSlices, (nB + kBlock - 1) / kBlock);
        int slice = (nB + slices - 1) / slices;
        slices = (nB + slice - 1) / slice;
        hipLaunchKernelGGL(sad_exact_rows_kernel, dim3(std::min(nA, 64), slices), dim3(kBlock), (size_t)D * sizeof(double), st,
                           A, lda, B, nB, ldb, D, flag_list, n_flag, nA, slice, fpi, fpd);
        hipLaunchKernelGGL(sad_fallback_finish_kernel, dim3(std::min((nA + 255) / 256, 64)), dim3(256), 0, st, flag_list, n_flag, nA, slices,
                           fpi, fpd, idx, dist);
        PCREG_HIP(hipGetLastError());
    }
    return PCREG_OK;
}


// ---- segmented getMatches: host side --------------------------------------------------------------------------
// workspace: [once] P and five scalars per row of both sets, reference norms, the two quantised operands, the score matrix
// | [per segment] constants, norms, candidate lists, top-2, filters, fallback partials
namespace {
struct SegLayout {
    size_t PS, PM, rowS, rowM, nrefS, nrefM, Aq, Bq, Sc, sc, nrmS, nrmM, part_idx, part_s, idx, dist, bidx, bdist, cand_q, cand_m, n_cand, n_flag,
           flag_list, flag2, n_flag2, fpi, fpd, grows, gnorms, qinfo, psum, map, n_slots, more_list, n_more, refpart, rowmin, total;
    int D2p, ldqa, ldqb, splits, chunk;
};
SegLayout seg_layout(int Q, int VM, int D, int Dp, int S, int tot, int n_max) {
    SegLayout L{};
    const size_t q = (size_t)std::max(Q, 1), vm = (size_t)std::max(VM, 1), ns = (size_t)std::max(S, 1);
    L.D2p = (int)align_up((size_t)(Dp + 1) / 2, DK2);
    L.ldqa = (int)align_up(q, BQ); L.ldqb = (int)align_up(vm, BQ);
    // as MANY chunks as the finalize wave can read (kMaxSplit): the certificate compares the exact 2nd-best with the 4th-best
    // integer score of every chunk, and only short chunks put that 4th-best clear of it (one chunk per segment left most
    // queries of the sweep's shape unproven)
    const int row_tiles = std::max(1, (std::max(n_max, Q) + BM - 1) / BM);
    const int sp = std::min(kMaxSplit, row_tiles);
    const int ct = (row_tiles + sp - 1) / sp;
    L.splits = (row_tiles + ct - 1) / ct; L.chunk = ct * BM;
    size_t b = 0;
    auto take = [&](size_t bytes) { size_t o = b; b += align_up(bytes, 256); return o; };
    L.PS = take(q * D * 8); L.PM = take(vm * D * 8);
    L.rowS = take(6 * q * 8); L.rowM = take(6 * vm * 8); L.nrefS = take(q * 8); L.nrefM = take(vm * 8);
    L.Aq = take((size_t)L.D2p * L.ldqa * 4); L.Bq = take((size_t)L.D2p * L.ldqb * 4); L.Sc = take((size_t)L.ldqb * L.ldqa * 4);
    L.sc = take((ns + 1) * sizeof(SegConst));
    L.nrmS = take(2 * ns * q * 8); L.nrmM = take(2 * (size_t)std::max(tot, 1) * 8);          // norms, then their reciprocals
    L.part_idx = take(ns * L.splits * q * KC * 4); L.part_s = take(ns * L.splits * q * KC * 4);
    L.idx = take(ns * q * 2 * 4); L.dist = take(ns * q * 2 * 8); L.bidx = take(ns * q * 2 * 4); L.bdist = take(ns * q * 2 * 8);
    L.cand_q = take(ns * q * 4); L.cand_m = take(ns * q * 4); L.n_cand = take(ns * 4); L.n_flag = take(ns * 4);
    L.flag_list = take(ns * q * 4); L.flag2 = take(ns * q * 4); L.n_flag2 = take(ns * 4);
    L.fpi = take(ns * kSegFbSlices * q * 2 * 4); L.fpd = take(ns * kSegFbSlices * q * 2 * 8);
    L.grows = take(ns * q * kNG * sizeof(SegGroupRows)); L.gnorms = take(ns * q * kNG * sizeof(SegGroupNorms)); L.qinfo = take(ns * q * sizeof(SegQueryInfo));
    L.refpart = take((kRefParts * 4 + 1) * 8); L.rowmin = take(vm * 4);
    L.psum = take(ns * q * kNG * kNC * 8); L.map = take(ns * q * kNG * 4); L.n_slots = take(ns * 4); L.more_list = take(ns * q * 4); L.n_more = take(ns * 4);
    L.total = b;
    return L;
}
}  // namespace

// ---- bounded workspace --------------------------------------------------------------------------------------------------
// The one-chain layout grows with VM x Q (the shared score matrix) and with S x Q x splits (the per-segment lists): 0.85 GB at the
// sweep's shape (60 k x 2 k, 329 spheres), tens of GB at 400 k x 6 k with 2000 spheres.  Above kSegBudget + kSegGatherBytes the
// call runs in BATCHES of consecutive segments: the model rows a batch names are marked, numbered in ascending order (their
// union) and gathered into a compact sub-model, the batch's lists are renumbered into it, and the one-chain form runs on
// (surface, sub-model, batch).  Spheres that follow one another in the sweep's grid order overlap heavily, so a batch's union is
// a small multiple of one sphere.  The pairs do not change: a pair's model index counts within its segment, and every exact
// distance is computed from the same rows by the same arithmetic.  Price of the bound: one host synchronisation at entry (the
// segment offsets) and one or two per batch (the size of its union); a model row is scored once per batch that names it.
constexpr size_t kSegBudget = (size_t)3 << 30;          // the one-chain workspace of a batch
constexpr size_t kSegGatherBytes = (size_t)1 << 30;     // the gathered sub-model of a batch
static size_t seg_batched_fixed_bytes(int VM, int tot, int S) {
    const size_t vm = (size_t)std::max(VM, 1);
    return 3 * align_up(vm * 4, 256) + align_up((size_t)std::max(tot, 1) * 4, 256) + align_up(((size_t)std::max(S, 1) + 1) * 4, 256) +
           align_up((vm + 1023) / 1024 * 4, 256) + 256;
}
size_t get_matches_segmented_workspace_bytes(int Q, int VM, int D, int S, int tot, int n_max) {
    const size_t one = seg_layout(Q, VM, D, D + 1, S, tot, n_max).total;
    if (one <= kSegBudget + kSegGatherBytes) return one;
    return kSegBudget + kSegGatherBytes + seg_batched_fixed_bytes(VM, tot, S);
}

// descS [Q][D], descM [VM][D] row-major doubles; segment z = model rows seg_rows[seg_off[z] .. seg_off[z + 1]) (0-based, ascending;
// seg_off [S + 1] on the device, tot = seg_off[S] and n_max = the longest segment known to the host).  pairs_all [S][Q][2]
// (1-based; the model index counts within the segment), n_pairs [S], metric_all [S][Q] or null.  SAD only.
static int launch_get_matches_segmented_one(const double* descS, int Q, const double* descM, int VM, int D, const int32_t* seg_rows,
                                            const int32_t* seg_off, int S, int tot, int n_max, const pcreg_match_opts& o, uint32_t* pairs_all,
                                            double* metric_all, int32_t* n_pairs, void* ws, size_t ws_bytes, hipStream_t st,
                                            const SegPreparedModel* prepared = nullptr) {
    if (S <= 0) return PCREG_OK;
    if (Q <= 0 || n_max <= 0 || VM <= 0) { PCREG_HIP(hipMemsetAsync(n_pairs, 0, (size_t)S * sizeof(int32_t), st)); return PCREG_OK; }
    const int Dp = D + (o.unnormalize ? 1 : 0);
    const SegLayout L = seg_layout(Q, VM, D, D + 1, S, tot, n_max);
    if (ws_bytes < L.total) { set_error("segmented get_matches workspace too small: %zu < %zu", ws_bytes, L.total); return PCREG_E_WORKSPACE; }
    if (L.splits * KC > 64 * kEPL) { set_error("segmented get_matches: %d list entries per query exceed the finalize wave", L.splits * KC); return PCREG_E_ARG; }
    char* w = (char*)ws;
    double *PS = (double*)(w + L.PS), *PM = (double*)(w + L.PM), *rS = (double*)(w + L.rowS), *rM = (double*)(w + L.rowM);
    double *nrefS = (double*)(w + L.nrefS), *nrefM = (double*)(w + L.nrefM);
    SegConst* sc = (SegConst*)(w + L.sc);
    double *nrmS = (double*)(w + L.nrmS), *nrmM = (double*)(w + L.nrmM);
    uint32_t *Aq = (uint32_t*)(w + L.Aq), *Bq = (uint32_t*)(w + L.Bq), *Sc = (uint32_t*)(w + L.Sc);
    int32_t* part_idx = (int32_t*)(w + L.part_idx); uint32_t* part_s = (uint32_t*)(w + L.part_s);
    int32_t *idx = (int32_t*)(w + L.idx), *bidx = (int32_t*)(w + L.bidx); double *dist = (double*)(w + L.dist), *bdist = (double*)(w + L.bdist);
    int32_t *cand_q = (int32_t*)(w + L.cand_q), *cand_m = (int32_t*)(w + L.cand_m), *n_cand = (int32_t*)(w + L.n_cand), *n_flag = (int32_t*)(w + L.n_flag);
    int32_t* flag_list = (int32_t*)(w + L.flag_list); int32_t* fpi = (int32_t*)(w + L.fpi); double* fpd = (double*)(w + L.fpd);
    int32_t *flag2 = (int32_t*)(w + L.flag2), *n_flag2 = (int32_t*)(w + L.n_flag2);
    const size_t q = (size_t)Q, vm = (size_t)VM;

    // once for all segments: powers, row scalars, the reference constant, the two quantised operands, ALL approximate scores
    hipLaunchKernelGGL(segp_rows_kernel, dim3((Q + kPR - 1) / kPR), dim3(kBlock), 0, st, descS, Q, D, o.change_metric, o.metric_factor, PS, rS, rS + q, rS + 2 * q, rS + 3 * q, rS + 4 * q, rS + 5 * q);
    if (prepared && prepared->VM == VM && prepared->D == D && prepared->change_metric == o.change_metric &&
        (!o.change_metric || prepared->metric_factor == o.metric_factor)) {
        PM = const_cast<double*>(prepared->P); rM = const_cast<double*>(prepared->r);          // read-only from here on
    } else {
        hipLaunchKernelGGL(segp_rows_kernel, dim3((VM + kPR - 1) / kPR), dim3(kBlock), 0, st, descM, VM, D, o.change_metric, o.metric_factor, PM, rM, rM + vm, rM + 2 * vm, rM + 3 * vm, rM + 4 * vm, rM + 5 * vm);
    }
    const SegSets sets{PS, rS, rS + q, rS + 2 * q, rS + 3 * q, rS + 4 * q, rS + 5 * q, PM, rM, rM + vm, rM + 2 * vm, rM + 3 * vm, rM + 4 * vm, rM + 5 * vm,
                       seg_rows, seg_off, Q, VM, D, Dp, (size_t)S * q, (size_t)std::max(tot, 1)};
    hipLaunchKernelGGL(segp_cc_kernel, dim3(S), dim3(kBlock), 0, st, sets, o, sc);
    {
        const int n_parts = std::max(1, std::min(kRefParts, (Q + VM + kBlock - 1) / kBlock));
        double* refpart = (double*)(w + L.refpart);
        hipLaunchKernelGGL(segp_ref_kernel, dim3(n_parts), dim3(kBlock), 0, st, sets, o, nrefS, nrefM, (const SegConst*)sc, S, refpart);
        hipLaunchKernelGGL(segp_ref_final_kernel, dim3(1), dim3(kBlock), 0, st, (const double*)refpart, n_parts, sc + S);
    }
    hipLaunchKernelGGL(segp_quantize_ref_kernel, dim3(L.ldqa / 64, L.D2p / 32), dim3(kBlock), 0, st, PS, nrefS, Q, D, Dp, sc + S, L.D2p, L.ldqa, Aq);
    hipLaunchKernelGGL(segp_quantize_ref_kernel, dim3(L.ldqb / 64, L.D2p / 32), dim3(kBlock), 0, st, PM, nrefM, VM, D, Dp, sc + S, L.D2p, L.ldqb, Bq);
    {
        const int n_tiles = (Q + BQ - 1) / BQ, row_tiles = (VM + BM - 1) / BM;
        const int sm = std::max(1, std::min(row_tiles, 6144 / std::max(n_tiles, 1)));          // ~8 rounds of resident workgroups (measured: 2 rounds 2.38 ms, 4: 2.28, 8: 2.24, 16: 2.22)
        const int ct = (row_tiles + sm - 1) / sm, s_eff = (row_tiles + ct - 1) / ct;
        hipLaunchKernelGGL(sad16_candidates_kernel<kSadMatrix>, dim3(n_tiles, s_eff), dim3(kBlock), 0, st, Aq, Q, L.ldqa, Bq, VM, L.ldqb, L.D2p, ct * BM,
                           (int32_t*)nullptr, (uint32_t*)nullptr, (unsigned long long*)nullptr, (const int32_t*)nullptr, SegZ{0, 0, 0, 0, nullptr, Sc, L.ldqa});
    }
    // per segment: constants and norms, lists out of the matrix, exact re-rank, exact rows for the unproven
    hipLaunchKernelGGL(segp_consts_kernel, dim3(S), dim3(kBlock), 0, st, sets, o, nrefS, nrefM, nrmS, nrmM, sc, S);
    PCREG_HIP(hipMemsetAsync(n_flag, 0, (size_t)S * sizeof(int32_t), st));
    PCREG_HIP(hipGetLastError());
    // PCREG_MATCH_FORCE_FALLBACK: 1 = every query counts as unproven (they take the refinement), 2 = and the refinement passes
    // them all on to the exhaustive kernel
    const int force = debug_flag(kDbgMatchForceFallback) != 0, skip_refine = debug_flag(kDbgMatchForceFallback) == 2;
    hipLaunchKernelGGL(segp_select_kernel, dim3(L.ldqa / 64, (L.splits + 3) / 4, S), dim3(kBlock), 0, st, Sc, L.ldqa, seg_rows, seg_off, Q, L.chunk, L.splits, part_idx, part_s);
    int32_t* dbg_hist = nullptr;
    unsigned long long* stats = match_stats_dev();
    if (stats) { hipLaunchKernelGGL(stats_bump_kernel, dim3(1), dim3(1), 0, st, stats, 6, 1ull); hipLaunchKernelGGL(stats_bump_kernel, dim3(1), dim3(1), 0, st, stats, 7, (unsigned long long)S); }
#ifdef PCREG_EXPERIMENTS
    if (debug_flag(kDbgSegDebug)) { PCREG_HIP(hipMalloc((void**)&dbg_hist, 2 * 130 * sizeof(int32_t))); PCREG_HIP(hipMemsetAsync(dbg_hist, 0, 2 * 130 * sizeof(int32_t), st)); }
#endif
    if (debug_flag(kDbgSegWaveFinalize) || dbg_hist) {
        hipLaunchKernelGGL(segp_finalize_kernel<false>, dim3((Q + 3) / 4, 1, S), dim3(kBlock), 0, st, sets, nrmS, nrmM, sc, (const int32_t*)nullptr, (const int32_t*)nullptr,
                           part_idx, part_s, L.splits, idx, dist, flag_list, n_flag, force, (o.matchThreshold * 0.01) * (2.0 * sqrt((double)Dp)), o.maxRatio, dbg_hist, stats);
    } else {
        SegGroupRows* grows = (SegGroupRows*)(w + L.grows); SegGroupNorms* gnorms = (SegGroupNorms*)(w + L.gnorms); SegQueryInfo* qinfo = (SegQueryInfo*)(w + L.qinfo);
        double* psum = (double*)(w + L.psum); int32_t *map = (int32_t*)(w + L.map), *n_slots = (int32_t*)(w + L.n_slots);
        int32_t *more_list = (int32_t*)(w + L.more_list), *n_more = (int32_t*)(w + L.n_more);
        PCREG_HIP(hipMemsetAsync(n_more, 0, (size_t)S * sizeof(int32_t), st));
        hipLaunchKernelGGL(segp_pick_kernel, dim3((Q + 3) / 4, 1, S), dim3(kBlock), 0, st, sets, nrmS, nrmM, sc, part_idx, part_s, L.splits, idx, dist, force,
                           (o.matchThreshold * 0.01) * (2.0 * sqrt((double)Dp)), o.maxRatio, grows, gnorms, qinfo);
        const int nb = (Q * kNG + kRQ * kRW - 1) / (kRQ * kRW);          // workgroups per segment; segments in groups of 8 (one per XCD)
        for (int round = 1; round <= 2; ++round) {                       // the first groups; then the second groups their exact second distance could not exclude
            hipLaunchKernelGGL(segp_scan_kernel, dim3(S), dim3(kBlock), 0, st, (const SegQueryInfo*)qinfo, Q, (const int32_t*)nullptr, round, map, n_slots);
            hipLaunchKernelGGL(segp_rerank_pairs_kernel, dim3((unsigned)(((S + 7) / 8) * 8 * nb)), dim3(64 * kRW), 0, st, (const double*)PS, (const double*)PM, Q, D, Dp, sc,
                               grows, gnorms, map, n_slots, nb, S, psum);
            hipLaunchKernelGGL(segp_decide_kernel, dim3((Q + kBlock - 1) / kBlock, 1, S), dim3(kBlock), 0, st, sc, qinfo, psum, Q, Dp, round, idx, dist, flag_list, n_flag,
                               more_list, n_more, force, stats);
        }
        hipLaunchKernelGGL(segp_decide_more_kernel, dim3(16, 1, S), dim3(kBlock), 0, st, sets, nrmS, nrmM, sc, part_idx, part_s, L.splits, qinfo, psum, more_list, n_more,
                           idx, dist, flag_list, n_flag, force, stats);
    }
    const int slice_f = (n_max + kSegFbSlices - 1) / kSegFbSlices;
    PCREG_HIP(hipMemsetAsync(n_flag2, 0, (size_t)S * sizeof(int32_t), st));
    hipLaunchKernelGGL(segp_refine_kernel<false>, dim3(16, 1, S), dim3(kBlock), 0, st, sets, nrmS, nrmM, sc, (const int32_t*)nullptr, (const uint32_t*)Sc, L.ldqa,
                       (const int32_t*)flag_list, (const int32_t*)n_flag, idx, dist, flag2, n_flag2, skip_refine, stats);
    hipLaunchKernelGGL(segp_exact_rows_kernel<false>, dim3(std::min(Q, 16), kSegFbSlices, S), dim3(kBlock), (size_t)Dp * sizeof(double), st, sets, nrmS, nrmM, sc,
                       (const int32_t*)nullptr, flag2, n_flag2, slice_f, fpi, fpd);
    hipLaunchKernelGGL(segp_fallback_finish_kernel, dim3(std::min((Q + 255) / 256, 8), 1, S), dim3(256), 0, st, flag2, n_flag2, Q, fpi, fpd, idx, dist);
#ifdef PCREG_EXPERIMENTS
    std::vector<int32_t> dbg_nf_fwd;
    if (dbg_hist) { dbg_nf_fwd.resize((size_t)S); PCREG_HIP(hipMemcpyAsync(dbg_nf_fwd.data(), n_flag, (size_t)S * 4, hipMemcpyDeviceToHost, st)); }
#endif
    const double maxval = 2.0 * sqrt((double)Dp);                 // percentToLevel, SAD
    const double thr = (o.matchThreshold * 0.01) * maxval;
    hipLaunchKernelGGL(segp_filter_kernel, dim3(S), dim3(kCompactThreads), 0, st, idx, dist, Q, seg_off, thr, o.maxRatio, cand_q, cand_m, n_cand, n_flag);
    PCREG_HIP(hipGetLastError());
    if (o.unique) {
        // back: every candidate's model row against the surface rows that can still beat its own query -- a row of the same matrix
        PCREG_HIP(hipMemsetAsync(n_flag2, 0, (size_t)S * sizeof(int32_t), st));
        if (debug_flag(kDbgSegWaveFinalize)) {
            hipLaunchKernelGGL(segp_back_direct_kernel, dim3(std::max(1, std::min((Q + 3) / 4, 128)), 1, S), dim3(kBlock), 0, st, sets, nrmS, nrmM, sc,
                               (const int32_t*)cand_q, (const int32_t*)cand_m, (const int32_t*)n_cand, (const uint32_t*)Sc, L.ldqa, (const double*)dist, bidx, bdist,
                               flag2, n_flag2, skip_refine, stats);
        } else {
            SegGroupRows* grows = (SegGroupRows*)(w + L.grows); SegGroupNorms* gnorms = (SegGroupNorms*)(w + L.gnorms); SegQueryInfo* qinfo = (SegQueryInfo*)(w + L.qinfo);
            double* psum = (double*)(w + L.psum); int32_t *map = (int32_t*)(w + L.map), *n_slots = (int32_t*)(w + L.n_slots);
            uint32_t* rowmin = (uint32_t*)(w + L.rowmin);
            hipLaunchKernelGGL(segp_rowmin_kernel, dim3((VM + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, st, (const uint32_t*)Sc, L.ldqa, Q, VM, rowmin);
            hipLaunchKernelGGL(segp_back_pick_kernel, dim3(std::max(1, std::min((Q + 3) / 4, 128)), 1, S), dim3(kBlock), 0, st, sets, nrmS, nrmM, sc,
                               (const int32_t*)cand_q, (const int32_t*)cand_m, (const int32_t*)n_cand, (const uint32_t*)Sc, L.ldqa, (const uint32_t*)rowmin, (const double*)dist, bidx, bdist,
                               flag2, n_flag2, skip_refine, grows, gnorms, qinfo, stats);
            hipLaunchKernelGGL(segp_scan_kernel, dim3(S), dim3(kBlock), 0, st, (const SegQueryInfo*)qinfo, Q, (const int32_t*)n_cand, 0, map, n_slots);
            const int nb = (Q * kNG + kRQ * kRW - 1) / (kRQ * kRW);
            hipLaunchKernelGGL(segp_rerank_pairs_kernel, dim3((unsigned)(((S + 7) / 8) * 8 * nb)), dim3(64 * kRW), 0, st, (const double*)PM, (const double*)PS, Q, D, Dp, sc,
                               grows, gnorms, map, n_slots, nb, S, psum);
            hipLaunchKernelGGL(segp_back_decide_kernel, dim3((Q + kBlock - 1) / kBlock, 1, S), dim3(kBlock), 0, st, qinfo, psum, (const int32_t*)n_cand, Q, bidx, bdist);
        }
        hipLaunchKernelGGL(segp_exact_rows_kernel<true>, dim3(std::min(Q, 16), kSegFbSlices, S), dim3(kBlock), (size_t)Dp * sizeof(double), st, sets, nrmS, nrmM, sc,
                           cand_m, flag2, n_flag2, 0, fpi, fpd);
        hipLaunchKernelGGL(segp_fallback_finish_kernel, dim3(std::min((Q + 255) / 256, 8), 1, S), dim3(256), 0, st, flag2, n_flag2, Q, fpi, fpd, bidx, bdist);
    }
    hipLaunchKernelGGL(segp_emit_kernel, dim3(S), dim3(kBlock), 0, st, cand_q, cand_m, n_cand, bidx, o.unique ? 1 : 0, dist, Q, pairs_all, metric_all, n_pairs);
    PCREG_HIP(hipGetLastError());
#ifdef PCREG_EXPERIMENTS
    if (dbg_hist) {            // candidates re-scored per query (forward, back) and unproven queries: the only host round trip, debugging only
        std::vector<int32_t> h(2 * 130), nf((size_t)S);
        PCREG_HIP(hipStreamSynchronize(st));
        PCREG_HIP(hipMemcpy(h.data(), dbg_hist, h.size() * 4, hipMemcpyDeviceToHost));
        PCREG_HIP(hipMemcpy(nf.data(), n_flag, nf.size() * 4, hipMemcpyDeviceToHost));
        (void)hipFree(dbg_hist);
        {
            long long q = 0, c = 0; for (int k = 0; k < 130; ++k) { q += h[k]; c += (long long)k * h[k]; }
            fprintf(stderr, "[pcreg] segmented forward: %lld queries, %.2f candidates inside the two-sided slack per query; histogram 0..8:", q, q ? (double)c / q : 0.0);
            for (int k = 0; k <= 8; ++k) fprintf(stderr, " %d", h[k]);
            long long big = 0; for (int k = 9; k < 130; ++k) big += h[k];
            fprintf(stderr, " >8: %lld\n", big);
        }
        long long f = 0, ff = 0; int mx = 0, nz = 0; for (int z = 0; z < S; ++z) { f += nf[z]; ff += dbg_nf_fwd[z]; mx = std::max(mx, dbg_nf_fwd[z]); nz += dbg_nf_fwd[z] > 0; }
        fprintf(stderr, "[pcreg] segmented: unproven queries forward %lld (in %d segments, at most %d in one)\n", ff, nz, mx); (void)f;
    }
#endif
    return PCREG_OK;
}

namespace {
// flag[r] = 1 for every model row the batch's lists name
__global__ void segb_mark_kernel(const int32_t* __restrict__ rows, int lo, int hi, int32_t* __restrict__ flag) {
    const int i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < hi) flag[rows[i]] = 1;
}
// the union's numbering = an exclusive scan of the flags in three deterministic steps (no atomics): counts per 1024 rows, their
// scan by one workgroup, the numbers
__global__ __launch_bounds__(256) void segb_count_kernel(const int32_t* __restrict__ flag, int VM, int32_t* __restrict__ bcnt) {
    const int i = blockIdx.x * 1024 + threadIdx.x * 4;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) c += (i + k < VM) ? flag[i + k] : 0;
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) c += __shfl_xor(c, o_);
    __shared__ int sw[4];
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) bcnt[blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}
__global__ __launch_bounds__(1024) void segb_scan_kernel(int32_t* __restrict__ bcnt, int nb, int32_t* __restrict__ n_union) {
    __shared__ int sh[1024];
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const int v = i < nb ? bcnt[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o_ = 1; o_ < 1024; o_ <<= 1) {
            const int t = (int)threadIdx.x >= o_ ? sh[threadIdx.x - o_] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nb) bcnt[i] = carry + sh[threadIdx.x] - v;
        const int tot = sh[1023];
        __syncthreads();
        carry += tot;
    }
    if (threadIdx.x == 0) *n_union = carry;
}
__global__ __launch_bounds__(256) void segb_number_kernel(const int32_t* __restrict__ flag, int VM, const int32_t* __restrict__ bcnt,
                                                          int32_t* __restrict__ newidx, int32_t* __restrict__ urows) {
    const int i = blockIdx.x * 1024 + threadIdx.x * 4;
    int f[4], c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { f[k] = (i + k < VM) ? flag[i + k] : 0; c += f[k]; }
    int incl = c;
#pragma unroll
    for (int o_ = 1; o_ < 64; o_ <<= 1) { const int t = __shfl_up(incl, o_); if ((int)(threadIdx.x & 63) >= o_) incl += t; }
    __shared__ int sw[4];
    if ((threadIdx.x & 63) == 63) sw[threadIdx.x >> 6] = incl;
    __syncthreads();
    int pos = bcnt[blockIdx.x] + incl - c;
    for (int w_ = 0; w_ < (int)(threadIdx.x >> 6); ++w_) pos += sw[w_];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i + k < VM) { newidx[i + k] = f[k] ? pos : -1; if (f[k]) urows[pos] = i + k; }
        pos += f[k];
    }
}
__global__ void segb_renumber_kernel(const int32_t* __restrict__ rows, int lo, int hi, const int32_t* __restrict__ newidx, int32_t* __restrict__ out) {
    const int i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < hi) out[i - lo] = newidx[rows[i]];
}
__global__ void segb_offsets_kernel(const int32_t* __restrict__ seg_off, int z0, int Sb, int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= Sb) out[i] = seg_off[z0 + i] - seg_off[z0];
}
// dst[u][:] = src[urows[u]][:], D doubles per row, one wave per row
__global__ __launch_bounds__(256) void segb_gather_kernel(const double* __restrict__ src, int D, const int32_t* __restrict__ urows, int n,
                                                          double* __restrict__ dst) {
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (u >= n) return;
    const double* a = src + (size_t)urows[u] * D; double* b = dst + (size_t)u * D;
    for (int d = lane; d < D; d += 64) b[d] = a[d];
}
}  // namespace

size_t segmented_prepared_model_bytes(int VM, int D) { const size_t vm = (size_t)std::max(VM, 1); return (vm * (size_t)D + 6 * vm) * sizeof(double); }
int launch_segmented_prepare_model(const double* descM_rows, int VM, int D, const pcreg_match_opts& o, double* P, double* r, hipStream_t st) {
    if (VM <= 0) return PCREG_OK;
    const size_t vm = (size_t)VM;
    hipLaunchKernelGGL(segp_rows_kernel, dim3((VM + kPR - 1) / kPR), dim3(kBlock), 0, st, descM_rows, VM, D, o.change_metric, o.metric_factor, P, r, r + vm, r + 2 * vm,
                       r + 3 * vm, r + 4 * vm, r + 5 * vm);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}
int launch_get_matches_segmented(const double* descS, int Q, const double* descM, int VM, int D, const int32_t* seg_rows,
                                 const int32_t* seg_off, int S, int tot, int n_max, const pcreg_match_opts& o, uint32_t* pairs_all,
                                 double* metric_all, int32_t* n_pairs, void* ws, size_t ws_bytes, hipStream_t st, const SegPreparedModel* prepared) {
    if (S <= 0) return PCREG_OK;
    if (Q <= 0 || n_max <= 0 || VM <= 0) { PCREG_HIP(hipMemsetAsync(n_pairs, 0, (size_t)S * sizeof(int32_t), st)); return PCREG_OK; }
    const size_t one = seg_layout(Q, VM, D, D + 1, S, tot, n_max).total;
    const size_t budget = PCREG_EXP_ENV("PCREG_SEG_BUDGET_MB", 0) > 0 ? (size_t)PCREG_EXP_ENV("PCREG_SEG_BUDGET_MB", 0) << 20 : kSegBudget;
    if (one <= kSegBudget + kSegGatherBytes && !debug_flag(kDbgSegBatched))
        return launch_get_matches_segmented_one(descS, Q, descM, VM, D, seg_rows, seg_off, S, tot, n_max, o, pairs_all, metric_all, n_pairs, ws, ws_bytes, st, prepared);
    // ---- batches of consecutive segments on gathered sub-models (they gather the RAW rows: a prepared model is not used)
    const size_t fixed = seg_batched_fixed_bytes(VM, tot, S);
    if (ws_bytes < fixed + 4096) { set_error("segmented get_matches workspace too small: %zu bytes", ws_bytes); return PCREG_E_WORKSPACE; }
    // whatever the caller's workspace holds beyond the fixed part is split 1 : 3 between the gathered rows and the one-chain form
    // (the workspace query returns the fixed part + kSegGatherBytes + kSegBudget for a problem that needs batches)
    const size_t room = ws_bytes - fixed;
    const size_t gather_bytes = std::min(kSegGatherBytes, room / 4) / 256 * 256, one_bytes = std::min(budget, room - gather_bytes);
    const size_t vm = (size_t)VM;
    char* w = (char*)ws;
    int32_t* flag = (int32_t*)w; w += align_up(vm * 4, 256);
    int32_t* newidx = (int32_t*)w; w += align_up(vm * 4, 256);
    int32_t* urows = (int32_t*)w; w += align_up(vm * 4, 256);
    int32_t* rows_b = (int32_t*)w; w += align_up((size_t)std::max(tot, 1) * 4, 256);
    int32_t* off_b = (int32_t*)w; w += align_up(((size_t)S + 1) * 4, 256);
    int32_t* bcnt = (int32_t*)w; w += align_up((vm + 1023) / 1024 * 4, 256);
    int32_t* n_union = (int32_t*)w; w += 256;
    double* descMb = (double*)w; w += gather_bytes;
    void* ws_one = w;
    const int nb = (VM + 1023) / 1024;
    const size_t cap_rows = gather_bytes / ((size_t)D * 8);
    std::vector<int32_t> off((size_t)S + 1);
    PCREG_HIP(hipMemcpyAsync(off.data(), seg_off, ((size_t)S + 1) * 4, hipMemcpyDeviceToHost, st));
    PCREG_HIP(hipStreamSynchronize(st));
    if (off[S] != tot) { set_error("segmented get_matches: seg_off[S] = %d, total_rows = %d", off[S], tot); return PCREG_E_ARG; }
    int z0 = 0;
    while (z0 < S) {
        int Sb = S - z0;
        for (;;) {                                   // the largest batch (halving) whose one-chain form fits
            int nmax_b = 0;
            for (int z = z0; z < z0 + Sb; ++z) nmax_b = std::max(nmax_b, off[z + 1] - off[z]);
            const int tot_b = off[z0 + Sb] - off[z0];
            if (tot_b == 0) { PCREG_HIP(hipMemsetAsync(n_pairs + z0, 0, (size_t)Sb * 4, st)); break; }
            // first the terms that do not depend on the union (the per-segment lists), on the host alone
            if (Sb > 1 && seg_layout(Q, 1, D, D + 1, Sb, tot_b, nmax_b).total > one_bytes / 2) { Sb = (Sb + 1) / 2; continue; }
            int VMb = 0;
            PCREG_HIP(hipMemsetAsync(flag, 0, vm * 4, st));
            hipLaunchKernelGGL(segb_mark_kernel, dim3((tot_b + 255) / 256), dim3(256), 0, st, seg_rows, off[z0], off[z0 + Sb], flag);
            hipLaunchKernelGGL(segb_count_kernel, dim3(nb), dim3(256), 0, st, flag, VM, bcnt);
            hipLaunchKernelGGL(segb_scan_kernel, dim3(1), dim3(1024), 0, st, bcnt, nb, n_union);
            PCREG_HIP(hipMemcpyAsync(&VMb, n_union, 4, hipMemcpyDeviceToHost, st));
            PCREG_HIP(hipStreamSynchronize(st));
            const size_t lay = seg_layout(Q, VMb, D, D + 1, Sb, tot_b, nmax_b).total;
            if (lay > one_bytes || (size_t)VMb > cap_rows) {
                if (Sb == 1) {
                    set_error("segmented get_matches: one segment of %d rows against %d queries needs %zu bytes; the workspace leaves %zu", nmax_b, Q, lay, one_bytes);
                    return PCREG_E_WORKSPACE;
                }
                Sb = (Sb + 1) / 2;
                continue;
            }
            hipLaunchKernelGGL(segb_number_kernel, dim3(nb), dim3(256), 0, st, flag, VM, bcnt, newidx, urows);
            hipLaunchKernelGGL(segb_renumber_kernel, dim3((tot_b + 255) / 256), dim3(256), 0, st, seg_rows, off[z0], off[z0 + Sb], newidx, rows_b);
            hipLaunchKernelGGL(segb_offsets_kernel, dim3((Sb + 256) / 256), dim3(256), 0, st, seg_off, z0, Sb, off_b);
            hipLaunchKernelGGL(segb_gather_kernel, dim3((VMb + 3) / 4), dim3(256), 0, st, descM, D, urows, VMb, descMb);
            PCREG_HIP(hipGetLastError());
            const int rc = launch_get_matches_segmented_one(descS, Q, descMb, VMb, D, rows_b, off_b, Sb, tot_b, nmax_b, o, pairs_all + (size_t)z0 * Q * 2,
                                                            metric_all ? metric_all + (size_t)z0 * Q : nullptr, n_pairs + z0, ws_one, one_bytes, st);
            if (rc) return rc;
            break;
        }
        z0 += Sb;
    }
    return PCREG_OK;
}

}  // namespace pcreg
