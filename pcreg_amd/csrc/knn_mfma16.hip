// pcreg_amd/csrc/knn_mfma16.hip -- the candidate stage of knn_fast.hip on the f16 matrix cores.
//
// s(q,m) = |m~|^2 - 2 q~.m~ is a dot product; v_mfma_f32_32x32x16_f16 scores 32 model points x 32
// queries (1024 pairs) per instruction at 16x the fp32 vector rate -- IF the operands are f16.  They
// are made f16 without giving up fp32-level accuracy by error-free splitting:
//     x = xh + xl (+ <= 2^-22 |x|),  xh = f16(x), xl = f16(x - xh)
//     Q.m = Qh mh + Qh ml + Ql mh + Ql ml          (Q = -2 sigma q~, m = sigma m~)
// i.e. 4 k-slots per coordinate, 12 in all, and |m~|^2 (fp32) as three f16 terms against 1,1,1:
// 15 of the instruction's 16 k-slots, ONE MFMA per 32 x 32 tile, accumulated in fp32 by the matrix
// core.  sigma is a power of two that brings the joint bounding box into [-64, 64): every entry
// stays inside f16's range and scores are exactly sigma^2 times the unscaled ones.
//     A (model, 32 x 16):   lane l gives point l%32, k-slots 8*(l/32)..+7
//                           k: [x: mh ml mh ml][y: ...] | [z: mh ml mh ml][w: wh wm wl 0]
//     B (queries, 16 x 32): lane l gives query l%32, the same k-slots
//                           k: [x: Qh Qh Ql Ql][y: ...] | [z: Qh Qh Ql Ql][1 1 1 0]
//     D (32 x 32 fp32):     lane l, register r: model row 8*(r/4) + 4*(l/32) + r%4, query l%32
// A lane thus sees 16 scores of ONE query per instruction; it folds them with a v_min3 tree and one
// compare against the query's threshold and only then touches its sorted top-4 -- the VALU work per
// pair drops from ~4.25 issue slots (3 FMA + selection) to ~1.1, and the FMAs moved to the matrix
// pipe, which runs concurrently.  The scores are approximations with a larger (but still rigorous)
// error bound than the fp32 FMA chain (knn_finalize_kernel, e_mode 1):
// nothing downstream changes -- candidates are re-ranked exactly and certified, failures fall back.
// Built with -fno-honor-nans and -amdgpu-mfma-vgpr-form (results straight into VGPRs).
#include "common.hpp"
#include "knn_fast_common.hpp"
#include <hip/hip_fp16.h>
#include <cstdlib>
#include <vector>
#include <utility>

namespace pcreg {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef PCREG_KT16
#define PCREG_KT16 512
#endif
constexpr int kT16 = PCREG_KT16;              // model points per tile: [2 k-halves][512 points][8 f16] = 16 KiB (768 / 1024 measured in round 3: see DESIGN 4.1)
#ifndef PCREG_KREFRESH
#define PCREG_KREFRESH 16
#endif
constexpr int kRefresh = PCREG_KREFRESH;       // tiles between two looks at the shared threshold words (a power of two)

// The error-free split needs ONE f16 rounding of ONE fp32 value: the stored high part and the high part the
// residual is taken against must be the same number.  hipcc folds `(_Float16)fma(a, b, c)` into v_fma_mixlo_f16 (the
// product-sum rounded ONCE, straight to f16) where the value feeds a conversion, while a second use of the same
// expression goes through the fp32 result (rounded twice): at an f16 rounding midpoint the two differ by one f16 ulp
// -- |m~|^2 = 216.9375 came out 0.125 too small, the point dropped out of its group's minimum, and 6 of 50 000
// queries of cfg 5's crop 0 got a wrong 2nd neighbour WITH a passing certificate (round 1 had the same bug; the
// full-size test tests/test_gpu_fullsize.py found it).  The empty asm makes the value opaque: no fold, one rounding.
__device__ __forceinline__ float opaque_f32(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ void split2(float x, _Float16& h, _Float16& l) {
    x = opaque_f32(x);
    h = (_Float16)x; l = (_Float16)opaque_f32(x - (float)h);
}

// model -> tiles of f16 operands; R_m^2 (unscaled) by atomicMax on the float bits; and the model-wide seeding grid
// (knn_fast.hip, stage 1c): up to kSeedSlots points per cell, whoever arrives first -- the thresholds it yields are
// hints, results never depend on them.  One pass over the model, once per PREPARED model.
__global__ __launch_bounds__(kBlock) void prep_model_f16_kernel(const float* __restrict__ m, int M, int ldm,
                                                                const Prep* __restrict__ prep, uint4* __restrict__ out,
                                                                int n_tiles, unsigned* __restrict__ rm2_bits,
                                                                int32_t* __restrict__ seed_cnt, float4* __restrict__ seed_slots) {
    const float cx = prep->cx, cy = prep->cy, cz = prep->cz, sg = prep->sigma;
    const float gx0 = prep->gx0, gy0 = prep->gy0, gz0 = prep->gz0, ih = prep->inv_h;
    const int nx = prep->nx, ny = prep->ny, nz = prep->nz;
    float mx = 0.0f;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n_tiles * kT16; i += gridDim.x * kBlock) {
        union { f16x8 v; uint4 u; } lo, hi;
        if (i < M) {
            const float px = m[i], py = m[i + (size_t)ldm], pz = m[i + 2 * (size_t)ldm];
            if (seed_cnt) {
                const float fx = floorf((px - gx0) * ih), fy = floorf((py - gy0) * ih), fz = floorf((pz - gz0) * ih);
                if (fx >= 0.0f && fx < (float)nx && fy >= 0.0f && fy < (float)ny && fz >= 0.0f && fz < (float)nz) {
                    const int cell = ((int)fz * ny + (int)fy) * nx + (int)fx;
                    const int k = atomicAdd(&seed_cnt[cell], 1);
                    if (k < kSeedSlots) seed_slots[(size_t)cell * kSeedSlots + k] = make_float4(px, py, pz, 0.0f);   // one 64-B line per cell
                }
            }
            float x = px - cx, y = py - cy, z = pz - cz;     // the m~ of the fp32 path
            mx = fmaxf(mx, __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));
            x *= sg; y *= sg; z *= sg;                                                         // exact
            const float w = opaque_f32(__builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));    // <= 3 * 64^2; opaque: see split2
            _Float16 xh, xl, yh, yl, zh, zl;
            split2(x, xh, xl); split2(y, yh, yl); split2(z, zh, zl);
            const _Float16 wh = (_Float16)w; const float w1 = opaque_f32(w - (float)wh);
            const _Float16 wm = (_Float16)w1; const _Float16 wl = (_Float16)opaque_f32(w1 - (float)wm);
            lo.v = f16x8{xh, xl, xh, xl, yh, yl, yh, yl};
            hi.v = f16x8{zh, zl, zh, zl, wh, wm, wl, (_Float16)0.0f};
        } else {                                   // padding: score +inf, never a candidate
            lo.u = make_uint4(0u, 0u, 0u, 0u);
            hi.u = make_uint4(0u, 0u, 0x00007C00u, 0u);
        }
        const size_t t = (size_t)(i / kT16), r = (size_t)(i % kT16);
        out[t * (2 * kT16) + r] = lo.u;
        out[t * (2 * kT16) + kT16 + r] = hi.u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    __shared__ float s_mx[kBlock / 64];                  // one atomic per workgroup: thousands on one word serialise
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(rm2_bits, __float_as_uint(fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]))));
}

// The search call builds the uniform grid over the QUERIES that the Unique back-check walks (knn_points.hip) as a
// by-product: seed_query_kernel leaves per-workgroup boxes of the queries, ONE surplus workgroup of the candidate
// kernel's grid folds them into the grid geometry (nothing waits for it: knn_finalize_kernel, the next launch, fills
// the cells), so the match stage needs no box / fill launches of its own.
__device__ void ug_reduce_boxes(const float* __restrict__ part /*[n][6]*/, int n, int Q, int cells_cap, UgPrep* __restrict__ prep) {
    __shared__ float s_box[kBlock / 64][6];
    float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int b = threadIdx.x; b < n; b += kBlock) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { v[c] = fminf(v[c], part[(size_t)b * 6 + c]); v[3 + c] = fmaxf(v[3 + c], part[(size_t)b * 6 + 3 + c]); }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { v[c] = fminf(v[c], __shfl_xor(v[c], o)); v[3 + c] = fmaxf(v[3 + c], __shfl_xor(v[3 + c], o)); }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) s_box[threadIdx.x >> 6][c] = v[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lo[3], hi[3];
        for (int c = 0; c < 3; ++c) {
            lo[c] = s_box[0][c]; hi[c] = s_box[0][3 + c];
            for (int w = 1; w < kBlock / 64; ++w) { lo[c] = fminf(lo[c], s_box[w][c]); hi[c] = fmaxf(hi[c], s_box[w][3 + c]); }
        }
        ug_make_prep(lo, hi, Q, cells_cap, prep);
    }
}

// ---- the candidate kernel: software-pipelined ------------------------------------------------------
// Round 1's loop (git history: knn_candidates_f16_kernel) issued MFMA -> (wait for it) -> min tree -> compare -> branch, so inside one wave the matrix
// pipe and the VALU strictly alternate (the ISA shows `v_mfma ... s_nop 9 ... v_min3 x8 ... v_cmp ... s_cbranch`)
// and all overlap was left to the co-resident waves; and its list update walked the 16 scores of a hit with one
// compare + branch each, ~1000 cycles during which the wave's three siblings end up waiting at the tile barrier
// (rocprofv3: 0.19 of 1.67 ms; without the barrier 1.48).  This form changes both:
//  * two accumulator tiles: the product of step j + 1 is issued first and the selection of step j (min tree +
//    compare) runs under it, so the MFMA's passes are covered by the wave's own VALU work and no `s_nop` sits on
//    the critical path (scripts/ubench/mfma_f16_valu.hip, modes 1 vs 2).  A step = (sub-tile of 32 model points,
//    query group g); steps run sub-major, one A operand (one ds_read_b128) serves QG consecutive steps;
//  * GROUP entries: a list entry is (min of the lane's 16 scores, first row of those 16 points) -- the lane's
//    tree result, which the hot loop already has.  The hot loop records a ballot per step and keeps the eight
//    minima of two sub-tiles; ONE uniform branch per two sub-tiles enters the update, which is a single sorted
//    insert per hit: no recomputation, no walk over the 16 scores.  knn_finalize_kernel expands an entry that can
//    still matter into its 16 points (exact distances, (distance, index) order), so the results are the same bits;
//    the certificate is unchanged: every point outside the listed groups sits in a group whose minimum was >= the
//    threshold it was compared with, hence >= the final word G.
// Rule for the inline-asm A reads (the compiler believes an asm output is ready at once): a ds_read's destination
// is never loop-carried in flight; it is waited for inside the iteration that issued it and only waited values
// cross the back edge (tests/test_isa_lint.py checks the emitted ISA for a VGPR read between load and wait).
template <int QG, bool DRY>   // DRY: timing only (no compare, no lists; PCREG_KNN_VARIANT=41)
__global__ __launch_bounds__(kBlock, kT16 > 768 ? 2 : (kT16 > 512 ? 3 : (QG <= 2 ? 5 : (QG <= 4 ? 4 : 2)))) void knn_candidates_f16_pipe_kernel(
    const float* __restrict__ q, int Q, int ldq, const uint4* __restrict__ mt, int n_tiles, int tiles_per_chunk,
    const Prep* __restrict__ prep, unsigned* __restrict__ gthr, uint2* __restrict__ cand_ent, int32_t* __restrict__ cand_cnt,
    int cap, int q_blocks, int xcd_map, int n_chunks, const float* __restrict__ ug_part, int ug_nparts, int ug_cells, UgPrep* __restrict__ ug_prep) {
    static_assert(QG % 2 == 0, "two accumulator tiles alternate: an even number of steps per sub-tile");
    if (blockIdx.x == gridDim.x - 1 && ug_prep != nullptr) {     // the surplus workgroup (the launcher adds it): query-grid geometry
        ug_reduce_boxes(ug_part, ug_nparts, Q, ug_cells, ug_prep);
        return;
    }
    __shared__ __attribute__((aligned(16))) uint4 tile[2][2 * kT16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    int qb, chunk;
    if (xcd_map) { const int j = (int)blockIdx.x >> 3; chunk = (j / q_blocks) * 8 + ((int)blockIdx.x & 7); qb = j % q_blocks; }
    else { qb = (int)blockIdx.x % q_blocks; chunk = (int)blockIdx.x / q_blocks; }
    if (chunk >= n_chunks) return;            // the grid rounds the chunk count up to a multiple of 8 for the XCD map
    const int q_base = (qb * (kBlock / 64) + wave) * (QG * 32);
    const float sg = prep->sigma, inv2 = prep->inv_sigma2, sg2 = sg * sg;

    f16x8 bq[QG];
    float thr[QG];
    unsigned gseen[QG];
    Cand cand[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int qi = q_base + g * 32 + col;
        float X = 0.0f, Y = 0.0f, Z = 0.0f, one = 0.0f;
        if (qi < Q) {
            const float sx = sg * (q[qi] - prep->cx), sy = sg * (q[qi + (size_t)ldq] - prep->cy), sz = sg * (q[qi + 2 * (size_t)ldq] - prep->cz);
            // a query far outside the prepared model's box (or not finite) is not scored here: all-zero operands, no list
            // entries; knn_finalize_kernel applies the same test and sends it to the exact fallback
            if (fabsf(sx) <= kQueryScaledMax && fabsf(sy) <= kQueryScaledMax && fabsf(sz) <= kQueryScaledMax) {
                X = -2.0f * sx; Y = -2.0f * sy; Z = -2.0f * sz; one = 1.0f;
            }
        }
        _Float16 Xh, Xl, Yh, Yl, Zh, Zl;
        split2(X, Xh, Xl); split2(Y, Yh, Yl); split2(Z, Zh, Zl);
        const _Float16 o1 = (_Float16)one;
        bq[g] = half == 0 ? f16x8{Xh, Xh, Xl, Xl, Yh, Yh, Yl, Yl} : f16x8{Zh, Zh, Zl, Zl, o1, o1, o1, (_Float16)0.0f};
#pragma unroll
        for (int k = 0; k < KC; ++k) { cand[g].s[k] = INFINITY; cand[g].i[k] = -1; }
        thr[g] = INFINITY; gseen[g] = 0xFFFFFFFFu;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&tile[0][0];
    const int t_begin = chunk * tiles_per_chunk, t_end = min(n_tiles, t_begin + tiles_per_chunk);
    const int ntile = t_end - t_begin;
#define PCREG_TILE_DMA(T, BUF)                                                                                     \
    _Pragma("unroll") for (int k = 0; k < kT16 / 128; ++k) {                                                        \
        const int seg = k * 4 + wave;                                                                              \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mt + (size_t)(T) * (2 * kT16) + seg * 64 + lane), \
                                         (__attribute__((address_space(3))) void*)(&tile[BUF][seg * 64]), 16, 0, 0);   \
    }
    if (ntile > 0) { PCREG_TILE_DMA(t_begin, 0) }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int kSubs = kT16 / 32;
    for (int t = 0; t < ntile; ++t) {
        if (t + 1 < ntile) { PCREG_TILE_DMA(t_begin + t + 1, (t + 1) & 1) }
        if ((t & (kRefresh - 1)) == 0) {          // chunks share one monotone threshold word per query (unscaled units)
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                const int qi = q_base + g * 32 + col;
                if (qi < Q) {
                    if (cand[g].s[KC - 1] < INFINITY) { unsigned k = f2ord(cand[g].s[KC - 1] * inv2); if (k < gseen[g]) atomicMin(&gthr[qi], k); }
                    unsigned gv = __hip_atomic_load(&gthr[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gseen[g] = gv;
                    thr[g] = fminf(fminf(thr[g], ord2f(gv) * sg2), __shfl_xor(thr[g], 32));       // and the sibling half's list
                }
            }
        }
        const unsigned cur = lds_base + (unsigned)((t & 1) * (2 * kT16) + half * kT16 + col) * 16u;
        const int jt = (t_begin + t) * kT16 + 4 * half;
        f16x8 av;
        {
            u32x4 a0;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a0) : "v"(cur) : "memory");
            av = __builtin_bit_cast(f16x8, a0);
        }
        f32x16 dprev = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bq[0], zero, 0, 0, 0);
        // the first tree of the loop reads dprev: the hazard recogniser gets its wait states HERE, once per tile,
        // or it pads the loop head in every iteration (there is no s_nop builtin: twelve one-wait-state scalar
        // instructions)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int sub = 0; sub < kSubs; sub += 2) {
            // ONE threshold test per two sub-tiles: the first step only keeps its minimum (mn0), the second folds it into
            // its tree as the 17th value (8 v_min3, the 16 scores alone take 7 v_min3 + 1 v_min) and tests min(mn0, mn1);
            // the rare path gets mn1 back by issuing that one product again (same operands, same bits)
            unsigned long long hit[QG];
            float mn0[QG];
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                // the next sub-tile's A operand: issued now, waited for in this half's last step (after sub 15 it
                // fetches sub 15 again: harmless, keeps the body branch-free)
                u32x4 an;
                asm volatile("ds_read_b128 %0, %1" : "=v"(an) : "v"(cur + (unsigned)min(sub + ss + 1, kSubs - 1) * 512u) : "memory");
                f16x8 avn = av;
#pragma unroll
                for (int g = 0; g < QG; ++g) {
                    f32x16 dnext;
                    if (g + 1 < QG) {
                        dnext = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bq[g + 1], zero, 0, 0, 0);
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(an) :: "memory");
                        avn = __builtin_bit_cast(f16x8, an);
                        dnext = __builtin_amdgcn_mfma_f32_32x32x16_f16(avn, bq[0], zero, 0, 0, 0);    // after sub 15: a product nobody reads
                    }
                    const f32x16 d = dprev;
                    const float m0 = fminf(fminf(d[0], d[1]), d[2]), m1 = fminf(fminf(d[3], d[4]), d[5]);
                    const float m2 = fminf(fminf(d[6], d[7]), d[8]), m3 = fminf(fminf(d[9], d[10]), d[11]);
                    const float m4 = fminf(fminf(d[12], d[13]), d[14]);
                    if (ss == 0) {
                        mn0[g] = fminf(fminf(fminf(m0, m1), m2), fminf(fminf(m3, m4), d[15]));
                    } else {
                        const float mn = fminf(fminf(fminf(fminf(m0, m1), m2), fminf(fminf(m3, m4), d[15])), mn0[g]);
                        if (DRY) { asm volatile("" :: "v"(mn)); hit[g] = 0; }
                        else hit[g] = __builtin_amdgcn_ballot_w64(mn < thr[g]);
                    }
                    dprev = dnext;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // the MFMA first, the selection under it
                    __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                av = avn;
            }
            unsigned long long any = 0;
#pragma unroll
            for (int g = 0; g < QG; ++g) any |= hit[g];
            if (!DRY && any != 0) {        // wave-uniform, rare
                u32x4 a1;                  // the second sub-tile's A operand again (av already holds the next pair's)
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a1) : "v"(cur + (unsigned)(sub + 1) * 512u) : "memory");
                const f16x8 av1 = __builtin_bit_cast(f16x8, a1);
#pragma unroll
                for (int g = 0; g < QG; ++g) {
                    if (hit[g] != 0) {               // scalar test; the insertions themselves are straight-line code
                        const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(av1, bq[g], zero, 0, 0, 0);
                        const float m0 = fminf(fminf(d[0], d[1]), d[2]), m1 = fminf(fminf(d[3], d[4]), d[5]);
                        const float m2 = fminf(fminf(d[6], d[7]), d[8]), m3 = fminf(fminf(d[9], d[10]), d[11]);
                        const float m4 = fminf(fminf(d[12], d[13]), d[14]);
                        const float mn1 = fminf(fminf(fminf(m0, m1), m2), fminf(fminf(m3, m4), d[15]));
                        cand_insert_branchless(cand[g], mn0[g], jt + sub * 32, mn0[g] < thr[g]);
                        thr[g] = fminf(thr[g], cand[g].s[KC - 1]);
                        cand_insert_branchless(cand[g], mn1, jt + (sub + 1) * 32, mn1 < thr[g]);
                        thr[g] = fminf(thr[g], cand[g].s[KC - 1]);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef PCREG_TILE_DMA
    // The two half-waves of a query (lanes l and l ^ 32) merge their sorted fours in registers and lane l < 32
    // writes ONE list of at most four GROUP entries per (chunk, query); the merged 4th-best is published too, so
    // every group dropped here still has a minimum >= the final threshold word G.
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int qi = q_base + g * 32 + col;
        Cand mine = cand[g];
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const float os = __shfl_xor(cand[g].s[k], 32);
            const int oi = __shfl_xor(cand[g].i[k], 32);
            if (oi >= 0 && (os < mine.s[KC - 1] || (os == mine.s[KC - 1] && (unsigned)oi < (unsigned)mine.i[KC - 1]))) {
                int pos = KC - 1;
#pragma unroll
                for (int t = KC - 2; t >= 0; --t) if (os < mine.s[t] || (os == mine.s[t] && (unsigned)oi < (unsigned)mine.i[t])) pos = t;
#pragma unroll
                for (int t = KC - 1; t > 0; --t) if (t > pos) { mine.s[t] = mine.s[t - 1]; mine.i[t] = mine.i[t - 1]; }
#pragma unroll
                for (int t = 0; t < KC; ++t) if (t == pos) { mine.s[t] = os; mine.i[t] = oi; }
            }
        }
        if (qi < Q && half == 0) {
            if (mine.s[KC - 1] < INFINITY) { unsigned k = f2ord(mine.s[KC - 1] * inv2); if (k < gseen[g]) atomicMin(&gthr[qi], k); }
            int nv = 0;
#pragma unroll
            for (int k = 0; k < KC; ++k) nv += mine.i[k] >= 0;
            if (nv > 0) {
                const int base = atomicAdd(&cand_cnt[qi], nv);
                uint2* dst = cand_ent + (size_t)qi * cap + base;
#pragma unroll
                for (int k = 0; k < KC; ++k) if (k < nv) dst[k] = make_uint2((unsigned)mine.i[k], __float_as_uint(mine.s[k] * inv2));
            }
        }
    }
}

}  // namespace

// ---- live timing of the dominant kernel (bench.py's roofline line) --------------------------------
// When enabled, the candidates kernel of every MAIN search (any size: a 125 k-row shard of an 8-GPU run as much as the
// 1 M-row model) is bracketed by two HIP events on the launch stream; pcreg_dev_search_kernel_ms() returns the mean over
// the launches since the last call.  A caller that searches for its own purposes passes timed = false.
static bool g_time_on = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_time_ev;
static size_t g_time_used = 0;
void knn_f16_timing_enable(bool on) { g_time_on = on; g_time_used = 0; }
int knn_f16_timing_read(float* mean_ms, int* launches) {
    double tot = 0.0; int n = 0;
    for (size_t k = 0; k < g_time_used; ++k) {
        float ms = 0.0f;
        if (hipEventSynchronize(g_time_ev[k].second) != hipSuccess) continue;
        if (hipEventElapsedTime(&ms, g_time_ev[k].first, g_time_ev[k].second) == hipSuccess) { tot += ms; ++n; }
    }
    g_time_used = 0;
    *mean_ms = n ? (float)(tot / n) : 0.0f; *launches = n;
    return PCREG_OK;
}

size_t knn_f16_prep_bytes(int M) { return (size_t)((M > 0 ? M : 1) + kT16 - 1) / kT16 * (2 * kT16) * sizeof(uint4); }

// the launch shape of the candidate kernel for Q queries against M model rows: q_blocks x S workgroups (+ surplus);
// a (chunk, query) pair lists at most KC group entries, so a query's list holds at most S * KC
void knn_f16_shape(int Q, int M, int target_blocks, int* q_blocks, int* S, int* tiles_per_chunk) {
    constexpr int QG = 4;
    const int n_tiles = (M + kT16 - 1) / kT16;
    const int qb = (Q + (kBlock / 64) * QG * 32 - 1) / ((kBlock / 64) * QG * 32);
    int s = target_blocks / (qb > 0 ? qb : 1); if (s < 1) s = 1;
    if (s > kF16MaxS) s = kF16MaxS;
    if (s > n_tiles) s = n_tiles > 0 ? n_tiles : 1;
    const int tpc = n_tiles > 0 ? (n_tiles + s - 1) / s : 1;
    s = n_tiles > 0 ? (n_tiles + tpc - 1) / tpc : 1;
    *q_blocks = qb; *S = s; *tiles_per_chunk = tpc;
}

// model -> f16 tiles (+ the seeding grid when seed_cnt != nullptr); once per prepared model
int launch_prep_model_f16(const float* m, int M, int ldm, const void* prep, unsigned* rm2, void* mtiles, int32_t* seed_cnt,
                          void* seed_slots, hipStream_t st) {
    if (M <= 0) return PCREG_OK;
    const int n_tiles = (M + kT16 - 1) / kT16;
    int pb = (n_tiles * kT16 + kBlock * 2 - 1) / (kBlock * 2); if (pb > 2048) pb = 2048;     // the fill's atomics want parallelism
    hipLaunchKernelGGL(prep_model_f16_kernel, dim3(pb), dim3(kBlock), 0, st, m, M, ldm, (const Prep*)prep, (uint4*)mtiles, n_tiles, rm2,
                       seed_cnt, (float4*)seed_slots);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

// The candidate stage against a prepared model.  cand_cnt [Q] must be zero (seed_query_kernel clears it).
// ug_*: the query-grid by-product (null: none).  Returns S (chunks) through S_out; list capacity per query = S * KC.
int launch_knn_candidates_f16(const float* q, int Q, int ldq, int M, const void* prep, const void* mtiles, unsigned* gthr,
                              void* cand_ent, int32_t* cand_cnt, int target_blocks, bool dry, bool timed, const float* ug_part,
                              int ug_nparts, int ug_cells, void* ug_prep, int* S_out, hipStream_t st) {
    int q_blocks, S, tiles_per_chunk;
    knn_f16_shape(Q, M, target_blocks, &q_blocks, &S, &tiles_per_chunk);
    *S_out = S;
    if (M <= 0 || Q <= 0) return PCREG_OK;
    const int n_tiles = (M + kT16 - 1) / kT16;
    // XCD-aware placement deals chunk c to the XCD that runs workgroups b = c (mod 8): the grid rounds the chunk count
    // up to a multiple of 8 and the surplus workgroups leave at once
    const int xcd_map = S >= 8 ? 1 : 0;
    const int grid_chunks = xcd_map ? (S + 7) / 8 * 8 : S;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (g_time_on && timed) {
        if (g_time_used == g_time_ev.size()) { hipEvent_t a, b; PCREG_HIP(hipEventCreate(&a)); PCREG_HIP(hipEventCreate(&b)); g_time_ev.emplace_back(a, b); }
        ev0 = g_time_ev[g_time_used].first; ev1 = g_time_ev[g_time_used].second; ++g_time_used;
        PCREG_HIP(hipEventRecord(ev0, st));
    }
    const int aux = ug_prep != nullptr ? 1 : 0;
#define PCREG_F16_LAUNCH(QGV, DRYV) hipLaunchKernelGGL((knn_candidates_f16_pipe_kernel<QGV, DRYV>), dim3(q_blocks * grid_chunks + aux), dim3(kBlock), 0, st, q, Q, ldq, \
                           (const uint4*)mtiles, n_tiles, tiles_per_chunk, (const Prep*)prep, gthr, (uint2*)cand_ent, cand_cnt, S * KC, q_blocks, xcd_map, S, \
                           ug_part, ug_nparts, ug_cells, (UgPrep*)ug_prep)
#ifdef PCREG_EXPERIMENTS
    if (dry) PCREG_F16_LAUNCH(4, true); else
#endif
    PCREG_F16_LAUNCH(4, false);
#undef PCREG_F16_LAUNCH
    (void)dry;
    if (ev1) PCREG_HIP(hipEventRecord(ev1, st));
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

}  // namespace pcreg
