// pcreg_amd/csrc/knn_mfma.hip -- the candidate kernel of knn_fast.hip on the matrix cores.
// Kept as a selectable variant (PCREG_KNN_VARIANT=5) and as evidence: on gfx950 the fp32 MFMA
// runs on the SIMD's fp32 datapath, so it does not overlap with the selection VALU work and
// loses to the 3-FMA VALU form.  Built with -fno-honor-nans and
// -amdgpu-mfma-vgpr-form (no canonicalising v_max, results straight into VGPRs).
#include "common.hpp"
#include "knn_fast_common.hpp"

namespace pcreg {
namespace {

// ---- 2b. candidate generation on the matrix cores ----------------------------------------
// s(q,m) = [m~x m~y m~z |m~|^2] . [-2q~x -2q~y -2q~z 1]^T is a K = 4 product, exactly the shape of
// v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate, an exact k-ordered fma chain; it runs at
// the fp32 VECTOR rate, so this is not a precision trade but a second issue port): one
// instruction scores 16 model points x 16 queries, 256 pairs per 32 SIMD cycles, and leaves
// the VALU free for the selection (v_min3 + v_min + one compare per 4 scores).
//   A (model)  : lane l holds component (l>>4) of model point (l&15) of the tile -> the model
//                is pre-laid out as [tile][component][16 points], one coalesced dword per lane
//   B (queries): lane l holds component (l>>4) of query (l&15): -2q~ and the constant 1
//   D          : lane l gets rows 4*(l>>4)+r (model points), column l&15 (its query)
// Each lane therefore sees one query per query tile and a quarter of the model points; it keeps
// a sorted top-4 for that (query, quarter).  Quarters and chunks share thresholds: lane groups
// through v_permlane/ds_bpermute every 8 model tiles, chunks through the threshold word in HBM.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kBlock) void prep_model_tiles_kernel(const float* __restrict__ m, int M, int ldm,
                                                                  const Prep* __restrict__ prep, float* __restrict__ out,
                                                                  int n_tiles, unsigned* __restrict__ rm2_bits) {
    const float cx = prep->cx, cy = prep->cy, cz = prep->cz;
    float mx = 0.0f;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n_tiles * 16; i += gridDim.x * kBlock) {
        float x = 0.0f, y = 0.0f, z = 0.0f, w = INFINITY;           // padding: s = +inf, never a candidate
        if (i < M) {
            x = m[i] - cx; y = m[i + (size_t)ldm] - cy; z = m[i + 2 * (size_t)ldm] - cz;
            w = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
            mx = fmaxf(mx, w);
        }
        float* t = out + (size_t)(i >> 4) * 64 + (i & 15);
        t[0] = x; t[16] = y; t[32] = z; t[48] = w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) atomicMax(rm2_bits, __float_as_uint(mx));
}

template <int NQ, bool DRY = false>
__global__ __launch_bounds__(kBlock) void knn_candidates_mfma_kernel(
    const float* __restrict__ q, int Q, int ldq, const float* __restrict__ mt, int n_tiles, int tiles_per_chunk,
    const Prep* __restrict__ prep, unsigned* __restrict__ gthr, int32_t* __restrict__ part_idx /*[S][Q][16]*/,
    float* __restrict__ part_s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int q_base = (blockIdx.x * (kBlock / 64) + wave) * (NQ * 16);
    if (q_base >= Q) return;
    const float cg = g == 0 ? prep->cx : (g == 1 ? prep->cy : prep->cz);

    float bq[NQ], thr[NQ];
    unsigned gseen[NQ];
    Cand cand[NQ];
#pragma unroll
    for (int T = 0; T < NQ; ++T) {
        const int qi = q_base + T * 16 + j;
        float v = 1.0f;
        if (g < 3) v = qi < Q ? -2.0f * (q[qi + (size_t)g * ldq] - cg) : 0.0f;
        bq[T] = v;
#pragma unroll
        for (int k = 0; k < KC; ++k) { cand[T].s[k] = INFINITY; cand[T].i[k] = -1; }
        thr[T] = INFINITY; gseen[T] = 0xFFFFFFFFu;
    }
    const int t_begin = blockIdx.y * tiles_per_chunk, t_end = min(n_tiles, t_begin + tiles_per_chunk);
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    // Software pipeline: the scores of model tile t-1 are consumed while the matrix core works on
    // tile t.  Every query tile T owns an accumulator d[T]; "consume d[T], then reissue d[T]" keeps
    // eight MFMAs in flight and puts the three selection VALU ops between consecutive MFMA issues.
    f32x4 d[NQ];
    {
        const float a0 = mt[(size_t)t_begin * 64 + lane];
#pragma unroll
        for (int T = 0; T < NQ; ++T) d[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bq[T], zero, 0, 0, 0);
    }
    for (int t = t_begin + 1; t <= t_end; ++t) {
        const bool more = t < t_end;
        const float a = more ? mt[(size_t)t * 64 + lane] : 0.0f;
        const int rel = t - 1 - t_begin;                 // tile whose scores are consumed now
        if ((rel & 7) == 0) {
            if ((rel & 63) == 0) {        // chunks: publish / pick up the threshold word in HBM
#pragma unroll
                for (int T = 0; T < NQ; ++T) {
                    const int qi = q_base + T * 16 + j;
                    if (qi < Q) {
                        if (cand[T].s[3] < INFINITY) { unsigned k = f2ord(cand[T].s[3]); if (k < gseen[T]) atomicMin(&gthr[qi], k); }
                        unsigned gv = __hip_atomic_load(&gthr[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        gseen[T] = gv;
                        thr[T] = fminf(thr[T], ord2f(gv));
                    }
                }
            }
            // lane groups of the same query: each group's 4th-best bounds the query's 4th-best
#pragma unroll
            for (int T = 0; T < NQ; ++T) {
                float x = fminf(thr[T], __shfl_xor(thr[T], 16));
                thr[T] = fminf(x, __shfl_xor(x, 32));
            }
        }
        const int jbase = (t - 1) * 16 + 4 * g;
#pragma unroll
        for (int T = 0; T < NQ; ++T) {
            const f32x4 v = d[T];
            float mn = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
            if (DRY) { asm volatile("" :: "v"(mn)); }
            else if (mn < thr[T]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (v[r] < thr[T]) cand_insert(cand[T], v[r], jbase + r);
                thr[T] = fminf(thr[T], cand[T].s[3]);
            }
            if (more) d[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq[T], zero, 0, 0, 0);
        }
    }
    const int chunk = blockIdx.y;
#pragma unroll
    for (int T = 0; T < NQ; ++T) {
        const int qi = q_base + T * 16 + j;
        if (qi < Q) {
            if (cand[T].s[3] < INFINITY) { unsigned k = f2ord(cand[T].s[3]); if (k < gseen[T]) atomicMin(&gthr[qi], k); }
            size_t o = ((size_t)chunk * Q + qi) * 16 + g * 4;
            *reinterpret_cast<int4*>(part_idx + o) = make_int4(cand[T].i[0], cand[T].i[1], cand[T].i[2], cand[T].i[3]);
            *reinterpret_cast<float4*>(part_s + o) = make_float4(cand[T].s[0], cand[T].s[1], cand[T].s[2], cand[T].s[3]);
        }
    }
}


}  // namespace

int launch_knn_candidates_mfma(const float* q, int Q, int ldq, const float* m, int M, int ldm, const void* prep,
                               unsigned* rm2, float* mtiles, int n_tiles, int tiles_per_chunk, int q_blocks, int S,
                               unsigned* gthr, int32_t* part_idx, float* part_s, bool dry, hipStream_t st) {
    constexpr int NQ = 8;
    int pb = (n_tiles * 16 + kBlock * 4 - 1) / (kBlock * 4); if (pb > 2048) pb = 2048;
    hipLaunchKernelGGL(prep_model_tiles_kernel, dim3(pb), dim3(kBlock), 0, st, m, M, ldm, (const Prep*)prep, mtiles, n_tiles, rm2);
    if (dry)
        hipLaunchKernelGGL((knn_candidates_mfma_kernel<NQ, true>), dim3(q_blocks, S), dim3(kBlock), 0, st, q, Q, ldq,
                           (const float*)mtiles, n_tiles, tiles_per_chunk, (const Prep*)prep, gthr, part_idx, part_s);
    else
        hipLaunchKernelGGL((knn_candidates_mfma_kernel<NQ>), dim3(q_blocks, S), dim3(kBlock), 0, st, q, Q, ldq,
                           (const float*)mtiles, n_tiles, tiles_per_chunk, (const Prep*)prep, gthr, part_idx, part_s);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

}  // namespace pcreg
