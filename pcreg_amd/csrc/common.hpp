// pcreg_amd/csrc/common.hpp -- shared host-side helpers of libpcreg_hip (gfx950 only).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <string>
#include "../../include/pcreg.h"

namespace pcreg {

void set_error(const char* fmt, ...);
int  ensure_device();   // PCREG_OK or PCREG_E_NODEVICE / PCREG_E_HIP

#define PCREG_HIP(call)                                                                  \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) {                                                         \
            ::pcreg::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),   \
                               __FILE__, __LINE__);                                      \
            return PCREG_E_HIP;                                                          \
        }                                                                                \
    } while (0)

#define PCREG_ARG(cond)                                                                  \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            ::pcreg::set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__);   \
            return PCREG_E_ARG;                                                          \
        }                                                                                \
    } while (0)

// Grow-only device scratch owned by the host tier (one per process; MEX calls arrive
// on MATLAB's interpreter thread).  Slots keep independent
// buffers alive across one call.
struct Scratch {
    static constexpr int kSlots = 24;
    void*  ptr[kSlots]  = {};
    size_t size[kSlots] = {};
    int get(int slot, size_t bytes, void** out);
    void release_all();
};
Scratch& scratch();
// device-tier temporaries: one Scratch per stream, so that calls on different streams may overlap
// (a stream runs its own kernels in order, which makes the reuse within it safe)
Scratch& stream_scratch(hipStream_t st);

// Switches.  Two kinds, and NEITHER reads the environment in the default build (a stray variable in a MATLAB worker's
// environment must not change which kernels run):
//  * result-preserving A/B switches (the direct-form search, the exhaustive fp64 SAD, the forced fallback of the certified
//    matcher, the fused / fp64-only RANSAC kernels, ...): process-wide integers set through pcreg_debug_set(key, value) --
//    the entry the parity tests call to run both sides of a certified path inside one process; every setting gives the same
//    indices and counts;
//  * experiment / debug switches that change the launch shape, print, synchronise, write files or INVALIDATE results
//    (timing-only kernels): environment variables that exist only in a build with -DPCREG_EXPERIMENTS (`make EXPERIMENTS=1`);
//    the default library compiles them to their default value.
enum DebugKey {
    kDbgKnnExact = 0,          // "knn_exact": direct-form fp32 search instead of the certified f16 matrix-core path
    kDbgMatchExact,            // "match_exact": exhaustive fp64 SAD instead of the certified u16 path
    kDbgMatchForceFallback,    // "match_force_fallback": 1 = every query takes the exact fallback, 2 = also skip the refine
    kDbgRansacFused,           // "ransac_fused": n >= 4096 on the fused tiled kernel instead of the staged chain
    kDbgRansacNoLane,          // "ransac_nolane": refit sums by the fp64 sweep kernel instead of the int8 matrix-core kernel
    kDbgRansacF64Score,        // "ransac_f64score": no fp32 screen in the staged chain
    kDbgRansacResidentF64,     // "ransac_resident_f64": the round-3 fp64 LDS-resident kernel instead of ransac_hyp32_kernel
    kDbgAlignTimes,            // "align_times": per-phase timestamps of align_points_knn (host read-back)
    kDbgAlignShape,            // "align_shape": launch-shape override of align_points_knn
    kDbgSegDebug,              // "seg_debug": histogram dump of the segmented matcher
    kDbgSegBatched,            // "seg_batched": the segmented matcher runs its bounded-workspace (batched) form whatever the size
    kDbgSegWaveFinalize,       // "seg_wave_finalize": the segmented matcher's forward re-rank as one wave per query (rounds 2-3) instead of pick / pairs / decide
    kDbgMatchStats,            // "match_stats": the certified matcher counts what it proves / re-scores / hands on (pcreg_debug_match_stats)
    kDbgCount
};
int debug_flag(DebugKey k);
// Device counters of the certified SAD matcher (match_sad16.hip), or null while "match_stats" is off:
//   [0] queries finalised   [1] candidates re-scored exactly (fp64)   [2] queries the certificate left unproven
//   [3] queries handed to the exhaustive exact-rows kernel   [4] Unique back-check items (segmented form)
//   [5] back-check items handed to the exhaustive kernel   [6] matcher calls   [7] segments
unsigned long long* match_stats_dev();
#ifdef PCREG_EXPERIMENTS
static inline int pcreg_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline const char* pcreg_env_str(const char* name) { return getenv(name); }
#define PCREG_EXP_ENV(name, dflt) pcreg_env_int(name, dflt)
#define PCREG_EXP_STR(name) pcreg_env_str(name)
#else
#define PCREG_EXP_ENV(name, dflt) (dflt)
#define PCREG_EXP_STR(name) ((const char*)nullptr)
#endif

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- kernel launchers implemented in the .hip files (device pointers, async) --------
struct RansacDims { int n_cap; int iters; int B; };

size_t ransac_workspace_bytes(int iters, int B, int n_cap);
int launch_ransac(const double* p1, const double* p2, int ld, const int32_t* offsets /*B+1 dev or null*/,
                  const int32_t* n_dev /*single registration: device n, or null*/, int n_cap, int B,
                  const pcreg_ransac_opts& o, const int32_t* sample_idx_dev,
                  pcreg_dev_ransac_result* out /*B*/, int32_t* inlier_idx, int32_t* iter_inl /*B*iters or null*/,
                  int32_t* iter_inl_ref, void* ws, size_t ws_bytes, hipStream_t st);
int launch_ransac_partial(const double* p1, const double* p2, int ld, const int32_t* n_dev, int n_cap,
                          const pcreg_ransac_opts& o, const int32_t* sample_idx_dev, int hyp_begin, int hyp_count,
                          pcreg_dev_ransac_part* part, void* ws, size_t ws_bytes, hipStream_t st);
int launch_ransac_finish(const double* p1, const double* p2, int ld, const int32_t* n_dev, int n_cap,
                         const pcreg_ransac_opts& o, const pcreg_dev_ransac_part* combined,
                         pcreg_dev_ransac_result* out, int32_t* inlier_idx, hipStream_t st, int n_parts = 1);
int launch_estimate_transform(const double* p1, const double* p2, int n, int ld, double* T16_dev,
                              int32_t* empty_dev, hipStream_t st);
int launch_calc_dists(const double* T16_dev, const double* p1, const double* p2, int n, int ld,
                      double* d, hipStream_t st);

// a prepared model (knn_fast.hip): pointers into one device block of model_prep_bytes(M) bytes + the caller's points
struct ModelView {
    const float* m; int M, ldm;
    void* prep; unsigned* rm2; float* box_part; void* tiles; int32_t* seed_cnt; void* seed_slots; int seeded;
};
size_t model_prep_bytes(int M);
ModelView model_view(const float* m, int M, int ldm, void* block);
int launch_model_prepare(const ModelView& v, hipStream_t st);
// the per-call workspace of a search and of the match stage that follows it
struct SearchWs {
    void* ctr; unsigned* gthr; int32_t* flag_list; int32_t* cand_cnt; void* cand_ent; int cap;
    int32_t* tail_idx; float* tail_dist;
    void* ug_prep; float* ug_part; int32_t* ug_cnt; void* ug_slots; int ug_cells, ug_nparts;
};
SearchWs search_ws_layout(int Q, int M, void* base, size_t* bytes);
size_t search_ws_bytes(int Q, int M);
int launch_model_search(const ModelView& v, const float* q, int Q, int ldq, int32_t idx_base, int32_t* idx, float* dist,
                        void* ws, size_t ws_bytes, bool with_grid, bool timed, hipStream_t st);
// the match stage on a finished search (knn_points.hip): threshold + ratio + Unique (query grid of the search's workspace)
// + ordered compaction in ONE launch; the two halves around the multi-GPU table exchange
int launch_match_finish(const ModelView& v, const float* q, int Q, int ldq, const int32_t* idx, const float* dist, float thr,
                        float ratio, int unique, void* ws, size_t ws_bytes, uint32_t* pairs, double* pts1, double* pts2,
                        int32_t* n_pairs, hipStream_t st);
int launch_match_table(const ModelView& v, int32_t m_lo, int M_total, const float* q, int Q, int ldq, const int32_t* idx,
                       const float* dist, float thr, float ratio, int unique, void* ws, size_t ws_bytes, int32_t* table,
                       hipStream_t st);
int launch_match_from_table(const float* q, int Q, int ldq, int M_total, const int32_t* idx, const float* dist, float thr,
                            float ratio, const int32_t* table, void* ws, size_t ws_bytes, uint32_t* pairs, double* pts1,
                            double* pts2, int32_t* n_pairs, hipStream_t st);

size_t knn2_points_workspace_bytes(int Q, int M);
int launch_knn2_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm,
                           int32_t idx_base, int32_t* idx, float* dist, void* ws, size_t ws_bytes,
                           hipStream_t st, bool timed = true);
int launch_merge_top2_f32(const int32_t* idx_in, const float* dist_in, int R, int Q, int32_t* idx,
                          float* dist, hipStream_t st, size_t rank_stride = 0);
// generic-D descriptor matching (fp64)
size_t match_features_workspace_bytes(int Q, int M, int D);
int launch_preprocess(const double* dS, int Q, int ldS, const double* dM, int M, int ldM, int D,
                      const pcreg_match_opts& o, double* outS, double* outM, void* ws, size_t ws_bytes,
                      hipStream_t st);
int launch_normalize_rows(double* f, int n, int ld, int D, hipStream_t st);
int launch_normalize_rows2(double* f, int n, int ld, double* f2, int n2, int ld2, int D, hipStream_t st);
int launch_match_features(const double* fS, int Q, int ldS, const double* fM, int M, int ldM, int D,
                          const pcreg_match_opts& o, uint32_t* pairs, double* metric, int32_t* P_dev,
                          void* ws, size_t ws_bytes, hipStream_t st);

// sphere-sweep driver pieces (sweep.hip) and the final refine (ransac.hip)
int launch_sphere_counts(const double* feat, int V, const double* centres, int S, double R, int32_t* counts, hipStream_t st);
size_t sphere_select_workspace_bytes(int V);
int launch_sphere_select(const double* feat, int V, const double c[3], double R, int32_t* idx, int32_t* n_out, void* ws, size_t ws_bytes,
                         hipStream_t st);
int launch_gather_rows_f64(const double* src, int D, const int32_t* idx, const int32_t* n, int cap, double* dst, hipStream_t st);
int launch_quick_tf(const double* pts, int n, int ld, const double T[16], double* out, int ldo, hipStream_t st);
int launch_refine_by_distance(const double* p1, const double* p2, const int32_t* n_dev, int cap, int ld, double maxDist,
                              double* T16_dev, int32_t* info_dev, hipStream_t st);

void knn_f16_timing_enable(bool on);
int knn_f16_timing_read(float* mean_ms, int* launches);
int launch_transpose_rows(const double* f, int n, int ld, int D, double* out, hipStream_t st);
int launch_widen_rows_u16(const uint16_t* rows, const int32_t* index, int n, int D, double* featmajor, hipStream_t st);   // rows[index[i]] (u16) -> feature-major f64
size_t local_points_workspace_bytes(int N);
int launch_local_points(const double* pts, int N, int ld, double R, const double c[3], int mode, double* out, int ldo, double* dists,
                        int32_t* totals, void* ws, size_t ws_bytes, hipStream_t st);
int launch_sphere_select_batched(const double* feat, int V, const double* centres, int S, double R, const int32_t* seg_off, int32_t* idx,
                                 double* feat_out, int32_t* n_out, hipStream_t st);
size_t get_matches_segmented_workspace_bytes(int Q, int VM, int D, int S, int tot, int n_max);
// The model side of the segmented matcher that does not depend on the surface or the segments: the powered rows and the six
// scalars per row (segp_rows_kernel).  A caller that matches MANY surfaces against one immutable model set (the host tier's
// descriptor sets) prepares them once: P [VM][D], r [6][VM]; valid for the (change_metric, metric_factor) they were made with.
struct SegPreparedModel { const double* P; const double* r; int VM, D, change_metric; double metric_factor; };
size_t segmented_prepared_model_bytes(int VM, int D);          // doubles of P and r together, in bytes
int launch_segmented_prepare_model(const double* descM_rows, int VM, int D, const pcreg_match_opts& o, double* P, double* r, hipStream_t st);
int launch_get_matches_segmented(const double* descS, int Q, const double* descM, int VM, int D, const int32_t* seg_rows,
                                 const int32_t* seg_off, int S, int tot, int n_max, const pcreg_match_opts& o, uint32_t* pairs_all,
                                 double* metric_all, int32_t* n_pairs, void* ws, size_t ws_bytes, hipStream_t st,
                                 const SegPreparedModel* prepared = nullptr);
int launch_sweep_plan(const int32_t* n_pairs, int S, int thresh, int32_t* trial_idx, int32_t* offsets, int32_t* n_trials, hipStream_t st);
int launch_sweep_gather(const uint32_t* pairs_all, int VS, const int32_t* n_pairs, const int32_t* trial_idx, const int32_t* offsets,
                        const int32_t* n_trials, int S, const double* featS, const double* featCur_all, const int64_t* row_off,
                        double* p1, double* p2, int ld, hipStream_t st);
int launch_gather_matched_rows(const uint32_t* pairs, const int32_t* n_pairs, int cap, const double* featS, const double* featM,
                               double* pts1, double* pts2, hipStream_t st);

int launch_align_points_knn(const double* pts, int ld, const int32_t* offsets_dev, int B, int max_n,
                            int C1, int C2, double* aligned, int ld_out, double* coeff, double* c,
                            int32_t* status, hipStream_t st);


// spatial-histogram descriptors (cfg 4)
size_t descriptors_workspace_bytes(int P, int S);
int launch_descriptors(const double* pts, int P, int ld, const double* kp, int S, int ldk, const pcreg_desc_opts& o, int single_mode,
                       double* feat, double* desc_f64, uint16_t* rows_u16, int32_t* row_index, int32_t* V_dev, int32_t* err_dev,
                       void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace pcreg
