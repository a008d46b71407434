// pcreg_amd/csrc/comm.hip -- the multi-GPU split of the hot path behind the C ABI (include/pcreg.h, "multi-GPU").
//
// north_star keeps the host MATLAB -> MEX -> C ABI: a MATLAB `parfor`/`spmd` pool is one process per worker
// (completeExperimentFast.m:134,201), so one worker per GPU + this file gives the model-row split of SURVEY 8e
// without any Python: worker 0 calls pcreg_comm_get_unique_id, the 128 bytes travel to the other workers by
// MATLAB's own means (labBroadcast / a file), every worker calls pcreg_comm_init(rank, world, id), and then
//     pcreg_match_points_sharded_f32   each rank passes ITS model rows [m_lo, m_lo + M_local) + the replicated surface
//     pcreg_ransac_sharded             every rank passes the same pairs; hypotheses are split over the ranks
// return the SAME result on every rank, bit for bit the single-GPU one.  The protocol is pcreg_amd/sharded.py's
// (which stays the test harness: gloo world-2 on CPU, two ranks on one GPU): per registration
//     1 ncclAllGather  of the per-rank top-2 lists ([2][Q][2] 4-byte words)      + merge kernel
//     1 ncclAllReduce  (int32 SUM) of the dense [4][Q] candidate table (one contributor per column: exact)
//     1 ncclAllGather  of the 112-byte partial RANSAC results                    + finish kernel
// RCCL is opened with dlopen at pcreg_comm_init (no link-time dependency: the library loads on boxes without
// RCCL, and lives next to a torch that bundles its own copy); collectives run on the library's comm stream.
#include "common.hpp"
#include <rccl/rccl.h>          // types and prototypes only; the functions are resolved with dlsym
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <vector>

namespace pcreg {
namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
std::mutex g_cmu;
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 0;
hipStream_t g_cstream = nullptr;
Scratch g_cs;                       // the sharded entry points' own device buffers

int load_rccl() {
    if (g_rccl.handle) return PCREG_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { set_error("RCCL not found (dlopen librccl.so.1: %s)", dlerror()); return PCREG_E_HIP; }
#define PCREG_SYM(F) g_rccl.F = (decltype(g_rccl.F))dlsym(h, "nccl" #F); if (!g_rccl.F) { set_error("librccl lacks nccl" #F); dlclose(h); return PCREG_E_HIP; }
    PCREG_SYM(GetUniqueId) PCREG_SYM(CommInitRank) PCREG_SYM(CommDestroy) PCREG_SYM(AllGather) PCREG_SYM(AllReduce) PCREG_SYM(GetErrorString)
#undef PCREG_SYM
    g_rccl.handle = h;
    return PCREG_OK;
}
#define PCREG_NCCL(call) do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) { set_error("RCCL: %s failed: %s", #call, g_rccl.GetErrorString(r__)); return PCREG_E_HIP; } } while (0)
#define CTRY(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)

int need_comm() {
    if (!g_comm) { set_error("pcreg_comm_init has not been called on this process"); return PCREG_E_ARG; }
    return PCREG_OK;
}

}  // namespace
}  // namespace pcreg

using namespace pcreg;

extern "C" {

int pcreg_comm_get_unique_id(pcreg_comm_id* id) {
    PCREG_ARG(id != nullptr);
    std::lock_guard<std::mutex> lock(g_cmu);
    static_assert(sizeof(pcreg_comm_id) == sizeof(ncclUniqueId), "pcreg_comm_id is an ncclUniqueId");
    CTRY(load_rccl());
    ncclUniqueId u;
    PCREG_NCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return PCREG_OK;
}

int pcreg_comm_init(int rank, int world, const pcreg_comm_id* id) {
    PCREG_ARG(id != nullptr && world >= 1 && rank >= 0 && rank < world);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available; libpcreg_hip has no CPU fallback");
        return PCREG_E_NODEVICE;
    }
    std::lock_guard<std::mutex> lock(g_cmu);
    if (g_comm) { set_error("pcreg_comm_init: a communicator is already open (pcreg_comm_destroy first)"); return PCREG_E_ARG; }
    CTRY(load_rccl());
    ncclUniqueId u; memcpy(&u, id, sizeof u);
    if (!g_cstream) PCREG_HIP(hipStreamCreateWithFlags(&g_cstream, hipStreamNonBlocking));
    PCREG_NCCL(g_rccl.CommInitRank(&g_comm, world, u, rank));          // on the calling thread's current device (pcreg_set_device)
    g_rank = rank; g_world = world;
    return PCREG_OK;
}

int pcreg_comm_rank(int* rank, int* world) {
    PCREG_ARG(rank && world);
    std::lock_guard<std::mutex> lock(g_cmu);
    CTRY(need_comm());
    *rank = g_rank; *world = g_world;
    return PCREG_OK;
}

int pcreg_comm_destroy(void) {
    std::lock_guard<std::mutex> lock(g_cmu);
    if (g_comm) { (void)hipStreamSynchronize(g_cstream); (void)g_rccl.CommDestroy(g_comm); g_comm = nullptr; }
    g_cs.release_all();
    g_rank = 0; g_world = 0;
    return PCREG_OK;
}

// getMatches' role on raw points (pcreg_match_points_f32) with the model rows split over the ranks.
int pcreg_match_points_sharded_f32(const float* q, int Q, int ldq, const float* m_local, int M_local, int ldm, int m_lo,
                                   int M_total, float thr_abs, float max_ratio, int unique, uint32_t* pairs, int* P) {
    PCREG_ARG(q && pairs && P && Q >= 0 && M_local >= 0 && ldq >= Q && (M_local == 0 || (m_local && ldm >= M_local)));
    PCREG_ARG(m_lo >= 0 && M_total >= 0 && (long long)m_lo + M_local <= M_total);
    std::lock_guard<std::mutex> lock(g_cmu);
    CTRY(need_comm());
    *P = 0;
    if (Q == 0 || M_total == 0) return PCREG_OK;
    const int R = g_world;
    hipStream_t st = g_cstream;
    const size_t q4 = (size_t)Q * 4;
    void *dq, *dm, *dpack, *dall, *didx, *ddist, *dcq, *dcm, *dkeep, *dcnt, *dtable, *dpairs, *ws, *wsu, *dident;
    const size_t wsb = pcreg_dev_knn2_points_f32_workspace(Q, M_local), wsu_b = pcreg_dev_unique_points_f32_workspace(Q);
    CTRY(g_cs.get(0, sizeof(float) * 3 * (size_t)Q, &dq));
    CTRY(g_cs.get(1, sizeof(float) * 3 * (size_t)(M_local > 0 ? M_local : 1), &dm));
    CTRY(g_cs.get(2, q4 * 4, &dpack));                        // [2][Q][2] words: indices, then the distances' bit patterns
    CTRY(g_cs.get(3, q4 * 4 * (size_t)R, &dall));
    CTRY(g_cs.get(4, q4 * 2, &didx));
    CTRY(g_cs.get(5, q4 * 2, &ddist));
    CTRY(g_cs.get(6, q4, &dcq));
    CTRY(g_cs.get(7, q4, &dcm));
    CTRY(g_cs.get(8, q4, &dkeep));
    CTRY(g_cs.get(9, 256, &dcnt));
    CTRY(g_cs.get(10, q4 * 4, &dtable));
    CTRY(g_cs.get(11, q4 * 2, &dpairs));
    CTRY(g_cs.get(12, wsb, &ws));
    CTRY(g_cs.get(13, wsu_b, &wsu));
    CTRY(g_cs.get(14, q4, &dident));
    int32_t* n_cand = (int32_t*)dcnt; int32_t* n_pairs = n_cand + 1;
    int32_t* idx_l = (int32_t*)dpack; float* dist_l = (float*)((int32_t*)dpack + 2 * (size_t)Q);
    if (ldq == Q) PCREG_HIP(hipMemcpyAsync(dq, q, sizeof(float) * 3 * (size_t)Q, hipMemcpyHostToDevice, st));
    else PCREG_HIP(hipMemcpy2DAsync(dq, sizeof(float) * (size_t)Q, q, sizeof(float) * (size_t)ldq, sizeof(float) * (size_t)Q, 3, hipMemcpyHostToDevice, st));
    if (M_local > 0) {
        if (ldm == M_local) PCREG_HIP(hipMemcpyAsync(dm, m_local, sizeof(float) * 3 * (size_t)M_local, hipMemcpyHostToDevice, st));
        else PCREG_HIP(hipMemcpy2DAsync(dm, sizeof(float) * (size_t)M_local, m_local, sizeof(float) * (size_t)ldm, sizeof(float) * (size_t)M_local, 3, hipMemcpyHostToDevice, st));
    }
    // 1. local top-2 (global row numbers), all_gather, merge by (distance, index)
    CTRY(pcreg_dev_knn2_points_f32((float*)dq, Q, Q, (float*)dm, M_local, M_local > 0 ? M_local : 1, m_lo, idx_l, dist_l, ws, wsb, st));
    PCREG_NCCL(g_rccl.AllGather(dpack, dall, q4, ncclInt32, g_comm, st));
    CTRY(pcreg_dev_merge_top2_strided_f32((int32_t*)dall, (float*)((int32_t*)dall + 2 * (size_t)Q), R, Q, q4, (int32_t*)didx, (float*)ddist, st));
    // 2. threshold + ratio test, redundantly on every rank
    CTRY(pcreg_dev_filter_top2_f32((int32_t*)didx, (float*)ddist, Q, M_total, thr_abs, max_ratio, (int32_t*)dcq, (int32_t*)dcm, n_cand, st));
    // 3. Unique verdict by the rank that owns the candidate's model row; 4. one integer SUM publishes verdicts + coordinates
    if (unique)
        CTRY(pcreg_dev_unique_points_f32((float*)dq, Q, Q, (float*)dm, M_local, M_local > 0 ? M_local : 1, m_lo, (int32_t*)dcq, (int32_t*)dcm, n_cand,
                                         (int32_t*)dkeep, wsu, wsu_b, st));
    CTRY(pcreg_dev_cand_table_f32((float*)dm, M_local, M_local > 0 ? M_local : 1, m_lo, (int32_t*)dcm, unique ? (int32_t*)dkeep : nullptr, n_cand, Q,
                                  (int32_t*)dtable, st));
    PCREG_NCCL(g_rccl.AllReduce(dtable, dtable, q4, ncclInt32, ncclSum, g_comm, st));
    // pair indices from the candidate lists, verdict from row 3 of the table
    CTRY(pcreg_dev_gather_pairs_f32((float*)dq, Q, Q, (float*)dtable, Q, (int32_t*)dcq, (int32_t*)dcm, unique ? (int32_t*)dtable + 3 * (size_t)Q : nullptr,
                                    n_cand, (uint32_t*)dpairs, nullptr, nullptr, n_pairs, st));
    int32_t np = 0;
    PCREG_HIP(hipMemcpyAsync(&np, n_pairs, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PCREG_HIP(hipStreamSynchronize(st));
    if (np > 0) PCREG_HIP(hipMemcpy(pairs, dpairs, sizeof(uint32_t) * 2 * (size_t)np, hipMemcpyDeviceToHost));
    *P = np;
    (void)dident;
    return PCREG_OK;
}

// ransac.m with the hypotheses of ONE registration split over the ranks (every rank passes the same pts1 / pts2).
int pcreg_ransac_sharded(const double* pts1, const double* pts2, int n, int ld, const pcreg_ransac_opts* opts,
                         double T[16], int32_t* inlier_idx, int* n_inliers, int* num_success, int* max_inliers, int* failed) {
    PCREG_ARG(pts1 && pts2 && opts && T && inlier_idx && n_inliers && num_success && max_inliers && failed && n >= 0 && ld >= n);
    PCREG_ARG(opts->iterNum >= 1 && opts->minPtNum == 3);       // the built-in sampler (global hypothesis index) is what makes the split exact
    std::lock_guard<std::mutex> lock(g_cmu);
    CTRY(need_comm());
    hipStream_t st = g_cstream;
    const int R = g_world, cap = n > 0 ? n : 1;
    const int share = (opts->iterNum + R - 1) / R, begin = std::min(g_rank * share, opts->iterNum), count = std::min(share, opts->iterNum - begin);
    void *d1, *d2, *dn, *dpart, *dall, *dres, *dinl, *ws;
    const size_t wsb = pcreg_dev_ransac_workspace(cap, count > 0 ? count : 1);
    CTRY(g_cs.get(15, sizeof(double) * 3 * (size_t)cap, &d1));
    CTRY(g_cs.get(16, sizeof(double) * 3 * (size_t)cap, &d2));
    CTRY(g_cs.get(17, 256, &dn));
    CTRY(g_cs.get(18, sizeof(pcreg_dev_ransac_part), &dpart));
    CTRY(g_cs.get(19, sizeof(pcreg_dev_ransac_part) * (size_t)R, &dall));
    CTRY(g_cs.get(20, sizeof(pcreg_dev_ransac_result), &dres));
    CTRY(g_cs.get(21, sizeof(int32_t) * (size_t)cap, &dinl));
    CTRY(g_cs.get(22, wsb, &ws));
    if (n > 0) {
        if (ld == n) {
            PCREG_HIP(hipMemcpyAsync(d1, pts1, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, st));
            PCREG_HIP(hipMemcpyAsync(d2, pts2, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, st));
        } else {
            PCREG_HIP(hipMemcpy2DAsync(d1, sizeof(double) * (size_t)cap, pts1, sizeof(double) * (size_t)ld, sizeof(double) * (size_t)n, 3, hipMemcpyHostToDevice, st));
            PCREG_HIP(hipMemcpy2DAsync(d2, sizeof(double) * (size_t)cap, pts2, sizeof(double) * (size_t)ld, sizeof(double) * (size_t)n, 3, hipMemcpyHostToDevice, st));
        }
    }
    const int32_t nh = n;
    PCREG_HIP(hipMemcpyAsync(dn, &nh, sizeof nh, hipMemcpyHostToDevice, st));
    PCREG_HIP(hipMemsetAsync(dpart, 0, sizeof(pcreg_dev_ransac_part), st));
    CTRY(pcreg_dev_ransac_partial((double*)d1, (double*)d2, (int32_t*)dn, cap, cap, opts, nullptr, begin, count, (pcreg_dev_ransac_part*)dpart, ws, wsb, st));
    static_assert(sizeof(pcreg_dev_ransac_part) == 112, "the gathered struct is 112 bytes");
    PCREG_NCCL(g_rccl.AllGather(dpart, dall, sizeof(pcreg_dev_ransac_part) / 4, ncclInt32, g_comm, st));
    CTRY(pcreg_dev_ransac_finish_parts((double*)d1, (double*)d2, (int32_t*)dn, cap, cap, opts, (pcreg_dev_ransac_part*)dall, R,
                                       (pcreg_dev_ransac_result*)dres, (int32_t*)dinl, st));
    pcreg_dev_ransac_result res;
    PCREG_HIP(hipMemcpyAsync(&res, dres, sizeof res, hipMemcpyDeviceToHost, st));
    if (n > 0) PCREG_HIP(hipMemcpyAsync(inlier_idx, dinl, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
    PCREG_HIP(hipStreamSynchronize(st));
    memcpy(T, res.T, sizeof(double) * 16);
    *n_inliers = res.n_inliers; *num_success = res.num_success; *max_inliers = res.max_inliers; *failed = res.failed;
    return PCREG_OK;
}

}  // extern "C"
