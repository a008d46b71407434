// pcreg_amd/csrc/api.hip -- the C ABI of libpcreg_hip.so (include/pcreg.h).
//
// Host tier: stage the caller's MATLAB-layout host arrays into HBM, enqueue the
// kernels on the library stream, copy the results back.  Device tier: thin argument
// checks around the launchers.  There is deliberately no CPU fallback anywhere in
// this file: without a gfx950 device every compute entry point fails with
// PCREG_E_NODEVICE.
#include "common.hpp"
#include <atomic>
#include <cstdarg>
#include <cmath>
#include <map>
#include <mutex>
#include <algorithm>
#include <vector>

namespace pcreg {

static thread_local char g_err[512] = "no error";
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

static std::mutex g_mu;
static int g_device_ok = -1;          // -1 unknown, 0 ok, else error code
static hipStream_t g_stream = nullptr;

int ensure_device() {
    if (g_device_ok == 0) return PCREG_OK;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s); libpcreg_hip has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        (void)hipGetLastError();
        return PCREG_E_NODEVICE;
    }
    int dev = 0;
    PCREG_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PCREG_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libpcreg_hip is built for gfx950 only", dev, prop.gcnArchName);
        return PCREG_E_NODEVICE;
    }
    if (!g_stream) PCREG_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    g_device_ok = 0;
    return PCREG_OK;
}

int Scratch::get(int slot, size_t bytes, void** out) {
    if (slot < 0 || slot >= kSlots) { set_error("scratch slot %d out of range", slot); return PCREG_E_ARG; }
    if (bytes == 0) bytes = 256;
    if (size[slot] < bytes) {
        if (ptr[slot]) { PCREG_HIP(hipFree(ptr[slot])); ptr[slot] = nullptr; size[slot] = 0; }
        size_t want = align_up(bytes + bytes / 4, 4096);       // grow geometrically
        PCREG_HIP(hipMalloc(&ptr[slot], want));
        size[slot] = want;
    }
    *out = ptr[slot];
    return PCREG_OK;
}
void Scratch::release_all() {
    for (int i = 0; i < kSlots; ++i) if (ptr[i]) { (void)hipFree(ptr[i]); ptr[i] = nullptr; size[i] = 0; }
}
Scratch& scratch() { static Scratch s; return s; }
static std::map<hipStream_t, Scratch>& stream_scratches() { static std::map<hipStream_t, Scratch> m; return m; }
Scratch& stream_scratch(hipStream_t st) { return stream_scratches()[st]; }

// copy an n x cols column-major host matrix (leading dimension ld) to a compact device
// matrix (leading dimension n)
template <typename T>
static int upload_cols(const T* host, int n, int ld, int cols, T* dev, hipStream_t st) {
    if (n <= 0 || cols <= 0) return PCREG_OK;
    if (ld == n) { PCREG_HIP(hipMemcpyAsync(dev, host, sizeof(T) * (size_t)n * cols, hipMemcpyHostToDevice, st)); }
    else PCREG_HIP(hipMemcpy2DAsync(dev, sizeof(T) * (size_t)n, host, sizeof(T) * (size_t)ld, sizeof(T) * (size_t)n, cols, hipMemcpyHostToDevice, st));
    return PCREG_OK;
}

}  // namespace pcreg

using namespace pcreg;

#define GUARD()                                                     \
    std::lock_guard<std::mutex> lock__(g_mu);                       \
    do { int rc__ = ensure_device(); if (rc__) return rc__; } while (0)
#define TRY(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)

namespace pcreg {
static std::atomic<int> g_debug[kDbgCount];
int debug_flag(DebugKey k) { return g_debug[k].load(std::memory_order_relaxed); }
static unsigned long long* g_match_stats = nullptr;
unsigned long long* match_stats_dev() {
    if (!debug_flag(kDbgMatchStats)) return nullptr;
    if (!g_match_stats) {
        if (hipMalloc((void**)&g_match_stats, 8 * sizeof(unsigned long long)) != hipSuccess) { g_match_stats = nullptr; return nullptr; }
        (void)hipMemset(g_match_stats, 0, 8 * sizeof(unsigned long long));
    }
    return g_match_stats;
}
}

extern "C" {

int pcreg_debug_set(const char* key, int value) {
    static const char* const names[pcreg::kDbgCount] = {"knn_exact", "match_exact", "match_force_fallback", "ransac_fused", "ransac_nolane",
                                                        "ransac_f64score", "ransac_resident_f64", "align_times", "align_shape", "seg_debug",
                                                        "seg_batched", "seg_wave_finalize", "match_stats"};
    PCREG_ARG(key != nullptr);
    for (int k = 0; k < pcreg::kDbgCount; ++k)
        if (!strcmp(key, names[k])) { pcreg::g_debug[k].store(value, std::memory_order_relaxed); return PCREG_OK; }
    pcreg::set_error("pcreg_debug_set: unknown key '%s'", key);
    return PCREG_E_ARG;
}

int pcreg_debug_match_stats(long long out[8], int reset) {
    PCREG_ARG(out != nullptr);
    for (int k = 0; k < 8; ++k) out[k] = 0;
    if (!pcreg::g_match_stats) return PCREG_OK;               // "match_stats" was never on
    PCREG_HIP(hipDeviceSynchronize());
    unsigned long long h[8];
    PCREG_HIP(hipMemcpy(h, pcreg::g_match_stats, sizeof h, hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; ++k) out[k] = (long long)h[k];
    if (reset) PCREG_HIP(hipMemset(pcreg::g_match_stats, 0, sizeof h));
    return PCREG_OK;
}

const char* pcreg_last_error(void) { return g_err; }
const char* pcreg_version(void) { return "pcreg-hip 0.1 (gfx950)"; }

int pcreg_device_count(int* count) {
    PCREG_ARG(count != nullptr);
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); }
    return PCREG_OK;
}

// page-locked staging of the host tier's result lists (pairs_fetch_*) and the event their counts are waited on
static void*  g_pin[2] = {nullptr, nullptr};
static size_t g_pin_bytes[2] = {0, 0};
static hipEvent_t g_pairs_ev = nullptr;

int pcreg_set_device(int ordinal) {
    std::lock_guard<std::mutex> lock(g_mu);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { set_error("no HIP device available"); (void)hipGetLastError(); return PCREG_E_NODEVICE; }
    PCREG_ARG(ordinal >= 0 && ordinal < count);
    scratch().release_all();
    for (auto& kv : stream_scratches()) kv.second.release_all();
    stream_scratches().clear();
    if (g_stream) { (void)hipStreamDestroy(g_stream); g_stream = nullptr; }
    if (g_pairs_ev) { (void)hipEventDestroy(g_pairs_ev); g_pairs_ev = nullptr; }
    for (int k = 0; k < 2; ++k) if (g_pin[k]) { (void)hipHostFree(g_pin[k]); g_pin[k] = nullptr; g_pin_bytes[k] = 0; }
    g_device_ok = -1;
    PCREG_HIP(hipSetDevice(ordinal));
    return ensure_device();
}

int pcreg_device_name(char* buf, int cap) {
    PCREG_ARG(buf != nullptr && cap > 0);
    GUARD();
    int dev = 0; PCREG_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop; PCREG_HIP(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, (size_t)cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return PCREG_OK;
}

// ------------------------------------------------------------------ host tier
int pcreg_estimate_transform(const double* pts1, const double* pts2, int n, int ld, double T[16], int* empty) {
    PCREG_ARG(pts1 && pts2 && T && empty && n >= 0 && ld >= n);
    GUARD();
    for (int k = 0; k < 16; ++k) T[k] = 0.0;
    *empty = 1;
    if (n < 3) return PCREG_OK;          // rank(pts1) < 3 -> [] (estimateTransform.m:11-14)
    void *d1, *d2, *dT;
    TRY(scratch().get(0, sizeof(double) * 3 * (size_t)n, &d1));
    TRY(scratch().get(1, sizeof(double) * 3 * (size_t)n, &d2));
    TRY(scratch().get(2, 256, &dT));
    TRY(upload_cols(pts1, n, ld, 3, (double*)d1, g_stream));
    TRY(upload_cols(pts2, n, ld, 3, (double*)d2, g_stream));
    int32_t* dE = (int32_t*)((char*)dT + 128);
    TRY(launch_estimate_transform((double*)d1, (double*)d2, n, n, (double*)dT, dE, g_stream));
    int32_t e = 1;
    PCREG_HIP(hipMemcpyAsync(T, dT, sizeof(double) * 16, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipMemcpyAsync(&e, dE, sizeof(int32_t), hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    *empty = e;
    return PCREG_OK;
}

int pcreg_calc_dists(const double T[16], const double* pts1, const double* pts2, int n, int ld, double* d) {
    PCREG_ARG(T && pts1 && pts2 && d && n >= 0 && ld >= n);
    GUARD();
    if (n == 0) return PCREG_OK;
    void *d1, *d2, *dT, *dd;
    TRY(scratch().get(0, sizeof(double) * 3 * (size_t)n, &d1));
    TRY(scratch().get(1, sizeof(double) * 3 * (size_t)n, &d2));
    TRY(scratch().get(2, 256, &dT));
    TRY(scratch().get(3, sizeof(double) * (size_t)n, &dd));
    TRY(upload_cols(pts1, n, ld, 3, (double*)d1, g_stream));
    TRY(upload_cols(pts2, n, ld, 3, (double*)d2, g_stream));
    PCREG_HIP(hipMemcpyAsync(dT, T, sizeof(double) * 16, hipMemcpyHostToDevice, g_stream));
    TRY(launch_calc_dists((double*)dT, (double*)d1, (double*)d2, n, n, (double*)dd, g_stream));
    PCREG_HIP(hipMemcpyAsync(d, dd, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    return PCREG_OK;
}

static int ransac_host(const double* pts1, const double* pts2, int total, int ld, const int32_t* offsets, int B,
                       const pcreg_ransac_opts* o, const int32_t* sample_idx, double* T, int32_t* inlier_idx,
                       int32_t* n_inliers, int32_t* num_success, int32_t* max_inliers, int32_t* failed,
                       int32_t* iter_inl, int32_t* iter_inl_ref) {
    PCREG_ARG(o->iterNum >= 1 && o->minPtNum >= 3);
    PCREG_ARG(o->minPtNum == 3 || sample_idx != nullptr);
    int max_n = 0;
    for (int b = 0; b < B; ++b) { int nb = offsets[b + 1] - offsets[b]; PCREG_ARG(nb >= 0); if (nb > max_n) max_n = nb; }
    PCREG_ARG(offsets[0] == 0 && offsets[B] == total);
    size_t hyps = (size_t)o->iterNum * B;
    size_t wsb = ransac_workspace_bytes(o->iterNum, B, max_n);
    void *d1, *d2, *dOff, *dS = nullptr, *dOut, *dInl, *ws, *dI1 = nullptr, *dI2 = nullptr;
    size_t tot = (size_t)(total > 0 ? total : 1);
    TRY(scratch().get(0, sizeof(double) * 3 * tot, &d1));
    TRY(scratch().get(1, sizeof(double) * 3 * tot, &d2));
    TRY(scratch().get(2, sizeof(int32_t) * ((size_t)B + 1), &dOff));
    TRY(scratch().get(3, sizeof(pcreg_dev_ransac_result) * (size_t)B, &dOut));
    TRY(scratch().get(4, sizeof(int32_t) * tot, &dInl));
    TRY(scratch().get(5, wsb, &ws));
    if (sample_idx) {
        TRY(scratch().get(6, sizeof(int32_t) * hyps * o->minPtNum, &dS));
        PCREG_HIP(hipMemcpyAsync(dS, sample_idx, sizeof(int32_t) * hyps * o->minPtNum, hipMemcpyHostToDevice, g_stream));
    }
    if (iter_inl) TRY(scratch().get(7, sizeof(int32_t) * hyps, &dI1));
    if (iter_inl_ref) TRY(scratch().get(8, sizeof(int32_t) * hyps, &dI2));
    TRY(upload_cols(pts1, total, ld, 3, (double*)d1, g_stream));
    TRY(upload_cols(pts2, total, ld, 3, (double*)d2, g_stream));
    PCREG_HIP(hipMemcpyAsync(dOff, offsets, sizeof(int32_t) * ((size_t)B + 1), hipMemcpyHostToDevice, g_stream));
    TRY(launch_ransac((double*)d1, (double*)d2, total, (int32_t*)dOff, nullptr, max_n, B, *o, (int32_t*)dS,
                      (pcreg_dev_ransac_result*)dOut, (int32_t*)dInl, (int32_t*)dI1, (int32_t*)dI2, ws, wsb, g_stream));
    std::vector<pcreg_dev_ransac_result> res((size_t)B);
    PCREG_HIP(hipMemcpyAsync(res.data(), dOut, sizeof(pcreg_dev_ransac_result) * (size_t)B, hipMemcpyDeviceToHost, g_stream));
    if (total > 0) PCREG_HIP(hipMemcpyAsync(inlier_idx, dInl, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, g_stream));
    if (iter_inl) PCREG_HIP(hipMemcpyAsync(iter_inl, dI1, sizeof(int32_t) * hyps, hipMemcpyDeviceToHost, g_stream));
    if (iter_inl_ref) PCREG_HIP(hipMemcpyAsync(iter_inl_ref, dI2, sizeof(int32_t) * hyps, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    for (int b = 0; b < B; ++b) {
        memcpy(T + 16 * (size_t)b, res[b].T, sizeof(double) * 16);
        n_inliers[b] = res[b].n_inliers; num_success[b] = res[b].num_success;
        max_inliers[b] = res[b].max_inliers; failed[b] = res[b].failed;
    }
    return PCREG_OK;
}

int pcreg_ransac(const double* pts1, const double* pts2, int n, int ld, const pcreg_ransac_opts* opts,
                 const int32_t* sample_idx, double T[16], int32_t* inlier_idx, int* n_inliers, int* num_success,
                 int* max_inliers, int* failed, int32_t* iter_inl, int32_t* iter_inl_ref) {
    PCREG_ARG(pts1 && pts2 && opts && T && inlier_idx && n_inliers && num_success && max_inliers && failed);
    PCREG_ARG(n >= 0 && ld >= n);
    GUARD();
    int32_t offsets[2] = {0, n};
    int32_t ni = 0, ns = 0, mi = 0, fl = 1;
    TRY(ransac_host(pts1, pts2, n, ld, offsets, 1, opts, sample_idx, T, inlier_idx, &ni, &ns, &mi, &fl, iter_inl, iter_inl_ref));
    *n_inliers = ni; *num_success = ns; *max_inliers = mi; *failed = fl;
    return PCREG_OK;
}

int pcreg_ransac_batched(const double* pts1, const double* pts2, int total, int ld, const int32_t* offsets, int B,
                         const pcreg_ransac_opts* opts, const int32_t* sample_idx, double* T, int32_t* inlier_idx,
                         int32_t* n_inliers, int32_t* num_success, int32_t* max_inliers, int32_t* failed) {
    PCREG_ARG(pts1 && pts2 && offsets && opts && T && inlier_idx && n_inliers && num_success && max_inliers && failed);
    PCREG_ARG(total >= 0 && ld >= total && B >= 1);
    GUARD();
    return ransac_host(pts1, pts2, total, ld, offsets, B, opts, sample_idx, T, inlier_idx, n_inliers, num_success,
                       max_inliers, failed, nullptr, nullptr);
}

int pcreg_knn2_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm, int32_t* idx, float* dist) {
    PCREG_ARG(q && m && idx && dist && Q >= 0 && M >= 0 && ldq >= Q && ldm >= M);
    GUARD();
    if (Q == 0) return PCREG_OK;
    void *dq, *dm, *di, *dd, *ws;
    size_t wsb = knn2_points_workspace_bytes(Q, M);
    TRY(scratch().get(0, sizeof(float) * 3 * (size_t)Q, &dq));
    TRY(scratch().get(1, sizeof(float) * 3 * (size_t)(M > 0 ? M : 1), &dm));
    TRY(scratch().get(2, sizeof(int32_t) * 2 * (size_t)Q, &di));
    TRY(scratch().get(3, sizeof(float) * 2 * (size_t)Q, &dd));
    TRY(scratch().get(4, wsb, &ws));
    TRY(upload_cols(q, Q, ldq, 3, (float*)dq, g_stream));
    TRY(upload_cols(m, M, ldm, 3, (float*)dm, g_stream));
    TRY(launch_knn2_points_f32((float*)dq, Q, Q, (float*)dm, M, M, 0, (int32_t*)di, (float*)dd, ws, wsb, g_stream));
    PCREG_HIP(hipMemcpyAsync(idx, di, sizeof(int32_t) * 2 * (size_t)Q, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipMemcpyAsync(dist, dd, sizeof(float) * 2 * (size_t)Q, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    return PCREG_OK;
}

// ---- prepared models ---------------------------------------------------------------------------------------------
}  // extern "C" (the handle types are C++ structs behind opaque C names)
struct pcreg_dev_model { pcreg::ModelView v; void* block; };
struct pcreg_model { pcreg_dev_model* dm; float* d_m; int M; };
extern "C" {

static int dev_model_create(const float* m, int M, int ldm, hipStream_t st, pcreg_dev_model** out) {
    *out = nullptr;
    void* block = nullptr;
    PCREG_HIP(hipMalloc(&block, model_prep_bytes(M)));
    pcreg_dev_model* h = new pcreg_dev_model{model_view(m, M, ldm, block), block};
    int rc = launch_model_prepare(h->v, st);
    if (rc) { (void)hipFree(block); delete h; return rc; }
    *out = h;
    return PCREG_OK;
}

int pcreg_dev_model_create(const float* m, int M, int ldm, void* stream, pcreg_dev_model** model) {
    PCREG_ARG(model != nullptr && M >= 0 && ldm >= M && (M == 0 || m != nullptr));
    GUARD();
    return dev_model_create(m, M, ldm, (hipStream_t)stream, model);
}
int pcreg_dev_model_destroy(pcreg_dev_model* model) {
    if (!model) return PCREG_OK;
    std::lock_guard<std::mutex> lock(g_mu);
    (void)hipDeviceSynchronize();                 // searches that still read the prepared block
    (void)hipFree(model->block);
    delete model;
    return PCREG_OK;
}
size_t pcreg_dev_model_search_workspace(int Q, int M) { return search_ws_bytes(Q, M); }
int pcreg_dev_model_search_f32(const pcreg_dev_model* model, const float* q, int Q, int ldq, int32_t idx_base, int32_t* idx,
                               float* dist, void* workspace, size_t workspace_bytes, void* stream) {
    PCREG_ARG(model && q && idx && dist && workspace);
    GUARD();
    return launch_model_search(model->v, q, Q, ldq, idx_base, idx, dist, workspace, workspace_bytes, true, true, (hipStream_t)stream);
}
int pcreg_dev_model_match_f32(const pcreg_dev_model* model, const float* q, int Q, int ldq, const int32_t* idx, const float* dist,
                              float thr_abs, float max_ratio, int unique, void* workspace, size_t workspace_bytes, uint32_t* pairs,
                              double* pts1, double* pts2, int32_t* n_pairs, void* stream) {
    PCREG_ARG(model && q && idx && dist && workspace && n_pairs);
    GUARD();
    return launch_match_finish(model->v, q, Q, ldq, idx, dist, thr_abs, max_ratio, unique, workspace, workspace_bytes, pairs, pts1, pts2,
                               n_pairs, (hipStream_t)stream);
}
int pcreg_dev_model_match_table_f32(const pcreg_dev_model* model, int32_t m_lo, int M_total, const float* q, int Q, int ldq,
                                    const int32_t* idx, const float* dist, float thr_abs, float max_ratio, int unique,
                                    void* workspace, size_t workspace_bytes, int32_t* table, void* stream) {
    PCREG_ARG(model && q && idx && dist && workspace && table);
    GUARD();
    return launch_match_table(model->v, m_lo, M_total, q, Q, ldq, idx, dist, thr_abs, max_ratio, unique, workspace, workspace_bytes, table,
                              (hipStream_t)stream);
}
int pcreg_dev_match_from_table_f32(const float* q, int Q, int ldq, int M_total, const int32_t* idx, const float* dist, float thr_abs,
                                   float max_ratio, const int32_t* table, void* workspace, size_t workspace_bytes, uint32_t* pairs,
                                   double* pts1, double* pts2, int32_t* n_pairs, void* stream) {
    PCREG_ARG(q && idx && dist && table && workspace && n_pairs);
    GUARD();
    return launch_match_from_table(q, Q, ldq, M_total, idx, dist, thr_abs, max_ratio, table, workspace, workspace_bytes, pairs, pts1, pts2,
                                   n_pairs, (hipStream_t)stream);
}

// matchFeatures' chain on raw points against a prepared model: upload the surface, search (4 launches), match (1)
static int match_points_on_view(const ModelView& v, const float* q, int Q, int ldq, float thr_abs, float max_ratio, int unique,
                                uint32_t* pairs, int* P) {
    *P = 0;
    if (Q == 0 || v.M == 0) return PCREG_OK;
    void *dq, *di, *dd, *ws, *dcnt, *dpairs;
    const size_t wsb = search_ws_bytes(Q, v.M);
    TRY(scratch().get(0, sizeof(float) * 3 * (size_t)Q, &dq));
    TRY(scratch().get(2, sizeof(int32_t) * 2 * (size_t)Q, &di));
    TRY(scratch().get(3, sizeof(float) * 2 * (size_t)Q, &dd));
    TRY(scratch().get(4, wsb, &ws));
    TRY(scratch().get(8, 256, &dcnt));
    TRY(scratch().get(9, sizeof(uint32_t) * 2 * (size_t)Q, &dpairs));
    int32_t* n_pairs = (int32_t*)dcnt;
    TRY(upload_cols(q, Q, ldq, 3, (float*)dq, g_stream));
    TRY(launch_model_search(v, (float*)dq, Q, Q, 0, (int32_t*)di, (float*)dd, ws, wsb, true, true, g_stream));
    TRY(launch_match_finish(v, (float*)dq, Q, Q, (int32_t*)di, (float*)dd, thr_abs, max_ratio, unique, ws, wsb, (uint32_t*)dpairs, nullptr,
                            nullptr, n_pairs, g_stream));
    int32_t np = 0;
    PCREG_HIP(hipMemcpyAsync(&np, n_pairs, sizeof(int32_t), hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    if (np > 0) PCREG_HIP(hipMemcpy(pairs, dpairs, sizeof(uint32_t) * 2 * (size_t)np, hipMemcpyDeviceToHost));
    *P = np;
    return PCREG_OK;
}

int pcreg_model_create(const float* m, int M, int ldm, pcreg_model** model) {
    PCREG_ARG(model != nullptr && M >= 0 && ldm >= M && (M == 0 || m != nullptr));
    GUARD();
    *model = nullptr;
    float* d_m = nullptr;
    PCREG_HIP(hipMalloc((void**)&d_m, sizeof(float) * 3 * (size_t)(M > 0 ? M : 1)));
    int rc = upload_cols(m, M, ldm, 3, d_m, g_stream);
    pcreg_dev_model* dm = nullptr;
    if (!rc) rc = dev_model_create(d_m, M, M > 0 ? M : 1, g_stream, &dm);
    if (!rc && hipStreamSynchronize(g_stream) != hipSuccess) { set_error("model preparation failed"); rc = PCREG_E_HIP; }
    if (rc) { if (dm) { (void)hipFree(dm->block); delete dm; } (void)hipFree(d_m); return rc; }
    *model = new pcreg_model{dm, d_m, M};
    return PCREG_OK;
}
int pcreg_model_destroy(pcreg_model* model) {
    if (!model) return PCREG_OK;
    std::lock_guard<std::mutex> lock(g_mu);
    (void)hipDeviceSynchronize();
    if (model->dm) { (void)hipFree(model->dm->block); delete model->dm; }
    (void)hipFree(model->d_m);
    delete model;
    return PCREG_OK;
}
int pcreg_model_match_points_f32(pcreg_model* model, const float* q, int Q, int ldq, float thr_abs, float max_ratio, int unique,
                                 uint32_t* pairs, int* P) {
    PCREG_ARG(model && model->dm && q && pairs && P && Q >= 0 && ldq >= Q);
    GUARD();
    return match_points_on_view(model->dm->v, q, Q, ldq, thr_abs, max_ratio, unique, pairs, P);
}

int pcreg_match_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm, float thr_abs,
                           float max_ratio, int unique, uint32_t* pairs, int* P) {
    PCREG_ARG(q && m && pairs && P && Q >= 0 && M >= 0 && ldq >= Q && ldm >= M);
    GUARD();
    *P = 0;
    if (Q == 0 || M == 0) return PCREG_OK;
    void *dm, *block;
    TRY(scratch().get(1, sizeof(float) * 3 * (size_t)M, &dm));
    TRY(scratch().get(10, model_prep_bytes(M), &block));
    TRY(upload_cols(m, M, ldm, 3, (float*)dm, g_stream));
    const ModelView v = model_view((float*)dm, M, M, block);
    TRY(launch_model_prepare(v, g_stream));
    return match_points_on_view(v, q, Q, ldq, thr_abs, max_ratio, unique, pairs, P);
}

// raw descriptor matrices on the device (column-major, ld = rows) -> getMatches.m:22-56 -> pairs on the host.  rawS / rawM are not
// modified: the preprocessing writes its own copies (scratch slots 0, 1).
static int match_dev_raw(const double* rawS, int Q, const double* rawM, int M, int D, const pcreg_match_opts* o, uint32_t* pairs, double* metric, int* P) {
    *P = 0;
    if (Q == 0 || M == 0) return PCREG_OK;
    const int Dp = D + (o->unnormalize ? 1 : 0);
    void *dS, *dM, *ws, *dpairs, *dmet, *dcnt, *pws = nullptr;
    size_t wsb = match_features_workspace_bytes(Q, M, Dp);
    TRY(scratch().get(0, sizeof(double) * (size_t)Q * Dp, &dS));
    TRY(scratch().get(1, sizeof(double) * (size_t)M * Dp, &dM));
    TRY(scratch().get(2, wsb, &ws));
    TRY(scratch().get(3, sizeof(uint32_t) * 2 * (size_t)Q, &dpairs));
    TRY(scratch().get(4, sizeof(double) * (size_t)Q, &dmet));
    TRY(scratch().get(5, 256, &dcnt));
    size_t pwb = sizeof(double) * ((size_t)Q + M + 1);
    TRY(scratch().get(8, pwb, &pws));
    TRY(launch_preprocess(rawS, Q, Q, rawM, M, M, D, *o, (double*)dS, (double*)dM, pws, pwb, g_stream));
    if (!o->prenormalized) {
        TRY(launch_normalize_rows2((double*)dS, Q, Q, (double*)dM, M, M, Dp, g_stream));
    }
    TRY(launch_match_features((double*)dS, Q, Q, (double*)dM, M, M, Dp, *o, (uint32_t*)dpairs,
                              metric ? (double*)dmet : nullptr, (int32_t*)dcnt, ws, wsb, g_stream));
    int32_t np = 0;
    PCREG_HIP(hipMemcpyAsync(&np, dcnt, sizeof(int32_t), hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    if (np > 0) {
        PCREG_HIP(hipMemcpy(pairs, dpairs, sizeof(uint32_t) * 2 * (size_t)np, hipMemcpyDeviceToHost));
        if (metric) PCREG_HIP(hipMemcpy(metric, dmet, sizeof(double) * (size_t)np, hipMemcpyDeviceToHost));
    }
    *P = np;
    return PCREG_OK;
}

static int match_host(const double* f1, int Q, int ld1, const double* f2, int M, int ld2, int D,
                      const pcreg_match_opts* o, bool preprocess, uint32_t* pairs, double* metric, int* P) {
    *P = 0;
    if (Q == 0 || M == 0) return PCREG_OK;
    if (preprocess) {
        void *rawS = nullptr, *rawM = nullptr;
        TRY(scratch().get(6, sizeof(double) * (size_t)Q * D, &rawS));
        TRY(scratch().get(7, sizeof(double) * (size_t)M * D, &rawM));
        TRY(upload_cols(f1, Q, ld1, D, (double*)rawS, g_stream));
        TRY(upload_cols(f2, M, ld2, D, (double*)rawM, g_stream));
        return match_dev_raw((const double*)rawS, Q, (const double*)rawM, M, D, o, pairs, metric, P);
    }
    const int Dp = D;
    void *dS, *dM, *ws, *dpairs, *dmet, *dcnt;
    size_t wsb = match_features_workspace_bytes(Q, M, Dp);
    TRY(scratch().get(0, sizeof(double) * (size_t)Q * Dp, &dS));
    TRY(scratch().get(1, sizeof(double) * (size_t)M * Dp, &dM));
    TRY(scratch().get(2, wsb, &ws));
    TRY(scratch().get(3, sizeof(uint32_t) * 2 * (size_t)Q, &dpairs));
    TRY(scratch().get(4, sizeof(double) * (size_t)Q, &dmet));
    TRY(scratch().get(5, 256, &dcnt));
    TRY(upload_cols(f1, Q, ld1, D, (double*)dS, g_stream));
    TRY(upload_cols(f2, M, ld2, D, (double*)dM, g_stream));
    if (!o->prenormalized) {
        TRY(launch_normalize_rows2((double*)dS, Q, Q, (double*)dM, M, M, Dp, g_stream));
    }
    TRY(launch_match_features((double*)dS, Q, Q, (double*)dM, M, M, Dp, *o, (uint32_t*)dpairs,
                              metric ? (double*)dmet : nullptr, (int32_t*)dcnt, ws, wsb, g_stream));
    int32_t np = 0;
    PCREG_HIP(hipMemcpyAsync(&np, dcnt, sizeof(int32_t), hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    if (np > 0) {
        PCREG_HIP(hipMemcpy(pairs, dpairs, sizeof(uint32_t) * 2 * (size_t)np, hipMemcpyDeviceToHost));
        if (metric) PCREG_HIP(hipMemcpy(metric, dmet, sizeof(double) * (size_t)np, hipMemcpyDeviceToHost));
    }
    *P = np;
    return PCREG_OK;
}

// ---- resident descriptor sets (host tier): one surface set against hundreds of row subsets of one model set
// (completeExperimentFast.m:101-150) without re-uploading ~28 MB of doubles per sphere
struct pcreg_desc_set {
    double* d; int n, D;            // n x D column-major on the device (ld = n), as uploaded
    double* rows;                   // the dense row-major copy the segmented matcher reads, made at its first use
    double* prep;                   // as a MODEL of the segmented matcher: powered rows + row scalars (SegPreparedModel), for ...
    int prep_cm; double prep_factor;   // ... these getMatches options (made at the first use, remade when they change)
};

__global__ void gather_cols_kernel(const double* __restrict__ src, int n_src, int D, const int32_t* __restrict__ rows, int n, double* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, d = blockIdx.y;
    if (i < n) dst[i + (size_t)d * n] = src[rows[i] + (size_t)d * n_src];
}

int pcreg_desc_set_create(const double* desc, int n, int ld, int D, pcreg_desc_set** set) {
    PCREG_ARG(set != nullptr && n >= 0 && D >= 1 && ld >= n && (n == 0 || desc != nullptr));
    GUARD();
    *set = nullptr;
    double* d = nullptr;
    PCREG_HIP(hipMalloc((void**)&d, sizeof(double) * (size_t)(n > 0 ? n : 1) * D));
    int rc = upload_cols(desc, n, ld, D, d, g_stream);
    if (!rc && hipStreamSynchronize(g_stream) != hipSuccess) { set_error("descriptor upload failed"); rc = PCREG_E_HIP; }
    if (rc) { (void)hipFree(d); return rc; }
    *set = new pcreg_desc_set{d, n, D, nullptr, nullptr, 0, 0.0};
    return PCREG_OK;
}
int pcreg_desc_set_destroy(pcreg_desc_set* set) {
    if (!set) return PCREG_OK;
    std::lock_guard<std::mutex> lock(g_mu);
    (void)hipDeviceSynchronize();
    (void)hipFree(set->d);
    if (set->rows) (void)hipFree(set->rows);
    if (set->prep) (void)hipFree(set->prep);
    delete set;
    return PCREG_OK;
}
int pcreg_desc_set_size(const pcreg_desc_set* set, int* n, int* D) {
    PCREG_ARG(set && n && D);
    *n = set->n; *D = set->D;
    return PCREG_OK;
}
int pcreg_get_matches_on_sets(const pcreg_desc_set* surface, const pcreg_desc_set* model, const int32_t* model_rows, int n_rows,
                              const pcreg_match_opts* par, uint32_t* pairs, double* metric, int* P) {
    PCREG_ARG(surface && model && par && pairs && P && surface->D == model->D && n_rows >= 0 && (model_rows || n_rows == 0));
    PCREG_ARG(par->metric == PCREG_METRIC_SAD || par->metric == PCREG_METRIC_SSD);
    GUARD();
    *P = 0;
    const int Q = surface->n, D = surface->D;
    if (!model_rows) return match_dev_raw(surface->d, Q, model->d, model->n, D, par, pairs, metric, P);      // descModel(:, :)
    for (int k = 0; k < n_rows; ++k) PCREG_ARG(model_rows[k] >= 0 && model_rows[k] < model->n);
    if (Q == 0 || n_rows == 0) return PCREG_OK;
    void *rawM, *drows;
    TRY(scratch().get(7, sizeof(double) * (size_t)n_rows * D, &rawM));
    TRY(scratch().get(9, sizeof(int32_t) * (size_t)n_rows, &drows));
    PCREG_HIP(hipMemcpyAsync(drows, model_rows, sizeof(int32_t) * (size_t)n_rows, hipMemcpyHostToDevice, g_stream));
    hipLaunchKernelGGL(gather_cols_kernel, dim3((n_rows + 255) / 256, D), dim3(256), 0, g_stream, model->d, model->n, D, (const int32_t*)drows, n_rows, (double*)rawM);   // descModel(rows, :)
    PCREG_HIP(hipGetLastError());
    return match_dev_raw(surface->d, Q, (const double*)rawM, n_rows, D, par, pairs, metric, P);
}

int pcreg_match_features(const double* f1, int Q, int ld1, const double* f2, int M, int ld2, int D,
                         const pcreg_match_opts* opts, uint32_t* pairs, double* metric, int* P) {
    PCREG_ARG(f1 && f2 && opts && pairs && P && Q >= 0 && M >= 0 && D >= 1 && ld1 >= Q && ld2 >= M);
    PCREG_ARG(opts->metric == PCREG_METRIC_SAD || opts->metric == PCREG_METRIC_SSD);
    GUARD();
    return match_host(f1, Q, ld1, f2, M, ld2, D, opts, false, pairs, metric, P);
}

int pcreg_get_matches(const double* descSurface, int Q, int ldS, const double* descModel, int M, int ldM, int D,
                      const pcreg_match_opts* par, uint32_t* pairs, double* metric, int* P) {
    PCREG_ARG(descSurface && descModel && par && pairs && P && Q >= 0 && M >= 0 && D >= 1 && ldS >= Q && ldM >= M);
    PCREG_ARG(par->metric == PCREG_METRIC_SAD || par->metric == PCREG_METRIC_SSD);
    GUARD();
    return match_host(descSurface, Q, ldS, descModel, M, ldM, D, par, true, pairs, metric, P);
}

// getLocalPoints.m:8-35, host tier.  *n_out = rows returned; 0 is MATLAB's [] (a gate failed, or nothing inside)
int pcreg_get_local_points(const double* pts, int N, int ld, double R, const double c[3], double min_points, double max_points,
                           int single_mode, double* pts_sphere, double* dists, int* n_out) {
    PCREG_ARG(pts && c && pts_sphere && n_out && N >= 0 && ld >= N && single_mode >= 0 && single_mode <= 2);
    GUARD();
    *n_out = 0;
    if (N == 0) return PCREG_OK;
    void *dp, *dout, *dd, *dt, *ws;
    TRY(scratch().get(0, sizeof(double) * 3 * (size_t)N, &dp));
    TRY(scratch().get(1, sizeof(double) * 3 * (size_t)N, &dout));
    TRY(scratch().get(2, sizeof(double) * (size_t)N, &dd));
    TRY(scratch().get(3, 256, &dt));
    TRY(scratch().get(4, local_points_workspace_bytes(N), &ws));
    TRY(upload_cols(pts, N, ld, 3, (double*)dp, g_stream));
    TRY(launch_local_points((const double*)dp, N, N, R, c, single_mode, (double*)dout, N, (double*)dd, (int32_t*)dt, ws, local_points_workspace_bytes(N), g_stream));
    int32_t tot[2] = {0, 0};
    PCREG_HIP(hipMemcpyAsync(tot, dt, sizeof(tot), hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    // :17 the box count against min_points, :31 the sphere count against both gates (comparisons as MATLAB makes them: inf allowed)
    if ((double)tot[0] < min_points || (double)tot[1] < min_points || (double)tot[1] > max_points || tot[1] == 0) return PCREG_OK;
    const size_t n = (size_t)tot[1];
    for (int k = 0; k < 3; ++k)
        PCREG_HIP(hipMemcpyAsync(pts_sphere + (size_t)k * n, (const double*)dout + (size_t)k * N, sizeof(double) * n, hipMemcpyDeviceToHost, g_stream));
    if (dists) PCREG_HIP(hipMemcpyAsync(dists, dd, sizeof(double) * n, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    *n_out = tot[1];
    return PCREG_OK;
}

// ---- the pair lists of a segmented call on their way back --------------------------------------------------------------------
// The device holds [S][vs][2] with the first n_pairs[z] pairs of row z valid (vs = the surface's rows: Unique's capacity); a sphere
// keeps a few hundred of its ~2000 slots, so copying the array as it is moved 5 MB of mostly nothing into pageable memory (~0.5 ms
// of the 7 ms sweep).  Now: the counts go to a page-locked buffer FIRST (begin: right behind the matcher, in front of whatever the
// caller launches next), the host waits for them alone (enqueue), packs the columns that hold pairs on the device and fetches that
// block through page-locked memory; finish() deals it into the caller's array (same layout; slots past n_pairs[z] are not written).
static int pinned_get(int k, size_t bytes, void** out) {
    if (g_pin_bytes[k] < bytes) {
        if (g_pin[k]) { PCREG_HIP(hipHostFree(g_pin[k])); g_pin[k] = nullptr; g_pin_bytes[k] = 0; }
        const size_t want = align_up(bytes + bytes / 4, 4096);
        PCREG_HIP(hipHostMalloc(&g_pin[k], want, hipHostMallocDefault));
        g_pin_bytes[k] = want;
    }
    *out = g_pin[k];
    return PCREG_OK;
}
struct PairFetch { int32_t* h_np = nullptr; void* h_packed = nullptr; int m = 0; bool whole = false; };
static int pairs_fetch_begin(const int32_t* dn, int S, PairFetch& f) {
    void* h;
    TRY(pinned_get(0, sizeof(int32_t) * (size_t)S, &h));
    f.h_np = (int32_t*)h;
    if (!g_pairs_ev) PCREG_HIP(hipEventCreateWithFlags(&g_pairs_ev, hipEventDisableTiming));
    PCREG_HIP(hipMemcpyAsync(f.h_np, dn, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipEventRecord(g_pairs_ev, g_stream));
    return PCREG_OK;
}
static int pairs_fetch_enqueue(PairFetch& f, const uint32_t* dp, size_t vs, int S, int pack_slot, uint32_t* pairs_all) {
    PCREG_HIP(hipEventSynchronize(g_pairs_ev));
    int m = 0;
    for (int z = 0; z < S; ++z) m = std::max(m, (int)f.h_np[z]);
    f.m = m;
    if (m == 0) return PCREG_OK;
    if ((size_t)m * 2 > vs) {                         // most slots hold pairs: the array as it is
        f.whole = true;
        PCREG_HIP(hipMemcpyAsync(pairs_all, dp, sizeof(uint32_t) * (size_t)S * vs * 2, hipMemcpyDeviceToHost, g_stream));
        return PCREG_OK;
    }
    void* packed;
    const size_t row = sizeof(uint32_t) * 2 * (size_t)m;
    TRY(scratch().get(pack_slot, row * (size_t)S, &packed));
    TRY(pinned_get(1, row * (size_t)S, &f.h_packed));
    PCREG_HIP(hipMemcpy2DAsync(packed, row, dp, sizeof(uint32_t) * 2 * vs, row, (size_t)S, hipMemcpyDeviceToDevice, g_stream));
    PCREG_HIP(hipMemcpyAsync(f.h_packed, packed, row * (size_t)S, hipMemcpyDeviceToHost, g_stream));
    return PCREG_OK;
}
// after the stream has been synchronised
static void pairs_fetch_finish(const PairFetch& f, size_t vs, int S, uint32_t* pairs_all, int32_t* n_pairs) {
    for (int z = 0; z < S; ++z) {
        n_pairs[z] = f.h_np[z];
        if (!f.whole && f.m > 0 && f.h_np[z] > 0)
            memcpy(pairs_all + (size_t)z * vs * 2, (const char*)f.h_packed + sizeof(uint32_t) * 2 * (size_t)f.m * z, sizeof(uint32_t) * 2 * (size_t)f.h_np[z]);
    }
}

// getMatches for S row subsets of one model set, host tier (the parfor of completeExperimentFast.m:131-149 as ONE call)
int pcreg_get_matches_segmented(const double* descSurface, int Q, int ldS, const double* descModel, int VM, int ldM, int D,
                                const int32_t* seg_rows, const int32_t* seg_off, int S, const pcreg_match_opts* par,
                                uint32_t* pairs_all, int32_t* n_pairs) {
    PCREG_ARG(descSurface && descModel && seg_off && par && pairs_all && n_pairs && Q >= 0 && VM >= 0 && D >= 1 && S >= 0 && ldS >= Q && ldM >= VM);
    PCREG_ARG(S <= 65535);
    if (par->metric != PCREG_METRIC_SAD) { set_error("pcreg_get_matches_segmented: Metric must be SAD (call pcreg_get_matches per segment for SSD)"); return PCREG_E_ARG; }
    GUARD();
    if (S == 0) return PCREG_OK;
    PCREG_ARG(seg_off[0] == 0);
    int n_max = 0;
    for (int z = 0; z < S; ++z) { const int n = seg_off[z + 1] - seg_off[z]; PCREG_ARG(n >= 0); if (n > n_max) n_max = n; }
    const int tot = seg_off[S];
    PCREG_ARG(tot == 0 || seg_rows);
    for (int k = 0; k < tot; ++k) PCREG_ARG(seg_rows[k] >= 0 && seg_rows[k] < VM);
    if (Q == 0 || VM == 0 || tot == 0) { for (int z = 0; z < S; ++z) n_pairs[z] = 0; return PCREG_OK; }
    const size_t q = (size_t)Q, vm = (size_t)VM;
    void *fS, *fM, *rS, *rM, *dr, *doff, *dp, *dn, *ws;
    TRY(scratch().get(0, sizeof(double) * q * D, &fS));
    TRY(scratch().get(1, sizeof(double) * vm * D, &fM));
    TRY(scratch().get(2, sizeof(double) * q * D, &rS));
    TRY(scratch().get(3, sizeof(double) * vm * D, &rM));
    TRY(scratch().get(4, sizeof(int32_t) * (size_t)tot, &dr));
    TRY(scratch().get(5, sizeof(int32_t) * ((size_t)S + 1), &doff));
    TRY(scratch().get(6, sizeof(uint32_t) * (size_t)S * q * 2, &dp));
    TRY(scratch().get(7, sizeof(int32_t) * (size_t)S, &dn));
    const size_t wsb = get_matches_segmented_workspace_bytes(Q, VM, D, S, tot, n_max);
    TRY(scratch().get(8, wsb, &ws));
    TRY(upload_cols(descSurface, Q, ldS, D, (double*)fS, g_stream));
    TRY(upload_cols(descModel, VM, ldM, D, (double*)fM, g_stream));
    TRY(launch_transpose_rows((const double*)fS, Q, Q, D, (double*)rS, g_stream));          // MATLAB's n x D -> dense rows
    TRY(launch_transpose_rows((const double*)fM, VM, VM, D, (double*)rM, g_stream));
    PCREG_HIP(hipMemcpyAsync(dr, seg_rows, sizeof(int32_t) * (size_t)tot, hipMemcpyHostToDevice, g_stream));
    PCREG_HIP(hipMemcpyAsync(doff, seg_off, sizeof(int32_t) * ((size_t)S + 1), hipMemcpyHostToDevice, g_stream));
    TRY(launch_get_matches_segmented((const double*)rS, Q, (const double*)rM, VM, D, (const int32_t*)dr, (const int32_t*)doff, S, tot, n_max, *par,
                                     (uint32_t*)dp, nullptr, (int32_t*)dn, ws, wsb, g_stream));
    PairFetch pf;
    TRY(pairs_fetch_begin((const int32_t*)dn, S, pf));
    TRY(pairs_fetch_enqueue(pf, (const uint32_t*)dp, q, S, 20, pairs_all));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    pairs_fetch_finish(pf, q, S, pairs_all, n_pairs);
    return PCREG_OK;
}

// the set's rows as the segmented matcher wants them (dense row-major), transposed once per set (caller holds the library lock)
static int desc_set_rows(const pcreg_desc_set* cs, const double** out) {
    pcreg_desc_set* s = const_cast<pcreg_desc_set*>(cs);
    if (!s->rows) {
        double* r = nullptr;
        PCREG_HIP(hipMalloc((void**)&r, sizeof(double) * (size_t)(s->n > 0 ? s->n : 1) * s->D));
        const int rc = s->n > 0 ? launch_transpose_rows(s->d, s->n, s->n, s->D, r, g_stream) : PCREG_OK;
        if (rc) { (void)hipFree(r); return rc; }
        s->rows = r;
    }
    *out = s->rows;
    return PCREG_OK;
}
// the set as the segmented matcher's prepared model for these options (caller holds the library lock)
static int desc_set_prepared(const pcreg_desc_set* cs, const pcreg_match_opts& o, SegPreparedModel* out) {
    pcreg_desc_set* s = const_cast<pcreg_desc_set*>(cs);
    const double* rows;
    TRY(desc_set_rows(cs, &rows));
    const bool have = s->prep && s->prep_cm == o.change_metric && (!o.change_metric || s->prep_factor == o.metric_factor);
    if (!have) {
        if (!s->prep) PCREG_HIP(hipMalloc((void**)&s->prep, segmented_prepared_model_bytes(s->n, s->D)));
        s->prep_cm = -1;                                         // not valid until the launch below has been enqueued
        TRY(launch_segmented_prepare_model(rows, s->n, s->D, o, s->prep, s->prep + (size_t)(s->n > 0 ? s->n : 1) * s->D, g_stream));
        s->prep_cm = o.change_metric; s->prep_factor = o.metric_factor;
    }
    *out = SegPreparedModel{s->prep, s->prep + (size_t)(s->n > 0 ? s->n : 1) * s->D, s->n, s->D, s->prep_cm, s->prep_factor};
    return PCREG_OK;
}
int pcreg_get_matches_segmented_on_sets(const pcreg_desc_set* surface, const pcreg_desc_set* model, const int32_t* seg_rows,
                                        const int32_t* seg_off, int S, const pcreg_match_opts* par, uint32_t* pairs_all, int32_t* n_pairs) {
    PCREG_ARG(surface && model && seg_off && par && pairs_all && n_pairs && S >= 0 && surface->D == model->D);
    PCREG_ARG(S <= 65535);
    if (par->metric != PCREG_METRIC_SAD) { set_error("pcreg_get_matches_segmented_on_sets: Metric must be SAD (call pcreg_get_matches_on_sets per segment for SSD)"); return PCREG_E_ARG; }
    GUARD();
    if (S == 0) return PCREG_OK;
    const int Q = surface->n, VM = model->n, D = surface->D;
    PCREG_ARG(seg_off[0] == 0);
    int n_max = 0;
    for (int z = 0; z < S; ++z) { const int n = seg_off[z + 1] - seg_off[z]; PCREG_ARG(n >= 0); if (n > n_max) n_max = n; }
    const int tot = seg_off[S];
    PCREG_ARG(tot == 0 || seg_rows);
    for (int k = 0; k < tot; ++k) PCREG_ARG(seg_rows[k] >= 0 && seg_rows[k] < VM);
    if (Q == 0 || VM == 0 || tot == 0) { for (int z = 0; z < S; ++z) n_pairs[z] = 0; return PCREG_OK; }
    const double *rS, *rM;
    TRY(desc_set_rows(surface, &rS));
    TRY(desc_set_rows(model, &rM));
    const size_t q = (size_t)Q;
    void *dr, *doff, *dp, *dn, *ws;
    TRY(scratch().get(4, sizeof(int32_t) * (size_t)tot, &dr));
    TRY(scratch().get(5, sizeof(int32_t) * ((size_t)S + 1), &doff));
    TRY(scratch().get(6, sizeof(uint32_t) * (size_t)S * q * 2, &dp));
    TRY(scratch().get(7, sizeof(int32_t) * (size_t)S, &dn));
    const size_t wsb = get_matches_segmented_workspace_bytes(Q, VM, D, S, tot, n_max);
    TRY(scratch().get(8, wsb, &ws));
    PCREG_HIP(hipMemcpyAsync(dr, seg_rows, sizeof(int32_t) * (size_t)tot, hipMemcpyHostToDevice, g_stream));
    PCREG_HIP(hipMemcpyAsync(doff, seg_off, sizeof(int32_t) * ((size_t)S + 1), hipMemcpyHostToDevice, g_stream));
    SegPreparedModel prep;
    TRY(desc_set_prepared(model, *par, &prep));
    TRY(launch_get_matches_segmented(rS, Q, rM, VM, D, (const int32_t*)dr, (const int32_t*)doff, S, tot, n_max, *par,
                                     (uint32_t*)dp, nullptr, (int32_t*)dn, ws, wsb, g_stream, &prep));
    PairFetch pf;
    TRY(pairs_fetch_begin((const int32_t*)dn, S, pf));
    TRY(pairs_fetch_enqueue(pf, (const uint32_t*)dp, q, S, 20, pairs_all));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    pairs_fetch_finish(pf, q, S, pairs_all, n_pairs);
    return PCREG_OK;
}

// n x 3 column-major host doubles -> [n][3] on the device (what the sphere kernels and the gather read)
static int upload_points_aos(const double* host, int n, int ld, double* cols_tmp, double* aos, hipStream_t st) {
    if (n <= 0) return PCREG_OK;
    TRY(upload_cols(host, n, ld, 3, cols_tmp, st));
    return launch_transpose_rows(cols_tmp, n, n, 3, aos, st);
}
int pcreg_sphere_counts(const double* featModel, int VM, int ldM, const double* centres, int S, int ldC, double R, int32_t* counts) {
    PCREG_ARG(VM >= 0 && S >= 0 && ldM >= VM && ldC >= S && (VM == 0 || featModel) && (S == 0 || (centres && counts)));
    GUARD();
    if (S == 0) return PCREG_OK;
    if (VM == 0) { for (int i = 0; i < S; ++i) counts[i] = 0; return PCREG_OK; }
    void *tmp, *fm, *tc, *cen, *cnt;
    TRY(scratch().get(0, sizeof(double) * 3 * (size_t)VM, &tmp));
    TRY(scratch().get(1, sizeof(double) * 3 * (size_t)VM, &fm));
    TRY(scratch().get(2, sizeof(double) * 3 * (size_t)S, &tc));
    TRY(scratch().get(3, sizeof(double) * 3 * (size_t)S, &cen));
    TRY(scratch().get(4, sizeof(int32_t) * (size_t)S, &cnt));
    TRY(upload_points_aos(featModel, VM, ldM, (double*)tmp, (double*)fm, g_stream));
    TRY(upload_points_aos(centres, S, ldC, (double*)tc, (double*)cen, g_stream));
    TRY(launch_sphere_counts((const double*)fm, VM, (const double*)cen, S, R, (int32_t*)cnt, g_stream));
    PCREG_HIP(hipMemcpyAsync(counts, cnt, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    return PCREG_OK;
}
int pcreg_sphere_sweep(const pcreg_desc_set* surface, const pcreg_desc_set* model, const double* featSurface, int ldS, const double* featModel, int ldM,
                       const double* centres, int S, int ldC, const int32_t* num_desc, double R_desc, const pcreg_match_opts* par, int putative_thresh,
                       const pcreg_ransac_opts* coef, int32_t* model_rows, uint32_t* pairs_all, int32_t* n_pairs, int32_t* trial, int* n_trials,
                       double* T, int32_t* num_success, int32_t* max_inliers, int32_t* failed) {
    PCREG_ARG(surface && model && featSurface && featModel && par && coef && n_trials && S >= 0 && surface->D == model->D);
    PCREG_ARG(S == 0 || (centres && num_desc && model_rows && pairs_all && n_pairs && trial && T && num_success && max_inliers && failed));
    PCREG_ARG(S <= 65535 && ldC >= S && ldS >= surface->n && ldM >= model->n && coef->minPtNum == 3 && coef->iterNum >= 1);
    if (par->metric != PCREG_METRIC_SAD) { set_error("pcreg_sphere_sweep: Metric must be SAD (every driver of the reference uses it)"); return PCREG_E_ARG; }
    GUARD();
    *n_trials = 0;
    if (S == 0) return PCREG_OK;
    const int VS = surface->n, VM = model->n, D = surface->D;
    std::vector<int32_t> off((size_t)S + 1, 0); std::vector<int64_t> roff((size_t)S, 0);
    int n_max = 0;
    for (int i = 0; i < S; ++i) {
        PCREG_ARG(num_desc[i] >= 0 && num_desc[i] <= VM && (long long)off[i] + num_desc[i] < 2147483647LL);
        off[i + 1] = off[i] + num_desc[i]; roff[i] = off[i]; n_max = std::max(n_max, num_desc[i]);
    }
    const int tot = off[S];
    if (VS == 0 || VM == 0 || tot == 0) { for (int i = 0; i < S; ++i) n_pairs[i] = 0; return PCREG_OK; }
    const size_t vs = (size_t)VS, ld = (size_t)S * vs;
    PCREG_ARG(ld <= 0x7FFFFFFFull);                              // the packed correspondences are indexed with int
    void *tmp, *fm, *fs, *tc, *cen, *doff, *droff, *rows, *fall, *nsel, *dp, *dn, *ws, *tidx, *toff, *nt, *p12, *res, *inl, *rws;
    TRY(scratch().get(0, sizeof(double) * 3 * (size_t)std::max(VM, VS), &tmp));
    TRY(scratch().get(1, sizeof(double) * 3 * (size_t)VM, &fm));
    TRY(scratch().get(2, sizeof(double) * 3 * vs, &fs));
    TRY(scratch().get(3, sizeof(double) * 3 * (size_t)S, &tc));
    TRY(scratch().get(9, sizeof(double) * 3 * (size_t)S, &cen));
    TRY(scratch().get(4, sizeof(int32_t) * (size_t)tot, &rows));
    TRY(scratch().get(5, sizeof(int32_t) * ((size_t)S + 1), &doff));
    TRY(scratch().get(6, sizeof(uint32_t) * (size_t)S * vs * 2, &dp));
    TRY(scratch().get(7, sizeof(int32_t) * (size_t)S, &dn));
    const size_t wsb = get_matches_segmented_workspace_bytes(VS, VM, D, S, tot, n_max);
    TRY(scratch().get(8, wsb, &ws));
    TRY(scratch().get(10, sizeof(int64_t) * (size_t)S, &droff));
    TRY(scratch().get(11, sizeof(double) * 3 * (size_t)tot, &fall));
    TRY(scratch().get(12, sizeof(int32_t) * (size_t)S, &nsel));
    TRY(scratch().get(13, sizeof(int32_t) * (3 * (size_t)S + 2), &tidx));          // trial_idx [S] | offsets [S + 1] | n_trials
    toff = (int32_t*)tidx + S; nt = (int32_t*)tidx + 2 * (size_t)S + 1;
    TRY(scratch().get(14, sizeof(double) * 6 * ld, &p12));
    TRY(scratch().get(15, sizeof(pcreg_dev_ransac_result) * (size_t)S, &res));
    TRY(scratch().get(16, sizeof(int32_t) * ld, &inl));
    const size_t rwsb = ransac_workspace_bytes(coef->iterNum, S, VS);
    TRY(scratch().get(17, rwsb, &rws));
    const double *rS, *rM;
    TRY(desc_set_rows(surface, &rS));
    TRY(desc_set_rows(model, &rM));
    // :52-125: the keypoints, the spheres' row lists and their keypoints back to back
    TRY(upload_points_aos(featModel, VM, ldM, (double*)tmp, (double*)fm, g_stream));
    TRY(upload_points_aos(featSurface, VS, ldS, (double*)tmp, (double*)fs, g_stream));
    TRY(upload_points_aos(centres, S, ldC, (double*)tc, (double*)cen, g_stream));
    PCREG_HIP(hipMemcpyAsync(doff, off.data(), sizeof(int32_t) * ((size_t)S + 1), hipMemcpyHostToDevice, g_stream));
    PCREG_HIP(hipMemcpyAsync(droff, roff.data(), sizeof(int64_t) * (size_t)S, hipMemcpyHostToDevice, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));                   // off / roff are locals: the copies must have read them before anything can fail and return
    TRY(launch_sphere_select_batched((const double*)fm, VM, (const double*)cen, S, R_desc, (const int32_t*)doff, (int32_t*)rows, (double*)fall, (int32_t*)nsel, g_stream));
    // :131-149: getMatches of the surface against every sphere's rows
    SegPreparedModel prep;
    TRY(desc_set_prepared(model, *par, &prep));
    TRY(launch_get_matches_segmented(rS, VS, rM, VM, D, (const int32_t*)rows, (const int32_t*)doff, S, tot, n_max, *par, (uint32_t*)dp, nullptr, (int32_t*)dn,
                                     ws, wsb, g_stream, &prep));
    // :166-216: the putative threshold, the trial spheres' correspondences, one batched ransac (registration t: seed + t)
    TRY(launch_sweep_plan((const int32_t*)dn, S, putative_thresh, (int32_t*)tidx, (int32_t*)toff, (int32_t*)nt, g_stream));
    double *p1 = (double*)p12, *p2 = (double*)p12 + 3 * ld;
    PCREG_HIP(hipMemsetAsync(p12, 0, sizeof(double) * 6 * ld, g_stream));
    TRY(launch_sweep_gather((const uint32_t*)dp, VS, (const int32_t*)dn, (const int32_t*)tidx, (const int32_t*)toff, (const int32_t*)nt, S, (const double*)fs,
                            (const double*)fall, (const int64_t*)droff, p1, p2, (int)ld, g_stream));
    PairFetch pf;
    TRY(pairs_fetch_begin((const int32_t*)dn, S, pf));          // the counts leave in front of the RANSAC launch; the lists are packed while it runs
    PCREG_HIP(hipMemsetAsync(res, 0, sizeof(pcreg_dev_ransac_result) * (size_t)S, g_stream));
    TRY(launch_ransac(p1, p2, (int)ld, (const int32_t*)toff, nullptr, VS, S, *coef, nullptr, (pcreg_dev_ransac_result*)res, (int32_t*)inl, nullptr, nullptr,
                      rws, rwsb, g_stream));
    std::vector<int32_t> h_nsel((size_t)S), h_tr((size_t)S);
    std::vector<pcreg_dev_ransac_result> h_res((size_t)S);
    int32_t h_nt = 0;
    TRY(pairs_fetch_enqueue(pf, (const uint32_t*)dp, vs, S, 20, pairs_all));
    PCREG_HIP(hipMemcpyAsync(h_nsel.data(), nsel, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipMemcpyAsync(model_rows, rows, sizeof(int32_t) * (size_t)tot, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipMemcpyAsync(h_tr.data(), tidx, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipMemcpyAsync(&h_nt, nt, sizeof(int32_t), hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipMemcpyAsync(h_res.data(), res, sizeof(pcreg_dev_ransac_result) * (size_t)S, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    pairs_fetch_finish(pf, vs, S, pairs_all, n_pairs);
    for (int i = 0; i < S; ++i)
        if (h_nsel[i] != num_desc[i]) { set_error("pcreg_sphere_sweep: num_desc[%d] = %d, but the sphere holds %d keypoints (pass pcreg_sphere_counts' values)", i, num_desc[i], h_nsel[i]); return PCREG_E_ARG; }
    *n_trials = h_nt;
    for (int t = 0; t < h_nt; ++t) {
        trial[t] = h_tr[t];
        const pcreg_dev_ransac_result& r = h_res[t];
        for (int k = 0; k < 16; ++k) T[(size_t)t * 16 + k] = r.failed ? 0.0 : r.T[k];
        num_success[t] = r.num_success; max_inliers[t] = r.max_inliers; failed[t] = r.failed;
    }
    return PCREG_OK;
}

// ---- the sphere sweep's model side as a handle: one model, many surfaces ----------------------------------------------------
// Everything of pcreg_sphere_sweep that does not depend on the surface: the spheres' row lists and gathered keypoints, the model
// set restricted to the union of those rows (the lists renumbered into it) and, per set of getMatches options, its powered rows.
struct pcreg_sphere_model {
    int S, VMu, D, tot, n_max;
    int32_t *seg_off, *rows_u; int64_t* roff; double *feat_all, *desc_u, *prep;
    int prep_cm; double prep_factor;
};
static void sphere_model_free(pcreg_sphere_model* m) {
    if (!m) return;
    void* p[] = {m->seg_off, m->rows_u, m->roff, m->feat_all, m->desc_u, m->prep};
    for (void* q : p) if (q) (void)hipFree(q);
    delete m;
}
int pcreg_sphere_model_create(const pcreg_desc_set* model, const double* featModel, int ldM, const double* centres, int S, int ldC,
                              const int32_t* num_desc, double R_desc, int32_t* model_rows, pcreg_sphere_model** out) {
    PCREG_ARG(model && featModel && out && S >= 0 && S <= 65535 && ldC >= S && ldM >= model->n && (S == 0 || (centres && num_desc && model_rows)));
    GUARD();
    *out = nullptr;
    const int VM = model->n, D = model->D;
    std::vector<int32_t> off((size_t)S + 1, 0); std::vector<int64_t> roff((size_t)S, 0);
    int n_max = 0;
    for (int i = 0; i < S; ++i) {
        PCREG_ARG(num_desc[i] >= 0 && num_desc[i] <= VM && (long long)off[i] + num_desc[i] < 2147483647LL);
        off[i + 1] = off[i] + num_desc[i]; roff[i] = off[i]; n_max = std::max(n_max, num_desc[i]);
    }
    const int tot = off[S];
    pcreg_sphere_model* m = new pcreg_sphere_model{S, 0, D, tot, n_max, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, -1, 0.0};
    auto fail = [&](int rc) { sphere_model_free(m); return rc; };
    if (S == 0 || tot == 0 || VM == 0) { *out = m; return PCREG_OK; }
    void *tmp, *fm, *tc, *cen, *nsel, *rows, *nd;
    int rc = PCREG_OK;
    if ((rc = scratch().get(0, sizeof(double) * 3 * (size_t)VM, &tmp)) || (rc = scratch().get(1, sizeof(double) * 3 * (size_t)VM, &fm)) ||
        (rc = scratch().get(3, sizeof(double) * 3 * (size_t)S, &tc)) || (rc = scratch().get(9, sizeof(double) * 3 * (size_t)S, &cen)) ||
        (rc = scratch().get(12, sizeof(int32_t) * (size_t)S, &nsel)) || (rc = scratch().get(4, sizeof(int32_t) * (size_t)tot, &rows)) ||
        (rc = scratch().get(5, 256, &nd)))
        return fail(rc);
    if (hipMalloc((void**)&m->seg_off, sizeof(int32_t) * ((size_t)S + 1)) != hipSuccess || hipMalloc((void**)&m->roff, sizeof(int64_t) * (size_t)S) != hipSuccess ||
        hipMalloc((void**)&m->rows_u, sizeof(int32_t) * (size_t)tot) != hipSuccess || hipMalloc((void**)&m->feat_all, sizeof(double) * 3 * (size_t)tot) != hipSuccess) {
        set_error("pcreg_sphere_model_create: out of device memory"); return fail(PCREG_E_HIP);
    }
    if ((rc = upload_points_aos(featModel, VM, ldM, (double*)tmp, (double*)fm, g_stream)) || (rc = upload_points_aos(centres, S, ldC, (double*)tc, (double*)cen, g_stream)))
        return fail(rc);
    if (hipMemcpyAsync(m->seg_off, off.data(), sizeof(int32_t) * ((size_t)S + 1), hipMemcpyHostToDevice, g_stream) != hipSuccess ||
        hipMemcpyAsync(m->roff, roff.data(), sizeof(int64_t) * (size_t)S, hipMemcpyHostToDevice, g_stream) != hipSuccess) { set_error("copy failed"); return fail(PCREG_E_HIP); }
    if ((rc = launch_sphere_select_batched((const double*)fm, VM, (const double*)cen, S, R_desc, m->seg_off, (int32_t*)rows, m->feat_all, (int32_t*)nsel, g_stream)))
        return fail(rc);
    std::vector<int32_t> h_nsel((size_t)S);
    if (hipMemcpyAsync(h_nsel.data(), nsel, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, g_stream) != hipSuccess ||
        hipMemcpyAsync(model_rows, rows, sizeof(int32_t) * (size_t)tot, hipMemcpyDeviceToHost, g_stream) != hipSuccess ||
        hipStreamSynchronize(g_stream) != hipSuccess) { set_error("pcreg_sphere_model_create: read-back failed"); return fail(PCREG_E_HIP); }
    for (int i = 0; i < S; ++i)
        if (h_nsel[i] != num_desc[i]) { set_error("pcreg_sphere_model_create: num_desc[%d] = %d, but the sphere holds %d keypoints (pass pcreg_sphere_counts' values)", i, num_desc[i], h_nsel[i]); return fail(PCREG_E_ARG); }
    // the union of the spheres' rows (ascending) and the lists renumbered into it
    std::vector<int32_t> uni(model_rows, model_rows + tot);
    std::sort(uni.begin(), uni.end());
    uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
    std::vector<int32_t> ren((size_t)tot);
    for (int k = 0; k < tot; ++k) ren[k] = (int32_t)(std::lower_bound(uni.begin(), uni.end(), model_rows[k]) - uni.begin());
    m->VMu = (int)uni.size();
    const double* rM;
    if ((rc = desc_set_rows(model, &rM))) return fail(rc);
    void* duni;
    if ((rc = scratch().get(6, sizeof(int32_t) * uni.size(), &duni))) return fail(rc);
    if (hipMalloc((void**)&m->desc_u, sizeof(double) * uni.size() * (size_t)D) != hipSuccess) { set_error("pcreg_sphere_model_create: out of device memory"); return fail(PCREG_E_HIP); }
    const int32_t nu = m->VMu;
    if (hipMemcpyAsync(duni, uni.data(), sizeof(int32_t) * uni.size(), hipMemcpyHostToDevice, g_stream) != hipSuccess ||
        hipMemcpyAsync(nd, &nu, sizeof(int32_t), hipMemcpyHostToDevice, g_stream) != hipSuccess ||
        hipMemcpyAsync(m->rows_u, ren.data(), sizeof(int32_t) * (size_t)tot, hipMemcpyHostToDevice, g_stream) != hipSuccess) { set_error("copy failed"); return fail(PCREG_E_HIP); }
    if ((rc = launch_gather_rows_f64(rM, D, (const int32_t*)duni, (const int32_t*)nd, m->VMu, m->desc_u, g_stream))) return fail(rc);
    if (hipStreamSynchronize(g_stream) != hipSuccess) { set_error("pcreg_sphere_model_create failed"); return fail(PCREG_E_HIP); }          // the host vectors go out of scope
    *out = m;
    return PCREG_OK;
}
int pcreg_sphere_model_destroy(pcreg_sphere_model* m) {
    if (!m) return PCREG_OK;
    std::lock_guard<std::mutex> lock(g_mu);
    (void)hipDeviceSynchronize();
    sphere_model_free(m);
    return PCREG_OK;
}
int pcreg_sphere_sweep_on_model(pcreg_sphere_model* m, const pcreg_desc_set* surface, const double* featSurface, int ldS, const pcreg_match_opts* par,
                                int putative_thresh, const pcreg_ransac_opts* coef, uint32_t* pairs_all, int32_t* n_pairs, int32_t* trial, int* n_trials,
                                double* T, int32_t* num_success, int32_t* max_inliers, int32_t* failed) {
    PCREG_ARG(m && surface && featSurface && par && coef && n_trials && surface->D == m->D && ldS >= surface->n && coef->minPtNum == 3 && coef->iterNum >= 1);
    PCREG_ARG(m->S == 0 || (pairs_all && n_pairs && trial && T && num_success && max_inliers && failed));
    if (par->metric != PCREG_METRIC_SAD) { set_error("pcreg_sphere_sweep_on_model: Metric must be SAD"); return PCREG_E_ARG; }
    GUARD();
    *n_trials = 0;
    const int S = m->S, VS = surface->n, D = m->D, tot = m->tot;
    if (S == 0) return PCREG_OK;
    if (VS == 0 || tot == 0 || m->VMu == 0) { for (int i = 0; i < S; ++i) n_pairs[i] = 0; return PCREG_OK; }
    const size_t vs = (size_t)VS, ld = (size_t)S * vs;
    PCREG_ARG(ld <= 0x7FFFFFFFull);
    void *tmp, *fs, *dp, *dn, *ws, *tidx, *toff, *nt, *p12, *res, *inl, *rws;
    TRY(scratch().get(0, sizeof(double) * 3 * vs, &tmp));
    TRY(scratch().get(2, sizeof(double) * 3 * vs, &fs));
    TRY(scratch().get(6, sizeof(uint32_t) * (size_t)S * vs * 2, &dp));
    TRY(scratch().get(7, sizeof(int32_t) * (size_t)S, &dn));
    const size_t wsb = get_matches_segmented_workspace_bytes(VS, m->VMu, D, S, tot, m->n_max);
    TRY(scratch().get(8, wsb, &ws));
    TRY(scratch().get(13, sizeof(int32_t) * (3 * (size_t)S + 2), &tidx));
    toff = (int32_t*)tidx + S; nt = (int32_t*)tidx + 2 * (size_t)S + 1;
    TRY(scratch().get(14, sizeof(double) * 6 * ld, &p12));
    TRY(scratch().get(15, sizeof(pcreg_dev_ransac_result) * (size_t)S, &res));
    TRY(scratch().get(16, sizeof(int32_t) * ld, &inl));
    const size_t rwsb = ransac_workspace_bytes(coef->iterNum, S, VS);
    TRY(scratch().get(17, rwsb, &rws));
    const double* rS;
    TRY(desc_set_rows(surface, &rS));
    if (!(m->prep && m->prep_cm == par->change_metric && (!par->change_metric || m->prep_factor == par->metric_factor))) {
        if (!m->prep) PCREG_HIP(hipMalloc((void**)&m->prep, segmented_prepared_model_bytes(m->VMu, D)));
        m->prep_cm = -1;
        TRY(launch_segmented_prepare_model(m->desc_u, m->VMu, D, *par, m->prep, m->prep + (size_t)m->VMu * D, g_stream));
        m->prep_cm = par->change_metric; m->prep_factor = par->metric_factor;
    }
    const SegPreparedModel prep{m->prep, m->prep + (size_t)m->VMu * D, m->VMu, D, m->prep_cm, m->prep_factor};
    TRY(upload_points_aos(featSurface, VS, ldS, (double*)tmp, (double*)fs, g_stream));
    TRY(launch_get_matches_segmented(rS, VS, m->desc_u, m->VMu, D, m->rows_u, m->seg_off, S, tot, m->n_max, *par, (uint32_t*)dp, nullptr, (int32_t*)dn,
                                     ws, wsb, g_stream, &prep));
    TRY(launch_sweep_plan((const int32_t*)dn, S, putative_thresh, (int32_t*)tidx, (int32_t*)toff, (int32_t*)nt, g_stream));
    double *p1 = (double*)p12, *p2 = (double*)p12 + 3 * ld;
    PCREG_HIP(hipMemsetAsync(p12, 0, sizeof(double) * 6 * ld, g_stream));
    TRY(launch_sweep_gather((const uint32_t*)dp, VS, (const int32_t*)dn, (const int32_t*)tidx, (const int32_t*)toff, (const int32_t*)nt, S, (const double*)fs,
                            m->feat_all, m->roff, p1, p2, (int)ld, g_stream));
    PairFetch pf;
    TRY(pairs_fetch_begin((const int32_t*)dn, S, pf));          // the counts leave in front of the RANSAC launch; the lists are packed while it runs
    PCREG_HIP(hipMemsetAsync(res, 0, sizeof(pcreg_dev_ransac_result) * (size_t)S, g_stream));
    TRY(launch_ransac(p1, p2, (int)ld, (const int32_t*)toff, nullptr, VS, S, *coef, nullptr, (pcreg_dev_ransac_result*)res, (int32_t*)inl, nullptr, nullptr,
                      rws, rwsb, g_stream));
    std::vector<int32_t> h_tr((size_t)S);
    std::vector<pcreg_dev_ransac_result> h_res((size_t)S);
    int32_t h_nt = 0;
    TRY(pairs_fetch_enqueue(pf, (const uint32_t*)dp, vs, S, 20, pairs_all));
    PCREG_HIP(hipMemcpyAsync(h_tr.data(), tidx, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipMemcpyAsync(&h_nt, nt, sizeof(int32_t), hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipMemcpyAsync(h_res.data(), res, sizeof(pcreg_dev_ransac_result) * (size_t)S, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    pairs_fetch_finish(pf, vs, S, pairs_all, n_pairs);
    *n_trials = h_nt;
    for (int t = 0; t < h_nt; ++t) {
        trial[t] = h_tr[t];
        const pcreg_dev_ransac_result& r = h_res[t];
        for (int k = 0; k < 16; ++k) T[(size_t)t * 16 + k] = r.failed ? 0.0 : r.T[k];
        num_success[t] = r.num_success; max_inliers[t] = r.max_inliers; failed[t] = r.failed;
    }
    return PCREG_OK;
}

int pcreg_align_points_knn_batched(const double* pts, int total, int ld, const int32_t* offsets, int B, int C1, int C2,
                                   double* aligned, double* coeff, double* c, int32_t* status) {
    PCREG_ARG(pts && offsets && aligned && coeff && c && status && total >= 0 && ld >= total && B >= 0);
    GUARD();
    if (B == 0) return PCREG_OK;
    int max_n = 0;
    for (int b = 0; b < B; ++b) { int nb = offsets[b + 1] - offsets[b]; PCREG_ARG(nb >= 0); if (nb > max_n) max_n = nb; }
    PCREG_ARG(offsets[0] == 0 && offsets[B] == total);
    size_t tot = (size_t)(total > 0 ? total : 1);
    void *dp, *da, *doff, *dco, *dc, *dst;
    TRY(scratch().get(0, sizeof(double) * 3 * tot, &dp));
    TRY(scratch().get(1, sizeof(double) * 3 * tot, &da));
    TRY(scratch().get(2, sizeof(int32_t) * ((size_t)B + 1), &doff));
    TRY(scratch().get(3, sizeof(double) * 9 * (size_t)B, &dco));
    TRY(scratch().get(4, sizeof(double) * 3 * (size_t)B, &dc));
    TRY(scratch().get(5, sizeof(int32_t) * (size_t)B, &dst));
    TRY(upload_cols(pts, total, ld, 3, (double*)dp, g_stream));
    PCREG_HIP(hipMemcpyAsync(doff, offsets, sizeof(int32_t) * ((size_t)B + 1), hipMemcpyHostToDevice, g_stream));
    PCREG_HIP(hipMemsetAsync(dco, 0, sizeof(double) * 9 * (size_t)B, g_stream));
    PCREG_HIP(hipMemsetAsync(dc, 0, sizeof(double) * 3 * (size_t)B, g_stream));
    PCREG_HIP(hipMemsetAsync(da, 0, sizeof(double) * 3 * tot, g_stream));
    TRY(launch_align_points_knn((double*)dp, total, (int32_t*)doff, B, max_n, C1, C2, (double*)da, total, (double*)dco,
                                (double*)dc, (int32_t*)dst, g_stream));
    if (total > 0) PCREG_HIP(hipMemcpyAsync(aligned, da, sizeof(double) * 3 * (size_t)total, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipMemcpyAsync(coeff, dco, sizeof(double) * 9 * (size_t)B, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipMemcpyAsync(c, dc, sizeof(double) * 3 * (size_t)B, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipMemcpyAsync(status, dst, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    return PCREG_OK;
}

int pcreg_dev_align_points_knn_batched(const double* pts, int total, int ld, const int32_t* offsets, int B, int max_n,
                                       int C1, int C2, double* aligned, double* coeff, double* c, int32_t* status,
                                       void* stream) {
    PCREG_ARG(pts && offsets && aligned && coeff && c && status && total >= 0 && ld >= total && B >= 0 && max_n >= 0);
    GUARD();
    if (B == 0) return PCREG_OK;
    return launch_align_points_knn(pts, ld, offsets, B, max_n, C1, C2, aligned, ld, coeff, c, status, (hipStream_t)stream);
}

int pcreg_align_points_knn(const double* pts, int n, int ld, int C1, int C2, double* aligned, double coeff[9],
                           double c[3]) {
    PCREG_ARG(pts && aligned && coeff && c && n >= 2 && ld >= n);
    int32_t offsets[2] = {0, n};
    int32_t status = 0;
    // compact the input so that "total == ld" holds for the batched entry point
    if (ld != n) {
        std::vector<double> tmp((size_t)n * 3);
        for (int k = 0; k < 3; ++k) memcpy(tmp.data() + (size_t)k * n, pts + (size_t)k * ld, sizeof(double) * (size_t)n);
        TRY(pcreg_align_points_knn_batched(tmp.data(), n, n, offsets, 1, C1, C2, aligned, coeff, c, &status));
    } else {
        TRY(pcreg_align_points_knn_batched(pts, n, n, offsets, 1, C1, C2, aligned, coeff, c, &status));
    }
    if (status != 0) { set_error("AlignPoints_KNN: support too small"); return PCREG_E_ARG; }
    return PCREG_OK;
}

// pts / sample_pts: double arrays (single data widened exactly by the caller); single_mode: see launch_descriptors
static int descriptors_host(const double* pts, int P, int ld, const double* sample_pts, int S, int lds, const pcreg_desc_opts* options,
                            int single_mode, double* feat, double* desc, int* V) {
    *V = 0;
    if (P == 0 || S == 0) return PCREG_OK;
    void *dp, *dk, *dfeat, *ddesc, *dcnt, *ws;
    size_t wsb = descriptors_workspace_bytes(P, S);
    TRY(scratch().get(0, sizeof(double) * 3 * (size_t)P, &dp));
    TRY(scratch().get(1, sizeof(double) * 3 * (size_t)S, &dk));
    TRY(scratch().get(2, sizeof(double) * 3 * (size_t)S, &dfeat));
    TRY(scratch().get(3, sizeof(double) * PCREG_DESC_LEN * (size_t)S, &ddesc));
    TRY(scratch().get(4, 256, &dcnt));
    TRY(scratch().get(5, wsb, &ws));
    int32_t* dV = (int32_t*)dcnt; int32_t* dErr = dV + 1;
    TRY(upload_cols(pts, P, ld, 3, (double*)dp, g_stream));
    TRY(upload_cols(sample_pts, S, lds, 3, (double*)dk, g_stream));
    TRY(launch_descriptors((double*)dp, P, P, (double*)dk, S, S, *options, single_mode, (double*)dfeat, (double*)ddesc, nullptr, nullptr, dV, dErr,
                           ws, wsb, g_stream));
    int32_t hv[2] = {0, 0};
    PCREG_HIP(hipMemcpyAsync(hv, dcnt, sizeof hv, hipMemcpyDeviceToHost, g_stream));
    PCREG_HIP(hipStreamSynchronize(g_stream));
    if (hv[1] != 0) { set_error("a support holds %d points: more than the %d an LDS-resident support may have (lower max_pts)", hv[1], 8191); return PCREG_E_ARG; }
    if (hv[0] > 0) {
        PCREG_HIP(hipMemcpy(feat, dfeat, sizeof(double) * 3 * (size_t)hv[0], hipMemcpyDeviceToHost));
        PCREG_HIP(hipMemcpy(desc, ddesc, sizeof(double) * PCREG_DESC_LEN * (size_t)hv[0], hipMemcpyDeviceToHost));
    }
    *V = hv[0];
    return PCREG_OK;
}

int pcreg_spatial_histogram_descriptors(const double* pts, int P, int ld, const double* sample_pts, int S, int lds,
                                        const pcreg_desc_opts* options, double* feat, double* desc, int* V) {
    PCREG_ARG(pts && sample_pts && options && feat && desc && V && P >= 0 && S >= 0 && ld >= P && lds >= S);
    GUARD();
    return descriptors_host(pts, P, ld, sample_pts, S, lds, options, 0, feat, desc, V);
}

// ---- `single` inputs (clouds read by pcread are single: upsampleMesh.m:21, GetPointcloudFromModel.m:269) ----------------
// AlignPoints_KNN keeps its class (AlignPoints_KNN.m:17,59: everything derives from pts): the float -> double widening is
// exact, the arithmetic is the double kernel's, the outputs are rounded once to float.
int pcreg_align_points_knn_f32(const float* pts, int n, int ld, int C1, int C2, float* aligned, float coeff[9], float c[3]) {
    PCREG_ARG(pts && aligned && coeff && c && n >= 2 && ld >= n);
    std::vector<double> in((size_t)n * 3), out((size_t)n * 3);
    for (int k = 0; k < 3; ++k) for (int i = 0; i < n; ++i) in[(size_t)k * n + i] = (double)pts[(size_t)k * ld + i];
    double co[9], cc[3];
    TRY(pcreg_align_points_knn(in.data(), n, n, C1, C2, out.data(), co, cc));
    for (size_t i = 0; i < (size_t)n * 3; ++i) aligned[i] = (float)out[i];
    for (int i = 0; i < 9; ++i) coeff[i] = (float)co[i];
    for (int i = 0; i < 3; ++i) c[i] = (float)cc[i];
    return PCREG_OK;
}

// getSpacialHistogramDescriptors with a `single` cloud and / or `single` keypoints.  The OUTPUTS are double whatever the
// inputs (getSpacialHistogramDescriptors.m:61-62 preallocates desc / feat with nan(...) and assigns into them).  What runs in
// single inside MATLAB and is element-wise -- hence reproducible -- is reproduced: getLocalPoints.m:8-31's open box test,
// pts_cube - c, sqrt(x^2 + y^2 + z^2), dists < R and with them WHICH keypoints survive and which points form a support;
// the support's coordinates are MATLAB's single pts_rel values.  mean / pca / the histogram then run in double on those
// values (the summation order of MATLAB's single mean and pca is not knowable: INTEGRATION.md).
int pcreg_spatial_histogram_descriptors_mixed(const void* pts, int pts_is_single, int P, int ld, const void* sample_pts,
                                              int sample_is_single, int S, int lds, const pcreg_desc_opts* options,
                                              double* feat, double* desc, int* V) {
    PCREG_ARG(pts && sample_pts && options && feat && desc && V && P >= 0 && S >= 0 && ld >= P && lds >= S);
    GUARD();
    std::vector<double> p, k;
    const double* pp = (const double*)pts; const double* kk = (const double*)sample_pts;
    int lp = ld, lk = lds;
    if (pts_is_single) {
        p.resize((size_t)P * 3);
        for (int c = 0; c < 3; ++c) for (int i = 0; i < P; ++i) p[(size_t)c * P + i] = (double)((const float*)pts)[(size_t)c * ld + i];
        pp = p.data(); lp = P;
    }
    if (sample_is_single) {
        k.resize((size_t)S * 3);
        for (int c = 0; c < 3; ++c) for (int i = 0; i < S; ++i) k[(size_t)c * S + i] = (double)((const float*)sample_pts)[(size_t)c * lds + i];
        kk = k.data(); lk = S;
    }
    const int mode = sample_is_single ? 1 : (pts_is_single ? 2 : 0);
    return descriptors_host(pp, P, lp, kk, S, lk, options, mode, feat, desc, V);
}
int pcreg_spatial_histogram_descriptors_f32(const float* pts, int P, int ld, const float* sample_pts, int S, int lds,
                                            const pcreg_desc_opts* options, double* feat, double* desc, int* V) {
    return pcreg_spatial_histogram_descriptors_mixed(pts, 1, P, ld, sample_pts, 1, S, lds, options, feat, desc, V);
}

// ------------------------------------------------------------------ device tier
size_t pcreg_dev_knn2_points_f32_workspace(int Q, int M) { return knn2_points_workspace_bytes(Q, M); }

int pcreg_dev_knn2_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm, int32_t idx_base,
                              int32_t* idx, float* dist, void* workspace, size_t workspace_bytes, void* stream) {
    PCREG_ARG(q && m && idx && dist && workspace);
    GUARD();
    return launch_knn2_points_f32(q, Q, ldq, m, M, ldm, idx_base, idx, dist, workspace, workspace_bytes, (hipStream_t)stream);
}

int pcreg_dev_merge_top2_f32(const int32_t* idx_in, const float* dist_in, int R, int Q, int32_t* idx, float* dist,
                             void* stream) {
    PCREG_ARG(idx_in && dist_in && idx && dist);
    GUARD();
    return launch_merge_top2_f32(idx_in, dist_in, R, Q, idx, dist, (hipStream_t)stream);
}

int pcreg_dev_merge_top2_strided_f32(const int32_t* idx_in, const float* dist_in, int R, int Q, size_t rank_stride,
                                     int32_t* idx, float* dist, void* stream) {
    PCREG_ARG(idx_in && dist_in && idx && dist);
    GUARD();
    return launch_merge_top2_f32(idx_in, dist_in, R, Q, idx, dist, (hipStream_t)stream, rank_stride);
}

size_t pcreg_dev_ransac_workspace(int n_cap, int iterNum) { return ransac_workspace_bytes(iterNum, 1, n_cap); }

int pcreg_dev_ransac(const double* pts1, const double* pts2, const int32_t* n_dev, int n_cap, int ld,
                     const pcreg_ransac_opts* opts, const int32_t* sample_idx, pcreg_dev_ransac_result* out,
                     int32_t* inlier_idx, void* workspace, size_t workspace_bytes, void* stream) {
    PCREG_ARG(pts1 && pts2 && opts && out && inlier_idx && workspace && n_cap >= 0 && ld >= n_cap);
    GUARD();
    return launch_ransac(pts1, pts2, ld, nullptr, n_dev, n_cap, 1, *opts, sample_idx, out, inlier_idx, nullptr, nullptr,
                         workspace, workspace_bytes, (hipStream_t)stream);
}


int pcreg_dev_ransac_partial(const double* pts1, const double* pts2, const int32_t* n_dev, int n_cap, int ld,
                             const pcreg_ransac_opts* opts, const int32_t* sample_idx, int hyp_begin, int hyp_count,
                             pcreg_dev_ransac_part* part, void* workspace, size_t workspace_bytes, void* stream) {
    PCREG_ARG(pts1 && pts2 && opts && part && workspace && n_cap >= 0 && ld >= n_cap);
    GUARD();
    return launch_ransac_partial(pts1, pts2, ld, n_dev, n_cap, *opts, sample_idx, hyp_begin, hyp_count, part, workspace,
                                 workspace_bytes, (hipStream_t)stream);
}
int pcreg_dev_ransac_finish(const double* pts1, const double* pts2, const int32_t* n_dev, int n_cap, int ld,
                            const pcreg_ransac_opts* opts, const pcreg_dev_ransac_part* combined,
                            pcreg_dev_ransac_result* out, int32_t* inlier_idx, void* stream) {
    PCREG_ARG(pts1 && pts2 && opts && combined && out && inlier_idx && n_cap >= 0 && ld >= n_cap);
    GUARD();
    return launch_ransac_finish(pts1, pts2, ld, n_dev, n_cap, *opts, combined, out, inlier_idx, (hipStream_t)stream);
}

int pcreg_dev_ransac_finish_parts(const double* pts1, const double* pts2, const int32_t* n_dev, int n_cap, int ld,
                                  const pcreg_ransac_opts* opts, const pcreg_dev_ransac_part* parts, int n_parts,
                                  pcreg_dev_ransac_result* out, int32_t* inlier_idx, void* stream) {
    PCREG_ARG(pts1 && pts2 && opts && parts && n_parts >= 1 && out && inlier_idx && n_cap >= 0 && ld >= n_cap);
    GUARD();
    return launch_ransac_finish(pts1, pts2, ld, n_dev, n_cap, *opts, parts, out, inlier_idx, (hipStream_t)stream, n_parts);
}

// live timing of the search's dominant kernel (HIP events on the launch stream), for bench.py's roofline
int pcreg_dev_search_kernel_timing(int enable) { GUARD(); knn_f16_timing_enable(enable != 0); return PCREG_OK; }
int pcreg_dev_search_kernel_ms(float* mean_ms, int* launches) {
    PCREG_ARG(mean_ms && launches);
    GUARD();
    return knn_f16_timing_read(mean_ms, launches);
}

// ---- descriptor stage, resident (speedyDescriptors.m:59 -> getMatches -> ransac without leaving HBM)
size_t pcreg_dev_spatial_histogram_descriptors_workspace(int P, int S) { return descriptors_workspace_bytes(P, S); }

int pcreg_dev_spatial_histogram_descriptors(const double* pts, int P, int ld, const double* sample_pts, int S, int lds,
                                            const pcreg_desc_opts* options, double* feat, double* desc, int32_t* counters,
                                            void* workspace, size_t workspace_bytes, void* stream) {
    PCREG_ARG(pts && sample_pts && options && feat && desc && counters && workspace && P >= 1 && S >= 1 && ld >= P && lds >= S);
    GUARD();
    return launch_descriptors(pts, P, ld, sample_pts, S, lds, *options, 0, feat, desc, nullptr, nullptr, counters, counters + 1, workspace,
                              workspace_bytes, (hipStream_t)stream);
}

int pcreg_dev_spatial_histogram_descriptors_rows_u16(const double* pts, int P, int ld, const double* sample_pts, int S, int lds,
                                                     const pcreg_desc_opts* options, int single_mode, double* feat, uint16_t* rows,
                                                     int32_t* row_index, int32_t* counters, void* workspace, size_t workspace_bytes,
                                                     void* stream) {
    PCREG_ARG(pts && sample_pts && options && feat && rows && row_index && counters && workspace && P >= 1 && S >= 1 && ld >= P && lds >= S);
    GUARD();
    return launch_descriptors(pts, P, ld, sample_pts, S, lds, *options, single_mode, feat, nullptr, rows, row_index, counters, counters + 1,
                              workspace, workspace_bytes, (hipStream_t)stream);
}

static constexpr int kLayoutRowMajorU16 = 1000;      // internal: dense uint16 rows (pcreg_dev_get_matches_rows_u16)
static size_t dev_get_matches_layout(int Q, int M, int D, int Dp, size_t off[6]) {
    size_t q = (size_t)(Q > 0 ? Q : 1), m = (size_t)(M > 0 ? M : 1), b = 0;
    off[0] = b; b += align_up(q * D * sizeof(double), 256);            // raw surface, feature-major
    off[1] = b; b += align_up(m * D * sizeof(double), 256);            // raw model
    off[2] = b; b += align_up(q * Dp * sizeof(double), 256);           // working copies
    off[3] = b; b += align_up(m * Dp * sizeof(double), 256);
    off[4] = b; b += align_up((q + m + 1) * sizeof(double), 256);      // preprocess
    off[5] = b; b += match_features_workspace_bytes(Q, M, Dp);
    return b;
}
size_t pcreg_dev_get_matches_workspace(int Q, int M, int D) {
    size_t off[6];
    return dev_get_matches_layout(Q, M, D, D + 1, off);
}

static int dev_get_matches_impl(const void* descSurface, int Q, int ldS, const void* descModel, int M, int ldM, int D,
                                int layout, const pcreg_match_opts* par, uint32_t* pairs, double* metric, int32_t* n_pairs,
                                void* workspace, size_t workspace_bytes, hipStream_t st, const int32_t* idxS = nullptr,
                                const int32_t* idxM = nullptr) {
    if (Q == 0 || M == 0) { PCREG_HIP(hipMemsetAsync(n_pairs, 0, sizeof(int32_t), st)); return PCREG_OK; }
    const int Dp = D + (par->unnormalize ? 1 : 0);
    size_t off[6];
    size_t need = dev_get_matches_layout(Q, M, D, D + 1, off);
    if (workspace_bytes < need) { set_error("get_matches workspace too small: %zu < %zu", workspace_bytes, need); return PCREG_E_WORKSPACE; }
    char* w = (char*)workspace;
    double *rawS = (double*)(w + off[0]), *rawM = (double*)(w + off[1]), *fS = (double*)(w + off[2]), *fM = (double*)(w + off[3]);
    const double *inS = (const double*)descSurface, *inM = (const double*)descModel;
    int ls = ldS, lm = ldM;
    if (layout == PCREG_LAYOUT_ROW_MAJOR) {        // [row][D] (ld = row pitch) -> feature-major
        if (ldS != D || ldM != D) { set_error("row-major descriptors must be dense (ld == D)"); return PCREG_E_ARG; }
        TRY(launch_transpose_rows((const double*)descSurface, D, D, Q, rawS, st));
        TRY(launch_transpose_rows((const double*)descModel, D, D, M, rawM, st));
        inS = rawS; inM = rawM; ls = Q; lm = M;
    } else if (layout == kLayoutRowMajorU16) {     // dense u16 rows (counts) -> feature-major doubles, exactly
        TRY(launch_widen_rows_u16((const uint16_t*)descSurface, idxS, Q, D, rawS, st));
        TRY(launch_widen_rows_u16((const uint16_t*)descModel, idxM, M, D, rawM, st));
        inS = rawS; inM = rawM; ls = Q; lm = M;
    }
    // getMatches.m:24-37 always works on private copies (the caller's descriptors stay untouched)
    TRY(launch_preprocess(inS, Q, ls, inM, M, lm, D, *par, fS, fM, w + off[4], align_up(((size_t)Q + M + 1) * sizeof(double), 256), st));
    if (!par->prenormalized) {
        TRY(launch_normalize_rows2(fS, Q, Q, fM, M, M, Dp, st));
    }
    return launch_match_features(fS, Q, Q, fM, M, M, Dp, *par, pairs, metric, n_pairs, w + off[5], workspace_bytes - off[5], st);
}

int pcreg_dev_get_matches(const double* descSurface, int Q, int ldS, const double* descModel, int M, int ldM, int D,
                          int layout, const pcreg_match_opts* par, uint32_t* pairs, double* metric, int32_t* n_pairs,
                          void* workspace, size_t workspace_bytes, void* stream) {
    PCREG_ARG(descSurface && descModel && par && pairs && n_pairs && workspace && Q >= 0 && M >= 0 && D >= 1);
    PCREG_ARG(layout == PCREG_LAYOUT_FEATURE_MAJOR || layout == PCREG_LAYOUT_ROW_MAJOR);
    PCREG_ARG(layout == PCREG_LAYOUT_ROW_MAJOR ? (ldS >= D && ldM >= D) : (ldS >= Q && ldM >= M));
    PCREG_ARG(par->metric == PCREG_METRIC_SAD || par->metric == PCREG_METRIC_SSD);
    GUARD();
    return dev_get_matches_impl(descSurface, Q, ldS, descModel, M, ldM, D, layout, par, pairs, metric, n_pairs, workspace,
                                workspace_bytes, (hipStream_t)stream);
}

int pcreg_dev_get_matches_rows_u16(const uint16_t* rowsSurface, const int32_t* indexSurface, int Q, const uint16_t* rowsModel,
                                   const int32_t* indexModel, int M, int D, const pcreg_match_opts* par, uint32_t* pairs, double* metric,
                                   int32_t* n_pairs, void* workspace, size_t workspace_bytes, void* stream) {
    PCREG_ARG(rowsSurface && rowsModel && par && pairs && n_pairs && workspace && Q >= 0 && M >= 0 && D >= 1);
    PCREG_ARG(par->metric == PCREG_METRIC_SAD || par->metric == PCREG_METRIC_SSD);
    GUARD();
    return dev_get_matches_impl(rowsSurface, Q, D, rowsModel, M, D, D, kLayoutRowMajorU16, par, pairs, metric, n_pairs, workspace,
                                workspace_bytes, (hipStream_t)stream, indexSurface, indexModel);
}

int pcreg_dev_gather_matched_rows(const uint32_t* pairs, const int32_t* n_pairs, int cap, const double* featSurface,
                                  const double* featModel, double* pts1, double* pts2, void* stream) {
    PCREG_ARG(pairs && n_pairs && featSurface && featModel && pts1 && pts2 && cap >= 0);
    GUARD();
    return launch_gather_matched_rows(pairs, n_pairs, cap, featSurface, featModel, pts1, pts2, (hipStream_t)stream);
}


// ---- sphere-sweep driver pieces (completeExperimentFast.m:46-225, :291, :356-394), device tier
int pcreg_dev_sphere_counts(const double* feat, int V, const double* centres, int S, double R, int32_t* counts, void* stream) {
    PCREG_ARG(feat && centres && counts && V >= 0 && S >= 0);
    GUARD();
    return launch_sphere_counts(feat, V, centres, S, R, counts, (hipStream_t)stream);
}
int pcreg_dev_sphere_select_batched(const double* feat, int V, const double* centres, int S, double R, const int32_t* seg_off,
                                    int32_t* idx, double* feat_out, int32_t* n_out, void* stream) {
    PCREG_ARG(feat && centres && seg_off && idx && V >= 0 && S >= 0);
    GUARD();
    return launch_sphere_select_batched(feat, V, centres, S, R, seg_off, idx, feat_out, n_out, (hipStream_t)stream);
}
size_t pcreg_dev_get_matches_segmented_workspace(int Q, int VM, int D, int S, int total_rows, int max_rows) {
    return get_matches_segmented_workspace_bytes(Q, VM, D, S, total_rows, max_rows);
}
int pcreg_dev_get_matches_segmented(const double* descSurface, int Q, const double* descModel, int VM, int D, const int32_t* seg_rows,
                                    const int32_t* seg_off, int S, int total_rows, int max_rows, const pcreg_match_opts* par,
                                    uint32_t* pairs_all, double* metric_all, int32_t* n_pairs, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    PCREG_ARG(descSurface && descModel && seg_rows && seg_off && par && pairs_all && n_pairs && workspace);
    PCREG_ARG(Q >= 0 && VM >= 0 && D >= 1 && S >= 0 && S <= 65535 && total_rows >= 0 && max_rows >= 0 && max_rows <= total_rows);
    if (par->metric != PCREG_METRIC_SAD) { set_error("pcreg_dev_get_matches_segmented: Metric must be SAD (one pcreg_dev_get_matches call per segment handles SSD)"); return PCREG_E_ARG; }
    GUARD();
    return launch_get_matches_segmented(descSurface, Q, descModel, VM, D, seg_rows, seg_off, S, total_rows, max_rows, *par, pairs_all,
                                        metric_all, n_pairs, workspace, workspace_bytes, (hipStream_t)stream);
}
size_t pcreg_dev_segmented_model_bytes(int VM, int D) { return segmented_prepared_model_bytes(VM, D); }
int pcreg_dev_segmented_model_prepare(const double* descModel, int VM, int D, const pcreg_match_opts* par, void* prepared, size_t prepared_bytes, void* stream) {
    PCREG_ARG(descModel && par && prepared && VM >= 0 && D >= 1 && prepared_bytes >= segmented_prepared_model_bytes(VM, D));
    GUARD();
    double* P = (double*)prepared;
    return launch_segmented_prepare_model(descModel, VM, D, *par, P, P + (size_t)(VM > 0 ? VM : 1) * D, (hipStream_t)stream);
}
int pcreg_dev_get_matches_segmented_prepared(const double* descSurface, int Q, const double* descModel, int VM, int D, const void* prepared,
                                             int prepared_change_metric, double prepared_metric_factor, const int32_t* seg_rows,
                                             const int32_t* seg_off, int S, int total_rows, int max_rows, const pcreg_match_opts* par,
                                             uint32_t* pairs_all, double* metric_all, int32_t* n_pairs, void* workspace, size_t workspace_bytes,
                                             void* stream) {
    PCREG_ARG(descSurface && descModel && prepared && seg_rows && seg_off && par && pairs_all && n_pairs && workspace);
    PCREG_ARG(Q >= 0 && VM >= 0 && D >= 1 && S >= 0 && S <= 65535 && total_rows >= 0 && max_rows >= 0 && max_rows <= total_rows);
    if (par->metric != PCREG_METRIC_SAD) { set_error("pcreg_dev_get_matches_segmented_prepared: Metric must be SAD"); return PCREG_E_ARG; }
    GUARD();
    const double* P = (const double*)prepared;
    const SegPreparedModel prep{P, P + (size_t)(VM > 0 ? VM : 1) * D, VM, D, prepared_change_metric, prepared_metric_factor};
    return launch_get_matches_segmented(descSurface, Q, descModel, VM, D, seg_rows, seg_off, S, total_rows, max_rows, *par, pairs_all,
                                        metric_all, n_pairs, workspace, workspace_bytes, (hipStream_t)stream, &prep);
}
size_t pcreg_dev_sphere_select_workspace(int V) { return sphere_select_workspace_bytes(V); }
int pcreg_dev_sphere_select(const double* feat, int V, const double centre[3], double R, int32_t* idx, int32_t* n_out,
                            void* workspace, size_t workspace_bytes, void* stream) {
    PCREG_ARG(feat && centre && idx && n_out && workspace && V >= 0);
    GUARD();
    return launch_sphere_select(feat, V, centre, R, idx, n_out, workspace, workspace_bytes, (hipStream_t)stream);
}
int pcreg_dev_gather_rows_f64(const double* src, int D, const int32_t* idx, const int32_t* n, int cap, double* dst, void* stream) {
    PCREG_ARG(src && idx && n && dst && D >= 1 && cap >= 0);
    GUARD();
    return launch_gather_rows_f64(src, D, idx, n, cap, dst, (hipStream_t)stream);
}
int pcreg_dev_sweep_plan(const int32_t* n_pairs, int S, int putative_thresh, int32_t* trial_idx, int32_t* offsets, int32_t* n_trials, void* stream) {
    PCREG_ARG(n_pairs && trial_idx && offsets && n_trials && S >= 0);
    GUARD();
    return launch_sweep_plan(n_pairs, S, putative_thresh, trial_idx, offsets, n_trials, (hipStream_t)stream);
}
int pcreg_dev_sweep_gather(const uint32_t* pairs_all, int VS, const int32_t* n_pairs, const int32_t* trial_idx, const int32_t* offsets,
                           const int32_t* n_trials, int S, const double* featSurface, const double* featCur_all, const int64_t* row_off,
                           double* pts1, double* pts2, int ld, void* stream) {
    PCREG_ARG(pairs_all && n_pairs && trial_idx && offsets && n_trials && featSurface && featCur_all && row_off && pts1 && pts2 && S >= 0 && VS >= 0 && ld >= 0);
    GUARD();
    return launch_sweep_gather(pairs_all, VS, n_pairs, trial_idx, offsets, n_trials, S, featSurface, featCur_all, row_off, pts1, pts2, ld, (hipStream_t)stream);
}
size_t pcreg_dev_ransac_batched_workspace(int n_cap, int iterNum, int B) { return ransac_workspace_bytes(iterNum, B > 0 ? B : 1, n_cap); }
int pcreg_dev_ransac_batched(const double* pts1, const double* pts2, int ld, const int32_t* offsets, int B, int n_cap,
                             const pcreg_ransac_opts* opts, pcreg_dev_ransac_result* out, int32_t* inlier_idx,
                             void* workspace, size_t workspace_bytes, void* stream) {
    PCREG_ARG(pts1 && pts2 && offsets && opts && out && inlier_idx && workspace && B >= 1 && n_cap >= 0 && ld >= 0);
    PCREG_ARG(opts->minPtNum == 3);                    // the built-in sampler, seed + b for registration b
    GUARD();
    return launch_ransac(pts1, pts2, ld, offsets, nullptr, n_cap, B, *opts, nullptr, out, inlier_idx, nullptr, nullptr,
                         workspace, workspace_bytes, (hipStream_t)stream);
}

int pcreg_dev_quick_tf(const double* pts, int n, int ld, const double T[16], double* out, int ldo, void* stream) {
    PCREG_ARG(pts && T && out && n >= 0 && ld >= n && ldo >= n);
    GUARD();
    return launch_quick_tf(pts, n, ld, T, out, ldo, (hipStream_t)stream);
}
int pcreg_dev_refine_by_distance(const double* pts1, const double* pts2, const int32_t* n_dev, int cap, int ld, double maxDist,
                                 double* T16, int32_t* info, void* stream) {
    PCREG_ARG(pts1 && pts2 && n_dev && T16 && info && cap >= 0 && ld >= cap);
    GUARD();
    return launch_refine_by_distance(pts1, pts2, n_dev, cap, ld, maxDist, T16, info, (hipStream_t)stream);
}

}  // extern "C"
