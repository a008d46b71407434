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
//     1 all-gather  of the per-rank top-2 lists ([2][Q][2] 4-byte words)      + merge kernel
//     1 all-reduce  (int32 SUM) of the dense [4][Q] table, column = query (one contributor per column: exact)
//     1 all-gather  of the 112-byte partial RANSAC results                    + finish kernel
// Two transports behind the same three calls:
//   * RCCL over xGMI (pcreg_comm_init): opened with dlopen (no link-time dependency: the library loads on boxes
//     without RCCL, and lives next to a torch that bundles its own copy);
//   * host-staged (pcreg_comm_init_host_staged): every rank copies its buffer into a POSIX shared-memory segment,
//     a sense-reversing barrier, every rank combines -- for workers that SHARE a GPU (RCCL refuses two ranks on one
//     device) and for boxes without RCCL.  Latency-sized messages (<= 16 Q bytes), so PCIe is not the issue; it is
//     also how the N > 1 path of this file is tested on the one-GPU box (tests/test_gpu_comm.py).
// A communicator lives on the device that was current at init: its stream and scratch are created there and
// released by pcreg_comm_destroy; calls from another device are refused.  Every argument check and every
// allocation of a sharded call happens BEFORE its first collective, so a rank either fails before anybody waits
// for it or goes through (a HIP launch failure in between is fatal for the whole group, as with any collective).
#include "common.hpp"
#include <rccl/rccl.h>          // types and prototypes only; the functions are resolved with dlsym
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
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
int g_rank = 0, g_world = 0, g_cdev = -1;
bool g_open = false;
hipStream_t g_cstream = nullptr;
Scratch g_cs;                       // the sharded entry points' own device buffers

// ---- host-staged transport -----------------------------------------------------------------------------------------
constexpr size_t kHsSlot = 64u << 20;               // bytes per rank (sparse: only touched pages exist)
// The segment's header.  Rank 0 always makes a FRESH segment (shm_unlink, then O_CREAT | O_EXCL: a new inode, zero-filled),
// fills the header and publishes `magic` last; the other ranks only open what exists.  A segment a crashed run left behind
// (attached == world, sense == 1, count != 0 ...) is therefore never reused by rank 0, and a rank that opened it before rank
// 0 replaced it finds out: either it is turned away at once (attached was already at world) or, while it waits for `ready`,
// it sees the name point at another inode and starts over.  The last rank to leave unlinks the name (if it is still ours).
constexpr unsigned long long kHsMagic = 0x70637265675f6873ull;      // "pcreg_hs"
struct HsHeader {
    std::atomic<unsigned long long> magic;     // kHsMagic once rank 0 has initialised the header
    std::atomic<int> world;
    std::atomic<int> attached;                 // ranks that have mapped this segment and been admitted
    std::atomic<int> ready;                    // rank 0: everybody is attached, the barrier may be used
    std::atomic<int> detached;
    std::atomic<int> count; std::atomic<int> sense;      // the sense-reversing barrier
};
struct HostStaged {
    bool on = false;
    std::string name;
    void* base = nullptr; size_t bytes = 0;
    ino_t ino = 0;                      // the segment this process mapped
    int local_sense = 0;
    std::vector<char> tmp;
    HsHeader* hdr() const { return (HsHeader*)base; }
    char* slot(int r) const { return (char*)base + 4096 + (size_t)r * kHsSlot; }
};
HostStaged g_hs;

int hs_barrier() {
    HsHeader* h = g_hs.hdr();
    g_hs.local_sense ^= 1;
    if (h->count.fetch_add(1, std::memory_order_acq_rel) == g_world - 1) {
        h->count.store(0, std::memory_order_relaxed);
        h->sense.store(g_hs.local_sense, std::memory_order_release);
        return PCREG_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->sense.load(std::memory_order_acquire) != g_hs.local_sense) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
            set_error("host-staged communicator: a rank did not arrive within 120 s"); return PCREG_E_HIP;
        }
    }
    return PCREG_OK;
}
// every rank's `bytes` (device, on st) -> all ranks' blocks back to back in `all` (gather) or their int32 sum in `buf`
int hs_exchange(void* buf, void* all, size_t bytes, bool sum, hipStream_t st) {
    if (bytes > kHsSlot) { set_error("host-staged communicator: %zu bytes exceed the %zu-byte slot", bytes, kHsSlot); return PCREG_E_ARG; }
    PCREG_HIP(hipMemcpyAsync(g_hs.slot(g_rank), buf, bytes, hipMemcpyDeviceToHost, st));
    PCREG_HIP(hipStreamSynchronize(st));
    int rc = hs_barrier(); if (rc) return rc;
    if (sum) {
        g_hs.tmp.assign(bytes, 0);
        int32_t* acc = (int32_t*)g_hs.tmp.data();
        for (int r = 0; r < g_world; ++r) { const int32_t* s = (const int32_t*)g_hs.slot(r); for (size_t k = 0; k < bytes / 4; ++k) acc[k] += s[k]; }
        PCREG_HIP(hipMemcpyAsync(buf, g_hs.tmp.data(), bytes, hipMemcpyHostToDevice, st));
    } else {
        g_hs.tmp.resize(bytes * (size_t)g_world);
        for (int r = 0; r < g_world; ++r) memcpy(g_hs.tmp.data() + (size_t)r * bytes, g_hs.slot(r), bytes);
        PCREG_HIP(hipMemcpyAsync(all, g_hs.tmp.data(), bytes * (size_t)g_world, hipMemcpyHostToDevice, st));
    }
    PCREG_HIP(hipStreamSynchronize(st));
    return hs_barrier();                     // nobody overwrites a slot that somebody still reads
}

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
    if (!g_open) { set_error("pcreg_comm_init has not been called on this process"); return PCREG_E_ARG; }
    int dev = -1;
    PCREG_HIP(hipGetDevice(&dev));
    if (dev != g_cdev) { set_error("the communicator was opened on device %d, the calling thread is on device %d", g_cdev, dev); return PCREG_E_ARG; }
    return PCREG_OK;
}
// n int32 words per rank
int all_gather_words(void* send, void* recv, size_t n, hipStream_t st) {
    if (g_hs.on) return hs_exchange(send, recv, n * 4, false, st);
    PCREG_NCCL(g_rccl.AllGather(send, recv, n, ncclInt32, g_comm, st));
    return PCREG_OK;
}
int all_reduce_sum_words(void* buf, size_t n, hipStream_t st) {
    if (g_hs.on) return hs_exchange(buf, nullptr, n * 4, true, st);
    PCREG_NCCL(g_rccl.AllReduce(buf, buf, n, ncclInt32, ncclSum, g_comm, st));
    return PCREG_OK;
}
int open_common(int rank, int world) {
    CTRY(ensure_device());
    PCREG_HIP(hipGetDevice(&g_cdev));
    PCREG_HIP(hipStreamCreateWithFlags(&g_cstream, hipStreamNonBlocking));
    g_rank = rank; g_world = world; g_open = true;
    return PCREG_OK;
}
void close_common() {
    if (g_cstream) { (void)hipStreamSynchronize(g_cstream); (void)hipStreamDestroy(g_cstream); g_cstream = nullptr; }
    g_cs.release_all();
    g_rank = 0; g_world = 0; g_cdev = -1; g_open = false;
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
    std::lock_guard<std::mutex> lock(g_cmu);
    if (g_open) { set_error("pcreg_comm_init: a communicator is already open (pcreg_comm_destroy first)"); return PCREG_E_ARG; }
    CTRY(load_rccl());
    CTRY(open_common(rank, world));
    ncclUniqueId u; memcpy(&u, id, sizeof u);
    ncclResult_t r = g_rccl.CommInitRank(&g_comm, world, u, rank);          // on the device recorded above (pcreg_set_device)
    if (r != ncclSuccess) { set_error("RCCL: ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); g_comm = nullptr; close_common(); return PCREG_E_HIP; }
    return PCREG_OK;
}

int pcreg_comm_init_host_staged(int rank, int world, const char* name) {
    PCREG_ARG(name != nullptr && name[0] == '/' && strlen(name) < 200 && world >= 1 && rank >= 0 && rank < world);
    std::lock_guard<std::mutex> lock(g_cmu);
    if (g_open) { set_error("pcreg_comm_init_host_staged: a communicator is already open (pcreg_comm_destroy first)"); return PCREG_E_ARG; }
    const size_t bytes = 4096 + (size_t)world * kHsSlot;
    const auto t0 = std::chrono::steady_clock::now();
    auto timed_out = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120); };
    auto name_ino = [&](ino_t* out) {            // the inode the name refers to NOW (false: no such segment)
        int f = shm_open(name, O_RDWR, 0600);
        if (f < 0) return false;
        struct stat sb; const bool ok = fstat(f, &sb) == 0;
        close(f);
        if (ok) *out = sb.st_ino;
        return ok;
    };
    void* base = MAP_FAILED; ino_t ino = 0;
    if (rank == 0) {
        (void)shm_unlink(name);                  // whatever a previous run left under this name is not ours
        int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) { set_error("shm_open(%s, O_CREAT | O_EXCL) failed: %s", name, strerror(errno)); return PCREG_E_HIP; }
        struct stat sb;
        if (ftruncate(fd, (off_t)bytes) != 0 || fstat(fd, &sb) != 0) { close(fd); shm_unlink(name); set_error("ftruncate(%s) failed", name); return PCREG_E_HIP; }   // zero-filled
        base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (base == MAP_FAILED) { shm_unlink(name); set_error("mmap(%s) failed", name); return PCREG_E_HIP; }
        ino = sb.st_ino;
        HsHeader* h = (HsHeader*)base;
        h->world.store(world); h->attached.store(1); h->ready.store(0); h->detached.store(0); h->count.store(0); h->sense.store(0);
        h->magic.store(kHsMagic, std::memory_order_release);
    } else {
        for (;;) {                               // open what rank 0 made; never create
            if (timed_out()) { set_error("host-staged communicator: rank %d found no live segment %s within 120 s", rank, name); return PCREG_E_HIP; }
            int fd = shm_open(name, O_RDWR, 0600);
            struct stat sb;
            if (fd < 0 || fstat(fd, &sb) != 0 || (size_t)sb.st_size < bytes) { if (fd >= 0) close(fd); sched_yield(); continue; }
            base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (base == MAP_FAILED) { set_error("mmap(%s) failed", name); return PCREG_E_HIP; }
            ino = sb.st_ino;
            HsHeader* h = (HsHeader*)base;
            bool stale = false;
            while (h->magic.load(std::memory_order_acquire) != kHsMagic && !stale) {      // rank 0 is still filling the header -- or this is debris
                sched_yield();
                ino_t now; stale = timed_out() || (name_ino(&now) && now != ino);
            }
            // admitted only while there is room: a complete segment of an earlier run (attached == world) turns the newcomer away
            if (!stale && h->world.load() == world && h->attached.fetch_add(1, std::memory_order_acq_rel) < world) {
                while (!h->ready.load(std::memory_order_acquire) && !stale) {              // rank 0 of THIS run has seen everybody
                    sched_yield();
                    ino_t now; stale = timed_out() || !name_ino(&now) || now != ino;       // ... unless the name has moved on: debris
                }
                if (!stale) break;
            }
            munmap(base, bytes); base = MAP_FAILED;
            sched_yield();
        }
    }
    int rc = open_common(rank, world);
    if (rc) { munmap(base, bytes); if (rank == 0) shm_unlink(name); return rc; }
    g_hs.on = true; g_hs.name = name; g_hs.base = base; g_hs.bytes = bytes; g_hs.ino = ino; g_hs.local_sense = 0;
    if (rank == 0) {
        while (g_hs.hdr()->attached.load(std::memory_order_acquire) < world) {
            sched_yield();
            if (timed_out()) {
                set_error("host-staged communicator: %d of %d ranks attached", g_hs.hdr()->attached.load(), world);
                munmap(base, bytes); shm_unlink(name); g_hs = HostStaged{}; close_common();
                return PCREG_E_HIP;
            }
        }
        g_hs.hdr()->ready.store(1, std::memory_order_release);
    }
    return PCREG_OK;
}

int pcreg_comm_rank(int* rank, int* world) {
    PCREG_ARG(rank && world);
    std::lock_guard<std::mutex> lock(g_cmu);
    if (!g_open) { set_error("pcreg_comm_init has not been called on this process"); return PCREG_E_ARG; }
    *rank = g_rank; *world = g_world;
    return PCREG_OK;
}

int pcreg_comm_destroy(void) {
    std::lock_guard<std::mutex> lock(g_cmu);
    if (g_cstream) (void)hipStreamSynchronize(g_cstream);
    if (g_comm) { (void)g_rccl.CommDestroy(g_comm); g_comm = nullptr; }
    if (g_hs.on) {
        // the last rank to leave removes the name -- if it still refers to the segment this process mapped
        const bool last = g_hs.hdr()->detached.fetch_add(1, std::memory_order_acq_rel) + 1 >= g_world;
        munmap(g_hs.base, g_hs.bytes);
        if (last) {
            int f = shm_open(g_hs.name.c_str(), O_RDWR, 0600);
            struct stat sb;
            const bool ours = f >= 0 && fstat(f, &sb) == 0 && sb.st_ino == g_hs.ino;
            if (f >= 0) close(f);
            if (ours) shm_unlink(g_hs.name.c_str());
        }
        g_hs = HostStaged{};
    }
    close_common();
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
    if (Q == 0 || M_total == 0) return PCREG_OK;                 // the same on every rank: nobody enters a collective
    const int R = g_world;
    hipStream_t st = g_cstream;
    const size_t q4 = (size_t)Q * 4;
    // every buffer of the call, before the first collective
    void *dq, *dm, *dpack, *dall, *didx, *ddist, *dcnt, *dtable, *dpairs, *ws, *block;
    const size_t wsb = search_ws_bytes(Q, M_local);
    CTRY(g_cs.get(0, sizeof(float) * 3 * (size_t)Q, &dq));
    CTRY(g_cs.get(1, sizeof(float) * 3 * (size_t)(M_local > 0 ? M_local : 1), &dm));
    CTRY(g_cs.get(2, q4 * 4, &dpack));                        // [2][Q][2] words: indices, then the distances' bit patterns
    CTRY(g_cs.get(3, q4 * 4 * (size_t)R, &dall));
    CTRY(g_cs.get(4, q4 * 2, &didx));
    CTRY(g_cs.get(5, q4 * 2, &ddist));
    CTRY(g_cs.get(9, 256, &dcnt));
    CTRY(g_cs.get(10, q4 * 4, &dtable));
    CTRY(g_cs.get(11, q4 * 2, &dpairs));
    CTRY(g_cs.get(12, wsb, &ws));
    CTRY(g_cs.get(13, model_prep_bytes(M_local), &block));
    int32_t* n_pairs = (int32_t*)dcnt;
    int32_t* idx_l = (int32_t*)dpack; float* dist_l = (float*)((int32_t*)dpack + 2 * (size_t)Q);
    if (ldq == Q) PCREG_HIP(hipMemcpyAsync(dq, q, sizeof(float) * 3 * (size_t)Q, hipMemcpyHostToDevice, st));
    else PCREG_HIP(hipMemcpy2DAsync(dq, sizeof(float) * (size_t)Q, q, sizeof(float) * (size_t)ldq, sizeof(float) * (size_t)Q, 3, hipMemcpyHostToDevice, st));
    if (M_local > 0) {
        if (ldm == M_local) PCREG_HIP(hipMemcpyAsync(dm, m_local, sizeof(float) * 3 * (size_t)M_local, hipMemcpyHostToDevice, st));
        else PCREG_HIP(hipMemcpy2DAsync(dm, sizeof(float) * (size_t)M_local, m_local, sizeof(float) * (size_t)ldm, sizeof(float) * (size_t)M_local, 3, hipMemcpyHostToDevice, st));
    }
    // 1. prepare this rank's shard, local top-2 (global row numbers) + the query grid, all-gather, merge by (distance, index)
    const ModelView v = model_view((float*)dm, M_local, M_local > 0 ? M_local : 1, block);
    CTRY(launch_model_prepare(v, st));
    CTRY(launch_model_search(v, (float*)dq, Q, Q, m_lo, idx_l, dist_l, ws, wsb, true, true, st));
    CTRY(all_gather_words(dpack, dall, q4, st));
    CTRY(launch_merge_top2_f32((int32_t*)dall, (float*)((int32_t*)dall + 2 * (size_t)Q), R, Q, (int32_t*)didx, (float*)ddist, st, q4));
    // 2. filters on every rank, the Unique verdict and the matched coordinates by the rank that owns the model row;
    //    one integer SUM publishes both; 3. ordered compaction from the summed table
    CTRY(launch_match_table(v, m_lo, M_total, (float*)dq, Q, Q, (int32_t*)didx, (float*)ddist, thr_abs, max_ratio, unique, ws, wsb, (int32_t*)dtable, st));
    CTRY(all_reduce_sum_words(dtable, q4, st));
    CTRY(launch_match_from_table((float*)dq, Q, Q, M_total, (int32_t*)didx, (float*)ddist, thr_abs, max_ratio, (int32_t*)dtable, ws, wsb,
                                 (uint32_t*)dpairs, nullptr, nullptr, n_pairs, st));
    int32_t np = 0;
    PCREG_HIP(hipMemcpyAsync(&np, n_pairs, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PCREG_HIP(hipStreamSynchronize(st));
    if (np > 0) PCREG_HIP(hipMemcpy(pairs, dpairs, sizeof(uint32_t) * 2 * (size_t)np, hipMemcpyDeviceToHost));
    *P = np;
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
    CTRY(launch_ransac_partial((double*)d1, (double*)d2, cap, (int32_t*)dn, cap, *opts, nullptr, begin, count, (pcreg_dev_ransac_part*)dpart, ws, wsb, st));
    static_assert(sizeof(pcreg_dev_ransac_part) == 112, "the gathered struct is 112 bytes");
    CTRY(all_gather_words(dpart, dall, sizeof(pcreg_dev_ransac_part) / 4, st));
    CTRY(launch_ransac_finish((double*)d1, (double*)d2, cap, (int32_t*)dn, cap, *opts, (pcreg_dev_ransac_part*)dall,
                              (pcreg_dev_ransac_result*)dres, (int32_t*)dinl, st, R));
    pcreg_dev_ransac_result res;
    PCREG_HIP(hipMemcpyAsync(&res, dres, sizeof res, hipMemcpyDeviceToHost, st));
    if (n > 0) PCREG_HIP(hipMemcpyAsync(inlier_idx, dinl, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
    PCREG_HIP(hipStreamSynchronize(st));
    memcpy(T, res.T, sizeof(double) * 16);
    *n_inliers = res.n_inliers; *num_success = res.num_success; *max_inliers = res.max_inliers; *failed = res.failed;
    return PCREG_OK;
}

}  // extern "C"
