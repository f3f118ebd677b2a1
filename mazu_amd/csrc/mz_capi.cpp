// mz_capi.cpp -- the extern "C" boundary declared in include/mazu_amd.h.
// Thin: argument checks, host<->device staging for the host-pointer entry points, and the
// lazily created per-device replica.  All query work is done by the kernels in mz_device.hip.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <algorithm>
#include <exception>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <new>

#include "mz_device.hpp"

using namespace mz;

struct mz_unitig_set {
    UnitigSet us;
};

// Staging context of one host-pointer call (mz_sshash_k2u_batch, mz_sshash_k2u_reads): per slot a stream, a pinned chunk
// buffer each way and their device twins.  Pooled per handle (a call takes one, concurrent callers get their own), freed by
// mz_sshash_free.
constexpr int kPipeSlots = 3; // chunk c is staged while chunk c-1 is on the GPU and chunk c-2 is drained
struct HostPipe {
    int device = -1;
    uint64_t cap = 0; // queries per chunk
    hipStream_t stream[kPipeSlots] = {}; // one per slot: the upload of a chunk overlaps the download of the one before
    hipEvent_t done[kPipeSlots] = {};
    uint64_t *h_in[kPipeSlots] = {}, *d_in[kPipeSlots] = {};
    mz_k2upos *h_out[kPipeSlots] = {}, *d_out[kPipeSlots] = {};
    mz_counts *d_counts = nullptr, *h_counts = nullptr;
    void release() {
        for (int j = 0; j < kPipeSlots; ++j) {
            if (h_in[j]) (void)hipHostFree(h_in[j]);
            if (h_out[j]) (void)hipHostFree(h_out[j]);
            if (d_in[j]) (void)hipFree(d_in[j]);
            if (d_out[j]) (void)hipFree(d_out[j]);
            if (done[j]) (void)hipEventDestroy(done[j]);
            if (stream[j]) (void)hipStreamDestroy(stream[j]);
            h_in[j] = d_in[j] = nullptr; h_out[j] = d_out[j] = nullptr; done[j] = nullptr; stream[j] = nullptr;
        }
        if (d_counts) (void)hipFree(d_counts);
        if (h_counts) (void)hipHostFree(h_counts);
        d_counts = h_counts = nullptr; cap = 0;
    }
};

struct mz_sshash {
    SSHashHost host;
    QueryOpts qopts = QueryOpts::from_env(); // kernel choices of the query entry points (mz_sshash_set_query_opts); read without a lock: set them before querying
    std::vector<HostPipe *> idle_pipes;   // guarded by mu
    DeviceIndex *dev = nullptr;           // default replica (first one placed, or the last mz_sshash_to_device target)
    std::vector<DeviceIndex *> replicas;  // one per device the index has been placed on (index replicated per GPU)
    bool built_on_device = false;         // the GPU builder made the arrays
    std::mutex mu;
};

struct mz_streaming {
    mz_sshash *ix;
    bool is_warm = false;
    mz_k2upos prev{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, MZ_NO_MATCH};
};

#define HIPC(expr)                                                                             \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            set_error("HIP error %d (%s): %s", (int)_e, hipGetErrorString(_e), #expr);         \
            return MZ_EHIP;                                                                    \
        }                                                                                      \
    } while (0)

// replica on `device`, uploaded on first use; *out is owned by the handle
// arrays of a GPU-built index live in HBM only until the host needs them (save, a replica on another GPU); caller holds ix->mu
static int host_arrays(mz_sshash *ix) {
    if (ix->host.arrays_on_host) return MZ_OK;
    if (ix->replicas.empty()) { set_error("index arrays are neither on the host nor on a device"); return MZ_EINVAL; }
    return device_download(*ix->replicas.front(), ix->host);
}
static int replica_on(mz_sshash *ix, int device, DeviceIndex **out) {
    std::lock_guard<std::mutex> g(ix->mu);
    for (DeviceIndex *r : ix->replicas)
        if (r->device == device) { *out = r; return MZ_OK; }
    DeviceIndex *nd = nullptr;
    int rc = host_arrays(ix);
    if (rc != MZ_OK) return rc;
    rc = device_upload(ix->host, device, &nd);
    if (rc != MZ_OK) return rc;
    ix->replicas.push_back(nd);
    if (!ix->dev) ix->dev = nd;
    *out = nd;
    return MZ_OK;
}
static int ensure_device(mz_sshash *ix) {
    if (ix->dev) return MZ_OK;
    DeviceIndex *r;
    return replica_on(ix, 0, &r);
}
// device-pointer entry points run on the replica of the GPU that owns the caller's buffer
static int replica_for_pointer(mz_sshash *ix, const void *dptr, DeviceIndex **out) {
    hipPointerAttribute_t attr;
    if (dptr && hipPointerGetAttributes(&attr, dptr) == hipSuccess && attr.type == hipMemoryTypeDevice)
        return replica_on(ix, attr.device, out);
    (void)hipGetLastError(); // not a device pointer we can classify: fall back to the default replica
    int rc = ensure_device(ix);
    *out = ix->dev;
    return rc;
}

int mz_internal_default_device(mz_sshash *ix, int *device) {
    int rc = ensure_device(ix);
    if (rc == MZ_OK) *device = ix->dev->device;
    return rc;
}

// fields of a caller's struct that lie within the struct_size it states (a caller built against an older, shorter struct leaves the rest at 0 = rule)
template <class T>
static bool copy_opts(const T *src, T *dst, const char *what) {
    *dst = T();
    if (!src) return true;
    if (src->struct_size < sizeof(uint32_t) || src->struct_size > 4096) { set_error("%s: struct_size %u is not a size of this struct", what, src->struct_size); return false; }
    std::memcpy(dst, src, std::min<size_t>(src->struct_size, sizeof(T)));
    dst->struct_size = (uint32_t)sizeof(T);
    return true;
}
extern "C" {

const char *mz_last_error(void) { return last_error(); }
uint32_t mz_abi_version(void) { return 1; }
int mz_device_count(void) { return device_count(); }
int mz_host_threads(void) { return hw_threads(); }

// ---------------------------------------------------------------- UnitigSet
#define NEW_US(call)                                  \
    if (!out) { set_error("null out pointer"); return MZ_EINVAL; } \
    *out = nullptr;                                   \
    mz_unitig_set *u = new (std::nothrow) mz_unitig_set(); \
    if (!u) return MZ_ENOMEM;                         \
    int rc = u->us.call;                              \
    if (rc != MZ_OK) { delete u; return rc; }         \
    *out = u;                                         \
    return MZ_OK;

int mz_unitig_set_from_seqs(const char *const *seqs, uint64_t n_seqs, uint32_t k, mz_unitig_set **out) { NEW_US(from_seqs(seqs, n_seqs, k)) }
int mz_unitig_set_from_cf_prefix(const char *prefix, mz_unitig_set **out) {
    if (!prefix) { set_error("null prefix"); return MZ_EINVAL; }
    NEW_US(from_cf_prefix(prefix))
}
int mz_unitig_set_from_packed(const uint64_t *words, uint64_t total_len, const uint64_t *starts, uint64_t n_unitigs,
                              uint32_t k, mz_unitig_set **out) { NEW_US(from_packed(words, total_len, starts, n_unitigs, k)) }
void mz_unitig_set_free(mz_unitig_set *us) { delete us; }
uint32_t mz_unitig_set_k(const mz_unitig_set *us) { return us->us.k; }
uint64_t mz_unitig_set_n_unitigs(const mz_unitig_set *us) { return us->us.n_unitigs; }
uint64_t mz_unitig_set_total_len(const mz_unitig_set *us) { return us->us.total_len; }
uint64_t mz_unitig_set_n_kmers(const mz_unitig_set *us) { return us->us.n_kmers(); }
uint64_t mz_unitig_set_unitig_len(const mz_unitig_set *us, uint64_t i) { return i < us->us.n_unitigs ? us->us.unitig_len(i) : 0; }
uint64_t mz_unitig_set_unitig_start_pos(const mz_unitig_set *us, uint64_t i) { return i <= us->us.n_unitigs ? us->us.unitig_start_pos(i) : 0; }
uint64_t mz_unitig_set_pos_to_id(const mz_unitig_set *us, uint64_t pos) { return pos < us->us.total_len ? us->us.pos_to_id(pos) : us->us.n_unitigs; }
const uint64_t *mz_unitig_set_useq_words(const mz_unitig_set *us) { return us->us.useq.data(); }

// ------------------------------------------------------------------- SSHash
static int build_host(const mz_unitig_set *us, uint32_t w, uint64_t skew_param, uint64_t seed, int n_threads, const BuildOpts &bo, mz_sshash **out) {
    if (!us || !out) { set_error("null argument"); return MZ_EINVAL; }
    *out = nullptr;
    mz_sshash *ix = new (std::nothrow) mz_sshash();
    if (!ix) return MZ_ENOMEM;
    int rc;
    try {
        rc = ix->host.build(us->us, w, skew_param, seed, n_threads, bo);
    } catch (const std::bad_alloc &) {
        set_error("out of host memory while building the index");
        rc = MZ_ENOMEM;
    }
    if (rc != MZ_OK) { delete ix; return rc; }
    *out = ix;
    return MZ_OK;
}
static int build_device(const mz_unitig_set *us, uint32_t w, uint64_t skew_param, uint64_t seed, int device, int n_threads, const BuildOpts &bo, mz_sshash **out);
int mz_sshash_from_unitig_set(const mz_unitig_set *us, uint32_t w, uint64_t skew_param, uint64_t seed, int n_threads,
                              mz_sshash **out) {
    return build_host(us, w, skew_param, seed, n_threads, BuildOpts(), out);
}
int mz_sshash_from_unitig_set_device(const mz_unitig_set *us, uint32_t w, uint64_t skew_param, uint64_t seed, int device, mz_sshash **out) {
    return build_device(us, w, skew_param, seed, device, 0, BuildOpts(), out);
}
int mz_sshash_from_unitig_set_opts(const mz_unitig_set *us, uint32_t w, uint64_t skew_param, uint64_t seed, const mz_build_opts *opts, mz_sshash **out) {
    mz_build_opts o;
    if (!copy_opts(opts, &o, "mz_build_opts")) return MZ_EINVAL;
    if (o.minimizer_lines < -1 || o.minimizer_lines > 1) { set_error("mz_build_opts.minimizer_lines must be -1, 0 or 1"); return MZ_EINVAL; }
    BuildOpts bo;
    bo.minimizer_lines = o.minimizer_lines;
    bo.lines_max_load = o.lines_max_load;
    bo.direct_bytes_per_key = o.direct_bytes_per_key;
    if (o.device >= 0 && opts) return build_device(us, w, skew_param, seed, o.device, o.n_threads, bo, out);
    return build_host(us, w, skew_param, seed, o.n_threads, bo, out);
}
int mz_sshash_set_query_opts(mz_sshash *ix, const mz_query_opts *opts) {
    if (!ix) { set_error("null index"); return MZ_EINVAL; }
    mz_query_opts o;
    if (!copy_opts(opts, &o, "mz_query_opts")) return MZ_EINVAL;
    if (o.point_kernel < 0 || o.point_kernel > 3 || o.reads_lines < -1 || o.reads_lines > 1 || o.reads_resolve < 0 || o.reads_resolve > 2 || o.reads_kernel < 0 ||
        o.reads_kernel > 2 || o.reads_verify < 0 || o.reads_verify > 2 || o.reads_grid < 0) { set_error("mz_query_opts: a field is out of its range"); return MZ_EINVAL; }
    QueryOpts q;
    q.point_kernel = o.point_kernel; q.reads_lines = o.reads_lines; q.reads_resolve = o.reads_resolve; q.reads_kernel = o.reads_kernel;
    q.reads_verify = o.reads_verify; q.reads_grid = o.reads_grid;
    ix->qopts = q;
    return MZ_OK;
}
int mz_sshash_get_query_opts(const mz_sshash *ix, mz_query_opts *opts) {
    if (!ix || !opts) { set_error("null argument"); return MZ_EINVAL; }
    const QueryOpts &q = ix->qopts;
    *opts = mz_query_opts{(uint32_t)sizeof(mz_query_opts), q.point_kernel, q.reads_lines, q.reads_resolve, q.reads_kernel, q.reads_verify, q.reads_grid};
    return MZ_OK;
}
static int build_device(const mz_unitig_set *us, uint32_t w, uint64_t skew_param, uint64_t seed, int device, int n_threads, const BuildOpts &bo, mz_sshash **out) {
    DeviceGuard guard;
    if (!us || !out) { set_error("null argument"); return MZ_EINVAL; }
    *out = nullptr;
    if (w == 0 || w > us->us.k) { set_error("SSHash: minimizer size w=%u must be in 1..=k (k=%u)", w, us->us.k); return MZ_EINVAL; }
    mz_sshash *ix = new (std::nothrow) mz_sshash();
    if (!ix) return MZ_ENOMEM;
    int rc = MZ_EINVAL;
    try {
        DeviceIndex *kept = nullptr; // the GPU builder leaves its arrays in HBM as the replica of `device`: no download, no second upload
        if (gpu_build_supported(us->us, w)) rc = build_on_device(ix->host, us->us, w, skew_param, seed, device, &kept, bo);
        if (rc == MZ_OK && kept) { ix->replicas.push_back(kept); ix->dev = kept; }
        ix->built_on_device = rc == MZ_OK;
        if (rc == MZ_ENOMEM) { // the device ran out of memory next to what already lives there: the host builder can still do it
            std::fprintf(stderr, "[mazu_amd] GPU index build ran out of device memory (%s); building on the host instead\n", last_error());
            ix->host = SSHashHost();
            rc = MZ_EINVAL;
        }
        if (rc == MZ_EINVAL) rc = ix->host.build(us->us, w, skew_param, seed, n_threads, bo); // configuration left to the host builder
    } catch (const std::bad_alloc &) {
        set_error("out of host memory while building the index");
        rc = MZ_ENOMEM;
    }
    if (rc != MZ_OK) { delete ix; return rc; }
    *out = ix;
    return MZ_OK;
}
void mz_sshash_free(mz_sshash *ix) {
    if (!ix) return;
    DeviceGuard guard;
    for (HostPipe *p : ix->idle_pipes) { p->release(); delete p; }
    for (DeviceIndex *r : ix->replicas) device_free(r);
    delete ix;
}
uint32_t mz_sshash_k(const mz_sshash *ix) { return ix->host.us.k; }
uint64_t mz_sshash_n_kmers(const mz_sshash *ix) { return ix->host.us.n_kmers(); }
uint64_t mz_sshash_sum_unitigs_len(const mz_sshash *ix) { return ix->host.us.total_len; }
uint64_t mz_sshash_unitig_len(const mz_sshash *ix, uint64_t id) { return id < ix->host.us.n_unitigs ? ix->host.us.unitig_len(id) : 0; }
uint64_t mz_sshash_n_unitigs(const mz_sshash *ix) { return ix->host.us.n_unitigs; }
int mz_sshash_unitig_seq(const mz_sshash *ix, uint64_t id, const uint64_t **words, uint64_t *base_offset, uint64_t *len) {
    if (!ix || id >= ix->host.us.n_unitigs) { set_error("unitig id out of range"); return MZ_EINVAL; }
    if (words) *words = ix->host.us.useq.data();
    if (base_offset) *base_offset = ix->host.us.unitig_start_pos(id);
    if (len) *len = ix->host.us.unitig_len(id);
    return MZ_OK;
}
const mz_unitig_set *mz_sshash_as_unitig_set(const mz_sshash *ix) {
    // mz_unitig_set is a struct holding exactly one UnitigSet, so the owned set can be lent in place
    static_assert(sizeof(mz_unitig_set) == sizeof(UnitigSet), "mz_unitig_set must wrap UnitigSet only");
    return reinterpret_cast<const mz_unitig_set *>(&ix->host.us);
}
uint32_t mz_sshash_w(const mz_sshash *ix) { return ix->host.w; }
uint64_t mz_sshash_skew_param(const mz_sshash *ix) { return ix->host.skew_param; }
uint64_t mz_sshash_n_minimizers(const mz_sshash *ix) { return ix->host.n_minimizers; }
uint64_t mz_sshash_n_minimizer_occs(const mz_sshash *ix) { return ix->host.pos.n; }
uint64_t mz_sshash_n_kmers_in_skew_index(const mz_sshash *ix) { return ix->host.has_skew ? ix->host.skew.n_keys : 0; }
uint64_t mz_sshash_num_bits(const mz_sshash *ix) { return ix->host.num_bits(); }
uint32_t mz_sshash_minimizer_mphf_kind(const mz_sshash *ix) { return ix->host.mm.direct ? 1u : 0u; }
uint32_t mz_sshash_has_minimizer_lines(const mz_sshash *ix) { return ix->host.front.present ? 1u : 0u; }
uint64_t mz_sshash_device_bytes(const mz_sshash *ix) { return ix->dev ? ix->dev->bytes : 0; } /* per replica */
uint32_t mz_sshash_built_on_device(const mz_sshash *ix) { return ix && ix->built_on_device ? 1u : 0u; }

int mz_sshash_to_device(mz_sshash *ix, int device) {
    DeviceGuard guard;
    if (!ix) { set_error("null index"); return MZ_EINVAL; }
    DeviceIndex *r = nullptr;
    int rc = replica_on(ix, device, &r);
    if (rc != MZ_OK) return rc;
    ix->dev = r; // host-pointer entry points use this replica from now on
    return MZ_OK;
}

int mz_sshash_k2u_batch_device(mz_sshash *ix, const uint64_t *d_fw_words, uint64_t n, mz_k2upos *d_out, mz_counts *d_counts,
                               void *stream) {
    DeviceGuard guard;
    if (!ix || (n && (!d_fw_words || !d_out))) { set_error("null argument"); return MZ_EINVAL; }
    DeviceIndex *r = nullptr;
    int rc = replica_for_pointer(ix, d_out, &r);
    if (rc != MZ_OK) return rc;
    return launch_k2u_batch(*r, ix->qopts, d_fw_words, n, d_out, d_counts, (hipStream_t)stream);
}

int mz_sshash_k2u_batch_compact_device(mz_sshash *ix, const uint64_t *d_fw_words, uint64_t n, uint64_t *d_hit_idx,
                                       mz_k2upos *d_hits, uint64_t *d_n_hits, void *stream) {
    DeviceGuard guard;
    if (!ix || !d_n_hits || (n && (!d_fw_words || !d_hit_idx || !d_hits))) { set_error("null argument"); return MZ_EINVAL; }
    DeviceIndex *r = nullptr;
    int rc = replica_for_pointer(ix, d_n_hits, &r);
    if (rc != MZ_OK) return rc;
    return launch_k2u_compact(*r, d_fw_words, n, d_hit_idx, d_hits, d_n_hits, (hipStream_t)stream);
}

// chunk size of the host-pointer pipeline: 16 MB of queries / 32 MB of results per slot stay cache-resident on the host
// (1 M: 2.54, 2 M: 2.69, 4 M: 2.56-2.70, 8 M: 2.54 G k-mers/s end to end into a reused result array; MZ_HOST_CHUNK overrides)
static uint64_t host_chunk_queries() {
    const char *e = std::getenv("MZ_HOST_CHUNK"); // read per call: tests shrink it to walk the multi-chunk path
    long long v = e ? atoll(e) : 0;
    return (uint64_t)(v > 0 ? v : (2ll << 20));
}
} // extern "C"
namespace mz {
// fn(0 .. n_tasks-1), task 0 on the calling thread and the others on threads of their own (a staging copy of a few megabytes
// lasts a fraction of a millisecond; a pool of kept threads measured no faster than starting them)
void parallel_for(int n_tasks, const std::function<void(int)> &fn) {
    std::vector<std::thread> th;
    int on_threads = 1; // tasks [1, on_threads) got a thread; the others run here (a thread that cannot be started is no error)
    try {
        th.reserve(n_tasks > 1 ? n_tasks - 1 : 0);
        for (int t = 1; t < n_tasks; ++t) { th.emplace_back([&fn, t] { fn(t); }); on_threads = t + 1; }
    } catch (...) {
    }
    if (n_tasks > 0) fn(0);
    for (int t = on_threads; t < n_tasks; ++t) fn(t);
    for (auto &x : th) x.join();
}
// copies with a few host threads (one memcpy stream is ~10 GB/s, the PCIe link wants ~50)
void parallel_copy(void *dst, const void *src, uint64_t bytes, int max_threads) {
    const uint64_t kMin = 2ull << 20;
    const int nt = (int)std::min<uint64_t>(std::max(1, std::min<int>(hw_threads(), max_threads)), bytes / kMin);
    if (nt <= 1) { std::memcpy(dst, src, bytes); return; }
    const uint64_t per = ((bytes + nt - 1) / nt + 4095) & ~4095ull;
    parallel_for(nt, [=](int t) {
        const uint64_t b = std::min<uint64_t>(bytes, per * t), e = std::min<uint64_t>(bytes, b + per);
        if (b < e) std::memcpy((char *)dst + b, (const char *)src + b, e - b);
    });
}
} // namespace mz
extern "C" {

static int pipe_acquire(mz_sshash *ix, uint64_t want, HostPipe **out) {
    HostPipe *p = nullptr;
    {
        std::lock_guard<std::mutex> g(ix->mu);
        for (size_t j = 0; j < ix->idle_pipes.size(); ++j)
            if (ix->idle_pipes[j]->device == ix->dev->device) { p = ix->idle_pipes[j]; ix->idle_pipes.erase(ix->idle_pipes.begin() + j); break; }
    }
    if (!p) { p = new HostPipe; p->device = ix->dev->device; }
    if (p->cap < want) { // (re)allocate for the larger chunk, in powers of two so that growing batch sizes settle quickly
        uint64_t r = 4096;
        while (r < want) r <<= 1;
        want = r;
        p->release();
        hipError_t e = hipSuccess;
        bool ok = true;
        for (int j = 0; j < kPipeSlots && ok; ++j)
            ok = (e = hipStreamCreateWithFlags(&p->stream[j], hipStreamNonBlocking)) == hipSuccess && // own streams: concurrent host threads do not serialise
                 (e = hipEventCreateWithFlags(&p->done[j], hipEventDisableTiming)) == hipSuccess &&
                 (e = hipHostMalloc((void **)&p->h_in[j], want * 8, hipHostMallocDefault)) == hipSuccess &&
                 (e = hipHostMalloc((void **)&p->h_out[j], want * sizeof(mz_k2upos), hipHostMallocDefault)) == hipSuccess &&
                 (e = hipMalloc((void **)&p->d_in[j], want * 8)) == hipSuccess && (e = hipMalloc((void **)&p->d_out[j], want * sizeof(mz_k2upos))) == hipSuccess;
        ok = ok && (e = hipMalloc((void **)&p->d_counts, sizeof(mz_counts))) == hipSuccess &&
             (e = hipHostMalloc((void **)&p->h_counts, sizeof(mz_counts), hipHostMallocDefault)) == hipSuccess;
        if (!ok) {
            set_error("staging buffers for %llu queries: %s", (unsigned long long)want, hipGetErrorString(e));
            p->release();
            delete p;
            return MZ_ENOMEM;
        }
        p->cap = want;
    }
    *out = p;
    return MZ_OK;
}
static void pipe_release(mz_sshash *ix, HostPipe *p) {
    std::lock_guard<std::mutex> g(ix->mu);
    ix->idle_pipes.push_back(p);
}


// Drives `n_chunks` chunks through the slots of a staging pipe.  stage(c, slot) fills the slot's pinned input and puts upload,
// kernel and download on the slot's stream (MZ_OK or an error code, message set); drain(c, slot) copies the slot's pinned
// results to the caller.  The tally is cleared first and copied to p->h_counts behind the last chunk.  Calls of up to two
// chunks run on the calling thread alone (stage c, then drain c - 1).  Longer ones drain on a helper thread, so that the
// calling thread stages chunk c while the GPU holds chunk c - 1 and the helper empties chunk c - 2: the link then carries
// uploads and downloads at the same time and neither host copy waits for the other.
struct PipeCopyThreads { int stage, drain; };
static PipeCopyThreads pipe_copy_threads(bool threaded) {
    const int hw = hw_threads();
    if (!threaded) return {8, 8};
    return {std::max(2, std::min(8, hw / 3)), std::max(2, std::min(12, hw - hw / 3 - 1))}; // results are twice the bytes of the queries
}
} // extern "C" (a template needs C++ linkage)
template <class Stage, class Drain>
static int run_pipe(HostPipe *p, uint64_t n_chunks, const char *what, Stage stage, Drain drain) {
    hipError_t e = hipMemsetAsync(p->d_counts, 0, sizeof(mz_counts), p->stream[0]);
    if (e == hipSuccess) e = hipEventRecord(p->done[0], p->stream[0]);
    for (int j = 1; j < kPipeSlots && (uint64_t)j < n_chunks && e == hipSuccess; ++j) e = hipStreamWaitEvent(p->stream[j], p->done[0], 0); // every stream in use starts after the tally is cleared
    int rc = MZ_OK;
    // stage + the pipe's own bookkeeping behind a chunk
    auto put = [&](uint64_t c) -> int {
        const int slot = (int)(c % kPipeSlots);
        hipStream_t st = p->stream[slot];
        int r = stage(c, slot);
        if (r != MZ_OK) return r;
        hipError_t pe = hipSuccess;
        if (c + 1 == n_chunks) { // the tally is complete once the last kernels of the other slots are done too
            for (int j = 1; j < kPipeSlots && pe == hipSuccess; ++j)
                if (c >= (uint64_t)j) pe = hipStreamWaitEvent(st, p->done[(c - j) % kPipeSlots], 0);
            if (pe == hipSuccess) pe = hipMemcpyAsync(p->h_counts, p->d_counts, sizeof(mz_counts), hipMemcpyDeviceToHost, st);
        }
        if (pe == hipSuccess) pe = hipEventRecord(p->done[slot], st);
        if (pe != hipSuccess) { set_error("HIP error in %s: %s", what, hipGetErrorString(pe)); return MZ_EHIP; }
        return MZ_OK;
    };
    auto take = [&](uint64_t c) -> int {
        const int slot = (int)(c % kPipeSlots);
        hipError_t te = hipEventSynchronize(p->done[slot]);
        if (te != hipSuccess) { set_error("HIP error in %s: %s", what, hipGetErrorString(te)); return MZ_EHIP; }
        drain(c, slot);
        return MZ_OK;
    };
    if (e != hipSuccess) {
        set_error("HIP error in %s: %s", what, hipGetErrorString(e));
        rc = MZ_EHIP;
    } else {
        std::mutex m;
        std::condition_variable cv;
        uint64_t staged = 0, drained = 0; // chunks put on the GPU / copied out, guarded by m
        int rc_drain = MZ_OK;
        bool stop = false;
        std::string drain_error;
        std::thread helper;
        auto helper_body = [&] {
            (void)hipSetDevice(p->device);
            for (uint64_t c = 0; c < n_chunks; ++c) {
                {
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] { return staged > c || stop; });
                    if (staged <= c) return; // the stager gave up
                }
                const int r = take(c);
                std::lock_guard<std::mutex> lk(m);
                if (r != MZ_OK) { rc_drain = r; drain_error = mz_last_error(); stop = true; cv.notify_all(); return; } // the message is thread-local: hand it over
                drained = c + 1;
                cv.notify_all();
            }
        };
        bool threaded = n_chunks > 2 && hw_threads() >= 4;
        if (threaded) {
            try { helper = std::thread(helper_body); } catch (...) { threaded = false; } // no thread to be had: the calling thread does both
        }
        if (!threaded) {
            for (uint64_t c = 0; c <= n_chunks && rc == MZ_OK; ++c) {
                if (c < n_chunks) rc = put(c);
                if (rc == MZ_OK && c >= 1) rc = take(c - 1);
            }
        }
        for (uint64_t c = 0; threaded && c < n_chunks && rc == MZ_OK; ++c) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return drained + kPipeSlots > c || stop; }); // the slot's previous chunk has left it
                if (stop) break;
            }
            rc = put(c);
            std::lock_guard<std::mutex> lk(m);
            if (rc != MZ_OK) stop = true; else staged = c + 1;
            cv.notify_all();
        }
        if (helper.joinable()) helper.join();
        if (rc == MZ_OK && rc_drain != MZ_OK) { rc = rc_drain; set_error("%s", drain_error.c_str()); }
    }
    if (rc != MZ_OK)
        for (int j = 0; j < kPipeSlots; ++j) (void)hipStreamSynchronize(p->stream[j]); // nothing may still be in flight when the pipe is reused
    return rc;
}
extern "C" {

// Host pointers in, host pointers out, through the staging pipe (run_pipe): a chunk is uploaded, looked up and downloaded on its
// slot's stream while its neighbours are staged and drained by the host.
int mz_sshash_k2u_batch(mz_sshash *ix, const uint64_t *fw_words, uint64_t n, mz_k2upos *out, mz_counts *counts) {
    DeviceGuard guard;
    if (!ix || (n && (!fw_words || !out))) { set_error("null argument"); return MZ_EINVAL; }
    int rc = ensure_device(ix);
    if (rc != MZ_OK) return rc;
    if (counts) *counts = mz_counts{0, 0, 0};
    if (n == 0) return MZ_OK;
    HIPC(hipSetDevice(ix->dev->device));
    const uint64_t chunk = std::min<uint64_t>(n, host_chunk_queries());
    HostPipe *p = nullptr;
    if ((rc = pipe_acquire(ix, chunk, &p)) != MZ_OK) return rc;
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    const PipeCopyThreads ct = pipe_copy_threads(n_chunks > 2);
    auto span = [&](uint64_t c, uint64_t &b, uint64_t &len) { b = c * chunk; len = std::min<uint64_t>(chunk, n - b); };
    rc = run_pipe(p, n_chunks, "k2u_batch",
        [&](uint64_t c, int slot) -> int {
            uint64_t b, len;
            span(c, b, len);
            hipStream_t st = p->stream[slot];
            parallel_copy(p->h_in[slot], fw_words + b, len * 8, ct.stage);
            hipError_t e = hipMemcpyAsync(p->d_in[slot], p->h_in[slot], len * 8, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) {
                const int r = launch_k2u_batch(*ix->dev, ix->qopts, p->d_in[slot], len, p->d_out[slot], p->d_counts, st);
                if (r != MZ_OK) return r;
                e = hipMemcpyAsync(p->h_out[slot], p->d_out[slot], len * sizeof(mz_k2upos), hipMemcpyDeviceToHost, st);
            }
            if (e != hipSuccess) { set_error("HIP error in k2u_batch: %s", hipGetErrorString(e)); return MZ_EHIP; }
            return MZ_OK;
        },
        [&](uint64_t c, int slot) {
            uint64_t b, len;
            span(c, b, len);
            parallel_copy(out + b, p->h_out[slot], len * sizeof(mz_k2upos), ct.drain);
        });
    if (rc == MZ_OK && counts) *counts = *p->h_counts;
    pipe_release(ix, p);
    return rc;
}

int mz_sshash_k2u(mz_sshash *ix, uint64_t fw_word, uint32_t k, mz_k2upos *out) {
    if (!ix || !out) { set_error("null argument"); return MZ_EINVAL; }
    if (k != ix->host.us.k) { // assert_eq!(km.k, self.k()), sshash.rs:473
        set_error("Got query k-mer size k=%u, expected k=%u.", k, ix->host.us.k);
        return MZ_EINVAL;
    }
    return mz_sshash_k2u_batch(ix, &fw_word, 1, out, nullptr);
}

uint64_t mz_reads_n_windows(const uint64_t *read_offsets, uint64_t n_reads, uint32_t k) {
    uint64_t t = 0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        uint64_t len = read_offsets[r + 1] - read_offsets[r];
        if (len >= k) t += len - k + 1;
    }
    return t;
}

int mz_sshash_k2u_reads_device(mz_sshash *ix, const uint8_t *d_bases, uint64_t n_bases, const uint64_t *d_read_offsets,
                               const uint64_t *d_kmer_offsets, uint64_t n_reads, int streaming, mz_k2upos *d_out,
                               mz_counts *d_counts, void *stream) {
    DeviceGuard guard;
    if (!ix || (n_reads && (!d_bases || !d_read_offsets || !d_kmer_offsets || !d_out))) { set_error("null argument"); return MZ_EINVAL; }
    DeviceIndex *r = nullptr;
    int rc = replica_for_pointer(ix, d_read_offsets, &r);
    if (rc != MZ_OK) return rc;
    return launch_k2u_reads(*r, ix->qopts, d_bases, n_bases, d_read_offsets, d_kmer_offsets, n_reads, streaming, d_out, d_counts,
                            (hipStream_t)stream);
}

// Reads through the staging pipe (run_pipe): a chunk is a run of whole reads whose windows fit the result slot and whose
// bases + offset tables fit the query slot.  Slot layout: [read offsets (nr+1)][k-mer offsets (nr+1)][bases], rebased to
// the chunk.  Returns MZ_EUNSUPPORTED_INTERNAL when a single read is too long for a chunk (the caller then takes the
// unchunked path).
static const int MZ_EUNSUPPORTED_INTERNAL = -1000;
static int k2u_reads_pipelined(mz_sshash *ix, const uint8_t *bases, const uint64_t *read_offsets, const uint64_t *ko, uint64_t n_reads,
                               int streaming, mz_k2upos *out, mz_counts *counts) {
    // chunk capacity in windows: the configured chunk, or less when the whole call is smaller
    const uint64_t all_bytes = (read_offsets[n_reads] - read_offsets[0]) + 16 * (n_reads + 1) + 64;
    const uint64_t cap = std::max<uint64_t>(1, std::min<uint64_t>(host_chunk_queries(), std::max<uint64_t>(ko[n_reads], (all_bytes + 7) / 8)));
    const uint64_t cap_in = cap * 8;
    // chunk boundaries
    std::vector<uint64_t> cut{0};
    for (uint64_t r0 = 0; r0 < n_reads;) {
        uint64_t r1 = r0;
        while (r1 < n_reads) {
            const uint64_t win = ko[r1 + 1] - ko[r0], bytes = (read_offsets[r1 + 1] - read_offsets[r0]) + 16 * (r1 + 1 - r0 + 1) + 64;
            if (win > cap || bytes > cap_in) break;
            ++r1;
        }
        if (r1 == r0) return MZ_EUNSUPPORTED_INTERNAL;
        cut.push_back(r1);
        r0 = r1;
    }
    HostPipe *p = nullptr;
    int rc = pipe_acquire(ix, cap, &p);
    if (rc != MZ_OK) return rc;
    const uint64_t n_chunks = cut.size() - 1;
    const PipeCopyThreads ct = pipe_copy_threads(n_chunks > 2);
    rc = run_pipe(p, n_chunks, "k2u_reads",
        [&](uint64_t c, int slot) -> int {
            hipStream_t st = p->stream[slot];
            const uint64_t r0 = cut[c], r1 = cut[c + 1], nr = r1 - r0, nb = read_offsets[r1] - read_offsets[r0];
            uint64_t *h_ro = p->h_in[slot], *h_ko = h_ro + (nr + 1);
            uint8_t *h_b = reinterpret_cast<uint8_t *>(h_ko + (nr + 1));
            for (uint64_t i = 0; i <= nr; ++i) { h_ro[i] = read_offsets[r0 + i] - read_offsets[r0]; h_ko[i] = ko[r0 + i] - ko[r0]; }
            parallel_copy(h_b, bases + read_offsets[r0], nb, ct.stage);
            const uint64_t in_bytes = 16 * (nr + 1) + nb, n_win = ko[r1] - ko[r0];
            uint64_t *d_ro = p->d_in[slot], *d_ko = d_ro + (nr + 1);
            const uint8_t *d_b = reinterpret_cast<const uint8_t *>(d_ko + (nr + 1));
            hipError_t e = hipMemcpyAsync(p->d_in[slot], p->h_in[slot], in_bytes, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) {
                const int r = launch_k2u_reads(*ix->dev, ix->qopts, d_b, nb, d_ro, d_ko, nr, streaming, p->d_out[slot], p->d_counts, st);
                if (r != MZ_OK) return r;
                if (n_win) e = hipMemcpyAsync(p->h_out[slot], p->d_out[slot], n_win * sizeof(mz_k2upos), hipMemcpyDeviceToHost, st);
            }
            if (e != hipSuccess) { set_error("HIP error in k2u_reads: %s", hipGetErrorString(e)); return MZ_EHIP; }
            return MZ_OK;
        },
        [&](uint64_t c, int slot) {
            const uint64_t r0 = cut[c], r1 = cut[c + 1], n_win = ko[r1] - ko[r0];
            if (n_win) parallel_copy(out + ko[r0], p->h_out[slot], n_win * sizeof(mz_k2upos), ct.drain);
        });
    if (rc == MZ_OK && counts) *counts = *p->h_counts;
    pipe_release(ix, p);
    return rc;
}

int mz_sshash_k2u_reads(mz_sshash *ix, const uint8_t *bases, const uint64_t *read_offsets, uint64_t n_reads, int streaming,
                        mz_k2upos *out, mz_counts *counts) {
    DeviceGuard guard;
    if (!ix || (n_reads && (!bases || !read_offsets))) { set_error("null argument"); return MZ_EINVAL; }
    int rc = ensure_device(ix);
    if (rc != MZ_OK) return rc;
    if (counts) *counts = mz_counts{0, 0, 0};
    if (n_reads == 0) return MZ_OK;
    const uint32_t k = ix->host.us.k;
    std::vector<uint64_t> ko(n_reads + 1, 0);
    for (uint64_t r = 0; r < n_reads; ++r) {
        if (read_offsets[r + 1] < read_offsets[r]) { set_error("read_offsets not monotone at %llu", (unsigned long long)r); return MZ_ENOTMONOTONE; }
        uint64_t len = read_offsets[r + 1] - read_offsets[r];
        ko[r + 1] = ko[r] + (len >= k ? len - k + 1 : 0);
    }
    const uint64_t base0 = read_offsets[0], n_bases = read_offsets[n_reads] - base0, n_win = ko[n_reads];
    if (n_win && !out) { set_error("null out"); return MZ_EINVAL; }
    std::vector<uint64_t> ro(n_reads + 1);
    for (uint64_t r = 0; r <= n_reads; ++r) ro[r] = read_offsets[r] - base0;
    HIPC(hipSetDevice(ix->dev->device));
    { // pipelined form (same staging pipe as mz_sshash_k2u_batch) when every read fits one chunk; else the plain form below
        int prc = k2u_reads_pipelined(ix, bases, read_offsets, ko.data(), n_reads, streaming, out, counts);
        if (prc != MZ_EUNSUPPORTED_INTERNAL) return prc;
    }
    uint8_t *d_b = nullptr;
    uint64_t *d_ro = nullptr, *d_ko = nullptr;
    mz_k2upos *d_o = nullptr;
    mz_counts *d_c = nullptr;
    hipStream_t s = nullptr;
    HIPC(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto cleanup = [&] {
        if (d_b) (void)hipFree(d_b);
        if (d_ro) (void)hipFree(d_ro);
        if (d_ko) (void)hipFree(d_ko);
        if (d_o) (void)hipFree(d_o);
        if (d_c) (void)hipFree(d_c);
        (void)hipStreamDestroy(s);
    };
    hipError_t e;
    if ((e = hipMalloc((void **)&d_b, std::max<uint64_t>(n_bases, 1))) != hipSuccess ||
        (e = hipMalloc((void **)&d_ro, (n_reads + 1) * 8)) != hipSuccess || (e = hipMalloc((void **)&d_ko, (n_reads + 1) * 8)) != hipSuccess ||
        (e = hipMalloc((void **)&d_o, std::max<uint64_t>(n_win, 1) * sizeof(mz_k2upos))) != hipSuccess ||
        (e = hipMalloc((void **)&d_c, sizeof(mz_counts))) != hipSuccess) {
        cleanup();
        set_error("hipMalloc failed: %s", hipGetErrorString(e));
        return MZ_ENOMEM;
    }
    rc = MZ_OK;
    if ((n_bases && (e = hipMemcpyAsync(d_b, bases + base0, n_bases, hipMemcpyHostToDevice, s)) != hipSuccess) ||
        (e = hipMemcpyAsync(d_ro, ro.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, s)) != hipSuccess ||
        (e = hipMemcpyAsync(d_ko, ko.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, s)) != hipSuccess ||
        (e = hipMemsetAsync(d_c, 0, sizeof(mz_counts), s)) != hipSuccess) rc = MZ_EHIP;
    if (rc == MZ_OK) rc = launch_k2u_reads(*ix->dev, ix->qopts, d_b, n_bases, d_ro, d_ko, n_reads, streaming, d_o, d_c, s);
    if (rc == MZ_OK && n_win && (e = hipMemcpyAsync(out, d_o, n_win * sizeof(mz_k2upos), hipMemcpyDeviceToHost, s)) != hipSuccess) rc = MZ_EHIP;
    mz_counts hc{0, 0, 0};
    if (rc == MZ_OK && (e = hipMemcpyAsync(&hc, d_c, sizeof hc, hipMemcpyDeviceToHost, s)) != hipSuccess) rc = MZ_EHIP;
    if (rc == MZ_OK && (e = hipStreamSynchronize(s)) != hipSuccess) rc = MZ_EHIP;
    if (rc == MZ_EHIP && e != hipSuccess) set_error("HIP error in k2u_reads: %s", hipGetErrorString(e));
    if (rc == MZ_OK && counts) *counts = hc;
    cleanup();
    return rc;
}

int mz_sshash_validate_self_duplicates(mz_sshash *ix, mz_validate_report *report, uint64_t *dup_pos, uint64_t cap, uint64_t *n_listed) {
    DeviceGuard guard;
    if (!ix || !report || (cap && !dup_pos)) { set_error("null argument"); return MZ_EINVAL; }
    int rc = ensure_device(ix);
    if (rc != MZ_OK) return rc;
    HIPC(hipSetDevice(ix->dev->device));
    uint64_t init[5] = {0, 0, 0, ~0ull, 0}, *d_r = nullptr; // report words followed by the duplicate list
    HIPC(hipMalloc((void **)&d_r, sizeof init + cap * 8));
    hipError_t e = hipMemcpy(d_r, init, sizeof init, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = launch_validate_self(*ix->dev, d_r, cap ? d_r + 5 : nullptr, cap, nullptr);
        if (rc == MZ_OK) e = hipMemcpy(init, d_r, sizeof init, hipMemcpyDeviceToHost); // synchronises with the null stream
        if (rc == MZ_OK && e == hipSuccess && cap && init[4]) e = hipMemcpy(dup_pos, d_r + 5, std::min<uint64_t>(init[4], cap) * 8, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_r);
    if (e != hipSuccess) { set_error("HIP error in validate_self: %s", hipGetErrorString(e)); return MZ_EHIP; }
    if (rc != MZ_OK) return rc;
    report->n_kmers = init[0];
    report->n_fail = init[1];
    report->n_duplicate = init[2];
    report->first_fail_pos = init[3];
    if (n_listed) *n_listed = std::min<uint64_t>(init[4], cap);
    return MZ_OK;
}
int mz_sshash_validate_self(mz_sshash *ix, mz_validate_report *report) { return mz_sshash_validate_self_duplicates(ix, report, nullptr, 0, nullptr); }

int mz_sshash_line_stats(mz_sshash *ix, mz_line_stats *out) {
    if (!ix || !out) { set_error("null argument"); return MZ_EINVAL; }
    std::memset(out, 0, sizeof(*out));
    if (!ix->host.front.present) return MZ_OK;
    DeviceGuard guard;
    int rc = ensure_device(ix);
    if (rc != MZ_OK) return rc;
    HIPC(hipSetDevice(ix->dev->device));
    uint64_t *d = nullptr;
    HIPC(hipMalloc((void **)&d, 8 * 8));
    uint64_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipError_t e = hipMemset(d, 0, 8 * 8);
    if (e == hipSuccess) { rc = launch_line_stats(*ix->dev, d, nullptr); e = hipMemcpy(h, d, 8 * 8, hipMemcpyDeviceToHost); }
    (void)hipFree(d);
    if (rc != MZ_OK) return rc;
    if (e != hipSuccess) { set_error("HIP error %d (%s) in mz_sshash_line_stats", (int)e, hipGetErrorString(e)); return MZ_EHIP; }
    out->n_lines = ix->dev->view.front.n_lines;
    out->g = ix->host.front.g; out->t = ix->host.front.t;
    out->entries_total = ix->host.pos.n;
    out->entries_inline = h[0]; out->buckets_spilled = h[1]; out->entries_spilled = h[2]; out->buckets_heavy = h[3];
    out->entries_heavy = ix->host.pos.n - h[0] - h[2]; // (a heavy marker's own count saturates at 65535: what is neither inline nor spilled is heavy)
    out->slots_empty = h[5];
    out->buckets_walk = h[7];
    if (front_is_hashed(ix->host.w)) { // what is neither inline nor behind a marker sits in lines the builder gave up (heavy counts saturate per marker: lower bound)
        out->lines_unusable = h[6];
        out->entries_heavy = h[4];
        out->entries_unusable = ix->host.pos.n - h[0] - h[2] - h[4];
    }
    return MZ_OK;
}

int mz_sshash_extract_kmers_device(mz_sshash *ix, const uint64_t *d_positions, const uint8_t *d_flip, uint64_t n,
                                   uint64_t *d_fw_words, void *stream) {
    DeviceGuard guard;
    if (!ix || (n && (!d_positions || !d_fw_words))) { set_error("null argument"); return MZ_EINVAL; }
    DeviceIndex *r = nullptr;
    int rc = replica_for_pointer(ix, d_fw_words, &r);
    if (rc != MZ_OK) return rc;
    return launch_extract_kmers(*r, d_positions, d_flip, n, d_fw_words, (hipStream_t)stream);
}

int mz_sshash_save(const mz_sshash *ix_, const char *path) {
    if (!ix_ || !path) { set_error("null argument"); return MZ_EINVAL; }
    mz_sshash *ix = const_cast<mz_sshash *>(ix_); // a GPU-built index fetches its arrays from HBM on first use (logically const)
    DeviceGuard guard;
    {
        std::lock_guard<std::mutex> g(ix->mu);
        int rc = host_arrays(ix);
        if (rc != MZ_OK) return rc;
    }
    return ix->host.save(path);
}
int mz_sshash_load(const char *path, mz_sshash **out) {
    if (!path || !out) { set_error("null argument"); return MZ_EINVAL; }
    *out = nullptr;
    mz_sshash *ix = new (std::nothrow) mz_sshash();
    if (!ix) return MZ_ENOMEM;
    int rc;
    try {
        rc = ix->host.load(path);
    } catch (const std::bad_alloc &) {
        set_error("out of host memory while loading %s", path);
        rc = MZ_ENOMEM;
    } catch (const std::exception &e) {
        set_error("%s: %s", path, e.what());
        rc = MZ_EIO;
    }
    if (rc != MZ_OK) { delete ix; return rc; }
    *out = ix;
    return MZ_OK;
}

// ------------------------------------------------------------ StreamingK2U
int mz_streaming_new(mz_sshash *ix, mz_streaming **out) {
    if (!ix || !out) { set_error("null argument"); return MZ_EINVAL; }
    *out = new (std::nothrow) mz_streaming{ix};
    return *out ? MZ_OK : MZ_ENOMEM;
}
void mz_streaming_free(mz_streaming *st) { delete st; }

// index/caching.rs:65-103.  The warm step compares the query with the k-mer that follows the previous hit on its unitig,
// read from the handle's own packed unitig sequence (the UnitigSet the reference's StreamingK2U borrows through
// AsRef<UnitigSet>, caching.rs:43-46,84-87): two host words, no device call.  Everything else is a cold lookup on the GPU.
int mz_streaming_k2u(mz_streaming *st, uint64_t fw_word, uint32_t k, mz_k2upos *out) {
    DeviceGuard guard;
    if (!st || !out) { set_error("null argument"); return MZ_EINVAL; }
    mz_sshash *ix = st->ix;
    if (k != ix->host.us.k) { set_error("Got query k-mer size k=%u, expected k=%u.", k, ix->host.us.k); return MZ_EINVAL; }
    int rc = MZ_OK;
    fw_word &= kmer_mask(k);
    if (st->is_warm) { // k2u_warm, :73-97
        const uint64_t next_pos = (uint64_t)st->prev.pos + 1, last_km_pos = (uint64_t)st->prev.unitig_len - k;
        if (next_pos <= last_km_pos) {
            const uint64_t gp = ix->host.us.starts[st->prev.unitig_id] + next_pos;
            const uint64_t kw = read_kmer(ix->host.us.useq.data(), gp, k);
            const uint32_t mt = word_equivalency(fw_word, kmer_rc(fw_word, k), kw);
            if (mt != MZ_NO_MATCH) {
                st->prev.pos = (uint32_t)next_pos;
                st->prev.match = mt;
                *out = st->prev;
                return MZ_OK;
            }
        }
    }
    // k2u_cold, :99-103 (`?`: a miss leaves the state untouched)
    rc = mz_sshash_k2u_batch(ix, &fw_word, 1, out, nullptr);
    if (rc == MZ_OK && out->match != MZ_NO_MATCH) {
        st->prev = *out;
        st->is_warm = true;
    }
    return rc;
}

} // extern "C"
