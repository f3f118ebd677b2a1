// mz_u2pos.hip -- the step right after k2u: unitig occurrences -> reference positions on the device.
// Mirrors reference src/index/dense_unitig_table.rs:55-76 (DenseUnitigTable as U2Pos, pufferfish 64-bit occurrence
// codec of src/index.rs:315-346), :110-153 (PiscemUnitigTable as U2Pos, occurrences packed into the minimal number of
// bits, src/spt_compact.rs:83-112) and src/index.rs:184-216 (project_onto_u_occ).  Variable-length output:
// count -> exclusive scan -> fill, all on the device.
#include <hip/hip_runtime.h>

#include <atomic>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>

#include "mz_device.hpp"

using namespace mz;

// how an occurrence is stored: element j at bit j * width of `words`; width 64 + piscem == 0 is the pufferfish word
// [63] = forward, [62:32] = position, [31:0] = reference (index.rs:334-346); piscem != 0 is
// (ref << ref_shift) | (pos << 1) | forward (spt_compact.rs:83-112)
struct OccCodec {
    const uint64_t *words;
    uint32_t width, ref_shift, piscem, pad;
    uint64_t pos_mask;
};
struct DecodedOcc { uint32_t ref_id, pos; bool fwd; };

// Staging of one chunk of mz_get_ref_pos_batch: pinned host buffers and their device twins, a stream, and the scan's tile sums.
struct RefPosSlot {
    hipStream_t stream = nullptr;
    hipEvent_t offsets_ready = nullptr;
    uint64_t *h_q = nullptr, *d_q = nullptr, *h_off = nullptr, *d_off = nullptr, *d_tiles = nullptr;
    mz_k2upos *d_recs = nullptr;
    mz_mapped_ref_pos *h_out = nullptr, *d_out = nullptr;
    uint64_t out_cap = 0; // records h_out / d_out hold
    uint64_t fit = 0, first_record = 0; // of the chunk in flight: records the caller has room for, and where they go
    void release_out() {
        if (h_out) (void)hipHostFree(h_out);
        if (d_out) (void)hipFree(d_out);
        h_out = d_out = nullptr; out_cap = 0;
    }
    void release() {
        release_out();
        if (h_q) (void)hipHostFree(h_q);
        if (h_off) (void)hipHostFree(h_off);
        if (d_q) (void)hipFree(d_q);
        if (d_off) (void)hipFree(d_off);
        if (d_tiles) (void)hipFree(d_tiles);
        if (d_recs) (void)hipFree(d_recs);
        if (offsets_ready) (void)hipEventDestroy(offsets_ready);
        if (stream) (void)hipStreamDestroy(stream);
        *this = RefPosSlot{};
    }
};
struct RefPosPipe {
    int device = -1;
    uint64_t cap = 0; // queries per chunk
    static constexpr int kSlots = 3; // chunk c + 2 is staged while the offsets of chunk c + 1 and the records of chunk c come back
    RefPosSlot slot[kSlots];
    void release() { for (RefPosSlot &s : slot) s.release(); cap = 0; }
};

struct mz_u2pos {
    std::vector<RefPosPipe *> idle_pipes; // guarded by mu; a call of mz_get_ref_pos_batch takes one, concurrent callers get their own
    std::mutex mu;
    std::vector<uint64_t> ctable, offsets; // host copies: occurrence words (packed for piscem, + 2 pad words) and plain unitig offsets
    uint64_t n_unitigs = 0, n_occs = 0;
    uint32_t width = 64, ref_shift = 0, piscem = 0;
    uint64_t pos_mask = 0;
    int device = -1;
    uint64_t *d_ctable = nullptr, *d_offsets = nullptr;
    OccCodec codec() const { return OccCodec{d_ctable, width, ref_shift, piscem, 0, pos_mask}; }
};

namespace {

constexpr int kB = 256, kItems = 4, kTileN = kB * kItems;

#define U2_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            set_error("HIP error %d (%s): %s", (int)_e, hipGetErrorString(_e), #expr);     \
            return MZ_EHIP;                                                                \
        }                                                                                  \
    } while (0)

unsigned grid_of(uint64_t n, unsigned cap = 256 * 16) { return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + kB - 1) / kB, cap)); }

// occurrence j of the table: UnitigOcc::decode_pf1 (index.rs:334-346) or decode_piscem (spt_compact.rs:97-112)
__device__ __forceinline__ DecodedOcc decode_occ(const OccCodec &c, uint64_t j) {
    if (c.piscem) {
        const uint64_t e = read_packed(c.words, j, c.width);
        return DecodedOcc{(uint32_t)(e >> c.ref_shift), (uint32_t)((e >> 1) & c.pos_mask), (e & 1) != 0};
    }
    const uint64_t e = c.words[j];
    const uint32_t pw = (uint32_t)(e >> 32);
    return DecodedOcc{(uint32_t)(e & 0xFFFFFFFFull), pw & 0x7FFFFFFFu, (pw & 0x80000000u) != 0};
}
// project_onto_u_occ (index.rs:184-216)
__device__ __forceinline__ mz_mapped_ref_pos project(const mz_k2upos &r, const DecodedOcc &oc, uint32_t k, uint64_t query) {
    const uint32_t pos = oc.fwd ? r.pos + oc.pos : oc.pos + (r.unitig_len - r.pos) - k;
    uint32_t o = r.match == MZ_IDENTITY_MATCH ? 1u : 0u;             // Forward = 1, Backward = 0 (lib.rs:50-64)
    if (!oc.fwd) o = 1u - o;
    return mz_mapped_ref_pos{oc.ref_id, pos, o, (uint32_t)query};
}

__global__ __launch_bounds__(kB) void count_kernel(const mz_k2upos *__restrict__ recs, uint64_t n, const uint64_t *__restrict__ offsets,
                                                   uint64_t n_unitigs, uint64_t *__restrict__ counts) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n; i += stride) {
        mz_k2upos r = recs[i];
        uint64_t c = 0;
        if ((r.match == MZ_IDENTITY_MATCH || r.match == MZ_TWIN_MATCH) && r.unitig_id < n_unitigs)
            c = offsets[r.unitig_id + 1] - offsets[r.unitig_id];
        counts[i] = c;
    }
}

// ---- exclusive scan of u64 counts (three small kernels; in place) ----
__device__ __forceinline__ uint64_t block_exclusive(uint64_t v, uint64_t *total) {
    __shared__ uint64_t wsum[kB / 64];
    uint64_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t o = __shfl_up(incl, d);
        if ((int)(threadIdx.x & 63) >= d) incl += o;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (int w = 0; w < kB / 64; ++w) {
        if (w < (int)(threadIdx.x >> 6)) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}
__global__ __launch_bounds__(kB) void scan_tile_sums(const uint64_t *__restrict__ counts, uint64_t n, uint64_t *__restrict__ tile_sums) {
    uint64_t base = (uint64_t)blockIdx.x * kTileN, s = 0;
    for (int q = 0; q < kItems; ++q) {
        uint64_t i = base + (uint64_t)threadIdx.x * kItems + q;
        if (i < n) s += counts[i];
    }
    uint64_t tot;
    block_exclusive(s, &tot);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(kB) void scan_tile_offsets(uint64_t *tile_sums, uint64_t n_tiles, uint64_t *grand_total) {
    uint64_t carry = 0;
    for (uint64_t b0 = 0; b0 < n_tiles; b0 += kB) { // one workgroup walks the tile sums
        uint64_t i = b0 + threadIdx.x;
        uint64_t v = i < n_tiles ? tile_sums[i] : 0, tot;
        uint64_t ex = block_exclusive(v, &tot);
        if (i < n_tiles) tile_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}
__global__ __launch_bounds__(kB) void scan_apply(uint64_t *__restrict__ counts, uint64_t n, const uint64_t *__restrict__ tile_sums,
                                                 const uint64_t *grand_total) {
    uint64_t base = (uint64_t)blockIdx.x * kTileN;
    uint64_t v[kItems], s = 0;
    for (int q = 0; q < kItems; ++q) {
        uint64_t i = base + (uint64_t)threadIdx.x * kItems + q;
        v[q] = i < n ? counts[i] : 0;
        s += v[q];
    }
    uint64_t tot;
    uint64_t ex = block_exclusive(s, &tot) + tile_sums[blockIdx.x];
    for (int q = 0; q < kItems; ++q) {
        uint64_t i = base + (uint64_t)threadIdx.x * kItems + q;
        if (i < n) counts[i] = ex;
        ex += v[q];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) counts[n] = *grand_total; // offsets[n] = total
}

// fill: one lane per query for short lists; a query with >= 64 occurrences is walked by its whole wave
__global__ __launch_bounds__(kB) void fill_kernel(const mz_k2upos *__restrict__ recs, uint64_t n, const uint64_t *__restrict__ table_off,
                                                  OccCodec ctable, const uint64_t *__restrict__ out_off, uint32_t k,
                                                  mz_mapped_ref_pos *__restrict__ out, uint64_t cap, uint64_t query_base) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    uint64_t n_round = (n + 63) & ~63ull;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n_round; i += stride) {
        mz_k2upos r = mz_k2upos{0, 0, 0, MZ_NO_MATCH};
        uint64_t o0 = 0, c = 0, t0 = 0;
        if (i < n) {
            r = recs[i];
            o0 = out_off[i];
            c = out_off[i + 1] - o0;
            if (c) t0 = table_off[r.unitig_id];
        }
        const bool longl = c >= 64;
        if (c && !longl)
            for (uint64_t j = 0; j < c; ++j)
                if (o0 + j < cap) out[o0 + j] = project(r, decode_occ(ctable, t0 + j), k, query_base + i);
        unsigned long long lb = __ballot(longl);
        while (lb) { // wave-cooperative walk of the long lists of this wave's 64 queries
            int src = __ffsll((long long)lb) - 1;
            lb &= lb - 1;
            mz_k2upos rr;
            rr.unitig_id = __shfl(r.unitig_id, src); rr.pos = __shfl(r.pos, src);
            rr.unitig_len = __shfl(r.unitig_len, src); rr.match = __shfl(r.match, src);
            uint64_t so = __shfl(o0, src), sc = __shfl(c, src), st = __shfl(t0, src), sq = __shfl(i, src);
            for (uint64_t j = __lane_id(); j < sc; j += 64)
                if (so + j < cap) out[so + j] = project(rr, decode_occ(ctable, st + j), k, query_base + sq);
        }
    }
}

// Validate::validate_self (src/index/validate.rs:24-52) on the device: thread per reference k-mer; its k2u
// record must project onto (ref_id, pos) through at least one occurrence of its unitig.
__global__ __launch_bounds__(kB) void check_refs_kernel(const mz_k2upos *__restrict__ recs, uint64_t n, const uint64_t *__restrict__ table_off,
                                                        OccCodec ctable, uint64_t n_unitigs, uint32_t k, uint32_t ref_id,
                                                        unsigned long long *report /* n, n_fail, first_fail */) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    unsigned fails = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n; i += stride) {
        mz_k2upos r = recs[i];
        if (r.match == MZ_SKIPPED) continue; // window with a non-ACGT byte: CanonicalKmerIterator never yields it (validate.rs:57)
        bool found = false;
        if ((r.match == MZ_IDENTITY_MATCH || r.match == MZ_TWIN_MATCH) && r.unitig_id < n_unitigs) {
            uint64_t a = table_off[r.unitig_id], b = table_off[r.unitig_id + 1];
            for (uint64_t j = a; j < b && !found; ++j) {
                mz_mapped_ref_pos m = project(r, decode_occ(ctable, j), k, i);
                found = m.ref_id == ref_id && m.pos == (uint32_t)i;
            }
        }
        if (!found) { ++fails; atomicMin(&report[2], (unsigned long long)i); }
    }
    for (int off = 32; off > 0; off >>= 1) fails += __shfl_down(fails, off);
    if ((threadIdx.x & 63) == 0 && fails) atomicAdd(&report[1], (unsigned long long)fails);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&report[0], (unsigned long long)n);
}


// ---- scratch of the entry point that allocates nothing (mz_u2pos_project_device_noalloc) ----------------------------
// The scan of the per-query counts needs one partial sum per tile of kTileN queries plus the grand total.  The existing
// entry point takes them from the stream-ordered allocator; callers that may not allocate (code under graph capture, tight
// serving loops) get them from this file-scope array instead.  Every GPU the library runs on holds its own copy of the
// array, at its own address, until the process ends: 128 KiB + 16 bytes.
constexpr uint64_t kNoAllocMaxQueries = 1ull << 24;                       // 16 Mi queries per call
constexpr uint64_t kNoAllocMaxTiles = kNoAllocMaxQueries / kTileN;       // 16384 tiles
static __device__ __align__(256) uint64_t scratch[kNoAllocMaxTiles + 2];
constexpr int kMaxDevices = 64;
std::atomic<uint64_t *> g_scratch_ptr[kMaxDevices]; // device address of `scratch` on each GPU, resolved by mz_u2pos_to_device (zero-initialised: static storage)
std::mutex g_scratch_mu;

// device address of GPU `device`'s copy of `scratch`.  Every GPU holds its own copy at its own address and
// hipGetSymbolAddress answers for the CURRENT device, so the device is made current here, whatever the caller left current, and
// the address is checked to belong to it before it is kept.
int resolve_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) { set_error("device %d beyond the %d devices the no-allocation path tracks", device, kMaxDevices); return MZ_EINVAL; }
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (g_scratch_ptr[device].load(std::memory_order_acquire)) return MZ_OK;
    DeviceGuard guard;
    U2_HIP(hipSetDevice(device));
    uint64_t *p = nullptr;
    U2_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
    if (!p) { set_error("hipGetSymbolAddress returned a null address for the scratch array"); return MZ_EHIP; }
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess) {
        if (attr.device != device) { set_error("scratch array resolved on device %d, wanted %d", attr.device, device); return MZ_EHIP; }
    } else {
        (void)hipGetLastError(); // (a runtime that does not classify symbol addresses: the device was made current above)
    }
    g_scratch_ptr[device].store(p, std::memory_order_release);
    return MZ_OK;
}
// Calls of the no-allocation entry point share their GPU's one array: a call whose kernels start while those of the previous one are
// still running on ANOTHER stream would corrupt that call's scan.  An event is recorded behind each call's last kernel, and a call
// on a different stream makes its stream wait for it first: calls on one GPU are then ordered on the device whatever streams and
// host threads they come from (no allocation: the event is created by mz_u2pos_to_device).  Not under stream capture, where events
// from outside the graph cannot be waited for: a captured graph is serialised by its owner.
struct NoAllocInFlight { hipEvent_t ev = nullptr; hipStream_t stream = nullptr; bool armed = false; };
NoAllocInFlight g_noalloc_busy[kMaxDevices];
std::mutex g_noalloc_mu[kMaxDevices];

__global__ void zero_word_kernel(uint64_t *p) { *p = 0; }

// count -> exclusive scan -> fill on `stream`, tile sums in d_tiles (n_tiles + 2 words); shared by both entry points
// (`query_base` is added to the query index a record carries: a caller that walks a batch in chunks passes the chunk's first query)
void launch_project(const mz_u2pos *u, uint32_t k, const mz_k2upos *d_recs, uint64_t n, uint64_t *d_offsets, mz_mapped_ref_pos *d_out, uint64_t out_cap,
                    uint64_t *d_tiles, hipStream_t stream, uint64_t query_base = 0) {
    const uint64_t n_tiles = (n + kTileN - 1) / kTileN;
    hipLaunchKernelGGL(count_kernel, dim3(grid_of(n)), dim3(kB), 0, stream, d_recs, n, u->d_offsets, u->n_unitigs, d_offsets);
    hipLaunchKernelGGL(scan_tile_sums, dim3((unsigned)n_tiles), dim3(kB), 0, stream, d_offsets, n, d_tiles);
    hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(kB), 0, stream, d_tiles, n_tiles, d_tiles + n_tiles);
    hipLaunchKernelGGL(scan_apply, dim3((unsigned)n_tiles), dim3(kB), 0, stream, d_offsets, n, d_tiles, d_tiles + n_tiles);
    if (d_out && out_cap)
        hipLaunchKernelGGL(fill_kernel, dim3(grid_of(n)), dim3(kB), 0, stream, d_recs, n, u->d_offsets, u->codec(), d_offsets, k, d_out, out_cap, query_base);
}

int ensure_u2pos_device(mz_u2pos *u, int device) {
    if (u->d_ctable && u->device == device) return MZ_OK;
    if (device_count() <= 0) { set_error("no HIP device visible: projection runs on the GPU only"); return MZ_EHIP; }
    U2_HIP(hipSetDevice(device));
    if (u->d_ctable) { (void)hipFree(u->d_ctable); (void)hipFree(u->d_offsets); u->d_ctable = u->d_offsets = nullptr; }
    U2_HIP(hipMalloc((void **)&u->d_ctable, std::max<size_t>(8, u->ctable.size() * 8)));
    U2_HIP(hipMalloc((void **)&u->d_offsets, u->offsets.size() * 8));
    if (!u->ctable.empty()) U2_HIP(hipMemcpy(u->d_ctable, u->ctable.data(), u->ctable.size() * 8, hipMemcpyHostToDevice));
    U2_HIP(hipMemcpy(u->d_offsets, u->offsets.data(), u->offsets.size() * 8, hipMemcpyHostToDevice));
    u->device = device;
    return MZ_OK;
}

int device_of(const void *p) {
    hipPointerAttribute_t a;
    if (p && hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeDevice) return a.device;
    (void)hipGetLastError();
    return 0;
}

} // namespace

extern "C" {

int mz_u2pos_dense_new(const uint64_t *ctable, uint64_t n_occs, const uint64_t *contig_offsets, uint64_t n_unitigs, mz_u2pos **out) {
    if (!out || !contig_offsets || (n_occs && !ctable)) { set_error("null argument"); return MZ_EINVAL; }
    *out = nullptr;
    for (uint64_t i = 0; i < n_unitigs; ++i)
        if (contig_offsets[i + 1] < contig_offsets[i]) { set_error("contig offsets not monotone at %llu", (unsigned long long)i); return MZ_ENOTMONOTONE; }
    if (contig_offsets[n_unitigs] > n_occs) { set_error("contig offsets exceed the occurrence table"); return MZ_EINVAL; }
    mz_u2pos *u = new (std::nothrow) mz_u2pos();
    if (!u) return MZ_ENOMEM;
    u->ctable.assign(ctable, ctable + n_occs);
    u->offsets.assign(contig_offsets, contig_offsets + n_unitigs + 1);
    u->n_unitigs = n_unitigs;
    u->n_occs = n_occs;
    *out = u;
    return MZ_OK;
}

// PiscemUnitigTable (dense_unitig_table.rs:110-153): `ctable_words` holds n_occs occurrences of `occ_width` bits each (the IntVector the
// reference fills in spt_compact.rs:145-160), `contig_offsets_words` n_unitigs + 1 offsets of `offsets_width` bits each (the packed
// IntVector of piscem_index.rs:30-31; 64 = plain words).  The occurrences stay packed in HBM; the offsets are unpacked once here.
int mz_u2pos_piscem_new(const uint64_t *ctable_words, uint64_t n_occs, uint32_t occ_width, uint32_t ref_shift, uint64_t pos_mask,
                        const uint64_t *contig_offsets_words, uint32_t offsets_width, uint64_t n_unitigs, mz_u2pos **out) {
    if (!out || !contig_offsets_words || (n_occs && !ctable_words)) { set_error("null argument"); return MZ_EINVAL; }
    *out = nullptr;
    if (occ_width == 0 || occ_width > 64 || offsets_width == 0 || offsets_width > 64 || ref_shift == 0 || ref_shift >= 64 ||
        occ_width - ref_shift > 32 || ref_shift - 1 > 32) { // reference id and position are 32-bit fields of MappedRefPos here
        set_error("piscem occurrence table: unsupported widths (occ %u, ref_shift %u, offsets %u)", occ_width, ref_shift, offsets_width);
        return MZ_EINVAL;
    }
    mz_u2pos *u = new (std::nothrow) mz_u2pos();
    if (!u) return MZ_ENOMEM;
    u->offsets.resize(n_unitigs + 1);
    {   // offsets: the caller's array is exactly ceil((n+1) * width / 64) words, so the last element is read without the pad word read_packed assumes
        const uint64_t nw = ((n_unitigs + 1) * offsets_width + 63) / 64;
        std::vector<uint64_t> tmp(contig_offsets_words, contig_offsets_words + nw);
        tmp.resize(nw + 2, 0);
        for (uint64_t i = 0; i <= n_unitigs; ++i) u->offsets[i] = read_packed(tmp.data(), i, offsets_width);
    }
    for (uint64_t i = 0; i < n_unitigs; ++i)
        if (u->offsets[i + 1] < u->offsets[i]) { delete u; set_error("contig offsets not monotone at %llu", (unsigned long long)i); return MZ_ENOTMONOTONE; }
    if (u->offsets[n_unitigs] > n_occs) { delete u; set_error("contig offsets exceed the occurrence table"); return MZ_EINVAL; }
    const uint64_t nw = (n_occs * occ_width + 63) / 64;
    u->ctable.assign(ctable_words, ctable_words + nw);
    u->ctable.resize(nw + 2, 0); // pad: a packed read may touch the following word
    u->n_unitigs = n_unitigs;
    u->n_occs = n_occs;
    u->width = occ_width;
    u->ref_shift = ref_shift;
    u->pos_mask = pos_mask;
    u->piscem = 1;
    *out = u;
    return MZ_OK;
}
void mz_u2pos_free(mz_u2pos *u) {
    if (!u) return;
    DeviceGuard guard;
    if (u->d_ctable) { (void)hipSetDevice(u->device); (void)hipFree(u->d_ctable); (void)hipFree(u->d_offsets); }
    for (RefPosPipe *p : u->idle_pipes) { (void)hipSetDevice(p->device); p->release(); delete p; }
    delete u;
}
uint64_t mz_u2pos_n_unitig_occs(const mz_u2pos *u, uint64_t ui) { return ui < u->n_unitigs ? u->offsets[ui + 1] - u->offsets[ui] : 0; }
uint64_t mz_u2pos_num_total_occs(const mz_u2pos *u) { return u->n_occs; }

int mz_u2pos_project_device(mz_u2pos *u, uint32_t k, const mz_k2upos *d_recs, uint64_t n, uint64_t *d_offsets, mz_mapped_ref_pos *d_out,
                            uint64_t out_cap, void *stream_) {
    if (!u || !d_offsets || (n && !d_recs)) { set_error("null argument"); return MZ_EINVAL; }
    if (n > 0xFFFFFFFFull) { set_error("mz_u2pos_project_device: %llu queries; a record's `query` field holds 32 bits: split the batch", (unsigned long long)n); return MZ_EINVAL; }
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard;
    int rc = ensure_u2pos_device(u, device_of(d_offsets));
    if (rc != MZ_OK) return rc;
    U2_HIP(hipSetDevice(u->device));
    uint64_t n_tiles = (n + kTileN - 1) / kTileN;
    uint64_t *d_tiles = nullptr;
    U2_HIP(hipMallocAsync((void **)&d_tiles, (n_tiles + 2) * 8, stream));
    if (n == 0) {
        U2_HIP(hipMemsetAsync(d_offsets, 0, 8, stream));
    } else {
        launch_project(u, k, d_recs, n, d_offsets, d_out, out_cap, d_tiles, stream);
    }
    U2_HIP(hipGetLastError());
    U2_HIP(hipFreeAsync(d_tiles, stream));
    return MZ_OK;
}

int mz_u2pos_to_device(mz_u2pos *u, int device) {
    if (!u) { set_error("null argument"); return MZ_EINVAL; }
    DeviceGuard guard;
    int rc = ensure_u2pos_device(u, device); // (returns early, without touching the current device, when the table is resident already)
    if (rc != MZ_OK) return rc;
    rc = resolve_scratch(device);
    if (rc != MZ_OK) return rc;
    std::lock_guard<std::mutex> busy_lock(g_noalloc_mu[device]);
    if (!g_noalloc_busy[device].ev) { // the in-flight guard's event of this GPU, created here so that the entry point itself creates nothing
        U2_HIP(hipSetDevice(device));
        U2_HIP(hipEventCreateWithFlags(&g_noalloc_busy[device].ev, hipEventDisableTiming));
    }
    return MZ_OK;
}

uint64_t mz_u2pos_noalloc_max_queries(void) { return kNoAllocMaxQueries; }

int mz_u2pos_project_device_noalloc(mz_u2pos *u, uint32_t k, const mz_k2upos *d_recs, uint64_t n, uint64_t *d_offsets, mz_mapped_ref_pos *d_out,
                                    uint64_t out_cap, void *stream_) {
    if (!u || !d_offsets || (n && !d_recs)) { set_error("null argument"); return MZ_EINVAL; }
    if (n > kNoAllocMaxQueries) {
        set_error("mz_u2pos_project_device_noalloc: %llu queries exceed the %llu its fixed scratch array covers; split the batch or use mz_u2pos_project_device",
                  (unsigned long long)n, (unsigned long long)kNoAllocMaxQueries);
        return MZ_EINVAL;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const int device = device_of(d_offsets);
    uint64_t *d_tiles = device >= 0 && device < kMaxDevices ? g_scratch_ptr[device].load(std::memory_order_acquire) : nullptr;
    if (!u->d_ctable || u->device != device || !d_tiles) {
        set_error("mz_u2pos_project_device_noalloc: the table is not resident on device %d; call mz_u2pos_to_device(u, %d) once beforehand", device, device);
        return MZ_EINVAL;
    }
    DeviceGuard guard;
    U2_HIP(hipSetDevice(device));
    // the in-flight guard: none under stream capture (events cannot be queried there; a captured graph is serialised by its owner)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    std::lock_guard<std::mutex> busy_lock(g_noalloc_mu[device]);
    NoAllocInFlight &fl = g_noalloc_busy[device];
    if (cap == hipStreamCaptureStatusNone && fl.armed && fl.stream != stream) U2_HIP(hipStreamWaitEvent(stream, fl.ev, 0));
    if (n == 0) hipLaunchKernelGGL(zero_word_kernel, dim3(1), dim3(1), 0, stream, d_offsets); // a kernel, not a memset node: safe under graph capture
    else launch_project(u, k, d_recs, n, d_offsets, d_out, out_cap, d_tiles, stream);
    U2_HIP(hipGetLastError());
    if (cap == hipStreamCaptureStatusNone && fl.ev) { // (the event is created by mz_u2pos_to_device: nothing is allocated here)
        if (hipEventRecord(fl.ev, stream) == hipSuccess) { fl.stream = stream; fl.armed = true; } else { (void)hipGetLastError(); fl.armed = false; }
    }
    return MZ_OK;
}

int mz_u2pos_check_refs_device(mz_u2pos *u, uint32_t k, const mz_k2upos *d_recs, uint64_t n, uint32_t ref_id, uint64_t *d_report, void *stream_) {
    if (!u || !d_report || (n && !d_recs)) { set_error("null argument"); return MZ_EINVAL; }
    DeviceGuard guard;
    int rc = ensure_u2pos_device(u, device_of(d_report));
    if (rc != MZ_OK) return rc;
    U2_HIP(hipSetDevice(u->device));
    hipLaunchKernelGGL(check_refs_kernel, dim3(grid_of(std::max<uint64_t>(n, 1))), dim3(kB), 0, (hipStream_t)stream_, d_recs, n, u->d_offsets,
                       u->codec(), u->n_unitigs, k, ref_id, (unsigned long long *)d_report);
    U2_HIP(hipGetLastError());
    return MZ_OK;
}

// GetRefPos::get_ref_pos_eager for a batch (src/index.rs:139-179), host buffers: k2u on the GPU, projection on the GPU.
// The batch is walked in chunks through three staging slots (pinned buffers, one stream each).  The number of records of a chunk
// is known only when its offsets are back on the host, so a chunk passes three steps, one per iteration of the host loop and
// each overlapping the other two of its neighbours: (1) upload, lookup, count + scan + fill, offsets back; (2) offsets rebased
// into the caller's array, records requested; (3) records copied to the caller.  A slot's record buffer starts at two records
// per query and grows to what a chunk needs (the fill is then repeated: the offsets stand).
static uint64_t refpos_chunk_queries() {
    const char *e = std::getenv("MZ_REFPOS_CHUNK"); // read per call: tests shrink it to walk the multi-chunk path
    const long long v = e ? atoll(e) : 0;
    return (uint64_t)(v > 0 ? v : (1ll << 20));
}
static int refpos_grow_out(RefPosSlot &s, uint64_t want) {
    if (s.out_cap >= want) return MZ_OK;
    uint64_t r = 4096;
    while (r < want) r <<= 1;
    s.release_out();
    hipError_t e;
    if ((e = hipHostMalloc((void **)&s.h_out, r * sizeof(mz_mapped_ref_pos), hipHostMallocDefault)) != hipSuccess ||
        (e = hipMalloc((void **)&s.d_out, r * sizeof(mz_mapped_ref_pos))) != hipSuccess) {
        s.release_out();
        (void)hipGetLastError();
        set_error("record buffers for %llu MappedRefPos records: %s", (unsigned long long)r, hipGetErrorString(e));
        return MZ_ENOMEM;
    }
    s.out_cap = r;
    return MZ_OK;
}
static int refpos_pipe_acquire(mz_u2pos *u, int device, uint64_t want, RefPosPipe **out) {
    RefPosPipe *p = nullptr;
    {
        std::lock_guard<std::mutex> g(u->mu);
        for (size_t j = 0; j < u->idle_pipes.size(); ++j)
            if (u->idle_pipes[j]->device == device) { p = u->idle_pipes[j]; u->idle_pipes.erase(u->idle_pipes.begin() + j); break; }
    }
    if (!p) { p = new RefPosPipe; p->device = device; }
    if (p->cap < want) {
        uint64_t r = 4096;
        while (r < want) r <<= 1;
        p->release();
        hipError_t e = hipSuccess;
        bool ok = true;
        for (int j = 0; j < RefPosPipe::kSlots && ok; ++j) {
            RefPosSlot &s = p->slot[j];
            ok = (e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking)) == hipSuccess &&
                 (e = hipEventCreateWithFlags(&s.offsets_ready, hipEventDisableTiming)) == hipSuccess &&
                 (e = hipHostMalloc((void **)&s.h_q, r * 8, hipHostMallocDefault)) == hipSuccess &&
                 (e = hipHostMalloc((void **)&s.h_off, (r + 1) * 8, hipHostMallocDefault)) == hipSuccess &&
                 (e = hipMalloc((void **)&s.d_q, r * 8)) == hipSuccess && (e = hipMalloc((void **)&s.d_off, (r + 1) * 8)) == hipSuccess &&
                 (e = hipMalloc((void **)&s.d_tiles, ((r + kTileN - 1) / kTileN + 2) * 8)) == hipSuccess &&
                 (e = hipMalloc((void **)&s.d_recs, r * sizeof(mz_k2upos))) == hipSuccess;
        }
        if (!ok) {
            (void)hipGetLastError();
            set_error("staging buffers for %llu queries: %s", (unsigned long long)r, hipGetErrorString(e));
            p->release();
            delete p;
            return MZ_ENOMEM;
        }
        p->cap = r;
    }
    *out = p;
    return MZ_OK;
}
int mz_get_ref_pos_batch(mz_sshash *ix, mz_u2pos *u, const uint64_t *fw_words, uint64_t n, uint64_t *hit_offsets, mz_mapped_ref_pos *out,
                         uint64_t out_cap, uint64_t *n_out) {
    if (!ix || !u || !hit_offsets || !n_out || (n && !fw_words)) { set_error("null argument"); return MZ_EINVAL; }
    *n_out = 0;
    hit_offsets[0] = 0;
    if (n == 0) return MZ_OK;
    if (n > 0xFFFFFFFFull) { set_error("mz_get_ref_pos_batch: %llu queries; a record's `query` field holds 32 bits: split the batch", (unsigned long long)n); return MZ_EINVAL; }
    if (!out) out_cap = 0;
    DeviceGuard guard;
    int device = 0; // the replica the handle's host-pointer entry points use; the handle is not retargeted
    int rc = mz_internal_default_device(ix, &device);
    if (rc != MZ_OK) return rc;
    if ((rc = ensure_u2pos_device(u, device)) != MZ_OK) return rc;
    U2_HIP(hipSetDevice(device));
    const uint32_t k = mz_sshash_k(ix);
    const uint64_t chunk = std::min<uint64_t>(n, refpos_chunk_queries()), n_chunks = (n + chunk - 1) / chunk;
    RefPosPipe *p = nullptr;
    if ((rc = refpos_pipe_acquire(u, device, chunk, &p)) != MZ_OK) return rc;
    auto span = [&](uint64_t c, uint64_t &b, uint64_t &len) { b = c * chunk; len = std::min<uint64_t>(chunk, n - b); };
    hipError_t e = hipSuccess;
    // upload, lookup, count + scan + fill, offsets back: all on the slot's stream
    auto stage = [&](uint64_t c) -> int {
        RefPosSlot &s = p->slot[c % RefPosPipe::kSlots];
        uint64_t b, len;
        span(c, b, len);
        if (out_cap) { const int r = refpos_grow_out(s, 2 * len); if (r != MZ_OK) return r; }
        parallel_copy(s.h_q, fw_words + b, len * 8, 8);
        if ((e = hipMemcpyAsync(s.d_q, s.h_q, len * 8, hipMemcpyHostToDevice, s.stream)) != hipSuccess) return MZ_EHIP;
        const int r = mz_sshash_k2u_batch_device(ix, s.d_q, len, s.d_recs, nullptr, s.stream);
        if (r != MZ_OK) return r;
        launch_project(u, k, s.d_recs, len, s.d_off, out_cap ? s.d_out : nullptr, out_cap ? s.out_cap : 0, s.d_tiles, s.stream, b);
        if ((e = hipGetLastError()) != hipSuccess) return MZ_EHIP;
        if ((e = hipMemcpyAsync(s.h_off, s.d_off, (len + 1) * 8, hipMemcpyDeviceToHost, s.stream)) != hipSuccess) return MZ_EHIP;
        if ((e = hipEventRecord(s.offsets_ready, s.stream)) != hipSuccess) return MZ_EHIP;
        return MZ_OK;
    };
    uint64_t total = 0; // records of the chunks whose offsets have been taken
    auto take_offsets = [&](uint64_t c) -> int {
        RefPosSlot &s = p->slot[c % RefPosPipe::kSlots];
        uint64_t b, len;
        span(c, b, len);
        if ((e = hipEventSynchronize(s.offsets_ready)) != hipSuccess) return MZ_EHIP;
        const uint64_t n_rec = s.h_off[len];
        s.first_record = total;
        s.fit = total < out_cap ? std::min<uint64_t>(n_rec, out_cap - total) : 0; // records of this chunk the caller has room for
        if (s.fit) {
            if (s.fit > s.out_cap) { // more occurrences than the slot was sized for: larger buffers, the fill again
                const int r = refpos_grow_out(s, s.fit);
                if (r != MZ_OK) return r;
                hipLaunchKernelGGL(fill_kernel, dim3(grid_of(len)), dim3(kB), 0, s.stream, s.d_recs, len, u->d_offsets, u->codec(), s.d_off, k, s.d_out, s.out_cap, b);
                if ((e = hipGetLastError()) != hipSuccess) return MZ_EHIP;
            }
            if ((e = hipMemcpyAsync(s.h_out, s.d_out, s.fit * sizeof(mz_mapped_ref_pos), hipMemcpyDeviceToHost, s.stream)) != hipSuccess) return MZ_EHIP;
        }
        {   // the chunk's offsets rebased onto the batch, while the records travel
            const uint64_t *src = s.h_off + 1, base = total;
            uint64_t *dst = hit_offsets + b + 1;
            const int nt = (int)std::min<uint64_t>(std::max(1, std::min(hw_threads(), 8)), len >> 17);
            auto rebase = [=](uint64_t i0, uint64_t i1) { for (uint64_t i = i0; i < i1; ++i) dst[i] = base + src[i]; };
            const uint64_t per = (len + nt - 1) / std::max(nt, 1);
            if (nt <= 1) rebase(0, len);
            else parallel_for(nt, [=](int t) { rebase(std::min<uint64_t>(len, per * t), std::min<uint64_t>(len, per * (t + 1))); });
        }
        total += n_rec;
        return MZ_OK;
    };
    auto take_records = [&](uint64_t c) -> int {
        RefPosSlot &s = p->slot[c % RefPosPipe::kSlots];
        if (!s.fit) return MZ_OK;
        if ((e = hipStreamSynchronize(s.stream)) != hipSuccess) return MZ_EHIP;
        parallel_copy(out + s.first_record, s.h_out, s.fit * sizeof(mz_mapped_ref_pos), 8);
        return MZ_OK;
    };
    for (uint64_t i = 0; i < n_chunks + 2 && rc == MZ_OK; ++i) {
        if (i < n_chunks) rc = stage(i); // its slot was emptied by take_records(i - 3), one iteration ago
        if (rc == MZ_OK && i >= 1 && i - 1 < n_chunks) rc = take_offsets(i - 1);
        if (rc == MZ_OK && i >= 2) rc = take_records(i - 2);
    }
    if (rc == MZ_EHIP && e != hipSuccess) set_error("HIP error in get_ref_pos_batch: %s", hipGetErrorString(e));
    if (rc != MZ_OK) { for (RefPosSlot &s : p->slot) (void)hipStreamSynchronize(s.stream); (void)hipGetLastError(); }
    else *n_out = total;
    {
        std::lock_guard<std::mutex> g(u->mu);
        u->idle_pipes.push_back(p);
    }
    return rc;
}

} // extern "C"
