// mz_device.hpp -- device replica of an SSHash index and the kernel launchers (gfx950).
#pragma once
#include <functional>
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <vector>

#include "../../include/mazu_amd.h"
#include "mz_host.hpp"

namespace mz {

struct DeviceIndex {
    int device = -1;
    IndexView view{};          // device pointers
    std::vector<void *> allocs; // everything hipMalloc'ed for this replica
    uint64_t bytes = 0;
    uint32_t n_cu = 256;
    uint64_t n_entries = 0;    // bucket entries of the index (size of `pos`)
};

// restores the caller's current HIP device when an entry point that switched devices returns
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

int device_count();
// copies every array of `h` into HBM of `device`
int device_upload(const SSHashHost &h, int device, DeviceIndex **out);
void device_free(DeviceIndex *d);

// all launchers are asynchronous on `stream`; pointers are device pointers on d.device
int launch_k2u_batch(const DeviceIndex &d, const QueryOpts &opts, const uint64_t *d_fw, uint64_t n, mz_k2upos *d_out, mz_counts *d_counts,
                     hipStream_t stream);
int launch_k2u_compact(const DeviceIndex &d, const uint64_t *d_fw, uint64_t n, uint64_t *d_hit_idx, mz_k2upos *d_hits,
                       uint64_t *d_n_hits, hipStream_t stream);
int launch_k2u_reads(const DeviceIndex &d, const QueryOpts &opts, const uint8_t *d_bases, uint64_t n_bases, const uint64_t *d_read_offsets,
                     const uint64_t *d_kmer_offsets, uint64_t n_reads, int streaming, mz_k2upos *d_out,
                     mz_counts *d_counts, hipStream_t stream);
// d_report: 5 x u64 {n_kmers, n_fail, n_duplicate, first_fail_pos, n_listed}; caller initialises {0,0,0,~0,0}.
// d_dup_list (may be null): receives up to dup_cap global positions of k-mers whose canonical word occurs earlier too
int launch_validate_self(const DeviceIndex &d, uint64_t *d_report, uint64_t *d_dup_list, uint64_t dup_cap, hipStream_t stream);
// d_out: 8 x u64, zeroed by the caller (see line_stats_kernel)
int launch_line_stats(const DeviceIndex &d, uint64_t *d_out, hipStream_t stream);
int launch_extract_kmers(const DeviceIndex &d, const uint64_t *d_positions, const uint8_t *d_flip, uint64_t n,
                         uint64_t *d_fw, hipStream_t stream);

// index construction on the device (mz_build_gpu.hip); false / MZ_EINVAL when the configuration needs the host builder
bool gpu_build_supported(const UnitigSet &us, uint32_t w);
// keep != nullptr: the arrays stay in HBM as the replica *keep and h.arrays_on_host is false (see device_download)
int build_on_device(SSHashHost &h, const UnitigSet &unitigs, uint32_t w, uint64_t skew_param, uint64_t seed, int device, DeviceIndex **keep = nullptr,
                    const BuildOpts &opts = BuildOpts());
// copies the index arrays of a replica into the (empty) host vectors of `h`; afterwards h.arrays_on_host is true
int device_download(const DeviceIndex &d, SSHashHost &h);

} // namespace mz

// device of the replica the host-pointer entry points of `ix` use (placed on device 0 first if the index is not resident yet)
int mz_internal_default_device(mz_sshash *ix, int *device);
namespace mz {
// memcpy split over up to `max_threads` host threads (pageable <-> pinned staging copies of the host-pointer entry points)
void parallel_copy(void *dst, const void *src, uint64_t bytes, int max_threads = 8);
// fn(0 .. n_tasks-1) on the library's helper threads and the calling thread; returns when all have run
void parallel_for(int n_tasks, const std::function<void(int)> &fn);
} // namespace mz
