/*
 * mazu_amd.h -- C ABI of the MI355X-native SSHash k-mer -> unitig dictionary.
 *
 * This is the drop-in boundary for ONE path of COMBINE-lab/mazu: the `K2U`
 * implementation `SSHash` (reference src/kphf/sshash.rs) behind the trait
 * surface of src/kphf/mod.rs:58-140.  A Rust `impl K2U for GpuSSHash` binds
 * exactly these symbols.  Plain pointers and
 * sizes only; no C++ or torch types.
 *
 * Conventions shared with the reference
 *   - k-mer words: 2 bits per base, A=0 C=1 G=2 T=3, first base in the LSBs
 *     (kmers crate / unitig_set.rs:226-229).  A query is the FORWARD word of a
 *     CanonicalKmer (`km.get_fw_mer()`, sshash.rs:472); `km.swap()` == pass the
 *     reverse complement.
 *   - unitig_id = 0-based order of insertion (line order of .cf_seg,
 *     unitig_set.rs:133-141); pos = 0-based offset of the k-mer's first base on
 *     the stored strand of the unitig, for Identity and Twin alike
 *     (kphf/mod.rs:79-98).
 *   - every query entry point runs on the GPU.  There is no CPU fallback: with
 *     no usable HIP device they return MZ_EHIP.
 *
 * Threading: an index is immutable after construction; all query entry points
 * may be called concurrently from several host threads (kphf/mod.rs:105-109
 * requires Self: Sync).  Streaming state lives in mz_streaming, one per thread
 * (index/caching.rs:65 takes &mut self).
 */
#ifndef MAZU_AMD_H
#define MAZU_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (reference: mazu::Error, src/err.rs:3-31; panics map to MZ_EINVAL) */
#define MZ_OK 0
#define MZ_EINVAL 1       /* wrong k (assert_eq!, sshash.rs:473), w > k (sshash.rs:92), k > 31, null/empty input */
#define MZ_ENOMEM 2
#define MZ_EHIP 3         /* no device / HIP runtime error */
#define MZ_EIO 4          /* Error::IO / InvalidData */
#define MZ_ENOTMONOTONE 5 /* Error::EFNotMonotone (elias_fano.rs:80-82) */
#define MZ_EEMPTY 6       /* Error::EFEmpty (elias_fano.rs:124-131) */

/* ---- MatchType (kmers crate; Orientation mapping in src/lib.rs:50-64) */
#define MZ_NO_MATCH 0u
#define MZ_IDENTITY_MATCH 1u
#define MZ_TWIN_MATCH 2u
#define MZ_SKIPPED 3u /* read window holding a non-ACGT byte: CanonicalKmerIterator skips it */

#define MZ_NO_SKEW_INDEX UINT64_MAX /* from_unitig_set_no_skew_index: skew_param = usize::MAX, sshash.rs:81-84 */

/* K2UPos (kphf/mod.rs:13-19) as a 16-byte record; match == MZ_NO_MATCH encodes `None`. */
typedef struct mz_k2upos {
    uint32_t unitig_id;
    uint32_t pos;
    uint32_t unitig_len;
    uint32_t match;
} mz_k2upos;

/* hit / miss / skipped tallies, as counted by the reference bench loop (bin/kphf/main.rs:283-320) */
typedef struct mz_counts {
    uint64_t n_hit;
    uint64_t n_miss;
    uint64_t n_skipped;
} mz_counts;

/* result of a device-side validate_self (kphf/mod.rs:69-103) */
typedef struct mz_validate_report {
    uint64_t n_kmers;     /* k-mers visited (2 queries each) */
    uint64_t n_fail;      /* assertions that would have fired in the reference */
    uint64_t n_duplicate; /* mismatches explained by a duplicate canonical k-mer in the INPUT (not a cdBG) */
    uint64_t first_fail_pos; /* global position of one failing k-mer, UINT64_MAX if none */
} mz_validate_report;

typedef struct mz_unitig_set mz_unitig_set; /* UnitigSet, unitig_set.rs:30-36 */
typedef struct mz_sshash mz_sshash;         /* SSHash<..>, sshash.rs:39-51 (owns its UnitigSet, :44) */
typedef struct mz_streaming mz_streaming;   /* StreamingK2U, index/caching.rs:12-17 */

/* thread-local description of the last error returned on this thread */
const char *mz_last_error(void);
/* ABI version of this header */
uint32_t mz_abi_version(void);
/* number of visible HIP devices (0 when there is none; never an error) */
int mz_device_count(void);
/* host threads the builder uses by default: hardware concurrency capped by the cgroup CPU quota
 * (override with MZ_NUM_THREADS) */
int mz_host_threads(void);

/* ---------------------------------------------------------------- UnitigSet */
/* UnitigSet::from_seqs(&[String], k), unitig_set.rs:74-106.  ACGT/acgt only (MZ_EINVAL otherwise). */
int mz_unitig_set_from_seqs(const char *const *seqs, uint64_t n_seqs, uint32_t k, mz_unitig_set **out);
/* UnitigSet::from_cf_prefix(prefix), unitig_set.rs:109-165: reads <prefix>.cf_seg and <prefix>.json */
int mz_unitig_set_from_cf_prefix(const char *prefix, mz_unitig_set **out);
/* same, from an already 2-bit packed concatenation: words holds ceil(total_len/32) u64,
 * starts holds n_unitigs+1 prefix sums (the accum_lens of unitig_set.rs:80-88).  Inputs are copied. */
int mz_unitig_set_from_packed(const uint64_t *words, uint64_t total_len, const uint64_t *starts,
                              uint64_t n_unitigs, uint32_t k, mz_unitig_set **out);
void mz_unitig_set_free(mz_unitig_set *us);
uint32_t mz_unitig_set_k(const mz_unitig_set *us);                         /* :247-249 */
uint64_t mz_unitig_set_n_unitigs(const mz_unitig_set *us);                 /* :168-170 */
uint64_t mz_unitig_set_total_len(const mz_unitig_set *us);                 /* :207-209 */
uint64_t mz_unitig_set_n_kmers(const mz_unitig_set *us);                   /* :212-214 */
uint64_t mz_unitig_set_unitig_len(const mz_unitig_set *us, uint64_t i);    /* :178-180 (Elias-Fano get) */
uint64_t mz_unitig_set_unitig_start_pos(const mz_unitig_set *us, uint64_t i); /* :196-199 */
uint64_t mz_unitig_set_pos_to_id(const mz_unitig_set *us, uint64_t pos);   /* :185-187 */
/* borrow of the packed sequence (valid until free): unitig i occupies bases [start,start+len) */
const uint64_t *mz_unitig_set_useq_words(const mz_unitig_set *us);

/* ------------------------------------------------------------------- SSHash */
/* SSHash::from_unitig_set(unitigs, w, skew_param, WyHashState::with_seed(seed)), sshash.rs:405-412.
 * skew_param == MZ_NO_SKEW_INDEX gives from_unitig_set_no_skew_index (:397-403).  The unitig set
 * is copied (the reference takes it by value; callers clone it out of a loaded index,
 * README.md:22).  n_threads <= 0 uses every host core. */
int mz_sshash_from_unitig_set(const mz_unitig_set *us, uint32_t w, uint64_t skew_param, uint64_t seed,
                              int n_threads, mz_sshash **out);
/* same construction with the heavy steps on the GPU `device` (minimizer scan with the window staged in LDS, radix sort,
 * minimizer records, Elias-Fano, entries, skew table): bit-identical arrays to the host builder.  Configurations it
 * does not cover (w > 16, or an occurrence not fitting one 64-bit sort key) are built on the host transparently. */
int mz_sshash_from_unitig_set_device(const mz_unitig_set *us, uint32_t w, uint64_t skew_param, uint64_t seed, int device,
                                     mz_sshash **out);
/* Construction with the layout choices that (w, skew_param) do not fix.  In the reference every construction parameter is an
 * argument (SSHash::from_unitig_set / from_unitig_set_no_skew_index, sshash.rs:397-412; CLI minimizer_size / skew_min,
 * bin/kphf/main.rs:75,178); the choices below are this library's own and never change a result.  Zero-initialise the struct,
 * set struct_size = sizeof(mz_build_opts), set what you want to decide; 0 leaves a choice to the size rules.  opts == NULL is
 * mz_sshash_from_unitig_set with all host threads. */
typedef struct mz_build_opts {
    uint32_t struct_size;
    int32_t device;                 /* >= 0: heavy steps on that GPU, index left resident there (mz_sshash_from_unitig_set_device); < 0: host builder */
    int32_t n_threads;              /* host threads; <= 0: all */
    int32_t minimizer_lines;        /* 1: always build the minimizer lines; -1: never; 0: when the table fits the size rule */
    uint32_t lines_max_load;        /* mean bucket entries per 16-slot line the table may carry (sets g: 3 -> 34 GB, 10 -> 9 GB on a GRCh38-sized set); 0: 3, else 5, else 10 */
    uint64_t direct_bytes_per_key;  /* budget of the direct-address minimizer records, bytes per distinct minimizer; 0: the rule of mz_sshash_minimizer_mphf_kind */
} mz_build_opts;
int mz_sshash_from_unitig_set_opts(const mz_unitig_set *us, uint32_t w, uint64_t skew_param, uint64_t seed, const mz_build_opts *opts,
                                   mz_sshash **out);
/* Kernel choices of the query entry points of one index handle; none changes a result.  Same struct conventions; 0 = the rule.
 * Set them before querying (they are read without a lock); opts == NULL restores the rules.  No counterpart in the reference
 * (its k2u has one implementation); the environment variables of the repository's A/B tools only pre-set a new handle. */
typedef struct mz_query_opts {
    uint32_t struct_size;
    int32_t point_kernel;  /* 1: wave-scoped tiles (the rule); 2: workgroup-scoped tiles; 3: one lane per query */
    int32_t reads_lines;   /* reads resolve through 1: the minimizer lines; -1: the minimizer records; rule: lines from 2^28 bucket entries on */
    int32_t reads_resolve; /* through the lines, 1: every window scans its line (the rule); 2: one lookup per run of windows */
    int32_t reads_kernel;  /* 1: wave-scoped reads kernel (the rule at k = 31, w = 15); 2: workgroup-staged */
    int32_t reads_verify;  /* 1: one sequence-record gather per candidate (the rule); 2: grouped verify-and-extend */
    int32_t reads_grid;    /* workgroups per CU of the wave-scoped reads kernel; rule: 24 */
} mz_query_opts;
int mz_sshash_set_query_opts(mz_sshash *ix, const mz_query_opts *opts);
int mz_sshash_get_query_opts(const mz_sshash *ix, mz_query_opts *opts);
void mz_sshash_free(mz_sshash *ix);

/* trait K2U accessors, kphf/mod.rs:58-67 (impl at sshash.rs:446-469) */
uint32_t mz_sshash_k(const mz_sshash *ix);
uint64_t mz_sshash_n_kmers(const mz_sshash *ix);
uint64_t mz_sshash_sum_unitigs_len(const mz_sshash *ix);
uint64_t mz_sshash_unitig_len(const mz_sshash *ix, uint64_t id);
uint64_t mz_sshash_n_unitigs(const mz_sshash *ix);
/* unitig_seq(id): borrow of the packed words plus the base offset/length of unitig id */
int mz_sshash_unitig_seq(const mz_sshash *ix, uint64_t id, const uint64_t **words, uint64_t *base_offset, uint64_t *len);
/* AsRef<UnitigSet>, sshash.rs:440-444 (borrow; do not free) */
const mz_unitig_set *mz_sshash_as_unitig_set(const mz_sshash *ix);
/* sshash.rs:333-339, :341-358 */
uint32_t mz_sshash_w(const mz_sshash *ix);
uint64_t mz_sshash_skew_param(const mz_sshash *ix);
uint64_t mz_sshash_n_minimizers(const mz_sshash *ix);
uint64_t mz_sshash_n_minimizer_occs(const mz_sshash *ix);
uint64_t mz_sshash_n_kmers_in_skew_index(const mz_sshash *ix);
uint64_t mz_sshash_num_bits(const mz_sshash *ix);
/* form of the minimizer MPHF (the reference's boomphf, sshash.rs:177): 1 = direct-address minimizer records (2w <= 32 and
 * the record table within max(64 MiB, 48 bytes per key, min(1 GiB, 1 KiB per key))), 0 = BBHash levels */
uint32_t mz_sshash_minimizer_mphf_kind(const mz_sshash *ix);
/* 1 when the index carries minimizer lines: membership and the entries of most buckets (sshash.rs:478-495: MPHF probe, two
 * Elias-Fano offsets, position array) answered by one 128-byte gather.  Built when the table (sized by the entries: at
 * most 3, else 5, else 10 per line of 16 slots) stays within max(64 MiB, 64 bytes per entry, min(9 GiB, 1 KiB per entry));
 * mz_build_opts.minimizer_lines / lines_max_load decide otherwise.  2w <= 32: lines addressed by the minimizer word itself
 * (direct-address minimizers only); 2w > 32: by a hash of the word, beside the BBHash levels, which stay the exact path for
 * the ~0.5 % of the words that share a 9-bit fingerprint with another word of their line (mz_line_stats.buckets_walk); reads use
 * them at (k, w) = (31, 17 / 19 / 21). */
uint32_t mz_sshash_has_minimizer_lines(const mz_sshash *ix);
/* what the minimizer lines hold (this layout's part of print_stats, sshash.rs:360-395): where the bucket entries of the index are answered from.
 * All zero for an index without lines.  Counted on the GPU (the index is placed on device 0 if it is nowhere yet). */
typedef struct mz_line_stats {
    uint64_t n_lines;         /* 128-byte lines of 16 slots */
    uint32_t g, t;            /* a minimizer word = tag (t bits) | line index | low (g bits) */
    uint64_t entries_total;   /* bucket entries of the index (n_minimizer_occs) */
    uint64_t entries_inline;  /* ... copied into their key's line: answered by the line gather alone */
    uint64_t buckets_spilled; /* light buckets that did not fit their line ... */
    uint64_t entries_spilled; /* ... and their entries: one more dependent access into the entry array */
    uint64_t buckets_heavy;   /* buckets beyond skew_param: answered by the skew index */
    uint64_t entries_heavy;
    uint64_t slots_empty;
    uint64_t lines_unusable;  /* 2w > 32 only (lines addressed by a hash of the word, g = 0, t = log2 n_lines): lines given up because more than */
    uint64_t entries_unusable; /* 16 words fell into them; entries whose queries take the exact walk through the MPHF (given-up lines and buckets_walk) */
    uint64_t buckets_walk;    /* markers without a size: words that share a fingerprint in their hash-addressed line, or a bucket of 65535 entries and more */
} mz_line_stats;
int mz_sshash_line_stats(mz_sshash *ix, mz_line_stats *out);
/* bytes resident in HBM once uploaded */
uint64_t mz_sshash_device_bytes(const mz_sshash *ix);
/* 1 when the arrays of this index were constructed by the GPU builder (mz_sshash_from_unitig_set_device / _opts with device >= 0), 0 when
 * the host builder made them -- asked for, or because the GPU builder handed the configuration over (it is silent about that: the arrays
 * are bit-identical either way).  No counterpart in the reference (its builder, sshash.rs:86-329, has one form). */
uint32_t mz_sshash_built_on_device(const mz_sshash *ix);

/* place a replica of the index in the HBM of `device` (kept alongside replicas on other devices) and make it
 * the one host-pointer entry points use; done implicitly on device 0 by the first query.  The *_device
 * entry points run on the replica of whichever GPU owns the caller's buffers, uploading it on first use. */
int mz_sshash_to_device(mz_sshash *ix, int device);

/* K2U::k2u(&CanonicalKmer) -> Option<K2UPos>, sshash.rs:471-554.  k must equal the index's k
 * (the reference panics; here MZ_EINVAL).  A one-query GPU launch -- use the batch forms for speed. */
int mz_sshash_k2u(mz_sshash *ix, uint64_t fw_word, uint32_t k, mz_k2upos *out);
/* n independent k2u queries, host buffers (copied over PCIe); counts may be NULL */
int mz_sshash_k2u_batch(mz_sshash *ix, const uint64_t *fw_words, uint64_t n, mz_k2upos *out, mz_counts *counts);
/* same with queries/results/counts already in HBM on the index's device; asynchronous on `stream`
 * (a hipStream_t; NULL = default stream).  d_counts may be NULL; otherwise it is ACCUMULATED into. */
int mz_sshash_k2u_batch_device(mz_sshash *ix, const uint64_t *d_fw_words, uint64_t n, mz_k2upos *d_out,
                               mz_counts *d_counts, void *stream);
/* hits only, compacted per wavefront: d_hit_idx[j] = query index, d_hits[j] = its K2UPos; *d_n_hits
 * (device, zeroed by the caller) receives the count.  Order of records is unspecified. */
int mz_sshash_k2u_batch_compact_device(mz_sshash *ix, const uint64_t *d_fw_words, uint64_t n, uint64_t *d_hit_idx,
                                       mz_k2upos *d_hits, uint64_t *d_n_hits, void *stream);

/* reads: every k-mer window of every read, as the bench loop of bin/kphf/main.rs:299-324 walks them
 * (CanonicalKmerIterator::from_u8_slice; windows with a non-ACGT byte are reported as MZ_SKIPPED).
 * bases = concatenated ASCII reads, read_offsets = n_reads+1 byte offsets.  Window j of read r is
 * written to out[kmer_offsets[r] + j] where kmer_offsets is the prefix sum of max(0, len_r - k + 1);
 * out must hold kmer_offsets[n_reads] records (mz_reads_n_windows computes it).  `streaming` != 0
 * requests the .as_streaming() path (index/caching.rs:65-103).  The batch path is the same either way -- every window gets
 * the answer of its own k2u -- so the flag changes nothing here; that equals the reference's streaming walk for unitig
 * sets whose canonical k-mers are unique (every cdBG).  On a set with repeated k-mers (UnitigSet::from_seqs on arbitrary
 * strings) the reference's warm step stays on the previous unitig where an independent lookup may answer with another
 * copy; mz_streaming_k2u keeps that state machine, one k-mer at a time. */
uint64_t mz_reads_n_windows(const uint64_t *read_offsets, uint64_t n_reads, uint32_t k);
int mz_sshash_k2u_reads(mz_sshash *ix, const uint8_t *bases, const uint64_t *read_offsets, uint64_t n_reads,
                        int streaming, mz_k2upos *out, mz_counts *counts);
int mz_sshash_k2u_reads_device(mz_sshash *ix, const uint8_t *d_bases, uint64_t n_bases, const uint64_t *d_read_offsets,
                               const uint64_t *d_kmer_offsets, uint64_t n_reads, int streaming, mz_k2upos *d_out,
                               mz_counts *d_counts, void *stream);

/* K2U::validate_self / validate_self_parallel, kphf/mod.rs:69-139, run on the device: every k-mer of
 * every unitig is queried in both orientations and compared with (unitig, offset, len, Identity|Twin). */
int mz_sshash_validate_self(mz_sshash *ix, mz_validate_report *report);
/* same pass; additionally lists up to `cap` global base positions of k-mers counted in n_duplicate (the copy the lookup
 * does NOT answer with), in no particular order; *n_listed (may be NULL) receives how many were written.  For inputs that
 * are not a cdBG (UnitigSet::from_seqs on arbitrary strings, unitig_set.rs:74-106) this says where the repeats are. */
int mz_sshash_validate_self_duplicates(mz_sshash *ix, mz_validate_report *report, uint64_t *dup_pos, uint64_t cap, uint64_t *n_listed);

/* forward words of the k-mers at global positions (device arrays); flip[i] != 0 returns the reverse
 * complement instead.  Used to synthesise positive queries without leaving HBM. */
int mz_sshash_extract_kmers_device(mz_sshash *ix, const uint64_t *d_positions, const uint8_t *d_flip, uint64_t n,
                                   uint64_t *d_fw_words, void *stream);

/* flat, versioned, little-endian index file (the reference bincode-serialises its structs,
 * bin/kphf/main.rs:156,193; that format depends on un-vendored crates and is not reproduced) */
int mz_sshash_save(const mz_sshash *ix, const char *path);
int mz_sshash_load(const char *path, mz_sshash **out);

/* ------------------------------------------------------------ U2Pos projection ("next" row: the step after k2u) */
/* MappedRefPos (src/index.rs:26-47) plus the index of the query it belongs to; o: 1 = Forward, 0 = Backward (lib.rs:50-64) */
typedef struct mz_mapped_ref_pos {
    uint32_t ref_id;
    uint32_t pos;
    uint32_t o;
    uint32_t query;
} mz_mapped_ref_pos;
typedef struct mz_u2pos mz_u2pos; /* DenseUnitigTable (src/index/dense_unitig_table.rs:12-76) or PiscemUnitigTable (:110-153) as U2Pos */
/* ctable: pufferfish 64-bit occurrences [63]=fwd [62:32]=pos [31:0]=ref (index.rs:315-346); contig_offsets: n_unitigs+1 */
int mz_u2pos_dense_new(const uint64_t *ctable, uint64_t n_occs, const uint64_t *contig_offsets, uint64_t n_unitigs, mz_u2pos **out);
/* PiscemUnitigTable as U2Pos (src/index/dense_unitig_table.rs:110-153): occurrences packed into occ_width bits each,
 * (ref_id << ref_shift) | (pos << 1) | forward (UnitigOcc::encode_piscem, src/spt_compact.rs:83-112; pos_mask as :134-141),
 * element j at bit j * occ_width of ctable_words; contig_offsets_words: n_unitigs+1 offsets of offsets_width bits each
 * (the packed IntVector of src/index/piscem_index.rs:30-31; pass 64 for plain words).  The projection entry points below
 * work on either kind of table; the occurrences stay packed in HBM. */
int mz_u2pos_piscem_new(const uint64_t *ctable_words, uint64_t n_occs, uint32_t occ_width, uint32_t ref_shift, uint64_t pos_mask,
                        const uint64_t *contig_offsets_words, uint32_t offsets_width, uint64_t n_unitigs, mz_u2pos **out);
void mz_u2pos_free(mz_u2pos *u);
uint64_t mz_u2pos_n_unitig_occs(const mz_u2pos *u, uint64_t ui); /* n_unitig_occs, dense_unitig_table.rs:69-71 */
uint64_t mz_u2pos_num_total_occs(const mz_u2pos *u);             /* :34-36 */
/* project_onto_u_occs (index.rs:184-216) for a batch of K2UPos records in HBM: d_offsets (n+1) receives the exclusive
 * prefix sums of the per-query occurrence counts (d_offsets[n] = total), d_out (may be NULL to only count) the
 * MappedRefPos records of query i at [d_offsets[i], d_offsets[i+1]); records beyond out_cap are not written.
 * n <= 2^32 - 1 (a record's `query` field holds 32 bits; MZ_EINVAL beyond; the same limit holds for mz_get_ref_pos_batch). */
int mz_u2pos_project_device(mz_u2pos *u, uint32_t k, const mz_k2upos *d_recs, uint64_t n, uint64_t *d_offsets,
                            mz_mapped_ref_pos *d_out, uint64_t out_cap, void *stream);
/* place the table in the HBM of `device` now (otherwise done on first use) and prepare that GPU for the entry point below */
int mz_u2pos_to_device(mz_u2pos *u, int device);
/* mz_u2pos_project_device for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  Same arguments and results; the partial sums of its scan live
 * in a fixed array inside the library (128 KiB per GPU, held until the process ends) instead of stream-ordered scratch, and
 * it calls no allocating routine.  Limits: n <= mz_u2pos_noalloc_max_queries() (2^24; MZ_EINVAL beyond), the table must
 * already be resident on the GPU that owns d_offsets (mz_u2pos_to_device; MZ_EINVAL otherwise).  Every call on one GPU uses
 * that GPU's one array.  Calls from different streams or host threads are ordered on the device by the library (each call's
 * stream waits for the previous call's last kernel); calls recorded under stream capture are not -- graphs that contain
 * this entry point and may run at the same time on one GPU race and must be serialised by their owner. */
int mz_u2pos_project_device_noalloc(mz_u2pos *u, uint32_t k, const mz_k2upos *d_recs, uint64_t n, uint64_t *d_offsets,
                                    mz_mapped_ref_pos *d_out, uint64_t out_cap, void *stream);
uint64_t mz_u2pos_noalloc_max_queries(void);
/* the assertion of Validate::validate_self (index/validate.rs:24-52) for the k-mers 0..n of reference ref_id whose K2UPos
 * records are d_recs: d_report[3] = {k-mers checked, failures, first failing position}, accumulated (init {0,0,~0}) */
int mz_u2pos_check_refs_device(mz_u2pos *u, uint32_t k, const mz_k2upos *d_recs, uint64_t n, uint32_t ref_id, uint64_t *d_report,
                               void *stream);
/* GetRefPos::get_ref_pos_eager for a batch (index.rs:139-179), host buffers, walked in chunks through pinned staging slots;
 * hit_offsets[n + 1] = prefix sums of the records per query, out = the records that fit out_cap (may be NULL to only count),
 * *n_out = total records (may exceed out_cap); a record's `query` is the index of its k-mer in fw_words */
int mz_get_ref_pos_batch(mz_sshash *ix, mz_u2pos *u, const uint64_t *fw_words, uint64_t n, uint64_t *hit_offsets,
                         mz_mapped_ref_pos *out, uint64_t out_cap, uint64_t *n_out);

/* ------------------------------------------------------------ StreamingK2U */
/* SSHash::as_streaming(), sshash.rs:435-437 -> StreamingK2U::new, index/caching.rs:47-59 */
int mz_streaming_new(mz_sshash *ix, mz_streaming **out);
void mz_streaming_free(mz_streaming *st);
/* StreamingK2U::k2u_streaming(&mut self, km), index/caching.rs:65-103 */
int mz_streaming_k2u(mz_streaming *st, uint64_t fw_word, uint32_t k, mz_k2upos *out);

#ifdef __cplusplus
}
#endif
#endif /* MAZU_AMD_H */
