/*
 * snacc_hip.h -- C-ABI of libsnacc_hip.so: the MI355X (gfx950) backend for the
 * all-pairs lz4 normalized-compression-distance hot path of alexsweeten/snacc.
 *
 * The reference has no FFI; its hot path is Python calling the third-party
 * codec once per file / ordered file pair:
 *
 *   ref:snacc/pairwise_ncd.py:69,80,90   bytes(seq) -> lz4framed.compress -> getsizeof
 *   ref:snacc/cli.py:108-116             phase A: N single compressions
 *   ref:snacc/cli.py:120-129             phase B: N*N ordered-pair compressions
 *
 * This library replaces exactly those calls with batched ones.  It returns raw
 * LZ4-frame lengths (liblz4 1.9.3 LZ4F_compressFrame, NULL preferences, bit
 * exact); the "+33" of sys.getsizeof and the float64 NCD formula
 * (ref:snacc/pairwise_ncd.py:93-111) stay in Python.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only; no C++/torch types cross the ABI.
 *  - every function returns 0 on success or a negative SNK_E_* code; it never
 *    throws or aborts.  snk_last_error() gives a message for the last failure.
 *  - the caller owns every host buffer; the library never keeps a host pointer
 *    after a call returns.  The library owns its device memory.
 *  - one snk_ctx is used from one thread at a time; contexts are independent.
 *  - streams: snk_pairs_device calls on different streams of one context overlap when every resident
 *    sequence is pure ACGT (dense tiles share nothing).  When a resident 2-bit sequence has exceptions
 *    (N runs, IUPAC codes, stretches in the other case) every 2-bit launch uses the context's ONE set of
 *    per-chain overflow tables (16 KiB per resident chain: 344 MB on a 256-CU device, allocated at the
 *    first such launch): such launches wait for each other on the device, whatever their streams.
 *    Launches that need a job list (mixed tiles, pair lists) share one list buffer per context and
 *    wait, on the host, for the previous launch that read it.
 *  - there is NO CPU fallback: without a HIP device every entry point that
 *    computes fails with SNK_E_HIP.
 */
#ifndef SNACC_HIP_H
#define SNACC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNK_ABI_VERSION 1

enum {
    SNK_OK        = 0,
    SNK_E_ARG     = -1,   /* bad argument                                   */
    SNK_E_HIP     = -2,   /* HIP runtime error (no device, OOM, launch ...) */
    SNK_E_STATE   = -3,   /* call order (e.g. pairs before upload)          */
    SNK_E_TOOBIG  = -4,   /* a sequence or a concatenation >= 0x7E000000 B  */
    SNK_E_KERNEL  = -5,   /* device-side consistency check failed           */
    SNK_E_EMPTY   = -6,   /* FASTA ingest: no sequence in the file (Python: ValueError)        */
    SNK_E_MIXED   = -7    /* FASTA ingest: T and U in one record under reverse complement      */
};

typedef struct snk_ctx snk_ctx;

/* ABI version of the loaded library (SNK_ABI_VERSION). */
int snk_version(void);

/* Message of the last failure on `ctx` (or of the last failed snk_ctx_create
 * when ctx == NULL).  Valid until the next call on the same context. */
const char *snk_last_error(const snk_ctx *ctx);

/* Create a context bound to HIP device `device` (index as seen by this process). */
int snk_ctx_create(int device, snk_ctx **out);
void snk_ctx_destroy(snk_ctx *ctx);

/* Tunables (all optional).  Keys:
 *   "fast_lanes"    chains (lanes) per wavefront in the 2-bit ACGT kernel; 0 (default) = as many as the LDS holds
 *   "fast_waves"    wavefronts per workgroup in the 2-bit ACGT kernel
 *   "bytes_lanes", "bytes_waves", "cbytes_*", "c2bytes_*"  the same for the byte kernels
 *                   (full table / compact 1024 slots / compact 2048 slots)
 *   "bytes_compact" -1 auto (default) / 0 never: compact table of the byte kernel when the
 *                   resident sequences use <= 2048 distinct 5-byte hashes (set before upload)
 *   "bytes_spec"    0 (default) / 1: LDS byte kernels on the slot stream with two lanes per chain (the second lane probes
 *                   5 bytes ahead and counts when the first lane's match ends there; exact).  Measured without gain
 *                   (DESIGN.md section 6.0); kept for experiments.
 *   "bytes_gt"      byte kernels with their tables in global memory, one chain per lane: waves per workgroup (1..8);
 *                   0 = tables in LDS; -1 (default) = 4 for the full table once the launch has more jobs than two rounds
 *                   of the LDS kernel, LDS otherwise (the compact tables are faster in LDS).  "bytes_gt_wgs": workgroups
 *                   per CU and launch (default 1; the host launches ceil(jobs / capacity) times, one table per chain)
 *   "bytes_legacy"  1 = linked-mode byte jobs use the legacy u32-table kernel (testing)
 *   "exc_limit"     a sequence with bytes other than ACGT (acgt in a set that is mostly lower case) -- N runs, IUPAC codes,
 *                   stretches in the other case -- stays on the 2-bit kernel while it has at most 4 + 1.25 * exc_limit RUNS of
 *                   such bytes per 2^20 bases (default 2048: 2564 runs per Mbp -- beyond that the byte kernels are faster);
 *                   how many bases the runs cover does not matter (round 4: soft-masked genomes run faster on the 2-bit kernel
 *                   at every density); 0 = pure ACGT only.  Set before snk_upload.
 *   "fast_dynamic"  -1 auto / 0 static round robin / 1 atomic queue: how the waves of the 2-bit kernel take their batches
 *   "far_lanes", "far_waves", "far_min", "far_stop_pct"  chains of the 2-bit kernel beyond the LDS (extra waves whose chains
 *                   keep their tables in global memory; far_lanes 0 = none, the default).  A measured negative kept for
 *                   reproduction (profiles/r03_far_chains.json): leave it off.  far_min: only launches of at least that many
 *                   jobs per LDS chain of the card use them (default 4); far_stop_pct: far waves take no new jobs once fewer
 *                   than this percentage of (LDS chains of the launch) jobs are left (default 140)
 *   "force_generic" 1 = route every pair through the byte kernel (testing)
 *   "deflate_serial" 1 = gzip / zlib: parse every single sequence with one wavefront from start to end
 *                   instead of in stitched parallel segments (testing; set before the first deflate call)
 *   "deflate_kmer"  0 = gzip / zlib: search every hash chain in full instead of first trying the members that
 *                   share six bytes with the probe (testing; same results)
 *   "deflate_norestart" 1 = gzip / zlib: a pair job parses x from its first byte instead of restarting from x's
 *                   stored stream shortly before the seam (testing; same results)
 *   "content_size"  1 = add the 8-byte content-size field to every frame
 *                   (py-lz4framed builds that set it)
 *   "fast_asm"      1 (default) = the hand-scheduled steady loop of the 2-bit kernel; 0 = its C++ statement (cross-checks)
 *   "fast_spec"     1 (default) = two lanes per chain in the 2-bit kernel's steady loop (snk_fast_steady_spec: the second
 *                   lane probes 5 bases ahead; exact); 0 = one lane per chain; 3 / 36 = three lanes per chain as a C++
 *                   statement (a third lane 10 / 6 bases ahead; 20 chains per wave; pure-ACGT sets; exact) -- built and
 *                   measured negative in round 4 (profiles/r04_third_lane.json), kept for reproduction
 *   "split_clean"   (round 4) in a resident set where only SOME 2-bit sequences carry exceptions, the pairs of two sequences without
 *                   any are listed first and run on the pure-ACGT kernel (an exception kernel's loop exits cost four times as much):
 *                   0 never, 1 (default) when those pairs alone fill the card 16 times -- a second launch leaves the card part-empty
 *                   once more --, 2 always (tests).  Sizes are the same either way.
 *   "defer_singles" 1 = snk_upload / snk_upload_fasta leave phase A (single sizes + prefix snapshots) to the calls that
 *                   need it: snk_singles / snk_singles_rows for the rows asked for, the snk_pairs* calls for the rows
 *                   (prefixes) they compute.  A rank of a row-sharded run thus computes its own rows only, and a
 *                   gzip / zlib run (snk_deflate_*) never runs the lz4 pass.  Default 0: phase A of every sequence
 *                   is part of the upload.        */
int snk_set_option(snk_ctx *ctx, const char *key, long value);

/* Replaces the per-task FASTA->bytes hand-off of ref:snacc/pairwise_ncd.py:59-69.
 * Copies `n_seq` byte strings (the concatenated residues of each input file, as
 * extract_sequences returns them) to the device, classifies them (pure
 * upper-case ACGT -> 2-bit packed), and computes every single-sequence frame
 * size plus the per-sequence prefix snapshots the pair kernels start from.
 * A second upload on the same context replaces the first. */
int snk_upload(snk_ctx *ctx, int n_seq, const uint8_t *const *seqs, const uint64_t *lens);

/* Number of sequences resident / how many of them took the 2-bit path / distinct 5-byte hash
 * values in the resident set when the byte kernel can use its compact table (0 otherwise). */
int snk_num_sequences(const snk_ctx *ctx);

/* Lengths (bytes) of the resident sequences, in upload order, into lens[snk_num_sequences()].
 * Host callers shard rows by work with them (snacc_amd/distributed.py).  Returns SNK_OK. */
int snk_lengths(const snk_ctx *ctx, uint64_t *lens);
int snk_num_packed(const snk_ctx *ctx);
/* Chains (ordered pairs in flight) of one workgroup of the 2-bit kernel under the current options and resident set:
 * a row tile of this many rows is a whole number of rounds on every compute unit (bench.py sizes its step by it). */
int snk_fast_chains(snk_ctx *ctx);
int snk_num_compact_hashes(const snk_ctx *ctx);

/* Phase A (ref:snacc/cli.py:108-116): sizes[i] = len(lz4framed.compress(seq_i)). */
int snk_singles(snk_ctx *ctx, uint32_t *sizes /* [n_seq], host */);

/* The same for the sequences [row_begin, row_end): sizes[i - row_begin].  With the option "defer_singles" phase A of
 * these rows (single sizes + the prefix snapshots their pair launches start from) runs now if it has not yet; a rank of
 * a row-sharded run asks for its own rows only and sends the sizes along with its tiles. */
int snk_singles_rows(snk_ctx *ctx, int row_begin, int row_end, uint32_t *sizes /* [row_end - row_begin], host */);

/* Host wall time (ms) of the stages of the last snk_upload / snk_upload_fasta on `ctx`, each ended by a stream
 * synchronisation: ms[0] device arena + host-to-device copies, [1] classification (exception granules, runs),
 * [2] 2-bit pack (+ class arena), [3] hash sets + slot stream of the byte kernels, [4] per-sequence tables,
 * [5] phase A (singles + snapshots; 0 when deferred), [6] the whole call.  Up to `cap` values; returns 7. */
int snk_upload_times(const snk_ctx *ctx, double *ms, int cap);

/* Phase B (ref:snacc/cli.py:120-129) for rows [row_begin, row_end):
 *   sizes[(i-row_begin)*n_seq + j] = len(lz4framed.compress(seq_i + seq_j)).
 * Blocking; result copied to host memory. */
int snk_pairs(snk_ctx *ctx, int row_begin, int row_end, uint32_t *sizes /* host */);

/* Same, asynchronous: launches on `hip_stream` (a hipStream_t passed as void*,
 * NULL = the context's own stream) and writes u32 sizes to DEVICE memory
 * `d_sizes` ((row_end-row_begin)*n_seq elements), e.g. a torch tensor that is
 * then all-gathered over RCCL.  Does not synchronise.  The device-side status
 * word is checked by the next blocking call or by snk_sync().  One call covers at most
 * 0xFFF00000 ordered pairs ((row_end-row_begin)*n_seq; SNK_E_TOOBIG beyond: tile the rows, as
 * snk_pairs does). */
int snk_pairs_device(snk_ctx *ctx, int row_begin, int row_end, void *d_sizes, void *hip_stream);

/* Arbitrary ordered pairs: sizes[t] = len(compress(seq[ij[2t]] + seq[ij[2t+1]])).
 * Used by compressed_size((a, b), "lz4") (one pair per call in the reference). */
int snk_pairs_list(snk_ctx *ctx, int n_pairs, const int32_t *ij, uint32_t *sizes /* host */);

/* The compressed FRAMES themselves (ref:snacc/pairwise_ncd.py:82-88 writes them
 * with -s/--save-compression).  Item t is the single sequence ij[2t] when ij[2t+1] == -1, else
 * the concatenation seq[ij[2t]] + seq[ij[2t+1]].  `offsets` has n_items+1 entries, offsets[t+1] -
 * offsets[t] = the frame size obtained from snk_singles / snk_pairs_list; frame t is written to
 * out[offsets[t] ...].  Bytes equal liblz4 1.9.3 LZ4F_compressFrame(prefs = NULL) or, with the
 * content_size option, LZ4F_compressFrame with frameInfo.contentSize set (tests/golden/frame_hashes.json
 * pins both). */
int snk_frames_list(snk_ctx *ctx, int n_items, const int32_t *ij, const uint64_t *offsets, uint8_t *out);

/* Wait for outstanding work on the context's stream / the given stream and
 * return SNK_E_KERNEL if any launch since the last check flagged an error. */
int snk_sync(snk_ctx *ctx, void *hip_stream);

/* Device time (ms, hipEvent pair around the kernels on their stream) of the
 * last snk_pairs / snk_pairs_device / snk_pairs_list call; for
 * snk_pairs_device it is valid after snk_sync().  Returns < 0 if unavailable. */
double snk_last_pairs_ms(snk_ctx *ctx);

/* Device times (ms) of every snk_pairs_device / snk_pairs_list launch since the previous call
 * of this function, oldest first: up to `cap` values into ms[], returns how many launches were
 * logged (bench.py averages them over its timed steps), or a negative error code -- SNK_E_STATE
 * when a logged launch has not completed yet (call snk_sync first).  Clears the log.  The log is kept
 * from the first call of this function on (a context whose caller never reads it holds one event pair,
 * not one per launch); that first call reports the most recent launch, if any. */
int snk_pairs_ms_log(snk_ctx *ctx, double *ms, int cap);

/* ---- FASTA ingest on the host ------------------------------------------
 * Replaces extract_sequences (ref:snacc/pairwise_ncd.py:15-39: Bio.SeqIO.parse + per-record
 * reverse_complement + concatenation) for the batched path: every file is parsed once, by
 * n_threads host threads, without creating Python strings.  Results are malloc'ed byte strings
 * (free with snk_free).  Errors: SNK_E_ARG (I/O), SNK_E_EMPTY (the reference's "No sequence
 * extracted" ValueError), SNK_E_MIXED (Biopython's "Mixed RNA/DNA found"); message via
 * snk_fasta_last_error() (thread-local). */
int snk_fasta_extract(const char *path, int reverse_complement, uint8_t **out, uint64_t *out_len);
int snk_fasta_extract_many(int n, const char *const *paths, int reverse_complement, int n_threads,
                           uint8_t **outs /* [n] */, uint64_t *lens /* [n] */);
const char *snk_fasta_last_error(void);
void snk_free(void *p);

/* Ingest + snk_upload in one call: the sequence bytes never enter the caller's language runtime. */
int snk_upload_fasta(snk_ctx *ctx, int n, const char *const *paths, int reverse_complement, int n_threads);

/* ---- CSV text of the distance matrix -----------------------------------------
 * The fields of `rows` x `cols` float64 values as the reference's DataFrame.to_csv writes them
 * (ref:snacc/cli.py:138-142: Python's repr of a float -- shortest round-trip digits, exponent form below
 * 1e-4 and from 1e16 on, ".0" on whole numbers, NaN as an empty field), joined by ','.  Row r is written
 * to out + r * stride (stride >= cols * SNK_CSV_FIELD_MAX bytes), len[r] = its length; no terminator, no
 * line end.  Host code only (n_threads host threads); the caller adds the label column and line ends. */
#define SNK_CSV_FIELD_MAX 25
int snk_csv_rows_f64(const double *m, uint64_t rows, uint64_t cols, char *out, uint64_t stride, uint32_t *len, int n_threads);

/* ---- NCD matrix from the integer sizes (ref:snacc/cli.py:131-136, ref:snacc/pairwise_ncd.py:93-111) ------------
 * out[i*n + j] = compute_distance(S_i, S_j, P_ij, P_ji) with S = singles + overhead and P = pairs + overhead
 * (overhead = sys.getsizeof(b"") = 33 plus, for gzip / zlib, nothing: their wrapper bytes are in the sizes already):
 * (min(P_ij, P_ji) - min(S_i, S_j)) / max(S_i, S_j), integer arithmetic and one float64 division -- bit-equal to the
 * reference's Python.  Host code only (n_threads host threads); `pairs` is the n x n row-major array of snk_pairs. */
int snk_ncd_matrix_u32(const uint32_t *singles, const uint32_t *pairs, uint64_t n, uint32_t overhead, double *out, int n_threads);

/* ---- gzip / zlib sizes -----------------------------------------------------
 * Replace, for the batched path, the codec calls of ref:snacc/pairwise_ncd.py:73-74
 * (gzip.compress -> deflate level 9) and :77-78 (zlib.compress -> deflate level 6) on the resident
 * sequences.  `level` is 9 or 6.  Sizes are the bytes of the RAW deflate stream zlib 1.2.11 writes
 * (memLevel 8, 32 KiB window, default strategy), bit exact; the caller adds the wrapper (gzip 18 B,
 * zlib 6 B) and sys.getsizeof's 33 where those semantics live (Python).
 *
 * snk_deflate_prepare builds, once per upload, the per-sequence match index and, once per level, the
 * symbol stream of every single sequence; the other calls do it implicitly.  Layout of the results
 * as for snk_singles / snk_pairs / snk_pairs_list (item (i, -1) of a list = sequence i alone). */
int snk_deflate_prepare(snk_ctx *ctx, int level);
int snk_deflate_singles(snk_ctx *ctx, int level, uint32_t *sizes /* [n_seq], host */);
int snk_deflate_pairs(snk_ctx *ctx, int level, int row_begin, int row_end, uint32_t *sizes /* host */);
int snk_deflate_pairs_list(snk_ctx *ctx, int level, int n_pairs, const int32_t *ij, uint32_t *sizes /* host */);
/* Same as snk_deflate_pairs, asynchronous (cf. snk_pairs_device): launches on `hip_stream` (NULL = the context's
 * stream) and writes the u32 raw stream sizes to DEVICE memory d_sizes ((row_end - row_begin) * n_seq elements),
 * e.g. a torch tensor that is then all-gathered over RCCL.  Does not synchronise (snk_sync); one stream at a time. */
int snk_deflate_pairs_device(snk_ctx *ctx, int level, int row_begin, int row_end, void *d_sizes, void *hip_stream);
/* Device time (ms) of the kernels of the last snk_deflate_pairs / _pairs_list / _pairs_device call (waits for the
 * launch of an asynchronous call to finish); < 0 if unavailable. */
double snk_deflate_last_ms(snk_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* SNACC_HIP_H */
