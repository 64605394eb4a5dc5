/*
 * aligntools_hip.h -- C ABI of the MI355X (gfx950) alignment shim.
 *
 * This is the drop-in boundary for the dynamic-programming hot path of
 * r3fang/alignTools: the O(l1*l2) matrix fill + pointer traceback of
 *
 *     align_gla               reference src/alignment.h:417-473 (+ :372-412)
 *     align_local_affine      reference src/alignment.h:805-847 (+ :766-800)
 *     align_fit_affine_jump   reference src/alignment.h:596-694 (+ :558-592)
 *     align_overlap           reference src/alignment.h:926-964 (+ :896-922)
 *     edit_dist               reference src/alignment.h:291-315
 *
 * The reference has no FFI of its own (it is one C translation unit); these
 * entry points are what its five main_* drivers (alignment.h:318,476,698,851,
 * 967) call instead of the align_*() functions -- see include/aligntools.h for
 * the reference-shaped single-pair wrappers and INTEGRATION.md for the patch.
 *
 * Plain C: pointers and sizes only, int return (0 = AT_OK, negative = error,
 * text via at_last_error), caller-owned buffers, one handle per thread.
 * There is NO CPU fallback: without a HIP device at_init fails.
 */
#ifndef ALIGNTOOLS_HIP_H
#define ALIGNTOOLS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sub-commands (reference src/main.c:39-43) */
enum { AT_MODE_GLOBAL = 0, AT_MODE_LOCAL = 1, AT_MODE_FIT = 2, AT_MODE_OVERLAP = 3, AT_MODE_EDIT = 4 };

/* traceback ops, emitted END -> START (the order the reference's trace_back_*
 * loops produce characters before strrev, alignment.h:372-412) */
enum { AT_OP_MID = 0,   /* (s1[i-1], s2[j-1]); i--, j--                    */
       AT_OP_LOW = 1,   /* (s1[i-1], '-');     i--                          */
       AT_OP_UPP = 2,   /* ('-', s2[j-1]);     j--                          */
       AT_OP_JUMP = 3   /* ('-', s2[j-1]);     j--   fit jump state         */ };

/* state the traceback starts in (reference LOW/MID/UPP, alignment.h:31-33) */
enum { AT_ST_LOW = 1, AT_ST_MID = 2, AT_ST_UPP = 3 };

enum { AT_OK = 0,
       AT_ERR_ARG = -1,        /* NULL / negative size / unknown mode                  */
       AT_ERR_NODEVICE = -2,   /* no HIP device / HIP runtime error                    */
       AT_ERR_RANGE = -3,      /* |score| could leave the exact int32 range            */
       AT_ERR_DOMAIN = -4,     /* input outside the domain the reference defines       */
       AT_ERR_FIT_ORDER = -5,  /* fit: first sequence longer than the second (:599)    */
       AT_ERR_NOMEM = -6 };

typedef struct at_handle at_handle;

/* Bind to one GPU.  One process (or thread) per GPU: n_devices must be 1.
 * device_ids == NULL selects the current device. */
int at_init(const int *device_ids, int n_devices, at_handle **out);
void at_destroy(at_handle *h);
/* Text of the last error on handle h; h == NULL: of the calling thread's last failure without a handle (at_init). */
const char *at_last_error(const at_handle *h);

/* Scoring block = the reference's opt_t (alignment.h:57-65, defaults :102-114:
 * o=-5 e=-1 m=1 u=-2 j=-10 s=false).  `sites` are 0-based positions on s2 as
 * parsed from the 2nd record's FASTA comment (alignment.h:243-256); only used
 * by AT_MODE_FIT with use_jump.  The reference's inverted junction predicate
 * (alignment.h:659) is reproduced: M->J may open at column j
 * iff (j-1) is NOT listed. */
int at_set_scoring(at_handle *h, int m, int u, int o, int e, int j,
                   int use_jump, const int *sites, int nsites);

/*
 * Align a batch of independent pairs held in HOST memory.
 *   seq_blob          raw sequence bytes (any alphabet; compared by byte
 *                     equality like alignment.h:449).  All-ACGT batches are packed
 *                     2 bits per base, others 8 bits, both on the GPU.
 *   off1/len1, off2/len2   per pair: byte offset and length of s1 and s2
 *   want_traceback    0 = scores and end cells only
 *   out_score[n]      reference return value (align_*: the double, always an
 *                     integer; edit: the distance)
 *   out_end_i/j[n]    cell the traceback starts from (1-based DP coordinates)
 *   out_state[n]      AT_ST_* start state
 *   out_ops, ops_off[n], out_nops[n]
 *                     ops of pair k are written to out_ops[ops_off[k] ..] (at
 *                     most len1+len2 of them), count in out_nops[k].  Exactly
 *                     out_nops[k] bytes of a slot are written; bytes between and
 *                     behind the slots are never touched, and the slots may come
 *                     in any order (they must not overlap).
 * Any out_* may be NULL if not wanted (out_ops/ops_off/out_nops together).
 */
int at_align_batch(at_handle *h, int mode, int64_t npairs,
                   const uint8_t *seq_blob,
                   const int64_t *off1, const int32_t *len1,
                   const int64_t *off2, const int32_t *len2,
                   int want_traceback,
                   int32_t *out_score, int32_t *out_end_i, int32_t *out_end_j, int32_t *out_state,
                   uint8_t *out_ops, const int64_t *ops_off, int32_t *out_nops);

/*
 * Same, with every buffer already resident in DEVICE memory (HBM) and the
 * sequences already packed (at_pack_batch): the entry the batch driver and
 * bench.py use.  Asynchronous on `stream` (a hipStream_t, NULL = default).
 * A handle owns one work counter, one workspace and one flag word: use it with
 * ONE stream at a time (calls queued on the same stream may follow each other
 * without waiting; a call on another stream needs the earlier ones finished --
 * or its own handle, which is how bench.py keeps three launches in flight).
 *   d_seq      packed words; bits = 2 (A,C,G,T -> 0..3, 16 bases per int32,
 *              base k of a sequence in bits [2k%32, 2k%32+1] of word k/16) or
 *              bits = 8 (4 bytes per int32, little endian)
 *   d_woff1/2  per pair WORD offset of s1 / s2 in d_seq
 *   max_len1/2 upper bounds of len1/len2 over the batch (sizes LDS / workspace); a pair
 *              longer than its bound is refused (score INT32_MIN, nops -1), not swept
 *   uniform_shape  non-zero = the caller guarantees len1[k] == max_len1 and
 *              len2[k] == max_len2 for every pair (fixed-length read batches);
 *              enables the packed two-pairs-per-wavefront kernel when the
 *              scores provably fit 16 bits.  0 is always safe: batches of 4096
 *              pairs or more are then checked on the device, and the packed or
 *              the int32 kernel runs accordingly (3 % slower than with the
 *              promise, not 2.5 times).  A pair that breaks
 *              the promise is not swept: it and the pairs sharing its work item
 *              (at most 16; 32 for reads of up to 76 bases) come back with
 *              score INT32_MIN and nops -1.
 */
int at_align_batch_device(at_handle *h, int mode, int64_t npairs,
                          const uint32_t *d_seq, int bits,
                          const int64_t *d_woff1, const int32_t *d_len1,
                          const int64_t *d_woff2, const int32_t *d_len2,
                          int32_t max_len1, int32_t max_len2, int uniform_shape,
                          int want_traceback,
                          int32_t *d_score, int32_t *d_end_i, int32_t *d_end_j, int32_t *d_state,
                          uint8_t *d_ops, const int64_t *d_ops_off, int32_t *d_nops,
                          void *stream);

/*
 * All-vs-all over one read set (BASELINE config "overlap, all-vs-all 50k x 1 kbp reads"): d_woff/d_len
 * describe `nreads` packed reads; work item p in [0, npairs) is the ordered pair (a < b) whose row-major
 * index in the strict upper triangle is first_pair + p, aligned as s1 = read a, s2 = read b.  No per-pair
 * descriptors exist (1.25e9 pairs would need 30 GB of them); outputs are indexed by p.  Ranks of a
 * multi-GPU job take disjoint [first_pair, first_pair + npairs) ranges.
 */
int at_align_allpairs_device(at_handle *h, int mode, int64_t nreads,
                             const uint32_t *d_seq, int bits,
                             const int64_t *d_woff, const int32_t *d_len, int32_t max_len,
                             int64_t first_pair, int64_t npairs, int want_traceback,
                             int32_t *d_score, int32_t *d_end_i, int32_t *d_end_j, int32_t *d_state,
                             uint8_t *d_ops, const int64_t *d_ops_off, int32_t *d_nops,
                             void *stream);

/*
 * All-vs-all with the read set in HOST memory (what `alignTools batch <cmd> --all-vs-all reads.fa` calls): the reads
 * (off[k], len[k] in seq_blob) go up and are packed once; pair p of [first_pair, first_pair + npairs) is aligned as
 * s1 = read a, s2 = read b with (a, b) as in at_align_allpairs_device; outputs are indexed by p - first_pair; pair p's
 * ops go to out_ops[ops_off[p - first_pair] ..] (a slot of len[a] + len[b] bytes).  Not for AT_MODE_FIT (l1 <= l2 is a
 * property of ordered pairs).
 */
int at_align_allpairs(at_handle *h, int mode, int64_t nreads,
                      const uint8_t *seq_blob, const int64_t *off, const int32_t *len,
                      int64_t first_pair, int64_t npairs, int want_traceback,
                      int32_t *out_score, int32_t *out_end_i, int32_t *out_end_j, int32_t *out_state,
                      uint8_t *out_ops, const int64_t *ops_off, int32_t *out_nops);

/*
 * A score threshold for all-vs-all overlap SCORES (at_align_allpairs_device / at_align_allpairs / at_align_allpairs_stream with
 * AT_MODE_OVERLAP and no tracebacks; reads of up to 1 024 bases, ACGT): with enabled != 0 a pair whose score -- the value of
 * align_overlap, alignment.h:926-964 -- is PROVEN to lie below min_score is not swept: its state is 0, its score an upper bound
 * (< min_score) and its end_j 0.  Every other pair (state 2) carries the exact results as without the threshold, whether its score
 * reaches min_score or not.  The proof is an upper bound from the bit-parallel edit distance of the pair with a free start in s1
 * (csrc/at_myers.hip.h): 2 score <= 2 m b - (2 c - m) D'(l1, b), c = min(m - u, m / 2 - o); it needs m >= 0 and 2 c > m (the default
 * scoring of `alignTools overlap`, m = 1 u = -2 o = -5, gives 2 b - 5 D'), otherwise every pair is swept.  enabled = 0 (the default)
 * sweeps every pair.
 */
int at_set_min_score(at_handle *h, int enabled, int32_t min_score);

/*
 * The same sweep with bounded memory, for triangles too large to hold (C5: 50 000 reads = 1.25e9 pairs = 20 GB of
 * results): scores and end cells only, the triangle cut into slices of at most chunk_pairs pairs (<= 0: 4 Mi); `fn`
 * is called once per slice, in pair order, on the calling thread, while the GPU already sweeps the next slice.  The
 * arrays it sees (indexed by pair - first) are valid during the call only; a non-zero return stops the sweep
 * (AT_ERR_ARG).  at_align_allpairs without tracebacks is this entry with a callback that copies into its out arrays.
 */
typedef int (*at_allpairs_chunk_fn)(void *user, int64_t first, int64_t npairs,
                                    const int32_t *score, const int32_t *end_i, const int32_t *end_j, const int32_t *state);
int at_align_allpairs_stream(at_handle *h, int mode, int64_t nreads,
                             const uint8_t *seq_blob, const int64_t *off, const int32_t *len,
                             int64_t first_pair, int64_t npairs, int64_t chunk_pairs,
                             at_allpairs_chunk_fn fn, void *user);

/*
 * Output rendering on the GPU: what trace_back_* + strrev produce (alignment.h:372-412,
 * 558-592, 766-800, 896-922, 172-184) -- the two gapped strings, in reading order -- from the op codes and end
 * cells at_align_batch_device left in HBM and the same packed sequences.  Pair k's strings are written to
 * d_r1/d_r2 at byte offset d_str_off[k] (NULL = d_ops_off[k]), d_nops[k] characters each, followed by a 0 byte
 * when nul_terminate (the slot then needs len1+len2+1 bytes).  Asynchronous on `stream`.
 */
int at_render_batch_device(at_handle *h, int64_t npairs,
                           const uint32_t *d_seq, int bits,
                           const int64_t *d_woff1, const int64_t *d_woff2,
                           const int32_t *d_end_i, const int32_t *d_end_j,
                           const uint8_t *d_ops, const int64_t *d_ops_off, const int32_t *d_nops,
                           uint8_t *d_r1, uint8_t *d_r2, const int64_t *d_str_off, int nul_terminate,
                           void *stream);

/*
 * CIGAR compaction for the result gather (sizes first, then one payload).  Writes the exclusive
 * prefix sums of d_nops to d_packed_off[0 .. npairs] (d_packed_off[npairs] = total bytes) and copies pair k's ops
 * from its slot to d_packed[d_packed_off[k] ..).  Pairs that would end beyond packed_cap are not copied: compare the
 * total with the capacity.  Asynchronous on `stream`.
 */
int at_compact_ops_device(at_handle *h, int64_t npairs,
                          const uint8_t *d_ops, const int64_t *d_ops_off, const int32_t *d_nops,
                          uint8_t *d_packed, int64_t packed_cap, int64_t *d_packed_off, void *stream);

/*
 * at_align_batch with the rendering done on the GPU: instead of op codes the caller receives the reference's
 * two strings per pair (0-terminated) at out_r1/out_r2 + str_off[k], slots of len1[k]+len2[k]+1 bytes, and
 * their common length in out_len[k].  Not for AT_MODE_EDIT (edit_dist returns a number only).
 */
int at_align_batch_strings(at_handle *h, int mode, int64_t npairs,
                           const uint8_t *seq_blob,
                           const int64_t *off1, const int32_t *len1,
                           const int64_t *off2, const int32_t *len2,
                           int32_t *out_score, int32_t *out_end_i, int32_t *out_end_j, int32_t *out_state,
                           char *out_r1, char *out_r2, const int64_t *str_off, int32_t *out_len);

/*
 * Multi-process batches: one process per GPU, each with its own handle.  The pairs are independent, so
 * the only communication is a broadcast of rank 0's scoring block and a gather of results, both RCCL collectives on the
 * handles' devices (over xGMI inside a node).  `dir` is a directory all ranks can reach: rank 0 leaves the RCCL id there.
 * RCCL is loaded at at_comm_init, not linked.  at_comm_allgather gathers a different number of bytes from every rank:
 * sizes first, then one payload padded to the largest (the two-phase CIGAR gather); every rank receives everything.
 */
int at_comm_init(at_handle *h, int rank, int world, const char *dir);
int at_comm_broadcast_scoring(at_handle *h);
int at_comm_allgather(at_handle *h, const void *mine, int64_t mine_bytes, void *all, int64_t all_cap, int64_t *bytes_of_rank);
void at_comm_destroy(at_handle *h);
/* 1 if the library's hand-written RCCL declarations were checked against the installed <rccl/rccl.h> when it was built
 * (static_asserts in csrc/at_comm.hip: id size, enum values, every entry point's argument list), 0 if no header was there */
int at_comm_abi_checked(void);

/* Host helper: pack `npairs` pairs of raw bytes into the word layout above.
 * bits = 0 picks 2 when every byte is one of ACGT, else 8; the choice is
 * returned in *bits_out.  words_out must hold at_pack_words(...) int32s. */
int64_t at_pack_words(int64_t npairs, const int32_t *len1, const int32_t *len2, int bits);
int at_pack_batch(int64_t npairs, const uint8_t *seq_blob,
                  const int64_t *off1, const int32_t *len1,
                  const int64_t *off2, const int32_t *len2,
                  int bits, int *bits_out,
                  uint32_t *words_out, int64_t *woff1_out, int64_t *woff2_out);

/* Host helper: ops (END -> START) + end cell -> the reference's two gapped
 * strings (what trace_back_* + strrev produce).  r1/r2 need nops+1 bytes. */
int at_render(const uint8_t *ops, int32_t nops,
              const uint8_t *s1, int32_t end_i, const uint8_t *s2, int32_t end_j,
              char *r1, char *r2);

/* Introspection for benchmarks / tests: name of the kernel configuration the
 * last batch ran with ("small"/"large", bits, waves), never NULL. */
const char *at_last_config(const at_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* ALIGNTOOLS_HIP_H */
