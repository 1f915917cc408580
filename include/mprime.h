/*
 * mprime.h — C ABI of the MI355X-native hot path of multiPrime's core step
 * (degenerate-primer candidate enumeration + mismatch-tolerant coverage scoring).
 *
 * The reference (joybio/multiPrime, scripts/multiPrime-core_V20.py, "V20" below) has no
 * FFI: everything is Python.  This header is the boundary a maintainer would bind with
 * ctypes from that script: every O(N_sequences) loop of V20 becomes
 * one call here, the O(1)-per-window control flow stays in Python.
 *
 * Two libraries export exactly these symbols:
 *   multiprime_amd/csrc/libmprime_hip.so   hand-written HIP for gfx950 (the product)
 *   oracle/_build/libmprime_oracle.so      plain-C restatement of V20 (test infrastructure only)
 *
 * Conventions: plain pointers and sizes, caller owns every buffer, the library owns only
 * the opaque context.  Every function returns MP_OK (0) or a negative MP_ERR_* code and
 * never throws; mp_last_error() gives a message.  One context per host thread / GPU.
 *
 * Encodings
 *   symbol code   4-bit IUPAC base-set mask: A=1 C=2 G=4 T=8, R=A|G ... ; '-' (and anything
 *                 V20:453 maps to '-', N included) = 0.
 *   window words  one (window,row) k-mer = three words: b0,b1 = low/high bit of the base
 *                 index (A=0,C=1,G=2,T=3; 0 where gap), g = gap flag; bit j = window position
 *                 j (0 = 5' end), k <= MP_MAX_K.  A word is a uint32 while k <= MP_NARROW_K (31)
 *                 and a uint64 for 32 <= k <= 63 (MP_WORD_BYTES(k)); arrays of words are passed
 *                 as void pointers.  The top bit of g (MP_WIN_SKIP, bit 31 / MP_WIN_SKIP64, bit
 *                 63) marks a slot that is not part of the evaluated universe (its window holds
 *                 an IUPAC code and was handed to the host as an exception, or the row does not
 *                 exist).
 *   candidate     k symbol codes (one uint8 each), 5'->3'.
 *   strict mask   uint64, bit j set = a mismatch at 0-based position j disqualifies (V20:1091-1101
 *                 get_Y; entries outside [0,k) are dropped by the host, as they can never
 *                 equal a mismatch index).
 */
#ifndef MPRIME_H
#define MPRIME_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP_MAX_K 63            /* the reference takes any -l (V20:64-65); here one k-mer fits one machine word per plane */
#define MP_NARROW_K 31         /* up to here window words are 32-bit */
#define MP_WORD_BYTES(k) ((k) <= MP_NARROW_K ? 4 : 8)
#define MP_WIN_SKIP 0x80000000u
#define MP_WIN_SKIP64 0x8000000000000000ull

#define MP_OK 0
#define MP_ERR_ARG (-1)          /* bad argument / call order */
#define MP_ERR_DEVICE (-2)       /* HIP runtime error (message has hipGetErrorString) */
#define MP_ERR_NOMEM (-3)
#define MP_ERR_CAPACITY (-4)     /* a caller-sized buffer was too small; message says what is needed */
#define MP_ERR_SHORT_WINDOW (-5) /* a ragged row leaves fewer than k residues (V20:683-687 falls through
                                    with a short k-mer there; behaviour of the reference is undefined) */

typedef struct mp_ctx mp_ctx;

/* lifetime ------------------------------------------------------------------------------- */
int mp_create(int device_ordinal, mp_ctx **out);
void mp_destroy(mp_ctx *ctx);
const char *mp_last_error(const mp_ctx *ctx);
/* "hip" or "oracle" */
const char *mp_backend_name(void);
/* Launch every kernel of this context on `hip_stream` (a hipStream_t; NULL = the default
 * stream).  bench.py passes torch's current stream so that torch.cuda events and RCCL
 * collectives order with the kernels.  A context works on ONE stream at a time: its stages hand
 * device blocks to each other without waiting for the device (a released block waits in the
 * context for the next request of its size), which is safe because everything is ordered on that
 * stream — so a change of stream first waits for what the old one still has in flight. */
int mp_set_stream(mp_ctx *ctx, void *hip_stream);

/* (1) alignment -> device ----------------------------------------------------------------- */
/* Row shards: the alignment is at least n_columns wide even if no row of this context's share is
 * that long (the window range comes from quantiles over ALL rows, V20:617-640).  Call before
 * mp_load_msa; windows may then start anywhere below max(longest local row, n_columns), and a row
 * that ends before a window is an ordinary ragged row. */
int mp_reserve_columns(mp_ctx *ctx, int32_t n_columns);

/* Replaces the per-character work of parse_seq (V20:441-455): `bytes` holds, back to back,
 * the residue characters of each record (the host has already joined a record's lines and
 * dropped '>' / '#' lines); row r is bytes[row_off[r] .. row_off[r+1]).  Per character:
 * upper-case, keep ACGTRYMKSWHBVD, everything else becomes '-' (V20:453).  Rows may be
 * ragged (unaligned input such as test_data/test.fa). */
int mp_load_msa(mp_ctx *ctx, const uint8_t *bytes, const int64_t *row_off, int32_t n_rows);

/* Replaces the row scans of seq_attribute (V20:622-627): lead_gap[r] = len - len(lstrip('-')),
 * rstrip_len[r] = len(rstrip('-')), row_len[r] = len.  Any pointer may be NULL. */
int mp_row_attributes(mp_ctx *ctx, int32_t *lead_gap, int32_t *rstrip_len, int32_t *row_len);

/* The same two quantities as histograms: lead_hist[x] = rows whose leading-gap length is x, rstrip_hist[x] = rows whose
 * right-stripped length is x, 0 <= x < n_bins (n_bins > the longest row, else MP_ERR_CAPACITY).  seq_attribute only takes one
 * order statistic of each (np.quantile, "higher" / "lower", V20:628-633), which a cumulative sum of the histogram gives
 * exactly — 8 x n_bins bytes leave the device instead of two values per row, and row shards add their histograms. */
int mp_row_histograms(mp_ctx *ctx, int32_t n_bins, int64_t *lead_hist, int64_t *rstrip_hist);

/* (2) window k-mers ------------------------------------------------------------------------ */
/* Replaces the slice + edge-gap repair of get_primers (V20:666-687) for every
 * (window p0+w, row), w in [0,n_windows): the k-mer of row r at window w is stored as window
 * words.  A k-mer that contains an IUPAC code (before or after repair) is not stored (its slot
 * gets MP_WIN_SKIP) but appended to the exception list, to be expanded by the host
 * (degenerate_seq, V20:368-380) and handed back through mp_set_extra_rows.
 * `v` = --variation: a k-mer with more than v gaps is a "gap row" (V20:689). */
int mp_build_windows(mp_ctx *ctx, int32_t p0, int32_t n_windows, int32_t k, int32_t v,
                     int32_t *n_exceptions);

/* Exception list of the last mp_build_windows, sorted by (window, row): ex_codes holds k symbol
 * codes per exception.  `cap` = capacity of the three arrays in exceptions. */
int mp_get_exceptions(mp_ctx *ctx, int32_t cap, int32_t *ex_window, int32_t *ex_row, uint8_t *ex_codes);

/* Concrete expansions of exception k-mers with <= v gaps, as extra rows of the evaluated
 * universe: window[i] ascending (relative to p0), words[3*i..3*i+2] = b0,b1,g. */
int mp_set_extra_rows(mp_ctx *ctx, int32_t n_extra, const int32_t *window, const void *words);

/* Parity/debug: window words of rows [row0,row0+n) of window w: out[0..n)=b0, [n..2n)=b1, [2n..3n)=g */
int mp_get_window_words(mp_ctx *ctx, int32_t w, int32_t row0, int32_t n, void *out);

/* (3) per-window k-mer histogram ----------------------------------------------------------- */
/* Replaces the dictionary building of get_primers (V20:689-711: cover / gap_sequence counts
 * and, through first_row, their first-seen order).  For every window: the distinct window
 * words over all non-SKIP rows with their multiplicity and the smallest row holding them.
 * cap_entries bounds the total number of entries over all windows (MP_ERR_CAPACITY if
 * exceeded; *n_entries then holds the number needed).  want_labels != 0 additionally keeps,
 * per (window,row), the index of the row's entry inside its window (-1 for SKIP rows) so the
 * host can rebuild the id lists (non_gap_seq_id / gap_seq_id, V20:698,707). */
int mp_window_unique(mp_ctx *ctx, int64_t cap_entries, int32_t want_labels, int64_t *n_entries);

/* The entropy gate of V20:723 decided on the device where that is certain.  More than half of the windows of a deep alignment end at
 * `Entropy of total (bit) > threshold` — after the host has read back, decoded, merged and ordered all their entries, and they are the
 * windows with the most entries.  mp_set_entropy_gate(threshold > 0) arms the gate for the following mp_window_unique calls WITHOUT labels
 * (0 disarms it): the call then takes the entropy of every window's table on the device and rejects the windows that exceed
 * threshold + 0.005 (the rounding to two decimals) by more than a bound on what the table cannot see — the rows holding an IUPAC code,
 * which the host adds with their expansions: total-variation distance theta = E / (T + E) between the two distributions, hence at most
 * theta log2(T + E) + h2(theta) between their entropies.  A rejected window has no entries (win_off[w] == win_off[w + 1]);
 * mp_plan_create_streamed reports it as MP_WIN_ENTROPY_DEVICE without planning it; every other window is decided by the host exactly as
 * before, so the planned windows and everything derived from them are unchanged.  mp_entropy_gate_result: how many windows the last
 * mp_window_unique rejected, and which (rejected[n_windows], may be NULL).  The checker accepts the calls and never rejects. */
int mp_set_entropy_gate(mp_ctx *ctx, double threshold);
int mp_entropy_gate_result(mp_ctx *ctx, int32_t *n_rejected, uint8_t *rejected);

/* Entries of window w are [win_off[w], win_off[w+1]); order inside a window is unspecified.
 * words: b0 at [0,n), b1 at [n,2n), g at [2n,3n) with n = total entries. */
int mp_get_unique(mp_ctx *ctx, int64_t *win_off, void *words, int32_t *count, int32_t *first_row);
int mp_get_labels(mp_ctx *ctx, int32_t w, int32_t *labels);
/* The labels of n windows at once: labels[i][n_rows] = those of windows[i] (one synchronisation for all of them). */
int mp_get_labels_many(mp_ctx *ctx, int32_t n, const int32_t *windows, int32_t *labels);

/* (3b) per-window base and nearest-neighbour counts ------------------------------------------ */
/* Replaces state_matrix (V20:541-554) and di_matrix / trans_matrix (V20:556-577), which the
 * reference builds as pandas frames of one row per sequence: over the window's universe — one
 * row per sequence whose k-mer has <= v gaps, plus the extra rows of mp_set_extra_rows —
 *   freq[w][b][j]  = rows whose symbol at position j is base b (A,C,G,T = 0..3; the '-' row is
 *                    dropped, V20:551-552)
 *   nn[w][j][a][b] = rows with base a at position j and base b at j+1 (pairs touching '-' are
 *                    not counted, V20:569-575)
 * Host arrays, int64: freq [n_windows][4][k], nn [n_windows][k-1][4][4]. */
int mp_window_stats(mp_ctx *ctx, int64_t *freq, int64_t *nn);
/* [r6] mp_window_stats in two halves: begin launches the kernel and the read-back of its counters on the context's second stream and
 * returns at once, end waits and fills freq / nn.  mp_plan_create_streamed (mprime_host.h) calls end itself when a begin is pending: its
 * read-back of the histogram entries then runs beside the statistics kernel.  One begin at a time. */
int mp_window_stats_begin(mp_ctx *ctx);
int mp_window_stats_end(mp_ctx *ctx, int64_t *freq, int64_t *nn);

/* (4) candidate x sequence coverage evaluation ---------------------------------------------- */
/* Replaces mis_primer_check + Y_distance (V20:1103-1130, 229-233), evaluated per sequence
 * instead of per distinct k-mer.  For candidate c (window cand_window[c], ascending) and
 * every row of the universe (non-SKIP rows with <= v gaps, plus the extra rows of that window):
 *   D = { j : symbol_j not in candidate_j }           ('-' is in no candidate symbol)
 *   out[3c+0] += |D| == 0                             perfect coverage  (V20:853, :954-956)
 *   out[3c+1] += 0 < |D| <= v and D & strictF == 0    F_mis_cover       (V20:1123)
 *   out[3c+2] += 0 < |D| <= v and D & strictR == 0    R_mis_cover       (V20:1127)
 * One call = one batched launch over all candidates. */
int mp_eval_candidates(mp_ctx *ctx, int32_t n_cand, const int32_t *cand_window, const uint8_t *cand_codes,
                       uint64_t strictF, uint64_t strictR, int64_t *out);

/* (4c) per-sequence coverage masks ----------------------------------------------------------- */
/* The bitset form of the two JSON side files (V20:1172-1177) for one candidate per entry: for
 * candidate c (window cand_window[c], ascending) bit r of not_f[c] (resp. not_r[c]) is set iff row r
 * would appear in gap_seq_id (more than v gaps, V20:689-698) or in the F (resp. R) dict of
 * non_coverage_seq_id (in `cover`, not perfectly matched, and not F- (resp. R-) admissible,
 * V20:1107-1127).  Rows whose window went to the exception list (IUPAC) get 0: the host owns them.
 * Each mask is (n_rows + 63) / 64 words.  This is what the pairing stage needs and it
 * scales as bits, not id strings. */
int mp_eval_masks(mp_ctx *ctx, int32_t n_cand, const int32_t *cand_window, const uint8_t *cand_codes,
                  uint64_t strictF, uint64_t strictR, uint64_t *not_f, uint64_t *not_r);

/* (4d) the same masks kept on the device — the hand-off to the pairing stage when core and pairing run in one process:
 * mp_eval_masks_resident computes them and leaves them in the context ([n_cand][row words]); mp_masks_set_bits applies the
 * host's verdict on single (mask, row) bits (which[i] = 0: not_f, 1: not_r — the rows whose window held an IUPAC code, which
 * the masks leave 0); mp_masks_fetch copies them out in mp_eval_masks's layout; mp_pair_coverage_resident returns, for every
 * pair (i, j), popcount(not_f[i] | not_r[j]) — the sequences a forward primer at window i or a reverse primer at window j
 * does not reach (get_multiPrime_V8.py:560-569) — without the masks ever leaving HBM. */
int mp_eval_masks_resident(mp_ctx *ctx, int32_t n_cand, const int32_t *cand_window, const uint8_t *cand_codes,
                           uint64_t strictF, uint64_t strictR);
int mp_masks_set_bits(mp_ctx *ctx, int64_t n, const int32_t *mask_index, const int32_t *row, const uint8_t *which, const uint8_t *value);
int mp_masks_fetch(mp_ctx *ctx, uint64_t *not_f, uint64_t *not_r);
int mp_pair_coverage_resident(mp_ctx *ctx, int64_t n_pairs, const int32_t *pairs, int32_t *out);

/* Device-resident form used by bench.py and the multi-GPU path: upload stages the candidate
 * tables once; launch enqueues the evaluation on the context's stream and leaves the
 * [n_cand][3] int64 counters in `device_out` (device memory owned by the caller, e.g. a torch
 * tensor that is then all-reduced over RCCL).  The oracle treats device_out as host memory. */
int mp_eval_upload(mp_ctx *ctx, int32_t n_cand, const int32_t *cand_window, const uint8_t *cand_codes,
                   uint64_t strictF, uint64_t strictR);
int mp_eval_launch(mp_ctx *ctx, int64_t *device_out);
/* The same evaluation for a caller that alternates between counter blocks (one step per alignment or per bucket of a pipelined
 * all-reduce): mp_eval_launch clears device_out with a dispatch of its own before the evaluation adds to it — 4-5 us on the stream,
 * a sixth of the evaluation of a 131072-row shard.  Here the CALLER vouches that device_out holds zeros (it was the device_clear of
 * an earlier rotating launch on this stream, or the caller zeroed it), and device_clear — another [n_cand][3] block, or NULL — is
 * set to zero by this launch's own workgroups beside their work, ready to be the device_out of the next launch.  Nothing else may be
 * using device_clear while the launch runs (a collective still reading it on another stream must have been waited for).
 * device_clear == device_out is an error. */
int mp_eval_launch_rotating(mp_ctx *ctx, int64_t *device_out, int64_t *device_clear);
/* mp_eval_launch on the context's SECOND stream (created on first use): a caller with a queue of independent evaluations of one
 * staged candidate set — steps of a benchmark, alignments of a batch that share their candidates' windows — alternates between
 * mp_eval_launch and this call, and consecutive launches overlap where they do not use the chip: the next kernel's dispatch, its
 * workgroups' first misses and warm-up columns run while the last workgroups of the kernel before it finish, and the clearing
 * dispatch of one block runs beside the evaluation into the other (a launch of a 1/8 share of config 4 is 17 us of window work in
 * a 27 us kernel).  The two launches in flight must write different counter blocks.  The first launch after mp_eval_upload has to be
 * an mp_eval_launch (it builds what the staged set needs once); results are complete when BOTH streams are (mp_eval_sync, or the
 * caller's device-wide synchronisation). */
int mp_eval_launch_alt(mp_ctx *ctx, int64_t *device_out);
int mp_eval_sync(mp_ctx *ctx);

/* HIP-event timing of the evaluation kernels themselves (recorded on the context's stream around
 * mp_eval_launch since the last reset): total milliseconds and number of launches timed.  Every
 * launch is timed unless the environment says MP_EVAL_TIMING_EVERY=n (every n-th launch counted
 * from the reset, 0 = none): an event pair idles the stream for a few microseconds. */
int mp_eval_timing(mp_ctx *ctx, int32_t reset, double *total_ms, int32_t *n_launches);
/* The individual durations (milliseconds) behind the totals of the LAST mp_eval_timing call, at most 4096 since its
 * reset: *n returns how many there are, the first min(*n, cap) are written (median / maximum for bench.py). */
int mp_eval_timing_samples(mp_ctx *ctx, int32_t cap, float *ms, int32_t *n);
/* How the staged candidates (mp_eval_upload) will be evaluated, for run logs and tests: info[0] = nested chain items, info[1] =
 * symbol-table items, info[2] = chain items the sliding kernel takes (0: the first-pass kernels run all of them), info[3] = chain
 * items left to the first-pass kernel beside it.  The checker reports zeros. */
int mp_eval_plan_info(mp_ctx *ctx, int32_t *info);

/* (5) 3'-end dimer scan — SURVEY §8a rows D and M ------------------------------------------- */
/* Replaces the search loops of Dimer.dimer_check (scripts/finDimer_V4.py:191-224) and of
 * dimer_examination (scripts/get_Maxprimerset_V1.3.py:193-215).  Primers are given as symbol
 * codes (IUPAC allowed, no gaps), primer p = codes[off[p] .. off[p+1]), length <= MP_DIMER_MAX_LEN.
 *
 * For a primer x and an "end" length l, every expansion e of x's 3' suffix of length l (in the
 * reference's expansion order, last position fastest) is searched as RC(e) in every expansion p
 * of the other primer y (same order); only the FIRST occurrence idx in p counts (str.find).  With
 * d2 = len(p) - l - idx and GC = #G+#C in e the pair is a dimer hit when
 *     loss_hit[l][GC][d2] != 0                  (Loss = Penalty_points(l,GC,0,d2) >= threshold)
 *  or d2 == 0 and deltaG(e) < dg_limit          (deltaG < -5 after rounding to 2 decimals)
 * The Loss decision table and the deltaG constants are supplied by the caller, computed with the
 * reference's own libm expressions, so that no device-side log10 / multiply can change a call:
 *   loss_hit  [MP_DIMER_MAX_LEN+1][MP_DIMER_MAX_LEN+1][64] bytes
 *   dg_params [16] stack terms indexed [next][this] (finDimer_V4.py:176-178), then [32] initiation
 *             terms indexed [first][last][ends with "TA"] (:179-183), then [MP_DIMER_MAX_LEN+1] the Na+ term
 *             times the length (:185), then [1] the symmetry correction (:186-187); deltaG is the
 *             left-to-right double sum of those.
 *   dg_limit  the smallest double whose 2-decimal rounding is not below -5.
 *
 * mode 0 (finDimer): unordered pairs i <= j, ends of i only, end lengths min(t, len_i) for
 *   t = 18 .. 5 (finDimer_V4.py:162-169), longest first, then expansion of the end, then
 *   expansion of j; the first passing combination is the pair's hit and the scan of the pair stops
 *   (at most one hit per pair).  hits[6h..] = {i, j, l, end expansion index, j expansion index, idx}.
 * mode 1 (Maxprimerset): ordered pairs (x, y) with x < n_new or y < n_new (the new primers are
 *   listed first; pairs among the old ones were verified when they were added), end lengths
 *   len_x-1 .. 5 (get_Maxprimerset_V1.3.py:149-154); same hit record, one per pair.
 * hits has room for cap_hits records; *n_hits returns the number found (may exceed cap_hits,
 * in which case only the first cap_hits written are valid).  Order of records is unspecified. */
#define MP_DIMER_MAX_LEN 64      /* primers of the dimer scans (adaptor-tailed primers included) */
#define MP_PATTERN_MAX_LEN 64    /* primers / patterns of the sequence scans (mp_pcr_scan, mp_kmm_scan): 2 bits per base in one or two 64-bit words */
int mp_dimer_scan(mp_ctx *ctx, int32_t n_primers, const uint8_t *codes, const int32_t *off, int32_t mode,
                  int32_t n_new, const uint8_t *loss_hit, const double *dg_params, double dg_limit,
                  int64_t cap_hits, int32_t *hits, int64_t *n_hits);

/* mode-0 style search for an explicit list of ORDERED primer pairs (x -> y): ends of x (lengths
 * min(t, len_x), t = 18..5) against the expansions of y, any passing combination sets flags[p] = 1.
 * Replaces the loops of Primers_filter.dimer_check (scripts/get_multiPrime_V8.py:419-438), which
 * is the union of the four ordered pairs (F,F), (F,R), (R,F), (R,R); its `Loss > 3.6` and its
 * one-term initiation in deltaG (:405-413) come in through loss_hit / dg_params. */
int mp_dimer_pairs(mp_ctx *ctx, int32_t n_primers, const uint8_t *codes, const int32_t *off, int64_t n_pairs,
                   const int32_t *pairs, const uint8_t *loss_hit, const double *dg_params, double dg_limit,
                   uint8_t *flags);

/* (6) coverage of primer pairs from per-window sequence bitsets — SURVEY §8f-1 -------------- */
/* Replaces the id-list unions of Primers_filter.primer_pairs (get_multiPrime_V8.py:560-569): set a
 * of `sets_a` holds the sequences a forward window does not cover (gap rows U F non-covered), set b
 * of `sets_b` the same for a reverse window; out[p] = popcount(sets_a[pairs[2p]] | sets_b[pairs[2p+1]]).
 * Each set is n_words 64-bit words. */
int mp_pair_coverage(mp_ctx *ctx, int32_t n_sets, int32_t n_words, const uint64_t *sets_a, const uint64_t *sets_b,
                     int64_t n_pairs, const int32_t *pairs, int32_t *out);

/* (7) exact in-silico PCR — SURVEY §8f-2 ---------------------------------------------------- */
/* Replaces the search of Product.get_PCR_PRODUCT (scripts/extract_PCR_product_V1.py:189-216, "PCR") for
 * every (primer pair, sequence).  `bytes`/`row_off` hold the sequence lines of the reference FASTA as they
 * stand in the file (no upper-casing: the reference's re.search is case sensitive, so only upper-case
 * A/C/G/T can match a primer expansion).  Primer 2p is the forward, 2p+1 the reverse primer of pair p
 * (symbol codes, IUPAC allowed, <= MP_PATTERN_MAX_LEN).  For each pair and sequence, in the reference's order:
 * the first forward expansion iF that occurs in the sequence AND for which a reverse expansion matches
 * inside its "Product" — the text from the first occurrence p1 of that expansion up to its next
 * non-overlapping occurrence (str.split) or the end of the line; inside it the first reverse expansion iR
 * (in expansion order) whose reverse complement occurs, at its first position q.  The amplicon is
 * sequence[p1 : q + len(R)].  out[(p * n_rows + r) * 4 ..] = {iF, p1, iR, q}, iF = -1 when there is none. */
int mp_pcr_scan(mp_ctx *ctx, const uint8_t *bytes, const int64_t *row_off, int32_t n_rows, int32_t n_pairs,
                const uint8_t *codes, const int32_t *off, int32_t *out);

/* (8) k-mismatch primer-site scan — SURVEY §8f-3 ------------------------------------------------ */
/* Replaces the mapping step of scripts/primer_coverage_validation_by_BWT_V9.py ("BWT": bowtie2 -N m -L 8 -a + samtools
 * + the MD:Z filter of build_dict, BWT:264-300, 241-262) by an exhaustive ungapped scan.  bowtie2 and samtools are absent
 * from this image, so the mapper cannot be run here; its acceptance rule below is pinned to the reference author's own
 * bowtie2 run instead (the shipped test_data/results/Core_primers_set/BWT_coverage/ output of rule BWT_validation: 1158
 * sequences, 485 with a product and 673 without, every decision reproduced — tests/golden/make_golden_bwt.py,
 * tests/test_validate_bwt.py); the MD:Z rule and everything else the script does are pinned to V9 itself
 * (tests/golden/validate.json.gz):
 * `bytes`/`row_off` hold the reference sequences (upper-cased by the scan; any character outside ACGT mismatches every
 * base, like bowtie2's N).  Pattern i = pat_codes[pat_off[i] .. pat_off[i+1]) is one CONCRETE primer expansion
 * (codes 1,2,4,8; length 4..MP_PATTERN_MAX_LEN).  For every sequence, start position p and strand s (0: the text reads the
 * pattern, 1: the text reads its reverse complement — SAM flag 16) the ungapped alignment is a hit when
 *   - it has at most max_mismatch mismatching positions (bowtie2 end-to-end, default scoring: the minimum score
 *     -0.6 - 0.6 L with 6 per mismatch admits floor((0.6 + 0.6 L) / 6) mismatches; the host passes that or its override), and
 *   - the last `term` positions of the alignment IN REFERENCE ORIENTATION all match: build_dict keeps an alignment when
 *     the trailing match count of its MD:Z string is >= the 3'-term threshold, and applies the same test to the
 *     reverse-strand file, where the end of the MD string is the primer's 5' end (quirk kept).
 * hits[4h..] = {sequence, p, pattern, strand}; cap_hits / *n_hits as in mp_dimer_scan; order unspecified. */
int mp_kmm_scan(mp_ctx *ctx, const uint8_t *bytes, const int64_t *row_off, int32_t n_rows, int32_t n_patterns,
                const uint8_t *pat_codes, const int32_t *pat_off, int32_t max_mismatch, int32_t term, int64_t cap_hits,
                int32_t *hits, int64_t *n_hits);

/* (8b) the resident sequence store shared by the scans of (7) and (8) — SURVEY §8f-4 ----------------------------------------- */
/* The steps either side of the core step read the SAME unaligned sequence database more than once: extract_PCR_product_V1.py:189-216
 * searches it for every primer pair, primer_coverage_validation_by_BWT_V9.py:264-300 maps every primer set against it.  mp_pcr_scan
 * and mp_kmm_scan take the database as ASCII bytes per call (1 byte per base over PCIe and through the kernel's packing loop, every
 * time).  mp_seq_load uploads it ONCE into the context and packs it on the device: per sequence, 64-bit words of 32 bases — a CODE
 * word (2 bits per base, A0 C1 G2 T3, case folded) and a FLAG word (bit 2j: base j is not A/C/G/T in either case, or lies past the
 * end of the sequence; bit 2j+1: base j is a lower-case letter) — 4 bits per base, everything both scans need: the PCR search is
 * case sensitive as the reference's re.search is (a lower-case base matches no expansion: it tests both flag bits), the k-mismatch
 * scan upper-cases as bowtie2 does (it tests bit 2j only).  The bytes stay in HBM beside the words for the rare fall-backs that walk
 * characters (a pair table of more than 4096 expansions, a sequence whose occurrence list overflows).  The *_resident scans are
 * mp_pcr_scan / mp_kmm_scan on the stored database: same arguments minus the text, same results bit for bit.  One store per context;
 * a second mp_seq_load replaces it, mp_seq_free / mp_destroy release it. */
int mp_seq_load(mp_ctx *ctx, const uint8_t *bytes, const int64_t *row_off, int32_t n_rows);
int mp_seq_free(mp_ctx *ctx);
/* n_rows of the store, its bases, and the bytes it holds on the device (words + characters + offsets) */
int mp_seq_info(mp_ctx *ctx, int32_t *n_rows, int64_t *n_bases, int64_t *device_bytes);
int mp_pcr_scan_resident(mp_ctx *ctx, int32_t n_pairs, const uint8_t *codes, const int32_t *off, int32_t *out);
int mp_kmm_scan_resident(mp_ctx *ctx, int32_t n_patterns, const uint8_t *pat_codes, const int32_t *pat_off, int32_t max_mismatch,
                         int32_t term, int64_t cap_hits, int32_t *hits, int64_t *n_hits);

/* (9) row shards over several GPUs — SURVEY §8e --------------------------------------------------------------------------- */
/* The reference is one process (its -p pool is inert, V20:1143).  Here every O(N) quantity of the path is a sum over sequences,
 * so N processes (one per GPU) each load a contiguous block of the alignment's rows (mp_reserve_columns + mp_load_msa), build
 * the same windows, and exchange: ONE all-reduce (sum, int64) of the [n_candidates x 3] counters of mp_eval_*, one of the
 * mp_window_stats tables, and variable-length all-gathers of the packed host tables (histogram entries, exceptions).  The
 * transport is RCCL over xGMI, opened on the first call.  A host in any language drives it:
 *     rank 0: mp_comm_unique_id(id)  ->  id to every rank (MPI_Bcast, a socket, a file — the host's business)
 *     all:    mp_comm_init(ctx, n_ranks, rank, id)                 (collective: every rank calls it)
 *     all:    mp_eval_candidates_allreduce(...) instead of mp_eval_candidates; mp_comm_allreduce_host_i64 on the
 *             mp_window_stats tables; mp_comm_allgather_i64 + mp_comm_allgatherv for the tables every rank needs in full,
 *             mp_comm_alltoall_counts + mp_comm_alltoallv for the tables with one consumer per row (histogram entries -> the rank
 *             that plans the window)
 * n_ranks = 1 is valid without RCCL (every collective is the identity).  Collectives are enqueued on the context's stream
 * (mp_set_stream) straight behind the kernels; the *_host_* / gather forms return when the result is in the caller's buffer.
 * Every rank must issue the same collectives in the same order.  The checker library implements n_ranks = 1 only. */
#define MP_COMM_ID_BYTES 128
int mp_comm_unique_id(uint8_t *id);                                        /* id[MP_COMM_ID_BYTES] */
int mp_comm_init(mp_ctx *ctx, int32_t n_ranks, int32_t rank, const uint8_t *id);
int mp_comm_destroy(mp_ctx *ctx);
/* what the communicator itself reports: ranks_seen[0] = its size, ranks_seen[1] = this rank (ncclCommCount / ncclCommUserRank; the
 * arguments of mp_comm_init for a world of one without RCCL), and the file the RCCL entry points were resolved from ("" when RCCL is
 * not in use).  For run logs: shows that N ranks really formed ONE communicator and which librccl carried it. */
int mp_comm_describe(mp_ctx *ctx, int32_t *ranks_seen, char *library_path, int32_t path_bytes);
/* in place, sum over ranks; `device_buf` is device memory, the call only enqueues */
int mp_comm_allreduce_i64(mp_ctx *ctx, int64_t *device_buf, int64_t n);
/* the same for a host buffer: returns with the sums in `host_buf` */
int mp_comm_allreduce_host_i64(mp_ctx *ctx, int64_t *host_buf, int64_t n);
/* out[r] = rank r's `value` (lengths of a variable-length gather) */
int mp_comm_allgather_i64(mp_ctx *ctx, int64_t value, int64_t *out);
/* recv = rank 0's bytes, rank 1's bytes, ...; counts[r] = bytes of rank r (counts[rank] == n_bytes); host buffers */
int mp_comm_allgatherv(mp_ctx *ctx, const void *send, int64_t n_bytes, const int64_t *counts, void *recv);
/* Personalised exchange (all-to-all-v) of host byte arrays: `send` holds, in rank order, send_counts[r] bytes for every rank r.
 * mp_comm_alltoall_counts tells every rank what it will receive (recv_counts[r] = what rank r sends to this rank: one all-gather of the
 * n_ranks x n_ranks count matrix); mp_comm_alltoallv then moves the bytes (ncclAllToAllv, or grouped ncclSend / ncclRecv where the
 * library lacks it): recv = rank 0's bytes for this rank, rank 1's, ...  Used where a table has ONE consumer per row — the histogram
 * entries of a window go to the rank that plans the window — so that a rank receives what it needs instead of everybody's everything
 * (the all-gather above: n_ranks times the bytes). */
int mp_comm_alltoall_counts(mp_ctx *ctx, const int64_t *send_counts, int64_t *recv_counts);
int mp_comm_alltoallv(mp_ctx *ctx, const void *send, const int64_t *send_counts, void *recv, const int64_t *recv_counts);
/* mp_eval_candidates over this rank's rows + the all-reduce of the counters on the same stream: out = the global counts */
int mp_eval_candidates_allreduce(mp_ctx *ctx, int32_t n_cand, const int32_t *cand_window, const uint8_t *cand_codes,
                                 uint64_t strictF, uint64_t strictR, int64_t *out);

/* Memory the context holds on the device, in bytes (window words, planes, tables). */
int mp_device_bytes(mp_ctx *ctx, int64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif /* MPRIME_H */
