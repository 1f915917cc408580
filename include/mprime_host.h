/*
 * mprime_host.h — C ABI of the native HOST stage of the core step: everything multiPrime-core_V20.py ("V20") does
 * per window once the O(N_sequences) loops have been reduced to tables by the device kernels of mprime.h —
 * the cover / gap_sequence dictionaries in first-seen order (V20:689-711), the gates (V20:713-740), entropy
 * (V20:602-614), the Viterbi and most-frequent seeds (V20:579-600), the greedy degeneracy refinement
 * (V20:860-1089) and the replay of its stopping rules on the batched evaluations — plus the FASTA record parser
 * (parse_seq, V20:441-455).  Plain C++ on the host cores (threads over windows / file chunks), no device calls:
 * these entry points take and return host arrays, so they work with any library serving mprime.h.
 *
 * Exported by multiprime_amd/csrc/libmprime_hip.so only (the product).  Their checker is test infrastructure:
 * oracle/core_ref.py (the pure-Python restatement that round 1 shipped, pinned to V20's recorded internals) and
 * the golden traces under tests/golden/.
 *
 * Conventions as in mprime.h: 0 or a negative MP_ERR_* code, never throws, caller owns every buffer, the library
 * owns only the opaque objects.  Symbol codes are 4-bit base-set masks (A=1 C=2 G=4 T=8, '-' = 0).
 */
#ifndef MPRIME_HOST_H
#define MPRIME_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------
 * (H1) FASTA records — replaces parse_seq (V20:441-455) minus its per-character mapping (that is mp_load_msa):
 * lines are split at \n, \r\n or \r (Python's universal newlines); a line starting with '#' is skipped; a line
 * starting with '>' sets the current id to the first space-delimited token of the stripped line ('>' included);
 * every other line is stripped of ASCII whitespace at both ends and appended to the current id's record — a
 * repeated id concatenates (the reference's defaultdict(str)); ids keep first-appearance order.
 * Sequence data before the first header is an error (the reference raises NameError there).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct mp_fasta mp_fasta;
int mp_fasta_parse_file(const char *path, int32_t n_threads, mp_fasta **out);
int mp_fasta_parse_buffer(const uint8_t *bytes, int64_t n_bytes, int32_t n_threads, mp_fasta **out);
void mp_fasta_destroy(mp_fasta *f);
const char *mp_fasta_error(const mp_fasta *f);
/* number of records, total residue bytes, total id bytes */
int mp_fasta_sizes(const mp_fasta *f, int32_t *n_rows, int64_t *n_residue_bytes, int64_t *n_id_bytes);
/* residues of all records back to back (row r = data[row_off[r] .. row_off[r+1])) — the input of mp_load_msa;
 * `data` may be NULL to fetch only the offsets.  The copy runs on n_threads threads. */
int mp_fasta_rows(const mp_fasta *f, uint8_t *data, int64_t *row_off);
/* ids back to back, id r = ids[id_off[r] .. id_off[r+1]) (raw bytes of the file) */
int mp_fasta_ids(const mp_fasta *f, uint8_t *ids, int64_t *id_off);
/* Residue bytes [byte0, byte1) of the rows laid end to end — mp_fasta_rows' data[byte0 .. byte1) — into dst, on n_threads threads
 * (0: the parser's).  The streamed load below is built on it. */
int mp_fasta_gather(const mp_fasta *f, int64_t byte0, int64_t byte1, uint8_t *dst, int32_t n_threads);
/* mp_load_msa(ctx, mp_fasta_rows(f)) without the intermediate copy [r6]: the records' residue bytes go from the parsed file straight
 * into a ring of transfer buffers the context keeps registered with the runtime (3 x 32 MB), chunk by chunk on the host's threads,
 * while the chunk before is on its way to the device by DMA — a 1 GB alignment used to be copied into a caller-owned array first
 * (1 GB of fresh pages), then crossed through the runtime's own staging buffers.  Same device state as mp_load_msa afterwards.
 * Product library only (it takes a device context); struct mp_ctx is the context of mprime.h. */
struct mp_ctx;
int mp_load_msa_fasta(struct mp_ctx *ctx, const mp_fasta *f);
/* Newlines of a file as Python's text mode counts them (\n, \r\n as one, a lone \r), on n_threads threads (0 = as many as pay
 * off) — get_multiPrime / get_degePrimer take "newlines / 2" of the whole input as the number of sequences
 * (get_multiPrime_V8.py:348-357, get_degePrimer_V6.py:260-271): a full pass over files of a gigabyte and more. */
int mp_file_count_newlines(const char *path, int32_t n_threads, int64_t *count);

/* ------------------------------------------------------------------------------------------------------------
 * (H2) per-window planning
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct mp_plan mp_plan;

typedef struct mp_plan_params {
    int32_t k;                   /* primer length (-l) */
    int32_t v;                   /* --variation */
    int32_t n_windows;
    int32_t n_threads;           /* 0 = one per host core (at most 32; MP_HOST_THREADS overrides) */
    int64_t total_sequences;     /* total_sequence_number (V20:713) */
    double coverage;             /* -f */
    double entropy_threshold;    /* after the length scaling of V20:642-649 */
    double max_degeneracy;       /* -d, score_of_dege_bases (V20:899-903) */
    int32_t max_dege_positions;  /* -n, number_of_dege_bases */
    int32_t keep_tables;         /* != 0: keep every window's ordered tables for mp_plan_window_table */
} mp_plan_params;

/* window status after planning */
#define MP_WIN_PLANNED 0
#define MP_WIN_GAP_GATE 1        /* round(gap_number / total, 2) >= 1 - coverage   (V20:713) */
#define MP_WIN_NO_COVER 2        /* len(cover) < 1                                  (V20:716) */
#define MP_WIN_ENTROPY 3         /* tBit > threshold                                (V20:723) */
#define MP_WIN_FEW_BASES 4       /* fewer than 4 bases in the frequency matrix      (V20:736) */
#define MP_WIN_GAP_COLUMN 5      /* an all-gap column                               (V20:738) */
#define MP_WIN_ENTROPY_DEVICE 6  /* tBit > threshold for certain, decided on the device from the window's histogram table before its entries
                                    were read back (mp_set_entropy_gate, mprime.h): the host would have stopped at V20:723 or at one of the two
                                    gates in front of it (gap fraction, empty cover); no entropy value is computed */

/* Builds, for every window, the insertion-ordered cover / gap_sequence tables from
 *   - histogram entries (any order, duplicates allowed: entries of several row shards are merged by key — counts
 *     add, the smallest first row wins): e_window[i] in [0, n_windows), e_words = b0 at [0,n), b1 at [n,2n),
 *     g at [2n,3n) (window words of mprime.h), e_count, e_first (GLOBAL row index of the first sighting);
 *   - exception k-mers (windows holding an IUPAC code, mp_get_exceptions): window, GLOBAL row, k symbol codes;
 *     each is expanded in the reference's order (itertools.product over the member lists of V20:105-107, last
 *     position fastest) unless it has more than v gaps (then it is a gap_sequence key as it stands, V20:689-698);
 * then applies the gates, computes the entropies, takes the seeds from freq [W][4][k] / nn [W][k-1][4][4]
 * (mp_window_stats, summed over shards) and derives the whole refinement chain of every seed. */
int mp_plan_create(const mp_plan_params *params, int64_t n_entries, const int32_t *e_window, const void *e_words,
                   const int64_t *e_count, const int64_t *e_first, int64_t n_exc, const int32_t *x_window,
                   const int64_t *x_row, const uint8_t *x_codes, const int64_t *freq, const int64_t *nn, mp_plan **out);
/* The same for the entries of ONE rank exactly as mp_get_unique returned them: window segments e_off [W+1] (entries of window w at
 * [e_off[w], e_off[w+1])), e_words as above, 32-bit counts and LOCAL first rows (row_base is added) — no per-entry window array, no
 * widening copies on the caller's side. */
int mp_plan_create_segments(const mp_plan_params *params, const int64_t *e_off, const void *e_words, const int32_t *e_count,
                            const int32_t *e_first, int64_t row_base, int64_t n_exc, const int32_t *x_window, const int64_t *x_row,
                            const uint8_t *x_codes, const int64_t *freq, const int64_t *nn, mp_plan **out);
/* The same straight from a context whose mp_window_unique has run (one rank, no JSON side files): the histogram entries are read
 * back in bands of windows while the planning threads already work on the bands that have arrived, instead of one blocking
 * mp_get_unique followed by mp_plan_create_segments — at 131072 x 1000 the read-back (58 MB) and the planning take about as long as each
 * other.  *n_entries returns the number of entries.  The plan is the same object, bit for bit (tests compare the two routes). */
struct mp_ctx;
int mp_plan_create_streamed(struct mp_ctx *ctx, const mp_plan_params *params, int64_t row_base, int64_t n_exc, const int32_t *x_window,
                            const int64_t *x_row, const uint8_t *x_codes, const int64_t *freq, const int64_t *nn, int64_t *n_entries,
                            mp_plan **out);
void mp_plan_destroy(mp_plan *p);
const char *mp_plan_error(const mp_plan *p);

/* status[W] (MP_WIN_*), cover_number[W], gap_number[W], cbit[W], tbit[W] (entropies are NaN where the window was
 * rejected before they were computed); any pointer may be NULL */
int mp_plan_windows(const mp_plan *p, int32_t *status, int64_t *cover_number, int64_t *gap_number, double *cbit, double *tbit);
/* number of planned windows and of candidates (all chain members of all seeds, window-ascending) */
int mp_plan_sizes(const mp_plan *p, int32_t *n_planned, int64_t *n_candidates);
/* the candidate list for mp_eval_candidates: cand_window[n] ascending, cand_codes[n][k] */
int mp_plan_candidates(const mp_plan *p, int32_t *cand_window, uint8_t *cand_codes);
/* seeds of window w: nm[k] / mm[k] base indices (0..3), *has_mm = 0 when there is no gap-free k-mer or MM == NM;
 * chain lengths (members incl. the seed) of both seeds */
int mp_plan_seeds(const mp_plan *p, int32_t w, uint8_t *nm, uint8_t *mm, int32_t *has_mm, int32_t *n_chain_nm, int32_t *n_chain_mm);

/* refinement chain of seed `seed` (0 = NM, 1 = MM) of window w: codes[n][k] (member 0 = the seed), the running perfect
 * coverage cov[n] and stops[n] (1 = a structural break rule of V20:899-904 ends the loop after this member).
 * *n returns the length; MP_ERR_CAPACITY if cap is too small. */
int mp_plan_chain(const mp_plan *p, int32_t w, int32_t seed, int32_t cap, uint8_t *codes, int64_t *cov, uint8_t *stops, int32_t *n);

/* Replays coverage_stast's stopping rules (V20:881-906) on ev [n_candidates][3] = {perfect, F_mis, R_mis} of
 * mp_eval_candidates (summed over shards), picks NM or MM (V20:816) and counts the nonsense expansions
 * (V20:846).  Fails with MP_ERR_ARG if a candidate's perfect coverage differs from the host's running sum
 * (the two are the same quantity). */
int mp_plan_finish(mp_plan *p, const int64_t *ev);
/* per planned window, ascending: window index, entropies (rounded as the reference prints them), the chosen primer's
 * symbol codes [n][k], its perfect coverage, F / R mis-coverage (perfect + admissible), nonsense_primer_number,
 * number of degenerate positions, cover_number */
int mp_plan_results(const mp_plan *p, int32_t *window, double *cbit, double *tbit, uint8_t *primer_codes, int64_t *cov,
                    int64_t *f_mis, int64_t *r_mis, int32_t *nonsense, int32_t *n_dege, int64_t *cover_number);

/* Ordered table of window w (needs keep_tables): which = 0 the cover dict, 1 gap_sequence; codes[n][k], counts[n],
 * first_row[n] in insertion order.  *n returns the size; MP_ERR_CAPACITY if cap is too small. */
int mp_plan_window_table(const mp_plan *p, int32_t w, int32_t which, int64_t cap, uint8_t *codes, int64_t *counts,
                         int64_t *first_row, int64_t *n);

/* Writes the two JSON side files of the core step (V20:1172-1177) — {out}.non_coverage_seq_id_json = {pos: [{k-mer: [ids]} (F),
 * {k-mer: [ids]} (R)]} and {out}.gap_seq_id_json = {pos: {expanded gap k-mer: [ids]}} — byte for byte what the reference's
 * json.dump(obj, fh, indent=4) writes (dict insertion orders, ids in file order, encode_basestring_ascii escapes), for the
 * n_out output windows out_window[] (ascending) at positions out_pos[] with final primers primer_codes [n_out][k]:
 * a cover k-mer is listed under F (R) when the primer misses it in 1..v positions one of which is F- (R-) strict, or in more
 * than v positions (V20:1107-1127).  The plan must have been created with keep_tables.  Sequences come from
 *   dev_off [W+1] / dev_words (b0 at [0,n_dev), b1, g) — the histogram entries as mp_get_unique returned them,
 *   labels [n_out][n_rows] — mp_get_labels of each output window (index of the row's entry inside its window, -1 = none),
 *   the exception list (x_window, x_row, x_codes) and the ids (raw bytes + offsets, decoded as UTF-8 / surrogateescape). */
int mp_plan_write_side_files(const mp_plan *p, int32_t n_out, const int32_t *out_window, const int64_t *out_pos,
                             const uint8_t *primer_codes, uint64_t strictF, uint64_t strictR, const int64_t *dev_off,
                             const void *dev_words, int64_t n_dev, const int32_t *labels, int32_t n_rows, int64_t n_exc,
                             const int32_t *x_window, const int64_t *x_row, const uint8_t *x_codes, const uint8_t *ids,
                             const int64_t *id_off, const char *noncov_path, const char *gap_path);
/* The same files written in several calls, each for a run of consecutive output windows with the labels of just those windows (a
 * 10^6-row alignment with 900 output windows would otherwise need 3.6 GB of labels at once): part bit 0 = this is the first run (the
 * files are created), bit 1 = the last one (the objects are closed); part = 3 is mp_plan_write_side_files.  Same bytes. */
int mp_plan_write_side_files_part(const mp_plan *p, int32_t n_out, const int32_t *out_window, const int64_t *out_pos,
                                  const uint8_t *primer_codes, uint64_t strictF, uint64_t strictR, const int64_t *dev_off,
                                  const void *dev_words, int64_t n_dev, const int32_t *labels, int32_t n_rows, int64_t n_exc,
                                  const int32_t *x_window, const int64_t *x_row, const uint8_t *x_codes, const uint8_t *ids,
                                  const int64_t *id_off, const char *noncov_path, const char *gap_path, int32_t part);

/* Expansions of n k-mers of symbol codes (degenerate_seq, V20:368-380) in the reference's order; out_src[i] = index
 * of the k-mer expansion i comes from.  *n_out returns the number needed; MP_ERR_CAPACITY if cap is too small. */
int mp_expand_kmers(int32_t k, int64_t n, const uint8_t *codes, int64_t cap, uint8_t *out_codes, int64_t *out_src, int64_t *n_out);
/* (H4) the per-primer numbers of the TSV for all output primers at once (csrc/primerstats.cpp): primer i = codes[i*k .. i*k+k), IUPAC symbol
 * codes, no gaps, at most 2^22 expansions each.
 * mp_primer_tm: tm[i] = round(mean over the expansions e of round(Calc_Tm_v2(e), 2), 2) (V20:849-852, 282-336).  `params` holds the
 *   reference's tables and constants as its own expressions evaluate them (multiprime_amd/thermo.py): dH[cur][prev] (16), dS[cur][prev] (16),
 *   dH of an end base A,C,G,T (4), dS of an end base (4), the symmetry term of dS, R ln(c) for a self-complementary and for any other
 *   sequence, the salt correction, 273.15 — 45 doubles.  Python's round() and statistics.mean (exact) are reproduced bit for bit.
 * mp_primer_filters: gc[i] = round(mean of r3[#G+#C of e], 2) with r3[g] = round(g / k, 3) supplied by the caller (k + 1 doubles,
 *   V20:401-407), repeat[i] = di_nucleotide (V20:410-416), hairpin[i] = hairpin_check at `distance` (V20:387-398). */
int mp_primer_tm(int32_t k, int64_t n, const uint8_t *codes, const double *params, double *tm);
int mp_primer_filters(int32_t k, int64_t n, const uint8_t *codes, const double *r3, int32_t distance, double *gc, uint8_t *repeat, uint8_t *hairpin);

/* (H5) coverage-bitset verdicts of the rows whose window held an IUPAC code (the hand-off to the pairing stage; V20:701-707 files
 * the row's id under every expansion's k-mer, V20:1107-1127 decides per k-mer).  Exception i: the row's symbol codes xc[i*k .. i*k+k) (0 = '-')
 * against output primer primer_of[i] (codes primers[p*k .. p*k+k), p < n_primers).  bad[2i] / bad[2i+1] = 1 when the forward / reverse primer
 * does NOT reach the row: more than v gaps, more than v positions that can mismatch (a gap, or a member of the row's symbol outside the
 * primer's), or a strict position (bit j of strictF / strictR) that can mismatch.  A few host threads from 16384 exceptions up. */
int mp_exception_verdicts(int32_t k, int32_t v, int64_t n, const uint8_t *xc, const int64_t *primer_of, int64_t n_primers, const uint8_t *primers,
                          uint64_t strictF, uint64_t strictR, uint8_t *bad);

/* (H5b) the verdicts of (H5) as the assignments mp_masks_set_bits (mprime.h) takes, the selection included (core.py _resident_bitsets: the
 * numpy selection + repeat / tile around mp_exception_verdicts cost 2.5 ms at 10^6 rows beside 0.1 ms of verdicts).  Of the n exception rows
 * (window x_window[i] < n_windows, global row x_row[i], codes xc[i*k ..)) those with slot_of[x_window[i]] >= 0 (the window's output row; its
 * primer is primers[slot*k ..), slot < n_primers) and row0 <= x_row[i] < row0 + n_rows give two assignments each, in exception order:
 * cand = slot, row = x_row[i] - row0, which = 0 / 1 (forward / reverse), value = the verdict.  The four outputs hold 2n entries;
 * *n_out = entries written. */
int mp_exception_assignments(int32_t k, int32_t v, int64_t n, const int32_t *x_window, const int64_t *x_row, const uint8_t *xc, int32_t n_windows,
                             const int32_t *slot_of, int64_t row0, int64_t n_rows, int64_t n_primers, const uint8_t *primers, uint64_t strictF,
                             uint64_t strictR, int32_t *cand, int32_t *row, uint8_t *which, uint8_t *value, int64_t *n_out);

/* The same expansions as window words (b0, b1, g of mprime.h, three per expansion) — what mp_set_extra_rows takes. */
int mp_expand_kmer_words(int32_t k, int64_t n, const uint8_t *codes, int64_t cap, void *out_words, int64_t *out_src,
                         int64_t *n_out);
/* [r6] The same for the exception list of mp_get_exceptions as it stands: rows with more than v gaps (code 0) are dropped — they are
 * gap_sequence entries, not k-mers (V20:689-707) —, every expansion comes with its row's window x_window[i] instead of the row index:
 * (out_window, out_words) is what mp_set_extra_rows takes (core.py did the selection and the indexing in numpy: 4.7 ms of a helper
 * thread's 8 at 10^6 rows).  MP_ERR_CAPACITY with *n_out = the expansions there are when cap is too small. */
int mp_expand_exception_words(int32_t k, int32_t v, int64_t n, const int32_t *x_window, const uint8_t *codes, int64_t cap, void *out_words,
                              int32_t *out_window, int64_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* MPRIME_HOST_H */
