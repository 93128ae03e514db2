/* chimeralm_feed.h -- C ABI of the native BAM feeder of the MI355X predict path.
 *
 * Replaces, behind plain pointers and sizes, the host data path the reference runs in Python before every predict batch
 * (file:line under /root/reference/chimeralm/):
 *   data/bam.py:21-38       is_chimeric + parse_bam_file: BAM records that are mapped, primary (neither secondary nor
 *                           supplementary) and carry an SA tag, in file order, as (query_name, sequence)
 *   data/tokenizer.py:85-114 tokenize_and_align_labels_and_quals_ids: characters -> token ids, truncation to the model
 *                           length, ONE trailing [SEP]; read name -> id row [len, code points ..., 0 ...] of 256 entries
 *   data/tokenizer.py:136-187 DataCollator.torch_call: pad to the longest read of the batch with [PAD] on the
 *                           tokenizer's padding side; id rows as int8 [B, 256]
 *   data/bam.py:142-174,287-299 batches of batch_size // world_size reads per device, rank r taking selected reads
 *                           r, r + world, r + 2 world, ... (the non-shuffling distributed sampler Lightning installs)
 * Worker threads inflate the BGZF blocks (zlib; the members are independent deflate streams), a decoder thread takes them in
 * file order, decodes records, tokenises straight from the 4-bit base codes and fills batches into a ring of (pinned) host slots; the consumer takes them in order and hands each slot back when its H2D copy
 * has been enqueued and completed (clm_stage_ids in chimeralm_hip.h does that copy on the engine's side stream).
 *
 * Every call returns 0 / a positive count on success and a negative CLM_E_* code on failure (chimeralm_hip.h);
 * clm_feeder_last_error() gives the message.  No exceptions cross the ABI.  One consumer thread per feeder.
 */
#ifndef CHIMERALM_FEED_H
#define CHIMERALM_FEED_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct clm_feeder clm_feeder;

typedef struct clm_feeder_config {
    int32_t struct_size;      /* sizeof(clm_feeder_config), checked */
    int32_t batch_size;       /* reads per batch on THIS device (reference: batch_size // world_size)           */
    int32_t max_tokens;       /* truncation length in tokens incl. the trailing [SEP]; the reference passes       */
                              /* tokenizer.max_len_single_sentence = model_max_length - 1 = 32769 (bam.py:155-166) */
    int32_t slots;            /* ring depth, >= 2                                                              */
    int32_t rank, world;      /* this device takes selected reads rank, rank + world, ...                      */
    int32_t pad_left;         /* 1: pad on the left (HyenaDNA tokenizer), 0: on the right                       */
    int32_t pinned;           /* 1: slots in page-locked host memory (hipHostMalloc; needs a HIP device)         */
                              /* 0: plain host memory (CPU-side tests of the decoder)                           */
    int64_t max_reads;        /* stop after this many SELECTED reads of the file (all ranks together); < 0: all  */
    int32_t inflate_threads;  /* BGZF members are inflated by this many worker threads, in file order for the     */
    int32_t reserved;         /* decoder (0: chosen from the host's core count and `world`, at most 8)            */
} clm_feeder_config;

typedef struct clm_feed_batch {
    int32_t slot;             /* give back with clm_feeder_release                                               */
    int32_t n_reads;          /* B (the last batch of a file may be short)                                       */
    int32_t n_tokens;         /* L = longest read of the batch incl. [SEP]                                        */
    int32_t reserved;
    int64_t row_stride;       /* elements between rows of ids (= n_tokens)                                        */
    const uint8_t* ids;       /* [B][row_stride] token ids, padded with [PAD] = 4                                  */
    const int8_t* names;      /* [B][256] id rows exactly as the reference collator builds them                   */
    int64_t first_index;      /* index, among this rank's reads, of row 0                                         */
} clm_feed_batch;

/* fills cfg with the reference's defaults: batch 12, 32769 tokens, 4 slots, rank 0 of 1, left padding, pinned, automatic
 * number of inflate threads */
int clm_feeder_default_config(clm_feeder_config* cfg);
/* opens the file, checks the BAM magic, starts the decoder thread */
int clm_feeder_open(const char* bam_path, const clm_feeder_config* cfg, clm_feeder** out);
/* next batch in file order: 1 = `out` filled, 0 = end of file (all batches delivered), < 0 = error (corrupt file, ...) */
int clm_feeder_next(clm_feeder* f, clm_feed_batch* out);
/* the consumer is done with the slot's memory (its copy to the device has completed) */
int clm_feeder_release(clm_feeder* f, int32_t slot);
/* counters so far: records seen, reads selected (all ranks), reads delivered to this rank, bases truncated away */
int clm_feeder_stats(const clm_feeder* f, int64_t* records, int64_t* selected, int64_t* delivered, int64_t* truncated);
const char* clm_feeder_last_error(const clm_feeder* f);   /* f may be NULL: error of the last failed clm_feeder_open */
int clm_feeder_close(clm_feeder* f);                       /* stops the thread, frees the ring; f may be NULL       */

/* ---- after predict: /root/reference/chimeralm/__main__.py:99-153 (filter_bam_by_predcition) ------------------------------
 * clm_bam_filter copies `in_bam` to `out_bam` (same header, same order) without the records whose read name is in `drop_names`
 * -- every record of such a read, primary or not, as the reference does (:131-134); counts of kept / dropped RECORDS come back.
 * clm_bam_sort_index is the `pysam.sort` + `pysam.index` pair of :147-153: coordinate order (reference id with unplaced reads
 * last, position, strand; stable), @HD SO:coordinate, and a BAI index next to the output (`out_bai` NULL: <out>.bai).
 * The sort works in memory up to a budget (CLM_SORT_MEM_MB; default a quarter of the host's RAM, 256 MiB .. 16 GiB) and beyond
 * it spills sorted runs to `<out>.tmp.N.run` files and merges them (same stable order whatever the budget), like `samtools
 * sort -m`.  Errors: negative CLM_E_* code, text from clm_bam_last_error() (per thread).
 * clm_bam_filter_ex adds: `flags` -- CLM_BAM_INPUT_SAM: `in_path` is SAM text (the reference opens any suffix but .bam in mode
 * "r", :127); CLM_BAM_KEEP_UNPLACED: also copy records without a reference placement (refID < 0).  By default those are left
 * out of a BAM's output and counted in `unplaced`: the reference walks the input with `bam_file.fetch()` (:131), which for a BAM
 * goes through the index reference by reference and never yields them; for SAM text it yields every record, so
 * CLM_BAM_INPUT_SAM implies keeping them.  clm_bam_filter is clm_bam_filter_ex with flags = 0. */
#define CLM_BAM_INPUT_SAM 1
#define CLM_BAM_KEEP_UNPLACED 2
int clm_bam_filter(const char* in_bam, const char* out_bam, const char* const* drop_names, int64_t n_drop, int64_t* kept,
                   int64_t* dropped);
int clm_bam_filter_ex(const char* in_path, const char* out_bam, const char* const* drop_names, int64_t n_drop, int flags,
                    int64_t* kept, int64_t* dropped, int64_t* unplaced);
int clm_bam_sort_index(const char* in_bam, const char* out_sorted_bam, const char* out_bai, int64_t* n_records);
const char* clm_bam_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
