/*
 * pq2hip -- MI355X (gfx950) decoder for Parquet's RLE/bit-packing hybrid
 * encoding and plain bit-unpacking, behind a C ABI.
 *
 * This header is the drop-in boundary: every entry point names the public
 * item of jorgecarleitao/parquet2 (v0.17.2) that it replaces.  Paths are
 * relative to the reference root.  The reference has no FFI of its own; its
 * boundary is the public Rust API re-exported at
 *   src/encoding/hybrid_rle/mod.rs:5-7, src/encoding/bitpacked/mod.rs:6-7 and
 *   src/deserialize/mod.rs:9-15.
 *
 * Conventions
 *  - plain pointers and sizes only; the caller owns every buffer it passes;
 *  - `d_*` arguments are device pointers on the context's GPU, `h_*` / bare
 *    pointers are host pointers;
 *  - device input buffers (`d_in`, `d_buf`, `d_packed`): the kernels read a
 *    stream in aligned 16-byte pieces, so they touch up to 15 bytes in front
 *    of a stream's first byte and up to 15 bytes behind its last one (never
 *    beyond the 16-byte line that holds that byte).  Those bytes have to be
 *    readable: allocate the buffer 16-byte aligned (hipMalloc does) and 16 bytes
 *    larger than `in_bytes`, or keep every stream inside 16-byte-aligned bounds of
 *    a larger allocation.  Their contents do not matter.  The host entry points
 *    pad their own staging copies;
 *  - every function returns a pq2 status (0 = OK); decode entry points also
 *    fill one status per page so that one malformed page does not poison a
 *    batch;
 *  - a context is bound to one GPU and is not thread-safe; use one context per
 *    (thread, GPU).  Distinct contexts may be used concurrently;
 *  - decoding always runs on the GPU.  There is no CPU fallback: if no HIP
 *    device is usable, pq2_ctx_create fails with PQ2_DEVICE_ERROR.
 */
#ifndef PQ2HIP_H
#define PQ2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: src/error.rs:23-34 (Error) plus two of our own -------- */
#define PQ2_OK 0
#define PQ2_OUT_OF_SPEC 1           /* Error::OutOfSpec */
#define PQ2_INVALID_PARAMETER 2     /* Error::InvalidParameter */
#define PQ2_FEATURE_NOT_SUPPORTED 3 /* Error::FeatureNotSupported */
#define PQ2_WOULD_OVER_ALLOCATE 4   /* Error::WouldOverAllocate */
#define PQ2_REFERENCE_WOULD_PANIC 5 /* defined, non-crashing answer for inputs on which the reference panics */
#define PQ2_DEVICE_ERROR 6          /* HIP runtime failure; see pq2_last_error */

/* Values are produced in blocks of this many u32; per-page output offsets
 * that are multiples of 4 values keep every store 16-byte aligned. */
#define PQ2_BLOCK_VALUES 1024u
/* Largest num_values / packed length a single page may have. */
#define PQ2_MAX_PAGE_VALUES 0x7fffffffu
#define PQ2_MAX_PAGE_BYTES 0xfff00000u

typedef struct pq2_ctx pq2_ctx;

/* One encoded stream ("page") of a batch: the arguments of
 * HybridRleDecoder::try_new(data, num_bits, num_values)
 * (src/encoding/hybrid_rle/mod.rs:68) plus where the decoded values go. */
typedef struct pq2_page_desc {
    uint64_t in_off;     /* byte offset of the stream inside the input buffer */
    uint32_t in_len;     /* stream length in bytes (`data.len()`) */
    uint32_t num_bits;   /* 0..=32 */
    uint64_t num_values; /* number of u32 to produce (`num_values`) */
    uint64_t out_off;    /* index (in u32) of the first output value */
} pq2_page_desc;

/* ---- context ------------------------------------------------------------ */
/* A context belongs to one GPU and is driven by one thread at a time.  Every
 * call that takes a context makes that GPU the calling thread's current device
 * (hipSetDevice) and leaves it so: a caller that works with several GPUs sets
 * its own device again before its next HIP call.  Streams handed to a call
 * must belong to the context's GPU. */
int32_t pq2_ctx_create(int32_t device_id, pq2_ctx **ctx);
int32_t pq2_ctx_destroy(pq2_ctx *ctx);
const char *pq2_strerror(int32_t status);
/* message of the last failing call on this context ("" if none) */
const char *pq2_last_error(const pq2_ctx *ctx);

/* Kernel timing for benchmarks (no counterpart in the reference).  While
 * enabled, every kernel launch of a batch is bracketed by HIP events on the
 * stream it runs on (a large batch is decoded in several rounds, each one
 * parse and one expand launch, on two internal streams).
 * pq2_ctx_kernel_times waits for them, returns the summed durations in
 * milliseconds with the number of batches and (if expand_launches is not
 * NULL) of expand launches, and resets the sums. */
int32_t pq2_ctx_enable_profiling(pq2_ctx *ctx, int32_t on);
/* Name of the kernel that produced the values of the last batch on this context ("ring_kernel": one
 * workgroup per page, the stream staged once through LDS; "decode_kernel": one workgroup per page, hand-off
 * through device memory; "expand_kernel": one wave per 4096 values behind parse_kernel;
 * "scout+expand_kernel": the same behind the scout kernels, which walk the headers of all pages in parallel over
 * tiles of the streams; "" before the first batch).  For benchmark reports. */
const char *pq2_ctx_last_schedule(const pq2_ctx *ctx);
/* Pages handed to pq2_decode_batch_device and pq2_decode_batch_host on this context since it was created (the
 * pages of a pq2_multi_decode call count for the context that got them).  For tests and reports: which part of a
 * batch a context was given.  No counterpart in the reference. */
uint64_t pq2_ctx_pages_total(const pq2_ctx *ctx);
/* What the default schedule went by: the largest number of run headers it expected of a page of the last batch, if
 * it looked at the streams before it chose (batches of 96 to 4096 pages that the descriptors send to ring_kernel:
 * a wave per page walks the headers of the stream's first 500 bytes, and a batch whose pages would take their one
 * walking wave longer than the scout kernels take for all of them goes to those); 0 if it did not look.  The
 * environment variable PQ2_DENSITY_PROBE=0 turns the look off.  No counterpart in the reference
 * (src/encoding/hybrid_rle/decoder.rs:30-64 is one decoder for every stream). */
uint32_t pq2_ctx_last_density(const pq2_ctx *ctx);
/* How the batches of this context are scheduled: "auto" (the default: the scout kernels for fewer than 96 pages, unless
 * every stream has at most 1 KiB, and
 * for up to 1024 pages that they are expected to get through before ring_kernel is done with its largest page,
 * parse_kernel + expand_kernel for more than 1024 pages of 8192 values or fewer on average, ring_kernel otherwise),
 * "scout", "ring", "tile" (parse_kernel + expand_kernel in rounds),
 * "page" (decode_kernel).  Every schedule gives the same values and statuses; which one is fastest depends on how
 * dense the run headers are, which the library cannot see from the descriptors: pages of very short runs (a few
 * values each) are faster through "scout" whatever their number, pages of long runs through "ring" from about a
 * hundred pages on.  The environment variable PQ2_MODE sets the same for every context created after it.
 * PQ2_INVALID_PARAMETER for another name.  No counterpart in the reference. */
int32_t pq2_ctx_set_schedule(pq2_ctx *ctx, const char *name);
/* Diagnostics of the scout kernels: for every page of the last batch on this context, the number of rounds its
 * resolve step took until every tile of the page's stream was entered where the one before it is left (0 for a
 * page without headers to walk).  Waits for the batch.  PQ2_INVALID_PARAMETER if the last batch was not decoded
 * through the scout kernels or n_pages is not its page count.  No counterpart in the reference. */
int32_t pq2_ctx_scout_rounds(pq2_ctx *ctx, uint32_t *rounds, uint64_t n_pages);
int32_t pq2_ctx_kernel_times(pq2_ctx *ctx, double *parse_ms, double *expand_ms, uint64_t *batches,
                             uint64_t *expand_launches);

/* ---- hybrid RLE / bit-packed decode -------------------------------------- */

/* Replaces, for every page i of the batch,
 *   HybridRleDecoder::try_new(&in[in_off..in_off+in_len], num_bits, num_values)?
 *       .collect::<Result<Vec<u32>, _>>()
 * (src/encoding/hybrid_rle/mod.rs:66-116, built on decoder.rs:8-65,
 * bitpacked/decode.rs:9-86, bitpacked/unpack.rs:21-95, uleb128.rs:3-23) with
 * input and output resident in HBM.  Exactly num_values u32 are written per
 * page at d_out[out_off..]; the stream is zero-padded when it is exhausted,
 * RLE values are not masked to num_bits, and a header that ends the stream is
 * dropped, exactly as the reference does.
 *
 * The work is enqueued on `stream` (a hipStream_t, NULL = default stream).
 * If `status` is non-NULL the call waits for completion and writes one status
 * per page; if it is NULL the call returns after enqueueing and statuses can
 * be collected later with pq2_fetch_status.  Page statuses:
 *   PQ2_OK, PQ2_REFERENCE_WOULD_PANIC (an RLE run without its value bytes,
 *   decoder.rs:59-60; a 10-byte varint whose last byte is > 1, uleb128.rs:10-12;
 *   a bit-packed header whose byte count overflows, decoder.rs:50),
 *   PQ2_INVALID_PARAMETER (num_bits > 32: unpack32 has no such width,
 *   unpack.rs:87).  The values of a page with a non-OK status are unspecified
 *   (the call stays memory-safe).
 * Returns non-zero only for errors of the call itself (bad descriptor bounds,
 * device failure). */
int32_t pq2_decode_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes, const pq2_page_desc *descs,
                                uint64_t n_pages, uint32_t *d_out, uint64_t out_values, int32_t *status,
                                void *stream);

/* Waits for the last batch enqueued on the context and copies its per-page
 * statuses (n_pages must match that batch). */
int32_t pq2_fetch_status(pq2_ctx *ctx, int32_t *status, uint64_t n_pages);

/* Same contract with host buffers: copies the streams to the GPU, decodes,
 * copies the values back.  Batches of 32 MiB and more (input + output) go in
 * chunks of whole pages, about 16 Mi values each, through three internal
 * streams, so that the upload of one chunk, the kernels of another and the
 * download of a third overlap (the copies are asynchronous only from and to
 * page-locked host memory).  This is what a per-page drop-in of
 * HybridRleDecoder / deserialize::utils::dict_indices_decoder
 * (src/deserialize/utils.rs:13-27) calls. */
int32_t pq2_decode_batch_host(pq2_ctx *ctx, const void *h_in, uint64_t in_bytes, const pq2_page_desc *descs,
                              uint64_t n_pages, uint32_t *h_out, uint64_t out_values, int32_t *status);

/* ---- the batch decode without allocation ------------------------------------- */

/* No counterpart in the reference.  For callers that cannot allocate: code
 * under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  The decode of pq2_decode_batch_device through the kernel that
 * takes its large batches (ring_kernel: one workgroup per page, walk and
 * expansion meet in LDS), with the device memory that kernel needs beyond the
 * inputs and outputs (its page descriptors, two status arrays, the
 * per-compute-unit tickets) inside one fixed array of the library's code
 * object, which every GPU the library runs on holds until the process ends
 * (about 340 KiB).  The call allocates nothing and frees nothing; it only
 * enqueues one memset and three kernels on `stream` and returns.
 *
 * Differences from pq2_decode_batch_device:
 *  - no context: `device_id` names the GPU, and `stream` must belong to it (the
 *    calling thread's current device is the same after the call as before);
 *  - `d_descs` and `d_status` are DEVICE arrays of n_pages entries (the caller's
 *    own buffers; d_descs 8-byte aligned): the descriptors are checked on the
 *    device.  A descriptor the host entry point would reject (stream or output
 *    outside the buffers, num_bits > 32 or > max_bits) fails its own page with
 *    PQ2_INVALID_PARAMETER instead of the call; the other pages are decoded;
 *  - `max_bits`: an upper bound of the pages' bit widths (0 = unknown: 32 is
 *    assumed; a bound of 16 or less selects the faster kernel for narrow pages);
 *  - limit: n_pages <= PQ2_NOALLOC_MAX_PAGES, else the call returns
 *    PQ2_WOULD_OVER_ALLOCATE and enqueues nothing;
 *  - every call on one GPU uses that GPU's one array: calls on one stream
 *    follow each other safely, calls that overlap on one GPU (two streams, two
 *    threads) race.  The call clears what has to start at zero itself.
 * The statuses are complete when the work on `stream` is. */
#define PQ2_NOALLOC_MAX_PAGES 4096u
int32_t pq2_decode_batch_device_noalloc(int32_t device_id, const void *d_in, uint64_t in_bytes,
                                        const pq2_page_desc *d_descs, uint64_t n_pages, uint32_t max_bits,
                                        uint32_t *d_out, uint64_t out_values, int32_t *d_status, void *stream);

/* ---- definition levels of flat nullable columns: validity bitmaps ----------- */

/* DefLevelsDecoder::Bitmap made whole.  For max_def_level == 1 the reference
 * never expands levels: DefLevelsDecoder::try_new
 * (src/deserialize/utils.rs:41-48) wraps the run iterator in HybridRleIter
 * (src/deserialize/hybrid_rle.rs:54-115), which hands the consumer
 * Bitmap(slice, len) for a bit-packed run (len = min(slice.len() * 8,
 * remaining), :97-105) and Repeated(value[0] == 1, min(run, remaining)) for an
 * RLE run (:106-113), and the consumer appends them to its validity bitmap.
 * This call does that for a batch of pages on the device: page i's level
 * stream (descs[i]: num_bits must be 1, num_values = the page's rows) becomes
 * a bitmap at d_bitmap + descs[i].out_off -- out_off counts BYTES here and is a
 * multiple of 4 -- with row r at bit r % 8 of byte r / 8 (LSB first, as
 * BitmapIter reads, src/encoding/hybrid_rle/bitmap.rs:46-62), written in whole
 * 32-bit words (a page takes 4 * ceil(num_values / 32) bytes; the bits behind
 * its last row are 0).
 * The iterator ends with the stream (hybrid_rle.rs:90-94: no zero padding,
 * unlike HybridRleDecoder): rows[i] receives the rows the stream holds (at most
 * num_values; the bits of the rows behind them are 0) and set_counts[i] the
 * number of set bits, i.e. of valid rows.  Either may be NULL.  An RLE run
 * whose value byte is not 1 repeats `false`, whatever its low bit (:107-108).
 * Statuses as pq2_decode_batch_device (only the 10-byte varint of
 * uleb128.rs:10-12 can fail a one-bit stream).  If rows, set_counts or status
 * is non-NULL the call waits for the batch. */
int32_t pq2_decode_validity_bitmap_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes,
                                                const pq2_page_desc *descs, uint64_t n_pages, void *d_bitmap,
                                                uint64_t bitmap_bytes, uint32_t *rows, uint32_t *set_counts,
                                                int32_t *status, void *stream);

/* ---- dictionary pages: indices -> values ---------------------------------- */

/* Where a page's dictionary lies inside the dictionary buffer: `dict_len`
 * PLAIN-encoded little-endian values of `value_size` bytes each from byte
 * `dict_off` on (a decompressed dictionary page's buffer as it is:
 * tests/it/read/dictionary/primitive.rs:30-50, src/types.rs:135-141).  Pages of
 * one column chunk share one dictionary. */
typedef struct pq2_dict_ref {
    uint64_t dict_off; /* multiple of 8 for 8-byte values, of 4 otherwise */
    uint32_t dict_len; /* entries */
    uint32_t reserved;
} pq2_dict_ref;

/* Decodes the dictionary indices of a batch of pages as
 * pq2_decode_batch_device does and replaces every index by its dictionary
 * value: what a consumer does with NativePageState::RequiredDictionary /
 * OptionalDictionary (src/deserialize/native.rs:28-69; the consumer loop
 * `dict.indexes.map(|x| dict.dict.value(x))`, tests/it/read/primitive.rs:99-108).
 * `value_size` is 4 (i32, f32), 8 (i64, f64) or 12 (int96); page i's values go to
 * d_out + descs[i].out_off * value_size (d_out 8-byte aligned).  Where the
 * lookup happens inside the decode kernel (pages of at most 16 bits, in batches
 * that take ring_kernel with dictionaries of at most 48 KiB, or that one round
 * of expand_kernel carries: few pages, or pages of dense headers) the indices
 * never exist in device memory; otherwise they pass through a buffer the
 * context owns.  An index that is not below
 * dict_len is Error::OutOfSpec in the reference
 * (tests/it/read/dictionary/primitive.rs:20-27): the page gets status
 * PQ2_OUT_OF_SPEC (unless the index stream itself already failed) and the
 * value written for that index is zero. */
int32_t pq2_decode_dict_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes, const pq2_page_desc *descs,
                                     uint64_t n_pages, const void *d_dict, uint64_t dict_bytes,
                                     const pq2_dict_ref *dicts, uint32_t value_size, void *d_out, uint64_t out_values,
                                     int32_t *status, void *stream);

/* Optional (nullable) dictionary-encoded pages of a flat column: decodes the
 * definition levels (def_descs[i]: the page's level stream, num_bits =
 * get_bit_width(max_def_levels[i]), num_values = rows) and the dictionary
 * indices (idx_descs[i]: num_bits = the page's index width, num_values = the
 * page header's num_values; its out_off is ignored), and writes for every row
 * r of page i, at d_out + (def_descs[i].out_off + r) * value_size and
 * d_validity + def_descs[i].out_off + r:
 *   level[r] == max_def_levels[i]: the dictionary value of the page's next unused
 *                                  index, validity 1;
 *   otherwise:                     zero, validity 0.
 * This is deserialize_optional over DefLevelsDecoder::Levels and
 * OptionalDictionary (src/deserialize/utils.rs:41-57, native.rs:28-69; the
 * consumer tests/it/read/utils.rs:7-66 with tests/it/read/primitive.rs:104-110):
 * indices are consumed by valid rows only, so an index beyond the dictionary
 * fails the page (PQ2_OUT_OF_SPEC) only if a valid row consumes it, and a
 * valid row behind the last index is null (`values.next()` is None).  For
 * max_def_level 1 the reference takes the levels as bitmap runs
 * (DefLevelsDecoder::Bitmap), which yields the same rows except that a level
 * stream that ends early yields fewer rows there and null rows here.
 * status[i]: the level stream's status if not OK, else the index stream's,
 * else PQ2_OUT_OF_SPEC for a consumed index beyond the dictionary.
 * Batches of 256 pages and more whose pages all have max_def_level 1 do not
 * expand their levels (validity bitmaps, as pq2_decode_validity_bitmap_batch_device
 * makes them), and with indices of at most 12 bits and dictionaries of at most
 * 16 KiB the decode kernel writes the rows itself: neither levels nor indices
 * pass through device memory. */
int32_t pq2_decode_optional_dict_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes,
                                              const pq2_page_desc *def_descs, const pq2_page_desc *idx_descs,
                                              const uint32_t *max_def_levels, uint64_t n_pages, const void *d_dict,
                                              uint64_t dict_bytes, const pq2_dict_ref *dicts, uint32_t value_size,
                                              void *d_out, uint8_t *d_validity, uint64_t out_rows, int32_t *status,
                                              void *stream);

/* ---- selected rows ---------------------------------------------------------- */

/* indexes::Interval (src/indexes/intervals.rs:11-16): `length` rows from row `start`. */
typedef struct pq2_interval {
    uint64_t start;
    uint64_t length;
} pq2_interval;

/* SliceFilteredIter::new(HybridRleDecoder::try_new(...), selected_rows)
 * (src/deserialize/utils.rs:92-148) for a batch of pages: page i keeps the
 * values of its intervals intervals[interval_offsets[i] .. interval_offsets[i + 1]),
 * one interval after the other, at d_out + descs[i].out_off (descs[i].num_values
 * stays the page's full count; the caller lays the pages out by their selected
 * counts).  The whole page is decoded into a buffer the context owns and the
 * selected ranges are copied from there.  The reference's iterator needs
 * ascending, non-overlapping intervals of at least one row (it subtracts
 * without checking) and ends early at the page's end; here such a list, or an
 * interval that reaches beyond num_values, makes the call fail with
 * PQ2_INVALID_PARAMETER before anything is enqueued. */
int32_t pq2_decode_selected_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes, const pq2_page_desc *descs,
                                         uint64_t n_pages, const pq2_interval *intervals,
                                         const uint64_t *interval_offsets, uint32_t *d_out, uint64_t out_values,
                                         int32_t *status, void *stream);

/* FilteredHybridBitmapIter::new(HybridDecoderBitmapIter::new(..), selected_rows)
 * (src/deserialize/filtered_rle.rs:53-274, :8-10) for a batch of pages: the
 * definition levels of optional flat columns (bit width 1: descs[i].num_bits
 * must be 1) intersected with the selected row intervals.  What the iterator
 * hands a reader run by run comes out whole:
 *   d_validity  one byte per selected row (1: the row holds a value), the rows of
 *               page i's intervals one after the other from
 *               d_validity[descs[i].out_off] on -- the bits of its Bitmap and
 *               Repeated items;
 *   d_counts    two words per interval, in the order of `intervals`:
 *               d_counts[2k] = valid rows between the end of the interval before
 *               (the page's first row for a page's first interval) and interval
 *               k's first row -- the sum of the Skipped(set_count) items in
 *               front of it, i.e. how many values the page's value decoder has
 *               to pass over; d_counts[2k + 1] = valid rows inside interval k.
 * Rows behind a page's last interval are not looked at, as the iterator ends
 * there.  Where a level stream holds fewer rows than its page, the reference's
 * iterator ends with the stream's last run; here the missing rows are null
 * rows (the zeros HybridRleDecoder pads with, mod.rs:95).  Intervals as for pq2_decode_selected_batch_device.  status[i]: the
 * level stream's status. */
int32_t pq2_filtered_validity_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes, const pq2_page_desc *descs,
                                           uint64_t n_pages, const pq2_interval *intervals,
                                           const uint64_t *interval_offsets, uint8_t *d_validity, uint64_t validity_rows,
                                           uint32_t *d_counts, int32_t *status, void *stream);

/* ---- whole V1 data pages, split on the device -------------------------------- */

/* A decompressed DataPage V1 buffer that lies in device memory (for instance
 * behind a GPU decompressor), which the host therefore cannot slice:
 * [u32 length][repetition levels] if max_rep_level > 0, [u32 length][definition
 * levels] if max_def_level > 0, then the values (src/page/mod.rs:350-398,
 * :414-420); for a dictionary-encoded page the values are [bit width][hybrid-RLE
 * indices] (src/deserialize/utils.rs:13-27). */
#define PQ2_RAW_DICT_INDICES 1u
typedef struct pq2_raw_page {
    uint64_t buf_off;       /* the page buffer inside d_buf */
    uint32_t buf_len;
    uint32_t num_values;    /* the page header's num_values */
    uint16_t max_rep_level; /* of the column's descriptor */
    uint16_t max_def_level;
    uint32_t flags;         /* PQ2_RAW_DICT_INDICES: decode the values as dictionary indices */
    uint64_t out_off;       /* the page's first value in each of the three outputs */
} pq2_raw_page;

/* split_buffer_v1 + get_bit_width + dict_indices_decoder on the device, then the
 * decode of up to three streams per page: repetition levels to d_rep, definition
 * levels to d_def (bit width get_bit_width(max level), src/read/levels.rs:3-5),
 * dictionary indices to d_idx, num_values each at out_off.  An output whose
 * stream no page has may be NULL.  `max_index_bits`: an upper bound of the pages'
 * index widths (a dictionary of n entries needs at most get_bit_width(n - 1)
 * bits; 0 = unknown, 32 is assumed, at the price of larger LDS windows).
 * status[i], the first that applies: PQ2_OUT_OF_SPEC for a level length prefix
 * that is missing or exceeds the buffer (page/mod.rs:358-391) or an index width
 * above 32 (utils.rs:19-23); PQ2_REFERENCE_WOULD_PANIC for a dictionary page
 * without values (utils.rs:18); PQ2_INVALID_PARAMETER for an index width above
 * max_index_bits; else the statuses of the repetition, definition and index
 * streams in this order.  The outputs of a page with a non-OK status are
 * unspecified.  `status` is required (n_pages entries) and the call waits for
 * completion: the statuses of the page front end are kept in scratch that the
 * next call on the context reuses, so this entry point has no enqueue-only form. */
int32_t pq2_decode_raw_pages_v1_device(pq2_ctx *ctx, const void *d_buf, uint64_t buf_bytes, const pq2_raw_page *pages,
                                       uint64_t n_pages, uint32_t max_index_bits, uint32_t *d_rep, uint32_t *d_def,
                                       uint32_t *d_idx, uint64_t out_values, int32_t *status, void *stream);

/* ---- plain bit-unpacking -------------------------------------------------- */

/* Replaces bitpacked::Decoder::<u32>::try_new(packed, num_bits, length)?.collect()
 * (src/encoding/bitpacked/decode.rs:28-86).  Errors as the reference:
 * PQ2_OUT_OF_SPEC for num_bits == 0 (:33-35) and for
 * packed_len * 8 < length * num_bits (:37-42); an empty `packed` yields no
 * values (:46-50).  num_bits > 32 -> PQ2_REFERENCE_WOULD_PANIC (unpack.rs:87). */
int32_t pq2_bitunpack_u32_device(pq2_ctx *ctx, const void *d_packed, uint64_t packed_len, uint32_t num_bits,
                                 uint64_t length, uint32_t *d_out, void *stream);
int32_t pq2_bitunpack_u32_host(pq2_ctx *ctx, const void *packed, uint64_t packed_len, uint32_t num_bits,
                               uint64_t length, uint32_t *out);

/* The same for T = u64: bitpacked::Decoder::<u64>::try_new(packed, num_bits, length)?.collect()
 * (decode.rs:28-86 with Unpackable for u64, src/encoding/bitpacked/mod.rs:153-166; unpack64,
 * unpack.rs:21-95), num_bits 1..64 -- what DELTA_BINARY_PACKED mini-blocks are made of
 * (src/encoding/delta_bitpacked/decoder.rs:85).  d_out 8-byte aligned.  num_bits > 64 ->
 * PQ2_REFERENCE_WOULD_PANIC. */
int32_t pq2_bitunpack_u64_device(pq2_ctx *ctx, const void *d_packed, uint64_t packed_len, uint32_t num_bits,
                                 uint64_t length, uint64_t *d_out, void *stream);
int32_t pq2_bitunpack_u64_host(pq2_ctx *ctx, const void *packed, uint64_t packed_len, uint32_t num_bits,
                               uint64_t length, uint64_t *out);

/* ---- DELTA_BINARY_PACKED --------------------------------------------------- */

/* Replaces, for every stream i of the batch,
 *   let mut d = delta_bitpacked::Decoder::try_new(&in[in_off..in_off+in_len])?;
 *   let values = d.by_ref().collect::<Result<Vec<i64>, _>>()?;  let consumed = d.consumed_bytes();
 * (src/encoding/delta_bitpacked/decoder.rs:121-244 over Block, :9-118; zigzag_leb128.rs:5-8;
 * uleb128.rs:3-23; mini-blocks are bitpacked::Decoder<u64>, decoder.rs:85) with the streams and
 * the values resident in HBM.  A descriptor's num_bits is ignored, its num_values is the room
 * (in i64) at d_out[out_off..]; the header's total_count says how many values the stream holds
 * and that many are written.  d_out 8-byte aligned.  The additions wrap as the reference's
 * wrapping_add does (:109, :241).
 *
 * The call waits for completion.  Per stream: `status` (required), and if non-NULL `num_values`
 * (total_count, also when it exceeds the room) and `consumed_bytes` (decoder.rs:182-185 after the
 * last value: the bytes of the mini-blocks that were opened, not those of the unused tail of the
 * last block).  Statuses: PQ2_OK; PQ2_OUT_OF_SPEC (a block without its bit widths, :46-50; a
 * mini-block without its bytes, :73-77; "Missing block", :199-206); PQ2_REFERENCE_WOULD_PANIC
 * (block_size not a multiple of 128, :141; zero mini-blocks, :158; values per mini-block not a
 * multiple of 8, :159; a bit width above 64, :85; a varint that overflows, uleb128.rs:10-12);
 * PQ2_INVALID_PARAMETER (total_count larger than the room).  The values of a stream with a
 * non-OK status are unspecified. */
int32_t pq2_delta_bitpacked_decode_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes,
                                                const pq2_page_desc *descs, uint64_t n_pages, int64_t *d_out,
                                                uint64_t out_values, uint64_t *num_values, uint64_t *consumed_bytes,
                                                int32_t *status, void *stream);
/* One stream in host memory: what a per-page drop-in of delta_bitpacked::Decoder calls.  Returns
 * the stream's status; `out` has room for `cap` values. */
int32_t pq2_delta_bitpacked_decode_host(pq2_ctx *ctx, const void *values, uint64_t len, int64_t *out, uint64_t cap,
                                        uint64_t *num_values, uint64_t *consumed_bytes);

/* Host-side write path of the same encoding (no device work).
 * delta_bitpacked::encode(iterator, &mut buffer) (src/encoding/delta_bitpacked/encoder.rs:11-78): blocks of 128
 * values with one mini-block each; byte for byte the reference's output, including the 128-slot array it reuses
 * from block to block (:27), so the packed slots behind a short last block hold the block before.  The
 * subtractions wrap as in the reference's release build.  `out_cap` must be at least
 * pq2_delta_bitpacked_encode_bound(len). */
uint64_t pq2_delta_bitpacked_encode_bound(uint64_t len);
int32_t pq2_delta_bitpacked_encode(const int64_t *values, uint64_t len, uint8_t *out, uint64_t out_cap,
                                   uint64_t *written);
/* src/encoding/zigzag_leb128.rs:5-8 and :10-15 */
int32_t pq2_zigzag_leb128_decode(const uint8_t *values, uint64_t len, int64_t *value, uint64_t *consumed);
uint64_t pq2_zigzag_leb128_encode(int64_t value, uint8_t container[10]);

/* ---- multi-GPU sharding ---------------------------------------------------- */

/* Splits a batch into `n_parts` contiguous page ranges balanced on what a page
 * costs its GPU: the bytes it moves through HBM, in_len + 4 * num_values.
 * Among contiguous splits the heaviest part (the one the batch waits for) is
 * as light as it can be.  Pages share no state, so there is no exchange step:
 * each rank decodes pages [bounds[r], bounds[r+1]) on its own GPU.  `bounds`
 * receives n_parts + 1 page indices.  The reference delegates this to its
 * callers (README.md:107-128: one thread per column chunk). */
int32_t pq2_partition_pages(const pq2_page_desc *descs, uint64_t n_pages, uint32_t n_parts, uint64_t *bounds);

/* The same split over costs the caller supplies (one per page, any unit), and
 * the costs this library would supply where the streams lie in host memory
 * (`h_in`; NULL: bytes only, what pq2_partition_pages uses): the bytes a page
 * moves plus 256 for every run header it is estimated to hold, from a walk
 * over its first 256 headers (src/encoding/hybrid_rle/decoder.rs:30-64) -- a
 * page of runs of a few values costs a GPU ten times what its bytes say.  No
 * counterpart in the reference, which leaves the split to its callers
 * (README.md:107-128). */
int32_t pq2_partition_by_cost(const uint64_t *costs, uint64_t n_pages, uint32_t n_parts, uint64_t *bounds);
int32_t pq2_page_costs(const void *h_in, uint64_t in_bytes, const pq2_page_desc *descs, uint64_t n_pages,
                       uint64_t *costs);

/* One process, several GPUs: the batch of pq2_decode_batch_host split with
 * pq2_page_costs / pq2_partition_by_cost over `n_ctx` contexts (one per GPU, from
 * pq2_ctx_create(device)), each driven by a host thread of its own; no data
 * moves between the GPUs.  This is the reference's thread-per-column-chunk
 * recipe (README.md:107-128) with GPUs in place of threads.  Page outputs
 * must be laid out in page order (the spans of the ranges must not
 * interleave); the values between two pages of one range are unspecified
 * after the call.  `status` receives one status per page.  Two contexts on
 * the same device are allowed (they share the GPU); the same context twice is
 * not (PQ2_INVALID_PARAMETER: a context is driven by one thread at a time). */
int32_t pq2_multi_decode(pq2_ctx *const *ctxs, uint32_t n_ctx, const void *h_in, uint64_t in_bytes,
                         const pq2_page_desc *descs, uint64_t n_pages, uint32_t *h_out, uint64_t out_values,
                         int32_t *status);

/* ---- host-side page glue (no device work) ---------------------------------- */

/* src/encoding/mod.rs:23-27 */
uint64_t pq2_ceil8(uint64_t value);
/* src/encoding/mod.rs:14-21; PQ2_OUT_OF_SPEC stands for `None` */
int32_t pq2_get_length(const uint8_t *values, uint64_t len, uint64_t *out);
/* src/read/levels.rs:3-5 */
uint32_t pq2_get_bit_width(int16_t max_level);
/* src/encoding/uleb128.rs:3-23 and :30-46 */
int32_t pq2_uleb128_decode(const uint8_t *values, uint64_t len, uint64_t *value, uint64_t *consumed);
uint64_t pq2_uleb128_encode(uint64_t value, uint8_t container[10]);

typedef struct pq2_split {
    uint64_t rep_off, rep_len, def_off, def_len, val_off, val_len; /* relative to the page buffer */
} pq2_split;
/* src/page/mod.rs:350-398 and :400-411 */
int32_t pq2_split_buffer_v1(const uint8_t *buffer, uint64_t len, int32_t has_rep, int32_t has_def, pq2_split *out);
int32_t pq2_split_buffer_v2(uint64_t len, uint64_t rep_len, uint64_t def_len, pq2_split *out);
/* src/deserialize/utils.rs:13-27: bit width byte + offset of the index stream */
int32_t pq2_dict_indices_locate(const uint8_t *values, uint64_t len, uint32_t *bit_width, uint64_t *stream_off);

/* Run iterator: hybrid_rle::Decoder::new(data, num_bits) collected
 * (src/encoding/hybrid_rle/decoder.rs:8-65).  Walks the run headers only (no
 * value is decoded), on the host: consumers such as DefLevelsDecoder::Bitmap
 * (src/deserialize/utils.rs:45-48) use the runs as zero-copy slices. */
#define PQ2_RUN_BITPACKED 0
#define PQ2_RUN_RLE 1
typedef struct pq2_run {
    int32_t kind;        /* PQ2_RUN_BITPACKED | PQ2_RUN_RLE */
    uint32_t reserved;
    uint64_t offset;     /* byte offset of the run's slice inside `data` */
    uint64_t len;        /* slice length in bytes */
    uint64_t run_length; /* RLE: number of repetitions (HybridEncoded::Rle(_, n)) */
} pq2_run;
/* Stores at most `cap` runs, always reports the full count in *n_runs. */
int32_t pq2_hybrid_rle_runs(const uint8_t *data, uint64_t len, uint32_t num_bits, pq2_run *runs, uint64_t cap,
                            uint64_t *n_runs);

/* ---- write side (host): byte-exact with the reference encoders ---------------- */

/* bitpacked::encode::<u32> (src/encoding/bitpacked/encode.rs:11-36, pack.rs:2-75):
 * `packed` must be zeroed and hold ceil(len / 32) * 4 * num_bits bytes. */
int32_t pq2_bitpacked_encode_u32(const uint32_t *unpacked, uint64_t len, uint32_t num_bits, uint8_t *packed,
                                 uint64_t packed_len);
/* hybrid_rle::encode_u32 (src/encoding/hybrid_rle/encoder.rs:9-72): one
 * bit-packed run for the whole input.  Needs out_cap >= 10 + 4*num_bits*ceil(len/32). */
int32_t pq2_hybrid_rle_encode_u32(const uint32_t *values, uint64_t len, uint32_t num_bits, uint8_t *out,
                                  uint64_t out_cap, uint64_t *written);
/* hybrid_rle::encode_bool (encoder.rs:75-93) and bitmap::encode_bool
 * (src/encoding/hybrid_rle/bitmap.rs:71-102); `bools` holds one byte per value. */
int32_t pq2_hybrid_rle_encode_bool(const uint8_t *bools, uint64_t len, uint8_t *out, uint64_t out_cap,
                                   uint64_t *written);
int32_t pq2_bitmap_encode_bool(const uint8_t *bools, uint64_t len, uint8_t *out, uint64_t out_cap,
                               uint64_t *written);

#ifdef __cplusplus
}
#endif
#endif /* PQ2HIP_H */
