//! Raw declarations of `libpq2hip.so`, one per export of `include/pq2hip.h` (which documents every entry point and
//! names the item of parquet2 it replaces).  NOT COMPILED in the build image (no rustc): kept in step with the header
//! by `tests/test_abi_symbols.py`, which compares names and parameter counts.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

#[repr(C)]
pub struct pq2_ctx {
    _private: [u8; 0],
}

/// The arguments of `HybridRleDecoder::try_new(data, num_bits, num_values)` plus where the values go.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pq2_page_desc {
    pub in_off: u64,
    pub in_len: u32,
    pub num_bits: u32,
    pub num_values: u64,
    pub out_off: u64,
}
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pq2_dict_ref {
    pub dict_off: u64,
    pub dict_len: u32,
    pub reserved: u32,
}
/// `indexes::Interval` (src/indexes/intervals.rs:11-16)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pq2_interval {
    pub start: u64,
    pub length: u64,
}
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pq2_raw_page {
    pub buf_off: u64,
    pub buf_len: u32,
    pub num_values: u32,
    pub max_rep_level: u16,
    pub max_def_level: u16,
    pub flags: u32,
    pub out_off: u64,
}
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pq2_split {
    pub rep_off: u64,
    pub rep_len: u64,
    pub def_off: u64,
    pub def_len: u64,
    pub val_off: u64,
    pub val_len: u64,
}
/// `hybrid_rle::HybridEncoded` as offsets into the stream (src/encoding/hybrid_rle/mod.rs:15-21)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pq2_run {
    pub kind: i32,
    pub reserved: u32,
    pub offset: u64,
    pub len: u64,
    pub run_length: u64,
}

pub const PQ2_OK: i32 = 0;
pub const PQ2_OUT_OF_SPEC: i32 = 1;
pub const PQ2_INVALID_PARAMETER: i32 = 2;
pub const PQ2_FEATURE_NOT_SUPPORTED: i32 = 3;
pub const PQ2_WOULD_OVER_ALLOCATE: i32 = 4;
pub const PQ2_REFERENCE_WOULD_PANIC: i32 = 5;
pub const PQ2_DEVICE_ERROR: i32 = 6;
pub const PQ2_BLOCK_VALUES: u32 = 1024;
pub const PQ2_MAX_PAGE_VALUES: u32 = 0x7fff_ffff;
pub const PQ2_MAX_PAGE_BYTES: u32 = 0xfff0_0000;
pub const PQ2_NOALLOC_MAX_PAGES: u32 = 4096;
pub const PQ2_RAW_DICT_INDICES: u32 = 1;
pub const PQ2_RUN_BITPACKED: i32 = 0;
pub const PQ2_RUN_RLE: i32 = 1;

extern "C" {
    // ---- context ----
    pub fn pq2_ctx_create(device_id: i32, ctx: *mut *mut pq2_ctx) -> i32;
    pub fn pq2_ctx_destroy(ctx: *mut pq2_ctx) -> i32;
    pub fn pq2_strerror(status: i32) -> *const c_char;
    pub fn pq2_last_error(ctx: *const pq2_ctx) -> *const c_char;
    pub fn pq2_ctx_enable_profiling(ctx: *mut pq2_ctx, on: i32) -> i32;
    pub fn pq2_ctx_last_schedule(ctx: *const pq2_ctx) -> *const c_char;
    pub fn pq2_ctx_pages_total(ctx: *const pq2_ctx) -> u64;
    pub fn pq2_ctx_last_density(ctx: *const pq2_ctx) -> u32;
    pub fn pq2_ctx_set_schedule(ctx: *mut pq2_ctx, name: *const c_char) -> i32;
    pub fn pq2_ctx_scout_rounds(ctx: *mut pq2_ctx, rounds: *mut u32, n_pages: u64) -> i32;
    pub fn pq2_ctx_kernel_times(ctx: *mut pq2_ctx, parse_ms: *mut f64, expand_ms: *mut f64, batches: *mut u64,
                                expand_launches: *mut u64) -> i32;
    // ---- hybrid RLE / bit-packed decode ----
    pub fn pq2_decode_batch_device(ctx: *mut pq2_ctx, d_in: *const c_void, in_bytes: u64, descs: *const pq2_page_desc,
                                   n_pages: u64, d_out: *mut u32, out_values: u64, status: *mut i32,
                                   stream: *mut c_void) -> i32;
    pub fn pq2_fetch_status(ctx: *mut pq2_ctx, status: *mut i32, n_pages: u64) -> i32;
    pub fn pq2_decode_batch_host(ctx: *mut pq2_ctx, h_in: *const c_void, in_bytes: u64, descs: *const pq2_page_desc,
                                 n_pages: u64, h_out: *mut u32, out_values: u64, status: *mut i32) -> i32;
    pub fn pq2_decode_batch_device_noalloc(device_id: i32, d_in: *const c_void, in_bytes: u64,
                                           d_descs: *const pq2_page_desc, n_pages: u64, max_bits: u32, d_out: *mut u32,
                                           out_values: u64, d_status: *mut i32, stream: *mut c_void) -> i32;
    pub fn pq2_decode_validity_bitmap_batch_device(ctx: *mut pq2_ctx, d_in: *const c_void, in_bytes: u64,
                                                   descs: *const pq2_page_desc, n_pages: u64, d_bitmap: *mut c_void,
                                                   bitmap_bytes: u64, rows: *mut u32, set_counts: *mut u32,
                                                   status: *mut i32, stream: *mut c_void) -> i32;
    pub fn pq2_decode_dict_batch_device(ctx: *mut pq2_ctx, d_in: *const c_void, in_bytes: u64,
                                        descs: *const pq2_page_desc, n_pages: u64, d_dict: *const c_void,
                                        dict_bytes: u64, dicts: *const pq2_dict_ref, value_size: u32, d_out: *mut c_void,
                                        out_values: u64, status: *mut i32, stream: *mut c_void) -> i32;
    pub fn pq2_decode_optional_dict_batch_device(ctx: *mut pq2_ctx, d_in: *const c_void, in_bytes: u64,
                                                 def_descs: *const pq2_page_desc, idx_descs: *const pq2_page_desc,
                                                 max_def_levels: *const u32, n_pages: u64, d_dict: *const c_void,
                                                 dict_bytes: u64, dicts: *const pq2_dict_ref, value_size: u32,
                                                 d_out: *mut c_void, d_validity: *mut u8, out_rows: u64,
                                                 status: *mut i32, stream: *mut c_void) -> i32;
    pub fn pq2_decode_selected_batch_device(ctx: *mut pq2_ctx, d_in: *const c_void, in_bytes: u64,
                                            descs: *const pq2_page_desc, n_pages: u64, intervals: *const pq2_interval,
                                            interval_offsets: *const u64, d_out: *mut u32, out_values: u64,
                                            status: *mut i32, stream: *mut c_void) -> i32;
    pub fn pq2_filtered_validity_batch_device(ctx: *mut pq2_ctx, d_in: *const c_void, in_bytes: u64,
                                              descs: *const pq2_page_desc, n_pages: u64, intervals: *const pq2_interval,
                                              interval_offsets: *const u64, d_validity: *mut u8, validity_rows: u64,
                                              d_counts: *mut u32, status: *mut i32, stream: *mut c_void) -> i32;
    pub fn pq2_decode_raw_pages_v1_device(ctx: *mut pq2_ctx, d_buf: *const c_void, buf_bytes: u64,
                                          pages: *const pq2_raw_page, n_pages: u64, max_index_bits: u32, d_rep: *mut u32,
                                          d_def: *mut u32, d_idx: *mut u32, out_values: u64, status: *mut i32,
                                          stream: *mut c_void) -> i32;
    // ---- bit-unpack ----
    pub fn pq2_bitunpack_u32_device(ctx: *mut pq2_ctx, d_packed: *const c_void, packed_len: u64, num_bits: u32,
                                    length: u64, d_out: *mut u32, stream: *mut c_void) -> i32;
    pub fn pq2_bitunpack_u32_host(ctx: *mut pq2_ctx, packed: *const c_void, packed_len: u64, num_bits: u32, length: u64,
                                  out: *mut u32) -> i32;
    pub fn pq2_bitunpack_u64_device(ctx: *mut pq2_ctx, d_packed: *const c_void, packed_len: u64, num_bits: u32,
                                    length: u64, d_out: *mut u64, stream: *mut c_void) -> i32;
    pub fn pq2_bitunpack_u64_host(ctx: *mut pq2_ctx, packed: *const c_void, packed_len: u64, num_bits: u32, length: u64,
                                  out: *mut u64) -> i32;
    // ---- DELTA_BINARY_PACKED ----
    pub fn pq2_delta_bitpacked_decode_batch_device(ctx: *mut pq2_ctx, d_in: *const c_void, in_bytes: u64,
                                                   descs: *const pq2_page_desc, n_pages: u64, d_out: *mut i64,
                                                   out_values: u64, num_values: *mut u64, consumed_bytes: *mut u64,
                                                   status: *mut i32, stream: *mut c_void) -> i32;
    pub fn pq2_delta_bitpacked_decode_host(ctx: *mut pq2_ctx, values: *const c_void, len: u64, out: *mut i64, cap: u64,
                                           num_values: *mut u64, consumed_bytes: *mut u64) -> i32;
    pub fn pq2_delta_bitpacked_encode_bound(len: u64) -> u64;
    pub fn pq2_delta_bitpacked_encode(values: *const i64, len: u64, out: *mut u8, out_cap: u64, written: *mut u64) -> i32;
    pub fn pq2_zigzag_leb128_decode(values: *const u8, len: u64, value: *mut i64, consumed: *mut u64) -> i32;
    pub fn pq2_zigzag_leb128_encode(value: i64, container: *mut u8) -> u64;
    // ---- several GPUs ----
    pub fn pq2_partition_pages(descs: *const pq2_page_desc, n_pages: u64, n_parts: u32, bounds: *mut u64) -> i32;
    pub fn pq2_partition_by_cost(costs: *const u64, n_pages: u64, n_parts: u32, bounds: *mut u64) -> i32;
    pub fn pq2_page_costs(h_in: *const c_void, in_bytes: u64, descs: *const pq2_page_desc, n_pages: u64,
                          costs: *mut u64) -> i32;
    pub fn pq2_multi_decode(ctxs: *const *mut pq2_ctx, n_ctx: u32, h_in: *const c_void, in_bytes: u64,
                            descs: *const pq2_page_desc, n_pages: u64, h_out: *mut u32, out_values: u64,
                            status: *mut i32) -> i32;
    // ---- host-side page glue (no device work) ----
    pub fn pq2_ceil8(value: u64) -> u64;
    pub fn pq2_get_length(values: *const u8, len: u64, out: *mut u64) -> i32;
    pub fn pq2_get_bit_width(max_level: i16) -> u32;
    pub fn pq2_uleb128_decode(values: *const u8, len: u64, value: *mut u64, consumed: *mut u64) -> i32;
    pub fn pq2_uleb128_encode(value: u64, container: *mut u8) -> u64;
    pub fn pq2_split_buffer_v1(buffer: *const u8, len: u64, has_rep: i32, has_def: i32, out: *mut pq2_split) -> i32;
    pub fn pq2_split_buffer_v2(len: u64, rep_len: u64, def_len: u64, out: *mut pq2_split) -> i32;
    pub fn pq2_dict_indices_locate(values: *const u8, len: u64, bit_width: *mut u32, stream_off: *mut u64) -> i32;
    pub fn pq2_hybrid_rle_runs(data: *const u8, len: u64, num_bits: u32, runs: *mut pq2_run, cap: u64,
                               n_runs: *mut u64) -> i32;
    // ---- encoders (the write side of the path, for test input) ----
    pub fn pq2_bitpacked_encode_u32(unpacked: *const u32, len: u64, num_bits: u32, packed: *mut u8, packed_len: u64) -> i32;
    pub fn pq2_hybrid_rle_encode_u32(values: *const u32, len: u64, num_bits: u32, out: *mut u8, out_cap: u64,
                                     written: *mut u64) -> i32;
    pub fn pq2_hybrid_rle_encode_bool(bools: *const u8, len: u64, out: *mut u8, out_cap: u64, written: *mut u64) -> i32;
    pub fn pq2_bitmap_encode_bool(bools: *const u8, len: u64, out: *mut u8, out_cap: u64, written: *mut u64) -> i32;
}
