//! `parquet2::deserialize::utils` (src/deserialize/utils.rs:13-58): which decoder a page's streams get.
use crate::encoding::hybrid_rle::{self, HybridRleDecoder};
use crate::error::Error;
use crate::page::{split_buffer, PageParts};
use crate::read::levels::get_bit_width;
use pq2hip_sys as sys;

/// src/deserialize/utils.rs:13-27: the values buffer of a dictionary-encoded page is `[bit width][hybrid-RLE stream]`.
pub fn dict_indices_decoder<'a>(page: &PageParts<'a>, num_values: usize) -> Result<HybridRleDecoder<'a>, Error> {
    let (_, _, indices_buffer) = split_buffer(page)?;
    let (mut bit_width, mut stream_off) = (0u32, 0u64);
    let rc = unsafe { sys::pq2_dict_indices_locate(indices_buffer.as_ptr(), indices_buffer.len() as u64, &mut bit_width, &mut stream_off) };
    if rc != sys::PQ2_OK {
        return Err(crate::error::from_status(rc, std::ptr::null()));
    }
    HybridRleDecoder::try_new(&indices_buffer[stream_off as usize..], bit_width, num_values)
}

/// src/deserialize/utils.rs:30-58
pub enum DefLevelsDecoder<'a> {
    /// max_def_level == 1: the runs of the one-bit stream (HybridRleIter over them is host-side iterator glue;
    /// whole batches of such pages become validity bitmaps in `pq2_decode_validity_bitmap_batch_device`).
    Bitmap(hybrid_rle::Decoder<'a>, usize),
    /// nested columns: the levels themselves, and the maximum
    Levels(HybridRleDecoder<'a>, u32),
}

impl<'a> DefLevelsDecoder<'a> {
    pub fn try_new(page: &PageParts<'a>, num_values: usize) -> Result<Self, Error> {
        let (_, def_levels, _) = split_buffer(page)?;
        let max_def_level = page.max_def_level;
        Ok(if max_def_level == 1 {
            Self::Bitmap(hybrid_rle::Decoder::new(def_levels, 1), num_values)
        } else {
            Self::Levels(HybridRleDecoder::try_new(def_levels, get_bit_width(max_def_level), num_values)?, max_def_level as u32)
        })
    }
}
