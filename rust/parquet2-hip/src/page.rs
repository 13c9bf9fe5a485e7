//! `parquet2::page::split_buffer` (src/page/mod.rs:350-431) on the parts of a `DataPage` that it reads.
use crate::error::Error;
use pq2hip_sys as sys;

/// What `split_buffer` needs of a data page: its (decompressed) buffer, whether the column has repetition and
/// definition levels, and for a V2 page the byte lengths of the two level streams from its header.
pub struct PageParts<'a> {
    pub buffer: &'a [u8],
    pub max_rep_level: i16,
    pub max_def_level: i16,
    pub v2_level_lengths: Option<(usize, usize)>,
}

/// Returns (repetition levels, definition levels, values).
pub fn split_buffer<'a>(page: &PageParts<'a>) -> Result<(&'a [u8], &'a [u8], &'a [u8]), Error> {
    let mut s = sys::pq2_split::default();
    let rc = match page.v2_level_lengths {
        None => unsafe {
            sys::pq2_split_buffer_v1(page.buffer.as_ptr(), page.buffer.len() as u64, (page.max_rep_level > 0) as i32,
                                     (page.max_def_level > 0) as i32, &mut s)
        },
        Some((rep, def)) => unsafe { sys::pq2_split_buffer_v2(page.buffer.len() as u64, rep as u64, def as u64, &mut s) },
    };
    if rc != sys::PQ2_OK {
        return Err(crate::error::from_status(rc, std::ptr::null()));
    }
    let part = |off: u64, len: u64| &page.buffer[off as usize..(off + len) as usize];
    Ok((part(s.rep_off, s.rep_len), part(s.def_off, s.def_len), part(s.val_off, s.val_len)))
}
