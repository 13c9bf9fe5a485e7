//! `parquet2::encoding::hybrid_rle` (src/encoding/hybrid_rle/{mod,decoder,bitmap,encoder}.rs).
use crate::error::Error;
use pq2hip_sys as sys;

/// src/encoding/hybrid_rle/mod.rs:15-21
#[derive(Debug, PartialEq, Eq, Clone, Copy)]
pub enum HybridEncoded<'a> {
    /// A bit-packed slice. The consumer must know its bit-width to unpack it.
    Bitpacked(&'a [u8]),
    /// A RLE-encoded slice: the value bytes and the run length.
    Rle(&'a [u8], usize),
}

/// The run iterator (src/encoding/hybrid_rle/decoder.rs:8-65) over `pq2_hybrid_rle_runs` (a host-side walk: the
/// runs of a page are what `DefLevelsDecoder::Bitmap` consumers iterate; batches of pages get their validity
/// bitmaps whole from `pq2_decode_validity_bitmap_batch_device`).
#[derive(Debug, Clone)]
pub struct Decoder<'a> {
    values: &'a [u8],
    num_bits: usize,
    runs: std::vec::IntoIter<sys::pq2_run>,
    failed: Option<i32>,
}

impl<'a> Decoder<'a> {
    pub fn new(values: &'a [u8], num_bits: usize) -> Self {
        // a run takes at least two bytes of the stream
        let mut runs = vec![sys::pq2_run::default(); values.len() / 2 + 2];
        let mut n = 0u64;
        let rc = unsafe {
            sys::pq2_hybrid_rle_runs(values.as_ptr(), values.len() as u64, num_bits as u32, runs.as_mut_ptr(), runs.len() as u64, &mut n)
        };
        runs.truncate(n as usize);
        Self { values, num_bits, runs: runs.into_iter(), failed: (rc != sys::PQ2_OK).then_some(rc) }
    }
    /// src/encoding/hybrid_rle/decoder.rs:23-27
    pub fn num_bits(&self) -> usize {
        self.num_bits
    }
}
impl<'a> Iterator for Decoder<'a> {
    type Item = Result<HybridEncoded<'a>, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        match self.runs.next() {
            Some(run) => {
                let slice = &self.values[run.offset as usize..(run.offset + run.len) as usize];
                Some(Ok(if run.kind == sys::PQ2_RUN_BITPACKED {
                    HybridEncoded::Bitpacked(slice)
                } else {
                    HybridEncoded::Rle(slice, run.run_length as usize)
                }))
            }
            // where the reference panics in the middle of a stream, the runs before come out and then an error
            None => self.failed.take().map(|rc| Err(crate::error::from_status(rc, std::ptr::null()))),
        }
    }
}

/// The value iterator (src/encoding/hybrid_rle/mod.rs:35-116) over `pq2_decode_batch_host`: one page, one call.
#[derive(Debug, Clone)]
pub struct HybridRleDecoder<'a> {
    values: std::vec::IntoIter<u32>,
    _borrow: std::marker::PhantomData<&'a [u8]>,
}

impl<'a> HybridRleDecoder<'a> {
    /// Decodes eagerly on the GPU (consumers drain the iterator with `collect`); exactly `num_values` items.
    pub fn try_new(data: &'a [u8], num_bits: u32, num_values: usize) -> Result<Self, Error> {
        let desc = sys::pq2_page_desc { in_off: 0, in_len: data.len() as u32, num_bits, num_values: num_values as u64, out_off: 0 };
        let mut out = vec![0u32; num_values];
        crate::with_ctx(|ctx| {
            let status = ctx.decode_batch_host(data, std::slice::from_ref(&desc), &mut out)?;
            if status[0] != sys::PQ2_OK {
                return Err(crate::error::from_status(status[0], ctx.as_ptr()));
            }
            Ok(())
        })?;
        Ok(Self { values: out.into_iter(), _borrow: Default::default() })
    }
}
impl<'a> Iterator for HybridRleDecoder<'a> {
    type Item = Result<u32, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        self.values.next().map(Ok)
    }
    fn size_hint(&self) -> (usize, Option<usize>) {
        self.values.size_hint()
    }
}
impl<'a> ExactSizeIterator for HybridRleDecoder<'a> {}

/// src/encoding/hybrid_rle/bitmap.rs:14-68 (host-side: a bitmap is read where it lies)
#[derive(Debug, Clone)]
pub struct BitmapIter<'a> {
    slice: &'a [u8],
    index: usize,
    end: usize,
}
impl<'a> BitmapIter<'a> {
    pub fn new(slice: &'a [u8], offset: usize, len: usize) -> Self {
        assert!(offset / 8 <= slice.len());
        Self { slice, index: offset, end: offset + len }
    }
}
impl<'a> Iterator for BitmapIter<'a> {
    type Item = bool;
    fn next(&mut self) -> Option<bool> {
        if self.index == self.end {
            return None;
        }
        // past the slice the last byte repeats (bitmap.rs:55-60); an empty slice reads as 0
        let byte = self.slice.get(self.index / 8).or(self.slice.last()).copied().unwrap_or(0);
        let bit = (byte >> (self.index % 8)) & 1 != 0;
        self.index += 1;
        Some(bit)
    }
    fn size_hint(&self) -> (usize, Option<usize>) {
        (self.end - self.index, Some(self.end - self.index))
    }
}

fn encode_with(bound: usize, call: impl FnOnce(*mut u8, u64, *mut u64) -> i32) -> Vec<u8> {
    let mut out = vec![0u8; bound];
    let mut written = 0u64;
    let rc = call(out.as_mut_ptr(), out.len() as u64, &mut written);
    assert_eq!(rc, sys::PQ2_OK);
    out.truncate(written as usize);
    out
}

/// src/encoding/hybrid_rle/encoder.rs:9-20: one bit-packed run, byte for byte as the reference writes it.
pub fn encode_u32<W: std::io::Write>(writer: &mut W, values: &[u32], num_bits: u32) -> std::io::Result<()> {
    let bound = 10 + (values.len() + 7) / 8 * num_bits as usize + 8;
    let bytes = encode_with(bound, |out, cap, written| unsafe {
        sys::pq2_hybrid_rle_encode_u32(values.as_ptr(), values.len() as u64, num_bits, out, cap, written)
    });
    writer.write_all(&bytes)
}

/// src/encoding/hybrid_rle/encoder.rs:72-93
pub fn encode_bool<W: std::io::Write>(writer: &mut W, values: &[bool]) -> std::io::Result<()> {
    let bools: Vec<u8> = values.iter().map(|&b| b as u8).collect();
    let bytes = encode_with(10 + bools.len() / 8 + 8, |out, cap, written| unsafe {
        sys::pq2_hybrid_rle_encode_bool(bools.as_ptr(), bools.len() as u64, out, cap, written)
    });
    writer.write_all(&bytes)
}

/// src/encoding/hybrid_rle/bitmap.rs:71-102
pub fn bitpacked_encode<W: std::io::Write>(writer: &mut W, values: &[bool]) -> std::io::Result<()> {
    let bools: Vec<u8> = values.iter().map(|&b| b as u8).collect();
    let bytes = encode_with(bools.len() / 8 + 8, |out, cap, written| unsafe {
        sys::pq2_bitmap_encode_bool(bools.as_ptr(), bools.len() as u64, out, cap, written)
    });
    writer.write_all(&bytes)
}
