//! `parquet2::encoding::delta_bitpacked` (src/encoding/delta_bitpacked/{decoder,encoder}.rs).
use crate::encoding::uleb128;
use crate::error::Error;
use pq2hip_sys as sys;

/// src/encoding/delta_bitpacked/decoder.rs:121-244
#[derive(Debug, Clone)]
pub struct Decoder<'a> {
    values: std::vec::IntoIter<i64>,
    consumed: usize,
    _borrow: std::marker::PhantomData<&'a [u8]>,
}
impl<'a> Decoder<'a> {
    pub fn try_new(values: &'a [u8]) -> Result<Self, Error> {
        // the header's third varint is the number of values: the room the stream needs
        let (_, a) = uleb128::decode(values)?;
        let (_, b) = uleb128::decode(&values[a.min(values.len())..])?;
        let (total_count, _) = uleb128::decode(&values[(a + b).min(values.len())..])?;
        let mut out = vec![0i64; total_count as usize];
        let (mut n, mut consumed) = (0u64, 0u64);
        crate::with_ctx(|ctx| {
            let rc = unsafe {
                sys::pq2_delta_bitpacked_decode_host(ctx.as_ptr(), values.as_ptr() as _, values.len() as u64, out.as_mut_ptr(),
                                                     total_count, &mut n, &mut consumed)
            };
            if rc != sys::PQ2_OK {
                return Err(crate::error::from_status(rc, ctx.as_ptr()));
            }
            Ok(())
        })?;
        out.truncate(n as usize);
        Ok(Self { values: out.into_iter(), consumed: consumed as usize, _borrow: Default::default() })
    }
    /// As the reference's once the iterator is drained (decoder.rs:206-218).
    pub fn consumed_bytes(&self) -> usize {
        self.consumed
    }
}
impl<'a> Iterator for Decoder<'a> {
    type Item = Result<i64, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        self.values.next().map(Ok)
    }
    fn size_hint(&self) -> (usize, Option<usize>) {
        self.values.size_hint()
    }
}

/// src/encoding/delta_bitpacked/encoder.rs:11-78 (blocks of 128 values, one mini-block)
pub fn encode(values: &[i64], buffer: &mut Vec<u8>) {
    let bound = unsafe { sys::pq2_delta_bitpacked_encode_bound(values.len() as u64) } as usize;
    let at = buffer.len();
    buffer.resize(at + bound, 0);
    let mut written = 0u64;
    let rc = unsafe { sys::pq2_delta_bitpacked_encode(values.as_ptr(), values.len() as u64, buffer[at..].as_mut_ptr(), bound as u64, &mut written) };
    assert_eq!(rc, sys::PQ2_OK);
    buffer.truncate(at + written as usize);
}
