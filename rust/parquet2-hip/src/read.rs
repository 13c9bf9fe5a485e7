//! `parquet2::read::levels` (src/read/levels.rs:3-5)
pub mod levels {
    pub fn get_bit_width(max_level: i16) -> u32 {
        unsafe { pq2hip_sys::pq2_get_bit_width(max_level) }
    }
}
