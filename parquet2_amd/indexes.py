"""`parquet2::indexes::Interval` (src/indexes/intervals.rs:11-16)."""
from dataclasses import dataclass


@dataclass
class Interval:
    start: int
    length: int

    @classmethod
    def new(cls, start, length):
        return cls(start, length)
