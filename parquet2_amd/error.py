"""Errors of the decoder, mirroring `parquet2::error::Error` (src/error.rs:23-34).

The reference panics on a few malformed inputs; this package answers those with
`ReferencePanic` instead of crashing.
"""


class Error(Exception):
    """Base class (parquet2::error::Error)."""


class OutOfSpec(Error):
    """Error::OutOfSpec: the data does not follow the Parquet specification."""


class FeatureNotActive(Error):
    """Error::FeatureNotActive"""


class FeatureNotSupported(Error):
    """Error::FeatureNotSupported"""


class InvalidParameter(Error):
    """Error::InvalidParameter"""


class WouldOverAllocate(Error):
    """Error::WouldOverAllocate"""


class ReferencePanic(Error):
    """The reference decoder panics on this input (status PQ2_REFERENCE_WOULD_PANIC)."""


class DeviceError(Error):
    """A HIP runtime call failed, or no GPU is usable (status PQ2_DEVICE_ERROR)."""


_BY_STATUS = {1: OutOfSpec, 2: InvalidParameter, 3: FeatureNotSupported, 4: WouldOverAllocate, 5: ReferencePanic,
              6: DeviceError}


def error_for_status(status, message=""):
    return _BY_STATUS.get(status, Error)(message or "status %d" % status)


def raise_for_status(status, message=""):
    if status != 0:
        raise error_for_status(status, message)
