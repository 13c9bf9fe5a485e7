from . import levels  # noqa: F401
