"""MI355X-native back end of the CompaCT codec: ctypes binding + batch API (no PyTorch)."""
from . import _ffi  # noqa: F401
from .batch import (DeviceBuffer, Event, PinnedArray, codec_params, decode_batch, decode_payload_dev,  # noqa: F401
                    zlib_compress_batch, zlib_decompress_batch,
                    default_config, device_info, encode_batch, encode_payload_dev)
