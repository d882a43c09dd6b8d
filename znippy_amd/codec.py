"""Codec layer — host-side mirror of znippy-common/src/codec.rs over the C ABI.

Same names, argument meaning and error behaviour as the reference:
  CompressCtx(level)            codec.rs:L16-28   (one context per worker; here: one HIP context/stream)
  .compress(input) -> bytes     codec.rs:L30-38
  .compress_into(input, out)    codec.rs:L43-55   out: bytearray, resized to the bytes written
  decompress_frame(frame)       codec.rs:L58-62
  decompress_into(frame, out)   codec.rs:L67-78
Errors raise ZnippyError (the reference returns anyhow::Error).  The compression level selects the
encoder's effort tier (znippy_ctx_set_level: 1-3 fast, 4-22 higher effort).
"""
from . import hip
from ._lib import ZnippyError  # noqa: F401

_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        import torch
        _default_ctx = hip.Context(torch.cuda.current_device())
    return _default_ctx


class CompressCtx:
    def __init__(self, compression_level: int = 19, ctx=None):
        self.level = compression_level
        self.ctx = ctx or default_context()

    def _apply_level(self):
        if not 1 <= int(self.level) <= 22:
            raise ZnippyError(-1, f"compression level {self.level} outside 1..22")
        if self.ctx.level != self.level:  # the HIP context may be shared between CompressCtx objects
            self.ctx.set_level(self.level)

    def compress(self, data) -> bytes:
        self._apply_level()
        return self.ctx.compress(bytes(data))

    def compress_into(self, data, out: bytearray) -> int:
        self._apply_level()
        b = self.ctx.compress(bytes(data))
        out[:] = b
        return len(b)


def decompress_frame(compressed, ctx=None) -> bytes:
    return (ctx or default_context()).decompress(bytes(compressed))


def decompress_into(compressed, out: bytearray, ctx=None) -> int:
    b = decompress_frame(compressed, ctx)
    out[:] = b
    return len(b)


def blake3_hash(data, ctx=None) -> bytes:
    """blake3::hash(&[u8]) (stream_packer.rs:L219, slot_packer.rs:L553, decompress.rs:L172)."""
    return (ctx or default_context()).blake3(bytes(data))
