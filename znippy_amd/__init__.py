"""znippy_amd — MI355X-native per-chunk codec + hash path for the Znippy archive format.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI), the ctypes binding,
and the host-side mirror of the reference's codec / worker-loop interface.
"""
__version__ = "0.1.0"
