"""Test shim: the reference's wrapper modules import `future.utils.with_metaclass` (python-future is not in this image)."""
