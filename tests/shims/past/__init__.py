"""Test shim: the reference imports `past.utils.old_div` (python-future is not in this image)."""
