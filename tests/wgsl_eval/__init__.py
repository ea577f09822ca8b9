"""TEST INFRASTRUCTURE: an independent executor of the WGSL the reference's effect compiler emits (see interp.py)."""
