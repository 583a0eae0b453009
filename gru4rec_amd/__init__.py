"""gru4rec_amd: GRU4Rec's session-parallel training / prediction hot path on MI355X (gfx950).

Python host (this package) -> ctypes -> libgru4rec_hip.so (hand-written HIP kernels).  No PyTorch, no Triton.
"""
__version__ = '0.1.0'
