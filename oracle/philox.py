"""Philox4x32-10 counter-based RNG, NumPy restatement (TEST INFRASTRUCTURE).

The reference draws its randomness (negative samples, dropout masks) from
Theano's MRG31k3p streams (gru4rec.py:25,298,559).  Bit-parity with MRG is not
required by the reference itself (README.md:359 accepts run-to-run variation),
so the HIP path uses Philox4x32-10 (Salmon et al., SC'11) and this file is the
CPU twin of `gru4rec_amd/csrc/philox.cuh` so that the oracle and the device
draw *identical* uniforms and parity can be checked step by step.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import anything under `oracle/`.
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)

# stream ids (counter word 3) -- must match csrc/philox.cuh
STREAM_SAMPLE = 0x53414D50   # 'SAMP'
STREAM_DROP_EMBED = 0x44454D42  # 'DEMB'
STREAM_DROP_HIDDEN = 0x44484944  # 'DHID' (+ layer index)
STREAM_TIEBREAK = 0x54494542  # 'TIEB': evaluation mode 'tiebreaking' (evaluation.py:55)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32 with 10 rounds.  All inputs broadcastable uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32).copy()
    c1 = np.asarray(c1, dtype=np.uint32).copy()
    c2 = np.asarray(c2, dtype=np.uint32).copy()
    c3 = np.asarray(c3, dtype=np.uint32).copy()
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _M0
            p1 = c2.astype(np.uint64) * _M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK).astype(np.uint32)
            n0 = hi1 ^ c1 ^ k0
            n1 = lo1
            n2 = hi0 ^ c3 ^ k1
            n3 = lo0
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def u32_to_unit_float(x):
    """uint32 -> float32 in [0,1): top 24 bits * 2^-24 (exactly representable)."""
    return ((np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float32)
            * np.float32(1.0 / 16777216.0))


def uniform_block(n, seed, c1, c2, stream):
    """n uniforms; element e comes from Philox call (c0=e>>2, c1, c2, c3=stream), lane e&3."""
    ncall = (n + 3) // 4
    c0 = np.arange(ncall, dtype=np.uint32)
    r = philox4x32_10(c0, np.uint32(c1 & 0xFFFFFFFF), np.uint32(c2 & 0xFFFFFFFF),
                      np.uint32(stream & 0xFFFFFFFF),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    out = np.stack(r, axis=1).reshape(-1)[:n]
    return u32_to_unit_float(out)


def uniform_rows(rows, cols, seed, step, stream):
    """float32 uniforms of shape (rows, cols): element (row, col) comes from Philox call
    (c0 = col>>2, c1 = row, c2 = step, c3 = stream), lane col&3."""
    ncall = (cols + 3) // 4
    c0 = np.arange(ncall, dtype=np.uint32)[None, :]
    c1 = np.arange(rows, dtype=np.uint32)[:, None]
    r = philox4x32_10(c0, c1, np.uint32(step & 0xFFFFFFFF), np.uint32(stream & 0xFFFFFFFF),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return u32_to_unit_float(np.stack(r, axis=2).reshape(rows, ncall * 4)[:, :cols])


def dropout_mask(rows, cols, retain, seed, step, stream):
    """Bernoulli(retain)/retain mask of shape (rows, cols) as float32.

    Element (row, col): Philox call (c0 = col>>2, c1 = row, c2 = step, c3 = stream), lane col&3;
    keep iff u < retain (Theano's binomial is `uniform < p`, gru4rec.py:298).
    """
    ncall = (cols + 3) // 4
    c0 = np.arange(ncall, dtype=np.uint32)[None, :]
    c1 = np.arange(rows, dtype=np.uint32)[:, None]
    r = philox4x32_10(c0, c1, np.uint32(step & 0xFFFFFFFF), np.uint32(stream & 0xFFFFFFFF),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = u32_to_unit_float(np.stack(r, axis=2).reshape(rows, ncall * 4)[:, :cols])
    keep = (u < np.float32(retain)).astype(np.float32)
    return keep / np.float32(retain)
