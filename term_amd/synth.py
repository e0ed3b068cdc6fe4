"""Synthetic Arrow-layout columns generated on the device (SURVEY.md section 8d workloads).

Counter-based: value(col, row) = f(mix64(seed ^ col * PHI ^ row)), so any row range of any column can be
produced independently (row-range sharding across ranks needs no communication).  torch is used only as
the device allocator / elementwise engine of the generator; the checks themselves run in libtgx.
"""
import numpy as np

PHI = 0x9E3779B97F4A7C15
MASK64 = (1 << 64) - 1


def _s64(x):
    """python int (mod 2^64) -> signed int64 value"""
    x &= MASK64
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(x, k):
    """logical shift right on torch int64"""
    return (x >> k) & ((1 << (64 - k)) - 1)


def mix64_torch(x):
    """splitmix64 finaliser on int64 tensors (two's complement wrapping arithmetic)"""
    x = x ^ _lsr(x, 30)
    x = x * _s64(0xBF58476D1CE4E5B9)
    x = x ^ _lsr(x, 27)
    x = x * _s64(0x94D049BB133111EB)
    x = x ^ _lsr(x, 31)
    return x


def mix64_numpy(x):
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


# the 16-column "null + range + unique" table of configs[3] / the north star:
#   kind, has_validity
COLUMNS_16 = [
    ("id_perm", False),      # bijective map of the global row index: all unique
    ("k_mod10", True),       # uniform in [0, N/10): ~10 duplicates per key
    ("i_wide", True), ("i_wide", True), ("i_wide", True), ("i_wide", True), ("i_small", True),
    ("i_wide", False),
    ("f_uniform", True), ("f_normal", True), ("f_uniform", True), ("f_expo", True), ("f_uniform", True),
    ("f_normal", True), ("f_uniform", False), ("f_normal", False),
]
UNIQUE_COLUMNS_16 = [0, 1]
NULL_RATE = 0.05


def perm_multiplier(n_total):
    """odd multiplier coprime to n_total with a*row < 2^63"""
    import math

    a = 6364136223
    while math.gcd(a, n_total) != 1:
        a += 2
    assert a * n_total < (1 << 63)
    return a


def gen_column(kind, col_index, row0, n, n_total, seed, device, chunk=1 << 26):
    """int64/float64 tensor of rows [row0, row0+n) of column `col_index`"""
    import torch

    is_float = kind.startswith("f_")
    out = torch.empty(n, dtype=torch.float64 if is_float else torch.int64, device=device)
    salt = _s64(seed ^ ((col_index + 1) * PHI))
    a = perm_multiplier(n_total)
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        rows = torch.arange(row0 + c0, row0 + c1, dtype=torch.int64, device=device)
        if kind == "id_perm":
            out[c0:c1] = (rows * a + 12345) % n_total
            continue
        h = mix64_torch(rows ^ salt)
        if kind == "k_mod10":
            out[c0:c1] = _lsr(h, 1) % max(1, n_total // 10)
        elif kind == "i_wide":
            out[c0:c1] = (h >> 23)  # arithmetic shift: uniform in [-2^40, 2^40)
        elif kind == "i_small":
            out[c0:c1] = _lsr(h, 1) % 1000 - 500
        else:
            u = _lsr(h, 11).to(torch.float64) * (1.0 / (1 << 53))
            if kind == "f_uniform":
                out[c0:c1] = u * 1000.0
            elif kind == "f_expo":
                out[c0:c1] = -torch.log1p(-u) * 50.0
            elif kind == "f_normal":
                h2 = mix64_torch(h ^ _s64(0xD1B54A32D192ED03))
                u2 = _lsr(h2, 11).to(torch.float64) * (1.0 / (1 << 53))
                r = torch.sqrt(-2.0 * torch.log1p(-u))
                out[c0:c1] = r * torch.cos(6.283185307179586 * u2)
            else:
                raise ValueError(kind)
        del rows, h
    return out


def gen_validity(col_index, row0, n, seed, device, null_rate=NULL_RATE, chunk=1 << 26):
    """LSB-first validity bitmap (uint8 tensor, padded to 64 bytes) for rows [row0, row0+n); row0 % 8 == 0"""
    import torch

    assert row0 % 8 == 0
    nbytes = (n + 7) // 8
    out = torch.zeros(((nbytes + 63) // 64 + 1) * 64, dtype=torch.uint8, device=device)
    salt = _s64(seed ^ ((col_index + 101) * PHI))
    thresh = int(null_rate * (1 << 53))
    weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=device)
    assert chunk % 8 == 0
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        m = c1 - c0
        rows = torch.arange(row0 + c0, row0 + c1, dtype=torch.int64, device=device)
        valid = (_lsr(mix64_torch(rows ^ salt), 11) >= thresh)
        pad = (-m) % 8
        if pad:
            valid = torch.cat([valid, torch.zeros(pad, dtype=torch.bool, device=device)])
        packed = (valid.view(-1, 8).to(torch.int32) * weights).sum(dim=1).to(torch.uint8)
        out[c0 // 8: c0 // 8 + packed.numel()] = packed
        del rows, valid, packed
    return out


def make_table(layout, row0, n, n_total, seed, device):
    """list of (values tensor, validity tensor or None) per column of `layout`"""
    cols = []
    for ci, (kind, has_validity) in enumerate(layout):
        vals = gen_column(kind, ci, row0, n, n_total, seed, device)
        validity = gen_validity(ci, row0, n, seed, device) if has_validity else None
        cols.append((vals, validity))
    if str(device).startswith("cuda"):
        # the columns are written by torch's stream; a tgx state works on a stream of its own and reads them as they
        # are when its kernels run (include/tgx.h): hand them over complete
        import torch

        torch.cuda.synchronize()
    return cols


def algorithmic_bytes(layout, n):
    """8 B/row of values + 1 bit/row of validity, each column counted once"""
    total = 0
    for _, has_validity in layout:
        total += 8 * n + ((n + 7) // 8 if has_validity else 0)
    return total
