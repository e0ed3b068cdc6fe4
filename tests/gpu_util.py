"""Helpers shared by the -m gpu parity tests: seeded Arrow-layout columns on the host (numpy) and on the
device (torch), fed to libtgx through the C ABI and to the oracle through oracle_binding."""
import numpy as np

import oracle_binding as orc
import term_amd as T


def make_i64(rng, n, lo, hi, null_frac=0.0):
    vals = rng.integers(lo, hi, size=n, dtype=np.int64)
    validity = None
    if null_frac > 0:
        mask = rng.random(n) >= null_frac
        validity = orc.pack_validity(mask)
    return vals, validity


def make_f64(rng, n, kind="uniform", null_frac=0.0):
    if kind == "uniform":
        vals = rng.random(n) * 1000.0
    elif kind == "normal":
        vals = rng.standard_normal(n)
    elif kind == "wide":
        vals = rng.standard_normal(n) * np.exp(rng.uniform(-20, 20, size=n))
    else:
        raise ValueError(kind)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    validity = None
    if null_frac > 0:
        mask = rng.random(n) >= null_frac
        validity = orc.pack_validity(mask)
    return vals, validity


def pad_validity(validity, multiple=64):
    """Arrow buffers are padded; keep the test buffers readable in whole 8-byte words."""
    if validity is None:
        return None
    pad = (-len(validity)) % multiple
    return np.concatenate([validity, np.zeros(pad + multiple, dtype=np.uint8)])


def to_device(arr):
    import torch

    if arr is None:
        return None
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda()


def numeric_column(vals, validity, device, offset=0, length=None):
    validity = pad_validity(validity)
    if device:
        vals_d, validity_d = to_device(vals), to_device(validity)
    else:
        vals_d, validity_d = vals, validity
    n = (len(vals) - offset) if length is None else length
    ctor = T.Column.int64 if vals.dtype == np.int64 else T.Column.float64
    return ctor(vals_d, validity_d, length=n, offset=offset)


def run_plan(specs, batches, hint=0):
    """specs: list of term_amd.CheckSpec; batches: list of column lists. Returns results."""
    T.init(distinct_capacity_hint=hint)
    plan = T.Plan(specs)
    st = T.State(plan)
    for cols in batches:
        st.update(cols)
    return st.finalize(), plan, st


def rel_err(a, b):
    if a == b:
        return 0.0
    return abs(a - b) / max(abs(a), abs(b), 1e-300)
