"""CPU tests of the drop-in boundary: libtgx.so loads without a GPU, exports every symbol include/*.h declares,
its structs match the ctypes binding, and every compute entry point fails loudly (no CPU fallback)."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

import term_amd as T
from term_amd._lib import CheckSpec, Result, _Column, _Error, _Options, spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def has_gpu():
    try:
        import torch

        return torch.cuda.device_count() > 0
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    lib = T.lib()
    names = T.abi_symbols()
    assert len(names) >= 30 and "tgx_update" in names and "tgx_host_run_suite_json" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.tgx_abi_version() == 6
    assert lib.tgx_status_name(6) == b"TGX_NO_DEVICE"


def test_struct_layouts_match_the_header():
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "tgx.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(tgx_column), sizeof(tgx_check_spec), sizeof(tgx_result),
             sizeof(tgx_error), sizeof(tgx_options), offsetof(tgx_result, distinct), offsetof(tgx_result, kll_n),
             offsetof(tgx_column, dictionary));
      return 0;
    }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        with open(c, "w") as f:
            f.write(src)
        exe = os.path.join(d, "s")
        subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        got = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(_Column), C.sizeof(CheckSpec), C.sizeof(Result), C.sizeof(_Error), C.sizeof(_Options),
            Result.distinct.offset, Result.kll_n.offset, _Column.dictionary.offset]
    assert got == want


@pytest.mark.skipif(has_gpu(), reason="checks the no-device behaviour")
def test_no_device_means_loud_failure_not_a_cpu_path():
    with pytest.raises(T.TgxError) as e:
        T.init()
    assert e.value.status == "TGX_NO_DEVICE" and "no CPU path" in str(e.value)
    plan = T.Plan([spec(T.NUMERIC_STATS, 0), spec(T.DISTINCT, 0), spec(T.REGEX_MATCH, 1, pattern="a+")])
    st = T.State(plan)
    a = np.arange(8, dtype=np.int64)
    with pytest.raises(T.TgxError) as e:
        st.update([T.Column.int64(a), T.Column.int64(a)])
    assert e.value.status == "TGX_NO_DEVICE"


def test_plan_validation_needs_no_device():
    with pytest.raises(T.TgxError) as e:
        T.Plan([spec(T.KLL, 0, kll_k=1)])
    assert "k must be at least 2" in str(e.value)  # kll_sketch.rs:167-169
    with pytest.raises(T.TgxError):
        T.Plan([spec(T.COMOMENTS, 0)])  # needs column2
    with pytest.raises(T.TgxError) as e:
        T.Plan([spec(T.REGEX_MATCH, 0, pattern=r"\b{start}word")])  # the \b{..} variants: outside the engine
    assert e.value.status == "TGX_UNSUPPORTED"  # the shim falls back to the stock SQL constraint
    # Unicode word boundaries are part of the engine since round 4 (whole characters on either side, \w as in the tables)
    T.Plan([spec(T.REGEX_MATCH, 0, pattern=r"\bword\b")])
    import ctypes as C

    for value, want in (("a word here", 1), ("swordfish", 0), ("wörd", 0), ("éword", 0), ("word é", 1), ("日本 word。", 1)):
        m, err, v, pat = C.c_int32(-1), T._lib._Error(), value.encode(), rb"\bword\b"
        assert T.lib().tgx_regex_is_match(pat, len(pat), 0, v, len(v), C.byref(m), C.byref(err)) == 0
        assert m.value == want, value
    p = T.Plan([spec(T.COUNT, 0), spec(T.NUMERIC_STATS, 0), spec(T.NUMERIC_STATS, 0, flags=T.FLAG_VARIANCE)])
    assert T.lib().tgx_plan_num_specs(p.h) == 3


def test_blob_roundtrip_without_device():
    from term_amd import wire

    plan = T.Plan([spec(T.COUNT, 0), spec(T.NUMERIC_STATS, 1), spec(T.COMOMENTS, 1, column2=2), spec(T.KLL, 1, kll_k=200),
                   spec(T.REGEX_MATCH, 3, pattern="@", flags=T.FLAG_NULL_IS_VALID)])
    blob = wire.pack(scan=[wire.scan_acc(10, 8, -3, 99, 400), wire.scan_acc(10, 10, 0.5, 2.5, 15.0, is_float=True)],
                     count=[wire.count_acc(10, 7)],
                     comoments=[wire.comoment_acc(10, 8, 1.0, 2.0, 3.0, 4.0, 5.0)],
                     kll=[wire.kll_state(200, 3, 1.0, 3.0, [[1.0, 2.0, 3.0]])], regex=[wire.regex_counts(10, 4)])
    # the plan has ONE scan task (column 1) + one for column 2? no: COMOMENTS does not scan; fix the expectation
    with pytest.raises(T.TgxError):
        T.State.deserialize(plan, blob)  # 2 scan accs for a plan with 1 scan task
    blob = wire.pack(scan=[wire.scan_acc(10, 8, -3, 99, 400)], count=[wire.count_acc(10, 7)],
                     comoments=[wire.comoment_acc(10, 8, 1.0, 2.0, 3.0, 4.0, 5.0)],
                     kll=[wire.kll_state(200, 3, 1.0, 3.0, [[1.0, 2.0, 3.0]])], regex=[wire.regex_counts(10, 4)])
    st = T.State.deserialize(plan, blob)
    res = st.finalize()
    assert (res[0].total, res[0].non_null) == (10, 7)
    assert (res[1].min_i, res[1].max_i, res[1].sum_i, res[1].mean) == (-3, 99, 400, 50.0)
    assert (res[2].non_null, res[2].sum_xy) == (8, 5.0)
    assert res[3].kll_n == 3 and st.kll_quantile(3, 0.5) == 2.0
    assert (res[4].total, res[4].matches) == (10, 4)
    assert st.serialize() == blob  # stable wire form
    with pytest.raises(T.TgxError):
        T.State.deserialize(plan, blob[:-3])
    with pytest.raises(T.TgxError):
        T.State.deserialize(plan, b"nope" + blob[4:])


def test_fingerprint_key_of_a_plan(monkeypatch):
    """tgx_plan_create draws a key from the OS (two plans differ), TGX_FINGERPRINT_KEY fixes it, the setter replaces it
    until the plan's first state exists; a blob packed from plain numbers carries none"""
    from term_amd import wire

    sp = [spec(T.DISTINCT, 0)]
    a, b = T.Plan(sp), T.Plan(sp)
    assert len(a.fingerprint_key()) == 16 and a.fingerprint_key() != b.fingerprint_key() and any(a.fingerprint_key())
    monkeypatch.setenv("TGX_FINGERPRINT_KEY", "000102030405060708090a0b0c0d0e0f")
    assert T.Plan(sp).fingerprint_key() == bytes(range(16))
    monkeypatch.setenv("TGX_FINGERPRINT_KEY", "not-hex")
    with pytest.raises(T.TgxError):
        T.Plan(sp)
    monkeypatch.delenv("TGX_FINGERPRINT_KEY")
    a.set_fingerprint_key(bytes(range(16, 32)))
    assert a.fingerprint_key() == bytes(range(16, 32))
    with pytest.raises(ValueError):
        a.set_fingerprint_key(b"short")
    p = T.Plan([spec(T.COUNT, 0)])
    blob = wire.pack(count=[wire.count_acc(10, 7)])
    assert T.blob_fingerprint_key(blob) is None
    st = T.State.deserialize(p, blob)  # (a host-only state: the key is fixed from here on)
    with pytest.raises(T.TgxError, match="fixed once a state"):
        p.set_fingerprint_key(bytes(16))
    assert st.finalize()[0].total == 10
    with pytest.raises(T.TgxError):
        T.blob_fingerprint_key(b"garbage")
