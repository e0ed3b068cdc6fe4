"""-m gpu: pattern checks through the HIP match kernel vs the oracle (bit-exact match counts)."""
import numpy as np
import pytest

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import pad_validity, run_plan, to_device
from test_regex_host import pattern_of

pytestmark = pytest.mark.gpu


def utf8_column(offsets, data, validity, device, offset=0, length=None, large=False):
    validity = pad_validity(validity)
    data = np.concatenate([data, np.zeros(16, np.uint8)])
    offs = offsets.astype(np.int64) if large else offsets
    if device:
        offs_d, data_d, val_d = to_device(offs), to_device(data), to_device(validity)
    else:
        offs_d, data_d, val_d = offs, data, validity
    n = (len(offsets) - 1 - offset) if length is None else length
    return T.Column(T.LARGE_UTF8 if large else T.UTF8, n, offsets=offs_d, data=data_d, validity=val_d,
                    offset=offset)


def synth_strings(rng, n, kind):
    out = []
    for i in range(n):
        r = rng.random()
        if kind == "email":
            if r < 0.80:
                s = "user%d@example%d.com" % (i, i % 1000)
            elif r < 0.85:
                s = "  padded%d@host.org " % i
            elif r < 0.90:
                s = "no-at-sign-%d.example.com" % i
            elif r < 0.93:
                s = "Ünï%d@exämple.com" % i
            elif r < 0.95:
                s = ""
            elif r < 0.97:
                s = "a@b@c%d.com" % i
            else:
                s = None
        elif kind == "mixed":
            choices = ["192.168.%d.%d" % (i % 256, (i * 7) % 300), "2023-12-25T10:30:%02dZ" % (i % 60),
                       "ABC%03d" % (i % 1000), "abc%03d" % (i % 1000), "550e8400-e29b-41d4-a716-%012x" % i,
                       "123-45-%04d" % (i % 10000), "٣٤-%d" % i, "x" * (i % 70), None, "line\nbreak%d" % i]
            s = choices[int(r * len(choices)) % len(choices)]
        out.append(s)
    return out


def test_reference_format_vectors_on_gpu(golden):
    for case in golden["format"]:
        vals = case["values"]
        if not vals:
            continue
        pat = pattern_of(case, golden["patterns"])
        flags = (T.FLAG_TRIM if case.get("trim") else 0) | \
                (T.FLAG_CASE_INSENSITIVE if case.get("case_sensitive") is False else 0) | \
                (T.FLAG_NULL_IS_VALID if case.get("null_is_valid", True) else 0)
        offs, data, validity = orc.utf8_from_list(vals)
        for device in (True, False):
            res, _, _ = run_plan([spec(T.REGEX_MATCH, 0, flags=flags, pattern=pat)],
                                 [[utf8_column(offs, data, validity, device)]])
            assert res[0].total == len(vals)
            assert res[0].matches / res[0].total == case["metric"], case["ref"]


@pytest.mark.parametrize("kind", ["email", "mixed"])
def test_seeded_columns_match_oracle_bit_exact(kind, golden):
    rng = np.random.default_rng(5 if kind == "email" else 6)
    n = 200_003
    vals = synth_strings(rng, n, kind)
    offs, data, validity = orc.utf8_from_list(vals)
    P = golden["patterns"]
    checks = [(r"@", 0), (r"^[^@]+@[^@]+\.[^@]+$", T.FLAG_NULL_IS_VALID), (P["email"], T.FLAG_TRIM),
              (P["email"], T.FLAG_NULL_IS_VALID), (P["ipv4"], 0), (P["iso8601_datetime"], T.FLAG_NULL_IS_VALID),
              (r"^[A-Z]{3}\d{3}$", T.FLAG_CASE_INSENSITIVE), (P["uuid"], 0), (P["social_security_number"], T.FLAG_TRIM),
              (r"^\d+-\d+$", 0), (r"^x{10,61}$", 0), (r"^.*$", 0)]
    specs = [spec(T.REGEX_MATCH, 0, flags=f, pattern=p) for p, f in checks]
    res, _, _ = run_plan(specs, [[utf8_column(offs, data, validity, True)]])
    for (p, f), r in zip(checks, res):
        want = orc.Regex(p, bool(f & T.FLAG_CASE_INSENSITIVE)).count_utf8(
            offs, data, validity, trim=bool(f & T.FLAG_TRIM), null_is_valid=bool(f & T.FLAG_NULL_IS_VALID))
        assert (r.total, r.matches) == (want.total, want.matches), p


def test_sliced_large_offsets_batches_and_merge(golden):
    rng = np.random.default_rng(9)
    n = 50_000
    vals = synth_strings(rng, n, "email")
    offs, data, validity = orc.utf8_from_list(vals)
    pat = golden["patterns"]["email"]
    flags = T.FLAG_TRIM
    T.init()
    plan = T.Plan([spec(T.REGEX_MATCH, 0, flags=flags, pattern=pat), spec(T.COUNT, 0)])
    a, b = T.State(plan), T.State(plan)
    cuts = [0, 7, 12_345, 12_346, 40_001, n]
    for i, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
        col = utf8_column(offs, data, validity, device=(i % 2 == 0), offset=lo, length=hi - lo, large=(i >= 3))
        (a if i < 3 else b).update([col])
    a.merge([T.State.deserialize(plan, b.serialize())])
    res = a.finalize()
    want = orc.Regex(pat).count_utf8(offs, data, validity, trim=True, null_is_valid=False)
    assert (res[0].total, res[0].matches) == (want.total, want.matches)
    c = orc.count(validity, n)
    assert (res[1].total, res[1].non_null) == (c.total, c.non_null)


def test_big_table_automaton_runs_from_global_memory():
    """a DFA larger than the 48 KiB LDS budget takes the global-memory table path"""
    pat = r"^(?:[a-z]{1,40}\d{1,40}[A-Z]{1,40}){1,3}$"
    vals = ["abc123XYZ", "a1A" * 3, "abc", "a1Ab2Bc3Cd4D", "", None, "zz99QQzz9Q"] * 1000
    offs, data, validity = orc.utf8_from_list(vals)
    res, _, _ = run_plan([spec(T.REGEX_MATCH, 0, pattern=pat)], [[utf8_column(offs, data, validity, True)]])
    want = orc.Regex(pat).count_utf8(offs, data, validity, null_is_valid=False)
    assert (res[0].total, res[0].matches) == (want.total, want.matches)


def test_unaligned_buffers(golden):
    """value bytes / offsets / validity that do not start on 16-byte boundaries (views into larger buffers)"""
    import torch

    rng = np.random.default_rng(12)
    vals = synth_strings(rng, 20_001, "email")
    offs, data, validity = orc.utf8_from_list(vals)
    want = orc.Regex(golden["patterns"]["email"]).count_utf8(offs, data, validity, trim=True, null_is_valid=True)
    wantd = orc.distinct_utf8(offs, data, validity)
    for shift in (1, 3, 8, 13):
        d = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
        d[shift:shift + len(data)] = torch.from_numpy(data).cuda()
        v = torch.zeros(len(validity) + 64, dtype=torch.uint8, device="cuda")
        v[shift:shift + len(validity)] = torch.from_numpy(validity).cuda()
        o = torch.zeros(len(offs) + 8, dtype=torch.int32, device="cuda")
        o[1:1 + len(offs)] = torch.from_numpy(offs).cuda()
        col = T.Column(T.UTF8, len(vals), offsets=o[1:], data=d[shift:], validity=v[shift:])
        res, _, _ = run_plan([spec(T.REGEX_MATCH, 0, pattern=golden["patterns"]["email"],
                                   flags=T.FLAG_TRIM | T.FLAG_NULL_IS_VALID), spec(T.DISTINCT, 0)], [[col]])
        assert (res[0].total, res[0].matches) == (want.total, want.matches), shift
        assert (res[1].non_null, res[1].distinct) == (wantd.non_null, wantd.distinct), shift


def test_chunk_walk_boundaries():
    """The staged walk takes a value eight bytes at a time from aligned LDS words (kernels/regex.hip: ChunkFeed,
    walk2_staged): every value length 0..40 at every start alignment, 128-row spans that just fit / just exceed the
    4 KiB stage (whole step, then half steps, then values read from global memory), values longer than the stage,
    patterns that decide early (dead / matched) and late (`$`-anchored), TRIM on both ends."""
    rng = np.random.default_rng(2024)
    alphabet = np.frombuffer(b"ab01@. -_Xz9", dtype=np.uint8)

    def rand_str(n):
        return bytes(alphabet[rng.integers(0, len(alphabet), size=n)]).decode()

    vals = []
    for length in range(0, 41):          # every length, shifting the alignment of what follows
        for _ in range(6):
            vals.append(rand_str(length))
    for total in (4096 - 64, 4096 - 16, 4096, 4096 + 16, 2 * 4096 - 16, 2 * 4096 + 64):
        # 128 consecutive values whose span is `total` bytes (plus up to 15 bytes of alignment slack)
        each, rest = divmod(total, 128)
        vals += [rand_str(each + (1 if k < rest else 0)) for k in range(128)]
    vals += [rand_str(5000), "x@y.z", rand_str(4097), None, rand_str(9000) + "@end.com", "", rand_str(33)] * 3
    vals += ["  " + rand_str(k) + "   " for k in range(0, 24)] + [" ", "  ", "   "]
    vals += [None if rng.random() < 0.1 else rand_str(int(rng.integers(0, 70))) for _ in range(20_000)]
    offs, data, validity = orc.utf8_from_list(vals)
    pats = [(r"@", 0), (r"^[ab01]+$", 0), (r"\.com$", T.FLAG_TRIM), (r"^[^@]+@[^@]+\.[^@]+$", T.FLAG_TRIM | T.FLAG_NULL_IS_VALID),
            (r"a.*9$", 0), (r"^$", T.FLAG_TRIM), (r"(?:ab|ba){2}", T.FLAG_CASE_INSENSITIVE), (r"X{2,}z?9", 0)]
    for large in (False, True):
        for lead in (0, 5):  # a sliced batch: the step no longer starts at row 0 / byte 0
            col = utf8_column(offs, data, validity, True, offset=lead, length=len(vals) - lead, large=large)
            res, _, _ = run_plan([spec(T.REGEX_MATCH, 0, pattern=p, flags=f) for p, f in pats], [[col]])
            for r, (p, f) in zip(res, pats):
                want = orc.Regex(p, case_insensitive=bool(f & T.FLAG_CASE_INSENSITIVE)).count_utf8(
                    offs, data, validity, n=len(vals) - lead, offset=lead, trim=bool(f & T.FLAG_TRIM),
                    null_is_valid=bool(f & T.FLAG_NULL_IS_VALID))
                assert (r.total, r.matches) == (want.total, want.matches), (p, large, lead)


@pytest.mark.parametrize("device", [True, False])
def test_multiline_word_boundary_and_set_operation_patterns_on_gpu(device):
    """(?m) anchors, (?-u) ASCII word boundaries / classes and class set operations (round 3): the match kernel walks
    the automaton whose states carry the context of the byte before -- counts equal the oracle's VM on a column of
    multi-line values, alone and as one product automaton of all patterns of the column"""
    rng = np.random.default_rng(11)
    words = ["foo", "food", "a foo b", "afoo", "foo_", "é foo", "fooé", "select 1", "SELECTED", "x\nfoo\ny", "abc\n123",
             "123", "a123", "12\n345", "\n", "", "abc\n", "m", "amz", "bcx", "x\nabc", "abc", " ", "tab\there"]
    vals = [None if rng.random() < 0.03 else rng.choice(words) + ("" if rng.random() < 0.5 else "\n" + rng.choice(words))
            for _ in range(60_000)]
    offs, data, validity = orc.utf8_from_list(vals)
    pats = [r"(?m)^foo$", r"(?m)^\d+$", r"(?-u:\b)foo(?-u:\b)", r"(?i)(?-u:\b)select(?-u:\b)", r"(?m)^[a-z--m]+$",
            r"^[a-z&&[^m]]+$", r"(?-u:\B)oo", r"(?m)^$"]
    specs = [spec(T.REGEX_MATCH, 0, pattern=p, flags=T.FLAG_NULL_IS_VALID if k % 2 else 0) for k, p in enumerate(pats)]
    res, _, _ = run_plan(specs, [[utf8_column(offs, data, validity, device)]])
    for k, p in enumerate(pats):
        want = orc.Regex(p).count_utf8(offs, data, validity, null_is_valid=bool(k % 2))
        assert (res[k].total, res[k].matches) == (want.total, want.matches), p
        assert 0 < res[k].matches < res[k].total, p  # (the column exercises both verdicts of every pattern)


@pytest.mark.parametrize("device", [True, False])
@pytest.mark.parametrize("view", [False, True])
def test_unicode_word_boundaries_on_gpu(device, view):
    """`\\b` / `\\B` as Rust's `regex` takes them by default (round 4): the automaton's context follows a character's
    bytes through a classifier of \\w and a thread past the assertion carries what the next character has to be -- ~650
    states x ~100 byte classes, so the kernel walks a table in global memory.  Counts equal the oracle's VM (code points
    on either side) on a column of words in several scripts, Utf8 and Utf8View."""
    rng = np.random.default_rng(12)
    words = ["word", "a word here", "swordfish", "wörd", "éword", "word é", "日本 word。", "naïve", "naïvely", "12", "a12",
             "x 12 y", "١٢", "x١٢", "αβγ δ", "xαβγ", "select", "xselect", "drop x", "", " ", "é", "日本", "a_b", "éx",
             "a‍b", "😀a", "on", "upon", "bonbon", "word, word", "WORD"]
    vals = [None if rng.random() < 0.03 else rng.choice(words) + ("" if rng.random() < 0.6 else " " + rng.choice(words))
            for _ in range(40_000)]
    offs, data, validity = orc.utf8_from_list(vals)
    pats = [r"\bword\b", r"\b\d+\b", r"\bé", r"\Bon\b", r"(?i)\bWORD\b", r"\b\p{Greek}+\b", r"\b(?:select|drop)\b", r"\w\b\W"]
    specs = [spec(T.REGEX_MATCH, 0, pattern=p, flags=T.FLAG_NULL_IS_VALID if k % 2 else 0) for k, p in enumerate(pats)]
    if view:
        from test_gpu_utf8view import view_column

        col = view_column(vals, rng, device)
    else:
        col = utf8_column(offs, data, validity, device)
    res, _, _ = run_plan(specs, [[col]])
    for k, p in enumerate(pats):
        want = orc.Regex(p).count_utf8(offs, data, validity, null_is_valid=bool(k % 2))
        assert (res[k].total, res[k].matches) == (want.total, want.matches), p
        assert 0 < res[k].matches < res[k].total, p


def test_case_insensitive_patterns_over_a_column_of_fold_edge_cases(golden):
    """`~*` (format.rs:756-776; TGX_FLAG_CASE_INSENSITIVE) on the device over a column that holds the characters where
    simple case folding, the Turkic lines and the Unicode version matter: counts bit-exact against the oracle (fold
    table of another origin: tests/test_unicode_tables.py) AND against the PyPI-`regex` vectors of
    tests/golden/regex_crosscheck_r5.json, summed per pattern."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "regex_crosscheck_r5.json")) as f:
        cases = json.load(f)["cases"]
    by_pattern = {}
    for c in cases:
        by_pattern.setdefault((c["pattern"], c["flags"]), {})[c["input"]] = c["match"]
    # one column: every input any pattern has a vector for, repeated with a NULL and an empty value in between
    inputs = sorted({c["input"] for c in cases})
    vals = []
    for rep in range(3):
        for i, s in enumerate(inputs):
            vals.append(s)
            if (i + rep) % 97 == 0:
                vals.append(None)
    offs, data, validity = orc.utf8_from_list(vals)
    keys = sorted(by_pattern)
    for g0 in range(0, len(keys), 6):
        group = keys[g0:g0 + 6]
        specs = [spec(T.REGEX_MATCH, 0, flags=f, pattern=p) for p, f in group]
        for device in (True, False):
            res, _, _ = run_plan(specs, [[utf8_column(offs, data, validity, device)]])
            for (p, f), r in zip(group, res):
                want = orc.Regex(p, bool(f & T.FLAG_CASE_INSENSITIVE)).count_utf8(offs, data, validity, null_is_valid=False)
                assert (r.total, r.matches) == (want.total, want.matches), (p, f)
                known = by_pattern[(p, f)]
                if all(s in known for s in inputs):   # (every pattern saw the seed inputs; mutated ones differ)
                    assert r.matches == 3 * sum(known[s] for s in inputs), (p, f)
    # the vectors themselves, pattern by pattern, each over exactly its own inputs
    for (p, f), known in by_pattern.items():
        ins = sorted(known)
        o2, d2, v2 = orc.utf8_from_list(ins)
        res, _, _ = run_plan([spec(T.REGEX_MATCH, 0, flags=f, pattern=p)], [[utf8_column(o2, d2, v2, True)]])
        assert res[0].matches == sum(known.values()), (p, f)


@pytest.mark.parametrize("layout", ["plain", "view", "dictionary", "host"])
def test_counted_unicode_classes_on_gpu(layout):
    """`^C{m,n}$` with a Unicode class -- the automaton of `^C*$` and a character count (length_filter_kernel; VERDICT r5
    "What's missing" 4: round 5 refused these four) -- against the oracle's VM over names, user names and junk of
    lengths on both sides of the bounds; with TRIM and NULL-is-valid; in one plan with a pattern that is walked as usual."""
    import random

    from test_regex_counted_classes import DIGITS, LETTERS, MARKS, OTHER, VERDICT_PATTERNS, subjects

    rng = random.Random(11)
    vals = []
    for m, n in ((1, 64), (2, 50), (1, 100)):
        vals += subjects(rng, m, n, 4000)
    vals += [None] * 300 + ["".join(rng.choice(LETTERS + DIGITS + MARKS) for _ in range(rng.randrange(0, 120))) for _ in range(8000)]
    rng.shuffle(vals)
    offs, data, validity = orc.utf8_from_list(vals)
    checks = [(p, f) for p in VERDICT_PATTERNS for f in (0, T.FLAG_TRIM, T.FLAG_NULL_IS_VALID)] + [(r"^\w+$", 0), (r"\d", 0)]
    specs = [spec(T.REGEX_MATCH, 0, flags=f, pattern=p) for p, f in checks]
    nrng = np.random.default_rng(3)
    if layout == "view":
        from test_gpu_utf8view import view_column
        col = view_column(vals, nrng, True)
    elif layout == "dictionary":
        from test_gpu_dictionary import encode
        col = encode(vals, nrng, device=True)
    else:
        col = utf8_column(offs, data, validity, layout != "host")
    res, _, _ = run_plan(specs, [[col]])
    for (p, f), r in zip(checks, res):
        want = orc.Regex(p).count_utf8(offs, data, validity, trim=bool(f & T.FLAG_TRIM), null_is_valid=bool(f & T.FLAG_NULL_IS_VALID))
        assert (r.total, r.matches) == (want.total, want.matches), (p, f)
