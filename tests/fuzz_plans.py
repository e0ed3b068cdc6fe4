"""A seeded differential tester for the whole C ABI path: random plans over random tables in random batchings, the
device's answers against the CPU oracle's on the logical table (tests/test_gpu_fuzz.py runs a fixed set of seeds;
tools/fuzz_device.py any range).  One seed fixes everything: columns (type, shape of the values, NULL rate), checks,
how the rows are cut into tgx_update calls (one batch, ragged cuts, a stream of DataFusion-sized batches; DEVICE or
HOST buffers; Arrow offsets), and what happens to the state before it is read (finalize / blob round trip / several
states merged).  Bit-exact wherever the reference is (counts, MIN / MAX, integer sums, DISTINCT, HyperLogLog
registers -> estimate); 1e-6 relative for floating-point aggregates (north star)."""
import math
import sys

import numpy as np

import oracle_binding as orc
import term_amd as T
from _lib_spec import spec
from gpu_util import numeric_column, pad_validity, rel_err, to_device

TOL = 1e-6


# ---- tables -----------------------------------------------------------------------------------------------------------
def int_values(rng, n, shape):
    if shape == "permutation":
        return rng.permutation(n).astype(np.int64) * int(rng.integers(1, 4)) + int(rng.integers(-10**6, 10**12))
    if shape == "ascending":
        return np.arange(n, dtype=np.int64) * int(rng.integers(1, 5)) + int(rng.integers(-10**9, 10**9))
    if shape == "descending":
        return int(rng.integers(0, 10**10)) - np.arange(n, dtype=np.int64) * int(rng.integers(1, 3))
    if shape == "few":
        return rng.integers(-3, int(rng.integers(1, 200)), size=n, dtype=np.int64)  # (includes -1: the all-ones key)
    if shape == "tenth":
        return rng.integers(0, max(2, n // 10), size=n, dtype=np.int64)
    if shape == "wide":
        return rng.integers(-(2**62), 2**62, size=n, dtype=np.int64)
    if shape == "constant":
        return np.full(n, int(rng.integers(-5, 5)), dtype=np.int64)
    if shape == "strays":  # dense and in order, a few keys from far away
        v = np.arange(n, dtype=np.int64)
        m = rng.random(n) < 0.002
        v[m] = rng.integers(-(2**40), 2**40, size=int(m.sum()), dtype=np.int64)
        return v
    if shape == "blocks":  # sorted blocks, shuffled
        v = np.arange(n, dtype=np.int64)
        cut = list(range(0, n, 7919)) or [0]
        return np.concatenate([v[c:c + 7919] for c in (cut[k] for k in rng.permutation(len(cut)))]) if n else v
    raise AssertionError(shape)


INT_SHAPES = ["permutation", "ascending", "descending", "few", "tenth", "wide", "constant", "strays", "blocks"]


def float_values(rng, n, shape):
    if shape == "uniform":
        return rng.random(n) * 1000.0
    if shape == "normal":
        return rng.standard_normal(n) * float(rng.choice([1.0, 1e-3, 1e6])) + float(rng.choice([0.0, 1e9]))
    if shape == "rounded":
        v = np.round(rng.standard_normal(n), 2)
        v[rng.random(n) < 0.01] = -0.0
        return v
    if shape == "ints":
        return rng.integers(-1000, 1000, size=n).astype(np.float64)
    if shape == "ascending":
        return np.arange(n, dtype=np.float64) * 0.5
    if shape == "specials":  # NaN (two payloads), infinities, signed zeros, denormals among ordinary values
        v = np.round(rng.standard_normal(n) * 10, 1)
        pool = np.array([np.nan, -np.nan, np.inf, -np.inf, 0.0, -0.0, 5e-324, -5e-324, 1e308], dtype=np.float64)
        m = rng.random(n) < 0.05
        v[m] = pool[rng.integers(0, len(pool), size=int(m.sum()))]
        return v
    raise AssertionError(shape)


FLOAT_SHAPES = ["uniform", "normal", "rounded", "ints", "ascending", "specials"]


# ---- strings ----------------------------------------------------------------------------------------------------------
WORD_PARTS = ["a", "Z", "user", "@", ".", "com", "é", "ß", "你", "🦀", " ", "0", "42", "-", "_", "x" * 17]
PATTERNS = [("@", 0), (r"^[a-zA-Z0-9]+$", 0), (r"^\s*user", 0), (r"(?i)^USER", 0), (r"[0-9]{2}", 0), (r"^$", 0),
            (r"é|你", 0), (r"^[^@]+@[^@]+\.[a-z]+$", 0), (r"com$", "trim"), (r"^user", "ci"), (r"\d", "nullvalid")]


def string_values(rng, n, vocab):
    """n strings drawn from `vocab` distinct words -> (offsets int32/int64-able, data uint8, list of the words, picks)"""
    words = []
    seen = set()
    while len(words) < vocab:
        k = int(rng.integers(0, 7))
        w = "".join(WORD_PARTS[int(j)] for j in rng.integers(0, len(WORD_PARTS), size=k)) + (
            "" if len(words) < 3 else "#%d" % len(words))
        if w in seen:
            continue
        seen.add(w)
        words.append(w)
    enc = [w.encode("utf-8") for w in words]
    picks = rng.integers(0, vocab, size=n)
    lens = np.array([len(e) for e in enc], dtype=np.int64)[picks]
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    data = np.frombuffer(b"".join(enc[k] for k in picks) or b"\0", dtype=np.uint8).copy()
    return offsets, data, words, picks


class Buffers:
    """the Arrow buffers of one column, on the host and (uploaded once, on first use) on the device"""

    def __init__(self, kind, vals, vb, extra, mask, seed, present=None):
        self.kind = kind
        self.present = present  # (round 5) an Int64 column handed over as a narrower / unsigned / Boolean Arrow type
        self.host = {"validity": pad_validity(vb)}
        self.layout = "plain"
        if kind == "s" and extra[2] != "plain":
            # Utf8View / Dictionary<Int32, Utf8>: built per batch from the Python values (tests/test_gpu_utf8view.py,
            # tests/test_gpu_dictionary.py); small tables only
            self.layout = extra[2]
            self.values = [extra[3][k] if m else None for k, m in zip(extra[4], mask)]
            self.rng = np.random.default_rng(seed)
        if kind == "s":
            self.large = extra[1]
            self.host["offsets"] = vals if self.large else vals.astype(np.int32)
            self.host["data"] = np.concatenate([extra[0], np.zeros(16, np.uint8)])
        elif present == "bool":
            self.host["values"] = np.concatenate([orc.pack_validity(vals != 0), np.zeros(64, np.uint8)])
        elif present is not None:
            narrow = vals.astype(present)
            assert (narrow.astype(np.int64) == vals).all() or present == np.uint64
            self.host["values"] = np.concatenate([narrow, np.zeros(64, narrow.dtype)]).view(np.uint8)
        else:
            self.host["values"] = vals
        self.dev = None

    def column(self, on_device, offset, length):
        if self.layout == "view":
            from test_gpu_utf8view import view_column

            lead = int(self.rng.integers(0, 70))  # (an Arrow offset into a longer array)
            return view_column([None] * lead + self.values[offset:offset + length], self.rng, on_device, offset=lead,
                               length=length)
        if self.layout == "dict":
            from test_gpu_dictionary import encode

            return encode(self.values[offset:offset + length], self.rng, repeat_entries=bool(self.rng.integers(0, 2)),
                          large=self.large, device=on_device)
        if on_device and self.dev is None:
            self.dev = {k: to_device(v) for k, v in self.host.items()}
        b = self.dev if on_device else self.host
        if self.kind == "s":
            return T.Column(T.LARGE_UTF8 if self.large else T.UTF8, length, offsets=b["offsets"], data=b["data"],
                            validity=b["validity"], offset=offset)
        if self.present is not None:
            type_id = {"bool": T.BOOL, np.int8: T.INT8, np.int16: T.INT16, np.uint8: T.UINT8, np.uint16: T.UINT16,
                       np.uint32: T.UINT32, np.uint64: T.UINT64}[self.present]
            return T.Column(type_id, length, values=b["values"], validity=b["validity"], offset=offset)
        ctor = {"i": T.Column.int64, "f": T.Column.float64, "i32": T.Column.int32, "f32": T.Column.float32}[self.kind]
        return ctor(b["values"], b["validity"], length=length, offset=offset)


def make_validity(rng, n, rate):
    if rate == 0.0:
        return None, np.ones(n, dtype=bool)
    mask = rng.random(n) >= rate if rate < 1.0 else np.zeros(n, dtype=bool)
    return orc.pack_validity(mask), mask


# ---- one case ---------------------------------------------------------------------------------------------------------
class Case:
    def __init__(self, seed, max_rows=2_600_000, host_only=False):
        rng = self.rng = np.random.default_rng(seed)
        self.seed = seed
        size_class = rng.choice(["tiny", "small", "medium", "big"], p=[0.15, 0.3, 0.3, 0.25])
        n = {"tiny": int(rng.integers(0, 70)), "small": int(rng.integers(70, 20_000)),
             "medium": int(rng.integers(20_000, 400_000)), "big": int(rng.integers(min(1_050_000, max_rows - 1), max_rows))}[size_class]
        self.n = n
        self.cols = []  # (kind 'i'/'f', values, validity bytes or None, mask)
        for _ in range(int(rng.integers(1, 4))):
            rate = float(rng.choice([0.0, 0.0, 0.02, 0.3, 1.0], p=[0.3, 0.2, 0.25, 0.2, 0.05]))
            kind = str(rng.choice(["i", "f", "i32", "f32", "s"], p=[0.45, 0.25, 0.08, 0.07, 0.15]))
            extra = None
            if kind == "i":
                vals = int_values(rng, n, str(rng.choice(INT_SHAPES)))
            elif kind == "f":
                vals = np.ascontiguousarray(float_values(rng, n, str(rng.choice(FLOAT_SHAPES))), dtype=np.float64)
            elif kind == "i32":
                vals = int_values(rng, n, str(rng.choice(["permutation", "ascending", "few", "tenth", "constant"])))
                vals = (vals % (2**31)).astype(np.int32) if rng.random() < 0.5 else vals.astype(np.int32)
            elif kind == "f32":
                with np.errstate(over="ignore"):  # (1e308 becomes inf: one more special value)
                    vals = float_values(rng, n, str(rng.choice(["uniform", "rounded", "ints", "specials"]))).astype(np.float32)
            else:
                vocab = int(rng.choice([1, 3, 100, max(1, n // 10), max(1, n)]))
                vocab = min(vocab, 200_000)
                offsets, data, words, picks = string_values(rng, n, vocab)
                # (data, LargeUtf8?, layout, the values as a Python list for the layouts built row by row)
                layout = str(rng.choice(["plain", "view", "dict"], p=[0.6, 0.2, 0.2])) if n <= 20_000 else "plain"
                vals, extra = offsets, (data, bool(rng.integers(0, 2)), layout, words, picks)
            vb, mask = make_validity(rng, n, rate)
            self.cols.append((kind, vals, vb, mask, extra))
        # checks
        self.specs, self.expect = [], []
        for ci, (kind, vals, vb, mask, extra) in enumerate(self.cols):
            numeric = kind != "s"
            menu = ["count", "stats", "var", "distinct", "mult", "approx"] if numeric else \
                   ["count", "distinct", "mult", "approx", "length", "regex", "regex"]
            picks = [str(x) for x in rng.choice(menu, size=int(rng.integers(1, 4)), replace=False)]
            if "count" in picks:
                self.add(spec(T.COUNT, ci), ("count", ci))
            if "var" in picks:
                self.add(spec(T.NUMERIC_STATS, ci, flags=T.FLAG_VARIANCE), ("stats", ci, True))
            elif "stats" in picks:
                self.add(spec(T.NUMERIC_STATS, ci), ("stats", ci, False))
            exact = not numeric  # (a string column's approximate count is its key set's)
            if "mult" in picks:
                self.add(spec(T.DISTINCT, ci, flags=T.FLAG_MULTIPLICITY), ("distinct", ci, True))
                exact = True
            elif "distinct" in picks:
                self.add(spec(T.DISTINCT, ci), ("distinct", ci, False))
                exact = True
            if "approx" in picks:
                # the lane's estimate -- or the exact count where the key set answers (an exact DISTINCT check or
                # variance lanes on the same column, any string column: tests/test_gpu_hll.py)
                self.add(spec(T.APPROX_DISTINCT, ci), ("approx", ci, exact or "var" in picks))
            if "length" in picks:
                lo = int(rng.integers(0, 6))
                hi = None if rng.random() < 0.4 else lo + int(rng.integers(0, 30))
                self.add(spec(T.LENGTH, ci, length_min=lo, length_max=hi), ("length", ci, lo, hi))
            for _ in range(picks.count("regex")):
                pat, opt = PATTERNS[int(rng.integers(0, len(PATTERNS)))]
                flags = {0: 0, "trim": T.FLAG_TRIM, "ci": T.FLAG_CASE_INSENSITIVE, "nullvalid": T.FLAG_NULL_IS_VALID}[opt]
                if rng.random() < 0.5:
                    flags |= T.FLAG_NULL_IS_VALID
                self.add(spec(T.REGEX_MATCH, ci, flags=flags, pattern=pat), ("regex", ci, pat, flags))
        numeric_cols = [ci for ci, c in enumerate(self.cols) if c[0] in ("i", "f")]
        if len(numeric_cols) >= 2 and rng.random() < 0.4:
            self.add(spec(T.COMOMENTS, numeric_cols[0], column2=numeric_cols[1]), ("comoments", numeric_cols[0], numeric_cols[1]))
        if len(numeric_cols) >= 2 and n <= 300_000 and rng.random() < 0.25:
            self.add(spec(T.SPEARMAN, numeric_cols[0], column2=numeric_cols[1]), ("spearman", numeric_cols[0], numeric_cols[1]))
        # (plain layouts here; tuples over Utf8View / dictionary columns are drawn further down, from a stream of their own)
        key_cols = [ci for ci, c in enumerate(self.cols) if c[0] in ("i", "f") or (c[0] == "s" and c[4][2] == "plain")]
        if len(key_cols) >= 2 and n <= 400_000 and rng.random() < 0.3:  # (the check counts tuples in a Python dict)
            mult = bool(rng.integers(0, 2))
            self.add(spec(T.DISTINCT, key_cols[0], columns=key_cols[:int(rng.integers(2, len(key_cols) + 1))],
                          flags=T.FLAG_MULTIPLICITY if mult else 0), ("tuple", tuple(key_cols), mult))
            self.expect[-1] = ("tuple", tuple(self.tuple_columns()), mult)
        for ci, (kind, vals, vb, mask, extra) in enumerate(self.cols):
            if kind in ("f", "i") and rng.random() < 0.15:
                self.add(spec(T.KLL, ci, kll_k=int(rng.choice([200, 2048]))), ("kll", ci))
        # (round 5, from a stream of its own -- earlier seeds keep their cases: tuples whose components are Utf8View or
        #  dictionary columns: the component is the row's string whatever the layout)
        rng6 = np.random.default_rng([seed, 6])
        laid_out = [ci for ci, c in enumerate(self.cols) if c[0] == "s" and c[4][2] != "plain"]
        if laid_out and len(self.cols) >= 2 and n <= 400_000 and not any(e[0] == "tuple" for e in self.expect) and rng6.random() < 0.35:
            first = laid_out[int(rng6.integers(0, len(laid_out)))]
            others = [ci for ci in range(len(self.cols)) if ci != first]
            rng6.shuffle(others)
            cols6 = [first] + others[:int(rng6.integers(1, min(3, len(others)) + 1))]
            rng6.shuffle(cols6)
            mult = bool(rng6.integers(0, 2))
            self.add(spec(T.DISTINCT, cols6[0], columns=cols6, flags=T.FLAG_MULTIPLICITY if mult else 0), ("tuple", tuple(cols6), mult))
            self.expect[-1] = ("tuple", tuple(self.tuple_columns()), mult)
        # thresholds that steer small batches into the big-batch paths (read per call / per state by the library)
        self.env = {}
        if rng.random() < 0.3:
            self.env["TGX_FP_LISTS_MIN_ROWS"] = str(int(rng.choice([1, 3000, 50_000])))
        if rng.random() < 0.3:
            self.env["TGX_PARTITION_MIN_ROWS"] = str(int(rng.choice([1, 5000, 100_000])))
        if rng.random() < 0.3:
            self.env["TGX_COALESCE_FLUSH_ROWS"] = str(int(rng.choice([10_000, 20_000, 100_000])))
        # batching
        self.mode = str(rng.choice(["one", "cuts", "stream"], p=[0.35, 0.4, 0.25]))
        self.device = str(rng.choice(["device", "host", "mixed"], p=[0.5, 0.3, 0.2]))
        has_spearman = any(e[0] == "spearman" for e in self.expect)
        self.after = str(rng.choice(["finalize", "blob", "merge", "ranks"], p=[0.4, 0.15, 0.25, 0.2]))
        if host_only:  # (no torch in the process: HOST buffers only, no threaded ranks -- tools/run_gpu_host_asan.sh)
            self.device = "host"
            if self.after == "ranks":
                self.after = "finalize"
        if has_spearman and self.after in ("blob", "merge"):
            self.after = "finalize"  # (not mergeable: TG/analyzers/advanced/correlation.rs:103-109)
        if self.mode == "one" or n == 0 or self.after == "ranks":
            self.cuts = [0, n]
        elif self.mode == "cuts":
            inner = sorted(int(x) for x in rng.integers(0, n + 1, size=int(rng.integers(1, 6))))
            self.cuts = [0] + inner + [n]
        else:
            step = int(rng.choice([8192, 8192, 65536, 1000]))
            self.cuts = list(range(0, n, step)) + [n]
        # what else happens to a state that is simply finalized: nothing / read half-way and fed on ("the state stays
        # usable", include/tgx.h) / tgx_state_sync between batches / reset and fed the same table again, backwards
        self.seq = str(rng.choice(["plain", "resume", "sync", "reuse"], p=[0.4, 0.25, 0.15, 0.2])) \
            if self.after == "finalize" else "plain"
        if self.after == "ranks":
            self.world = int(rng.integers(2, 5))
            inner = sorted(int(x) // 64 * 64 for x in rng.integers(0, n + 1, size=self.world - 1))
            self.cuts = [0] + inner + [n]  # one shard per rank (validity bytes are shared: shards start on whole words)
        # ---- round 6, from a stream of its own: the DISTINCT checks ask for EXACT key sets (TGX_FLAG_EXACT_KEYS: string /
        # tuple keys kept with their bytes, equal fingerprints confirmed on them) -- nothing about the results may change,
        # whatever else happens to the state (lists, flushes, merges, blobs, ranks, resume / sync / reuse)
        rng7 = np.random.default_rng([seed, 7])
        self.exact_keys = bool(rng7.random() < 0.5)
        if self.exact_keys:
            for sp in self.specs:
                if sp.kind == T.DISTINCT:
                    sp.flags |= T.FLAG_EXACT_KEYS
        # ---- round 5, drawn from a stream of their own (the cases of earlier seeds stay what they were) ----
        rng5 = np.random.default_rng([seed, 5])
        # HOST batches handed over as TGX_MEM_HOST_RETAINED (kept until the flush): nothing about the results may change
        self.retain = bool(rng5.random() < 0.4)
        # an Int64 column presented as the narrowest Arrow type that holds its values (Int8 .. UInt32: widened on the
        # device; UInt64 / Boolean: COUNT and DISTINCT only) -- the oracle keeps seeing the Int64 values
        self.present = [None] * len(self.cols)
        for ci, (kind, vals, vb, mask, extra) in enumerate(self.cols):
            if kind != "i" or rng5.random() >= 0.3:
                continue
            lo, hi = (int(vals.min()), int(vals.max())) if n else (0, 0)
            keys_only = all(e[0] in ("count", "distinct", "tuple") for e in self.expect if ci in self.columns_of_expect(e))
            fits = [t for t in (np.int8, np.uint8, np.int16, np.uint16, np.uint32)
                    if np.iinfo(t).min <= lo and hi <= np.iinfo(t).max]
            if keys_only and lo >= 0 and hi <= 1 and rng5.random() < 0.5:
                self.present[ci] = "bool"
            elif keys_only and lo >= 0 and rng5.random() < 0.5:
                self.present[ci] = np.uint64
            elif fits:
                self.present[ci] = fits[int(rng5.integers(0, len(fits)))]

    @staticmethod
    def columns_of_expect(e):
        if e[0] == "tuple":
            return tuple(e[1])
        if e[0] in ("comoments", "spearman"):
            return (e[1], e[2])
        return (e[1],)

    def prefix_case(self, m):
        """the same case over the table's first m rows (what a state read half-way must report)"""
        import copy

        c = copy.copy(self)
        c.n = m
        c.cols = [(k, (v[:m + 1] if k == "s" else v[:m]), vb, mask[:m], extra) for k, v, vb, mask, extra in self.cols]
        return c

    def tuple_columns(self):
        s = self.specs[-1]
        return [s.columns[k] for k in range(s.n_columns)]

    def add(self, s, e):
        self.specs.append(s)
        self.expect.append(e)

    def describe(self):
        cols = ", ".join("%s%s/nulls=%d" % (c[0], "" if p is None else "as" + (p if isinstance(p, str) else p.__name__),
                                            int((~c[3]).sum())) for c, p in zip(self.cols, self.present))
        return "seed %d: n=%d cols=[%s] checks=%s batching=%s(%d) buffers=%s after=%s env=%s" % (
            self.seed, self.n, cols, [e[0] for e in self.expect], self.mode, len(self.cuts) - 1, self.device,
            self.after + ("" if self.seq == "plain" else "/" + self.seq) + (" kept" if self.retain else "") +
            (" exact-keys" if self.exact_keys else ""), self.env)

    # ---- the device side ----
    def columns_of(self, lo, hi, on_device):
        if not hasattr(self, "buffers"):
            self.buffers = [Buffers(kind, vals, vb, extra, mask, self.seed + k, self.present[k])
                            for k, (kind, vals, vb, mask, extra) in enumerate(self.cols)]
        cols = [b.column(on_device, lo, hi - lo) for b in self.buffers]
        if self.retain and not on_device:
            for c in cols:
                if c.c.mem == T.MEM_HOST:
                    c.c.mem = T.MEM_HOST_RETAINED
                if c.c.dictionary and c.c.dictionary.contents.mem == T.MEM_HOST:
                    c.c.dictionary.contents.mem = T.MEM_HOST_RETAINED
        return cols

    def run_device(self):
        import os

        for k, v in self.env.items():
            os.environ[k] = v
        try:
            return self.run_device_inner()
        finally:
            for k in self.env:
                os.environ.pop(k, None)

    def run_device_inner(self):
        T.init()
        plan = T.Plan(self.specs)
        if self.after == "ranks":
            from test_gpu_distributed_sim import _run_ranks

            on_device = self.device != "host"
            shards_of = lambda rank: self.columns_of(self.cuts[rank], self.cuts[rank + 1], on_device)  # noqa: E731
            out = _run_ranks(self.world, plan, shards_of)
            return [r for r, _st in out]  # every rank's view of the whole table
        n_states = int(self.rng.integers(2, 4)) if self.after == "merge" else 1
        states = [T.State(plan) for _ in range(n_states)]
        # a declared value range (tgx_distinct_range_hint: what ranks agree on before a sharded run) for some of the
        # single-column uniqueness checks over Int64 columns: the column's true MIN / MAX
        hints = []
        for k, e in enumerate(self.expect):
            if e[0] == "distinct" and self.cols[e[1]][0] == "i" and self.cols[e[1]][3].any() and self.rng.random() < 0.2:
                v = self.cols[e[1]][1][self.cols[e[1]][3]]
                if int(v.max()) - int(v.min()) < 2**33:
                    hints.append((k, int(v.min()), int(v.max())))
        for st_k in states:
            for k, lo, hi in hints:
                st_k.distinct_range_hint(k, lo, hi)
        keep = []  # device tensors stay alive until the states have been read
        n_batches = len(self.cuts) - 1
        order = list(range(n_batches))
        stop_at = int(self.rng.integers(0, n_batches + 1)) if self.seq == "resume" else -1
        # a blob taken half-way: the rest of the table is fed to the state rebuilt from it (what an incremental run does
        # with a stored partial state); Spearman states hold their pairs on the device and stay as they are
        blob_at = int(self.rng.integers(0, n_batches + 1)) if self.after == "blob" and not any(
            e[0] == "spearman" for e in self.expect) else -1
        if self.seq == "reuse":  # a first round in the other order, read and thrown away
            for b in reversed(order):
                on_device = self.device == "device" or (self.device == "mixed" and bool(self.rng.integers(0, 2)))
                cols = self.columns_of(self.cuts[b], self.cuts[b + 1], on_device)
                keep.append(cols)
                states[0].update(cols)
            self.check_one(states[0].finalize())
            states[0].reset()
            for k, lo, hi in hints:
                states[0].distinct_range_hint(k, lo, hi)
        for b in order:
            if b == stop_at:  # the table so far
                self.prefix_case(self.cuts[b]).check_one(states[0].finalize())
            if b == blob_at:
                states[0] = T.State.deserialize(plan, states[0].serialize())
            lo, hi = self.cuts[b], self.cuts[b + 1]
            on_device = self.device == "device" or (self.device == "mixed" and bool(self.rng.integers(0, 2)))
            cols = self.columns_of(lo, hi, on_device)
            keep.append(cols)
            states[b * n_states // max(1, n_batches)].update(cols)
            if self.seq == "sync" and self.rng.random() < 0.3:
                states[0].sync()
        st = states[0]
        if self.after == "merge":
            order = list(self.rng.permutation(n_states))
            st = T.State(plan)
            st.merge([states[k] for k in order])
        elif self.after == "blob":
            st = T.State.deserialize(plan, st.serialize())
        res = st.finalize()
        del keep
        return [res]

    # ---- the oracle side + comparison ----
    def key_bits(self, ci):
        """(values as the 64-bit patterns DISTINCT compares, validity) of a numeric column"""
        kind, vals, vb, _, _ = self.cols[ci]
        if kind == "i32":
            return vals.astype(np.int64).view(np.uint64), vb
        if kind == "f32":
            return vals.astype(np.float64).view(np.uint64), vb
        return vals.view(np.uint64), vb

    def exact_distinct(self, ci):
        kind, vals, vb, _, extra = self.cols[ci]
        if kind == "s":
            return orc.distinct_utf8(vals.astype(np.int32) if not extra[1] else self.offsets32(vals), extra[0], vb, n=self.n)
        bits, vb = self.key_bits(ci)
        return orc.distinct_bits64(bits, vb, n=self.n)

    @staticmethod
    def offsets32(offsets):
        assert offsets[-1] < 2**31
        return offsets.astype(np.int32)

    def check(self, all_res):
        for res in all_res:
            self.check_one(res)

    def check_one(self, res):
        n = self.n
        for r, e in zip(res, self.expect):
            what = e[0]
            if what == "tuple":
                self.check_tuple(r, e)
                continue
            kind, vals, vb, mask, extra = self.cols[e[1]]
            wide = vals if kind in ("i", "f", "s") else vals.astype(np.int64 if kind == "i32" else np.float64)
            if what == "count":
                c = orc.count(vb, n)
                assert (r.total, r.non_null) == (c.total, c.non_null), (e, r.total, r.non_null)
            elif what == "stats":
                exact_var = None
                if e[2] and int(mask.sum()) > 1 and np.isfinite(wide[mask].astype(np.float64)).all():
                    x = wide[mask].astype(np.longdouble)  # two passes in extended precision
                    exact_var = float(((x - x.mean()) ** 2).sum() / (len(x) - 1))
                self.check_stats(r, orc.stats(wide, vb, n=n), e[2], e, exact_var)
            elif what == "distinct":
                d = self.exact_distinct(e[1])
                assert (r.total, r.non_null, r.distinct) == (d.total, d.non_null, d.distinct), (e, r.distinct, d.distinct)
                if e[2]:
                    assert r.groups_once == d.groups_once, (e, r.groups_once, d.groups_once)
            elif what == "approx":
                c = orc.count(vb, n)
                assert (r.total, r.non_null) == (c.total, c.non_null), e
                if e[2]:
                    want = self.exact_distinct(e[1]).distinct
                else:
                    bits, _ = self.key_bits(e[1])
                    want = orc.hll_estimate(orc.hll_registers(bits.view(np.int64), vb, n=n))
                assert r.distinct == want, (e, r.distinct, want)
            elif what == "length":
                want = orc.length_count_utf8(self.offsets32(vals), extra[0], vb, n=n, min_chars=e[2], max_chars=e[3])
                assert (r.total, r.matches) == (n, want.matches), (e, r.matches, want.matches)
            elif what == "regex":
                rx = orc.Regex(e[2], case_insensitive=bool(e[3] & T.FLAG_CASE_INSENSITIVE))
                want = rx.count_utf8(self.offsets32(vals), extra[0], vb, n=n, trim=bool(e[3] & T.FLAG_TRIM),
                                     null_is_valid=bool(e[3] & T.FLAG_NULL_IS_VALID))
                assert (r.total, r.matches) == (n, want.matches), (e, r.matches, want.matches)
            elif what == "kll":
                x = vals[mask].astype(np.float64)
                assert r.kll_n == int((~np.isnan(x)).sum()), (e, r.kll_n)
            elif what == "comoments":
                _, y, yb, _, _ = self.cols[e[2]]
                o = orc.comoments(vals, y, vb, yb, n=n)
                assert int(r.non_null) == o.n, (e, r.non_null, o.n)
                with np.errstate(all="ignore"):
                    xa, ya = np.abs(np.asarray(vals, dtype=np.float64)), np.abs(np.asarray(y, dtype=np.float64))
                    overflowing = bool(np.isinf(xa).any() or np.isinf(ya).any() or np.isinf(xa.max(initial=0.0) * ya.max(initial=0.0))
                                       or np.isinf(xa.max(initial=0.0) ** 2) or np.isinf(ya.max(initial=0.0) ** 2))
                with np.errstate(all="ignore"):
                    mx, my = float(xa.max(initial=0.0)), float(ya.max(initial=0.0))
                    scales = (mx, my, mx * mx, my * my, mx * my)
                for (got, want), scale in zip(((r.sum_x, o.sum_x), (r.sum_y, o.sum_y), (r.sum_x2, o.sum_x2),
                                               (r.sum_y2, o.sum_y2), (r.sum_xy, o.sum_xy)), scales):
                    # what ANY sum of n doubles of that magnitude can be off by: terms next to DBL_MAX that cancel (seed
                    # 1244830: x = 1e308 twice, against y = -0.47 and 0.47 -- the oracle's extended precision cancels
                    # them exactly, sum_xy = 0.84; products about a pivot of 1e307 leave -35.5, an error of 1e-306 of
                    # the terms) are not a difference between two implementations in doubles
                    # (a product of magnitudes beyond DBL_MAX: the terms are as large as doubles get)
                    slack = 64.0 * max(1, o.n) * 2.0 ** -53 * (scale if math.isfinite(scale) else sys.float_info.max)
                    if math.isfinite(want) and math.isfinite(got) and abs(got - want) <= slack:
                        pass
                    elif math.isinf(got) and overflowing:
                        # (whatever the oracle's extended precision makes of it -- NaN, an infinity, or a finite sum of
                        #  raw products next to DBL_MAX, seed 2302961: -1e307 -- the products about a pivot, and the
                        #  re-basing of a merge, pass through infinity with the sign the pivot's side gives them)
                        pass
                    elif math.isnan(want) and math.isinf(got) and overflowing:
                        # an infinity (or a product beyond DBL_MAX) among the pairs: the raw products the oracle -- and
                        # DataFusion -- adds come out as +inf and -inf and cancel to NaN, the kernels' products about
                        # the pair's pivot overflow with other signs and stay infinite (seed 2001332)
                        pass
                    elif math.isinf(want) and math.isinf(got) and overflowing:
                        # the same with ONE such pair: the raw product is an infinity of the product's sign, the product
                        # about the pivot one of the other sign when the pivot lies on the other side of the small
                        # factor (seed 1006038: x = 2.5, y = 1e308, pivot of x above 2.5: -inf for +inf)
                        pass
                    elif math.isnan(want) or math.isinf(want):
                        assert math.isnan(got) or got == want, (e, got, want)
                    elif abs(want) > 1e300 and math.isinf(got) and (got > 0) == (want > 0):
                        # values next to DBL_MAX (the pool's 1e308, several times): the oracle adds in extended
                        # precision, a sum in doubles -- the kernels', DataFusion's -- passes through infinity for
                        # some orders of the same rows and stays there (seed 705910: -inf for -1.6e308)
                        pass
                    else:
                        assert rel_err(got, want) < TOL or abs(got - want) < 1e-6, (e, got, want)
            elif what == "spearman":
                _, y, yb, _, _ = self.cols[e[2]]
                o = orc.spearman_state(vals, y, vb, yb, n=n)
                got = (int(r.non_null), r.sum_x, r.sum_y, r.sum_x2, r.sum_y2, r.sum_xy)
                want = (int(o.n), o.sum_x, o.sum_y, o.sum_x2, o.sum_y2, o.sum_xy)
                assert got == want, (e, got, want)

    def check_tuple(self, r, e):
        """COUNT(DISTINCT (a, b, ...)): a tuple is a value of its own, NULL components included"""
        n = self.n
        parts = []
        for ci in e[1]:
            kind, vals, vb, mask, extra = self.cols[ci]
            if kind == "s":
                data = extra[0].tobytes()
                col = [data[vals[i]:vals[i + 1]] if mask[i] else None for i in range(n)]
            else:
                bits, _ = self.key_bits(ci)
                col = [int(bits[i]) if mask[i] else None for i in range(n)]
            parts.append(col)
        counts = {}
        for t in zip(*parts):
            counts[t] = counts.get(t, 0) + 1
        assert (r.total, r.distinct) == (n, len(counts)), (e, r.total, r.distinct, len(counts))
        if e[2]:
            once = sum(1 for v in counts.values() if v == 1)
            assert r.groups_once == once, (e, r.groups_once, once)

    @staticmethod
    def check_stats(r, st, variance, e, exact_var=None):
        assert (r.total, r.non_null, bool(r.has_value)) == (st.total, st.non_null, bool(st.has_value)), e
        if not st.has_value:
            return
        if st.is_float:
            assert orc.nan_equal(r.min_f, st.min_f) and orc.nan_equal(r.max_f, st.max_f), (e, r.min_f, st.min_f)
            if not math.isnan(st.min_f):
                assert math.copysign(1, r.min_f) == math.copysign(1, st.min_f), e
            if math.isnan(st.sum_hi) or math.isinf(st.sum_hi):
                assert orc.nan_equal(r.sum_f, st.sum_hi), (e, r.sum_f, st.sum_hi)
                return
            assert rel_err(r.sum_f, st.sum_hi) < TOL or abs(r.sum_f - st.sum_hi) < 1e-6 * st.non_null, (e, r.sum_f, st.sum_hi)
        else:
            assert (r.min_i, r.max_i, r.sum_i) == (st.min_i, st.max_i, st.sum_i_wrapping), (e, r.min_i, st.min_i)
        assert rel_err(r.mean, st.mean) < TOL or abs(r.mean - st.mean) < 1e-9, (e, r.mean, st.mean)
        if variance:
            assert bool(r.has_variance) == bool(st.has_variance), e
            if st.has_variance and not (math.isnan(st.var_samp) or math.isinf(st.var_samp)):
                # The oracle restates DataFusion's online update, which loses digits on offset data (values near 6e11
                # with a spread of 40: 143.50026 for a true 143.5); the device's pivot-shifted sums do not.  So: within
                # 1e-6 of the reference's value, give or take the reference's own distance from the exact one -- and
                # never further from the exact value than the reference is, beyond 1e-7 relative.
                # States and partitions are merged with Chan's update in doubles -- here as in DataFusion's
                # VarianceAccumulator::merge_batch -- whose `delta` of two means carries the means' rounding: an absolute
                # eps * |mean| * sqrt(var) or so, whatever the batching (8 rows near 1e12: 2e-5 on a variance of 6).
                slack = 0.0 if exact_var is None else 4 * abs(st.var_samp - exact_var)
                merge_noise = 32 * 2.2e-16 * abs(st.mean) * math.sqrt(abs(st.var_samp))
                assert abs(r.var_samp - st.var_samp) <= 1e-6 * abs(st.var_samp) + slack + merge_noise + 1e-12, (
                    e, r.var_samp, st.var_samp, exact_var)
                if exact_var is not None:
                    assert abs(r.var_samp - exact_var) <= abs(st.var_samp - exact_var) + 1e-7 * abs(exact_var) + \
                        merge_noise + 1e-12, (e, r.var_samp, st.var_samp, exact_var)


def run_seed(seed, max_rows=2_600_000, host_only=False):
    case = Case(seed, max_rows, host_only)
    try:
        case.check(case.run_device())
    except AssertionError as err:
        raise AssertionError("%s\n%s" % (case.describe(), err)) from None
    except T.TgxError as err:
        raise AssertionError("%s\n%s" % (case.describe(), err)) from None
    return case
