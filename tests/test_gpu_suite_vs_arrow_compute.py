"""-m gpu: ValidationSuite.run on the HIP path against Arrow C++ compute (pyarrow) on the same pyarrow table -- end to
end, past the oracle: every metric of a suite over a 300 000-row mixed table (NULLs, repeated keys, multi-byte text,
8192-row record batches AND one batch) is recomputed with pyarrow.compute / RE2 and compared.  The metric formulas are
the reference's (completeness nv / N, uniqueness D / N, distinctness D / N, unique value ratio
U / N, format M / N with NULL rows counted as matching, length likewise); the aggregates underneath come from an
implementation that shares no code with this repo (tests/test_oracle_independent.py holds the oracle against the same)."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from term_amd.suite import Assertion, Check, CompletenessOptions, Level, ValidationSuite

pytestmark = pytest.mark.gpu

N = 300_000


def table():
    rng = np.random.default_rng(20250104)
    ids = rng.permutation(N).astype(np.int64)
    qty = rng.integers(-50, 5000, size=N, dtype=np.int64)
    price = np.round(rng.gamma(2.0, 40.0, size=N), 2)
    price[rng.random(N) < 0.01] = 0.0
    words = ["alpha", "Beta", "gämma", "δelta", "user@example.com", "a@b.co", "日本語", "", "x" * 40, "bad@@mail"]
    text = [words[int(i)] + ("" if k % 4 else str(int(k))) for i, k in zip(rng.integers(0, len(words), size=N), rng.integers(0, 50, size=N))]
    codes = ["AA", "BB", "CC", "DD"]
    text_arr = pa.array(text, pa.string(), mask=rng.random(N) < 0.07)
    return pa.table({
        "id": pa.array(ids),
        "qty": pa.array(qty, mask=rng.random(N) < 0.05),
        "price": pa.array(price, mask=rng.random(N) < 0.10),
        "text": text_arr,
        "text_view": text_arr.cast(pa.string_view()),    # (what DataFusion reads Parquet strings as)
        "text_large": text_arr.cast(pa.large_string()),
        "code": pa.array([codes[int(i)] for i in rng.integers(0, 4, size=N)], pa.string()).dictionary_encode(),
        "grp": pa.array(rng.integers(0, 1000, size=N, dtype=np.int64), mask=rng.random(N) < 0.02),
        "sp": pa.array(rng.integers(-10**12, 10**12, size=N, dtype=np.int64) // 3_000_000 * 7, mask=rng.random(N) < 0.03),
    })


def suite():
    def one(name, f):
        return f(Check.builder(name).level(Level.WARNING)).build()

    any_ = Assertion.GreaterThanOrEqual(-1e300)
    b = ValidationSuite.builder("vs_arrow")
    for c in ("qty", "price", "text", "grp"):
        b = b.check(one("complete_" + c, lambda k, c=c: k.completeness(c, CompletenessOptions.threshold(0.5))))
    for c in ("qty", "price"):
        for stat in ("min", "max", "mean", "sum", "standard_deviation", "variance"):
            b = b.check(one("%s_%s" % (stat, c), lambda k, c=c, stat=stat: k.statistic(c, stat, any_)))
    b = b.check(one("uniq_id", lambda k: k.validates_uniqueness(["id"], 1.0)))
    b = b.check(one("uniq_grp", lambda k: k.validates_uniqueness(["grp"], 0.0)))
    b = b.check(one("dist_grp", lambda k: k.validates_distinctness(["grp"], any_)))
    b = b.check(one("uvr_qty", lambda k: k.validates_unique_value_ratio(["qty"], any_)))
    b = b.check(one("uvr_sp", lambda k: k.validates_unique_value_ratio(["sp"], any_)))
    b = b.check(one("dist_sp", lambda k: k.validates_distinctness(["sp"], any_)))
    b = b.check(one("dist_text", lambda k: k.validates_distinctness(["text"], any_)))
    b = b.check(one("dist_code", lambda k: k.validates_distinctness(["code"], any_)))
    b = b.check(one("at_sign", lambda k: k.validates_regex("text", r"^[^@]+@[^@]+\.[a-z]+$", 0.0)))
    b = b.check(one("upper_first", lambda k: k.validates_regex("text", r"^\p{Lu}", 0.0)))
    b = b.check(one("code_fmt", lambda k: k.validates_regex("code", r"^[A-C]{2}$", 0.0)))
    b = b.check(one("len_text", lambda k: k.has_length_between("text", 2, 12)))
    # tuples: a tuple is a value of its own, NULL components included (what GROUP BY makes of the same columns)
    b = b.check(one("uniq_pair", lambda k: k.validates_uniqueness(["grp", "text_large"], 0.0)))
    b = b.check(one("uniq_triple", lambda k: k.validates_uniqueness(["qty", "price", "text"], 0.0)))
    # (a dictionary column inside a tuple: the component is the row's dictionary ENTRY, as if the column were decoded)
    b = b.check(one("uniq_dict_pair", lambda k: k.validates_uniqueness(["grp", "code"], 0.0)))
    b = b.check(one("uniq_dict_triple", lambda k: k.validates_uniqueness(["code", "text_view", "qty"], 0.0)))
    for c in ("text_view", "text_large"):  # the same values in the other two string layouts: the same metrics
        b = b.check(one("dist_" + c, lambda k, c=c: k.validates_distinctness([c], any_)))
        b = b.check(one("at_sign_" + c, lambda k, c=c: k.validates_regex(c, r"^[^@]+@[^@]+\.[a-z]+$", 0.0)))
        b = b.check(one("upper_first_" + c, lambda k, c=c: k.validates_regex(c, r"^\p{Lu}", 0.0)))
        b = b.check(one("complete_" + c, lambda k, c=c: k.completeness(c, CompletenessOptions.threshold(0.5))))
    b = b.check(one("size", lambda k: k.has_size(Assertion.Equals(float(N)))))
    b = b.check(one("corr", lambda k: k.has_correlation("qty", "price", any_)))
    return b.build()


def expected(t):
    m = {}
    n = t.num_rows
    for c in ("qty", "price", "text", "grp"):
        m["complete_%s.completeness" % c] = (n - t[c].null_count) / n
    for c in ("qty", "price"):
        col = t[c]
        mm = pc.min_max(col).as_py()
        m["min_%s.min" % c] = float(mm["min"])
        m["max_%s.max" % c] = float(mm["max"])
        m["mean_%s.mean" % c] = pc.mean(col).as_py()
        m["sum_%s.sum" % c] = float(pc.sum(col).as_py())
        m["standard_deviation_%s.standard_deviation" % c] = pc.stddev(col, ddof=1).as_py()
        m["variance_%s.variance" % c] = pc.variance(col, ddof=1).as_py()
    d = lambda c: pc.count_distinct(t[c], mode="only_valid").as_py()
    m["uniq_id.full_uniqueness"] = d("id") / n
    m["uniq_grp.full_uniqueness"] = d("grp") / n
    m["dist_grp.distinctness"] = d("grp") / n
    counts = pc.value_counts(t["qty"].combine_chunks())
    m["uvr_qty.unique_value_ratio"] = sum(1 for c in counts.field("counts").to_pylist() if c == 1) / n  # (the NULL group counts as a group)
    counts = pc.value_counts(t["sp"].combine_chunks())
    m["uvr_sp.unique_value_ratio"] = sum(1 for c in counts.field("counts").to_pylist() if c == 1) / n
    m["dist_sp.distinctness"] = d("sp") / n
    m["dist_text.distinctness"] = d("text") / n
    m["dist_code.distinctness"] = pc.count_distinct(t["code"].cast(pa.string()), mode="only_valid").as_py() / n

    def fmt(c, pat):
        hits = pc.match_substring_regex(t[c].cast(pa.string()), pat)
        return (pc.sum(hits).as_py() + hits.null_count) / n  # (a NULL row counts as matching: null_is_valid)

    m["at_sign.regex"] = fmt("text", r"^[^@]+@[^@]+\.[a-z]+$")
    m["upper_first.regex"] = fmt("text", r"^\p{Lu}")
    m["code_fmt.regex"] = fmt("code", r"^[A-C]{2}$")
    lens = pc.utf8_length(t["text"])
    ok = pc.and_(pc.greater_equal(lens, 2), pc.less_equal(lens, 12))
    m["len_text.length_between"] = (pc.sum(ok).as_py() + ok.null_count) / n
    m["size.size"] = float(n)
    plain = t.set_column(t.schema.get_field_index("code"), "code", t["code"].cast(pa.string()))
    m["uniq_pair.full_uniqueness"] = plain.group_by(["grp", "text_large"]).aggregate([]).num_rows / n
    m["uniq_triple.full_uniqueness"] = plain.group_by(["qty", "price", "text"]).aggregate([]).num_rows / n
    m["uniq_dict_pair.full_uniqueness"] = plain.group_by(["grp", "code"]).aggregate([]).num_rows / n
    m["uniq_dict_triple.full_uniqueness"] = plain.group_by(["code", "text", "qty"]).aggregate([]).num_rows / n
    for c in ("text_view", "text_large"):
        m["dist_%s.distinctness" % c] = m["dist_text.distinctness"]
        m["at_sign_%s.regex" % c] = m["at_sign.regex"]
        m["upper_first_%s.regex" % c] = m["upper_first.regex"]
        m["complete_%s.completeness" % c] = m["complete_text.completeness"]
    both = pc.and_(pc.is_valid(t["qty"]), pc.is_valid(t["price"]))
    x = t["qty"].filter(both).to_numpy().astype(np.float64)
    y = t["price"].filter(both).to_numpy()
    m["corr.correlation"] = float(np.corrcoef(x, y)[0, 1])
    return m


@pytest.mark.parametrize("batch_rows", [None, 8192])
def test_suite_metrics_against_arrow_compute(batch_rows):
    t = table()
    fed = t if batch_rows is None else pa.Table.from_batches(t.to_batches(max_chunksize=batch_rows))
    out = suite().run(fed)
    got = out.report.metrics.custom_metrics
    want = expected(t)
    missing = [k for k in want if k not in got]
    assert not missing, (missing, sorted(got))
    for k, v in want.items():
        assert got[k] == pytest.approx(v, rel=1e-9, abs=1e-9 if k.startswith("corr") else 1e-12), k
    # (the LENGTH constraint wants every row inside the bounds: it is the one that fails, with the ratio above as its metric)
    assert out.report.metrics.failed_checks == 1 and [i.check_name for i in out.report.issues] == ["len_text"]
    assert out.report.metrics.total_checks == len(want)


def test_a_constraint_the_library_refuses_does_not_cost_the_others_their_verdicts():
    """All constraints of a suite share one pass; one the library refuses (here MIN of a Boolean column with the
    reference's result-type rule switched off: COUNT and DISTINCT checks only) keeps the refusal as ITS error -- as an
    Err from evaluate() reads in the reference, where every constraint is a query of its own (core/suite.rs:84-257) --
    and the rest are evaluated all the same."""
    n = 50_000
    t = pa.table({"id": pa.array(np.arange(n, dtype=np.int64)), "flag": pa.array(np.arange(n) % 3 == 0),
                  "v": pa.array(np.arange(n, dtype=np.float64) % 100)})
    check = (Check.builder("c").level(Level.ERROR).validates_uniqueness(["id"], 1.0).has_min("flag", Assertion.Equals(0.0))
             .has_mean("v", Assertion.Equals(49.5)).completeness("flag").build())
    out = ValidationSuite.builder("s").strict_reference_types(False).check(check).build().run(t)
    m = out.report.metrics
    assert (m.total_checks, m.passed_checks, m.failed_checks) == (4, 3, 1)
    (issue,) = out.report.issues
    assert issue.constraint_name == "min" and issue.message.startswith("Error evaluating constraint:") and "TGX_UNSUPPORTED" in issue.message
    assert m.custom_metrics["c.full_uniqueness"] == 1.0 and m.custom_metrics["c.mean"] == 49.5


def test_tuples_over_boolean_narrow_and_unsigned_columns_against_group_by():
    n = 100_000
    rng = np.random.default_rng(5)
    t = pa.table({"g": pa.array(rng.integers(0, 50, n)), "flag": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1),
                  "u8": pa.array(rng.integers(0, 7, n).astype(np.uint8)), "u64": pa.array(rng.integers(0, 5, n).astype(np.uint64))})
    c = Check.builder("c").level(Level.ERROR).validates_uniqueness(["g", "flag"], 0.0).build()
    d = Check.builder("d").level(Level.ERROR).validates_uniqueness(["u8", "u64", "flag"], 0.0).build()
    out = ValidationSuite.builder("s").check(c).check(d).build().run(pa.Table.from_batches(t.to_batches(max_chunksize=8192)))
    m = out.report.metrics.custom_metrics
    assert not out.report.issues
    assert m["c.full_uniqueness"] == t.group_by(["g", "flag"]).aggregate([]).num_rows / n
    assert m["d.full_uniqueness"] == t.group_by(["u8", "u64", "flag"]).aggregate([]).num_rows / n
