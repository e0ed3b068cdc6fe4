"""Python face of the host-side mirror of term-guard's builder API (term_amd/csrc/host/term_guard.h).

Same names and argument meaning as the reference (core/check.rs, core/suite.rs, constraints/assertion.rs,
core/builder_extensions.rs); a suite is serialised to the JSON of include/tgx_host.h and run by libtgx:

    suite = (ValidationSuite.builder("users").table_name("users")
             .check(Check.builder("required").level(Level.ERROR)
                    .completeness("user_id", CompletenessOptions.full())
                    .has_min("age", Assertion.GreaterThanOrEqual(0.0)).build())
             .build())
    result = suite.run({"user_id": col, "age": col})     # dict of term_amd.Column, or a pyarrow Table
    result.is_success(), result.report.issues, result.to_json()
"""
import ctypes as C
import json

from . import _lib
from ._lib import MEM_HOST, MEM_HOST_RETAINED, Column, TgxError, _Column, _Error


class Level:
    INFO, WARNING, ERROR = "info", "warning", "error"
    Info, Warning, Error = INFO, WARNING, ERROR


class Assertion:
    """constraints/assertion.rs:27-46"""

    def __init__(self, kind, *args):
        self.kind, self.args = kind, [float(a) for a in args]

    @staticmethod
    def Equals(v): return Assertion("equals", v)
    @staticmethod
    def NotEquals(v): return Assertion("not_equals", v)
    @staticmethod
    def GreaterThan(v): return Assertion("greater_than", v)
    @staticmethod
    def GreaterThanOrEqual(v): return Assertion("greater_than_or_equal", v)
    @staticmethod
    def LessThan(v): return Assertion("less_than", v)
    @staticmethod
    def LessThanOrEqual(v): return Assertion("less_than_or_equal", v)
    @staticmethod
    def Between(lo, hi): return Assertion("between", lo, hi)
    @staticmethod
    def NotBetween(lo, hi): return Assertion("not_between", lo, hi)

    def to_json(self):
        return {"kind": self.kind, "args": self.args}

    def evaluate(self, value):
        holds = C.c_int32()
        err = _Error()
        _host_check(_host().tgx_host_assertion_json(json.dumps(self.to_json()).encode(), float(value), C.byref(holds),
                                                    None, C.byref(err)), err)
        return bool(holds.value)

    def description(self):
        out = C.c_char_p()
        err = _Error()
        _host_check(_host().tgx_host_assertion_json(json.dumps(self.to_json()).encode(), 0.0, None, C.byref(out),
                                                    C.byref(err)), err)
        return _take(out)

    __str__ = description


class LogicalOperator:
    """core/logical.rs:32-45"""
    All, Any = "all", "any"

    @staticmethod
    def Exactly(n): return {"exactly": int(n)}
    @staticmethod
    def AtLeast(n): return {"at_least": int(n)}
    @staticmethod
    def AtMost(n): return {"at_most": int(n)}


class CompletenessOptions:
    """core/builder_extensions.rs:14-80"""

    def __init__(self, threshold=1.0, operator=LogicalOperator.All):
        self.threshold_, self.operator_ = threshold, operator

    @staticmethod
    def full(): return CompletenessOptions(1.0)
    @staticmethod
    def threshold(t): return CompletenessOptions(t)
    @staticmethod
    def at_least(n): return CompletenessOptions(1.0, LogicalOperator.AtLeast(n))
    @staticmethod
    def any(): return CompletenessOptions(1.0, LogicalOperator.Any)

    def with_operator(self, op):
        self.operator_ = op
        return self


class ConstraintOptions(CompletenessOptions):
    """core/unified.rs:131-191"""

    @staticmethod
    def new(): return ConstraintOptions()

    def with_threshold(self, t):
        self.threshold_ = t
        return self


class FormatOptions:
    """constraints/format.rs:367-470"""

    def __init__(self, case_sensitive=True, trim_before_check=False, null_is_valid=True):
        self.case_sensitive_, self.trim_, self.null_is_valid_ = case_sensitive, trim_before_check, null_is_valid

    @staticmethod
    def new(): return FormatOptions()
    @staticmethod
    def case_insensitive(): return FormatOptions(case_sensitive=False)
    @staticmethod
    def strict(): return FormatOptions(null_is_valid=False)
    @staticmethod
    def lenient(): return FormatOptions(False, True, True)
    @staticmethod
    def with_trimming(): return FormatOptions(trim_before_check=True)

    def case_sensitive(self, v):
        self.case_sensitive_ = v
        return self

    def trim_before_check(self, v):
        self.trim_ = v
        return self

    def null_is_valid(self, v):
        self.null_is_valid_ = v
        return self

    def to_json(self):
        return {"case_sensitive": self.case_sensitive_, "trim_before_check": self.trim_,
                "null_is_valid": self.null_is_valid_}


class NullHandling:
    Exclude, Include, Distinct = "exclude", "include", "distinct"


def _cols(columns):
    return [columns] if isinstance(columns, str) else list(columns)


class CheckBuilder:
    """core/check.rs:217-2310 (the hot-path subset listed in SURVEY.md section 8a)"""

    def __init__(self, name):
        self._c = {"name": name, "level": Level.WARNING, "constraints": []}

    def level(self, level):
        self._c["level"] = level
        return self

    def description(self, d):
        self._c["description"] = d
        return self

    def _add(self, **kw):
        self._c["constraints"].append(kw)
        return self

    def has_size(self, assertion):
        return self._add(type="size", assertion=assertion.to_json())

    def has_approx_count_distinct(self, column, assertion):
        """core/check.rs:379-390; the metric is the exact number of distinct non-NULL values"""
        return self._add(type="approx_count_distinct", column=column, assertion=assertion.to_json())

    def completeness(self, columns, options=None):
        o = options or CompletenessOptions.full()
        return self._add(type="completeness", columns=_cols(columns), operator=o.operator_, threshold=o.threshold_)

    def any_complete(self, columns):
        return self._add(type="completeness", columns=_cols(columns), operator="any", threshold=1.0)

    def at_least_complete(self, n, columns, threshold):
        return self._add(type="completeness", columns=_cols(columns), operator={"at_least": n}, threshold=threshold)

    def exactly_complete(self, n, columns, threshold):
        return self._add(type="completeness", columns=_cols(columns), operator={"exactly": n}, threshold=threshold)

    def statistic(self, column, statistic, assertion, p=0.5):
        return self._add(type="statistic", column=column, statistic=statistic, p=p, assertion=assertion.to_json())

    def has_min(self, column, assertion): return self.statistic(column, "min", assertion)
    def has_max(self, column, assertion): return self.statistic(column, "max", assertion)
    def has_mean(self, column, assertion): return self.statistic(column, "mean", assertion)
    def has_sum(self, column, assertion): return self.statistic(column, "sum", assertion)
    def has_standard_deviation(self, column, assertion): return self.statistic(column, "standard_deviation", assertion)
    def has_variance(self, column, assertion): return self.statistic(column, "variance", assertion)

    def uniqueness(self, columns, kind, threshold=1.0, assertion=None, null_handling=NullHandling.Exclude):
        kw = dict(type="uniqueness", columns=_cols(columns), kind=kind, threshold=threshold, null_handling=null_handling)
        if assertion is not None:
            kw["assertion"] = assertion.to_json()
        return self._add(**kw)

    def validates_uniqueness(self, columns, threshold): return self.uniqueness(columns, "full_uniqueness", threshold)
    def validates_distinctness(self, columns, assertion): return self.uniqueness(columns, "distinctness", assertion=assertion)
    def validates_unique_value_ratio(self, columns, assertion):
        return self.uniqueness(columns, "unique_value_ratio", assertion=assertion)
    def validates_primary_key(self, columns): return self.uniqueness(columns, "primary_key")
    def validates_uniqueness_with_nulls(self, columns, threshold, null_handling):
        return self.uniqueness(columns, "unique_with_nulls", threshold, null_handling=null_handling)

    def primary_key(self, columns):
        return self.completeness(columns, CompletenessOptions.full()).validates_uniqueness(columns, 1.0)

    def is_contained_in(self, column, allowed_values):
        """constraints/values.rs:200-330: every non-NULL value is one of `allowed_values`"""
        return self._add(type="containment", column=column, allowed_values=list(allowed_values))

    # check.rs:518-623, 1777-1785 (constraints/length.rs)
    def length(self, column, kind, a=0, b=0):
        """kind: min | max | between | exactly | not_empty"""
        return self._add(type="length", column=column, kind=kind, a=a, b=b)

    def has_min_length(self, column, n): return self.length(column, "min", n)
    def has_max_length(self, column, n): return self.length(column, "max", n)
    def has_length_between(self, column, lo, hi): return self.length(column, "between", lo, hi)
    def has_exact_length(self, column, n): return self.length(column, "exactly", n)
    def is_not_empty(self, column): return self.length(column, "not_empty")

    def has_format(self, column, fmt, threshold, options=None, **kw):
        return self._add(type="format", column=column, format=fmt, threshold=threshold,
                         options=(options or FormatOptions()).to_json(), **kw)

    def validates_regex(self, column, pattern, threshold): return self.has_format(column, "regex", threshold, pattern=pattern)
    def validates_regex_with_options(self, column, pattern, threshold, options):
        return self.has_format(column, "regex", threshold, options, pattern=pattern)
    def validates_email(self, column, threshold): return self.has_format(column, "email", threshold)
    def validates_url(self, column, threshold, allow_localhost=False):
        return self.has_format(column, "url", threshold, allow_localhost=allow_localhost)
    def validates_credit_card(self, column, threshold, detect_only=False):
        return self.has_format(column, "credit_card", threshold, detect_only=detect_only)
    def validates_phone(self, column, threshold, country=None):
        kw = {"country": country} if country else {}
        return self.has_format(column, "phone", threshold, FormatOptions.with_trimming(), **kw)
    def validates_postal_code(self, column, threshold, country):
        return self.has_format(column, "postal_code", threshold, FormatOptions.with_trimming(), country=country)
    def validates_uuid(self, column, threshold): return self.has_format(column, "uuid", threshold)
    def validates_ipv4(self, column, threshold): return self.has_format(column, "ipv4", threshold)
    def validates_ipv6(self, column, threshold): return self.has_format(column, "ipv6", threshold)
    def validates_json(self, column, threshold): return self.has_format(column, "json", threshold)
    def validates_iso8601_datetime(self, column, threshold): return self.has_format(column, "iso8601_datetime", threshold)
    def email(self, column, threshold): return self.has_format(column, "email", threshold, FormatOptions(True, True, False))
    def contains_ssn(self, column, threshold):
        return self.has_format(column, "social_security_number", threshold, FormatOptions.with_trimming())

    def has_approx_quantile(self, column, quantile, assertion):
        return self._add(type="quantile", column=column, quantile=quantile, assertion=assertion.to_json())

    def has_correlation(self, column1, column2, assertion):
        return self._add(type="correlation", column1=column1, column2=column2, assertion=assertion.to_json())

    def constraint(self, c):
        """core/check.rs:263: a constraint object (MultiStatisticalConstraint, QuantileConstraint,
        CorrelationConstraint below) instead of a builder shorthand"""
        return self._add(**c.spec)

    def build(self):
        return Check(self._c)


class StatisticType:
    """constraints/statistics.rs:24-43"""
    Min, Max, Mean, Sum = "min", "max", "mean", "sum"
    StandardDeviation, Variance, Median = "standard_deviation", "variance", "median"

    @staticmethod
    def Percentile(p): return ("percentile", p)


class MultiStatisticalConstraint:
    """constraints/statistics.rs:376-417: `statistics` = [(StatisticType, Assertion), ...] of one column"""

    def __init__(self, column, statistics):
        stats = []
        for st, a in statistics:
            name, p = st if isinstance(st, tuple) else (st, 0.5)
            stats.append({"statistic": name, "p": p, "assertion": a.to_json()})
        self.spec = {"type": "multi_statistic", "column": column, "statistics": stats}


class QuantileCheck:
    """constraints/quantile.rs:36-58"""

    def __init__(self, quantile, assertion):
        self.quantile, self.assertion = quantile, assertion

    def to_json(self):
        return {"quantile": self.quantile, "assertion": self.assertion.to_json()}


class QuantileConstraint:
    """constraints/quantile.rs:144-225 (QuantileValidation::Single / Multiple / Monotonic / Distribution / Custom)"""

    def __init__(self, column, **spec):
        self.spec = dict(type="quantile", column=column, **spec)

    @staticmethod
    def median(column, assertion):
        return QuantileConstraint.percentile(column, 0.5, assertion)

    @staticmethod
    def percentile(column, quantile, assertion):
        return QuantileConstraint(column, validation="single", quantile=quantile, assertion=assertion.to_json())

    @staticmethod
    def multiple(column, checks):
        return QuantileConstraint(column, validation="multiple", checks=[c.to_json() for c in checks])

    @staticmethod
    def monotonic(column, quantiles, strict):
        return QuantileConstraint(column, validation="monotonic", quantiles=list(quantiles), strict=bool(strict))

    @staticmethod
    def distribution(column):
        return QuantileConstraint(column, validation="distribution")


class CorrelationType:
    """constraints/correlation.rs:19-36"""
    Pearson, Spearman, KendallTau = "pearson", "spearman", "kendall_tau"
    MutualInformation, Covariance, Custom = "mutual_information", "covariance", "custom"


class CorrelationConstraint:
    """constraints/correlation.rs:147-263 (CorrelationValidation::Pairwise / Range / Independence / MultiColumn /
    Stability)"""

    def __init__(self, **spec):
        self.spec = dict(type="correlation", **spec)

    @staticmethod
    def pairwise(column1, column2, correlation_type, assertion, sql_expression=""):
        return CorrelationConstraint(validation="pairwise", column1=column1, column2=column2,
                                     correlation_type=correlation_type, assertion=assertion.to_json(),
                                     sql_expression=sql_expression)

    @staticmethod
    def pearson(column1, column2, assertion):
        return CorrelationConstraint.pairwise(column1, column2, CorrelationType.Pearson, assertion)

    @staticmethod
    def spearman(column1, column2, assertion):
        return CorrelationConstraint.pairwise(column1, column2, CorrelationType.Spearman, assertion)

    @staticmethod
    def covariance(column1, column2, assertion):
        return CorrelationConstraint.pairwise(column1, column2, CorrelationType.Covariance, assertion)

    @staticmethod
    def range(column1, column2, correlation_type, min, max):
        return CorrelationConstraint(validation="range", column1=column1, column2=column2,
                                     correlation_type=correlation_type, min=min, max=max)

    @staticmethod
    def independence(column1, column2, max_correlation):
        return CorrelationConstraint(validation="independence", column1=column1, column2=column2,
                                     max_correlation=max_correlation)

    @staticmethod
    def multi_column(columns, correlation_type=CorrelationType.Pearson):
        return CorrelationConstraint(validation="multi_column", columns=list(columns), correlation_type=correlation_type)


class Check:
    def __init__(self, spec):
        self.spec = spec

    @staticmethod
    def builder(name):
        return CheckBuilder(name)

    def name(self): return self.spec["name"]
    def level(self): return self.spec["level"]


class Issue:
    def __init__(self, d):
        self.check_name, self.constraint_name = d["check_name"], d["constraint_name"]
        self.level, self.message, self.metric = d["level"], d["message"], d.get("metric")


class Metrics:
    def __init__(self, d):
        self.total_checks, self.passed_checks = d["total_checks"], d["passed_checks"]
        self.failed_checks, self.skipped_checks = d["failed_checks"], d["skipped_checks"]
        self.execution_time_ms, self.custom_metrics = d["execution_time_ms"], d.get("custom_metrics", {})


class Report:
    def __init__(self, d):
        self.suite_name, self.timestamp = d["suite_name"], d["timestamp"]
        self.metrics = Metrics(d["metrics"])
        self.issues = [Issue(i) for i in d["issues"]]

    def has_errors(self): return any(i.level == Level.ERROR for i in self.issues)
    def has_warnings(self): return any(i.level == Level.WARNING for i in self.issues)


class ValidationResult:
    """core/result.rs:123-196"""

    def __init__(self, text):
        self._json = text
        d = json.loads(text)
        self.status = d["status"]
        self.report = Report(d["report"])

    def is_success(self): return self.status == "success"
    def is_failure(self): return self.status == "failure"
    def metrics(self): return self.report.metrics if self.is_success() else None
    def to_json(self): return self._json


class ValidationSuiteBuilder:
    def __init__(self, name):
        self._s = {"name": name, "table_name": "data", "checks": []}

    def description(self, d):
        self._s["description"] = d
        return self

    def table_name(self, t):
        self._s["table_name"] = t
        return self

    def check(self, c):
        self._s["checks"].append(c.spec)
        return self

    def with_optimizer(self, _enabled):
        return self

    def strict_reference_types(self, on):
        """True (default): statistics the reference cannot read off a column's type are errors, as there
        (constraints/statistics.rs:277-308: the aggregate must come back Int64 or Float64 -- MIN / MAX of an Int32,
        Date32, Float32 or Timestamp column does not); False: the widened column's value answers (a deviation)"""
        self._s["strict_reference_types"] = bool(on)
        return self

    def exact_string_keys(self, on):
        """True (default): uniqueness checks over string / binary / tuple keys count by VALUE (equal fingerprints are
        confirmed byte by byte, TGX_FLAG_EXACT_KEYS) -- COUNT(DISTINCT c) as the reference computes it
        (constraints/uniqueness.rs:612-617); False: by keyed 128-bit fingerprint alone (deviations)"""
        self._s["exact_string_keys"] = bool(on)
        return self

    def column_type(self, column, arrow_type):
        """the Arrow DataType of a column of the table ("Int32", "Date32", "Timestamp(Nanosecond, None)", ...); a
        pyarrow table handed to run() declares its own"""
        self._s.setdefault("column_types", {})[column] = arrow_type
        return self

    def build(self):
        return ValidationSuite(self._s)


def _arrow_batches(table):
    """pyarrow Table / RecordBatch -> (names, [[Column, ...] per batch])"""
    import pyarrow as pa

    if isinstance(table, pa.RecordBatch):
        table = pa.Table.from_batches([table])
    names = table.column_names
    batches = []
    def view(arr):
        try:
            return Column.from_arrow(arr)
        except TgxError:
            # a type outside the path (lists, structs, ...): its validity bitmap and length travel all the same --
            # completeness / size checks need nothing else -- and a check that reads its values fails with the library's
            # own error ("values is NULL")
            return Column.validity_only(arr)

    for rb in table.to_batches():
        batches.append([view(rb.column(i)) for i in range(rb.num_columns)])
    if not batches:
        batches = []
    return names, batches


class ValidationSuite:
    def __init__(self, spec):
        self.spec = spec

    @staticmethod
    def builder(name):
        return ValidationSuiteBuilder(name)

    def name(self): return self.spec["name"]

    def run(self, table):
        """table: dict name -> term_amd.Column (one batch), a list of such dicts (batches), a pyarrow Table /
        RecordBatch, or None (no table registered)."""
        name_arr, n_cols, col_arr, n_batches, _keep = _flatten_table(table)
        out = C.c_char_p()
        err = _Error()
        spec = self.spec
        declared = _arrow_type_names(table)
        if declared:  # (what the builder declared wins)
            spec = dict(spec, column_types=dict(declared, **spec.get("column_types", {})))
        _host_check(_host().tgx_host_run_suite_json(json.dumps(spec).encode(), name_arr, n_cols, col_arr,
                                                    n_batches, C.byref(out), C.byref(err)), err)
        return ValidationResult(_take(out))


def _arrow_type_names(table):
    """{column: DataType in arrow-rs's Debug spelling} of a pyarrow table (the reference's result-type rule needs the
    type the caller holds, not the 4- / 8-byte layout it is handed over as); {} for anything else"""
    try:
        import pyarrow as pa
    except ImportError:
        return {}
    if not isinstance(table, (pa.Table, pa.RecordBatch)):
        return {}
    simple = {pa.int8(): "Int8", pa.int16(): "Int16", pa.int32(): "Int32", pa.int64(): "Int64", pa.uint8(): "UInt8",
              pa.uint16(): "UInt16", pa.uint32(): "UInt32", pa.uint64(): "UInt64", pa.float16(): "Float16",
              pa.float32(): "Float32", pa.float64(): "Float64", pa.date32(): "Date32", pa.date64(): "Date64",
              pa.bool_(): "Boolean"}
    out = {}
    for f in table.schema:
        t = f.type
        if t in simple:
            out[f.name] = simple[t]
        elif pa.types.is_timestamp(t):
            unit = {"s": "Second", "ms": "Millisecond", "us": "Microsecond", "ns": "Nanosecond"}[t.unit]
            out[f.name] = "Timestamp(%s, %s)" % (unit, "None" if t.tz is None else 'Some("%s")' % t.tz)
        elif pa.types.is_time(t) or pa.types.is_duration(t) or pa.types.is_decimal(t):
            out[f.name] = str(t)
    return out


def _arrow_flat(table):
    """pyarrow Table / RecordBatch -> the arguments of the host calls, the column structs filled in place: a table that
    arrives as thousands of 8192-row record batches costs a few microseconds per (batch, column) here instead of the
    11 us a Column object takes (16 M rows x 3 columns as 8192-row batches: 67 ms of Python before the library saw a
    row).  The first batch's columns go through Column.from_arrow -- which knows every layout -- and say what the
    column's chunks are; chunks of fixed-width and string / binary columns are then described from their buffers'
    addresses alone, every other layout keeps taking the long way."""
    import pyarrow as pa

    if isinstance(table, pa.RecordBatch):
        table = pa.Table.from_batches([table])
    names = table.column_names
    batches = table.to_batches()
    n_cols, n_batches = len(names), len(batches)
    col_arr = (_Column * max(1, n_cols * n_batches))()
    keep = [table, batches]
    if n_batches == 0:
        return (C.c_char_p * max(1, n_cols))(*[n.encode() for n in names]), n_cols, col_arr, 0, keep

    def long_way(arr):
        try:
            return Column.from_arrow(arr)
        except TgxError:
            # a type outside the path (lists, structs, ...): its validity bitmap and length travel all the same --
            # completeness / size checks need nothing else -- and a check that reads its values fails with the library's
            # own error ("values is NULL")
            return Column.validity_only(arr)

    lean = []  # per column: None (the long way), ("fixed", type id) or ("string", type id)
    for ci, field in enumerate(table.schema):
        t = field.type
        first = long_way(batches[0].column(ci))
        if pa.types.is_primitive(t) and not pa.types.is_boolean(t) and first.c.values:
            lean.append(("fixed", first.c.type))
        elif (pa.types.is_string(t) or pa.types.is_large_string(t) or pa.types.is_binary(t) or pa.types.is_large_binary(t)) \
                and first.c.type in (_lib.UTF8, _lib.LARGE_UTF8):
            lean.append(("string", first.c.type))
        else:
            lean.append(None)
    import struct

    fmt = struct.Struct("<iiqqqQQQQQQQii")  # tgx_column, field by field (its size is checked below)
    assert fmt.size == C.sizeof(_Column)
    pack, size = fmt.pack_into, fmt.size
    # the columns' chunks are the record batches' columns when every column is cut alike (a table made of record
    # batches is): reading them column by column spares a RecordBatch.column() call per (batch, column), 1 us each
    chunked = [table.column(ci).chunks for ci in range(n_cols)]
    if not all(len(ch) == n_batches and all(len(a) == len(rb) for a, rb in zip(ch, batches)) for ch in chunked):
        chunked = [[rb.column(ci) for rb in batches] for ci in range(n_cols)]
    slow = []  # the slots filled the long way: marked as retained afterwards (the packed ones already are)
    for ci in range(n_cols):
        how = lean[ci]
        at = ci
        for arr in chunked[ci]:
            slot_at = at
            at += n_cols
            bufs = None if how is None else arr.buffers()
            if how is None or bufs[1] is None:  # (an array without rows may come without buffers)
                if how is None or len(arr):
                    col = long_way(arr)
                    keep.append(col)
                    C.memmove(C.byref(col_arr[slot_at]), C.byref(col.c), C.sizeof(_Column))
                else:
                    col_arr[slot_at].type = how[1]
                slow.append(slot_at)
                continue
            nulls = arr.null_count
            validity = bufs[0].address if (nulls and bufs[0] is not None) else 0
            if how[0] == "fixed":
                pack(col_arr, slot_at * size, how[1], MEM_HOST_RETAINED, len(arr), arr.offset, nulls, validity, bufs[1].address, 0, 0, 0, 0, 0, 0, 0)
            else:
                pack(col_arr, slot_at * size, how[1], MEM_HOST_RETAINED, len(arr), arr.offset, nulls, validity, 0, bufs[1].address,
                     bufs[2].address if bufs[2] is not None else 0, 0, 0, 0, 0, 0)
    name_arr = (C.c_char_p * max(1, n_cols))(*[n.encode() for n in names])
    keep.append(_mark_retained(col_arr, slow))
    return name_arr, n_cols, col_arr, n_batches, keep


def _mark_retained(col_arr, which):
    """The host calls run the whole table and return: every buffer handed over is alive and unmodified until then, which
    is all TGX_MEM_HOST_RETAINED asks for -- a table that arrives as 8192-row record batches is then copied by the
    library's copy threads beside the noting of the next batches instead of inside every tgx_update.  (The structs in
    `col_arr` are copies: the caller's Column objects are not touched; a dictionary's struct is copied too.)"""
    dict_copies = []
    for i in (range(which) if isinstance(which, int) else which):
        if col_arr[i].mem != MEM_HOST:
            continue
        if col_arr[i].dictionary:
            if col_arr[i].dictionary.contents.mem != MEM_HOST:
                continue
            d = _Column()
            C.memmove(C.byref(d), col_arr[i].dictionary, C.sizeof(_Column))
            d.mem = MEM_HOST_RETAINED
            dict_copies.append(d)
            col_arr[i].dictionary = C.pointer(d)
        col_arr[i].mem = MEM_HOST_RETAINED
    return dict_copies


def _flatten_table(table):
    if table is None:
        names, batches = [], []
    elif isinstance(table, dict):
        names, batches = list(table.keys()), [list(table.values())]
    elif isinstance(table, (list, tuple)):
        names = list(table[0].keys()) if table else []
        batches = [[b[n] for n in names] for b in table]
    else:
        return _arrow_flat(table)
    n_cols, n_batches = len(names), len(batches)
    name_arr = (C.c_char_p * max(1, n_cols))(*[n.encode() for n in names])
    flat = [c.c for b in batches for c in b]
    col_arr = (_Column * max(1, len(flat)))(*flat)
    return name_arr, n_cols, col_arr, n_batches, (batches, _mark_retained(col_arr, len(flat)))


# ---------------------------------------------------------------------------------------------- analyzers
class _Analyzer:
    """mirror of TG/analyzers/traits.rs Analyzer: name / metric_key / merge_states / compute_metric_from_state"""

    def __init__(self, spec):
        self.spec = spec

    def name(self):
        return self.spec["type"]

    def metric_key(self):
        t = self.spec["type"]
        if t in ("size", "standard_deviation"):  # standard_deviation.rs does not override metric_key
            return t
        if t == "correlation":
            return "correlation_%s_%s_%s" % (self.spec.get("method", "pearson"), self.spec["column1"], self.spec["column2"])
        return "%s.%s" % (t, self.spec["column"])

    def merge_states(self, states):
        out = C.c_char_p()
        err = _Error()
        _host_check(_host().tgx_host_merge_states_json(json.dumps(self.spec).encode(), json.dumps(states).encode(),
                                                       C.byref(out), C.byref(err)), err)
        return json.loads(_take(out))

    def merge_states_text(self, states):
        """merge_states, but the bytes the library wrote (what serde_json would be handed)"""
        out = C.c_char_p()
        err = _Error()
        _host_check(_host().tgx_host_merge_states_json(json.dumps(self.spec).encode(), json.dumps(states).encode(),
                                                       C.byref(out), C.byref(err)), err)
        return _take(out)

    def compute_metric_from_state(self, state):
        """-> MetricValue as {"type": "Double"|"Long"|"Map", "value": ...}; raises TgxError with the
        reference's AnalyzerError text (e.g. 'No data available for analysis')"""
        out = C.c_char_p()
        err = _Error()
        _host_check(_host().tgx_host_metric_from_state_json(json.dumps(self.spec).encode(), json.dumps(state).encode(),
                                                            C.byref(out), C.byref(err)), err)
        return json.loads(_take(out))


def SizeAnalyzer(): return _Analyzer({"type": "size"})
def CompletenessAnalyzer(column): return _Analyzer({"type": "completeness", "column": column})
def DistinctnessAnalyzer(column): return _Analyzer({"type": "distinctness", "column": column})
def ApproxCountDistinctAnalyzer(column): return _Analyzer({"type": "approx_count_distinct", "column": column})
def MeanAnalyzer(column): return _Analyzer({"type": "mean", "column": column})
def MinAnalyzer(column): return _Analyzer({"type": "min", "column": column})
def MaxAnalyzer(column): return _Analyzer({"type": "max", "column": column})
def SumAnalyzer(column): return _Analyzer({"type": "sum", "column": column})
def StandardDeviationAnalyzer(column): return _Analyzer({"type": "standard_deviation", "column": column})


def CorrelationAnalyzer(column1, column2, method="pearson"):
    return _Analyzer({"type": "correlation", "column1": column1, "column2": column2, "method": method})


class AnalyzerContext:
    """TG/analyzers/context.rs: metrics by key, errors; plus the analyzers' states (for merges across shards)"""

    def __init__(self, text):
        d = json.loads(text)
        self.text = text  # as written by the library (integer / float token kinds matter to serde_json readers)
        self.metrics, self.states, self._errors = d["metrics"], d["states"], d["errors"]

    def get_metric(self, key):
        return self.metrics.get(key)

    def has_errors(self):
        return bool(self._errors)

    def errors(self):
        return self._errors


class AnalysisRunner:
    """TG/analyzers/runner.rs:64-202 -- but ONE pass over the table for all analyzers"""

    def __init__(self):
        self._analyzers, self._continue, self._table = [], True, "data"

    def add(self, analyzer):
        self._analyzers.append(analyzer)
        return self

    def continue_on_error(self, flag):
        self._continue = bool(flag)
        return self

    def table_name(self, name):
        self._table = name
        return self

    def analyzer_count(self):
        return len(self._analyzers)

    def run(self, table):
        spec = {"table_name": self._table, "continue_on_error": self._continue,
                "analyzers": [a.spec for a in self._analyzers]}
        name_arr, n_cols, col_arr, n_batches, _keep = _flatten_table(table)
        out = C.c_char_p()
        err = _Error()
        _host_check(_host().tgx_host_run_analysis_json(json.dumps(spec).encode(), name_arr, n_cols, col_arr, n_batches,
                                                       C.byref(out), C.byref(err)), err)
        return AnalyzerContext(_take(out))


# ---------------------------------------------------------------------------------------------- bridge
_HOST = None


def _host():
    global _HOST
    if _HOST is None:
        L = _lib.lib()
        E = C.POINTER(_Error)
        L.tgx_host_run_suite_json.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(_Column),
                                              C.c_size_t, C.POINTER(C.c_char_p), E]
        L.tgx_host_constraint_plan_json.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), E]
        L.tgx_host_constraint_verdict_json.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), E]
        L.tgx_host_validate_identifier.argtypes = [C.c_char_p, E]
        L.tgx_host_assertion_json.argtypes = [C.c_char_p, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_char_p), E]
        L.tgx_host_run_analysis_json.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(_Column),
                                                 C.c_size_t, C.POINTER(C.c_char_p), E]
        L.tgx_host_merge_states_json.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), E]
        L.tgx_host_metric_from_state_json.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), E]
        L.tgx_host_free.argtypes = [C.c_void_p]
        L.tgx_host_free.restype = None
        _HOST = L
    return _HOST


def _host_check(status, err):
    if status != 0:
        raise TgxError(status, err.msg.decode("utf-8", "replace"))


def _take(out):
    text = out.value.decode("utf-8")
    _host().tgx_host_free(C.cast(out, C.c_void_p))
    return text


def constraint_plan(constraint):
    """the aggregates one constraint (a dict as CheckBuilder builds them) asks for"""
    out = C.c_char_p()
    err = _Error()
    _host_check(_host().tgx_host_constraint_plan_json(json.dumps(constraint).encode(), C.byref(out), C.byref(err)), err)
    return json.loads(_take(out))


def constraint_verdict(constraint, results):
    """Constraint::evaluate's verdict half on given aggregates (list of dicts with tgx_result field names)"""
    out = C.c_char_p()
    err = _Error()
    _host_check(_host().tgx_host_constraint_verdict_json(json.dumps(constraint).encode(), json.dumps(results).encode(),
                                                         C.byref(out), C.byref(err)), err)
    return json.loads(_take(out))


def validate_identifier(identifier):
    err = _Error()
    _host_check(_host().tgx_host_validate_identifier(identifier.encode(), C.byref(err)), err)


# ---------------------------------------------------------------------------------------------- incremental analysis
class InMemoryStateStore:
    """TG/analyzers/incremental/state_store.rs StateStore: partition -> {metric_key: state}"""

    def __init__(self):
        self._p = {}

    def load_state(self, partition):
        return dict(self._p.get(partition, {}))

    def save_state(self, partition, state_map):
        self._p[partition] = dict(state_map)

    def list_partitions(self):
        return sorted(self._p)

    def delete_partition(self, partition):
        self._p.pop(partition, None)


class FileSystemStateStore:
    """state_store.rs:37-190: <base>/<partition>/<metric_key>.json, each file the serde_json form of the analyzer's
    state struct -- the states produced here use the reference's field names, so a directory written by either side
    can be read by the other."""

    def __init__(self, base_path):
        import os

        self._base = str(base_path)
        os.makedirs(self._base, exist_ok=True)

    def _dir(self, partition):
        import os

        return os.path.join(self._base, partition)

    def load_state(self, partition):
        import os

        out, d = {}, self._dir(partition)
        if not os.path.isdir(d):
            return out
        for name in sorted(os.listdir(d)):
            if name.endswith(".json"):
                with open(os.path.join(d, name)) as f:
                    out[name[:-5]] = json.load(f)
        return out

    def save_state(self, partition, state_map):
        import os

        d = self._dir(partition)
        os.makedirs(d, exist_ok=True)
        for key, state in state_map.items():
            with open(os.path.join(d, key + ".json"), "w") as f:
                json.dump(state, f)

    def list_partitions(self):
        import os

        return sorted(n for n in os.listdir(self._base) if os.path.isdir(os.path.join(self._base, n)))

    def delete_partition(self, partition):
        import shutil

        shutil.rmtree(self._dir(partition), ignore_errors=True)


def _state_is_empty(analyzer, state):
    """AnalyzerState::is_empty of the mirrored state types"""
    t = analyzer.spec["type"]
    if t in ("size", "mean", "standard_deviation"):
        return state.get("count", 0) == 0
    if t in ("completeness", "distinctness", "approx_count_distinct"):
        return state.get("total_count", 0) == 0
    if t in ("min", "max"):
        return state.get("min") is None and state.get("max") is None
    if t == "sum":
        return not state.get("has_values", False)
    return False  # correlation: the trait default


class IncrementalAnalysisRunner:
    """TG/analyzers/incremental/runner.rs:102-430.  Each partition's states come from ONE fused pass on the GPU
    (AnalysisRunner); merging and metrics are the reference's state algebra (csrc/host/analyzers.cpp)."""

    def __init__(self, state_store, fail_fast=True, save_empty_states=False, max_merge_batch_size=100):
        self._store, self._analyzers = state_store, []
        self._fail_fast, self._save_empty, self._batch = fail_fast, save_empty_states, max_merge_batch_size

    def add_analyzer(self, analyzer):
        self._analyzers.append(analyzer)
        return self

    def analyzer_count(self):
        return len(self._analyzers)

    def list_partitions(self):
        return self._store.list_partitions()

    def delete_partition(self, partition):
        self._store.delete_partition(partition)

    def _fresh(self, table):
        r = AnalysisRunner().continue_on_error(True)
        for a in self._analyzers:
            r.add(a)
        ctx = r.run(table)
        if ctx.has_errors() and self._fail_fast:
            raise TgxError(1, ctx.errors()[0]["error"])
        return ctx

    def analyze_partition(self, table, partition):
        """runner.rs:139-213: states of `table` saved under `partition`, metrics returned"""
        ctx = self._fresh(table)
        states = {a.metric_key(): ctx.states[a.metric_key()] for a in self._analyzers if a.metric_key() in ctx.states}
        self._store.save_state(partition, {k: v for k, v in states.items()
                                           if self._save_empty or not _state_is_empty(self._by_key(k), v)})
        return ctx

    def _by_key(self, key):
        return next(a for a in self._analyzers if a.metric_key() == key)

    def _finish(self, merged_by_key, errors):
        metrics = {}
        for a in self._analyzers:
            key = a.metric_key()
            if key not in merged_by_key:
                continue
            try:
                metrics[key] = a.compute_metric_from_state(merged_by_key[key])
            except TgxError as e:
                if self._fail_fast:
                    raise
                errors.append({"analyzer_name": a.name(), "error": e.msg})
        return AnalyzerContext(json.dumps({"metrics": metrics, "states": merged_by_key, "errors": errors}))

    def analyze_incremental(self, table, partition):
        """runner.rs:216-317: new data's states merged into the partition's stored states"""
        existing = self._store.load_state(partition)
        ctx = self._fresh(table)
        merged, errors = {}, list(ctx.errors())
        for a in self._analyzers:
            key = a.metric_key()
            if key not in ctx.states:
                continue
            new = ctx.states[key]
            try:
                merged[key] = a.merge_states([existing[key], new]) if key in existing else new
            except TgxError as e:
                if self._fail_fast:
                    raise
                errors.append({"analyzer_name": a.name(), "error": e.msg})
        self._store.save_state(partition, {k: v for k, v in merged.items()
                                           if self._save_empty or not _state_is_empty(self._by_key(k), v)})
        return self._finish(merged, errors)

    def analyze_partitions(self, partitions):
        """runner.rs:320-414: metrics over the union of stored partitions, no data access"""
        partitions = list(partitions)
        by_key, errors = {}, []
        for i in range(0, len(partitions), self._batch):
            for p in partitions[i:i + self._batch]:
                for key, st in self._store.load_state(p).items():
                    by_key.setdefault(key, []).append(st)
        merged = {}
        for a in self._analyzers:
            key = a.metric_key()
            if not by_key.get(key):
                continue
            try:
                merged[key] = a.merge_states(by_key[key])
            except TgxError as e:
                if self._fail_fast:
                    raise
                errors.append({"analyzer_name": a.name(), "error": e.msg})
        return self._finish(merged, errors)

