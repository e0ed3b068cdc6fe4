#!/usr/bin/env python3
"""Writes tests/golden/reference_vectors.json.

The vectors are the known-answer cases of the reference's own unit tests (inputs and expected
outputs only -- SURVEY.md section 4), transcribed by hand; `ref` gives the test's file:line under
/root/reference/term-guard/src.  Nothing here reads /root/reference at run time.

Run:  python tests/golden/make_reference_vectors.py
"""
import json
import os

N = None

completeness = [
    dict(ref="constraints/completeness.rs:339-354", columns={"id": [1, 2, 3, 4]}, cols=["id"],
         operator="all", threshold=1.0, status="success", metric=1.0),
    dict(ref="constraints/completeness.rs:356-377", columns={"email": [1, 2, N, 4, 5]}, cols=["email"],
         operator="all", threshold=0.8, status="success", metric=0.8),
    dict(ref="constraints/completeness.rs:379-396", columns={"phone": [1, N, N, 4]}, cols=["phone"],
         operator="all", threshold=0.8, status="failure", metric=0.5, message_contains="50.00%"),
    dict(ref="constraints/completeness.rs:398-423",
         columns={"first_name": [1, 2, 3], "last_name": [10, 20, 30]}, cols=["first_name", "last_name"],
         operator="all", threshold=1.0, status="success", metric=1.0),
    dict(ref="constraints/completeness.rs:425-450",
         columns={"col1": [1, 2, 3], "col2": [N, 20, 30], "col3": [100, 200, 300]},
         cols=["col1", "col2", "col3"], operator="all", threshold=1.0, status="failure",
         message_contains="col2"),
    dict(ref="constraints/completeness.rs:452-476",
         columns={"phone": [1, N, N], "email": [N, 2, N], "address": [N, N, N]},
         cols=["phone", "email", "address"], operator="any", threshold=0.3, status="success"),
    dict(ref="constraints/completeness.rs:478-504",
         columns={"col1": [1, 2, 3, 4], "col2": [10, 20, 30, 40], "col3": [N, 200, 300, 400],
                  "col4": [100, N, 3000, 4000]},
         cols=["col1", "col2", "col3", "col4"], operator="at_least:2", threshold=0.8, status="success"),
    dict(ref="constraints/completeness.rs:506-529",
         columns={"a": [1, 2, 3], "b": [10, N, 30], "c": [N, N, N]}, cols=["a", "b", "c"],
         operator="exactly:1", threshold=1.0, status="success"),
    dict(ref="constraints/completeness.rs:531-540", columns={"id": []}, cols=["id"], operator="all",
         threshold=1.0, status="skipped"),
]

statistics = [
    dict(ref="constraints/statistics.rs:563-573", values=[10.0, 20.0, 30.0], stat="mean",
         assertion=["equals", 20.0], status="success", metric=20.0),
    dict(ref="constraints/statistics.rs:575-593", values=[5.0, 10.0, 15.0], stat="min",
         assertion=["equals", 5.0], status="success", metric=5.0),
    dict(ref="constraints/statistics.rs:575-593", values=[5.0, 10.0, 15.0], stat="max",
         assertion=["equals", 15.0], status="success", metric=15.0),
    dict(ref="constraints/statistics.rs:595-605", values=[10.0, 20.0, 30.0], stat="sum",
         assertion=["equals", 60.0], status="success", metric=60.0),
    dict(ref="constraints/statistics.rs:607-616", values=[10.0, N, 20.0], stat="mean",
         assertion=["equals", 15.0], status="success", metric=15.0),
    dict(ref="constraints/statistics.rs:618-628", values=[N, N, N], stat="mean",
         assertion=["equals", 0.0], status="failure", message_contains="null"),
]

uniqueness = [
    dict(ref="constraints/uniqueness.rs:907-918", kind="full_uniqueness", values=["A", "B", "C", "A"],
         threshold=0.7, status="success", metric=0.75),
    dict(ref="constraints/uniqueness.rs:920-934", kind="full_uniqueness", values=["A", "B", N, "A"],
         threshold=0.4, status="success", metric=0.5),
    dict(ref="constraints/uniqueness.rs:936-949", kind="distinctness", values=["A", "B", "C", "A"],
         assertion=["equals", 0.75], status="success", metric=0.75),
    dict(ref="constraints/uniqueness.rs:951-965", kind="unique_value_ratio", values=["A", "B", "C", "A"],
         assertion=["equals", 0.5], status="success", metric=0.5),
    dict(ref="constraints/uniqueness.rs:967-979", kind="primary_key", values=["A", "B", "C"],
         status="success", metric=1.0),
    dict(ref="constraints/uniqueness.rs:981-993", kind="primary_key", values=["A", "B", N],
         status="failure", message_contains="NULL values"),
    dict(ref="constraints/uniqueness.rs:995-1007", kind="primary_key", values=["A", "B", "A"],
         status="failure", message_contains="duplicate values"),
    dict(ref="constraints/uniqueness.rs:1043-1057", kind="unique_with_nulls_include",
         values=["A", "B", N, N], threshold=0.4, status="success", metric=0.75),
    dict(ref="constraints/uniqueness.rs:1059-1070", kind="full_uniqueness", values=[], threshold=1.0,
         status="skipped"),
]

# multi-column: COUNT(DISTINCT (col1, col2)) / the '|'-concatenation of COALESCE(CAST(c AS VARCHAR), '<NULL>')
uniqueness_multi = [
    dict(ref="constraints/uniqueness.rs:1009-1023", kind="full_uniqueness", col1=["A", "B", "A"], col2=["1", "2", "2"],
         threshold=0.9, status="success", metric=1.0),
    dict(ref="constraints/uniqueness.rs:1025-1041", kind="distinctness", col1=["A", "B", "A"], col2=["1", "2", "1"],
         assertion=["greater_than", 0.5], status="success", metric=2.0 / 3.0),
]

# constraints/length.rs:246-438
length = [
    dict(ref="constraints/length.rs:248-267", kind="min", a=5, values=["hello", "world", "testing", "great", N],
         status="success", metric=1.0, name="min_length"),
    dict(ref="constraints/length.rs:269-288", kind="min", a=5, values=["hi", "hello", "a", "testing", N],
         status="failure", metric=0.6, message_contains="at least 5 characters"),
    dict(ref="constraints/length.rs:290-302", kind="max", a=10, values=["hi", "hey", "test", N],
         status="success", metric=1.0, name="max_length"),
    dict(ref="constraints/length.rs:304-322", kind="max", a=10,
         values=["short", "this is a very long string that exceeds the limit", "ok", N],
         status="failure", metric=0.75, message_contains="at most 10 characters"),
    dict(ref="constraints/length.rs:324-347", kind="between", a=3, b=10,
         values=["hello", "testing", "hi", "this is way too long", N],
         status="failure", metric=0.6, message_contains="between 3 and 10 characters", name="length_between"),
    dict(ref="constraints/length.rs:349-369", kind="exactly", a=5, values=["hello", "world", "test", "testing", N],
         status="failure", metric=0.6, message_contains="exactly 5 characters", name="exact_length"),
    dict(ref="constraints/length.rs:371-391", kind="not_empty", values=["hello", "a", "", "testing", N],
         status="failure", metric=0.8, message_contains="not empty", name="not_empty"),
    dict(ref="constraints/length.rs:393-412", kind="min", a=2, values=["hello", "\u4f60\u597d", "\U0001f980\U0001f525", "caf\u00e9", N],
         status="success", metric=1.0),
    dict(ref="constraints/length.rs:414-426", kind="min", a=5, values=[N, N, N], status="success", metric=1.0),
    dict(ref="constraints/length.rs:428-438", kind="min", a=5, values=[], status="skipped"),
]

# constraints/values.rs:520-601 (ContainmentConstraint)
containment = [
    dict(ref="constraints/values.rs:522-542", values=["active", "inactive", "pending", "invalid_status"],
         allowed=["active", "inactive", "pending", "archived"], status="failure", metric=0.75,
         message_contains="values are not in the allowed set"),
    dict(ref="constraints/values.rs:544-559", values=["active", "inactive", "pending"],
         allowed=["active", "inactive", "pending", "archived"], status="success", metric=1.0),
    dict(ref="constraints/values.rs:589-601", values=["active", N, "inactive", N], allowed=["active", "inactive"],
         status="success", metric=1.0),
]

# constraints/approx_count_distinct.rs:186-347 (the reference's tests bound the HyperLogLog estimate; `exact` is the
# number of distinct non-NULL values, which lies inside every one of those bounds)
approx_count_distinct = [
    dict(ref="constraints/approx_count_distinct.rs:190-203", dtype="int64", values=list(range(1000)),
         assertion=["greater_than", 990.0], status="success", bounds=[990.0, 1e18], exact=1000.0),
    dict(ref="constraints/approx_count_distinct.rs:206-225", dtype="int64", values=[1, 2, 3] * 100,
         assertion=["less_than", 10.0], status="success", bounds=[0.0, 10.0], exact=3.0),
    dict(ref="constraints/approx_count_distinct.rs:228-255", dtype="int64", values=[1, N, 2, N, 3, N, 1, 2, 3, N],
         assertion=["between", 2.0, 5.0], status="success", bounds=[2.0, 5.0], exact=3.0),
    dict(ref="constraints/approx_count_distinct.rs:258-273", dtype="int64", values=[i % 10 for i in range(50)],
         assertion=["greater_than", 100.0], status="failure", bounds=[0.0, 20.0], exact=10.0,
         message="Approximate distinct count 10 does not satisfy assertion greater than 100 for column 'test_col'"),
    dict(ref="constraints/approx_count_distinct.rs:276-297", dtype="string",
         values=["apple", "banana", "cherry", "apple", "banana", "date", "elderberry", N],
         assertion=["between", 4.0, 6.0], status="success", bounds=[4.0, 6.0], exact=5.0),
    dict(ref="constraints/approx_count_distinct.rs:300-311", dtype="int64", values=[],
         assertion=["equals", 0.0], status="success", bounds=[0.0, 0.0], exact=0.0),
    dict(ref="constraints/approx_count_distinct.rs:314-326", dtype="int64", values=[N, N, N, N, N],
         assertion=["equals", 0.0], status="success", bounds=[0.0, 0.0], exact=0.0),
    # "Should be within 3% of 1000": 10 000 rows, i % 1000
    dict(ref="constraints/approx_count_distinct.rs:329-348", dtype="int64", values=[i % 1000 for i in range(10000)],
         assertion=["between", 970.0, 1030.0], status="success", bounds=[970.0, 1030.0], exact=1000.0),
]

EMAIL = "email"
fmt = [
    dict(ref="constraints/format.rs:917-934", format="email", threshold=0.7,
         values=["test@example.com", "user@domain.org", "invalid-email", "another@test.net"],
         status="success", metric=0.75, name="email"),
    dict(ref="constraints/format.rs:937-955", format="url", allow_localhost=False, threshold=0.7,
         values=["https://example.com", "http://test.org", "not-a-url", "https://another.site.net/path"],
         status="success", metric=0.75, name="url"),
    dict(ref="constraints/format.rs:957-974", format="url", allow_localhost=True, threshold=0.7,
         values=["https://localhost:3000", "http://localhost", "https://example.com", "not-a-url"],
         status="success", metric=0.75),
    dict(ref="constraints/format.rs:976-994", format="credit_card", detect_only=True, threshold=0.8,
         values=["4111-1111-1111-1111", "5555 5555 5555 4444", "normal text", "4111111111111111"],
         status="success", metric=0.75, name="credit_card"),
    dict(ref="constraints/format.rs:996-1013", format="phone", country="US", threshold=0.7, trim=True,
         values=["(555) 123-4567", "555-123-4567", "5551234567", "invalid-phone"],
         status="success", metric=0.75, name="phone"),
    dict(ref="constraints/format.rs:1015-1033", format="postal_code", country="US", threshold=0.7, trim=True,
         values=["12345", "12345-6789", "invalid", "98765"], status="success", metric=0.75,
         name="postal_code"),
    dict(ref="constraints/format.rs:1035-1053", format="uuid", threshold=0.7,
         values=["550e8400-e29b-41d4-a716-446655440000", "6ba7b810-9dad-11d1-80b4-00c04fd430c8",
                 "invalid-uuid", "6ba7b811-9dad-11d1-80b4-00c04fd430c8"],
         status="success", metric=0.75, name="uuid"),
    dict(ref="constraints/format.rs:1055-1073", format="ipv4", threshold=0.7,
         values=["192.168.1.1", "10.0.0.1", "256.256.256.256", "172.16.0.1"],
         status="success", metric=0.75, name="ipv4"),
    dict(ref="constraints/format.rs:1075-1093", format="ipv6", threshold=0.7,
         values=["2001:0db8:85a3:0000:0000:8a2e:0370:7334", "2001:db8:85a3::8a2e:370:7334",
                 "invalid-ipv6", "::1"],
         status="success", metric=0.75, name="ipv6"),
    dict(ref="constraints/format.rs:1095-1113", format="json", threshold=0.7,
         values=['{"key": "value"}', "[1, 2, 3]", "not json", '{"nested": {"key": "value"}}'],
         status="success", metric=0.75, name="json"),
    dict(ref="constraints/format.rs:1115-1133", format="iso8601_datetime", threshold=0.7,
         values=["2023-12-25T10:30:00Z", "2023-12-25T10:30:00.123Z", "invalid-datetime",
                 "2023-12-25T10:30:00+05:30"],
         status="success", metric=0.75, name="iso8601_datetime"),
    dict(ref="constraints/format.rs:1135-1153", format="regex", pattern=r"^[A-Z]{3}\d{3}$", threshold=0.7,
         values=["ABC123", "DEF456", "invalid", "GHI789"], status="success", metric=0.75, name="regex"),
    dict(ref="constraints/format.rs:1155-1179", format="regex", pattern=r"^[A-Z]{3}\d{3}$", threshold=0.7,
         case_sensitive=False, values=["abc123", "DEF456", "invalid", "ghi789"],
         status="success", metric=0.75),
    dict(ref="constraints/format.rs:1181-1204", format="email", threshold=0.7, trim=True,
         values=["  test@example.com  ", "user@domain.org", "  invalid-email  ", " another@test.net "],
         status="success", metric=0.75),
    dict(ref="constraints/format.rs:1206-1224", format="email", threshold=0.6, null_is_valid=True,
         values=["test@example.com", N, "invalid-email", N], status="success", metric=0.75),
    dict(ref="constraints/format.rs:1226-1240", format="email", threshold=0.2, null_is_valid=False,
         values=["test@example.com", N, "invalid-email", N], status="success", metric=0.25),
    dict(ref="constraints/format.rs:1242-1260", format="email", threshold=0.5,
         values=["invalid", "also_invalid", "nope", "still_invalid"], status="failure", metric=0.0),
    dict(ref="constraints/format.rs:1262-1271", format="email", threshold=0.9, values=[], status="skipped"),
    dict(ref="constraints/format.rs:1389-1407", format="social_security_number", threshold=0.95, trim=True,
         values=["123-45-6789", "123456789", "456-78-9012", "789012345"], status="success", metric=1.0,
         name="social_security_number"),
    dict(ref="constraints/format.rs:1409-1431", format="social_security_number", threshold=0.0, trim=True,
         values=["000-12-3456", "666-12-3456", "900-12-3456", "123-00-4567", "123-45-0000"],
         status="success", metric=0.0),
    dict(ref="constraints/format.rs:1433-1458", format="social_security_number", threshold=0.5, trim=True,
         values=["123-45-6789", "not-an-ssn", "666-12-3456", "456789012", "123 45 6789", "789-01-2345",
                 N, "234-56-7890"],
         status="success", metric=0.625),
    dict(ref="constraints/format.rs:1460-1476", format="social_security_number", threshold=0.8, trim=True,
         values=["123-45-6789", "invalid", "234-56-7890", "not-ssn"], status="failure", metric=0.5),
    dict(ref="constraints/format.rs:1478-1484", format="social_security_number", threshold=0.4, trim=True,
         values=["123-45-6789", "invalid", "234-56-7890", "not-ssn"], status="success", metric=0.5),
    dict(ref="constraints/format.rs:1486-1508", format="social_security_number", threshold=0.3, trim=True,
         values=["078-05-1120", "219-09-9999", "457-55-5462", "999-99-9999", "123-45-67890",
                 "12-345-6789", "ABC-DE-FGHI", ""],
         status="success", metric=0.375),
]

# built-in pattern strings are DATA the reference's tests exercise; kept here so the oracle's and the
# product's copies can both be checked against one transcription (constraints/format.rs:237-294)
patterns = {
    "email": r"^[a-zA-Z0-9.!#$%&'*+/=?^_`{|}~-]+@[a-zA-Z0-9](?:[a-zA-Z0-9-]{0,61}[a-zA-Z0-9])?(?:\.[a-zA-Z0-9](?:[a-zA-Z0-9-]{0,61}[a-zA-Z0-9])?)*$",
    "url": r"^https?://[a-zA-Z0-9.-]+\.[a-zA-Z]{2,}(?::\d+)?(?:/[^\s]*)?$",
    "url_localhost": r"^https?://(?:localhost|(?:[a-zA-Z0-9.-]+\.?[a-zA-Z]{2,}|(?:\d{1,3}\.){3}\d{1,3}))(?::\d+)?(?:/[^\s]*)?$",
    "credit_card": r"^(?:4[0-9]{12}(?:[0-9]{3})?|5[1-5][0-9]{14}|3[47][0-9]{13}|3[0-9]{13}|6(?:011|5[0-9]{2})[0-9]{12})$|^(?:\d{4}[-\s]?){3}\d{4}$",
    "phone_US": r"^(\+?1[-.\s]?)?\(?([0-9]{3})\)?[-.\s]?([0-9]{3})[-.\s]?([0-9]{4})$",
    "phone_CA": r"^(\+?1[-.\s]?)?\(?([0-9]{3})\)?[-.\s]?([0-9]{3})[-.\s]?([0-9]{4})$",
    "phone_UK": r"^(\+44\s?)?(?:\(?0\d{4}\)?\s?\d{6}|\(?0\d{3}\)?\s?\d{7}|\(?0\d{2}\)?\s?\d{8})$",
    "phone_DE": r"^(\+49\s?)?(?:\(?0\d{2,5}\)?\s?\d{4,12})$",
    "phone_FR": r"^(\+33\s?)?(?:\(?0\d{1}\)?\s?\d{8})$",
    "phone": r"^[\+]?[1-9][\d]{0,15}$",
    "postal_code_US": r"^\d{5}(-\d{4})?$",
    "postal_code_CA": r"^[A-Za-z]\d[A-Za-z][ -]?\d[A-Za-z]\d$",
    "postal_code_UK": r"^[A-Z]{1,2}\d[A-Z\d]?\s?\d[A-Z]{2}$",
    "postal_code_DE": r"^\d{5}$",
    "postal_code_FR": r"^\d{5}$",
    "postal_code_JP": r"^\d{3}-\d{4}$",
    "postal_code_AU": r"^\d{4}$",
    "postal_code": r"^[A-Za-z0-9\s-]{3,10}$",
    "uuid": r"^[0-9a-fA-F]{8}-[0-9a-fA-F]{4}-[1-5][0-9a-fA-F]{3}-[89abAB][0-9a-fA-F]{3}-[0-9a-fA-F]{12}$",
    "ipv4": r"^(?:(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.){3}(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)$",
    "ipv6": r"^([0-9a-fA-F]{0,4}:){1,7}([0-9a-fA-F]{0,4})?$|^::$|^::1$|^([0-9a-fA-F]{1,4}:)*::([0-9a-fA-F]{1,4}:)*[0-9a-fA-F]{1,4}$",
    "json": r"^\s*[\{\[].*[\}\]]\s*$",
    "iso8601_datetime": r"^\d{4}-\d{2}-\d{2}T\d{2}:\d{2}:\d{2}(?:\.\d+)?(?:Z|[+-]\d{2}:\d{2})$",
    "social_security_number": r"^(00[1-9]|0[1-9][0-9]|[1-5][0-9]{2}|6[0-5][0-9]|66[0-5]|667|66[89]|6[7-9][0-9]|[7-8][0-9]{2})-?(0[1-9]|[1-9][0-9])-?(000[1-9]|00[1-9][0-9]|0[1-9][0-9]{2}|[1-9][0-9]{3})$",
}

analyzers = dict(
    ref="analyzers/basic/tests.rs:12-36",
    table={"id": [1, 2, 3, 4, N], "value": [10.0, 20.0, N, 30.0, 40.0], "name": ["a", "b", "a", N, "c"]},
    expect=[
        dict(ref="analyzers/basic/tests.rs:72-89", analyzer="completeness", column="id",
             state={"total_count": 5, "non_null_count": 4}, metric=0.8),
        dict(ref="analyzers/basic/tests.rs:116-128", analyzer="distinctness", column="name",
             state={"total_count": 4, "distinct_count": 3}, metric=0.75),
        dict(ref="analyzers/basic/tests.rs:135-146", analyzer="mean", column="value",
             state={"sum": 100.0, "count": 4}, metric=25.0),
        dict(ref="analyzers/basic/tests.rs:172-194", analyzer="min", column="value", metric=10.0),
        dict(ref="analyzers/basic/tests.rs:172-194", analyzer="max", column="value", metric=40.0),
        dict(ref="analyzers/basic/tests.rs:218-233", analyzer="sum", column="value", metric=100.0),
        dict(ref="analyzers/basic/tests.rs:40-53", analyzer="size", metric=5),
    ],
    merges=[
        dict(ref="analyzers/basic/tests.rs:55-64", analyzer="size", states=[10, 20, 15], merged=45),
        dict(ref="analyzers/basic/tests.rs:92-109", analyzer="completeness",
             states=[[10, 8], [20, 18]], merged=[30, 26]),
        dict(ref="analyzers/basic/tests.rs:148-165", analyzer="mean", states=[[100.0, 4], [50.0, 2]],
             merged=[150.0, 6], metric=25.0),
        dict(ref="analyzers/basic/tests.rs:196-216", analyzer="minmax",
             states=[[10.0, 30.0], [5.0, 40.0]], merged=[5.0, 40.0]),
        dict(ref="analyzers/basic/tests.rs:236-256", analyzer="sum", states=[100.0, 50.0], merged=150.0),
    ],
    edge=[
        dict(ref="analyzers/basic/tests.rs:263-293", case="empty table", completeness=1.0, size=0),
        dict(ref="analyzers/basic/tests.rs:296-326", case="all null", rows=3, completeness=0.0, size=3),
    ],
)

correlation = dict(
    ref="analyzers/advanced/correlation.rs:470-548",
    x="i for i in 0..99", y="2x+1", n=100,
    pearson=dict(value=1.0, tol=1e-4), covariance=dict(lo=1600.0, hi=1700.0),
    spearman=dict(value=1.0, tol=1e-4),
)

kll = [
    dict(ref="analyzers/advanced/kll_sketch.rs:406-469", k=100, input="0..999", count=1000,
         checks=[dict(phi=0.5, expected=500.0, rel_err_lt=0.85), dict(phi=0.9, expected=900.0, rel_err_lt=0.85)]),
    dict(ref="analyzers/advanced/kll_sketch.rs:478-486", k=100, input=[42.0], count=1,
         checks=[dict(phi=0.0, equals=42.0), dict(phi=0.5, equals=42.0), dict(phi=1.0, equals=42.0)]),
    dict(ref="analyzers/advanced/kll_sketch.rs:488-523", k=100, merge=["0..499", "500..999"], count=1000,
         checks=[dict(phi=0.5, expected=500.0, rel_err_lt=0.6)]),
    dict(ref="analyzers/advanced/kll_sketch.rs:525-533", k=100, input=[1.0, "nan", 2.0], count=2, checks=[]),
    dict(ref="analyzers/advanced/kll_sketch.rs:535-541", k=200, error_bound=1.65 / (200 ** 0.5)),
]

assertion = [
    # constraints/assertion.rs:48-62 (Equals uses |v-e| < 1e-10) and :64-76 descriptions
    dict(kind="equals", args=[20.0], value=20.0 + 5e-11, ok=True, text="equals 20"),
    dict(kind="equals", args=[20.0], value=20.0 + 2e-10, ok=False, text="equals 20"),
    dict(kind="not_equals", args=[1.5], value=1.5, ok=False, text="not equals 1.5"),
    dict(kind="greater_than", args=[5.0], value=5.0, ok=False, text="greater than 5"),
    dict(kind="greater_than_or_equal", args=[5.0], value=5.0, ok=True, text="greater than or equal to 5"),
    dict(kind="less_than", args=[5.0], value=4.0, ok=True, text="less than 5"),
    dict(kind="less_than_or_equal", args=[5.0], value=5.5, ok=False, text="less than or equal to 5"),
    dict(kind="between", args=[10.0, 20.0], value=20.0, ok=True, text="between 10 and 20"),
    dict(kind="not_between", args=[10.0, 20.0], value=15.0, ok=False, text="not between 10 and 20"),
]

# ---- the unified constraints' own unit tests: MultiStatisticalConstraint, QuantileConstraint (Single / Multiple /
# Monotonic), CorrelationConstraint (Pairwise / Range / Independence).  `constraint` is the suite-JSON form of the
# constructor call the test makes (term_amd/suite.py); `table` the columns its fixture registers.
_x = [float(i) for i in range(100)]
_corr = dict(x=_x, y=[2.0 * i + (i % 10) - 5.0 for i in range(100)])      # constraints/correlation.rs:520-552
_indep = dict(x=_x, y=[float((i * 37) % 100) for i in range(100)])          # constraints/correlation.rs:554-584
_q100 = dict(value=[float(i) for i in range(1, 101)])                       # constraints/quantile.rs:529-530


def _a(kind, *args):
    return dict(kind=kind, args=list(args))


constraint_variants = [
    dict(ref="constraints/statistics.rs:642-662", table=dict(value=[10.0, 20.0, 30.0, 40.0]),
         constraint=dict(type="multi_statistic", column="value", statistics=[
             dict(statistic="min", p=0.5, assertion=_a("greater_than_or_equal", 10.0)),
             dict(statistic="max", p=0.5, assertion=_a("less_than_or_equal", 40.0)),
             dict(statistic="mean", p=0.5, assertion=_a("equals", 25.0)),
             dict(statistic="sum", p=0.5, assertion=_a("equals", 100.0))]),
         status="success", name="multi_statistical"),
    dict(ref="constraints/statistics.rs:664-682", table=dict(value=[10.0, 20.0, 30.0]),
         constraint=dict(type="multi_statistic", column="value", statistics=[
             dict(statistic="min", p=0.5, assertion=_a("equals", 5.0)),
             dict(statistic="max", p=0.5, assertion=_a("equals", 30.0))]),
         status="failure", message_contains="minimum is 10", name="multi_statistical"),
    dict(ref="constraints/quantile.rs:527-539", table=_q100,
         constraint=dict(type="quantile", column="value", validation="single", quantile=0.5,
                         assertion=_a("between", 45.0, 55.0)),
         status="success", name="quantile"),
    dict(ref="constraints/quantile.rs:541-553", table=_q100,
         constraint=dict(type="quantile", column="value", validation="single", quantile=0.95,
                         assertion=_a("between", 94.0, 96.0)),
         status="success", name="quantile"),
    dict(ref="constraints/quantile.rs:555-573", table=_q100,
         constraint=dict(type="quantile", column="value", validation="multiple", checks=[
             dict(quantile=0.25, assertion=_a("between", 24.0, 26.0)),
             dict(quantile=0.75, assertion=_a("between", 74.0, 76.0))]),
         status="success", name="quantile"),
    dict(ref="constraints/quantile.rs:575-593", table=_q100,
         constraint=dict(type="quantile", column="value", validation="monotonic", quantiles=[0.1, 0.5, 0.9],
                         strict=True),
         status="success", name="quantile"),
    dict(ref="constraints/correlation.rs:586-598", table=_corr,
         constraint=dict(type="correlation", validation="pairwise", correlation_type="pearson", column1="x",
                         column2="y", assertion=_a("greater_than", 0.9)),
         status="success", metric_gt=0.9, name="correlation"),
    dict(ref="constraints/correlation.rs:600-612", table=_indep,
         constraint=dict(type="correlation", validation="independence", column1="x", column2="y",
                         max_correlation=0.3),
         status="success", name="independence"),
    dict(ref="constraints/correlation.rs:614-631", table=_corr,
         constraint=dict(type="correlation", validation="range", correlation_type="pearson", column1="x",
                         column2="y", min=0.8, max=1.0),
         status="success", name="correlation_range"),
]
# constructor errors the same tests pin (text of the TermError)
constraint_variant_errors = [
    dict(ref="constraints/quantile.rs:595-603",
         constraint=dict(type="quantile", column="value", validation="single", quantile=1.5,
                         assertion=_a("less_than", 100.0)),
         error_contains="Quantile must be between 0.0 and 1.0"),
    dict(ref="constraints/correlation.rs:633-641",
         constraint=dict(type="correlation", validation="independence", column1="x", column2="y",
                         max_correlation=1.5),
         error_contains="Max correlation must be between 0.0 and 1.0"),
    dict(ref="constraints/correlation.rs:656-668",
         constraint=dict(type="correlation", validation="multi_column", columns=["a"], correlation_type="pearson"),
         error_contains="At least 2 columns required"),
    dict(ref="constraints/statistics.rs:684-699",
         constraint=dict(type="statistic", column="value", statistic="percentile", p=1.5,
                         assertion=_a("less_than", 100.0)),
         error_contains="Percentile must be between 0.0 and 1.0"),
]

out = dict(constraint_variants=constraint_variants, constraint_variant_errors=constraint_variant_errors,
           completeness=completeness, statistics=statistics, uniqueness=uniqueness,
           uniqueness_multi=uniqueness_multi, length=length, containment=containment, approx_count_distinct=approx_count_distinct, format=fmt,
           patterns=patterns, analyzers=analyzers, correlation=correlation, kll=kll, assertion=assertion)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print("wrote", path)
