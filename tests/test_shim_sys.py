"""The Rust shim's raw bindings (shim/src/sys.rs) are generated from include/tgx.h: the committed file must be what
the generator makes of the header today, bind every prototype, and mirror every struct field for field."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sys_rs_is_up_to_date():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_shim_sys.py"), "--check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def test_every_prototype_and_field_of_the_header_is_bound():
    header = open(os.path.join(ROOT, "include", "tgx.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    sys_rs = open(os.path.join(ROOT, "shim", "src", "sys.rs")).read()
    protos = set(re.findall(r"\b(tgx_[a-z0-9_]+)\s*\(", header))
    bound = set(re.findall(r"pub fn (tgx_[a-z0-9_]+)\(", sys_rs))
    assert protos == bound, (protos - bound, bound - protos)
    structs = [(m.group(2), m.group(1))
               for m in re.finditer(r"typedef\s+struct\s+[A-Za-z_0-9]+\s*\{(.*?)\}\s*([A-Za-z_0-9]+)\s*;", header, flags=re.S)]
    assert len(structs) >= 6
    for name, body in structs:
        rust = re.search(r"pub struct %s \{(.*?)\n\}" % name, sys_rs, flags=re.S)
        assert rust, name
        rust_fields = re.findall(r"pub ([a-z_0-9]+):", rust.group(1))
        c_fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fp = re.match(r".*\(\s*\*\s*([A-Za-z_0-9]+)\s*\)\s*\(", decl)
            if fp:
                c_fields.append(fp.group(1))
                continue
            for part in decl.split(","):
                c_fields.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[[^\]]*\])?\s*$", part.strip())[0])
        assert [f.rstrip("_") for f in rust_fields] == c_fields, name


def test_planner_covers_the_host_layers_constraint_table():
    """every constraint class of the C++ host layer (the table the GPU tests exercise) has its verdict in the shim"""
    host = open(os.path.join(ROOT, "term_amd", "csrc", "host", "term_guard.cpp")).read()
    planner = open(os.path.join(ROOT, "shim", "src", "planner.rs")).read()
    classes = set(re.findall(r"class (\w+)Constraint : public Constraint", host))
    assert classes == {"Size", "Completeness", "Statistical", "MultiStatistical", "Uniqueness", "Format", "Length",
                       "Containment", "ApproxCountDistinct", "Quantile", "Correlation"}
    for verdict in ("Size", "Completeness", "Statistic", "Uniqueness", "Format", "Length", "Containment",
                    "ApproxCountDistinct", "Quantile", "Pearson", "Independence"):
        assert "Verdict::%s" % verdict in planner, verdict
    # the messages are the reference's: spot-check the ones its tests pin
    for text in ("is below threshold", "Uniqueness ratio", "Primary key columns contain", "Format validation ratio",
                 "Length constraint failed", "values are not in the allowed set", "is null (no non-null values)"):
        assert text in planner and text in host, text
