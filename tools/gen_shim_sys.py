#!/usr/bin/env python3
"""include/tgx.h -> shim/src/sys.rs: the raw Rust bindings of the C ABI, generated so that they cannot drift.

    python tools/gen_shim_sys.py            # writes shim/src/sys.rs
    python tools/gen_shim_sys.py --check    # exit code 1 when the committed file differs (tests/test_shim_sys.py)

The header is plain C in a regular style (no macros in declarations, one declarator per parameter), so a small
hand-written reader is enough: enums and #defines become constants, structs become #[repr(C)] structs with the same
field order, prototypes become one `extern "C"` block.  C names are kept (tgx_column, tgx_update, ..) as in any -sys
crate."""
import argparse
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
HEADER = os.path.join(ROOT, "include", "tgx.h")
OUT = os.path.join(ROOT, "shim", "src", "sys.rs")

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "uint8_t": "u8", "int8_t": "i8",
           "uint16_t": "u16", "int16_t": "i16", "size_t": "usize", "double": "f64", "float": "f32", "char": "c_char",
           "void": "c_void", "int": "c_int"}
KEYWORDS = {"type", "ref", "fn", "mod", "use", "in", "match", "move", "box", "loop", "impl", "where", "self"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def rust_ident(name):
    return name + "_" if name in KEYWORDS else name


def rust_type(ctype, enums, structs):
    """ctype: tokens of a C type without the declarator name, e.g. 'const uint8_t * const *'"""
    toks = ctype.replace("*", " * ").split()
    toks = [t for t in toks if t != "struct"]
    # base type = first non-const token; then a chain of (const?) pointers, read left to right
    i, const_base = 0, False
    if toks[i] == "const":
        const_base, i = True, i + 1
    base = toks[i]
    i += 1
    if i < len(toks) and toks[i] == "const":  # "uint8_t const"
        const_base, i = True, i + 1
    if base in SCALARS:
        rt = SCALARS[base]
    elif base in enums:
        rt = "i32"  # C enums of this header are ints; the structs themselves use int32_t fields
    elif base in structs or base.startswith("tgx_"):
        rt = base
    else:
        raise ValueError("unknown C type %r in %r" % (base, ctype))
    constness = const_base
    while i < len(toks):
        assert toks[i] == "*", ctype
        i += 1
        rt = ("*const " if constness else "*mut ") + rt
        constness = False
        if i < len(toks) and toks[i] == "const":
            constness, i = True, i + 1
    return rt


def split_decl(decl):
    """'const uint8_t *const *variadic' -> ('const uint8_t *const *', 'variadic', array_len or None)"""
    decl = decl.strip()
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[\s*([A-Za-z0-9_]+)\s*\])?$", decl, flags=re.S)
    if not m:
        raise ValueError("cannot read declarator %r" % decl)
    return m.group(1).strip(), m.group(2), m.group(4)


def parse(text):
    text = strip_comments(text)
    consts, enums, structs, opaque, funcs = [], {}, {}, [], []
    for m in re.finditer(r"^#define\s+(TGX_[A-Z0-9_]+)\s+([0-9]+)\s*$", text, flags=re.M):
        consts.append((m.group(1), "u32", m.group(2)))
    # enums (named through typedef, or anonymous flag sets)
    for m in re.finditer(r"(typedef\s+)?enum\s*([A-Za-z_0-9]*)\s*\{(.*?)\}\s*([A-Za-z_0-9]*)\s*;", text, flags=re.S):
        name = m.group(4) or m.group(2)
        flags = "<<" in m.group(3)
        if name:
            enums[name] = True
        for item in m.group(3).split(","):
            item = item.strip()
            if not item:
                continue
            k, v = [x.strip() for x in item.split("=")]
            v = v.replace("u", "")
            consts.append((k, "u32" if flags else "i32", v))
    # structs
    for m in re.finditer(r"typedef\s+struct\s+([A-Za-z_0-9]+)\s*\{(.*?)\}\s*([A-Za-z_0-9]+)\s*;", text, flags=re.S):
        structs[m.group(3)] = m.group(2)
    for m in re.finditer(r"typedef\s+struct\s+([A-Za-z_0-9]+)\s+([A-Za-z_0-9]+)\s*;", text):
        if m.group(2) not in structs:
            opaque.append(m.group(2))
    # prototypes: everything outside braces that looks like `ret name(args);`
    flat = re.sub(r"\{.*?\}", "{}", text, flags=re.S)
    for m in re.finditer(r"^([A-Za-z_][A-Za-z0-9_ \*]*?)\b(tgx_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", flat, flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef"):
            continue
        funcs.append((ret, name, args))
    return consts, enums, structs, opaque, funcs


def field_lines(body, enums, structs):
    out = []
    # function-pointer fields first cut out: `int32_t (*name)(args);`
    pos = 0
    for part in re.split(r";", body):
        part = " ".join(part.split())
        if not part:
            continue
        fp = re.match(r"^(.*?)\(\s*\*\s*([A-Za-z_0-9]+)\s*\)\s*\((.*)\)$", part)
        if fp:
            ret = rust_type(fp.group(1).strip(), enums, structs)
            args = []
            for a in fp.group(3).split(","):
                t, n, _ = split_decl(a)
                args.append("%s: %s" % (rust_ident(n), rust_type(t, enums, structs)))
            out.append("    pub %s: Option<unsafe extern \"C\" fn(%s) -> %s>," % (rust_ident(fp.group(2)), ", ".join(args), ret))
            continue
        # `int64_t min_i, max_i` / `char msg[256]` / `const uint8_t *validity`
        first, *more = [x.strip() for x in part.split(",")]
        t, n, arr = split_decl(first)
        names = [(n, arr)]
        for extra in more:
            stars = len(extra) - len(extra.lstrip("*"))
            assert stars == 0, part
            _, n2, arr2 = split_decl("x " + extra) if " " not in extra and "[" not in extra else split_decl(t + " " + extra)
            names.append((n2, arr2))
        rt = rust_type(t, enums, structs)
        for n, arr in names:
            ty = "[%s; %s]" % (rt, arr) if arr else rt
            out.append("    pub %s: %s," % (rust_ident(n), ty))
        pos += 1
    return out


def generate():
    consts, enums, structs, opaque, funcs = parse(open(HEADER).read())
    o = ["// GENERATED by tools/gen_shim_sys.py from include/tgx.h -- do not edit (tests/test_shim_sys.py diffs it).",
         "// Raw bindings of libtgx's C ABI: constants, #[repr(C)] structs in the header's field order, the prototypes.",
         "#![allow(non_camel_case_types, non_snake_case, dead_code)]",
         "use std::os::raw::{c_char, c_int, c_void};", ""]
    for name, ty, val in consts:
        o.append("pub const %s: %s = %s;" % (name, ty, val))
    o.append("")
    for name in opaque:
        o += ["#[repr(C)]", "pub struct %s {" % name, "    _private: [u8; 0],", "}", ""]
    for name, body in structs.items():
        o += ["#[repr(C)]", "#[derive(Clone, Copy)]", "pub struct %s {" % name]
        o += field_lines(body, enums, structs)
        o += ["}", ""]
    o.append("extern \"C\" {")
    for ret, name, args in funcs:
        params = []
        if args.strip() not in ("", "void"):
            for a in args.split(","):
                t, n, arr = split_decl(a)
                rt = rust_type(t, enums, structs)
                if arr:  # an array parameter decays to a pointer
                    rt = ("*const " if t.startswith("const") else "*mut ") + rust_type(t.replace("const", "").strip(), enums, structs)
                params.append("%s: %s" % (rust_ident(n), rt))
        rret = "" if ret == "void" else " -> %s" % rust_type(ret, enums, structs)
        o.append("    pub fn %s(%s)%s;" % (name, ", ".join(params), rret))
    o += ["}", ""]
    return "\n".join(o)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    text = generate()
    if args.check:
        have = open(OUT).read() if os.path.exists(OUT) else ""
        if have != text:
            print("shim/src/sys.rs is out of date: run python tools/gen_shim_sys.py", file=sys.stderr)
            return 1
        return 0
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(text)
    return 0


if __name__ == "__main__":
    sys.exit(main())
