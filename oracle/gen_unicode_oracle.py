#!/usr/bin/env python3
"""Generates oracle/unicode_oracle_tables.h: the Unicode tables of the ORACLE's Pike VM (oracle/regex_oracle.c).
Test infrastructure, like everything under oracle/.

The product's compiler has tables of its own (term_amd/csrc/regex/unicode_tables.h, made by tools/gen_unicode_tables.py
by probing the PyPI `regex` module).  These are derived ANOTHER way, so that the parity suite does not hold a table
against itself (round-4 verdict: `(?i)` folded i with dotless i in both engines and no test could see it):

  * case folding -- ICU 70's `u_foldCase(c, U_FOLD_CASE_DEFAULT)`: simple case folding, status C + S of CaseFolding.txt
    without the Turkic lines, as Rust's regex-syntax folds (format.rs:756-760 `~*`), at Unicode 14.0 -- plus the lines
    CaseFolding-15.1.0 and -16.0.0 added, written out in tools/unicode_versions.py (3 + 27 lines).  Nothing of the
    `regex` module goes into the fold table.
  * classes (General_Category, Alphabetic, White_Space, Lowercase, Uppercase, Join_Control, Script) -- ICU 70's
    character database for every code point Unicode 14.0 assigns.  The 10 301 code points 15.0 - 16.0 added are known
    to ONE database in this image, the `regex` module's (17.0): for those, and only those, the module is asked; the
    4 803 of 17.0 stay unassigned (regex-syntax 0.8.8 is Unicode 16.0).

tests/test_unicode_tables.py compares the two headers: the fold tables must be the same set of pairs, the class tables
may differ in the few older characters whose properties 15.0 - 17.0 changed (listed there).

    python oracle/gen_unicode_oracle.py
"""
import ctypes
import os
import sys

import regex

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
import unicode_versions as UV  # noqa: E402

MAXCP = 0x10FFFF
GCS = ["L", "Lu", "Ll", "Lt", "Lm", "Lo", "M", "Mn", "Mc", "Me", "N", "Nd", "Nl", "No", "P", "Pc", "Pd", "Ps", "Pe",
       "Pi", "Pf", "Po", "S", "Sm", "Sc", "Sk", "So", "Z", "Zs", "Zl", "Zp", "C", "Cc", "Cf", "Co", "Cn"]
PROPS = ["Alphabetic", "White_Space", "Lowercase", "Uppercase"]
SCRIPTS = ["Latin", "Greek", "Cyrillic", "Han", "Hiragana", "Katakana", "Arabic", "Hebrew"]


class Icu:
    def __init__(self):
        lib = None
        for v in (70,):
            try:
                lib = ctypes.CDLL("libicuuc.so.%d" % v)
                self.sfx = "_%d" % v
                break
            except OSError:
                pass
        if lib is None:
            raise SystemExit("libicuuc.so.70 (ICU 70 = Unicode 14.0) is needed to regenerate the oracle's tables")
        self.lib = lib
        ver = (ctypes.c_uint8 * 4)()
        self.fn("u_getUnicodeVersion", None, [ctypes.c_void_p])(ver)
        assert list(ver)[:2] == [14, 0], list(ver)
        self.fold = self.fn("u_foldCase", ctypes.c_int32, [ctypes.c_int32, ctypes.c_uint32])
        self.char_type = self.fn("u_charType", ctypes.c_int8, [ctypes.c_int32])
        self.has_prop = self.fn("u_hasBinaryProperty", ctypes.c_int8, [ctypes.c_int32, ctypes.c_int32])
        self.prop_enum = self.fn("u_getPropertyEnum", ctypes.c_int32, [ctypes.c_char_p])
        self.value_enum = self.fn("u_getPropertyValueEnum", ctypes.c_int32, [ctypes.c_int32, ctypes.c_char_p])
        self.value_name = self.fn("u_getPropertyValueName", ctypes.c_char_p, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32])
        self.script = self.fn("uscript_getScript", ctypes.c_int32, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)])

    def fn(self, name, restype, argtypes):
        f = getattr(self.lib, name + self.sfx)
        f.restype, f.argtypes = restype, argtypes
        return f


def is_scalar(cp):
    return not (0xD800 <= cp <= 0xDFFF)


def ranges_of(member):
    out, start = [], None
    for cp in range(MAXCP + 2):
        ok = cp <= MAXCP and is_scalar(cp) and member[cp]
        if ok and start is None:
            start = cp
        elif not ok and start is not None:
            out.append((start, cp - 1))
            start = None
    return out


def main():
    icu = Icu()
    gc_prop = icu.prop_enum(b"General_Category")
    sc_prop = icu.prop_enum(b"Script")
    gc_name = {}  # UCharCategory value -> "Lu"
    for v in range(30):
        gc_name[v] = icu.value_name(gc_prop, v, 0).decode()  # (U_SHORT_PROPERTY_NAME = 0)
    assert gc_name[0] == "Cn" and gc_name[1] == "Lu" and gc_name[9] == "Nd", gc_name
    prop_id = {p: icu.prop_enum(p.encode()) for p in PROPS + ["Join_Control"]}
    script_id = {s: icu.value_enum(sc_prop, s.encode()) for s in SCRIPTS}
    assert all(v >= 0 for v in prop_id.values()) and all(v >= 0 for v in script_id.values())

    # per code point: general category, the binary properties, the script -- ICU for Unicode 14's repertoire, the
    # `regex` module for what 15.0 - 16.0 added, "unassigned" for 17.0's additions
    gc = [None] * (MAXCP + 1)
    props = {p: bytearray(MAXCP + 1) for p in prop_id}
    scripts = {s: bytearray(MAXCP + 1) for s in SCRIPTS}
    late = []
    err = ctypes.c_int32(0)
    rx_cn = regex.compile(r"\p{Cn}")
    for cp in range(MAXCP + 1):
        if not is_scalar(cp):
            gc[cp] = "Cs"
            continue
        t = icu.char_type(cp)
        if t == 0 and not UV.added_in_17(cp) and not rx_cn.fullmatch(chr(cp)):
            late.append(cp)
            continue
        gc[cp] = gc_name[t]
        for p, pid in prop_id.items():
            if icu.has_prop(cp, pid):
                props[p][cp] = 1
        sid = icu.script(cp, ctypes.byref(err))
        for s, want in script_id.items():
            if sid == want:
                scripts[s][cp] = 1
    assert len(late) == 4489 + 627 + 5185, len(late)  # the additions of Unicode 15.0, 15.1 and 16.0
    two_letter = [g for g in GCS if len(g) == 2]
    rx_gc = {g: regex.compile(r"\p{%s}" % g) for g in two_letter}
    rx_prop = {p: regex.compile(r"\p{%s}" % p) for p in prop_id}
    rx_script = {s: regex.compile(r"\p{Script=%s}" % s) for s in SCRIPTS}
    for cp in late:
        c = chr(cp)
        hit = [g for g in two_letter if rx_gc[g].fullmatch(c)]
        assert len(hit) == 1, hex(cp)
        gc[cp] = hit[0]
        for p in prop_id:
            if rx_prop[p].fullmatch(c):
                props[p][cp] = 1
        for s in SCRIPTS:
            if rx_script[s].fullmatch(c):
                scripts[s][cp] = 1

    tables = {}
    for g in GCS:
        tables["gc_" + g] = ranges_of([x is not None and x.startswith(g) for x in gc])
    for p in PROPS:
        tables["prop_" + p] = ranges_of(props[p])
    for s in SCRIPTS:
        tables["script_" + s] = ranges_of(scripts[s])
    # the Perl classes as regex-syntax defines them: \d = Nd, \s = White_Space, \w = Alphabetic + M + Nd + Pc + Join_Control
    tables["perl_digit"] = ranges_of([x == "Nd" for x in gc])
    tables["perl_space"] = ranges_of(props["White_Space"])
    tables["perl_word"] = ranges_of([bool(props["Alphabetic"][cp] or props["Join_Control"][cp] or
                                          (gc[cp] is not None and (gc[cp][0] == "M" or gc[cp] in ("Nd", "Pc"))))
                                     for cp in range(MAXCP + 1)])

    # simple case folding: code points with one fold are one orbit
    fold = {}
    for cp in range(MAXCP + 1):
        if is_scalar(cp):
            f = icu.fold(cp, 0)  # (U_FOLD_CASE_DEFAULT = 0: no Turkic mappings; simple = C + S)
            if f != cp:
                fold[cp] = f
    for code, status, mapping in UV.CASEFOLDING_15_1 + UV.CASEFOLDING_16_0:
        assert status in ("C", "S") and code not in fold, hex(code)
        fold[code] = mapping
    classes = {}
    for cp, f in fold.items():
        assert f not in fold, hex(f)  # (a fold is folded)
        classes.setdefault(f, {f}).add(cp)
    fold_pairs = sorted((a, b) for members in classes.values() for a in members for b in members if a != b)

    names = sorted(tables)
    lines = ["// GENERATED by oracle/gen_unicode_oracle.py -- do not edit.  Test infrastructure (the oracle's tables).",
             "// Classes: ICU 70 (Unicode 14.0) + the additions of 15.0 - 16.0 as the `regex` module %s has them;" % regex.__version__,
             "// fold pairs: ICU 70 u_foldCase (simple, default) + the C / S lines of CaseFolding-15.1.0 and -16.0.0.",
             "#pragma once", "#include <stdint.h>",
             "typedef struct { uint32_t lo, hi; } tgx_urange;",
             "typedef struct { const char *name; const tgx_urange *ranges; uint32_t count; } tgx_utable;"]
    for n in names:
        lines.append("static const tgx_urange tgx_ur_%s[] = {%s};" %
                     (n, ",".join("{0x%X,0x%X}" % r for r in tables[n])))
    lines.append("static const tgx_utable tgx_utables[] = {")
    for n in names:
        lines.append('  {"%s", tgx_ur_%s, %d},' % (n, n, len(tables[n])))
    lines.append("};")
    lines.append("static const uint32_t tgx_n_utables = %d;" % len(names))
    lines.append("// simple case folding: every (code point, equivalent code point) pair, sorted by the first")
    lines.append("static const uint32_t tgx_fold_pairs[][2] = {%s};" % ",".join("{0x%X,0x%X}" % p for p in fold_pairs))
    lines.append("static const uint32_t tgx_n_fold_pairs = %d;" % len(fold_pairs))
    path = os.path.join(HERE, "unicode_oracle_tables.h")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", path, os.path.getsize(path), "bytes,", len(fold_pairs), "fold pairs")


if __name__ == "__main__":
    main()
