"""Which code points Unicode 17.0 added -- written down by hand, because nothing in this image knows Unicode 16.

The reference's pattern engine is Rust `regex` 1.12.2 = regex-syntax 0.8.8 (/root/reference/Cargo.lock:3637-3661), whose
tables are Unicode 16.0.  The databases this image holds are Unicode 13 (CPython 3.10 `unicodedata`), 14 (ICU 70, glibc
2.35) and 17 (PyPI `regex` 2026.7.19, fontTools, idna): the 16.0 repertoire is "what `regex` knows minus what 17.0
added".  The list below is that subtrahend.  How it was pinned: the code points `regex` has assigned and ICU 70 has not
are the additions of 15.0 + 15.1 + 16.0 + 17.0 = 4 489 + 627 + 5 185 + 4 803 = 15 104 (the published totals of the four
releases); sorted by block (fontTools' Blocks-17.0.0) and attributed to their release, every one of the four totals
closes exactly (tests/test_unicode_tables.py re-does the arithmetic for 17.0 where ICU 70 can be loaded).
"""

# whole blocks new in 17.0, then the additions 17.0 made to older blocks
UNICODE_17_ADDITIONS = [
    (0x10940, 0x10959),  # Sidetic
    (0x11DB0, 0x11DDB), (0x11DE0, 0x11DE9),  # Tolong Siki
    (0x16EA0, 0x16EB8), (0x16EBB, 0x16ED3),  # Beria Erfe (a cased script: 25 pairs of CaseFolding-17.0.0.txt)
    (0x1E6C0, 0x1E6DE), (0x1E6E0, 0x1E6F5), (0x1E6FE, 0x1E6FF),  # Tai Yo
    (0x323B0, 0x33479),  # CJK Unified Ideographs Extension J
    (0x18D80, 0x18DF2),  # Tangut Components Supplement
    (0x11B60, 0x11B67),  # Sharada Supplement
    (0x1CEC0, 0x1CED0), (0x1CEE0, 0x1CEF0),  # Miscellaneous Symbols Supplement
    (0x088F, 0x088F),  # Arabic Extended-B
    (0x0C5C, 0x0C5C),  # Telugu
    (0x0CDC, 0x0CDC),  # Kannada
    (0x1ACF, 0x1ADD), (0x1AE0, 0x1AEB),  # Combining Diacritical Marks Extended
    (0x20C1, 0x20C1),  # SAUDI RIYAL SIGN
    (0x2B96, 0x2B96),  # Miscellaneous Symbols and Arrows
    (0xA7CE, 0xA7CF), (0xA7D2, 0xA7D2), (0xA7D4, 0xA7D4), (0xA7F1, 0xA7F1),  # Latin Extended-D (A7D3 / A7D5 got capitals)
    (0xFBC3, 0xFBD2), (0xFD90, 0xFD91), (0xFDC8, 0xFDCE),  # Arabic Presentation Forms-A
    (0x10EC5, 0x10EC7), (0x10ED0, 0x10ED8), (0x10EFA, 0x10EFB),  # Arabic Extended-C
    (0x16FF2, 0x16FF6),  # Ideographic Symbols and Punctuation
    (0x187F8, 0x187FF),  # Tangut
    (0x18D09, 0x18D1E),  # Tangut Supplement
    (0x1CCFA, 0x1CCFC), (0x1CEBA, 0x1CEBF),  # Symbols for Legacy Computing Supplement
    (0x1F6D8, 0x1F6D8),  # Transport and Map Symbols
    (0x1F777, 0x1F77A),  # Alchemical Symbols
    (0x1F8D0, 0x1F8D8),  # Supplemental Arrows-C
    (0x1FA54, 0x1FA57),  # Chess Symbols
    (0x1FA8A, 0x1FA8A), (0x1FA8E, 0x1FA8E), (0x1FAC8, 0x1FAC8), (0x1FACD, 0x1FACD), (0x1FAEA, 0x1FAEA),
    (0x1FAEF, 0x1FAEF),  # Symbols and Pictographs Extended-A
    (0x1FBFA, 0x1FBFA),  # Symbols for Legacy Computing
    (0x2B73A, 0x2B73F),  # CJK Unified Ideographs Extension C
    (0x2CEA2, 0x2CEAD),  # CJK Unified Ideographs Extension E
]
UNICODE_17_COUNT = 4803

# CaseFolding.txt lines (status C or S) that 15.1 and 16.0 added to what Unicode 14 (ICU 70) folds: `code; status; mapping`
CASEFOLDING_15_1 = [(0x1FD3, "S", 0x0390), (0x1FE3, "S", 0x03B0), (0xFB05, "S", 0xFB06)]
CASEFOLDING_16_0 = ([(0x1C89, "C", 0x1C8A), (0xA7CB, "C", 0x0264), (0xA7CC, "C", 0xA7CD), (0xA7DA, "C", 0xA7DB),
                     (0xA7DC, "C", 0x019B)] +
                    [(0x10D50 + i, "C", 0x10D70 + i) for i in range(22)])  # Garay
# the two lines of status T (Turkic) Rust's simple folding leaves out: 0049; T; 0131 and 0130; T; 0069


def added_in_17(cp):
    for lo, hi in UNICODE_17_ADDITIONS:
        if lo <= cp <= hi:
            return True
    return False


def count_17():
    return sum(hi - lo + 1 for lo, hi in UNICODE_17_ADDITIONS)


assert count_17() == UNICODE_17_COUNT, count_17()
