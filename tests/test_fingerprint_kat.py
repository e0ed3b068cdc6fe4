"""The keyed fingerprint is Chaskey-8 as published (N. Mouha, B. Mennink, A. Van Herrewege, D. Watanabe, B. Preneel,
I. Verbauwhede: "Chaskey: An Efficient MAC Algorithm for 32-bit Microcontrollers", SAC 2014): the Python statement of
it the GPU tests hold the kernels to (tests/fp_reference.py; tests/test_gpu_exact_keys.py compares it with
`distinct128.hip` record by record) reproduces the known answers of the authors' reference implementation -- key
833D3433 009F389F 2398E64F 417ACF39 (little-endian words), messages m[i] = i of 0 .. 4 bytes, the tag as four
little-endian words.  CPU only: no kernel is called here."""
import struct

import fp_reference as R

KEY = struct.pack("<4I", 0x833D3433, 0x009F389F, 0x2398E64F, 0x417ACF39)
# (length of the message, the 128-bit tag's four words)
KNOWN = [
    (0, (0x792E8FE5, 0x75CE87AA, 0x2D1450B5, 0x1191970B)),
    (1, (0x13A9307B, 0x50E62C89, 0x4577BD88, 0xC0BBDC18)),
    (2, (0x55DF8922, 0x2C7FF577, 0x73809EF4, 0x4E5084C0)),
    (3, (0x1BDBB264, 0xA07680D8, 0x8E5B2AB8, 0x20660413)),
    (4, (0x30B2D171, 0xE38532FB, 0x16707C16, 0x73ED45F0)),
]
# (all five are one padded block under K2: what they pin is the permutation, its eight rounds, the subkeys and the
#  padding; blocks chained and a full last block under K1 are the construction on top, held by the tests below and by
#  the GPU tests' comparison of the kernels with tests/fp_reference.py over every length class)


def test_reference_reproduces_the_published_chaskey8_vectors():
    for n, want in KNOWN:
        fa, fb = R.fingerprint(KEY, bytes(range(n)))
        got = (fa & 0xFFFFFFFF, fa >> 32, fb & 0xFFFFFFFF, fb >> 32)
        assert got == want, (n, [hex(x) for x in got])


def test_subkeys_are_the_doublings_in_gf_2_128():
    # K1 = 2 K, K2 = 4 K with the reduction x^128 + x^7 + x^2 + x + 1 (0x87 into the low word when the top bit leaves)
    k, k1, k2 = R.subkeys(struct.pack("<4I", 0x80000000, 0, 0, 0x80000000))
    assert k1 == [0x00000087, 1, 0, 0]
    assert k2 == [0x0000010E, 2, 0, 0]


def test_a_full_last_block_and_a_padded_one_differ():
    # 16 bytes (K1, no padding) against the same 15 bytes + what the padding would append (0x01): not the same tag
    m15 = bytes(range(15))
    assert R.fingerprint(KEY, m15) != R.fingerprint(KEY, m15 + b"\x01")
