"""Test infrastructure: a Python statement of the library's keyed fingerprint (term_amd/csrc/kernels/distinct128.hip:
Chaskey-8 over the value's 16-byte blocks), and -- because the function is keyed, not collision-free for whoever HOLDS the
key -- a generator of distinct values with equal fingerprints under a given key.  The GPU tests hand a plan a known key,
feed such pairs, and expect a fingerprint set to count them once (the deviation, stated) and an EXACT set
(TGX_FLAG_EXACT_KEYS) to count them as the oracle does."""
import struct

M32 = 0xFFFFFFFF


def _rotl(x, r):
    return ((x << r) | (x >> (32 - r))) & M32


def _permute(v):
    v0, v1, v2, v3 = v
    for _ in range(8):
        v0 = (v0 + v1) & M32
        v1 = _rotl(v1, 5) ^ v0
        v0 = _rotl(v0, 16)
        v2 = (v2 + v3) & M32
        v3 = _rotl(v3, 8) ^ v2
        v0 = (v0 + v3) & M32
        v3 = _rotl(v3, 13) ^ v0
        v2 = (v2 + v1) & M32
        v1 = _rotl(v1, 7) ^ v2
        v2 = _rotl(v2, 16)
    return [v0, v1, v2, v3]


def _times_two(k):
    carry = 0x87 if k[3] >> 31 else 0
    return [((k[0] << 1) & M32) ^ carry, ((k[1] << 1) | (k[0] >> 31)) & M32, ((k[2] << 1) | (k[1] >> 31)) & M32,
            ((k[3] << 1) | (k[2] >> 31)) & M32]


def subkeys(key):
    k = list(struct.unpack("<4I", key))
    k1 = _times_two(k)
    return k, k1, _times_two(k1)


def state_after(key, blocks):
    """the state after the full 16-byte `blocks` (none of them the value's last)"""
    k, _, _ = subkeys(key)
    v = list(k)
    for b in blocks:
        m = struct.unpack("<4I", b)
        v = _permute([a ^ x for a, x in zip(v, m)])
    return v


def fingerprint(key, value):
    """(fa, fb) as the kernels compute them (before the table's free-slot marker is stepped aside)"""
    value = bytes(value)
    k, k1, k2 = subkeys(key)
    n_full = max(0, (len(value) - 1) // 16)  # every block but the last
    v = state_after(key, [value[16 * i:16 * i + 16] for i in range(n_full)])
    last = value[16 * n_full:]
    if len(last) == 16:
        lk = k1
    else:
        last = last + b"\x01" + b"\x00" * (15 - len(last))
        lk = k2
    m = struct.unpack("<4I", last)
    v = _permute([a ^ x ^ y for a, x, y in zip(v, m, lk)])
    v = [a ^ y for a, y in zip(v, lk)]
    fa, fb = v[0] | (v[1] << 32), v[2] | (v[3] << 32)
    if fa == 0xFFFFFFFFFFFFFFFF:
        fa -= 1
    if fb == 0xFFFFFFFFFFFFFFFF:
        fb -= 1
    return fa, fb


def colliding_pair(key, rng, tail_len=16, alphabet=None):
    """two distinct values of 16 + tail_len bytes (1 <= tail_len <= 16) with ONE fingerprint under `key`: first blocks
    m, m' chosen freely; the second blocks differ by the XOR of the two states, so the states meet again before the last
    permutation.  With tail_len < 16 the difference must vanish on the padded bytes: the first blocks are searched until
    it does on the last 16 - tail_len bytes (2^(8 * (16 - tail_len)) tries: keep tail_len >= 14)."""
    def draw(n):
        if alphabet is None:
            return bytes(int(x) for x in rng.integers(0, 256, size=n))
        return bytes(alphabet[int(x)] for x in rng.integers(0, len(alphabet), size=n))
    while True:
        m1, m2 = draw(16), draw(16)
        if m1 == m2:
            continue
        d = [a ^ b for a, b in zip(state_after(key, [m1]), state_after(key, [m2]))]
        delta = struct.pack("<4I", *d)
        if any(delta[tail_len:]):
            continue
        t1 = draw(tail_len)
        t2 = bytes(a ^ b for a, b in zip(t1, delta[:tail_len]))
        a, b = m1 + t1, m2 + t2
        assert a != b and fingerprint(key, a) == fingerprint(key, b)
        return a, b


def tuple_fingerprint(key, components):
    """components: None | int (the 64 bits of an Int64 / Float64) | bytes"""
    blocks = []
    for c in components:
        if c is None:
            ca, cb = 0x4E554C4C4E554C4C, 0
        elif isinstance(c, int):
            ca, cb = c & 0xFFFFFFFFFFFFFFFF, 1
        else:
            ca, cb = fingerprint(key, c)
            cb |= 2
        blocks.append(struct.pack("<2Q", ca, cb))
    # the message is the component blocks: its last block is the last component's, a full one (K1)
    k, k1, _ = subkeys(key)
    v = state_after(key, blocks[:-1])
    m = struct.unpack("<4I", blocks[-1])
    v = _permute([a ^ x ^ y for a, x, y in zip(v, m, k1)])
    v = [a ^ y for a, y in zip(v, k1)]
    return v[0] | (v[1] << 32), v[2] | (v[3] << 32)


# (No crafted pair of Int64 TUPLES: a numeric component is the block (value, 1) -- its high half is fixed, so steering
#  the states together again needs their difference to vanish on 8 chosen bytes, a 2^64 search even with the key.)


def colliding_string_tuples(key, rng):
    """two distinct tuples (Int64, Utf8) with one tuple fingerprint: the string components are a colliding pair, so the
    component blocks (fa, fb | 2) are equal and everything after them is too"""
    a, b = colliding_pair(key, rng)
    n = int(rng.integers(0, 1 << 40))
    return (n, a), (n, b)
