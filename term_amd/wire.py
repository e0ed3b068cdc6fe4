"""Wire form of a tgx state (tgx_state_serialize / tgx_state_deserialize), version 3.

The blob is what ranks exchange (one all-gather of a few KiB) and what a checkpoint stores; it is the
counterpart of the serde_json analyzer states of the reference's IncrementalAnalysisRunner
(analyzers/incremental/runner.rs:71-111).  Layout, little-endian, in plan-task order:

    u32 magic 'TGXS', u32 version, u32 n_scan, n_count, n_comoments, n_distinct, n_kll, n_regex, n_hll
    u32 keyed, u8 key[16]      keyed = 1: the blob holds string / tuple keys as fingerprints made under `key`
                               (tgx_plan_set_fingerprint_key); 0: no such keys, key = zeros
    n_scan      x ScanAcc      i64 total, non_null, min_key, max_key; u64 sum_lo; i64 sum_hi; f64 sum, comp;
                               i64 var_n; f64 var_mean, var_m2; i32 is_float, pad                    (96 B)
    n_count     x CountAcc     i64 total, non_null                                                      (16 B)
    n_comoments x ComomentAcc  i64 total, n; f64 s[5]; f64 c[5]; f64 px, py; i32 pivot_set, pad          (120 B)
                               s + c = sums of x', y', x'x', y'y', x'y' with x' = x - px, y' = y - py: about the
                               pivots (px, py); (0, 0) makes them the raw sum_x, sum_y, sum_x2, sum_y2, sum_xy
    n_distinct  x { u32 owner_partitioned, u32 wide_keys; u64 total, non_null, distinct, twice, empty_rows;
                    u64 n_records; records (16 B {key, count} or 32 B {hash_a, hash_b, count, 0}) }
    n_kll       x { u32 k, u32 n_levels; u64 n; f64 min, max; n_levels x { u32 count; f64 items[count] } }
    n_regex     x { u64 total, u64 matches }
    n_hll       x { u32 mode (0 nothing seen, 1 registers, 2 the exact key set answers), u32 has_registers;
                    has_registers x 16384 u8 HyperLogLog registers (rank 0 .. 33) }

min_key / max_key are the Int64 values themselves, or the IEEE totalOrder keys of Float64 values
(bits ^ ((bits >> 63) >>> 1)).  This module packs partial states from plain numbers; libtgx does the parsing.
"""
import struct

MAGIC, VERSION = 0x53584754, 3
I64_MAX, I64_MIN = (1 << 63) - 1, -(1 << 63)


def f64_total_key(x):
    bits = struct.unpack("<q", struct.pack("<d", x))[0]
    return bits ^ ((bits >> 63) & 0x7FFFFFFFFFFFFFFF)


def scan_acc(total, non_null, minimum=None, maximum=None, total_sum=0, is_float=False, var=None):
    """var = (n, mean, m2) or None.  total_sum: exact int for Int64 columns, float for Float64 columns."""
    if non_null == 0 or minimum is None:
        mn, mx = I64_MAX, I64_MIN
    elif is_float:
        mn, mx = f64_total_key(float(minimum)), f64_total_key(float(maximum))
    else:
        mn, mx = int(minimum), int(maximum)
    if is_float:
        lo, hi, s = 0, 0, float(total_sum)
    else:
        v = int(total_sum) & ((1 << 128) - 1)
        lo, hi = v & ((1 << 64) - 1), v >> 64
        hi = hi - (1 << 64) if hi >= (1 << 63) else hi
        s = 0.0
    vn, vmean, vm2 = var if var else (0, 0.0, 0.0)
    return struct.pack("<qqqqQqddqddii", total, non_null, mn, mx, lo, hi, s, 0.0, vn, vmean, vm2,
                       1 if is_float else 0, 0)


def count_acc(total, non_null):
    return struct.pack("<qq", total, non_null)


def comoment_acc(total, n, sum_x, sum_y, sum_x2, sum_y2, sum_xy, px=0.0, py=0.0):
    """the five sums are taken about the pivots (px, py); the default (0, 0) means raw sums"""
    return struct.pack("<qq5d5dddii", total, n, sum_x, sum_y, sum_x2, sum_y2, sum_xy, 0.0, 0.0, 0.0, 0.0, 0.0,
                       px, py, 1, 0)


def distinct_counts(total, non_null, distinct, twice=0):
    """an owner-partitioned partial: this rank's keys are disjoint from every other rank's"""
    return struct.pack("<II5QQ", 1, 0, total, non_null, distinct, twice, 0, 0)


def kll_state(k, n, minimum, maximum, levels):
    out = struct.pack("<IIQdd", k, len(levels), n, minimum, maximum)
    for items in levels:
        out += struct.pack("<I", len(items)) + struct.pack("<%dd" % len(items), *items)
    return out


def regex_counts(total, matches):
    return struct.pack("<QQ", total, matches)


def hll_state(registers=None, mode=1):
    """registers: 16384 bytes (or None: nothing seen yet)"""
    if registers is None:
        return struct.pack("<II", mode, 0)
    assert len(registers) == 16384
    return struct.pack("<II", mode, 1) + bytes(registers)


def pack(scan=(), count=(), comoments=(), distinct=(), kll=(), regex=(), hll=()):
    head = struct.pack("<9I", MAGIC, VERSION, len(scan), len(count), len(comoments), len(distinct), len(kll), len(regex),
                       len(hll))
    head += struct.pack("<I16s", 0, bytes(16))  # (no string keys in a state packed from plain numbers)
    return (head + b"".join(scan) + b"".join(count) + b"".join(comoments) + b"".join(distinct) + b"".join(kll) +
            b"".join(regex) + b"".join(hll))
