"""Row-range sharding across ranks: the merge step.

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  Every rank scans its own row range; then
  * exact DISTINCT columns exchange their key sets once -- slices of congruent range bitmaps (all columns in one
    all-to-all) where the agreed value range is dense, fixed-size key records by hash owner otherwise -- after
    which each rank holds a disjoint part of the global key set, and
  * the packed partial states (a few KiB) are all-gathered and folded in rank order on every rank, so all
    ranks finish with the same result (`AnalyzerState::merge`, analyzers/traits.rs:160-170).
No other collective touches the data path.
"""
import term_amd as T


class _DevPtr:
    """raw device pointer -> torch tensor view through __cuda_array_interface__ (no copy)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def shard_rows(n_total, world, rank, align=64):
    """contiguous row range of `rank`, boundaries aligned to `align` rows (validity words stay whole)"""
    per = (n_total // world) // align * align
    lo = rank * per
    hi = n_total if rank == world - 1 else lo + per
    return lo, hi


def exchange_distinct(state, spec_indices, dist, world):
    """hash-owner all-to-all of the local key sets, then import the owned keys (device tensors, RCCL)"""
    import torch

    for si in spec_indices:
        rec = state.distinct_record_bytes(si)
        words = rec // 8
        ptr, counts = state.distinct_export(si, world)
        total = sum(counts)
        send = torch.as_tensor(_DevPtr(ptr, max(total, 1) * rec), device="cuda").view(torch.int64)[: total * words]
        send_counts = torch.tensor(counts, dtype=torch.int64, device="cuda")
        recv_counts = torch.empty(world, dtype=torch.int64, device="cuda")
        dist.all_to_all_single(recv_counts, send_counts)
        rc = recv_counts.tolist()
        recv = torch.empty(sum(rc) * words, dtype=torch.int64, device="cuda")
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=[c * words for c in rc],
                               input_split_sizes=[c * words for c in counts])
        torch.cuda.synchronize()
        state.distinct_import(si, recv.data_ptr(), sum(rc))


def exchange_distinct_bitmaps(state, spec_indices, dist, world, rank):
    """Range-bitmap form of the exchange for one or more DISTINCT columns in ONE all-to-all: all ranks hold
    congruent bitmaps (tgx_distinct_range_hint); each sends slice r of every column to rank r (equal splits),
    ORs what it receives and keeps the owned slices.  ~range/8 bytes per rank and column instead of 16 bytes per
    key.  Raises TgxError(TGX_UNSUPPORTED) when a set is a hash table (nothing has been exchanged then)."""
    import torch

    if isinstance(spec_indices, int):
        spec_indices = [spec_indices]
    views = [state.distinct_bitmap_view(si) for si in spec_indices]  # raises before any communication
    parts = []  # (spec index, base, n_words, ptr, slice_words, is_twice)
    for si, (base, n_words, seen_ptr, twice_ptr) in zip(spec_indices, views):
        sw = ((n_words + world - 1) // world + 3) // 4 * 4
        parts.append((si, base, n_words, seen_ptr, sw, False))
        if twice_ptr:
            parts.append((si, base, n_words, twice_ptr, sw, True))
    row = sum(p[4] for p in parts)  # words every rank sends to every other rank
    # everything below is ordered on torch's CURRENT stream only (no device-wide synchronize), so a caller can
    # keep an unrelated scan running on another stream while the slices travel
    send = torch.empty(world, row, dtype=torch.int32, device="cuda")
    col = 0
    for si, base, n_words, ptr, sw, _ in parts:
        src = torch.as_tensor(_DevPtr(ptr, n_words * 4), device="cuda").view(torch.int32)
        full = n_words // sw  # whole slices present in the bitmap
        if full:
            send[:full, col:col + sw] = src[: full * sw].view(full, sw)
        rest = n_words - full * sw
        if full < world:
            send[full:, col:col + sw] = 0  # the padding past the end of the bitmap
        if rest:
            send[full, col:col + rest] = src[full * sw:]
        col += sw
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv.view(-1), send.view(-1))
    torch.cuda.current_stream().synchronize()
    col = 0
    done = set()
    for si, base, n_words, ptr, sw, is_twice in parts:
        if si in done:
            continue
        done.add(si)
        seen_ptr = recv.data_ptr() + 4 * col
        has_twice = any(p[0] == si and p[5] for p in parts)
        twice_ptr = recv.data_ptr() + 4 * (col + sw) if has_twice else None
        # the received slices are used in place: slice of peer i at word offset i * row
        state.distinct_adopt_slices(si, base + rank * sw * 32, seen_ptr, twice_ptr, world, sw, row)
        col += sw * (2 if has_twice else 1)
    # (tgx_distinct_adopt_slices returns after its kernel has finished: `recv` may be freed now)


def exchange_distinct_auto(state, spec_indices, dist, world, rank):
    """bitmap slices (one all-to-all for all columns) where the key sets are range bitmaps, 16/32-byte key
    records otherwise"""
    try:
        exchange_distinct_bitmaps(state, list(spec_indices), dist, world, rank)
        return
    except T.TgxError as e:
        if e.status != "TGX_UNSUPPORTED":
            raise
    for si in spec_indices:
        try:
            exchange_distinct_bitmaps(state, [si], dist, world, rank)
        except T.TgxError as e:
            if e.status != "TGX_UNSUPPORTED":
                raise
            exchange_distinct(state, [si], dist, world)


def agree_on_ranges(local_minmax, dist, world, device="cuda"):
    """local_minmax: list of (has_value, min, max) per DISTINCT column -> list of global (lo, hi) or None.
    One all-gather of 2 x columns int64 values."""
    import torch

    i64max, i64min = (1 << 63) - 1, -(1 << 63)
    mine = []
    for has, lo, hi in local_minmax:
        mine += [lo if has else i64max, hi if has else i64min]
    t = torch.tensor(mine, dtype=torch.int64, device=device)
    out = torch.empty(world * len(mine), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, t)
    allv = out.view(world, len(local_minmax), 2).cpu()
    res = []
    for c in range(len(local_minmax)):
        lo, hi = int(allv[:, c, 0].min()), int(allv[:, c, 1].max())
        res.append((lo, hi) if lo <= hi else None)
    return res


_GATHER_CAPACITY = {}  # cache_key -> bytes per rank agreed for the one-collective fast path


def allgather_blobs(blob, dist, world, device="cpu", cache_key=None):
    """all-gather one byte string per rank.

    Every rank sends a 16-byte header (payload length, capacity it needs) + payload in a buffer of an agreed
    capacity: ONE all_gather_into_tensor, one host->device and one device->host copy.  The capacity is agreed by a
    size round the first time (and remembered under `cache_key`, e.g. the plan, when given); if any rank's payload
    has outgrown it, every rank sees that in the headers and all repeat the round with the larger capacity."""
    import struct

    import torch

    blob = bytes(blob)

    def agree():
        sizes = torch.empty(world, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(sizes, torch.tensor([len(blob)], dtype=torch.int64, device=device))
        need = int(sizes.max().item())
        return (need + need // 2 + 16 + 255) // 256 * 256

    cap = _GATHER_CAPACITY.get(cache_key) if cache_key is not None else None
    if cap is None:
        cap = agree()
    while True:
        fits = len(blob) + 16 <= cap
        buf = bytearray(cap)
        struct.pack_into("<QQ", buf, 0, len(blob) if fits else 0, len(blob) + 16)
        if fits:
            buf[16:16 + len(blob)] = blob
        mine = torch.frombuffer(buf, dtype=torch.uint8).to(device)
        out = torch.empty(world * cap, dtype=torch.uint8, device=device)
        dist.all_gather_into_tensor(out, mine)
        raw = bytes(out.cpu().numpy())
        heads = [struct.unpack_from("<QQ", raw, r * cap) for r in range(world)]
        need = max(h[1] for h in heads)
        if need <= cap:
            if cache_key is not None:
                _GATHER_CAPACITY[cache_key] = cap
            return [raw[r * cap + 16: r * cap + 16 + heads[r][0]] for r in range(world)]
        cap = (need + need // 2 + 255) // 256 * 256  # the same on every rank: all saw the same headers


def allgather_many(blobs, dist, world, device="cpu", cache_key=None):
    """several blobs per rank (e.g. the stats state and the distinct state) in the collectives of one"""
    import struct

    packed = b"".join(struct.pack("<Q", len(b)) + bytes(b) for b in blobs)
    out = []
    for raw in allgather_blobs(packed, dist, world, device, cache_key):
        parts, pos = [], 0
        for _ in blobs:
            (n,) = struct.unpack("<Q", raw[pos:pos + 8])
            parts.append(raw[pos + 8:pos + 8 + n])
            pos += 8 + n
        out.append(parts)
    return out


def merge_blobs(plan, blobs):
    """fold partial states in rank order; the same on every rank"""
    merged = None
    for blob in blobs:
        part = T.State.deserialize(plan, blob)
        if merged is None:
            merged = part
        else:
            merged.merge([part])
    return merged


def allgather_merge(plan, state_or_blob, dist, world, device="cpu"):
    blob = state_or_blob if isinstance(state_or_blob, (bytes, bytearray)) else state_or_blob.serialize()
    return merge_blobs(plan, allgather_blobs(blob, dist, world, device))


def sharded_suite_step(plan, state, plan_d, state_d, stat_specs, distinct_specs, columns, dist, world, rank,
                       cache_key=None):
    """One step of a row-sharded suite on this rank (what bench.py times for N > 1):

        scan of the local shard -> ranks agree on the DISTINCT columns' global value ranges -> congruent range
        bitmaps (tgx_distinct_range_hint) -> one all-to-all of bitmap slices (hash-owner key exchange where a set
        is not a bitmap) -> one all-gather of the packed partial states -> identical rank-ordered merge everywhere.

    `plan` / `state` hold the additive checks (`stat_specs`), `plan_d` / `state_d` the DISTINCT checks
    (`distinct_specs`); returns the merged results of both, in that order."""
    state.reset()
    state.update(columns)
    local = state.finalize()
    state_d.reset()
    minmax = []
    for s in distinct_specs:
        r = next((x for sp, x in zip(stat_specs, local) if sp.kind == T.NUMERIC_STATS and sp.column == s.column), None)
        minmax.append((bool(r.has_value) and not r.is_float, r.min_i, r.max_i) if r is not None else (False, 0, 0))
    for j, rng in enumerate(agree_on_ranges(minmax, dist, world)):
        if rng is not None:
            state_d.distinct_range_hint(j, rng[0], rng[1])
    state_d.update(columns)
    exchange_distinct_auto(state_d, list(range(len(distinct_specs))), dist, world, rank)
    per_rank = allgather_many([state.serialize(), state_d.serialize()], dist, world, device="cuda", cache_key=cache_key)
    merged = merge_blobs(plan, [p[0] for p in per_rank])
    merged_d = merge_blobs(plan_d, [p[1] for p in per_rank])
    return merged.finalize() + merged_d.finalize()

