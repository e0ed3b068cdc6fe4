"""Row-range sharding across ranks: the merge step.

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  Every rank scans its own row range; then
  * exact DISTINCT columns exchange their key sets once (hash-owner all-to-all of fixed-size records), after
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
    send = torch.zeros(world, row, dtype=torch.int32, device="cuda")
    col = 0
    for si, base, n_words, ptr, sw, _ in parts:
        src = torch.as_tensor(_DevPtr(ptr, n_words * 4), device="cuda").view(torch.int32)
        full = n_words // sw  # whole slices present in the bitmap
        if full:
            send[:full, col:col + sw] = src[: full * sw].view(full, sw)
        rest = n_words - full * sw
        if rest:
            send[full, col:col + rest] = src[full * sw:]
        col += sw
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv.view(-1), send.view(-1))
    col = 0
    slices = {}
    for si, base, n_words, ptr, sw, is_twice in parts:
        slices.setdefault(si, {})["twice" if is_twice else "seen"] = recv[:, col:col + sw].contiguous()
        slices[si]["base"], slices[si]["sw"] = base, sw
        col += sw
    torch.cuda.synchronize()
    for si, d in slices.items():
        tw = d.get("twice")
        state.distinct_adopt_slices(si, d["base"] + rank * d["sw"] * 32, d["seen"].data_ptr(),
                                    tw.data_ptr() if tw is not None else None, world, d["sw"])


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


def allgather_blobs(blob, dist, world, device="cpu"):
    """all-gather one byte string per rank.  Blobs of owner-partitioned states have the same size on every rank,
    so the common case is ONE collective: the payload travels with an 8-byte length prefix in a buffer of the
    local size; only if the sizes turn out to differ is a second, padded round needed."""
    import struct

    import torch

    payload = struct.pack("<Q", len(blob)) + bytes(blob)
    mine = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([len(payload)], dtype=torch.int64, device=device))
    sizes = [int(x.item()) for x in sizes]
    mx = max(sizes)
    if mx != len(payload):
        mine = torch.cat([mine, torch.zeros(mx - len(payload), dtype=torch.uint8, device=device)])
    gathered = [torch.empty(mx, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(gathered, mine)
    out = []
    for r in range(world):
        raw = bytes(gathered[r].cpu().numpy())
        (n,) = struct.unpack("<Q", raw[:8])
        out.append(raw[8:8 + n])
    return out


def allgather_many(blobs, dist, world, device="cpu"):
    """several blobs per rank (e.g. the stats state and the distinct state) in the collectives of one"""
    import struct

    packed = b"".join(struct.pack("<Q", len(b)) + bytes(b) for b in blobs)
    out = []
    for raw in allgather_blobs(packed, dist, world, device):
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
