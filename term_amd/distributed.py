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


def exchange_distinct_bitmaps(state, spec_index, dist, world, rank):
    """Range-bitmap form of the exchange: all ranks hold congruent bitmaps (tgx_distinct_range_hint); each sends
    slice r to rank r (equal splits), ORs what it receives and keeps the owned slice.  ~range/8 bytes per rank
    instead of 16 bytes per key.  Raises TgxError(TGX_UNSUPPORTED) when the set is a hash table."""
    import torch

    base, n_words, seen_ptr, twice_ptr = state.distinct_bitmap_view(spec_index)
    slice_words = ((n_words + world - 1) // world + 3) // 4 * 4
    padded = slice_words * world

    def swap(ptr):
        src = torch.as_tensor(_DevPtr(ptr, n_words * 4), device="cuda").view(torch.int32)
        send = torch.zeros(padded, dtype=torch.int32, device="cuda")
        send[:n_words] = src
        recv = torch.empty(padded, dtype=torch.int32, device="cuda")
        dist.all_to_all_single(recv, send)
        return recv

    recv_seen = swap(seen_ptr)
    recv_twice = swap(twice_ptr) if twice_ptr else None
    torch.cuda.synchronize()
    state.distinct_adopt_slices(spec_index, base + rank * slice_words * 32, recv_seen.data_ptr(),
                                recv_twice.data_ptr() if recv_twice is not None else None, world, slice_words)


def exchange_distinct_auto(state, spec_indices, dist, world, rank):
    """bitmap slices where the key set is a range bitmap, 16/32-byte key records otherwise"""
    for si in spec_indices:
        try:
            exchange_distinct_bitmaps(state, si, dist, world, rank)
        except T.TgxError as e:
            if e.status != "TGX_UNSUPPORTED":
                raise
            exchange_distinct(state, [si], dist, world)


def agree_on_ranges(local_minmax, dist, device="cuda"):
    """local_minmax: list of (has_value, min, max) per DISTINCT column -> list of global (lo, hi) or None"""
    import torch

    i64max, i64min = (1 << 63) - 1, -(1 << 63)
    los = torch.tensor([m[1] if m[0] else i64max for m in local_minmax], dtype=torch.int64, device=device)
    his = torch.tensor([m[2] if m[0] else i64min for m in local_minmax], dtype=torch.int64, device=device)
    dist.all_reduce(los, op=dist.ReduceOp.MIN)
    dist.all_reduce(his, op=dist.ReduceOp.MAX)
    out = []
    for lo, hi in zip(los.tolist(), his.tolist()):
        out.append((lo, hi) if lo <= hi else None)
    return out


def allgather_blobs(blob, dist, world, device="cpu"):
    """all-gather variable-size byte strings (sizes first, then padded payloads)"""
    import torch

    n = torch.tensor([len(blob)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    mine = torch.zeros(mx, dtype=torch.uint8, device=device)
    mine[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    gathered = [torch.empty(mx, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(gathered, mine)
    return [bytes(gathered[r][: sizes[r]].cpu().numpy()) for r in range(world)]


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
