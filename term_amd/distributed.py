"""Row-range sharding across ranks: thin callers of the C entry point.

The whole cross-rank step is `tgx_allreduce(plan, state, comm)` in libtgx (term_amd/csrc/allreduce.cpp): range
agreement, ONE all-to-all of re-based range-bitmap slices for the dense Int64 DISTINCT columns (hash-owner key
records otherwise), one all-gather of the packed partial states and the rank-ordered merge -- the cross-rank form of
`AnalyzerState::merge` (analyzers/traits.rs:160-170).  This module only picks the transport:

  rccl_comm(dist)        RCCL over xGMI; the library owns the communicator (torch.distributed carries the 128-byte
                         unique id to the other ranks, nothing else)
  torch_dist_comm(dist)  any torch.distributed backend through host buffers (the gloo world-2 CPU test)
  ThreadComm             N ranks as threads of one process sharing one GPU (tests/, tools/sim_bench_ranks.py)
"""
import ctypes as C
import threading

import term_amd as T
from term_amd._lib import Comm


def shard_rows(n_total, world, rank, align=64):
    """contiguous row range of `rank`, boundaries aligned to `align` rows (validity words stay whole)"""
    per = (n_total // world) // align * align
    lo = rank * per
    hi = n_total if rank == world - 1 else lo + per
    return lo, hi


def sharded_suite_step(plan, state, columns, comm):
    """One step of a row-sharded suite on this rank (what bench.py times for N > 1): scan of the local shard with
    the fused plan, then the cross-rank step; every rank returns the results of the whole table."""
    state.reset()
    state.update(columns)
    state.allreduce(comm)
    return state.finalize()


def shared_fingerprint_key(dist, rank, device=None):
    """One fingerprint key for all ranks (tgx_plan_set_fingerprint_key): string / tuple keys travel between ranks as
    fingerprints, which mean the same everywhere only under one key -- tgx_allreduce refuses ranks whose keys differ.
    Rank 0 draws 16 bytes from the operating system, torch.distributed broadcasts them; pass the result as
    `T.Plan(specs, fingerprint_key=...)` on every rank.  (Plans without string / tuple DISTINCT checks need none.)"""
    import os

    import torch

    key = torch.zeros(16, dtype=torch.uint8)
    if rank == 0:
        key = torch.frombuffer(bytearray(os.urandom(16)), dtype=torch.uint8).clone()
    if device is not None:
        key = key.to(device)
    dist.broadcast(key, src=0)
    return bytes(key.cpu().numpy())


# ---------------------------------------------------------------------------------------------- RCCL
def rccl_comm(dist, rank, world, device="cuda"):
    """tgx_comm over RCCL: rank 0 draws the unique id, torch.distributed broadcasts it, every rank joins"""
    import torch

    uid = torch.zeros(T._lib.RCCL_UNIQUE_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        uid = torch.frombuffer(bytearray(Comm.rccl_unique_id()), dtype=torch.uint8).clone()
    uid = uid.to(device)
    dist.broadcast(uid, src=0)
    return Comm.rccl(bytes(uid.cpu().numpy()), rank, world)


# ---------------------------------------------------------------------------------------------- host-buffer transports
def _view(ptr, nbytes):
    """raw HOST pointer -> numpy uint8 view (no copy)"""
    import numpy as np

    if nbytes == 0:
        return np.zeros(0, dtype=np.uint8)
    return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr))


def torch_dist_comm(dist, rank, world):
    """tgx_comm over any torch.distributed process group, through HOST buffers (gloo has no all-to-all: every
    collective here is an all_gather of equal-sized blocks, which is all the CPU test needs)"""
    import numpy as np
    import torch

    def gather_blocks(block):
        out = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(out, block)
        return out

    def alltoall(send, recv, per_peer, _stream):
        mine = torch.from_numpy(_view(send, per_peer * world).copy())
        dst = _view(recv, per_peer * world)
        for r, blk in enumerate(gather_blocks(mine)):
            dst[r * per_peer:(r + 1) * per_peer] = blk.numpy()[rank * per_peer:(rank + 1) * per_peer]

    def alltoallv(send, send_counts, recv, recv_counts, elem, _stream):
        sc = [int(send_counts[r]) * elem for r in range(world)]
        rc = [int(recv_counts[r]) * elem for r in range(world)]
        sizes = gather_blocks(torch.tensor(sc, dtype=torch.int64))
        cap = max(1, max(int(t.sum()) for t in sizes))
        mine = torch.zeros(cap, dtype=torch.uint8)
        mine[: sum(sc)] = torch.from_numpy(_view(send, sum(sc)).copy())
        dst = _view(recv, sum(rc))
        pos = 0
        for r, blk in enumerate(gather_blocks(mine)):
            start = int(sizes[r][:rank].sum())
            assert int(sizes[r][rank]) == rc[r]
            dst[pos:pos + rc[r]] = blk.numpy()[start:start + rc[r]]
            pos += rc[r]

    def allgather(send, recv, nbytes, _stream):
        mine = torch.from_numpy(_view(send, nbytes).copy())
        dst = _view(recv, nbytes * world)
        for r, blk in enumerate(gather_blocks(mine)):
            dst[r * nbytes:(r + 1) * nbytes] = blk.numpy()

    del np
    return Comm.custom(rank, world, alltoall, alltoallv, allgather, device_buffers=False)


class ThreadGroup:
    """what N threaded ranks share: a barrier and one mailbox slot per rank"""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


def thread_comm(group, rank, device_buffers=False):
    """tgx_comm between threads of one process (a stand-in for RCCL with the same call pattern and data movement).
    device_buffers=True hands the callbacks DEVICE pointers, as RCCL gets them: the ranks share one GPU, so a peer's
    buffer is copied with a device-to-device memcpy once every rank has published its pointers."""
    world = group.world

    def exchange(item):
        group.slots[rank] = item
        group.barrier.wait()
        seen = list(group.slots)
        group.barrier.wait()
        return seen

    if device_buffers:
        import torch

        class _DevPtr:
            def __init__(self, ptr, nbytes):
                self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

        def dview(ptr, nbytes):
            return torch.as_tensor(_DevPtr(ptr, nbytes), device="cuda") if nbytes else torch.empty(0, dtype=torch.uint8, device="cuda")

        def fence():
            torch.cuda.synchronize()  # the library's stream is not torch's: drain everything before peers read

        def alltoall(send, recv, per_peer, _stream):
            fence()
            peers = exchange(send)
            dst = dview(recv, per_peer * world)
            for r in range(world):
                dst[r * per_peer:(r + 1) * per_peer] = dview(peers[r], per_peer * world)[rank * per_peer:(rank + 1) * per_peer]
            fence()
            group.barrier.wait()  # nobody reuses its send buffer before every peer has read it

        def alltoallv(send, send_counts, recv, recv_counts, elem, _stream):
            fence()
            sc = [int(send_counts[r]) * elem for r in range(world)]
            rc = [int(recv_counts[r]) * elem for r in range(world)]
            peers = exchange((send, sc))
            dst = dview(recv, sum(rc))
            pos = 0
            for r in range(world):
                p, psc = peers[r]
                start = sum(psc[:rank])
                assert psc[rank] == rc[r]
                if rc[r]:
                    dst[pos:pos + rc[r]] = dview(p, sum(psc))[start:start + rc[r]]
                pos += rc[r]
            fence()
            group.barrier.wait()

        def allgather(send, recv, nbytes, _stream):
            fence()
            peers = exchange(send)
            dst = dview(recv, nbytes * world)
            for r in range(world):
                dst[r * nbytes:(r + 1) * nbytes] = dview(peers[r], nbytes)
            fence()
            group.barrier.wait()
    else:
        def alltoall(send, recv, per_peer, _stream):
            peers = exchange(bytes(_view(send, per_peer * world)))
            dst = _view(recv, per_peer * world)
            for r in range(world):
                dst[r * per_peer:(r + 1) * per_peer] = memoryview(peers[r])[rank * per_peer:(rank + 1) * per_peer]

        def alltoallv(send, send_counts, recv, recv_counts, elem, _stream):
            sc = [int(send_counts[r]) * elem for r in range(world)]
            rc = [int(recv_counts[r]) * elem for r in range(world)]
            peers = exchange((bytes(_view(send, sum(sc))), sc))
            dst = _view(recv, sum(rc))
            pos = 0
            for r in range(world):
                data, psc = peers[r]
                start = sum(psc[:rank])
                assert psc[rank] == rc[r]
                dst[pos:pos + rc[r]] = memoryview(data)[start:start + rc[r]]
                pos += rc[r]

        def allgather(send, recv, nbytes, _stream):
            peers = exchange(bytes(_view(send, nbytes)))
            dst = _view(recv, nbytes * world)
            for r in range(world):
                dst[r * nbytes:(r + 1) * nbytes] = memoryview(peers[r])

    return Comm.custom(rank, world, alltoall, alltoallv, allgather, device_buffers=device_buffers)
