"""Communicators for the partitioned multi-GPU run (include/abyss_amd.h, ``abg_comm``).

* ``RcclComm`` -- the product path: the library's own RCCL communicator
  (``abg_rccl_comm_create``), collectives enqueued on the engine's HIP stream over xGMI.  The
  128-byte unique id travels from rank 0 to the others through the torch.distributed process
  group that launched the ranks (one process per GPU).
* ``StagedTorchComm`` -- the same two collectives over ``torch.distributed`` with the buffers
  staged through host memory.  It lets the partitioned algorithm run where RCCL cannot: on CPU
  against tests/hostcheck (gloo, world_size 2) and with two ranks sharing one GPU (RCCL refuses
  duplicate devices).  Test infrastructure; ``bench.py --comm staged`` exposes it for diagnosis.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

U8, U32, U64 = 0, 1, 2
SUM, MAX, MIN = 0, 1, 2

AGV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p)
AR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p)
A2A_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64),
                     C.POINTER(C.c_uint64), C.c_void_p)


class CommStruct(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("stream_ordered", C.c_int32), ("struct_size", C.c_int32),
                ("user", C.c_void_p), ("all_gather_v", AGV_FN), ("all_reduce", AR_FN), ("all_to_all_v", A2A_FN)]


class RcclComm:
    """abg_rccl_comm_create over an id broadcast through torch.distributed."""

    def __init__(self, device: int, single: bool = False):
        """single: a communicator of one rank, no process group needed."""
        self._lib = _lib.load()
        if single:
            self.rank, self.world = 0, 1
        else:
            import torch
            import torch.distributed as dist
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        ident = (C.c_uint8 * 128)()
        err = ""
        if self.rank == 0:
            rc = self._lib.abg_rccl_unique_id(ident)
            if rc != 0:
                err = "abg_rccl_unique_id failed (%d): %s" % (rc, self._lib.abg_last_error(None).decode())
        if not single:
            # (a failure on rank 0 travels with the id, so that every rank raises instead of waiting for it)
            box = [(bytes(ident), err)]
            dist.broadcast_object_list(box, src=0, device=torch.device("cuda", device) if dist.get_backend() == "nccl" else None)
            ident = (C.c_uint8 * 128).from_buffer_copy(box[0][0])
            err = box[0][1]
        if err:
            raise RuntimeError(err)
        self.struct = CommStruct()
        rc = self._lib.abg_rccl_comm_create(ident, self.rank, self.world, device, C.byref(self.struct))
        if rc != 0:
            raise RuntimeError("abg_rccl_comm_create failed (%d): %s" % (rc, self._lib.abg_last_error(None).decode()))

    def close(self):
        if self.struct is not None:
            self._lib.abg_rccl_comm_destroy(C.byref(self.struct))
            self.struct = None


class LocalComm:
    """A communicator of one rank whose collectives are identities (no transport at all): with
    ABG_FORCE_DIST=1 it drives the partitioned code path in a single process."""

    def __init__(self):
        self.rank, self.world = 0, 1
        self.calls = {"all_gather_v": 0, "all_reduce": 0, "all_to_all_v": 0}
        self._agv = AGV_FN(lambda *_a: self._count("all_gather_v"))
        self._ar = AR_FN(lambda *_a: self._count("all_reduce"))
        self._a2a = A2A_FN(lambda *_a: self._count("all_to_all_v"))  # (never called: one rank's exchange is a copy inside the engine)
        self.struct = CommStruct(0, 1, 0, C.sizeof(CommStruct), None, self._agv, self._ar, self._a2a)

    def _count(self, what):
        self.calls[what] += 1
        return 0


class StagedTorchComm:
    """all_gather_v / all_reduce of ``abg_comm`` through torch.distributed on host copies.
    read(ptr, nbytes) -> uint8 array and write(ptr, uint8 array) move bytes between the
    engine's memory space and the host (memmove for tests/hostcheck, abg_dev_copy for a GPU)."""

    def __init__(self, read, write, group=None):
        import torch.distributed as dist
        self.read, self.write, self.group = read, write, group
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.calls = {"all_gather_v": 0, "all_reduce": 0, "all_to_all_v": 0, "bytes": 0}
        self._agv = AGV_FN(self._all_gather_v)
        self._ar = AR_FN(self._all_reduce)
        self._a2a = A2A_FN(self._all_to_all_v)
        self.struct = CommStruct(self.rank, self.world, 0, C.sizeof(CommStruct), None, self._agv, self._ar, self._a2a)

    def _all_gather_v(self, _user, buf, counts, displs, _stream):
        import torch
        import torch.distributed as dist
        try:
            cnt = [int(counts[q]) for q in range(self.world)]
            dsp = [int(displs[q]) for q in range(self.world)]
            width = max(cnt)
            if width == 0:
                return 0
            mine = np.zeros(width, dtype=np.uint8)
            if cnt[self.rank]:
                mine[:cnt[self.rank]] = self.read(buf + dsp[self.rank], cnt[self.rank])
            parts = [torch.empty(width, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, torch.from_numpy(mine), group=self.group)
            for q in range(self.world):
                if q != self.rank and cnt[q]:
                    self.write(buf + dsp[q], parts[q].numpy()[:cnt[q]])
            self.calls["all_gather_v"] += 1
            self.calls["bytes"] += sum(cnt)
            self.calls["bytes_all_gather_v"] = self.calls.get("bytes_all_gather_v", 0) + sum(cnt)
            return 0
        except Exception as e:  # an exception must not unwind through the C caller
            print("StagedTorchComm.all_gather_v:", repr(e), flush=True)
            return -1

    def _all_to_all_v(self, _user, send, scounts, sdispls, recv, rcounts, rdispls, _stream):
        import torch
        import torch.distributed as dist
        try:
            sc = [int(scounts[q]) for q in range(self.world)]
            sd = [int(sdispls[q]) for q in range(self.world)]
            rc = [int(rcounts[q]) for q in range(self.world)]
            rd = [int(rdispls[q]) for q in range(self.world)]
            parts = [self.read(send + sd[q], sc[q]) if sc[q] else np.zeros(0, dtype=np.uint8) for q in range(self.world)]
            out = torch.empty(sum(rc), dtype=torch.uint8)
            dist.all_to_all_single(out, torch.from_numpy(np.concatenate(parts)) if sum(sc) else torch.empty(0, dtype=torch.uint8),
                                   output_split_sizes=rc, input_split_sizes=sc, group=self.group)
            o = out.numpy()
            at = 0
            for q in range(self.world):
                if rc[q]:
                    self.write(recv + rd[q], np.ascontiguousarray(o[at:at + rc[q]]))
                at += rc[q]
            self.calls["all_to_all_v"] += 1
            off_rank = sum(sc) - sc[self.rank]  # (what leaves this rank: its own part is a local copy)
            self.calls["bytes"] += off_rank
            self.calls["bytes_all_to_all_v"] = self.calls.get("bytes_all_to_all_v", 0) + off_rank
            return 0
        except Exception as e:
            print("StagedTorchComm.all_to_all_v:", repr(e), flush=True)
            return -1

    def _all_reduce(self, _user, buf, count, dtype, op, _stream):
        import torch
        import torch.distributed as dist
        try:
            npdt = {U8: np.uint8, U32: np.uint32, U64: np.uint64}[dtype]
            raw = self.read(buf, int(count) * np.dtype(npdt).itemsize).view(npdt)
            # torch.distributed has no unsigned 32/64-bit reductions: widen (values here are far below 2^63)
            t = torch.from_numpy(raw.copy() if dtype == U8 else raw.astype(np.int64))
            dist.all_reduce(t, op={SUM: dist.ReduceOp.SUM, MAX: dist.ReduceOp.MAX, MIN: dist.ReduceOp.MIN}[op], group=self.group)
            self.write(buf, np.ascontiguousarray(t.numpy().astype(npdt)).view(np.uint8))
            self.calls["all_reduce"] += 1
            self.calls["bytes"] += int(count) * np.dtype(npdt).itemsize
            self.calls["bytes_all_reduce"] = self.calls.get("bytes_all_reduce", 0) + int(count) * np.dtype(npdt).itemsize
            return 0
        except Exception as e:
            print("StagedTorchComm.all_reduce:", repr(e), flush=True)
            return -1


def host_memory_io():
    """read/write for buffers that already live in host memory (tests/hostcheck)."""
    def read(ptr, n):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (n,)).copy()

    def write(ptr, arr):
        C.memmove(ptr, arr.ctypes.data, arr.size)
    return read, write


def device_memory_io(ctx):
    """read/write through abg_dev_copy of the context `ctx` (an api.BloomDBG)."""
    lib = _lib.load()

    def read(ptr, n):
        out = np.empty(n, dtype=np.uint8)
        assert lib.abg_dev_copy(ctx._ctx, out.ctypes.data, ptr, n, 1) == 0
        return out

    def write(ptr, arr):
        arr = np.ascontiguousarray(arr)
        assert lib.abg_dev_copy(ctx._ctx, ptr, arr.ctypes.data, arr.size, 0) == 0
    return read, write


# ---------------------------------------------------------------------------------------------
# A communicator checked against the host before a job trusts it with its filter: every entry of abg_comm with the
# shapes the engine uses (uneven byte counts, empty parts, in-place all-gather, u8 / u32 / u64 reductions), the
# expected bytes worked out locally from (source, destination, index).  bench.py runs it on the library's RCCL
# communicator before the timed steps of every N > 1 run; tests/test_dist_partition.py runs it over gloo.
class HostBuf:
    """n bytes of host memory (tests/hostcheck's memory space)."""

    def __init__(self, n):
        self.a = np.zeros(max(n, 1), dtype=np.uint8)
        self.ptr = self.a.ctypes.data

    def put(self, arr):
        self.a[:arr.size] = arr

    def get(self, n):
        return self.a[:n].copy()


class TorchBuf:
    """n bytes of device memory (a torch tensor: what RCCL moves)."""

    def __init__(self, n, device):
        import torch
        self.t = torch.zeros(max(n, 1), dtype=torch.uint8, device=device)
        self.ptr = self.t.data_ptr()

    def put(self, arr):
        import torch
        self.t[:arr.size].copy_(torch.from_numpy(np.ascontiguousarray(arr)))
        torch.cuda.synchronize()

    def get(self, n):
        import torch
        torch.cuda.synchronize()
        return self.t[:n].cpu().numpy().copy()


def selftest(comm, make_buf, stream=None, sync=lambda: None):
    """Run every collective of `comm` (anything with .struct / .rank / .world) once per shape and compare with what the
    host says the result must be.  make_buf(nbytes) -> HostBuf / TorchBuf.  Returns {"ok", "checks", "failed"}."""
    st, r, W = comm.struct, comm.rank, comm.world
    failed, checks = [], 0
    U64A = C.c_uint64 * W

    def pat(src, dst, n, salt):
        return ((np.arange(n, dtype=np.uint64) * 7 + np.uint64(src * 31 + dst * 17 + salt)) & np.uint64(255)).astype(np.uint8)

    # ---- all_to_all_v: uneven counts, zeros among them (also a rank's own part)
    for salt in (0, 1):
        def cnt(a, b):
            return 0 if (a + 2 * b + salt) % 4 == 0 else ((a * 7 + b * 3 + salt) % 5) * 1000 + 13 * (b + 1)
        sc = [cnt(r, q) for q in range(W)]
        rc = [cnt(q, r) for q in range(W)]
        sd = [sum(sc[:q]) for q in range(W)]
        rd = [sum(rc[:q]) for q in range(W)]
        send, recv = make_buf(sum(sc)), make_buf(sum(rc))
        if sum(sc):
            send.put(np.concatenate([pat(r, q, sc[q], salt) for q in range(W)]))
        rcode = st.all_to_all_v(st.user, send.ptr, U64A(*sc), U64A(*sd), recv.ptr, U64A(*rc), U64A(*rd), stream)
        sync()
        want = np.concatenate([pat(q, r, rc[q], salt) for q in range(W)]) if sum(rc) else np.zeros(0, dtype=np.uint8)
        checks += 1
        if rcode != 0 or not np.array_equal(recv.get(sum(rc)), want):
            failed.append("all_to_all_v[%d] rc=%d" % (salt, rcode))
    # ---- all_gather_v in place: every rank's part at its displacement, one part empty
    cn = [0 if q == 1 % W and W > 1 else (q % 3) * 4096 + 5 * (q + 1) for q in range(W)]
    dp = [sum(cn[:q]) for q in range(W)]
    buf = make_buf(sum(cn))
    whole = np.zeros(sum(cn), dtype=np.uint8)
    whole[dp[r]:dp[r] + cn[r]] = pat(r, 0, cn[r], 5)
    buf.put(whole)
    rcode = st.all_gather_v(st.user, buf.ptr, U64A(*cn), U64A(*dp), stream)
    sync()
    want = np.concatenate([pat(q, 0, cn[q], 5) for q in range(W)])
    checks += 1
    if rcode != 0 or not np.array_equal(buf.get(sum(cn)), want):
        failed.append("all_gather_v rc=%d" % rcode)
    # ---- all_reduce: the (type, operation) pairs the engine uses
    n = 3001
    for dt, npdt, ops in ((U8, np.uint8, (MIN, MAX)), (U32, np.uint32, (SUM, MIN, MAX)), (U64, np.uint64, (SUM, MIN))):
        for op in ops:
            def vals(q):
                x = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(q * 40503 + dt * 7 + op)) >> np.uint64(7)
                return (x % np.uint64(251 if dt == U8 else 1 << 20)).astype(npdt)
            b = make_buf(n * np.dtype(npdt).itemsize)
            b.put(vals(r).view(np.uint8))
            rcode = st.all_reduce(st.user, b.ptr, n, dt, op, stream)
            sync()
            allv = np.stack([vals(q).astype(np.uint64) for q in range(W)])
            want = {SUM: allv.sum(axis=0), MAX: allv.max(axis=0), MIN: allv.min(axis=0)}[op].astype(npdt)
            checks += 1
            if rcode != 0 or not np.array_equal(b.get(n * np.dtype(npdt).itemsize).view(npdt), want):
                failed.append("all_reduce dtype=%d op=%d rc=%d" % (dt, op, rcode))
    return {"ok": not failed, "checks": checks, "failed": failed}
