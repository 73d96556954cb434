"""
Multi-GPU use of the batched solver (row (e) of SURVEY.md section 8): instances are independent, so a
batch is cut into contiguous shards, one per rank (one process per GPU), the family plan is replicated,
and nothing is exchanged while solving.  The only exchange is the FINAL gather of the results to the
consumer (rank `root`), for which this module has two transports, neither of them PyTorch:

  RcclGather   results stay on the device; every rank `ncclSend`s its result rows to the root, whose
               `ncclRecv`s land at the shard's offset of ONE device buffer (RCCL through ctypes on the
               solver's own HIP stream: ordered behind the solve kernel, point-to-point over each GPU's
               xGMI link to the root, no ring), then one D2H copy at the root.
  HostGather   every rank copies its result rows D2H straight into its slice of one POSIX shared-memory
               array that the root reads -- no collective at all; the realistic case when the consumer
               is numpy on the host, and the transport the CPU test tier can run.

Both are single-node (the scope of BASELINE.json: the 8 GPUs of one node).  Rendezvous: the RCCL unique id
travels through whatever the launcher already has (`uid_exchange`, e.g. a broadcast over its process group:
bench.py) or, by default, through a /dev/shm file keyed by `key`, which must be unique per JOB
(`job_key()`: the launcher's port plus the pid of the launcher process every rank is a child of) -- a key
reused by a later job would let its ranks read the id a crashed run left behind.  The reference has no
counterpart (single process, single thread).
"""

from __future__ import annotations

import ctypes as C
import os
import time
from multiprocessing import shared_memory
from typing import Dict, Optional, Tuple

import numpy as np


def shard_bounds(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of instances owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def job_key(port=None) -> str:
    """rendezvous key of this job: MASTER_PORT of the launcher + the pid of the parent process (the launcher's
    agent, the same for all ranks of one job and different for the next job on the same port)"""
    return f"{port if port is not None else os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"


def _wait_for(path: str, timeout: float = 120.0) -> None:
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise TimeoutError(f'rendezvous file {path} did not appear')
        time.sleep(0.002)


# --------------------------------------------------------------------------------------------------
class HostGather:
    """Gather of row blocks through one shared-memory array per call: rank r writes rows
    [lo_r, hi_r) (shard_bounds), raises its flag; the root returns the full array."""

    def __init__(self, rank: int, world: int, key: str, root: int = 0):
        self.rank, self.world, self.root, self.key = rank, world, root, str(key)
        self._seq = 0
        self.name = 'host_shm'

    def gather_rows(self, local: np.ndarray, B: int) -> Optional[np.ndarray]:
        local = np.ascontiguousarray(local)
        lo, hi = shard_bounds(B, self.rank, self.world)
        if local.shape[0] != hi - lo:
            raise ValueError(f'rank {self.rank} owns {hi - lo} rows, got {local.shape[0]}')
        tail = local.shape[1:]
        row_bytes = int(np.prod(tail, dtype=np.int64)) * local.dtype.itemsize if tail else local.dtype.itemsize
        hdr = 64 * self.world
        name = f'cpg_{self.key}_{self._seq}'
        self._seq += 1
        ready = f'/dev/shm/{name}.ready'
        if self.rank == self.root:
            try:
                shm = shared_memory.SharedMemory(name=name, create=True, size=hdr + max(1, B * row_bytes))
            except FileExistsError:
                # a segment a killed run left under the same key: nobody of THIS job can be attached yet (the
                # ready file is written after the segment exists), so it is safe to replace
                if os.path.exists(ready):
                    os.remove(ready)
                stale = shared_memory.SharedMemory(name=name)
                stale.close(); stale.unlink()
                shm = shared_memory.SharedMemory(name=name, create=True, size=hdr + max(1, B * row_bytes))
            shm.buf[:hdr] = bytes(hdr)
            open(ready, 'w').close()
        else:
            _wait_for(ready)
            shm = shared_memory.SharedMemory(name=name)
        try:
            full = np.ndarray((B,) + tail, dtype=local.dtype, buffer=shm.buf, offset=hdr)
            full[lo:hi] = local
            shm.buf[64 * self.rank] = 1
            if self.rank != self.root:
                del full
                return None
            t0 = time.time()
            while not all(shm.buf[64 * r] == 1 for r in range(self.world)):
                if time.time() - t0 > 600:
                    raise TimeoutError('HostGather: a rank did not deliver its shard')
                time.sleep(0.0005)
            out = np.array(full)
            del full
            return out
        finally:
            shm.close()
            if self.rank == self.root:
                shm.unlink()
                os.remove(ready)

    def close(self) -> None:
        pass


# --------------------------------------------------------------------------------------------------
class _NcclUniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]


class RcclGather:
    """Gather-to-root over RCCL (librccl through ctypes) of arrays that live in device memory of the
    solver's handle.  `solver` provides the C-ABI handle whose HIP stream the transfers are queued on."""

    NCCL_UINT8 = 1

    def __init__(self, solver, rank: int, world: int, key: str, root: int = 0, lib: str = 'librccl.so',
                 uid_exchange=None):
        """uid_exchange(raw: bytes | None) -> bytes: delivers the root's 128-byte RCCL unique id to every rank (the
        root passes it, the others pass None); default: a /dev/shm file keyed by `key` (see job_key)."""
        self.rank, self.world, self.root, self.key = rank, world, root, str(key)
        self.s = solver
        self.name = 'rccl'
        self.L = C.CDLL(lib)
        L = self.L
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
        L.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        stream = C.c_void_p()
        solver.lib.check(solver.lib.L.cpg_hip_get_stream(solver.h, C.byref(stream)), 'cpg_hip_get_stream')
        self.stream = stream
        uid = _NcclUniqueId()
        path = f'/dev/shm/cpg_rccl_{self.key}.id'
        self._id_path = None
        if rank == root:
            self._ck(L.ncclGetUniqueId(C.byref(uid)), 'ncclGetUniqueId')
        if uid_exchange is not None:
            raw = uid_exchange(C.string_at(C.byref(uid), 128) if rank == root else None)     # (all 128 bytes: a c_char field would stop at a NUL)
            if rank != root:
                C.memmove(C.byref(uid), raw, min(128, len(raw)))
        elif rank == root:
            with open(path + '.tmp', 'wb') as f:
                f.write(C.string_at(C.byref(uid), 128))
            os.replace(path + '.tmp', path)          # atomic: a reader never sees a partial id
            self._id_path = path
        else:
            _wait_for(path)
            raw = open(path, 'rb').read()
            C.memmove(C.byref(uid), raw, min(128, len(raw)))
        self.comm = C.c_void_p()
        self._ck(L.ncclCommInitRank(C.byref(self.comm), world, uid, rank), 'ncclCommInitRank')
        self._gbufs: Dict[str, list] = {}      # root: one device gather buffer per named array [ptr, bytes]

    def _ck(self, rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError(f'{what} failed: {self.L.ncclGetErrorString(rc).decode()}')

    def _gbuf(self, name: str, need: int) -> C.c_void_p:
        s = self.s
        ent = self._gbufs.get(name)
        if ent is None or ent[1] < need:
            if ent is not None and ent[0].value:
                s.lib.check(s.lib.L.cpg_hip_free(s.h, ent[0]), 'cpg_hip_free')
            p = C.c_void_p()
            s.lib.check(s.lib.L.cpg_hip_malloc(s.h, need, C.byref(p)), 'cpg_hip_malloc')
            ent = [p, need]
            self._gbufs[name] = ent
        return ent[0]

    def enqueue(self, arrays, B: int) -> None:
        """arrays: list of (name, device pointer of this rank's block, rows, row_bytes).  Queues, on the solver's
        stream (behind the solve kernel), this rank's sends to the root -- or, on the root, the receives from
        every other rank at their shard's offset of the array's device gather buffer -- as ONE RCCL group.
        Asynchronous; `solver.synchronize()` completes it."""
        L = self.L
        lo, hi = shard_bounds(B, self.rank, self.world)
        for name, d_ptr, rows, row_bytes in arrays:
            if rows != hi - lo:
                raise ValueError(f'rank {self.rank} owns {hi - lo} rows, got {rows}')
        if self.world == 1:
            return
        bufs = {}
        if self.rank == self.root:
            for name, d_ptr, rows, row_bytes in arrays:
                bufs[name] = self._gbuf(name, max(1, B * row_bytes))
        self._ck(L.ncclGroupStart(), 'ncclGroupStart')
        for name, d_ptr, rows, row_bytes in arrays:
            if self.rank == self.root:
                for r in range(self.world):
                    rlo, rhi = shard_bounds(B, r, self.world)
                    if r != self.root and rhi > rlo:
                        self._ck(L.ncclRecv(C.c_void_p(bufs[name].value + rlo * row_bytes), (rhi - rlo) * row_bytes,
                                            self.NCCL_UINT8, r, self.comm, self.stream), 'ncclRecv')
            elif rows:
                self._ck(L.ncclSend(d_ptr, rows * row_bytes, self.NCCL_UINT8, self.root, self.comm, self.stream), 'ncclSend')
        self._ck(L.ncclGroupEnd(), 'ncclGroupEnd')

    def fetch(self, name: str, d_ptr, row_bytes: int, B: int, dtype, tail=()) -> Optional[np.ndarray]:
        """after enqueue + synchronize: the gathered [B, ...] array on the root's host (one D2H per shard
        source: the gather buffer for remote shards, the root's own block directly); None elsewhere"""
        s = self.s
        if self.rank != self.root:
            return None
        out = np.empty((B,) + tuple(tail), dtype=dtype)
        flat = out.reshape(-1).view(np.uint8)
        for r in range(self.world):
            rlo, rhi = shard_bounds(B, r, self.world)
            if rhi == rlo:
                continue
            src = d_ptr if r == self.root else C.c_void_p(self._gbufs[name][0].value + rlo * row_bytes)
            dst = flat[rlo * row_bytes:rhi * row_bytes]
            s.lib.check(s.lib.L.cpg_hip_memcpy_d2h(s.h, dst.ctypes.data_as(C.c_void_p), src, (rhi - rlo) * row_bytes), 'd2h')
        return out

    def close(self) -> None:
        for ent in self._gbufs.values():
            if ent[0].value:
                self.s.lib.L.cpg_hip_free(self.s.h, ent[0])
        self._gbufs = {}
        if self.comm.value:
            self.L.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
        if self._id_path and os.path.exists(self._id_path):
            os.remove(self._id_path)


# --------------------------------------------------------------------------------------------------
def result_spec(dev) -> Dict[str, tuple]:
    """device result arrays of a DeviceBatch: key -> (row bytes, dtype, trailing shape, name in the result dict)"""
    return dict(prim=(dev.n_prim * 8, np.float64, (dev.n_prim,), 'prim'), dual=(dev.n_dual * 8, np.float64, (dev.n_dual,), 'dual'),
                obj=(8, np.float64, (), 'obj_val'), iter=(4, np.int32, (), 'iter'), status=(4, np.int32, (), 'status'),
                pri=(8, np.float64, (), 'pri_res'), dua=(8, np.float64, (), 'dua_res'))


def solve_sharded(solver, theta_var: np.ndarray, rank: int, world: int, gather, **kwargs) -> Optional[Dict[str, np.ndarray]]:
    """Every rank passes the FULL theta_var [B, np_var] (or any array whose rows [lo, hi) are its shard's);
    each solves its shard on its own GPU; the flat results are gathered on the root through `gather`
    (HostGather or RcclGather).  Returns the dict of full arrays on the root, None elsewhere."""
    from .runtime import DeviceBatch
    B = theta_var.shape[0]
    lo, hi = shard_bounds(B, rank, world)
    local = np.ascontiguousarray(theta_var[lo:hi])
    out: Dict[str, np.ndarray] = {}
    if isinstance(gather, RcclGather):
        solver.apply_settings(**kwargs)
        dev = DeviceBatch(solver, hi - lo)
        dev.upload(local)
        solver.solve_device(dev)
        st = np.empty(hi - lo, dtype=np.int32)
        if hi > lo:
            solver.lib.check(solver.lib.L.cpg_hip_memcpy_d2h(solver.h, st.ctypes.data_as(C.c_void_p), dev._ptrs['status'],
                                                             st.nbytes), 'd2h')
        if (st == -2).any():
            # rows that changed class are re-solved through the per-instance factor path on this rank
            # (BatchSolver._resolve_class_changes) and written back before they travel
            fixed = dev.download()
            raw_obj = np.where(np.isinf(fixed.obj_val), np.sign(fixed.obj_val) * 1e30, fixed.obj_val)
            for k, a in (('prim', fixed.prim_flat), ('dual', fixed.dual_flat), ('obj', raw_obj), ('iter', fixed.iter),
                         ('status', fixed.status), ('pri', fixed.pri_res), ('dua', fixed.dua_res)):
                a = np.ascontiguousarray(a)
                solver.lib.check(solver.lib.L.cpg_hip_memcpy_h2d(solver.h, dev._ptrs[k], a.ctypes.data_as(C.c_void_p),
                                                                 a.nbytes), 'h2d')
        spec = result_spec(dev)
        gather.enqueue([(k, dev._ptrs[k], hi - lo, rb) for k, (rb, dt, tail, nm) in spec.items()], B)
        solver.synchronize()
        for k, (rb, dt, tail, nm) in spec.items():
            out[nm] = gather.fetch(k, dev._ptrs[k], rb, B, dt, tail)
        dev.free()
        if out['obj_val'] is not None:          # +-1e30 -> +-inf as the reference shim does (templates/cpg_solver.py.jinja2:98-101)
            o = out['obj_val']
            out['obj_val'] = np.where(np.abs(o) >= 1e30, np.sign(o) * np.inf, o)
    else:
        res = solver.solve(theta_var=local, B=hi - lo, **kwargs)
        for name, arr in (('prim', res.prim_flat), ('dual', res.dual_flat), ('obj_val', res.obj_val),
                          ('iter', res.iter), ('status', res.status), ('pri_res', res.pri_res),
                          ('dua_res', res.dua_res)):
            out[name] = gather.gather_rows(np.ascontiguousarray(arr), B)
    return out if rank == gather.root else None
