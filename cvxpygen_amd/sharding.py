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
    [lo_r, hi_r) (shard_bounds), raises its flag; the root returns the full array.

    A segment (and its ready file) that a killed run left under the same key must not receive a shard of THIS
    job: every segment carries a fresh nonce (also the content of its ready file), a rank delivers a random
    token with its shard and returns only after the root echoed that token into the SAME mapping -- a stale
    mapping never echoes a fresh token, and the rank re-attaches when the ready file's nonce changes under it."""

    SLOT = 64                                   # header bytes per rank: [0] flag, [8:16] token of the rank, [16:24] echo of the root

    def __init__(self, rank: int, world: int, key: str, root: int = 0, timeout: float = 600.0):
        self.rank, self.world, self.root, self.key = rank, world, root, str(key)
        self._seq = 0
        self.name = 'host_shm'
        self.timeout = timeout

    @staticmethod
    def _read(path: str) -> Optional[bytes]:
        try:
            with open(path, 'rb') as f:
                return f.read()
        except OSError:
            return None

    def gather_rows(self, local: np.ndarray, B: int) -> Optional[np.ndarray]:
        local = np.ascontiguousarray(local)
        lo, hi = shard_bounds(B, self.rank, self.world)
        if local.shape[0] != hi - lo:
            raise ValueError(f'rank {self.rank} owns {hi - lo} rows, got {local.shape[0]}')
        tail = local.shape[1:]
        row_bytes = int(np.prod(tail, dtype=np.int64)) * local.dtype.itemsize if tail else local.dtype.itemsize
        S = self.SLOT
        hdr = S * self.world + S                # (+ one slot for the segment's nonce)
        size = hdr + max(1, B * row_bytes)
        name = f'cpg_{self.key}_{self._seq}'
        self._seq += 1
        ready = f'/dev/shm/{name}.ready'
        t0 = time.time()
        if self.rank == self.root:
            nonce = os.urandom(16)
            if os.path.exists(ready):           # left by a killed run: take it away BEFORE the segment is replaced
                os.remove(ready)
            try:
                shm = shared_memory.SharedMemory(name=name, create=True, size=size)
            except FileExistsError:
                stale = shared_memory.SharedMemory(name=name)
                stale.close(); stale.unlink()
                shm = shared_memory.SharedMemory(name=name, create=True, size=size)
            try:
                shm.buf[:hdr] = bytes(hdr)
                shm.buf[S * self.world:S * self.world + 16] = nonce
                with open(ready + '.tmp', 'wb') as f:
                    f.write(nonce)
                os.replace(ready + '.tmp', ready)
                full = np.ndarray((B,) + tail, dtype=local.dtype, buffer=shm.buf, offset=hdr)
                full[lo:hi] = local
                others = [r for r in range(self.world) if r != self.root]
                while not all(shm.buf[S * r] == 1 for r in others):
                    if time.time() - t0 > self.timeout:
                        raise TimeoutError('HostGather: a rank did not deliver its shard')
                    time.sleep(0.0005)
                out = np.array(full)
                del full
                for r in others:                # the echo: "this mapping is the one the root read"
                    shm.buf[S * r + 16:S * r + 24] = bytes(shm.buf[S * r + 8:S * r + 16])
                return out
            finally:
                shm.close()
                shm.unlink()
                if os.path.exists(ready):
                    os.remove(ready)
        token = os.urandom(8)
        while True:
            if time.time() - t0 > self.timeout:
                raise TimeoutError('HostGather: the root did not take this rank\'s shard')
            nonce = self._read(ready)
            if not nonce or len(nonce) != 16:
                time.sleep(0.001)
                continue
            try:
                shm = shared_memory.SharedMemory(name=name)
            except FileNotFoundError:
                time.sleep(0.001)
                continue
            try:
                if shm.size < size or bytes(shm.buf[S * self.world:S * self.world + 16]) != nonce:
                    time.sleep(0.001)           # segment and ready file of different generations
                    continue
                full = np.ndarray((B,) + tail, dtype=local.dtype, buffer=shm.buf, offset=hdr)
                full[lo:hi] = local
                del full
                o = S * self.rank
                shm.buf[o + 8:o + 16] = token
                shm.buf[o] = 1
                while True:
                    if bytes(shm.buf[o + 16:o + 24]) == token:
                        return None
                    if self._read(ready) != nonce and bytes(shm.buf[o + 16:o + 24]) != token:
                        break                   # the root replaced (or never owned) this segment: deliver again
                    if time.time() - t0 > self.timeout:
                        raise TimeoutError('HostGather: the root did not take this rank\'s shard')
                    time.sleep(0.0005)
            finally:
                shm.close()

    def close(self) -> None:
        pass


class HostDeviceGather(HostGather):
    """HostGather behind the interface of RcclGather (enqueue / fetch on device result arrays): every rank copies
    its result rows D2H after its solve and delivers them through the shared-memory slices.  The transport
    `make_gather` falls back to when the RCCL communicator cannot be created."""

    def __init__(self, solver, rank: int, world: int, key: str, root: int = 0):
        super().__init__(rank, world, key, root)
        self.s = solver
        self._full: Dict[str, Optional[np.ndarray]] = {}

    def enqueue(self, arrays, B: int) -> None:
        """same arguments as RcclGather.enqueue; synchronous (waits for the solve, copies, gathers)"""
        s = self.s
        s.synchronize()
        for name, d_ptr, rows, row_bytes in arrays:
            blk = np.empty((rows, row_bytes), dtype=np.uint8)
            if rows:
                s.lib.check(s.lib.L.cpg_hip_memcpy_d2h(s.h, blk.ctypes.data_as(C.c_void_p), d_ptr, blk.nbytes), 'd2h')
            self._full[name] = blk if self.world == 1 else self.gather_rows(blk, B)

    def fetch(self, name: str, d_ptr, row_bytes: int, B: int, dtype, tail=()) -> Optional[np.ndarray]:
        if self.rank != self.root:
            return None
        return np.ascontiguousarray(self._full[name]).reshape(-1).view(dtype).reshape((B,) + tuple(tail))


# --------------------------------------------------------------------------------------------------
class _NcclUniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]


class RcclGather:
    """Gather-to-root over RCCL (librccl through ctypes) of arrays that live in device memory of the
    solver's handle.  `solver` provides the C-ABI handle whose HIP stream the transfers are queued on."""

    NCCL_UINT8 = 1

    def __init__(self, solver, rank: int, world: int, key: str, root: int = 0, lib: str = 'librccl.so',
                 uid_exchange=None, init_timeout: float = 120.0):
        """uid_exchange(raw: bytes | None) -> bytes: delivers the root's 128-byte RCCL unique id to every rank (the
        root passes it, the others pass None); default: a /dev/shm file keyed by `key` (see job_key).
        init_timeout: seconds ncclCommInitRank may take (it is a collective: a peer that never arrives, or a
        fabric RCCL cannot bring up, would otherwise hang the job) -- TimeoutError after that."""
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
        self._gbufs: Dict[str, list] = {}      # root: one device gather buffer per named array [ptr, bytes]
        import threading
        box: Dict[str, object] = {}

        def init():                            # (a foreign call releases the GIL: the wait below can time out)
            try:
                box['rc'] = L.ncclCommInitRank(C.byref(self.comm), world, uid, rank)
            except BaseException as e:         # noqa: BLE001 -- handed to the caller's thread
                box['exc'] = e
        th = threading.Thread(target=init, daemon=True)
        th.start()
        th.join(init_timeout)
        if th.is_alive():
            raise TimeoutError(f'ncclCommInitRank did not return within {init_timeout:.0f} s')
        if 'exc' in box:
            raise box['exc']
        self._ck(int(box['rc']), 'ncclCommInitRank')

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
def make_gather(solver, rank: int, world: int, key: str, root: int = 0, lib: str = 'librccl.so', uid_exchange=None,
                agree=None, init_timeout: float = 120.0):
    """The gather of a multi-GPU job: RcclGather when EVERY rank could create its communicator, else
    HostDeviceGather (D2H + shared-memory slices).  `agree(ok: bool) -> bool` must return the AND over all ranks
    (e.g. an all-reduce over the launcher's process group); without it each rank decides alone, which is only
    safe when the failure is the same everywhere (library missing).  Returns (gather, kind, note): kind 'rccl' or
    'host', note = the reason of the fallback ('' when none)."""
    g, err = None, ''
    called = []

    def exchange(raw):
        called.append(1)
        return uid_exchange(raw)
    try:
        g = RcclGather(solver, rank, world, key=key, root=root, lib=lib, uid_exchange=exchange if uid_exchange is not None else None,
                       init_timeout=init_timeout)
    except (OSError, RuntimeError, TimeoutError, AttributeError) as e:
        err = f'{type(e).__name__}: {e}'
        if uid_exchange is not None and not called:     # the exchange is a collective of the launcher: take part in it
            uid_exchange(bytes(128) if rank == root else None)
    ok = g is not None
    all_ok = agree(ok) if agree is not None else ok
    if all_ok:
        return g, 'rccl', ''
    if g is not None:                          # another rank failed: this rank's communicator is of no use
        try:
            g.close()
        except Exception:                      # noqa: BLE001
            pass
    return HostDeviceGather(solver, rank, world, key=key, root=root), 'host', (err or 'another rank could not create its communicator')


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
    if isinstance(gather, (RcclGather, HostDeviceGather)):
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
