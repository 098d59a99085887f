"""
Voice sharding across the GPUs of one node.

Voices are independent until the sum, so the voice table is partitioned contiguously over the
ranks (one process per GPU); every rank renders the float64 partial stereo bus of its shard for the
same frame range and the partial buses are summed by RCCL over xGMI (``ncclReduce``, float64, root
0) -- the only collective on the path.  The reference is single-process: this is new work asked for
by BASELINE.json, not a translation of anything upstream.

Rendezvous: rank 0 creates the ``ncclUniqueId`` and broadcasts its 128 bytes -- by default through a
tiny TCP exchange on MASTER_ADDR:MASTER_PORT+1 (no dependency beyond the standard library); a launcher
that already has a channel passes its own ``broadcast(payload, rank, world, nbytes) -> bytes`` callable to
``init`` (bench.py, started by the distributed launcher, hands over its gloo process group that way).  This package
imports no ML framework.
"""
from __future__ import annotations

import ctypes as C
import os
import socket
import time
from typing import Callable, Optional, Sequence, Tuple

import numpy as np

from . import _native as N

__all__ = ["shard_range", "shard_sizes", "init_from_env", "init", "shutdown", "DistVoiceBank", "Rendezvous", "rank_env", "free_port",
           "resample_ranges", "resample_span", "resample_shard"]


def shard_sizes(nvoices: int, world: int) -> list:
    """Voices per rank: contiguous blocks, the first ``nvoices % world`` ranks get one extra."""
    if world < 1:
        raise ValueError("world must be >= 1")
    base, extra = divmod(nvoices, world)
    return [base + (1 if r < extra else 0) for r in range(world)]


def shard_range(nvoices: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the voice table owned by ``rank``."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    sizes = shard_sizes(nvoices, world)
    lo = sum(sizes[:rank])
    return lo, lo + sizes[rank]


# -- rendezvous ----------------------------------------------------------------------------------

def _tcp_broadcast(payload: Optional[bytes], rank: int, world: int, nbytes: int) -> bytes:
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("SYNTHHIP_RDZV_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((host, port))
        srv.listen(world)
        for _ in range(world - 1):
            conn, _addr = srv.accept()
            conn.sendall(payload)
            conn.close()
        srv.close()
        return payload
    deadline = time.time() + 120
    while True:
        try:
            c = socket.create_connection((host, port), timeout=5)
            break
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.1)
    data = b""
    while len(data) < nbytes:
        chunk = c.recv(nbytes - len(data))
        if not chunk:
            raise ConnectionError("rendezvous closed early")
        data += chunk
    c.close()
    return data


class Rendezvous:
    """The control channel of a one-node, one-process-per-GPU job -- no ML framework, the standard library only.

    * data (``broadcast``, ``gather``, ``allmax``): a TCP star on MASTER_ADDR : MASTER_PORT + 1 (``SYNTHHIP_RDZV_PORT`` overrides);
      rank 0 listens, the others connect once and keep the connection.  Every operation is one length-prefixed message per rank
      to rank 0 and one back: ~100 us on loopback, used OUTSIDE timed regions.
    * ``barrier``: the ranks share a page of /dev/shm (one 64-byte slot per rank, written by its owner only): a rank publishes
      its generation number and spins until every slot has reached it -- a few microseconds, against ~0.1-1 ms for a barrier over
      TCP; a timed pass of twenty 36-us blocks is bracketed by two of them.  Without /dev/shm the barrier goes over the sockets.

    ``world == 1``: every operation is the identity and nothing is opened."""

    def __init__(self, rank: int, world: int, addr: Optional[str] = None, port: Optional[int] = None, timeout: float = 120.0,
                 shm_barrier: bool = True) -> None:
        self.rank, self.world = int(rank), int(world)
        self._conns = []                  # rank 0: socket of rank r at [r - 1]
        self._sock = None                 # others: the socket to rank 0
        self._shm = None
        self._slots = None
        self._gen = 0
        self._shm_path = None
        if self.world <= 1:
            return
        host = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port if port is not None else os.environ.get("SYNTHHIP_RDZV_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
        token = os.environ.get("SYNTHHIP_RDZV_TOKEN", "")      # a per-job secret from the launcher (rank_env); "" under a launcher that sets none
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((host, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            got = {}
            while len(got) < self.world - 1:
                conn, _addr = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                # a connector that does not say hello properly (a port scan, another job's rank, a wrong token) is dropped; the job's own
                # ranks are still awaited
                try:
                    hello = self._recv_obj(conn)
                    r = int(hello["rank"])
                    if hello.get("token") != token or not 0 < r < self.world or r in got:
                        raise ConnectionError("rendezvous: unexpected hello %r" % (hello,))
                except (ConnectionError, OSError, KeyError, TypeError, ValueError):
                    conn.close()
                    continue
                got[r] = conn
            srv.close()
            self._conns = [got[r] for r in range(1, self.world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    c = socket.create_connection((host, port), timeout=5)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(timeout)
            self._sock = c
            self._send_obj(c, {"rank": self.rank, "token": token})
        if shm_barrier:
            self._open_shm()

    # -- framing ------------------------------------------------------------------------------------
    # Messages are JSON (ints, floats, strings, lists, dicts, None; bytes as {"__b64__": ...}) behind a 4-byte length, capped at
    # MAX_MESSAGE: nothing that arrives on the port is ever unpickled (unpickling what a TCP port delivers is code execution for
    # whoever can reach it), and a bogus length cannot ask for more than MAX_MESSAGE bytes.
    MAX_MESSAGE = 1 << 20

    @staticmethod
    def _encode(obj):
        if isinstance(obj, (bytes, bytearray)):
            import base64
            return {"__b64__": base64.b64encode(bytes(obj)).decode("ascii")}
        if isinstance(obj, dict):
            return {str(k): Rendezvous._encode(v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return [Rendezvous._encode(v) for v in obj]
        if isinstance(obj, (np.integer,)):
            return int(obj)
        if isinstance(obj, (np.floating,)):
            return float(obj)
        return obj

    @staticmethod
    def _decode(obj):
        if isinstance(obj, dict):
            if len(obj) == 1 and "__b64__" in obj:
                import base64
                return base64.b64decode(obj["__b64__"], validate=True)
            return {k: Rendezvous._decode(v) for k, v in obj.items()}
        if isinstance(obj, list):
            return [Rendezvous._decode(v) for v in obj]
        return obj

    @classmethod
    def _send_obj(cls, sock, obj) -> None:
        import json
        data = json.dumps(cls._encode(obj), allow_nan=True, separators=(",", ":")).encode("utf-8")
        if len(data) > cls.MAX_MESSAGE:
            raise ValueError("rendezvous: message of %d bytes (limit %d)" % (len(data), cls.MAX_MESSAGE))
        sock.sendall(len(data).to_bytes(4, "little") + data)

    @staticmethod
    def _recv_exact(sock, n: int) -> bytes:
        buf = bytearray()
        while len(buf) < n:
            chunk = sock.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("rendezvous: a peer closed the connection")
            buf += chunk
        return bytes(buf)

    @classmethod
    def _recv_obj(cls, sock):
        import json
        n = int.from_bytes(cls._recv_exact(sock, 4), "little")
        if n > cls.MAX_MESSAGE:
            raise ConnectionError("rendezvous: a peer announced a message of %d bytes (limit %d)" % (n, cls.MAX_MESSAGE))
        try:
            return cls._decode(json.loads(cls._recv_exact(sock, n).decode("utf-8")))
        except (ValueError, UnicodeDecodeError) as e:
            raise ConnectionError("rendezvous: malformed message (%s)" % e) from None

    def _exchange(self, obj, combine):
        """Every rank contributes obj; rank 0 applies combine(list by rank) and everyone gets the result."""
        if self.world <= 1:
            return combine([obj])
        if self.rank == 0:
            parts = [obj] + [self._recv_obj(c) for c in self._conns]
            result = combine(parts)
            for c in self._conns:
                self._send_obj(c, result)
            return result
        self._send_obj(self._sock, obj)
        return self._recv_obj(self._sock)

    # -- collectives --------------------------------------------------------------------------------
    def broadcast(self, payload: Optional[bytes], nbytes: Optional[int] = None) -> bytes:
        """Rank 0's bytes on every rank."""
        out = self._exchange(payload if self.rank == 0 else None, lambda parts: parts[0])
        if nbytes is not None and len(out) != nbytes:
            raise ConnectionError("rendezvous: broadcast of %d bytes, expected %d" % (len(out), nbytes))
        return out

    def gather(self, obj) -> list:
        """[rank 0's obj, rank 1's, ...] on every rank."""
        return self._exchange(obj, lambda parts: list(parts))

    def allmax(self, *vals: float) -> Tuple[float, ...]:
        return tuple(self._exchange([float(v) for v in vals], lambda parts: [max(col) for col in zip(*parts)]))

    def as_broadcast(self) -> Callable[[Optional[bytes], int, int, int], bytes]:
        """The callable ``init(broadcast=...)`` takes."""
        return lambda payload, rank, world, nbytes: self.broadcast(payload, nbytes)

    # -- barrier ------------------------------------------------------------------------------------
    def _open_shm(self) -> None:
        import mmap
        import secrets
        path = None
        if self.rank == 0:
            try:
                path = "/dev/shm/synthhip_rdzv_%d_%s" % (os.getpid(), secrets.token_hex(4))
                with open(path, "wb") as f:
                    f.write(b"\0" * (64 * self.world))
            except OSError:
                path = None
        path = self._exchange(path, lambda parts: parts[0])
        if path is None:
            return
        try:
            f = open(path, "r+b")
            self._shm = mmap.mmap(f.fileno(), 64 * self.world)
            f.close()
            self._slots = np.frombuffer(self._shm, dtype=np.int64)[::8]       # slot r at byte 64 r: a cache line per writer
            self._shm_path = path
        except OSError:
            self._shm, self._slots = None, None
        ok = self._exchange(self._slots is not None, lambda parts: all(parts))
        if not ok:
            self._slots = None
        if self.rank == 0:                                                     # everybody has it mapped (or gave up): the name can go
            try:
                os.unlink(path)
            except OSError:
                pass

    def barrier(self, timeout: float = 300.0) -> None:
        if self.world <= 1:
            return
        if self._slots is None:
            self._exchange(None, lambda parts: None)
            return
        self._gen += 1
        g = self._gen
        self._slots[self.rank] = g                       # (an aligned 8-byte store: seen whole or not at all)
        slots = self._slots
        t_end = None
        spins = 0
        while int(slots.min()) < g:
            spins += 1
            if spins & 0xFFFF == 0:
                now = time.time()
                if t_end is None:
                    t_end = now + timeout
                elif now > t_end:
                    raise TimeoutError("rendezvous: barrier %d timed out on rank %d (slots %s)" % (g, self.rank, slots.tolist()))

    def close(self) -> None:
        for c in self._conns:
            try:
                c.close()
            except OSError:
                pass
        self._conns = []
        if self._sock is not None:
            try:
                self._sock.close()
            except OSError:
                pass
            self._sock = None
        self._slots = None
        if self._shm is not None:
            try:
                self._shm.close()
            except (BufferError, OSError):
                pass
            self._shm = None


_TOKEN = None


def _launcher_token() -> str:
    """One random token per launcher process (every rank_env() of a job comes from the same process)."""
    global _TOKEN
    if _TOKEN is None:
        import secrets
        _TOKEN = secrets.token_hex(16)
    return _TOKEN


def rank_env(rank: int, world: int, master_port: int, base: Optional[dict] = None, master_addr: str = "127.0.0.1") -> dict:
    """The environment of rank ``rank`` of a one-node job of ``world`` processes, one per GPU: what a launcher sets (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR, MASTER_PORT) plus the IPC mode the host driver needs -- so that ``device_for_rank`` maps rank r to GPU
    ordinal r.  bench.py and tests/test_gpu_multi.py start their ranks with it."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR=master_addr, MASTER_PORT=str(master_port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SYNTHHIP_RDZV_TOKEN"] = _launcher_token()       # the ranks one launcher starts share a secret: Rendezvous drops any other connector
    env.pop("SYNTHHIP_DEVICE", None)
    env.pop("SYNTHHIP_RDZV_PORT", None)
    return env


def free_port(addr: str = "127.0.0.1") -> int:
    """A port p such that p and p + 1 are free right now (MASTER_PORT and the rendezvous port behind it)."""
    for _ in range(64):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as a:
            a.bind((addr, 0))
            p = a.getsockname()[1]
            if p >= 65535:
                continue
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as b:
                try:
                    b.bind((addr, p + 1))
                except OSError:
                    continue
            return p
    raise OSError("no pair of free ports found")


def init(rank: int, world: int, broadcast: Optional[Callable[[Optional[bytes], int, int, int], bytes]] = None) -> None:
    """Create the RCCL communicator on this process's GPU (sh_init must select the right device first)."""
    N.ensure_init()
    L = N.lib()
    if broadcast is None:
        broadcast = _tcp_broadcast
    payload = None
    if rank == 0:
        buf = (C.c_char * N.SH_DIST_ID_BYTES)()
        N.check(L.sh_dist_unique_id(buf))
        payload = bytes(buf.raw)
    ident = broadcast(payload, rank, world, N.SH_DIST_ID_BYTES)
    raw = (C.c_char * N.SH_DIST_ID_BYTES).from_buffer_copy(ident)
    N.check(L.sh_dist_init(rank, world, raw))


def init_from_env() -> Tuple[int, int]:
    """RANK / WORLD_SIZE / LOCAL_RANK as the launcher sets them (one process per GPU); selects GPU LOCAL_RANK."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    N.ensure_init(device_for_rank())
    if world > 1:
        init(rank, world)
    return rank, world


def shutdown() -> None:
    N.check(N.lib().sh_dist_shutdown())


def device_for_rank(env=None, visible: Optional[int] = None) -> int:
    """The GPU ordinal of this process: LOCAL_RANK as the launcher set it (one process per GPU on one node), else RANK, else 0 -- unless
    the launcher shows every rank exactly ONE GPU (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES set per process): then that one, ordinal 0.
    ``visible``: how many GPUs this process sees (asked of the library for the real environment when it matters: LOCAL_RANK >= 1).
    Pure host logic otherwise (tests/test_host_logic.py maps eight ranks to eight distinct ordinals with it)."""
    real = env is None
    env = os.environ if env is None else env
    r = int(env.get("LOCAL_RANK", env.get("RANK", "0")))
    if visible is None and real and r > 0:
        try:
            visible = int(N.lib().sh_device_count())
        except Exception:
            visible = None
    if visible == 1 and r > 0:
        return 0
    return r


def comm_info() -> dict:
    """What RCCL itself reports about this process's communicator (ncclCommCount / ncclCommUserRank / ncclGetVersion), beside
    what the library was told: a bench line can show that RCCL SAW n ranks.  Without a communicator (single GPU): world 1, rank 0."""
    L = N.lib()
    v = (C.c_int32 * 5)()
    L.sh_dist_comm_info(v, 5)
    has = bool(v[0])
    ver = int(v[3])
    return {"communicator": has, "world": int(v[1]) if has else 1, "rank": int(v[2]) if has else 0,
            "told_world": int(L.sh_dist_world()) if has else 1, "told_rank": int(L.sh_dist_rank()) if has else 0,
            "version": ("%d.%d.%d" % (ver // 10000, (ver // 100) % 100, ver % 100)) if ver else None,
            "librccl_loaded": bool(v[4])}


class _HipBackend:
    """What DistVoiceBank's slot ring needs from the device side: libsynthhip (RCCL on the communication stream)."""

    def __init__(self, voices: Sequence, gains: Sequence[Tuple[float, float]]) -> None:
        from .mixer import VoiceBank
        self.local = VoiceBank(list(voices), gains=list(gains))

    def nslots(self) -> int:
        return N.lib().sh_dist_slots()

    def alloc(self, nbytes: int):
        return N.DeviceBuffer(nbytes)

    def view(self, buf, offset: int, nbytes: int):
        return buf.view(offset, nbytes)

    def render(self, nframes: int, start: int, bus_f32, bus_f64) -> None:
        self.local.render_device(nframes, start, bus_f32=bus_f32, bus_f64=bus_f64)

    def wait_slot(self, slot: int) -> None:
        N.check(N.lib().sh_dist_wait_slot(slot))

    def reduce_async(self, bus_f64, nvalues: int, root: int, bus_f32, slot: int) -> None:
        N.check(N.lib().sh_dist_reduce_bus_async(bus_f64.handle, nvalues, root, bus_f32.handle, slot))

    def mark_slot(self, slot: int) -> None:
        """The slot's last render lies two launches back: record where the render streams stand."""
        N.check(N.lib().sh_dist_mark_slot(slot))

    def reduce_lagged(self, bus_f64, nvalues: int, root: int, bus_f32, slot: int) -> None:
        """The reduce of a marked slot, a few launches after the mark: the run of pipelined renders goes on."""
        N.check(N.lib().sh_dist_reduce_bus_lagged(bus_f64.handle, nvalues, root, bus_f32.handle, slot))

    def wait_slot_keep(self, slot: int) -> None:
        N.check(N.lib().sh_dist_wait_slot_keep(slot))

    def sync(self) -> None:
        N.sync()

    def download(self, buf, nvalues: int) -> np.ndarray:
        return buf.download(np.float32, nvalues)


class DistVoiceBank:
    """This rank's shard of a voice table + the reduce of the partial buses.

    With several ranks the exchange is pipelined and batched: blocks are rendered into a ring of slots, each
    slot holding ``batch`` consecutive blocks; when a slot is full one ``ncclReduce`` (and, on root, the
    rounding to float32) is enqueued for it on the communication stream, overlapping the render of the
    following blocks.  ``flush()`` sends a partly filled slot; ``sync()`` releases every FULL slot whose reduce is still held
    back and waits for everything enqueued -- a partly filled slot is only sent by ``flush()`` (call ``flush()`` then ``sync()``
    before reading the last blocks).

    The reduce of a full slot is held back: a render's partial buses are folded into the slot's float64 bus by the render two
    launches on, so two renders later the slot is complete on the device without anybody having ended the run of pipelined
    renders (``mark_slot`` records where the render streams stand), and ``_REDUCE_LAG`` renders later still the collective is
    enqueued behind those marks (``reduce_lagged``) -- by then they have fired, and nothing in a hardware queue waits.  Enqueued
    at once the reduce costs a pipeline drain per batch (1024 voices, 8 blocks per batch: 47.8 instead of 38 us per block).
    ``flush()`` enqueues what is still held back.

    ``backend`` (tests): an object with _HipBackend's methods that renders / reduces some other way -- the world-2
    gloo test on CPU drives this very ring (slot rotation, batching, back-pressure, flush) with the oracle as the
    renderer and gloo as the collective."""

    def __init__(self, voices: Sequence, gains: Sequence[Tuple[float, float]], rank: int, world: int, batch: int = 1,
                 backend=None) -> None:
        self.rank, self.world = rank, world
        self.batch = max(1, int(batch)) if world > 1 else 1
        self.total_voices = len(voices)
        lo, hi = shard_range(len(voices), rank, world)
        self.lo, self.hi = lo, hi
        if hi <= lo:
            raise ValueError("rank %d owns no voices (%d voices over %d ranks)" % (rank, len(voices), world))
        self.backend = backend if backend is not None else _HipBackend(voices[lo:hi], gains[lo:hi])
        self.local = getattr(self.backend, "local", None)
        self._bus64 = []          # per slot: float64 partial buses of `batch` blocks
        self._bus32 = []          # per slot: float32 results (meaningful on root)
        self._views = []          # per slot, per block: (float64 view, float32 view)
        self._cap = 0
        self._slot = 0
        self._fill = 0            # blocks rendered into the current slot
        self._root = 0
        self._launches = 0        # renders so far
        self._held = []           # full slots whose reduce is held back: [slot, values, root, launches when it filled up, marked]

    def _buffers(self, nframes: int) -> None:
        if nframes != self._cap:
            self.flush()
            self.backend.sync()
            B = self.backend
            nslots = B.nslots() if self.world > 1 else 1
            self._bus64 = [B.alloc(self.batch * nframes * 16) for _ in range(nslots)]
            self._bus32 = [B.alloc(self.batch * nframes * 8) for _ in range(nslots)]
            self._views = [[(B.view(b64, j * nframes * 16, nframes * 16), B.view(b32, j * nframes * 8, nframes * 8))
                            for j in range(self.batch)] for b64, b32 in zip(self._bus64, self._bus32)]
            self._cap = nframes
            self._slot = 0
            self._fill = 0

    def _close_slot(self) -> None:
        k = self._slot
        n = self._fill * self._cap * 2
        if hasattr(self.backend, "reduce_lagged"):
            self._held.append([k, n, self._root, self._launches, False])
        else:
            self.backend.reduce_async(self._bus64[k], n, self._root, self._bus32[k], k)
        self._slot = (k + 1) % len(self._bus64)
        self._fill = 0

    _REDUCE_LAG = 4               # renders between a slot's mark and its reduce (a ring of 4 slots leaves 3 batches of room)

    def _release_held(self, everything: bool = False) -> None:
        for h in self._held:
            if not h[4] and self._launches >= h[3] + 2:
                self.backend.mark_slot(h[0])
                h[4] = True
        lag = min(self._REDUCE_LAG, max(0, 2 * self.batch - 2))      # (a slot is reused 3 batches after it filled up)
        while self._held and (everything or (self._held[0][4] and self._launches >= self._held[0][3] + 2 + lag)):
            k, n, root, _at, _m = self._held.pop(0)
            self.backend.reduce_lagged(self._bus64[k], n, root, self._bus32[k], k)      # (falls back to the draining form by itself if a fold is still owed)

    def flush(self) -> None:
        """Enqueue the reduce of the current, partly filled slot (and of the full slots held back)."""
        if self.world > 1:
            if self._fill:
                self._close_slot()
            self._release_held(everything=True)

    def sync(self) -> None:
        """Enqueue the reduces of the full slots that are still held back, then wait for the device: every block of a FULL slot
        is final in its float32 view afterwards (before, a caller that rendered whole batches and called sync()
        without flush() read views whose collective had never been enqueued)."""
        if self.world > 1:
            self._release_held(everything=True)
        self.backend.sync()

    def render_device(self, nframes: int, start: int = 0, root: int = 0):
        """Render the shard's block; returns the float32 bus buffer of this block (meaningful on root once the
        slot's reduce has run: after ``flush()`` + ``sync()``)."""
        self._buffers(nframes)
        if self.world == 1:                      # nothing to exchange: the kernel rounds to float32 itself
            self.backend.render(nframes, start, self._bus32[0], None)
            return self._bus32[0]
        if self._fill and root != self._root:
            self.flush()                         # a slot is reduced to ONE root
        self._root = root
        k, j = self._slot, self._fill
        if j == 0:                               # the reduce that last used this slot must have finished
            (getattr(self.backend, "wait_slot_keep", None) or self.backend.wait_slot)(k)
        v64, v32 = self._views[k][j]
        self.backend.render(nframes, start, None, v64)
        self._launches += 1
        self._fill += 1
        if self._fill == self.batch:
            self._close_slot()
        self._release_held()
        return v32

    def render(self, nframes: int, start: int = 0, root: int = 0) -> Optional[np.ndarray]:
        buf = self.render_device(nframes, start, root)
        self.flush()
        self.backend.sync()
        if self.rank != root and self.world > 1:
            return None
        return self.backend.download(buf, nframes * 2).reshape(nframes, 2)


# -- Sample.resample sharded by output-frame range (no collective) ---------------------------

_RANGE_ALIGN = 256          # output ranges start at multiples of this many frames (the library needs 16)


def resample_ranges(out_frames: int, world: int) -> list:
    """[(first, count)] per rank: contiguous output-frame ranges, starts aligned to 256 frames, sizes within one
    alignment unit of each other; trailing ranks may be empty for tiny outputs."""
    if world < 1:
        raise ValueError("world must be >= 1")
    units = -(-out_frames // _RANGE_ALIGN)
    sizes = shard_sizes(units, world)
    out, first = [], 0
    for u in sizes:
        start = min(first * _RANGE_ALIGN, out_frames)
        end = min((first + u) * _RANGE_ALIGN, out_frames)
        out.append((start, end - start))
        first += u
    return out


def resample_span(in_frames: int, inrate: int, outrate: int, out_first: int, out_n: int) -> Tuple[int, int]:
    """(first, count) of the input frames an output range reads (host arithmetic in the library; needs no GPU)."""
    a, b = C.c_size_t(), C.c_size_t()
    N.check(N.lib().sh_resample_span(in_frames, inrate, outrate, out_first, out_n, C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


def resample_shard(frames, samplewidth: int, nchannels: int, inrate: int, outrate: int, rank: int, world: int,
                   is_float: bool = False) -> Tuple[int, bytes]:
    """This rank's part of ``audioop.ratecv(frames, width, nchannels, inrate, outrate, None)[0]``.

    ``frames`` is the whole input (bytes-like) as every rank sees it on the host (a file, shared memory); only the
    span this rank's output range reads -- its share plus a halo of one input frame -- is uploaded.  Returns
    (first output frame, PCM bytes); the ranks' byte strings concatenated in rank order are the full result, bit for
    bit, and no rank talks to another."""
    fb = samplewidth * nchannels
    view = memoryview(frames).cast("B")
    if len(view) % fb:
        raise ValueError("frames data is not a whole number of frames")
    in_frames = len(view) // fb
    L = N.lib()
    nout = L.sh_resample_out_frames(in_frames, inrate, outrate)
    first, count = resample_ranges(nout, world)[rank]
    if count == 0:
        return first, b""
    in_first, in_count = resample_span(in_frames, inrate, outrate, first, count)
    src = N.DeviceBuffer(in_count * fb)
    src.upload(np.frombuffer(view[in_first * fb:(in_first + in_count) * fb], dtype=np.uint8))
    dst = N.DeviceBuffer(count * fb)
    N.check(L.sh_resample_range(src.handle, in_first, in_count, nchannels, samplewidth, 1 if is_float else 0,
                                inrate, outrate, first, count, dst.handle))
    out = dst.download_bytes(count * fb)
    src.free()
    dst.free()
    return first, out
