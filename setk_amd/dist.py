"""
Utterance sharding over the GPUs of one node.

The reference parallelises this path with ``split_scp.pl`` + ``run.pl JOB=1:nj``
(scripts/run_adapt_beamformer.sh:69-92): contiguous scp shards, one process each, no
communication.  Here: one process per GPU (any launcher that exports RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT, e.g. ``python -m torch.distributed.run``); utterances
are independent units dealt to ranks by duration (longest first, each to the least loaded rank),
and the ranks exchange only the start / finish barrier and the counters of the final "Processed N
utterances" line.  There is no data-path collective because the path has no exchange step.

Backends of those few bytes (``Shard(backend=...)`` or SETK_DIST_BACKEND):

  rccl   (default on a GPU node) RCCL over xGMI through the library's own C entry points
         (setk_comm_*, csrc/comm.hip): no ``import torch`` in any rank -- 1 - 2 s of start-up
         per rank saved.  Rendezvous: rank 0's ncclUniqueId travels over a TCP socket on
         MASTER_ADDR:MASTER_PORT.
  tcp    the same exchange over that socket alone (rank 0 gathers and answers): for hosts
         without a GPU per rank -- the CPU tests of the multi-rank command line -- and as the
         fallback when librccl cannot be loaded.
  nccl / gloo   torch.distributed, as in rounds 1 - 4; taken only when asked for, or when the
         loaded library predates setk_comm_*.
"""
import os
import socket
import struct
import time


def _recv_exact(sock, n):
    buf = b""
    while len(buf) < n:
        part = sock.recv(n - len(buf))
        if not part:
            raise ConnectionError("peer closed the rendezvous socket")
        buf += part
    return buf


_MAGIC = b"SETK"


class _Star(object):
    """Rank 0 listens on MASTER_ADDR:MASTER_PORT, ranks 1..W-1 connect (with retries while the
    listener comes up) and stay connected: a W-way exchange is one round trip through rank 0.
    Carries the RCCL rendezvous (128 bytes) and, as backend "tcp", the barrier and the sums."""

    def __init__(self, rank, world, addr, port, timeout=None):
        # The rendezvous is lazy (first collective), and for most command lines that is the
        # closing barrier: ranks may arrive as far apart as their shards' run times differ.  So the
        # wait for peers is long (SETK_DIST_TIMEOUT seconds, default a day: the launcher ends the
        # other ranks when one dies) and an established socket never times out -- TCP keep-alive
        # reports a peer that vanished.
        if timeout is None:
            timeout = float(os.environ.get("SETK_DIST_TIMEOUT", "86400"))
        self.rank, self.world = rank, world
        self.peers = []
        self.sock = None
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", port))
            srv.listen(world)
            deadline = time.time() + timeout
            got = {}
            while len(got) < world - 1:
                srv.settimeout(max(0.05, deadline - time.time()))
                c, _ = srv.accept()
                try:
                    c.settimeout(10.0)   # a stray connection must not stall the job
                    magic, r = struct.unpack("<4si", _recv_exact(c, 8))
                except (OSError, ConnectionError, struct.error):
                    c.close()
                    continue
                if magic != _MAGIC or not 1 <= r < world or r in got:
                    c.close()    # not one of this job's ranks (or a duplicate): refused
                    continue
                self._established(c)
                got[r] = c
            srv.close()
            self.peers = [got[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.sendall(_MAGIC + struct.pack("<i", rank))
            self._established(s)
            self.sock = s

    @staticmethod
    def _established(sock):
        sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        sock.setsockopt(socket.SOL_SOCKET, socket.SO_KEEPALIVE, 1)
        sock.settimeout(None)

    def broadcast(self, payload, nbytes):
        """rank 0's `payload` (bytes) to everybody."""
        if self.rank == 0:
            for c in self.peers:
                c.sendall(payload)
            return payload
        return _recv_exact(self.sock, nbytes)

    def allreduce(self, values, op="sum"):
        n = len(values)
        fmt = "<%dd" % n
        if self.rank == 0:
            acc = [float(v) for v in values]
            for c in self.peers:
                other = struct.unpack(fmt, _recv_exact(c, 8 * n))
                acc = [max(a, b) if op == "max" else a + b for a, b in zip(acc, other)]
            out = struct.pack(fmt, *acc)
            for c in self.peers:
                c.sendall(out)
            return acc
        self.sock.sendall(struct.pack(fmt, *[float(v) for v in values]))
        return list(struct.unpack(fmt, _recv_exact(self.sock, 8 * n)))

    def close(self):
        for c in self.peers + ([self.sock] if self.sock else []):
            try:
                c.close()
            except OSError:
                pass
        self.peers, self.sock = [], None


class Shard:
    """rank/world view of the job; degenerates to a single process."""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self._dist = None    # torch.distributed (legacy backends)
        self._star = None    # the TCP star (rendezvous, backend "tcp")
        self._comm = None    # setk_comm_t (backend "rccl")
        self._lib = None
        self.backend = None
        self.assigned_weight = 0.0
        if self.world <= 1:
            return
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # the ranks can only agree on a port through their launcher
            raise RuntimeError("WORLD_SIZE > 1 without MASTER_PORT: start the ranks with "
                               "`python -m torch.distributed.run --master-addr 127.0.0.1 "
                               "--master-port <free port> ...` (or export MASTER_PORT)")
        backend = backend or os.environ.get("SETK_DIST_BACKEND") or "rccl"
        if backend in ("rccl", "tcp"):
            # connected at the first collective: a command line decides whether it runs
            # torch-free (which fixes how the library is loaded) after it has seen its inputs
            self._want = backend
            return
        self._init_torch(backend)

    _want = None

    def _ensure(self):
        if self._want is None:
            return
        backend, self._want = self._want, None
        # (the launcher's own store listens on MASTER_PORT: the star takes the next port)
        addr, port = os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]) + 1 + \
            int(os.environ.get("SETK_DIST_PORT_OFFSET", "0"))
        self._star = _Star(self.rank, self.world, addr, port)
        if backend == "rccl" and not self._init_rccl():
            backend = "tcp"
        self.backend = backend

    def _init_rccl(self):
        """RCCL through the C ABI.  Every rank learns whether ALL of them can (library with
        setk_comm_*, librccl loadable, a GPU): one cannot fall back alone."""
        import ctypes
        from . import _ffi
        lib, ok, uid = None, 1.0, b"\0" * 128
        try:
            lib = _ffi.load_library()
            if not hasattr(lib, "setk_comm_create") or not os.path.exists("/dev/kfd"):
                ok = 0.0  # (a library from before round 5, or no GPU driver on this host)
            else:
                # this rank's GPU must be usable BEFORE the collective ncclCommInitRank: a rank that
                # fails locally there would leave the others waiting in it.  (The context is the one
                # the command line creates anyway: `default_context` caches it per device.)
                _ffi.default_context(self.local_rank)
            if ok and self.rank == 0:
                buf = ctypes.create_string_buffer(128)
                if lib.setk_comm_unique_id(buf) != 0:
                    ok = 0.0
                uid = buf.raw
        except Exception:  # noqa: BLE001  (no library / no GPU: the star alone carries the job)
            ok = 0.0
        if sum(self._star.allreduce([ok])) < self.world:
            return False
        uid = self._star.broadcast(uid, 128)
        comm = ctypes.c_void_p()
        lib.setk_comm_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_char_p,
                                         ctypes.c_int, ctypes.c_int]
        lib.setk_comm_allreduce_f64.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                                ctypes.c_int, ctypes.c_int]
        lib.setk_comm_barrier.argtypes = [ctypes.c_void_p]
        lib.setk_comm_destroy.argtypes = [ctypes.c_void_p]
        lib.setk_comm_last_error.restype = ctypes.c_char_p
        rc = lib.setk_comm_create(ctypes.byref(comm), self.local_rank, uid, self.rank, self.world)
        good = 1.0 if rc == 0 else 0.0
        if sum(self._star.allreduce([good])) < self.world:
            if rc == 0:
                lib.setk_comm_destroy(comm)
            return False
        self._lib, self._comm = lib, comm
        return True

    def _init_torch(self, backend):
        import torch
        import torch.distributed as dist
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if not dist.is_initialized():
            kw = {}
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                kw["device_id"] = torch.device("cuda", self.local_rank)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
        self._dist = dist
        self.backend = backend

    @property
    def device(self):
        return self.local_rank

    @property
    def torch_free_ok(self):
        """A command line may run without torch: alone, or with the library's own collectives
        (the torch.distributed backends bring torch in themselves)."""
        return self.world <= 1 or self._dist is None

    def assign(self, keys, weights=None):
        """Keys owned by this rank.  With weights (e.g. durations) the keys are
        dealt longest first, each to the least loaded rank, which balances the sum
        of weights; without, plain round robin in table order."""
        return assign_keys(keys, self.rank, self.world, weights)

    def assign_by_duration(self, wav_reader, keys=None):
        """Keys of this rank, dealt longest-first / least-loaded on the sample counts
        in the wave headers (utterances whose length cannot be read from a header --
        pipes, per-channel globs -- count as the mean of the others)."""
        keys = list(wav_reader.index_keys if keys is None else keys)
        if self.world <= 1:
            return keys
        lens = [wav_reader.peek_nsamps(k) for k in keys]
        known = [n for n in lens if n]
        mean = (sum(known) / len(known)) if known else 1.0
        weights = [n if n else mean for n in lens]
        mine = assign_keys(keys, self.rank, self.world, weights)
        wsum = dict(zip(keys, weights))
        self.assigned_weight = sum(wsum[k] for k in mine)
        return mine

    def _allreduce(self, values, op="sum"):
        self._ensure()
        values = [float(v) for v in values]
        if self._comm is not None:
            import ctypes
            out = []
            for i in range(0, len(values), 64):
                chunk = values[i:i + 64]
                arr = (ctypes.c_double * len(chunk))(*chunk)
                rc = self._lib.setk_comm_allreduce_f64(self._comm, arr, len(chunk), 1 if op == "max" else 0)
                if rc != 0:
                    raise RuntimeError("setk_comm_allreduce_f64: " +
                                       (self._lib.setk_comm_last_error() or b"?").decode())
                out.extend(arr)
            return out
        if self._star is not None:
            return self._star.allreduce(values, op)
        import torch
        dev = torch.device("cuda", self.local_rank) if self.backend == "nccl" else "cpu"
        t = torch.tensor(values, dtype=torch.float64, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX if op == "max" else self._dist.ReduceOp.SUM)
        return t.cpu().tolist()

    def barrier(self):
        if self.world > 1:
            self._allreduce([1.0])

    def sum_counts(self, values):
        """Element-wise sum of a short list of python numbers over all ranks."""
        if self.world <= 1:
            return list(values)
        return self._allreduce(values, "sum")

    def max_values(self, values):
        """Element-wise maximum over all ranks (wall clocks)."""
        if self.world <= 1:
            return list(values)
        return self._allreduce(values, "max")

    def close(self):
        if self._comm is not None:
            self._lib.setk_comm_destroy(self._comm)
            self._comm = None
        if self._star is not None:
            self._star.close()
            self._star = None
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
            self._dist = None


def assign_keys(keys, rank, world, weights=None):
    keys = list(keys)
    if world <= 1:
        return keys
    if weights is None:
        return keys[rank::world]
    # longest first, each to the least loaded rank so far (LPT; ties -> lowest
    # rank): every rank evaluates the same deterministic deal
    order = sorted(range(len(keys)), key=lambda i: (-float(weights[i]), i))
    load = [0.0] * world
    owner = {}
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += float(weights[i])
    mine = sorted(i for i, r in owner.items() if r == rank)  # table order inside a rank
    return [keys[i] for i in mine]
