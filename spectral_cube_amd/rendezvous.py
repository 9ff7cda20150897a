"""Host-side rendezvous for the one-process-per-GPU driver, without torch.

The only things the ranks of one node have to tell each other on the host are
tiny: the 128-byte RCCL unique id (rank 0 -> everyone), a few statistics records,
and "I am here" for barriers.  A shared directory does that: every collective is
one small file per rank, written atomically (temp file + rename) and polled by
the others.  No sockets, no third-party package; works for any launcher that gives
every rank RANK / WORLD_SIZE (``python -m torch.distributed.run`` does - its
environment is read, torch itself is not imported).

A directory may outlive a launch (a fixed SPC_RDV_DIR, a torchrun restart after a crash that never
reached close()): the ranks therefore open every launch with a HANDSHAKE that gives it a fresh
session id, and every collective's file names carry that id - files of an earlier launch are never
read, whatever they hold (see FileRendezvous._handshake).

The transport protocol (what ``distributed.py`` asks of a rendezvous object):

    rank, world_size
    bcast_bytes(payload_or_None) -> bytes        rank 0's payload on every rank
    allgather_bytes(payload) -> [bytes] * world  in rank order
    barrier()

``tests/test_distributed_cpu.py`` runs the same drivers over a torch.distributed
(gloo) transport with this protocol, and over this class, with world_size 2.
"""
import json
import os
import secrets
import stat
import shutil
import tempfile
import time


class RendezvousTimeout(RuntimeError):
    pass


# ---- payload encoding ----------------------------------------------------------------------------
# The directory is shared state on the node: what is read back from it is DATA, never code.  Objects travel as JSON
# (deserialising Python object streams from a planted file would run them); tuples and numpy scalars / small arrays -
# what the drivers exchange: ranks, timings, statistics records, digests - keep their type through tags.
def _enc(o):
    import numpy as np
    if isinstance(o, tuple):
        return {"__tuple__": [_enc(x) for x in o]}
    if isinstance(o, list):
        return [_enc(x) for x in o]
    if isinstance(o, dict):
        if not all(isinstance(k, str) for k in o):
            raise TypeError("rendezvous objects take string dictionary keys")
        return {"__dict__": {k: _enc(v) for k, v in o.items()}}
    if isinstance(o, (bytes, bytearray)):
        return {"__bytes__": bytes(o).hex()}
    if isinstance(o, np.ndarray):
        return {"__nd__": [o.dtype.str, list(o.shape), np.ascontiguousarray(o).tobytes().hex()]}
    if isinstance(o, np.generic):
        return _enc(o.item())
    if isinstance(o, float):
        if o != o or o in (float("inf"), float("-inf")):
            return {"__float__": repr(o)}
        return o
    if o is None or isinstance(o, (bool, int, str)):
        return o
    raise TypeError("cannot send %s through the rendezvous" % type(o).__name__)


def _dec(o):
    if isinstance(o, list):
        return [_dec(x) for x in o]
    if isinstance(o, dict):
        if "__tuple__" in o:
            return tuple(_dec(x) for x in o["__tuple__"])
        if "__dict__" in o:
            return {k: _dec(v) for k, v in o["__dict__"].items()}
        if "__bytes__" in o:
            return bytes.fromhex(o["__bytes__"])
        if "__float__" in o:
            return float(o["__float__"])
        if "__nd__" in o:
            import numpy as np
            dt, shape, hx = o["__nd__"]
            return np.frombuffer(bytes.fromhex(hx), dtype=np.dtype(dt)).reshape(shape).copy()
        raise ValueError("unknown rendezvous tag")
    return o


def dumps(obj):
    return json.dumps(_enc(obj), separators=(",", ":")).encode("ascii")


def loads(raw):
    return _dec(json.loads(bytes(raw).decode("ascii")))


class FileRendezvous:
    """Collectives of small host payloads through files in *path* (single node)."""

    def __init__(self, path, rank, world_size, timeout=300.0):
        if world_size < 1 or not (0 <= rank < world_size):
            raise ValueError("bad rank %d / world %d" % (rank, world_size))
        self.path, self.rank, self.world_size, self.timeout = os.fspath(path), int(rank), int(world_size), timeout
        self._seq = 0
        os.makedirs(self.path, mode=0o700, exist_ok=True)
        # the directory name is predictable (/dev/shm/spc_rdv_<port>_<ppid>_...): it must be OURS and closed to others,
        # or whoever made it first could feed the ranks their payloads
        st = os.stat(self.path)
        if st.st_uid != os.geteuid():
            raise PermissionError("rendezvous directory %s belongs to uid %d, not to this process (uid %d)"
                                  % (self.path, st.st_uid, os.geteuid()))
        if stat.S_IMODE(st.st_mode) & 0o077:
            os.chmod(self.path, 0o700)
        self._sid = self._handshake()

    # ---- construction from the launcher's environment ----------------------------------------
    @classmethod
    def from_env(cls, env=None, timeout=300.0):
        """RANK / WORLD_SIZE from the environment; the directory is SPC_RDV_DIR, or one that
        every rank of THIS launch derives alike: the launcher's pid (all ranks are its children)
        and MASTER_PORT keep two launches on one box apart."""
        env = os.environ if env is None else env
        rank, world = int(env.get("RANK", 0)), int(env.get("WORLD_SIZE", 1))
        path = env.get("SPC_RDV_DIR")
        if not path:
            base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
            # the launcher's pid and port keep two launches apart, the elastic run id / restart count two
            # ATTEMPTS of one launch (torchrun restarts its workers under the same agent pid)
            attempt = "%s_%s" % (env.get("TORCHELASTIC_RUN_ID", "none"), env.get("TORCHELASTIC_RESTART_COUNT", "0"))
            attempt = "".join(ch if ch.isalnum() or ch in "_-" else "_" for ch in attempt)
            path = os.path.join(base, "spc_rdv_%s_%d_%s" % (env.get("MASTER_PORT", "0"), os.getppid(), attempt))
        return cls(path, rank, world, timeout)

    # ---- handshake: a session id no earlier launch in this directory can have ----------------------
    def _put(self, name, payload):
        tmp = os.path.join(self.path, name + ".tmp%d" % os.getpid())
        with open(tmp, "wb") as fh:
            fh.write(payload)
        os.replace(tmp, os.path.join(self.path, name))          # atomic: readers never see a partial file

    def _get(self, name):
        try:
            with open(os.path.join(self.path, name), "rb") as fh:
                return fh.read()
        except FileNotFoundError:
            return None

    def _handshake(self):
        """Every rank publishes a random nonce (hello.<rank>); rank 0 publishes `session` = its own fresh
        session id + the nonces it has seen, and republishes whenever a hello changes; a rank accepts a session
        only if it lists the rank's OWN nonce, and acknowledges it by id; rank 0 returns once every rank has
        acknowledged the current id.  A stale hello / session / ack left by an earlier launch holds another
        nonce or id, so it is ignored until its owner overwrites it - nothing depends on clocks or on the
        directory being empty."""
        nonce = secrets.token_hex(8)
        self._put("hello.%d" % self.rank, nonce.encode())
        t0, delay = time.monotonic(), 0.0002

        def wait():
            nonlocal delay
            if time.monotonic() - t0 > self.timeout:
                raise RendezvousTimeout("rank %d: no handshake within %.0f s in %s" % (self.rank, self.timeout, self.path))
            time.sleep(delay)
            delay = min(delay * 1.5, 0.005)

        if self.rank == 0:
            sid, seen = None, None
            while True:
                hellos = [nonce.encode()] + [self._get("hello.%d" % r) for r in range(1, self.world_size)]
                if all(h is not None for h in hellos):
                    if hellos != seen:
                        seen, sid = hellos, secrets.token_hex(8)
                        self._put("session", dumps((sid, [h.decode("ascii", "replace") for h in hellos])))
                    if all(self._get("ack.%d" % r) == sid.encode() for r in range(1, self.world_size)):
                        self._put("go", sid.encode())
                        return sid
                wait()
        while True:
            raw = self._get("session")
            if raw is not None:
                sid, hellos = self._session(raw, None)
                if sid is not None and len(hellos) == self.world_size and hellos[self.rank] == nonce:
                    self._put("ack.%d" % self.rank, sid.encode())
                    # rank 0 may still re-issue the session if ANOTHER rank's hello was stale when it read it:
                    # the first collective's file name carries the id, so wait until rank 0 has settled on it
                    return self._settled(sid, nonce)
            wait()

    @staticmethod
    def _session(raw, default):
        """(session id, [nonce per rank]) of a `session` file; (default, ()) for anything that is not one"""
        try:
            sid, hellos = loads(raw)
            if not isinstance(sid, str) or not isinstance(hellos, list) or not all(isinstance(h, str) for h in hellos):
                raise ValueError
            return sid, hellos
        except Exception:
            return default, ()

    def _settled(self, sid, nonce):
        """rank 0 writes `go.<sid>` once every acknowledgement matches; until then a newer session may appear"""
        t0, delay = time.monotonic(), 0.0002
        while True:
            if self._get("go") == sid.encode():
                return sid
            raw = self._get("session")
            if raw is not None:
                cur, hellos = self._session(raw, sid)
                if cur != sid and len(hellos) == self.world_size and hellos[self.rank] == nonce:
                    sid = cur
                    self._put("ack.%d" % self.rank, sid.encode())
            if time.monotonic() - t0 > self.timeout:
                raise RendezvousTimeout("rank %d: session never settled in %s" % (self.rank, self.path))
            time.sleep(delay)
            delay = min(delay * 1.5, 0.005)

    # ---- files -------------------------------------------------------------------------------
    def _name(self, seq, rank):
        return os.path.join(self.path, "%s.%08d.%d" % (self._sid, seq, rank))

    def _write(self, seq, payload):
        tmp = self._name(seq, self.rank) + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as fh:
            fh.write(payload)
        os.replace(tmp, self._name(seq, self.rank))          # atomic: readers never see a partial file

    def _read(self, seq, rank):
        name = self._name(seq, rank)
        t0, delay = time.monotonic(), 0.0002
        while True:
            try:
                with open(name, "rb") as fh:
                    return fh.read()
            except FileNotFoundError:
                if time.monotonic() - t0 > self.timeout:
                    raise RendezvousTimeout("rank %d waited %.0f s for rank %d (step %d) in %s"
                                            % (self.rank, self.timeout, rank, seq, self.path))
                time.sleep(delay)
                delay = min(delay * 1.5, 0.005)

    def _retire(self, seq):
        # every rank has written step `seq`, so every rank has finished READING step seq - 1
        if seq >= 1:
            try:
                os.unlink(self._name(seq - 1, self.rank))
            except FileNotFoundError:
                pass

    # ---- collectives -------------------------------------------------------------------------
    def allgather_bytes(self, payload):
        seq = self._seq
        self._seq += 1
        self._write(seq, bytes(payload))
        out = [self._read(seq, r) if r != self.rank else bytes(payload) for r in range(self.world_size)]
        self._retire(seq)
        return out

    def bcast_bytes(self, payload=None):
        """rank 0's payload (others pass None)"""
        if self.rank == 0 and payload is None:
            raise ValueError("rank 0 must provide the payload")
        return self.allgather_bytes(payload if self.rank == 0 else b"")[0]

    def allgather_object(self, obj):
        return [loads(b) for b in self.allgather_bytes(dumps(obj))]

    def barrier(self):
        self.allgather_bytes(b"")

    def close(self):
        """last collective: every rank leaves a marker once it needs nothing from the directory any
        more; rank 0 waits for all of them and removes it"""
        try:
            self.barrier()
            self._seq += 1
            self._write(self._seq, b"bye")
            if self.rank == 0:
                for r in range(1, self.world_size):
                    self._read(self._seq, r)
        except (RendezvousTimeout, OSError):
            pass
        if self.rank == 0:
            shutil.rmtree(self.path, ignore_errors=True)


class SingleProcess:
    """world_size 1: the same protocol without any exchange"""

    rank, world_size = 0, 1

    def allgather_bytes(self, payload):
        return [bytes(payload)]

    def bcast_bytes(self, payload=None):
        return payload

    def allgather_object(self, obj):
        return [obj]

    def barrier(self):
        pass

    def close(self):
        pass
