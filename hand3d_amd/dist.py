"""Batch sharding across the GPUs of one node: every image is independent, so the path shards with NO
data-path collective.  Two tiny exchanges remain, both RCCL over xGMI on the engine's own stream through the C ABI
(include/hp3d.h: hp3d_comm_init / hp3d_bcast_weights / hp3d_allgather[_dev]):
  * once: broadcast of the packed weight blob from rank 0;
  * per batch: all-gather of the [B/n,21,3] keypoints (252 B/image).
The reference has no counterpart (single tf.Session everywhere, run.py:50).

No PyTorch is needed: one process per GPU is started by any launcher that exports RANK / WORLD_SIZE / LOCAL_RANK /
MASTER_ADDR / MASTER_PORT (torch.distributed.run does; so does a shell loop), and the only thing the processes have to
agree on before RCCL exists -- the 128-byte communicator id -- travels over a plain TCP socket (`Rendezvous`), which
also carries the host-side scalars of the benchmark protocol (barrier, max of the per-rank wall times).

Nothing in this package imports torch.  Callers that already live inside a torch process group find torch.distributed
variants of the two exchanges in `examples/torch_dist_helpers.py` (outside the product package; two-rank gloo test).
"""
import hashlib
import hmac
import os
import socket
import threading
import struct
import time

import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of rank `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_items, world):
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


# ------------------------------------------------------------------------------------------- TCP rendezvous
_MAGIC = b'HP3DRDZV2'
_MAX_MSG = 1 << 28            # 256 MB: far above anything the protocol carries (ids, scalars, keypoint arrays)


# Wire format of the payloads: a small tagged encoding of exactly the value kinds the protocol carries (None, bool, int,
# float, str, bytes, ndarray, list / tuple, dict with str keys).  Nothing a peer sends is ever executed or un-pickled.
def _enc(obj, out):
    if obj is None:
        out.append(b'N')
    elif isinstance(obj, (bool, np.bool_)):
        out.append(b'T' if obj else b'F')
    elif isinstance(obj, (int, np.integer)):
        out.append(b'I' + struct.pack('<q', int(obj)))
    elif isinstance(obj, (float, np.floating)):
        out.append(b'D' + struct.pack('<d', float(obj)))
    elif isinstance(obj, str):
        b = obj.encode('utf-8')
        out.append(b'S' + struct.pack('<Q', len(b)) + b)
    elif isinstance(obj, (bytes, bytearray, memoryview)):
        b = bytes(obj)
        out.append(b'B' + struct.pack('<Q', len(b)) + b)
    elif isinstance(obj, np.ndarray):
        if obj.dtype.hasobject:
            raise TypeError("rendezvous: object arrays cannot be sent")
        ds = obj.dtype.str.encode('ascii')
        raw = np.ascontiguousarray(obj).tobytes()
        out.append(b'A' + struct.pack('<BB', len(ds), obj.ndim) + ds + struct.pack('<%dq' % obj.ndim, *obj.shape) +
                   struct.pack('<Q', len(raw)) + raw)
    elif isinstance(obj, (list, tuple)):
        out.append(b'L' + struct.pack('<Q', len(obj)))
        for x in obj:
            _enc(x, out)
    elif isinstance(obj, dict):
        out.append(b'M' + struct.pack('<Q', len(obj)))
        for k, v in obj.items():
            if not isinstance(k, str):
                raise TypeError("rendezvous: dict keys must be str")
            _enc(k, out)
            _enc(v, out)
    else:
        raise TypeError("rendezvous: cannot send %r" % type(obj))


_MAX_DEPTH = 16          # nesting of lists / maps a message may have (a hostile peer must not be able to exhaust the stack)


def _dec(buf, pos, depth=0):
    if depth > _MAX_DEPTH:
        raise ValueError("rendezvous: message nested deeper than %d" % _MAX_DEPTH)
    tag = buf[pos:pos + 1]
    pos += 1
    if tag == b'N':
        return None, pos
    if tag in (b'T', b'F'):
        return tag == b'T', pos
    if tag == b'I':
        return struct.unpack_from('<q', buf, pos)[0], pos + 8
    if tag == b'D':
        return struct.unpack_from('<d', buf, pos)[0], pos + 8
    if tag in (b'S', b'B'):
        (n,) = struct.unpack_from('<Q', buf, pos)
        pos += 8
        if n > len(buf) - pos:
            raise ValueError("rendezvous: truncated message")
        raw = bytes(buf[pos:pos + n])
        return (raw.decode('utf-8') if tag == b'S' else raw), pos + n
    if tag == b'A':
        ld, nd = struct.unpack_from('<BB', buf, pos)
        pos += 2
        dt = np.dtype(bytes(buf[pos:pos + ld]).decode('ascii'))
        pos += ld
        if dt.hasobject:
            raise ValueError("rendezvous: object arrays are refused")
        shape = struct.unpack_from('<%dq' % nd, buf, pos)
        pos += 8 * nd
        (n,) = struct.unpack_from('<Q', buf, pos)
        pos += 8
        if n > len(buf) - pos or any(d < 0 for d in shape) or n != int(np.prod(shape, dtype=np.int64)) * dt.itemsize:
            raise ValueError("rendezvous: bad array header")
        return np.frombuffer(buf, dtype=dt, count=n // dt.itemsize, offset=pos).reshape(shape).copy(), pos + n
    if tag in (b'L', b'M'):
        (n,) = struct.unpack_from('<Q', buf, pos)
        pos += 8
        if n > len(buf) - pos:              # every element takes at least one byte
            raise ValueError("rendezvous: truncated message")
        if tag == b'L':
            out = []
            for _ in range(n):
                x, pos = _dec(buf, pos, depth + 1)
                out.append(x)
            return out, pos
        d = {}
        for _ in range(n):
            k, pos = _dec(buf, pos, depth + 1)
            v, pos = _dec(buf, pos, depth + 1)
            if not isinstance(k, str):
                raise ValueError("rendezvous: bad dict key")
            d[k] = v
        return d, pos
    raise ValueError("rendezvous: unknown tag %r" % tag)


def _send_msg(sock, obj):
    parts = []
    _enc(obj, parts)
    data = b''.join(parts)
    sock.sendall(struct.pack('<Q', len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    if n > _MAX_MSG:
        raise ValueError("rendezvous: message of %d bytes refused" % n)
    buf = _recv_exact(sock, n)
    obj, pos = _dec(buf, 0)
    if pos != n:
        raise ValueError("rendezvous: trailing bytes in message")
    return obj


def rendezvous_ports(master_port):
    """The launcher's own store listens on MASTER_PORT; the rendezvous takes the first free port of a fixed sequence
    derived from it (every rank walks the same sequence; a handshake tells a foreign listener from rank 0)."""
    base = int(master_port)
    return [1024 + (base - 1024 + 977 + 131 * k) % (65536 - 1024) for k in range(8)]


def _is_loopback(addr):
    return addr in ('localhost', '::1') or addr.startswith('127.')


def _mac(secret, *parts):
    return hmac.new(secret, b'|'.join(parts), hashlib.sha256).digest()


class Rendezvous(object):
    """Star over TCP with rank 0 as the hub: allgather / broadcast / barrier of small values (None, numbers, str, bytes,
    ndarrays, lists, str-keyed dicts -- a fixed tagged encoding, nothing is un-pickled).  World size 1 needs no socket.

    Joining is a fixed-size challenge-response, checked BEFORE any payload is parsed: the hub sends a 16-byte nonce, the
    spoke answers magic + rank + HMAC-SHA256(secret, nonce | rank | world), the hub proves itself with
    HMAC(secret, nonce | "hub").  `secret` comes from the launcher (env HP3D_RDZV_SECRET: bench.py's self-launch draws a
    random one per run); without it the key is only the public "port:world" string, which keeps strangers' stray
    connections and foreign listeners apart but authenticates nobody -- so the hub then insists on a loopback address
    unless HP3D_RDZV_ALLOW_REMOTE=1."""

    def __init__(self, rank, world, addr='127.0.0.1', port=29500, timeout=300.0, token=None, secret=None):
        self.rank, self.world = int(rank), int(world)
        self.peers = {}          # hub: rank -> socket
        self.sock = None         # spoke: socket to the hub
        self.token = (token if token is not None else '%s:%d' % (port, self.world))
        if self.world == 1:
            return
        if secret is None:
            secret = os.environ.get('HP3D_RDZV_SECRET')
        has_secret = bool(secret)
        key = (secret if has_secret else self.token)
        key = key if isinstance(key, bytes) else str(key).encode('utf-8')
        wtag = str(self.world).encode('ascii')
        ports = rendezvous_ports(port)
        deadline = time.time() + timeout
        if self.rank == 0:
            if not has_secret and not _is_loopback(addr) and os.environ.get('HP3D_RDZV_ALLOW_REMOTE') != '1':
                raise RuntimeError("rendezvous: refusing to listen on %s without HP3D_RDZV_SECRET (set it on every rank, or "
                                   "HP3D_RDZV_ALLOW_REMOTE=1 on a trusted network)" % addr)
            srv = None
            for p in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, p))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError("rendezvous: none of the ports %s could be bound on %s" % (ports, addr))
            srv.listen(self.world)
            srv.settimeout(1.0)
            while len(self.peers) < self.world - 1:
                if time.time() > deadline:
                    raise TimeoutError("rendezvous: %d of %d ranks joined" % (len(self.peers) + 1, self.world))
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    continue
                try:
                    c.settimeout(2.0)            # the handshake is two short messages on a local link: a silent stranger costs 2 s, not 10
                    nonce = os.urandom(16)
                    c.sendall(_MAGIC + nonce)
                    hello = _recv_exact(c, len(_MAGIC) + 4 + 32)          # fixed size; nothing is parsed before the MAC holds
                    (prank,) = struct.unpack('<I', hello[len(_MAGIC):len(_MAGIC) + 4])
                    ok = hello[:len(_MAGIC)] == _MAGIC and 0 < prank < self.world and prank not in self.peers and \
                        hmac.compare_digest(hello[len(_MAGIC) + 4:], _mac(key, nonce, str(prank).encode('ascii'), wtag))
                    if not ok:
                        c.close()
                        continue
                    c.sendall(_mac(key, nonce, b'hub'))
                    c.settimeout(timeout)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self.peers[prank] = c
                except Exception:                # a stranger, a scanner, a half-open connection: drop it, keep listening
                    try:
                        c.close()
                    except Exception:
                        pass
            srv.close()
        else:
            k = 0
            while self.sock is None:
                if time.time() > deadline:
                    raise TimeoutError("rendezvous: rank %d could not reach rank 0 on %s ports %s" % (self.rank, addr, ports))
                p = ports[k % len(ports)]
                k += 1
                s = None
                try:
                    s = socket.create_connection((addr, p), timeout=2.0)
                    s.settimeout(10.0)
                    first = _recv_exact(s, len(_MAGIC) + 16)
                    if first[:len(_MAGIC)] != _MAGIC:
                        raise ConnectionError("not a rendezvous hub")
                    nonce = first[len(_MAGIC):]
                    s.sendall(_MAGIC + struct.pack('<I', self.rank) + _mac(key, nonce, str(self.rank).encode('ascii'), wtag))
                    if not hmac.compare_digest(_recv_exact(s, 32), _mac(key, nonce, b'hub')):
                        raise ConnectionError("the listener does not hold the rendezvous secret")
                    s.settimeout(timeout)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self.sock = s
                except (OSError, ConnectionError):
                    try:
                        if s is not None:
                            s.close()
                    except Exception:
                        pass
                    if k % len(ports) == 0:
                        time.sleep(0.2)

    @classmethod
    def from_env(cls, timeout=300.0):
        return cls(int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
                   os.environ.get('MASTER_ADDR', '127.0.0.1'), int(os.environ.get('MASTER_PORT', '29500')), timeout)

    def allgather(self, obj):
        """[obj of rank 0, ..., obj of rank world-1] on every rank."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            objs = [obj] + [None] * (self.world - 1)
            for r, s in self.peers.items():
                objs[r] = _recv_msg(s)
            for s in self.peers.values():
                _send_msg(s, objs)
            return objs
        _send_msg(self.sock, obj)
        return _recv_msg(self.sock)

    def broadcast(self, obj, src=0):
        return self.allgather(obj if self.rank == src else None)[src]

    def barrier(self):
        self.allgather(None)

    def max(self, x):
        return max(self.allgather(float(x)))

    def close(self):
        for s in list(self.peers.values()) + ([self.sock] if self.sock else []):
            try:
                s.close()
            except Exception:
                pass
        self.peers, self.sock = {}, None


# ------------------------------------------------------------------------------------------- native (RCCL via the C ABI)
class ShardedPipeline(object):
    """One rank's share of a sharded batch: the engine, its RCCL communicator and the rendezvous that set it up.
    `weights` is needed on rank 0 only."""

    def __init__(self, engine, rank=0, world=1, rdzv=None):
        self.engine, self.rank, self.world = engine, int(rank), int(world)
        self.rdzv = rdzv if rdzv is not None else Rendezvous(rank, world) if world == 1 else None
        if self.rdzv is None:
            raise ValueError("world > 1 needs a Rendezvous (Rendezvous.from_env())")
        self.comm_ready = False
        self.comm_abandoned = False   # a set-up call ran into its deadline: the communicator is left alone for good
        self.tcp_only = False         # set by use_tcp_only(): gathers travel over the rendezvous sockets, not RCCL

    def use_tcp_only(self):
        """Degraded mode for a box whose RCCL communicator cannot be built: every rank loads its own weights and the (tiny)
        per-step keypoint gather travels over the rendezvous' TCP sockets.  Must be entered by ALL ranks."""
        if self.comm_ready and not self.comm_abandoned:
            try:
                self.engine.comm_destroy()
            except Exception:
                pass
        self.comm_ready = False
        self.tcp_only = True

    def _with_deadline(self, what, fn, *args):
        """RCCL calls that set a communicator up (ncclCommInitRank, the first broadcast) have been seen neither to return nor to fail
        (profiles/r05_tuning_notes.md section 15: two ranks on one device); at eight ranks that is the launcher's 1800 s and no JSON
        line.  So they run in a worker thread (ctypes releases the GIL for the call) and the caller waits at most HP3D_RCCL_TIMEOUT
        seconds (default 120; 0 = wait forever).  On expiry a TimeoutError whose text starts with "rccl init timeout" is raised: the
        caller (bench.py) then agrees with the other ranks over the rendezvous and takes the TCP path.  The stuck call is abandoned:
        its thread is a daemon, and the communicator is never touched again (close() does not destroy it)."""
        timeout = float(os.environ.get('HP3D_RCCL_TIMEOUT', '120'))
        if timeout <= 0:
            return fn(*args)
        box = {}

        def run():
            try:
                box['value'] = fn(*args)
            except BaseException as e:          # noqa: B902 -- handed to the waiting thread as it is
                box['error'] = e
        t = threading.Thread(target=run, name='hp3d-' + what, daemon=True)
        t.start()
        t.join(timeout)
        if t.is_alive():
            self.comm_abandoned = True
            raise TimeoutError("rccl init timeout: %s did not return within %.0f s on rank %d of %d (HP3D_RCCL_TIMEOUT)" % (what, timeout, self.rank, self.world))
        if 'error' in box:
            raise box['error']
        return box.get('value')

    def comm_init(self):
        """hp3d_comm_init on every rank: rank 0 draws the 128-byte id, the rendezvous hands it round.  Bounded by HP3D_RCCL_TIMEOUT."""
        if not self.comm_ready:
            msg = None
            if self.rank == 0:
                try:
                    msg = self.engine.comm_unique_id()
                except Exception as e:          # no RCCL on this box: the OTHER ranks are already waiting for the id -- tell them
                    msg = 'hp3d_comm_unique_id failed on rank 0: %s: %s' % (type(e).__name__, e)
            msg = self.rdzv.broadcast(msg, 0)
            if isinstance(msg, str):
                raise RuntimeError(msg)
            self._with_deadline('hp3d_comm_init', self.engine.comm_init, self.rank, self.world, msg)
            self.comm_ready = True

    def sync_weights(self, weights=None, dtype=0, use_comm=None):
        """Rank 0 packs the weight dictionary; every other rank receives the packed device blob (and rank 0's nets mask
        and precision) by hp3d_bcast_weights -- no un-pickling or re-packing on the other ranks.  use_comm: None = only
        when world > 1; True forces the RCCL path at world size 1 as well (hardware smoke test of the exchange)."""
        if self.tcp_only and self.world > 1:
            # no communicator (use_tcp_only): rank 0's weight DICTIONARY travels over the rendezvous (tagged arrays, nothing un-pickled;
            # 140 MB) and every rank packs its own copy -- still one source of truth, just not the packed blob over xGMI
            weights = self.rdzv.broadcast(weights if self.rank == 0 else None, 0)
            self.engine.load_weight_dict(weights)
            self.engine.finalize_weights(dtype)
            return
        if self.rank == 0:
            self.engine.load_weight_dict(weights)
            self.engine.finalize_weights(dtype)
        if use_comm or (use_comm is None and self.world > 1):
            self.comm_init()
            self._with_deadline('hp3d_bcast_weights', self.engine.bcast_weights, 0)      # (the first collective: where a half-built communicator would hang)

    sync_weights_native = sync_weights        # round-1 name

    def gather_keypoints(self, coord_dev, n_local):
        """hp3d_allgather_dev of this rank's [n_local,21,3] device-resident keypoints -> float32 [world*n_local,21,3]
        on the host of every rank (equal shard sizes: the benchmark's weak-scaling layout)."""
        if self.world == 1 and not self.comm_ready:
            return self.engine.to_host(coord_dev, (n_local, 21, 3))
        if self.tcp_only:
            return np.concatenate(self.rdzv.allgather(self.engine.to_host(coord_dev, (n_local, 21, 3))), 0)
        return self.engine.allgather_dev(coord_dev, n_local * 63, self.world).reshape(self.world * n_local, 21, 3)

    def gather_ragged(self, local_kp, n_total):
        """Ragged shards (shard_range): pad to the largest shard, hp3d_allgather, cut the padding."""
        sizes = shard_sizes(n_total, self.world)
        mx = max(sizes)
        pad = np.zeros((mx,) + tuple(local_kp.shape[1:]), np.float32)
        pad[:local_kp.shape[0]] = local_kp
        if self.world == 1 and not self.comm_ready:
            return pad[:sizes[0]]
        if self.tcp_only:
            return np.concatenate([np.asarray(x)[:s] for x, s in zip(self.rdzv.allgather(pad), sizes)], 0)
        full = self.engine.allgather(pad, self.world).reshape((self.world, mx) + tuple(local_kp.shape[1:]))
        return np.concatenate([full[r, :s] for r, s in enumerate(sizes)], 0)

    def close(self):
        if self.comm_ready and not self.comm_abandoned:
            self.engine.comm_destroy()
        self.comm_ready = False
        self.rdzv.close()
