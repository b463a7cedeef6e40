"""Deterministic stand-ins for the policy/value network, identical bit for bit in three places:
  * here in NumPy  -- drives the REFERENCE player when generating golden vectors (StubPipe)
  * oracle/xq_mcts.c (xqo_stub_uniform / xqo_stub_hash) -- drives the C oracle
  * here in torch  -- replaces the ResNet on the GPU in the -m gpu parity tests
so that "identical visit counts with the net stubbed to constants" can be checked exactly.

hash stub:  h = mix64(salt + sum_{o: planes[o] != 0} mix64(o + 1))      (o over 14*90 or 28*90 elements)
            policy[a] = x^8 (three float32 squarings), x = ((mix64(h + (a+1)*GOLD) >> 40 & 0xFFFF) + 1) / 65536
            value     = ((mix64(h ^ C2) >> 40 & 0xFFFF) - 32768) / 32768
Also: the counter-based uniform stream (Philox4x32-10) shared by the oracle and the engine.
"""
import numpy as np

M64 = (1 << 64) - 1
GOLD = 0x9E3779B97F4A7C15
C2 = 0x5851F42D4C957F2D
N_LABELS = 2086


def mix64_int(z):
    z &= M64
    z ^= z >> 30
    z = (z * 0xbf58476d1ce4e5b9) & M64
    z ^= z >> 27
    z = (z * 0x94d049bb133111eb) & M64
    z ^= z >> 31
    return z


def _mix64_np(z):
    z = z.astype(np.uint64)
    with np.errstate(over="ignore"):
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xbf58476d1ce4e5b9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94d049bb133111eb)
        z ^= z >> np.uint64(31)
    return z


_C1 = _mix64_np(np.arange(1, 2521, dtype=np.uint64))      # 14 or 28 planes of 90 squares
_A = (np.arange(1, N_LABELS + 1, dtype=np.uint64) * np.uint64(GOLD))


def hash_stub_numpy(planes, salt=0):
    """planes [n,14,10,9] (any dtype, 0/1) -> (policy float32 [n,2086], value float32 [n])"""
    pl = np.asarray(planes).reshape(len(planes), -1) != 0
    with np.errstate(over="ignore"):
        h0 = (pl.astype(np.uint64) * _C1[None, :pl.shape[1]]).sum(axis=1, dtype=np.uint64) + np.uint64(salt)
        h = _mix64_np(h0)
        u = (_mix64_np(h[:, None] + _A[None, :]) >> np.uint64(40)) & np.uint64(0xFFFF)
    x = (u + np.uint64(1)).astype(np.float32) / np.float32(65536.0)
    x = x * x
    x = x * x
    x = x * x
    uv = (_mix64_np(h ^ np.uint64(C2)) >> np.uint64(40)) & np.uint64(0xFFFF)
    v = (uv.astype(np.float32) - np.float32(32768.0)) / np.float32(32768.0)
    return x.astype(np.float32), v.astype(np.float32)


def uniform_stub_numpy(planes, value=0.0):
    n = len(planes)
    return (np.full((n, N_LABELS), np.float32(1.0 / 2086.0), dtype=np.float32),
            np.full((n,), np.float32(value), dtype=np.float32))


# ---- torch (runs on the GPU; int64 arithmetic wraps like uint64) ---------------------------------
def _s64(x):
    x &= M64
    return x - (1 << 64) if x >= (1 << 63) else x


def _mix64_torch(z):
    import torch
    def lsr(t, k):
        return (t >> k) & ((1 << (64 - k)) - 1)
    z = z ^ lsr(z, 30)
    z = z * _s64(0xbf58476d1ce4e5b9)
    z = z ^ lsr(z, 27)
    z = z * _s64(0x94d049bb133111eb)
    z = z ^ lsr(z, 31)
    return z


_torch_consts = {}


def hash_stub_torch(planes, salt=0):
    """planes: tensor [n,14,10,9] on any device -> (policy float32 [n,2086], value float32 [n])"""
    import torch
    dev = planes.device
    key = str(dev)
    if key not in _torch_consts:
        c1 = torch.from_numpy(_C1.view(np.int64).copy()).to(dev)
        a = torch.from_numpy(_A.view(np.int64).copy()).to(dev)
        _torch_consts[key] = (c1, a)
    c1, a = _torch_consts[key]
    n = planes.shape[0]
    pl = (planes.reshape(n, -1) != 0).to(torch.int64)
    h0 = (pl * c1[None, :pl.shape[1]]).sum(dim=1) + _s64(salt)
    h = _mix64_torch(h0)
    u = (_mix64_torch(h[:, None] + a[None, :]) >> 40) & 0xFFFF
    x = (u + 1).to(torch.float32) / 65536.0
    x = x * x
    x = x * x
    x = x * x
    uv = (_mix64_torch(h ^ _s64(C2)) >> 40) & 0xFFFF
    v = (uv.to(torch.float32) - 32768.0) / 32768.0
    return x, v


# ---- pipe protocol of the reference (agent/api.py:37-74): send list[planes] -> recv list[(p, v)] ----
class StubPipe:
    def __init__(self, fn):
        self.fn = fn
        self.queue = []
        self.n_batches = 0
        self.n_positions = 0

    def send(self, data):
        p, v = self.fn(np.asarray(data, dtype=np.float32))
        self.n_batches += 1
        self.n_positions += len(data)
        self.queue.append([(p[i], float(v[i])) for i in range(len(data))])

    def poll(self, timeout=None):
        return bool(self.queue)

    def recv(self):
        return self.queue.pop(0)


# ---- Philox4x32-10 uniform stream: u(seed, game_id, stream, idx) ---------------------------------------
def philox_uniform(seed, game_id, stream, idx):
    c = [idx & 0xFFFFFFFF, (idx >> 32) & 0xFFFFFFFF, stream & 0xFFFFFFFF, game_id & 0xFFFFFFFF]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF,
             ((p0 >> 32) ^ c[3] ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return ((c[0] >> 5) * 67108864.0 + (c[1] >> 6)) / 9007199254740992.0


def numpy_choice(p, u):
    """np.random.choice(len(p), p=p) given its uniform draw u (legacy RandomState.choice algorithm:
    cdf = p.cumsum(); cdf /= cdf[-1]; idx = cdf.searchsorted(u, side='right'))."""
    cdf = np.asarray(p, dtype=np.float64).cumsum()
    cdf /= cdf[-1]
    return int(cdf.searchsorted(u, side='right'))
