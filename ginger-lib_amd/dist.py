"""Multi-GPU MSM: one process per GPU, pairs sharded by contiguous index range, one exchange.

The reference has no distributed component.  sum_i s_i P_i is a sum in a
commutative group, so rank r computes a complete single-GPU MSM over its shard and the partial
sums are combined.  EC addition is not an RCCL reduction operator (a limb-wise ncclSum of
projective coordinates is meaningless), hence the "all-reduce of partial sums" is realised as ONE
all-gather of the raw projective limbs (288 / 576 / 864 bytes per rank for G1 / MNT4-G2 / MNT6-G2)
followed by a world_size-1 addition fold in rank order on every rank (gh_proj_add, host side).
The affine image of the result is independent of the shard count.
"""
import numpy as np


def shard_bounds(n, rank, world):
    """Contiguous [lo, hi) of rank `rank` when n items are split over `world` ranks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_fold(partial_xyz, proj_add, dist=None, device=None):
    """partial_xyz: 1-D uint64 numpy array (this rank's projective partial sum).
    Returns the fold over all ranks (same on every rank).  `proj_add(acc, p) -> acc'`."""
    import torch
    import torch.distributed as td
    dist = dist or td
    world = dist.get_world_size()
    if world == 1:
        return np.array(partial_xyz, dtype=np.uint64)
    t = torch.from_numpy(np.ascontiguousarray(partial_xyz, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    acc = parts[0].cpu().numpy().view(np.uint64).copy()
    for r in range(1, world):
        acc = proj_add(acc, parts[r].cpu().numpy().view(np.uint64))
    return acc


class CDist:
    """The exchange behind the C ABI (include/ginger_hip_dist.h): gh_dist_init_rccl / gh_dist_init_custom +
    gh_partials_allgather_fold.  transport = "rccl": RCCL over xGMI -- the 128-byte unique id travels from rank 0
    through `store` (a torch.distributed store, e.g. the TCPStore torchrun's MASTER_ADDR / MASTER_PORT give);
    transport = "callback": any all-gather as a Python callable (gloo in the CPU tests)."""

    @staticmethod
    def probe_rccl(gl):
        """True if librccl can be loaded and bound on THIS rank (no collective: safe to call before the ranks agree)"""
        return gl.load_library().gh_dist_probe_rccl() == 0

    def __init__(self, gl, rank, world, transport="rccl", store=None, allgather=None):
        import ctypes
        self.gl, self.rank, self.world = gl, rank, world
        lib = gl.load_library()
        if transport == "rccl":
            uid = (ctypes.c_char * 128)()
            if rank == 0:
                gl._check(lib.gh_dist_unique_id(uid))
                store.set("gh_dist_uid", bytes(uid.raw))
            else:
                uid.raw = store.get("gh_dist_uid")
            gl._check(lib.gh_dist_init_rccl(uid, rank, world))
        else:
            def _cb(_ctx, send, recv, nbytes):
                try:
                    data = ctypes.string_at(send, nbytes)
                    out = allgather(data)
                    assert len(out) == nbytes * world
                    ctypes.memmove(recv, out, len(out))
                    return 0
                except Exception:       # noqa: a Python error must not unwind through the C frame
                    return 1
            self._cb = gl.ALLGATHER_FN(_cb)       # keep the thunk alive
            gl._check(lib.gh_dist_init_custom(self._cb, None, rank, world))
        r, w = ctypes.c_int(), ctypes.c_int()
        gl._check(lib.gh_dist_info(ctypes.byref(r), ctypes.byref(w)))
        self.world_seen = w.value              # ranks the transport itself reports (ncclCommCount)
        self.last_exchange_us = 0.0
        rr, rv = ctypes.c_int(), ctypes.c_int()
        path = ctypes.create_string_buffer(512)
        gl._check(lib.gh_dist_transport(ctypes.byref(rr), ctypes.byref(rv), path, 512))
        self.rccl_ranks = rr.value             # 0 on the callback transport: nothing went over RCCL
        self.rccl_version = rv.value
        self.rccl_path = path.value.decode()

    def allgather_fold(self, curve, partial_xyz):
        import ctypes
        gl = self.gl
        p = np.ascontiguousarray(partial_xyz, dtype=np.uint64)
        out = np.zeros_like(p)
        us = ctypes.c_double()
        gl._check(gl.load_library().gh_partials_allgather_fold(gl.CURVES[curve], gl._ptr(p), gl._ptr(out), ctypes.byref(us)))
        self.last_exchange_us = us.value
        return out

    def allgather_fold_batch(self, curve, partials):
        """the results of a pipelined batch in ONE exchange: list of partial sums in, list of global sums out"""
        import ctypes
        gl = self.gl
        if not partials:
            return []
        p = np.ascontiguousarray(np.stack([np.asarray(x, dtype=np.uint64) for x in partials]))
        out = np.zeros_like(p)
        us = ctypes.c_double()
        gl._check(gl.load_library().gh_partials_allgather_fold_batch(gl.CURVES[curve], gl._ptr(p), len(partials), gl._ptr(out), ctypes.byref(us)))
        self.last_exchange_us = us.value
        return [out[i].copy() for i in range(len(partials))]

    def shutdown(self):
        self.gl.load_library().gh_dist_shutdown()


def gloo_allgather_bytes(dist):
    """all-gather of a byte string over a torch.distributed (gloo) group, for CDist(transport="callback")"""
    import torch

    def fn(data):
        t = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t)
        return b"".join(bytes(p.numpy().tobytes()) for p in parts)
    return fn


class ShardedMSM:
    """Holds this rank's shard of the bases resident on its GPU; multi_scalar_mul() takes this
    rank's shard of the scalars and returns the global sum on every rank.

    local_msm / proj_add are injectable so that the exchange logic can be exercised on CPU ranks
    (gloo) with a checker standing in for the device (tests/test_dist_cpu.py)."""

    def __init__(self, curve, local_msm, proj_add, device=None):
        self.curve, self.local_msm, self.proj_add, self.device = curve, local_msm, proj_add, device

    def multi_scalar_mul(self, scalars_shard):
        partial = self.local_msm(scalars_shard)
        return all_gather_fold(partial, self.proj_add, device=self.device)
