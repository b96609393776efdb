"""sim.py -- executes an isa.Prog on the CPU, one 64-lane wave at a time (numpy over the lanes).

Purpose: the generated kernels are checked against Python integers / the group law BEFORE they are assembled for the
GPU (tests/test_asmgen.py), including every memory access: a load or store outside a registered buffer raises instead
of faulting a card.  Only the instruction subset the generators emit is implemented; an unknown mnemonic raises.
Semantics follow the CDNA3/4 ISA manual (operand order of the *rev forms, v_cndmask, v_alignbit, carry conventions).
"""
import numpy as np

from .isa import Reg

U32 = np.uint32
U64 = np.uint64
M32 = 0xFFFFFFFF
LANES = 64
_AR = np.arange(LANES, dtype=U64)


def mask_arr(m):
    return ((U64(m) >> _AR) & U64(1)).astype(bool)


def arr_mask(a):
    return int(np.packbits(a.astype(np.uint8), bitorder="little").view(U64)[0])


class MemFault(Exception):
    pass


class Memory:
    """Global memory as named regions at synthetic 64-bit addresses (dword granularity)."""

    def __init__(self):
        self.regions = []       # (base, nbytes, array(uint32), name, writable)
        self.next = 0x7F0000000000
        self.loads = 0
        self.stores = 0

    def add(self, name, arr_u32, writable=False):
        arr = np.ascontiguousarray(arr_u32, dtype=U32).reshape(-1)
        base = self.next
        self.next += (arr.nbytes + 0xFFFF) // 0x10000 * 0x10000 + 0x10000
        self.regions.append((base, arr.nbytes, arr, name, writable))
        return base

    def get(self, name):
        for r in self.regions:
            if r[3] == name:
                return r[2]
        raise KeyError(name)

    def _find(self, addr, nbytes, write):
        for base, size, arr, name, wr in self.regions:
            if base <= addr and addr + nbytes <= base + size:
                if write and not wr:
                    raise MemFault("store to read-only region %s at +%d" % (name, addr - base))
                return arr, (addr - base) >> 2
        raise MemFault("%s of %d bytes at 0x%x is outside every buffer" % ("store" if write else "load", nbytes, addr))

    def load(self, addr, ndw):
        if addr & 3:
            raise MemFault("unaligned load 0x%x" % addr)
        arr, i = self._find(addr, 4 * ndw, False)
        self.loads += 1
        return arr[i:i + ndw]

    def store(self, addr, vals):
        if addr & 3:
            raise MemFault("unaligned store 0x%x" % addr)
        arr, i = self._find(addr, 4 * len(vals), True)
        self.stores += 1
        arr[i:i + len(vals)] = vals


class Wave:
    def __init__(self, prog, mem, lds_words=0):
        self.prog = prog
        self.mem = mem
        self.V = np.zeros((512, LANES), dtype=U32)      # VGPRs 0..255, AGPRs at 256 + i
        self.S = [0] * 128
        self.scc = 0
        self.lds = np.zeros(max(lds_words, 1), dtype=U32)
        self.labels = {}
        for n, i in enumerate(prog.ins):
            if i.op == "label":
                self.labels[i.args[0]] = n
        self.executed = 0
        self.hist = {}
        self.set_exec((1 << 64) - 1)

    # ---- special registers
    def set_exec(self, m):
        self.S[126] = m & M32
        self.S[127] = (m >> 32) & M32

    def exec_mask(self):
        return self.S[126] | (self.S[127] << 32)

    def em(self):
        return mask_arr(self.exec_mask())

    # ---- operand access
    def ridx(self, r):
        k = r.kind
        if k == "v":
            return r.idx
        if k == "a":
            return 256 + r.idx
        raise TypeError(r)

    def rd(self, o):
        """32-bit source as uint32 array (or numpy scalar broadcast)."""
        if isinstance(o, Reg):
            if o.kind in ("v", "a"):
                assert o.n == 1, o
                return self.V[self.ridx(o)]
            assert o.n == 1 or o.kind in ("vcc", "exec"), o
            return U32(self.S[o.idx])
        return U32(o & M32)

    def rd64(self, o):
        if isinstance(o, Reg):
            if o.kind in ("v", "a"):
                assert o.n == 2, o
                i = self.ridx(o)
                return self.V[i].astype(U64) | (self.V[i + 1].astype(U64) << U64(32))
            assert o.n == 2, o
            return U64(self.S[o.idx] | (self.S[o.idx + 1] << 32))
        return U64(o & 0xFFFFFFFFFFFFFFFF)

    def rd_smask(self, o):
        assert isinstance(o, Reg) and o.n == 2 and o.kind in ("s", "vcc", "exec"), o
        return self.S[o.idx] | (self.S[o.idx + 1] << 32)

    def wr_smask(self, o, m):
        assert isinstance(o, Reg) and o.n == 2 and o.kind in ("s", "vcc", "exec"), o
        self.S[o.idx] = m & M32
        self.S[o.idx + 1] = (m >> 32) & M32

    def wr(self, o, val):
        assert isinstance(o, Reg) and o.kind in ("v", "a") and o.n == 1, o
        em = self.em()
        val = np.broadcast_to(np.asarray(val, dtype=U32), (LANES,))
        self.V[self.ridx(o)][em] = val[em]

    def wr64(self, o, val):
        assert isinstance(o, Reg) and o.kind in ("v", "a") and o.n == 2, o
        em = self.em()
        val = np.broadcast_to(np.asarray(val, dtype=U64), (LANES,))
        i = self.ridx(o)
        self.V[i][em] = (val & U64(M32)).astype(U32)[em]
        self.V[i + 1][em] = (val >> U64(32)).astype(U32)[em]

    def wr_lanemask(self, o, arr):
        """VALU compare / carry-out result: bits of inactive lanes are written as 0."""
        self.wr_smask(o, arr_mask(np.asarray(arr, dtype=bool) & self.em()))

    def ws(self, o, val):
        assert isinstance(o, Reg) and o.kind == "s" and o.n == 1, o
        self.S[o.idx] = int(val) & M32

    def rs(self, o):
        if isinstance(o, Reg):
            assert o.n == 1 and o.kind in ("s",), o
            return self.S[o.idx]
        return o & M32

    # ---- asynchronous results: a register that a memory / LDS instruction will write is "in flight" until the s_waitcnt that
    #      covers it (vmcnt(0) / lgkmcnt(0); partial counts are not modelled: they clear nothing).  Any other instruction that reads or
    #      writes such a register in between is a missing wait -- on the card it would see (or be overwritten by) stale data.
    def _check_inflight(self, i):
        op = i.op
        if op == "s_waitcnt":
            if "vmcnt" in i.mods and getattr(self, "_inflight_vm", None) is not None:
                # vector memory results return in order: vmcnt(N) leaves the N youngest loads (and stores) outstanding
                keep = int(i.mods["vmcnt"])
                self._vm_fifo = self._vm_fifo[len(self._vm_fifo) - keep:] if keep else []
                self._inflight_vm = set().union(*self._vm_fifo) if self._vm_fifo else set()
            if i.mods.get("lgkmcnt") == 0:
                self._inflight_lgkm = set()
            return
        vm = getattr(self, "_inflight_vm", None)
        if vm is None:
            vm = self._inflight_vm = set()
            self._inflight_lgkm = set()
            self._vm_fifo = []
        lg = self._inflight_lgkm
        regs = set()
        for a in i.args:
            if isinstance(a, Reg) and a.kind in ("v", "a"):
                base = a.idx + (256 if a.kind == "a" else 0)
                regs.update(range(base, base + a.n))
            elif isinstance(a, Reg) and a.kind == "s":
                regs.update(-1 - r for r in range(a.idx, a.idx + a.n))       # SGPRs as negative keys (s_load results)
        own = set()
        if op.startswith("global_load") and isinstance(i.args[0], Reg):       # the same destination again under another EXEC mask (a
            d0 = i.args[0]                                                    # gather a quarter of the wave at a time): loads return in order
            own = set(range(d0.idx, d0.idx + d0.n)) & vm
        hit = (regs - own) & (vm | lg)
        if hit:
            raise RuntimeError("%s touches a register with a result in flight (no s_waitcnt): %s" % (
                i.text().strip(), sorted(("v%d" % r) if r >= 0 else ("s%d" % (-1 - r)) for r in hit)[:6]))
        dst = None
        if op.startswith("global_load") or (op.startswith("global_atomic") and i.mods.get("sc0")):
            dst, pend = i.args[0], vm
        elif op.startswith("ds_read") or op in ("ds_bpermute_b32", "ds_permute_b32", "ds_swizzle_b32"):
            dst, pend = i.args[0], lg
        elif op.startswith("s_load"):
            dst, pend = i.args[0], lg
        if op.startswith("global_store") or (op.startswith("global_atomic") and not i.mods.get("sc0")):
            self._vm_fifo.append(set())                                       # counted by vmcnt, nothing to protect
        if dst is not None:
            if dst.kind in ("v", "a"):
                base = dst.idx + (256 if dst.kind == "a" else 0)
                pend.update(range(base, base + dst.n))
                if pend is vm:
                    self._vm_fifo.append(set(range(base, base + dst.n)))
            else:
                pend.update(-1 - r for r in range(dst.idx, dst.idx + dst.n))

    # ---- run
    def run(self, max_steps=50_000_000):
        pc = 0
        ins = self.prog.ins
        n = len(ins)
        old = np.seterr(over="ignore")
        try:
            while pc < n:
                i = ins[pc]
                pc += 1
                op = i.op
                if op in ("label", "comment"):
                    continue
                self.executed += 1
                self.hist[op] = self.hist.get(op, 0) + 1
                self._check_inflight(i)
                if self.executed > max_steps:
                    raise RuntimeError("step limit")
                if op == "s_endpgm":
                    return
                h = HANDLERS.get(op)
                if h is None:
                    raise NotImplementedError(op)
                t = h(self, i)
                if t is not None:
                    pc = self.labels[t]
        finally:
            np.seterr(**old)
        raise RuntimeError("fell off the end of the program")


HANDLERS = {}


def op(*names):
    def deco(f):
        for n in names:
            HANDLERS[n] = f
        return f
    return deco


# ------------------------------------------------------------------ VALU
def _bin(fn):
    def h(w, i):
        d, a, b = i.args
        w.wr(d, fn(w.rd(a), w.rd(b)))
    return h


HANDLERS["v_add_u32"] = _bin(lambda a, b: a + b)
HANDLERS["v_sub_u32"] = _bin(lambda a, b: a - b)
HANDLERS["v_subrev_u32"] = _bin(lambda a, b: b - a)
HANDLERS["v_and_b32"] = _bin(lambda a, b: a & b)
HANDLERS["v_or_b32"] = _bin(lambda a, b: a | b)
HANDLERS["v_xor_b32"] = _bin(lambda a, b: a ^ b)
HANDLERS["v_lshlrev_b32"] = _bin(lambda a, b: b << (a & U32(31)))
HANDLERS["v_lshrrev_b32"] = _bin(lambda a, b: b >> (a & U32(31)))
HANDLERS["v_ashrrev_i32"] = _bin(lambda a, b: (np.asarray(b).astype(np.int32) >> (a & U32(31)).astype(np.int32)).astype(U32))
HANDLERS["v_mul_lo_u32"] = _bin(lambda a, b: (a.astype(U64) * np.asarray(b).astype(U64) & U64(M32)).astype(U32))
HANDLERS["v_mul_hi_u32"] = _bin(lambda a, b: ((np.asarray(a).astype(U64) * np.asarray(b).astype(U64)) >> U64(32)).astype(U32))
HANDLERS["v_min_u32"] = _bin(lambda a, b: np.minimum(a, b))
HANDLERS["v_max_u32"] = _bin(lambda a, b: np.maximum(a, b))


@op("v_mov_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_accvgpr_mov_b32")
def _mov(w, i):
    d, a = i.args
    w.wr(d, w.rd(a))


@op("v_bfrev_b32")
def _bfrev(w, i):
    d, a = i.args
    x = np.asarray(np.broadcast_to(w.rd(a), (LANES,)), dtype=U32)
    r = np.zeros(LANES, dtype=U32)
    for b in range(32):
        r |= ((x >> U32(b)) & U32(1)) << U32(31 - b)
    w.wr(d, r)


@op("v_not_b32")
def _not(w, i):
    d, a = i.args
    w.wr(d, ~np.asarray(w.rd(a), dtype=U32))


@op("v_add3_u32")
def _add3(w, i):
    d, a, b, c = i.args
    w.wr(d, w.rd(a) + w.rd(b) + w.rd(c))


@op("v_or3_b32")
def _or3(w, i):
    d, a, b, c = i.args
    w.wr(d, w.rd(a) | w.rd(b) | w.rd(c))


@op("v_bfi_b32")
def _bfi(w, i):
    d, a, b, c = i.args
    m = np.asarray(w.rd(a), dtype=U32)
    w.wr(d, (m & w.rd(b)) | (~m & w.rd(c)))


@op("v_and_or_b32")
def _and_or(w, i):
    d, a, b, c = i.args
    w.wr(d, (w.rd(a) & w.rd(b)) | w.rd(c))


@op("v_lshl_or_b32")
def _lshl_or(w, i):
    d, a, b, c = i.args
    w.wr(d, (np.asarray(w.rd(a), dtype=U32) << (w.rd(b) & U32(31))) | w.rd(c))


@op("v_lshl_add_u32")
def _lshl_add(w, i):
    d, a, b, c = i.args
    w.wr(d, (np.asarray(w.rd(a), dtype=U32) << (w.rd(b) & U32(31))) + w.rd(c))


@op("v_add_lshl_u32")
def _add_lshl(w, i):
    d, a, b, c = i.args
    w.wr(d, np.asarray(w.rd(a) + w.rd(b), dtype=U32) << (w.rd(c) & U32(31)))


@op("v_alignbit_b32")
def _alignbit(w, i):
    d, hi, lo, sh = i.args
    v = (np.asarray(w.rd(hi)).astype(U64) << U64(32)) | np.asarray(w.rd(lo)).astype(U64)
    s = np.asarray(w.rd(sh)).astype(U64) & U64(31)
    w.wr(d, ((v >> s) & U64(M32)).astype(U32))


@op("v_bfe_u32")
def _bfe(w, i):
    d, a, off, width = i.args
    o = np.asarray(w.rd(off)) & U32(31)
    wd = int(np.asarray(w.rd(width)).reshape(-1)[0]) & 31
    w.wr(d, (np.asarray(w.rd(a), dtype=U32) >> o) & U32((1 << wd) - 1))


@op("v_mad_u64_u32")
def _mad64(w, i):
    d, sd, a, b, c = i.args
    prod = np.asarray(w.rd(a)).astype(U64) * np.asarray(w.rd(b)).astype(U64)
    c64 = w.rd64(c)
    res = prod + c64
    w.wr64(d, res)
    w.wr_lanemask(sd, np.broadcast_to(res < prod, (LANES,)))


@op("v_mad_u32_u24")
def _mad24(w, i):
    d, a, b, c = i.args
    w.wr(d, ((np.asarray(w.rd(a)) & U32(0xFFFFFF)).astype(U64) * (np.asarray(w.rd(b)) & U32(0xFFFFFF)).astype(U64)
             + np.asarray(w.rd(c)).astype(U64) & U64(M32)).astype(U32))


@op("v_lshl_add_u64")
def _lshl_add_u64(w, i):
    d, a, sh, c = i.args
    w.wr64(d, (np.asarray(w.rd64(a), dtype=U64) << U64(w.rs(sh) & 7)) + w.rd64(c))


@op("v_lshrrev_b64")
def _lshr64(w, i):
    d, sh, a = i.args
    w.wr64(d, np.asarray(w.rd64(a), dtype=U64) >> (np.asarray(w.rd(sh)).astype(U64) & U64(63)))


@op("v_ashrrev_i64")
def _ashr64(w, i):
    d, sh, a = i.args
    v = np.asarray(w.rd64(a), dtype=U64).astype(np.int64) >> (np.asarray(w.rd(sh)).astype(np.int64) & np.int64(63))
    w.wr64(d, v.astype(U64))


@op("v_lshlrev_b64")
def _lshl64(w, i):
    d, sh, a = i.args
    w.wr64(d, np.asarray(w.rd64(a), dtype=U64) << (np.asarray(w.rd(sh)).astype(U64) & U64(63)))


@op("v_cndmask_b32")
def _cnd(w, i):
    d, a, b, m = i.args
    sel = mask_arr(w.rd_smask(m))
    w.wr(d, np.where(sel, w.rd(b), w.rd(a)))


def _cmp(fn):
    def h(w, i):
        d, a, b = i.args
        w.wr_lanemask(d, np.broadcast_to(fn(np.asarray(w.rd(a), dtype=U32), np.asarray(w.rd(b), dtype=U32)), (LANES,)))
    return h


for _n, _f in (("eq", lambda a, b: a == b), ("ne", lambda a, b: a != b), ("lt", lambda a, b: a < b),
               ("le", lambda a, b: a <= b), ("gt", lambda a, b: a > b), ("ge", lambda a, b: a >= b)):
    HANDLERS["v_cmp_%s_u32" % _n] = _cmp(_f)
for _n, _f in (("lt", lambda a, b: a.astype(np.int32) < b.astype(np.int32)), ("gt", lambda a, b: a.astype(np.int32) > b.astype(np.int32)),
               ("ge", lambda a, b: a.astype(np.int32) >= b.astype(np.int32)), ("le", lambda a, b: a.astype(np.int32) <= b.astype(np.int32))):
    HANDLERS["v_cmp_%s_i32" % _n] = _cmp(_f)


@op("v_add_co_u32")
def _add_co(w, i):
    d, co, a, b = i.args
    r = np.asarray(w.rd(a)).astype(U64) + np.asarray(w.rd(b)).astype(U64)
    w.wr(d, (r & U64(M32)).astype(U32))
    w.wr_lanemask(co, np.broadcast_to(r >> U64(32) != 0, (LANES,)))


@op("v_addc_co_u32")
def _addc_co(w, i):
    d, co, a, b, ci = i.args
    cin = mask_arr(w.rd_smask(ci)).astype(U64)
    r = np.asarray(w.rd(a)).astype(U64) + np.asarray(w.rd(b)).astype(U64) + cin
    w.wr(d, (r & U64(M32)).astype(U32))
    w.wr_lanemask(co, np.broadcast_to(r >> U64(32) != 0, (LANES,)))


@op("v_sub_co_u32")
def _sub_co(w, i):
    d, co, a, b = i.args
    x, y = np.asarray(w.rd(a)).astype(np.int64), np.asarray(w.rd(b)).astype(np.int64)
    r = x - y
    w.wr(d, (r & 0xFFFFFFFF).astype(U32))
    w.wr_lanemask(co, np.broadcast_to(r < 0, (LANES,)))


@op("v_subrev_co_u32")
def _subrev_co(w, i):
    d, co, a, b = i.args
    x, y = np.asarray(w.rd(b)).astype(np.int64), np.asarray(w.rd(a)).astype(np.int64)
    r = x - y
    w.wr(d, (r & 0xFFFFFFFF).astype(U32))
    w.wr_lanemask(co, np.broadcast_to(r < 0, (LANES,)))


@op("v_subb_co_u32")
def _subb_co(w, i):
    d, co, a, b, ci = i.args
    cin = mask_arr(w.rd_smask(ci)).astype(np.int64)
    r = np.asarray(w.rd(a)).astype(np.int64) - np.asarray(w.rd(b)).astype(np.int64) - cin
    w.wr(d, (r & 0xFFFFFFFF).astype(U32))
    w.wr_lanemask(co, np.broadcast_to(r < 0, (LANES,)))


@op("v_subbrev_co_u32")
def _subbrev_co(w, i):
    d, co, a, b, ci = i.args
    cin = mask_arr(w.rd_smask(ci)).astype(np.int64)
    r = np.asarray(w.rd(b)).astype(np.int64) - np.asarray(w.rd(a)).astype(np.int64) - cin
    w.wr(d, (r & 0xFFFFFFFF).astype(U32))
    w.wr_lanemask(co, np.broadcast_to(r < 0, (LANES,)))


@op("v_readfirstlane_b32")
def _rfl(w, i):
    d, a = i.args
    em = w.exec_mask()
    lane = (em & -em).bit_length() - 1 if em else 0
    w.ws(d, int(np.asarray(np.broadcast_to(w.rd(a), (LANES,)))[lane]))


@op("v_mov_b32_dpp")
def _mov_dpp(w, i):
    # quad_perm only, bound_ctrl:0 row_mask/bank_mask 0xf: every lane reads lane (quad base + perm[lane & 3])
    d, a = i.args
    perm = i.mods["quad_perm"]
    src = np.asarray(w.rd(a))
    idx = np.array([(l & ~3) + perm[l & 3] for l in range(LANES)])
    w.wr(d, src[idx])


# ------------------------------------------------------------------ SALU
def _sbin(fn, scc=lambda r: r != 0):
    def h(w, i):
        d, a, b = i.args
        r = fn(w.rs(a), w.rs(b)) & M32
        w.ws(d, r)
        if scc is not None:
            w.scc = 1 if scc(r) else 0
    return h


HANDLERS["s_and_b32"] = _sbin(lambda a, b: a & b)
HANDLERS["s_or_b32"] = _sbin(lambda a, b: a | b)
HANDLERS["s_lshl_b32"] = _sbin(lambda a, b: a << (b & 31))
HANDLERS["s_lshr_b32"] = _sbin(lambda a, b: a >> (b & 31))
HANDLERS["s_mul_i32"] = _sbin(lambda a, b: a * b, scc=None)


@op("s_add_u32")
def _s_add(w, i):
    d, a, b = i.args
    r = w.rs(a) + w.rs(b)
    w.ws(d, r)
    w.scc = 1 if r >> 32 else 0


@op("s_addc_u32")
def _s_addc(w, i):
    d, a, b = i.args
    r = w.rs(a) + w.rs(b) + w.scc
    w.ws(d, r)
    w.scc = 1 if r >> 32 else 0


@op("s_sub_u32")
def _s_sub(w, i):
    d, a, b = i.args
    r = w.rs(a) - w.rs(b)
    w.ws(d, r)
    w.scc = 1 if r < 0 else 0


@op("s_mov_b32")
def _s_mov(w, i):
    d, a = i.args
    w.ws(d, w.rs(a))


def _s64(o, w):
    if isinstance(o, Reg):
        return w.rd_smask(o)
    return o & 0xFFFFFFFFFFFFFFFF if o >= 0 else o & 0xFFFFFFFFFFFFFFFF


@op("s_mov_b64")
def _s_mov64(w, i):
    d, a = i.args
    w.wr_smask(d, _s64(a, w))


@op("s_cselect_b64")
def _s_csel64(w, i):
    d, a, b = i.args
    w.wr_smask(d, _s64(a, w) if w.scc else _s64(b, w))


def _s64bin(fn):
    def h(w, i):
        d, a, b = i.args
        r = fn(_s64(a, w), _s64(b, w)) & 0xFFFFFFFFFFFFFFFF
        w.wr_smask(d, r)
        w.scc = 1 if r else 0
    return h


HANDLERS["s_and_b64"] = _s64bin(lambda a, b: a & b)
HANDLERS["s_or_b64"] = _s64bin(lambda a, b: a | b)
HANDLERS["s_xor_b64"] = _s64bin(lambda a, b: a ^ b)
HANDLERS["s_andn2_b64"] = _s64bin(lambda a, b: a & ~b)
HANDLERS["s_orn2_b64"] = _s64bin(lambda a, b: a | ~b)


@op("s_bfm_b64")
def _s_bfm64(w, i):
    d, a, b = i.args
    w.wr_smask(d, (((1 << (w.rs(a) & 63)) - 1) << (w.rs(b) & 63)) & 0xFFFFFFFFFFFFFFFF)


@op("s_not_b64")
def _s_not64(w, i):
    d, a = i.args
    r = ~_s64(a, w) & 0xFFFFFFFFFFFFFFFF
    w.wr_smask(d, r)
    w.scc = 1 if r else 0


@op("s_and_saveexec_b64")
def _s_and_saveexec(w, i):
    d, a = i.args
    old = w.exec_mask()
    src = _s64(a, w)
    w.wr_smask(d, old)
    w.set_exec(src & old)
    w.scc = 1 if (src & old) else 0


@op("s_or_saveexec_b64")
def _s_or_saveexec(w, i):
    d, a = i.args
    old = w.exec_mask()
    src = _s64(a, w)
    w.wr_smask(d, old)
    w.set_exec(src | old)
    w.scc = 1 if (src | old) else 0


def _scmp(fn):
    def h(w, i):
        a, b = i.args
        w.scc = 1 if fn(w.rs(a), w.rs(b)) else 0
    return h


HANDLERS["s_cmp_eq_u32"] = _scmp(lambda a, b: a == b)
HANDLERS["s_cmp_lg_u32"] = _scmp(lambda a, b: a != b)
HANDLERS["s_cmp_lt_u32"] = _scmp(lambda a, b: a < b)
HANDLERS["s_cmp_ge_u32"] = _scmp(lambda a, b: a >= b)
HANDLERS["s_cmp_gt_u32"] = _scmp(lambda a, b: a > b)


@op("s_cmp_eq_u64")
def _s_cmp_eq64(w, i):
    a, b = i.args
    w.scc = 1 if _s64(a, w) == _s64(b, w) else 0


@op("s_cmp_lg_u64")
def _s_cmp_lg64(w, i):
    a, b = i.args
    w.scc = 1 if _s64(a, w) != _s64(b, w) else 0


@op("s_branch")
def _s_branch(w, i):
    return i.args[0]


@op("long_branch")
def _long_branch(w, i):
    w.scc = 0
    return i.args[0]


@op("s_cbranch_scc0")
def _b_scc0(w, i):
    return i.args[0] if not w.scc else None


@op("s_cbranch_scc1")
def _b_scc1(w, i):
    return i.args[0] if w.scc else None


@op("s_cbranch_vccz")
def _b_vccz(w, i):
    return i.args[0] if (w.S[106] | w.S[107]) == 0 else None


@op("s_cbranch_vccnz")
def _b_vccnz(w, i):
    return i.args[0] if (w.S[106] | w.S[107]) != 0 else None


@op("s_cbranch_execz")
def _b_execz(w, i):
    return i.args[0] if w.exec_mask() == 0 else None


@op("s_cbranch_execnz")
def _b_execnz(w, i):
    return i.args[0] if w.exec_mask() != 0 else None


@op("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio", "s_clause")
def _nopish(w, i):
    return None


def _s_load(ndw):
    def h(w, i):
        d, base, off = i.args
        addr = w.rd_smask(base) + (off if isinstance(off, int) else w.rs(off))
        vals = w.mem.load(addr, ndw)
        for k in range(ndw):
            w.S[d.idx + k] = int(vals[k])
    return h


HANDLERS["s_load_dword"] = _s_load(1)
HANDLERS["s_load_dwordx2"] = _s_load(2)
HANDLERS["s_load_dwordx4"] = _s_load(4)
HANDLERS["s_load_dwordx8"] = _s_load(8)
HANDLERS["s_load_dwordx16"] = _s_load(16)


# ------------------------------------------------------------------ vector memory
def _vaddr(w, i, addr_op, saddr):
    off = int(i.mods.get("offset", 0))
    if saddr == "off" or saddr is None:
        a = w.rd64(addr_op).astype(np.int64)
        return a + off
    base = w.rd_smask(saddr)
    return np.asarray(np.broadcast_to(w.rd(addr_op), (LANES,))).astype(np.int64) + base + off


def _gload(ndw):
    def h(w, i):
        d, addr_op, saddr = i.args
        addrs = np.broadcast_to(_vaddr(w, i, addr_op, saddr), (LANES,))
        em = w.em()
        base = w.ridx(d)
        for l in np.nonzero(em)[0]:
            vals = w.mem.load(int(addrs[l]), ndw)
            for k in range(ndw):
                w.V[base + k][l] = vals[k]
    return h


def _gstore(ndw):
    def h(w, i):
        addr_op, d, saddr = i.args
        addrs = np.broadcast_to(_vaddr(w, i, addr_op, saddr), (LANES,))
        em = w.em()
        base = w.ridx(d)
        for l in np.nonzero(em)[0]:
            w.mem.store(int(addrs[l]), [w.V[base + k][l] for k in range(ndw)])
    return h


for _k, _n in (("dword", 1), ("dwordx2", 2), ("dwordx3", 3), ("dwordx4", 4)):
    HANDLERS["global_load_" + _k] = _gload(_n)
    HANDLERS["global_store_" + _k] = _gstore(_n)


# ------------------------------------------------------------------ LDS
def _lds_idx(w, addr_op, byte_off, nbytes):
    a = np.asarray(np.broadcast_to(w.rd(addr_op), (LANES,))).astype(np.int64) + byte_off
    em = w.em()
    act = a[em]
    if act.size and (act.min() < 0 or act.max() + nbytes > w.lds.size * 4 or (act & 3).any()):
        raise MemFault("LDS access out of range: [%d, %d) of %d bytes" % (act.min(), act.max() + nbytes, w.lds.size * 4))
    return a >> 2, em


@op("ds_read_b32")
def _ds_read_b32(w, i):
    d, addr = i.args
    idx, em = _lds_idx(w, addr, int(i.mods.get("offset", 0)), 4)
    w.V[w.ridx(d)][em] = w.lds[idx[em]]


@op("ds_write_b32")
def _ds_write_b32(w, i):
    addr, d = i.args
    idx, em = _lds_idx(w, addr, int(i.mods.get("offset", 0)), 4)
    w.lds[idx[em]] = w.V[w.ridx(d)][em]


def _ds_read_n(n):
    def h(w, i):
        d, addr = i.args
        idx, em = _lds_idx(w, addr, int(i.mods.get("offset", 0)), 4 * n)
        for k in range(n):
            w.V[w.ridx(d) + k][em] = w.lds[idx[em] + k]
    return h


def _ds_write_n(n):
    def h(w, i):
        addr, d = i.args
        idx, em = _lds_idx(w, addr, int(i.mods.get("offset", 0)), 4 * n)
        for k in range(n):
            w.lds[idx[em] + k] = w.V[w.ridx(d) + k][em]
    return h


HANDLERS["ds_read_b64"] = _ds_read_n(2)
HANDLERS["ds_read_b128"] = _ds_read_n(4)
HANDLERS["ds_write_b64"] = _ds_write_n(2)
HANDLERS["ds_write_b128"] = _ds_write_n(4)


@op("ds_read2st64_b32")
def _ds_read2st64(w, i):
    d, addr = i.args
    for k, key in enumerate(("offset0", "offset1")):
        idx, em = _lds_idx(w, addr, int(i.mods.get(key, 0)) * 256, 4)
        w.V[w.ridx(d) + k][em] = w.lds[idx[em]]


@op("ds_write2st64_b32")
def _ds_write2st64(w, i):
    addr, d0, d1 = i.args
    for d, key in ((d0, "offset0"), (d1, "offset1")):
        idx, em = _lds_idx(w, addr, int(i.mods.get(key, 0)) * 256, 4)
        w.lds[idx[em]] = w.V[w.ridx(d)][em]


# ------------------------------------------------------------------ round 4: ops of the G2 round kernels (g2_rounds.py)
@op("v_mad_i64_i32")
def _mad_i64(w, i):
    d, sd, a, b, c = i.args
    x = np.asarray(np.broadcast_to(w.rd(a), (LANES,))).astype(np.int32).astype(np.int64)
    y = np.asarray(np.broadcast_to(w.rd(b), (LANES,))).astype(np.int32).astype(np.int64)
    c64 = np.asarray(np.broadcast_to(w.rd64(c), (LANES,)), dtype=U64).astype(np.int64)
    res = x * y + c64
    w.wr64(d, res.astype(U64))
    w.wr_lanemask(sd, np.zeros(LANES, dtype=bool))


def _f64_src(w, o):
    if isinstance(o, str):
        return np.float64(float(o))
    if isinstance(o, Reg):
        return np.asarray(np.broadcast_to(w.rd64(o), (LANES,)), dtype=U64).view(np.float64)
    return np.float64(o)


@op("v_cvt_f64_u32")
def _cvt_f64_u32(w, i):
    d, a = i.args
    w.wr64(d, np.asarray(np.broadcast_to(w.rd(a), (LANES,))).astype(np.float64).view(U64))


@op("v_add_f64")
def _add_f64(w, i):
    d, a, b = i.args
    w.wr64(d, np.asarray(np.broadcast_to(_f64_src(w, a) + _f64_src(w, b), (LANES,)), dtype=np.float64).view(U64))


@op("v_mul_f64")
def _mul_f64(w, i):
    d, a, b = i.args
    w.wr64(d, np.asarray(np.broadcast_to(_f64_src(w, a) * _f64_src(w, b), (LANES,)), dtype=np.float64).view(U64))


@op("v_cvt_u32_f64")
def _cvt_u32_f64(w, i):
    d, a = i.args
    v = np.trunc(np.asarray(np.broadcast_to(_f64_src(w, a), (LANES,)), dtype=np.float64))
    w.wr(d, np.clip(v, 0, 4294967295.0).astype(U64).astype(U32))


@op("ds_bpermute_b32")
def _ds_bpermute(w, i):
    # dst[lane] = src[(addr[lane] >> 2) & 63]; a source lane that is disabled by EXEC delivers 0 (ISA manual)
    d, addr, src = i.args
    a = (np.asarray(np.broadcast_to(w.rd(addr), (LANES,))).astype(np.int64) + int(i.mods.get("offset", 0))) >> 2 & 63
    v = np.asarray(np.broadcast_to(w.rd(src), (LANES,)), dtype=U32)
    em = w.em()
    w.wr(d, np.where(em[a], v[a], U32(0)))


@op("s_lshl_b64")
def _s_lshl64(w, i):
    d, a, b = i.args
    r = (_s64(a, w) << (w.rs(b) & 63)) & 0xFFFFFFFFFFFFFFFF
    w.wr_smask(d, r)
    w.scc = 1 if r else 0


@op("s_mul_hi_u32")
def _s_mul_hi(w, i):
    d, a, b = i.args
    w.ws(d, (w.rs(a) * w.rs(b)) >> 32)


@op("s_min_u32")
def _s_min(w, i):
    d, a, b = i.args
    x, y = w.rs(a), w.rs(b)
    w.ws(d, min(x, y))
    w.scc = 1 if x < y else 0


@op("s_sub_i32")
def _s_sub_i32(w, i):
    d, a, b = i.args
    w.ws(d, w.rs(a) - w.rs(b))


@op("s_subb_u32")
def _s_subb(w, i):
    d, a, b = i.args
    r = w.rs(a) - w.rs(b) - w.scc
    w.ws(d, r)
    w.scc = 1 if r < 0 else 0


HANDLERS["v_mul_u32_u24"] = _bin(lambda a, b: ((np.asarray(a) & U32(0xFFFFFF)).astype(U64) * (np.asarray(b) & U32(0xFFFFFF)).astype(U64)
                                                & U64(M32)).astype(U32))


@op("global_atomic_add")
def _gatomic_add(w, i):
    d, addr_op, data, saddr = i.args
    addrs = np.broadcast_to(_vaddr(w, i, addr_op, saddr), (LANES,))
    em = w.em()
    vals = np.asarray(np.broadcast_to(w.rd(data), (LANES,)), dtype=U32)
    for l in np.nonzero(em)[0]:
        old = int(w.mem.load(int(addrs[l]), 1)[0])
        w.mem.store(int(addrs[l]), [(old + int(vals[l])) & M32])
        w.V[w.ridx(d)][l] = old
