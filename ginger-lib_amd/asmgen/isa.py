"""isa.py -- a small gfx950 program builder: instructions are kept as structured tuples so that the same program
can be (a) printed as assembler text for clang -x assembler -mcpu=gfx950 and (b) executed by sim.py on the CPU.

Why hand-allocated assembly: the Montgomery product kernels are register-allocation bound under
hipcc (spills in the G1 update, one wave per SIMD and an out-of-line call ABI for the towers).  Here every field element
has a fixed home (a "slot" of 26 consecutive VGPRs, or AGPRs), the prime lives in SGPRs, and the instruction order is
the generator's: a product is ONE chain of v_mad_u64_u32, each taking its 64-bit addend from its predecessor (which the
pipeline forwards: 4 cycles per mad; alternating two accumulators costs a fifth cycle for the register-file read of the
addend -- tools/asm_mb).

Operands:  V(i) / V(i, n)  VGPR or VGPR range;  A(i) AGPR;  S(i) / S(i, n) SGPR (range);  an int = immediate;
           VCC, EXEC;  a str = label.
"""


class Reg(tuple):
    __slots__ = ()

    def __new__(cls, kind, idx, n=1):
        return tuple.__new__(cls, (kind, idx, n))

    kind = property(lambda s: s[0])
    idx = property(lambda s: s[1])
    n = property(lambda s: s[2])

    def __repr__(self):
        k, i, n = self
        if k in ("vcc", "exec"):
            return k
        return "%s%d" % (k, i) if n == 1 else "%s[%d:%d]" % (k, i, i + n - 1)

    def sub(self, j, n=1):
        assert 0 <= j and j + n <= self.n, (self, j, n)
        return Reg(self.kind, self.idx + j, n)

    def lo(self):
        return self.sub(0)

    def hi(self):
        return self.sub(1)


def V(i, n=1):
    return Reg("v", i, n)


def A(i, n=1):
    return Reg("a", i, n)


def S(i, n=1):
    return Reg("s", i, n)


VCC = Reg("vcc", 106, 2)     # s[106:107] on gfx9
EXEC = Reg("exec", 126, 2)
OFF = "off"


def fmt(o):
    if isinstance(o, Reg):
        return repr(o)
    if isinstance(o, bool):
        raise TypeError(o)
    if isinstance(o, int):
        if -16 <= o <= 64:
            return str(o)
        return "0x%x" % (o & 0xFFFFFFFF)
    return str(o)


class Ins:
    __slots__ = ("op", "args", "mods", "comment")

    def __init__(self, op, args, mods, comment=None):
        self.op, self.args, self.mods, self.comment = op, args, mods, comment

    def text(self):
        if self.op == "label":
            return "%s:" % self.args[0]
        if self.op == "comment":
            return "\t; %s" % self.args[0]
        if self.op == "long_branch":
            # s_branch reaches +-2^15 dwords; the update loop is longer.  PC-relative jump through an SGPR pair (what LLVM's
            # branch relaxation emits); clobbers the pair and SCC.
            target, tmp, post = self.args
            return "\n".join([
                "\ts_getpc_b64 %s" % fmt(tmp),
                "%s:" % post,
                "\ts_add_u32 %s, %s, (%s-%s)&4294967295" % (fmt(tmp.lo()), fmt(tmp.lo()), target, post),
                "\ts_addc_u32 %s, %s, (%s-%s)>>32" % (fmt(tmp.hi()), fmt(tmp.hi()), target, post),
                "\ts_setpc_b64 %s" % fmt(tmp)])
        if self.op == "s_waitcnt":
            parts = []
            for k in ("vmcnt", "lgkmcnt", "expcnt"):
                if k in self.mods:
                    parts.append("%s(%d)" % (k, self.mods[k]))
            return "\ts_waitcnt " + " ".join(parts)
        s = "\t%s %s" % (self.op, ", ".join(fmt(a) for a in self.args))
        for k, v in self.mods.items():
            if v is True:
                s += " %s" % k
            else:
                s += " %s:%s" % (k, v)
        if self.comment:
            s += "\t; " + self.comment
        return s


VALU_PREFIX = ("v_",)


class Prog:
    """An instruction list plus the kernel descriptor fields."""

    def __init__(self, name):
        self.name = name
        self.ins = []
        self.lds_bytes = 0
        self.kernarg_bytes = 0
        self.args_meta = []          # (offset, size, kind)
        self.max_v = 0
        self.max_a = -1
        self.max_s = 0
        self._uniq = 0

    # ---- emission
    def emit(self, op, *args, comment=None, **mods):
        for a in args:
            if isinstance(a, Reg):
                top = a.idx + a.n
                if a.kind == "v":
                    self.max_v = max(self.max_v, top)
                elif a.kind == "a":
                    self.max_a = max(self.max_a, top)
                elif a.kind == "s":
                    self.max_s = max(self.max_s, top)
        self.ins.append(Ins(op, args, mods, comment))

    def __getattr__(self, op):
        if op.startswith(("v_", "s_", "ds_", "global_", "buffer_")):
            return lambda *a, **m: self.emit(op, *a, **m)
        raise AttributeError(op)

    def long_branch(self, target, tmp):
        self.emit("long_branch", target, tmp, self.uniq("post_getpc"))

    def label(self, name):
        self.ins.append(Ins("label", (name,), {}))

    def comment(self, text):
        self.ins.append(Ins("comment", (text,), {}))

    def uniq(self, stem):
        self._uniq += 1
        return ".L%s_%s_%d" % (self.name, stem, self._uniq)

    def count(self, pred=None):
        n = 0
        for i in self.ins:
            if i.op in ("label", "comment"):
                continue
            if pred is None or pred(i):
                n += 1
        return n

    # ---- output
    def body_text(self):
        return "\n".join(i.text() for i in self.ins)

    def kernel_text(self, waves_per_simd_hint=None):
        nv = (self.max_v + 7) // 8 * 8
        na = max(self.max_a, 0)
        accum_offset = max(4, (self.max_v + 3) // 4 * 4)
        next_free_vgpr = accum_offset + na if na else self.max_v
        out = []
        out.append("\t.text")
        out.append("\t.protected %s" % self.name)
        out.append("\t.globl %s" % self.name)
        out.append("\t.p2align 8")
        out.append("\t.type %s,@function" % self.name)
        out.append("%s:" % self.name)
        out.append(self.body_text())
        out.append(".L%s_end:" % self.name)
        out.append("\t.size %s, .L%s_end-%s" % (self.name, self.name, self.name))
        out.append("\t.section .rodata,\"a\",@progbits")
        out.append("\t.p2align 6, 0x0")
        out.append("\t.amdhsa_kernel %s" % self.name)
        kd = [
            ("group_segment_fixed_size", self.lds_bytes),
            ("private_segment_fixed_size", 0),
            ("kernarg_size", self.kernarg_bytes),
            ("user_sgpr_count", 2),
            ("user_sgpr_kernarg_segment_ptr", 1),
            ("uses_dynamic_stack", 0),
            ("enable_private_segment", 0),
            ("system_sgpr_workgroup_id_x", 1),
            ("system_sgpr_workgroup_id_y", 0),
            ("system_sgpr_workgroup_id_z", 0),
            ("system_vgpr_workitem_id", 0),
            ("next_free_vgpr", max(next_free_vgpr, 1)),
            ("next_free_sgpr", max(self.max_s, 1)),
            ("accum_offset", accum_offset),
            ("reserve_vcc", 1),
            ("float_denorm_mode_32", 3),
            ("float_denorm_mode_16_64", 3),
            ("dx10_clamp", 1),
            ("ieee_mode", 1),
        ]
        for k, v in kd:
            out.append("\t\t.amdhsa_%s %d" % (k, v))
        out.append("\t.end_amdhsa_kernel")
        out.append("\t.text")
        return "\n".join(out), dict(vgprs=self.max_v, agprs=na, sgprs=self.max_s, accum_offset=accum_offset, alloc=nv)

    def metadata_yaml(self):
        na = max(self.max_a, 0)
        lines = []
        lines.append("  - .agpr_count: %d" % na)
        lines.append("    .args:")
        for off, size, kind in self.args_meta:
            lines.append("      - .offset: %d" % off)
            lines.append("        .size: %d" % size)
            if kind == "ptr":
                lines.append("        .value_kind: global_buffer")
                lines.append("        .address_space: global")
            else:
                lines.append("        .value_kind: by_value")
        lines.append("    .group_segment_fixed_size: %d" % self.lds_bytes)
        lines.append("    .kernarg_segment_align: 8")
        lines.append("    .kernarg_segment_size: %d" % self.kernarg_bytes)
        lines.append("    .max_flat_workgroup_size: 256")
        lines.append("    .name: %s" % self.name)
        lines.append("    .private_segment_fixed_size: 0")
        lines.append("    .sgpr_count: %d" % (self.max_s + 6))
        lines.append("    .sgpr_spill_count: 0")
        lines.append("    .symbol: %s.kd" % self.name)
        lines.append("    .uses_dynamic_stack: false")
        lines.append("    .vgpr_count: %d" % self.max_v)
        lines.append("    .vgpr_spill_count: 0")
        lines.append("    .wavefront_size: 64")
        return "\n".join(lines)

    def add_arg(self, size, kind):
        align = size
        off = (self.kernarg_bytes + align - 1) // align * align
        self.args_meta.append((off, size, kind))
        self.kernarg_bytes = off + size
        return off


def module_text(progs):
    out = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.amdhsa_code_object_version 6"]
    infos = {}
    for p in progs:
        t, info = p.kernel_text()
        out.append(t)
        infos[p.name] = info
    out.append("\t.amdgpu_metadata")
    out.append("---")
    out.append("amdhsa.kernels:")
    for p in progs:
        out.append(p.metadata_yaml())
    out.append("amdhsa.target: amdgcn-amd-amdhsa--gfx950")
    out.append("amdhsa.version:")
    out.append("  - 1")
    out.append("  - 2")
    out.append("...")
    out.append("\t.end_amdgpu_metadata")
    return "\n".join(out) + "\n", infos


# ---------------------------------------------------------------------------------------------------------------------------
# gfx940 / gfx950 data hazards that the hardware does not interlock (LLVM GCNHazardRecognizer::checkVALUHazards, the
# hasVDecCoExecHazard() block -- hipcc pads the same places with s_nop):
#   * a VALU instruction writes an SGPR (v_cmp, carry-out, v_readfirstlane) or VCC -> a VALU instruction reads that SGPR / VCC
#     (select mask, carry-in, scalar source): 2 wait states in between;
#   * a VALU instruction writes a VGPR -> v_readlane / v_readfirstlane reads it: 1 wait state;
#   * a VALU instruction writes a VGPR -> a DPP instruction reads it: 2 wait states (all of GFX9).
# A wait state is one issued instruction of the wave (s_nop N counts N + 1).  Found on the card: a v_readfirstlane right behind
# the shift that made its operand read the register's OLD value in one kernel and the new one in its twin with identical code.
# fix_hazards() pads a finished program with the s_nop the rules ask for; labels are joins: the padding after a label assumes
# the worst predecessor (a writer right before the branch, the branch being one wait state).
CARRY_OUT_OPS = {"v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32",
                 "v_mad_u64_u32", "v_mad_i64_i32"}
SGPR_WAIT, READLANE_WAIT, DPP_WAIT = 2, 1, 2


def _sregs(o):
    if isinstance(o, Reg) and o.kind in ("s", "vcc"):
        return range(o.idx, o.idx + o.n)
    return ()


def _vregs(o):
    if isinstance(o, Reg) and o.kind in ("v", "a"):
        base = o.idx + (256 if o.kind == "a" else 0)
        return range(base, base + o.n)
    return ()


def _wait_states(i):
    if i.op in ("label", "comment"):
        return 0
    if i.op == "s_nop":
        return int(i.args[0]) + 1
    if i.op == "long_branch":
        return 4
    return 1


def hazard_scan(prog, fix):
    """returns (list of findings, new instruction list); fix: insert the missing wait states"""
    out, found = [], []
    pos = 0
    s_write, v_write = {}, {}        # register -> wait-state position of the last VALU write
    join = None                      # position of the last label: stands for a VALU write of every register right before the branch
    for i in prog.ins:
        if i.op == "label":
            join = pos
            out.append(i)
            continue
        if i.op == "comment":
            out.append(i)
            continue
        valu = i.op.startswith("v_")
        need = 0
        if valu:
            if i.op in ("v_cmp_eq_u32",) or i.op.startswith("v_cmp_") or i.op in ("v_readfirstlane_b32", "v_readlane_b32"):
                wr_s, srcs = list(_sregs(i.args[0])), i.args[1:]
            elif i.op in CARRY_OUT_OPS:
                wr_s, srcs = list(_sregs(i.args[1])), i.args[2:]
            else:
                wr_s, srcs = [], i.args[1:]
            wr_v = list(_vregs(i.args[0]))
            for o in srcs:
                for r in _sregs(o):
                    last = s_write.get(r)
                    if join is not None and (last is None or join - 1 > last):
                        last = join - 1
                    if last is not None:
                        need = max(need, SGPR_WAIT - (pos - last - 1))
            vwait = READLANE_WAIT if i.op in ("v_readfirstlane_b32", "v_readlane_b32") else (DPP_WAIT if i.op.endswith("_dpp") else 0)
            if vwait:
                for o in srcs:
                    for r in _vregs(o):
                        last = v_write.get(r)
                        if join is not None and (last is None or join - 1 > last):
                            last = join - 1
                        if last is not None:
                            need = max(need, vwait - (pos - last - 1))
        if need > 0:
            found.append((len(out), i.op, need))
            if fix:
                out.append(Ins("s_nop", (need - 1,), {}))
                pos += need
        out.append(i)
        if valu:
            for r in wr_s:
                s_write[r] = pos
            for r in wr_v:
                v_write[r] = pos
        pos += _wait_states(i)
    return found, out


def fix_hazards(prog):
    """pads prog in place; returns the number of s_nop inserted"""
    found, out = hazard_scan(prog, True)
    prog.ins = out
    return len(found)
