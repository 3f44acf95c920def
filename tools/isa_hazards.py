#!/usr/bin/env python3
"""Static check of the MFMA hazards hipcc does not see inside `asm volatile` statements, on the ISA that actually ships.

The scan kernels place MFMAs and the VALU / DS instructions that consume their results by hand (k_scan_hist_m2: inline-asm
statements with `s_nop`, early-clobber operands and statement order; k_scan_hist_b: an inline-asm `ds_add_u32` fed by MFMA
results; DESIGN.md section 3.1 lists the five hazards, "each a wrong result on hardware first").  hipcc inserts wait states only
between instructions it scheduled itself, so a compiler or source change can silently break one of them.  This module
disassembles the gfx950 code objects inside libxmh.so (llvm-objdump, no GPU needed) and checks, instruction by instruction:

  R1  MFMA result -> first VALU / DS / VMEM read of it: at least `RAW_WAIT` wait states (one per instruction, N + 1 per `s_nop N`);
  R2  (hand-scheduled kernels only) VALU write of a register an MFMA read as srcC / A / B: not within `WAR_WAIT` wait states behind
      that MFMA -- hipcc itself overwrites A / B operands in the very next slot elsewhere, so this pins the distances the
      hand-placed statements were validated with on hardware rather than an architectural minimum;
  R3  VALU write -> MFMA reading the register (srcC, A or B): at least `SRCC_WAIT` wait states (the `s_nop 3` that opens every statement);
  R4  two SDWA byte inserts (dst_unused:UNUSED_PRESERVE) into one register: never back to back (dst_sel forwarding).

Straight-line analysis in address order: state is dropped at unconditional branches (a taken branch costs more than any of these
distances) and kept across conditional ones (the fall-through of a branch not taken follows one slot later).  `python tools/isa_hazards.py [libxmh.so]` prints a report;
tests/test_isa_hazards.py asserts it."""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "clip-based-cross-modal-hash_amd", "xmh", "libxmh.so")
LLVM_BIN = os.environ.get("XMH_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

RAW_WAIT = 8      # v_mfma_i32_16x16x64_i8 result -> VALU / DS read (xmh_scan.hip: "needs 8 wait states")
WAR_WAIT = 3      # VALU write onto the A / B / srcC registers of an MFMA issued this many (or fewer) slots earlier
SRCC_WAIT = 2     # VALU write -> MFMA srcC read

Violation = collections.namedtuple("Violation", "rule kernel index text detail")
Insn = collections.namedtuple("Insn", "mnem dst src text addr", defaults=(None,))

_REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def _regs(operand):
    out = []
    for m in _REG.finditer(operand):
        bank = m.group(1)
        if m.group(2) is not None:
            out.append((bank, int(m.group(2))))
        else:
            out.extend((bank, i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


_NO_VDST = ("ds_write", "ds_add_u32", "ds_add_f32", "ds_inc_u32", "ds_dec_u32", "ds_or_b32", "ds_and_b32", "ds_max_u32", "ds_min_u32", "ds_add_u64",
            "global_store", "buffer_store", "flat_store", "scratch_store", "global_load_lds", "global_atomic", "flat_atomic", "buffer_atomic",
            "v_cmp", "v_readlane", "v_readfirstlane", "ds_gws", "ds_nop", "buffer_wbl2", "buffer_inv", "buffer_gl")
_DST_ALSO_READ = ("v_fmac", "v_mac", "v_dot2c", "v_dot4c", "v_dot8c", "v_pk_fmac", "v_movrel", "v_writelane")


def parse(text):
    """llvm-objdump -d text -> {mangled kernel name: [Insn]}"""
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        am = re.search(r"//\s*([0-9A-Fa-f]{8,16}):", line)
        addr = int(am.group(1), 16) if am else None
        parts = body.split(None, 1)
        mnem = parts[0]
        ops = parts[1] if len(parts) > 1 else ""
        operands = [o.strip() for o in ops.split(",")] if ops else []
        dst, src = [], []
        if mnem.startswith(("s_", "buffer_wbl2", "buffer_inv")) and not mnem.startswith("s_nop"):
            pass
        has_dst = not mnem.startswith(_NO_VDST) or "_rtn" in mnem
        if mnem.startswith(("global_atomic", "flat_atomic", "buffer_atomic")):
            has_dst = " sc0" in ops or " glc" in ops                  # returning form
            if has_dst:
                has_dst = bool(operands) and bool(_regs(operands[0])) and len(operands) >= 3
        for i, o in enumerate(operands):
            r = _regs(o.split(" ")[0] if i == len(operands) - 1 else o)
            if i == len(operands) - 1:                                # modifiers (offset:, dst_sel:, ...) follow the last operand
                r = _regs(o.split(" ")[0])
            if i == 0 and has_dst and r:
                dst = r
            else:
                src.extend(r)
        if dst and (mnem.startswith(_DST_ALSO_READ) or "UNUSED_PRESERVE" in ops):
            src.extend(dst)
        cur.append(Insn(mnem, tuple(dst), tuple(src), body, addr))
    return kernels


def _wait_states(insn):
    if insn.mnem == "s_nop":
        return int(insn.text.split()[1], 0) + 1
    return 1


def check(insns, kernel="", war=None):
    """-> (violations, stats).  stats: smallest distance seen per rule (None when the pattern does not occur).
    war: apply R2 (default: only to k_scan_hist_m2, the kernel whose MFMAs are inline asm)."""
    if war is None:
        war = "k_scan_hist_r2" in kernel or "k_scan_ap_r2" in kernel
    out = []
    stats = {"R1": None, "R2": None, "R3": None, "n_mfma": 0, "n_snop3": 0, "n_sdwa_preserve": 0}
    writer = {}           # reg -> ("mfma" | "valu" | "other", wait-state clock at issue)
    mfma_reads = []       # (clock, regs read as A/B/C, regs read as srcC)
    last_sdwa = None      # (index, dst regs) of the immediately preceding instruction if it was a preserving SDWA insert
    clock = 0
    for idx, ins in enumerate(insns):
        m = ins.mnem
        if m.startswith(("s_branch", "s_endpgm", "s_setpc", "s_swappc")):
            writer.clear()
            mfma_reads = []
            last_sdwa = None
            clock += 16
            continue
        if m.startswith("s_cbranch"):                                  # not taken: the next instruction follows one slot later with everything still in flight
            last_sdwa = None
            clock += 1
            continue
        is_mfma = m.startswith(("v_mfma", "v_smfmac"))
        is_valu = m.startswith("v_") and not is_mfma
        if m == "s_nop" and ins.text.split()[1] == "3":
            stats["n_snop3"] += 1
        if is_mfma:
            stats["n_mfma"] += 1
            srcc = _regs(ins.text.split(",")[-1].split()[0]) if ins.text.count(",") >= 3 else []
            for r in set(srcc) | set(ins.src):                        # R3: srcC, and A / B alike (k_scan_hist_r2 builds its A tiles with VALU instructions)
                w = writer.get(r)
                if w and w[0] == "valu":
                    d = clock - w[1] - 1
                    stats["R3"] = d if stats["R3"] is None else min(stats["R3"], d)
                    if d < SRCC_WAIT:
                        out.append(Violation("R3", kernel, idx, ins.text, "operand %s%d written by a VALU %d wait states earlier" % (r[0], r[1], d)))
            mfma_reads.append((clock, set(ins.src), set(srcc)))
            mfma_reads = [x for x in mfma_reads if clock - x[0] <= 32]
        else:
            for r in ins.src:                                         # R1
                w = writer.get(r)
                if w and w[0] == "mfma":
                    d = clock - w[1] - 1
                    stats["R1"] = d if stats["R1"] is None else min(stats["R1"], d)
                    if d < RAW_WAIT:
                        out.append(Violation("R1", kernel, idx, ins.text, "%s%d is an MFMA result only %d wait states old" % (r[0], r[1], d)))
            if is_valu and ins.dst:                                   # R2
                for (c0, regs, _srcc) in mfma_reads:
                    hit = regs.intersection(ins.dst)
                    if hit:
                        d = clock - c0 - 1
                        stats["R2"] = d if stats["R2"] is None else min(stats["R2"], d)
                        if war and d < WAR_WAIT:
                            r = sorted(hit)[0]
                            out.append(Violation("R2", kernel, idx, ins.text, "writes %s%d, an operand of the MFMA %d wait states earlier" % (r[0], r[1], d)))
        if "_sdwa" in m and "UNUSED_PRESERVE" in ins.text:            # R4
            stats["n_sdwa_preserve"] += 1
            if last_sdwa is not None and set(last_sdwa[1]).intersection(ins.dst):
                out.append(Violation("R4", kernel, idx, ins.text, "SDWA insert directly behind another insert into the same register"))
        last_sdwa = (idx, ins.dst) if ("_sdwa" in m and ins.dst) else None
        kind = "mfma" if is_mfma else ("valu" if is_valu else "other")
        for r in ins.dst:
            writer[r] = (kind, clock)
        clock += _wait_states(ins)
    return out, stats


def check_lds_returns(insns, kernel=""):
    """R5: nothing touches the destination of a returning LDS operation before an `s_waitcnt lgkmcnt` has covered it.
    The scan's pass 2 issues its returning atomics from `asm volatile` statements and waits for them a group later with a counted
    lgkmcnt; hipcc believes the result is there when the statement ends, so any copy or use it schedules in between moves garbage
    (round 4: a `v_mov_b64` of eight result pairs on the path of a chunk with exactly one whole batch -- wrong APs a few runs in a
    thousand).  LDS operations of a wave complete in order, so the outstanding ones are a FIFO: `lgkmcnt(N)` retires all but the N newest.
    Scalar loads share the counter and may return out of order: one of them in flight makes every counted wait inexact, which is flagged
    as well.  Pending sets travel along branches (target address = next instruction + 4 * simm16) as well as along the fall-through."""
    out = []
    fifo = []                                   # [(kind, frozenset(dst regs), text)] oldest first; kind "ds" / "smem"
    at_target = collections.defaultdict(list)   # address -> FIFOs arriving by a branch
    stats = {"n_returning": 0, "max_outstanding": 0}

    def merge(a, b):                            # union of two arrival states: keep every pending destination, order by the longer list
        longer, other = (a, b) if len(a) >= len(b) else (b, a)
        extra = [x for x in other if x not in longer]
        return extra + list(longer)

    for idx, ins in enumerate(insns):
        if ins.addr is not None and ins.addr in at_target:
            for f in at_target.pop(ins.addr):
                fifo = merge(fifo, f)
        m = ins.mnem
        if m == "s_waitcnt":
            g = re.search(r"lgkmcnt\((\d+)\)", ins.text)
            if g:
                n = int(g.group(1))
                if n and any(k == "smem" for k, _, _ in fifo):
                    out.append(Violation("R5", kernel, idx, ins.text, "counted lgkmcnt with a scalar load in flight (it may return out of order)"))
                fifo = fifo[len(fifo) - n:] if n < len(fifo) else fifo
                if n == 0:
                    fifo = []
            continue
        pending = set().union(*[d for _, d, _ in fifo]) if fifo else set()
        touched = pending.intersection(set(ins.src) | set(ins.dst))
        if touched:
            r = sorted(touched)[0]
            out.append(Violation("R5", kernel, idx, ins.text, "%s%d is the destination of an LDS operation no s_waitcnt has covered yet" % (r[0], r[1])))
        if m.startswith(("ds_", "s_load", "s_buffer_load")) and not m.startswith("ds_nop"):
            returning = m.startswith(("s_load", "s_buffer_load")) or "_rtn" in m or m.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append"))
            dst = frozenset(ins.dst) if (returning and not m.startswith("s_")) else frozenset()
            fifo.append(("smem" if m.startswith("s_") else "ds", dst, ins.text))
            if dst:
                stats["n_returning"] += 1
            stats["max_outstanding"] = max(stats["max_outstanding"], len(fifo))
        if m.startswith(("s_branch", "s_cbranch")) and ins.addr is not None:
            g = re.match(r"\S+\s+(-?\d+)", ins.text)
            if g:
                off = int(g.group(1))
                if off >= 32768:
                    off -= 65536
                at_target[ins.addr + 4 + 4 * off].append(list(fifo))
            if m.startswith("s_branch"):
                fifo = []
        if m.startswith(("s_endpgm", "s_setpc", "s_swappc")):
            fifo = []
    return out, stats


def code_objects(lib=LIB, workdir=None):
    """extract the gfx950 code objects of the fat binary into workdir (a copy of the library is unbundled there) -> file paths"""
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    if not os.path.exists(objdump):
        raise FileNotFoundError(objdump)
    work = workdir or tempfile.mkdtemp(prefix="xmh_isa_")
    local = os.path.join(work, "lib.so")
    shutil.copy(lib, local)
    subprocess.run([objdump, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=work)
    return sorted(os.path.join(work, f) for f in os.listdir(work) if f.startswith("lib.so.") and f.endswith("gfx950"))


def disassemble(path):
    return subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", path], check=True, capture_output=True, text=True).stdout


# pass 2 (asm-issued returning atomics) and the pass-1 kernel that reads its A tiles with asm ds_read_b128 ahead of a counted wait
LDS_RETURN_KERNELS = ("k_scan_ap_c", "k_scan_ap_s", "k_scan_ap_r2")


def analyse_lds_returns(lib=LIB, name_filter=LDS_RETURN_KERNELS):
    """-> {kernel: (violations, stats)} of rule R5 for the pass-2 kernels (asm-issued returning atomics)"""
    work = tempfile.mkdtemp(prefix="xmh_isa_")
    try:
        res = {}
        for co in code_objects(lib, work):
            text = disassemble(co)
            if not any(n in text for n in name_filter):
                continue
            for name, insns in parse(text).items():
                if any(n in name for n in name_filter):
                    res[name] = check_lds_returns(insns, name)
        return res
    finally:
        shutil.rmtree(work, ignore_errors=True)


def analyse(lib=LIB, name_filter=("k_scan_hist_r2", "k_scan_hist_b", "k_scan_ap_r2", "k_topk_filter_mfma")):
    """-> {kernel: (violations, stats)} for every kernel whose mangled name contains one of name_filter"""
    work = tempfile.mkdtemp(prefix="xmh_isa_")
    try:
        res = {}
        for co in code_objects(lib, work):
            text = disassemble(co)
            if not any(n in text for n in name_filter):
                continue
            for name, insns in parse(text).items():
                if any(n in name for n in name_filter):
                    res[name] = check(insns, name)
        return res
    finally:
        shutil.rmtree(work, ignore_errors=True)


def hipcc_version():
    try:
        out = subprocess.run([os.path.join(LLVM_BIN, "clang"), "--version"], capture_output=True, text=True).stdout
        return out.splitlines()[0] if out else "unknown"
    except OSError:
        return "unknown"


if __name__ == "__main__":
    res = analyse(sys.argv[1] if len(sys.argv) > 1 else LIB)
    print("compiler:", hipcc_version())
    bad = 0
    for name in sorted(res):
        v, st = res[name]
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:70]
        print("%-72s mfma %4d  s_nop3 %3d  sdwa %3d  min R1 %s R2 %s R3 %s  violations %d" % (short, st["n_mfma"], st["n_snop3"], st["n_sdwa_preserve"], st["R1"], st["R2"], st["R3"], len(v)))
        for x in v[:6]:
            print("    ", x.rule, x.index, x.text[:100], "--", x.detail)
        bad += len(v)
    r5 = analyse_lds_returns(sys.argv[1] if len(sys.argv) > 1 else LIB)
    for name in sorted(r5):
        v, st = r5[name]
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:70]
        print("%-72s returning LDS ops %4d  max outstanding %2d  R5 violations %d" % (short, st["n_returning"], st["max_outstanding"], len(v)))
        for x in v[:4]:
            print("    ", x.rule, x.index, x.text[:100], "--", x.detail)
        bad += len(v)
    sys.exit(1 if bad else 0)
