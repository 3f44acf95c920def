#!/usr/bin/env python3
"""Count the VALU instructions per 64 (query, item) pairs in the steady-state loops of the scan kernels, from the ISA hipcc emits
for the very sources that are built into libxmh.so (-save-temps).  Writes profiles/<tag>_scan_isa.json (read by bench_roofline.py)
and profiles/<tag>_scan_isa.txt (the loop bodies themselves, so a reader can check the counts).

    python tools/isa_count.py r02

The hot path of a kernel = the straight run of code (no branch inside) with the most LDS atomics: hipcc emits the unrolled full
64-item batch that way.  One LDS atomic = one step of a wave = 64 pairs, so VALU per 64 pairs = VALU in the run / atomics in
the run.  This is a STATIC count of the hot path: per-batch prologue code (loads, staging) and conditionally executed pieces
that hipcc moved out of the run (pass 2: the credit of the first group of a batch sits behind the `have_prev` test) are not in
it; the dynamic count per launch is SQ_INSTS_VALU in the rocprofv3 PMC summary (bench_roofline.py reports both)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "clip-based-cross-modal-hash_amd", "csrc", "xmh_scan.hip")

# kernel instances as xmh_scan_describe / rocprofv3 spell them; the json key is the same string without blanks (bench_roofline.py)
#   k_scan_hist_s<W, LW, TERN, S, NW, CACHE>;  k_scan_ap_s<W, LW, TERN, CAPPED, S, P32, MASKED, NW, CACHE>;
#   k_scan_hist_m<NMC, NML, NW, CACHE, BYTE>;  k_scan_hist_m2<NML, NW, NQ, CACHE, STAMP>;  k_scan_ap_c<CAPPED>
KERNELS = [
    "k_scan_hist_r2<2, 4, 4, true>", "k_scan_hist_r2<2, 4, 4, false>", "k_scan_hist_r2w<2, 4, 2, true>", "k_scan_ap_c<false, 8, false>", "k_scan_ap_c<false, 8, true>",
    "k_scan_ap_r2<2, 4, 2, false>",
    "k_scan_hist_s<2, 3, false, 4, 1, true>", "k_scan_hist_s<2, 3, false, 4, 1, false>", "k_scan_hist_s<1, 3, false, 2, 1, false>",
    "k_scan_ap_s<2, 1, false, false, 4, false, false, 1, true>", "k_scan_ap_s<2, 3, false, true, 4, false, false, 1, false>",
    "k_scan_ap_s<1, 3, false, true, 2, false, false, 1, false>",
    "k_scan_ap_s<4, 1, false, false, 8, true, false, 1, true>", "k_scan_ap_s<4, 1, false, false, 8, false, false, 1, true>",
    "k_scan_ap_s<8, 1, false, false, 8, false, false, 1, true>", "k_scan_ap_s<8, 3, false, true, 8, false, false, 1, false>",
]


def mangled(instance):
    """("k_scan_ap_c", "ILb0ELi8EE") for "k_scan_ap_c<false, 8>": the Itanium spelling of integer / bool template arguments"""
    m = re.match(r"(\w+)<(.*)>", instance)
    args = [a.strip() for a in m.group(2).split(",")]
    enc = "".join(("Lb%dE" % (a == "true")) if a in ("true", "false") else "Li%sE" % a for a in args)
    return m.group(1), "I" + enc + "E"


def kernel_bodies(asm):
    """mangled name -> list of instruction/label lines"""
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):                 # (not the first s_endpgm: gated kernels return early)
                cur = None
                continue
            cur.append(line)
    return out


def best_loop(lines):
    """-> (instruction lines executed per full 64-item batch, LDS atomics among them).
    hipcc lays the unrolled full-batch path out as one straight run of code (possibly after the loop's own span, jumping back to
    the loop header), and keeps the ragged-batch path inside the span.  So: the straight run with the most LDS atomics = the hot
    path; the per-batch prologue = the code from the label its final branch returns to, up to the branch that enters it."""
    is_label = lambda l: re.match(r"^(\.LBB\d+_\d+):", l)                     # noqa: E731
    is_branch = lambda l: re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)    # noqa: E731
    is_atomic = lambda l: re.match(r"\s+ds_(add|inc)", l)                      # noqa: E731
    # straight runs: split at branches only (labels that are merely fall-through targets stay inside a run)
    runs, start = [], 0
    for i, l in enumerate(lines):
        if is_branch(l):
            runs.append((start, i))
            start = i + 1
    runs.append((start, len(lines) - 1))
    a, b = max(runs, key=lambda r: sum(1 for l in lines[r[0]:r[1] + 1] if is_atomic(l)))
    natom = sum(1 for l in lines[a:b + 1] if is_atomic(l))
    if not natom:
        return None, 0
    # trim the run to start at the last label before its first atomic that some branch targets (the entry of the hot path)
    first_atomic = next(i for i in range(a, b + 1) if is_atomic(lines[i]))
    targets = {is_branch(l).group(1) for l in lines if is_branch(l)}
    entry = a
    for i in range(a, first_atomic):
        m = is_label(lines[i])
        if m and m.group(1) in targets:
            entry = i
    hot = lines[entry:b + 1]
    entry_label = is_label(lines[entry]).group(1) if is_label(lines[entry]) else None
    back = is_branch(lines[b]).group(1) if is_branch(lines[b]) else None
    pro = []
    if entry_label and back:
        idx = {is_label(l).group(1): i for i, l in enumerate(lines) if is_label(l)}
        if back in idx:
            for i in range(idx[back], len(lines)):
                pro.append(lines[i])
                m = is_branch(lines[i])
                if m and m.group(1) == entry_label:
                    break
            else:
                pro = []
        if pro and sum(1 for l in pro if is_atomic(l)):
            pro = []                                           # walked through another path: do not guess
    return pro + hot, natom


def count(body):
    ops = collections.Counter()
    for l in body:
        m = re.match(r"\s+(v_\w+|ds_\w+|s_\w+|global_\w+|buffer_\w+)", l)
        if m:
            ops[m.group(1)] += 1
    return ops


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-save-temps", "-c", SRC,
                        "-o", os.path.join(tmp, "x.o")], cwd=tmp, check=True, stderr=subprocess.DEVNULL)
        asm = open([os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]).read()
    bodies = kernel_bodies(asm)
    result, text = {}, []
    for inst in KERNELS:
        key = re.sub(r"\s+", "", inst)
        kern, targs = mangled(inst)
        names = [n for n in bodies if kern + targs in n]
        if not names:
            continue
        body, natom = best_loop(bodies[names[0]])
        if not body or not natom:
            continue
        ops = count(body)
        is_valu = lambda k: k.startswith("v_") and not k.startswith("v_mfma")      # noqa: E731  (MFMAs issue to the matrix pipe)
        valu = {re.sub(r"_e(32|64)$|_sdwa$|_dpp$", "", k): 0 for k in ops if is_valu(k)}
        for k, v in ops.items():
            if is_valu(k):
                valu[re.sub(r"_e(32|64)$|_sdwa$|_dpp$", "", k)] += v
        mfma = sum(v for k, v in ops.items() if k.startswith("v_mfma"))
        per_step = {k: v / natom for k, v in sorted(valu.items())}
        result[key] = per_step
        text.append("=== %s  %s\n    loop: %d instructions, %d LDS atomics (= steps of 64 pairs); VALU per step %.2f, DS per step %.2f, SALU per step %.2f\n"
                    % (key, names[0], sum(ops.values()), natom, sum(valu.values()) / natom,
                       sum(v for k, v in ops.items() if k.startswith("ds_")) / natom, sum(v for k, v in ops.items() if k.startswith("s_")) / natom))
        text.append("    per step: " + ", ".join("%s %.2f" % kv for kv in per_step.items()) + (", MFMA %.3f" % (mfma / natom) if mfma else "") + "\n")
        text.extend(l + "\n" for l in body if l.strip() and not l.strip().startswith(";"))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(result, open(os.path.join(ROOT, "profiles", "%s_scan_isa.json" % tag), "w"), indent=1, sort_keys=True)
    open(os.path.join(ROOT, "profiles", "%s_scan_isa.txt" % tag), "w").writelines(text)
    for k, v in result.items():
        print("%-64s VALU per 64 pairs: %.2f" % (k, sum(v.values())))


if __name__ == "__main__":
    main()
