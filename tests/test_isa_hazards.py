"""The hand-placed MFMA hazards of the scan kernels ("each item was a wrong result on hardware
first") as a regression gate that needs no GPU: disassemble the gfx950 code objects inside the libxmh.so the build just produced
and check wait states, operand overwrites and SDWA spacing instruction by instruction (tools/isa_hazards.py).  A compiler upgrade,
or an edit that drops one `s_nop` from xmh_scan.hip, fails here instead of changing mAP on a GPU box."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazards as H  # noqa: E402

# the compiler the pinned counts below were read from; the hazard rules hold for any compiler, the counts are only asserted for this one
PINNED_COMPILER = "AMD clang version 22.0.0git"

pytestmark = pytest.mark.skipif(not (os.path.exists(H.LIB) and os.path.exists(os.path.join(H.LLVM_BIN, "llvm-objdump"))),
                                reason="needs the built libxmh.so and ROCm's llvm-objdump")


@pytest.fixture(scope="module")
def report():
    return H.analyse()


def _template_ints(name, kernel):
    m = re.search(kernel + r"I((?:L[ib]\d+E)+)E", name)
    return [int(x) for x in re.findall(r"L[ib](\d+)E", m.group(1))]


def test_every_hand_scheduled_kernel_is_in_the_library(report):
    names = "\n".join(report)
    for k in ("k_scan_hist_r2ILi2ELi4ELi4ELb1E", "k_scan_hist_r2ILi1ELi4ELi4ELb0E", "k_scan_hist_r2wILi2ELi4ELi2ELb1E", "k_scan_hist_r2wILi1ELi4ELi2ELb0E",
              "k_scan_ap_r2ILi2ELi4ELi2ELb0E", "k_scan_ap_r2ILi1ELi4ELi2ELb1E", "k_scan_hist_bILi4E", "k_topk_filter_mfmaILi8E"):
        assert k in names, k
    assert sum(st["n_mfma"] for _, st in report.values()) > 400
    assert not any("k_scan_hist_m2" in n or "k_scan_hist_mI" in n or "k_scan_ap_mI" in n for n in report)      # round 5: the superseded generations are gone


def test_no_mfma_hazard_in_the_shipped_isa(report):
    """R1: MFMA result -> VALU / DS read at least 8 wait states later; R2 (k_scan_hist_m2): no VALU write onto an MFMA's operands within
    3 slots behind it; R3: VALU write -> srcC read at least 2 wait states; R4: SDWA byte inserts into one register never back to back."""
    bad = [v for vs, _ in report.values() for v in vs]
    assert not bad, "\n".join("%s %s[%d] %s -- %s" % (v.rule, v.kernel[:60], v.index, v.text[:80], v.detail) for v in bad[:20])


def test_pass2_without_a_pair_cache_keeps_the_statement_margins(report):
    """k_scan_ap_r2 (round 5) = k_scan_hist_r2's statements with returning atomics as consumers: the consumers of an MFMA result sit a full
    statement behind it (>= 12 wait states, 8 are required), every MFMA operand a VALU instruction wrote is at least 2 wait states old, no
    tile is rewritten within 3 slots of the last MFMA that read it; every statement opens with `s_nop 3` (4 NQ + 1 per instance)."""
    seen = 0
    for name, (bad, st) in report.items():
        if "k_scan_ap_r2" not in name:
            continue
        nml, nw, nq, capped = _template_ints(name, "k_scan_ap_r2")
        assert not bad, (name, bad[:3])
        assert st["R1"] is not None and st["R1"] >= 12, (name, st)
        assert st["R3"] is not None and st["R3"] >= H.SRCC_WAIT, (name, st)
        assert st["R2"] is None or st["R2"] >= H.WAR_WAIT, (name, st)
        if H.hipcc_version().startswith(PINNED_COMPILER):
            assert st["n_snop3"] == 4 * nq + 1, (name, st)
        seen += 1
    assert seen == 4


def test_register_built_operands_keep_the_same_margins(report):
    """k_scan_hist_r2: the consumers of an MFMA result sit a full statement behind it (>= 12 wait states, 8 are required); the pair-cache
    variants carry 12 preserving SDWA inserts per query group of a batch body; the A tiles are built by VALU instructions from the packed
    words, so every MFMA operand a VALU instruction wrote is at least 2 wait states old (R3 on A / B, covered by the `s_nop 3`
    that opens each statement) and a tile is not overwritten within 3 slots of the last MFMA that read it (R2: each group has its own
    tile set, kept alive one statement longer by an empty asm)."""
    seen = wide = 0
    for name, (bad, st) in report.items():
        if "k_scan_hist_r2w" in name:                                  # 65..128 bits: six MFMAs per statement, the cache word packed by a second statement
            nml, nw, nq, cache = _template_ints(name, "k_scan_hist_r2w")
            assert not bad, (name, bad[:3])
            assert st["R1"] is not None and st["R1"] >= 8 and st["R3"] is not None and st["R3"] >= H.SRCC_WAIT, (name, st)
            assert st["R2"] is None or st["R2"] >= H.WAR_WAIT, (name, st)
            assert st["n_sdwa_preserve"] == (8 * nq if cache else 0), (name, st)
            wide += 1
            continue
        if "k_scan_hist_r2" not in name:
            continue
        nml, nw, nq, cache = _template_ints(name, "k_scan_hist_r2")
        assert not bad, (name, bad[:3])
        assert st["R1"] is not None and st["R1"] >= 12, (name, st)
        assert st["R3"] is not None and st["R3"] >= H.SRCC_WAIT, (name, st)
        assert st["R2"] is None or st["R2"] >= H.WAR_WAIT, (name, st)
        assert st["n_sdwa_preserve"] == (12 * nq if cache else 0), (name, st)
        if H.hipcc_version().startswith(PINNED_COMPILER):
            assert st["n_snop3"] == 4 * nq + 1, (name, st)
        seen += 1
    assert seen == 4 and wide == 4


def test_no_result_of_a_returning_lds_operation_is_touched_before_its_wait():
    """R5.  Pass 2 issues its returning atomics from asm statements and waits for them a group later with a counted lgkmcnt; hipcc takes
    an asm's result as present when the statement ends.  Round 4: in the one-group-per-batch variants of k_scan_ap_c it copied eight
    result pairs (v_mov_b64) in front of the wait on the path of a chunk with exactly one whole batch -- wrong APs a few evaluations in
    a thousand, found by a long fuzz run, not by any test.  Every pass-2 kernel of the library is checked here instruction by instruction."""
    res = H.analyse_lds_returns()
    names = "\n".join(res)
    for k in ("k_scan_ap_cILb0ELi8ELb0E", "k_scan_ap_cILb0ELi8ELb1E", "k_scan_ap_cILb1ELi8ELb1E", "k_scan_ap_sI", "k_scan_ap_r2ILi2ELi4ELi2ELb0E"):
        assert k in names, k
    bad = [v for vs, _ in res.values() for v in vs]
    assert not bad, "\n".join("%s %s[%d] %s -- %s" % (v.rule, v.kernel[:60], v.index, v.text[:80], v.detail) for v in bad[:20])
    assert sum(st["n_returning"] for _, st in res.values()) > 3000 and len(res) > 140     # every pass-2 instance of the library (156 in round 5)


def test_the_lds_return_rule_sees_a_planted_copy():
    def stream(*lines):
        return H.parse("0000 <probe>:\n" + "".join("\t%s // %08X: 0\n" % (l, 0x1000 + 4 * i) for i, l in enumerate(lines)))["probe"]
    rules = lambda ins: [v.detail[:20] for v in H.check_lds_returns(ins, "probe")[0]]      # noqa: E731
    at = "ds_add_rtn_u64 v[%d:%d], v2, v[4:5]"
    assert rules(stream(at % (10, 11), "v_mov_b64_e32 v[20:21], v[10:11]")) != []                              # copied before any wait
    assert rules(stream(at % (10, 11), "s_waitcnt lgkmcnt(0)", "v_mov_b64_e32 v[20:21], v[10:11]")) == []
    assert rules(stream(at % (10, 11), at % (12, 13), "s_waitcnt lgkmcnt(1)", "v_add_f32_e32 v30, v10, v31")) == []   # the older of two is in
    assert rules(stream(at % (10, 11), at % (12, 13), "s_waitcnt lgkmcnt(1)", "v_add_f32_e32 v30, v12, v31")) != []   # the newer is not
    assert rules(stream(at % (10, 11), "s_cbranch_scc1 1", "s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 v20, v10")) != []   # the branch skips the wait
    assert rules(stream(at % (10, 11), "s_load_dword s4, s[0:1], 0x0", at % (12, 13), "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v20, v10")) != []   # scalar load in flight


def test_the_checker_sees_a_planted_hazard():
    """the rules fire on a three-instruction stream with each hazard planted (the checker itself is not vacuous)"""
    def stream(*lines):
        return H.parse("0000 <k_scan_hist_r2_probe>:\n" + "".join("\t%s // 0: 0\n" % l for l in lines))["k_scan_hist_r2_probe"]
    mf = "v_mfma_i32_16x16x64_i8 v[0:3], v[4:7], v[8:11], v[12:15]"
    rules = lambda ins: sorted({v.rule for v in H.check(ins, "k_scan_hist_r2_probe")[0]})      # noqa: E731
    assert rules(stream(mf, "s_nop 5", "v_min_u32_e32 v20, 0x10001, v1")) == ["R1"]                      # 6 wait states < 8
    assert rules(stream(mf, "s_nop 7", "v_min_u32_e32 v20, 0x10001, v1")) == []
    assert rules(stream(mf, "s_nop 6", "ds_add_u32 v2, v21")) == ["R1"]                                     # DS address from the result
    assert rules(stream(mf, "v_mov_b32_e32 v5, v30")) == ["R2"]                                            # lands on the A operand
    assert rules(stream("v_mov_b32_e32 v12, v30", mf)) == ["R3"]                                           # srcC written in the slot before
    assert rules(stream("v_mov_b32_e32 v12, v30", "s_nop 3", mf)) == []
    assert rules(stream("v_and_b32_e32 v5, 0x1010101, v30", mf)) == ["R3"]                                 # an A operand a VALU wrote in the slot before
    assert rules(stream("v_and_b32_e32 v5, 0x1010101, v30", "s_nop 1", mf)) == []
    sd = "v_or_b32_sdwa v40, v41, v42 dst_sel:BYTE_%d dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0"
    assert rules(stream(sd % 1, sd % 2)) == ["R4"]
    assert rules(stream(sd % 1, "ds_add_u32 v50, v51", sd % 2)) == []
