"""The build's guard for hazards inside hand-written assembly (mistra_amd/build.py: isa_hazard_report): the compiler's hazard recogniser
and waitcnt insertion do not look inside asm statements, and the kernel carries ~4 000 lines of them.  Three classes — a scalar base
written by v_readfirstlane feeding a vector-memory instruction too early (the round-3 GPU memory fault), a DPP read of a register the
previous VALU instruction wrote, a vmcnt wait that cannot cover the function's look-ahead ring — are refused at link time.  Here: the
scanner on crafted ISA (each class caught, the legal forms passed), and on the product kernel's real ISA, clean as built and caught
again once the hand-placed wait states are taken out of the text."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import REPO
from mistra_amd import build

CRAFTED = """
_ZN6mistra4bad1Ev:
	v_mov_b32_e32 v1, v2
	;;#ASMSTART
	v_readfirstlane_b32 s0, v0
	v_readfirstlane_b32 s1, v1
	s_nop 2
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[64:67], v10, s[0:1] offset:0
	;;#ASMEND
	s_endpgm
_ZN6mistra4bad2Ev:
	;;#ASMSTART
	v_fmac_f64_e32 v[2:3], v[4:5], v[6:7]
	v_fmac_f64_dpp v[8:9], -v[2:3], v[6:7] row_newbcast:3 row_mask:0x1 bank_mask:0xf
	;;#ASMEND
_ZN6mistra4bad3Ev:
	;;#ASMSTART
	global_load_dwordx4 v[64:67], v10, s[0:1] offset:0
	global_load_dwordx4 v[68:71], v10, s[0:1] offset:16
	s_waitcnt vmcnt(2)
	;;#ASMEND
_ZN6mistra4goodEv:
	;;#ASMSTART
	v_readfirstlane_b32 s0, v0
	v_readfirstlane_b32 s1, v1
	s_nop 4
	global_load_dwordx4 v[64:67], v10, s[0:1] offset:0
	global_load_dwordx4 v[68:71], v10, s[0:1] offset:16
	s_waitcnt vmcnt(1)
	v_fmac_f64_e32 v[2:3], v[4:5], v[6:7]
	s_nop 1
	v_fmac_f64_dpp v[2:3], -v[2:3], v[6:7] row_newbcast:3 row_mask:0x1 bank_mask:0xf
	v_fmac_f64_dpp v[2:3], -v[8:9], v[6:7] row_newbcast:4 row_mask:0x1 bank_mask:0xf
	;;#ASMEND
"""


def test_each_hazard_class_is_caught_on_crafted_isa():
    hits = build.isa_hazard_report(CRAFTED)
    assert len(hits) == 3, hits
    assert "bad1" in hits[0] and "scalar base" in hits[0] and "3 wait state(s)" in hits[0]
    assert "bad2" in hits[1] and "DPP-reads" in hits[1]
    assert "bad3" in hits[2] and "vmcnt(2)" in hits[2]
    assert not any("good" in h for h in hits)


@pytest.fixture(scope="module")
def product_isa(tmp_path_factory):
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc here")
    out = tmp_path_factory.mktemp("isa") / "ros3_kernel.s"
    cmd = [build.hipcc(), "--offload-arch=" + build.ARCH] + [f for f in build.COMMON if f != "-fPIC"] + \
          ["-S", "--offload-device-only", os.path.join(build.CSRC, "ros3_kernel.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out.read_text()


def test_product_kernel_isa_is_clean_and_the_guard_sees_it(product_isa):
    seen = {}
    assert build.isa_hazard_report(product_isa, seen=seen) == []
    # the scanner is not blind: it examined the ring loads, the DPP chains and the counted waits of every asm-carrying function
    assert seen["vmem_scalar_base"] > 300 and seen["dpp"] > 500 and seen["vmcnt"] > 200 and seen["functions"] >= 20, seen


def test_guard_fires_when_the_hand_placed_wait_states_are_removed(product_isa):
    # (i) the `s_nop 4` behind ring_base's v_readfirstlane pair (ros3_kernel.hip: ring_base; vm_exec_asm.inc / gsum_exec_asm.inc head)
    no_nop4 = re.sub(r"(v_readfirstlane_b32 s\d+, v\d+\n)\ts_nop 4\n", r"\1", product_isa)
    assert no_nop4 != product_isa
    hits = build.isa_hazard_report(no_nop4)
    assert any("scalar base" in h for h in hits), hits[:3]
    # (ii) the `s_nop 1` in front of the tail chain's dependent DPP steps
    no_nop1 = product_isa.replace("\ts_nop 1\n\tv_fmac_f64_dpp", "\tv_fmac_f64_dpp")
    assert no_nop1 != product_isa
    hits = build.isa_hazard_report(no_nop1)
    assert sum("DPP-reads" in h for h in hits) > 50
    # (iii) a counted wait that no longer covers the ring
    deep = product_isa.replace("s_waitcnt vmcnt(7)", "s_waitcnt vmcnt(9)", 1)
    assert any("vmcnt(9)" in h for h in build.isa_hazard_report(deep))
