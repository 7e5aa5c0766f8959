"""Compile-time check of the one place where the kernels rely on something the compiler cannot see:
the scalar half-record requests of eval_segment_mix (is_kernels.h: srec_request / srec_arrived) are
inline-assembly s_load_dwordx16 whose destination SGPRs are in flight until the inline-assembly
s_waitcnt that follows.  tools/check_srec.py walks every path of the generated gfx950 ISA between
the two and fails on any instruction that reads or writes those registers (a spill, a copy on a
loop back edge, a reuse after a loop exit)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [("is_k_unary_fast", "k_dp_unary_fastILb%dELi%dELb%dE" % (inv, nvr, pre), 2)
         for inv in (0, 1) for nvr in (2, 4) for pre in (0, 1)] + \
        [("is_k_pairwise", "k_pw_phase1ILb0ELi2", 4), ("is_k_pairwise", "k_pw_phase1ILb0ELi0", 4),
         ("is_k_pairwise", "k_pw_phase1ILb1ELi2", 0), ("is_k_pairwise", "k_pw_phase1ILb1ELi0", 0)]


@pytest.fixture(scope="module")
def isa():
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    # the default target compiles every source with -save-temps and runs the check on the ISA of the
    # objects it links (the library is not linked on a violation); the cases below look at the same
    # files kernel by kernel and make sure every kernel that should hold requests does
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "instance_stixels_amd", "csrc"), "-j8"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return os.path.join(ROOT, "instance_stixels_amd", "csrc", "build")


def test_the_build_runs_the_check_and_fails_closed(isa, tmp_path):
    """check_srec.py on a doctored listing: a touched destination register, an indirect jump and
    a branch to an unknown label between request and wait are violations; no request at all in
    --all mode fails too (a build that silently lost its requests must not pass)."""
    good = """_Z4testv:
	;;#ASMSTART
	s_load_dwordx16 s[16:31], s[2:3], 0x0
	;;#ASMEND
	v_add_f32_e32 v1, v2, v3
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	s_endpgm
.Lfunc_end0:
"""
    tool = os.path.join(ROOT, "tools", "check_srec.py")
    def run(text):
        (tmp_path / "x-hip-amdgcn-amd-amdhsa-gfx950.s").write_text(text)
        return subprocess.run([sys.executable, tool, "--dir", str(tmp_path), "--all", "x"],
                              capture_output=True, text=True)
    assert run(good).returncode == 0
    assert run(good.replace("v_add_f32_e32 v1, v2, v3", "s_mov_b32 s20, 0")).returncode == 1
    assert run(good.replace("v_add_f32_e32 v1, v2, v3", "s_setpc_b64 s[0:1]")).returncode == 1
    assert run(good.replace("v_add_f32_e32 v1, v2, v3", "s_cbranch_scc1 .LBB9_99")).returncode == 1
    assert run("_Z4testv:\n\ts_endpgm\n.Lfunc_end0:\n").returncode == 1


@pytest.mark.parametrize("src,kernel,min_requests", CASES)
def test_scalar_requests_untouched_while_in_flight(isa, src, kernel, min_requests):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_srec.py"), "--dir", isa, src, kernel],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    n = int(r.stdout.strip().splitlines()[-1].split()[0])
    assert n >= min_requests, r.stdout
