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
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "instance_stixels_amd", "csrc"), "asm",
                        "KERNELS=is_k_unary_fast.hip is_k_pairwise.hip"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return True


@pytest.mark.parametrize("src,kernel,min_requests", CASES)
def test_scalar_requests_untouched_while_in_flight(isa, src, kernel, min_requests):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_srec.py"), src, kernel],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    n = int(r.stdout.strip().splitlines()[-1].split()[0])
    assert n >= min_requests, r.stdout
