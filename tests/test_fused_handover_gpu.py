"""GPU tests of the hand-over inside the fused LUT + DP launch of the unary DP (is_k_unary_fast.hip, LUTF) under the
conditions its assumptions do not cover by themselves: another PROCESS on the card, a stream with a reduced CU
mask, and the policy after a repair.  The reference gets the order of its table kernel and its DP kernel from the
stream (Stixels.cu:535-590); the fused launch checks what it relies on per call and repairs the call otherwise --
so every case asserts the same bits as the two ordinary launches (IS_LUT_FUSED=0) and records the repair count."""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(seed=59, frames=8):
    case2 = helpers.build_case("drn_d_22_unary", 1024, 2048, 128, seed=seed, n_images=2)
    return helpers.sub_case(case2, [i % 2 for i in range(frames)])


def _core(case):
    from instance_stixels_amd.core import Core
    return Core(case["params"], case["lut"], case["odr"], max_batch=len(case["frames"]))


def _run(core, case, want_tables=True):
    cfg = case["cfg"]
    return core.run(disparity_big=case["disparity"], segmentation=case["segmentation"], ground_function=case["gf"],
                    normalization_ground=case["ng"], inv_sigma2_ground=case["ig"], vhor=case["vhor"],
                    pairwise=False, median_join=bool(cfg.median_join), want_tables=want_tables, want_instances=False)


def _same(a, b):
    assert np.array_equal(a["cost_table"].view(np.uint32), b["cost_table"].view(np.uint32))
    assert np.array_equal(a["index_table"], b["index_table"])
    for img in range(len(a["sections"])):
        assert helpers.sections_equal(a["sections"][img], b["sections"][img])


def _reference_launches(case, monkeypatch):
    monkeypatch.setenv("IS_LUT_FUSED", "0")
    core = _core(case)
    try:
        return _run(core, case)
    finally:
        core.close()
        monkeypatch.delenv("IS_LUT_FUSED")


def test_fused_handover_with_a_second_process_on_the_gpu(monkeypatch, tmp_path):
    """The default (fused) launch of an 8-frame unary call, ten times, while a second process with its own context
    keeps the card busy with pairwise batches: the same bits as the ordinary launches every time; the number of
    repaired calls is printed (0 expected: the dispatch order inside ONE launch does not depend on other queues)."""
    monkeypatch.delenv("IS_LUT_FUSED", raising=False)
    case = _case()
    ref = _reference_launches(case, monkeypatch)
    flag = tmp_path / "busy.flag"
    flag.write_text("x")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "fused_busy_child.py"), str(flag), "120"],
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=ROOT)
    try:
        seen = []
        for _ in range(50):     # (the runtime may print warnings first)
            seen.append(child.stdout.readline())
            if "BUSY_READY" in seen[-1] or not seen[-1]:
                break
        assert "BUSY_READY" in seen[-1], "".join(seen)
        core = _core(case)
        core.set_eval_counters(True)
        try:
            for it in range(10):
                got = _run(core, case)
                _same(got, ref)
            counters = core.eval_counters()
            repairs = core.lut_fused_repairs()
        finally:
            core.close()
        assert child.poll() is None, "the busy process ended before the fused calls did"
    finally:
        flag.unlink()
        tail = child.communicate(timeout=180)[0]
    assert "BUSY_DONE" in tail, tail
    assert counters["lutf_unit_cycles"] > 0            # the fused form did run
    print("second process busy: repaired calls", repairs, "of 10; polls", counters["lutf_spins"], tail.strip())
    assert repairs <= 1    # (a first distrust turns the fused launch off for the context: never more than one)
    ref1 = helpers.run_oracle(case, image=1)
    errs = helpers.compare(ref1, got, 1, case["cfg"])
    assert not errs, "\n".join(errs[:10])


def _masked_stream(words):
    """hipExtStreamCreateWithCUMask: a stream whose queues may only use the CUs of the mask."""
    hip = ctypes.CDLL("libamdhip64.so")
    stream = ctypes.c_void_p()
    mask = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), ctypes.c_uint32(len(words)), mask)
    return hip, stream, rc


@pytest.mark.parametrize("name,words", [("lower_half", [0xFFFFFFFF] * 4 + [0] * 4),
                                        ("every_other_cu", [0x55555555] * 8),
                                        ("one_word", [0xFFFFFFFF] + [0] * 7),
                                        ("ragged", [0x0000FFFF, 0xFFFFFFFF, 0x3, 0, 0xF0F0F0F0, 0, 0, 0x1])])
def test_fused_handover_on_a_stream_with_a_reduced_cu_mask(name, words, monkeypatch):
    """The fused launch on streams that may use only part of the chip (32 ... 128 of the 256 CUs, among them masks
    that leave the XCDs unequal shares): whatever the dispatcher does with the block order there, the output has the
    bits of the ordinary launches; the repair count and the polls are recorded.  A context that had to repair keeps
    the table in the prepare launch afterwards (second call: no unit runs)."""
    import torch
    monkeypatch.delenv("IS_LUT_FUSED", raising=False)
    case = _case(seed=61)
    ref = _reference_launches(case, monkeypatch)
    hip, stream, rc = _masked_stream(words)
    if rc != 0:
        pytest.skip(f"hipExtStreamCreateWithCUMask is not available here (rc {rc})")
    try:
        ext = torch.cuda.ExternalStream(stream.value)
        core = _core(case)
        core.set_eval_counters(True)
        try:
            with torch.cuda.stream(ext):
                got = _run(core, case)
                c1 = core.eval_counters()
                r1 = core.lut_fused_repairs()
                _same(got, ref)
                got2 = _run(core, case)
                c2 = core.eval_counters()
                r2 = core.lut_fused_repairs()
                _same(got2, ref)
        finally:
            core.close()
    finally:
        hip.hipStreamDestroy(stream)
    print(f"CU mask {name}: repaired calls {r1} then {r2}; unit cycles {c1['lutf_unit_cycles']} then "
          f"{c2['lutf_unit_cycles']}; polls {c1['lutf_spins']}")
    assert c1["lutf_unit_cycles"] > 0
    if r1:   # distrusted once -> off for the rest of the context's life
        assert r2 == r1 and c2["lutf_unit_cycles"] == c1["lutf_unit_cycles"]


def test_context_keeps_the_table_in_the_prepare_launch_after_a_repair(monkeypatch):
    """IS_LUT_FUSED=3 = the default policy with a wrong XCC id published by the units: the first 8-frame call is
    repaired (same bits), is_lut_fused_repairs counts it, and the calls after it do not run the fused launch any
    more -- a repaired call costs 2.8 x an ordinary one, so a context whose dispatcher does not behave as observed
    pays that once.  IS_LUT_FUSED=2 (explicit) keeps repairing: the count goes up with every call."""
    case = _case(seed=63)
    ref = _reference_launches(case, monkeypatch)
    monkeypatch.setenv("IS_LUT_FUSED", "3")
    core = _core(case)
    core.set_eval_counters(True)
    try:
        _same(_run(core, case), ref)
        c1, r1, last1 = core.eval_counters(), core.lut_fused_repairs(), core.lut_fused_repaired()
        _same(_run(core, case), ref)
        c2, r2, last2 = core.eval_counters(), core.lut_fused_repairs(), core.lut_fused_repaired()
    finally:
        core.close()
    assert (r1, last1) == (1, 1) and c1["lutf_unit_cycles"] > 0
    assert (r2, last2) == (1, 0) and c2["lutf_unit_cycles"] == c1["lutf_unit_cycles"]
    monkeypatch.setenv("IS_LUT_FUSED", "2")
    core = _core(case)
    try:
        for k in range(3):
            _same(_run(core, case), ref)
            assert core.lut_fused_repairs() == k + 1
    finally:
        core.close()


def test_distrusting_workgroups_do_not_wait_out_their_polls(monkeypatch):
    """The poll of a DP workgroup ends as soon as ANY workgroup has distrusted the hand-over (it reads the word in its
    loop) and is bounded to tens of milliseconds by itself: a forced repair of 64 frames -- 262144 DP workgroups --
    finishes in well under a second."""
    import torch
    monkeypatch.setenv("IS_LUT_FUSED", "2")
    case = _case(seed=65, frames=64)
    core = _core(case)
    try:
        _run(core, case, want_tables=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _run(core, case, want_tables=False)
        dt = time.perf_counter() - t0
        assert core.lut_fused_repaired() == 1
    finally:
        core.close()
    print(f"forced repair of 64 frames incl. H2D / D2H of the test harness: {dt * 1e3:.1f} ms")
    assert dt < 1.5
