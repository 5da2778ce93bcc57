"""CPU test of GENERATED code: the 64-row forward (csrc/fa_fwd_w64.hpp) issues every MFMA from inline asm, so hipcc neither
pads wait states behind them nor knows that a register-allocator copy placed behind one reads an accumulator tile whose last
pass has not landed.  tools/audit_w64.py models the instruction stream (control flow included) and reports such reads, spills,
scratch and stray M0 uses; this test runs it on the shipped sources -- and on a build with the source-level fences compiled
out, which it must flag (that is the bug the fences fix: stale accumulator registers 13..15 of one O tile at E = 64)."""
import os

import pytest
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUDIT = os.path.join(ROOT, "tools", "audit_w64.py")


def test_shipped_w64_kernels_pass_the_audit():
    r = subprocess.run([sys.executable, AUDIT, "--only", "fwd"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(": OK") == 20         # {bf16, f16} x ({E64, E128} x {plain, masked} x {scale folded into Q, exact} + E256 x {plain, masked})


@pytest.fixture(scope="module", autouse=True)
def compiled_once():
    """the five device-assembly compiles this file's tests look at (a minute each), side by side, into the cache of tools/audit_w64.py"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from audit_w64 import prewarm
    prewarm([("fa_fwd_bf16.hip", []), ("fa_fwd_f16.hip", []), ("fa_bwd_bf16.hip", []), ("fa_bwd_f16.hip", []),
             ("fa_fwd_bf16.hip", ["-DNNOP_W64_NO_LEAVE_FENCE=1"]), ("fa_fwd_f16.hip", ["-DNNOP_W64_NO_LEAVE_FENCE=1"])])


def test_shipped_backward_w64_kernels_pass_the_audit():
    """csrc/fa_bwd_w64.hpp: the same rules for the one-wave-per-SIMD backward -- in particular `VALU-written register -> operand of
    an asm MFMA` (its first version fed row constants to the MFMAs as a C operand the compiler had copied with v_mov right in front
    of them: the tiles of the first key block came out wrong, found on the GPU; this model flags that placement)."""
    r = subprocess.run([sys.executable, AUDIT, "--only", "bwd"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    # {bf16, f16} x {dK/dV, dQ} x {plain, masked} x ({E64, E128, E256} + the narrow shape at {E64, E128})
    assert r.stdout.count(": OK") == 40


def test_audit_flags_the_build_without_leave_fences():
    r = subprocess.run([sys.executable, AUDIT, "--only", "fwd", "--flags", "-DNNOP_W64_NO_LEAVE_FENCE=1"], capture_output=True, text=True)
    assert r.returncode == 1 and "before the MFMA writing it is done" in r.stdout, r.stdout + r.stderr


def test_schedule_of_the_generated_loop_stays_balanced():
    """The slots of the hand-placed loop are dealt out by issue cost (W64Plan).  tools/w64_gaps.py prices the MFMA-to-MFMA gaps
    of the GENERATED code with the calibrated costs (profiles/r02/gapcost.log); a change that lets hipcc pile work into a few
    gaps, or adds instructions to the loop, shows up as predicted cycles per MFMA (45.2 at E = 64, 42.1 at E = 128 with an LDS-DMA
    piece priced at its calibrated 40 cycles -- round 2 priced it as a plain VALU instruction and printed 43.1 / 40.0; the measured
    46.2 / 42.5 track the new numbers)."""
    import re
    gaps = os.path.join(ROOT, "tools", "w64_gaps.py")
    r = subprocess.run([sys.executable, gaps, "Li0ELb1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    per = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"Li(\d+)ELi0ELb1\w*: .*? ([\d.]+) per MFMA", r.stdout)}
    assert set(per) == {64, 128}, r.stdout
    assert per[64] <= 46.5 and per[128] <= 43.0, per
    # the backward kernels (csrc/fa_bwd_w64.hpp; round 3, measured 49.8 / 50.6 cycles per algorithmic MFMA at E = 64, 38.2 in the dQ pass at E = 128): one compile
    r = subprocess.run([sys.executable, gaps, "IDF16b", "--bwd"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = {(int(m.group(1)), int(m.group(2)), int(m.group(3))): float(m.group(4))
           for m in re.finditer(r"Li(\d+)ELi(\d)ELi0ELi(\d)E\w*: .*? ([\d.]+) per MFMA", r.stdout)}
    # (E, kind: 0 dK/dV, 1 dQ, narrow shape), plain mode.  The narrow shape (32 stationary rows per wave: one MFMA per fragment read) pays
    # 20-26 % more issue per MFMA -- it is launched only where it doubles the number of busy SIMDs.
    lim = {(64, 0, 0): 45.0, (64, 1, 0): 49.5, (128, 0, 0): 40.0, (128, 1, 0): 39.5, (256, 0, 0): 45.0, (256, 1, 0): 44.0,
           (64, 0, 1): 53.5, (64, 1, 1): 59.5, (128, 0, 1): 44.5, (128, 1, 1): 47.0}
    assert set(got) == set(lim), r.stdout
    assert all(got[k] <= lim[k] for k in lim), got


def test_scratch_only_where_it_is_known_and_never_in_the_e256_kernels():
    """Register budget of the one-wave-per-SIMD kernels: no scratch anywhere (16-bit E = 256 in particular: spill-free on both
    passes), except the persistent masked dQ kernel of E = 128, which parks a few prologue values (early-requested fragment
    loads) around -- never inside, tools/audit_w64.py rule 1 -- the hand-placed loop, once per block."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from audit_w64 import compile_asm, kernels
    seen = 0
    for src, pat in (("fa_fwd_bf16.hip", "fa_fwd_w64_kernel"), ("fa_bwd_bf16.hip", "fa_bwd_w64_kernel")):
        for name, body, meta in kernels(compile_asm(src, []), pat):
            scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1))
            spills = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1))
            parked = "fa_bwd_w64_kernelIDF16bLi128ELi1ELi1E" in name                                           # E = 128, dQ pass, masked mode
            if parked:
                assert scratch <= 128 and spills <= 24, (name, scratch, spills)
            else:
                assert scratch == 0 and spills == 0, (name, scratch, spills)
            seen += 1
    assert seen == 10 + 12 + 8
