"""CPU tests of the two-waves-per-SIMD forward (csrc/fa_fwd_duo.hpp): its phase loop is GENERATED inline-asm text with fixed physical
registers (tools/gen_duo_asm.py -> csrc/fa_fwd_duo_asm.inc).  hipcc neither pads wait states nor tracks LDS requests inside an asm
statement, so the generator audits its own streams (check_stream: fragment reads waited for, transcendental / permlane / M0 wait
states, the distance between the last QK^T MFMAs and the first vector read of their score registers); this test runs that audit,
pins the committed include to the generator, and checks on the compiled kernels that hipcc allocates nothing it has to spill
around the loop in plain mode."""
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "nnop.jl_amd", "csrc")


def test_committed_streams_are_the_generators_output():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_duo_asm.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, "csrc/fa_fwd_duo_asm.inc is stale: run tools/gen_duo_asm.py\n" + r.stdout + r.stderr


def test_generated_streams_pass_the_static_audit():
    import gen_duo_asm as g
    for masked in (False, True):
        for prof in (False, True):
            assert g.check_stream(g.loop(masked, prof))


def test_audit_flags_a_missing_wait_and_a_short_distance():
    import gen_duo_asm as g
    lines = g.loop(False)
    # drop the first counted LDS wait in front of an MFMA: the MFMA then reads a fragment that may not have landed
    i = next(k for k, ln in enumerate(lines) if ln.startswith("s_waitcnt lgkmcnt"))
    with pytest.raises(AssertionError, match="not waited for"):
        g.check_stream(lines[:i] + lines[i + 1:])
    # a vector read of the key-block-1 logits right behind the MFMAs that write them: they may not have left the matrix pipe yet
    kb1 = g.vr(g.S(1, 1), 16)
    j = max(k for k, ln in enumerate(lines[:len(lines) // 2]) if ln.startswith("v_mfma_f32_32x32x16") and ln.split()[1].startswith(kb1))
    bad = lines[:j + 1] + [f"v_max_f32 v212, {g.vr(g.S(0, 1))}, {g.vr(g.S(0, 1) + 1)}"] + lines[j + 1:]
    with pytest.raises(AssertionError, match="key block 1"):
        g.check_stream(bad)


def test_phase_structure_of_the_loop():
    """the matrix phase holds nothing but MFMAs, fragment reads, their waits and (at its end) the LDS-DMA batch: a VALU instruction
    between MFMAs stalls the wave's in-order issue while its SIMD partner's vector phase holds the port (measured: profiles/r04)"""
    import gen_duo_asm as g
    both = g.m_phase(True, True, False)
    ops = [ln.split()[0] for ln in both if not ln.startswith("@")]
    first_mfma = next(i for i, o in enumerate(ops) if o.startswith("v_mfma"))
    last_mfma = max(i for i, o in enumerate(ops) if o.startswith("v_mfma"))
    inside = set(ops[first_mfma:last_mfma + 1])
    assert inside <= {"v_mfma_f32_16x16x32_@T@", "v_mfma_f32_32x32x16_@T@", "ds_read_b128", "ds_read_b64_tr_b16", "s_waitcnt"}, inside
    assert sum(o.startswith("v_mfma_f32_32x32x16") for o in ops) == 32 and sum(o.startswith("v_mfma_f32_16x16x32") for o in ops) == 8
    assert sum(o.startswith("buffer_load_dwordx4") for o in ops) == 4


def test_compiled_kernels_allocate_within_the_loops_register_map():
    from audit_w64 import compile_asm                        # (the library's flags; shares its cache with tests/test_w64_codegen.py)
    text = compile_asm("fa_fwd_bf16.hip", [])
    got = {}
    for m in re.finditer(r"\.set _ZN4nnop17fa_fwd_duo_kernelIDF16bLi64ELi(\d)ELi(\d)EEEvNS_9FwdParamsE\.(num_vgpr|num_agpr|private_seg_size), (\d+)", text):
        got[(int(m.group(1)), int(m.group(2)), m.group(3))] = int(m.group(4))
    for m in re.finditer(r"\.set _ZN4nnop17fa_fwd_duo_kernelIDF16bLi128ELi(\d)ELi1EEEvNS_9FwdParamsE\.(num_vgpr|num_agpr|private_seg_size), (\d+)", text):
        got[(int(m.group(1)), 128, m.group(2))] = int(m.group(3))
    for m in re.finditer(r"\.set _ZN4nnop17fa_fwd_duo_kernelIDF16bLi32ELi(\d)ELi2EEEvNS_9FwdParamsE\.(num_vgpr|num_agpr|private_seg_size), (\d+)", text):
        got[(int(m.group(1)), 32, m.group(2))] = int(m.group(3))
    for nz in (1, 2, 128, 32):                                # rows per wave / 32 at E = 64; the E = 128 loop; the E = 32 loop
        assert got[(0, nz, "num_vgpr")] <= 256 and got[(1, nz, "num_vgpr")] <= 256 and got[(0, nz, "num_agpr")] == 0 and got[(1, nz, "num_agpr")] == 0, got
        # plain mode: nothing spilled; masked mode (persistent block loop around the statement): a few prologue / epilogue values park in
        # scratch around -- never inside -- the loop, once per block
        assert got[(0, nz, "private_seg_size")] == 0 and got[(1, nz, "private_seg_size")] <= 64, got
