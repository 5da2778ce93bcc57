"""CPU tests of the drop-in boundary: the library loads, exports every symbol include/nnop_hip.h
declares, and validates descriptors the way the reference's host functions do
(src/attention.jl:141-144).  No compute call is made (no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nnop_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nnop_[a-z_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points(pkg):
    names = declared_functions()
    assert set(names) == set(pkg._lib.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/nnop_hip.h but not exported"
    out = subprocess.run(["nm", "-D", "--defined-only", pkg._lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (nnop_\w+)", out))
    # the only exports beyond the public header are the test hooks of csrc/nnop_debug.h
    assert exported - set(pkg._lib.DEBUG_SYMBOLS) == set(declared_functions()), \
        "exported C symbols must be exactly the header's (+ the nnop_debug_* test hooks)"
    assert set(pkg._lib.DEBUG_SYMBOLS) <= exported


def test_release_build_has_no_lab_equipment(pkg):
    """The shipped library is the release build: no ablation / experiment code, and the launchers never call
    getenv (the NNOP_* knobs are parsed once into a table, csrc/tuning.hpp)."""
    assert not pkg._lib.dev_build()
    out = subprocess.run(["nm", "-D", "--undefined-only", pkg._lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" in out          # ... referenced by the one-time parser only; launch paths use tune_get()
    syms = subprocess.run(["nm", "-C", pkg._lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "fa_fwd_split16_kernel" not in syms


def test_debug_hook_round_trip(pkg):
    lib = pkg._lib.load()
    assert lib.nnop_debug_set(99, 1) == -(2 ** 31)
    prev = pkg._lib.debug_set("fwd_nw", 4)
    assert pkg._lib.debug_set("fwd_nw", prev) == 4


def test_header_compiles_as_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "nnop_hip.h"\nint main(void){ nnop_fa_desc d; (void)d; return NNOP_HIP_ABI_VERSION > 0 ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c),
                    "-o", str(tmp_path / "t")], check=True)
    assert subprocess.run([str(tmp_path / "t")]).returncode == 0


def test_abi_version_and_strerror(pkg):
    lib = pkg._lib.load()
    assert lib.nnop_abi_version() == pkg._lib.ABI_VERSION
    assert int(re.search(r"#define NNOP_HIP_ABI_VERSION (\d+)", open(HEADER).read()).group(1)) == pkg._lib.ABI_VERSION
    assert pkg._lib.strerror(0) == "success"
    assert "power-of-2" in pkg._lib.strerror(pkg._lib.NNOP_ERR_EMB_NOT_POW2)
    assert "divisible" in pkg._lib.strerror(pkg._lib.NNOP_ERR_HEADS)
    assert pkg._lib.strerror(-999) == "unknown nnop status"


def _desc(pkg, **kw):
    base = dict(dtype=2, emb=64, ql=128, kl=128, qh=4, kh=4, batch=1, causal=0, emb_k=0, emb_v=0, kl_v=0, kh_v=0)
    base.update(kw)
    return pkg._lib.FaDesc(**base)


@pytest.mark.parametrize("kw,status", [
    (dict(emb_k=32), "NNOP_ERR_EMB_MISMATCH"),
    (dict(kl_v=64), "NNOP_ERR_KV_SHAPE"),
    (dict(kh_v=2), "NNOP_ERR_KV_SHAPE"),
    (dict(emb=48, emb_k=48), "NNOP_ERR_EMB_NOT_POW2"),
    (dict(qh=6, kh=4), "NNOP_ERR_HEADS"),
    (dict(dtype=7), "NNOP_ERR_DTYPE"),
    (dict(emb=1024), "NNOP_ERR_EMB_UNSUPPORTED"),       # powers of two up to 512 run (csrc/fa_generic.hpp outside 16..128)
    (dict(ql=0), "NNOP_ERR_SHAPE"),
    (dict(), "NNOP_ERR_NULL"),                # valid descriptor, NULL tensors
])
def test_descriptor_validation_order_and_codes(pkg, kw, status):
    """Validation happens before anything touches the device, in the reference's order."""
    lib = pkg._lib.load()
    d = _desc(pkg, **kw)
    null = C.c_void_p(0)
    st = lib.nnop_fa_fwd(C.byref(d), null, null, null, null, null, null, null, null, null)
    assert st == getattr(pkg._lib, status)
    st = lib.nnop_fa_bwd(C.byref(d), *([null] * 13), null, 0, null)
    assert st == getattr(pkg._lib, status)
    if status != "NNOP_ERR_NULL":
        assert lib.nnop_fa_bwd_workspace_bytes(C.byref(d)) == 0


def test_workspace_bytes(pkg):
    lib = pkg._lib.load()
    d = _desc(pkg, ql=4096, qh=4, kh=4, batch=4)
    # two fp32 per (padded) query row; 16-bit E = 64 / 128 problems carry the same pair once more as 2 x 8 elements (32 bytes)
    rows = 4 * 4 * 4096
    frag = 32 if (d.dtype != pkg._lib.NNOP_F32 and d.emb in (64, 128)) else 0
    assert lib.nnop_fa_bwd_workspace_bytes(C.byref(d)) == rows * (8 + frag)
    assert lib.nnop_fa_bwd_workspace_bytes(None) == 0
    assert lib.nnop_fa_fwd(None, *([C.c_void_p(0)] * 9)) == pkg._lib.NNOP_ERR_NULL


def test_shared_memory_null_pointer(pkg):
    assert pkg._lib.load().nnop_shared_memory(0, None) == pkg._lib.NNOP_ERR_NULL


def test_debug_hook_is_locked_without_the_environment_switch():
    """nnop_debug_set flips process-wide kernel selection (and, through bwd_stages, which passes nnop_fa_bwd runs): a host that merely
    links the library must not reach it.  A fresh process WITHOUT NNOP_DEBUG_HOOKS gets INT_MIN back; with it, the previous value."""
    code = ("import ctypes as C, sys; lib = C.CDLL(sys.argv[1]); lib.nnop_debug_set.restype = C.c_int; "
            "print(lib.nnop_debug_set(1, 4), lib.nnop_debug_set(1, -1))")
    import __graft_entry__ as ge
    path = ge.load_package()._lib.LIB_PATH
    env = {k: v for k, v in os.environ.items() if k != "NNOP_DEBUG_HOOKS"}
    locked = subprocess.run([os.sys.executable, "-c", code, path], capture_output=True, text=True, env=env)
    assert locked.stdout.split() == [str(-(2 ** 31))] * 2, locked.stdout + locked.stderr
    env["NNOP_DEBUG_HOOKS"] = "1"
    open_ = subprocess.run([os.sys.executable, "-c", code, path], capture_output=True, text=True, env=env)
    assert open_.stdout.split() == ["-1", "4"], open_.stdout + open_.stderr


def test_misaligned_bases_are_refused_before_any_launch(pkg):
    """NNOP_ERR_ALIGN: the MFMA kernels move q, k, v, o, the gradients, the bias and the workspace with 16-byte vector accesses and
    LDS-DMA from the raw base.  Checked behind the descriptor and NULL checks, in front of the launch (so this runs without a GPU:
    the call returns before it touches the device)."""
    lib = pkg._lib.load()
    d = _desc(pkg)
    ok = 0x7f0000001000                                     # fake, aligned, never dereferenced: every call below fails earlier
    vp = C.c_void_p
    for bad_pos in range(6):                                # o, ms, ls, q, k, v
        args = [ok] * 6
        args[bad_pos] = ok + (2 if bad_pos in (1, 2) else 8) - (1 if bad_pos in (1, 2) else 0)   # ms / ls: element (2-byte) alignment
        st = lib.nnop_fa_fwd(C.byref(d), *[vp(a) for a in args], None, None, None)
        assert st == pkg._lib.NNOP_ERR_ALIGN, (bad_pos, st)
    # the plain-HIP kernels (embedding dims outside the tiled set) need element alignment only: 8 bytes off is fine for E = 8 ...
    d8 = _desc(pkg, emb=8)
    st = lib.nnop_fa_bwd(C.byref(d8), vp(ok + 8), vp(ok), vp(ok), None, vp(ok), vp(ok), vp(ok), vp(ok), vp(ok), vp(ok), vp(ok), None, None,
                         vp(ok + 4), C.c_size_t(1 << 30), None)
    assert st == pkg._lib.NNOP_ERR_ALIGN                    # ... but the workspace is always 16-byte aligned
    st = lib.nnop_fa_bwd(C.byref(d), vp(ok), vp(ok), vp(ok + 8), None, vp(ok), vp(ok), vp(ok), vp(ok), vp(ok), vp(ok), vp(ok), None, None,
                         vp(ok), C.c_size_t(1 << 30), None)
    assert st == pkg._lib.NNOP_ERR_ALIGN
    assert "misaligned" in pkg._lib.strerror(pkg._lib.NNOP_ERR_ALIGN)
    assert "1 ... 512" in pkg._lib.strerror(pkg._lib.NNOP_ERR_EMB_UNSUPPORTED)


@pytest.mark.parametrize("B,KH,rep", [(4, 4, 1), (8, 2, 4), (3, 5, 2), (1, 8, 1), (64, 32, 1), (7, 3, 3)])
def test_shard_entry_equals_the_python_sharding(pkg, B, KH, rep):
    """nnop_fa_shards (the host-side entry a Julia caller loops over, one device per rank) against nnop.jl_amd/shard.py: the same
    rectangles for every world size and rank, offsets = the start of the rectangle in each tensor, every unit covered exactly once."""
    shard = pkg.shard
    QH, QL, KL, E = KH * rep, 96, 160, 64
    d = _desc(pkg, batch=B, kh=KH, qh=QH, ql=QL, kl=KL, emb=E)
    for world in (1, 2, 3, 8, B * KH, B * KH + 3):
        seen = []
        for rank in range(world):
            got = pkg._lib.fa_shards(d, world, rank)
            want = shard.rectangles(B, KH, world, rank)
            assert [(s.b0, s.b1, s.kh0, s.kh1) for s in got] == [(r.b0, r.b1, r.kh0, r.kh1) for r in want]
            for s in got:
                assert (s.desc.batch, s.desc.kh, s.desc.qh) == (s.b1 - s.b0, s.kh1 - s.kh0, (s.kh1 - s.kh0) * rep)
                assert (s.desc.ql, s.desc.kl, s.desc.emb, s.desc.dtype, s.desc.causal) == (QL, KL, E, d.dtype, d.causal)
                qhead = s.b0 * QH + s.kh0 * rep
                assert s.q_off == qhead * QL * E and s.kv_off == (s.b0 * KH + s.kh0) * KL * E
                assert s.row_off == qhead * QL and s.mask_off == s.b0 * KL
                whole = s.kh0 == 0 and s.kh1 == KH
                assert s.pair_off == (s.b0 * KL * QL * QH if whole else -1)
                seen += [(b, h) for b in range(s.b0, s.b1) for h in range(s.kh0, s.kh1)]
        assert sorted(seen) == [(b, h) for b in range(B) for h in range(KH)]
    assert pkg._lib.load().nnop_fa_shards(C.byref(d), 2, 2, (pkg._lib.FaShard * 3)()) == pkg._lib.NNOP_ERR_SHAPE


def test_launcher_form_rules_without_a_device(pkg):
    """the launcher's choice of forward kernel is a plain function of the descriptor (csrc/fa_fwd_inst.hpp fwd_form_of; with no device the
    rules assume 256 CUs): the two-waves-per-SIMD form from KL = 1024 on grids that fill the chip, its 32-row-wave loops (E = 64 and
    E = 128) on launches that would leave CUs idle with 256-row blocks, the one-wave form at E = 128 otherwise"""
    mk = lambda **kw: pkg._lib.FaDesc(**dict(dict(dtype=2, emb=64, ql=4096, kl=4096, qh=4, kh=4, batch=4, causal=0, emb_k=0, emb_v=0, kl_v=0, kh_v=0), **kw))
    f = pkg._lib.fwd_form
    assert f(mk()) == "fa_fwd_duo_kernel"                                                  # C2
    assert f(mk(emb=128, ql=8192, kl=8192, qh=32, kh=32, batch=8, causal=1)) == "fa_fwd_w64_kernel"      # C3
    assert f(mk(ql=512, kl=512)) == "fa_fwd_duo_kernel" and f(mk(ql=512, kl=512, batch=64)) != "fa_fwd_duo_kernel"
    assert f(mk(emb=128, ql=2048, kl=2048, causal=1)) == "fa_fwd_duo_kernel"               # 256 blocks of 128 rows: one round
    assert f(mk(emb=128, ql=2048, kl=2048, batch=16)) == "fa_fwd_w64_kernel"
    assert f(mk(dtype=0)) not in ("fa_fwd_duo_kernel", "fa_fwd_w64_kernel")                # fp32: the 32-row tiled kernels
    assert f(mk(), True, False) != "fa_fwd_duo_kernel"                                     # pair bias: 32-row kernel
