"""Mechanical guard for the built gfx950 code (runs in the CPU container: hipcc cross-compiles, the
checks read the code objects with llvm-readelf / llvm-objdump -- tools/isa_guard.py).

The inline-asm GEMMs rely on compiler behaviour that bit three times in round 2 (a re-pack of in-flight fragment
registers, a VALU->SGPR->VMEM hazard hipcc does not pad for an asm consumer, a flat_load where a global_load was
written); each has a numerics regression test on the GPU, this file checks the ISA and the resource usage themselves, so a
hipcc bump that re-opens one of them fails HERE, before any GPU is involved."""
import os
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_guard as G  # noqa: E402

LDS_BYTES_PER_CU = 160 * 1024


@pytest.fixture(scope="module")
def built():
    """{object name: (metadata, code)} of the library's objects (rebuilt first if a source is newer)"""
    from npi_gnn_amd.build import HERE, HOST_ONLY, SOURCES, build_library
    build_library()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for src in SOURCES:
            if src in HOST_ONLY:                              # no device code: nothing to guard
                continue
            out[src] = G.analyse(os.path.join(HERE, "build", src.replace(".hip", ".o")), tmp)
    return out


def _pretty(meta):
    names = list(meta)
    return dict(zip(G.demangle(names), (meta[n] for n in names)))


def test_no_kernel_uses_scratch_or_spills_vector_registers(built):
    for src, (meta, _) in built.items():
        for name, m in meta.items():
            assert m["scratch"] == 0 and m["vgpr_spill"] == 0 and not m["dynamic_stack"], (src, name, m)
            assert m["lds"] <= LDS_BYTES_PER_CU, (src, name, m)


def test_hot_kernels_keep_the_occupancy_the_design_assumes(built):
    seg = _pretty(built["segsum.hip"][0])
    hot = {n: m for n, m in seg.items() if n.startswith("void npi::segsum_kernel<float, 4, 1, 0, 1>")}
    assert len(hot) == 1, list(seg)[:5]                       # f32 x 4, one chunk, unweighted, unguarded: the C4 kernel
    for n, m in hot.items():
        assert m["vgpr"] <= 64 and m["agpr"] == 0, (n, m)     # 8 waves / SIMD (512 VGPRs / 64)
    # the same kernel when it also writes the finished rows' power-of-two scales (npi_segsum_ex; the fp16 x 2 projection behind
    # it): 7 waves / SIMD, no scratch -- at 8 it spilled three registers
    sc = {n: m for n, m in seg.items() if n.startswith("void npi::segsum_kernel<float, 4, 1, 0, 2>")}
    assert len(sc) == 1
    for n, m in sc.items():
        assert m["vgpr"] <= 72 and m["agpr"] == 0 and m["scratch"] == 0, (n, m)
    gemm = _pretty(built["gemm_f32.hip"][0])
    # (<4, 1>: the same kernel with the rank-2 store epilogue of GATConv's dX -- 22 registers and 8 KB of LDS more; <4, 2>: with
    # the row dots of GATConv's scores in the store epilogue -- 10 KB of LDS more)
    for fam in ("gemm_split_ws_kernel<4, 0, false>", "gemm_split_ws_kernel<4, 1, false>", "gemm_split_ws_kernel<4, 2, false>",
                "gemm_split_ws_kernel<4, 0, true>", "gemm_split_ws_kernel<4, 1, true>", "gemm_split_ws_kernel<4, 2, true>",     # fp16 x 2
                "gemm_dw_split_kernel<4, false, false>", "gemm_dw_split_kernel<4, false, true>",
                "gemm_dw_split_kernel<4, true, false>",
                "gemm_bf16_ws_kernel<4>"):
        ks = {n: m for n, m in gemm.items() if fam in n}
        assert len(ks) == 1, fam
        for n, m in ks.items():
            assert m["vgpr"] + m["agpr"] <= 256, (n, m)       # 2 waves / SIMD: one producer + one consumer wave
            assert m["lds"] <= LDS_BYTES_PER_CU, (n, m)
    # the segsum instantiation grid stays pruned (482 before round 3: an 88 s build)
    assert len(seg) <= 220, len(seg)


def test_asm_gemms_hold_no_flat_access_no_sgpr_hazard_and_touch_no_in_flight_lds_destination(built):
    meta, code = built["gemm_f32.hip"]
    checked = 0
    for sym, instrs in code.items():
        if any(k in sym for k in ("gemm_split_ws_kernel", "gemm_dw_split_kernel", "gemm_bf16_ws_kernel")):
            checked += 1
            assert G.find_flat(instrs) == [], (sym, G.find_flat(instrs)[:3])
            wide = "global_load_dwordx2" if "gemm_dw_split_kernelILi4ELb1" in sym or "gemm_dw_split_kernelILi2ELb1" in sym else "global_load_dwordx4"
            assert any(i.startswith(wide) for i in instrs), sym                     # the asm loads are there at all (bf16 dW: 8 bytes)
            f16 = ("gemm_split_ws_kernel" in sym and sym.endswith("Lb1EEEvNS_9SplitArgsE")) or \
                  ("gemm_dw_split_kernel" in sym and sym.endswith("Lb0ELb1EEEvNS_6DwArgsE"))        # <TN, BF16IN = false, F16 = true>
            mfma = "v_mfma_f32_32x32x16_f16" if f16 else "v_mfma_f32_32x32x16_bf16"
            assert any(i.startswith("ds_read_b128") for i in instrs) and any(mfma in i for i in instrs), (sym, mfma)
    assert checked == 20                                      # <4> and <2> of each family, the split kernel also with the rank-2 and the row-dot epilogue and each of those on fp16 x 2, dW also for bf16 operands and on fp16 x 2
    # no FLAT memory instruction anywhere: a flat access counts on lgkmcnt as well as vmcnt (every LDS / scalar-load wait then
    # drains the gathers too) -- round 4 found all 128 aggregation kernels gathering through flat_load because the second
    # part of the table was addressed through a pointer biased with integer arithmetic
    for src, (_, code) in built.items():
        for sym, instrs in code.items():
            assert G.find_flat(instrs) == [], (src, sym, G.find_flat(instrs)[:3])
    for src, (_, code) in built.items():
        for sym, instrs in code.items():
            hz = G.find_sgpr_hazards(instrs)
            assert hz == [], (src, sym, hz[:3])
            touch = G.find_inflight_touch_linear(instrs)
            assert touch == [], (src, sym, touch[:3])


def _code(lines):
    c = G.Code()
    for k, l in enumerate(lines):
        c.append(l)
        c.addr.append(4 * k)
    return c


def test_the_checks_catch_what_they_are_for():
    """The analysers on hand-written snippets of the three bug patterns (and of their fixed forms)."""
    # 1. re-pack of a fragment between the LDS read and its wait (round 2: dW 3 % off under concurrent load)
    bad = _code(["ds_read_b128 v[10:13], v2", "v_lshrrev_b32_e32 v20, 16, v10", "v_perm_b32 v10, v20, v11, s4",
                 "s_waitcnt lgkmcnt(0)", "v_mfma_f32_32x32x16_bf16 v[30:45], v[10:13], v[14:17], v[30:45]"])
    hits = G.find_inflight_touch_linear(bad)
    assert [h[0] for h in hits] == [1, 2] and G.find_inflight_touch(bad)
    good = _code(["ds_read_b128 v[10:13], v2", "s_waitcnt lgkmcnt(0)", "v_perm_b32 v10, v20, v11, s4"])
    assert G.find_inflight_touch_linear(good) == [] and G.find_inflight_touch(good) == []
    # counted waits: LDS returns in order -- lgkmcnt(1) retires the older of two reads, not the younger
    two = _code(["ds_read_b128 v[10:13], v2", "ds_read_b128 v[14:17], v2 offset:64", "s_waitcnt lgkmcnt(1)",
                 "v_mov_b32_e32 v40, v10", "v_mov_b32_e32 v41, v14"])
    assert [h[0] for h in G.find_inflight_touch_linear(two)] == [4]
    assert [h[0] for h in G.find_inflight_touch(two)] == [4]
    # a scalar load in flight returns out of order: only lgkmcnt(0) retires the LDS read then
    mixed = _code(["ds_read_b32 v5, v2", "s_load_dword s8, s[0:1], 0x0", "s_waitcnt lgkmcnt(1)", "v_add_u32_e32 v6, v5, v5"])
    assert len(G.find_inflight_touch_linear(mixed)) == 1 and len(G.find_inflight_touch(mixed)) == 1
    # across a loop: the read at the bottom is consumed at the top of the next iteration without a wait (CFG analysis)
    loop = _code(["v_add_u32_e32 v6, v5, v5", "ds_read_b32 v5, v2", "s_cmp_lg_u32 s4, s5", "s_cbranch_scc1 65532", "s_endpgm"])
    assert [h[0] for h in G.find_inflight_touch(loop)] == [0]
    # 2. VALU-written SGPR as the scalar base of a VMEM access within 5 wait states (the first split-dW build faulted)
    hz = _code(["v_readfirstlane_b32 s4, v1", "v_readfirstlane_b32 s5, v2", "s_nop 1", "global_load_dwordx4 v[4:7], v3, s[4:5]"])
    assert [h[0] for h in G.find_sgpr_hazards(hz)] == [0, 1]
    ok = _code(["v_readfirstlane_b32 s4, v1", "v_readfirstlane_b32 s5, v2", "s_nop 4", "global_load_dwordx4 v[4:7], v3, s[4:5]"])
    assert G.find_sgpr_hazards(ok) == []
    scalar = _code(["s_add_u32 s4, s6, s8", "s_addc_u32 s5, s7, 0", "global_load_dwordx4 v[4:7], v3, s[4:5]"])
    assert G.find_sgpr_hazards(scalar) == []                  # scalar arithmetic: no hazard (what the fixed kernel does)
    # 3. a base pointer that lost its address space
    assert len(G.find_flat(_code(["flat_load_dwordx4 v[4:7], v[2:3]", "global_load_dwordx4 v[4:7], v3, s[4:5]"]))) == 1
