"""Build-time guard for the kernels that synchronise LDS-DMA with hand-counted `s_waitcnt vmcnt(N)` across a raw
`s_barrier` (csrc/narrow_fwd.hip, csrc/shared_mlp_x3.hip).  Their waits assume an exact number and ORDER of vector
memory instructions per loop body; if the compiler emitted others (a spill, a hoisted load, a re-ordered DMA) the wait
would be too short and a tile would be read before its DMA landed -- silently.  hipcc cross-compiles gfx950 without a
GPU, so the ISA is checked here, on CPU, every time the suite runs:

  narrow_fwd_kernel<OTW, ...>   no scratch; LDS-DMA instructions come in groups of NDMA = 4 per basic block; the only
                                vmcnt constants are {0, NDMA, NST, NDMA + NST} with NST = 16 * OTW stores per tile
  gemm_x3p_kernel<..., NPL>     no scratch; in the main loop every LDS-DMA precedes every register load of the streamed
                                operand, there are NPL * NA DMAs and NXL loads, and the wait in front of the barrier is
                                vmcnt(NXL) -- "all but the register loads", i.e. the DMA has landed
"""
import glob
import os
import re
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
sys.path.insert(0, ROOT)
from usip_amd.build import FLAGS as BUILD_FLAGS  # noqa: E402   (the ISA checked here is the ISA that ships)

FLAGS = [f for f in BUILD_FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only"]

pytestmark = pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not present")


_ASM_CACHE = {}
_DIAG_CACHE = {}


def _asm(src, tmp_path):
    if src not in _ASM_CACHE:
        out = str(tmp_path / (os.path.basename(src) + ".s"))
        r = subprocess.run([HIPCC] + FLAGS + ["-x", "hip", os.path.join(ROOT, "usip_amd", "csrc", src), "-o", out],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
        _ASM_CACHE[src] = open(out).read()
        _DIAG_CACHE[src] = r.stderr.decode(errors="replace")
    return _ASM_CACHE[src]


def _functions(asm, prefix):
    for m in re.finditer(r"^(%s\w+):" % prefix, asm, re.M):
        body = asm[m.start():asm.index(".Lfunc_end", m.start())]
        ins = [ln.split(";")[0].strip() for ln in body.split("\n")]
        yield m.group(1), [i for i in ins if i and (not i.startswith(".") or i.endswith(":"))]


def _blocks(ins):
    cur = []
    for i in ins:
        if re.match(r"^\.LBB\d+_\d+:$", i) or i.endswith(":"):
            if cur:
                yield cur
            cur = []
        else:
            cur.append(i)
    if cur:
        yield cur


def test_narrow_forward_kernel_vmcnt_assumptions(tmp_path):
    asm = _asm("narrow_fwd.hip", tmp_path)
    seen = 0
    for name, ins in _functions(asm, "_ZN12_GLOBAL__N_117narrow_fwd_kernel"):
        seen += 1
        otw = int(re.search(r"narrow_fwd_kernelILi(\d)E", name).group(1))
        ndma, nst = 4, 16 * otw
        assert not any("scratch_" in i for i in ins), name
        for blk in _blocks(ins):
            d = sum("global_load_lds_dwordx4" in i for i in blk)
            assert d in (0, ndma), (name, d)
        consts = {int(c) for i in ins if i.startswith("s_waitcnt") for c in re.findall(r"vmcnt\((\d+)\)", i)}
        assert consts <= {0, ndma, nst, ndma + nst}, (name, consts)
        assert {ndma, nst, ndma + nst} <= consts, (name, consts)          # the counted waits are really there
    assert seen == 16                                                       # OTW x PRO x STATS x RB


def test_split_gemm_main_loop_order_and_counts(tmp_path):
    asm = _asm("shared_mlp_x3.hip", tmp_path)
    seen = deep = 0
    for name, ins in _functions(asm, "_ZN12_GLOBAL__N_115gemm_x3p_kernel"):
        pro, epi, tm, wn, npl, aslots = (int(v) for v in
                                         re.search(r"kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)E", name).groups())
        assert not any("scratch_" in i for i in ins), name
        # the main loop: the block with the most MFMAs that ends in a backward branch
        per_stage = tm * 2 * (6 if npl == 3 else 3)                         # MFMAs of one stage of one wave
        loops = [b for b in _blocks(ins) if sum("v_mfma" in i for i in b) >= per_stage and any(i.startswith("s_cbranch") for i in b)]
        assert loops, name
        body = max(loops, key=len)
        dma = [n for n, i in enumerate(body) if "global_load_lds_dwordx4" in i]
        loads = [n for n, i in enumerate(body) if i.startswith("buffer_load_dword")]
        na = (64 * tm * 2) // (128 * wn)
        nxl = {0: 8, 1: 8, 2: 16, 3: 24}[pro]
        assert len(dma) == npl * na, (name, len(dma))
        assert len(loads) == nxl, (name, len(loads))
        if aslots == 2:
            assert max(dma) < min(loads), name                              # "DMA first, then the register loads"
        # (three slots: the wait below lets ALL of this stage's DMAs and register loads stay in flight -- it only needs the
        # previous stage's to have landed -- so their order among themselves does not matter, their NUMBER does)
        bar = max(n for n, i in enumerate(body) if i.startswith("s_barrier"))
        waits = [i for i in body[:bar] if i.startswith("s_waitcnt") and "vmcnt" in i]
        # three-slot weight ring (round 5): this stage's DMA (for stage kt+2) may stay in flight with the register loads
        allowed = nxl + (npl * na if aslots == 3 else 0)
        assert waits and re.search(r"vmcnt\((\d+)\)", waits[-1]).group(1) == str(allowed), (name, waits[-3:])
        deep += int(aslots == 3)
        assert max(loads + dma) < body.index(waits[-1]), name               # ... and the wait comes after all of them
        seen += 1
    assert seen >= 20 and deep >= 10                           # every 128-row-tile instantiation also exists with the deep ring


@pytest.mark.parametrize("src", ["gemm_x2d.hip", "gemm_x2e.hip", "gemm_x2f.hip"])
def test_lds_dma_statements_save_and_restore_m0(src, tmp_path):
    """The weight / operand DMA of the direct GEMMs is inline asm (`buffer_load_dwordx4 ... offen lds`; a builtin would make
    hipcc drain vmcnt in front of every LDS read).  Its LDS base travels in m0.  Round 5 listed "m0" as a clobber
    and hipcc answered 344 times that a reserved register in a clobber list may be ignored; now every
    statement saves m0 to a scalar register, sets it, issues the DMA one wait state later and restores it -- it clobbers
    nothing, whatever the compiler keeps in m0.  Checked on the ISA that ships: no such diagnostic, every DMA is wrapped
    exactly so, and nothing else in the translation unit touches m0."""
    asm = _asm(src, tmp_path)
    assert "reserved registers" not in _DIAG_CACHE[src] and "inline asm clobber" not in _DIAG_CACHE[src], _DIAG_CACHE[src][:400]
    ins = [ln.split(";")[0].strip() for ln in asm.split("\n")]
    ins = [i for i in ins if i and not i.startswith(".") and not i.startswith("#")]
    dma = [n for n, i in enumerate(ins) if i.startswith("buffer_load_dwordx4") and i.endswith(" lds")]
    assert len(dma) >= 8, (src, len(dma))
    # a statement holds one DMA instruction or several back to back (gemm_x2f.hip: two, the second with an immediate offset)
    runs = [[n] for n in dma if n - 1 not in dma]
    for r in runs:
        while r[-1] + 1 in dma:
            r.append(r[-1] + 1)
    for r in runs:
        n, e = r[0], r[-1]
        save = re.match(r"s_mov_b32 (s\d+|vcc_lo|vcc_hi|ttmp\d+), m0$", ins[n - 3])
        assert save, (src, ins[n - 4:e + 2])
        assert re.match(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi)$", ins[n - 2]) and ins[n - 1] == "s_nop 0", (src, ins[n - 4:e + 2])
        assert ins[n - 2] != "s_mov_b32 m0, %s" % save.group(1), (src, ins[n - 4:e + 2])      # (early clobber: another register)
        assert ins[e + 1] == "s_mov_b32 m0, %s" % save.group(1), (src, ins[n - 4:e + 2])
    assert sum(bool(re.search(r"\bm0\b", i)) for i in ins) == 3 * len(runs), src    # nobody else reads or writes m0


def test_one_wave_per_simd_gemm_registers_and_counted_waits(tmp_path):
    """csrc/gemm_x2f.hip (round 6): 256 accumulator registers in AGPRs (16 tiles of a 64-position wave) and at most 256
    vector registers, nothing of the main loop in scratch; per two stages the loop holds 96 MFMAs, 32 + 8 (16 with the
    BatchNorm-backward prologue) fragment / coefficient reads, 2 x NX operand loads and 8 DMA pieces, and the wait in front of
    each of its two barriers is vmcnt(2 NX + 4): this wave's pieces of the NEXT stage have landed, everything younger --
    two stages of operand loads, one stage of pieces -- stays in flight."""
    asm = _asm("gemm_x2f.hip", tmp_path)
    seen = 0
    for name, ins in _functions(asm, "_ZN12_GLOBAL__N_115gemm_x2f_kernel"):
        pro = int(re.search(r"kernelILi(\d)E", name).group(1))
        nx = {1: 8, 2: 16, 3: 24}[pro]
        # two stages per trip; the first trip of a tile is peeled (its stage 0 starts every accumulator with C = 0); hipcc may
        # cut a trip into two blocks at a stage boundary (scalar selects of the next tile's offsets compiled to branches)
        mblocks = [b for b in _blocks(ins) if any("v_mfma" in i for i in b)]
        peeled = [i for b in mblocks if any(re.search(r"v_mfma.*\], 0$", i) for i in b) for i in b]
        loop = [i for b in mblocks if not any(re.search(r"v_mfma.*\], 0$", i) for i in b) for i in b]
        assert sum(bool(re.search(r"v_mfma.*\], 0$", i)) for i in peeled) == 16, name
        for what, body in (("peeled", peeled), ("loop", loop)):
            assert sum("v_mfma" in i for i in body) == 96, (name, what)
            # (tile-level state the epilogue displaced may be reloaded once per tile in the peeled trip; the LOOP touches no scratch)
            assert not any("scratch_store" in i for i in body), (name, what)
            assert what == "peeled" or not any("scratch_" in i for i in body), name
            assert sum(i.startswith("buffer_load_dwordx4") and i.endswith(" lds") for i in body) == 8, (name, what)
            assert sum(i.startswith("buffer_load_dword") and not i.endswith(" lds") for i in body) == 2 * nx, (name, what)
            bars = [n for n, i in enumerate(body) if i.startswith("s_barrier")]
            assert len(bars) == 2, (name, what)
            for n in bars:
                w = [i for i in body[max(0, n - 3):n] if i.startswith("s_waitcnt") and "vmcnt" in i]
                assert w and re.search(r"vmcnt\((\d+)\)", w[-1]).group(1) == str(2 * nx + 4), (name, body[n - 3:n + 1])
        assert any(i.startswith("s_cbranch") for i in loop), name
        seen += 1
    for m in re.finditer(r"\.amdhsa_kernel (\S*gemm_x2f_kernel\S*)", asm):
        seg = asm[m.start():m.start() + 4000]
        assert int(re.search(r"amdhsa_next_free_vgpr (\d+)", seg).group(1)) <= 512, m.group(1)
        assert int(re.search(r"amdhsa_accum_offset (\d+)", seg).group(1)) <= 256, m.group(1)
    assert seen == 4


def test_fused_layer_backward_has_no_scratch(tmp_path):
    """layer_bwd_x2_kernel lives at the edge of the register file (weight fragments, two sets of accumulators and a
    prefetched tile: 230-244 VGPRs); a spill costs it more than the fusion gains (measured: 224 -> 641 us with 400 B)."""
    asm = _asm("layer_bwd_x2.hip", tmp_path)
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S*layer_bwd_x2_kernel\S*)", asm):
        seg = asm[m.start():m.start() + 4000]
        assert int(re.search(r"amdhsa_private_segment_fixed_size (\d+)", seg).group(1)) == 0, m.group(1)
        assert int(re.search(r"amdhsa_next_free_vgpr (\d+)", seg).group(1)) <= 256, m.group(1)
        seen += 1
    assert seen == 7
    # ... and the form that also takes the first layer's weight-gradient sums (round 6): packed FMAs on natural register
    # pairs (test_no_packed_fp32_instruction_selects_halves covers their operand selection), no scratch
    m = re.search(r"\.amdhsa_kernel (\S*layer_bwd_x2ws_kernel\S*)", asm)
    seg = asm[m.start():m.start() + 4000]
    assert int(re.search(r"amdhsa_private_segment_fixed_size (\d+)", seg).group(1)) == 0
    assert int(re.search(r"amdhsa_next_free_vgpr (\d+)", seg).group(1)) <= 256
    body = [b for n, b in _functions(asm, "_ZN12_GLOBAL__N_121layer_bwd_x2ws_kernel")][0]
    assert sum(1 for ln in body if ln.startswith("v_pk_fma_f32")) >= 64


def test_gathered_knn_layer_kernels_registers_and_streaming_loads(tmp_path):
    """csrc/knn_layer.hip (round 6): no scratch; the backward reads (dZ, Y) -- their last use in the step -- with
    non-temporal 16-byte loads and stays at two workgroups of eight waves per CU (<= 128 registers: the form with every
    piece's loads ahead of the first use needed 210 and ran slower)."""
    asm = _asm("knn_layer.hip", tmp_path)
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S*knn_layer_(?:fwd|bwd)_kernel\S*)", asm):
        seg = asm[m.start():m.start() + 4000]
        assert int(re.search(r"amdhsa_private_segment_fixed_size (\d+)", seg).group(1)) == 0, m.group(1)
        assert int(re.search(r"amdhsa_next_free_vgpr (\d+)", seg).group(1)) <= 128, m.group(1)
        seen += 1
    assert seen == 3
    for name, body in _functions(asm, "_ZN12_GLOBAL__N_120knn_layer_bwd_kernel"):
        nt = sum(1 for ln in body if re.search(r"global_load_dwordx4 .* nt", ln))
        assert nt >= 2, (name, nt)


@pytest.mark.parametrize("src", sorted(os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "usip_amd", "csrc", "*.hip"))))
def test_no_packed_fp32_instruction_selects_halves(src, tmp_path):
    """Packed fp32 arithmetic whose operands pick their halves with op_sel is what hipcc's SLP vectoriser emits when it
    pairs the operations of neighbouring channels and finds the registers in the other order.  One such `v_pk_fma_f32`
    (op_sel:[0,1,0] op_sel_hi:[1,0,1]) in the 64 -> 128 fused layer backward returned c3 for c2 * y + c3 in its low half
    on lanes 48-63, rarely and only with two waves on a SIMD (gfx950, ROCm 7.2).  The library is built with
    -fno-slp-vectorize; the packed operations written out by hand never read a high register into a low half.  This
    keeps it that way, for every translation unit of the library."""
    assert "-fno-slp-vectorize" in FLAGS
    asm = _asm(src, tmp_path)
    # op_sel:[..1..] = the LOW half of the result reads the HIGH register of that operand (the failing pattern; a swap
    # or a broadcast of the high half).  op_sel_hi:[..0..] alone -- the low register broadcast to both halves, what
    # `q - (f32x2){p, p}` in the hand-written distance loops compiles to -- is left alone: nearest_kernel's distances
    # are compared bit for bit with the oracle in every run of the suite and have never differed.
    bad = [ln.strip() for ln in asm.split("\n")
           if re.search(r"\bv_pk_(fma|mul|add)_f32\b.*\bop_sel:\[[0-9,]*1", ln)]
    assert not bad, bad[:4]
