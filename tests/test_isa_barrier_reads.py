"""The shipped code objects are free of the LDS-DMA ring hazard: no kernel that issues LDS-DMA reaches an s_barrier with one of its own ds_reads
still outstanding (tools/check_barrier_reads.py: the barrier orders issue, not return, and another wave's DMA refill behind it can overtake a queued
read -- the cause of the batch-16 encode results that differed from run to run in round 4).  Static: disassembles libpcd_hip.so, needs no GPU."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_barrier_reads", os.path.join(ROOT, "tools", "check_barrier_reads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_merge_and_wait_model():
    t = _tool()
    assert t.merge(("r",), ("o", "o", "o")) == ("o", "o", "r")
    assert t.merge(None, ("r",)) == ("r",)
    # a loop whose body leaves one read in flight at the barrier that follows the back edge is found; with the wait in front of the barrier it is not
    def body(wait):
        return [(None, "global_load_lds_dwordx4", "v[0:1], off"), ("L0", None, None), (None, "s_waitcnt", wait), (None, "s_barrier", ""),
                (None, "ds_read_b128", "v[0:3], v4"), (None, "ds_read_b128", "v[4:7], v4"), (None, "s_waitcnt", "lgkmcnt(1)"),
                (None, "s_cbranch_scc1", "L0"), (None, "s_endpgm", "")]
    assert t.check_kernel(body("vmcnt(0)")) == [((1, 1), 1)]
    assert t.check_kernel(body("vmcnt(0) lgkmcnt(0)")) == []
    # no LDS-DMA in the kernel: register-staged rings write with ds_write, which queues behind the reads
    assert t.check_kernel(body("vmcnt(0)")[1:]) == []


def test_load_destination_model():
    """The second check: a load's destination registers may not be touched before a wait that covers the load (in-order vmcnt)."""
    t = _tool()
    ld = lambda d, a: (None, "global_load_dwordx4", f"v[{d}:{d + 3}], v[{a}:{a + 1}], off")
    use = (None, "v_mfma_f32_16x16x32_f16", "a[0:3], v[8:11], v[20:23], a[0:3]")
    end = (None, "s_endpgm", "")
    # two loads, wait for all but the youngest, use the OLDER one's registers: fine; use the younger one's: reported
    assert t.check_vmem_kernel([ld(8, 0), ld(20, 2), (None, "s_waitcnt", "vmcnt(1)"), (None, "v_add_f32_e32", "v1, v8, v9"), end]) == []
    hits = t.check_vmem_kernel([ld(8, 0), ld(20, 2), (None, "s_waitcnt", "vmcnt(1)"), use, end])
    assert len(hits) == 1 and ("v", 20) in hits[0][1]
    # a store in between counts in vmcnt like a load; an LDS-DMA has no destination registers
    seq = [ld(8, 0), (None, "global_store_dwordx4", "v[0:1], v[30:33], off"), (None, "global_load_lds_dwordx4", "v[2:3], off"),
           (None, "s_waitcnt", "vmcnt(2)"), (None, "v_add_f32_e32", "v1, v8, v9"), end]
    assert t.check_vmem_kernel(seq) == []
    seq[3] = (None, "s_waitcnt", "vmcnt(3)")
    assert len(t.check_vmem_kernel(seq)) == 1
    # a copy of an in-flight register (what a spill or a coalescing move of an asm-loaded value would be) is reported
    assert len(t.check_vmem_kernel([ld(8, 0), (None, "v_mov_b32_e32", "v40, v9"), (None, "s_waitcnt", "vmcnt(0)"), end])) == 1
    # the second arm of a lane-divergent load into the same register is not
    assert t.check_vmem_kernel([ld(8, 0), (None, "global_load_dword", "v8, v[4:5], off"), (None, "s_waitcnt", "vmcnt(0)"), end]) == []


def test_built_library_has_no_read_outstanding_at_a_ring_barrier(capsys):
    t = _tool()
    lib = os.path.join(ROOT, "3d-shape-generation_amd", "libpcd_hip.so")
    if not os.path.exists(t.OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not present")
    assert os.path.exists(lib), "build the library first: python __graft_entry__.py build"
    images = list(t.device_objects(lib))
    assert len(images) >= 10
    import sys
    argv = sys.argv
    sys.argv = ["check_barrier_reads.py", lib]
    try:
        rc = t.main()
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    assert rc == 0, out
    assert "0 barrier(s)" in out and " 0 use(s)" in out
