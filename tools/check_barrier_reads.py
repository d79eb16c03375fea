"""Static check of the BUILT library for the LDS-DMA ring hazard ("reads landed before the barrier").

A ring stage is refilled by another wave's LDS-DMA (`global_load_lds_*`) right behind the `s_barrier` that every wave reaches after reading the
stage.  The barrier orders instruction ISSUE, not the return of `ds_read`s: a wave that passes it with fragment reads still queued in the LDS pipe
can have them overtaken by the DMA's write (seen in conv3d_k4s2_halo_kernel at batch 16: one encode in seven differed by 1e-3..5e-3).  hipcc sinks
a stage's last MFMAs -- and the `s_waitcnt lgkmcnt` in front of them -- below a raw `__builtin_amdgcn_s_barrier()`, so the source has to say
`s_waitcnt lgkmcnt(0)` before such a barrier, and this script checks that the code that ships does:

    python tools/check_barrier_reads.py [path/to/libpcd_hip.so]

It unbundles every gfx950 code object of the library, disassembles it (llvm-objdump), builds each kernel's control-flow graph and propagates the
queue of outstanding LGKM operations (ds_read / other) through it to a fixed point (`s_waitcnt lgkmcnt(N)` keeps the N youngest).  In a kernel that
issues LDS-DMA, an `s_barrier` reached with a `ds_read` possibly outstanding is reported; exit code 1 if any is.  The rule is stricter than the
hazard (a read of a region no DMA ever writes would be harmless) and that is the point: no case-by-case reasoning in the kernels.

Second check, same disassembly: no instruction touches the destination registers of a vector-memory load that may still be in flight.  The compiler
keeps that rule for the loads it sees; a load issued from `asm volatile(... : "=v"(x) ...)` is, to the compiler, complete at the asm statement, and
a copy, spill or early use of `x` before the kernel's own counted `s_waitcnt vmcnt(N)` reads a register the data has not reached (the weights-in-
registers kernels, gemm_xw_kernel and conv3d_k4s2_halo_kernel issue such loads and fence them with `asm volatile("" : "+v"(x))` behind the wait).
Model: the in-order queue of outstanding vector-memory operations (loads with their destination registers, stores, atomics, LDS-DMA);
`s_waitcnt vmcnt(N)` keeps the N youngest (`flat_*` returns out of order: with one pending only `vmcnt(0)` retires); every other instruction's
VGPR / AGPR operands are checked against the pending destinations.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_objects(lib_path):
    """The gfx950 ELF images inside the library's uncompressed offload bundles."""
    blob = open(lib_path, "rb").read()
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        (count,) = struct.unpack_from("<Q", blob, i + 24)
        o = i + 32
        for _ in range(count):
            off, size, tl = struct.unpack_from("<QQQ", blob, o)
            o += 24
            triple = blob[o:o + tl].decode()
            o += tl
            if "gfx950" in triple and size:
                yield blob[i + off:i + off + size]
        pos = i + len(MAGIC)


def kernels(disassembly):
    """(name, [(label | None, opcode, operands)]) per function of one llvm-objdump -d --symbolize-operands listing."""
    name, body = None, []
    for line in disassembly.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
        if m:
            if re.fullmatch(r"L\d+", m.group(1)):
                body.append((m.group(1), None, None))
            else:
                if name:
                    yield name, body
                name, body = m.group(1), []
            continue
        if name is None or not line.startswith("\t"):
            continue
        text = line.split("//")[0].strip()
        if text:
            parts = text.split(None, 1)
            body.append((None, parts[0], parts[1] if len(parts) > 1 else ""))
    if name:
        yield name, body


def merge(a, b):
    """Join of two outstanding-queues, aligned at the youngest entry: a position holds a read if it may in either."""
    if a is None:
        return b
    if b is None:
        return a
    n = max(len(a), len(b))
    pa, pb = ("o",) * (n - len(a)) + a, ("o",) * (n - len(b)) + b
    return tuple("r" if x == "r" or y == "r" else "o" for x, y in zip(pa, pb))


def check_kernel(body):
    # basic blocks: a label starts one, a branch / s_endpgm ends one
    blocks, labels, cur = [], {}, []
    for lab, op, arg in body:
        if lab is not None:
            if cur:
                blocks.append(cur)
            cur = []
            labels[lab] = len(blocks)
            continue
        cur.append((op, arg))
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    uses_dma = any("global_load_lds" in op or (op.startswith("buffer_load") and " lds" in arg) for blk in blocks for op, arg in blk)
    if not uses_dma:
        return []
    succ = []
    for i, blk in enumerate(blocks):
        op, arg = blk[-1] if blk else ("", "")
        s = []
        if op == "s_branch":
            s.append(labels[arg.strip()])
        elif op.startswith("s_cbranch"):
            s.append(labels[arg.strip()])
            s.append(i + 1)
        elif op != "s_endpgm":
            s.append(i + 1)
        succ.append([t for t in s if t < len(blocks)])
    state_in = [None] * len(blocks)
    state_in[0] = ()
    work, found = [0], {}
    while work:
        i = work.pop()
        q = state_in[i]
        for k, (op, arg) in enumerate(blocks[i]):
            if op.startswith("ds_read") or op.startswith("ds_load"):
                q = (q + ("r",))[-64:]
            elif op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op in ("s_sendmsg", "s_memtime", "s_memrealtime"):
                q = (q + ("o",))[-64:]
            elif op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", arg)
                if m:
                    n = int(m.group(1))
                    q = q[len(q) - n:] if n else ()
            elif op == "s_barrier" and "r" in q:
                found[(i, k)] = max(found.get((i, k), 0), q.count("r"))
        for t in succ[i]:
            new = merge(state_in[t], q)
            if new != state_in[t]:
                state_in[t] = new
                work.append(t)
    return sorted(found.items())


REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


VMEM_PREFIX = ("global_", "buffer_", "scratch_", "flat_", "tbuffer_", "image_")


def vmem_entry(op, arg):
    """None if `op` is not a vector-memory instruction, else (destination registers, is_flat)."""
    if not op.startswith(VMEM_PREFIX):
        return None
    dst = frozenset()
    first = arg.split(",")[0] if arg else ""
    if "_load_" in op and "_lds_" not in op and " lds" not in arg:
        dst = frozenset(regs_of(first))
    elif "_atomic_" in op and (" sc0" in arg or " glc" in arg):
        dst = frozenset(regs_of(first))
    return dst, op.startswith("flat_")


def merge_vm(a, b):
    if a is None:
        return b
    if b is None:
        return a
    n = max(len(a), len(b))
    pad = (frozenset(), False)
    pa, pb = (pad,) * (n - len(a)) + a, (pad,) * (n - len(b)) + b
    return tuple((x[0] | y[0], x[1] or y[1]) for x, y in zip(pa, pb))


def check_vmem_kernel(body):
    """[(instruction text, registers touched while their load may be in flight)] for one function body."""
    blocks, labels, cur = [], {}, []
    for lab, op, arg in body:
        if lab is not None:
            if cur:
                blocks.append(cur)
            cur = []
            labels[lab] = len(blocks)
            continue
        cur.append((op, arg))
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for i, blk in enumerate(blocks):
        op, arg = blk[-1] if blk else ("", "")
        s = []
        if op == "s_branch":
            s.append(labels[arg.strip()])
        elif op.startswith("s_cbranch"):
            s.append(labels[arg.strip()])
            s.append(i + 1)
        elif op != "s_endpgm":
            s.append(i + 1)
        succ.append([t for t in s if t < len(blocks)])
    state_in = [None] * len(blocks)
    state_in[0] = ()
    work, found = [0], {}
    while work:
        i = work.pop()
        q = state_in[i]
        for k, (op, arg) in enumerate(blocks[i]):
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", arg)
                if m:
                    n = int(m.group(1))
                    if n == 0:
                        q = ()
                    elif not any(e[1] for e in q):
                        q = q[len(q) - n:] if len(q) > n else q
                continue
            pending = set().union(*(e[0] for e in q)) if q else set()
            if pending:
                # (a second load INTO a pending destination is what the compiler emits for the two arms of a lane-divergent branch -- vector path and
                # element path of the same value under complementary exec masks -- and returns in order: only its address operands are checked)
                operands = arg.split(",", 1)[1] if vmem_entry(op, arg) is not None and vmem_entry(op, arg)[0] and "," in arg else arg
                hit = regs_of(operands) & pending
                if hit:
                    found[(i, k)] = (f"{op} {arg}", sorted(hit))
            e = vmem_entry(op, arg)
            if e is not None:
                q = (q + (e,))[-64:]
        for t in succ[i]:
            new = merge_vm(state_in[t], q)
            if new != state_in[t]:
                state_in[t] = new
                work.append(t)
    return [found[key] for key in sorted(found)]


# Kernels whose counted waits depend on WHICH path was taken (a wait of vmcnt(4 + NST) exactly on the paths that issued NST more operations): a
# path-insensitive join of the queues cannot verify them and reports phantom pending loads.  They are listed, counted and not failed on; what covers
# them is dynamic: bitwise equality with the LDS-staged kernel on every GEMM test shape and the repeated-launch screens.
PATH_SENSITIVE = {
    "gemm_xw_kernel": "wait_vmcnt<4 + NST> / <4> by `behind_atomics`, <8> / <10> by the bias pieces of a tile boundary, <12> / <4> by which wave group requested the "
                      "K tile's pieces (csrc/gemm_f16.hip)",
    "gemm_xs_kernel": "the same structure with the epilogue's ND stores and the R drip stores of a K tile in the counts; the k-step-1 wait takes its count at run time "
                      "(wait_vmcnt_rt: requesting group, tile boundary, dripping, behind an epilogue); covered by "
                      "tests/test_gpu_kernels.py::test_gemm_store_with_fragment_order_weights_exact (bitwise v. gemm_xp_kernel, every combination, repeated launches)",
}


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "3d-shape-generation_amd", "libpcd_hip.so")
    bad = checked = early = skipped = 0
    with tempfile.TemporaryDirectory() as tmp:
        for n, image in enumerate(device_objects(lib)):
            path = os.path.join(tmp, f"dev{n}.co")
            open(path, "wb").write(image)
            dis = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", path], capture_output=True, text=True, check=True).stdout
            for name, body in kernels(dis):
                checked += 1
                for (blk, k), reads in check_kernel(body):
                    bad += 1
                    print(f"{name}: s_barrier (block {blk}, instruction {k}) reached with up to {reads} ds_read outstanding")
                hits = check_vmem_kernel(body)
                if any(k in name for k in PATH_SENSITIVE):
                    skipped += len(hits)
                    continue
                for text, regs in hits:
                    early += 1
                    print(f"{name}: `{text.strip()}` touches {regs[0][0]}{regs[0][1]}.. ({len(regs)} registers) of a load that may still be in flight")
    print(f"{checked} kernels checked, {bad} barrier(s) with reads outstanding in LDS-DMA kernels, {early} use(s) of a load's registers before its wait"
          f" ({skipped} unverifiable reports in path-sensitive kernels: {', '.join(PATH_SENSITIVE)})")
    return 1 if bad or early or not checked else 0


if __name__ == "__main__":
    sys.exit(main())
