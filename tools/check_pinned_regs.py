"""The LDS-DMA conv kernels (conv_ring_kernel.h, conv_b2b_kernel.h) read their MFMA fragments with inline-asm ds_read_b128 into
FIXED registers v[100:115] (+ v[116:131] for the prologue form) and wait for them with explicit lgkmcnt waits.  Between a read and
its wait the compiler believes the register already holds the data: a register COPY out of a pinned register in that window copies
stale contents (this happened in a first version of conv_b2b_f32_kernel, where a fragment read sat on only some control-flow paths).
This script compiles csrc/conv.hip to gfx950 assembly and fails if any of those kernels contains a move whose SOURCE is a pinned
fragment register.  `python tools/check_pinned_regs.py` (about 40 s)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "deep-online-video-stabilization_amd", "csrc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-Wno-inline-asm", "-Wno-unused-function"]
PINNED = set(range(100, 116))


def regs_of(operand):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", operand)
    return {int(m.group(1))} if m else set()


def check(text, prefixes=("_Z20conv_ring_f32_kernel", "_Z19conv_b2b_f32_kernel")):
    """fp32 instantiations only: the bf16-operand and split variants (second template argument 1 .. 4) convert their fragments with
    vector ALU instructions AFTER the wait, which legitimately reads the pinned registers (check_inflight covers them)."""
    bad = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not name.startswith(prefixes) or re.match(r"_Z20conv_ring_f32_kernelILi\dELi[1-5]E", name):
            continue
        for line in body.splitlines():
            t = line.strip()
            mm = re.match(r"(v_mov_b32_e32|v_mov_b64_e32|v_accvgpr_write_b32)\s+([^,]+),\s*(\S+)", t)
            if mm and regs_of(mm.group(3)) & PINNED:
                bad.append((name, t))
    return bad


def check_inflight(text, prefix="_Z20conv_ring_f32_kernel"):
    """Stronger check for every ring kernel: walk the instruction stream in layout order, keep the ds_read_b128 destinations that
    have not been waited for (LDS operations return in order: s_waitcnt lgkmcnt(n) retires all but the youngest n), and report any
    instruction that READS such a register.  Layout order is an approximation of control flow; a hit is a reason to look."""
    bad = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not name.startswith(prefix):
            continue
        fifo = []                                           # destination register sets of the LDS reads in flight, oldest first
        for line in body.splitlines():
            t = line.strip()
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            op, _, rest = t.partition(" ")
            ops = [o.strip() for o in rest.split(",")] if rest else []
            if op == "s_waitcnt":
                mm = re.search(r"lgkmcnt\((\d+)\)", t)
                if mm:
                    n = int(mm.group(1))
                    while len(fifo) > n:
                        fifo.pop(0)
                continue
            if op.startswith("ds_read"):
                fifo.append(regs_of(ops[0]))
                continue
            if op.startswith("ds_write") or op.startswith("ds_"):
                fifo.append(set())                          # counts in lgkmcnt, writes no register
            flying = set().union(*fifo) if fifo else set()
            if not flying:
                continue
            srcs = ops[1:] if (op.startswith("v_") or op.startswith("global_") or op.startswith("buffer_")) else ops
            if op.startswith("v_mfma") or op.startswith("global_store") or op.startswith("ds_write"):
                srcs = ops[1:] if op.startswith("v_mfma") else ops
            for o in srcs:
                o = o.split(" ")[0]
                if regs_of(o) & flying:
                    bad.append((name, t))
                    break
    return bad


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "conv.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", os.path.join(CSRC, "conv.hip"), "-o", out],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
        bad = check(text)
        bad2 = check_inflight(text)
    for name, t in bad[:20]:
        print("copy out of a pinned fragment register in %s: %s" % (name, t))
    for name, t in bad2[:20]:
        print("read of a register whose LDS read is in flight in %s: %s" % (subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip(), t))
    print("pinned-register check: %s" % ("FAILED (%d + %d)" % (len(bad), len(bad2)) if (bad or bad2) else "ok"))
    return 1 if (bad or bad2) else 0


if __name__ == "__main__":
    sys.exit(main())
