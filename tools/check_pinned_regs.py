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
    """fp32 instantiations only: the bf16-operand variants (second template argument 1) convert their fragments with vector ALU
    instructions AFTER the wait, which legitimately reads the pinned registers."""
    bad = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not name.startswith(prefixes) or re.match(r"_Z20conv_ring_f32_kernelILi\dELi1E", name):
            continue
        for line in body.splitlines():
            t = line.strip()
            mm = re.match(r"(v_mov_b32_e32|v_mov_b64_e32|v_accvgpr_write_b32)\s+([^,]+),\s*(\S+)", t)
            if mm and regs_of(mm.group(3)) & PINNED:
                bad.append((name, t))
    return bad


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "conv.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", os.path.join(CSRC, "conv.hip"), "-o", out],
                              stderr=subprocess.DEVNULL)
        bad = check(open(out).read())
    for name, t in bad[:20]:
        print("copy out of a pinned fragment register in %s: %s" % (name, t))
    print("pinned-register check: %s" % ("FAILED (%d)" % len(bad) if bad else "ok"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
