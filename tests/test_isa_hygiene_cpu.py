"""CPU (hipcc cross-compiles gfx950 without a GPU): the LDS-DMA conv kernels keep their MFMA fragments in FIXED registers between an
inline-asm ds_read and its explicit wait; a compiler-inserted copy out of such a register in that window would copy stale contents
(it happened in a first version of conv_b2b_f32_kernel).  tools/check_pinned_regs.py compiles conv.hip to
assembly and looks for such copies in the fp32 instantiations."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_copies_out_of_pinned_fragment_registers():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_pinned_regs.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "pinned-register check: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_the_checker_sees_a_planted_copy():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_pinned_regs as C
    asm = ("_Z19conv_b2b_f32_kernelILi2EEv7B2bArgs: ; @x\n\tds_read_b128 v[100:103], v5\n\tv_mov_b64_e32 v[60:61], v[100:101]\n"
           "\ts_waitcnt lgkmcnt(0)\n.Lfunc_end0:\n"
           "_Z20conv_ring_f32_kernelILi0ELi1ELi1ELi0EEv8ConvArgs: ; bf16 variant: converts after the wait, not checked\n"
           "\tv_mov_b32_e32 v4, v101\n.Lfunc_end1:\n"
           "_Z20conv_ring_f32_kernelILi1ELi0ELi1ELi0EEv8ConvArgs: ; @y\n\tv_mov_b32_e32 v4, v99\n\tv_accvgpr_write_b32 a0, v115\n.Lfunc_end2:\n")
    bad = C.check(asm)
    assert [b[0][:24] for b in bad] == ["_Z19conv_b2b_f32_kernelI", "_Z20conv_ring_f32_kernel"] and "v[100:101]" in bad[0][1] and "v115" in bad[1][1]
