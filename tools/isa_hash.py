"""Per-kernel hash of the gfx950 ISA of one csrc/*.hip file: `python tools/isa_hash.py conv.hip > before.txt`, change the source,
run again, diff.  Used to show that adding a kernel (or a template parameter) leaves the existing kernels instruction-identical
(the fp32 inference kernels are register-allocation sensitive)."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "deep-online-video-stabilization_amd", "csrc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-Wno-inline-asm", "-Wno-unused-function"]


def main():
    src = os.path.join(CSRC, sys.argv[1])
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", src, "-o", out])
        text = open(out).read()
    # a function body: from "<name>:" up to its ".Lfunc_end"
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        body = "\n".join(l for l in body.splitlines() if not l.strip().startswith(";") and not l.strip().startswith(".loc"))
        body = re.sub(r"\.LBB\d+_\d+", ".LBB", body)         # block labels carry the function's ordinal
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print("%s  %6d lines  %s" % (hashlib.sha256(body.encode()).hexdigest()[:16], body.count("\n"), demangled))


if __name__ == "__main__":
    main()
