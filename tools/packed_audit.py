"""Which kernels of the shipped library contain packed-fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 / v_pk_mov_b32)?  DESIGN.md 5.4: those are the
instructions whose results went wrong in a kernel co-resident with another queue's matrix + vector kernel.  Extracts the gfx950 code objects from the .so
(llvm-objdump --offloading), disassembles them and counts per kernel.  CPU.
usage: python tools/packed_audit.py [library.so]   ->   table on stdout;  packed_counts(path) -> {kernel: count}"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def packed_counts(lib=None):
    lib = lib or os.path.join(ROOT, "gshell_amd", "lib", "libgshell_hip.so")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            dis = subprocess.run([OBJDUMP, "-d", "--demangle", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            name = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    name = m.group(1)
                    out.setdefault(name, 0)
                elif name is not None and re.search(r"\bv_pk_(mul|add|fma)_f32\b|\bv_pk_mov_b32\b", line):
                    out[name] += 1
    return out


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*", "", name)


if __name__ == "__main__":
    c = packed_counts(sys.argv[1] if len(sys.argv) > 1 else None)
    with_pk = sorted(((v, short(k)) for k, v in c.items() if v), reverse=True)
    without = sorted(short(k) for k, v in c.items() if not v)
    print(f"{len(c)} kernels; {len(with_pk)} contain packed-fp32 instructions:")
    for v, k in with_pk:
        print(f"  {v:5d}  {k}")
    print(f"{len(without)} without: " + ", ".join(without))
