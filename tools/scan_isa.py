"""Disassembles the gfx950 code objects inside a built shared library (the .hip_fatbin section: one clang offload bundle per
translation unit) and lists packed fp32 VALU instructions whose LOW result takes the HIGH half of the second source
(op_sel[1] = 1): the forms that return wrong values in lanes 48-63 next to another stream's bf16-MFMA GEMM on gfx950
(tools/ubench/pk_opsel_repro.hip, docs/history/DESIGN_rounds_1-5.md 9).  Used by tests/test_host.py; `python tools/scan_isa.py <lib.so>` prints a census."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
UNSAFE = re.compile(r"v_pk_(?:add|mul|fma)_f32\s.*op_sel:\[[01],1")
PACKED = re.compile(r"v_pk_(?:add|mul|fma)_f32\s")


def code_objects(so_path):
    """-> list of (triple, ELF bytes) of the device code objects bundled into the library"""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path])
        data = open(fat, "rb").read()
    out, pos = [], data.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", data, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "amdgcn" in triple and size:
                out.append((triple, data[pos + off:pos + off + size]))
        pos = data.find(MAGIC, pos + len(MAGIC))
    return out


def disassemble(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes)
        f.flush()
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout


def census(so_path):
    """-> (number of code objects, {kernel symbol: unsafe count}, total packed fp32 instructions)"""
    per, packed, objs = {}, 0, code_objects(so_path)
    for _, elf in objs:
        cur = None
        for line in disassemble(elf).splitlines():
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
            if m:
                cur = m.group(1)
            if PACKED.search(line):
                packed += 1
                if UNSAFE.search(line):
                    per[cur] = per.get(cur, 0) + 1
    return len(objs), per, packed


if __name__ == "__main__":
    n, per, packed = census(sys.argv[1])
    print(f"{n} code objects, {packed} packed fp32 instructions, {sum(per.values())} with op_sel[1] = 1")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1]):
        print(f"  {v:5d}  {k}")
