"""Scan the gfx950 code of libotgan_hip.so for the one instruction form that was seen computing wrong values.

DESIGN section 3 "Four hazards", item 3 (round 4): a packed fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32, VOP3P) whose
SRC1 has OP_SEL set -- the LOW half of the result takes the HIGH register of the source pair, what the compiler emits for
`acc2 += w2 * x4.y` -- returns a wrong low half in lanes 48-63 when a wave with MFMAs and LDS-DMA loads in flight (the
256 x 128 Winograd-domain GEMM) shares its SIMD.  Every other form measured clean (tools/debug/corun_probe.py form2..form10).
This script lists the packed fp32 instructions of every kernel in the library by form; `bad` are the ones with the src1
op_sel bit.  tests/test_isa_cpu.py asserts there are none.

    python tools/isa_check.py [path/to/libotgan_hip.so]
"""
import collections
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PK = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\b(.*)")
OPSEL = re.compile(r"op_sel:\[([01,]+)\]")


def code_objects(lib):
    """gfx950 code objects of every offload bundle in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib])
        data = open(fat, "rb").read()
        if data[:4] == b"CCOB":     # compressed bundle (e.g. librccl.so): let the bundler decompress the gfx950 entry
            co = os.path.join(td, "gfx950.co")
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
            return [open(co, "rb").read()]
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(data[base + off:base + off + size])
    return out


def scan(lib):
    """-> (counts by form, list of (kernel, instruction text) with the src1 op_sel bit)."""
    forms, bad = collections.Counter(), []
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name],
                                 capture_output=True, text=True, check=True).stdout
        kernel = "?"
        for line in txt.splitlines():
            if line.endswith(">:"):
                kernel = line.split("<")[-1][:-2]
                continue
            m = PK.search(line)
            if not m:
                continue
            sel = OPSEL.search(m.group(2))
            bits = sel.group(1).split(",") if sel else []
            src1 = len(bits) > 1 and bits[1] == "1"
            forms[(m.group(1), "op_sel src1" if src1 else ("op_sel other" if sel else "plain / op_sel_hi only"))] += 1
            if src1:
                bad.append((kernel, line.strip()))
    return forms, bad


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "ot-gan_amd", "csrc", "libotgan_hip.so")
    forms, bad = scan(lib)
    for k, v in sorted(forms.items()):
        print(f"{v:7d}  {k[0]:14s} {k[1]}")
    print("instructions with the failing form:", len(bad))
    for k, l in bad[:40]:
        print("  ", k, "|", l)
    sys.exit(1 if bad else 0)
