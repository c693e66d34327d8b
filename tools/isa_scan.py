"""Disassemble the gfx950 code objects inside a built libcloudaae_hip.so and check instruction-level rules on what was
actually built (tests/test_isa_rules.py; also a CLI:  python tools/isa_scan.py [lib.so]).

Rule 1 (profiles/notes_two_processes_one_gpu.md, round 6): no packed-fp32 instruction takes the LOW half of its result from
the HIGH register of a source pair -- `op_sel:[...]` with a 1 in it on v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32.  On MI355X
such an instruction now and then returns `src0 + 0` in lanes 48-63 when a wave of another process shares the SIMD.

Needs llvm-objcopy, clang-offload-bundler and llvm-objdump of the ROCm LLVM (/opt/rocm/lib/llvm/bin); no GPU."""
import os
import re
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def tools_present():
    return all(os.path.exists(os.path.join(LLVM_BIN, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"))


def disassemble(lib_path):
    """-> list of (kernel name, instruction text) over every gfx950 code object of the library (one per translation unit)"""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat],
                       check=True)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(_MAGIC), data)]
        assert starts, "no offload bundle in " + lib_path
        for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(data)])):
            piece = os.path.join(tmp, "bundle%d.bin" % n)
            co = os.path.join(tmp, "dev%d.co" % n)
            open(piece, "wb").write(data[a:b])
            subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + piece,
                            "--targets=" + _TARGET, "--output=" + co], check=True, capture_output=True)
            if not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            text = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True,
                                  capture_output=True, text=True).stdout
            fn = "?"
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    fn = m.group(1)
                    continue
                ins = line.strip()
                if ins and not ins.startswith(("Disassembly", dev_prefix(co))):
                    out.append((fn, re.sub(r"\s*//.*$", "", ins)))
    return out


def dev_prefix(path):
    return path + ":"


_PK_F32 = re.compile(r"^v_pk_(add|mul|fma)_f32\b")
_OP_SEL = re.compile(r"\bop_sel:\[([01,]+)\]")


def packed_f32_low_from_high(instructions):
    """the instructions rule 1 forbids: [(kernel, instruction)]"""
    bad = []
    for fn, ins in instructions:
        if _PK_F32.match(ins):
            m = _OP_SEL.search(ins)
            if m and "1" in m.group(1):
                bad.append((fn, ins))
    return bad


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "cloudaae_amd", "libcloudaae_hip.so")
    ins = disassemble(lib)
    kernels = len(set(fn for fn, _ in ins))
    pk = sum(1 for _, i in ins if _PK_F32.match(i))
    bad = packed_f32_low_from_high(ins)
    print("%s: %d instructions in %d functions, %d packed-fp32 instructions, %d with the low half taken from a high register"
          % (lib, len(ins), kernels, pk, len(bad)))
    for fn, i in bad[:40]:
        print("   ", fn, "|", i)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
