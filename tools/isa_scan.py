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


_MFMA = re.compile(r"^(v_mfma_\S+)\s+([av])\[(\d+):(\d+)\]")
_REG = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+))\b")
_NOP = re.compile(r"^s_nop\s+(\d+)")


def xdl_results_read_too_early(instructions, kernel_substring, reader=r"^v_min3_f32\b", need=11):
    """Rule 2 (csrc/nn_distance.hip, the digest of the Chamfer scores): the `v_min3_f32` there are inline assembly, which the
    compiler's hazard recogniser does not see -- it neither counts the wait states an XDL result needs before a vector
    instruction may read it (v_mfma_f32_32x32x16_bf16: 11) nor keeps its scheduler from placing the read right behind the
    write.  The source pins the order in groups; this checks the BUILT code: between a v_mfma that writes a register and the
    first `reader` instruction that reads it (in text order, inside the named kernels) lie at least `need` wait states --
    an instruction counts 1, `s_nop N` N + 1, another v_mfma its 8 passes.  -> [(kernel, mfma, reader, wait states)]"""
    rd = re.compile(reader)
    bad = []
    by_fn = {}
    for fn, ins in instructions:
        if kernel_substring in fn:
            by_fn.setdefault(fn, []).append(ins)
    for fn, body in by_fn.items():
        wrote = {}                      # register -> (clock at the write's issue, text of the mfma)
        clock = 0
        for ins in body:
            m = _MFMA.match(ins)
            if rd.match(ins):
                srcs = ins.split(",", 1)[1] if "," in ins else ""
                for kind, lo, hi, one in _REG.findall(srcs):
                    regs = range(int(lo), int(hi) + 1) if one == "" else (int(one),)
                    for r in regs:
                        w = wrote.get((kind, r))
                        if w is not None and clock - w[0] - w[2] < need:
                            bad.append((fn, w[1], ins, clock - w[0] - w[2]))
            if m:
                for r in range(int(m.group(3)), int(m.group(4)) + 1):
                    wrote[(m.group(2), r)] = (clock, ins, 8)
                clock += 8
            else:
                n = _NOP.match(ins)
                clock += int(n.group(1)) + 1 if n else 1
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
    early = xdl_results_read_too_early(ins, "nn_distance_filter_kernel")
    reads = sum(1 for fn, i in ins if "nn_distance_filter_kernel" in fn and i.startswith("v_min3_f32"))
    print("nn_distance_filter_kernel: %d v_min3_f32, %d of them read an XDL result less than 11 wait states after its v_mfma" % (reads, len(early)))
    for e in early[:10]:
        print("   ", e)
    return 1 if (bad or early) else 0


if __name__ == "__main__":
    sys.exit(main())
