"""CPU: the C-ABI library loads and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(cloudaae_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import torch  # noqa: F401  (binds the library to torch's HIP runtime, as the product does)
    path = os.path.join(ROOT, "cloudaae_amd", "libcloudaae_hip.so")
    assert os.path.exists(path), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(path)
    declared = _declared()
    assert len(declared) >= 8
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    # one ABI revision: the header's, the library's and the Python host's signature table (cloudaae_amd/_lib.py)
    header = open(os.path.join(ROOT, "include", "cloudaae_hip.h")).read()
    abi = int(re.search(r"#define\s+CLOUDAAE_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.cloudaae_version() == abi
    from cloudaae_amd import _lib
    assert _lib.ABI_VERSION == abi


def test_header_is_plain_c():
    # no torch / C++ types may leak into the boundary
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        assert "torch" not in text.lower().replace("pytorch", "")
        assert "std::" not in text and "at::" not in text


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "cloudaae_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "liboracle" not in text, f


def test_fc_ticket_is_taken_after_the_atomics_are_drained():
    """csrc/fc.hip, a product cut over K (and over row tiles) and finished by the last workgroup to arrive: every
    wave must wait for what it publishes -- the agent-scope stores of its partial tile -- with s_waitcnt vmcnt(0)
    BEFORE the barrier that precedes the ticket: s_barrier does not drain the counter, and a ticket published
    early lets the last workgroup read incomplete sums.
    Checked on the emitted gfx950 ISA: between the last publishing instruction (global_store_dword ... sc1) and
    the ticket (global_atomic_add ... sc0) there is an s_waitcnt vmcnt(0) in front of the s_barrier; and the
    partial tiles are read back with agent-scope loads (sc1)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc on this machine")
    src = os.path.join(ROOT, "cloudaae_amd", "csrc", "fc.hip")
    asm = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                          "-munsafe-fp-atomics", "-S", "--cuda-device-only", src, "-o", "-"],
                         check=True, capture_output=True, text=True).stdout.splitlines()
    tickets = [i for i, l in enumerate(asm) if re.search(r"\bglobal_atomic_add\s.*\bsc0\b", l)]
    assert len(tickets) >= 6, "expected a ticket per instantiation of the forward body in fc.hip's ISA"
    publish = re.compile(r"global_store_dword\s.*\bsc1\b")
    kinds = set()
    for t in tickets:
        pubs = [i for i in range(t) if publish.search(asm[i])]
        assert pubs, "ticket without preceding publication"
        kinds.add("store")
        window = [l.strip() for l in asm[pubs[-1] + 1:t]]
        bar = [i for i, l in enumerate(window) if l.startswith("s_barrier")]
        assert bar, "no barrier between the publication and the ticket"
        waits = [i for i, l in enumerate(window[:bar[0]]) if re.match(r"s_waitcnt\s+vmcnt\(0\)", l)]
        assert waits, "the publication is not drained before the barrier:\n" + "\n".join(window)
    assert kinds == {"store"}, kinds
    assert any(re.search(r"buffer_load_dwordx4\s.*\bsc1\b", l) for l in asm), "partial tiles not read at agent scope"
