"""Register / scratch audit of the device code inside a built library: every kernel's VGPR / SGPR counts, spills and
private-segment (scratch) bytes from the code objects' AMDGPU metadata notes.
Usage: python tools/check_codeobj.py [simgan_amd/libsimgan_hip.so]      (needs /opt/rocm/lib/llvm/bin; no GPU)
Prints one line per kernel that spills VGPRs or uses scratch memory and a summary; exit status 1 if any does."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """-> list of gfx code-object byte strings inside the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(td, "x")], check=True)
        data = open(fat, "rb").read()
    out = []
    for m in re.finditer(MAGIC, data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + len(MAGIC))
        off = base + len(MAGIC) + 8
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24:off + 24 + tl].decode()
            off += 24 + tl
            if "amdgcn" in triple and sz:
                out.append(data[base + o:base + o + sz])
    return out


def kernels(lib):
    res = []
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            g = lambda key: (re.search(rf"\.{key}:\s*(\S+)", blk) or [None, "0"])[1]  # noqa: E731
            res.append({"name": g("name"), "vgpr": int(g("vgpr_count")), "sgpr": int(g("sgpr_count")), "vgpr_spill": int(g("vgpr_spill_count")),
                        "sgpr_spill": int(g("sgpr_spill_count")), "scratch": int(g("private_segment_fixed_size")), "lds": int(g("group_segment_fixed_size"))})
    return res


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "simgan_amd", "libsimgan_hip.so")
    ks = kernels(lib)
    bad = [k for k in ks if k["vgpr_spill"] or k["scratch"]]     # (SGPR spills go to VGPR lanes, not to memory: reported, not failed)
    for k in bad:
        print(f"{k['name']}: {k['vgpr_spill']} VGPR / {k['sgpr_spill']} SGPR spills, {k['scratch']} B scratch")
    print(f"{len(ks)} kernels in {lib}: {len(bad)} with VGPR spills or scratch; {sum(1 for k in ks if k['sgpr_spill'])} spill SGPRs into VGPR lanes "
          f"(max {max(k['sgpr_spill'] for k in ks)}); max VGPRs {max(k['vgpr'] for k in ks)}")
    sys.exit(1 if bad else 0)
