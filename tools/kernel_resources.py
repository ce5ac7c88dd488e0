#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of the gfx950 code objects embedded in libvoxtral_hip.so (no GPU needed).

The .so carries one clang offload bundle per translation unit in its `.hip_fatbin` section; every bundle holds the gfx950 code object, whose
AMDGPU metadata note lists, per kernel, `.vgpr_count`, `.vgpr_spill_count`, `.sgpr_spill_count`, `.private_segment_fixed_size` (scratch bytes per
lane) and `.group_segment_fixed_size`.  `kernel_resources(path)` returns them as a list of dicts; run as a script it prints the table
(`--spills`: only kernels that spill or use scratch).  tests/test_abi_cpu.py::test_no_kernel_spills_or_scratch holds the shipped library to zero.
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so_path: str, arch: str = "gfx950"):
    """The `arch` code objects (bytes) of every offload bundle in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path, os.devnull])
        blob = open(fat, "rb").read()
    out = []
    for m in re.finditer(MAGIC, blob):
        s = m.start()
        (n,) = struct.unpack_from("<Q", blob, s + len(MAGIC))
        o = s + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, o); o += 24
            triple = blob[o:o + tl].decode(); o += tl
            if arch in triple and size:
                out.append(blob[s + off:s + off + size])
    return out


def demangle(names):
    try:
        exe = os.path.join(LLVM_BIN, "llvm-cxxfilt")
        r = subprocess.run([exe if os.path.exists(exe) else "c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        out = r.stdout.splitlines()
        return out if len(out) == len(names) else list(names)
    except Exception:
        return list(names)


def kernel_resources(so_path: str, arch: str = "gfx950"):
    import yaml
    keys = (".name", ".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
            ".group_segment_fixed_size", ".uses_dynamic_stack", ".max_flat_workgroup_size")
    rows = []
    for co in code_objects(so_path, arch):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co); f.flush()
            txt = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True, check=True).stdout
        m = re.search(r"^\s*---\s*$(.*?)^\s*\.\.\.\s*$", txt, re.S | re.M)      # the metadata note is one YAML document
        if not m:
            continue
        meta = yaml.safe_load(m.group(1))
        for k in meta.get("amdhsa.kernels", []):
            rows.append({key: k[key] for key in keys if key in k})
    for r, dn in zip(rows, demangle([r[".name"] for r in rows])):
        r["demangled"] = dn
    return rows


PK_SRC1_SWAP = re.compile(r"v_pk_(?:mul|add|fma)_f32\b.*\bop_sel:\[[01],1|v_pk_mov_b32\b.*\bop_sel:\[1,0\]")      # low lane <- src1.high; v_pk_mov_b32: low <- src0.high AND high <- src1.low


def packed_src1_swaps(so_path: str, arch: str = "gfx950"):
    """[(kernel, instruction)] for every packed-FP32 instruction of the shipped code objects whose op_sel takes SRC1's high dword for the low lane -- the encoding that
    returns a wrong low lane on MI355X while another wave of the CU runs MFMAs (csrc/vox_kernels.h VOX_NO_PK_F32, tools/repro/pk_fp32_corun.cpp)."""
    out = []
    for co in code_objects(so_path, arch):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co); f.flush()
            txt = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for ln in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
            if m:
                cur = m.group(1); continue
            if PK_SRC1_SWAP.search(ln):
                out.append((cur, ln.strip().split("//")[0].strip()))
    names = demangle([k for k, _ in out])
    return [(n, i) for n, (_, i) in zip(names, out)]


def main(argv):
    here = os.path.dirname(os.path.abspath(__file__))
    so = next((a for a in argv if not a.startswith("--")), os.path.join(here, "..", "voxtral-mini-realtime-rs_amd", "libvoxtral_hip.so"))
    rows = kernel_resources(so)
    only = "--spills" in argv
    n_bad = 0
    for r in sorted(rows, key=lambda r: -(r.get(".vgpr_spill_count", 0) * 1000 + r.get(".private_segment_fixed_size", 0))):
        bad = r.get(".vgpr_spill_count", 0) or r.get(".sgpr_spill_count", 0) or r.get(".private_segment_fixed_size", 0)
        n_bad += bool(bad)
        if only and not bad:
            continue
        print(f"vgpr {r.get('.vgpr_count', 0):4d} agpr {r.get('.agpr_count', 0):3d} vspill {r.get('.vgpr_spill_count', 0):4d} sspill {r.get('.sgpr_spill_count', 0):3d} "
              f"scratch {r.get('.private_segment_fixed_size', 0):5d} lds {r.get('.group_segment_fixed_size', 0):6d}  {r['demangled'][:150]}")
    print(f"{len(rows)} kernels, {n_bad} with spills or scratch", file=sys.stderr)


if __name__ == "__main__":
    main(sys.argv[1:])
