#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel of libsart, read from the metadata of the gfx950 code object inside
csrc/build/*.o (no GPU needed):  python tools/kernel_resources.py [substring of the kernel name]

A trace kernel must show scratch 0 (no spills to memory) and <= 128 VGPRs (4 waves / SIMD at 1024 threads per workgroup)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object_notes(obj_path: str) -> str:
    with tempfile.TemporaryDirectory() as tmp:
        obj = os.path.join(tmp, os.path.basename(obj_path))
        shutil.copy(obj_path, obj)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", obj], check=True, capture_output=True, cwd=tmp)
        dev = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not dev:          # a translation unit without kernels
            return ""
        assert len(dev) == 1, dev
        return subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, dev[0])], capture_output=True,
                              text=True, check=True).stdout


def kernels(notes: str):
    out = []
    for blk in re.split(r"\n  - \.a", notes):
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1)) if re.search(r"\.%s:\s+(\d+)" % key, blk) else -1
        out.append({"name": name.group(1), "vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"), "scratch": g("private_segment_fixed_size"),
                    "lds": g("group_segment_fixed_size"), "vgpr_spill": g("vgpr_spill_count"), "sgpr_spill": g("sgpr_spill_count"),
                    "kernarg": g("kernarg_segment_size")})
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines() if p.returncode == 0 else names


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    build = os.path.join(ROOT, "solaraxionraytracing_amd", "csrc", "build")
    for o in sorted(f for f in os.listdir(build) if f.endswith(".o") and f != "build_id.o"):
        ks = kernels(code_object_notes(os.path.join(build, o)))
        for k, nice in zip(ks, demangle([k["name"] for k in ks])):
            if pat in nice:
                print("%-100s vgpr %3d sgpr %3d scratch %4d lds %6d spills v%d s%d kernarg %d" % (
                    nice[:100], k["vgpr"], k["sgpr"], k["scratch"], k["lds"], k["vgpr_spill"], k["sgpr_spill"], k["kernarg"]))


if __name__ == "__main__":
    main()
