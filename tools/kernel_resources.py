#!/usr/bin/env python3
"""Register / scratch / LDS figures of every kernel in a built libvp_amd*.so, read from the code objects' own metadata
(the AMDGPU notes the loader uses) -- no compiler run, no GPU.

    python tools/kernel_resources.py [path/to/lib.so] [--md]

`.private_segment_fixed_size` is the scratch (spill) bytes per lane; tests/test_kernel_resources.py fails when a kernel that a
BASELINE config launches has any."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(lib):
    """{demangled kernel name: {vgpr, agpr, sgpr, scratch, lds, vgpr_spill, sgpr_spill}} for the gfx950 code objects of `lib`."""
    out = {}
    with tempfile.TemporaryDirectory(prefix="vp_co_") as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)                                   # llvm-objdump --offloading extracts next to its input
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
                blk = ".agpr_count:" + blk
                def num(key):
                    m = re.search(r"\." + key + r":\s*(\d+)", blk)
                    return int(m.group(1)) if m else 0
                m = re.search(r"\.name:\s*(\S+)", blk)
                if not m:
                    continue
                out[m.group(1)] = dict(vgpr=num("vgpr_count"), agpr=num("agpr_count"), sgpr=num("sgpr_count"), scratch=num("private_segment_fixed_size"),
                                       lds=num("group_segment_fixed_size"), vgpr_spill=num("vgpr_spill_count"), sgpr_spill=num("sgpr_spill_count"))
    names = list(out)
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    dem = subprocess.run([filt] + names, capture_output=True, text=True).stdout.split("\n") if (names and filt) else names
    res = {}
    for n, dn in zip(names, dem):
        short = re.sub(r"\(.*", "", dn.replace("void ", "")).strip() or n
        res[short] = out[n]
    return res


def fp64_op_counts(lib, kernel):
    """{opcode family: static count} of the fp64 vector arithmetic in `kernel` (demangled name as kernel_resources() prints it),
    from a disassembly of the code object in `lib`.  For a kernel whose loop body is straight-line (vp_k_stft_fused<false, false>:
    one pass per frame) this is what a wavefront executes per pass."""
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    with tempfile.TemporaryDirectory(prefix="vp_co_") as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            for m in re.finditer(r"^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", txt, re.S | re.M):
                dn = subprocess.run([filt, m.group(1)], capture_output=True, text=True).stdout.strip() if filt else m.group(1)
                short = re.sub(r"\(.*", "", dn.replace("void ", "")).strip()
                if short != kernel:
                    continue
                ops = re.findall(r"\b(v_(?:add|mul|fma|fmac|max|min)_f64)", m.group(2))
                out = {}
                for o in ops:
                    out[o] = out.get(o, 0) + 1
                return out
    return None


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(ROOT, "vocoderproject_amd", "libvp_amd.so")
    res = kernel_resources(lib)
    md = "--md" in sys.argv
    hdr = ("kernel", "VGPRs", "AGPRs", "SGPRs", "scratch B/lane", "static LDS", "VGPR spills", "SGPR spills")
    if md:
        print("| " + " | ".join(hdr) + " |\n|" + "---|" * len(hdr))
    else:
        print("%-44s %6s %6s %6s %15s %11s %12s %12s" % hdr)
    for k in sorted(res):
        r = res[k]
        row = (k, r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["lds"], r["vgpr_spill"], r["sgpr_spill"])
        print(("| " + " | ".join(str(x) for x in row) + " |") if md else "%-44s %6d %6d %6d %15d %11d %12d %12d" % row)


if __name__ == "__main__":
    main()
