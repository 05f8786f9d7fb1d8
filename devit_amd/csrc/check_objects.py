#!/usr/bin/env python3
"""Build gate of csrc/build.sh over the gfx950 code objects (it fails the build, it is not a test):
  1. no kernel of the library spills registers or uses scratch: a scratch reload inside a GEMM K-step is a vector-memory operation that lands in the
     kernel's own counted vmcnt waits;
  2. in the kernels whose K loop is a generated inline-asm statement that leaves its accumulators in literal AGPRs (gemm4_kernel, gemmfr_kernel: the
     compiler is told they are clobbered, not that they are LIVE between the loop and the read-out asm statements), nothing but an MFMA writes an
     a-register: no v_accvgpr_write, no load into an AGPR (advisor r04: a compiler that used AGPRs as spill space there would corrupt the tile).
A missing tool or an object that cannot be taken apart is an ERROR, never a pass; an object is skipped only when it holds no device code at all
(no .hip_fatbin section: api / comm / encoder today -- found by looking, not by name, so device code added to one of them later is gated too).
usage: check_objects.py BUILD_DIR obj..."""
import os
import re
import subprocess
import sys

LLVM = os.environ.get("DEVIT_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ASM_ACC_KERNELS = ("gemm4_kernel", "gemmfr_kernel", "wgradfr_kernel")


def run(*cmd):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        sys.exit(f"check_objects: `{' '.join(cmd)}` failed ({p.returncode}): {p.stderr.strip()[:400]}")
    return p.stdout


def scan_agpr_writes(disassembly):
    """(violations, MFMAs seen) over `llvm-objdump -d` text: inside the asm-accumulator kernels every instruction whose destination is an
    a-register must be an MFMA (tests/test_abi.py feeds it synthetic text)."""
    bad, cur, seen = [], None, 0
    for line in disassembly.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            if not m.group(1).startswith("L_"):          # (labels of the asm statements stay inside their kernel)
                cur = m.group(1)
            continue
        if cur is None or not any(k in cur for k in ASM_ACC_KERNELS):
            continue
        m = re.match(r"^\s+(\S+)\s+([^,\s]+)", line)
        if not m:
            continue
        op, dst = m.group(1), m.group(2)
        if op.startswith("v_accvgpr_write") or (re.match(r"^a(\d+|\[)", dst) and not op.startswith("v_mfma")):
            bad.append(f"{cur}: `{line.strip()[:90]}` writes an AGPR outside an MFMA")
        seen += op.startswith("v_mfma")
    return bad, seen


def main(build, names):
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf", "llvm-objdump"):
        if not os.access(os.path.join(LLVM, tool), os.X_OK):
            sys.exit(f"check_objects: {LLVM}/{tool} is missing: the no-spill / AGPR gates cannot run (set DEVIT_LLVM_BIN)")
    bad = []
    checked = 0
    for f in names:
        obj, fat, co = (os.path.join(build, f + e) for e in (".o", ".fatbin", ".gfx950.co"))
        if not re.search(r"\s\.hip_fatbin\s", run(os.path.join(LLVM, "llvm-readelf"), "-S", "-W", obj)):
            continue                                  # host code only
        checked += 1
        try:
            run(os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj)
            run(os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
                f"--output={co}", "--unbundle")
            notes = run(os.path.join(LLVM, "llvm-readelf"), "--notes", co)
            kernels = re.findall(r"\.name:\s+(\S+)", notes)
            if not kernels:
                sys.exit(f"check_objects: {f}.o: no kernel metadata found (expected device code)")
            name = None
            for line in notes.splitlines():
                m = re.search(r"\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                m = re.search(r"\.vgpr_spill_count:\s+(\d+)", line)
                if m and int(m.group(1)) > 0:
                    bad.append(f"{f}.hip: {name} spills {m.group(1)} VGPRs")
                m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", line)
                if m and int(m.group(1)) > 0:
                    bad.append(f"{f}.hip: {name} uses {m.group(1)} B of scratch")
            if any(any(k in n for k in ASM_ACC_KERNELS) for n in kernels):
                found, seen = scan_agpr_writes(run(os.path.join(LLVM, "llvm-objdump"), "-d", co))
                bad += [f"{f}.hip: {x}" for x in found]
                if seen == 0:
                    sys.exit(f"check_objects: {f}.o: the disassembly of the asm-accumulator kernels shows no MFMA: the gate is not seeing them")
        finally:
            for p in (fat, co):
                if os.path.exists(p):
                    os.remove(p)
    if checked == 0:
        sys.exit("check_objects: none of the objects holds device code: the gate is not seeing the build")
    if bad:
        print("check_objects: build gate failed:")
        print("\n".join(bad[:40]))
        sys.exit(1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
