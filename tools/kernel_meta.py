#!/usr/bin/env python3
"""Register / LDS / spill numbers of the gfx950 kernels from the code object's own metadata (hipcc -S: the
.amdhsa_* directives and the amdhsa.kernels YAML) -- what rocprofv3's trace columns do not report faithfully.

    python tools/kernel_meta.py [source.hip ...] > profiles/r03_kernel_meta.txt

Dynamic LDS is requested at launch (csrc: LDS_TOTAL / LDS_TOTAL_COMP / LDS_BYTES) and is printed from the
source constants the launchers use."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerf-simple_amd", "csrc")
DEFAULT = [("mlp_bf16_16.hip", []), ("mlp_bf16_16.hip", ["-DNERF_HALF"]), ("mlp_f32.hip", []), ("mlp_bwd_16.hip", []),
           ("dw_gemm.hip", []), ("dw_gemm_f8.hip", []), ("composite.hip", [])]
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".group_segment_fixed_size",
          ".private_segment_fixed_size", ".max_flat_workgroup_size")


def assemble(src, defs):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++20", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", "-",
           os.path.join(CSRC, src)] + defs
    return subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()


def demangle(names):
    out = subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE).stdout.decode().split("\n")
    return dict(zip(names, out))


def kernels(asm):
    # the amdhsa.kernels metadata: one YAML map per kernel, '- .agpr_count: ...' ... '.name: _Z...'
    meta = asm[asm.index("amdhsa.kernels:"):] if "amdhsa.kernels:" in asm else ""
    for block in re.split(r"\n  - ", meta)[1:]:
        d = {}
        for line in block.split("\n"):
            m = re.match(r"\s*(\.[a-z_]+):\s+(\S+)\s*$", line)
            if m and (m.group(1) in FIELDS or m.group(1) == ".name"):
                d[m.group(1)] = m.group(2)
        if ".name" in d:
            yield d


def lane_moves(asm, mangled):
    """SGPR spills become v_writelane_b32 / v_readlane_b32 on the vector ALU: how many sit INSIDE the kernel's MFMA stream
    (first to last v_mfma: the layer loop of a tile) and how many outside it (tile prologue, compositing, kernel prologue)."""
    try:
        a = asm.index(mangled + ":")
        body = asm[a:asm.index(".Lfunc_end", a)].split("\n")
    except ValueError:
        return None
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    if not mf:
        return None
    inside = lambda i: mf[0] < i < mf[-1]                    # noqa: E731
    rd = [i for i, l in enumerate(body) if "v_readlane_b32" in l]
    wr = [i for i, l in enumerate(body) if "v_writelane_b32" in l]
    return len(mf), sum(map(inside, rd)), len(rd), sum(map(inside, wr)), len(wr)


def main():
    srcs = [(a, []) for a in sys.argv[1:]] or DEFAULT
    for src, defs in srcs:
        asm = assemble(src, defs)
        ks = list(kernels(asm))
        names = demangle([k[".name"] for k in ks])
        print(f"== {src} {' '.join(defs)}")
        for k in ks:
            nm = names.get(k[".name"], k[".name"]).replace("(anonymous namespace)::", "")
            print(f"  {nm}")
            print("     " + "  ".join(f"{f[1:]}={k.get(f, '?')}" for f in FIELDS))
            lm = lane_moves(asm, k[".name"])
            if lm and (lm[2] or lm[4]):
                print(f"     lane moves (SGPR spill traffic): v_readlane {lm[2]} ({lm[1]} between the first and the last of the "
                      f"{lm[0]} MFMAs), v_writelane {lm[4]} ({lm[3]} between them)")
        consts = re.findall(r"constexpr int (LDS_TOTAL(?:_COMP)?|LDS_BYTES)\b", open(os.path.join(CSRC, src)).read())
        if consts:
            print(f"     dynamic LDS is added at launch: see {', '.join(sorted(set(consts)))} in {src}")
    print("(group_segment_fixed_size = static LDS only; the MLP kernels request 138 KiB / 158 KiB (fused render) dynamically)")


if __name__ == "__main__":
    main()
