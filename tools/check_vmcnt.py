"""Check the counted vmcnt waits of the MLP kernels against the generated ISA.

chunk_barrier<N> (csrc/nerf_device.h) waits for vmcnt(N) before each chunk's s_barrier,
claiming that exactly N vector-memory instructions follow the chunk's LDS-DMA pieces.  If
fewer followed, the wait would not cover the DMA (a race on the weight buffer); more only
costs time.  This walks the gfx950 assembly of a kernel: for every inline-asm
`s_waitcnt vmcnt(N)` directly in front of an `s_barrier`, the vector-memory instructions
between the preceding `... lds` DMA instruction and the wait are counted and compared with N.

usage: python tools/check_vmcnt.py file.hip [more.hip ...]   (needs hipcc; no GPU)
"""
import os
import re
import subprocess
import sys
import tempfile

VMEM = re.compile(r"^\s*(buffer|global|flat|scratch)_(load|store|atomic)\w*\s")
DMA = re.compile(r"^\s*buffer_load_dword\w*\s.*\blds\s*$")
WAIT = re.compile(r"^\s*s_waitcnt vmcnt\((\d+)\)\s*$")


def kernels_of(asm_text):
    """{kernel symbol: [instruction lines]} for every .amdhsa kernel body in the file."""
    out, cur, name = {}, None, None
    for line in asm_text.splitlines():
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            name, cur = m.group(1), []
            out[name] = cur
        elif cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                cur = None
    return {k: v for k, v in out.items() if any("s_endpgm" in l for l in v)}


def check_kernel(lines):
    """-> (counted waits checked, list of (line number, N, counted)) mismatches."""
    checked, bad = 0, []
    since_dma, seen_dma, in_asm = 0, False, False
    pending = None                      # (line number, N, count) of an asm wait awaiting its s_barrier
    for i, line in enumerate(lines):
        text = line.split(";")[0].rstrip()
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        if not text.strip():
            continue
        if DMA.match(text):
            seen_dma, since_dma, pending = True, 0, None
            continue
        if VMEM.match(text):
            since_dma += 1
            pending = None
            continue
        w = WAIT.match(text)
        if w and in_asm and seen_dma:
            pending = (i, int(w.group(1)), since_dma)
            continue
        if text.strip() == "s_barrier" and pending is not None:
            checked += 1
            if pending[1] != pending[2]:
                bad.append(pending)
            pending = None
    return checked, bad


def assemble(src, extra=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "--cuda-device-only", "-S",
               *extra, src, "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        return open(out).read()


def check_source(src, extra=()):
    """-> {kernel: (checked, mismatches)}"""
    return {k: check_kernel(v) for k, v in kernels_of(assemble(src, extra)).items()}


if __name__ == "__main__":
    rc = 0
    for src in sys.argv[1:]:
        for k, (n, bad) in check_source(src).items():
            print(f"{os.path.basename(src)} {k[:70]}: {n} counted waits, {len(bad)} mismatches")
            for ln, want, got in bad[:10]:
                print(f"    line {ln}: vmcnt({want}) but {got} vector-memory instructions follow the DMA")
            rc |= bool(bad)
    sys.exit(rc)
