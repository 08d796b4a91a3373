"""Check the counted vmcnt waits of the MLP kernels against the generated ISA.

chunk_barrier<N> (csrc/nerf_device.h) waits for vmcnt(N) before each chunk's s_barrier,
claiming that exactly N vector-memory instructions follow the chunk's LDS-DMA pieces.  If
fewer followed, the wait would not cover the DMA (a race on the weight buffer); more only
costs time.  This walks the gfx950 assembly of a kernel: for every inline-asm
`s_waitcnt vmcnt(N)` directly in front of an `s_barrier`, the vector-memory instructions
between the preceding `... lds` DMA instruction and the wait are counted and compared with N.

It also scans for the store-data hazard found on MI355X (csrc/nerf_device.h store_granule): a
wide store whose data registers the next instruction overwrites.

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


# data registers: first operand of a buffer store, second of a global store
STORE128 = re.compile(r"^\s*(?:buffer_store_dwordx[34]\s+|global_store_dwordx[34]\s+v(?:\[\d+:\d+\]|\d+),\s*)()v\[(\d+):(\d+)\]")
VALU_DST = re.compile(r"^\s*v_\w+\s+v(?:\[(\d+):(\d+)\]|(\d+))")


def check_store_data_hazard(lines):
    """Wide stores whose data registers are written by the very next instruction (the MI355X
    store-data hazard documented at store_granule in csrc/nerf_device.h).  -> (stores, offenders)"""
    stores, bad = 0, []
    real = [l.split(";")[0].rstrip() for l in lines]
    real = [l for l in real if l.strip() and not l.strip().startswith((".", "#"))]
    for i, text in enumerate(real[:-1]):
        m = STORE128.match(text)
        if not m:
            continue
        stores += 1
        lo, hi = int(m.group(2)), int(m.group(3))
        w = VALU_DST.match(real[i + 1])
        if w:
            d0 = int(w.group(1) if w.group(1) is not None else w.group(3))
            d1 = int(w.group(2)) if w.group(2) is not None else d0
            if d0 <= hi and d1 >= lo:
                bad.append((text.strip(), real[i + 1].strip()))
    return stores, bad


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


def check_source_stores(src, extra=()):
    """-> {kernel: (wide stores, offenders)}"""
    return {k: check_store_data_hazard(v) for k, v in kernels_of(assemble(src, extra)).items()}


if __name__ == "__main__":
    rc = 0
    for src in sys.argv[1:]:
        for k, (n, bad) in check_source(src).items():
            print(f"{os.path.basename(src)} {k[:70]}: {n} counted waits, {len(bad)} mismatches")
            for ln, want, got in bad[:10]:
                print(f"    line {ln}: vmcnt({want}) but {got} vector-memory instructions follow the DMA")
            rc |= bool(bad)
        for k, (n, bad) in check_source_stores(src).items():
            print(f"{os.path.basename(src)} {k[:70]}: {n} wide stores, {len(bad)} with data overwritten by the next instruction")
            rc |= bool(bad)
    sys.exit(rc)
