#!/usr/bin/env python3
"""Scan device assembly for the store-data hazard of profiles/r03_store_hazard.md: a vector-memory store of MORE than 64 bits
directly followed by an instruction that writes one of its data registers, with no wait state (s_nop) in between.
    tools/isa_store_hazard.py csrc/fft_big64.hip [extra hipcc flags]      (compiles with the library's flags, device only)
    tools/isa_store_hazard.py file.s                                        (scans an assembly dump)
Exit status 1 when a pattern is found."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
WIDE_STORE = re.compile(r'^(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)$')


def data_regs(op, rest):
    """register range of the store's data operand"""
    ops = [o.strip() for o in rest.split(',')]
    src = ops[0] if op.startswith('buffer') else ops[1]  # buffer: vdata first; global/flat/scratch: vaddr, vdata
    m = re.match(r'v\[(\d+):(\d+)\]', src)
    return (int(m.group(1)), int(m.group(2))) if m else None


def written_regs(line):
    m = re.match(r'(v_\w+|ds_read\w*|buffer_load\w*|global_load\w*)\s+(v\[(\d+):(\d+)\]|v(\d+))\b', line)
    if not m or line.startswith(('v_cmp', 'v_cmpx')):
        return None
    return (int(m.group(3)), int(m.group(4))) if m.group(3) else (int(m.group(5)), int(m.group(5)))


def scan(text):
    lines = [l.strip() for l in text.splitlines()]
    lines = [l for l in lines if l and not l.startswith((';', '.', '//')) and not l.endswith(':')]
    stores = guarded = 0
    bad = []
    for i, l in enumerate(lines[:-1]):
        m = WIDE_STORE.match(l)
        if not m:
            continue
        rng = data_regs(m.group(1), m.group(2))
        if not rng:
            continue
        stores += 1
        nxt = lines[i + 1]
        if nxt.startswith('s_nop'):
            guarded += 1
            continue
        w = written_regs(nxt)
        if w and not (w[1] < rng[0] or w[0] > rng[1]):
            bad.append((l, nxt))
    return stores, guarded, bad


def main():
    src = Path(sys.argv[1])
    if src.suffix == '.s':
        text = src.read_text()
    else:
        with tempfile.TemporaryDirectory() as tmp:
            out = Path(tmp) / 'k.s'
            cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', f'-I{ROOT / "include"}',
                   f'-I{ROOT / "simpledsp_amd" / "csrc"}', '-fno-slp-vectorize', *sys.argv[2:], '--cuda-device-only', '-S', '-o', str(out), str(src)]
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            text = out.read_text()
    stores, guarded, bad = scan(text)
    print(f"{src.name}: {stores} stores of more than 64 bits; followed by s_nop: {guarded}; unguarded overwrites of store data: {len(bad)}")
    for st, nx in bad[:20]:
        print("   ", st, "  ->  ", nx)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
