#!/usr/bin/env python3
"""Instruction mix per kernel of a device-only assembly dump (hipcc --cuda-device-only -S):  tools/isa_mix.py k.s [filter]"""
import collections
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\S+):\s*; @\S+\n(.*?)s_endpgm', text, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(HIP_vector.*', '', name).replace('sdsp_hip::(anonymous namespace)::', '')
    if flt not in name:
        continue
    c = collections.Counter()
    for line in m.group(2).split('\n'):
        line = line.strip()
        if not line or line[0] in '.;/' or line.endswith(':'):
            continue
        op = line.split()[0]
        if op.startswith('v_'):
            c['valu'] += 1
            if re.match(r'v_(fma|mul|add|sub|fmac|mac|pk_)\w*_f(32|64)', op):
                c['v_fp'] += 1
            elif op.startswith('v_mov') or op.startswith('v_accvgpr'):
                c['v_mov'] += 1
            else:
                c['v_other'] += 1
        elif op.startswith('ds_'):
            c[op] += 1
        elif op.startswith(('global_', 'buffer_', 'scratch_')):
            c[op] += 1
        elif op.startswith('s_barrier'):
            c['s_barrier'] += 1
        elif op.startswith('s_waitcnt'):
            c['s_waitcnt'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
    print(name, dict(sorted(c.items())))
