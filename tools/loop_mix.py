#!/usr/bin/env python3
"""Development helper: instruction mix of every loop (backward branch) of one kernel of the built library.
Usage: python tools/loop_mix.py <substring of the mangled kernel name> [lib.so]"""
import re
import subprocess
import sys
import tempfile
from collections import Counter

LLVM = "/opt/rocm/lib/llvm/bin/"
name = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else "nmma_amd/libnmma_hip.so"
data = open(lib, "rb").read()
for st in [m.start() for m in re.finditer(b"\x7fELF", data)][1:]:
    with tempfile.NamedTemporaryFile(suffix=".co") as tmp:
        tmp.write(data[st:]); tmp.flush()
        dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", tmp.name], capture_output=True, text=True).stdout
    blocks = re.split(r"\n(?=[0-9a-f]+ <)", dis)
    for blk in blocks:
        head = blk.split("\n", 1)[0]
        if name not in head:
            continue
        print(head[:120])
        rows = []
        for ln in blk.splitlines()[1:]:
            m = re.match(r"\s*(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
            if m:
                rows.append((int(m.group(3), 16), m.group(1), m.group(2)))
        addr = {a: i for i, (a, _, _) in enumerate(rows)}
        base = rows[0][0]
        for i, (a, op, args) in enumerate(rows):
            if op.startswith("s_cbranch") or op == "s_branch":
                t = re.search(r"\+0x([0-9a-f]+)>", blk.splitlines()[1 + i] if False else "")
        # branch targets: objdump prints "<sym+0xOFF>" in the comment part; re-scan raw lines
        raw = [ln for ln in blk.splitlines()[1:] if re.search(r"//\s*[0-9A-Fa-f]+:", ln)]
        for i, ln in enumerate(raw):
            m = re.match(r"\s*(s_cbranch\S+|s_branch)\s", ln)
            t = re.search(r"\+0x([0-9a-f]+)>", ln)
            if m and t:
                ta = base + int(t.group(1), 16)
                if ta in addr and addr[ta] < i:
                    body = rows[addr[ta]:i + 1]
                    c = Counter(op for _, op, _ in body)
                    valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_readlane") and not k.startswith("v_writelane"))
                    print(f"  loop {addr[ta]}..{i} ({len(body)} instrs): VALU {valu}, SALU/other {len(body) - valu}, global/scratch loads "
                          f"{sum(v for k, v in c.items() if k.startswith(('global_load', 'scratch_load', 'buffer_load', 'flat_load')))}, "
                          f"stores {sum(v for k, v in c.items() if 'store' in k)}, s_load {sum(v for k, v in c.items() if k.startswith('s_load'))}, "
                          f"waitcnt {c.get('s_waitcnt', 0)}")
                    print("     ", ", ".join(f"{k} {v}" for k, v in c.most_common(14)))
