#!/usr/bin/env python3
"""Development helper: the loops of one kernel of the built library (backward branches) with their instruction mix --
`python tools/loop_census.py <mangled-name-fragment> [min_fma]`."""
import collections
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
key = sys.argv[1]
min_fma = int(sys.argv[2]) if len(sys.argv) > 2 else 20
data = open("/root/repo/nmma_amd/libnmma_hip.so", "rb").read()
for st in [m.start() for m in re.finditer(b"\x7fELF", data)][1:]:
    with tempfile.NamedTemporaryFile(suffix=".co") as tmp:
        tmp.write(data[st:])
        tmp.flush()
        d = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", tmp.name], capture_output=True, text=True).stdout
    if key not in d:
        continue
    i = d.index("<" + key) if ("<" + key) in d else d.index(key)
    j = d.index("\n\n", i + 10)
    ins = []
    for line in d[i:j].splitlines()[1:]:
        m = re.match(r"\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m:
            ins.append((int(m.group(2), 16), m.group(1)))
    base = ins[0][0]
    idx = {a: k for k, (a, _) in enumerate(ins)}
    for k, (a, t) in enumerate(ins):
        if t.startswith(("s_cbranch", "s_branch")):
            full = d[i:j].splitlines()[1:][k]
            m = re.search(r"\+0x([0-9a-f]+)>", full)
            if m and base + int(m.group(1), 16) in idx and idx[base + int(m.group(1), 16)] < k:
                a0 = idx[base + int(m.group(1), 16)]
                body = [x.split()[0] for _, x in ins[a0:k + 1]]
                nf = sum(1 for x in body if x.startswith(("v_fma_f64", "v_mul_f64")))
                if nf >= min_fma:
                    c = collections.Counter()
                    for x in body:
                        c["valu" if x.startswith("v_") else "lds" if x.startswith("ds_") else "wait" if x.startswith("s_waitcnt") else
                          "salu" if x.startswith("s_") else "other"] += 1
                    top = collections.Counter(x for x in body if x.startswith("v_")).most_common(14)
                    print(f"loop [{a0}, {k}] {k - a0 + 1} instructions, {nf} f64 fma/mul: {dict(c)}")
                    print("    ", top)
    break
