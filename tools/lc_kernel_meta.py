#!/usr/bin/env python3
"""Development helper: instruction mix and register use of the em_lc_loglike instantiations of a built library."""
import collections
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
lib = sys.argv[1] if len(sys.argv) > 1 else "nmma_amd/libnmma_hip.so"
pat = sys.argv[2] if len(sys.argv) > 2 else "em_lc_loglike"
data = open(lib, "rb").read()
for st in [m.start() for m in re.finditer(b"\x7fELF", data)][1:]:
    with tempfile.NamedTemporaryFile(suffix=".co") as tmp:
        tmp.write(data[st:])
        tmp.flush()
        dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", tmp.name], capture_output=True, text=True).stdout
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", tmp.name], capture_output=True, text=True).stdout
    if pat not in dis:
        continue
    cur, cnt = None, collections.defaultdict(collections.Counter)
    for line in dis.splitlines():
        m = re.match(r"^<(\S+)>:$", line.strip())
        if m:
            cur = m.group(1)
            continue
        if cur and pat in cur and line.strip():
            op = line.split()[0]
            for pre in ("flat_load", "flat_store", "global_load", "global_store", "scratch_", "ds_read", "ds_write", "v_rcp_f64", "v_div_fmas_f64", "v_log_f32", "v_exp_f32"):
                if op.startswith(pre):
                    cnt[cur][pre] += 1
            cnt[cur]["total"] += 1
    meta, name = collections.defaultdict(dict), None
    for line in notes.splitlines():
        m = re.search(r"\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        for key in ("vgpr_spill_count", "vgpr_count", "sgpr_count", "private_segment_fixed_size"):
            m = re.search(r"\." + key + r":\s+(\d+)", line)
            if m and name and pat in name:
                meta[name][key] = int(m.group(1))
    for k, v in cnt.items():
        demangled = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(demangled[:70].ljust(72), meta.get(k, {}), dict(v))
