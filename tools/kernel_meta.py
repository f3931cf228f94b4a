#!/usr/bin/env python3
"""Development helper: per-kernel code-object metadata of a built library (VGPRs, spills, scratch) and a hash of every
kernel's disassembly -- `python tools/kernel_meta.py [lib.so] [--save file]`, `--diff file` compares with a saved listing
(used to check that a source change leaves the other instantiations of em_logl byte-identical)."""
import hashlib
import json
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def kernel_table(lib):
    data = open(lib, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF", data)]
    out = {}
    notes, dis = "", ""
    for start in starts[1:]:          # one embedded code object per translation unit (the first ELF is the host library itself)
        with tempfile.NamedTemporaryFile(suffix=".co") as tmp:
            tmp.write(data[start:])
            tmp.flush()
            n = subprocess.run([LLVM + "llvm-readelf", "--notes", tmp.name], capture_output=True, text=True).stdout
            if "amdhsa.kernels" not in n:
                continue
            notes += n
            dis += subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", tmp.name], capture_output=True, text=True).stdout
    name = None
    for line in notes.splitlines():
        m = re.search(r"\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
        for key in ("private_segment_fixed_size", "vgpr_spill_count", "vgpr_count", "sgpr_count"):
            m = re.search(r"\." + key + r":\s+(\d+)", line)
            if m and name:
                out[name][key] = int(m.group(1))
    cur, body = None, []

    def flush():
        if cur in out:
            # (branch targets are printed as absolute offsets: drop them, they move with the position in the code object)
            # (and pc-relative literals -- s_getpc_b64 + s_add_u32 sN, sN, 0xffxxxxxx -- which move with the layout of the unit, and
            #  the padding behind the last instruction: a kernel's code is the same whatever else its translation unit holds)
            while body and body[-1].split()[0] in ("s_nop", "s_code_end"):
                body.pop()
            text = "\n".join(re.sub(r"^(s_add_u32 s\d+, s\d+,) 0xff[0-9a-f]{6}$", r"\1 <pcrel>",
                                    re.sub(r"^(s_c?branch\S*)\s+\S+", r"\1", re.sub(r"<[^>]*>", "", ln).split("//")[0].rstrip())) for ln in body)
            out[cur]["insts"] = len(body)
            out[cur]["sha"] = hashlib.sha1(text.encode()).hexdigest()[:12]
    for line in dis.splitlines():
        m = re.match(r"^<(\S+)>:$", line.strip())
        if m:
            flush()
            cur, body = m.group(1), []
        elif line.strip():
            body.append(line.strip())
    flush()
    return out


def main():
    argv = sys.argv[1:]
    args = [a for i, a in enumerate(argv) if not a.startswith("--") and (i == 0 or argv[i - 1] not in ("--save", "--diff"))]
    lib = args[0] if args else "nmma_amd/libnmma_hip.so"
    tab = kernel_table(lib)
    if "--save" in sys.argv:
        json.dump(tab, open(sys.argv[sys.argv.index("--save") + 1], "w"), indent=1)
    old = json.load(open(sys.argv[sys.argv.index("--diff") + 1])) if "--diff" in sys.argv else None
    for k in sorted(tab):
        v = tab[k]
        short = re.sub(r"^_ZN4nmma", "", k)[:48]
        mark = ""
        if old is not None:
            mark = "  NEW" if k not in old else ("" if old[k].get("sha") == v.get("sha") else "  CHANGED (was %d insts, %d vgprs, %d B scratch)" % (
                old[k].get("insts", -1), old[k].get("vgpr_count", -1), old[k].get("private_segment_fixed_size", -1)))
        print("%-50s vgpr %3d spill %4d scratch %5d insts %6d %s%s" % (short, v.get("vgpr_count", -1), v.get("vgpr_spill_count", -1),
                                                                     v.get("private_segment_fixed_size", -1), v.get("insts", -1), v.get("sha", "?"), mark))


if __name__ == "__main__":
    main()
