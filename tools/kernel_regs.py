#!/usr/bin/env python3
"""Register / scratch / occupancy table of the conv kernels from hipcc's resource-usage remarks.
   hipcc ... -c pn2_conv.hip -Rpass-analysis=kernel-resource-usage 2> remarks.txt ; python tools/kernel_regs.py remarks.txt [name filter]"""
import re, subprocess, sys
def parse(f):
    out, cur = {}, None
    for l in open(f):
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = m.group(1); out[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", l)
        if m and cur:
            out[cur][m.group(1).split()[0]] = int(m.group(2))
    return out
def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, r))
if __name__ == "__main__":
    tabs = [parse(f) for f in sys.argv[1:] if f.endswith(".txt")]
    flt = [a for a in sys.argv[1:] if not a.endswith(".txt")]
    dm = demangle(list(tabs[0]))
    for k, v in tabs[0].items():
        name = re.sub(r"\(anonymous namespace\)::", "", dm[k]).split("(")[0].replace("void ", "")
        if flt and not all(a in name for a in flt): continue
        row = f"{name:64s}" + "".join(f"  | V {t.get(k, {}).get('VGPRs', -1):3d} A {t.get(k, {}).get('AGPRs', -1):3d} scr {t.get(k, {}).get('ScratchSize', -1):4d} occ {t.get(k, {}).get('Occupancy', -1)}" for t in tabs)
        print(row)
