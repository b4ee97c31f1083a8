"""Tabulate hipcc -Rpass-analysis=kernel-resource-usage output for libplship's kernels."""
import re
import subprocess
import sys

csrc = sys.argv[1] if len(sys.argv) > 1 else "projected-langevin-sampling_amd/csrc"
out = subprocess.run(["make", "-C", csrc, "resource-usage"], capture_output=True, text=True)
txt = out.stdout + out.stderr
rows, cur = [], None
for line in txt.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark: (.*?) \[-Rpass", line)
    if not m:
        m = re.search(r":\s+(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize.*|Occupancy.*|SGPRs Spill|VGPRs Spill|LDS Size.*):\s*(\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
    else:
        kv = m.group(1).split(":")
        key, val = kv[0].strip(), kv[-1].strip()
    if key in ("Function Name", "Name"):
        cur = {"name": val}
        rows.append(cur)
    elif cur is not None:
        cur[key] = val
def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n
print(f"{'kernel':100s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'occ':>4s} {'spillV':>6s} {'scratch':>7s}")
for r in rows:
    name = demangle(r["name"])
    name = re.sub(r"plship::", "", name)[:100]
    occ = next((v for k, v in r.items() if k.startswith("Occupancy")), "?")
    scr = next((v for k, v in r.items() if k.startswith("ScratchSize")), "?")
    print(f"{name:100s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {occ:>4s} {r.get('VGPRs Spill','?'):>6s} {scr:>7s}")
