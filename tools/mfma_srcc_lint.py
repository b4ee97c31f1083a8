"""Lint the gfx950 ISA of libplship for a write-after-read hazard on the accumulator INPUT of an fp64 MFMA.

`v_mfma_f64_16x16x4_f64 vD, vA, vB, vC` with vD != vC leaves vC dead as far as the register allocator is concerned, so
it may place a copy into vC right behind the MFMA.  On gfx950 the instruction streams its C operand over its 16 passes:
a VALU / load write into vC inside that window corrupts the sum (found in round 3: a k-tail code path of
gemm_tn_f64_kg.h accumulated garbage in one 16 x 16 block; the hazard recogniser of ROCm 7.2's LLVM does not pad this
case for the DGEMM opcodes).  The lint flags every write into the C range between such an MFMA and the next MFMA (or
within WINDOW instructions).  Usage: python tools/mfma_srcc_lint.py file.s [...]   (exit status 1 if anything is flagged)
"""
import re
import sys

WINDOW = 12
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = REG.fullmatch(tok.strip())
    if not m:
        return None
    if m.group(1) is not None:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def dst_regs(line):
    """VGPRs an instruction writes (first operand of VALU / loads that return data); None if it writes none"""
    ins = line.split()
    if not ins:
        return None
    op = ins[0]
    if op.startswith(("s_", "ds_write", "global_store", "buffer_store", "scratch_store", ";", ".")) or op.endswith(":"):
        return None
    if op.startswith("buffer_load") and " lds" in line:
        return None
    rest = line[len(op):].split(",")
    return regs(rest[0]) if rest else None


def lint(path):
    bad = []
    kernel = "?"
    lines = open(path).read().splitlines()
    for n, raw in enumerate(lines):
        line = raw.split(";")[0].strip()
        if raw.startswith("_Z") and raw.rstrip().endswith(":"):
            kernel = raw.rstrip(":")
        if not line.startswith("v_mfma_f64"):
            continue
        ops = line[len(line.split()[0]):].split(",")
        d, c = regs(ops[0]), regs(ops[3].split()[0]) if len(ops) > 3 else None
        if not d or not c or d == c:
            continue
        seen = 0
        for m in range(n + 1, min(len(lines), n + 1 + 4 * WINDOW)):
            nxt = lines[m].split(";")[0].strip()
            if not nxt or nxt.endswith(":"):
                continue
            if nxt.startswith("v_mfma"):
                break
            seen += 1
            if seen > WINDOW:
                break
            w = dst_regs(nxt)
            if w and (w & c):
                bad.append((path, kernel, n + 1, line, m + 1, nxt))
    return bad


if __name__ == "__main__":
    allbad = []
    for p in sys.argv[1:]:
        allbad += lint(p)
    for path, kernel, n, line, m, nxt in allbad:
        print(f"{path}:{n}: {kernel}\n    {line}\n  {m}: {nxt}")
    print(f"{len(allbad)} suspicious write(s) into a live MFMA C operand")
    sys.exit(1 if allbad else 0)
