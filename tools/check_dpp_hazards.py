#!/usr/bin/env python3
"""Static check of the hand-placed DPP instructions of csrc/hamming.hip (the row-broadcast xor of the mAP scans).

gfx9 rule: a VGPR written by a VALU instruction must not be read as the DPP source operand within the next two wait states (the
hardware does not interlock).  The compiler keeps the rule for DPP instructions it emits itself; the ones inside inline asm are
invisible to it, and stay safe only because their DPP sources are registers written by loads (see `xor_bcnt2_row_bcast`).  This
script disassembles the built object and verifies that for EVERY DPP instruction of every kernel:
    python tools/check_dpp_hazards.py [path/to/hamming.o]
Exit status 0 = no hazard; prints the count of DPP instructions checked."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(obj):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "k.o")
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "-d", "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not dev:
            raise RuntimeError("no device code object found in " + obj)
        text = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, dev[0])], stdout=subprocess.PIPE, check=True).stdout.decode()
    ins = []
    for line in text.splitlines():
        line = line.split("//")[0].strip()
        if not line or line.endswith(":") or re.match(r"^[0-9a-f]+ <", line) or not re.match(r"^[sv]_|^ds_|^global_|^buffer_|^scratch_|^flat_", line):
            continue
        ins.append(line)
    total, bad = 0, []
    for i, l in enumerate(ins):
        if "_dpp" not in l:
            continue
        total += 1
        ops = l.split(None, 1)[1].split(",")
        src0 = regs(ops[1])
        ws, j = 0, i - 1
        while j >= 0 and ws < 2:
            p = ins[j]
            if p.startswith("s_nop"):
                ws += int(p.split()[1]) + 1
                j -= 1
                continue
            op = p.split()[0]
            if op.startswith("v_") and not op.startswith("v_cmp") and len(p.split(None, 1)) > 1:
                if regs(p.split(None, 1)[1].split(",")[0]) & src0:
                    bad.append((p, l))
            ws += 1
            j -= 1
    return total, bad


if __name__ == "__main__":
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "concepthash_amd", "csrc", "build", "hamming.o")
    total, bad = check(obj)
    for p, l in bad:
        print("HAZARD:", p, "->", l)
    print(f"{total} DPP instructions checked, {len(bad)} hazards")
    sys.exit(1 if bad else 0)
