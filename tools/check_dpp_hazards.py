#!/usr/bin/env python3
"""Static checks of the hand-placed instructions of csrc/hamming.hip on the BUILT object (the mAP scans: DPP row-broadcast xor,
inline-asm VMEM ring behind counted `s_waitcnt vmcnt(N)`), over the control-flow graph of every kernel, not the listing order.

    python tools/check_dpp_hazards.py [path/to/hamming.o]          exit status 0 = clean   (rules 1-4)
    python tools/check_dpp_hazards.py --all                        rules 1-4 on hamming.o + rule 5 on every object of the product library

1. DPP read after VALU write (gfx9: 2 wait states, no hardware interlock): no VALU instruction writes a DPP instruction's source
   VGPR within the two wait states in front of it, along ANY path into it (branch targets and loop back-edges included).  The
   compiler keeps the rule for the DPP instructions it emits; the ones inside inline asm are invisible to it and stay safe only
   because their sources are registers written by loads (`xor_bcnt2_row_bcast`).
2. DPP after a VALU write of EXEC (`v_cmpx*`, `v_readlane`-free forms: 5 wait states).
3. VMEM ring ordering: hipcc considers the destination of an inline-asm `global_load_*` valid right after the statement and does
   not count the load in its own `s_waitcnt` bookkeeping.  A forward dataflow over the CFG tracks, for every VGPR that is the
   destination of an outstanding vector-memory load, the least number of younger vector-memory operations on any path (loads,
   stores and atomics share vmcnt and retire in issue order); `s_waitcnt vmcnt(N)` retires the entries with at least N younger
   ones.  ANY instruction that reads or writes such a register before the wait that covers it is reported -- a compiler copy,
   spill or re-use of a ring register, or a compiler wait that counted only its own loads.
4. No scratch traffic (register spills) in any `map_scan_kernel` instantiation.
5. (every object, `--all` / check_pk_opsel) No packed-fp32 VALU instruction (`v_pk_fma_f32`, `v_pk_mul_f32`, `v_pk_add_f32`) whose
   `op_sel` routes the HIGH half of SRC1 into the low lane (`op_sel:[x,1]` / `[x,1,x]`).  Measured on MI355X in round 3 (DESIGN.md
   section 3.10, profiles/r03_pk_opsel_hazard.txt, tools/pk_opsel_repro.py): that spelling returns a wrong LOW lane, intermittently,
   whenever a wave of ANOTHER kernel executes MFMAs on the same SIMD -- first seen as one wrong output column x 16 rows of the LN-fold
   GEMM epilogues under two launch chains, then reproduced stand-alone (a VALU-only victim kernel next to an MFMA-only co-tenant:
   ~1e5 wrong results per 1e12; 0 alone; 0 next to LDS-DMA / ds_read / VALU / global-memory / barrier co-tenants), for all three
   opcodes, with or without op_sel_hi.  SRC0's and SRC2's high half (`op_sel:[1,0,0]`, `[0,0,1]`) and every `op_sel_hi = 0` broadcast
   are exact under the same co-tenant (tests/test_isa_forms_gpu.py).  Which spelling the compiler picks for `vector * pair.hi` follows
   its operand canonicalisation (it flipped when a kernel body moved into an inlined function), so the product sources keep such
   scalars in LOW halves and this check rejects the spelling in whatever the compiler emitted.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load", "flat_store",
        "flat_atomic", "scratch_load", "scratch_store")


def vregs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def first_operand(ins):
    parts = ins.split(None, 1)
    return parts[1].split(",")[0] if len(parts) > 1 else ""


def disassemble(obj):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "k.o")
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "-d", "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        if not dev:
            raise RuntimeError("no device code object found in " + obj)
        # EVERY extracted amdgcn code object (one per offload arch / translation unit), not only the first
        return "\n".join(subprocess.run([OBJDUMP, "-d", os.path.join(tmp, d)], stdout=subprocess.PIPE, check=True).stdout.decode()
                         for d in dev)


def functions(text):
    """-> {name: [(address, instruction text)]}"""
    funcs, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(2), [])
            continue
        if cur is None or "//" not in line:
            continue
        body, _, cmt = line.partition("//")
        body = body.strip()
        am = re.match(r"\s*([0-9A-Fa-f]+):", cmt)
        if body and am and re.match(r"^[sv]_|^ds_|^global_|^buffer_|^scratch_|^flat_", body):
            cur.append((int(am.group(1), 16), body, cmt))
    return funcs


class Kernel:
    def __init__(self, name, rows):
        self.name = name
        self.addr = [r[0] for r in rows]
        self.ins = [r[1] for r in rows]
        index = {a: i for i, a in enumerate(self.addr)}
        n = len(rows)
        self.succ = [[] for _ in range(n)]
        for i, (a, body, cmt) in enumerate(rows):
            op = body.split()[0]
            tgt = None
            if op.startswith("s_cbranch") or op == "s_branch":
                m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", cmt)
                if m:
                    tgt = index.get(rows[0][0] + int(m.group(1), 16))
                else:                                   # raw simm16 (words, relative to the next instruction)
                    simm = int(body.split()[1]) & 0xFFFF
                    simm = simm - 0x10000 if simm >= 0x8000 else simm
                    tgt = index.get(a + 4 + 4 * simm)
                if tgt is not None:
                    self.succ[i].append(tgt)
            if op not in ("s_branch", "s_endpgm", "s_setpc_b64") and i + 1 < n:
                self.succ[i].append(i + 1)
        self.pred = [[] for _ in range(n)]
        for i, ss in enumerate(self.succ):
            for s in ss:
                self.pred[s].append(i)

    # ---- rules 1 and 2: look back `budget` wait states along every path into instruction i --------------------------------------
    def lookback(self, i, budget, hit):
        bad, seen, stack = [], set(), [(p, 0) for p in self.pred[i]]
        while stack:
            j, ws = stack.pop()
            if ws >= budget or (j, ws) in seen:
                continue
            seen.add((j, ws))
            p = self.ins[j]
            if p.startswith("s_nop"):
                cost = int(p.split()[1]) + 1
            else:
                cost = 1
                if hit(p):
                    bad.append(p)
            stack.extend((q, ws + cost) for q in self.pred[j])
        return bad

    def check_dpp(self):
        total, bad = 0, []
        for i, l in enumerate(self.ins):
            if "_dpp" not in l.split()[0] and " row_" not in l and "quad_perm" not in l and "row_newbcast" not in l:
                continue
            total += 1
            ops = l.split(None, 1)[1].split(",")
            src0 = vregs(ops[1]) if len(ops) > 1 else set()

            def writes_src(p):
                return p.startswith("v_") and not p.startswith("v_cmp") and bool(vregs(first_operand(p)) & src0)

            def writes_exec(p):
                return p.startswith("v_cmpx") or (p.startswith("v_") and re.match(r"exec", first_operand(p).strip()) is not None)

            bad += [("valu->dpp", p, l) for p in self.lookback(i, 2, writes_src)]
            bad += [("exec->dpp", p, l) for p in self.lookback(i, 5, writes_exec)]
        return total, bad

    # ---- rule 3: forward dataflow of outstanding VMEM load destinations ---------------------------------------------------------
    def check_vmem_order(self):
        n = len(self.ins)
        if n == 0:
            return 0, []
        leaders = {0} | {s for ss in self.succ for s in ss if True}
        leaders = sorted({0} | {t for i, ss in enumerate(self.succ) for t in ss if t != i + 1} | {i + 1 for i, ss in enumerate(self.succ)
                                                                                                  if len(ss) != 1 or ss[0] != i + 1 if i + 1 < n})
        block_of, blocks = {}, []
        for bi, st in enumerate(leaders):
            en = leaders[bi + 1] if bi + 1 < len(leaders) else n
            blocks.append((st, en))
            block_of[st] = bi
        state_in = [None] * len(blocks)
        state_in[0] = {}
        work, bad, loads = [0], {}, 0

        def transfer(bi, st_in, report):
            nonlocal loads
            pend = dict(st_in)
            st, en = blocks[bi]
            for i in range(st, en):
                l = self.ins[i]
                op = l.split()[0]
                if op == "s_waitcnt":
                    m = re.search(r"vmcnt\((\d+)\)", l)
                    if m:
                        k = int(m.group(1))
                        pend = {r: a for r, a in pend.items() if a < k}
                    elif re.match(r"s_waitcnt\s+\d+$", l):
                        pend = {}
                    continue
                touched = vregs(l.split(None, 1)[1]) if len(l.split(None, 1)) > 1 else set()
                hit = touched & set(pend)
                if hit and report:
                    bad.setdefault(i, (l, sorted(hit)))
                if op.startswith(VMEM):
                    pend = {r: a + 1 for r, a in pend.items() if a + 1 < 64}
                    is_load = "_load" in op or ("_atomic" in op and (" sc0" in l or " glc" in l))
                    if is_load and not re.search(r"\blds\b", l):
                        if report and "_load" in op:
                            loads += 1
                        for r in vregs(first_operand(l)):
                            pend[r] = 0
            return pend

        while work:
            bi = work.pop()
            out = transfer(bi, state_in[bi], False)
            last = blocks[bi][1] - 1
            for s in self.succ[last]:
                sb = block_of[s]
                cur = state_in[sb]
                if cur is None:
                    state_in[sb] = dict(out)
                    work.append(sb)
                else:
                    merged = dict(cur)
                    changed = False
                    for r, a in out.items():
                        if r not in merged or a < merged[r]:
                            merged[r] = a
                            changed = True
                    if changed:
                        state_in[sb] = merged
                        work.append(sb)
        for bi in range(len(blocks)):
            if state_in[bi] is not None:
                transfer(bi, state_in[bi], True)
        return loads, [(self.name, hex(self.addr[i]), l, regs) for i, (l, regs) in sorted(bad.items())]


def check_all(obj):
    """-> dict(dpp_total, dpp_bad, vmem_loads, vmem_bad, scratch_bad, kernels)"""
    funcs = functions(disassemble(obj))
    res = dict(dpp_total=0, dpp_bad=[], vmem_loads=0, vmem_bad=[], scratch_bad=[], kernels=len(funcs))
    for name, rows in funcs.items():
        k = Kernel(name, rows)
        t, b = k.check_dpp()
        res["dpp_total"] += t
        res["dpp_bad"] += [(name,) + x for x in b]
        ld, vb = k.check_vmem_order()
        res["vmem_loads"] += ld
        res["vmem_bad"] += vb
        if "map_scan_kernel" in name:
            res["scratch_bad"] += [(name, l) for l in k.ins if l.startswith("scratch_")]
    return res


PK_F32 = re.compile(r"^(v_pk_(?:fma|mul|add)_f32)\b.*\bop_sel:\[(\d),(\d)(?:,(\d))?\]")


def check_pk_opsel(obj):
    """Rule 5 on one object -> (packed-fp32 instructions seen, [(kernel, address, instruction)] with op_sel set on SRC1)."""
    total, bad = 0, []
    for name, rows in functions(disassemble(obj)).items():
        for addr, ins, _ in rows:
            if not ins.startswith("v_pk_") or "_f32" not in ins.split()[0]:
                continue
            total += 1
            m = PK_F32.match(ins)
            if m and m.group(3) == "1":
                bad.append((name, addr, ins))
    return total, bad


def product_objects():
    """Objects the library on disk was linked from: the product sources, plus the experiment kernels when the last build was an
    experiments build (csrc/build/experiments.flag, written by concepthash_amd.build) -- they are dispatchable there."""
    build = os.path.join(ROOT, "concepthash_amd", "csrc", "build")
    sys.path.insert(0, ROOT)
    from concepthash_amd.build import EXPERIMENT_SOURCES, SOURCES
    stamp = os.path.join(build, "experiments.flag")
    experiments = os.path.exists(stamp) and open(stamp).read().strip() == "1"
    return [os.path.join(build, os.path.splitext(os.path.basename(s))[0] + ".o") for s in SOURCES + (EXPERIMENT_SOURCES if experiments else [])]


def expects_device_code(obj):
    """True when the object's source defines kernels (`__global__`): then a failed extraction is an error, not 'host-only object'."""
    sys.path.insert(0, ROOT)
    from concepthash_amd.build import CSRC, EXPERIMENT_SOURCES, SOURCES
    stem = os.path.splitext(os.path.basename(obj))[0]
    for s in SOURCES + EXPERIMENT_SOURCES:
        if os.path.splitext(os.path.basename(s))[0] == stem:
            return s.endswith(".hip") and "__global__" in open(os.path.join(CSRC, s)).read()   # plain C++ sources have no device code
    return True


def has_device_code(obj):
    try:
        disassemble(obj)
        return True
    except RuntimeError:
        return False


def check(obj):
    """(number of DPP instructions checked, hazards): the round-2 interface, now over the CFG and with the EXEC rule."""
    r = check_all(obj)
    return r["dpp_total"], r["dpp_bad"]


if __name__ == "__main__":
    if "--all" in sys.argv:      # rule 5 over every object of the product library
        n_bad = 0
        for o in product_objects():
            if not has_device_code(o):
                if expects_device_code(o):
                    print(f"{os.path.basename(o)}: NO device code could be extracted although its source defines kernels")
                    n_bad += 1
                continue
            t, b = check_pk_opsel(o)
            for x in b[:10]:
                print("PK OP_SEL:", x)
            print(f"{os.path.basename(o)}: {t} packed-fp32 instructions, {len(b)} with op_sel on SRC1")
            n_bad += len(b)
        r = check_all(os.path.join(ROOT, "concepthash_amd", "csrc", "build", "hamming.o"))      # rules 1-4 on the map scans
        print(f"hamming.o: {r['dpp_total']} DPP instructions, {len(r['dpp_bad'])} hazards; {r['vmem_loads']} ring loads, {len(r['vmem_bad'])} "
              f"uses before the covering wait; {len(r['scratch_bad'])} scratch instructions in the map scans")
        sys.exit(1 if (n_bad or r["dpp_bad"] or r["vmem_bad"] or r["scratch_bad"]) else 0)
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "concepthash_amd", "csrc", "build", "hamming.o")
    r = check_all(obj)
    for x in r["dpp_bad"][:20]:
        print("DPP HAZARD:", x)
    for x in r["vmem_bad"][:20]:
        print("VMEM ORDER:", x)
    for x in r["scratch_bad"][:20]:
        print("SCRATCH:", x)
    print(f"{r['kernels']} kernels; {r['dpp_total']} DPP instructions checked, {len(r['dpp_bad'])} hazards; "
          f"{r['vmem_loads']} vector-memory loads tracked, {len(r['vmem_bad'])} uses before the covering vmcnt wait; "
          f"{len(r['scratch_bad'])} scratch instructions in the map scans")
    sys.exit(1 if (r["dpp_bad"] or r["vmem_bad"] or r["scratch_bad"]) else 0)
