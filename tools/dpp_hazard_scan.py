#!/usr/bin/env python3
"""Static scan of the gfx950 ISA for the one hazard the compiler does not see inside inline asm: a VALU instruction that
writes a VGPR followed within two wait states by a DPP instruction that READS that VGPR through its DPP operand (src0).
gfx9: "VALU writes VGPR -> DPP reads that VGPR: 2 wait states"; every instruction issued in between is one wait state,
`s_nop N` is N + 1.  Also reports VALU writes of EXEC within five wait states of a DPP instruction.

    python tools/dpp_hazard_scan.py            -> the lean kernels (all instantiations the library launches)
Exit code 1 when a hazard is found.  (tests/test_host_abi.py runs it: no GPU needed, ~20 s.)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "spgemm/common.h"
#include "spgemm/lean.h"
namespace nsp { namespace spgemm {
#define NUM1(BS, T, U, F) template __global__ void k_num_lean<BS, T, U, F>(const int *, const int *, const real *, const int *, const int *, const real *, const int *, int *, real *, const int *, const int *, const int *, int, int, int, int);
#define SYM1(BS, T, U, F) template __global__ void k_sym_lean<BS, T, U, F>(const int *, const int *, const int *, const int *, const int *, const int *, const int *, int *, int, int, int, BinState *, int *, long long *, const int *, int, int);
// all four forms (FORM bit 0: branch-free retry rounds, bit 1: pipelined walk): they share the inline-asm sorts
#define NUM(BS, T, U) NUM1(BS, T, U, 0) NUM1(BS, T, U, 1) NUM1(BS, T, U, 2) NUM1(BS, T, U, 3)
#define SYM(BS, T, U) SYM1(BS, T, U, 0) SYM1(BS, T, U, 1) SYM1(BS, T, U, 2) SYM1(BS, T, U, 3)
NUM(64, 256, 2) NUM(256, 1024, 2) NUM(512, 4096, 4) NUM(1024, 8192, 4)
SYM(64, 1024, 2) SYM(128, 2048, 2) SYM(512, 8192, 4) SYM(1024, 32768, 4)
}}
'''
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in VREG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(asm_text):
    hazards, n_dpp = [], 0
    kernel = "?"
    window = []  # (wait states this instruction is worth, set of VGPRs it writes as VALU, writes exec, text)
    for raw in asm_text.splitlines():
        ln = raw.split(";")[0].strip()
        if not ln:
            continue
        m = re.match(r"^(_ZN3nsp\S+):", ln)
        if m:
            kernel, window = m.group(1)[:60], []
            continue
        if ln.endswith(":") or ln.startswith("."):
            if ln.endswith(":"):
                window = []  # a branch target: what ran before is unknown -- (the asm blocks carry their own s_nop)
            continue
        op, _, rest = ln.partition(" ")
        ops = [t.strip() for t in rest.split(",")]
        if op == "s_nop":
            window.append((int(ops[0], 0) + 1, set(), False, ln))
            continue
        is_valu = op.startswith("v_")
        if "_dpp" in op or " quad_perm:" in rest or " row_" in rest or " wave_" in rest:
            n_dpp += 1
            # v_xxx_dpp vdst, src0(dpp), [src1] ...: the DPP operand is the first source
            src0 = regs(ops[1].split(" ")[0]) if len(ops) > 1 else set()
            ws = 0
            for w, wr, ex, text in reversed(window):
                if ws < 2 and wr & src0:
                    hazards.append((kernel, text, ln, ws))
                if ws < 5 and ex:
                    hazards.append((kernel, text, ln, ws))
                ws += w
                if ws >= 5:
                    break
        writes = regs(ops[0]) if is_valu and ops and not op.startswith(("v_cmp", "v_cmpx")) else set()
        writes_exec = is_valu and (op.startswith("v_cmpx") or (ops and ops[0].strip() in ("exec", "exec_lo", "exec_hi")))
        window.append((1, writes, writes_exec, ln))
        window = window[-8:]
    return hazards, n_dpp


def main():
    td = tempfile.mkdtemp(prefix="dpp_")
    src, asm = os.path.join(td, "k.hip"), os.path.join(td, "k.s")
    open(src, "w").write(SRC)
    csrc = sys.argv[sys.argv.index("--csrc") + 1] if "--csrc" in sys.argv else os.path.join(ROOT, "nsparse_amd", "csrc")
    total_h, total_d = [], 0
    for prec in ("-DDOUBLE", "-DFLOAT"):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", '-DNSPARSE_SRC_HASH="x"',
                               "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"), "-I" + csrc, prec,
                               "--cuda-device-only", "-S", src, "-o", asm], stderr=subprocess.DEVNULL)
        h, d = scan(open(asm).read())
        total_h += h
        total_d += d
    print(f"dpp_hazard_scan: {total_d} DPP instructions in the lean kernels (8 kernels x 4 forms x 2 precisions), {len(total_h)} hazards")
    for k, w, r, ws in total_h[:20]:
        print(f"  {k}: `{w}` then `{r}` after {ws} wait state(s)")
    return 1 if total_h else 0


if __name__ == "__main__":
    sys.exit(main())
