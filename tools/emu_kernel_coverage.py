#!/usr/bin/env python3
"""Which kernels of the library did the tests launch on the CPU emulation?

    EMU_COVERAGE=/tmp/cov.txt bash tools/emu_corpus.sh r05      (every process appends the kernels it launched)
    python tools/emu_kernel_coverage.py /tmp/cov.txt [d|s]       -> launched / never launched, per instantiation

The kernels of the library = the functions named k_* in tests/emu/lib/libnsparse_<prec>.so (kernels are ordinary
functions there; every instantiation the host code can launch is a symbol)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("nsp::spgemm::", "").replace("nsp::spmv::", "").replace("nsp::amb::", "").replace("nsp::", "")


def main():
    cov_file = sys.argv[1]
    prec = sys.argv[2] if len(sys.argv) > 2 else "d"
    lib = os.path.join(ROOT, "tests", "emu", "lib", f"libnsparse_{prec}.so")
    out = subprocess.run(["nm", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    mangled = sorted({ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TtWw"})
    dem = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True, check=True).stdout.splitlines()
    # the function ITSELF is a k_*: "void nsp::spgemm::k_num_tb<64, 256, 256>(int const*, ...)" -- not the launch lambdas and
    # std::function handlers that carry a kernel's name in their template arguments
    kernels = {m: d for m, d in zip(mangled, dem) if re.match(r"^(void )?(nsp::)?(\w+::)*k_\w+(<[^()]*>)?\(", d) and d.endswith(")")}
    launched = {ln.strip() for ln in open(cov_file) if ln.strip()}
    hit = sorted(short(d) for m, d in kernels.items() if m in launched)
    miss = sorted(short(d) for m, d in kernels.items() if m not in launched)
    print(f"# kernel instantiations in libnsparse_{prec}.so: {len(kernels)}; launched by the tests on the emulation: {len(hit)}; never: {len(miss)}")
    print("## never launched")
    for k in miss:
        print("  " + k)
    print("## launched")
    for k in hit:
        print("  " + k)


if __name__ == "__main__":
    main()
