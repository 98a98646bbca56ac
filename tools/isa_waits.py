#!/usr/bin/env python3
"""Where does a kernel wait for memory?  Compiles the library's device code for gfx950 with -save-temps (no GPU
needed), cuts one kernel out of the assembly and prints its vector-memory requests and `s_waitcnt vmcnt(N)` in program
order, with the registers the kernel takes.  What to look for: a `vmcnt(0)` right behind a group
of requests that were meant to stay in flight - hipcc waits for a load where its value is first read (a sum, a select),
where a value that exists on one path only meets the other path (a copy), and in front of a loop that holds loads.

    python tools/isa_waits.py k_bc_emit_tileILb1ELi10E        # mangled-name fragment
    python tools/isa_waits.py k_stream_lines --all            # every instruction class, not only memory
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    frag, show_all = sys.argv[1], "--all" in sys.argv[2:]
    src = os.path.join(REPO, "fastq_utils_amd", "csrc", "fqg_abi.hip")
    with tempfile.TemporaryDirectory() as d:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only",
                        "-save-temps=obj", "-c", src, "-o", os.path.join(d, "x.o")], check=True, cwd=d,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(d) if f.endswith(".s") and "amdgcn" in f]
        text = open(os.path.join(d, asm[0])).read()
    names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", text, re.M) if frag in m.group(1)]
    if not names:
        sys.exit(f"no kernel matches {frag!r}")
    for name in names:
        body = text[text.index(f"\n{name}:"):]
        body = body[:body.index("s_endpgm")]
        regs = {k: re.search(r"\." + k + r":\s+(\d+)", text[text.index(".name:           " + name):][:1500]) for k in
                ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count")}
        print(f"== {name}")
        print("   " + ", ".join(f"{k} {v.group(1)}" for k, v in regs.items() if v))
        mem = re.compile(r"^\s+(global_load\w*\b|global_store\w*\b|global_atomic\w*\b|buffer_\w+\b|flat_\w+\b|s_waitcnt .*vmcnt\(\d+\)|s_cbranch\w*\b|s_branch\b)", re.M)
        last = None
        for i, line in enumerate(body.splitlines()):
            t = line.strip()
            if not t or t.startswith(";"):
                continue
            if t.startswith(".LBB"):
                label = t.split(":")[0]
                last = label
                continue
            m = mem.match(line)
            if m and (show_all or not t.startswith(("s_cbranch", "s_branch"))):
                if last:
                    print(f"  {last}:")
                    last = None
                print(f"  {i:6d}  {t.split(';')[0].strip()[:90]}")
        print()


if __name__ == "__main__":
    main()
