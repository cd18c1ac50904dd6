"""Static facts of the compiled gfx950 kernels that the hand-counted waits and the occupancy plan rely on, per kernel
instantiation of aa_kernels.hip: number of global_store / global_load_lds / scratch instructions, s_waitcnt vmcnt(N)
values, VGPR count and scratch bytes.  `python tools/kernel_static.py > tests/golden/kernel_static.json` records
them for the compiler at hand (after tools/check_counted_waits.py has shown that build bit-identical to the
full-wait build); tests/test_kernel_static.py compares a fresh compile with the record."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "audio_analyzer_rs_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-fno-strict-aliasing"]      # as csrc/Makefile
WAVE = re.compile(r"stft_wave(1024|2048|_small)(_mel)?_kernel")


def compile_asm(out_s, src="aa_kernels.hip", extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *FLAGS, *extra, "-S", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", "-o", out_s, os.path.join(CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-2000:])
    return r.stderr


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines()


def facts():
    with tempfile.TemporaryDirectory() as td:
        s_path = os.path.join(td, "k.s")
        remarks = compile_asm(s_path)
        text = open(s_path).read()
        # the fused MFCC kernel is a translation unit of its own, with its own scheduler strategy (csrc/Makefile)
        m_path = os.path.join(td, "m.s")
        remarks += compile_asm(m_path, "aa_kernels_mfcc.hip", ("-mllvm", "-amdgpu-sched-strategy=max-ilp"))
        text += open(m_path).read()
    res = {}
    cur = None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
        m = re.search(r"remark:\s+VGPRs: (\d+)", line)
        if m and cur:
            res[cur]["vgprs"] = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and cur:
            res[cur]["scratch_bytes"] = int(m.group(1))
    for name in list(res):
        m = re.search(r"^%s:[^\n]*\n(.*?)s_endpgm" % re.escape(name), text, re.S | re.M)
        body = m.group(1) if m else ""
        res[name]["global_store"] = len(re.findall(r"^\s+global_store_", body, re.M))
        res[name]["global_load_lds"] = len(re.findall(r"^\s+global_load_lds_", body, re.M))
        res[name]["vmcnt_waits"] = sorted(set(int(v) for v in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)))
        # lds_dma_16 writes m0 by hand (s_mov_b32 m0, sN) without telling the compiler: no other instruction of the kernel may
        # read or write it
        res[name]["m0_other_uses"] = len([l for l in body.splitlines() if re.search(r"\bm0\b", l.split(";")[0])
                                          and not re.match(r"\s+s_mov_b32 m0, s\d+", l)])
    names = sorted(n for n in res if WAVE.search(n))
    pretty = demangle(names)
    out = {}
    for n, p in zip(names, pretty):
        p = re.sub(r"\(.*", "", p.replace("void aa::(anonymous namespace)::", ""))
        out[p] = res[n]
    ver = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.splitlines()
    return {"compiler": next((l for l in ver if "clang version" in l), ver[0] if ver else ""), "kernels": out}


if __name__ == "__main__":
    json.dump(facts(), sys.stdout, indent=1, sort_keys=True)
    print()
