"""bindings/rust/aa_stft_sys.rs against include/aa_stft.h (CPU test, no rustc in this image): the Rust side a maintainer
of audio-analyzer-rs would add (the seam: /root/reference/src/dsp/fft.rs:6-42) must declare every exported function
with the header's argument count and types, every plain struct with the header's field order, and the constants with
the header's values.  The parser below is independent of tools/gen_rust_sys.py (which wrote the file): a bug in the
generator does not cancel out."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "aa_stft.h")
SYS_RS = os.path.join(ROOT, "bindings", "rust", "aa_stft_sys.rs")
FFT_RS = os.path.join(ROOT, "bindings", "rust", "fft.rs")

C_TO_RUST = {"int": "c_int", "int32_t": "i32", "int64_t": "i64", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "float": "f32",
             "double": "f64", "char": "c_char", "void": "c_void"}


def _c_type_to_rust(ctype, array_param=False):
    toks = ctype.replace("*", " * ").split()
    stars = toks.count("*") + (1 if array_param else 0)
    const = "const" in toks
    base = [t for t in toks if t not in ("*", "const", "struct")]
    assert len(base) == 1, ctype
    b = C_TO_RUST.get(base[0], base[0])
    if stars == 0:
        return b
    t = ("*const " if const else "*mut ") + b
    return "*mut " * (stars - 1) + t


def _header_functions():
    text = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    text = re.sub(r"typedef[^;{]*;", " ", text)
    text = re.sub(r"enum\s*\{.*?\}\s*;", " ", text, flags=re.S)
    text = text.replace('extern "C" {', " ")
    out = {}
    for stmt in text.split(";"):
        m = re.search(r"(\w[\w\s\*]*?)\b(aa_\w+)\s*\((.*)\)\s*$", stmt.strip(), re.S)
        if not m:
            continue
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                am = re.match(r"^(.*?)(\w+)((?:\[\w+\])*)$", a.strip())
                params.append(_c_type_to_rust(am.group(1), array_param=bool(am.group(3))))
        out[name] = (_c_type_to_rust(ret), params)
    return out


def _rust_functions():
    text = re.sub(r"//.*$", "", open(SYS_RS).read(), flags=re.M)
    block = re.search(r'extern "C" \{(.*?)\n\}', text, re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (aa_\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block, re.S):
        params = [p.split(":", 1)[1].strip() for p in m.group(2).split(",") if p.strip()]
        out[m.group(1)] = ((m.group(3) or "c_void").strip(), params)
    return out


def test_every_header_function_is_declared_with_the_same_signature():
    c, r = _header_functions(), _rust_functions()
    assert len(c) >= 56, len(c)
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in c:
        assert c[name][0] == r[name][0], (name, "return", c[name][0], r[name][0])
        assert len(c[name][1]) == len(r[name][1]), (name, "argument count", len(c[name][1]), len(r[name][1]))
        assert c[name][1] == r[name][1], (name, c[name][1], r[name][1])


def test_exported_symbols_of_the_library_are_the_declared_ones():
    so = os.path.join(ROOT, "audio_analyzer_rs_amd", "libaa_stft.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    syms = subprocess.check_output(["nm", "-D", "--defined-only", so]).decode().split("\n")
    exported = sorted(s.split()[-1] for s in syms if s.strip() and s.split()[-1].startswith("aa_") and " T " in s)
    assert exported == sorted(_rust_functions())


def test_structs_and_constants_match_the_header():
    htext = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    rtext = re.sub(r"//.*$", "", open(SYS_RS).read(), flags=re.M)
    # plain structs: field names in order, field types, array dimensions
    n_structs = 0
    for m in re.finditer(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", htext, re.S):
        name, want = m.group(2), []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            first, *more = decl.split(",")
            fm = re.match(r"^(.*?)(\w+)((?:\[\w+\])*)$", first.strip())
            ty, dims = _c_type_to_rust(fm.group(1)), re.findall(r"\[(\w+)\]", fm.group(3))
            want.append((fm.group(2), ty, dims))
            want += [(x.strip(), ty, []) for x in more]
        rm = re.search(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct %s \{(.*?)\}" % name, rtext, re.S)
        assert rm, name
        got = [(f.split(":")[0].replace("pub", "").strip(), f.split(":", 1)[1].strip())
               for f in rm.group(1).split(",\n") if ":" in f]
        assert [g[0] for g in got] == [w[0] for w in want], (name, got, want)
        for (fname, rty), (_, ty, dims) in zip(got, want):
            expect = ty
            for d in reversed(dims):
                expect = f"[{expect}; {d}]" if d.isdigit() else f"[{expect}; {d} as usize]"
            assert rty.rstrip(",") == expect, (name, fname, rty, expect)
        n_structs += 1
    assert n_structs >= 3                                   # aa_stft_config, aa_tuner_output, aa_comm_id
    # opaque handles are zero-sized repr(C) structs
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", htext):
        assert re.search(r"#\[repr\(C\)\]\s*pub struct %s \{\s*_private: \[u8; 0\],\s*\}" % m.group(2), rtext), m.group(2)
    # constants: #defines and enum members with their values
    consts = dict(re.findall(r"^#define\s+(AA_\w+)\s+(-?\d+)\s*$", htext, re.M))
    for body in re.findall(r"enum\s*\{(.*?)\}", htext, re.S):
        for item in body.split(","):
            if "=" in item:
                k, v = item.split("=")
                consts[k.strip()] = v.strip()
    assert len(consts) >= 25
    for k, v in consts.items():
        rm = re.search(r"pub const %s: \w+ = (-?\d+);" % k, rtext)
        assert rm and rm.group(1) == v, (k, v, rm and rm.group(1))


def test_generated_file_is_up_to_date():
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"]) == 0, \
        "bindings/rust/aa_stft_sys.rs is stale: run python tools/gen_rust_sys.py"


def test_wrapper_calls_exist_with_the_right_argument_count():
    """bindings/rust/fft.rs (the FftProcessor-shaped wrapper over the raw declarations): every aa_* call names a
    declared function and passes as many arguments as the header takes."""
    decl = _header_functions()
    text = re.sub(r"//.*$", "", open(FFT_RS).read(), flags=re.M)
    calls = 0
    for m in re.finditer(r"\b(aa_\w+)\s*\(", text):
        name = m.group(1)
        if name not in decl:
            assert name in ("aa_sys",) or re.search(r"struct %s\b|\*mut %s\b" % (name, name), text), name
            continue
        depth, i, args, cur = 1, m.end(), [], ""
        while depth:
            ch = text[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
                if depth == 0:
                    break
            if ch == "," and depth == 1:
                args.append(cur)
                cur = ""
            else:
                cur += ch
            i += 1
        if cur.strip():
            args.append(cur)
        assert len(args) == len(decl[name][1]), (name, len(args), len(decl[name][1]))
        calls += 1
    assert calls >= 12
    for fn in ("pub fn new(len: usize) -> Self", "pub fn process_forward(&mut self, windowed: &mut [f32]) -> &[Complex<f32>]",
               "pub fn process_inverse(&mut self, spec: &mut [Complex<f32>]) -> &[f32]"):
        assert fn in text, fn                                  # the reference's signatures, fft.rs:14, 33, 39
