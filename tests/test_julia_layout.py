"""Static check of the Julia side of the boundary: the `struct` mirrors in
julia/HedgehogMC.jl must have the C layout of include/hedgehog_mc.h.  No Julia is needed: the C side
is measured (a generated offsetof / sizeof dump compiled with gcc), the Julia side is computed from
the field types with the C layout rules Julia uses for isbits structs; the ctypes mirror of the
Python host is held to the same dump.  Also: every C entry point the Julia file `ccall`s exists in
the header with that name."""
import ctypes as C
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hedgehog_mc.h")
JULIA = os.path.join(ROOT, "julia", "HedgehogMC.jl")
PAIRS = {"hh_model": "HHModel", "hh_config": "HHConfig", "hh_result": "HHResult",
         "hh_lsm_result": "HHLsmResult"}


def c_structs():
    """{struct: [field names in order]} parsed from the header (comments stripped)."""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(None, 1)[1] if not decl.startswith("const") else decl.split(None, 2)[2]
            for n in names.split(","):
                fields.append(re.sub(r"\[.*\]", "", n).replace("*", "").strip())
        out[m.group(1)] = fields
    return out


def measured_layout(tmp_path):
    structs = c_structs()
    lines = ['#include <stddef.h>', '#include <stdio.h>', f'#include "{HEADER}"', "int main(void) {"]
    for s, fields in structs.items():
        lines.append(f'  printf("{s} sizeof %zu\\n", sizeof({s}));')
        for f in fields:
            lines.append(f'  printf("{s} {f} %zu %zu\\n", offsetof({s}, {f}), sizeof((({s}*)0)->{f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", str(src), "-o", str(exe)], check=True)
    out = {}
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        s, f, *nums = ln.split()
        out.setdefault(s, {"fields": []})
        if f == "sizeof":
            out[s]["size"] = int(nums[0])
        else:
            out[s]["fields"].append((f, int(nums[0]), int(nums[1])))
    return out


JL_TYPES = {"Cdouble": (8, 8), "Float64": (8, 8), "Int32": (4, 4), "UInt32": (4, 4), "Cint": (4, 4),
            "UInt64": (8, 8), "Int64": (8, 8)}


def jl_type(t):
    t = t.strip()
    if t.startswith("Ptr{"):
        return 8, 8
    m = re.fullmatch(r"NTuple\{(\d+),\s*(\w+)\}", t)
    if m:
        sz, al = JL_TYPES[m.group(2)]
        return int(m.group(1)) * sz, al
    return JL_TYPES[t]


def julia_layout():
    src = open(JULIA).read()
    out = {}
    for m in re.finditer(r"^struct (HH\w+)\n(.*?)^end", src, flags=re.S | re.M):
        off, fields, maxal = 0, [], 1
        for decl in re.split(r"[;\n]", m.group(2)):
            decl = decl.split("#")[0].strip()
            if not decl:
                continue
            name, typ = decl.split("::")
            size, al = jl_type(typ)
            off = (off + al - 1) // al * al
            fields.append((name.strip(), off, size))
            off += size
            maxal = max(maxal, al)
        out[m.group(1)] = {"fields": fields, "size": (off + maxal - 1) // maxal * maxal}
    return out


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_julia_struct_mirrors_have_the_c_layout(tmp_path):
    c, jl = measured_layout(tmp_path), julia_layout()
    for cname, jname in PAIRS.items():
        assert jname in jl, f"julia/HedgehogMC.jl has no struct {jname}"
        assert jl[jname]["size"] == c[cname]["size"], (cname, jl[jname]["size"], c[cname]["size"])
        assert jl[jname]["fields"] == c[cname]["fields"], (cname, jl[jname]["fields"], c[cname]["fields"])


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_ctypes_mirrors_have_the_c_layout(tmp_path):
    from hedgehog_jl_amd import _ffi
    c = measured_layout(tmp_path)
    for cname in PAIRS:
        cls = getattr(_ffi, cname)
        assert C.sizeof(cls) == c[cname]["size"]
        got = [(n, getattr(cls, n).offset, getattr(cls, n).size) for n, _ in cls._fields_]
        assert got == c[cname]["fields"], cname


def test_every_ccall_of_the_julia_layer_is_a_declared_entry_point():
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    declared = set(re.findall(r"\b(hh_\w+)\s*\(", hdr))
    called = set(re.findall(r"ccall\(\(:(hh_\w+),", open(JULIA).read()))
    assert called and called <= declared, called - declared
    # what VERDICT r1 #5 asks the Julia layer to bind
    assert {"hh_mc_solve", "hh_mc_accumulate", "hh_mc_finalize", "hh_carr_madan", "hh_mc_solve_basket",
            "hh_lsm_solve", "hh_heston_exact_grid", "hh_abi_version",
            "hh_mgpu_create", "hh_mgpu_solve", "hh_mgpu_destroy", "hh_mgpu_reduce_mode"} <= called
    src = open(JULIA).read()
    for needle in ("function solve_batch_greeks_hip", "BatchGreekProblem{P,L}, ::ForwardAD, pricing_method::MonteCarlo",
                   "function solve_sharded_hip", "replay_layout", "HH_NOISE_REPLAY",
                   "devices = nothing", "function install!(; devices = nothing)"):
        assert needle in src, needle
    abi = re.search(r"#define HH_ABI_VERSION (\d+)", open(HEADER).read()).group(1)
    assert "const HH_ABI_VERSION = " + abi in src


def _split_top(args):
    """'Ref{HHModel}, Ptr{Cdouble}, UInt32' -> the comma-separated items at brace depth 0"""
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch in "{(":
            depth += 1
        elif ch in "})":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def test_every_ccall_signature_matches_the_header_prototype():
    """The Julia layer has never run (no toolchain): a ccall whose argument tuple does not match the C
    prototype — one argument short, a Cint where the header takes a uint64_t, a value where it takes a
    pointer — would corrupt the call silently.  Every `ccall((:hh_x, LIB[]), Ret, (T1, T2, …), …)` of
    julia/HedgehogMC.jl is held against `Ret hh_x(T1, T2, …)` of include/hedgehog_mc.h by arity and by
    the width / kind of each argument."""
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    protos = {n: (r, a) for r, n, a in
              re.findall(r"^\s*((?:const\s+)?[\w\*]+(?:\s*\*)?)\s+(hh_\w+)\s*\(([^;{]*?)\)\s*;", hdr, re.M | re.S)}

    def c_kind(t):
        t = re.sub(r"\s+", " ", t.strip())
        if "*" in t:
            return "ptr"
        base = re.sub(r"\b(const|enum|struct)\b", "", t).split()
        base = base[0] if base else "void"   # the parameter name follows the type
        return {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64",
                "size_t": "u64", "double": "f64", "void": "void"}[base]

    def jl_kind(t):
        t = t.strip()
        if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring",):
            return "ptr"
        return {"Cint": "i32", "Int32": "i32", "UInt32": "u32", "Cuint": "u32", "Int64": "i64", "UInt64": "u64",
                "Csize_t": "u64", "Cdouble": "f64", "Float64": "f64", "Cvoid": "void"}[t]

    src = open(JULIA).read()
    # the argument tuple holds types only — braces, never parentheses
    calls = re.findall(r"ccall\(\(:(hh_\w+),\s*LIB\[\]\),\s*([\w\{\}]+),\s*\(([^()]*)\),\s*\n?\s*[\w\.\(]", src, re.S)
    assert len(calls) >= 20
    for name, ret, args in calls:
        c_ret, c_args = protos[name]
        want = [] if c_args.strip() in ("", "void") else [c_kind(a) for a in _split_top(c_args)]
        got = [jl_kind(a) for a in _split_top(args)]
        assert got == want, (name, got, want)
        assert jl_kind(ret) == c_kind(c_ret + " x"), (name, ret, c_ret)


def test_every_docstring_sits_directly_above_a_definition():
    """Julia attaches a top-level string literal to the NEXT expression only when nothing (no comment, no
    blank line, no other statement) stands between them; a docstring that drifted away from its function
    documents the wrong object or none."""
    lines = open(JULIA).read().split("\n")
    opened = None
    checked = 0
    for i, ln in enumerate(lines):
        if ln == '"""':
            if opened is None:
                opened = i
            else:
                nxt = lines[i + 1]
                assert re.match(r"^(function |struct |mutable struct |const |macro |abstract type |[\w!]+\()", nxt), \
                    (i + 2, nxt)
                opened = None
                checked += 1
    assert opened is None and checked >= 8
    # the batched Greeks honour install!(devices = …) like every routed solve
    assert "devices = DEVICES[]" in open(JULIA).read()


# ---- block balance of the Julia sources (no Julia here: a missing `end` must show up on the CPU) ------------------
_JL_OPEN = {"function","if","for","while","let","do","begin","struct","module","try","macro","quote","baremodule"}
def _jl_strip(src):
    # remove triple-quoted strings, then ordinary strings, then comments
    src = re.sub(r'"""(?:.|\n)*?"""', '""', src)
    out=[]
    for line in src.split("\n"):
        res=""; i=0; instr=False
        while i < len(line):
            ch=line[i]
            if instr:
                if ch=="\\": i+=2; continue
                if ch=='"': instr=False
                i+=1; continue
            if ch=='"': instr=True; res+='""'; i+=1; continue
            if ch=="'" and i+2 < len(line) and line[i+2]=="'": res+="' '"; i+=3; continue
            if ch=="#": break
            res+=ch; i+=1
        out.append(res)
    return out
def _jl_block_balance(path):
    lines=_jl_strip(open(path).read())
    stack=[]; depth=0
    for ln,line in enumerate(lines,1):
        for tok in re.finditer(r"[A-Za-z_@!][A-Za-z_0-9!]*|[\[\]\(\)\{\}]", line):
            t=tok.group(0)
            if t in "[({": depth+=1; continue
            if t in "])}": depth-=1; continue
            if depth>0: continue                      # generators / comprehensions / a[end]
            prev=line[:tok.start()].rstrip()
            if t=="mutable": continue
            if t in _JL_OPEN:
                if prev.endswith(".") or prev.endswith(":"): continue    # x.begin / :if symbols
                stack.append((t,ln))
            elif t=="end":
                if not stack: return f"{path}:{ln}: unmatched end"
                stack.pop()
        if depth<0: return f"{path}:{ln}: bracket underflow"
    if stack: return f"{path}: unclosed {stack[-1]}"
    if depth: return f"{path}: bracket depth {depth} at EOF"
    return None


def test_julia_sources_have_balanced_blocks(tmp_path):
    """function / if / for / let / do / begin / struct / module … `end`, brackets and string quotes of both Julia
    files pair up (generators, comprehensions and `a[end]` inside brackets are skipped); a copy with one `end`
    removed is caught."""
    root = os.path.dirname(JULIA)
    for name in ("HedgehogMC.jl", "parity_replay.jl"):
        assert _jl_block_balance(os.path.join(root, name)) is None
    src = open(os.path.join(root, "parity_replay.jl")).read()
    bad = tmp_path / "bad.jl"
    bad.write_text(src.replace("end  # probe only", "", 1))
    assert src != bad.read_text() and _jl_block_balance(str(bad)) is not None


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference sources are only in the build container")
def test_every_hedgehog_name_the_julia_layer_uses_exists_in_the_reference():
    """`Hedgehog.X` and the names of `import Hedgehog: …` in julia/*.jl against the definitions in the reference's
    source text (struct / function / abstract type / const / assignment / re-imported names)."""
    import glob
    root = os.path.dirname(JULIA)
    src = open(JULIA).read() + open(os.path.join(root, "parity_replay.jl")).read()
    names = set(re.findall(r"Hedgehog\.([A-Za-z_]\w*!?)", src)) - {"jl"}
    imp = re.search(r"import Hedgehog: (.*?)\n\n", src, re.S).group(1)
    names |= {n.strip() for n in re.sub(r"\s+", " ", imp).split(",")}
    ref = "\n".join(open(f).read() for f in glob.glob("/root/reference/src/**/*.jl", recursive=True))
    missing = []
    for n in sorted(names):
        e = re.escape(n)
        if not re.search(rf"(struct\s+{e}\b|function\s+{e}\b|^\s*{e}\s*\(|abstract type\s+{e}\b|const\s+{e}\b|"
                         rf"^\s*{e}\s*=|import \w+: .*\b{e}\b)", ref, re.M):
            missing.append(n)
    assert len(names) >= 25 and not missing, missing


def _header_constants():
    """NAME -> int for every enumerator and integer #define of include/hedgehog_mc.h"""
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for body in re.findall(r"enum\s+\w+\s*\{(.*?)\}", hdr, re.S):
        nxt = 0
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            m = re.match(r"(\w+)\s*(?:=\s*(-?\w+))?$", item)
            assert m, item
            nxt = int(m.group(2), 0) if m.group(2) else nxt
            out[m.group(1)] = nxt
            nxt += 1
    for name, val in re.findall(r"^#define\s+(HH_\w+)\s+(-?\d+)\s", hdr, re.M):
        out[name] = int(val)
    return out


def test_enumerators_agree_between_the_header_the_ctypes_binding_and_the_julia_layer():
    """HH_* constants are spelled out three times (C header, _ffi.py, HedgehogMC.jl): every one that appears in a
    binding must carry the header's value."""
    from hedgehog_jl_amd import _ffi
    hdr = _header_constants()
    assert hdr["HH_MGPU_RCCL"] == 2 and hdr["HH_ERR_RCCL"] == -5 and hdr["HH_ACC_LEN"] == 16
    seen = 0
    for name, val in vars(_ffi).items():
        if name.startswith("HH_") and isinstance(val, int) and name in hdr:
            assert val == hdr[name], (name, val, hdr[name])
            seen += 1
    assert seen >= 30
    jl = open(JULIA).read()
    n_jl = 0
    for m in re.finditer(r"^const\s+((?:HH_\w+\s*,\s*)*HH_\w+)\s*=\s*(.+?)\s*(?:#.*)?$", jl, re.M):
        names = [x.strip() for x in m.group(1).split(",")]
        vals = re.findall(r"(?:Int32|Cint|UInt32)?\(?\s*(-?\d+)\s*\)?", m.group(2))
        vals = [int(v) for v in vals][:len(names)]
        assert len(vals) == len(names), m.group(0)
        for nm, v in zip(names, vals):
            if nm in hdr:
                assert v == hdr[nm], (nm, v, hdr[nm])
                n_jl += 1
    assert n_jl >= 10
