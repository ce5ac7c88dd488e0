#!/usr/bin/env python3
"""Generates the Rust `-sys` declarations of INTEGRATION.md section 2 from include/voxtral_hip.h (every constant, struct, opaque handle and
function), so the binding a maintainer would paste can never drift from the header:  python tools/gen_sys_block.py [--check]
(--check: exit 1 if INTEGRATION.md is stale; tests/test_abi_cpu.py runs it)."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "voxtral_hip.h"); DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED -sys (tools/gen_sys_block.py) -->", "<!-- END GENERATED -sys -->"
PRIM = {"int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "size_t": "usize", "float": "f32", "double": "f64",
        "uint8_t": "u8", "uint16_t": "u16", "char": "c_char", "void": "c_void", "int": "i32"}
KEYWORDS = {"in": "inp", "type": "ty", "ref": "r", "box": "bx", "move": "mv", "loop": "lp", "match": "m_"}


def rust_type(c: str) -> str:
    """`const float* const*` -> `*const *const f32`, `vox_ctx**` -> `*mut *mut vox_ctx`, `int32_t* const*` -> `*const *mut i32`."""
    c = c.strip()
    toks = re.findall(r"const|\*|[A-Za-z_][A-Za-z0-9_]*", c)
    base = None; base_const = False; i = 0
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            base_const = True
        else:
            base = toks[i]
        i += 1
    t = PRIM.get(base, base)
    ptrs = []                     # constness of what each '*' points to
    pointee_const = base_const
    while i < len(toks):
        assert toks[i] == "*"
        ptrs.append(pointee_const)
        pointee_const = False; i += 1
        if i < len(toks) and toks[i] == "const":
            pointee_const = True; i += 1
    for pc in ptrs:
        t = ("*const " if pc else "*mut ") + t
    return t


def parse_header():
    src = open(HDR).read()
    body = src[src.index('extern "C" {') + 12: src.rindex("#ifdef __cplusplus")]
    consts = re.findall(r"#define\s+(VOX_[A-Z_]+)\s+(\d+)(u?)", src)      # a `u` suffix (flag bits passed as uint32_t) -> u32
    opaque = re.findall(r"typedef struct (vox_[a-z0-9_]+) \1;", body)
    nocom = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    structs = []
    for m in re.finditer(r"typedef struct \{(.*?)\}\s*(vox_[a-z0-9_]+);", nocom, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ty, names = decl.rsplit(None, 1)[0], decl
            mm = re.match(r"((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*)\s+(.*)", decl)
            ty, names = mm.group(1), mm.group(2)
            for n in names.split(","):
                fields.append((n.strip(), rust_type(ty)))
        structs.append((m.group(2), fields))
    funcs = []
    for m in re.finditer(r"(?:^|\n)\s*(const char\*|int32_t)\s+(vox_[a-z0-9_]+)\s*\((.*?)\)\s*;", nocom, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*\[\d*\]$", a)
                if arr:
                    params.append((arr.group(2), rust_type(arr.group(1).strip() + "*"))); continue
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
                params.append((mm.group(2), rust_type(mm.group(1))))
        funcs.append((name, params, "*const c_char" if "char" in ret else "i32"))
    return consts, opaque, structs, funcs


def generate() -> str:
    consts, opaque, structs, funcs = parse_header()
    out = ["```rust", "// generated from include/voxtral_hip.h by tools/gen_sys_block.py -- do not edit by hand", "#![allow(non_camel_case_types)]",
           "use std::os::raw::{c_char, c_void};", ""]
    for o in opaque:
        out.append(f"#[repr(C)] pub struct {o} {{ _p: [u8; 0] }}")
    out.append("")
    for name, fields in structs:
        out.append("#[repr(C)] #[derive(Clone, Copy, Debug, Default)]")
        out.append(f"pub struct {name} {{ " + ", ".join(f"pub {KEYWORDS.get(n, n)}: {t}" for n, t in fields) + " }")
    out.append("")
    for n, v, u in consts:
        out.append(f"pub const {n}: {'u32' if u else 'i32'} = {v};")
    out += ["", 'extern "C" {']
    for name, params, ret in funcs:
        ps = ", ".join(f"{KEYWORDS.get(n, n)}: {t}" for n, t in params)
        out.append(f"    pub fn {name}({ps}) -> {ret};")
    out += ["}", "```"]
    return "\n".join(out)


def main():
    doc = open(DOC).read()
    a, b = doc.index(BEGIN) + len(BEGIN), doc.index(END)
    new = doc[:a] + "\n" + generate() + "\n" + doc[b:]
    if "--check" in sys.argv:
        sys.exit(0 if new == doc else 1)
    open(DOC, "w").write(new)
    print(f"INTEGRATION.md: {len(parse_header()[3])} functions")


if __name__ == "__main__":
    main()
