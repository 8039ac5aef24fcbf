#!/usr/bin/env python3
"""Rust `extern "C"` declarations for EVERY export of include/vdbhip.h, derived from the header itself.

`python tools/gen_rust_ffi.py` prints the block INTEGRATION.md section 2 carries; tests/test_integration_doc_cpu.py
re-derives it and compares (names, arity, every parameter and return type), so the document cannot drift from the ABI.
No Rust toolchain exists in the build image: this is the only check possible short of compiling the binding."""
from __future__ import annotations

import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

OPAQUE = {"vdb_index": "VdbIndex", "vdb_ctx": "VdbCtx", "vdb_sharded": "VdbSharded", "vdb_pending": "VdbPending"}
SCALAR = {"int": "c_int", "uint64_t": "u64", "int64_t": "i64", "uint32_t": "u32", "uint8_t": "u8", "float": "f32",
          "double": "f64", "char": "c_char", "void": "c_void", "size_t": "usize"}


def c_decls(header: str | None = None):
    """[(name, ret_ctype, [(ctype, pname), ...]), ...] for every function prototype of the header, in order"""
    src = open(header or os.path.join(ROOT, "include", "vdbhip.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    out = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(vdb_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef"):
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)([A-Za-z_]\w*)$", a)
                ctype, pname = mm.group(1).strip(), mm.group(2)
                params.append((" ".join(ctype.replace("*", " * ").split()), pname))
        out.append((name, " ".join(ret.replace("*", " * ").split()), params))
    return out


def rust_type(ctype: str) -> str:
    toks = ctype.split()
    stars = toks.count("*")
    toks = [t for t in toks if t != "*"]
    const = "const" in toks
    base = [t for t in toks if t not in ("const", "struct")]
    assert len(base) == 1, ctype
    b = OPAQUE.get(base[0]) or SCALAR[base[0]]
    if stars == 0:
        return b
    inner = b
    for level in range(stars):
        # the innermost pointer carries the const of the pointee; outer levels are `*mut` (out-parameters)
        inner = ("*const " if (const and level == 0) else "*mut ") + inner
    return inner


RUST_KEYWORDS = {"type", "ref", "in", "box", "move", "match", "loop", "fn", "mod", "use", "self", "super", "where", "as"}


def rust_block() -> str:
    lines = ["use std::os::raw::{c_char, c_int, c_void};", ""]
    for c, r in OPAQUE.items():
        lines.append(f"#[repr(C)] pub struct {r} {{ _private: [u8; 0] }}   // {c}")
    lines += ["", 'extern "C" {']
    for name, ret, params in c_decls():
        ps = ", ".join(f"{(p + '_') if p in RUST_KEYWORDS else p}: {rust_type(t)}" for t, p in params)
        lines.append(f"    pub fn {name}({ps}) -> {rust_type(ret)};")
    lines.append("}")
    return "\n".join(lines)


def rewrite_doc() -> None:
    """replace INTEGRATION.md's extern "C" block in place (`python tools/gen_rust_ffi.py --write`)"""
    path = os.path.join(ROOT, "INTEGRATION.md")
    text = open(path).read()
    blocks = [m for m in re.finditer(r"```rust\n(.*?)```", text, flags=re.S) if 'extern "C" {' in m.group(1) and "pub fn vdb_last_error" in m.group(1)]
    assert len(blocks) == 1
    m = blocks[0]
    open(path, "w").write(text[:m.start(1)] + rust_block() + "\n" + text[m.end(1):])


if __name__ == "__main__":
    import sys

    if "--write" in sys.argv:
        rewrite_doc()
    else:
        print(rust_block())
