"""CPU: INTEGRATION.md's Rust `extern "C"` block lists EVERY export of include/vdbhip.h with the right arity and types.
No rustc exists in the build image, so the block is checked the only way possible: it is re-derived from the header
(tools/gen_rust_ffi.py: C type -> Rust FFI type) and compared declaration by declaration; independently, the names are
checked against the symbols libvdbhip.so really exports and the ctypes table of the Python binding."""
import os
import re
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def _doc_block():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```rust\n(.*?)```", text, flags=re.S)
    ffi = [b for b in blocks if 'extern "C" {' in b and "pub fn vdb_last_error" in b]
    assert len(ffi) == 1, "INTEGRATION.md must hold exactly one full extern \"C\" block"
    return ffi[0]


def _parse(block):
    out = []
    for m in re.finditer(r"pub fn (vdb_[a-z0-9_]+)\((.*?)\)\s*->\s*([^;]+);", block, flags=re.S):
        params = [tuple(x.strip() for x in p.split(":", 1)) for p in m.group(2).split(",") if p.strip()]
        out.append((m.group(1), params, m.group(3).strip()))
    return out


def test_extern_block_matches_header():
    import gen_rust_ffi as G

    doc = _parse(_doc_block())
    want = _parse(G.rust_block())
    assert [d[0] for d in doc] == [w[0] for w in want], "names / order differ from include/vdbhip.h"
    for d, w in zip(doc, want):
        assert len(d[1]) == len(w[1]), f"{d[0]}: arity {len(d[1])} in INTEGRATION.md, {len(w[1])} in the header"
        assert [t for _, t in d[1]] == [t for _, t in w[1]], f"{d[0]}: parameter types differ"
        assert d[2] == w[2], f"{d[0]}: return type differs"
    # every C prototype was understood by the generator (nothing silently dropped)
    hdr = open(os.path.join(ROOT, "include", "vdbhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(vdb_[a-z0-9_]+)\s*\(", hdr)))
    assert sorted(d[0] for d in doc) == names
    assert len(names) >= 75


def test_type_mapping_rules():
    import gen_rust_ffi as G

    assert G.rust_type("const float *") == "*const f32"
    assert G.rust_type("uint64_t *") == "*mut u64"
    assert G.rust_type("vdb_index * *") == "*mut *mut VdbIndex"
    assert G.rust_type("const vdb_index *") == "*const VdbIndex"
    assert G.rust_type("const char *") == "*const c_char"
    assert G.rust_type("void *") == "*mut c_void"
    assert G.rust_type("int64_t") == "i64"


def test_block_names_are_exported_and_bound():
    from lab_1806_vec_db_amd import _lib

    lib = _lib.load()
    for name, _, _ in _parse(_doc_block()):
        assert hasattr(lib, name), f"{name} is in INTEGRATION.md but libvdbhip.so does not export it"
        assert name in _lib.SIGNATURES or name in ("vdb_last_error", "vdb_version")
        if name in _lib.SIGNATURES:
            n_c = len(_lib.SIGNATURES[name])
            n_doc = len([d for d in _parse(_doc_block()) if d[0] == name][0][1])
            assert n_c == n_doc, f"{name}: ctypes table has {n_c} arguments, the header {n_doc}"
