"""cargo / rustc do not exist in this image, so the Rust binding
(bindings/rust/crispy-hip-sys/src/lib.rs) is checked mechanically against include/crispy_hip.h instead of being
compiled: every declared function (name, return type, argument count, argument types in order), every #[repr(C)]
struct (field names, types, order), the status codes and the size constants."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "crispy_hip.h")
RS = os.path.join(ROOT, "bindings", "rust", "crispy-hip-sys", "src", "lib.rs")

OPAQUE = {"crispy_rn", "crispy_mel", "crispy_asr", "crispy_resampler"}
STRUCTS = {"crispy_asr_hparams", "crispy_asr_specials", "crispy_asr_opts", "crispy_asr_segment", "crispy_asr_window",
           "crispy_asr_result"}
SCALARS_C = {"int": "i32", "long": "i64", "float": "f32", "double": "f64", "size_t": "usize", "char": "c_char", "void": "c_void",
             "int8_t": "i8", "int16_t": "i16", "unsigned char": "u8", "crispy_rn_layout": "i32",
             "crispy_asr_progress_fn": "crispy_asr_progress_fn"}
SCALARS_RS = {"c_int": "i32", "c_long": "i64", "c_float": "f32", "f32": "f32", "f64": "f64", "c_double": "f64", "usize": "usize", "c_char": "c_char",
              "c_void": "c_void", "i8": "i8", "c_uchar": "u8", "u8": "u8", "crispy_rn_layout": "i32", "i32": "i32", "i16": "i16",
              "crispy_asr_progress_fn": "crispy_asr_progress_fn"}


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def canon_c(t: str) -> str:
    """'const float *const *' -> '*const *const f32'; 'crispy_rn **' -> '*mut *mut crispy_rn'; 'int' -> 'i32'."""
    t = " ".join(t.replace("*", " * ").split())
    toks = [x for x in t.split() if x != "volatile"]      # Rust has no volatile types: the binding reads through a raw pointer
    # base type = everything before the first '*', minus a leading/trailing const
    first = toks.index("*") if "*" in toks else len(toks)
    base_toks = [x for x in toks[:first] if x != "const"]
    base_const = "const" in toks[:first]
    base = " ".join(base_toks)
    base = SCALARS_C.get(base, base)
    assert base in set(SCALARS_C.values()) | OPAQUE | STRUCTS, f"unknown C type {t!r}"
    # pointers, left to right: each '*' optionally followed by 'const' (constness of that pointer itself)
    ptrs = []
    i = first
    while i < len(toks):
        assert toks[i] == "*"
        self_const = i + 1 < len(toks) and toks[i + 1] == "const"
        ptrs.append(self_const)
        i += 2 if self_const else 1
    # pointee constness: first pointer points at base (base_const); pointer k+1 points at pointer k (its self_const)
    out = base
    pointee_const = base_const
    for self_const in ptrs:
        out = ("*const " if pointee_const else "*mut ") + out
        pointee_const = self_const
    return out


def canon_rs(t: str) -> str:
    t = " ".join(t.split())
    m = re.fullmatch(r"((?:\*(?:const|mut) )*)(\w+)", t)
    assert m, f"unparsed Rust type {t!r}"
    base = SCALARS_RS.get(m.group(2), m.group(2))
    assert base in set(SCALARS_RS.values()) | OPAQUE | STRUCTS, f"unknown Rust type {t!r}"
    return m.group(1) + base


def c_functions():
    src = _strip_c_comments(open(HDR).read())
    out = {}
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?[A-Za-z_][\w ]*?\s*\*?)\s*(crispy_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                pm = re.fullmatch(r"(.*?)(\w+)", a)                      # last identifier is the parameter name
                params.append(canon_c(pm.group(1).strip()))
        out[name] = (None if ret == "void" else canon_c(ret), params)
    return out


def rs_functions():
    src = re.sub(r"//[^\n]*", "", open(RS).read())
    blk = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (crispy_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", blk):
        name, args, ret = m.group(1), m.group(2).strip(), m.group(3)
        params = [canon_rs(a.split(":", 1)[1].strip()) for a in args.split(",") if a.strip()]
        out[name] = (canon_rs(ret.strip()) if ret else None, params)
    return out


def test_every_function_of_the_header_is_declared_identically():
    C, R = c_functions(), rs_functions()
    assert len(C) >= 55
    assert set(C) == set(R), (sorted(set(C) - set(R)), sorted(set(R) - set(C)))
    for name in sorted(C):
        assert C[name] == R[name], f"{name}: header {C[name]} vs lib.rs {R[name]}"


def test_type_canonicalisation_itself():
    assert canon_c("const float *") == "*const f32" and canon_c("float *") == "*mut f32"
    assert canon_c("crispy_rn **") == "*mut *mut crispy_rn"
    assert canon_c("const float *const *") == "*const *const f32"
    assert canon_c("const char **") == "*mut *const c_char"
    assert canon_c("const crispy_asr *") == "*const crispy_asr"
    assert canon_rs("*const *const c_float") == "*const *const f32"
    assert canon_rs("*mut *const c_char") == "*mut *const c_char"


def test_structs_fields_and_constants_match():
    csrc = _strip_c_comments(open(HDR).read())
    rsrc = re.sub(r"//[^\n]*", "", open(RS).read())
    for st in sorted(STRUCTS):
        cm = re.search(r"typedef struct %s \{(.*?)\} %s;" % (st, st), csrc, flags=re.S)
        cfields = []
        for decl in cm.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # "int a, b, c" or "const char *text"
            ty = re.match(r"((?:const )?(?:unsigned )?\w+)", decl).group(1)
            for nm in decl[len(ty):].split(","):
                nm = nm.strip()
                stars = nm.count("*")
                cfields.append((nm.replace("*", "").strip(), canon_c(ty + " " + "*" * stars)))
        rm = re.search(r"#\[repr\(C\)\](?:\s*#\[derive[^\]]*\])?\s*pub struct %s \{(.*?)\n\}" % st, rsrc, flags=re.S)
        assert rm, f"{st}: no #[repr(C)] struct in lib.rs"
        rfields = [(n, canon_rs(t.strip())) for n, t in re.findall(r"pub (\w+): ([^,\n]+),", rm.group(1))]
        assert cfields == rfields, (st, cfields, rfields)
    for st in OPAQUE:
        assert re.search(r"#\[repr\(C\)\]\s*pub struct %s \{\s*_private: \[u8; 0\],\s*\}" % st, rsrc), st
    for name, val in re.findall(r"(CRISPY_(?:OK|ERR_\w+)) = (-?\d+)", csrc):
        assert re.search(r"pub const %s: c_int = %s;" % (name, val), rsrc), name
    for name, val in re.findall(r"#define (CRISPY_(?:RN_FRAME_SIZE|RN_WEIGHT_BYTES|RN_TAPS|MEL_FRAMES|MEL_BINS)) (\d+)", csrc):
        assert re.search(r"pub const %s: usize = %s;" % (name, val), rsrc), name
    for name, val in (("CRISPY_RN_LAYOUT_TBF", 0), ("CRISPY_RN_LAYOUT_BTF", 1)):
        assert re.search(r"%s = %d" % (name, val), csrc) and re.search(r"pub const %s: crispy_rn_layout = %d;" % (name, val), rsrc)


def test_crate_files_exist_and_link_the_library():
    d = os.path.join(ROOT, "bindings", "rust", "crispy-hip-sys")
    toml = open(os.path.join(d, "Cargo.toml")).read()
    assert 'links = "crispy_hip"' in toml and 'name = "crispy-hip-sys"' in toml
    assert "rustc-link-lib=dylib=crispy_hip" in open(os.path.join(d, "build.rs")).read()
    rs = open(RS).read()
    for needle in ("pub struct DenoiseState", "pub fn process_frame(&mut self, output: &mut [f32], input: &[f32]) -> f32",
                   "pub struct GpuWhisperEngine", "impl SpeechModel for GpuWhisperEngine", "unsafe impl Send for DenoiseState"):
        assert needle in rs, needle
