"""Type-level comparison of include/zjhip.h with the extern block of bindings/rust/src/lib.rs, without a Rust compiler:
every prototype, fn-pointer typedef and #[repr(C)] struct is reduced to a canonical form on both sides.  TEST ONLY.

Canonical type = (base, chain) where chain lists, from the OUTERMOST pointer inwards, whether the pointee is const:
  C    `const int16_t *const *y`   -> ("i16", ("const", "const"))      Rust `*const *const i16`
  C    `uint8_t *const *outs`      -> ("u8",  ("const", "mut"))        Rust `*const *mut u8`
  C    `const int32_t qt[64]`      -> ("i32", ("const",))              Rust `*const i32`   (an array parameter is a pointer)
"""
import re

C_BASE = {"int16_t": "i16", "int32_t": "i32", "int64_t": "i64", "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "size_t": "usize",
          "int": "c_int", "unsigned": "c_uint", "float": "f32", "double": "f64", "char": "c_char", "void": "c_void"}
RUST_ALIAS = {"u32": "c_uint"}  # `unsigned` is declared as u32 in the shim: the same 32 bits on every target of the library


def strip_c_comments(src):
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def c_type(decl, named=True):
    """One C parameter / field / return declaration -> (base, chain, name)."""
    decl = decl.replace("ZJ_API", " ").strip()
    arr = re.findall(r"\[[^\]]*\]", decl)
    decl = re.sub(r"\[[^\]]*\]", " ", decl)
    toks = re.findall(r"\*|\w+", decl)
    name = None
    if named and toks and toks[-1] != "*" and toks[-1] not in C_BASE and toks[-1] != "const" and not toks[-1].startswith("zj_"):
        name = toks.pop()
    elif named and len([t for t in toks if t not in ("const", "*")]) > 1:
        name = toks.pop()
    base, base_const, levels = None, False, []  # levels[i] = constness of pointer i itself
    for t in toks:
        if t == "*":
            levels.append(False)
        elif t == "const":
            if levels:
                levels[-1] = True
            else:
                base_const = True
        elif t in ("struct", "enum"):
            continue
        else:
            assert base is None, (decl, toks)
            base = t
    for _ in arr[:1]:  # T x[N] as a parameter: one more pointer level (inner dimensions do not occur in the header's prototypes)
        levels.append(False)
    consts = [base_const] + levels  # c_0 .. c_n
    chain = tuple("const" if consts[i] else "mut" for i in range(len(levels) - 1, -1, -1))
    return (C_BASE.get(base, base), chain, name)


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out if a.strip()]


def c_prototypes(hdr):
    """{name: (ret, [args])} of every `ZJ_API ret name(args);`"""
    src = strip_c_comments(hdr)
    protos = {}
    for m in re.finditer(r"ZJ_API\s+([^;(]*?)\b(zj_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        r = c_type(ret, named=False)
        a = [] if args.strip() in ("", "void") else [c_type(x)[:2] for x in split_args(args)]
        protos[name] = ((r[0], r[1]), a)
    return protos


def c_fn_typedefs(hdr):
    src = strip_c_comments(hdr)
    out = {}
    for m in re.finditer(r"typedef\s+(\w+)\s*\(\s*\*\s*(zj_\w+)\s*\)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        r = c_type(m.group(1), named=False)
        out[m.group(2)] = ((r[0], r[1]), [c_type(x, named=False)[:2] for x in split_args(m.group(3))])
    return out


def c_structs(hdr):
    """{name: [(field, base, dims)]} of every `typedef struct name { ... } name;` with a body"""
    src = strip_c_comments(hdr)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(zj_\w+)\s*\{(.*?)\}\s*\1\s*;", src, flags=re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            base, rest = stmt.split(None, 1)
            for d in rest.split(","):
                d = d.strip()
                dims = tuple(int(x) for x in re.findall(r"\[(\d+)\]", d))
                fields.append((re.sub(r"\[.*", "", d).strip(), C_BASE.get(base, base), dims))
        out[m.group(1)] = fields
    return out


def rust_type(t):
    t = t.strip()
    chain = []
    while True:
        m = re.match(r"\*(const|mut)\s+(.*)", t, flags=re.S)
        if not m:
            break
        chain.append(m.group(1))
        t = m.group(2).strip()
    m = re.match(r"Option<(\w+)>$", t)
    if m:
        t = m.group(1)
    return (RUST_ALIAS.get(t, t), tuple(chain))


def rust_externs(rs):
    code = re.sub(r"//[^\n]*", "", rs)
    block = code[code.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    out = {}
    for m in re.finditer(r"pub fn (zj_\w+)\(([^)]*)\)(?:\s*->\s*([^;]+))?;", block, flags=re.S):
        args = [rust_type(a.split(":", 1)[1]) for a in split_args(m.group(2))]
        ret = rust_type(m.group(3)) if m.group(3) else ("c_void", ())
        out[m.group(1)] = (ret, args)
    return out


def rust_fn_types(rs):
    code = re.sub(r"//[^\n]*", "", rs)
    out = {}
    for m in re.finditer(r'pub type (zj_\w+) = unsafe extern "C" fn\(([^)]*)\)\s*->\s*([^;]+);', code):
        out[m.group(1)] = (rust_type(m.group(3)), [rust_type(a) for a in split_args(m.group(2))])
    return out


def rust_structs(rs):
    code = re.sub(r"//[^\n]*", "", rs)
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*pub struct (zj_\w+)\s*\{(.*?)\n\}", code, flags=re.S):
        fields = []
        for fm in re.finditer(r"pub (\w+):\s*((?:\[+[^,\]]*(?:;\s*\d+\])+)|[\w:]+)", m.group(2)):
            t = fm.group(2).strip()
            dims = []
            while t.startswith("["):
                inner, n = t[1:-1].rsplit(";", 1)
                dims.append(int(n))
                t = inner.strip()
            fields.append((fm.group(1), RUST_ALIAS.get(t, t) if t != "u32" else "u32", tuple(dims)))
        out[m.group(1)] = fields
    return out


def compare(hdr, rs):
    """List of human-readable mismatches between the header and the shim (empty = they agree)."""
    bad = []
    cp, rp = c_prototypes(hdr), rust_externs(rs)
    for name in sorted(set(cp) | set(rp)):
        if name not in cp or name not in rp:
            bad.append(f"{name}: declared on one side only")
            continue
        (cr, ca), (rr, ra) = cp[name], rp[name]
        if (cr[0], cr[1]) != (rr[0], rr[1]) and not (cr == ("c_void", ()) and rr == ("c_void", ())):
            bad.append(f"{name}: return type {cr} vs {rr}")
        if len(ca) != len(ra):
            bad.append(f"{name}: {len(ca)} vs {len(ra)} parameters")
            continue
        for i, (x, y) in enumerate(zip(ca, ra)):
            if x != y:
                bad.append(f"{name}: parameter {i}: C {x} vs Rust {y}")
    ct, rt = c_fn_typedefs(hdr), rust_fn_types(rs)
    for name in sorted(set(ct) | set(rt)):
        if ct.get(name) != rt.get(name):
            bad.append(f"fn type {name}: C {ct.get(name)} vs Rust {rt.get(name)}")
    cs, rst = c_structs(hdr), rust_structs(rs)
    for name, fields in cs.items():
        if name not in rst:
            bad.append(f"struct {name}: missing in the shim")
            continue
        cf = [(n, "u32" if b == "u32" else b, d) for n, b, d in fields]
        if cf != rst[name]:
            for a, b in zip(cf, rst[name]):
                if a != b:
                    bad.append(f"struct {name}: field C {a} vs Rust {b}")
                    break
            else:
                bad.append(f"struct {name}: {len(cf)} vs {len(rst[name])} fields")
    return bad
