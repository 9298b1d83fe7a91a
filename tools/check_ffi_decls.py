#!/usr/bin/env python3
"""Check that the Rust shim's `extern "C"` block (rust/poly-commit-hip/src/ffi.rs) declares EVERY function of
include/pc_hip.h with a matching signature, and that its constants agree with the header's enums.

There is no Rust toolchain in the authoring image, so nothing compiles the crate here; this check is what keeps the
FFI declarations from drifting away from the C ABI (it runs in the CPU test suite: tests/test_abi_cpu.py).

    python tools/check_ffi_decls.py            # exit 0 / 1, mismatches on stderr
    python tools/check_ffi_decls.py --emit     # print an extern block generated from the header (starting point for ffi.rs)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pc_hip.h")
FFI_RS = os.path.join(ROOT, "rust", "poly-commit-hip", "src", "ffi.rs")

OPAQUE = ("pc_ctx", "pc_srs", "pc_job", "pc_group", "pc_group_srs", "pc_group_job")
ENUMS = ("pc_curve", "pc_scalar_form", "pc_mem", "pc_hash", "pc_status")
CALLBACKS = ("pc_ipa_challenge_fn",)          # function-pointer typedefs of the header, declared as `pub type` aliases in ffi.rs


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def c_type_to_rust(t):
    """One C parameter / return type (already without the parameter name) -> Rust FFI type."""
    t = " ".join(t.split())
    arr = re.match(r"^(.*)\[(\d*)\]$", t)            # `float out[8]` decays to a pointer
    if arr:
        t = arr.group(1).strip() + " *"
    t = t.replace(" *", "*").replace("* ", "*")
    # split pointer levels: read right to left
    m = re.match(r"^(const )?([A-Za-z_0-9 ]+?)((?:\*(?: ?const)?)*)$", t)
    if not m:
        raise ValueError("cannot parse C type: " + t)
    base_const, base, ptrs = bool(m.group(1)), m.group(2).strip(), m.group(3)
    if base.endswith(" const"):
        base, base_const = base[:-6].strip(), True
    prim = {"int": "c_int", "unsigned": "c_uint", "unsigned int": "c_uint", "size_t": "usize", "void": "c_void", "char": "c_char",
            "float": "f32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8"}
    if base in prim:
        r = prim[base]
    elif base in OPAQUE:
        r = base
    elif base in ENUMS:
        r = "c_int"
    elif base in CALLBACKS:
        r = base
    else:
        raise ValueError("unknown C base type: " + base)
    levels = re.findall(r"\*( ?const)?", ptrs)
    # innermost pointer's pointee constness = base_const; each further level's pointee constness = the `const` after the previous '*'
    const_of_pointee = base_const
    for lv in levels:
        r = ("*const " if const_of_pointee else "*mut ") + r
        const_of_pointee = bool(lv)
    if not levels and r == "c_void":
        return "()"
    return r


def parse_header(path=HEADER):
    src = strip_comments(open(path).read())
    src = re.sub(r"#[^\n]*", " ", src)
    body = src[src.index('extern "C" {') + len('extern "C" {'):]
    body = re.sub(r"typedef\s+[A-Za-z_][A-Za-z_0-9 \*]*\(\s*\*\s*[a-z_]+\s*\)\s*\([^;]*\)\s*;", " ", body)      # function-pointer typedefs
    funcs = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 \*]*?)\b(pc_hip_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", body, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        args = []
        if params and params != "void":
            for p in params.split(","):
                p = p.strip()
                am = re.match(r"^(.*?)([A-Za-z_][A-Za-z_0-9]*)(\[\d*\])?$", p)
                ctype = (am.group(1).strip() + (am.group(3) or "")) if am else p
                args.append((am.group(2) if am else "_", c_type_to_rust(ctype)))
        funcs[name] = (args, c_type_to_rust(ret))
    enums = {}
    for m in re.finditer(r"typedef\s+enum\s*\{([^}]*)\}\s*([a-z_]+)\s*;", src, flags=re.S):
        for item in m.group(1).split(","):
            item = item.strip()
            if item:
                k, v = [x.strip() for x in item.split("=")]
                enums[k] = int(v)
    defines = {k: int(v) for k, v in re.findall(r"#define\s+(PC_HIP_[A-Z_]+)\s+(-?\d+)", open(path).read())}
    return funcs, enums, defines


def parse_ffi_rs(path=FFI_RS):
    src = re.sub(r"//[^\n]*", " ", open(path).read())
    m = re.search(r'extern\s+"C"\s*\{(.*)\}', src, flags=re.S)
    block = m.group(1) if m else ""
    funcs = {}
    for fm in re.finditer(r"pub\s+fn\s+(pc_hip_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        name, params, ret = fm.group(1), " ".join(fm.group(2).split()), (fm.group(3) or "()").strip()
        args = []
        for p in [x.strip() for x in params.split(",") if x.strip()]:
            an, at = p.split(":", 1)
            args.append((an.strip(), " ".join(at.split())))
        funcs[name] = (args, " ".join(ret.split()))
    consts = {k: int(v) for k, v in re.findall(r"pub\s+const\s+([A-Z_0-9]+)\s*:\s*[a-z_0-9]+\s*=\s*(-?\d+)\s*;", src)}
    return funcs, consts


def emit(funcs):
    out = ['#[link(name = "pc_hip")]', 'extern "C" {']
    for name, (args, ret) in funcs.items():
        a = ", ".join(f"{n if n != 'where' else 'where_'}: {t}" for n, t in args)
        out.append(f"    pub fn {name}({a})" + (f" -> {ret};" if ret != "()" else ";"))
    out.append("}")
    return "\n".join(out)


def check():
    hfuncs, henums, hdefs = parse_header()
    problems = []
    if not os.path.exists(FFI_RS):
        return [f"{FFI_RS} does not exist"], hfuncs
    rfuncs, rconsts = parse_ffi_rs()
    for name, (hargs, hret) in hfuncs.items():
        if name not in rfuncs:
            problems.append(f"missing in ffi.rs: {name}")
            continue
        rargs, rret = rfuncs[name]
        if rret != hret:
            problems.append(f"{name}: return type {rret!r} != {hret!r} (header)")
        if len(rargs) != len(hargs):
            problems.append(f"{name}: {len(rargs)} parameters, header has {len(hargs)}")
            continue
        for i, ((hn, ht), (rn, rt)) in enumerate(zip(hargs, rargs)):
            if ht != rt:
                problems.append(f"{name}: parameter {i} ({hn}): {rt!r} != {ht!r} (header)")
    for name in rfuncs:
        if name not in hfuncs:
            problems.append(f"declared in ffi.rs but not in the header: {name}")
    for k, v in list(henums.items()) + list(hdefs.items()):
        if k not in rconsts:
            problems.append(f"constant {k} missing in ffi.rs")
        elif rconsts[k] != v:
            problems.append(f"constant {k}: {rconsts[k]} != {v} (header)")
    problems += check_conventions_constants()
    return problems, hfuncs


def check_conventions_constants():
    """ROOT_* of rust/poly-commit-hip/tests/conventions.rs == TWO_ADIC_ROOT_OF_UNITY limbs of csrc/field_constants.h."""
    test = os.path.join(ROOT, "rust", "poly-commit-hip", "tests", "conventions.rs")
    if not os.path.exists(test):
        return [f"{test} does not exist"]
    fc = open(os.path.join(ROOT, "poly_commit_amd", "csrc", "field_constants.h")).read()
    rs = open(test).read()
    out = []
    for struct, const in (("pc_bls12_381_fr", "ROOT_BLS12_381_FR"), ("pc_bn254_fr", "ROOT_BN254_FR"), ("pc_pallas_fr", "ROOT_PALLAS_FR")):
        i = fc.index("struct " + struct + " {")
        body = fc[i:fc.index("NAME =", i)]
        w = [int(x.strip().rstrip("u"), 16) for x in re.search(r"ROOT\[\d*\]\s*=\s*\{([^}]*)\}", body).group(1).split(",")]
        want = [w[2 * k] | (w[2 * k + 1] << 32) for k in range(len(w) // 2)]
        m = re.search(const + r"\s*:\s*\[u64;\s*4\]\s*=\s*\[([^\]]*)\]", rs)
        got = [int(x.strip(), 16) for x in m.group(1).split(",")] if m else None
        if got != want:
            out.append(f"{const} in tests/conventions.rs differs from {struct}::ROOT")
    return out


if __name__ == "__main__":
    if "--emit" in sys.argv:
        print(emit(parse_header()[0]))
        sys.exit(0)
    probs, hf = check()
    for p in probs:
        print("check_ffi_decls:", p, file=sys.stderr)
    print(f"check_ffi_decls: {len(hf)} functions in the header, {len(probs)} problem(s)")
    sys.exit(1 if probs else 0)
