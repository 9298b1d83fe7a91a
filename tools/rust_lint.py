#!/usr/bin/env python3
"""The checks this image allows for Rust sources that have never met `rustc` (rust/poly-commit-hip, rust/ref-golden).

Not a parser and no substitute for `cargo check` (rust/README.md: the first-contact checklist): it finds the mistakes that do not need
type inference to be seen.

  1. delimiters: (), [], {} balance per file, with comments, strings, chars and lifetimes lexed away
  2. generics: `<` / `>` balance inside every `fn` signature, `impl` header, `struct` / `type` / `trait` header (up to the `{`, `;` or `=`)
  3. lifetimes: every named lifetime used in a `fn` signature is declared in its own `<..>`, in a `for<..>` binder of that
     signature, or in the enclosing `impl` header
  4. ffi: every `ffi::NAME` used anywhere exists in src/ffi.rs (a function of the extern block or a constant)
  5. ffi calls: every call `ffi::pc_hip_*(...)` passes as many arguments as the declaration takes; an argument that is visibly a
     pointer (`.as_ptr()`, `.as_mut_ptr()`, `as *const`, `as *mut`, `null()`, `&mut x`) sits at a pointer parameter and an
     argument that is visibly a length / integer (`.len()`, a literal, `as usize`, `as c_int`, `as u32`) at a non-pointer one -- the pointer/length
     ORDER of include/pc_hip.h (tools/check_ffi_decls.py ties ffi.rs to the header)
  6. modules: every `mod x;` has its file; every `crate::a::b` / `poly_commit_hip::a::b` / `use super::..` path names a module file
     that exists and an item (`fn`, `struct`, `enum`, `trait`, `type`, `const`, `static`, `mod`, macro) that module defines
  7. items: no two `fn`s of the same name at the top level of one file or inside one `impl` block

usage: python tools/rust_lint.py [crate_dir ...]     (default: both crates under rust/); exit 0 / 1, findings on stdout
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATES = [os.path.join(ROOT, "rust", "poly-commit-hip"), os.path.join(ROOT, "rust", "ref-golden")]


# ---- lexing -----------------------------------------------------------------------------------------------------------------------
def blank_noncode(src):
    """Same length as src; comments, string / char literal CONTENTS replaced by spaces (newlines kept), lifetimes kept."""
    out = list(src)
    i, n = 0, len(src)

    def blank(a, b):
        for k in range(a, b):
            if out[k] != "\n":
                out[k] = " "
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            blank(i, j)
            i = j
        elif src.startswith("/*", i):
            depth, j = 1, i + 2
            while j < n and depth:
                if src.startswith("/*", j):
                    depth += 1; j += 2
                elif src.startswith("*/", j):
                    depth -= 1; j += 2
                else:
                    j += 1
            blank(i, j)
            i = j
        elif c == '"' or (c == "r" and re.match(r'r#*"', src[i:]) and (i == 0 or not (src[i - 1].isalnum() or src[i - 1] == "_"))) or \
                (c == "b" and i + 1 < n and src[i + 1] == '"' and (i == 0 or not (src[i - 1].isalnum() or src[i - 1] == "_"))):
            m = re.match(r'b?r(#*)"', src[i:])
            if m:                                  # raw string
                close = '"' + m.group(1)
                j = src.find(close, i + m.end())
                j = n if j < 0 else j + len(close)
                blank(i + m.end(), j - len(close))
                i = j
            else:
                j = i + (2 if c == "b" else 1)
                while j < n and src[j] != '"':
                    j += 2 if src[j] == "\\" else 1
                blank(i + 1, j)
                i = j + 1
        elif c == "'":
            m = re.match(r"'(\\.[^']*|[^\\'])'", src[i:])           # char literal
            if m:
                blank(i + 1, i + m.end() - 1)
                i += m.end()
            else:                                                  # lifetime / label
                i += 1
        else:
            i += 1
    return "".join(out)


def line_of(src, pos):
    return src.count("\n", 0, pos) + 1


# ---- 1. delimiters ----------------------------------------------------------------------------------------------------------------
def check_delimiters(path, code, out):
    stack = []
    pairs = {")": "(", "]": "[", "}": "{"}
    for i, ch in enumerate(code):
        if ch in "([{":
            stack.append((ch, i))
        elif ch in ")]}":
            if not stack or stack[-1][0] != pairs[ch]:
                out.append(f"{path}:{line_of(code, i)}: unbalanced '{ch}'")
                return
            stack.pop()
    for ch, i in stack:
        out.append(f"{path}:{line_of(code, i)}: '{ch}' never closed")


# ---- 2./3. headers ----------------------------------------------------------------------------------------------------------------
HEADER_RE = re.compile(r"(?<![A-Za-z0-9_])(fn|impl|struct|enum|trait|type)\b")


def header_end(code, start):
    """Index of the `{`, `;` or (for `type`) `=` that ends the header starting at `start`, skipping (..) [..] groups and `where` clauses."""
    depth, i = 0, start
    while i < len(code):
        ch = code[i]
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        elif depth == 0 and ch in "{;":
            return i
        i += 1
    return len(code)


def angle_balance(text):
    t = text.replace("->", "  ").replace("=>", "  ")
    depth = 0
    for ch in t:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
            if depth < 0:
                return depth
    return depth


def check_headers(path, code, out):
    impl_stack = []          # (brace depth at which the impl body opened, lifetimes of its header)
    depth = 0
    events = {m.start(): m for m in HEADER_RE.finditer(code)}
    i = 0
    while i < len(code):
        ch = code[i]
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
            while impl_stack and impl_stack[-1][0] > depth:
                impl_stack.pop()
        m = events.get(i)
        if m:
            kw = m.group(1)
            end = header_end(code, m.end())
            head = code[m.start():end]
            if kw == "type" and "=" in head:
                head = head[:head.index("=")] + head[head.index("="):]          # generics on both sides must balance as a whole
            if kw == "fn" and re.match(r"fn\s*\(", head):                        # `fn(..) -> ..` TYPE, not an item
                i += 1
                continue
            if kw == "impl":
                # `impl Trait` in type position (an argument, a return type, a generic argument) is not an item
                k = m.start() - 1
                while k >= 0 and code[k].isspace():
                    k -= 1
                prev_word = re.search(r"(\w+)$", code[max(0, k - 10):k + 1])
                item_pos = k < 0 or code[k] in "};]{" or (prev_word and prev_word.group(1) in ("unsafe", "default"))
                if not item_pos or not re.match(r"impl\s*[<A-Za-z_!]", head):
                    i += 1
                    continue
            if kw == "impl" and end < len(code) and code[end] != "{":            # `impl Trait` in argument position
                i += 1
                continue
            STATS["headers_checked"] += 1
            bal = angle_balance(head)
            if bal != 0:
                out.append(f"{path}:{line_of(code, m.start())}: '<' / '>' do not balance in `{' '.join(head.split())[:90]}`")
            lts = set(re.findall(r"'([a-z_][A-Za-z0-9_]*)\b", head))
            if kw == "impl" and end < len(code) and code[end] == "{":
                g = re.match(r"impl\s*<([^{]*?)>\s", head)
                declared = set(re.findall(r"'([a-z_][A-Za-z0-9_]*)", g.group(1))) if g else set()
                declared |= set(re.findall(r"for\s*<\s*'([a-z_][A-Za-z0-9_]*)", head))
                impl_stack.append((depth + 1, declared))
                undeclared = lts - declared - {"static", "_"} - set(x for b in re.findall(r"for\s*<([^>]*)>", head) for x in re.findall(r"'([a-z_]\w*)", b))
                if undeclared:
                    out.append(f"{path}:{line_of(code, m.start())}: lifetime(s) {sorted(undeclared)} not declared in the impl header")
            if kw == "fn":
                g = re.match(r"fn\s+\w+\s*<", head)
                declared = set()
                if g:
                    j, d = g.end(), 1
                    while j < len(head) and d:
                        d += head[j] == "<"
                        d -= head[j] == ">" and head[j - 1] != "-"
                        j += 1
                    declared = set(re.findall(r"'([a-z_][A-Za-z0-9_]*)", head[g.end():j]))
                for b in re.findall(r"for\s*<([^>]*)>", head):
                    declared |= set(re.findall(r"'([a-z_]\w*)", b))
                for _, l in impl_stack:
                    declared |= l
                undeclared = lts - declared - {"static", "_"}
                if undeclared:
                    fname = re.match(r"fn\s+(\w+)", head).group(1)
                    out.append(f"{path}:{line_of(code, m.start())}: lifetime(s) {sorted(undeclared)} used in `fn {fname}` but declared nowhere")
        i += 1


# ---- 4./5. ffi ----------------------------------------------------------------------------------------------------------------------
def parse_ffi(ffi_code):
    """name -> list of parameter types (functions), plus the set of constant names."""
    fns, consts = {}, set(re.findall(r"pub\s+const\s+(\w+)\s*:", ffi_code))
    consts |= set(re.findall(r"pub\s+(?:struct|enum|type)\s+(\w+)", ffi_code))
    for m in re.finditer(r"(?:pub\s+)?use\s+([^;]+);", ffi_code):            # names the module brings in (core::ffi::c_void ..): visible as ffi::NAME only if `pub use`
        for pth in expand_use(m.group(1)):
            consts.add(pth.split("::")[-1].strip())
    for m in re.finditer(r"opaque!\s*\(([^)]*)\)", ffi_code):                 # the opaque handle types
        consts |= {x.strip() for x in m.group(1).split(",") if x.strip()}
    for m in re.finditer(r"pub\s+fn\s+(\w+)\s*\(", ffi_code):
        j, d = m.end(), 1
        while d:
            d += ffi_code[j] == "("
            d -= ffi_code[j] == ")"
            j += 1
        params = split_top(ffi_code[m.end():j - 1])
        fns[m.group(1)] = [p.split(":", 1)[1].strip() if ":" in p else p.strip() for p in params if p.strip()]
    return fns, consts


def split_top(s):
    out, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append("".join(cur)); cur = []
        else:
            cur.append(ch)
    if "".join(cur).strip():
        out.append("".join(cur))
    return out


PTR_ARG = re.compile(r"\.as_ptr\(\)|\.as_mut_ptr\(\)|as\s+\*(const|mut)\b|null(_mut)?\(\)|^&mut\s|^&\w|\.raw\b|\.srs\b|\.ptr\b|\.dev\b")
INT_ARG = re.compile(r"\.len\(\)\s*$|^\d+\s*$|as\s+(usize|c_int|c_uint|u32|i32|u64)\s*$|^(true|false)\s+as\s")


STATS = {"ffi_calls_checked": 0, "ffi_names_checked": 0, "paths_checked": 0, "headers_checked": 0}


def check_ffi(crate, files, out):
    ffi_path = os.path.join(crate, "src", "ffi.rs")
    if not os.path.exists(ffi_path):
        return
    fns, consts = parse_ffi(blank_noncode(open(ffi_path).read()).replace("->", "  "))
    for path, code in files.items():
        if path == ffi_path:
            continue
        for m in re.finditer(r"(?<![:\w])ffi::(\w+)", code):
            name = m.group(1)
            STATS["ffi_names_checked"] += 1
            if name not in fns and name not in consts:
                out.append(f"{path}:{line_of(code, m.start())}: ffi::{name} is not declared in src/ffi.rs")
                continue
            k = m.end()
            while k < len(code) and code[k].isspace():
                k += 1
            if name in fns and k < len(code) and code[k] == "(":
                j, d = k + 1, 1
                while d and j < len(code):
                    d += code[j] in "([{"
                    d -= code[j] in ")]}"
                    j += 1
                args = [a.strip() for a in split_top(code[k + 1:j - 1]) if a.strip()]
                STATS["ffi_calls_checked"] += 1
                if len(args) != len(fns[name]):
                    out.append(f"{path}:{line_of(code, m.start())}: ffi::{name} takes {len(fns[name])} arguments, the call passes {len(args)}")
                    continue
                for pos, (a, t) in enumerate(zip(args, fns[name])):
                    is_ptr_param = t.startswith("*")
                    if PTR_ARG.search(a) and not INT_ARG.search(a) and not is_ptr_param:
                        out.append(f"{path}:{line_of(code, m.start())}: ffi::{name} argument {pos + 1} `{a[:50]}` is a pointer, the parameter is `{t}`")
                    if INT_ARG.search(a) and not PTR_ARG.search(a) and is_ptr_param:
                        out.append(f"{path}:{line_of(code, m.start())}: ffi::{name} argument {pos + 1} `{a[:50]}` is an integer, the parameter is `{t}`")


# ---- 6. module paths --------------------------------------------------------------------------------------------------------------
ITEM_RE = r"(?:pub(?:\([a-z: ]+\))?\s+)?(?:unsafe\s+)?(?:const\s+)?(?:async\s+)?(?:extern\s+\"C\"\s+)?(?:fn|struct|enum|trait|type|const|static|mod|union)\s+{name}\b|macro_rules!\s*{name}\b|pub\s+use\s+[^;]*\b{name}\s*[;,}}]"


def module_file(crate, mod):
    for cand in (os.path.join(crate, "src", mod + ".rs"), os.path.join(crate, "src", mod, "mod.rs")):
        if os.path.exists(cand):
            return cand
    return None


def defines(code, name):
    return re.search(ITEM_RE.format(name=re.escape(name)), code) is not None


def expand_use(path_expr):
    """`a::{b, c::d}` -> ['a::b', 'a::c::d']"""
    path_expr = " ".join(path_expr.split())
    m = re.match(r"^(.*?)\{(.*)\}$", path_expr)
    if not m:
        return [path_expr.split(" as ")[0].strip()]
    out = []
    for part in split_top(m.group(2)):
        for e in expand_use(part.strip()):
            out.append(m.group(1) + e)
    return out


def check_modules(crate, files, out):
    lib = os.path.join(crate, "src", "lib.rs")
    crate_name = None
    toml = os.path.join(crate, "Cargo.toml")
    if os.path.exists(toml):
        m = re.search(r'^name\s*=\s*"([^"]+)"', open(toml).read(), re.M)
        crate_name = m.group(1).replace("-", "_") if m else None
    for path, code in files.items():
        in_src = os.path.dirname(path) == os.path.join(crate, "src")
        for m in re.finditer(r"^\s*(?:pub\s+)?mod\s+(\w+)\s*;", code, re.M):
            if in_src and not module_file(crate, m.group(1)):
                out.append(f"{path}:{line_of(code, m.start())}: `mod {m.group(1)};` has no file")
        roots = ["crate"] if in_src else []
        if crate_name and not in_src:
            roots.append(crate_name)
        refs = []
        for m in re.finditer(r"\buse\s+([^;]+);", code):
            for p in expand_use(m.group(1)):
                refs.append((m.start(), p))
        for m in re.finditer(r"\b((?:crate|" + (crate_name or "crate") + r")(?:::\w+)+)", code):
            refs.append((m.start(), m.group(1)))
        for pos, p in refs:
            segs = [s.strip() for s in p.split("::")]
            if not segs or segs[0] not in roots or len(segs) < 2 or not os.path.exists(lib):
                continue
            STATS["paths_checked"] += 1
            mf = module_file(crate, segs[1])
            if mf is None:
                if not defines(files.get(lib, ""), segs[1]):
                    out.append(f"{path}:{line_of(code, pos)}: `{p}`: no module or item `{segs[1]}` in the crate root")
                continue
            if len(segs) >= 3 and segs[2] not in ("*", "self") and not defines(files[mf] if mf in files else blank_noncode(open(mf).read()), segs[2]):
                out.append(f"{path}:{line_of(code, pos)}: `{p}`: `{segs[2]}` is not defined in {os.path.relpath(mf, crate)}")


# ---- 7. duplicate fns ---------------------------------------------------------------------------------------------------------------
def check_duplicates(path, code, out):
    depth, scopes = 0, [dict()]
    scope_depth = [0]
    i = 0
    fn_re = re.compile(r"(?<![A-Za-z0-9_])fn\s+(\w+)")
    opens_scope = {}
    for m in re.finditer(r"(?<![A-Za-z0-9_])(impl|trait|mod)\b", code):
        e = header_end(code, m.end())
        if e < len(code) and code[e] == "{":
            opens_scope[e] = True
    while i < len(code):
        ch = code[i]
        if ch == "{":
            depth += 1
            if opens_scope.get(i):
                scopes.append(dict()); scope_depth.append(depth)
        elif ch == "}":
            if len(scope_depth) > 1 and scope_depth[-1] == depth:
                scopes.pop(); scope_depth.pop()
            depth -= 1
        else:
            m = fn_re.match(code, i)
            if m and depth == scope_depth[-1]:
                name = m.group(1)
                if name in scopes[-1]:
                    out.append(f"{path}:{line_of(code, i)}: `fn {name}` defined twice in one scope (first at line {scopes[-1][name]})")
                scopes[-1][name] = line_of(code, i)
                i = m.end()
                continue
        i += 1


def lint_crate(crate):
    out, files = [], {}
    for base, _, names in os.walk(crate):
        if os.sep + "target" in base:
            continue
        for n in names:
            if n.endswith(".rs"):
                p = os.path.join(base, n)
                files[p] = blank_noncode(open(p).read())
    for path, code in sorted(files.items()):
        check_delimiters(path, code, out)
        check_headers(path, code, out)
        check_duplicates(path, code, out)
    check_ffi(crate, files, out)
    check_modules(crate, files, out)
    return out, len(files)


def main():
    crates = sys.argv[1:] or CRATES
    bad = 0
    for c in crates:
        findings, n = lint_crate(c)
        print(f"{os.path.relpath(c, ROOT)}: {n} files, {len(findings)} finding(s)")
        for f in findings:
            print("  " + os.path.relpath(f, ROOT) if f.startswith(ROOT) else "  " + f)
        bad += len(findings)
    print("checked:", ", ".join(f"{k} = {v}" for k, v in STATS.items()))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
