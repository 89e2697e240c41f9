"""The Go binding (go/internal/pattern/pattern_hip.go) cannot be compiled here
(no Go toolchain), so its contract with the C ABI is checked textually: every
C.oval_* function it calls is declared in include/*.h with the same number of
arguments, it defines every symbol the reference's pattern_test.go uses, and it
carries no placeholder."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO_DIR = os.path.join(ROOT, "go", "internal", "pattern")
GO_FILE = os.path.join(GO_DIR, "pattern_hip.go")
GO_FILES = sorted(os.path.join(GO_DIR, f) for f in os.listdir(GO_DIR) if f.endswith(".go") and not f.endswith("_test.go"))
GO_TEST_FILES = sorted(os.path.join(GO_DIR, f) for f in os.listdir(GO_DIR) if f.endswith("_test.go"))


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def _split_top_level(args):
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _declared_functions():
    """name -> number of parameters, from every header in include/."""
    decls = {}
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        if not fn.endswith(".h"):
            continue
        text = _strip_c_comments(open(os.path.join(inc, fn)).read())
        for m in re.finditer(r"\b(oval_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
            name, args = m.group(1), m.group(2).strip()
            n = 0 if args in ("", "void") else len(_split_top_level(args))
            decls[name] = n
    return decls


def _go_calls(src):
    """(name, number of arguments) of every C.oval_*( ... ) call."""
    calls = []
    for m in re.finditer(r"\bC\.(oval_[a-z0-9_]+)\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        args = src[m.end():i - 1].strip()
        calls.append((m.group(1), 0 if not args else len(_split_top_level(args))))
    return calls


def test_every_c_call_matches_a_declaration():
    assert len(GO_FILES) >= 3
    go_code = ""
    for f in GO_FILES:
        src = open(f).read()
        assert src.startswith("//go:build ovalhip\n"), f
        go_code += re.sub(r"//[^\n]*", "", src.split('import "C"', 1)[1])
    decls = _declared_functions()
    calls = _go_calls(go_code)
    assert len(calls) >= 16
    types = {"oval_read_fn"}                                   # a conversion, not a call
    for name, n_args in calls:
        if name in types:
            continue
        assert name in decls, f"{name} is called from Go but not declared in include/*.h"
        assert decls[name] == n_args, f"{name}: Go passes {n_args} arguments, the header declares {decls[name]}"
    used = {n for n, _ in calls}
    for must in ("oval_pattern_generate", "oval_pattern_valid_stream", "oval_pattern_decide_size",
                 "oval_pattern_generate_data_unit", "oval_pattern_valid_data_unit", "oval_pattern_set_batching",
                 "oval_pattern_generate_parts", "oval_parts_count", "oval_parts_get", "oval_parts_release",
                 "oval_parts_close", "oval_pattern_generate_batch", "oval_pattern_valid_batch"):
        assert must in used, f"the Go binding does not call {must}"


def test_go_file_defines_the_reference_surface_and_has_no_placeholders():
    src = open(GO_FILE).read()
    assert src.startswith("//go:build ovalhip\n")
    # internal/pattern/pattern.go:22,73,94,118,136 and the constant pattern_test.go:40,90 read
    for sig in (r"func DecideSize\(minSize, maxSize int\) \(int, error\)",
                r"func Generate\(dataSize, workerID int, bucketName string, obj \*object\.Object\) \(\[\]byte, error\)",
                r"func generateDataUnit\(unitCount, workerID int, bucketName string, obj \*object\.Object\) \(\[\]byte, error\)",
                r"func Valid\(workerID int, expectedBucketName string, obj \*object\.Object, reader io\.Reader\) error",
                r"func validDataUnit\(unitCount, workerID int, expectedBucketName string, obj \*object\.Object, data \[\]byte\) error",
                r"dataUnitSize\s+= 256"):
        assert re.search(sig, src), sig
    assert "//export ovalGoRead" in src and "size_t ovalGoRead(void* user, uint8_t* p, size_t n);" in src
    for f in GO_FILES:
        text = open(f).read()
        body = text.split('import "C"', 1)[1]
        assert "/*" not in body and "TODO" not in text and "..." not in body, f
    # the read callback has the type the header gives oval_read_fn
    hdr = _strip_c_comments(open(os.path.join(ROOT, "include", "oval_pattern.h")).read())
    assert re.search(r"typedef\s+size_t\s*\(\*oval_read_fn\)\(void\*\s*user,\s*uint8_t\*\s*p,\s*size_t\s+n\)", hdr)


def test_go_tests_use_only_what_the_binding_defines():
    """The Go tests of the MI355X build cannot be run here either: what can be
    checked is that they carry the build tag, sit in the package, and name only
    functions, methods and struct fields that the binding files define."""
    assert GO_TEST_FILES
    binding = "".join(open(f).read() for f in GO_FILES)
    funcs = set(re.findall(r"^func (\w+)\(", binding, flags=re.M))
    methods = set(re.findall(r"^func \(\w+ \*?\w+\) (\w+)\(", binding, flags=re.M))
    for f in GO_TEST_FILES:
        src = open(f).read()
        assert src.startswith("//go:build ovalhip\n") and "\npackage pattern\n" in src, f
        code = re.sub(r"//[^\n]*", "", src)
        for name in ("GenerateBatch", "ValidBatch", "GenerateParts", "Generate", "Valid", "DecideSize"):
            if re.search(r"\b%s\(" % name, code):
                assert name in funcs, f"{name} is used by {os.path.basename(f)} but not defined by the binding"
        for m in re.findall(r"\bparts\.(\w+)\(", code):
            assert m in methods, f"Parts.{m} is not defined by the binding"
        for struct, body in re.findall(r"\b(GenerateItem|ValidItem)\{([^}]*)\}", code):
            decl = re.search(r"type %s struct \{(.*?)\n\}" % struct, binding, flags=re.S).group(1)
            for field in re.findall(r"(\w+):", body):
                assert re.search(r"^\s*%s\s" % field, decl, flags=re.M), f"{struct} has no field {field}"
        # every import is used and every used package imported (the mistakes a compiler would refuse first)
        imports = re.findall(r'^\t"([\w/.]+)"$', src, flags=re.M)
        for imp in imports:
            pkg = imp.rsplit("/", 1)[-1]
            assert re.search(r"\b%s\." % pkg, code), f"{os.path.basename(f)} imports {imp} without using it"
        for pkg in ("bytes", "io", "strings", "sync", "object", "testing"):
            if re.search(r"\b%s\.\w" % pkg, code):
                assert any(i.rsplit("/", 1)[-1] == pkg for i in imports), f"{os.path.basename(f)} uses {pkg} without importing it"


def _preamble(src):
    """The C text cgo compiles for a file: the comment block right before import "C",
    without the #cgo directives (they are for the go tool, not for the compiler)."""
    head = src.split('import "C"', 1)[0]
    body = head[head.rindex("/*") + 2:head.rindex("*/")]
    return "\n".join(line for line in body.splitlines() if not line.lstrip().startswith("#cgo"))


def test_cgo_preambles_compile_as_c_and_exports_match_their_declarations(tmp_path):
    """What a C compiler can say about the Go files without a Go toolchain: every
    preamble is valid C11 against include/ (cgo hands it to the C compiler as it
    is), and the prototype cgo will generate for each //export function (Go
    parameter types mapped to C as cgo maps them) is the one the preamble declares,
    so the two declarations of the symbol that meet in cgo's generated C file agree."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of this image"
    go_to_c = {"unsafe.Pointer": "void*", "*C.uint8_t": "uint8_t*", "C.size_t": "size_t", "C.int": "int",
               "*C.char": "char*", "C.int64_t": "int64_t"}
    for f in GO_FILES:
        src = open(f).read()
        pre = _preamble(src)
        exports = re.findall(r"^//export (\w+)\nfunc (\w+)\(([^)]*)\)\s*([\w.*]*)\s*\{", src, flags=re.M)
        checks = ""
        for tag, name, params, ret in exports:
            assert tag == name, f"//export {tag} sits on func {name}"
            c_params = []
            for p in _split_top_level(params):
                pname, ptype = p.strip().split(None, 1)
                c_params.append(f"{go_to_c[ptype.strip()]} {pname}")
            proto = f"{go_to_c[ret] if ret else 'void'} {name}({', '.join(c_params)});"
            # a second, compatible declaration is legal C; an incompatible one is an error
            checks += f"extern {proto}\n"
            want = re.sub(r"\s+", " ", proto)
            assert want in re.sub(r"\s+", " ", pre), f"{os.path.basename(f)}: the preamble does not declare `{want}`"
        if "//export" in src:
            # cgo refuses definitions in the preamble of a file that exports
            assert not re.search(r"\)\s*\{", pre), f"{os.path.basename(f)}: a definition in the preamble of an exporting file"
        c = tmp_path / (os.path.basename(f) + ".c")
        c.write_text(pre + "\n" + checks)
        p = subprocess.run([gcc, "-std=c11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                            "-I" + os.path.join(ROOT, "include"), str(c)], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
    assert any("//export" in open(f).read() for f in GO_FILES)
