"""Parser for include/epi.h: the prototypes and struct definitions of the C ABI as data.

Used by tests/test_shim_signatures_cpu.py (the JVM binding in shim/scala must agree with the header) and by
tools/gen_scala_bindings.py (prints the `fn(...)` table of Epi.scala from the header)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "epi.h")

SCALARS = {"int32_t": "JAVA_INT", "int64_t": "JAVA_LONG", "size_t": "JAVA_LONG", "double": "JAVA_DOUBLE",
           "uint8_t": "JAVA_BYTE", "int": "JAVA_INT"}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)          # preprocessor lines
    return text.replace('extern "C" {', " ")


def layout_of(ctype):
    """C parameter / return type -> java.lang.foreign ValueLayout name."""
    t = ctype.replace("const", " ").strip()
    if "*" in t:
        return "ADDRESS"
    t = " ".join(t.split())
    if t == "void":
        return None
    if t in SCALARS:
        return SCALARS[t]
    raise ValueError("no layout for C type %r" % ctype)


def prototypes(path=HEADER):
    """[(name, return C type, [(C type, parameter name), ...]), ...] in header order."""
    text = _strip_comments(open(path).read())
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    text = re.sub(r"typedef\s+enum\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    out = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(epi_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef") or not ret:
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                pm = re.match(r"(.*?)(\w+)$", a)
                params.append((pm.group(1).strip(), pm.group(2)))
        out.append((name, ret, params))
    return out


def structs(path=HEADER):
    """{struct name: [(C type, field name), ...]} for every `typedef struct name { ... } name;` with a body."""
    text = _strip_comments(open(path).read())
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"((?:const\s+)?\w+)\s+(.*)$", decl)
            base, rest = fm.group(1), fm.group(2)
            for item in rest.split(","):
                item = item.strip()
                stars = item.count("*")
                fields.append((base + " " + "*" * stars if stars else base, item.replace("*", "").strip()))
        out[m.group(3)] = fields
    return out


if __name__ == "__main__":
    for name, ret, params in prototypes():
        print(name, "|", ret, "|", params)
    for s, f in structs().items():
        print(s, f)
