"""The JVM shim under shim/scala cannot be compiled here (no JVM toolchain), so it is checked mechanically:

* every function include/epi.h declares is bound in Epi.scala by exactly one `fn("name", ...)`, with the same
  arity and with the ValueLayout of every argument and of the result matching the C type;
* every struct layout in Epi.scala has the size and the field offsets a C compiler gives the struct of epi.h
  (java.lang.foreign inserts no padding by itself: a missing paddingLayout shows up here);
* the downcalls the shim files make pass as many arguments as the handle was declared with;
* (when the reference tree is present) every public signature of the reference's env.scala, WorldToy.scala,
  Calibrate.scala and DDQNAgent.scala appears unchanged in the shim file that replaces it, so their callers
  (MutliTrainer.scala, App.scala, the specs) compile against the shim as they are."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import epi_header as H  # noqa: E402

SHIM = os.path.join(ROOT, "shim", "scala", "epidemic")
REFERENCE = "/root/reference/src/main/scala/epidemic"
SIZES = {"JAVA_INT": 4, "JAVA_LONG": 8, "JAVA_DOUBLE": 8, "JAVA_BYTE": 1, "ADDRESS": 8}


def _epi_scala():
    return open(os.path.join(SHIM, "Epi.scala")).read()


def _bindings():
    """{C name: (scala val, [result layout, argument layouts...])} from the fn(...) table."""
    out = {}
    for m in re.finditer(r'val\s+(\w+)\s*=\s*fn\("(\w+)"\s*,\s*([^)]*)\)', _epi_scala()):
        assert m.group(2) not in out, "%s bound twice" % m.group(2)
        out[m.group(2)] = (m.group(1), [x.strip() for x in m.group(3).split(",")])
    return out


def test_every_function_of_the_header_is_bound_with_matching_types():
    protos = H.prototypes()
    assert len(protos) >= 60
    bound = _bindings()
    assert sorted(bound) == sorted(name for name, _, _ in protos)
    for name, ret, params in protos:
        want = [H.layout_of(ret)] + [H.layout_of(t) for t, _ in params]
        assert bound[name][1] == want, "%s: header %s, binding %s" % (name, want, bound[name][1])


def _scala_layouts():
    """{VAL: [(field name or None for padding, size, alignment)]} from the structLayout definitions, in order."""
    text = _epi_scala()
    out = {}
    for m in re.finditer(r"val\s+(\w+)\s*:\s*StructLayout\s*=\s*MemoryLayout\.structLayout\((.*?)\)\n", text, flags=re.S):
        fields = []
        body = re.sub(r"//[^\n]*", "", m.group(2))
        for item in [x.strip() for x in body.split(",\n") if x.strip()]:
            pad = re.match(r"MemoryLayout\.paddingLayout\((\d+)\)$", item)
            if pad:
                fields.append((None, int(pad.group(1)), 1))
                continue
            fm = re.match(r'(\w+)\.withName\("(\w+)"\)$', item)
            assert fm, item
            lay = fm.group(1)
            if lay in SIZES:
                fields.append((fm.group(2), SIZES[lay], SIZES[lay]))
            else:
                inner = out[lay]
                fields.append((fm.group(2), sum(f[1] for f in inner), max(f[2] for f in inner)))
        out[m.group(1)] = fields
    return out


def test_struct_layouts_match_the_c_compiler(tmp_path):
    structs = H.structs()
    lines = ['#include "epi.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(void) {"]
    for sname, fields in structs.items():
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (sname, sname))
        for _, fname in fields:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (sname, fname, sname, fname))
    lines += ["  return 0;", "}"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    c_side = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        s, f, v = line.split()
        c_side[(s, f)] = int(v)
    layouts = _scala_layouts()
    assert sorted(layouts) == sorted(s[len("epi_"):].upper() for s in structs)
    for sname, fields in structs.items():
        lay = layouts[sname[len("epi_"):].upper()]
        off = 0
        seen = []
        for fname, size, align in lay:
            if fname is not None:
                assert off % align == 0, "%s.%s at %d is misaligned: java.lang.foreign rejects the layout" % (sname, fname, off)
                assert c_side[(sname, fname)] == off, "%s.%s: C offset %d, Scala layout %d" % (sname, fname, c_side[(sname, fname)], off)
                seen.append(fname)
            off += size
        assert seen == [f for _, f in fields], sname
        assert off == c_side[(sname, "size")], "%s: C size %d, Scala layout %d" % (sname, c_side[(sname, "size")], off)


def test_offsets_written_by_hand_in_the_helpers_match_the_header():
    """putState / putHyper / putLearnHyper / putParams write fields at literal offsets."""
    text = _epi_scala() + open(os.path.join(SHIM, "EpidemicEnv.scala")).read()
    lay = _scala_layouts()

    def offsets(name):
        off, out = 0, []
        for fname, size, _ in lay[name]:
            if fname is not None:
                out.append((off, size))
            off += size
        return out

    for fn_name, struct in (("putState", "STATE"), ("putHyper", "HYPER"), ("putLearnHyper", "LEARN_HYPER"), ("putParams", "ENV_PARAMS")):
        body = re.search(r"def %s\(.*?\n  \}" % fn_name, text, flags=re.S).group(0)
        base = "off" if fn_name == "putHyper" else "o"
        wrote = sorted((int(m.group(2) or 0), 8 if m.group(1) == "JAVA_DOUBLE" else 4)
                       for m in re.finditer(r"seg\.set\((JAVA_DOUBLE|JAVA_INT), %s(?: \+ (\d+))?," % base, body))
        assert wrote == offsets(struct), "%s writes %s, %s has %s" % (fn_name, wrote, struct, offsets(struct))


def _count_args(text, start):
    """Number of top-level arguments of the call whose '(' is at text[start]."""
    depth, args, k, any_char = 0, 0, start, False
    while k < len(text):
        ch = text[k]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                return args + (1 if any_char else 0)
        elif ch == "," and depth == 1:
            args += 1
        elif depth >= 1 and not ch.isspace():
            any_char = True
        k += 1
    raise AssertionError("unbalanced call")


def test_downcalls_in_the_shim_pass_the_declared_number_of_arguments():
    by_val = {v: (c, len(lay) - 1) for c, (v, lay) in _bindings().items()}
    calls = 0
    for fname in sorted(os.listdir(SHIM)):
        text = open(os.path.join(SHIM, fname)).read()
        for m in re.finditer(r"(?:Epi\.)?(\w+)\.invoke\(", text):
            if m.group(1) not in by_val:
                continue
            got = _count_args(text, m.end() - 1)
            cname, want = by_val[m.group(1)]
            assert got == want, "%s: %s.invoke passes %d arguments, %s takes %d" % (fname, m.group(1), got, cname, want)
            calls += 1
    assert calls >= 20


def _public_signatures(text):
    """Normalised public `def` signatures and primary-constructor / case-class parameter lists of a Scala source."""
    text = re.sub(r"//[^\n]*", "", text)
    sigs = set()
    for m in re.finditer(r"^[ \t]*((?:final |case |private(?:\[\w+\])? |override |@inline |inline )*)def\s+(\w+)\s*((?:\([^)]*\))*)\s*(?::\s*([^={\n]+?))?\s*(?:=|\{|$)",
                         text, flags=re.M):
        if "private" in m.group(1):
            continue
        params = " ".join(m.group(3).split())
        ret = " ".join((m.group(4) or "").split())
        sigs.add("def %s%s: %s" % (m.group(2), params, ret))
    for m in re.finditer(r"(?:final\s+)?(?:case\s+)?class\s+(\w+)\s*(?:private(?:\[\w+\])?\s*)?\(", text):
        start = m.end() - 1
        depth, k = 0, start
        while True:
            depth += text[k] == "("
            depth -= text[k] == ")"
            if depth == 0:
                break
            k += 1
        params = [" ".join(p.split()) for p in _split_top(text[start + 1:k])]
        sigs.add("class %s(%s)" % (m.group(1), ", ".join(params)))
    return sigs


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
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


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree is not on this machine")
@pytest.mark.parametrize("ref_file,shim_file", [("env.scala", "EpidemicEnv.scala"), ("WorldToy.scala", "WorldToy.scala"),
                                                ("Calibrate.scala", "Calibrate.scala"), ("DDQNAgent.scala", "DDQNAgent.scala")])
def test_public_signatures_of_the_reference_survive_in_the_shim(ref_file, shim_file):
    want = _public_signatures(open(os.path.join(REFERENCE, ref_file)).read())
    got = _public_signatures(open(os.path.join(SHIM, shim_file)).read())
    # the agent's public constructor is the shim's secondary one
    if shim_file == "DDQNAgent.scala":
        got.add("class DDQNAgent(hp: HyperParams, stateSize: Int, actionSize: Int)")
        assert "def this(hp: HyperParams, stateSize: Int, actionSize: Int)" in open(os.path.join(SHIM, shim_file)).read()
    missing = sorted(s for s in want if s not in got)
    assert not missing, "missing from %s:\n  %s\n(shim has:\n  %s)" % (shim_file, "\n  ".join(missing), "\n  ".join(sorted(got)))
    assert len(want) >= 2


def test_shim_files_say_they_are_uncompiled():
    for fname in os.listdir(SHIM):
        assert "UNCOMPILED HERE" in open(os.path.join(SHIM, fname)).read().splitlines()[0], fname


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree is not on this machine")
def test_public_values_and_the_action_enum_survive():
    def cases(text):
        return sorted(re.findall(r'case\s+(\w+)\s+extends\s+Action\((\d+),\s*"(\w+)"\)', text))
    ref_env = open(os.path.join(REFERENCE, "env.scala")).read()
    shim_env = open(os.path.join(SHIM, "EpidemicEnv.scala")).read()
    assert len(cases(ref_env)) == 8 and cases(shim_env) == cases(ref_env)
    assert "val all: Array[Action]" in shim_env
    for needle in ("val nodes: Vector[Node]",):
        assert needle in open(os.path.join(REFERENCE, "WorldToy.scala")).read()
        assert needle in open(os.path.join(SHIM, "WorldToy.scala")).read()
