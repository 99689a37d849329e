"""Prints the downcall table and the struct layouts of shim/scala/epidemic/Epi.scala from include/epi.h.

The Scala file is kept by hand (it also holds the helpers around the table); run this after changing the header and
paste the two blocks.  tests/test_shim_signatures_cpu.py fails until header and binding agree again."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import epi_header as H


def camel(name):
    parts = name[len("epi_"):].split("_")
    return parts[0] + "".join(p.capitalize() for p in parts[1:])


def main():
    print("  // ---- downcall handles: one per function of include/epi.h, in header order")
    for name, ret, params in H.prototypes():
        res = H.layout_of(ret)
        args = [H.layout_of(t) for t, _ in params]
        print('  val %s = fn("%s", %s)' % (camel(name), name, ", ".join([res] + args)))
    print()
    print("  // ---- struct layouts of include/epi.h")
    st = H.structs()
    size, align = {}, {}
    for sname, fields in st.items():
        items, off, amax = [], 0, 1
        for ctype, fname in fields:
            base = ctype.replace("const", "").strip()
            if "*" in base:
                lay, sz, al = "ADDRESS", 8, 8
            elif base in st:
                lay, sz, al = sname_to_val(base), size[base], align[base]
            else:
                lay = H.SCALARS[base]
                sz = al = {"JAVA_INT": 4, "JAVA_LONG": 8, "JAVA_DOUBLE": 8, "JAVA_BYTE": 1}[lay]
            if off % al:   # java.lang.foreign inserts no padding by itself
                items.append("MemoryLayout.paddingLayout(%d)" % (al - off % al))
                off += al - off % al
            items.append('%s.withName("%s")' % (lay, fname))
            off += sz
            amax = max(amax, al)
        if off % amax:
            items.append("MemoryLayout.paddingLayout(%d)" % (amax - off % amax))
            off += amax - off % amax
        size[sname], align[sname] = off, amax
        print("  val %s: StructLayout = MemoryLayout.structLayout(   // %d bytes\n    %s)" % (sname_to_val(sname), off, ",\n    ".join(items)))


def sname_to_val(sname):
    return sname[len("epi_"):].upper()


if __name__ == "__main__":
    main()
