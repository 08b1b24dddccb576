"""Readable names for the library's kernel symbols: `ln_bwd_kernel<bf16, 1>` from `_Z13ln_bwd_kernelIDF16bLi1EEv...`.

bench.py's launch trace (focal_trace_read) and `rocprofv3 -M --stats` both give the MANGLED symbol; the image's demanglers do not
know `DF16b` (__bf16) and print `bool _Accum` or give up, so both sides go through this one decoder instead: function name plus
the template arguments the library uses (types bf16 / float / int, integer and bool literals).  Anything it cannot decode is
returned unchanged, so two sides still compare equal.
"""
import re

_TYPES = {"DF16b": "bf16", "f": "float", "i": "int", "j": "unsigned", "b": "bool", "d": "double", "l": "long", "h": "uint8", "t": "uint16"}


def _ident(s, i):
    m = re.match(r"(\d+)", s[i:])
    if not m:
        return None, i
    n = int(m.group(1))
    j = i + len(m.group(1))
    return s[j:j + n], j + n


def short_kernel_name(sym):
    """`name<args>` of a mangled kernel symbol (or the symbol itself when it is not one of ours / not mangled)."""
    if not sym.startswith("_Z"):  # already demangled (by a demangler that may not know __bf16): name + template arguments, no parameter list
        s, depth, out = sym.replace("(anonymous namespace)::", "").replace("bool _Accum", "bf16"), 0, []
        if s.startswith("void "):
            s = s[5:]
        for ch in s:
            if ch == "(" and depth == 0:
                break
            depth += ch == "<"
            depth -= ch == ">"
            out.append(ch)
        return "".join(out).strip().split("::")[-1] if "<" not in "".join(out) else "".join(out).strip()
    i = 2
    name = None
    if sym[i] == "N":  # nested name: take the last identifier (the anonymous-namespace kernels)
        i += 1
        while i < len(sym) and sym[i].isdigit():
            name, i = _ident(sym, i)
    else:
        name, i = _ident(sym, i)
    if name is None:
        return sym
    if i >= len(sym) or sym[i] != "I":
        return name
    i += 1
    args = []
    while i < len(sym) and sym[i] != "E":
        if sym.startswith("DF16b", i):
            args.append("bf16")
            i += 5
        elif sym[i] == "L":  # literal: L <type> <value> E
            m = re.match(r"L([a-z])(n?\d+)E", sym[i:])
            if not m:
                return sym
            ty, val = m.group(1), m.group(2).replace("n", "-")
            args.append(("true" if val != "0" else "false") if ty == "b" else val)
            i += len(m.group(0))
        elif sym[i] in _TYPES:
            args.append(_TYPES[sym[i]])
            i += 1
        elif sym[i].isdigit():
            a, i = _ident(sym, i)
            args.append(a)
        else:
            return sym
    return f"{name}<{', '.join(args)}>"


if __name__ == "__main__":
    import sys
    for line in sys.stdin:
        print(short_kernel_name(line.strip()))
