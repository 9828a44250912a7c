"""Partial preprocessor: resolves `#if / #ifdef / #ifndef / #elif / #else / #endif` blocks whose condition is made of
the given macros only (name=value pairs; value "undef" = not defined) and leaves every other conditional untouched.
Used once in round 3 to delete the measured-and-rejected kernel variants from mpnn_amd/csrc (their numbers live in
profiles/r02_ab_*.txt); kept as the record of how the sources were reduced.

  python tools/strip_switches.py file.hip NAME=0 OTHER=2 THIRD=undef ...   (rewrites the file in place)
"""
import re
import sys


def evaluate(expr, macros):
    """-> True / False, or None when the expression mentions anything but the known macros."""
    e = re.sub(r"//.*$", "", expr).strip()
    e = re.sub(r"/\*.*?\*/", "", e)

    def defined(m):
        name = m.group(1) or m.group(2)
        if name not in macros:
            raise KeyError(name)
        return "1" if macros[name] != "undef" else "0"

    try:
        e = re.sub(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)", defined, e)
        names = set(re.findall(r"[A-Za-z_]\w*", e))
        for n in names:
            if n not in macros:
                return None
        for n in names:
            e = re.sub(r"\b%s\b" % n, "0" if macros[n] == "undef" else macros[n], e)
        e = e.replace("&&", " and ").replace("||", " or ")
        e = re.sub(r"!(?!=)", " not ", e)
        return bool(eval(e, {"__builtins__": {}}))
    except KeyError:
        return None


def strip(lines, macros):
    out = []
    # stack entries: dict(known: bool, taken: bool (a branch already emitted), active: bool (current branch emitted),
    #                     parent_active: bool)
    stack = []

    def emitting():
        return all(s["active"] for s in stack if s["known"]) and all(s.get("parent_active", True) for s in stack)

    i = 0
    while i < len(lines):
        line = lines[i]
        # join continuation lines of a directive
        full = line
        j = i
        while full.rstrip().endswith("\\") and re.match(r"\s*#", line) and j + 1 < len(lines):
            j += 1
            full = full.rstrip()[:-1] + " " + lines[j]
        m = re.match(r"\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", full, re.S)
        if not m:
            if emitting():
                out.append(line)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2).strip()
        if kind in ("if", "ifdef", "ifndef"):
            if kind == "if":
                val = evaluate(rest, macros)
            else:
                name = re.match(r"(\w+)", rest).group(1)
                val = None if name not in macros else ((macros[name] != "undef") == (kind == "ifdef"))
            parent = emitting()
            if val is None:
                stack.append(dict(known=False, active=True, parent_active=parent))
                if parent:
                    out.extend(lines[i:j + 1])
            else:
                stack.append(dict(known=True, taken=val, active=val, parent_active=parent))
        elif kind == "elif":
            top = stack[-1]
            if not top["known"]:
                if emitting():
                    out.extend(lines[i:j + 1])
            else:
                if top["taken"]:
                    top["active"] = False
                else:
                    val = evaluate(rest, macros)
                    if val is None:
                        raise SystemExit(f"line {i + 1}: #elif with an unknown condition after known ones: not handled")
                    top["active"] = top["taken"] = val
        elif kind == "else":
            top = stack[-1]
            if not top["known"]:
                if emitting():
                    out.extend(lines[i:j + 1])
            else:
                top["active"] = not top["taken"]
                top["taken"] = True
        else:  # endif
            top = stack.pop()
            if not top["known"] and top["parent_active"] and emitting():
                out.extend(lines[i:j + 1])
        i = j + 1
    assert not stack, "unbalanced conditionals"
    return out


def main():
    path = sys.argv[1]
    macros = dict(a.split("=", 1) for a in sys.argv[2:])
    src = open(path).read().split("\n")
    res = strip(src, macros)
    # drop `#define NAME value` of resolved switches (their guards are gone)
    res = [ln for ln in res if not any(re.match(r"\s*#\s*define\s+%s\b" % n, ln) for n in macros)]
    open(path, "w").write("\n".join(res))


if __name__ == "__main__":
    main()
