#!/usr/bin/env python3
"""One-off source tool (round 6 prune): resolve experiment switches in the csrc headers to their shipped setting and delete the dead branches.
usage: strip_switches.py file... ; UNDEF names are treated as not defined, ZERO names as the constant 0."""
import re, sys
UNDEF = {"MDRP_EXP_STATS", "MDRP_EXP_COSTSPLIT", "MDRP_EXP_SAMEPAIR", "MDRP_NO_ASM_FMA", "MDRP_NO_PRUNE", "MDRP_NO_ASM_MASK", "MDRP_NO_CLASSIFY",
         "MDRP_NO_BOUND", "MDRP_NO_F32", "MDRP_NO_SGPR_STATE", "MDRP_LM_FENCE", "MDRP_5PT_STAGES", "MDRP_FAST_BUILD", "MDRP_LO_TRACE"}
ZERO = {"MDRP_SOLVER_IEEE_DIV", "MDRP_LM_IEEE_DIV"}

def evaluate(cond):
    """returns True / False when the condition is decided by the switches, else a simplified condition string"""
    c = cond
    for n in UNDEF:
        c = re.sub(r"defined\s*\(\s*%s\s*\)" % n, "0", c)
    for n in ZERO:
        c = re.sub(r"\b%s\b" % n, "0", c)
    c = re.sub(r"!\s*0\b", "1", c)
    # fold "X && 1" / "1 && X" / "X && 0"
    parts = [p.strip() for p in c.split("&&")]
    if any(p == "0" for p in parts):
        return False
    parts = [p for p in parts if p != "1"]
    if not parts:
        return True
    return " && ".join(parts)

def process(text):
    out, stack = [], []  # stack entries: [state, emitted_if] ; state: 'keep' (condition unknown: directives kept), 'true', 'false'
    def active():
        return all(s[0] != 'false' for s in stack)
    for line in text.split("\n"):
        st = line.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", st)
        if not m:
            if active():
                out.append(line)
            continue
        d, rest = m.group(1), m.group(2)
        rest_nc = re.sub(r"//.*$", "", rest).strip()
        if d in ("ifdef", "ifndef", "if"):
            if d == "ifdef":
                name = rest_nc.split()[0]
                v = False if name in UNDEF else None
                cond = None
            elif d == "ifndef":
                name = rest_nc.split()[0]
                v = True if name in UNDEF else None
                if name in ZERO: v = "drop_define"
                cond = None
            else:
                r = evaluate(rest_nc)
                v = r if isinstance(r, bool) else None
                cond = r if not isinstance(r, bool) else None
            if v == "drop_define":
                stack.append(['false', False, 'dropdef'])
                continue
            if v is None:
                if active():
                    if d == "if" and cond is not None and cond != rest_nc:
                        out.append(re.sub(r"#\s*if\b.*", "#if " + cond, line))
                    else:
                        out.append(line)
                stack.append(['keep', True])
            else:
                stack.append(['true' if v else 'false', False])
        elif d == "else":
            top = stack[-1]
            if top[0] == 'keep':
                if all(s[0] != 'false' for s in stack[:-1]): out.append(line)
            elif len(top) > 2: pass
            else:
                top[0] = 'false' if top[0] == 'true' else 'true'
        elif d == "elif":
            top = stack[-1]
            assert top[0] == 'keep', "elif on a resolved switch: " + line
            if all(s[0] != 'false' for s in stack[:-1]): out.append(line)
        else:  # endif
            top = stack.pop()
            if top[0] == 'keep' and active():
                out.append(line)
    assert not stack
    return "\n".join(out)

for f in sys.argv[1:]:
    s = open(f).read()
    t = process(s)
    if t != s:
        open(f, "w").write(t)
        print(f, len(s.split("\n")), "->", len(t.split("\n")))
