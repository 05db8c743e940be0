#!/usr/bin/env python3
"""Re-flow a markdown file to <= 120 columns (characters): paragraphs and list items are re-wrapped with hanging indents,
tables whose rows are wider than that become bullet lists (first cell in bold, the other cells named by their column),
headings and code blocks are left alone.  usage: wrap_md.py FILE..."""
import re
import sys
import textwrap

W = 120
BULLET = re.compile(r"^(\s*)((?:[*\-+]|\d+\.)\s+)")


def wrap(first, hang, body):
    w = textwrap.TextWrapper(width=W, initial_indent=first, subsequent_indent=hang, break_long_words=False, break_on_hyphens=False)
    return w.wrap(body) or [first.rstrip()]


def flush(par, out):
    if not par:
        return
    m = BULLET.match(par[0])
    if m:
        first = m.group(1) + m.group(2)
        body = " ".join([par[0][len(first):].strip()] + [p.strip() for p in par[1:]])
        out.extend(wrap(first, " " * len(first), body))
    else:
        lead = re.match(r"^\s*", par[0]).group(0)
        out.extend(wrap(lead, lead, " ".join(p.strip() for p in par)))
    par.clear()


def reflow(text):
    lines, out, par, in_code, i = text.split("\n"), [], [], False, 0
    while i < len(lines):
        ln = lines[i]
        if ln.strip().startswith("```"):
            flush(par, out); in_code = not in_code; out.append(ln); i += 1; continue
        if in_code:
            out.append(ln); i += 1; continue
        if ln.startswith("#") or not ln.strip():
            flush(par, out); out.append(ln); i += 1; continue
        if ln.startswith("|"):
            flush(par, out)
            tbl = []
            while i < len(lines) and lines[i].startswith("|"):
                tbl.append(lines[i]); i += 1
            if max(len(t) for t in tbl) <= W:
                out.extend(tbl); continue
            rows = [[c.strip() for c in t.strip().strip("|").split("|")] for t in tbl]
            header = rows[0]
            for r in rows[1:]:
                if all(re.match(r"^:?-+:?$", c) for c in r):
                    continue
                rest = ["%s: %s" % (h, c) if h else c for h, c in zip(header[1:], r[1:]) if c]
                out.extend(wrap("* ", "  ", "**%s** -- %s" % (r[0], "; ".join(rest))))
            continue
        if BULLET.match(ln):
            flush(par, out)
        par.append(ln)
        i += 1
    flush(par, out)
    return "\n".join(out)


for path in sys.argv[1:]:
    text = open(path).read()
    open(path, "w").write(reflow(text))
