#!/usr/bin/env python
"""Re-flows a markdown file to at most W columns: paragraphs and list items (with their continuation lines) are joined and wrapped again; tables whose rows
exceed W become nested lists (first cell = item, the other cells = "- *header*: text"); fenced code blocks, headings and short tables are left alone.
usage: tools/wrap_md.py FILE"""
import re, sys, textwrap
W = 156


def split_cells(line):
    cells, cur, tick = [], "", False
    for ch in line.strip()[1:]:
        if ch == "`":
            tick = not tick
        if ch == "|" and not tick:
            cells.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        cells.append(cur.strip())
    return cells


def wrap(text, first, rest):
    return textwrap.fill(text, width=W, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


MARK = re.compile(r"^(\s*)((?:[*-]|\d+\.)\s+)")
src = open(sys.argv[1]).read().split("\n")
out, i, fence = [], 0, False
while i < len(src):
    l = src[i]
    if l.lstrip().startswith("```"):
        fence = not fence; out.append(l); i += 1; continue
    if fence or l.strip() == "" or l.startswith("#"):
        out.append(l); i += 1; continue
    if l.startswith("|"):
        j = i
        while j < len(src) and src[j].startswith("|"):
            j += 1
        tbl = src[i:j]
        if max(len(x) for x in tbl) > W and len(tbl) >= 2 and re.match(r"^\|[\s:|-]+\|?\s*$", tbl[1]):
            hdr = split_cells(tbl[0])
            for row in tbl[2:]:
                c = split_cells(row)
                if not c:
                    continue
                out.append(wrap("* " + (c[0] if c[0] else "(—)"), "", "  "))
                for k in range(1, len(c)):
                    if c[k]:
                        out.append(wrap(f"- *{hdr[k] if k < len(hdr) and hdr[k] else 'col %d' % (k + 1)}*: {c[k]}", "  ", "    "))
            out.append("")
        else:
            out.extend(tbl)
        i = j; continue
    # a paragraph or a list item: this line plus the lines that continue it (deeper or equally indented text without a marker of its own)
    m = MARK.match(l)
    lead = len(l) - len(l.lstrip())
    cont = lead + (len(m.group(2)) if m else 0)
    parts = [l.strip()]
    j = i + 1
    while j < len(src):
        n = src[j]
        if n.strip() == "" or n.startswith("#") or n.startswith("|") or n.lstrip().startswith("```") or MARK.match(n):
            break
        nl = len(n) - len(n.lstrip())
        if m and nl < cont and nl <= lead:
            break
        if not m and nl != lead:
            break
        parts.append(n.strip()); j += 1
    out.append(wrap(" ".join(parts), " " * lead, " " * cont))
    i = j
open(sys.argv[1], "w").write("\n".join(out))
