#!/usr/bin/env python3
"""Keep the design documents readable in a terminal: no line over 200 characters.
   python tools/reflow_md.py FILE...      rewrites in place
 * a table with any row over the limit becomes a list: one bullet per row headed by its first cell, the other cells as "header: text" sub-items
 * any other line over the limit (paragraph, list item) is wrapped at 150 characters, continuation lines indented under the item's text
Code fences are left alone."""
import re
import sys
import textwrap

LIMIT, WIDTH = 200, 150


def cells(row):
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|"):
        row = row[:-1]
    return [c.strip() for c in re.split(r"(?<!\\)\|", row)]


def wrap(text, first, rest):
    return textwrap.wrap(text, WIDTH, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False) or [first.rstrip()]


def table_to_list(rows):
    head = cells(rows[0])
    out = []
    for r in rows[2:]:
        c = cells(r)
        out += wrap(f"**{c[0]}**" if c and c[0] else "(row)", "* ", "  ")
        for h, v in zip(head[1:], c[1:]):
            if v:
                out += wrap(f"*{h}*: {v}", "  - ", "    ")
    return out + [""]


def reflow(lines):
    out, i, fence = [], 0, False
    while i < len(lines):
        ln = lines[i]
        if ln.lstrip().startswith("```"):
            fence = not fence
            out.append(ln)
            i += 1
            continue
        if not fence and ln.lstrip().startswith("|") and i + 1 < len(lines) and re.match(r"^\s*\|[\s:|-]+\|?\s*$", lines[i + 1]):
            j = i
            while j < len(lines) and lines[j].lstrip().startswith("|"):
                j += 1
            rows = lines[i:j]
            out += table_to_list(rows) if any(len(r) > LIMIT for r in rows) else rows
            i = j
            continue
        if not fence and len(ln) > LIMIT:
            m = re.match(r"^(\s*(?:[*+-]|\d+\.)\s+|\s*)", ln)
            first = m.group(1)
            out += wrap(ln[len(first):], first, " " * len(first))
        else:
            out.append(ln)
        i += 1
    return out


for path in sys.argv[1:]:
    src = open(path, encoding="utf-8").read().split("\n")
    dst = reflow(src)
    open(path, "w", encoding="utf-8").write("\n".join(dst))
    print(path, "longest line", max(len(l) for l in dst))
