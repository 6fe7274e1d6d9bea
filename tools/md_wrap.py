#!/usr/bin/env python
"""Keep the docs readable in a terminal: tables whose rows exceed the width become bullet lists (first cell bold, the other cells
labelled by their column header), and every paragraph / bullet is re-wrapped at 118 columns.  Code blocks and narrow tables are left
alone.  Usage: tools/md_wrap.py FILE.md [...]  (in place)"""
import re
import sys
import textwrap

W = 118


def cells(line):
    parts = re.split(r"(?<!\\)\|", line.strip())
    return [c.strip() for c in parts[1:-1]]


def convert_table(rows):
    head, body = cells(rows[0]), [cells(r) for r in rows[2:]]
    out = []
    for r in body:
        first = r[0] if r[0].startswith("**") or not r[0] else f"**{r[0]}**"
        text = first
        for h, c in zip(head[1:], r[1:]):
            if not c or c == "—":
                continue
            text += (f" — *{h}*: " if h else " — ") + c
        out.append("* " + text)
    return out


def wrap_block(lines):
    """lines: one paragraph or one list item (first line may start with a bullet)."""
    first = lines[0]
    m = re.match(r"^(\s*)([*\-] |\d+\. )?", first)
    indent = m.group(1) + (" " * len(m.group(2)) if m.group(2) else "")
    lead = m.group(0)
    text = " ".join([first[len(lead):].strip()] + [l.strip() for l in lines[1:]])
    return textwrap.wrap(text, width=W, initial_indent=lead, subsequent_indent=indent, break_long_words=False,
                         break_on_hyphens=False) or [lead.rstrip()]


def process(src):
    lines = src.split("\n")
    out, i, n = [], 0, len(lines)
    while i < n:
        l = lines[i]
        if l.startswith("```"):
            j = i + 1
            while j < n and not lines[j].startswith("```"):
                j += 1
            out += lines[i:j + 1]
            i = j + 1
            continue
        if l.lstrip().startswith("|") and i + 1 < n and re.match(r"^\s*\|[\s:|-]+\|\s*$", lines[i + 1]):
            j = i
            while j < n and lines[j].lstrip().startswith("|"):
                j += 1
            rows = lines[i:j]
            if max(len(r) for r in rows) > W + 7:
                for item in convert_table(rows):
                    out += wrap_block([item])
            else:
                out += rows
            i = j
            continue
        if not l.strip() or l.startswith("#") or l.startswith(">"):
            out.append(l)
            i += 1
            continue
        # paragraph or list item: gather continuation lines
        j = i + 1
        while j < n and lines[j].strip() and not re.match(r"^\s*([*\-] |\d+\. |\||#|```|>)", lines[j]):
            j += 1
        out += wrap_block(lines[i:j])
        i = j
    return "\n".join(out)


for path in sys.argv[1:]:
    s = open(path).read()
    t = process(s)
    open(path, "w").write(t)
    print(path, max(len(x) for x in t.split("\n")))
