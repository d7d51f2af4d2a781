#!/usr/bin/env python3
"""Keep a markdown file readable in a terminal: tables that hold a row of more than `limit` characters become bullet lists
(**first cell** -- *header:* further cells), and paragraphs / list items are re-flowed at `width` columns.  Code blocks,
headings and short tables are left alone.      python tools/wrap_md.py DESIGN.md [width] [limit]"""
import re
import sys
import textwrap

BULLET = re.compile(r"^(\s*)([*-]|\d+\.)\s+")


def cells(row):
    parts = re.split(r"(?<!\\)\|", row.strip())
    return [c.strip() for c in parts[1:-1]]


def flow(text, width, first, rest):
    return textwrap.wrap(" ".join(text.split()), width, initial_indent=first, subsequent_indent=rest, break_long_words=False,
                         break_on_hyphens=False)


def main(path, width=124, limit=300):
    lines = open(path).read().split("\n")
    out, i, in_code = [], 0, False
    special = lambda s: s.startswith("|") or s.startswith("#") or s.startswith("```") or not s.strip()
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("```"):
            in_code = not in_code
        if in_code or ln.startswith("```") or ln.startswith("#") or not ln.strip():
            out.append(ln)
            i += 1
            continue
        if ln.startswith("|"):
            j = i
            while j < len(lines) and lines[j].startswith("|"):
                j += 1
            block = lines[i:j]
            is_table = len(block) > 1 and re.match(r"^\|[\s:|-]+\|\s*$", block[1])
            if not is_table or max(len(b) for b in block) <= limit:
                out += block
            else:
                head = cells(block[0])
                for row in block[2:]:
                    c = cells(row)
                    txt = "**%s**" % c[0].strip("*") if c and c[0] else ""
                    for h, v in zip(head[1:], c[1:]):
                        if v:
                            txt += (" -- *%s:* %s" % (h, v)) if h else (" -- " + v)
                    out += flow(txt, width, "* ", "  ")
            i = j
            continue
        # a paragraph or a list item with its continuation lines
        m = BULLET.match(ln)
        j = i + 1
        while j < len(lines) and not special(lines[j]) and not BULLET.match(lines[j]):
            j += 1
        body = " ".join(l.strip() for l in lines[i:j])
        if m:
            first = m.group(1) + m.group(2) + " "
            out += flow(body[len(m.group(2)):].strip(), width, first, " " * len(first))
        else:
            out += flow(body, width, "", "")
        i = j
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], *(int(a) for a in sys.argv[2:]))
