"""Reflow a Markdown file to <= WIDTH columns: paragraphs and list items are wrapped, tables whose rows are too long become one bold title + a bullet per column,
code fences and short tables stay.  Usage: python scripts/r06/reflow_md.py FILE [WIDTH]"""
import re
import sys
import textwrap

path = sys.argv[1]; W = int(sys.argv[2]) if len(sys.argv) > 2 else 160
src = open(path).read().split("\n")
out = []


def wrap(text, first="", rest=""):
    return textwrap.wrap(text, width=W, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False) or [first.rstrip()]


def cells(row):
    parts = re.split(r"(?<!\\)\|", row.strip())
    if parts and parts[0] == "":
        parts = parts[1:]
    if parts and parts[-1].strip() == "":
        parts = parts[:-1]
    return [p.strip() for p in parts]


i = 0; fence = False
while i < len(src):
    line = src[i]
    if line.strip().startswith("```"):
        fence = not fence; out.append(line); i += 1; continue
    if fence or len(line) <= W and not line.lstrip().startswith("|"):
        # a short line: it may still be the first line of a paragraph with long followers; paragraphs are handled below
        pass
    if fence:
        out.append(line); i += 1; continue
    if line.lstrip().startswith("|"):
        blk = []
        while i < len(src) and src[i].lstrip().startswith("|"):
            blk.append(src[i]); i += 1
        if all(len(b) <= W for b in blk):
            out.extend(blk); continue
        head = cells(blk[0]); rows = [cells(b) for b in blk[2:]] if len(blk) > 1 and re.match(r"^\s*\|[\s:|-]+\|?\s*$", blk[1]) else [cells(b) for b in blk[1:]]
        for r in rows:
            title = r[0] if r and r[0] else (head[0] if head else "")
            out.extend(wrap(f"**{title}**" if title else "**·**"))
            for k in range(1, len(r)):
                if not r[k]:
                    continue
                name = head[k] if k < len(head) and head[k] else f"col {k}"
                out.extend(wrap(f"- *{name}*: {r[k]}", "", "  "))
            out.append("")
        continue
    if line.startswith("#") or line.strip() == "" or re.match(r"^\s*([-*_]\s*){3,}$", line):
        out.append(line); i += 1; continue
    # paragraph or list item: gather its continuation lines
    m = re.match(r"^(\s*)([-*+]|\d+[.)])\s+", line)
    if m:
        indent = m.group(1); marker = m.group(2); body = line[m.end():]; i += 1
        while i < len(src) and src[i].strip() and not re.match(r"^\s*([-*+]|\d+[.)])\s+", src[i]) and not src[i].startswith("#") and not src[i].lstrip().startswith("|") and not src[i].strip().startswith("```"):
            body += " " + src[i].strip(); i += 1
        out.extend(wrap(body, f"{indent}{marker} ", indent + " " * (len(marker) + 1)))
        continue
    indent = re.match(r"^(\s*)", line).group(1); body = line.strip(); i += 1
    while i < len(src) and src[i].strip() and not re.match(r"^\s*([-*+]|\d+[.)])\s+", src[i]) and not src[i].startswith("#") and not src[i].lstrip().startswith("|") and not src[i].strip().startswith("```"):
        body += " " + src[i].strip(); i += 1
    out.extend(wrap(body, indent, indent))
open(path, "w").write("\n".join(out))
long = [k + 1 for k, l in enumerate(out) if len(l) > W]
print(f"{path}: {len(out)} lines, {len(long)} still over {W} columns: {long[:10]}")
