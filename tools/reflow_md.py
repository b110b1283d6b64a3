"""Reflow the prose of a markdown file to at most WIDTH columns; tables, code fences, headings and list structure are kept
(list items and their continuation lines are wrapped with a hanging indent).  usage: python tools/reflow_md.py FILE [WIDTH]"""
import re
import sys
import textwrap


def reflow(text, width=100):
    out, para, indent, first = [], [], "", ""
    fence = False

    def flush():
        nonlocal para, indent, first
        if para:
            body = " ".join(s.strip() for s in para)
            out.extend(textwrap.wrap(body, width=width, initial_indent=first, subsequent_indent=indent,
                                     break_long_words=False, break_on_hyphens=False) or [first.rstrip()])
        para, indent, first = [], "", ""

    for line in text.split("\n"):
        if line.strip().startswith("```"):
            flush()
            fence = not fence
            out.append(line)
            continue
        if fence or line.lstrip().startswith("|") or line.startswith("#") or not line.strip() or line.strip() == "---":
            flush()
            out.append(line)
            continue
        m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", line)
        if m:
            flush()
            first = m.group(0)
            indent = " " * len(first)
            para = [line[len(first):]]
            continue
        if not para:
            lead = re.match(r"^\s*", line).group(0)
            first = indent = lead
        para.append(line)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    src = open(path).read()
    open(path, "w").write(reflow(src, width))
