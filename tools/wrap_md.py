"""Re-flow a markdown file to lines of at most WIDTH columns: paragraphs and list items are wrapped (continuation lines indented under the
item), table rows are turned into list items (`* cell 1 -- cell 2 -- ...`: a 3-6 KB table cell is not readable as a table), fenced code
blocks and headings are left alone.  Usage: python tools/wrap_md.py IN.md OUT.md [WIDTH=160]"""
import re
import sys
import textwrap


def cells_of(row):
    parts = re.split(r"(?<!\\)\|", row.strip())
    if parts and parts[0] == "":
        parts = parts[1:]
    if parts and parts[-1] == "":
        parts = parts[:-1]
    return [c.strip().replace("\\|", "|") for c in parts]


def main(src, dst, width=160):
    out, fence, header = [], False, None
    for line in open(src).read().split("\n"):
        if line.startswith("```"):
            fence = not fence
            out.append(line)
            continue
        if fence or line.startswith("#") or len(line) <= width and not line.startswith("|"):
            if not line.startswith("|"):
                header = None
            out.append(line)
            continue
        if line.startswith("|"):
            cells = cells_of(line)
            if all(re.fullmatch(r":?-+:?", c) for c in cells if c) and any(cells):
                continue  # the |---|---| rule
            if header is None:
                header = cells
                if any(cells):
                    out.append("")
                    out.append("(" + " / ".join(c for c in cells if c) + ")")
                    out.append("")
                continue
            text = " -- ".join(c for c in cells if c)
            out.extend(textwrap.wrap(text, width, initial_indent="* ", subsequent_indent="  ", break_long_words=False, break_on_hyphens=False))
            continue
        m = re.match(r"(\s*(?:[*+-]|\d+\.)\s+)", line)
        lead = m.group(1) if m else re.match(r"\s*", line).group(0)
        body = line[len(lead):]
        out.extend(textwrap.wrap(body, width, initial_indent=lead, subsequent_indent=" " * len(lead), break_long_words=False, break_on_hyphens=False))
    open(dst, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 160)
