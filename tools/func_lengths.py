#!/usr/bin/env python3
"""Longest function bodies of the given C++ files (brace matching from lines that look like a function head)."""
import re, sys
for path in sys.argv[1:]:
    src = open(path).read().split("\n")
    i = 0
    out = []
    while i < len(src):
        l = src[i]
        m = re.match(r"^(\t?)(?!\s)(?:[\w:<>\*&~,\s\[\]]+?)\b([\w:~]+)\((.*)$", l)
        if m and not re.match(r"^\t?(if|for|while|switch|return|else|do|struct|class|namespace|#|//|typedef|using)\b", l.strip()) and not l.rstrip().endswith(";"):
            # find opening brace
            j = i
            while j < len(src) and "{" not in src[j] and not src[j].rstrip().endswith(";"):
                j += 1
            if j < len(src) and "{" in src[j]:
                depth = 0
                k = j
                while k < len(src):
                    depth += src[k].count("{") - src[k].count("}")
                    if depth <= 0 and k >= j:
                        break
                    k += 1
                out.append((k - i + 1, m.group(2), i + 1))
                if m.group(1) == "":  # top-level: skip its body (members are listed when indented)
                    pass
        i += 1
    for n, name, at in sorted(out, reverse=True)[:8]:
        print("%5d  %s:%d  %s" % (n, path.split("/")[-1], at, name))
