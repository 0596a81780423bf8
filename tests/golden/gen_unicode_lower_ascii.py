#!/usr/bin/env python3
"""Enumerates every non-ASCII code point whose FULL lowercase mapping contains
an ASCII character, from the interpreter's Unicode database, and writes
tests/golden/unicode_lower_ascii.json.

Why it matters: the reference lowercases with str::to_lowercase (Unicode) and
then splits on every char that is not ASCII alphanumeric
(/root/reference/src/adapters/analyzer/lexicon.rs:54-58).  A byte-level
tokenizer is therefore exact iff it special-cases exactly these code points.
Expected output: U+0130 -> 'i' U+0307, U+212A -> 'k'.  (The set has been stable
across Unicode versions; Rust's tables agree.)
"""
import json
import os
import unicodedata

out = []
for c in range(0x80, 0x110000):
    if 0xD800 <= c <= 0xDFFF:
        continue
    low = chr(c).lower()
    if any(ord(x) < 128 for x in low):
        out.append(dict(cp=c, utf8=list(chr(c).encode("utf-8")), lower=[ord(x) for x in low]))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "unicode_lower_ascii.json")
with open(path, "w") as f:
    json.dump(dict(unicode_version=unicodedata.unidata_version, entries=out), f, indent=1)
    f.write("\n")
print(out)
