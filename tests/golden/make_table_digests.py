#!/usr/bin/env python3
"""Generate tests/golden/table_digests.json from the reference's literal constant tables.

The reference sources are read AS TEXT / AS DATA (regex over array initialisers); nothing from
the reference is compiled, imported or executed.  Only digests (sha256 of a canonical binary
encoding) and element counts are committed -- the oracle and the product must reproduce the
same tables independently (generated from their definitions, or typed in as data), and
tests/test_tables.py compares digests.

Canonical encoding: integers -> little-endian int32; float-literal taps -> the float32 value
widened to float64, little-endian (that is what `double[] x = { 1.5F, ... }` holds in Java).

Run (in the build container, where /root/reference exists):
    python tests/golden/make_table_digests.py
"""
import hashlib
import json
import os
import re
import struct
import sys

import numpy as np

REF = os.environ.get("JSDR_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    return src


def array_body(src: str, name: str) -> str:
    """text between the outermost braces of `name ... = { ... };`"""
    m = re.search(r"\b" + re.escape(name) + r"\b[^=;]*=\s*\{", src)
    if not m:
        raise KeyError(name)
    i = m.end()
    depth = 1
    j = i
    while depth:
        c = src[j]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
        j += 1
    return src[i:j - 1]


def ints(body: str):
    return [int(t, 0) for t in re.findall(r"[-+]?(?:0x[0-9a-fA-F]+|\d+)", body)]


def float_literals(body: str):
    toks = re.findall(r"[-+]?\d+\.\d+(?:[eE][-+]?\d+)?[fF]", body)
    return [float(np.float32(t[:-1])) for t in toks]


def dig_ints(vals):
    return hashlib.sha256(b"".join(struct.pack("<i", v) for v in vals)).hexdigest()


def dig_f64(vals):
    return hashlib.sha256(b"".join(struct.pack("<d", v) for v in vals)).hexdigest()


def main():
    out = {}
    fec = strip_comments(open(os.path.join(REF, "FECDecoder.java")).read())
    for name in ["Partab", "mettab", "Syms", "Scrambler", "ALPHA_TO", "INDEX_OF", "RS_poly"]:
        vals = ints(array_body(fec, name))
        out[name] = {"n": len(vals), "sha256": dig_ints(vals), "kind": "int32"}
    dem = strip_comments(open(os.path.join(REF, "FUNcubeBPSKDemod.java")).read())
    for name in ["dsFilter", "dmFilter"]:
        vals = float_literals(array_body(dem, name))
        out[name] = {"n": len(vals), "sha256": dig_f64(vals), "kind": "float32-as-float64"}
    vals = ints(array_body(dem, "SYNC_VECTOR"))
    out["SYNC_VECTOR"] = {"n": len(vals), "sha256": dig_ints(vals), "kind": "int32"}
    # input-only fixtures shipped by the reference (copied as data files into tests/golden/)
    for fn in ["sine4410.raw", "sine4410-short.raw"]:
        data = open(os.path.join(REF, fn), "rb").read()
        out[fn] = {"n": len(data), "md5": hashlib.md5(data).hexdigest(), "kind": "file"}
    path = os.path.join(HERE, "table_digests.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", path)
    for k, v in sorted(out.items()):
        print(f"  {k:20s} n={v['n']}")


if __name__ == "__main__":
    sys.exit(main())
