#!/usr/bin/env python3
"""Extract the command-line flag surface of the REFERENCE's train.py (build container only;
needs /root/reference) into tests/golden/cli_flags.json.

The reference file is PARSED (python `ast`), never executed or copied: for every
`parser.add_argument(...)` call (reference train.py:14-33) the flag name, type, default,
action and dest are recorded as data.  tests/test_cli_cpu.py diffs
`otgan_amd.train.build_parser()` against this table.

    python oracle/make_golden_cli.py        # rewrites tests/golden/cli_flags.json
"""
import ast
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("OTGAN_REFERENCE", "/root/reference")


def flags_of(path):
    tree = ast.parse(open(path).read())
    out = []
    for node in ast.walk(tree):
        if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute)
                and node.func.attr == "add_argument"):
            continue
        name = ast.literal_eval(node.args[0])
        rec = {"name": name, "line": node.lineno, "type": None, "default": None, "action": None, "dest": None}
        for kw in node.keywords:
            if kw.arg == "type":
                rec["type"] = kw.value.id
            elif kw.arg in ("default", "action", "dest"):
                rec[kw.arg] = ast.literal_eval(kw.value)
        out.append(rec)
    out.sort(key=lambda r: r["line"])
    return out


def main():
    flags = flags_of(os.path.join(REF, "train.py"))
    dst = os.path.join(ROOT, "tests", "golden", "cli_flags.json")
    with open(dst, "w") as f:
        json.dump({"source": "reference train.py (argparse calls, parsed with ast)", "flags": flags}, f, indent=1)
    print(len(flags), "flags ->", dst)


if __name__ == "__main__":
    main()
