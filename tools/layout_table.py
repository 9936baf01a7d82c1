"""The layout of every struct of include/rt_api.h as the C compiler sees it (tests/helpers/layout_probe.c):
    python tools/layout_table.py           # markdown table for INTEGRATION.md
    layout()                               # {struct: {"size": n, "fields": {name: [offset, size]}}}"""
import json
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def layout():
    exe = os.path.join(tempfile.mkdtemp(), "layout_probe")
    subprocess.check_call(["gcc", "-std=c99", "-o", exe, os.path.join(ROOT, "tests", "helpers", "layout_probe.c")])
    return json.loads(subprocess.check_output([exe]))


def table(lay=None):
    lay = lay or layout()
    lines = ["| struct | size | fields: name @offset (size) |", "|---|---|---|"]
    for name, s in lay.items():
        fields = ", ".join("`%s` @%d (%d)" % (f, o, n) for f, (o, n) in s["fields"].items())
        lines.append("| `%s` | %d | %s |" % (name, s["size"], fields))
    return "\n".join(lines)


if __name__ == "__main__":
    print(table())
