#!/usr/bin/env python3
"""Run a slice of tests/golden/simple_tests.json (records whose arguments or name match the given substrings) through the
product binaries and compare with the recorded reference SAM -- a quick device check of selected regression cases."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bt2test import ref_bin
pats = sys.argv[1:] or ["-X"]
cases = [r for r in json.load(open(os.path.join(ROOT, "tests", "golden", "simple_tests.json"))) if any(p in " ".join(r["args"]) or p in r["name"] for p in pats)]
b = os.path.join(ROOT, "bowtie2_amd", "bin")
bad = n = 0
built = {}
with tempfile.TemporaryDirectory() as tmp:
    for rec in cases:
        for width, exe in (("s", "bowtie2-align-s"), ("l", "bowtie2-align-l")):
            key = (tuple(rec["ref"]), width)
            if key not in built:
                d = os.path.join(tmp, "i%d" % len(built)); os.makedirs(d)
                open(d + "/r.fa", "w").write("".join(">%d\n%s\n" % (i, s) for i, s in enumerate(rec["ref"])))
                subprocess.check_call([ref_bin("bowtie2-build-l" if width == "l" else "bowtie2-build-s"), "--quiet", d + "/r.fa", d + "/i"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                built[key] = d + "/i"
            cmd = [os.path.join(b, exe)] + rec["args"] + ["-x", built[key]]
            if "m1" in rec:
                if rec["flag"] == "-c": cmd += ["-c", "-1", rec["m1"].strip(), "-2", rec["m2"].strip()]
                else:
                    open(tmp + "/1", "w").write(rec["m1"]); open(tmp + "/2", "w").write(rec["m2"])
                    cmd += [rec["flag"], "-1", tmp + "/1", "-2", tmp + "/2"]
            elif rec["flag"] == "-c": cmd += ["-c", "-U", rec["input"].strip()]
            else:
                open(tmp + "/r", "w").write(rec["input"])
                cmd += ([rec["flag"], tmp + "/r"] if rec["flag"] == "--tab5" else ([rec["flag"], "-U", tmp + "/r"] if rec["flag"] else ["-U", tmp + "/r"]))
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
            got = [l for l in p.stdout.splitlines() if not l.startswith("@PG")]
            n += 1
            if got != rec["sam"][width]:
                bad += 1
                print("BAD", rec["name"], width, " ".join(rec["args"]), p.stderr.strip()[-200:])
print("%d runs, %d bad" % (n, bad))
