#!/usr/bin/env python3
"""Re-host of the reference's scripts/test/simple_tests.pl for the unpaired cases.

The Perl script holds ~270 inline cases (reference sequences, reads, arguments, expected hits).  This tool asks
perl to dump that table as JSON, keeps the unpaired cases in input formats this build reads, runs the *reference*
binaries (oracle/_ref) on each -- forward reads and, as the Perl harness does, reverse-complemented reads -- and
records the SAM they print.  tests/test_simple_tests.py then demands the same SAM from our binaries.
Writes tests/golden/simple_tests.json.  Needs /root/reference and oracle/_ref (run where the reference exists)."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
SRC = "/root/reference/scripts/test/simple_tests.pl"
PAIRED_KEYS = {"mate1s", "mate2s", "pairhits", "pairhits_orig", "fastq1", "fastq2", "fasta1", "fasta2", "raw1", "raw2", "qseq1", "qseq2",
               "cline_reads1", "cline_reads2", "tabbed1", "tabbed2", "paired", "mate1fw", "mate2fw", "tlen_map", "pnext_map", "rnext_map"}
SKIP_KEYS = {"cont_fasta_reads", "should_abort"}


def dump_cases():
    src = open(SRC).read()
    a = src.index("my @cases = (")
    b = src.index("\n);\n", a)
    pl = ("use strict; use warnings; use JSON::PP; my $should_test_bam=0; my $compiled_with_sra=0;\n" + src[a:b + 3] +
          "\nprint JSON::PP->new->canonical->encode(\\@cases);\n")
    with tempfile.NamedTemporaryFile("w", suffix=".pl", delete=False) as f:
        f.write(pl)
    out = subprocess.run(["perl", f.name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True)
    os.unlink(f.name)
    return json.loads(out.stdout)


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTacgtNn", "TGCAtgcaNn"))


def case_inputs(c, fw):
    """-> (format flag, file text or command-line string) or None if the case is out of scope"""
    if c.get("reads") is not None:
        reads, quals, names = c["reads"], c.get("quals") or [], c.get("names") or []
        recs = []
        for i, s in enumerate(reads):
            q = quals[i] if i < len(quals) and quals[i] else "I" * len(s)
            if not fw:
                s, q = revcomp(s), q[::-1]
            nm = names[i] if i < len(names) and names[i] else "r%d" % i
            recs.append("@%s\n%s\n+\n%s\n" % (nm, s, q))
        return "-q", "".join(recs)
    if not fw:
        return None                     # file-based cases run forward only (simple_tests.pl: `next unless $fw`)
    for key, flag in (("fastq", "-q"), ("fasta", "-f"), ("raw", "-r"), ("tabbed", "--tab5"), ("cline_reads", "-c"), ("qseq", "--qseq")):
        if c.get(key) is not None:
            if key == "tabbed" and any(len(l.split("\t")) > 3 for l in c[key].splitlines() if l.strip()):
                return None             # paired tab5 records
            return flag, c[key]
    return None


def run_ref(exe, large, fa_text, flag, payload, args, tmp):
    fa = os.path.join(tmp, "ref.fa")
    open(fa, "w").write(fa_text)
    base = os.path.join(tmp, "idx")
    subprocess.check_call([os.path.join(REF, "bowtie2-build-l" if large else "bowtie2-build-s"), "--quiet", fa, base], stdout=subprocess.DEVNULL)
    cmd = [os.path.join(REF, exe)] + args + ["-x", base]
    if flag == "-c":
        cmd += ["-c", "-U", payload.strip()]
    else:
        rf = os.path.join(tmp, "reads.txt")
        open(rf, "w").write(payload)
        cmd += ([flag, rf] if flag == "--tab5" else [flag, "-U", rf])
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    if p.returncode != 0:
        return None
    return [l for l in p.stdout.splitlines() if not l.startswith("@PG")]


def main():
    cases = dump_cases()
    out = []
    skipped = 0
    for ci, c in enumerate(cases):
        if PAIRED_KEYS & set(c) or SKIP_KEYS & set(c):
            skipped += 1
            continue
        args = (c.get("args") or "").split() + ["--quiet"] + (c["report"].split() if c.get("report") else ["-a"])
        fa_text = "".join(">%d\n%s\n" % (i, s) for i, s in enumerate(c["ref"]))
        for fw in ([True] if c.get("norc") else []) + ([False] if not c.get("nofw") else []) if (c.get("norc") or c.get("nofw")) else (True, False):
            inp = case_inputs(c, fw)
            if inp is None:
                continue
            flag, payload = inp
            rec = {"case": ci, "name": c.get("name", "case%d" % ci), "fw": fw, "ref": c["ref"], "flag": flag, "input": payload, "args": args, "sam": {}}
            ok = True
            for large in (False, True):
                with tempfile.TemporaryDirectory() as tmp:
                    sam = run_ref("bowtie2-align-l" if large else "bowtie2-align-s", large, fa_text, flag, payload, args, tmp)
                if sam is None:
                    ok = False
                    break
                rec["sam"]["l" if large else "s"] = sam
            if ok:
                out.append(rec)
    dst = os.path.join(ROOT, "tests", "golden", "simple_tests.json")
    json.dump(out, open(dst, "w"), indent=0, sort_keys=True)
    print("%d sub-cases from %d cases written to %s (%d cases skipped as paired/out of scope)" % (len(out), len(cases), dst, skipped))


if __name__ == "__main__":
    sys.exit(main())
