#!/usr/bin/env python3
"""Re-host of the reference's scripts/test/simple_tests.pl for the unpaired cases.

The Perl script holds ~270 inline cases (reference sequences, reads, arguments, expected hits).  This tool asks
perl to dump that table as JSON, keeps the unpaired cases in input formats this build reads, runs the *reference*
binaries (oracle/_ref) on each -- forward reads and, as the Perl harness does, reverse-complemented reads -- and
records the SAM they print.  tests/test_simple_tests.py then demands the same SAM from our binaries.
Writes tests/golden/simple_tests.json.  Needs /root/reference and oracle/_ref (run where the reference exists)."""
import json
import os
import shlex
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
SRC = "/root/reference/scripts/test/simple_tests.pl"
PAIRED_KEYS = {"mate1s", "mate2s", "pairhits", "pairhits_orig", "fastq1", "fastq2", "fasta1", "fasta2", "raw1", "raw2", "qseq1", "qseq2",
               "cline_reads1", "cline_reads2", "tabbed1", "tabbed2", "paired", "mate1fw", "mate2fw", "tlen_map", "pnext_map", "rnext_map"}
SKIP_KEYS = set()


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
    if c.get("cont_fasta_reads") is not None:
        return "", c["cont_fasta_reads"]       # the case's own arguments carry -F <len>,<freq>
    for key, flag in (("fastq", "-q"), ("fasta", "-f"), ("raw", "-r"), ("tabbed", "--tab5"), ("cline_reads", "-c"), ("qseq", "--qseq")):
        if c.get(key) is not None:
            return flag, c[key]        # --tab5 files may hold paired (5-field) records
    return None


def pair_inputs(c, fw):
    """-> (format flag, mate-1 text, mate-2 text, extra args) for a paired case, or None if out of scope"""
    m1fw = c.get("mate1fw", 1)
    m2fw = c.get("mate2fw", 0)
    extra = ["--" + ("f" if m1fw else "r") + ("f" if m2fw else "r")]
    if c.get("mate1s") is not None:
        m1, m2 = list(c["mate1s"]), list(c["mate2s"])
        q1, q2 = list(c.get("qual1s") or []), list(c.get("qual2s") or [])
        names = c.get("names") or []
        if not fw:      # simple_tests.pl:4895-4921: mates swap roles; same-strand policies also reverse-complement them
            if bool(m1fw) == bool(m2fw):
                m1, m2 = [revcomp(x) for x in m1], [revcomp(x) for x in m2]
                q1, q2 = [x[::-1] for x in q1], [x[::-1] for x in q2]
            m1, m2, q1, q2 = m2, m1, q2, q1
        if any(x == "" for x in m1 + m2):
            return None
        f1 = f2 = ""
        for i in range(len(m1)):
            nm = names[i] if i < len(names) and names[i] else "r%d" % i
            a = q1[i] if i < len(q1) and q1[i] else "I" * len(m1[i])
            b = q2[i] if i < len(q2) and q2[i] else "I" * len(m2[i])
            f1 += "@%s/1\n%s\n+\n%s\n" % (nm, m1[i], a)
            f2 += "@%s/2\n%s\n+\n%s\n" % (nm, m2[i], b)
        return "-q", f1, f2, extra
    if not fw:
        return None
    for key, flag in (("fastq", "-q"), ("fasta", "-f"), ("raw", "-r"), ("cline_reads", "-c"), ("qseq", "--qseq")):
        if c.get(key + "1") is not None and c.get(key + "2") is not None:
            return flag, c[key + "1"], c[key + "2"], extra
    return None


def run_ref_pair(exe, large, fa_text, flag, p1, p2, args, tmp):
    fa = os.path.join(tmp, "ref.fa")
    open(fa, "w").write(fa_text)
    base = os.path.join(tmp, "idx")
    subprocess.check_call([os.path.join(REF, "bowtie2-build-l" if large else "bowtie2-build-s"), "--quiet", fa, base], stdout=subprocess.DEVNULL)
    cmd = [os.path.join(REF, exe)] + args + ["-x", base]
    if flag == "-c":
        cmd += ["-c", "-1", p1.strip(), "-2", p2.strip()]
    else:
        f1, f2 = os.path.join(tmp, "m1.txt"), os.path.join(tmp, "m2.txt")
        open(f1, "w").write(p1)
        open(f2, "w").write(p2)
        cmd += [flag, "-1", f1, "-2", f2]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    if p.returncode != 0:
        return None
    return [l for l in p.stdout.splitlines() if not l.startswith("@PG")]


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
        cmd += ([flag, rf] if flag == "--tab5" else ([flag, "-U", rf] if flag else ["-U", rf]))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    if p.returncode != 0:
        return None
    return [l for l in p.stdout.splitlines() if not l.startswith("@PG")]


def main():
    cases = dump_cases()
    out = []
    skipped = 0
    for ci, c in enumerate(cases):
        if SKIP_KEYS & set(c):
            skipped += 1
            continue
        # the Perl harness hands the argument string to sh: quotes group, and inside double quotes a backslash before ";" stays
        args = shlex.split(c.get("args") or "") + ["--quiet"] + (shlex.split(c["report"]) if c.get("report") else ["-a"])
        fa_text = "".join(">%d\n%s\n" % (i, s) for i, s in enumerate(c["ref"]))
        # a paired case given as one --tab5 file goes through the single-file path below (pair policy arguments appended)
        tab_only = c.get("tabbed") is not None and not any(c.get(k) is not None for k in ("mate1s", "fastq1", "fasta1", "raw1", "cline_reads1", "reads", "fastq", "fasta", "raw", "cline_reads"))
        if tab_only and (PAIRED_KEYS & set(c)):
            args = args + ["--" + ("f" if c.get("mate1fw", 1) else "r") + ("f" if c.get("mate2fw", 0) else "r")]
        if (PAIRED_KEYS & set(c)) and not tab_only:
            # paired cases: mate lists (forward and role-swapped) or mate files (forward only)
            if any(c.get(k) is not None for k in ("tabbed1", "tabbed2", "tabbed", "reads", "fastq", "fasta", "raw", "cline_reads")):
                skipped += 1
                continue
            n_here = 0
            for fw in (True, False):
                inp = pair_inputs(c, fw)
                if inp is None:
                    continue
                flag, p1, p2, extra = inp
                rec = {"case": ci, "name": c.get("name", "case%d" % ci), "fw": fw, "ref": c["ref"], "flag": flag, "m1": p1, "m2": p2,
                       "args": args + extra, "sam": {}}
                ok = True
                for large in (False, True):
                    with tempfile.TemporaryDirectory() as tmp:
                        sam = run_ref_pair("bowtie2-align-l" if large else "bowtie2-align-s", large, fa_text, flag, p1, p2, args + extra, tmp)
                    if sam is None:
                        ok = False
                        break
                    rec["sam"]["l" if large else "s"] = sam
                if ok:
                    out.append(rec)
                    n_here += 1
            if n_here == 0:
                skipped += 1
            continue
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
            if ok and not c.get("should_abort"):
                out.append(rec)
            elif not ok and c.get("should_abort"):
                rec["sam"] = {}
                rec["abort"] = True         # the reference refuses / aborts on this input: so must we
                out.append(rec)
    dst = os.path.join(ROOT, "tests", "golden", "simple_tests.json")
    json.dump(out, open(dst, "w"), indent=0, sort_keys=True)
    print("%d sub-cases from %d cases written to %s (%d cases skipped as paired/out of scope)" % (len(out), len(cases), dst, skipped))


if __name__ == "__main__":
    sys.exit(main())
