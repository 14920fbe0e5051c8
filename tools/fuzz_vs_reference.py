#!/usr/bin/env python3
"""Differential fuzzing of the worker source against the reference binaries (oracle/_ref): random genomes (repeats, N runs,
low-complexity stretches), random unpaired or paired reads (substitutions, indels, junk and mis-oriented mates), a random
subset of options, either index width.  Runs the host-compiled worker (tests/hostsim -- build it first, e.g. by running
pytest tests/test_hostsim_golden.py) and the reference on the same input and logs every case whose SAM differs and that
our side did not flag as over a capacity limit.  usage: fuzz_vs_reference.py <seed> <iterations>   (scratch under /tmp/fuzz)
BT2G_HOSTSIM=<exe> picks the engine: tests/hostsim/hostsim (the worker under a plain main) or tests/hostsim/hostsim_driver_twin (the product's
driver, bt2g_search.cpp, on the CPU -- built by tests/test_driver_twin.py; wrap it in a script to add e.g. --batch 16 -p 3)."""
import sys, os, subprocess, random, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bt2test import ref_bin, write_fasta, write_fastq, build_index, revcomp, bam_record, write_bam
HS = os.environ.get('BT2G_HOSTSIM', os.path.join(ROOT, 'tests', 'hostsim', 'hostsim'))
os.makedirs('/tmp/fuzz', exist_ok=True)
seed0=int(sys.argv[1]); nit=int(sys.argv[2])
MODE=int(os.environ.get('FUZZ_MODE','1'))     # 2: small repeat-dense genomes, short reads, more options per case
LONG=os.environ.get('FUZZ_LONG')              # half of the unpaired reads 513 ... 1 999 bp long
LOCAL=os.environ.get('FUZZ_LOCAL')            # every case in local mode, reads up to 500 bp (with BT2G_CHECK_LOCAL_PK=1: the packed local fill replayed on every window)
out=open('/tmp/fuzz/fail_%d.log'%seed0,'w')
flog=open('/tmp/fuzz/flagged_%d.log'%seed0,'w')
def rnd_genome(rnd):
    nref=rnd.randrange(1,4)
    refs=[]
    elem="".join(rnd.choice("ACGT") for _ in range(rnd.randrange(60,400) if MODE==1 else rnd.randrange(20,150)))
    for i in range(nref):
        L=rnd.randrange(300,20000) if MODE==1 else rnd.randrange(200,3000)
        if LONG: L=rnd.randrange(2500,20000)
        s=[rnd.choice("ACGT") for _ in range(L)]
        for _ in range(rnd.randrange(0,6) if MODE==1 else rnd.randrange(3,12)):
            p=rnd.randrange(0,max(1,L-len(elem)-1))
            s[p:p+len(elem)]=[c if rnd.random()>0.03 else rnd.choice("ACGT") for c in elem]
        if rnd.random()<0.3:
            p=rnd.randrange(0,L-50); s[p:p+rnd.randrange(5,40)]=list("N"*rnd.randrange(5,40))
        if rnd.random()<0.3:
            p=rnd.randrange(0,L-100); s[p:p+80]=list(("AC" if rnd.random()<0.5 else "A")*80)[:80]
        refs.append(("c%d"%i,"".join(s)))
    return refs
def mut(rnd,s,sub,indel):
    o=[]
    for c in s:
        r=rnd.random()
        if r<sub: o.append(rnd.choice("ACGTN" if rnd.random()<0.05 else "ACGT"))
        elif r<sub+indel: continue
        elif r<sub+2*indel: o.append(c); o.append(rnd.choice("ACGT"))
        else: o.append(c)
    return "".join(o) or "A"
POOL_SE=[[],["--local"],["-k","2"],["-k","7"],["-a"],["--very-fast"],["--very-sensitive"],["--fast-local"],["--very-sensitive-local"],["-N","1"],["-L","12"],["-L","28"],["-i","C,5,0"],["-i","L,2,0.1"],
 ["--ignore-quals"],["--mp","4,1"],["--np","3"],["--rdg","3,2"],["--rfg","7,4"],["--score-min","L,-3,-0.3"],["--n-ceil","L,2,0.3"],["--nofw"],["--norc"],["--no-1mm-upfront"],["--no-exact-upfront"],
 ["-D","4"],["-R","1"],["-R","3"],["--gbar","8"],["--dpad","6"],["-5","3"],["-3","4"],["--overhang"],["--seed","17"],["-M","2"],["--xeq"],["--no-unal"],
 ["-d","-a","--no-exact-upfront","--no-1mm-upfront"],["--bwa-sw-like"],["--policy","MMP=C3;NP=C2"],["--trim-to","5:40"],["--trim-to","60"],["--passthrough"],["--omit-sec-seq","-k","3"],["--ma","3","--local"],["--qc-filter"],["--phred64"],["--policy","MMP=R"],["--policy","NP=Q;RDG=4"],["-F","30,7"],["--sam-append-comment"],["--solexa-quals"],["--extends","30"],["--dp-fails","8"],["--ug-fails","5"],["--seed-boost","50"],["--tighten","1"],["--tighten","2"],["--no-extend"],["--no-ungapped"]]
POOL_PE=[["--ff"],["--rf"],["--no-mixed"],["--no-discordant"],["--dovetail"],["--no-contain"],["--no-overlap"],["-I","80"],["-X","300"],["-X","700"],["--soft-clipped-unmapped-tlen"]]
def conflicts(a):
    flat=" ".join(" ".join(x) for x in a)
    if "--local" in flat or "-local" in flat:
        if "--score-min" in flat: return True
    if flat.count("-k ")+flat.count("-a")+flat.count("-M ")>1: return True
    if ("--very" in flat)+("--fast-local" in flat)>1: return True
    if "--trim-to" in flat and ("-5 " in flat+" " or "-3 " in flat+" "): return True
    if flat.count("--trim-to")>1: return True
    if "--bwa-sw-like" in flat and ("--ma" in flat or "--mp" in flat or "--rdg" in flat or "--rfg" in flat or "--score-min" in flat or "--policy" in flat): return True
    if "-d " in flat+" " and ("--no-1mm-upfront" not in flat): return True
    if "--phred64" in flat: return True      # the generated qualities are phred33
    if "-F " in flat+" ": return True         # needs FASTA input (covered by the regression table)
    if flat.count("--policy")>1: return True
    if "--soft-clipped-unmapped-tlen" in flat and not ("--local" in flat or "-local" in flat or "--bwa-sw-like" in flat): return True
    return False
nfail=0; nwarn=0; t0=time.time()
for it in range(nit):
    rnd=random.Random(seed0*100003+it)
    d='/tmp/fuzz/w%d'%seed0; os.makedirs(d,exist_ok=True)
    refs=rnd_genome(rnd)
    large=rnd.random()<0.3
    fa=d+"/g.fa"; base=d+"/g"
    for f in os.listdir(d):
        if f.startswith("g."): os.unlink(d+"/"+f)
    write_fasta(fa,refs); build_index(fa,base,large)
    paired=rnd.random()<0.5 and not LONG
    mixed_run=False
    n=rnd.randrange(20,120)
    sub=rnd.choice([0.0,0.01,0.03,0.08]); indel=rnd.choice([0.0,0.002,0.01])
    opts=[]
    for _ in range(rnd.randrange(0,4) if MODE==1 else rnd.randrange(1,7)):
        o=rnd.choice(POOL_SE+(POOL_PE if paired else []))
        if o not in opts: opts.append(o)
    if LOCAL and not any('local' in x for o in opts for x in o) and not any(x in ('--bwa-sw-like','--score-min') for o in opts for x in o): opts.append(['--local'])
    if conflicts(opts): continue
    args=[x for o in opts for x in o]
    exe=ref_bin("bowtie2-align-l" if large else "bowtie2-align-s")
    if paired:
        r1=[];r2=[]
        for i in range(n):
            _,s=refs[rnd.randrange(len(refs))]
            L1=rnd.randrange(20,200) if MODE==1 else rnd.randrange(12,90); L2=rnd.randrange(20,200) if MODE==1 else rnd.randrange(12,90)
            frag=max(int(rnd.gauss(250,60) if MODE==1 else rnd.gauss(120,40)),max(L1,L2)+1); frag=min(frag,len(s)-1)
            if frag<max(L1,L2)+1: L1=L2=max(5,frag-1)
            p=rnd.randrange(0,len(s)-frag) if len(s)>frag else 0
            f=s[p:p+frag]; m1=f[:L1]; m2=revcomp(f[-L2:])
            k=rnd.random()
            if k>0.93: m2=revcomp(m2)
            elif k>0.86: m2="".join(rnd.choice("ACGT") for _ in range(L2))
            if rnd.random()<0.5: m1,m2=m2,m1
            m1=mut(rnd,m1,sub,indel); m2=mut(rnd,m2,sub,indel)
            r1.append(("p%d/1"%i,m1,"".join(rnd.choice("IIIH?5#") for _ in m1))); r2.append(("p%d/2"%i,m2,"".join(rnd.choice("IIIH?5#") for _ in m2)))
        if "--sam-append-comment" in args:
            cm=lambda m: rnd.choice(["", " %d:N:0:ACGT"%m, " %d:Y:18:TTAGGC x"%m, " free text", " 3:N:0:A"])
            r1=[(n+cm(1),s_,q) for n,s_,q in r1]; r2=[(n+cm(2),s_,q) for n,s_,q in r2]
        if rnd.random()<0.15 and "--sam-append-comment" not in args:
            # the same pairs as an unaligned BAM file (mates flagged 0x40 / 0x80, records to be skipped in between)
            recs=[]
            for (n1,s1,q1),(n2,s2,q2) in zip(r1,r2):
                recs.append(bam_record(n1,77,s1,q1,b"BCZAC\0"))
                if rnd.random()<0.1: recs.append(bam_record("solo",4,s1,q1))
                recs.append(bam_record(n2,141,s2,q2))
            write_bam(d+"/p.bam",recs,block=rnd.choice([300,4000,60000]))
            inp=["-b","--align-paired-reads"]+(["--preserve-tags"] if rnd.random()<0.5 else [])+["-1",d+"/p.bam","-2",d+"/p.bam"]
        else:
            write_fastq(d+"/1.fq",r1); write_fastq(d+"/2.fq",r2)
            inp=["-1",d+"/1.fq","-2",d+"/2.fq"]
            if rnd.random()<0.15:
                # pairs and unpaired reads in one run (the reference only finishes on such input single-threaded)
                us=[("u%d"%i,s_,q) for i,(_,s_,q) in enumerate(r1[:rnd.randrange(1,40)])]
                write_fastq(d+"/u.fq",us); inp+=["-U",d+"/u.fq"]; mixed_run=True
    else:
        rs=[]
        for i in range(n):
            _,s=refs[rnd.randrange(len(refs))]
            L=rnd.randrange(1,260) if MODE==1 else rnd.randrange(1,80)
            if LOCAL and rnd.random()<0.5: L=rnd.randrange(200,500)
            if LONG and rnd.random()<0.5: L=rnd.randrange(513,1900)      # the long-read class (reads of 513 ... 1 999 bp)
            L=min(L,len(s)-1)
            p=rnd.randrange(0,len(s)-L); m=s[p:p+L]
            if rnd.random()<0.5: m=revcomp(m)
            if rnd.random()<0.05: m="".join(rnd.choice("ACGT") for _ in range(L))
            m=mut(rnd,m,sub,indel)
            rs.append(("r%d"%i,m,"".join(rnd.choice("IIIH?5#") for _ in m)))
        if "--sam-append-comment" in args:
            rs=[(n+rnd.choice(["", " 1:N:0:ACGT", " 2:Y:18:TTAGGC x", " free text", " nocolon"]),s_,q) for n,s_,q in rs]
        if rnd.random()<0.15 and "--sam-append-comment" not in args:
            recs=[bam_record(n_,rnd.choice([4,4,4,4,0,69]),s_,q,rnd.choice([b"",b"NHC\x03",b"RGZg1\0XSs\xf9\xff"])) for n_,s_,q in rs]
            write_bam(d+"/r.bam",recs,block=rnd.choice([300,4000,60000]))
            inp=["-b"]+(["--preserve-tags"] if rnd.random()<0.5 else [])+["-U",d+"/r.bam"]
        else:
            write_fastq(d+"/r.fq",rs); inp=["-U",d+"/r.fq"]
    a=subprocess.run([exe]+args+["-x",base]+inp+(["-p","1"] if mixed_run else ["-p","2","--reorder"]),capture_output=True,text=True,errors="replace")
    b=subprocess.run([HS]+args+["-x",base]+inp,capture_output=True,text=True,errors="replace")
    body=lambda t:[l for l in t.splitlines() if not l.startswith("@PG")]
    if a.returncode!=0: continue
    warn="Warning: " in b.stderr and ("overflow" in b.stderr or "exceeded" in b.stderr)
    if warn:      # which capacity: the sites of the flagged reads go to the log, the case is not compared
        nwarn+=1
        sites=sorted(set(l.split("site")[-1].strip(" )") for l in b.stderr.splitlines() if l.startswith("Warning: read") and "site" in l))
        flog.write(json.dumps({"it":it,"args":args,"paired":paired,"sites":sites,"long":bool(LONG)})+"\n"); flog.flush()
    if body(a.stdout)!=body(b.stdout) and not warn:
        nfail+=1
        keep='/tmp/fuzz/case_%d_%d'%(seed0,it); os.system("rm -rf %s; cp -r %s %s"%(keep,d,keep))
        out.write(json.dumps({"it":it,"args":args,"paired":paired,"large":large,"rc":b.returncode,"err":b.stderr[-300:]})+"\n"); out.flush()
print("seed",seed0,"iters",nit,"fails",nfail,"flagged (not compared)",nwarn,"%.0fs"%(time.time()-t0))
